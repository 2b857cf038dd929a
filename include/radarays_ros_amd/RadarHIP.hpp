// RadarHIP.hpp -- C++ host side above the C ABI, mirroring the reference's backend interface.
//
// Reference (C++):  include/radarays_ros/Radar.hpp:34-105   abstract `Radar` (the seam:
//                   `virtual sensor_msgs::ImagePtr simulate(ros::Time) = 0`, :64)
//                   src/radarays_ros/Radar.cpp:10-41,188-226 ctor constants, updateDynCfg, loadParams
//                   include/radarays_ros/RadarCPU.hpp:21-28  a concrete backend
// This header keeps the same member names, argument meaning and error behaviour, minus the
// ROS / OpenCV / rmagine types this image does not have: TF lookup becomes updateTsm(pose),
// sensor_msgs::Image becomes the plain `Image` struct with the same fields, rm::Transform
// becomes float[7] (quaternion xyzw + translation).  INTEGRATION.md shows the ROS-typed twin.
#pragma once
#include <cmath>
#include <cstdint>
#include <iostream>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../radarays_mi355.h"

namespace radarays_ros_amd {

// msg/RadarMaterial.msg, msg/RadarModel.msg, msg/RadarParams.msg
struct RadarMaterial { float velocity = 0.3f, ambient = 1.0f, diffuse = 0.0f, specular = 1.0f; };
struct RadarModel { float beam_width = 8.0f * (float)M_PI / 180.0f; uint32_t n_samples = 200; uint32_t n_reflections = 2; };
struct RadarParams { std::vector<RadarMaterial> materials; RadarModel model; };

// cfg/RadarModel.cfg:11-85 (the fields the hot path reads; same names and defaults)
struct RadarModelConfig {
    double beam_width = 8.0, resolution = 0.0438;
    int n_cells = 3424, n_samples = 10, beam_sample_dist = 2;
    double beam_sample_dist_normal_p_in_cone = 0.8;
    int n_reflections = 4;
    double energy_max = 0.5, signal_max = 120.0;
    int signal_denoising = 1;
    int signal_denoising_triangular_width = 50; double signal_denoising_triangular_mode = 0.35;
    int signal_denoising_gaussian_width = 50;   double signal_denoising_gaussian_mode = 0.5;
    int signal_denoising_mb_width = 50;         double signal_denoising_mb_mode = 0.4;
    int ambient_noise = 2;
    double ambient_noise_at_signal_0 = 0.3, ambient_noise_at_signal_1 = 0.03;
    double ambient_noise_energy_max = 0.5, ambient_noise_energy_min = 0.1, ambient_noise_energy_loss = 0.05;
    int scroll_image = 0;
    double multipath_threshold = 0.5;
    bool record_multi_reflection = true, record_multi_path = false, include_motion = true;
};

// sensor_msgs/Image as RadarCPU.cpp:555-561 fills it
struct Image {
    double stamp = 0.0; std::string frame_id;
    uint32_t height = 0, width = 0, step = 0; std::string encoding = "mono8";
    std::vector<uint8_t> data;
};
using ImagePtr = std::shared_ptr<Image>;

class Radar {   // Radar.hpp:34
public:
    Radar(std::string map_frame, std::string sensor_frame)
    : m_map_frame(std::move(map_frame)), m_sensor_frame(std::move(sensor_frame)) {}   // Radar.cpp:10-41
    virtual ~Radar() = default;

    void loadParams(const std::vector<RadarMaterial>& materials, const std::vector<int>& object_materials,
                    int material_id_air)   // Radar.cpp:220-226 (ROS parameter server -> arguments)
    { m_params.materials = materials; m_object_materials = object_materials; m_material_id_air = material_id_air; m_dirty_mat = true; }
    RadarParams getParams() const { return m_params; }
    void setParams(const RadarParams& p) { m_params = p; m_dirty_mat = true; m_dirty_cfg = true; }

    void updateDynCfg(const RadarModelConfig& config)   // Radar.cpp:188-218
    {
        if (config.beam_sample_dist != m_cfg.beam_sample_dist || std::abs(config.beam_width - m_cfg.beam_width) > 0.001 ||
            config.n_samples != m_cfg.n_samples ||
            std::abs(config.beam_sample_dist_normal_p_in_cone - m_cfg.beam_sample_dist_normal_p_in_cone) > 0.001)
            m_resample = true;
        m_params.model.beam_width = (float)(config.beam_width * M_PI / 180.0);
        m_params.model.n_samples = (uint32_t)config.n_samples;
        m_params.model.n_reflections = (uint32_t)config.n_reflections;
        m_cfg = config; m_dirty_cfg = true;
    }
    bool updateTsm() const { return has_last; }                 // Radar.cpp:80-132: TF lookup, last pose as fallback
    bool updateTsm(const float pose_qxyzw_t[7])
    {
        for (int k = 0; k < 7; k++) if (!std::isfinite(pose_qxyzw_t[k])) return has_last;
        for (int k = 0; k < 7; k++) Tsm_last[k] = pose_qxyzw_t[k];
        has_last = true; return true;
    }
    virtual ImagePtr simulate(double stamp) = 0;                 // Radar.hpp:64

protected:
    float Tsm_last[7] = { 0, 0, 0, 1, 0, 0, 0 }; bool has_last = false;
    std::string m_map_frame, m_sensor_frame;
    RadarParams m_params;
    RadarModelConfig m_cfg;
    int m_material_id_air = 0;                 // Radar.cpp:23
    std::vector<int> m_object_materials;
    float m_wave_energy_threshold = 0.001f;    // Radar.cpp:24
    std::vector<float> m_waves_start;          // beam sample directions [n][3]
    bool m_resample = true;                    // Radar.cpp:25
    bool m_dirty_cfg = true, m_dirty_mat = true;
};

class RadarHIP : public Radar {   // sibling of RadarCPU (RadarCPU.hpp:16-37)
public:
    // One backend object per process like the reference (radar_simulator.cpp:145-176); `devices` lists the GPUs
    // of this node it fans out over (rr_multi: azimuth blocks, one RCCL collective per frame).
    RadarHIP(std::string map_frame, std::string sensor_frame, const std::vector<float>& verts,
             const std::vector<uint32_t>& faces, const std::vector<uint32_t>& face_object, const std::vector<int>& devices,
             bool build_on_gpu = false /* rr_set_mesh_gpu: the map loads in 0.35 s instead of 1.9 s at 10M triangles, frames take 1.2x as long */)
    : Radar(std::move(map_frame), std::move(sensor_frame))
    {
        m_multi = rr_create_multi(devices.data(), (int)devices.size());
        if (!m_multi) throw std::runtime_error(rr_multi_last_error(nullptr));
        m_ctx = rr_multi_ctx(m_multi, 0);
        if ((build_on_gpu ? rr_multi_set_mesh_gpu : rr_multi_set_mesh)(m_multi, verts.data(), verts.size() / 3, faces.data(), faces.size() / 3,
                                                                        face_object.empty() ? nullptr : face_object.data())) {
            std::string e = rr_multi_last_error(m_multi); rr_destroy_multi(m_multi); throw std::runtime_error(e);
        }
    }
    RadarHIP(std::string map_frame, std::string sensor_frame, const std::vector<float>& verts,
             const std::vector<uint32_t>& faces, const std::vector<uint32_t>& face_object, int device = 0)
    : RadarHIP(std::move(map_frame), std::move(sensor_frame), verts, faces, face_object, std::vector<int>{ device }) {}
    ~RadarHIP() override { rr_destroy_multi(m_multi); }
    RadarHIP(const RadarHIP&) = delete;
    RadarHIP& operator=(const RadarHIP&) = delete;

    // m_waves_start: RadarCPU::simulate draws them with sample_cone_local whenever m_resample is set (RadarCPU.cpp:136-145),
    // seeding from std::random_device.  push() does the same (rr_sample_cone_local); setBeamSeed() makes the draw
    // reproducible, setBeamSamples() injects directions from any other generator -- [n][3], local frame -- until the next
    // dynamic reconfigure of the beam asks for a re-draw again
    void setBeamSeed(uint32_t seed) { m_beam_seed = seed; m_have_seed = true; }
    void setBeamSamples(const std::vector<float>& dirs) { m_waves_start = dirs; m_resample = false; m_push_beams = true; }
    const std::vector<float>& beamSamples() const { return m_waves_start; }
    void setNoiseOffsets(const std::vector<float>& rnd) { rr_multi_set_noise_offsets(m_multi, rnd.data(), rnd.size()); }
    // include_motion (cfg/RadarModel.cfg:85, RadarCPU.cpp:190-196): the reference looks Tsm up once PER AZIMUTH while
    // the antenna turns; the TF lookups of one sweep arrive here as [n_angles][7] and are used while
    // m_cfg.include_motion is set (an empty vector: one pose per frame again)
    void setMotionPoses(const std::vector<float>& poses_per_azimuth) { m_motion = poses_per_azimuth; m_push_motion = true; }

    ImagePtr simulate(double stamp) override   // RadarCPU.cpp:30-564
    {
        ImagePtr msg;
        if (!updateTsm()) {
            std::cout << "Couldn't get Transform between sensor and map. Skipping..." << std::endl;   // RadarCPU.cpp:131
            return msg;
        }
        if (!push()) return msg;
        msg = std::make_shared<Image>();
        msg->height = (uint32_t)m_cfg.n_cells; msg->width = (uint32_t)m_n_angles; msg->step = msg->width;
        msg->data.assign((size_t)msg->height * msg->width, 0);
        if (rr_multi_device_count(m_multi) > 1) {
            if (rr_multi_simulate(m_multi, Tsm_last, msg->data.data())) { msg.reset(); m_err = rr_multi_last_error(m_multi); std::cout << "[RadarHIP] " << m_err << std::endl; return {}; }
        } else if (rr_simulate(m_ctx, Tsm_last, 0, m_n_angles, msg->data.data(), nullptr, &m_stats)) { msg.reset(); return fail(); }
        msg->stamp = stamp; msg->frame_id = m_sensor_frame;   // RadarCPU.cpp:560-561
        return msg;
    }
    // Offline generation (the twin of integration/src/radarays_ros/RadarHIP.cpp: simulateBatch / simulateSweeps): one image
    // per pose [n][7], up to RR_MAX_BATCH poses per set of launches; with per-azimuth pose tables (include_motion,
    // RadarCPU.cpp:190-196) sweeps = [n][n_angles][7], one table per frame (rr_multi_set_motion_poses)
    std::vector<ImagePtr> simulateBatch(const std::vector<float>& poses, double stamp) { return batch(poses, false, stamp); }
    std::vector<ImagePtr> simulateSweeps(const std::vector<float>& sweeps, double stamp) { return batch(sweeps, true, stamp); }

    // The gen_radar_image action of the optimisation loop (action/GenRadarImage.action,
    // scripts/radaray_opti.py:170-200), batched: one image per material table, same pose, one call.
    std::vector<ImagePtr> simulateMaterialSets(const std::vector<std::vector<RadarMaterial>>& sets, double stamp)
    {
        std::vector<ImagePtr> out;
        if (!updateTsm()) {
            std::cout << "Couldn't get Transform between sensor and map. Skipping..." << std::endl;
            return out;
        }
        if (!push()) return out;
        const size_t n_mat = m_params.materials.size();
        std::vector<rr_material> flat;
        for (const auto& set : sets) {
            if (set.size() != n_mat) { m_err = "every material set needs as many entries as loadParams() gave"; return out; }
            for (const RadarMaterial& m : set) flat.push_back({ m.velocity, m.ambient, m.diffuse, m.specular });
        }
        const size_t npx = (size_t)m_cfg.n_cells * m_n_angles;
        std::vector<uint8_t> px(sets.size() * npx);
        if (rr_simulate_material_sets(m_ctx, Tsm_last, flat.data(), (int)sets.size(), n_mat, px.data())) { fail(); return out; }
        for (size_t k = 0; k < sets.size(); k++) {
            ImagePtr msg = std::make_shared<Image>();
            msg->height = (uint32_t)m_cfg.n_cells; msg->width = (uint32_t)m_n_angles; msg->step = msg->width;
            msg->data.assign(px.begin() + k * npx, px.begin() + (k + 1) * npx);
            msg->stamp = stamp; msg->frame_id = m_sensor_frame;
            out.push_back(msg);
        }
        return out;
    }
    // The same action with the optimiser's WHOLE parameter vector (scripts/radaray_opti.py:36-113: model.beam_width,
    // model.n_reflections, the material values): one RadarParams per evaluation, one call for all of them
    // (rr_simulate_param_sets: sets with the same beam_width share pass 0, sets with fewer passes stop early).  The beam of
    // a set is drawn like push() draws it -- with the very seed push() used for the current beam, so equal widths give equal
    // directions and sets that differ only in beam_width share their variates;
    // model.n_samples must be the current one.  `real` given: the objective values of radaray_opti.py:196 come back
    // (PSNR against the real image, skimage's formula; the optimiser minimises its negative) and, with want_images false,
    // no image leaves the GPU.
    bool simulateParamSets(const std::vector<RadarParams>& sets, double stamp, std::vector<ImagePtr>* images,
                           const Image* real = nullptr, std::vector<double>* psnr = nullptr)
    {
        if (!updateTsm()) { std::cout << "Couldn't get Transform between sensor and map. Skipping..." << std::endl; return false; }
        if (!push()) return false;
        const size_t n_mat = m_params.materials.size(), nb = m_params.model.n_samples;
        const size_t npx = (size_t)m_cfg.n_cells * m_n_angles;
        if (real && (real->data.size() != npx || !psnr)) { m_err = "the real image must be n_cells x n_angles mono8 (and psnr given)"; return false; }
        std::vector<rr_material> flat; flat.reserve(sets.size() * n_mat);
        std::vector<std::vector<float>> dirs(sets.size());
        std::vector<rr_param_set> ps(sets.size());
        const uint32_t seed = m_beam_seed;     // the seed push() drew the CURRENT beam with: sets that differ only in beam_width see the same variates
        for (size_t k = 0; k < sets.size(); k++) {
            const RadarParams& p = sets[k];
            if (p.materials.size() != n_mat || p.model.n_samples != nb) { m_err = "every parameter set needs the loaded number of materials and the current n_samples"; return false; }
            for (const RadarMaterial& m : p.materials) flat.push_back({ m.velocity, m.ambient, m.diffuse, m.specular });
            if (std::abs(p.model.beam_width - m_params.model.beam_width) > 1e-7f) {
                dirs[k].assign(3 * nb, 0.0f);
                if (rr_sample_cone_local(seed, p.model.beam_width, nb, m_cfg.beam_sample_dist, (float)m_cfg.beam_sample_dist_normal_p_in_cone, dirs[k].data())) { m_err = "sample_cone_local failed"; return false; }
            }
            ps[k].n_reflections = (int32_t)p.model.n_reflections; ps[k].reserved_ = 0;
        }
        for (size_t k = 0; k < sets.size(); k++) { ps[k].materials = flat.data() + k * n_mat; ps[k].beam_dirs = dirs[k].empty() ? nullptr : dirs[k].data(); }
        std::vector<uint8_t> px(images ? sets.size() * npx : 0);
        if (psnr) psnr->assign(sets.size(), 0.0);
        if (rr_simulate_param_sets(m_ctx, Tsm_last, ps.data(), (int)sets.size(), n_mat, images ? px.data() : nullptr,
                                   real ? real->data.data() : nullptr, real ? psnr->data() : nullptr)) { fail(); return false; }
        if (images) {
            images->clear();
            for (size_t k = 0; k < sets.size(); k++) {
                ImagePtr msg = std::make_shared<Image>();
                msg->height = (uint32_t)m_cfg.n_cells; msg->width = (uint32_t)m_n_angles; msg->step = msg->width;
                msg->data.assign(px.begin() + k * npx, px.begin() + (k + 1) * npx);
                msg->stamp = stamp; msg->frame_id = m_sensor_frame;
                images->push_back(msg);
            }
        }
        return true;
    }
    const std::string& lastError() const { return m_err; }
    const rr_stats& lastStats() const { return m_stats; }

private:
    std::vector<ImagePtr> batch(const std::vector<float>& poses, bool sweeps, double stamp)
    {
        std::vector<ImagePtr> out;
        if (!push()) return out;
        const size_t per = sweeps ? 7 * (size_t)m_n_angles : 7;
        if (poses.empty() || poses.size() % per) { m_err = "poses must be [n][7] (sweeps: [n][n_angles][7])"; return out; }
        const size_t n_total = poses.size() / per, npx = (size_t)m_cfg.n_cells * m_n_angles;
        std::vector<uint8_t> px((size_t)RR_MAX_BATCH * npx);
        std::vector<float> first;
        for (size_t at = 0; at < n_total; at += RR_MAX_BATCH) {
            const size_t n = std::min(n_total - at, (size_t)RR_MAX_BATCH);
            const float* p = poses.data() + at * per;
            if (sweeps) {      // table k of the call = the per-azimuth poses of its frame k; the pose arguments are ignored but must be valid
                first.clear();
                for (size_t k = 0; k < n; k++) first.insert(first.end(), p + k * per, p + k * per + 7);
                if (rr_multi_set_motion_poses(m_multi, p, n * (size_t)m_n_angles)) { mfail(); break; }
            } else if (rr_multi_set_motion_poses(m_multi, nullptr, 0)) { mfail(); break; }
            if (rr_multi_simulate_batch(m_multi, sweeps ? first.data() : p, (int)n, px.data())) { mfail(); break; }
            for (size_t k = 0; k < n; k++) {
                ImagePtr msg = std::make_shared<Image>();
                msg->height = (uint32_t)m_cfg.n_cells; msg->width = (uint32_t)m_n_angles; msg->step = msg->width;
                msg->data.assign(px.begin() + k * npx, px.begin() + (k + 1) * npx);
                msg->stamp = stamp; msg->frame_id = m_sensor_frame;
                out.push_back(msg);
            }
        }
        m_push_motion = true;      // simulate() re-installs its own table (or none)
        return out;
    }
    // marshal the protected state of Radar into the context (what simulate() reads, Radar.hpp:66-105)
    bool push()
    {
        if (m_resample || m_waves_start.empty()) {    // RadarCPU.cpp:136-145
            const size_t n = m_params.model.n_samples;
            m_waves_start.assign(3 * n, 0.0f);
            // the reference seeds every re-draw from std::random_device (radar_algorithms.cpp:258-259); so does this, unless
            // setBeamSeed fixed one -- and the seed that was USED is kept, so that a parameter batch can repeat the draw for
            // other beam widths on the same variates (advisor, round 4)
            if (!m_have_seed) m_beam_seed = (uint32_t)std::random_device{}();
            const uint32_t seed = m_beam_seed;
            if (rr_sample_cone_local(seed, m_params.model.beam_width, n, m_cfg.beam_sample_dist,
                                     (float)m_cfg.beam_sample_dist_normal_p_in_cone, m_waves_start.data())) {
                m_err = "sample_cone_local: beam_sample_dist must be 0..3"; std::cout << "[RadarHIP] " << m_err << std::endl; return false;
            }
            m_resample = false; m_push_beams = true;
        }
        if (m_dirty_cfg) {
            rr_config c; rr_default_config(&c);
            c.n_cells = m_cfg.n_cells; c.n_reflections = (int)m_params.model.n_reflections;
            c.signal_denoising = m_cfg.signal_denoising;
            c.signal_denoising_triangular_width = m_cfg.signal_denoising_triangular_width;
            c.signal_denoising_triangular_mode = m_cfg.signal_denoising_triangular_mode;
            c.signal_denoising_gaussian_width = m_cfg.signal_denoising_gaussian_width;
            c.signal_denoising_gaussian_mode = m_cfg.signal_denoising_gaussian_mode;
            c.signal_denoising_mb_width = m_cfg.signal_denoising_mb_width;
            c.signal_denoising_mb_mode = m_cfg.signal_denoising_mb_mode;
            c.ambient_noise = m_cfg.ambient_noise; c.scroll_image = m_cfg.scroll_image;
            c.record_multi_reflection = m_cfg.record_multi_reflection; c.record_multi_path = m_cfg.record_multi_path;
            c.multipath_threshold = m_cfg.multipath_threshold;
            c.resolution = m_cfg.resolution; c.energy_max = m_cfg.energy_max; c.signal_max = m_cfg.signal_max;
            c.ambient_noise_at_signal_0 = m_cfg.ambient_noise_at_signal_0;
            c.ambient_noise_at_signal_1 = m_cfg.ambient_noise_at_signal_1;
            c.ambient_noise_energy_max = m_cfg.ambient_noise_energy_max;
            c.ambient_noise_energy_min = m_cfg.ambient_noise_energy_min;
            c.ambient_noise_energy_loss = m_cfg.ambient_noise_energy_loss;
            c.wave_energy_threshold = m_wave_energy_threshold;
            c.range_max = 1000.0f;                  // make_model: range.max of the OnDn model (radar_algorithms.cpp:158)
            if (rr_multi_set_config(m_multi, &c)) { mfail(); return false; }
            m_n_angles = c.n_angles; m_dirty_cfg = false; m_push_motion = true;
        }
        if (m_push_motion) {
            const bool on = m_cfg.include_motion && m_motion.size() == 7 * (size_t)m_n_angles;
            if (rr_multi_set_motion_poses(m_multi, on ? m_motion.data() : nullptr, on ? (size_t)m_n_angles : 0)) { mfail(); return false; }
            m_push_motion = false;
        }
        if (m_dirty_mat) {
            std::vector<rr_material> mats(m_params.materials.size());
            for (size_t i = 0; i < mats.size(); i++) {
                const RadarMaterial& m = m_params.materials[i];
                mats[i] = { m.velocity, m.ambient, m.diffuse, m.specular };
            }
            std::vector<int32_t> om(m_object_materials.begin(), m_object_materials.end());
            if (rr_multi_set_materials(m_multi, mats.data(), mats.size(), om.data(), om.size(), m_material_id_air)) { mfail(); return false; }
            m_dirty_mat = false;
        }
        if (m_push_beams) {
            if (rr_multi_set_beam_samples(m_multi, m_waves_start.data(), m_waves_start.size() / 3)) { mfail(); return false; }
            m_push_beams = false;
        }
        return true;
    }
    ImagePtr fail() { m_err = rr_last_error(m_ctx); std::cout << "[RadarHIP] " << m_err << std::endl; return {}; }
    void mfail() { m_err = rr_multi_last_error(m_multi); std::cout << "[RadarHIP] " << m_err << std::endl; }
    rr_multi* m_multi = nullptr;     // owns one context per device (and the RCCL communicator when there are several)
    rr_ctx* m_ctx = nullptr;         // = the context of the first device (statistics, parameter batches)
    std::vector<float> m_motion; bool m_push_motion = false;
    int m_n_angles = 400;
    bool m_push_beams = false;
    uint32_t m_beam_seed = 0; bool m_have_seed = false;
    rr_stats m_stats{};
    std::string m_err;
};

}  // namespace radarays_ros_amd
