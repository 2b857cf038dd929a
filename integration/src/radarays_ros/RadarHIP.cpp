#include "radarays_ros/RadarHIP.hpp"

#include <cv_bridge/cv_bridge.h>


#include <cmath>
#include <cstring>
#include <random>
#include <stdexcept>

namespace rm = rmagine;

namespace radarays_ros
{

namespace
{

inline void to_pose7(const rm::Transform& T, float* p)
{
    p[0] = T.R.x; p[1] = T.R.y; p[2] = T.R.z; p[3] = T.R.w;
    p[4] = T.t.x; p[5] = T.t.y; p[6] = T.t.z;
}

} // namespace

RadarHIP::RadarHIP(
    std::shared_ptr<ros::NodeHandle> nh_p,
    std::shared_ptr<tf2_ros::Buffer> tf_buffer,
    std::shared_ptr<tf2_ros::TransformListener> tf_listener,
    std::string map_frame,
    std::string sensor_frame,
    const std::string& map_file,
    const std::vector<int>& devices,
    bool build_on_gpu)
:Base(nh_p, tf_buffer, tf_listener, map_frame, sensor_frame)
{
    rr_mesh mesh;
    char err[512];
    if(rr_load_mesh_file(map_file.c_str(), &mesh, err, sizeof(err)))
    {
        throw std::runtime_error(std::string("[RadarHIP] map '") + map_file + "': " + err);
    }
    // Object ids index the material list (m_object_materials[obj_id], RadarCPU.cpp:268).  The loader numbers objects in
    // depth-first scene order; whether rm::import_embree_map (radar_simulator.cpp:149) numbers this file the same way cannot be
    // known without rmagine.  ~hip_object_order: the object NAMES in the order the material list of the scene's yaml was
    // written for (config/oru4_test.yaml:37-56) -- the objects are renumbered to it (rr_mesh_reorder_objects)
    std::vector<std::string> object_order;
    nh_p->getParam("hip_object_order", object_order);
    if(!object_order.empty())
    {
        std::vector<const char*> names;
        for(size_t i = 0; i < object_order.size(); i++) { names.push_back(object_order[i].c_str()); }
        if(rr_mesh_reorder_objects(&mesh, names.data(), names.size(), err, sizeof(err)))
        {
            rr_free_mesh(&mesh);
            throw std::runtime_error(std::string("[RadarHIP] ~hip_object_order: ") + err);
        }
    }
    std::cout << "[RadarHIP] " << map_file << ": " << mesh.n_faces << " triangles, " << mesh.n_objects << " objects" << std::endl;
    for(size_t i = 0; mesh.object_names && i < mesh.n_objects; i++)
    {
        std::cout << "[RadarHIP]   object " << i << ": " << mesh.object_names[i] << std::endl;
    }
    std::vector<float> verts(mesh.verts, mesh.verts + 3 * mesh.n_verts);
    std::vector<uint32_t> faces(mesh.faces, mesh.faces + 3 * mesh.n_faces);
    std::vector<uint32_t> face_object(mesh.face_object_id, mesh.face_object_id + mesh.n_faces);
    rr_free_mesh(&mesh);
    init(verts, faces, face_object, devices, build_on_gpu);
}

RadarHIP::RadarHIP(
    std::shared_ptr<ros::NodeHandle> nh_p,
    std::shared_ptr<tf2_ros::Buffer> tf_buffer,
    std::shared_ptr<tf2_ros::TransformListener> tf_listener,
    std::string map_frame,
    std::string sensor_frame,
    const std::vector<float>& verts,
    const std::vector<uint32_t>& faces,
    const std::vector<uint32_t>& face_object,
    const std::vector<int>& devices,
    bool build_on_gpu)
:Base(nh_p, tf_buffer, tf_listener, map_frame, sensor_frame)
{
    init(verts, faces, face_object, devices, build_on_gpu);
}

void RadarHIP::init(
    const std::vector<float>& verts,
    const std::vector<uint32_t>& faces,
    const std::vector<uint32_t>& face_object,
    const std::vector<int>& devices,
    bool build_on_gpu)
{
    if(rr_abi_version() != RR_ABI_VERSION)
    {
        throw std::runtime_error("[RadarHIP] libradarays_mi355.so does not match radarays_mi355.h");
    }
    m_multi = rr_create_multi(devices.data(), (int)devices.size());
    if(!m_multi)
    {
        // no HIP device, no CPU fallback: the node falls back by configuration (~hip:=false), not silently
        throw std::runtime_error(std::string("[RadarHIP] ") + rr_multi_last_error(nullptr));
    }
    m_ctx = rr_multi_ctx(m_multi, 0);
    // built once (host SAH with spatial splits, or LBVH on device 0), then copied device to device
    int rc;
    if(build_on_gpu)
    {
        rc = rr_multi_set_mesh_gpu(m_multi, verts.data(), verts.size() / 3, faces.data(), faces.size() / 3,
                                   face_object.empty() ? nullptr : face_object.data());
    } else {
        rc = rr_multi_set_mesh(m_multi, verts.data(), verts.size() / 3, faces.data(), faces.size() / 3,
                               face_object.empty() ? nullptr : face_object.data());
    }
    if(rc)
    {
        std::string e = rr_multi_last_error(m_multi);
        rr_destroy_multi(m_multi);
        m_multi = nullptr;
        throw std::runtime_error("[RadarHIP] " + e);
    }
    m_nh_p->param<double>("hip_sweep_duration", m_sweep_duration, 0.0);
}

RadarHIP::~RadarHIP()
{
    if(m_multi)
    {
        rr_destroy_multi(m_multi);
    }
}

sensor_msgs::ImagePtr RadarHIP::fail()
{
    ROS_WARN_STREAM("[RadarHIP] " << rr_multi_last_error(m_multi));
    return sensor_msgs::ImagePtr();
}

sensor_msgs::ImagePtr RadarHIP::wrap(const unsigned char* pixels, ros::Time stamp) const
{
    // RadarCPU.cpp:555-561: mono8, height n_cells, width n_angles, step n_angles
    cv::Mat view(m_cfg.n_cells, m_n_angles, CV_8UC1, const_cast<unsigned char*>(pixels));
    sensor_msgs::ImagePtr msg = cv_bridge::CvImage(std_msgs::Header(), "mono8", view).toImageMsg();
    msg->header.stamp = stamp;
    msg->header.frame_id = m_sensor_frame;
    return msg;
}

bool RadarHIP::pushState()
{
    // ---- m_cfg + m_params.model + the constants of Radar::Radar (Radar.cpp:22-32) -> rr_config ----
    rr_config c;
    rr_default_config(&c);
    c.n_cells = m_cfg.n_cells;
    c.n_angles = m_radar_model.theta.size;
    c.n_reflections = m_params.model.n_reflections;
    c.signal_denoising = m_cfg.signal_denoising;
    c.signal_denoising_triangular_width = m_cfg.signal_denoising_triangular_width;
    c.signal_denoising_triangular_mode = m_cfg.signal_denoising_triangular_mode;
    c.signal_denoising_gaussian_width = m_cfg.signal_denoising_gaussian_width;
    c.signal_denoising_gaussian_mode = m_cfg.signal_denoising_gaussian_mode;
    c.signal_denoising_mb_width = m_cfg.signal_denoising_mb_width;
    c.signal_denoising_mb_mode = m_cfg.signal_denoising_mb_mode;
    c.ambient_noise = m_cfg.ambient_noise;
    c.scroll_image = m_cfg.scroll_image;
    c.record_multi_reflection = m_cfg.record_multi_reflection;
    c.record_multi_path = m_cfg.record_multi_path;
    c.multipath_threshold = m_cfg.multipath_threshold;
    c.resolution = m_cfg.resolution;
    c.energy_max = m_cfg.energy_max;
    c.signal_max = m_cfg.signal_max;
    c.ambient_noise_at_signal_0 = m_cfg.ambient_noise_at_signal_0;
    c.ambient_noise_at_signal_1 = m_cfg.ambient_noise_at_signal_1;
    c.ambient_noise_energy_max = m_cfg.ambient_noise_energy_max;
    c.ambient_noise_energy_min = m_cfg.ambient_noise_energy_min;
    c.ambient_noise_energy_loss = m_cfg.ambient_noise_energy_loss;
    c.wave_energy_threshold = m_wave_energy_threshold;
    c.theta_min = m_radar_model.theta.min;
    c.theta_inc = m_radar_model.theta.inc;
    c.range_max = 1000.0f;     // make_model gives every pass's OnDn model range [0, 1000] (radar_algorithms.cpp:157-158)
    if(rr_multi_set_config(m_multi, &c))
    {
        fail();
        return false;
    }
    m_n_angles = c.n_angles;

    // ---- materials, re-read from the parameter server before every frame by the node (radar_simulator.cpp:85,200) ----
    std::vector<rr_material> mats(m_params.materials.data.size());
    for(size_t i = 0; i < mats.size(); i++)
    {
        const RadarMaterial& m = m_params.materials.data[i];
        mats[i].velocity = m.velocity;
        mats[i].ambient = m.ambient;
        mats[i].diffuse = m.diffuse;
        mats[i].specular = m.specular;
    }
    std::vector<int32_t> object_materials(m_object_materials.begin(), m_object_materials.end());
    if(rr_multi_set_materials(m_multi, mats.data(), mats.size(), object_materials.data(), object_materials.size(), m_material_id_air))
    {
        fail();
        return false;
    }

    // ---- beam samples: RadarCPU.cpp:136-145 ----
    if(m_resample)
    {
        // the reference's sample_cone_local seeds itself from std::random_device (radar_algorithms.cpp:258-259); the seeded
        // twin of the library draws the same distribution and lets parameter batches repeat the draw for other widths
        m_beam_seed = std::random_device()();
        std::vector<float> dirs(3 * (size_t)m_params.model.n_samples);
        if(rr_sample_cone_local(m_beam_seed, m_params.model.beam_width, m_params.model.n_samples, m_cfg.beam_sample_dist,
                                m_cfg.beam_sample_dist_normal_p_in_cone, dirs.data()))
        {
            ROS_WARN_STREAM("[RadarHIP] beam_sample_dist " << m_cfg.beam_sample_dist << " is not one of 0..3");
            return false;
        }
        DirectedWave wave;                 // RadarCPU.cpp:106-114: what every wave of the beam starts as
        wave.energy = 1.0;
        wave.polarization = 0.5;
        wave.frequency = 76.5;
        wave.velocity = 0.3;
        wave.material_id = 0;
        wave.time = 0.0;
        wave.ray.orig = {0.0, 0.0, 0.0};
        m_waves_start.assign(m_params.model.n_samples, wave);
        for(size_t i = 0; i < m_waves_start.size(); i++)
        {
            m_waves_start[i].ray.dir = {dirs[3 * i + 0], dirs[3 * i + 1], dirs[3 * i + 2]};
        }
        if(rr_multi_set_beam_samples(m_multi, dirs.data(), m_waves_start.size()))
        {
            fail();
            return false;
        }
        m_resample = false;
    }

    // ---- ambient noise: one U(0,1) * 1000 per column and frame from std::random_device (RadarCPU.cpp:461-472) ----
    if(m_cfg.ambient_noise)
    {
        std::mt19937 gen(std::random_device{}());
        std::uniform_real_distribution<float> dist_uni(0.0, 1.0);
        std::vector<float> rnd((size_t)RR_MAX_BATCH * m_n_angles);     // a fresh row for every frame of a batch
        for(size_t i = 0; i < rnd.size(); i++)
        {
            rnd[i] = dist_uni(gen) * 1000.0;
        }
        if(rr_multi_set_noise_offsets(m_multi, rnd.data(), rnd.size()))
        {
            fail();
            return false;
        }
    }
    return true;
}

int RadarHIP::lookupSweep(ros::Time stamp, std::vector<float>& sweep, std::vector<char>& skipped)
{
    // RadarCPU.cpp:190-196: with include_motion the loop calls updateTsm() once per azimuth and `continue`s -- column left
    // zero, no ros::spinOnce() -- while no transform has ever arrived; afterwards ros::spinOnce() (RadarCPU.cpp:544-547)
    // lets TF move on before the next column
    const ros::Time end = (stamp == ros::Time(0)) ? ros::Time::now() : stamp;
    int n_ok = 0;
    sweep.assign(7 * (size_t)m_n_angles, 0.0f);
    skipped.assign(m_n_angles, 0);
    for(int angle_id = 0; angle_id < m_n_angles; angle_id++)
    {
        bool ok;
        if(m_sweep_duration > 0.0)
        {
            ok = updateTsm(end - ros::Duration((1.0 - (double)angle_id / m_n_angles) * m_sweep_duration));
        } else {
            ok = updateTsm();
        }
        float* p = &sweep[7 * (size_t)angle_id];
        if(!ok)
        {
            skipped[angle_id] = 1;
            p[3] = 1.0f;                    // identity: rendered, then blanked
            continue;
        }
        to_pose7(Tsm_last, p);
        n_ok++;
        ros::spinOnce();
    }
    return n_ok;
}

sensor_msgs::ImagePtr RadarHIP::simulate(ros::Time stamp)
{
    sensor_msgs::ImagePtr msg;

    if(m_polar_image.rows != m_cfg.n_cells)
    {
        m_polar_image = cv::Mat_<unsigned char>(m_cfg.n_cells, m_radar_model.theta.size);
    }
    m_polar_image.setTo(cv::Scalar(0));

    // without motion: update Tsm only once (RadarCPU.cpp:127-134)
    if(!m_cfg.include_motion)
    {
        if(!updateTsm())
        {
            std::cout << "Couldn't get Transform between sensor and map. Skipping..." << std::endl;
            return msg;
        }
    }

    if(!pushState())
    {
        return msg;
    }

    std::vector<float> sweep;
    std::vector<char> skipped;
    int n_ok = m_n_angles;
    if(m_cfg.include_motion)
    {
        n_ok = lookupSweep(stamp, sweep, skipped);
    }
    if(rr_multi_set_motion_poses(m_multi, m_cfg.include_motion ? sweep.data() : nullptr, m_cfg.include_motion ? (size_t)m_n_angles : 0))
    {
        return fail();
    }

    if(n_ok > 0)
    {
        float pose[7];
        to_pose7(Tsm_last, pose);
        // one device: the same bytes as rr_simulate; several: azimuth blocks + one RCCL gather to device 0
        if(rr_multi_simulate(m_multi, pose, m_polar_image.data))
        {
            return fail();
        }
        for(int angle_id = 0; m_cfg.include_motion && angle_id < m_n_angles; angle_id++)
        {
            if(skipped[angle_id])
            {
                m_polar_image.col((m_cfg.scroll_image + angle_id) % m_n_angles).setTo(cv::Scalar(0));   // RadarCPU.cpp:457
            }
        }
    }

    return wrap(m_polar_image.data, stamp);
}

std::vector<sensor_msgs::ImagePtr> RadarHIP::simulateBatch(const std::vector<rm::Transform>& poses, ros::Time stamp)
{
    std::vector<sensor_msgs::ImagePtr> out;
    if(!pushState() || rr_multi_set_motion_poses(m_multi, nullptr, 0))
    {
        return out;
    }
    const size_t npx = (size_t)m_cfg.n_cells * m_n_angles;
    std::vector<unsigned char> pixels((size_t)RR_MAX_BATCH * npx);
    std::vector<float> flat(7 * (size_t)RR_MAX_BATCH);
    for(size_t first = 0; first < poses.size(); first += RR_MAX_BATCH)
    {
        const size_t n = std::min(poses.size() - first, (size_t)RR_MAX_BATCH);
        for(size_t k = 0; k < n; k++)
        {
            to_pose7(poses[first + k], &flat[7 * k]);
        }
        if(rr_multi_simulate_batch(m_multi, flat.data(), (int)n, pixels.data()))
        {
            fail();
            return out;
        }
        for(size_t k = 0; k < n; k++)
        {
            out.push_back(wrap(&pixels[k * npx], stamp));
        }
    }
    return out;
}

std::vector<sensor_msgs::ImagePtr> RadarHIP::simulateSweeps(const std::vector<std::vector<rm::Transform> >& sweeps, ros::Time stamp)
{
    std::vector<sensor_msgs::ImagePtr> out;
    if(!pushState())
    {
        return out;
    }
    const size_t npx = (size_t)m_cfg.n_cells * m_n_angles;
    std::vector<unsigned char> pixels((size_t)RR_MAX_BATCH * npx);
    std::vector<float> table, flat;
    for(size_t first = 0; first < sweeps.size(); first += RR_MAX_BATCH)
    {
        const size_t n = std::min(sweeps.size() - first, (size_t)RR_MAX_BATCH);
        table.assign(n * 7 * (size_t)m_n_angles, 0.0f);
        flat.assign(7 * n, 0.0f);
        for(size_t k = 0; k < n; k++)
        {
            if(sweeps[first + k].size() != (size_t)m_n_angles)
            {
                ROS_WARN_STREAM("[RadarHIP] a sweep needs one pose per azimuth (" << m_n_angles << ")");
                return out;
            }
            for(int a = 0; a < m_n_angles; a++)
            {
                to_pose7(sweeps[first + k][a], &table[(k * m_n_angles + a) * 7]);
            }
            to_pose7(sweeps[first + k][0], &flat[7 * k]);     // ignored while a table is set, but must be a valid pose
        }
        // row k of the table = the per-azimuth poses of frame k of the batch
        if(rr_multi_set_motion_poses(m_multi, table.data(), n * (size_t)m_n_angles)
           || rr_multi_simulate_batch(m_multi, flat.data(), (int)n, pixels.data()))
        {
            fail();
            break;
        }
        for(size_t k = 0; k < n; k++)
        {
            out.push_back(wrap(&pixels[k * npx], stamp));
        }
    }
    rr_multi_set_motion_poses(m_multi, nullptr, 0);
    return out;
}

bool RadarHIP::simulateParamSets(const std::vector<RadarParams>& sets, ros::Time stamp, std::vector<sensor_msgs::ImagePtr>* images,
                                 const sensor_msgs::Image* real, std::vector<double>* psnr)
{
    if(!updateTsm())
    {
        std::cout << "Couldn't get Transform between sensor and map. Skipping..." << std::endl;
        return false;
    }
    if(!pushState() || rr_multi_set_motion_poses(m_multi, nullptr, 0))
    {
        return false;
    }
    const size_t n_mat = m_params.materials.data.size();
    const size_t n_beam = m_params.model.n_samples;
    const size_t npx = (size_t)m_cfg.n_cells * m_n_angles;
    if(sets.empty() || sets.size() > (size_t)RR_MAX_BATCH)
    {
        ROS_WARN_STREAM("[RadarHIP] 1.." << RR_MAX_BATCH << " parameter sets per call");
        return false;
    }
    if(real && (real->data.size() != npx || !psnr))
    {
        ROS_WARN_STREAM("[RadarHIP] the real image must be n_cells x n_angles mono8 (and psnr given)");
        return false;
    }
    std::vector<rr_material> mats;
    mats.reserve(sets.size() * n_mat);
    std::vector<std::vector<float> > dirs(sets.size());
    std::vector<rr_param_set> ps(sets.size());
    for(size_t k = 0; k < sets.size(); k++)
    {
        const RadarParams& p = sets[k];
        if(p.materials.data.size() != n_mat || p.model.n_samples != n_beam)
        {
            ROS_WARN_STREAM("[RadarHIP] every parameter set needs the loaded number of materials and the current n_samples");
            return false;
        }
        for(size_t i = 0; i < n_mat; i++)
        {
            rr_material m;
            m.velocity = p.materials.data[i].velocity;
            m.ambient = p.materials.data[i].ambient;
            m.diffuse = p.materials.data[i].diffuse;
            m.specular = p.materials.data[i].specular;
            mats.push_back(m);
        }
        if(std::abs(p.model.beam_width - m_params.model.beam_width) > 1e-7f)
        {
            // same seed as the current beam: sets that differ only in beam_width see the same variates
            dirs[k].assign(3 * n_beam, 0.0f);
            if(rr_sample_cone_local(m_beam_seed, p.model.beam_width, n_beam, m_cfg.beam_sample_dist,
                                    m_cfg.beam_sample_dist_normal_p_in_cone, dirs[k].data()))
            {
                return false;
            }
        }
        ps[k].n_reflections = (int32_t)p.model.n_reflections;
        ps[k].reserved_ = 0;
    }
    for(size_t k = 0; k < sets.size(); k++)
    {
        ps[k].materials = &mats[k * n_mat];
        ps[k].beam_dirs = dirs[k].empty() ? nullptr : dirs[k].data();
    }
    std::vector<unsigned char> pixels(images ? sets.size() * npx : 0);
    if(psnr)
    {
        psnr->assign(sets.size(), 0.0);
    }
    float pose[7];
    to_pose7(Tsm_last, pose);
    if(rr_simulate_param_sets(m_ctx, pose, ps.data(), (int)sets.size(), n_mat, images ? pixels.data() : nullptr,
                              real ? real->data.data() : nullptr, real ? psnr->data() : nullptr))
    {
        ROS_WARN_STREAM("[RadarHIP] " << rr_last_error(m_ctx));
        return false;
    }
    if(images)
    {
        images->clear();
        for(size_t k = 0; k < sets.size(); k++)
        {
            images->push_back(wrap(&pixels[k * npx], stamp));
        }
    }
    return true;
}

} // namespace radarays_ros
