// C++ driver of the C++ host mirror (include/radarays_ros_amd/RadarHIP.hpp) -- used by
// tests/test_gpu_cpp_host.py: reads a scene + parameters from a simple binary file written by
// the test, runs RadarHIP::simulate() the way radar_simulator.cpp:200-208 would, writes the
// mono8 image.  No Python, no torch: only the C ABI.
#include <radarays_ros_amd/RadarHIP.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>

using namespace radarays_ros_amd;

template <typename T>
static std::vector<T> rd(std::ifstream& f)
{
    uint64_t n = 0; f.read((char*)&n, 8);
    std::vector<T> v(n); f.read((char*)v.data(), (std::streamsize)(n * sizeof(T)));
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin [map.obj]\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    auto verts = rd<float>(f); auto faces = rd<uint32_t>(f); auto fobj = rd<uint32_t>(f);
    auto mats = rd<float>(f); auto objmat = rd<int32_t>(f); auto beams = rd<float>(f); auto pose = rd<float>(f);
    auto cfgv = rd<double>(f);   // n_reflections, ambient_noise, scroll_image, tri_width, energy_max, signal_max, resolution

    try {
        RadarHIP radar("map", "navtech", verts, faces, fobj, 0);
        ImagePtr none = radar.simulate(0.0);                       // no transform yet -> null ImagePtr
        if (none) { std::fprintf(stderr, "expected a null image before updateTsm\n"); return 3; }
        std::vector<RadarMaterial> m(mats.size() / 4);
        for (size_t i = 0; i < m.size(); i++) m[i] = { mats[4 * i], mats[4 * i + 1], mats[4 * i + 2], mats[4 * i + 3] };
        radar.loadParams(m, std::vector<int>(objmat.begin(), objmat.end()), 0);
        RadarModelConfig cfg;                                       // cfg/mulran_kaist_dyncfg.yaml values come from the file
        cfg.n_reflections = (int)cfgv[0]; cfg.ambient_noise = (int)cfgv[1]; cfg.scroll_image = (int)cfgv[2];
        cfg.signal_denoising_triangular_width = (int)cfgv[3]; cfg.energy_max = cfgv[4]; cfg.signal_max = cfgv[5];
        cfg.resolution = cfgv[6]; cfg.beam_width = 10.0; cfg.n_samples = (int)(beams.size() / 3); cfg.include_motion = false;
        radar.updateDynCfg(cfg);
        radar.setBeamSamples(beams);
        radar.updateTsm(pose.data());
        ImagePtr img = radar.simulate(42.5);
        if (!img) { std::fprintf(stderr, "simulate failed: %s\n", radar.lastError().c_str()); return 4; }
        if (img->encoding != "mono8" || img->step != img->width || img->stamp != 42.5 || img->frame_id != "navtech") return 5;
        std::ofstream o(argv[2], std::ios::binary);
        uint32_t hw[2] = { img->height, img->width };
        o.write((const char*)hw, 8);
        o.write((const char*)img->data.data(), (std::streamsize)img->data.size());
        // parameter batch: table 0 = the loaded materials (must reproduce img), table 1 = ambient halved
        std::vector<std::vector<RadarMaterial>> sets(2, m);
        for (auto& x : sets[1]) x.ambient *= 0.5f;
        std::vector<ImagePtr> batch = radar.simulateMaterialSets(sets, 43.0);
        if (batch.size() != 2 || batch[0]->data != img->data || batch[1]->data == img->data) {
            std::fprintf(stderr, "simulateMaterialSets mismatch: %s\n", radar.lastError().c_str()); return 7;
        }
        // the same backend constructed over a device LIST (rr_multi; one device here): identical bytes
        {
            RadarHIP multi("map", "navtech", verts, faces, fobj, std::vector<int>{ 0 }, /*build_on_gpu=*/true);   // the tree does not change an image
            multi.loadParams(m, std::vector<int>(objmat.begin(), objmat.end()), 0);
            multi.updateDynCfg(cfg); multi.setBeamSamples(beams); multi.updateTsm(pose.data());
            ImagePtr img2 = multi.simulate(42.5);
            if (!img2 || img2->data != img->data) { std::fprintf(stderr, "device-list backend differs: %s\n", multi.lastError().c_str()); return 8; }
            // include_motion with the SAME pose for every azimuth is the same frame
            std::vector<float> sweep; for (int a = 0; a < 400; a++) sweep.insert(sweep.end(), pose.begin(), pose.end());
            RadarModelConfig cm = cfg; cm.include_motion = true;
            multi.updateDynCfg(cm); multi.setMotionPoses(sweep);
            ImagePtr img3 = multi.simulate(42.5);
            if (!img3 || img3->data != img->data) { std::fprintf(stderr, "include_motion sweep differs: %s\n", multi.lastError().c_str()); return 9; }
            // offline generation: three poses in one set of launches; frame 0 is `pose`, frame 2 another one
            std::vector<float> three;
            for (int k = 0; k < 3; k++) { three.insert(three.end(), pose.begin(), pose.end()); three[7 * k + 4] += 0.25f * k; }
            multi.updateDynCfg(cfg);
            std::vector<ImagePtr> b3 = multi.simulateBatch(three, 50.0);
            if (b3.size() != 3 || b3[0]->data != img->data || b3[2]->data == img->data || b3[1]->stamp != 50.0) {
                std::fprintf(stderr, "simulateBatch: %s\n", multi.lastError().c_str()); return 19;
            }
            // ... and the same three frames as sweeps whose 400 poses are all equal: the same bytes, one table per frame
            std::vector<float> sweeps3;
            for (int k = 0; k < 3; k++) for (int a = 0; a < 400; a++) sweeps3.insert(sweeps3.end(), three.begin() + 7 * k, three.begin() + 7 * k + 7);
            multi.updateDynCfg(cm);
            std::vector<ImagePtr> s3 = multi.simulateSweeps(sweeps3, 51.0);
            if (s3.size() != 3) { std::fprintf(stderr, "simulateSweeps: %s\n", multi.lastError().c_str()); return 20; }
            for (int k = 0; k < 3; k++) if (s3[k]->data != b3[k]->data) { std::fprintf(stderr, "simulateSweeps: frame %d differs from simulateBatch\n", k); return 20; }
            ImagePtr img4 = multi.simulate(42.5);                 // simulate() re-installs ITS sweep table
            if (!img4 || img4->data != img->data) { std::fprintf(stderr, "simulate() after simulateSweeps differs\n"); return 21; }
        }
        // the optimiser's evaluation, batched (radaray_opti.py): three RadarParams -- the current ones, a narrower beam with
        // two passes, other materials with one pass -- as images and as scores against the first image
        {
            RadarHIP opt("map", "navtech", verts, faces, fobj, 0);
            opt.setBeamSeed(7);
            opt.loadParams(m, std::vector<int>(objmat.begin(), objmat.end()), 0);
            opt.updateDynCfg(cfg); opt.updateTsm(pose.data());
            ImagePtr cur = opt.simulate(5.0);
            if (!cur) { std::fprintf(stderr, "optimiser backend failed: %s\n", opt.lastError().c_str()); return 15; }
            RadarParams p0 = opt.getParams(), p1 = p0, p2 = p0;
            p1.model.beam_width = (float)(5.0 * M_PI / 180.0); p1.model.n_reflections = 2;     // as Radar::updateDynCfg converts it
            for (auto& x : p2.materials) x.ambient *= 0.5f;
            p2.model.n_reflections = 1;
            std::vector<ImagePtr> imgs; std::vector<double> psnr;
            if (!opt.simulateParamSets({ p0, p1, p2 }, 6.0, &imgs, cur.get(), &psnr) || imgs.size() != 3 || psnr.size() != 3) {
                std::fprintf(stderr, "simulateParamSets: %s\n", opt.lastError().c_str()); return 15;
            }
            if (imgs[0]->data != cur->data || !std::isinf(psnr[0]) || !(psnr[1] > 0 && psnr[1] < 100) || !(psnr[2] > 0 && psnr[2] < 100)) {
                std::fprintf(stderr, "simulateParamSets: set 0 must reproduce the current image (psnr %g %g %g)\n", psnr[0], psnr[1], psnr[2]); return 16;
            }
            // set 1 one by one: the same parameters through setParams / updateDynCfg + simulate()
            RadarModelConfig c1 = cfg; c1.beam_width = 5.0; c1.n_reflections = 2;
            opt.updateDynCfg(c1);
            ImagePtr one = opt.simulate(7.0);
            if (!one || one->data != imgs[1]->data) { std::fprintf(stderr, "simulateParamSets: set 1 differs from the same parameters one by one\n"); return 17; }
            std::vector<double> only;
            opt.updateDynCfg(cfg);
            if (!opt.simulateParamSets({ p0, p1, p2 }, 8.0, nullptr, cur.get(), &only) || only != psnr) { std::fprintf(stderr, "scores without images differ\n"); return 18; }
        }
        // dynamic reconfigure of the beam (Radar.cpp:188-218 sets m_resample, RadarCPU.cpp:136-145 re-draws): a backend that
        // was never given samples draws them itself (rr_sample_cone_local, seeded here), and draws again when beam_width changes
        {
            RadarHIP fresh("map", "navtech", verts, faces, fobj, 0);
            fresh.setBeamSeed(42);
            fresh.loadParams(m, std::vector<int>(objmat.begin(), objmat.end()), 0);
            fresh.updateDynCfg(cfg); fresh.updateTsm(pose.data());
            ImagePtr a = fresh.simulate(1.0);
            if (!a) { std::fprintf(stderr, "resampling backend failed: %s\n", fresh.lastError().c_str()); return 10; }
            const std::vector<float> b0 = fresh.beamSamples();
            if (b0.size() != beams.size()) { std::fprintf(stderr, "resampled %zu values, expected %zu\n", b0.size(), beams.size()); return 10; }
            uint64_t nb = b0.size();
            o.write((const char*)&nb, 8); o.write((const char*)b0.data(), (std::streamsize)(nb * sizeof(float)));
            o.write((const char*)a->data.data(), (std::streamsize)a->data.size());
            RadarModelConfig narrow = cfg; narrow.beam_width = 6.0;
            fresh.updateDynCfg(narrow);
            ImagePtr b = fresh.simulate(2.0);
            if (!b || fresh.beamSamples() == b0 || b->data == a->data) { std::fprintf(stderr, "beam_width change did not resample\n"); return 11; }
            fresh.updateDynCfg(narrow);                                  // nothing changed: no re-draw
            const std::vector<float> b1 = fresh.beamSamples();
            ImagePtr c2 = fresh.simulate(3.0);
            if (!c2 || fresh.beamSamples() != b1 || c2->data != b->data) { std::fprintf(stderr, "unchanged config re-drew the beam\n"); return 12; }
        }
        // the map file route (rm::import_embree_map, radar_simulator.cpp:149): rr_load_mesh_file gives the arrays back
        if (argc > 3) {
            rr_mesh mesh; char err[256];
            if (rr_load_mesh_file(argv[3], &mesh, err, sizeof(err))) { std::fprintf(stderr, "rr_load_mesh_file: %s\n", err); return 13; }
            const bool same = mesh.n_verts * 3 == verts.size() && mesh.n_faces * 3 == faces.size() &&
                              std::equal(verts.begin(), verts.end(), mesh.verts) && std::equal(faces.begin(), faces.end(), mesh.faces) &&
                              std::equal(fobj.begin(), fobj.end(), mesh.face_object_id);
            rr_free_mesh(&mesh);
            if (!same) { std::fprintf(stderr, "rr_load_mesh_file: arrays differ from the scene\n"); return 14; }
        }
        const rr_stats& st = radar.lastStats();
        std::printf("ok %u x %u wave_passes %llu signals %llu\n", img->height, img->width,
                    (unsigned long long)st.wave_passes, (unsigned long long)st.signals);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 6;
    }
    return 0;
}
