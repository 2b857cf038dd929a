// C++ driver of the C++ host mirror (include/radarays_ros_amd/RadarHIP.hpp) -- used by
// tests/test_gpu_cpp_host.py: reads a scene + parameters from a simple binary file written by
// the test, runs RadarHIP::simulate() the way radar_simulator.cpp:200-208 would, writes the
// mono8 image.  No Python, no torch: only the C ABI.
#include <radarays_ros_amd/RadarHIP.hpp>

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
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 2; }
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
