// Pure C++ caller of the C ABI (no Python, no torch): frames per second of rr_simulate_device on its frame
// lanes and of the frame-batch entry points, images resident in HBM.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/cpp_bench.cpp \
//       -o /tmp/cpp_bench -L radarays_ros_amd -lradarays_mi355 -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/radarays_ros_amd
//   /tmp/cpp_bench scene.bin [frames] [batch]      (scene.bin as written by tests/test_cpp_host.py: write_scene)
#include <hip/hip_runtime_api.h>
#include <radarays_mi355.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <vector>

template <typename T> static std::vector<T> rd(std::ifstream& f)
{
    uint64_t n = 0; f.read((char*)&n, 8); std::vector<T> v(n); f.read((char*)v.data(), (std::streamsize)(n * sizeof(T))); return v;
}
#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s: %s\n", #x, rr_last_error(c)); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s scene.bin [frames] [batch]\n", argv[0]); return 2; }
    const int frames = argc > 2 ? atoi(argv[2]) : 2000, batch = argc > 3 ? atoi(argv[3]) : 4;
    std::ifstream f(argv[1], std::ios::binary);
    auto verts = rd<float>(f); auto faces = rd<uint32_t>(f); auto fobj = rd<uint32_t>(f);
    auto mats = rd<float>(f); auto objmat = rd<int32_t>(f); auto beams = rd<float>(f); auto pose = rd<float>(f); auto cfgv = rd<double>(f);
    rr_ctx* c = rr_create(0);
    if (!c) { std::fprintf(stderr, "%s\n", rr_last_error(nullptr)); return 6; }
    CK(rr_set_mesh(c, verts.data(), verts.size() / 3, faces.data(), faces.size() / 3, fobj.data()));
    CK(rr_set_materials(c, (const rr_material*)mats.data(), mats.size() / 4, objmat.data(), objmat.size(), 0));
    rr_config cfg; rr_default_config(&cfg);
    cfg.n_reflections = (int)cfgv[0]; cfg.ambient_noise = (int)cfgv[1]; cfg.scroll_image = (int)cfgv[2];
    cfg.signal_denoising_triangular_width = (int)cfgv[3]; cfg.energy_max = cfgv[4]; cfg.signal_max = cfgv[5]; cfg.resolution = cfgv[6];
    CK(rr_set_config(c, &cfg));
    CK(rr_set_beam_samples(c, beams.data(), beams.size() / 3));
    std::vector<float> rnd(cfg.n_angles); for (int i = 0; i < cfg.n_angles; i++) rnd[i] = 1000.f * (float)((i * 2654435761u % 1000u) / 1000.0);
    CK(rr_set_noise_offsets(c, rnd.data(), rnd.size()));
    const size_t npx = (size_t)cfg.n_cells * cfg.n_angles;
    uint8_t *d_img = nullptr, *d_cols = nullptr;
    if (hipMalloc((void**)&d_img, npx * 32) != hipSuccess || hipMalloc((void**)&d_cols, npx * 32) != hipSuccess) return 3;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<float> poses(7 * 32);
    for (int k = 0; k < 32; k++) { for (int j = 0; j < 7; j++) poses[7 * k + j] = pose[j]; poses[7 * k + 4] += 0.05f * k; }
    // (1) one frame per call, pipelined over the context's frame lanes
    for (int k = 0; k < 20; k++) CK(rr_simulate_device(c, &poses[7 * (k % 16)], d_img, s));
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; k++) CK(rr_simulate_device(c, &poses[7 * (k % 16)], d_img + (size_t)(k % 4) * npx, s));
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    std::printf("rr_simulate_device: %.0f images/s (host enqueue %.1f us/frame)\n", frames / std::chrono::duration<double>(t2 - t0).count(),
                1e6 * std::chrono::duration<double>(t1 - t0).count() / frames);
    // (2) `batch` poses per call + one assemble launch, 4 calls in flight on 4 streams
    hipStream_t st[4]; for (auto& x : st) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    const int steps = frames / batch;
    for (int k = 0; k < 8; k++) {
        CK(rr_simulate_batch_columns_device(c, poses.data(), batch, 0, cfg.n_angles, d_cols + (size_t)(k % 4) * 8 * npx, st[k % 4]));
        CK(rr_assemble_frames_device(c, d_cols + (size_t)(k % 4) * 8 * npx, cfg.n_angles, npx, batch, npx, d_img + (size_t)(k % 4) * 8 * npx, st[k % 4]));
    }
    hipDeviceSynchronize();
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; k++) {
        CK(rr_simulate_batch_device(c, &poses[7 * (k % (32 - batch + 1))], batch, d_img + (size_t)(k % 4) * 8 * npx, st[k % 4]));
    }
    t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    t2 = std::chrono::steady_clock::now();
    std::printf("batch of %d poses per call, 4 streams: %.0f images/s (host enqueue %.1f us/step)\n", batch,
                steps * batch / std::chrono::duration<double>(t2 - t0).count(), 1e6 * std::chrono::duration<double>(t1 - t0).count() / steps);
    rr_destroy(c);
    return 0;
}
