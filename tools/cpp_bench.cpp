// Pure C++ caller of the C ABI (no Python, no torch): frames per second of rr_simulate_device on its frame
// lanes and of the frame-batch entry points, images resident in HBM.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/cpp_bench.cpp \
//       -o /tmp/cpp_bench -L radarays_ros_amd -lradarays_mi355 -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/radarays_ros_amd
//   /tmp/cpp_bench scene.bin [frames] [batch] [mode]   (scene.bin as written by tests/test_cpp_host.py: write_scene)
//   mode: (none) device-resident rates | sync: rr_simulate one frame at a time (the reference's call shape) |
//         multi: rr_multi_simulate_batch_async on devices {0} against rr_simulate_batch_host_async (host-resident rates) |
//         graph: the launch chain of one frame captured in a hipGraph against the same chain launched kernel by kernel
//                (run with RR_LANES=1: the single-lane route of rr_simulate_device puts every launch on the caller's stream)
#include <hip/hip_runtime_api.h>
#include <radarays_mi355.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

template <typename T> static std::vector<T> rd(std::ifstream& f)
{
    uint64_t n = 0; f.read((char*)&n, 8); std::vector<T> v(n); f.read((char*)v.data(), (std::streamsize)(n * sizeof(T))); return v;
}
#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s: %s\n", #x, rr_last_error(c)); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s scene.bin [frames] [batch] [sync|multi|graph]\n", argv[0]); return 2; }
    const int frames = argc > 2 ? atoi(argv[2]) : 2000, batch = argc > 3 ? atoi(argv[3]) : 4;
    const std::string mode = argc > 4 ? argv[4] : "";
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
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    if (mode == "sync") {
        // the reference's call shape: one synchronous simulate() per frame, image in (pageable) host memory
        std::vector<uint8_t> img(npx); rr_stats st;
        for (int k = 0; k < 10; k++) CK(rr_simulate(c, &poses[7 * (k % 16)], 0, cfg.n_angles, img.data(), nullptr, &st));
        std::vector<double> ts;
        for (int k = 0; k < frames; k++) {
            auto a = clk::now(); CK(rr_simulate(c, &poses[7 * (k % 16)], 0, cfg.n_angles, img.data(), nullptr, &st)); ts.push_back(secs(a, clk::now()));
        }
        std::sort(ts.begin(), ts.end());
        std::printf("rr_simulate (sync, host image, stats): median %.3f ms  p10 %.3f  p90 %.3f  (%llu wave-passes)\n", 1e3 * ts[ts.size() / 2],
                    1e3 * ts[ts.size() / 10], 1e3 * ts[ts.size() * 9 / 10], (unsigned long long)st.wave_passes);
        rr_destroy(c); return 0;
    }
    if (mode == "graph") {
        hipStream_t gs; hipStreamCreateWithFlags(&gs, hipStreamNonBlocking);
        for (int k = 0; k < 10; k++) CK(rr_simulate_device(c, &poses[0], d_img, gs));
        hipStreamSynchronize(gs);
        std::vector<double> ts;
        for (int k = 0; k < frames; k++) { auto a = clk::now(); CK(rr_simulate_device(c, &poses[0], d_img, gs)); hipStreamSynchronize(gs); ts.push_back(secs(a, clk::now())); }
        std::sort(ts.begin(), ts.end());
        std::printf("kernel by kernel : median %.3f ms per frame (p10 %.3f)\n", 1e3 * ts[ts.size() / 2], 1e3 * ts[ts.size() / 10]);
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        hipError_t e = hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { std::printf("capture refused: %s\n", hipGetErrorString(e)); return 0; }
        const int rc = rr_simulate_device(c, &poses[0], d_img, gs);
        e = hipStreamEndCapture(gs, &g);
        if (rc || e != hipSuccess || !g) { std::printf("capture failed: rc %d (%s), %s\n", rc, rr_last_error(c), hipGetErrorString(e)); return 0; }
        size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e != hipSuccess) { std::printf("instantiate failed: %s\n", hipGetErrorString(e)); return 0; }
        for (int k = 0; k < 10; k++) hipGraphLaunch(ge, gs);
        hipStreamSynchronize(gs);
        ts.clear();
        for (int k = 0; k < frames; k++) { auto a = clk::now(); hipGraphLaunch(ge, gs); hipStreamSynchronize(gs); ts.push_back(secs(a, clk::now())); }
        std::sort(ts.begin(), ts.end());
        std::printf("hipGraph (%zu nodes): median %.3f ms per frame (p10 %.3f)\n", nn, 1e3 * ts[ts.size() / 2], 1e3 * ts[ts.size() / 10]);
        // HOST time of the two ways to enqueue the chain (what a graph can save for a caller that issues many chains: rr_multi)
        {
            const int burst = 8;
            double t_k = 0.0, t_g = 0.0, t_s = 0.0;
            for (int rep = 0; rep < 40; rep++) {
                hipStreamSynchronize(gs);
                auto a = clk::now();
                for (int k = 0; k < burst; k++) CK(rr_simulate_device(c, &poses[0], d_img, gs));
                t_k += secs(a, clk::now());
                hipStreamSynchronize(gs);
                a = clk::now();
                for (int k = 0; k < burst; k++) hipGraphLaunch(ge, gs);
                t_g += secs(a, clk::now());
            }
            // one kernel node's parameters replaced before every launch (how a replay would get its poses)
            std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
            hipGraphNode_t kn = nullptr; hipKernelNodeParams kp{};
            for (auto nd : nodes) { hipGraphNodeType ty; hipGraphNodeGetType(nd, &ty); if (ty == hipGraphNodeTypeKernel) { kn = nd; break; } }
            if (kn && hipGraphKernelNodeGetParams(kn, &kp) == hipSuccess) {
                for (int rep = 0; rep < 40; rep++) {
                    hipStreamSynchronize(gs);
                    auto a = clk::now();
                    for (int k = 0; k < burst; k++) hipGraphExecKernelNodeSetParams(ge, kn, &kp);
                    t_s += secs(a, clk::now());
                }
            }
            std::printf("host time to enqueue one chain: kernel by kernel %.1f us, hipGraphLaunch %.1f us, hipGraphExecKernelNodeSetParams %.1f us per node\n",
                        1e6 * t_k / (40 * burst), 1e6 * t_g / (40 * burst), 1e6 * t_s / (40 * burst));
        }
        rr_destroy(c); return 0;
    }
    if (mode == "multi") {
        // host-resident rates: the C++ drop-in's route (rr_multi over {0}) against the ctx's own host delivery
        const int ring = 8, steps = frames / batch;
        std::vector<uint8_t*> host(ring);
        for (auto& h : host) { h = (uint8_t*)rr_host_alloc((size_t)batch * npx); if (!h) return 4; }
        hipStream_t st4[4]; for (auto& x : st4) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        auto run_ctx = [&](int n) -> int {
            for (int k = 0; k < n; k++) {
                uint8_t* h = host[k % ring];
                if (rr_wait_host(c, h)) return 1;
                if (rr_simulate_batch_host_async(c, &poses[7 * (k % (32 - batch + 1))], batch, h, st4[k % 4])) return 1;
            }
            return rr_wait_host(c, nullptr) || rr_synchronize(c, nullptr);
        };
        if (run_ctx(64)) { std::fprintf(stderr, "%s\n", rr_last_error(c)); return 1; }
        auto a = clk::now();
        if (run_ctx(steps)) { std::fprintf(stderr, "%s\n", rr_last_error(c)); return 1; }
        const double t_ctx = secs(a, clk::now());
        std::printf("rr_simulate_batch_host_async, %d poses per call, 4 streams: %.0f images/s host-resident\n", batch, steps * batch / t_ctx);
        rr_destroy(c);
        // CPP_BENCH_NDEV=n with RR_MULTI_LOOPBACK=1: device 0 listed n times -- the n-device orchestration (n contexts, n x the
        // launches per call from ONE host thread, plan-driven copies in place of the collective) on one GPU: what the host side costs
        const int ndev = getenv("CPP_BENCH_NDEV") ? std::max(1, atoi(getenv("CPP_BENCH_NDEV"))) : 1;
        std::vector<int> devs((size_t)ndev, 0);
        rr_multi* m = rr_create_multi(devs.data(), ndev);
        if (!m) { std::fprintf(stderr, "%s\n", rr_multi_last_error(nullptr)); return 6; }
#define MK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s: %s\n", #x, rr_multi_last_error(m)); return 1; } } while (0)
        MK(rr_multi_set_mesh(m, verts.data(), verts.size() / 3, faces.data(), faces.size() / 3, fobj.data()));
        MK(rr_multi_set_materials(m, (const rr_material*)mats.data(), mats.size() / 4, objmat.data(), objmat.size(), 0));
        MK(rr_multi_set_config(m, &cfg));
        MK(rr_multi_set_beam_samples(m, beams.data(), beams.size() / 3));
        MK(rr_multi_set_noise_offsets(m, rnd.data(), rnd.size()));
        double t_enq = 0.0;
        auto run_multi = [&](int n) -> int {
            t_enq = 0.0;
            for (int k = 0; k < n; k++) {
                uint8_t* h = host[k % ring];
                if (rr_multi_wait(m, h)) return 1;
                auto q0 = clk::now();
                if (rr_multi_simulate_batch_async(m, &poses[7 * (k % (32 - batch + 1))], batch, h)) return 1;
                t_enq += secs(q0, clk::now());
            }
            return rr_multi_wait(m, nullptr);
        };
        MK(run_multi(64));
        a = clk::now();
        MK(run_multi(steps));
        const double t_m = secs(a, clk::now());
        std::printf("rr_multi_simulate_batch_async over %d device entr%s, %d poses per call: %.0f images/s host-resident = %.1f %% of the ctx route; "
                    "host time per call %.1f us (%.1f us per frame)\n", ndev, ndev == 1 ? "y" : "ies", batch,
                    steps * batch / t_m, 100.0 * t_ctx / t_m, 1e6 * t_enq / steps, 1e6 * t_enq / steps / batch);
        // the CPU cost of a call proper: bursts of as many calls as there are slots, issued onto a DRAINED object -- no call
        // of a burst waits for a slot, so the time is what the host spends issuing the launches (the figure above is the
        // period of the steady state, which contains the wait for the GPU as soon as the GPU is the slower side)
        {
            const int slots = getenv("RR_MULTI_SLOTS") ? std::max(1, std::min(8, atoi(getenv("RR_MULTI_SLOTS")))) : 4;
            double t_cpu = 0.0; int n_calls = 0;
            for (int rep = 0; rep < 50; rep++) {
                MK(rr_multi_wait(m, nullptr));
                auto q0 = clk::now();
                for (int k = 0; k < slots; k++)
                    if (rr_multi_simulate_batch_async(m, &poses[7 * (k % (32 - batch + 1))], batch, host[k % ring])) return 1;
                t_cpu += secs(q0, clk::now()); n_calls += slots;
            }
            MK(rr_multi_wait(m, nullptr));
            std::printf("host CPU time per call, slots free (bursts of %d onto a drained object): %.1f us = %.1f us per device entry\n",
                        slots, 1e6 * t_cpu / n_calls, 1e6 * t_cpu / n_calls / ndev);
        }
        rr_destroy_multi(m);
        for (auto h : host) rr_host_free(h);
        return 0;
    }
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
