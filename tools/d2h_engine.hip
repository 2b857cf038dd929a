// d2h_engine.hip -- which engine carries a hipMemcpyAsync from HBM to page-locked host memory on this box, what it costs the
// kernels that run beside it, and whether an environment switch changes either.
//
// The images of a batch leave the GPU either as plain copies (one-pass workloads, the flush at the end of a run) or trickled
// out by a few waves of the next batch's trace launches (rr_simulate_batch_host_async) -- the trickle exists because a plain
// copy beside the kernels cost ~7 % of the frame rate (DESIGN_EXPERIMENTS.md, round 3).  rocprofv3 shows those plain copies
// as `__amd_rocclr_copyBuffer` KERNELS (a blit shader writing over PCIe), not as SDMA transfers.  This tool measures, for one
// process environment (run it once per setting of GPU_FORCE_BLIT_COPY_SIZE / HSA_ENABLE_SDMA / ...):
//   A  copy alone            : GB/s of 11 MB (8 images) device -> pinned host, 40 copies back to back on one stream
//   B  kernel alone          : a VALU-bound kernel filling the chip at 8 waves per SIMD, 40 launches on another stream
//   C  both at once          : the same 40 + 40; time until both streams are done, kernel slow-down, copy GB/s
// Modes (argv[1]): how a copy is ordered behind the kernel that produced its data --
//   plain       the copy stream carries nothing but copies
//   samestream  a small kernel ahead of every copy on the copy's own stream (what a frame lane does: k_assemble, then the copy)
//   event       the small kernel runs on a third stream; the copy stream waits for its event, then copies
//   hipcc --offload-arch=gfx950 -O3 tools/d2h_engine.hip -o /tmp/d2h_engine && /tmp/d2h_engine [mode]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_busy(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; i++) {
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
    }
    // ... and a little memory traffic, as every real kernel has
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ void k_small(uint8_t* d, size_t n) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = 7; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "plain";
    const int M = !strcmp(mode, "samestream") ? 1 : !strcmp(mode, "event") ? 2 : 0;
    const size_t bytes = 8ull * 3424 * 400;         // one batch of mono8 images
    const int n = 40;
    uint8_t* d = nullptr; uint8_t* h = nullptr; float* o = nullptr;
    CK(hipMalloc(&d, bytes)); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    const int blocks = 256 * 8 * 4;                 // 8 waves per SIMD on 256 CUs, 4 rounds
    CK(hipMalloc(&o, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMemset(d, 7, bytes));
    hipStream_t sc, sk, sp; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    hipEvent_t evp[64]; for (int i = 0; i < 64; i++) CK(hipEventCreateWithFlags(&evp[i], hipEventDisableTiming));
    int evk = 0;
    // one copy, ordered behind its producer as the mode says
    auto copy = [&]() -> hipError_t {
        if (M == 1) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, sc, d, (size_t)16384);
        if (M == 2) {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, sp, d, (size_t)16384);
            hipEvent_t e = evp[evk++ & 63];
            hipError_t r = hipEventRecord(e, sp); if (r != hipSuccess) return r;
            r = hipStreamWaitEvent(sc, e, 0); if (r != hipSuccess) return r;
        }
        return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, sc);
    };
    const int iters = 20000;
    // warm up
    for (int i = 0; i < 3; i++) { CK(copy()); hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, sk, o, iters); }
    CK(hipDeviceSynchronize());
    const char* envs[] = { "GPU_FORCE_BLIT_COPY_SIZE", "HSA_ENABLE_SDMA", "GPU_BLIT_ENGINE_TYPE", "DEBUG_CLR_LIMIT_BLIT_WG", "ROC_P2P_SDMA_SIZE" };
    printf("mode %s; env:", mode);
    for (const char* e : envs) printf(" %s=%s", e, getenv(e) ? getenv(e) : "-");
    printf("\n");
    // A
    double t0 = now();
    for (int i = 0; i < n; i++) CK(copy());
    CK(hipStreamSynchronize(sc));
    const double tA = now() - t0;
    // B
    t0 = now();
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, sk, o, iters);
    CK(hipStreamSynchronize(sk));
    const double tB = now() - t0;
    // C: as many copies as fit beside the kernels (the copies re-issued until the kernels are done would blur the
    // picture: issue n of each, time each stream by an event)
    hipEvent_t e0, ec, ek; CK(hipEventCreate(&e0)); CK(hipEventCreate(&ec)); CK(hipEventCreate(&ek));
    t0 = now();
    for (int i = 0; i < n; i++) {
        hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, sk, o, iters);
        CK(copy());
    }
    CK(hipStreamSynchronize(sc));
    const double tCc = now() - t0;
    CK(hipStreamSynchronize(sk));
    const double tCk = now() - t0;
    printf("A copy alone   : %7.3f ms per 11 MB = %5.1f GB/s\n", 1e3 * tA / n, bytes * n / tA / 1e9);
    printf("B kernel alone : %7.3f ms per launch\n", 1e3 * tB / n);
    printf("C both         : copies done after %7.3f ms (%5.1f GB/s while sharing), kernels after %7.3f ms (%.3f x alone)\n",
           1e3 * tCc, bytes * n / tCc / 1e9, 1e3 * tCk, tCk / tB);
    // did the bytes arrive?
    int bad = 0; for (size_t i = 0; i < bytes; i += 4099) bad += h[i] != 7;
    printf("check: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
