// lanes.hip -- what does a wave64 VALU instruction cost on gfx950 as a function of HOW MANY lanes are active?
//
// tools/halfwave.hip found something nobody asked for: with at most 8 lanes of a wave active the same instruction stream
// takes 4.7x as long (16 lanes: 1.2x; 32 or 64: 1.0x), at an unchanged shader clock.  k_trace's waves spend their last
// iterations with one or two live quads (4 - 8 lanes), so this sweep measures it properly: active lanes 64 .. 1, as quads
// packed at the bottom of the wave and spread over it, for independent FMA chains (issue-bound) and for ONE dependent chain
// (latency-bound), with 8 / 4 / 1 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/lanes.hip -o /tmp/lanes && /tmp/lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long mask, int iters)
{
    float a[8];
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 1e-3f + k;
    const float m = 0.999f, c = 1e-3f;
#ifdef LANES_BY_BRANCH
    // the mask as the compiler makes it: a branch around the loop (s_and_saveexec), not a hand-written s_mov to EXEC
    if ((mask >> (threadIdx.x & 63)) & 1ull)
#else
    const unsigned long long saved = __builtin_amdgcn_read_exec();
    asm volatile("s_mov_b64 exec, %0" :: "s"(mask & saved));
#endif
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64 / CHAINS; u++) {
            if (CHAINS == 8)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
            else
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        }
    }
#ifndef LANES_BY_BRANCH
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
#endif
    float s = 0; for (int k = 0; k < 8; k++) s += a[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static unsigned long long quads_mask(int n_quads, bool spread)
{
    unsigned long long m = 0;
    for (int k = 0; k < n_quads; k++) { const int q = spread ? (k * 16) / n_quads : k; m |= 0xFull << (4 * q); }
    return m;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* d; CK(hipMalloc(&d, (size_t)cus * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 8000;
    printf("%s, %d CUs; 64 x %d v_fma_f32 per wave; ns per wave-instruction per SIMD (8 independent chains | 1 dependent chain)\n", p.gcnArchName, cus, iters);
    for (int wps : { 8, 4, 1 }) {
        const int blocks = cus * wps;          // 4 waves per block = one per SIMD
        printf("-- %d wave%s per SIMD\n", wps, wps == 1 ? "" : "s");
        for (int spread = 0; spread < 2; spread++)
            for (int nq : { 16, 12, 8, 6, 4, 3, 2, 1 }) {
                const unsigned long long mask = quads_mask(nq, spread != 0);
                double t[2];
                for (int v = 0; v < 2; v++) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipEventRecord(e0));
                        if (v == 0) hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, mask, iters);
                        else        hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(256), 0, 0, d, mask, iters);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    t[v] = best * 1e6 / ((double)iters * 64 * wps);
                }
                printf("   %2d quads (%2d lanes) %s   %6.2f | %6.2f\n", nq, 4 * nq, spread ? "spread " : "packed ", t[0], t[1]);
            }
    }
    return 0;
}
