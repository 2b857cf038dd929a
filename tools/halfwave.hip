// halfwave.hip -- does gfx950 skip an all-inactive 32-lane half of a wave64 VALU instruction?
//
// A wave64 VALU op issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md "Wave scheduling").  If the hardware skipped a
// half whose 32 EXEC bits are all zero, k_trace could keep the quads that hold a LEAF in one half of the wave and the
// quads that hold a NODE in the other, and a mixed iteration's two paths would each cost half.
// Method: 8 waves per SIMD (issue-bound, like k_trace), each running ITER rounds of 8 independent v_fma chains with the
// EXEC mask set once before the loop (s_mov exec); the time of the loop under different masks, per kernel launch.
//   hipcc --offload-arch=gfx950 -O3 tools/halfwave.hip -o /tmp/halfwave && /tmp/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long mask, int iters, unsigned long long* clk)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    const unsigned long long saved = __builtin_amdgcn_read_exec();
    // restrict EXEC for the whole loop: lanes outside `mask` execute nothing
    asm volatile("s_mov_b64 exec, %0" :: "s"(mask & saved));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 8;                 // 8 blocks x 4 waves per CU = 8 waves per SIMD
    float* d; CK(hipMalloc(&d, (size_t)blocks * 256 * sizeof(float)));
    unsigned long long* clk; CK(hipHostMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } cases[] = {
        { "all 64 lanes            ", ~0ull },
        { "lower half (lanes 0-31) ", 0x00000000FFFFFFFFull },
        { "upper half (lanes 32-63)", 0xFFFFFFFF00000000ull },
        { "even lanes (both halves)", 0x5555555555555555ull },
        { "one lane per half       ", 0x0000000100000001ull },
        { "lane 0 only             ", 0x1ull },
        { "lanes 0-15              ", 0xFFFFull },
        { "one quad per half       ", 0x0000000F0000000Full },
    };
    const int iters = 20000;
    printf("%s, %d CUs, %d blocks x 256 threads, %d x 64 v_fma_f32 per wave\n", p.gcnArchName, cus, blocks, iters);
    double t_all = 0;
    for (auto& c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, c.mask, iters, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        if (c.mask == ~0ull) t_all = best;
        // cycles per wave-instruction per SIMD: time x clock / (instructions per wave x waves per SIMD)
        const double instr_per_simd = (double)iters * 64 * 8;
        // shader clock of block 0's wave: s_memtime ticks per 100 MHz s_memrealtime tick
        printf("%s  %8.3f ms  = %.2fx of all-lanes   (%.2f ns per wave-instruction per SIMD; shader clock %.0f MHz, %.2f cycles per instruction)\n",
               c.name, best, best / t_all, best * 1e6 / instr_per_simd, 100.0 * (double)clk[0] / (double)clk[1],
               (double)clk[0] / instr_per_simd);
    }
    printf("verdict: a half-masked VALU op costs %s\n", "see the ratios above: ~0.5x = the empty half is skipped, ~1.0x = it is not");
    return 0;
}
