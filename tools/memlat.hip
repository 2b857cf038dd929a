// Dependent-load latency of one lane on MI355X (pointer chase over a random cycle of 128-B lines): what ONE traversal step
// pays at least for its fetch, by the level that serves it.  build: hipcc --offload-arch=gfx950 -O2 tools/memlat.hip -o memlat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>

__global__ void chase(const uint32_t* __restrict__ next, uint32_t start, int hops, uint32_t* out, long long* cycles)
{
    uint32_t i = start;
    for (int k = 0; k < 1024; k++) { asm volatile("" : "+v"(i)); i = next[(size_t)i * 32]; }          // warm the path into whatever cache holds it
    const long long t0 = wall_clock64();
    for (int k = 0; k < hops; k++) { asm volatile("" : "+v"(i)); i = next[(size_t)i * 32]; }      // the index in a VGPR: a vector load through L1
    const long long t1 = wall_clock64();
    *out = i; *cycles = t1 - t0;
}

int main()
{
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const size_t sizes[] = { 16u << 10, 256u << 10, 2u << 20, 16u << 20, 128u << 20, 1024u << 20, 2047u << 20 };
    for (size_t bytes : sizes) {
        const size_t lines = bytes / 128;
        std::vector<uint32_t> perm(lines);
        for (size_t i = 0; i < lines; i++) perm[i] = (uint32_t)i;
        std::mt19937_64 g(1); std::shuffle(perm.begin(), perm.end(), g);
        std::vector<uint32_t> h(lines * 32, 0);
        for (size_t i = 0; i < lines; i++) h[(size_t)perm[i] * 32] = perm[(i + 1) % lines];       // one cycle through all lines
        uint32_t* d = nullptr; uint32_t* out = nullptr; long long* cyc = nullptr;
        hipMalloc((void**)&d, h.size() * 4); hipMalloc((void**)&out, 4); hipMalloc((void**)&cyc, 8);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const int hops = (int)std::min<size_t>(lines * 4, 200000);
        hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, perm[0], hops, out, cyc);
        hipDeviceSynchronize();
        long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%8.2f MB: %7.1f ns per dependent 4-B load (%d hops)\n", bytes / 1048576.0, (double)c / hops * 1e6 / rate_khz, hops);
        (void)hipFree(d); (void)hipFree(out); (void)hipFree(cyc);
    }
    return 0;
}
