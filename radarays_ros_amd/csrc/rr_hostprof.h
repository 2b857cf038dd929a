// rr_hostprof.h -- where the HOST time of a call goes (RR_HOST_PROFILE=1): named scopes accumulate wall time and counts,
// a table goes to stderr when the process ends.  Developer tooling for the host cost of the rr_multi call
// (tools/cpp_bench.cpp multi); off, a scope costs one predictable branch.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace rr {

struct HostProf {
    static constexpr int kMax = 24;
    bool on = false;
    const char* name[kMax] = {};
    double secs[kMax] = {};
    uint64_t calls[kMax] = {};
    HostProf()
    {
        on = getenv("RR_HOST_PROFILE") && atoi(getenv("RR_HOST_PROFILE")) != 0;
    }
    ~HostProf()
    {
        if (!on) return;
        std::fprintf(stderr, "[rr host profile] %-34s %10s %12s %10s\n", "scope", "calls", "total ms", "us / call");
        for (int k = 0; k < kMax; k++)
            if (calls[k]) std::fprintf(stderr, "[rr host profile] %-34s %10llu %12.3f %10.2f\n", name[k], (unsigned long long)calls[k], 1e3 * secs[k], 1e6 * secs[k] / (double)calls[k]);
    }
};
inline HostProf g_host_prof;

// (not thread-safe by design: the profile is read with RR_MULTI_THREADS=0, one thread issuing everything)
struct HostProfScope {
    int k; std::chrono::steady_clock::time_point t0;
    HostProfScope(int k_, const char* name) : k(g_host_prof.on ? k_ : -1)
    {
        if (k >= 0) { g_host_prof.name[k] = name; t0 = std::chrono::steady_clock::now(); }
    }
    ~HostProfScope()
    {
        if (k >= 0) { g_host_prof.secs[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g_host_prof.calls[k]++; }
    }
};

}  // namespace rr
