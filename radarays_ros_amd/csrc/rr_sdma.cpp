// rr_sdma.cpp -- images to host memory over the SDMA engines, whichever HIP runtime serves the process (round 6).
//
// The reference's simulate() ends with the image in host memory (RadarCPU.cpp:542,555-561).  Which engine carries a
// hipMemcpyAsync to page-locked memory is the HIP runtime's choice: ROCm 7.2's uses SDMA (54 GB/s beside a chip-filling kernel,
// which it slows by 0.5 %), the ROCm 7.0.2 runtime bundled in the PyTorch wheel launches a blit KERNEL that competes with the
// frame kernels -- 27-35k images/s on config 2 where the link allows 39.4k -- and a copy kernel of this library's own does no
// better (24-31k in every shape tried: workgroup count, size, unrolling, stores in flight, one XCD, a stream of its own;
// DESIGN_EXPERIMENTS.md round 6): PCIe-paced stores issued by shader cores are the problem, not who launches them.
// Both runtimes sit on the same ROCr (HSA) layer, and ROCr's hsa_amd_memory_async_copy IS the SDMA path.  This file uses it
// directly: two worker threads per context take (event, device source, host destination, bytes) jobs in turn; for each, a
// worker waits for the HIP event (the batch's assemble), submits the copy with a completion signal and waits for that signal.  A
// copy is therefore ordered behind the kernels that produce the image by the event and ahead of the lane's next use by the
// job's completion (rr_api.hip waits for it before it lets a lane's next batch overwrite the image).
//
// The HSA instance is never initialised here: the one the active HIP runtime initialised is looked up among the loaded
// objects (a process can hold two copies of libhsa-runtime64 -- the wheel's and the system's -- of which only one is live)
// and recognised by the fact that it knows a device pointer.  Anything unexpected (no live instance, a host pointer ROCr does
// not know, an error from the copy) switches the path off for the context; the caller falls back to the stream-ordered copies.
#include "rr_sdma.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <dlfcn.h>
#include <link.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace rr {

namespace {

struct HsaApi {
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    decltype(&hsa_amd_memory_async_copy) async_copy = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
    decltype(&hsa_system_get_info) system_info = nullptr;       // (optional: the timestamp frequency, for the length of the active wait)
    bool ok() const { return pointer_info && async_copy && signal_create && signal_destroy && signal_store && signal_wait; }
};

int collect_hsa_objects(struct dl_phdr_info* info, size_t, void* data)
{
    if (info->dlpi_name && std::strstr(info->dlpi_name, "libhsa-runtime64")) static_cast<std::vector<std::string>*>(data)->push_back(info->dlpi_name);
    return 0;
}

// the loaded libhsa-runtime64 whose runtime is LIVE (initialised by the HIP runtime in use): it knows `device_ptr`
bool find_live_hsa(const void* device_ptr, HsaApi& api, hsa_agent_t& gpu, std::string& why)
{
    std::vector<std::string> objs;
    dl_iterate_phdr(collect_hsa_objects, &objs);
    if (objs.empty()) { why = "no libhsa-runtime64 is loaded in this process"; return false; }
    for (const std::string& path : objs) {
        void* h = dlopen(path.c_str(), RTLD_NOLOAD | RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        HsaApi a;
        a.pointer_info = (decltype(a.pointer_info))dlsym(h, "hsa_amd_pointer_info");
        a.async_copy = (decltype(a.async_copy))dlsym(h, "hsa_amd_memory_async_copy");
        a.signal_create = (decltype(a.signal_create))dlsym(h, "hsa_signal_create");
        a.signal_destroy = (decltype(a.signal_destroy))dlsym(h, "hsa_signal_destroy");
        a.signal_store = (decltype(a.signal_store))dlsym(h, "hsa_signal_store_relaxed");
        a.signal_wait = (decltype(a.signal_wait))dlsym(h, "hsa_signal_wait_scacquire");
        a.system_info = (decltype(a.system_info))dlsym(h, "hsa_system_get_info");
        if (!a.ok()) continue;
        hsa_amd_pointer_info_t pi; std::memset(&pi, 0, sizeof(pi)); pi.size = sizeof(pi);
        if (a.pointer_info(const_cast<void*>(device_ptr), &pi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
            pi.type == HSA_EXT_POINTER_TYPE_HSA && pi.agentOwner.handle != 0) {
            api = a; gpu = pi.agentOwner;
            return true;            // (the handle stays open: the object is the process' runtime anyway)
        }
    }
    why = "no loaded libhsa-runtime64 knows the device buffer (is its runtime initialised?)";
    return false;
}

}  // namespace

struct SdmaCopier {
    // TWO workers take jobs off one queue in turn, each with its own completion signal: while one waits for its copy to finish
    // the other already waits for the next batch's event and submits behind it, so the engine goes from copy to copy without
    // the wake-up of a thread in between (one worker: 35.6k images/s on config 2 from a C++ caller, where the HIP runtime's own
    // SDMA path -- several copies queued with dependency signals -- reaches 39.4k)
    static constexpr int kWorkers = 2;
    HsaApi api;
    hsa_agent_t gpu{};
    hsa_signal_t sig[kWorkers]{};
    int device = 0;
    std::thread th[kWorkers];
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    struct Job { uint64_t id; hipEvent_t after; const void* src; void* dst; size_t bytes; };
    std::deque<Job> q;                        // jobs no worker has taken yet
    uint64_t next_id = 1, done_upto = 0;      // every job <= done_upto is complete ...
    std::set<uint64_t> done_ahead;            // ... and these beyond it (two copies may finish out of order)
    bool quit = false;
    // RR_SDMA_ACTIVE_US > 0: the worker first waits for a copy's completion signal ACTIVELY for that long (in ticks of the HSA
    // timestamp clock) before it sleeps on it -- a blocked wait is woken by an interrupt, possibly tens of microseconds late on a
    // busy host.  Measured on the pool (config 2, 12 runs): 39.0-39.1k images/s with 600 or 2000 us, 39.0-39.25k with 0 -- with two
    // workers the other one has the next copy queued already, so the default is 0: no core is burnt
    uint64_t active_ticks = 0;
    std::atomic<int> failed{0};
    std::string err;

    bool is_done(uint64_t job) const { return job <= done_upto || done_ahead.count(job) != 0; }     // (mu held)
    void mark_done(uint64_t job)                                                                     // (mu held)
    {
        done_ahead.insert(job);
        while (!done_ahead.empty() && *done_ahead.begin() == done_upto + 1) { done_upto++; done_ahead.erase(done_ahead.begin()); }
    }

    void run(int w)
    {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return quit || !q.empty(); });
                if (q.empty()) return;                 // quit, nothing left
                j = q.front(); q.pop_front();
                if (quit) {                            // the context is being destroyed: images nobody waited for are dropped (their
                    mark_done(j.id);                   // buffers may be gone already), the header's rule for rr_destroy
                    cv_done.notify_all();
                    continue;
                }
            }
            // the kernels that produce the image first (the streams are non-blocking: nothing else orders a copy behind them)
            bool ok = true;
            std::string why;
            if (hipEventSynchronize(j.after) != hipSuccess) {
                // (e.g. the event's stream is being captured by the caller just now: the runtime refuses to wait on it.)  The image
                // must not be copied before the kernels that write it are done: wait for the whole device instead, then deliver
                // by a blocking copy below; the route is switched off, the caller's next deliveries take the stream-ordered one
                ok = false; why = "hipEventSynchronize failed (is the stream being captured?)"; (void)hipGetLastError();
                for (int tries = 0; tries < 2000 && hipDeviceSynchronize() != hipSuccess; tries++) { (void)hipGetLastError(); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
            }
            if (ok && failed.load()) { ok = false; why = "an earlier job failed"; }
            if (ok) {
                hsa_amd_pointer_info_t pi; std::memset(&pi, 0, sizeof(pi)); pi.size = sizeof(pi);
                if (api.pointer_info(j.dst, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN || pi.agentOwner.handle == 0) {
                    ok = false; why = "the host buffer is not known to the HSA runtime (page-locked memory is needed)";
                } else {
                    api.signal_store(sig[w], 1);
                    const hsa_status_t st = api.async_copy(j.dst, pi.agentOwner, j.src, gpu, j.bytes, 0, nullptr, sig[w]);
                    if (st != HSA_STATUS_SUCCESS) { ok = false; why = "hsa_amd_memory_async_copy: status " + std::to_string((int)st); }
                    else {
                        // (bounded waits: a copy that never completes must not hang the caller's rr_wait_host for ever)
                        hsa_signal_value_t v = 1;
                        const auto t0 = std::chrono::steady_clock::now();
                        if (active_ticks) v = api.signal_wait(sig[w], HSA_SIGNAL_CONDITION_LT, 1, active_ticks, HSA_WAIT_STATE_ACTIVE);
                        while (v >= 1) {
                            v = api.signal_wait(sig[w], HSA_SIGNAL_CONDITION_LT, 1, 50000000ull, HSA_WAIT_STATE_BLOCKED);
                            if (v >= 1 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;
                        }
                        if (v >= 1) { ok = false; why = "the SDMA copy did not complete within 20 s"; }
                        else if (v < 0) { ok = false; why = "the SDMA copy reported an error through its completion signal"; }
                    }
                }
            }
            // whatever went wrong, the image is delivered: a blocking copy by the HIP runtime (the event has completed or failed)
            if (!ok) { (void)hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost); (void)hipGetLastError(); }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok && !failed.load()) { failed.store(1); err = why; }
                mark_done(j.id);
            }
            cv_done.notify_all();
        }
    }
};

SdmaCopier* sdma_create(int hip_device, const void* any_device_ptr, std::string& why)
{
    SdmaCopier* s = new SdmaCopier();
    s->device = hip_device;
    if (!find_live_hsa(any_device_ptr, s->api, s->gpu, why)) { delete s; return nullptr; }
    {
        uint64_t freq = 100000000ull;          // 100 MHz unless the runtime says otherwise
        if (s->api.system_info) { uint64_t f = 0; if (s->api.system_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &f) == HSA_STATUS_SUCCESS && f) freq = f; }
        const long us = getenv("RR_SDMA_ACTIVE_US") ? atol(getenv("RR_SDMA_ACTIVE_US")) : 0;
        s->active_ticks = us > 0 ? (uint64_t)((double)freq * 1e-6 * (double)us) : 0;
    }
    int made = 0;
    for (; made < SdmaCopier::kWorkers; made++) if (s->api.signal_create(1, 0, nullptr, &s->sig[made]) != HSA_STATUS_SUCCESS) break;
    int started = 0;
    if (made == SdmaCopier::kWorkers) {
        try { for (; started < SdmaCopier::kWorkers; started++) s->th[started] = std::thread([s, started] { s->run(started); }); }
        catch (...) { }
    }
    if (started != SdmaCopier::kWorkers) {
        why = made != SdmaCopier::kWorkers ? "hsa_signal_create failed" : "no thread for the copy workers";
        { std::lock_guard<std::mutex> lk(s->mu); s->quit = true; }
        s->cv_job.notify_all();
        for (int k = 0; k < started; k++) if (s->th[k].joinable()) s->th[k].join();
        for (int k = 0; k < made; k++) (void)s->api.signal_destroy(s->sig[k]);
        delete s;
        return nullptr;
    }
    return s;
}

void sdma_destroy(SdmaCopier* s)
{
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->quit = true; }
    s->cv_job.notify_all();
    for (std::thread& t : s->th) if (t.joinable()) t.join();      // (copies in flight complete; queued ones are dropped: rr_destroy drops images nobody waited for)
    for (hsa_signal_t& g : s->sig) (void)s->api.signal_destroy(g);
    delete s;
}

uint64_t sdma_submit(SdmaCopier* s, hipEvent_t after, const void* d_src, void* h_dst, size_t bytes)
{
    uint64_t id;
    { std::lock_guard<std::mutex> lk(s->mu); id = s->next_id++; s->q.push_back({ id, after, d_src, h_dst, bytes }); }
    s->cv_job.notify_one();
    return id;
}

bool sdma_done(SdmaCopier* s, uint64_t job)
{
    std::lock_guard<std::mutex> lk(s->mu);
    return s->is_done(job);
}

void sdma_wait(SdmaCopier* s, uint64_t job)
{
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_done.wait(lk, [&] { return s->is_done(job); });
}

void sdma_wait_all(SdmaCopier* s)
{
    std::unique_lock<std::mutex> lk(s->mu);
    const uint64_t last = s->next_id - 1;
    s->cv_done.wait(lk, [&] { return s->done_upto >= last; });
}

bool sdma_failed(SdmaCopier* s, std::string* why)
{
    if (!s->failed.load()) return false;
    if (why) { std::lock_guard<std::mutex> lk(s->mu); *why = s->err; }
    return true;
}

}  // namespace rr
