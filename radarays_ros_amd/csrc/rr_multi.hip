// rr_multi.hip -- multi-GPU fan-out BEHIND the C ABI (SURVEY.md §8b "Threading": one backend object per process,
// radar_simulator.cpp:145-176; §8e: azimuth columns are independent, RadarCPU.cpp:155).
//
// One process, one rr_ctx per device, mesh / parameters replicated, device i renders the contiguous azimuth block
// rr_partition(n_angles, n, i) of every frame of a call in ONE set of launches on its own stream; ONE RCCL
// collective per call GATHERS the blocks on device 0 (the root) -- one ncclGroup of send / recv pairs along
// rr_multi_plan: one piece per device for equal blocks, one per device and frame for ragged ones; nobody but the
// root receives anything -- then the root transposes into mono8 images and copies them to the caller's host buffer.
// Calls are PIPELINED (round 4): rr_multi_simulate_batch_async enqueues a batch on one of RR_MULTI_SLOTS (4) slots --
// own stream + block buffer per device, own receive / image buffers on the root -- and returns; rr_multi_wait is the
// consumer's fence.  The render of batch k+1 overlaps the collective, transpose and D2H copy of batch k, like the
// slots of dist.py's step loop.  Per-batch error bits travel to page-locked host words (rr_peek_error_bits_async), so
// no call drains a device unless something failed.
// RCCL is loaded at run time (librccl.so.1): the library itself has no link-time dependency on it and
// rr_create_multi() fails with a clear message where it is missing.  Built on the public entry points of
// radarays_mi355.h only.
#include "../../include/radarays_mi355.h"
#include "rr_hostprof.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// the part of the RCCL API this file uses (rccl.h: same prototypes as NCCL 2)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclDataType_t kNcclUint8 = 1;   // ncclUint8 / ncclChar family: ncclInt8 = 0, ncclUint8 = 1
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    int version = 0;                       // ncclGetVersion's code (2.27.7 -> 22707), 0: the library does not say
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* n : { "librccl.so.1", "librccl.so" }) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = "rr_create_multi: librccl.so.1 not found (RCCL is needed for more than one device)"; return false; }
#define RR_SYM(f) f = (decltype(f))dlsym(lib, "nccl" #f); if (!f) { err = "rr_create_multi: librccl lacks nccl" #f; return false; }
        RR_SYM(CommInitAll) RR_SYM(CommDestroy) RR_SYM(Send) RR_SYM(Recv) RR_SYM(GroupStart) RR_SYM(GroupEnd) RR_SYM(GetErrorString)
#undef RR_SYM
        // The prototypes above are declared BY HAND (the library is loaded at run time, rccl.h is not included): they are those of
        // the NCCL 2 API from 2.7 on (ncclSend / ncclRecv; rccl.h of ROCm 7.2: 2.27.7, checked against the header by
        // tests/test_abi.py).  A library that reports a version outside [2.7, 3.0) is refused instead of being called
        // through signatures nobody has compared; one that reports none is accepted and left to the self-test below
        GetVersion = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
        if (GetVersion && GetVersion(&version) == 0) {
            const bool ok = (version >= 20900 && version < 30000) || (version >= 2700 && version < 2900);    // (2.9 changed the code's format)
            if (!ok) {
                err = "rr_create_multi: librccl reports NCCL version code " + std::to_string(version) +
                      ", outside the range this library's hand-declared prototypes were checked for (2.7 .. 2.x)";
                return false;
            }
        } else version = 0;
        return true;
    }
};
Rccl g_rccl;
std::string g_multi_create_error;

template <typename T>
struct Buf {
    T* p = nullptr; size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

// one batch in flight: its own stream and block buffer on every device, the root's receive / image buffers, the host
// words the devices' error bits arrive in
struct MultiSlot {
    std::vector<hipStream_t> streams;     // per device
    std::vector<Buf<uint8_t>> block;      // per device: [n_frames][n_loc_i][n_cells]
    std::vector<hipEvent_t> ev_block;     // per device: block rendered (loopback: the root's copies wait for it)
    Buf<uint8_t> gathered;                // root: what the collective delivers
    Buf<uint8_t> d_imgs;                  // root: [n_frames][n_cells][n_angles]
    hipEvent_t ev_done = nullptr;         // root: images in the caller's buffer
    uint32_t* h_bits = nullptr;           // page-locked [n_devices]: rr_peek_error_bits_async
    const void* dst = nullptr;            // the caller's buffer of the batch in flight
    bool pending = false;
    int failed = 0;                       // != 0: this batch was in flight when another one's error drained the object -- its
                                          // images are not to be trusted (error bits are per frame lane, not per batch, and a
                                          // drain reads and clears them all); reported by the wait for its buffer / the next use of the slot
    bool owns_streams = true;             // one device: twice as many records as streams (see rr_create_multi)
};

// One enqueue thread per device: a call's launches for device i (a dozen kernels, an event, a copy: 60-100 us of host time)
// are issued by worker i while the others issue theirs -- from ONE thread the host side of a call grows with the number
// of devices (measured in loopback, config 2, 8 entries: 450 us per 8-frame call = a cap of 17.6k images/s whatever the GPUs
// do).  A context is used by one thread at a time (the header's rule): worker i is the only one that touches context i
// while a call renders; the caller's thread takes over (collective, root) only after every worker has reported back.
struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
};

struct rr_multi {
    std::vector<std::unique_ptr<MultiWorker>> workers;     // empty: the caller's thread issues everything (the default; RR_MULTI_THREADS=1 starts them)
    std::vector<int> devices;
    std::vector<rr_ctx*> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<MultiSlot> slots;         // batches in flight (RR_MULTI_SLOTS, default 4)
    size_t next_slot = 0;
    rr_config cfg;
    bool have_cfg = false;
    bool loopback = false;                // see rr_create_multi
    bool self_rccl = false;               // RR_MULTI_SELF_RCCL=1 with ONE device: its block travels to itself through RCCL (test switch)
    bool self_rccl_frames = false;        // ... =2: frame by frame (the ragged plan's many pieces in one group)
    std::string err;
};

namespace {

void stop_workers(rr_multi* m)
{
    for (auto& w : m->workers) {
        { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    m->workers.clear();
}

int mfail(rr_multi* m, int code, const std::string& msg) { if (m) m->err = msg; else g_multi_create_error = msg; return code; }

#define RRM_HIP(m, expr)                                                                       \
    do { hipError_t e_ = (expr);                                                               \
         if (e_ != hipSuccess) return mfail((m), -100, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define RRM_NCCL(m, expr)                                                                      \
    do { ncclResult_t r_ = (expr);                                                             \
         if (r_ != 0) return mfail((m), -101, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)
// a setter that failed on device i: keep that context's message
#define RRM_EACH(m, call)                                                                      \
    do { for (size_t i_ = 0; i_ < (m)->ctx.size(); i_++) { rr_ctx* c = (m)->ctx[i_]; int rc_ = (call);   \
             if (rc_) return mfail((m), rc_, std::string("device ") + std::to_string((m)->devices[i_]) + ": " + rr_last_error(c)); } } while (0)

}  // namespace

extern "C" {

namespace {
// Before the first real collective: does the hand-declared ABI mean what this file thinks it means?  Rank 0 of the
// communicator sends 16 bytes of a 64-byte pattern to ITSELF (a send / recv pair to one's own rank inside a group is legal
// NCCL) with the datatype constant this file calls ncclUint8 -- if that constant named a wider type, more than 16 bytes
// would move and the guard bytes behind them would change; if the call signatures were off, nothing sensible would arrive.
// With several devices every other rank then exchanges the same 16 bytes with rank 0, so the first gather of a frame is
// not the first time two devices talk.
bool rccl_selftest(rr_multi* m, std::string& why)
{
    const int n = (int)m->devices.size();
    unsigned char pat[64], back[64];
    for (int i = 0; i < 64; i++) pat[i] = (unsigned char)(0xA5 ^ (i * 7));
    std::vector<unsigned char*> src((size_t)n, nullptr), dst((size_t)n, nullptr);
    auto cleanup = [&]() { for (int i = 0; i < n; i++) { (void)hipSetDevice(m->devices[(size_t)i]); if (src[(size_t)i]) (void)hipFree(src[(size_t)i]); if (dst[(size_t)i]) (void)hipFree(dst[(size_t)i]); } (void)hipSetDevice(m->devices[0]); };
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++) {
        e = hipSetDevice(m->devices[(size_t)i]);
        if (e == hipSuccess) e = hipMalloc((void**)&src[(size_t)i], 64);
        if (e == hipSuccess) e = hipMalloc((void**)&dst[(size_t)i], 64 * (size_t)(i == 0 ? n : 1));
        if (e == hipSuccess) e = hipMemcpy(src[(size_t)i], pat, 64, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(dst[(size_t)i], 0, 64 * (size_t)(i == 0 ? n : 1));
    }
    if (e != hipSuccess) { why = std::string("buffers: ") + hipGetErrorString(e); cleanup(); return false; }
    MultiSlot& S = m->slots[0];
    ncclResult_t gr = g_rccl.GroupStart(), r1 = 0;
    for (int i = 0; i < n && gr == 0 && r1 == 0 && e == hipSuccess; i++) {       // rank i -> rank 0 (i = 0: to itself), 16 of its 64 bytes
        e = hipSetDevice(m->devices[(size_t)i]);
        if (e == hipSuccess) r1 = g_rccl.Send(src[(size_t)i], 16, kNcclUint8, 0, m->comms[(size_t)i], S.streams[(size_t)i]);
        if (e == hipSuccess && r1 == 0) e = hipSetDevice(m->devices[0]);
        if (e == hipSuccess && r1 == 0) r1 = g_rccl.Recv(dst[0] + 64 * (size_t)i, 16, kNcclUint8, i, m->comms[0], S.streams[0]);
    }
    const ncclResult_t ger = g_rccl.GroupEnd();
    if (e != hipSuccess || gr != 0 || r1 != 0 || ger != 0) {
        why = e != hipSuccess ? std::string("hipSetDevice: ") + hipGetErrorString(e) : std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(gr ? gr : (r1 ? r1 : ger));
        cleanup(); return false;
    }
    for (int i = 0; i < n && e == hipSuccess; i++) { e = hipSetDevice(m->devices[(size_t)i]); if (e == hipSuccess) e = hipStreamSynchronize(S.streams[(size_t)i]); }
    bool ok = e == hipSuccess;
    if (!ok) why = std::string("synchronize: ") + hipGetErrorString(e);
    (void)hipSetDevice(m->devices[0]);
    for (int i = 0; i < n && ok; i++) {
        if (hipMemcpy(back, dst[0] + 64 * (size_t)i, 64, hipMemcpyDeviceToHost) != hipSuccess) { why = "read-back failed"; ok = false; break; }
        if (std::memcmp(back, pat, 16) != 0) { why = "the 16 bytes rank " + std::to_string(i) + " sent did not arrive"; ok = false; }
        for (int k = 16; k < 64 && ok; k++) if (back[k] != 0) { why = "ncclSend(count = 16, the constant taken for ncclUint8) moved more than 16 bytes: the datatype enum of this librccl differs"; ok = false; }
    }
    cleanup();
    return ok;
}
}  // namespace

rr_multi* rr_create_multi(const int* devices, int n_devices)
{
    if (!devices || n_devices < 1 || n_devices > 64) { g_multi_create_error = "rr_create_multi: need 1..64 device indices"; return nullptr; }
    // RR_MULTI_LOOPBACK=1 (tests on a one-GPU box): a device may be listed several times; every listed entry gets its own
    // context, block and stream as usual, and the ONE collective of a call is replaced by device-to-device copies that
    // follow the same plan (rr_multi_plan) -- everything of the n > 1 path runs except the RCCL calls themselves
    const bool loopback = getenv("RR_MULTI_LOOPBACK") && atoi(getenv("RR_MULTI_LOOPBACK")) != 0;
    for (int i = 0; i < n_devices && !loopback; i++) for (int j = 0; j < i; j++)
        if (devices[i] == devices[j]) { g_multi_create_error = "rr_create_multi: a device is listed twice"; return nullptr; }
    rr_multi* m = new rr_multi();
    m->loopback = loopback;
    m->devices.assign(devices, devices + n_devices);
    rr_default_config(&m->cfg);
    for (int i = 0; i < n_devices; i++) {
        rr_ctx* c = rr_create(devices[i]);
        if (!c) { g_multi_create_error = std::string("rr_create_multi: ") + rr_last_error(nullptr); rr_destroy_multi(m); return nullptr; }
        m->ctx.push_back(c);
    }
    // batches in flight: 4 streams per device = its 4 hardware queues, the measured optimum of the one-GPU step loop
    int n_slots = getenv("RR_MULTI_SLOTS") ? atoi(getenv("RR_MULTI_SLOTS")) : 4;
    n_slots = std::max(1, std::min(n_slots, 8));
    // One device: a batch owns no buffers here (its images wait on the ctx's frame lane until the lane's next batch carries
    // them out), so the records outnumber the streams two to one -- a call then waits for the batch EIGHT back, not for
    // the one whose deferred images its own launches are about to carry (that wait would force the plain copy every time)
    const bool self_rccl_early = n_devices == 1 && !loopback && getenv("RR_MULTI_SELF_RCCL") && atoi(getenv("RR_MULTI_SELF_RCCL")) != 0;
    m->slots.resize((size_t)((n_devices == 1 && !self_rccl_early) ? 2 * n_slots : n_slots));
    for (size_t si = 0; si < m->slots.size(); si++) {
        MultiSlot& S = m->slots[si];
        S.block.resize((size_t)n_devices);
        S.streams.assign((size_t)n_devices, nullptr); S.ev_block.assign((size_t)n_devices, nullptr);
        bool ok = true;
        S.owns_streams = si < (size_t)n_slots;
        if (!S.owns_streams) S.streams = m->slots[si - (size_t)n_slots].streams;
        for (int i = 0; i < n_devices && ok; i++)
            ok = hipSetDevice(devices[i]) == hipSuccess &&
                 (!S.owns_streams || hipStreamCreateWithFlags(&S.streams[(size_t)i], hipStreamNonBlocking) == hipSuccess) &&
                 hipEventCreateWithFlags(&S.ev_block[(size_t)i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipSetDevice(devices[0]) == hipSuccess && hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipHostMalloc((void**)&S.h_bits, sizeof(uint32_t) * (size_t)n_devices, hipHostMallocDefault) == hipSuccess;
        if (!ok) { g_multi_create_error = "rr_create_multi: stream / event creation failed"; rr_destroy_multi(m); return nullptr; }
        for (int i = 0; i < n_devices; i++) S.h_bits[i] = 0;
    }
    // RR_MULTI_SELF_RCCL=1 (test switch for one-GPU boxes, the complement of the loopback): ONE device whose block goes
    // through the REAL RCCL calls -- ncclCommInitAll with one rank, one group of ncclSend / ncclRecv to itself on the slot's
    // stream -- instead of the single-device route: what the loopback leaves out (library loading, symbols, datatype,
    // group semantics, stream ordering of the collective against render and transpose) runs here
    m->self_rccl = n_devices == 1 && !loopback && getenv("RR_MULTI_SELF_RCCL") && atoi(getenv("RR_MULTI_SELF_RCCL")) != 0;
    m->self_rccl_frames = m->self_rccl && atoi(getenv("RR_MULTI_SELF_RCCL")) == 2;
    if (n_devices > 1 && getenv("RR_MULTI_THREADS") && atoi(getenv("RR_MULTI_THREADS")) != 0) {
        for (int i = 0; i < n_devices; i++) {
            m->workers.emplace_back(new MultiWorker());
            MultiWorker* w = m->workers.back().get();
            const int dev = devices[i];
            try {
                w->th = std::thread([w, dev] {
                    (void)hipSetDevice(dev);
                    std::unique_lock<std::mutex> lk(w->mu);
                    for (;;) {
                        w->cv.wait(lk, [w] { return w->has_job || w->quit; });
                        if (w->quit) return;
                        w->has_job = false;
                        lk.unlock();
                        const int rc = w->job();
                        lk.lock();
                        w->rc = rc; w->done = true;
                        w->cv.notify_all();
                    }
                });
            } catch (...) { stop_workers(m); break; }          // no thread to be had: the caller's thread does the work
        }
    }
    if ((n_devices > 1 && !loopback) || m->self_rccl) {
        // the communicator is owned here: one rank per device of this process
        if (!g_rccl.load(g_multi_create_error)) { rr_destroy_multi(m); return nullptr; }
        m->comms.resize((size_t)n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n_devices, devices);
        if (r != 0) { g_multi_create_error = std::string("rr_create_multi: ncclCommInitAll: ") + g_rccl.GetErrorString(r); m->comms.clear(); rr_destroy_multi(m); return nullptr; }
        std::string why;
        if (!rccl_selftest(m, why)) { g_multi_create_error = "rr_create_multi: RCCL self-test failed: " + why; rr_destroy_multi(m); return nullptr; }
    }
    return m;
}

void rr_destroy_multi(rr_multi* m)
{
    if (!m) return;
    stop_workers(m);
    for (size_t i = 0; i < m->ctx.size(); i++) { (void)hipSetDevice(m->devices[i]); (void)hipDeviceSynchronize(); }
    for (size_t i = 0; i < m->comms.size(); i++) if (m->comms[i]) g_rccl.CommDestroy(m->comms[i]);
    for (MultiSlot& S : m->slots) {
        for (size_t i = 0; i < S.streams.size(); i++) {
            (void)hipSetDevice(m->devices[i]);
            if (i < S.block.size()) S.block[i].release();
            if (S.ev_block[i]) (void)hipEventDestroy(S.ev_block[i]);
            if (S.owns_streams && S.streams[i]) (void)hipStreamDestroy(S.streams[i]);
        }
        (void)hipSetDevice(m->devices[0]);
        S.gathered.release(); S.d_imgs.release();
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        if (S.h_bits) (void)hipHostFree(S.h_bits);
    }
    for (size_t i = 0; i < m->ctx.size(); i++) rr_destroy(m->ctx[i]);
    delete m;
}

const char* rr_multi_last_error(const rr_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
int rr_multi_device_count(const rr_multi* m) { return m ? (int)m->ctx.size() : 0; }
int rr_multi_rccl_version(const rr_multi* m) { return (m && !m->comms.empty()) ? g_rccl.version : 0; }
rr_ctx* rr_multi_ctx(rr_multi* m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }

// ---- replicated state: every setter goes to every device ------------------------------------------------
int rr_multi_set_mesh(rr_multi* m, const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object_id)
{
    if (!m) return -1;
    // ONE build (1.9 s of host time at 10M triangles), then the finished tree goes from device to device (xGMI)
    { rr_ctx* c = m->ctx[0]; const int rc = rr_set_mesh(c, verts, nv, faces, nf, face_object_id);
      if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[0]) + ": " + rr_last_error(c)); }
    for (size_t i = 1; i < m->ctx.size(); i++) {
        rr_ctx* c = m->ctx[i]; const int rc = rr_copy_mesh(c, m->ctx[0]);
        if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[i]) + ": " + rr_last_error(c));
    }
    return 0;
}
int rr_multi_set_mesh_gpu(rr_multi* m, const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object_id)
{
    if (!m) return -1;
    { rr_ctx* c = m->ctx[0]; const int rc = rr_set_mesh_gpu(c, verts, nv, faces, nf, face_object_id);
      if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[0]) + ": " + rr_last_error(c)); }
    for (size_t i = 1; i < m->ctx.size(); i++) {
        rr_ctx* c = m->ctx[i]; const int rc = rr_copy_mesh(c, m->ctx[0]);
        if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[i]) + ": " + rr_last_error(c));
    }
    return 0;
}
int rr_multi_set_materials(rr_multi* m, const rr_material* materials, size_t n_materials,
                           const int32_t* object_materials, size_t n_objects, int32_t material_id_air)
{
    if (!m) return -1;
    RRM_EACH(m, rr_set_materials(c, materials, n_materials, object_materials, n_objects, material_id_air));
    return 0;
}
int rr_multi_set_config(rr_multi* m, const rr_config* cfg)
{
    if (!m) return -1;
    if (!cfg) return mfail(m, -3, "rr_multi_set_config: null config");
    RRM_EACH(m, rr_set_config(c, cfg));
    m->cfg = *cfg; m->have_cfg = true;
    return 0;
}
int rr_multi_set_beam_samples(rr_multi* m, const float* dirs, size_t n)
{
    if (!m) return -1;
    RRM_EACH(m, rr_set_beam_samples(c, dirs, n));
    return 0;
}
int rr_multi_set_noise_offsets(rr_multi* m, const float* rnd, size_t n)
{
    if (!m) return -1;
    RRM_EACH(m, rr_set_noise_offsets(c, rnd, n));
    return 0;
}
int rr_multi_set_motion_poses(rr_multi* m, const float* poses, size_t n)
{
    if (!m) return -1;
    RRM_EACH(m, rr_set_motion_poses(c, poses, n));
    return 0;
}

// ---- the data plan of one call: pure arithmetic, exported so that it can be checked without a GPU ------------
// equal blocks : device r's whole block buffer (`bytes_per_device`) travels as ONE piece and lands at r * bytes_per_device
//                of the root's buffer, laid out [device][frame][n_loc][n_cells]
// ragged blocks: device r sends, for every frame f, the n_loc_r * n_cells bytes at send_off[r][f] of ITS block buffer
//                ([frame][n_loc_r][n_cells]) to the root, which receives them at recv_off[r][f] of [frame][n_angles][n_cells]
int rr_multi_plan(int n_angles, int n_cells, int n_devices, int n_frames, int* equal_blocks, size_t* bytes_per_device,
                  size_t* send_off, size_t* recv_off, size_t* piece_bytes)
{
    if (n_angles < 1 || n_cells < 1 || n_devices < 1 || n_frames < 1) return -3;
    bool equal = true; int b0 = 0, e0 = 0;
    rr_partition(n_angles, n_devices, 0, &b0, &e0);
    for (int r = 0; r < n_devices; r++) {
        int b = 0, e = 0; rr_partition(n_angles, n_devices, r, &b, &e);
        equal = equal && (e - b) == (e0 - b0);
        for (int f = 0; f < n_frames; f++) {
            const size_t k = (size_t)r * n_frames + f;
            if (send_off) send_off[k] = (size_t)f * (size_t)(e - b) * n_cells;
            if (recv_off) recv_off[k] = ((size_t)f * n_angles + (size_t)b) * n_cells;
            if (piece_bytes) piece_bytes[k] = (size_t)(e - b) * n_cells;
        }
    }
    if (equal_blocks) *equal_blocks = equal ? 1 : 0;
    if (bytes_per_device) *bytes_per_device = (size_t)n_frames * (size_t)(e0 - b0) * n_cells;
    return 0;
}

// ---- frames ------------------------------------------------------------------------------------------------
namespace {

// after an error: nothing of this object may still be in flight when the caller gets the code back (a late D2H copy
// into a buffer the caller frees on error; sticky error bits that would fail the next, healthy call) -- every device is
// drained, its error bits are read and cleared, every slot is free again.  Returns the first error a device reports.
// An error invalidates EVERY batch in flight (advisor, round 4): the drain reads and clears the sticky bits of all frame
// lanes, so a second overflowing batch could no longer be told from a healthy one -- the other pending slots are marked
// failed with `code` and report it from their own rr_multi_wait / the next use of their slot.
int drain_all(rr_multi* m, std::string* first_msg, const MultiSlot* culprit = nullptr, int code = 0)
{
    int first = 0;
    for (size_t i = 0; i < m->ctx.size(); i++) { (void)hipSetDevice(m->devices[i]); (void)hipDeviceSynchronize(); }
    for (size_t i = 0; i < m->ctx.size(); i++) {
        const int rc = rr_synchronize(m->ctx[i], nullptr);
        if (rc && !first) { first = rc; if (first_msg) *first_msg = std::string("device ") + std::to_string(m->devices[i]) + ": " + rr_last_error(m->ctx[i]); }
    }
    if (!code) code = first ? first : -7;
    for (MultiSlot& S : m->slots) {
        if (S.pending && &S != culprit) S.failed = code;        // keeps its dst: the wait for that buffer reports it
        else if (!S.failed) S.dst = nullptr;
        S.pending = false;
        for (size_t i = 0; i < m->ctx.size(); i++) S.h_bits[i] = 0;
    }
    return first;
}

// a launch-time failure: keep ITS message, but hand the object back drained
int fail_drained(rr_multi* m, int code, const std::string& msg)
{
    (void)drain_all(m, nullptr, nullptr, code);
    return mfail(m, code, msg);
}

int wait_slot(rr_multi* m, MultiSlot& S)
{
    if (S.failed) {
        const int rc = S.failed;
        S.failed = 0; S.dst = nullptr;
        return mfail(m, rc, "this batch was in flight when another batch's error drained the object: its images are invalid (an error invalidates every batch in flight)");
    }
    if (!S.pending) return 0;
    const int n = (int)m->ctx.size();
    hipError_t e = hipSetDevice(m->devices[0]);
    if (e == hipSuccess) {
        // one device: the image may still sit on its frame lane or be on its way over SDMA (rr_simulate_batch_host_async);
        // several: the root's copy is an rr_deliver_to_host_async job -- either way the context's fence completes it
        if (rr_wait_host(m->ctx[0], S.dst)) return fail_drained(m, -100, std::string("device ") + std::to_string(m->devices[0]) + ": " + rr_last_error(m->ctx[0]));
    }
    if (e == hipSuccess) e = hipEventSynchronize(S.ev_done);
    if (e != hipSuccess) return fail_drained(m, -100, std::string("rr_multi_wait: ") + hipGetErrorString(e));
    S.pending = false; S.dst = nullptr;
    // the root's stream is ordered behind every device's block (collective / events), each block behind its error bits
    uint32_t bits = 0;
    for (int i = 0; i < n; i++) bits |= S.h_bits[i];
    if (bits) {
        std::string msg;
        const int code = (bits & 1u) ? -7 : -8;
        const int rc = drain_all(m, &msg, &S, code);
        return mfail(m, rc ? rc : code, rc ? msg : "a device reported an overflow / bad id");
    }
    return 0;
}

}  // namespace

int rr_multi_wait(rr_multi* m, const void* h_imgs_u8)
{
    if (!m) return -1;
    // oldest first, so that an error is reported for the batch that caused it
    int first = 0;
    for (size_t k = 0; k < m->slots.size(); k++) {
        MultiSlot& S = m->slots[(m->next_slot + k) % m->slots.size()];
        if ((!S.pending && !S.failed) || (h_imgs_u8 && S.dst != h_imgs_u8)) continue;
        const int rc = wait_slot(m, S);
        if (rc) { if (!first) first = rc; break; }       // (a failed wait drained everything)
    }
    // "every outstanding batch": the error is reported once for all of them
    if (!h_imgs_u8 && first) for (MultiSlot& S : m->slots) if (S.failed) { S.failed = 0; S.dst = nullptr; }
    return first;
}

int rr_multi_simulate_batch_async(rr_multi* m, const float* poses, int n_frames, uint8_t* out_imgs_u8)
{
    if (!m) return -1;
    if (!m->have_cfg) return mfail(m, -2, "rr_multi_set_config has not been called");
    if (!poses || !out_imgs_u8) return mfail(m, -3, "rr_multi_simulate_batch: null poses/output");
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return mfail(m, -3, "rr_multi_simulate_batch: n_frames must be 1..64");
    const int n = (int)m->ctx.size();
    const int A = m->cfg.n_angles; const size_t C = (size_t)m->cfg.n_cells;
    MultiSlot& S = m->slots[m->next_slot];
    m->next_slot = (m->next_slot + 1) % m->slots.size();
    rr::HostProfScope hp_all(8, "multi: whole call");
    { rr::HostProfScope hp(9, "multi: wait for the slot"); const int rc = wait_slot(m, S); if (rc) return rc; }        // the batch that used this slot's buffers last
    const auto dev_msg = [&](int i) { return std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rr_last_error(m->ctx[(size_t)i]); };
    if (n == 1 && !m->self_rccl) {
        // one device: no collective; the images take the ctx's own host delivery (deferred, trickled out by the next
        // batch's trace launches: within 1 % of leaving them in HBM)
        RRM_HIP(m, hipSetDevice(m->devices[0]));
        int rc = rr_simulate_batch_host_async(m->ctx[0], poses, n_frames, out_imgs_u8, S.streams[0]);
        if (rc) return fail_drained(m, rc, dev_msg(0));
        rc = rr_peek_error_bits_async(m->ctx[0], &S.h_bits[0], S.streams[0]);
        if (rc) return fail_drained(m, rc, dev_msg(0));
        hipError_t e = hipEventRecord(S.ev_done, S.streams[0]);
        if (e != hipSuccess) return fail_drained(m, -100, std::string("hipEventRecord: ") + hipGetErrorString(e));
        S.pending = true; S.dst = out_imgs_u8;
        return 0;
    }
#define RRM_TRY_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail_drained(m, -100, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define RRM_TRY_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != 0) return fail_drained(m, -101, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)
    std::vector<int> b((size_t)n), e((size_t)n);
    bool equal = true;
    for (int i = 0; i < n; i++) { rr_partition(A, n, i, &b[(size_t)i], &e[(size_t)i]); equal = equal && (e[(size_t)i] - b[(size_t)i]) == (e[0] - b[0]); }
    if (m->self_rccl_frames) equal = false;      // (test switch: one send / recv pair per frame, as ragged blocks travel)
    // 1. every device renders its block of all frames (one set of launches each, all devices concurrently)
    //    -- each device's launches from its own enqueue thread where there are several devices (MultiWorker)
    std::vector<std::string> hip_err((size_t)n);
    auto render = [&](int i) -> int {          // rc of the context, or -100 with hip_err[i] set
        const size_t nl = (size_t)(e[(size_t)i] - b[(size_t)i]);
        hipError_t he = hipSetDevice(m->devices[(size_t)i]);
        if (he == hipSuccess) he = S.block[(size_t)i].ensure(std::max<size_t>(1, (size_t)n_frames * nl * C));
        S.h_bits[i] = 0;
        int rc = 0;
        if (he == hipSuccess && nl > 0) {
            rr_ctx* c = m->ctx[(size_t)i];
            { rr::HostProfScope hp(10, "multi: device entry: render"); rc = rr_simulate_batch_columns_device(c, poses, n_frames, b[(size_t)i], e[(size_t)i], S.block[(size_t)i].p, S.streams[(size_t)i]); }
            { rr::HostProfScope hp(11, "multi: device entry: error bits"); if (!rc) rc = rr_peek_error_bits_async(c, &S.h_bits[i], S.streams[(size_t)i]); }
            if (rc) return rc;
        }
        { rr::HostProfScope hp(12, "multi: device entry: block event"); if (he == hipSuccess) he = hipEventRecord(S.ev_block[(size_t)i], S.streams[(size_t)i]); }
        if (he != hipSuccess) { hip_err[(size_t)i] = hipGetErrorString(he); return -100; }
        return 0;
    };
    std::vector<int> rcs((size_t)n, 0);
    if (!m->workers.empty()) {
        for (int i = 0; i < n; i++) {
            MultiWorker* w = m->workers[(size_t)i].get();
            { std::lock_guard<std::mutex> lk(w->mu); w->job = [&render, i] { return render(i); }; w->done = false; w->has_job = true; }
            w->cv.notify_all();
        }
        for (int i = 0; i < n; i++) {
            MultiWorker* w = m->workers[(size_t)i].get();
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return w->done; });
            rcs[(size_t)i] = w->rc;
        }
    } else {
        for (int i = 0; i < n && (i == 0 || rcs[(size_t)i - 1] == 0); i++) rcs[(size_t)i] = render(i);
    }
    for (int i = 0; i < n; i++)
        if (rcs[(size_t)i]) return fail_drained(m, rcs[(size_t)i], hip_err[(size_t)i].empty() ? dev_msg(i) : std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + hip_err[(size_t)i]);
    // 2. ONE collective: a GATHER to the root (device 0), nobody else receives anything.  RCCL has no plain gather in
    //    every version, so it is one group of send / recv pairs along rr_multi_plan: equal blocks travel as one piece
    //    per device into the layout [device][frame][n_loc][n_cells], ragged ones frame by frame into
    //    [frame][n_angles][n_cells]; the root's own block is a device-to-device copy on its stream
    rr::HostProfScope hp_root(13, "multi: gather + assemble + D2H");
    RRM_TRY_HIP(hipSetDevice(m->devices[0]));
    const uint8_t* d_cols = nullptr; int n_loc = A; size_t block_stride = (size_t)A * C, frame_stride = (size_t)A * C;
    struct Piece { int dev; size_t so, ro, bytes; };
    std::vector<Piece> pieces;
    if (equal) {
        const size_t nl = (size_t)(e[0] - b[0]), per = (size_t)n_frames * nl * C;
        RRM_TRY_HIP(S.gathered.ensure((size_t)n * per));
        for (int i = 0; i < n; i++) pieces.push_back({ i, 0, (size_t)i * per, per });
        n_loc = (int)nl; block_stride = per; frame_stride = nl * C;
    } else {
        RRM_TRY_HIP(S.gathered.ensure((size_t)n_frames * A * C));
        std::vector<size_t> so((size_t)n * n_frames), ro((size_t)n * n_frames), pb((size_t)n * n_frames);
        (void)rr_multi_plan(A, (int)C, n, n_frames, nullptr, nullptr, so.data(), ro.data(), pb.data());
        for (int i = 0; i < n; i++) for (int f = 0; f < n_frames; f++) {
            const size_t k = (size_t)i * n_frames + f;
            if (pb[k]) pieces.push_back({ i, so[k], ro[k], pb[k] });
        }
    }
    d_cols = S.gathered.p;
    for (const Piece& p : pieces)          // the root's own pieces (and, in loopback, everybody's): plain copies on the root's stream
        if ((p.dev == 0 && !m->self_rccl) || m->loopback) {
            if (p.dev != 0) RRM_TRY_HIP(hipStreamWaitEvent(S.streams[0], S.ev_block[(size_t)p.dev], 0));
            RRM_TRY_HIP(hipMemcpyAsync(S.gathered.p + p.ro, S.block[(size_t)p.dev].p + p.so, p.bytes, hipMemcpyDeviceToDevice, S.streams[0]));
        }
    if (!m->loopback) {
        RRM_TRY_NCCL(g_rccl.GroupStart());
        ncclResult_t gr = 0; hipError_t ge = hipSuccess;          // a failure inside the group still has to close the group
        for (const Piece& p : pieces) {
            if ((p.dev == 0 && !m->self_rccl) || gr != 0 || ge != hipSuccess) continue;
            // (one thread drives every device: the current device follows the communicator a call is made on)
            ge = hipSetDevice(m->devices[(size_t)p.dev]);
            if (ge == hipSuccess) gr = g_rccl.Send(S.block[(size_t)p.dev].p + p.so, p.bytes, kNcclUint8, 0, m->comms[(size_t)p.dev], S.streams[(size_t)p.dev]);
            if (ge == hipSuccess && gr == 0) ge = hipSetDevice(m->devices[0]);
            if (ge == hipSuccess && gr == 0) gr = g_rccl.Recv(S.gathered.p + p.ro, p.bytes, kNcclUint8, p.dev, m->comms[0], S.streams[0]);
        }
        const ncclResult_t ger = g_rccl.GroupEnd();
        if (ge != hipSuccess) return fail_drained(m, -100, std::string("hipSetDevice (collective): ") + hipGetErrorString(ge));
        if (gr != 0 || ger != 0) return fail_drained(m, -101, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(gr != 0 ? gr : ger));
    }
    // 3. root: transpose into mono8 images, copy to the caller's host buffer
    RRM_TRY_HIP(hipSetDevice(m->devices[0]));
    const size_t bytes = (size_t)n_frames * C * A;
    RRM_TRY_HIP(S.d_imgs.ensure(bytes));
    { const int rc = rr_assemble_frames_device(m->ctx[0], d_cols, n_loc, block_stride, n_frames, frame_stride, S.d_imgs.p, S.streams[0]);
      if (rc) return fail_drained(m, rc, std::string("root: ") + rr_last_error(m->ctx[0])); }
    // (over the SDMA engines, whichever HIP runtime serves the process: rr_deliver_to_host_async; fenced in wait_slot)
    { const int rc = rr_deliver_to_host_async(m->ctx[0], S.d_imgs.p, out_imgs_u8, bytes, S.streams[0]);
      if (rc) return fail_drained(m, rc, std::string("root: ") + rr_last_error(m->ctx[0])); }
    RRM_TRY_HIP(hipEventRecord(S.ev_done, S.streams[0]));
#undef RRM_TRY_HIP
#undef RRM_TRY_NCCL
    S.pending = true; S.dst = out_imgs_u8;
    return 0;
}

int rr_multi_simulate_batch(rr_multi* m, const float* poses, int n_frames, uint8_t* out_imgs_u8)
{
    const int rc = rr_multi_simulate_batch_async(m, poses, n_frames, out_imgs_u8);
    if (rc) return rc;
    return rr_multi_wait(m, out_imgs_u8);
}

int rr_multi_simulate(rr_multi* m, const float pose_qxyzw_t[7], uint8_t* out_u8)
{
    return rr_multi_simulate_batch(m, pose_qxyzw_t, 1, out_u8);
}

}  // extern "C"
