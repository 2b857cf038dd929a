// rr_multi.hip -- multi-GPU fan-out BEHIND the C ABI (SURVEY.md §8b "Threading": one backend object per process,
// radar_simulator.cpp:145-176; §8e: azimuth columns are independent, RadarCPU.cpp:155).
//
// One process, one rr_ctx per device, mesh / parameters replicated, device i renders the contiguous azimuth block
// rr_partition(n_angles, n, i) of every frame of a call in ONE set of launches on its own stream; ONE RCCL
// collective per call assembles the frames on device 0:
//     equal blocks  : ncclAllGather of [n_frames][n_loc][n_cells] per device (the single gather of north_star)
//     ragged blocks : one ncclGroup of send/recv pairs (block of frame f, device r -> its place in frame f on the root)
// then the root transposes into mono8 images and copies them to the caller's host buffer.
// RCCL is loaded at run time (librccl.so.1): the library itself has no link-time dependency on it and
// rr_create_multi() fails with a clear message where it is missing.  Built on the public entry points of
// radarays_mi355.h only.
#include "../../include/radarays_mi355.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
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
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* n : { "librccl.so.1", "librccl.so" }) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = "rr_create_multi: librccl.so.1 not found (RCCL is needed for more than one device)"; return false; }
#define RR_SYM(f) f = (decltype(f))dlsym(lib, "nccl" #f); if (!f) { err = "rr_create_multi: librccl lacks nccl" #f; return false; }
        RR_SYM(CommInitAll) RR_SYM(CommDestroy) RR_SYM(AllGather) RR_SYM(Send) RR_SYM(Recv) RR_SYM(GroupStart) RR_SYM(GroupEnd) RR_SYM(GetErrorString)
#undef RR_SYM
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

struct rr_multi {
    std::vector<int> devices;
    std::vector<rr_ctx*> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<Buf<uint8_t>> block;      // per device: [n_frames][n_loc_i][n_cells]
    std::vector<Buf<uint8_t>> gathered;   // per device (all-gather) / root only (send/recv)
    Buf<uint8_t> d_imgs;                  // root: [n_frames][n_cells][n_angles]
    rr_config cfg;
    bool have_cfg = false;
    bool loopback = false;                // see rr_create_multi
    std::string err;
};

namespace {

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
        hipStream_t s = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            g_multi_create_error = "rr_create_multi: stream creation failed"; rr_destroy_multi(m); return nullptr;
        }
        m->streams.push_back(s);
    }
    m->block.resize((size_t)n_devices); m->gathered.resize((size_t)n_devices);
    if (n_devices > 1 && !loopback) {
        // the communicator is owned here: one rank per device of this process
        if (!g_rccl.load(g_multi_create_error)) { rr_destroy_multi(m); return nullptr; }
        m->comms.resize((size_t)n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n_devices, devices);
        if (r != 0) { g_multi_create_error = std::string("rr_create_multi: ncclCommInitAll: ") + g_rccl.GetErrorString(r); m->comms.clear(); rr_destroy_multi(m); return nullptr; }
    }
    return m;
}

void rr_destroy_multi(rr_multi* m)
{
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        (void)hipDeviceSynchronize();
        if (i < m->comms.size() && m->comms[i]) g_rccl.CommDestroy(m->comms[i]);
        if (i < m->block.size()) m->block[i].release();
        if (i < m->gathered.size()) m->gathered[i].release();
        if (i == 0) m->d_imgs.release();
        if (i < m->streams.size() && m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
        rr_destroy(m->ctx[i]);
    }
    delete m;
}

const char* rr_multi_last_error(const rr_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
int rr_multi_device_count(const rr_multi* m) { return m ? (int)m->ctx.size() : 0; }
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
// equal blocks : every device all-gathers `bytes_per_device`; device r's block lands at r * bytes_per_device of the
//                gathered buffer, laid out [device][frame][n_loc][n_cells]
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
int rr_multi_simulate_batch(rr_multi* m, const float* poses, int n_frames, uint8_t* out_imgs_u8)
{
    if (!m) return -1;
    if (!m->have_cfg) return mfail(m, -2, "rr_multi_set_config has not been called");
    if (!poses || !out_imgs_u8) return mfail(m, -3, "rr_multi_simulate_batch: null poses/output");
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return mfail(m, -3, "rr_multi_simulate_batch: n_frames must be 1..64");
    const int n = (int)m->ctx.size();
    const int A = m->cfg.n_angles; const size_t C = (size_t)m->cfg.n_cells;
    std::vector<int> b((size_t)n), e((size_t)n);
    bool equal = true;
    for (int i = 0; i < n; i++) { rr_partition(A, n, i, &b[(size_t)i], &e[(size_t)i]); equal = equal && (e[(size_t)i] - b[(size_t)i]) == (e[0] - b[0]); }
    // 1. every device renders its block of all frames (one set of launches each, all devices concurrently)
    for (int i = 0; i < n; i++) {
        const size_t nl = (size_t)(e[(size_t)i] - b[(size_t)i]);
        RRM_HIP(m, hipSetDevice(m->devices[(size_t)i]));
        RRM_HIP(m, m->block[(size_t)i].ensure(std::max<size_t>(1, (size_t)n_frames * nl * C)));
        if (nl == 0) continue;
        rr_ctx* c = m->ctx[(size_t)i];
        const int rc = rr_simulate_batch_columns_device(c, poses, n_frames, b[(size_t)i], e[(size_t)i], m->block[(size_t)i].p, m->streams[(size_t)i]);
        if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rr_last_error(c));
    }
    // 2. ONE collective: the blocks meet on the root
    const uint8_t* d_cols = nullptr; int n_loc = A; size_t block_stride = (size_t)A * C, frame_stride = (size_t)A * C;
    if (n == 1) {
        d_cols = m->block[0].p;                            // [n_frames][A][C]
    } else if (equal) {
        const size_t nl = (size_t)(e[0] - b[0]), per = (size_t)n_frames * nl * C;
        for (int i = 0; i < n; i++) { RRM_HIP(m, hipSetDevice(m->devices[(size_t)i])); RRM_HIP(m, m->gathered[(size_t)i].ensure((size_t)n * per)); }
        if (m->loopback) {
            // what the all-gather leaves on the root: block r at r * per
            for (int i = 0; i < n; i++) RRM_HIP(m, hipStreamSynchronize(m->streams[(size_t)i]));
            for (int i = 0; i < n; i++)
                RRM_HIP(m, hipMemcpyAsync(m->gathered[0].p + (size_t)i * per, m->block[(size_t)i].p, per, hipMemcpyDeviceToDevice, m->streams[0]));
        } else {
        RRM_NCCL(m, g_rccl.GroupStart());
        for (int i = 0; i < n; i++)
            RRM_NCCL(m, g_rccl.AllGather(m->block[(size_t)i].p, m->gathered[(size_t)i].p, per, kNcclUint8, m->comms[(size_t)i], m->streams[(size_t)i]));
        RRM_NCCL(m, g_rccl.GroupEnd());
        }
        d_cols = m->gathered[0].p;                         // [device][n_frames][nl][C]
        n_loc = (int)nl; block_stride = per; frame_stride = nl * C;
    } else {
        RRM_HIP(m, hipSetDevice(m->devices[0]));
        RRM_HIP(m, m->gathered[0].ensure((size_t)n_frames * A * C));
        std::vector<size_t> so((size_t)n * n_frames), ro((size_t)n * n_frames), pb((size_t)n * n_frames);
        (void)rr_multi_plan(A, (int)C, n, n_frames, nullptr, nullptr, so.data(), ro.data(), pb.data());
        if (m->loopback) {
            for (int i = 0; i < n; i++) RRM_HIP(m, hipStreamSynchronize(m->streams[(size_t)i]));
            for (int i = 0; i < n; i++)
                for (int f = 0; f < n_frames; f++) {
                    const size_t k = (size_t)i * n_frames + f;
                    if (pb[k] == 0) continue;
                    RRM_HIP(m, hipMemcpyAsync(m->gathered[0].p + ro[k], m->block[(size_t)i].p + so[k], pb[k], hipMemcpyDeviceToDevice, m->streams[0]));
                }
        } else {
        RRM_NCCL(m, g_rccl.GroupStart());
        for (int i = 0; i < n; i++)
            for (int f = 0; f < n_frames; f++) {
                const size_t k = (size_t)i * n_frames + f;
                if (pb[k] == 0) continue;
                RRM_NCCL(m, g_rccl.Send(m->block[(size_t)i].p + so[k], pb[k], kNcclUint8, 0, m->comms[(size_t)i], m->streams[(size_t)i]));
                RRM_NCCL(m, g_rccl.Recv(m->gathered[0].p + ro[k], pb[k], kNcclUint8, i, m->comms[0], m->streams[0]));
            }
        RRM_NCCL(m, g_rccl.GroupEnd());
        }
        d_cols = m->gathered[0].p;                         // [n_frames][A][C]
    }
    // 3. root: transpose into mono8 images, copy to the caller's host buffer
    RRM_HIP(m, hipSetDevice(m->devices[0]));
    const size_t bytes = (size_t)n_frames * C * A;
    RRM_HIP(m, m->d_imgs.ensure(bytes));
    int rc = rr_assemble_frames_device(m->ctx[0], d_cols, n_loc, block_stride, n_frames, frame_stride, m->d_imgs.p, m->streams[0]);
    if (rc) return mfail(m, rc, std::string("root: ") + rr_last_error(m->ctx[0]));
    RRM_HIP(m, hipMemcpyAsync(out_imgs_u8, m->d_imgs.p, bytes, hipMemcpyDeviceToHost, m->streams[0]));
    // 4. drain; per-device error bits (queue overflow / bad ids) surface here
    for (int i = n - 1; i >= 0; i--) {
        rr_ctx* c = m->ctx[(size_t)i];
        rc = rr_synchronize(c, m->streams[(size_t)i]);
        if (rc) return mfail(m, rc, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rr_last_error(c));
    }
    return 0;
}

int rr_multi_simulate(rr_multi* m, const float pose_qxyzw_t[7], uint8_t* out_u8)
{
    return rr_multi_simulate_batch(m, pose_qxyzw_t, 1, out_u8);
}

}  // extern "C"
