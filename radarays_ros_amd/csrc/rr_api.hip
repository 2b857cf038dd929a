// rr_api.hip -- C ABI of libradarays_mi355.so (include/radarays_mi355.h):
// context, buffers, frame orchestration.  No CPU fallback: every compute call
// runs the gfx950 kernels of rr_kernels.hip or fails with an error string.
#include "../../include/radarays_mi355.h"
#include "rr_device.h"
#include "rr_hostprof.h"
#include "rr_sdma.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>

namespace rr {
void launch_trace(const Params& P, int pass, const PoseArgs* poses, bool stats, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                  hipEvent_t ev_rep_start = nullptr, hipEvent_t ev_rep_stop = nullptr, bool* repair_launched = nullptr);
void launch_shade(const Params& P, int pass, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_scan(const Params& P, int pass, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_column(const Params& P, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_decay_table(float* decay, int n_cells, double resolution, double energy_loss, hipStream_t s);
void launch_assemble_u8(const uint8_t* cols, uint8_t* img, int n_angles, int n_cells, int scroll, hipStream_t s,
                        int n_loc = 0, size_t block_stride = 0, int n_frames = 1, size_t frame_stride = 0);
void launch_assemble_f32(const float* cols, float* img, int n_angles, int n_cells, int scroll, hipStream_t s);
bool build_bvh4_gpu(const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object,
                    Node4** d_nodes_out, size_t* n_nodes_out, TriRec** d_tris_out, size_t* n_tris_out,
                    uint32_t* depth_out, uint32_t* stack_need_out, float* inflate_out,
                    std::string& err, hipStream_t stream);
void launch_debug_trace(const Params& P, const float* origs, const float* dirs, int n,
                        float* out_t, uint32_t* out_face, hipStream_t s, unsigned long long* steps = nullptr);
void launch_encode_refs(Node4* nodes, size_t n_nodes, uint32_t tri_base4, hipStream_t s, size_t n_tris);
void launch_mat_limits(const float4* materials, size_t n, double* limits, hipStream_t s);
void* trace0_kernel(bool spill, bool stackless);
Params trace0_params(const Params& P);
void launch_score(const uint8_t* imgs, const uint8_t* ref, size_t npx, int n_images, unsigned long long* sse, hipStream_t s);
void launch_copy_host(const void* src, void* dst, size_t bytes, int blocks, int inflight, int xcd, hipStream_t s, int threads);
void launch_copy_words(const void* src, void* dst, size_t bytes, hipStream_t s);
void launch_debug_brdf(size_t n, const float* in, int model, float* out, hipStream_t s);
void launch_store_u32(const uint32_t* src, uint32_t* h_dst, hipStream_t s);
void launch_debug_fresnel(size_t n, const float* normals, const float* dirs, const double* energy, const double* v1, const float* v2,
                          float* out_rdir, double* out_re, float* out_tdir, double* out_te, hipStream_t s);
}  // namespace rr

using namespace rr;

namespace {

std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    uint64_t launches = 0;
    std::vector<float> samples_ms;   // every launch since the last reset (median / percentiles)
};

}  // namespace

struct Lane {
    int buf_seg = 0, buf_cap = 0, buf_sigcap = 0, buf_cells = 0, buf_passes = 0;
    DevBuf<float4> d_wA[2], d_wB[2];
    DevBuf<double2> d_wC[2];
    DevBuf<uint32_t> d_idx[2], d_count[2], d_refpos, d_sig_count, d_spill;
    DevBuf<uint2> d_torder[2];
    DevBuf<uint2> d_hit;
    DevBuf<uint8_t> d_cflag, d_cols_u8;
    DevBuf<SigRec> d_sigtmp, d_sig;
    DevBuf<float> d_cols_f32;
    DevBuf<Counters> d_counters;
    DevBuf<uint32_t> d_sticky;    // error bits of ALL frames since the last rr_synchronize (async entry points); the synchronous entry points clear them when they report an error themselves
    DevBuf<uint8_t> d_img_u8;     // host-buffer path: assembled image before the D2H copy
    DevBuf<float> d_img_f32;
    DevBuf<SegStats> d_seg_stats;
    DevBuf<float4> d_matsets;     // material sets of a parameter batch [n_sets][n_materials]
    DevBuf<double> d_matset_limits;   // ... and their angles of total reflection (k_mat_limits)
    // ... and its beam tables [n_groups][n_beam] with their two trace orders; the host arrays they are copied from stay
    // alive with the lane (a copy from pageable memory may still be staged when the call returns)
    DevBuf<float4> d_set_beams; DevBuf<uint32_t> d_set_order, d_set_order2;
    std::vector<float4> h_set_beams, h_matsets; std::vector<uint32_t> h_set_order, h_set_order2;
    int last_n_seg = 0, last_n_passes = 0;
    int spill_stride = 0, stack_lds = 1;
    // tight later-pass trace grids (rr_device.h: GridHint): the lane's history / overflow counters, the overflow lists,
    // and the page-locked copy of the history that arrives behind every batch (read without a fence: it is a hint)
    DevBuf<GridHint> d_hint; DevBuf<uint32_t> d_ovf_list; int ovf_stride = 0;
    uint32_t* h_hist = nullptr; int hist_gen = 0;
    // Launch graphs (round 5): the launch chain of a batch -- n_reflections x {trace [+ repair], shade, scan}, column, history
    // copy -- captured once per (azimuth block, frames, output buffer, trace rows) and replayed with ONE hipGraphLaunch; the
    // poses are the only thing that changes between replays (the third argument of the pass-0 trace node).  Host time per
    // chain: 46 us launched kernel by kernel (16 launches) against ~11 us replayed (tools/cpp_bench.cpp graph)
    struct FrameGraph {
        int az_begin = 0, az_end = 0, n_frames = 0; const void* cols = nullptr; unsigned short rows[kMaxPasses] = {};
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr; hipGraphNode_t pose_node = nullptr; uint64_t last_use = 0; int hits = 0;
        // Replays must not touch a launch that is still queued or running: whether hipGraphExecKernelNodeSetParams rewrites the
        // kernel arguments of an exec IN PLACE is the runtime's business (advisor, round 5: lane reuse is ordered on the device
        // only, the host never waits), so the library does not depend on it -- TWO execs per shape, used alternately, each with
        // an event behind its last launch; the host waits for that event before it re-sets the exec's poses or destroys it.
        // The exec about to be updated was launched two uses of this shape ago: the wait is over before it starts, except for a
        // caller that runs more than a whole lane rotation ahead of the GPU
        hipGraphExec_t ge2 = nullptr; hipEvent_t ev[2] = { nullptr, nullptr }; bool ev_pending[2] = { false, false }; int flip = 0;
        hipKernelNodeParams pose_kp{};     // the pass-0 trace node as captured (grid, block, LDS) ...
        Params pose_P;                     // ... and the Params bytes it was captured with
    };
    std::vector<FrameGraph> graphs; int graph_gen = 0;
    DevBuf<float> d_poses;       // [RR_MAX_BATCH][8]: Params::pose_table, written by the pass-0 trace launch of every chain
    unsigned short last_rows[kMaxPasses] = {};     // rows the lane's last batch was launched with (0: the bound)

    hipStream_t stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_consumed = nullptr;
    bool pending_consume = false;
    // host delivery (rr_simulate_batch_host_async): the copies of this lane that may still be in flight.  Two records:
    // a lane's copies complete in order, and a record is reused only after its copy has completed (take_rec waits), so a
    // buffer that is in no record any more has been delivered.
    struct CopyRec { const void* dst = nullptr; hipEvent_t ev = nullptr; bool pending = false; } rec[2];
    int rec_next = 0;
    // a batch's images stay in d_img_u8 ("deferred") until the lane's NEXT host-delivery batch, whose later-pass trace
    // launches carry the copy (see Params::copy_src); rr_wait_host / rr_synchronize / any other use of the lane flush a
    // deferred copy with a plain hipMemcpyAsync
    // ... or, the default: the batch's images leave at once over SDMA (rr_sdma.cpp).  TWO image buffers per lane, used alternately
    // (d_img_u8 and d_img_u8_b), each with its event (behind the assemble that filled it) and the job that empties it: a buffer
    // is written again two uses of the lane later (eight batches with four lanes), by which time its copy has long left -- the
    // host checks the job before it reuses the buffer and practically never has to wait (with ONE buffer it waited for the lane's
    // previous batch every time: the lane's stream ran dry while the host issued the next chain -- 35.9k instead of 39.4k images/s
    // on config 2 from a C++ caller, 2.5k instead of 4.3k with one pose per batch on the target)
    DevBuf<uint8_t> d_img_u8_b; hipEvent_t ev_img[2] = { nullptr, nullptr }; int img_flip = 0;
    uint64_t sdma_job[2] = { 0, 0 }; const void* sdma_dst[2] = { nullptr, nullptr };
    bool deferred = false;
    uint8_t* def_dst = nullptr; size_t def_bytes = 0; hipStream_t def_stream = nullptr; bool def_foldable = false;
};

struct rr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    bool have_mesh = false;
    // ONE allocation: the BVH4 nodes, then the leaf-order triangles -- child references are float4 offsets
    // from its base (rr_bvh.h), so a traversal step forms its address the same way for a node and a leaf
    DevBuf<float4> d_bvh;
    uint32_t tri_base4 = 0;        // float4 offset of triangle 0
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t depth = 0, stack_need = 0;
    float hit_pad = 0.f;           // grazing guard of the triangle test (traverse): 1e-5 x the extent of the faces' vertices = half the builders' box padding

    // params
    rr_config cfg;
    bool have_cfg = false;
    std::vector<rr_material> materials;
    std::vector<int32_t> object_materials;
    int32_t material_id_air = 0;
    bool have_materials = false;
    std::vector<float> beams;   // xyz
    std::vector<float> noise;
    int noise_rows = 1;
    int motion_rows = 1;
    bool motion_live = false;    // a motion table was in use at the last upload (Params::motion_poses non-null)
    std::vector<float> motion;   // [n_angles][7] or empty
    std::vector<float> smear;
    int smear_mode = 0;

    DevBuf<float4> d_qas, d_beams, d_materials;
    DevBuf<double> d_mat_limits;   // [n_materials]: angle of total reflection per material
    double limit_same = 0.0;       // ... and for v2 = 0.3f (the same material on both sides): computed once, at rr_create
    DevBuf<uint32_t> d_beam_order, d_beam_order2;
    DevBuf<int32_t> d_objmat;
    DevBuf<float> d_smear, d_noise, d_motion, d_decay;
    DevBuf<uint8_t> d_param_imgs;   // rr_simulate_material_sets: images before the D2H copy
    // what upload_tables() has to refresh (the reference node re-reads its parameters before EVERY
    // frame, radar_simulator.cpp:85,200: setters that bring nothing new must cost nothing)
    enum : unsigned { D_CFG = 1, D_BEAMS = 2, D_MAT = 4, D_NOISE = 8, D_MOTION = 16, D_ALL = 31 };
    unsigned tables_dirty = D_ALL;

    // frame lanes: each owns a full set of frame buffers + a stream, so consecutive
    // frames overlap on the GPU (the tail of one frame's k_trace runs beside the next frame)
    std::vector<Lane> lanes;
    size_t next_lane = 0, last_lane = 0;
    size_t next_stream_lane = 0;
    int stream_lanes = 3;          // lanes whose own stream rr_simulate_device uses

    bool stats_mode = false;
    int pass0_az = 16;
    int stack_lds_max = 64;      // traversal stack entries kept in LDS (RR_STACK_LDS lowers it: tests of the spill path)
    int timing = 0;   // 0 off, 1 every kernel, 2 k_trace only
    std::map<std::string, KernelTimer> timers;
    // timing events are pooled: created once, handed out in the frame path, returned when rr_get_kernel_time
    // reads them (no hipEventCreate / hipEventDestroy between the synchronisation points of a timed region)
    std::vector<hipEvent_t> event_pool;
    hipEvent_t take_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
    }

    int passes_override = -1;    // a parameter batch in the making: the largest n_reflections of its sets sizes queues and launch loops
    DevBuf<unsigned long long> d_sse; DevBuf<uint8_t> d_ref_img;     // rr_score_images / rr_simulate_param_sets
    void* h_rb = nullptr; size_t h_rb_bytes = 0;         // page-locked: read_back()
    void* h_frame = nullptr; size_t h_frame_bytes = 0;   // page-locked: error bits + per-pass counters of rr_simulate's frame

    bool roctx = false;
    int fold_min_busy = 2;       // other lanes that must have a batch in flight for a host copy to be folded (RR_FOLD_MIN_BUSY)
    int seg_chunk = 16;          // later-pass trace grids in chunks of S neighbouring segments, segment-fast inside a chunk (RR_TRACE_CHUNK; 0: rows of one segment)
    int stackless = 0;           // RR_STACKLESS=1: the stack-free traversal (no LDS; DESIGN.md §3 says what it costs)
    int cull_pop = 1;            // k_trace's later passes drop stack entries at pop time (RR_CULL_POP=0: off; the images are the same either way)
    // a deferred host copy that cannot ride on a later-pass trace launch (one-pass frames, the last batch of a run, a caller
    // with a single batch in flight) is stored by the library's own kernel (k_copy_host) when the destination is page-locked:
    // flush_blocks one-wave workgroups with at most flush_inflight 1-KB stores outstanding each (RR_FLUSH_BLOCKS, RR_FLUSH_INFLIGHT;
    // RR_FLUSH_KERNEL=0: hipMemcpyAsync, i.e. whichever engine the process' HIP runtime picks)
    int flush_kernel = 1, flush_blocks = 8, flush_inflight = 0;
    int flush_threads = 256;     // threads per workgroup of the copy kernel (RR_FLUSH_THREADS)
    int flush_xcd = 0;           // the copy kernel's workgroups all on this XCD (RR_FLUSH_XCD 0..7; -1: dealt out over all eight)
    // RR_HOST_COPY_STREAM=1 (experiment, round 6): a batch that cannot fold its predecessor's images into a trace launch (one-pass
    // frames) sends its OWN images at once on one dedicated copy stream -- copies then run one at a time, in order, beside the
    // batches instead of in front of the lane's next one
    int host_copy_stream = 0; hipStream_t copy_stream = nullptr;
    // RR_HOST_SDMA (1): rr_simulate_batch_host_async hands a batch's images to ROCr's SDMA path (rr_sdma.cpp: one worker thread,
    // copies in order, each behind its batch's last kernel) instead of a copy the HIP runtime would pick an engine for; 0, a
    // pageable destination or a runtime ROCr cannot be reached through: the deferred / trickled copies below
    int host_sdma = 1; SdmaCopier* sdma = nullptr; bool sdma_tried = false;
    // rr_deliver_to_host_async: copies of caller-owned device buffers that rr_wait_host fences (an SDMA job, or -- fallback -- an
    // event behind a stream-ordered copy); events are pooled
    struct Delivery { const void* dst; uint64_t job; hipEvent_t ev; };
    std::vector<Delivery> deliveries;
    std::vector<hipEvent_t> delivery_events;
    int copy_blocks = 8;         // workgroups (one wave each) of a later-pass trace launch that trickle a folded host copy (RR_COPY_BLOCKS; 0: never fold)
    int tight_grid = 1;          // later-pass trace rows sized by what earlier batches needed (RR_TIGHT_GRID=0: the doubling bound)
    int tight_force = 0;         // RR_TIGHT_FORCE=n: rows of n workgroups whatever the history says (tests of the repair path)
    int hist_gen = 1;            // bumped whenever mesh / materials / beam / config change: the lanes' histories start over
    int graph_guard = 1;         // RR_GRAPH_GUARD=0 (probe): replay ONE exec per shape without waiting for its previous launch, as round 5 did
    int use_graphs = 1;          // RR_GRAPHS=0: every launch chain is issued kernel by kernel
    int graph_gen = 1;           // bumped whenever anything a captured launch bakes in may have changed (tables, tree, lane buffers)
    uint64_t graph_clock = 0, graph_replays = 0, graph_captures = 0;
};

namespace {

// roctx ranges around the enqueue of trace / shade / scan / column / assemble (readable rocprofv3
// timelines with --marker-trace).  Optional: RR_ROCTX=1 loads librocprofiler-sdk-roctx / libroctx64 at run time.
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
bool roctx_load()
{
    static int state = 0;   // 0 untried, 1 ok, -1 missing
    if (state == 0) {
        state = -1;
        for (const char* n : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
            g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
            if (g_roctx_push && g_roctx_pop) { state = 1; break; }
        }
    }
    return state == 1;
}
inline void roctx_push(const char* name) { if (g_roctx_push) g_roctx_push(name); }
inline void roctx_pop() { if (g_roctx_pop) g_roctx_pop(); }

int fail(rr_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define RR_HIP(c, expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), -100, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// radar_algorithms.h:283-351 + RadarCPU.cpp:48-93 (host, same float/double mix)
void make_smear(const rr_config& cfg, std::vector<float>& w, int& mode)
{
    w.clear(); mode = 0;
    int width = 0; double mfrac = 0.0;
    if (cfg.signal_denoising == 1) { width = cfg.signal_denoising_triangular_width; mfrac = cfg.signal_denoising_triangular_mode; }
    else if (cfg.signal_denoising == 2) { width = cfg.signal_denoising_gaussian_width; mfrac = cfg.signal_denoising_gaussian_mode; }
    else if (cfg.signal_denoising == 3) { width = cfg.signal_denoising_mb_width; mfrac = cfg.signal_denoising_mb_mode; }
    if (width <= 0) return;
    mode = (int)(mfrac * width);
    w.resize((size_t)width);
    if (cfg.signal_denoising == 3) {
        const float fm = (float)mode;
        const float a = (float)((double)fm / M_SQRT2);
        for (int i = 0; i < width; i++) {
            const float x = (float)i;
            const float xx = x * x, aa = a * a, aaa = a * a * a;
            w[i] = (float)(std::sqrt(2.0 / M_PI) * (double)xx * (double)expf(-xx / (2 * aa)) / (double)aaa);
        }
    } else {
        for (int i = 0; i < width; i++) {
            float p;
            if (i <= mode) p = (float)i / (float)mode;
            else p = (float)(1.0 - (double)(((float)i - (float)mode) / ((float)width - (float)mode)));
            w[i] = (float)((double)(p * 1.0f) + (1.0 - (double)p) * (double)0.0f);
        }
    }
    float sum = 0.0f;
    for (int i = 0; i < width; i++) sum += w[i];
    for (int i = 0; i < width; i++) w[i] /= sum;
    const double mode_val = w[mode];
    for (int i = 0; i < width; i++) w[i] = (float)((double)w[i] / mode_val);
}

// number of ray-cast passes the frame buffers and the launch loop are sized for: the config's, or the largest of a
// parameter batch while rr_simulate_param_sets_device assembles it
inline int eff_passes(const rr_ctx* c) { return c->passes_override >= 0 ? c->passes_override : c->cfg.n_reflections; }
inline rr_config eff_config(const rr_ctx* c) { rr_config g = c->cfg; g.n_reflections = eff_passes(c); return g; }

int wave_capacity(const rr_config& cfg, int n_beam)
{
    long cap = cfg.max_waves_per_azimuth;
    if (cap <= 0) {
        cap = n_beam;
        for (int p = 1; p < cfg.n_reflections && cap < 65536; p++) cap *= 2;
        cap = std::min<long>(cap, 65536);
    }
    cap = std::max<long>(cap, n_beam);
    return (int)cap;
}

int signal_capacity(const rr_config& cfg, int n_beam, int cap)
{
    long tot = 0, w = n_beam;
    for (int p = 0; p < cfg.n_reflections; p++) { tot += std::min<long>(w, cap); w = std::min<long>(2 * w, cap); }
    if (cfg.record_multi_path) tot *= 2;
    return (int)std::max<long>(tot, 1);
}

// trace orders of a beam table (results are always stored under the reference index, so they only change speed):
//   pass 0     : k_trace tiles (beam sample, azimuth) into waves itself (see there); this order
//                decides which samples share a tile / are neighbours in the launch: rows of nearly
//                equal elevation, sorted by yaw inside a row
//   pass 1 ... : inherited through torder from a second order of the beam samples, yaw-major rows
//                (the reflected fan of a yaw slice stays together; measured against elevation-major
//                and Morton orders)
void beam_trace_orders(const float* beams, size_t nb, std::vector<uint32_t>& order, std::vector<uint32_t>& order2)
{
    auto make_order = [&](int major, std::vector<uint32_t>& o) {   // major: 2 = elevation (z), 1 = yaw (y)
        o.resize(nb);
        for (size_t i = 0; i < nb; i++) o[i] = (uint32_t)i;
        const int minor = 3 - major;
        std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return beams[3 * a + major] < beams[3 * b + major]; });
        for (size_t i = 0; i < nb; i += 16)
            std::stable_sort(o.begin() + i, o.begin() + std::min(nb, i + 16), [&](uint32_t a, uint32_t b) { return beams[3 * a + minor] < beams[3 * b + minor]; });
    };
    make_order(2, order);
    make_order(1, order2);
}

// a set-up table goes up through a page-locked staging block and a word-copy kernel on the NULL stream (ordered exactly like the
// hipMemcpy it replaces, and complete on return): no dispatch of the runtime's own copy kernel is left in a run's kernel trace.
// Larger than 4 MB, or not whole words: hipMemcpy
hipError_t upload_table(rr_ctx* c, void* d_dst, const void* src, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    if (!c->flush_kernel || bytes % 4 != 0 || bytes > ((size_t)4 << 20)) return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice);
    if (c->h_rb_bytes < bytes) {
        if (c->h_rb) (void)hipHostFree(c->h_rb);
        c->h_rb = nullptr; c->h_rb_bytes = 0;
        hipError_t e = hipHostMalloc(&c->h_rb, bytes + 4096, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        c->h_rb_bytes = bytes + 4096;
    }
    std::memcpy(c->h_rb, src, bytes);
    launch_copy_words(c->h_rb, d_dst, bytes, nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

int upload_tables(rr_ctx* c)
{
    if (!c->tables_dirty) return 0;
    RR_HIP(c, hipDeviceSynchronize());   // frames in flight on the lanes still read the old tables
    const rr_config& g = c->cfg;
    const unsigned dirty = c->tables_dirty;
    if (dirty & (rr_ctx::D_CFG | rr_ctx::D_BEAMS | rr_ctx::D_MAT)) c->hist_gen++;     // wave counts per pass change: the trace-grid history starts over
    // captured launches (launch graphs) hold table pointers and scalars of the old parameters.  Fresh noise offsets or motion
    // tables of the SAME shape -- what a node sets before every frame (RadarCPU.cpp:461-472, :190-196) -- only change the
    // contents of a buffer the graphs already point at
    bool regen = (dirty & (rr_ctx::D_CFG | rr_ctx::D_BEAMS | rr_ctx::D_MAT)) != 0;
    const void* noise_before = c->d_noise.p; const int noise_rows_before = c->noise_rows;
    const void* motion_before = c->motion_live ? (const void*)c->d_motion.p : nullptr; const int motion_rows_before = c->motion_rows;
    if (dirty & rr_ctx::D_CFG) {
    // Tas.R = EulerAngles{0,0,theta(angle)} -> quaternion (rmagine ZYX), RadarCPU.cpp:202
    std::vector<float4> qas((size_t)g.n_angles);
    for (int k = 0; k < g.n_angles; k++) {
        const float theta = g.theta_min + (float)k * g.theta_inc;
        const float roll = 0.0f, pitch = 0.0f, yaw = theta;
        const float cr = cosf(roll / 2.0f), sr = sinf(roll / 2.0f);
        const float cp = cosf(pitch / 2.0f), sp = sinf(pitch / 2.0f);
        const float cy = cosf(yaw / 2.0f), sy = sinf(yaw / 2.0f);
        float4 q;
        q.w = cr * cp * cy + sr * sp * sy;
        q.x = sr * cp * cy - cr * sp * sy;
        q.y = cr * sp * cy + sr * cp * sy;
        q.z = cr * cp * sy - sr * sp * cy;
        qas[k] = q;
    }
    RR_HIP(c, c->d_qas.ensure(qas.size()));
    RR_HIP(c, upload_table(c, c->d_qas.p, qas.data(), qas.size() * sizeof(float4)));
    }

    if (dirty & rr_ctx::D_BEAMS) {
    const size_t nb = c->beams.size() / 3;
    std::vector<float4> b4(nb);
    for (size_t i = 0; i < nb; i++) b4[i] = make_float4(c->beams[3 * i], c->beams[3 * i + 1], c->beams[3 * i + 2], 0.0f);
    RR_HIP(c, c->d_beams.ensure(nb));
    if (nb) RR_HIP(c, upload_table(c, c->d_beams.p, b4.data(), nb * sizeof(float4)));
    {
        std::vector<uint32_t> order, order2;
        beam_trace_orders(c->beams.data(), nb, order, order2);
        RR_HIP(c, c->d_beam_order2.ensure(nb));
        if (nb) RR_HIP(c, upload_table(c, c->d_beam_order2.p, order2.data(), nb * sizeof(uint32_t)));
        RR_HIP(c, c->d_beam_order.ensure(nb));
        if (nb) RR_HIP(c, upload_table(c, c->d_beam_order.p, order.data(), nb * sizeof(uint32_t)));
    }
    }

    if (dirty & rr_ctx::D_MAT) {
    std::vector<float4> m4(c->materials.size());
    for (size_t i = 0; i < m4.size(); i++)
        m4[i] = make_float4(c->materials[i].velocity, c->materials[i].ambient, c->materials[i].diffuse, c->materials[i].specular);
    RR_HIP(c, c->d_materials.ensure(m4.size()));
    if (!m4.empty()) RR_HIP(c, upload_table(c, c->d_materials.p, m4.data(), m4.size() * sizeof(float4)));
    // angles of total reflection, tabulated on the device (the very asin the kernels used to call per wave-pass)
    RR_HIP(c, c->d_mat_limits.ensure(m4.size()));
    launch_mat_limits(c->d_materials.p, m4.size(), c->d_mat_limits.p, nullptr);
    RR_HIP(c, hipGetLastError());
    // the frame streams are non-blocking: nothing orders them behind the NULL stream this table was launched on
    RR_HIP(c, hipStreamSynchronize(nullptr));
    RR_HIP(c, c->d_objmat.ensure(c->object_materials.size()));
    if (!c->object_materials.empty())
        RR_HIP(c, upload_table(c, c->d_objmat.p, c->object_materials.data(), c->object_materials.size() * sizeof(int32_t)));
    }

    if (dirty & rr_ctx::D_CFG) {
    make_smear(g, c->smear, c->smear_mode);
    RR_HIP(c, c->d_smear.ensure(c->smear.size()));
    if (!c->smear.empty()) RR_HIP(c, upload_table(c, c->d_smear.p, c->smear.data(), c->smear.size() * sizeof(float)));

    RR_HIP(c, c->d_decay.ensure((size_t)std::max(1, g.n_cells)));
    launch_decay_table(c->d_decay.p, g.n_cells, g.resolution, g.ambient_noise_energy_loss, nullptr);
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipDeviceSynchronize());
    }

    if (dirty & (rr_ctx::D_NOISE | rr_ctx::D_CFG)) {
    // one row of n_angles offsets, or k rows: frame f of a batch then takes row f % k (the reference draws
    // fresh offsets for every frame, RadarCPU.cpp:461-472)
    const size_t A = (size_t)g.n_angles;
    if (!c->noise.empty() && c->noise.size() % A != 0)
        return fail(c, -3, "rr_set_noise_offsets: the number of offsets must be a multiple of n_angles (one row per frame of a batch)");
    c->noise_rows = c->noise.size() >= 2 * A ? (int)(c->noise.size() / A) : 1;
    std::vector<float> nz((size_t)c->noise_rows * A, 0.0f);
    for (size_t i = 0; i < nz.size() && i < c->noise.size(); i++) nz[i] = c->noise[i];
    RR_HIP(c, c->d_noise.ensure(nz.size()));
    RR_HIP(c, upload_table(c, c->d_noise.p, nz.data(), nz.size() * sizeof(float)));
    }
    if ((dirty & (rr_ctx::D_MOTION | rr_ctx::D_CFG)) && !c->motion.empty()) {
        // one table of n_angles poses, or k tables: frame f of a batch then takes table f % k (one sweep of the antenna per frame)
        if (c->motion.size() % (7 * (size_t)g.n_angles) != 0)
            return fail(c, -3, "rr_set_motion_poses: the number of poses must be a multiple of n_angles (one table per frame of a batch)");
        c->motion_rows = (int)(c->motion.size() / (7 * (size_t)g.n_angles));
        RR_HIP(c, c->d_motion.ensure(c->motion.size()));
        RR_HIP(c, upload_table(c, c->d_motion.p, c->motion.data(), c->motion.size() * sizeof(float)));
    }
    c->motion_live = !c->motion.empty();
    const void* motion_after = c->motion_live ? (const void*)c->d_motion.p : nullptr;
    if (regen || noise_before != (const void*)c->d_noise.p || noise_rows_before != c->noise_rows ||
        motion_before != motion_after || motion_rows_before != c->motion_rows) c->graph_gen++;
    c->tables_dirty = 0;
    return 0;
}

void drop_graph(Lane::FrameGraph& fg)
{
    for (int k = 0; k < 2; k++) {       // an exec is destroyed only after its last launch has left the GPU
        if (fg.ev[k]) { if (fg.ev_pending[k]) (void)hipEventSynchronize(fg.ev[k]); (void)hipEventDestroy(fg.ev[k]); fg.ev[k] = nullptr; fg.ev_pending[k] = false; }
    }
    if (fg.ge) (void)hipGraphExecDestroy(fg.ge);
    if (fg.ge2) (void)hipGraphExecDestroy(fg.ge2);
    if (fg.g) (void)hipGraphDestroy(fg.g);
    fg.ge = fg.ge2 = nullptr; fg.g = nullptr;
}
void drop_graphs(Lane& L)
{
    for (Lane::FrameGraph& fg : L.graphs) drop_graph(fg);
    L.graphs.clear();
}

int ensure_frame_buffers(rr_ctx* c, Lane& L, int n_seg, bool want_f32)
{
    const rr_config g = eff_config(c);      // (a parameter batch may ask for more passes than the config)
    const int n_beam = (int)(c->beams.size() / 3);
    const int cap = wave_capacity(g, n_beam);
    const int sigcap = signal_capacity(g, n_beam, cap);
    const size_t S = (size_t)n_seg;
    const size_t per_seg = (size_t)cap * (2 * 2 * 48 + 2 * 4 + 2 * 9 + 8) + (size_t)sigcap * 8 + (size_t)g.n_cells * 5;
    size_t free_b = 0, total_b = 0;
    RR_HIP(c, hipMemGetInfo(&free_b, &total_b));
    if (S * per_seg > total_b / 2)
        return fail(c, -6, "wave queue capacity needs more than half of device memory; lower max_waves_per_azimuth");
    for (int k = 0; k < 2; k++) {
        RR_HIP(c, L.d_wA[k].ensure(S * 2 * cap));
        RR_HIP(c, L.d_wB[k].ensure(S * 2 * cap));
        RR_HIP(c, L.d_wC[k].ensure(S * 2 * cap));
        RR_HIP(c, L.d_idx[k].ensure(S * cap));
        RR_HIP(c, L.d_torder[k].ensure(S * cap));
        RR_HIP(c, L.d_count[k].ensure(S));
    }
    RR_HIP(c, L.d_cflag.ensure(S * 2 * cap));
    RR_HIP(c, L.d_refpos.ensure(S * 2 * cap));
    RR_HIP(c, L.d_sigtmp.ensure(S * 2 * cap));
    RR_HIP(c, L.d_hit.ensure(S * cap));
    RR_HIP(c, L.d_sig.ensure(S * sigcap));
    RR_HIP(c, L.d_sig_count.ensure(S));
    if (!L.d_counters.p) { RR_HIP(c, L.d_counters.ensure(1)); RR_HIP(c, hipMemset(L.d_counters.p, 0, sizeof(Counters))); }
    if (!L.d_sticky.p) { RR_HIP(c, L.d_sticky.ensure(1)); RR_HIP(c, hipMemset(L.d_sticky.p, 0, sizeof(uint32_t))); }
    RR_HIP(c, L.d_seg_stats.ensure(S * (size_t)std::max(1, g.n_reflections)));
    if (!L.d_hint.p) { RR_HIP(c, L.d_hint.ensure(1)); RR_HIP(c, hipMemset(L.d_hint.p, 0, sizeof(GridHint))); L.hist_gen = 0; }
    if (!L.h_hist) { RR_HIP(c, hipHostMalloc((void**)&L.h_hist, kMaxPasses * sizeof(uint32_t), hipHostMallocDefault)); std::memset(L.h_hist, 0, kMaxPasses * sizeof(uint32_t)); }
    RR_HIP(c, L.d_ovf_list.ensure(S * (size_t)kMaxPasses)); L.ovf_stride = (int)S;
    RR_HIP(c, L.d_cols_u8.ensure(S * g.n_cells));
    if (want_f32) RR_HIP(c, L.d_cols_f32.ensure(S * g.n_cells));
    // traversal stack: LDS part + spill
    L.stack_lds = (int)std::max<uint32_t>(1, std::min<uint32_t>(c->stack_need, (uint32_t)c->stack_lds_max));   // 64 B of LDS per entry per wave
    const int spill_depth = (int)c->stack_need - L.stack_lds;
    // one spill column per ray slot a launch can address: later passes ceil(cap/32)*32 slots per segment,
    // pass 0 its (sample x azimuth) tiles, whose padding can exceed S * n_beam (e.g. ONE segment: 16 x n_beam)
    const size_t A0 = (size_t)c->pass0_az, Sw0 = 16 / A0;
    const size_t slots0 = ((((S + A0 - 1) / A0) * (((size_t)n_beam + Sw0 - 1) / Sw0) + 1) / 2) * 32;      // (rounded up to pairs of waves: covers 64- and 128-thread workgroups)
    const size_t threads = std::max(S * (size_t)((cap + 63) / 64) * 64, slots0);
    L.spill_stride = (int)threads;
    if (spill_depth > 0) RR_HIP(c, L.d_spill.ensure((size_t)spill_depth * threads));
    else RR_HIP(c, L.d_spill.ensure(1));
    L.buf_seg = n_seg; L.buf_cap = cap; L.buf_sigcap = sigcap; L.buf_cells = g.n_cells; L.buf_passes = std::max(1, g.n_reflections);
    RR_HIP(c, L.d_poses.ensure((size_t)RR_MAX_BATCH * 8));
    L.graph_gen = 0;           // the lane's buffers moved: its captured launches point at the old ones
    return 0;
}

// Size the lane's frame buffers for n_seg segments under the CURRENT config.  This is the one place
// that decides whether the buffers fit (segments, wave / signal capacity, n_cells, traversal stack):
// every entry point sizes through here BEFORE it takes a pointer into the lane, and run_frame()
// resolves "the lane's own column buffer" only after it -- a reallocation can never leave a caller
// with a stale pointer.  Frames still in flight may use the old buffers: drain the device first.
int prepare_lane(rr_ctx* c, Lane& L, int n_seg, bool want_f32 = false)
{
    const rr_config g = eff_config(c);      // (a parameter batch may ask for more passes than the config)
    const int n_beam = (int)(c->beams.size() / 3);
    const int cap = wave_capacity(g, n_beam);
    const int sigcap = signal_capacity(g, n_beam, cap);
    // a parameter batch (its sets bring their own numbers of passes, so the nominal capacity changes from call to call)
    // also runs in buffers that are LARGER than it needs: the kernels take every stride from the lane (Params::cap), and
    // with the default capacity nothing can overflow that would not have overflowed the nominal one.  Ordinary frames keep
    // the exact layout (a user-lowered max_waves_per_azimuth must be reported when exceeded).
    const bool roomy = c->passes_override >= 0 && c->cfg.max_waves_per_azimuth <= 0 && L.buf_cap >= cap && L.buf_sigcap >= sigcap &&
                       L.buf_passes >= g.n_reflections;
    const bool fits = L.buf_seg >= n_seg && g.n_cells == L.buf_cells &&
                      ((cap == L.buf_cap && sigcap == L.buf_sigcap && L.buf_passes >= g.n_reflections) || roomy) &&
                      (!want_f32 || (L.d_cols_f32.p && L.d_cols_f32.n >= (size_t)L.buf_seg * g.n_cells));
    if (fits) return 0;
    RR_HIP(c, hipDeviceSynchronize());
    return ensure_frame_buffers(c, L, std::max(n_seg, L.buf_seg), want_f32);
}

void fill_params(rr_ctx* c, Lane& L, Params& P, const float pose[7], int az_begin, int n_seg,
                 uint8_t* d_cols_u8, float* d_cols_f32)
{
    const rr_config g = eff_config(c);      // (a parameter batch may ask for more passes than the config)
    std::memset(&P, 0, sizeof(P));
    P.nodes = reinterpret_cast<const Node4*>(c->d_bvh.p); P.tris = reinterpret_cast<const TriRec*>(c->d_bvh.p + c->tri_base4);
    P.tri_base4 = c->tri_base4;
    P.q_as = c->d_qas.p; P.beams = c->d_beams.p; P.beam_order = c->d_beam_order.p; P.beam_order2 = c->d_beam_order2.p; P.materials = c->d_materials.p;
    P.mat_limits = c->d_mat_limits.p; P.limit_same = c->limit_same;
    P.object_materials = c->d_objmat.p; P.smear = c->d_smear.p;
    P.noise_rnd = g.ambient_noise ? c->d_noise.p : nullptr; P.noise_rows = c->noise_rows;
    P.decay = c->d_decay.p;
    P.motion_poses = c->motion.empty() ? nullptr : c->d_motion.p; P.motion_rows = c->motion_rows;
    for (int k = 0; k < 2; k++) {
        P.waves[k].A = L.d_wA[k].p; P.waves[k].B = L.d_wB[k].p; P.waves[k].C = L.d_wC[k].p;
        P.idx[k] = L.d_idx[k].p; P.count[k] = L.d_count[k].p; P.torder[k] = L.d_torder[k].p;
    }
    P.refpos = L.d_refpos.p;
    P.cflag = L.d_cflag.p; P.sigtmp = L.d_sigtmp.p; P.hit = L.d_hit.p;
    P.sig = L.d_sig.p; P.sig_count = L.d_sig_count.p; P.spill = L.d_spill.p; P.counters = L.d_counters.p; P.sticky = L.d_sticky.p; P.seg_stats = L.d_seg_stats.p;
    P.cols_u8 = d_cols_u8; P.cols_f32 = d_cols_f32;
    P.az_begin = az_begin; P.n_seg = n_seg;
    P.n_beam = (int)(c->beams.size() / 3); P.cap = L.buf_cap; P.sigcap = L.buf_sigcap;
    P.n_cells = g.n_cells; P.n_angles = g.n_angles;
    P.n_materials = (int)c->materials.size(); P.n_objects = (int)c->object_materials.size();
    P.material_id_air = c->material_id_air;
    P.n_passes = g.n_reflections;
    P.record_multi_reflection = g.record_multi_reflection; P.record_multi_path = g.record_multi_path;
    P.brdf_model = g.brdf_model;
    P.signal_denoising = c->smear.empty() ? 0 : g.signal_denoising;
    P.smear_w = (int)c->smear.size(); P.smear_mode = c->smear_mode;
    P.ambient_noise = g.ambient_noise; P.scroll = g.scroll_image;
    P.thr = g.wave_energy_threshold; P.range_max = g.range_max; P.hit_pad = c->hit_pad;
    P.resolution = g.resolution; P.multipath_threshold = g.multipath_threshold;
    P.energy_max_f = (float)g.energy_max; P.signal_max = g.signal_max;
    P.noise_at_0 = g.ambient_noise_at_signal_0; P.noise_at_1 = g.ambient_noise_at_signal_1;
    P.noise_e_max = g.ambient_noise_energy_max; P.noise_e_min = g.ambient_noise_energy_min;
    P.noise_e_loss = g.ambient_noise_energy_loss;
    P.spill_stride = L.spill_stride; P.stack_lds = L.stack_lds;
    P.spill_depth = std::max(0, (int)c->stack_need - L.stack_lds);
    P.pass0_az = c->pass0_az;
    P.cull_pop = c->cull_pop; P.seg_chunk = c->seg_chunk; P.stackless = c->stackless;
    P.grid_hint = L.d_hint.p; P.ovf_list = L.d_ovf_list.p; P.ovf_stride = L.ovf_stride;     // rows stay at the bound until run_frame tightens them
    P.hist_host = (c->tight_grid && g.n_reflections > 1) ? L.h_hist : nullptr;              // the chain's k_column stores the history there (read without a fence by later batches)
}

// a free copy record of the lane (waits for the oldest copy if both are still in flight)
int take_rec(rr_ctx* c, Lane& L, Lane::CopyRec** out)
{
    Lane::CopyRec& r = L.rec[L.rec_next];
    L.rec_next ^= 1;
    if (r.pending) { RR_HIP(c, hipEventSynchronize(r.ev)); r.pending = false; r.dst = nullptr; }
    *out = &r;
    return 0;
}

// device -> host on stream s: the library's own copy kernel when the destination is page-locked (`visible`) and everything is
// 16-byte aligned, else hipMemcpyAsync (rr_copy_to_host_async in the header says why)
int copy_out(rr_ctx* c, const void* d_src, void* h_dst, size_t bytes, bool visible, hipStream_t s)
{
    if (bytes == 0) return 0;
    if (c->flush_kernel && visible && bytes % 16 == 0 && ((uintptr_t)h_dst | (uintptr_t)d_src) % 16 == 0) {
        launch_copy_host(d_src, h_dst, bytes, c->flush_blocks, c->flush_inflight, c->flush_xcd, s, c->flush_threads);
        RR_HIP(c, hipGetLastError());
    } else RR_HIP(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    return 0;
}
bool host_visible(const void* p)
{
    hipPointerAttribute_t at;
    const bool v = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();     // a pageable pointer makes hipPointerGetAttributes fail: not an error of the caller's call
    return v;
}

// a small synchronous read-back of device words (counters, per-pass statistics) without asking the runtime for a copy: a kernel
// stores them into a page-locked block of the context, the host copies from there.  The device must be idle on these words
// (the callers have synchronised).  Sizes that are not multiples of 16 take hipMemcpy
int read_back(rr_ctx* c, void* dst, const void* d_src, size_t bytes)
{
    if (bytes == 0) return 0;
    if (!c->flush_kernel || bytes % 16 != 0 || (uintptr_t)d_src % 16 != 0) { RR_HIP(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost)); return 0; }
    if (c->h_rb_bytes < bytes) {
        if (c->h_rb) (void)hipHostFree(c->h_rb);
        c->h_rb = nullptr; c->h_rb_bytes = 0;
        RR_HIP(c, hipHostMalloc(&c->h_rb, bytes + 4096, hipHostMallocDefault));
        c->h_rb_bytes = bytes + 4096;
    }
    launch_copy_host(d_src, c->h_rb, bytes, 4, 0, -1, c->stream, 256);
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(dst, c->h_rb, bytes);
    return 0;
}

// the lane's deferred host copy, now, as a plain copy on the stream its batch ran on
// the images the lane's last host-delivery batch handed to the SDMA worker have left d_img_u8 (host wait; over long before a
// lane comes round again)
void settle_sdma(rr_ctx* c, Lane& L, int slot = -1, const void* only_dst = nullptr)
{
    for (int b = 0; b < 2; b++) {
        if ((slot >= 0 && b != slot) || !L.sdma_job[b]) continue;
        if (only_dst && L.sdma_dst[b] != only_dst) continue;
        if (c->sdma) sdma_wait(c->sdma, L.sdma_job[b]);
        L.sdma_job[b] = 0; L.sdma_dst[b] = nullptr;
    }
}

int flush_deferred(rr_ctx* c, Lane& L, bool settle = true)
{
    if (settle) settle_sdma(c, L);       // (every user of the lane but the SDMA route itself, which looks after its two buffers)
    if (!L.deferred) return 0;
    Lane::CopyRec* r = nullptr;
    int rc = take_rec(c, L, &r); if (rc) return rc;
    { const int rcc = copy_out(c, L.d_img_u8.p, L.def_dst, L.def_bytes, L.def_foldable, L.def_stream); if (rcc) return rcc; }
    RR_HIP(c, hipEventRecord(r->ev, L.def_stream));
    r->dst = L.def_dst; r->pending = true;
    RR_HIP(c, hipEventRecord(L.ev_consumed, L.def_stream));
    L.pending_consume = true;
    L.deferred = false;
    return 0;
}

struct TimedScope {
    rr_ctx* c; hipStream_t s; const char* name; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    TimedScope(rr_ctx* c_, hipStream_t s_, const char* n_) : c(c_), s(s_), name(n_) {
        on = c->timing == 1;
        if (on) { a = c->take_event(); b = c->take_event(); (void)hipEventRecord(a, s); }
        if (c->roctx) roctx_push(name);
    }
    ~TimedScope() {
        if (on) { (void)hipEventRecord(b, s); c->timers[name].pending.emplace_back(a, b); }
        if (c->roctx) roctx_pop();
    }
};

// timing mode 1, kernels of the frame chain: the launch's own begin / end timestamps (hipExtLaunchKernel events) -- a kernel's
// duration as rocprofv3 reports it, whatever it waited for before it started (TimedScope's stream events include that wait)
struct KernelEvents {
    rr_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr;
    KernelEvents(rr_ctx* c_, const char* n_) : c(c_), name(n_) {
        if (c->timing == 1) { a = c->take_event(); b = c->take_event(); }
        if (c->roctx) roctx_push(name);
    }
    ~KernelEvents() {
        if (a) c->timers[name].pending.emplace_back(a, b);
        if (c->roctx) roctx_pop();
    }
};

// child references are 28-bit float4 offsets from the base of the tree allocation (rr_bvh.h)
int check_bvh_size(rr_ctx* c, size_t n_nodes, size_t n_tris)
{
    if (n_nodes * 8 + (n_tris + 4) * 3 >= (1ull << 28))
        return fail(c, -4, "rr_set_mesh: tree too large for 28-bit references (8 x nodes + 3 x triangles must stay below 2^28: about 60M triangles)");
    return 0;
}

int check_ready(rr_ctx* c)
{
    if (!c) return -1;
    if (!c->have_mesh) return fail(c, -2, "rr_set_mesh has not been called");
    if (!c->have_cfg) return fail(c, -2, "rr_set_config has not been called");
    if (!c->have_materials) return fail(c, -2, "rr_set_materials has not been called");
    if (c->beams.empty() && c->cfg.n_reflections > 0) return fail(c, -2, "rr_set_beam_samples has not been called");
    return 0;
}

// a parameter batch as run_frame sees it: per frame its passes and beam group, per group the frame pass 0 is traced for
struct SetPlan {
    int n_groups = 1;
    unsigned char frame_passes[64], frame_beam[64], group_frame[64];
    const float4* d_beams = nullptr; const uint32_t* d_order = nullptr; const uint32_t* d_order2 = nullptr;   // [n_groups][n_beam]; null: the ctx's tables
};

int run_frame(rr_ctx* c, Lane& L, const float* pose, int az_begin, int az_end,
              uint8_t* d_cols_u8 /* null: the lane's own buffer */, float* d_cols_f32, hipStream_t s, int n_frames = 1,
              const float4* d_matsets = nullptr, int mat_stride = 0, bool lane_f32 = false,
              const uint8_t* copy_src = nullptr, uint8_t* copy_dst = nullptr, size_t copy_bytes = 0,
              const SetPlan* plan = nullptr)
{
    const rr_config g = eff_config(c);      // (a parameter batch may ask for more passes than the config)
    if (az_begin < 0 || az_end > g.n_angles || az_begin > az_end) return fail(c, -3, "azimuth range out of bounds");
    const int n_loc = az_end - az_begin;
    const int n_seg = n_loc * n_frames;
    if (n_seg == 0) return 0;
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return fail(c, -3, "frame batch must be 1..64");
    for (int k = 0; k < 7 * (d_matsets ? 1 : n_frames); k++) if (!std::isfinite(pose[k])) return fail(c, -3, "non-finite pose");
    int rc = upload_tables(c); if (rc) return rc;
    // ONE per-azimuth pose table and several frames: every frame would be the same sweep and the call's poses would be ignored
    // without a word (advisor, round 5) -- a batch under include_motion brings one table per frame (or k tables, frame f -> f % k)
    if (!d_matsets && n_frames > 1 && !c->motion.empty() && c->motion_rows == 1)
        return fail(c, -3, "a pose batch while ONE per-azimuth pose table is set (rr_set_motion_poses): give one table per frame (k x n_angles poses) or clear the table");
    rc = prepare_lane(c, L, n_seg, lane_f32); if (rc) return rc;
    if (!d_cols_u8) d_cols_u8 = L.d_cols_u8.p;       // the lane's own column buffer, valid only from here on
    if (lane_f32) d_cols_f32 = L.d_cols_f32.p;
    Params P;
    fill_params(c, L, P, pose, az_begin, n_seg, d_cols_u8, d_cols_f32);
    P.n_loc = n_loc; P.n_frames = n_frames;
    if (d_matsets) {   // parameter batch: one pose, one material table per frame
        P.materials = d_matsets; P.mat_limits = L.d_matset_limits.p; P.mat_stride = mat_stride;
        P.set_mode = 1;
        P.noise_rows = 1;     // every set is the SAME frame under another parameter set: one noise realisation (row 0)
        P.motion_rows = 1;    // ... and one sweep of the antenna (table 0); every set the SAME pose: q_sm / t_sm (no pose table)
        SetPlan dflt;
        if (!plan) {          // material sets only: one beam, every frame the config's passes
            for (int f = 0; f < n_frames; f++) { dflt.frame_passes[f] = (unsigned char)g.n_reflections; dflt.frame_beam[f] = 0; }
            dflt.group_frame[0] = 0; plan = &dflt;
        }
        P.n_groups = plan->n_groups;
        std::memcpy(P.frame_passes, plan->frame_passes, (size_t)n_frames);
        std::memcpy(P.frame_beam, plan->frame_beam, (size_t)n_frames);
        std::memcpy(P.group_frame, plan->group_frame, (size_t)plan->n_groups);
        if (plan->d_beams) { P.beams = plan->d_beams; P.beam_order = plan->d_order; P.beam_order2 = plan->d_order2; }
    }
    if (c->stats_mode || g.n_reflections == 0) RR_HIP(c, hipMemsetAsync(L.d_counters.p, 0, sizeof(Counters), s));
    L.last_n_seg = n_seg; L.last_n_passes = g.n_reflections;
    // later-pass trace rows as long as earlier batches needed (rr_device.h: GridHint).  Not for the statistics build (the
    // repair launch does not count), the spill path (its columns are laid out for the full rows) or parameter batches
    // (frames with their own beams and passes)
    if (L.hist_gen != c->hist_gen) {
        RR_HIP(c, hipMemsetAsync(L.d_hint.p, 0, sizeof(GridHint), s));
        std::memset(L.h_hist, 0, kMaxPasses * sizeof(uint32_t));
        L.hist_gen = c->hist_gen;
    }
    const bool tight = c->tight_grid && !c->stats_mode && !P.set_mode && P.spill_depth == 0 && g.n_reflections <= kMaxPasses;
    if (tight) {
        for (int pass = 1; pass < g.n_reflections; pass++) {
            const long bound = std::min<long>((long)P.cap, pass < 20 ? (long)P.n_beam << pass : (long)P.cap);
            const long full = (bound + 15) / 16;
            uint32_t h = 0;
            for (const Lane& o : c->lanes) if (o.h_hist && o.hist_gen == c->hist_gen) h = std::max(h, o.h_hist[pass]);
            long row = h ? std::min<long>(full, ((long)h + (long)h / 16 + 32 + 15) / 16) : full;
            // an ODD number of workgroups per row: the hardware deals workgroups out to the 8 XCDs round robin in flat order
            // (y * row + x), so with a row length that shares a factor with 8 the same x always lands on the same XCDs -- and the
            // tail of every row (few or no live rays) always on the same ones.  Rows rounded to a multiple of four: 435 -> 457 us
            // per launch alone, -2.4 % images/s on the target (measured by accident, DESIGN_EXPERIMENTS.md)
            row |= 1;
            if (c->tight_force) row = std::min<long>(full, c->tight_force);
            P.tight_groups[pass] = (row < full && row < 65535) ? (unsigned short)row : 0;
        }
    }
    std::memcpy(L.last_rows, P.tight_groups, sizeof(L.last_rows));
    // the poses of the call ride in the pass-0 trace launch (by value), which also writes them into the lane's pose table for
    // the launches behind it; a parameter batch (every set the same pose) and a single frame use row 0
    PoseArgs pa;
    std::memset(&pa, 0, sizeof(pa));
    pa.n = d_matsets ? 1 : n_frames;
    for (int f = 0; f < pa.n; f++) for (int k = 0; k < 7; k++) pa.p[f][k] = pose[7 * f + k];
    P.pose_table = reinterpret_cast<float4*>(L.d_poses.p);
    // the launch chain of the batch
    auto enqueue = [&](Params& Q) -> int {
    for (int pass = 0; pass < g.n_reflections; pass++) {
        // the previous batch's images ride on the later-pass launches, one slice each (rr_simulate_batch_host_async)
        Q.copy_blocks = 0;
        if (pass >= 1 && copy_src) {
            const size_t n16 = copy_bytes / 16, slices = (size_t)g.n_reflections - 1, k = (size_t)pass - 1;
            const size_t b = n16 * k / slices, e = n16 * (k + 1) / slices;
            Q.copy_src = reinterpret_cast<const uint4*>(copy_src) + b; Q.copy_dst = reinterpret_cast<uint4*>(copy_dst) + b;
            Q.copy_n16 = e - b; Q.copy_blocks = c->copy_blocks;
        }
        if (c->roctx) roctx_push(pass == 0 ? "trace pass 0" : "trace");
        if (c->timing) {
            // the kernel's own begin/end timestamps (hipExtLaunchKernel events), on its launch stream
            hipEvent_t a = c->take_event(), b = c->take_event();
            // ... and of the repair launch behind a tightened row (timer "trace_repair": the trace figure does not contain it)
            const bool rep = pass > 0 && pass < kMaxPasses && Q.tight_groups[pass];
            hipEvent_t ra = rep ? c->take_event() : nullptr, rb = rep ? c->take_event() : nullptr;
            bool launched = false;
            launch_trace(Q, pass, &pa, c->stats_mode, s, a, b, ra, rb, &launched);
            c->timers[pass == 0 ? "trace0" : "trace"].pending.emplace_back(a, b);
            if (launched) c->timers["trace_repair"].pending.emplace_back(ra, rb);
            else if (rep) { c->event_pool.push_back(ra); c->event_pool.push_back(rb); }
        } else {
            launch_trace(Q, pass, &pa, c->stats_mode, s);
        }
        if (c->roctx) roctx_pop();
        { KernelEvents t(c, "shade"); launch_shade(Q, pass, s, t.a, t.b); }
        if (pass < g.n_reflections - 1) { KernelEvents t(c, "scan"); launch_scan(Q, pass, s, t.a, t.b); }
    }
    { KernelEvents t(c, "column"); launch_column(Q, s, t.a, t.b); }
    return 0;
    };
    // Launch graphs: a chain that has been issued before with the same shape is captured once and replayed -- one
    // hipGraphLaunch instead of 4..20 launches (host time per device entry of rr_multi: 45-81 -> ~25 us).  Only plain pose
    // batches: no carried host copy (its pointers move from call to call), no parameter batch, no timing / statistics /
    // roctx instrumentation; whatever a captured launch bakes in is covered by graph_gen (tables, tree, lane buffers) or
    // by the key (azimuth block, frames, output buffer, trace rows)
    if (L.graph_gen != c->graph_gen) { drop_graphs(L); L.graph_gen = c->graph_gen; }
    const bool graphable = !d_matsets && c->use_graphs && !copy_src && !c->timing && !c->stats_mode && !c->roctx && !d_cols_f32 && g.n_reflections > 0;
    if (graphable) {
        Lane::FrameGraph* fg = nullptr;
        for (Lane::FrameGraph& x : L.graphs)
            if (x.az_begin == az_begin && x.az_end == az_end && x.n_frames == n_frames && x.cols == (const void*)d_cols_u8 &&
                std::memcmp(x.rows, P.tight_groups, sizeof(x.rows)) == 0) { fg = &x; break; }
        if (!fg) {
            if (L.graphs.size() >= 12) {           // forget the least recently used shape
                size_t lru = 0;
                for (size_t k = 1; k < L.graphs.size(); k++) if (L.graphs[k].last_use < L.graphs[lru].last_use) lru = k;
                drop_graph(L.graphs[lru]);
                L.graphs.erase(L.graphs.begin() + (long)lru);
            }
            Lane::FrameGraph n;
            n.az_begin = az_begin; n.az_end = az_end; n.n_frames = n_frames; n.cols = d_cols_u8;
            std::memcpy(n.rows, P.tight_groups, sizeof(n.rows));
            L.graphs.push_back(n);
            fg = &L.graphs.back();
        }
        fg->last_use = ++c->graph_clock;
        if (!fg->ge && fg->hits >= 1) {            // the second call with this shape: worth a capture
            // (No SDMA worker may be waiting on an event of this stream while it captures: the runtime treats a
            // hipEventSynchronize on an event whose stream is capturing as an error and invalidates the capture -- found by
            // fuzz_batch in round 6.  Captures are rare, once per shape: let the deliveries in flight finish first.)
            if (c->sdma) sdma_wait_all(c->sdma);
            Params Q = P;
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                const int rcq = enqueue(Q);
                hipGraph_t gph = nullptr;
                e = hipStreamEndCapture(s, &gph);
                if (rcq == 0 && e == hipSuccess && gph) {
                    hipGraphExec_t ge = nullptr, ge2 = nullptr;
                    if (hipGraphInstantiate(&ge, gph, nullptr, nullptr, 0) == hipSuccess && hipGraphInstantiate(&ge2, gph, nullptr, nullptr, 0) == hipSuccess &&
                        hipEventCreateWithFlags(&fg->ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&fg->ev[1], hipEventDisableTiming) == hipSuccess) {
                        size_t nn = 0; (void)hipGraphGetNodes(gph, nullptr, &nn);
                        std::vector<hipGraphNode_t> nodes(nn); (void)hipGraphGetNodes(gph, nodes.data(), &nn);
                        for (hipGraphNode_t nd : nodes) {
                            hipGraphNodeType ty; hipKernelNodeParams kp{};
                            if (hipGraphNodeGetType(nd, &ty) == hipSuccess && ty == hipGraphNodeTypeKernel &&
                                hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess && kp.func == trace0_kernel(P.spill_depth > 0, P.stackless != 0)) {
                                fg->pose_node = nd; fg->pose_kp = kp; fg->pose_P = trace0_params(P); break;
                            }
                        }
                        if (fg->pose_node) { fg->g = gph; fg->ge = ge; fg->ge2 = ge2; c->graph_captures++; }
                        else { (void)hipGraphExecDestroy(ge); (void)hipGraphExecDestroy(ge2); (void)hipGraphDestroy(gph); }
                    } else { if (ge) (void)hipGraphExecDestroy(ge); if (ge2) (void)hipGraphExecDestroy(ge2); (void)hipGraphDestroy(gph); }
                    if (!fg->ge) for (int k = 0; k < 2; k++) if (fg->ev[k]) { (void)hipEventDestroy(fg->ev[k]); fg->ev[k] = nullptr; }
                } else if (gph) (void)hipGraphDestroy(gph);
            }
            (void)hipGetLastError();
            if (!fg->ge) fg->hits = -1000000;      // capture is not available here: stay with plain launches for this shape
        }
        if (fg->ge) {
            int pass0 = 0;
            void* args[3] = { (void*)&fg->pose_P, (void*)&pass0, (void*)&pa };
            hipKernelNodeParams kp = fg->pose_kp;
            kp.kernelParams = args; kp.extra = nullptr;
            hipError_t e = hipSuccess;
            const int w = c->graph_guard ? fg->flip : 0; fg->flip ^= 1;
            hipGraphExec_t ex = w ? fg->ge2 : fg->ge;
            if (c->graph_guard && fg->ev_pending[w]) { HostProfScope hp(6, "ctx:   graph: wait for the exec's previous launch"); e = hipEventSynchronize(fg->ev[w]); fg->ev_pending[w] = false; }
            { HostProfScope hp(3, "ctx:   graph: set the poses"); if (e == hipSuccess) e = hipGraphExecKernelNodeSetParams(ex, fg->pose_node, &kp); }
            { HostProfScope hp(4, "ctx:   graph: launch"); if (e == hipSuccess) e = hipGraphLaunch(ex, s); }
            if (e == hipSuccess && c->graph_guard) { e = hipEventRecord(fg->ev[w], s); fg->ev_pending[w] = e == hipSuccess; }
            if (e != hipSuccess) return fail(c, -100, std::string("launch graph replay: ") + hipGetErrorString(e));
            c->graph_replays++;
            return 0;
        }
        fg->hits++;
    }
    { HostProfScope hp(5, "ctx:   chain issued kernel by kernel"); const int rcq = enqueue(P); if (rcq) return rcq; }
    RR_HIP(c, hipGetLastError());
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int rr_abi_version(void) { return RR_ABI_VERSION; }

void rr_partition(int n_angles, int world, int rank, int* begin, int* end)
{
    // contiguous azimuth blocks that differ by at most one column (radarays_ros_amd/dist.py: partition)
    if (world < 1) world = 1;
    const int base = n_angles / world, rem = n_angles % world;
    const int b = rank * base + (rank < rem ? rank : rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

void rr_default_config(rr_config* cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_cells = 3424; cfg->n_angles = 400; cfg->n_reflections = 4;
    cfg->signal_denoising = 1;
    cfg->signal_denoising_triangular_width = 50; cfg->signal_denoising_triangular_mode = 0.35;
    cfg->signal_denoising_gaussian_width = 50; cfg->signal_denoising_gaussian_mode = 0.5;
    cfg->signal_denoising_mb_width = 50; cfg->signal_denoising_mb_mode = 0.4;
    cfg->ambient_noise = 2; cfg->scroll_image = 0;
    cfg->record_multi_reflection = 1; cfg->record_multi_path = 0;
    cfg->max_waves_per_azimuth = 0;
    cfg->resolution = 0.0438; cfg->energy_max = 0.5; cfg->signal_max = 120.0;
    cfg->ambient_noise_at_signal_0 = 0.3; cfg->ambient_noise_at_signal_1 = 0.03;
    cfg->ambient_noise_energy_max = 0.5; cfg->ambient_noise_energy_min = 0.1;
    cfg->ambient_noise_energy_loss = 0.05; cfg->multipath_threshold = 0.5;
    cfg->wave_energy_threshold = 0.001f;
    cfg->theta_min = 0.0f; cfg->theta_inc = (float)(-(2 * M_PI) / 400);
    cfg->range_max = 1000.0f;
}

rr_ctx* rr_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("rr_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                         "); this library has no CPU fallback";
        return nullptr;
    }
    if (device < 0 || device >= n) { g_create_error = "rr_create: device index out of range"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "rr_create: hipSetDevice failed"; return nullptr; }
    rr_ctx* c = new rr_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_error = "rr_create: hipStreamCreate failed"; delete c; return nullptr;
    }
    rr_default_config(&c->cfg);
    // 4 buffer sets: the sharded step loop (dist.py) keeps 4 steps in flight on its own streams
    // (measured optimum: 4 streams = 4 hardware queues); rr_simulate_device, whose frames run on the
    // lanes' OWN streams beside the caller's stream, rotates over the first 3 only (same reason)
    int n_lanes = getenv("RR_LANES") ? atoi(getenv("RR_LANES")) : 4;
    n_lanes = std::max(1, std::min(n_lanes, 8));
    c->stream_lanes = getenv("RR_STREAM_LANES") ? std::max(1, std::min(atoi(getenv("RR_STREAM_LANES")), n_lanes))
                                                : std::min(3, n_lanes);
    if (getenv("RR_PASS0_AZ")) { const int a = atoi(getenv("RR_PASS0_AZ")); if (a == 1 || a == 2 || a == 4 || a == 8 || a == 16) c->pass0_az = a; }
    if (getenv("RR_STACK_LDS")) c->stack_lds_max = std::max(1, std::min(64, atoi(getenv("RR_STACK_LDS"))));
    if (getenv("RR_ROCTX") && atoi(getenv("RR_ROCTX")) != 0) c->roctx = roctx_load();
    if (getenv("RR_FOLD_MIN_BUSY")) c->fold_min_busy = std::max(0, atoi(getenv("RR_FOLD_MIN_BUSY")));
    if (getenv("RR_CULL_POP")) c->cull_pop = atoi(getenv("RR_CULL_POP")) != 0;
    if (getenv("RR_STACKLESS")) c->stackless = atoi(getenv("RR_STACKLESS")) != 0;
    if (getenv("RR_TRACE_CHUNK")) c->seg_chunk = std::max(0, std::min(1024, atoi(getenv("RR_TRACE_CHUNK"))));
    if (getenv("RR_COPY_BLOCKS")) c->copy_blocks = std::max(0, std::min(26, atoi(getenv("RR_COPY_BLOCKS"))));
    if (getenv("RR_GRAPHS")) c->use_graphs = atoi(getenv("RR_GRAPHS")) != 0;
    if (getenv("RR_GRAPH_GUARD")) c->graph_guard = atoi(getenv("RR_GRAPH_GUARD")) != 0;
    if (getenv("RR_FLUSH_KERNEL")) c->flush_kernel = atoi(getenv("RR_FLUSH_KERNEL")) != 0;
    if (getenv("RR_HOST_COPY_STREAM")) c->host_copy_stream = atoi(getenv("RR_HOST_COPY_STREAM"));
    if (getenv("RR_HOST_SDMA")) c->host_sdma = atoi(getenv("RR_HOST_SDMA")) != 0;
    if (getenv("RR_FLUSH_BLOCKS")) c->flush_blocks = std::max(1, std::min(1024, atoi(getenv("RR_FLUSH_BLOCKS"))));
    if (getenv("RR_FLUSH_INFLIGHT")) c->flush_inflight = std::max(0, std::min(64, atoi(getenv("RR_FLUSH_INFLIGHT"))));
    if (getenv("RR_FLUSH_XCD")) c->flush_xcd = std::max(-1, std::min(7, atoi(getenv("RR_FLUSH_XCD"))));
    if (getenv("RR_FLUSH_THREADS")) c->flush_threads = std::max(64, std::min(1024, atoi(getenv("RR_FLUSH_THREADS"))));
    if (getenv("RR_TIGHT_GRID")) c->tight_grid = atoi(getenv("RR_TIGHT_GRID")) != 0;
    if (getenv("RR_TIGHT_FORCE")) c->tight_force = std::max(0, atoi(getenv("RR_TIGHT_FORCE")));
    {   // the one angle of total reflection that does not depend on the material table
        const float4 same = make_float4(0.3f, 0.f, 0.f, 0.f);
        DevBuf<float4> m1; DevBuf<double> l1;
        bool ok = m1.ensure(1) == hipSuccess && l1.ensure(2) == hipSuccess &&
                  upload_table(c, m1.p, &same, sizeof(same)) == hipSuccess;
        if (ok) {
            launch_mat_limits(m1.p, 1, l1.p, nullptr);
            double two[2] = {0.0, 0.0};
            ok = hipStreamSynchronize(nullptr) == hipSuccess && read_back(c, two, l1.p, sizeof(two)) == 0;
            c->limit_same = two[0];
        }
        m1.release(); l1.release();
        if (!ok) { g_create_error = "rr_create: device set-up failed"; rr_destroy(c); return nullptr; }
    }
    c->lanes.resize((size_t)n_lanes);
    for (Lane& L : c->lanes) {
        if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&L.ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.ev_consumed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.ev_img[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.ev_img[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.rec[0].ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.rec[1].ev, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "rr_create: lane stream/event creation failed"; rr_destroy(c); return nullptr;
        }
    }
    return c;
}

void rr_destroy(rr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();   // frames may still be in flight on the lanes' or the caller's streams
    if (c->sdma) { sdma_destroy(c->sdma); c->sdma = nullptr; }      // (its queued copies wait for events that have completed by now)
    for (rr_ctx::Delivery& d : c->deliveries) (void)hipEventDestroy(d.ev);
    for (hipEvent_t e : c->delivery_events) (void)hipEventDestroy(e);
    for (auto& kv : c->timers) for (auto& p : kv.second.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->d_bvh.release(); c->d_qas.release(); c->d_beams.release(); c->d_materials.release(); c->d_mat_limits.release();
    c->d_objmat.release(); c->d_smear.release(); c->d_noise.release(); c->d_decay.release(); c->d_param_imgs.release(); c->d_sse.release(); c->d_ref_img.release(); c->d_beam_order.release(); c->d_beam_order2.release(); c->d_motion.release();
    for (Lane& L : c->lanes) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        for (int k = 0; k < 2; k++) { L.d_wA[k].release(); L.d_wB[k].release(); L.d_wC[k].release(); L.d_idx[k].release(); L.d_count[k].release(); L.d_torder[k].release(); }
        L.d_refpos.release();
        L.d_hit.release(); L.d_sig_count.release(); L.d_spill.release(); L.d_cflag.release(); L.d_cols_u8.release();
        L.d_sigtmp.release(); L.d_sig.release(); L.d_cols_f32.release(); L.d_counters.release(); L.d_sticky.release(); L.d_seg_stats.release(); L.d_matsets.release(); L.d_matset_limits.release(); L.d_set_beams.release(); L.d_set_order.release(); L.d_set_order2.release(); L.d_img_u8.release(); L.d_img_f32.release();
        L.d_hint.release(); L.d_ovf_list.release(); if (L.h_hist) { (void)hipHostFree(L.h_hist); L.h_hist = nullptr; }
        drop_graphs(L); L.d_poses.release();
        if (L.ev_ready) (void)hipEventDestroy(L.ev_ready);
        if (L.ev_consumed) (void)hipEventDestroy(L.ev_consumed);
        for (hipEvent_t e : L.ev_img) if (e) (void)hipEventDestroy(e);
        L.d_img_u8_b.release();
        for (Lane::CopyRec& r : L.rec) if (r.ev) (void)hipEventDestroy(r.ev);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_frame) (void)hipHostFree(c->h_frame);
    if (c->h_rb) (void)hipHostFree(c->h_rb);
    delete c;
}

const char* rr_last_error(const rr_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

namespace {

// The grazing guard's padding (traverse, rr_kernels.hip): 1e-5 x max(extent, largest |coordinate|) of the vertices the faces
// use -- half of what both builders pad their boxes with (2e-5 x the same measure + 1e-6; the GPU builder measures ALL
// vertices, which can only give more).  One multiplication: nothing a compiler could contract; the oracle forms the same value
float guard_pad(const float* verts, const uint32_t* faces, size_t nf)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < 3 * nf; i++) {
        const float* v = verts + 3 * (size_t)faces[i];
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
    }
    if (nf == 0) return 0.f;
    float ext = 0.f, mag = 0.f;
    for (int k = 0; k < 3; k++) { ext = std::max(ext, hi[k] - lo[k]); mag = std::max(mag, std::max(std::fabs(lo[k]), std::fabs(hi[k]))); }
    return 1e-5f * std::max(ext, mag);
}

// the finished host tree -> the ctx's one allocation (nodes, then triangles; references re-encoded as offsets)
int upload_tree(rr_ctx* c, const Bvh4& bvh)
{
    const size_t nn = bvh.nodes.size(), nt = bvh.tris.size();
    int rc = check_bvh_size(c, nn, nt); if (rc) return rc;
    // from here on the old tree is being overwritten: no mesh until the new one is complete (an error
    // return below leaves the context without a mesh, never with a half-written one)
    c->have_mesh = false;
    for (Lane& L : c->lanes) L.buf_seg = 0;
    c->tri_base4 = (uint32_t)(nn * 8);
    RR_HIP(c, c->d_bvh.ensure(nn * 8 + (nt + 4) * 3));   // +4 triangles: a quad may fetch past a short leaf
    RR_HIP(c, hipMemcpy(c->d_bvh.p, bvh.nodes.data(), nn * sizeof(Node4), hipMemcpyHostToDevice));
    if (nt) RR_HIP(c, hipMemcpy(c->d_bvh.p + c->tri_base4, bvh.tris.data(), nt * sizeof(TriRec), hipMemcpyHostToDevice));
    launch_encode_refs(reinterpret_cast<Node4*>(c->d_bvh.p), nn, c->tri_base4, nullptr, nt);
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipDeviceSynchronize());
    c->n_nodes = nn; c->n_tris = nt;
    c->depth = bvh.depth; c->stack_need = bvh.stack_need;
    c->have_mesh = true; c->hist_gen++; c->graph_gen++;
    for (Lane& L : c->lanes) L.buf_seg = 0;   // stack geometry may have changed
    return 0;
}

// traversal steps (node + leaf) the uploaded tree costs a fixed sample of radar-like rays: origins in the middle of the
// map's footprint and the lower half of its height, directions within +-5 degrees of horizontal (a radar's beam; reflections
// off walls stay level) -- a deterministic sample, the same for every candidate tree of a mesh
int measure_tree_steps(rr_ctx* c, const float lo[3], const float hi[3], double* steps_per_ray)
{
    const int n = 16384;
    std::vector<float> o(3 * (size_t)n), d(3 * (size_t)n);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto u01 = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) * (1.0 / 16777216.0)); };
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 2; k++) o[3 * i + k] = lo[k] + (0.25f + 0.5f * u01()) * (hi[k] - lo[k]);
        o[3 * i + 2] = lo[2] + (0.05f + 0.45f * u01()) * (hi[2] - lo[2]);
        const float yaw = 6.2831853f * u01(), el = (u01() - 0.5f) * 0.1745f;
        d[3 * i] = cosf(el) * cosf(yaw); d[3 * i + 1] = cosf(el) * sinf(yaw); d[3 * i + 2] = sinf(el);
    }
    const int stack_lds = (int)std::max<uint32_t>(1, std::min<uint32_t>(c->stack_need, (uint32_t)c->stack_lds_max));
    const int spill_depth = (int)c->stack_need - stack_lds;
    DevBuf<float> d_o, d_d; DevBuf<uint32_t> d_spill; DevBuf<unsigned long long> d_steps;
    hipError_t e = d_o.ensure(3 * (size_t)n);
    if (e == hipSuccess) e = d_d.ensure(3 * (size_t)n);
    if (e == hipSuccess) e = d_spill.ensure(spill_depth > 0 ? (size_t)spill_depth * n : 1);
    if (e == hipSuccess) e = d_steps.ensure(1);
    if (e == hipSuccess) e = hipMemcpy(d_o.p, o.data(), o.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d.p, d.data(), d.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_steps.p, 0, sizeof(unsigned long long));
    unsigned long long h = 0;
    if (e == hipSuccess) {
        Params P; std::memset(&P, 0, sizeof(P));
        P.nodes = reinterpret_cast<const Node4*>(c->d_bvh.p); P.tris = reinterpret_cast<const TriRec*>(c->d_bvh.p + c->tri_base4);
        P.tri_base4 = c->tri_base4; P.range_max = c->have_cfg ? c->cfg.range_max : 1000.0f; P.hit_pad = c->hit_pad;
        P.spill = d_spill.p; P.spill_stride = n; P.stack_lds = stack_lds; P.spill_depth = std::max(0, spill_depth);
        P.cull_pop = c->cull_pop;
        launch_debug_trace(P, d_o.p, d_d.p, n, nullptr, nullptr, c->stream, d_steps.p);
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(&h, d_steps.p, sizeof(h), hipMemcpyDeviceToHost);
    }
    d_o.release(); d_d.release(); d_spill.release(); d_steps.release();
    if (e != hipSuccess) return fail(c, -100, std::string("rr_set_mesh (tree choice): ") + hipGetErrorString(e));
    *steps_per_ray = (double)h / n;
    return 0;
}

}  // namespace

int rr_set_mesh(rr_ctx* c, const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                const uint32_t* face_object_id)
{
    if (!c) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    Bvh4 bvh; std::string err;
    // Which tree?  The default -- SAH over references with spatial splits and the vertical weight (rr_bvh.h) -- halves the
    // traversal steps of maps that mix 0.2 m terrain with 10 m building faces, but on a small regular mesh its few
    // spatial splits disturb the packing (the 100k-triangle heightfield of config 2: 12.3 steps per ray against 10.9 for
    // the plain SAH).  Images do not depend on the tree, so for meshes that build in a fraction of a second the choice
    // is MEASURED: the candidates are uploaded one after the other, each traces the same sample of radar-like rays, the
    // one with the fewest traversal steps stays.  RR_BVH_CHOOSE=0 (or any RR_BVH_ALPHA / _WZ experiment): default only.
    const bool choose = nf > 0 && nf <= (size_t)2000000 && !(getenv("RR_BVH_CHOOSE") && atoi(getenv("RR_BVH_CHOOSE")) == 0) &&
                        !getenv("RR_BVH_ALPHA") && !getenv("RR_BVH_WZ");
    // the builder allocates hundreds of MB and starts threads: whatever it throws (bad_alloc, system_error) stops here
    try {
    if (!build_bvh4(verts, nv, faces, nf, face_object_id, bvh, err)) return fail(c, -4, err);
    if (bvh.spatial_splits > 0 && bvh.nodes.size() * 8 + (bvh.tris.size() + 4) * 3 >= (1ull << 28)) {
        // the parts spatial splits add pushed the tree over the 28-bit reference range: build without them
        BvhOptions plain; plain.sbvh_alpha = -1.0f;
        if (!build_bvh4(verts, nv, faces, nf, face_object_id, bvh, err, 0, &plain)) return fail(c, -4, err);
    }
    // frames in flight on the lane streams or a caller's stream (all non-blocking: a blocking hipMemcpy
    // does not order against them) still trace the old tree
    RR_HIP(c, hipDeviceSynchronize());
    c->hit_pad = guard_pad(verts, faces, nf);       // (build_bvh4 has checked the indices)
    int rc = upload_tree(c, bvh); if (rc) return rc;
    if (choose) {
        float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
        for (size_t i = 0; i < nv; i++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], verts[3 * i + k]); hi[k] = std::max(hi[k], verts[3 * i + k]); }
        double best = 0.0;
        rc = measure_tree_steps(c, lo, hi, &best); if (rc) return rc;
        const bool verbose = getenv("RR_BVH_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "[rr bvh] tree choice: SAH + spatial splits, vertical weight 0.5: %.2f steps per sample ray\n", best);
        int kept = 0;
        for (int cand = 1; cand <= 2; cand++) {
            BvhOptions o; o.sbvh_alpha = -1.0f; o.vertical_weight = cand == 1 ? 0.5f : 1.0f;
            Bvh4 alt;
            if (!build_bvh4(verts, nv, faces, nf, face_object_id, alt, err, 0, &o)) continue;
            rc = upload_tree(c, alt); if (rc) return rc;
            double st = 0.0;
            rc = measure_tree_steps(c, lo, hi, &st); if (rc) return rc;
            if (verbose) fprintf(stderr, "[rr bvh] tree choice: plain SAH, vertical weight %.1f: %.2f steps per sample ray\n", o.vertical_weight, st);
            if (st < best * 0.98) { best = st; kept = cand; bvh = std::move(alt); }      // (2 %: do not trade trees over noise in the sample)
        }
        if (kept != 2) { rc = upload_tree(c, bvh); if (rc) return rc; }                  // the last candidate uploaded is not the winner
        if (verbose) fprintf(stderr, "[rr bvh] tree choice: kept candidate %d\n", kept);
    }
    } catch (const std::exception& ex) { c->have_mesh = false; return fail(c, -4, std::string("rr_set_mesh: host BVH build failed: ") + ex.what());
    } catch (...) { c->have_mesh = false; return fail(c, -4, "rr_set_mesh: host BVH build failed"); }
    return 0;
}

int rr_set_mesh_gpu(rr_ctx* c, const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                    const uint32_t* face_object_id)
{
    if (!c) return -1;
    if (nf == 0) return rr_set_mesh(c, verts, nv, faces, nf, face_object_id);
    if (!verts || !faces) return fail(c, -4, "rr_set_mesh_gpu: null vertex/face pointer");
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    Node4* dn = nullptr; TriRec* dt = nullptr; size_t nn = 0, nt = 0; uint32_t depth = 0, need = 0; float inflate = 0.f;
    std::string err;
    if (!build_bvh4_gpu(verts, nv, faces, nf, face_object_id, &dn, &nn, &dt, &nt, &depth, &need, &inflate, err, c->stream))
        return fail(c, -4, err);
    {
        // the builder hands over two arrays: move them into the one allocation the traversal addresses
        int rc = check_bvh_size(c, nn, nt);
        hipError_t e = hipSuccess;
        if (!rc) {
            c->have_mesh = false;           // see rr_set_mesh: no mesh while the tree is being replaced
            for (Lane& L : c->lanes) L.buf_seg = 0;
            c->tri_base4 = (uint32_t)(nn * 8);
            e = c->d_bvh.ensure(nn * 8 + (nt + 4) * 3);
            if (e == hipSuccess) e = hipMemcpy(c->d_bvh.p, dn, nn * sizeof(Node4), hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(c->d_bvh.p + c->tri_base4, dt, nt * sizeof(TriRec), hipMemcpyDeviceToDevice);
        }
        (void)hipFree(dn); (void)hipFree(dt);
        if (rc) return rc;
        RR_HIP(c, e);
        launch_encode_refs(reinterpret_cast<Node4*>(c->d_bvh.p), nn, c->tri_base4, nullptr, nt);
        RR_HIP(c, hipGetLastError());
        RR_HIP(c, hipDeviceSynchronize());
    }
    c->n_nodes = nn; c->n_tris = nt; c->depth = depth; c->stack_need = need;
    c->hit_pad = guard_pad(verts, faces, nf);
    c->have_mesh = true; c->hist_gen++; c->graph_gen++;
    for (Lane& L : c->lanes) L.buf_seg = 0;
    return 0;
}

int rr_copy_mesh(rr_ctx* c, rr_ctx* src)
{
    if (!c) return -1;
    if (!src || src == c) return fail(c, -3, "rr_copy_mesh: need another context as the source");
    if (!src->have_mesh) return fail(c, -2, "rr_copy_mesh: the source context has no mesh");
    // nothing may still trace the old tree here, nothing may still write the source's
    RR_HIP(c, hipSetDevice(src->device));
    RR_HIP(c, hipDeviceSynchronize());
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    const size_t n4 = (size_t)src->n_nodes * 8 + ((size_t)src->n_tris + 4) * 3;     // float4 records, as rr_set_mesh sizes them
    c->have_mesh = false;
    for (Lane& L : c->lanes) L.buf_seg = 0;
    RR_HIP(c, c->d_bvh.ensure(n4));
    // child references are offsets from the base of the allocation: the tree is position independent
    if (src->device == c->device) RR_HIP(c, hipMemcpy(c->d_bvh.p, src->d_bvh.p, n4 * sizeof(float4), hipMemcpyDeviceToDevice));
    else RR_HIP(c, hipMemcpyPeer(c->d_bvh.p, c->device, src->d_bvh.p, src->device, n4 * sizeof(float4)));
    RR_HIP(c, hipDeviceSynchronize());
    c->tri_base4 = src->tri_base4; c->n_nodes = src->n_nodes; c->n_tris = src->n_tris;
    c->depth = src->depth; c->stack_need = src->stack_need; c->hit_pad = src->hit_pad;
    c->have_mesh = true; c->hist_gen++; c->graph_gen++;
    return 0;
}

int rr_set_materials(rr_ctx* c, const rr_material* materials, size_t n_materials,
                     const int32_t* object_materials, size_t n_objects, int32_t material_id_air)
{
    if (!c) return -1;
    if (!materials || n_materials == 0) return fail(c, -3, "rr_set_materials: empty material table");
    if (n_objects && !object_materials) return fail(c, -3, "rr_set_materials: null object_materials");
    if (material_id_air < 0 || (size_t)material_id_air >= n_materials) return fail(c, -3, "rr_set_materials: material_id_air out of range");
    for (size_t i = 0; i < n_objects; i++)
        if (object_materials[i] < 0 || (size_t)object_materials[i] >= n_materials)
            return fail(c, -3, "rr_set_materials: object_materials entry out of range");
    if (c->have_materials && c->materials.size() == n_materials && c->object_materials.size() == n_objects &&
        c->material_id_air == material_id_air &&
        std::memcmp(c->materials.data(), materials, n_materials * sizeof(rr_material)) == 0 &&
        (n_objects == 0 || std::memcmp(c->object_materials.data(), object_materials, n_objects * sizeof(int32_t)) == 0))
        return 0;   // the per-frame loadParams() of the reference node, nothing new
    c->materials.assign(materials, materials + n_materials);
    c->object_materials.assign(object_materials, object_materials + n_objects);
    c->material_id_air = material_id_air;
    c->have_materials = true; c->tables_dirty |= rr_ctx::D_MAT;
    return 0;
}

int rr_set_config(rr_ctx* c, const rr_config* cfg)
{
    if (!c) return -1;
    if (!cfg) return fail(c, -3, "rr_set_config: null config");
    if (cfg->n_cells < 1 || cfg->n_cells > 8192) return fail(c, -3, "rr_set_config: n_cells must be in [1, 8192]");
    if (cfg->n_angles < 1 || cfg->n_angles > 65536) return fail(c, -3, "rr_set_config: n_angles must be in [1, 65536]");
    if (cfg->n_reflections < 0 || cfg->n_reflections > 16) return fail(c, -3, "rr_set_config: n_reflections must be in [0, 16]");
    if (cfg->signal_denoising < 0 || cfg->signal_denoising > 3) return fail(c, -3, "rr_set_config: signal_denoising must be 0..3");
    const int w = cfg->signal_denoising == 1 ? cfg->signal_denoising_triangular_width
                : cfg->signal_denoising == 2 ? cfg->signal_denoising_gaussian_width
                : cfg->signal_denoising == 3 ? cfg->signal_denoising_mb_width : 0;
    if (w < 0 || w > 256) return fail(c, -3, "rr_set_config: smear width must be in [0, 256]");
    if (cfg->ambient_noise < 0 || cfg->ambient_noise > 2) return fail(c, -3, "rr_set_config: ambient_noise must be 0..2");
    if (cfg->brdf_model < 0 || cfg->brdf_model > 1) return fail(c, -3, "rr_set_config: brdf_model must be 0 (A + B cos^C) or 1 (Cook-Torrance lobe)");
    if (!(cfg->resolution > 0.0)) return fail(c, -3, "rr_set_config: resolution must be > 0");
    // mode = (int)(fraction * width) indexes the weight table (RadarCPU.cpp:48-93): the reference's
    // sliders keep the fraction in [0, 1) (cfg/RadarModel.cfg:47-51); anything else would read outside it.
    // fraction * width < 1 (mode 0) is accepted and gives the reference's 0/0 weights
    const double mf = cfg->signal_denoising == 1 ? cfg->signal_denoising_triangular_mode
                    : cfg->signal_denoising == 2 ? cfg->signal_denoising_gaussian_mode
                    : cfg->signal_denoising == 3 ? cfg->signal_denoising_mb_mode : 0.0;
    if (!(mf >= 0.0 && mf < 1.0)) return fail(c, -3, "rr_set_config: denoising mode fraction must be in [0, 1)");
    // tfar of the ray cast; must stay far below the coordinate that marks an empty BVH child (3e38)
    if (!(cfg->range_max > 0.0f && cfg->range_max <= 1.0e30f)) return fail(c, -3, "rr_set_config: range_max must be in (0, 1e30]");
    if (!std::isfinite(cfg->wave_energy_threshold) || !std::isfinite(cfg->theta_min) || !std::isfinite(cfg->theta_inc))
        return fail(c, -3, "rr_set_config: non-finite wave_energy_threshold / theta_min / theta_inc");
    if (c->have_cfg && std::memcmp(&c->cfg, cfg, sizeof(rr_config)) == 0) return 0;
    c->cfg = *cfg;
    c->have_cfg = true; c->tables_dirty |= rr_ctx::D_CFG;
    return 0;
}

int rr_set_beam_samples(rr_ctx* c, const float* dirs, size_t n)
{
    if (!c) return -1;
    if (n && !dirs) return fail(c, -3, "rr_set_beam_samples: null dirs");
    if (n > 65536) return fail(c, -3, "rr_set_beam_samples: more than 65536 samples");
    for (size_t i = 0; i < 3 * n; i++) if (!std::isfinite(dirs[i])) return fail(c, -3, "rr_set_beam_samples: non-finite direction");
    if (c->beams.size() == 3 * n && (n == 0 || std::memcmp(c->beams.data(), dirs, 3 * n * sizeof(float)) == 0)) return 0;
    c->beams.assign(dirs, dirs + 3 * n);
    c->tables_dirty |= rr_ctx::D_BEAMS;
    return 0;
}

int rr_set_noise_offsets(rr_ctx* c, const float* rnd, size_t n)
{
    if (!c) return -1;
    if (n && !rnd) return fail(c, -3, "rr_set_noise_offsets: null pointer");
    for (size_t i = 0; i < n; i++) if (!std::isfinite(rnd[i])) return fail(c, -3, "rr_set_noise_offsets: non-finite offset");
    c->noise.assign(rnd, rnd + n);
    c->tables_dirty |= rr_ctx::D_NOISE;
    return 0;
}

int rr_set_motion_poses(rr_ctx* c, const float* poses, size_t n)
{
    if (!c) return -1;
    if (n && !poses) return fail(c, -3, "rr_set_motion_poses: null pointer");
    for (size_t i = 0; i < 7 * n; i++) if (!std::isfinite(poses[i])) return fail(c, -3, "rr_set_motion_poses: non-finite pose");
    c->motion.assign(poses, poses + 7 * n);
    c->tables_dirty |= rr_ctx::D_MOTION;
    return 0;
}

int rr_simulate_columns_device(rr_ctx* c, const float pose[7], int az_begin, int az_end,
                               uint8_t* d_cols_u8, float* d_cols_f32, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!pose || !d_cols_u8) return fail(c, -3, "rr_simulate_columns_device: null pose/output");
    RR_HIP(c, hipSetDevice(c->device));
    // rotate over the frame lanes so that calls issued on DIFFERENT streams (pipelined
    // multi-GPU slots) can overlap; a lane is reused only after its previous frame finished
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = li;
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));
    rc = run_frame(c, L, pose, az_begin, az_end, d_cols_u8, d_cols_f32, s); if (rc) return rc;
    RR_HIP(c, hipEventRecord(L.ev_consumed, s));
    L.pending_consume = true;
    return 0;
}

int rr_simulate_batch_columns_device(rr_ctx* c, const float* poses, int n_frames, int az_begin, int az_end,
                                     uint8_t* d_cols_u8, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!poses || !d_cols_u8) return fail(c, -3, "rr_simulate_batch_columns_device: null poses/output");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = li;
    { HostProfScope hp(0, "ctx: wait for the lane's event");
      if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0)); }
    { HostProfScope hp(1, "ctx: run_frame");
      rc = run_frame(c, L, poses, az_begin, az_end, d_cols_u8, nullptr, s, n_frames); if (rc) return rc; }
    { HostProfScope hp(2, "ctx: record the lane's event");
      RR_HIP(c, hipEventRecord(L.ev_consumed, s)); }
    L.pending_consume = true;
    return 0;
}

int rr_simulate_batch_columns_carry_device(rr_ctx* c, const float* poses, int n_frames, int az_begin, int az_end,
                                           uint8_t* d_cols_u8, void* stream, const void* d_carry_src, void* h_carry_dst, size_t carry_bytes)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!poses || !d_cols_u8) return fail(c, -3, "rr_simulate_batch_columns_carry_device: null poses/output");
    if (carry_bytes && (!d_carry_src || !h_carry_dst)) return fail(c, -3, "rr_simulate_batch_columns_carry_device: null carry pointer");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    rc = upload_tables(c); if (rc) return rc;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }
    c->last_lane = li;
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));
    // the carried copy rides on the later-pass trace launches (a few waves, one 1-KB store per wave in flight: Params::copy_src)
    // when there are such launches and the destination is page-locked; else it is a plain copy ahead of the batch
    const bool visible = carry_bytes > 0 && host_visible(h_carry_dst);
    const bool fold = visible && c->copy_blocks > 0 && eff_config(c).n_reflections >= 2 && carry_bytes % 16 == 0 && !c->stats_mode &&
                      ((uintptr_t)d_carry_src | (uintptr_t)h_carry_dst) % 16 == 0;
    if (carry_bytes && !fold) { const int rcc = copy_out(c, d_carry_src, h_carry_dst, carry_bytes, visible, s); if (rcc) return rcc; }
    rc = run_frame(c, L, poses, az_begin, az_end, d_cols_u8, nullptr, s, n_frames, nullptr, 0, false,
                   fold ? (const uint8_t*)d_carry_src : nullptr, fold ? (uint8_t*)h_carry_dst : nullptr, fold ? carry_bytes : 0);
    if (rc) {
        if (fold) (void)copy_out(c, d_carry_src, h_carry_dst, carry_bytes, true, s);    // refused before any launch: the copy still happens
        return rc;
    }
    RR_HIP(c, hipEventRecord(L.ev_consumed, s));
    L.pending_consume = true;
    return 0;
}

int rr_simulate_batch_device(rr_ctx* c, const float* poses, int n_frames, uint8_t* d_imgs_u8, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!poses || !d_imgs_u8) return fail(c, -3, "rr_simulate_batch_device: null poses/output");
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_batch_device: n_frames must be 1..64");
    RR_HIP(c, hipSetDevice(c->device));
    const rr_config& g = c->cfg;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    rc = upload_tables(c); if (rc) return rc;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = li;
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));
    rc = run_frame(c, L, poses, 0, g.n_angles, nullptr, nullptr, s, n_frames); if (rc) return rc;
    { TimedScope t(c, s, "assemble");
      launch_assemble_u8(L.d_cols_u8.p, d_imgs_u8, g.n_angles, g.n_cells, g.scroll_image, s, g.n_angles,
                         (size_t)g.n_angles * g.n_cells, n_frames, (size_t)g.n_angles * g.n_cells); }
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipEventRecord(L.ev_consumed, s));
    L.pending_consume = true;
    return 0;
}

void* rr_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void rr_host_free(void* p) { if (p) (void)hipHostFree(p); }

namespace {
// the SDMA worker of the context, made on first use (nullptr: not available / switched off)
SdmaCopier* sdma_of(rr_ctx* c, const void* any_device_ptr)
{
    if (!c->host_sdma) return nullptr;
    if (!c->sdma && !c->sdma_tried) {
        c->sdma_tried = true;
        std::string why;
        c->sdma = sdma_create(c->device, any_device_ptr, why);
        if (!c->sdma && getenv("RR_HOST_SDMA_VERBOSE")) fprintf(stderr, "[rr] SDMA delivery not available: %s\n", why.c_str());
    }
    if (c->sdma && sdma_failed(c->sdma, nullptr)) {
        if (getenv("RR_HOST_SDMA_VERBOSE")) { std::string why; (void)sdma_failed(c->sdma, &why); fprintf(stderr, "[rr] SDMA delivery switched off: %s\n", why.c_str()); }
        c->host_sdma = 0;
        return nullptr;
    }
    return c->sdma;
}
}  // namespace

int rr_deliver_to_host_async(rr_ctx* c, const void* d_src, void* h_dst, size_t bytes, void* stream)
{
    if (!c) return -1;
    if (bytes == 0) return 0;
    if (!d_src || !h_dst) return fail(c, -3, "rr_deliver_to_host_async: null pointer");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    hipEvent_t ev = nullptr;
    if (!c->delivery_events.empty()) { ev = c->delivery_events.back(); c->delivery_events.pop_back(); }
    else RR_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const bool visible = host_visible(h_dst);
    SdmaCopier* sd = visible ? sdma_of(c, d_src) : nullptr;
    uint64_t job = 0;
    if (sd) {
        const hipError_t e = hipEventRecord(ev, s);        // the copy starts once the stream has got here
        if (e != hipSuccess) { c->delivery_events.push_back(ev); RR_HIP(c, e); }
        job = sdma_submit(sd, ev, d_src, h_dst, bytes);
    } else {
        const int rc = copy_out(c, d_src, h_dst, bytes, visible, s);
        if (rc) { c->delivery_events.push_back(ev); return rc; }
        const hipError_t e = hipEventRecord(ev, s);        // ... is complete once the stream has got here
        if (e != hipSuccess) { c->delivery_events.push_back(ev); RR_HIP(c, e); }
    }
    c->deliveries.push_back({ h_dst, job, ev });
    return 0;
}

int rr_host_delivery_route(rr_ctx* c)
{
    if (!c) return -1;
    if (c->sdma && c->host_sdma && !sdma_failed(c->sdma, nullptr)) return 2;      // SDMA through ROCr: in use
    if (c->host_sdma && !c->sdma_tried) return 1;                                  // ... will be tried by the first delivery
    return 0;                                                                      // stream-ordered copies (deferred / trickled / copy kernel)
}

int rr_copy_to_host_async(rr_ctx* c, const void* d_src, void* h_dst, size_t bytes, void* stream)
{
    if (!c) return -1;
    if (bytes && (!d_src || !h_dst)) return fail(c, -3, "rr_copy_to_host_async: null pointer");
    RR_HIP(c, hipSetDevice(c->device));
    return copy_out(c, d_src, h_dst, bytes, bytes > 0 && host_visible(h_dst), stream ? (hipStream_t)stream : c->stream);
}

int rr_simulate_batch_host_async(rr_ctx* c, const float* poses, int n_frames, uint8_t* h_imgs_u8, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!poses || !h_imgs_u8) return fail(c, -3, "rr_simulate_batch_host_async: null poses/output");
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_batch_host_async: n_frames must be 1..64");
    RR_HIP(c, hipSetDevice(c->device));
    const rr_config& g = c->cfg;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    rc = upload_tables(c); if (rc) return rc;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    c->last_lane = li;
    const size_t bytes = (size_t)n_frames * g.n_cells * g.n_angles;
    // The default route: over the SDMA engines through ROCr, at once, behind this batch's assemble -- no shader core stores a
    // byte of it, so nothing has to be deferred or trickled, and it is the same engine under every HIP runtime
    const bool device_visible = host_visible(h_imgs_u8);
    if (SdmaCopier* sd = (device_visible && !c->stats_mode) ? sdma_of(c, c->d_bvh.p) : nullptr) {
        rc = flush_deferred(c, L, false); if (rc) return rc;     // (images an earlier batch left on the lane by the other route)
        const int b = L.img_flip;
        settle_sdma(c, L, b);                                   // the job that empties THIS buffer: two uses of the lane ago
        DevBuf<uint8_t>& img = b ? L.d_img_u8_b : L.d_img_u8;
        if (img.n < bytes) {
            settle_sdma(c, L);
            RR_HIP(c, hipDeviceSynchronize());
            RR_HIP(c, img.ensure(bytes));
        }
        if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));   // the lane's previous batch (its frame buffers)
        rc = run_frame(c, L, poses, 0, g.n_angles, nullptr, nullptr, s, n_frames); if (rc) return rc;
        { TimedScope t(c, s, "assemble");
          launch_assemble_u8(L.d_cols_u8.p, img.p, g.n_angles, g.n_cells, g.scroll_image, s, g.n_angles,
                             (size_t)g.n_angles * g.n_cells, n_frames, (size_t)g.n_angles * g.n_cells); }
        RR_HIP(c, hipGetLastError());
        RR_HIP(c, hipEventRecord(L.ev_consumed, s));
        L.pending_consume = true;
        RR_HIP(c, hipEventRecord(L.ev_img[b], s));
        L.sdma_job[b] = sdma_submit(sd, L.ev_img[b], img.p, h_imgs_u8, bytes);
        L.sdma_dst[b] = h_imgs_u8;
        L.img_flip ^= 1;
        return 0;
    }
    settle_sdma(c, L);
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));   // the lane's previous batch, incl. its assemble
    // the images the lane's previous batch left behind ride on this batch's later-pass launches when possible
    // ... and worthwhile: the trickle (one 1-KB store per wave in flight, a few waves) needs about 1 ms per launch for 8
    // images, which hides behind a launch only while other batches share the GPU with it; a caller with a single batch
    // in flight gets the plain copy
    int busy = 0;
    for (Lane& M : c->lanes) if (&M != &L && M.pending_consume && hipEventQuery(M.ev_consumed) == hipErrorNotReady) busy++;
    (void)hipGetLastError();
    const bool fold = L.deferred && L.def_foldable && c->copy_blocks > 0 && g.n_reflections >= 2 && L.def_bytes % 16 == 0 &&
                      L.d_img_u8.n >= bytes && !c->stats_mode && busy >= c->fold_min_busy;
    const uint8_t* job_src = nullptr; uint8_t* job_dst = nullptr; size_t job_bytes = 0;
    Lane::CopyRec* rec = nullptr;
    if (fold) {
        rc = take_rec(c, L, &rec); if (rc) return rc;
        job_src = L.d_img_u8.p; job_dst = L.def_dst; job_bytes = L.def_bytes; L.deferred = false;
    }
    else { rc = flush_deferred(c, L); if (rc) return rc; if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0)); }
    if (L.d_img_u8.n < bytes) {
        RR_HIP(c, hipDeviceSynchronize());      // an earlier copy may still read the old buffer
        RR_HIP(c, L.d_img_u8.ensure(bytes));
    }
    rc = run_frame(c, L, poses, 0, g.n_angles, nullptr, nullptr, s, n_frames, nullptr, 0, false, job_src, job_dst, job_bytes);
    if (rc) {
        if (fold) {     // the frame was refused before any launch: the folded copy still has to happen
            (void)copy_out(c, job_src, job_dst, job_bytes, true, s);
            RR_HIP(c, hipEventRecord(rec->ev, s)); rec->dst = job_dst; rec->pending = true;
        }
        return rc;
    }
    if (fold) { RR_HIP(c, hipEventRecord(rec->ev, s)); rec->dst = job_dst; rec->pending = true; }   // behind the launches that carried it
    { TimedScope t(c, s, "assemble");
      launch_assemble_u8(L.d_cols_u8.p, L.d_img_u8.p, g.n_angles, g.n_cells, g.scroll_image, s, g.n_angles,
                         (size_t)g.n_angles * g.n_cells, n_frames, (size_t)g.n_angles * g.n_cells); }
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipEventRecord(L.ev_consumed, s));
    L.pending_consume = true;
    // Where do the images go from here?  A copy issued behind the batch costs the frame rate about the PCIe transfer
    // time of the images, whoever stores the bytes (tools/probe_hostpath.py, 10M-triangle target, 8 poses per batch, 4
    // streams; 3,950-4,050 images/s with the images left in HBM): hipMemcpyAsync on this stream 3,670-3,775; a separate
    // copy stream 3,250-3,840 (a fifth stream shares a hardware queue with a batch stream); the assemble kernel writing
    // straight into the host buffer 3,515; an own copy kernel of 4..128 workgroups 3,590-3,650; the copy folded into a
    // trace launch with all its stores in flight 3,580-3,650 -- while copies of 1 MB per batch cost nothing
    // (tools/probe_fence.py).  What stalls is the memory pipeline: stores to host memory drain at PCIe speed, and once
    // they fill its write queues the stores of every other kernel wait behind them.  So the copy is DEFERRED to the lane's
    // next batch and trickled out by a few waves of its later-pass trace launches with ONE store per wave in flight
    // (k_trace, Params::copy_src): 3,980-4,025 images/s, within 1 % of the HBM-resident rate.
    if (c->host_copy_stream && g.n_reflections < 2) {
        // nothing later could carry these images: out they go now, on the copy stream, behind this batch's assemble; the lane's
        // next batch waits for the copy (device side) before it touches the lane
        if (!c->copy_stream) RR_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        Lane::CopyRec* r = nullptr;
        rc = take_rec(c, L, &r); if (rc) return rc;
        RR_HIP(c, hipStreamWaitEvent(c->copy_stream, L.ev_consumed, 0));
        rc = copy_out(c, L.d_img_u8.p, h_imgs_u8, bytes, device_visible, c->copy_stream); if (rc) return rc;
        RR_HIP(c, hipEventRecord(r->ev, c->copy_stream));
        r->dst = h_imgs_u8; r->pending = true;
        RR_HIP(c, hipEventRecord(L.ev_consumed, c->copy_stream));      // what the lane's next user waits for
        return 0;
    }
    L.deferred = true; L.def_dst = h_imgs_u8; L.def_bytes = bytes; L.def_stream = s; L.def_foldable = device_visible;
    return 0;
}

int rr_wait_host(rr_ctx* c, const void* h_imgs_u8)
{
    if (!c) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    // oldest batch first (lanes are handed out round robin: the next one to be used holds the oldest batch): its images leave
    // while the younger batches still render, and only the youngest batch's copy is left when the kernels are done -- in lane
    // order the youngest batch may come first, and the copies of all the others then queue up behind the end of the run
    for (size_t i = 0; i < c->deliveries.size();) {       // rr_deliver_to_host_async's copies
        rr_ctx::Delivery& d = c->deliveries[i];
        if (h_imgs_u8 != nullptr && d.dst != h_imgs_u8) { i++; continue; }
        if (d.job && c->sdma) sdma_wait(c->sdma, d.job);
        else RR_HIP(c, hipEventSynchronize(d.ev));
        c->delivery_events.push_back(d.ev);
        c->deliveries.erase(c->deliveries.begin() + (long)i);
    }
    const size_t nl = c->lanes.size();
    for (size_t k = 0; k < nl; k++) {
        Lane& L = c->lanes[(c->next_lane + k) % nl];
        settle_sdma(c, L, -1, h_imgs_u8);
        if (L.deferred && (h_imgs_u8 == nullptr || L.def_dst == h_imgs_u8)) { int rc = flush_deferred(c, L); if (rc) return rc; }
        for (Lane::CopyRec& r : L.rec)
            if (r.pending && (h_imgs_u8 == nullptr || r.dst == h_imgs_u8)) {
                RR_HIP(c, hipEventSynchronize(r.ev));
                r.pending = false; r.dst = nullptr;
            }
    }
    return 0;
}

int rr_simulate_param_sets_device(rr_ctx* c, const float pose[7], const rr_param_set* sets, int n_sets, size_t n_materials,
                                  uint8_t* d_imgs_u8, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!pose || !sets || !d_imgs_u8) return fail(c, -3, "rr_simulate_param_sets_device: null pose/sets/output");
    if (n_sets < 1 || n_sets > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_param_sets_device: n_sets must be 1..64");
    const size_t n_mat = c->materials.size();
    if (n_materials != n_mat)
        return fail(c, -3, "rr_simulate_param_sets_device: every set must hold as many materials as the table given to rr_set_materials");
    const size_t nb = c->beams.size() / 3;
    if (nb == 0) return fail(c, -2, "rr_set_beam_samples has not been called");
    int p_max = 0;
    SetPlan plan; plan.n_groups = 0;
    std::vector<const float*> group_dirs;         // beam table of each group (null: the ctx's own samples)
    for (int k = 0; k < n_sets; k++) {
        const rr_param_set& S = sets[k];
        const int np = S.n_reflections < 0 ? c->cfg.n_reflections : S.n_reflections;
        if (np > 16) return fail(c, -3, "rr_simulate_param_sets_device: n_reflections must be <= 16 (negative: the config's)");
        p_max = std::max(p_max, np);
        plan.frame_passes[k] = (unsigned char)np;
        if (S.materials)
            for (size_t i = 0; i < n_mat; i++)
                if (!std::isfinite(S.materials[i].velocity) || !std::isfinite(S.materials[i].ambient) || !std::isfinite(S.materials[i].diffuse) ||
                    !std::isfinite(S.materials[i].specular))
                    return fail(c, -3, "rr_simulate_param_sets_device: non-finite material parameter");
        if (S.beam_dirs) for (size_t i = 0; i < 3 * nb; i++) if (!std::isfinite(S.beam_dirs[i])) return fail(c, -3, "rr_simulate_param_sets_device: non-finite beam direction");
        // sets with the same directions (the same pointer, or the same bytes) form a group and share pass 0
        const float* dirs = S.beam_dirs;
        if (dirs && std::memcmp(dirs, c->beams.data(), 3 * nb * sizeof(float)) == 0) dirs = nullptr;
        int g = -1;
        for (int j = 0; j < plan.n_groups && g < 0; j++) {
            const float* o = group_dirs[(size_t)j];
            if (o == dirs || (o && dirs && std::memcmp(o, dirs, 3 * nb * sizeof(float)) == 0)) g = j;
        }
        if (g < 0) { g = plan.n_groups++; group_dirs.push_back(dirs); plan.group_frame[g] = (unsigned char)k; }
        plan.frame_beam[k] = (unsigned char)g;
    }
    RR_HIP(c, hipSetDevice(c->device));
    const rr_config& g0 = c->cfg;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    rc = upload_tables(c); if (rc) return rc;
    const size_t li = c->next_lane++ % c->lanes.size();
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = li;
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(s, L.ev_consumed, 0));
    static_assert(sizeof(rr_material) == sizeof(float4), "rr_material is {velocity, ambient, diffuse, specular}");
    const size_t G = (size_t)plan.n_groups;
    const bool own_beams = !(G == 1 && group_dirs[0] == nullptr);
    if (L.d_matsets.n < (size_t)n_sets * n_mat || (own_beams && L.d_set_beams.n < G * nb)) {
        RR_HIP(c, hipDeviceSynchronize());      // the tables of this lane may still be read by an earlier step
        RR_HIP(c, L.d_matsets.ensure((size_t)n_sets * n_mat));
        RR_HIP(c, L.d_matset_limits.ensure((size_t)n_sets * n_mat));
        if (own_beams) { RR_HIP(c, L.d_set_beams.ensure(G * nb)); RR_HIP(c, L.d_set_order.ensure(G * nb)); RR_HIP(c, L.d_set_order2.ensure(G * nb)); }
    }
    // the lane's previous use of these host arrays: its copies were enqueued on a stream this stream now waits for
    // (ev_consumed), but a staged copy reads the host side at an unknown time: wait for the lane's last batch before reuse
    if (L.pending_consume) RR_HIP(c, hipEventSynchronize(L.ev_consumed));
    L.h_matsets.resize((size_t)n_sets * n_mat);
    for (int k = 0; k < n_sets; k++) {
        const rr_material* m = sets[k].materials ? sets[k].materials : c->materials.data();
        for (size_t i = 0; i < n_mat; i++) L.h_matsets[(size_t)k * n_mat + i] = make_float4(m[i].velocity, m[i].ambient, m[i].diffuse, m[i].specular);
    }
    if (own_beams) {
        L.h_set_beams.resize(G * nb); L.h_set_order.resize(G * nb); L.h_set_order2.resize(G * nb);
        std::vector<uint32_t> o1, o2;
        for (size_t gi = 0; gi < G; gi++) {
            const float* d = group_dirs[gi] ? group_dirs[gi] : c->beams.data();
            for (size_t i = 0; i < nb; i++) L.h_set_beams[gi * nb + i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], 0.0f);
            beam_trace_orders(d, nb, o1, o2);
            std::copy(o1.begin(), o1.end(), L.h_set_order.begin() + (std::ptrdiff_t)(gi * nb));
            std::copy(o2.begin(), o2.end(), L.h_set_order2.begin() + (std::ptrdiff_t)(gi * nb));
        }
        plan.d_beams = L.d_set_beams.p; plan.d_order = L.d_set_order.p; plan.d_order2 = L.d_set_order2.p;
    }
    // sizes follow the largest number of passes of the batch
    c->passes_override = p_max;
    rc = prepare_lane(c, L, n_sets * g0.n_angles);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(L.d_matsets.p, L.h_matsets.data(), L.h_matsets.size() * sizeof(float4), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && own_beams) e = hipMemcpyAsync(L.d_set_beams.p, L.h_set_beams.data(), G * nb * sizeof(float4), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && own_beams) e = hipMemcpyAsync(L.d_set_order.p, L.h_set_order.data(), G * nb * sizeof(uint32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && own_beams) e = hipMemcpyAsync(L.d_set_order2.p, L.h_set_order2.data(), G * nb * sizeof(uint32_t), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) rc = fail(c, -100, std::string("rr_simulate_param_sets_device: ") + hipGetErrorString(e));
    }
    if (!rc) {
        launch_mat_limits(L.d_matsets.p, (size_t)n_sets * n_mat, L.d_matset_limits.p, s);
        rc = run_frame(c, L, pose, 0, g0.n_angles, nullptr, nullptr, s, n_sets, L.d_matsets.p, (int)n_mat, false, nullptr, nullptr, 0, &plan);
    }
    c->passes_override = -1;
    if (rc) {
        // staged copies from the lane's host vectors may already be enqueued (advisor, round 4): the next call on this lane
        // must not rewrite them underneath -- it waits for ev_consumed like after a complete batch
        (void)hipEventRecord(L.ev_consumed, s);
        L.pending_consume = true;
        return rc;
    }
    { TimedScope t(c, s, "assemble");
      launch_assemble_u8(L.d_cols_u8.p, d_imgs_u8, g0.n_angles, g0.n_cells, g0.scroll_image, s, g0.n_angles,
                         (size_t)g0.n_angles * g0.n_cells, n_sets, (size_t)g0.n_angles * g0.n_cells); }
    RR_HIP(c, hipGetLastError());
    RR_HIP(c, hipEventRecord(L.ev_consumed, s));
    L.pending_consume = true;
    return 0;
}

int rr_simulate_material_sets_device(rr_ctx* c, const float pose[7], const rr_material* sets, int n_sets,
                                     size_t n_materials, uint8_t* d_imgs_u8, void* stream)
{
    if (!c) return -1;
    if (!pose || !sets || !d_imgs_u8) return fail(c, -3, "rr_simulate_material_sets_device: null pose/sets/output");
    if (n_sets < 1 || n_sets > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_material_sets_device: n_sets must be 1..64");
    if (n_materials != c->materials.size())
        return fail(c, -3, "rr_simulate_material_sets_device: every set must hold as many materials as the table given to rr_set_materials");
    // the parameter batch with only the material tables varying: one beam group, the config's passes
    rr_param_set ps[RR_MAX_BATCH];
    for (int k = 0; k < n_sets; k++) { ps[k].materials = sets + (size_t)k * n_materials; ps[k].beam_dirs = nullptr; ps[k].n_reflections = -1; ps[k].reserved_ = 0; }
    return rr_simulate_param_sets_device(c, pose, ps, n_sets, n_materials, d_imgs_u8, stream);
}

namespace {
// the images of a parameter batch in c->d_param_imgs: copy out and / or score, report the frame's error bits
int finish_param_batch(rr_ctx* c, int n_sets, uint8_t* out_imgs_u8, const uint8_t* ref_img_u8, double* out_psnr)
{
    const size_t npx = (size_t)c->cfg.n_cells * c->cfg.n_angles;
    if (out_imgs_u8) RR_HIP(c, hipMemcpyAsync(out_imgs_u8, c->d_param_imgs.p, (size_t)n_sets * npx, hipMemcpyDeviceToHost, c->stream));
    if (ref_img_u8 && out_psnr) {
        RR_HIP(c, c->d_ref_img.ensure(npx));
        RR_HIP(c, hipMemcpyAsync(c->d_ref_img.p, ref_img_u8, npx, hipMemcpyHostToDevice, c->stream));
        const int rc = rr_score_images_device(c, c->d_param_imgs.p, n_sets, c->d_ref_img.p, out_psnr, nullptr, c->stream);   // synchronises the stream
        if (rc) return rc;
    }
    RR_HIP(c, hipStreamSynchronize(c->stream));
    Counters h;
    { const int rcb = read_back(c, &h, c->lanes[c->last_lane].d_counters.p, sizeof(h)); if (rcb) return rcb; }
    if (h.overflow) RR_HIP(c, hipMemset(c->lanes[c->last_lane].d_sticky.p, 0, sizeof(uint32_t)));
    if (h.overflow & 1u) return fail(c, -7, "wave/signal queue capacity exceeded; raise rr_config.max_waves_per_azimuth");
    if (h.overflow & 2u) return fail(c, -8, "object id or material id out of range of the tables given to rr_set_materials");
    return 0;
}
}  // namespace

int rr_simulate_material_sets(rr_ctx* c, const float pose[7], const rr_material* sets, int n_sets, size_t n_materials,
                              uint8_t* out_imgs_u8)
{
    if (!c) return -1;
    if (!out_imgs_u8) return fail(c, -3, "rr_simulate_material_sets: null output");
    if (n_sets < 1 || n_sets > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_material_sets: n_sets must be 1..64");
    RR_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n_sets * c->cfg.n_cells * c->cfg.n_angles;
    RR_HIP(c, c->d_param_imgs.ensure(bytes));
    int rc = rr_simulate_material_sets_device(c, pose, sets, n_sets, n_materials, c->d_param_imgs.p, c->stream); if (rc) return rc;
    return finish_param_batch(c, n_sets, out_imgs_u8, nullptr, nullptr);
}

int rr_simulate_param_sets(rr_ctx* c, const float pose[7], const rr_param_set* sets, int n_sets, size_t n_materials,
                           uint8_t* out_imgs_u8, const uint8_t* ref_img_u8, double* out_psnr)
{
    if (!c) return -1;
    if (!out_imgs_u8 && !(ref_img_u8 && out_psnr)) return fail(c, -3, "rr_simulate_param_sets: neither an image buffer nor a reference image + score buffer");
    if ((ref_img_u8 == nullptr) != (out_psnr == nullptr)) return fail(c, -3, "rr_simulate_param_sets: ref_img_u8 and out_psnr go together");
    if (n_sets < 1 || n_sets > RR_MAX_BATCH) return fail(c, -3, "rr_simulate_param_sets: n_sets must be 1..64");
    RR_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n_sets * c->cfg.n_cells * c->cfg.n_angles;
    RR_HIP(c, c->d_param_imgs.ensure(bytes));
    int rc = rr_simulate_param_sets_device(c, pose, sets, n_sets, n_materials, c->d_param_imgs.p, c->stream); if (rc) return rc;
    return finish_param_batch(c, n_sets, out_imgs_u8, ref_img_u8, out_psnr);
}

int rr_score_images_device(rr_ctx* c, const uint8_t* d_imgs_u8, int n_images, const uint8_t* d_ref_u8, double* out_psnr,
                           uint64_t* out_sse, void* stream)
{
    if (!c) return -1;
    if (!c->have_cfg) return fail(c, -2, "rr_set_config has not been called");
    if (!d_imgs_u8 || !d_ref_u8 || (!out_psnr && !out_sse)) return fail(c, -3, "rr_score_images_device: null buffer");
    if (n_images < 1 || n_images > 65535) return fail(c, -3, "rr_score_images_device: n_images must be 1..65535");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t npx = (size_t)c->cfg.n_cells * c->cfg.n_angles;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "sse words");
    if (c->d_sse.n < (size_t)n_images) { RR_HIP(c, hipStreamSynchronize(s)); RR_HIP(c, c->d_sse.ensure((size_t)n_images)); }
    RR_HIP(c, hipMemsetAsync(c->d_sse.p, 0, (size_t)n_images * sizeof(uint64_t), s));
    launch_score(d_imgs_u8, d_ref_u8, npx, n_images, c->d_sse.p, s);
    RR_HIP(c, hipGetLastError());
    std::vector<uint64_t> sse((size_t)n_images);
    RR_HIP(c, hipMemcpyAsync(sse.data(), c->d_sse.p, (size_t)n_images * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    RR_HIP(c, hipStreamSynchronize(s));
    for (int k = 0; k < n_images; k++) {
        if (out_sse) out_sse[k] = sse[(size_t)k];
        if (out_psnr) {
            // skimage.metrics.peak_signal_noise_ratio for uint8 (scripts/radaray_opti.py:196): data_range 255,
            // err = mean of the squared differences in f64 (exact here: an integer sum below 2^53), 10 log10(255^2 / err)
            const double err = (double)sse[(size_t)k] / (double)npx;
            out_psnr[k] = err > 0.0 ? 10.0 * std::log10((255.0 * 255.0) / err) : INFINITY;
        }
    }
    return 0;
}

int rr_assemble_image_device(rr_ctx* c, const uint8_t* d_cols_u8, uint8_t* d_img_u8, void* stream)
{
    if (!c) return -1;
    if (!c->have_cfg) return fail(c, -2, "rr_set_config has not been called");
    if (!d_cols_u8 || !d_img_u8) return fail(c, -3, "rr_assemble_image_device: null buffer");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    { TimedScope t(c, s, "assemble"); launch_assemble_u8(d_cols_u8, d_img_u8, c->cfg.n_angles, c->cfg.n_cells, c->cfg.scroll_image, s); }
    RR_HIP(c, hipGetLastError());
    return 0;
}

int rr_assemble_frames_device(rr_ctx* c, const uint8_t* d_cols_u8, int n_loc, size_t block_stride,
                              int n_frames, size_t frame_stride, uint8_t* d_imgs_u8, void* stream)
{
    if (!c) return -1;
    if (!c->have_cfg) return fail(c, -2, "rr_set_config has not been called");
    if (!d_cols_u8 || !d_imgs_u8) return fail(c, -3, "rr_assemble_frames_device: null buffer");
    if (n_loc < 1 || c->cfg.n_angles % n_loc != 0) return fail(c, -3, "rr_assemble_frames_device: n_loc must divide n_angles");
    if (n_frames < 1 || n_frames > RR_MAX_BATCH) return fail(c, -3, "rr_assemble_frames_device: n_frames must be 1..64");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    { TimedScope t(c, s, "assemble"); launch_assemble_u8(d_cols_u8, d_imgs_u8, c->cfg.n_angles, c->cfg.n_cells, c->cfg.scroll_image, s, n_loc, block_stride, n_frames, frame_stride); }
    RR_HIP(c, hipGetLastError());
    return 0;
}

int rr_assemble_blocks_device(rr_ctx* c, const uint8_t* d_cols_u8, int n_loc, size_t block_stride,
                              uint8_t* d_img_u8, void* stream)
{
    if (!c) return -1;
    if (!c->have_cfg) return fail(c, -2, "rr_set_config has not been called");
    if (!d_cols_u8 || !d_img_u8) return fail(c, -3, "rr_assemble_blocks_device: null buffer");
    if (n_loc < 1 || c->cfg.n_angles % n_loc != 0) return fail(c, -3, "rr_assemble_blocks_device: n_loc must divide n_angles");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    { TimedScope t(c, s, "assemble"); launch_assemble_u8(d_cols_u8, d_img_u8, c->cfg.n_angles, c->cfg.n_cells, c->cfg.scroll_image, s, n_loc, block_stride); }
    RR_HIP(c, hipGetLastError());
    return 0;
}

int rr_simulate_device(rr_ctx* c, const float pose[7], uint8_t* d_img_u8, void* stream)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!pose || !d_img_u8) return fail(c, -3, "rr_simulate_device: null pose/output");
    RR_HIP(c, hipSetDevice(c->device));
    hipStream_t user = stream ? (hipStream_t)stream : c->stream;
    rc = upload_tables(c); if (rc) return rc;
    const int A = c->cfg.n_angles;
    if (c->lanes.size() == 1) {
        Lane& L = c->lanes[0];
        { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
        c->last_lane = 0;
        // With ONE lane every launch of the frame goes to the caller's stream, so the call can be CAPTURED into a hipGraph
        // (hipStreamBeginCapture on `user`, this call, hipStreamEndCapture) and replayed -- tools/cpp_bench.cpp `graph`.  While
        // capturing, the lane's hand-over event stays out of it (an event recorded outside the capture cannot be waited
        // for inside): the caller keeps other work off the context while such a graph runs.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(user, &cap);
        const bool capturing = cap == hipStreamCaptureStatusActive;
        // the lane's previous frame may have run on ANOTHER caller stream (or a flushed host copy may still read the lane)
        if (L.pending_consume && !capturing) RR_HIP(c, hipStreamWaitEvent(user, L.ev_consumed, 0));
        rc = run_frame(c, L, pose, 0, A, nullptr, nullptr, user); if (rc) return rc;
        rc = rr_assemble_image_device(c, L.d_cols_u8.p, d_img_u8, user); if (rc) return rc;
        if (!capturing) { RR_HIP(c, hipEventRecord(L.ev_consumed, user)); L.pending_consume = true; }
        return 0;
    }
    // Frame pipelining: trace/shade/scan/column of this frame run on the lane's own stream
    // (no dependency on the caller's stream), only the assemble -- the one kernel that touches
    // the caller's buffer -- is ordered on the caller's stream.  The lane is reused only after
    // that assemble has consumed its columns.
    const size_t li = c->next_stream_lane++ % (size_t)c->stream_lanes;
    Lane& L = c->lanes[li];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = li;
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(L.stream, L.ev_consumed, 0));
    rc = run_frame(c, L, pose, 0, A, nullptr, nullptr, L.stream); if (rc) return rc;
    RR_HIP(c, hipEventRecord(L.ev_ready, L.stream));
    RR_HIP(c, hipStreamWaitEvent(user, L.ev_ready, 0));
    rc = rr_assemble_image_device(c, L.d_cols_u8.p, d_img_u8, user); if (rc) return rc;
    RR_HIP(c, hipEventRecord(L.ev_consumed, user));
    L.pending_consume = true;
    return 0;
}

int rr_synchronize(rr_ctx* c, void* stream)
{
    if (!c) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    for (Lane& L : c->lanes) { int rc = flush_deferred(c, L); if (rc) return rc; }
    if (!c->deliveries.empty()) { const int rc = rr_wait_host(c, nullptr); if (rc) return rc; }
    for (Lane& L : c->lanes) RR_HIP(c, hipStreamSynchronize(L.stream));
    RR_HIP(c, hipStreamSynchronize(stream ? (hipStream_t)stream : c->stream));
    // batches may run on OTHER caller streams as well (the header recommends four): a frame there could set a
    // bit between the read and the clear below, so the whole device is drained first -- after this call no
    // frame of this context is in flight anywhere and every error bit raised so far is reported exactly once
    RR_HIP(c, hipDeviceSynchronize());
    for (Lane& L : c->lanes) for (Lane::CopyRec& r : L.rec) { r.pending = false; r.dst = nullptr; }
    // error bits of every frame the asynchronous entry points enqueued since the last call (a frame that
    // overflowed its wave queue or met a bad material id is truncated, never silently)
    uint32_t bits = 0;
    for (Lane& L : c->lanes) {
        if (!L.d_sticky.p) continue;
        uint32_t b = 0;
        RR_HIP(c, hipMemcpy(&b, L.d_sticky.p, sizeof(b), hipMemcpyDeviceToHost));
        if (b) { bits |= b; RR_HIP(c, hipMemset(L.d_sticky.p, 0, sizeof(b))); }
    }
    if (bits & 1u) return fail(c, -7, "wave/signal queue capacity exceeded in a frame since the last rr_synchronize; raise rr_config.max_waves_per_azimuth");
    if (bits & 2u) return fail(c, -8, "object id or material id out of range of the tables given to rr_set_materials (a frame since the last rr_synchronize)");
    return 0;
}

int rr_peek_error_bits_async(rr_ctx* c, uint32_t* h_bits, void* stream)
{
    if (!c) return -1;
    if (!h_bits) return fail(c, -3, "rr_peek_error_bits_async: null pointer");
    RR_HIP(c, hipSetDevice(c->device));
    Lane& L = c->lanes[c->last_lane];
    if (!L.d_sticky.p) { *h_bits = 0; return 0; }     // no frame has run on this lane yet
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->flush_kernel && host_visible(h_bits)) { launch_store_u32(L.d_sticky.p, h_bits, s); RR_HIP(c, hipGetLastError()); }     // (a kernel's store: no copy engine involved)
    else RR_HIP(c, hipMemcpyAsync(h_bits, L.d_sticky.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return 0;
}

int rr_get_stats(rr_ctx* c, rr_stats* st)
{
    if (!c || !st) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    std::memset(st, 0, sizeof(*st));
    Lane& L = c->lanes[c->last_lane];
    if (!L.d_counters.p) return 0;
    Counters h;
    { const int rcb = read_back(c, &h, L.d_counters.p, sizeof(h)); if (rcb) return rcb; }
    st->nodes_visited = h.nodes; st->tris_tested = h.tris; st->overflow = h.overflow;
    if (getenv("RR_TRACE_STATS")) fprintf(stderr, "[rr stats] waves %u wave_iters %llu (avg %.1f) max_iters %u\n", h.n_waves, h.wave_iters, h.n_waves ? (double)h.wave_iters / h.n_waves : 0.0, h.max_iters);
    if (getenv("RR_TRACE_STATS") && h.n_waves)
        fprintf(stderr, "[rr stats] per wave: iterations %.2f, issuing node path %.2f, leaf path %.2f, live quad-steps %.1f (of 16 x iterations = %.1f)\n",
                (double)h.it_all / h.n_waves, (double)h.it_node / h.n_waves, (double)h.it_leaf / h.n_waves,
                (double)h.quad_steps / h.n_waves, 16.0 * h.it_all / h.n_waves);
    const size_t n = (size_t)L.last_n_seg * (size_t)L.last_n_passes;
    if (n && L.d_seg_stats.p) {
        std::vector<SegStats> ss(n);
        { const int rcb = read_back(c, ss.data(), L.d_seg_stats.p, n * sizeof(SegStats)); if (rcb) return rcb; }
        for (const SegStats& x : ss) { st->wave_passes += x.wave_passes; st->hits += x.hits; st->signals += x.signals; }
    }
    return 0;
}

int rr_simulate(rr_ctx* c, const float pose[7], int az_begin, int az_end,
                uint8_t* out_u8, float* out_f32, rr_stats* stats)
{
    int rc = check_ready(c); if (rc) return rc;
    if (!pose || (!out_u8 && !out_f32)) return fail(c, -3, "rr_simulate: null pose/output");
    RR_HIP(c, hipSetDevice(c->device));
    const rr_config& g = c->cfg;
    if (az_begin < 0 || az_end > g.n_angles || az_begin > az_end) return fail(c, -3, "azimuth range out of bounds");
    const int n_seg = az_end - az_begin;
    if (n_seg == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return 0; }
    rc = upload_tables(c); if (rc) return rc;
    Lane& L = c->lanes[0];
    { int rcf = flush_deferred(c, L); if (rcf) return rcf; }   // images a host-delivery batch left on this lane
    c->last_lane = 0;
    // The reference's call shape: one synchronous simulate() per frame (radar_simulator.cpp:197-212).  Its latency is
    // the chain of kernels plus what the host adds around it, so the host adds as little as it can: the frame is
    // ordered behind the lane's previous user by an event (no device-wide drain), the error bits and the per-pass
    // counters ride home behind the image on the same stream, and ONE hipStreamSynchronize ends the call.
    if (L.pending_consume) RR_HIP(c, hipStreamWaitEvent(c->stream, L.ev_consumed, 0));
    rc = run_frame(c, L, pose, az_begin, az_end, nullptr, nullptr, c->stream, 1, nullptr, 0, out_f32 != nullptr);
    if (rc) return rc;
    const size_t n_st = (size_t)n_seg * (size_t)std::max(1, g.n_reflections);
    const size_t need = sizeof(Counters) + (stats ? n_st * sizeof(SegStats) : 0);
    if (c->h_frame_bytes < need) {
        if (c->h_frame) (void)hipHostFree(c->h_frame);
        c->h_frame = nullptr; c->h_frame_bytes = 0;
        RR_HIP(c, hipHostMalloc(&c->h_frame, need + 4096, hipHostMallocDefault));
        c->h_frame_bytes = need + 4096;
    }
    Counters* h_cnt = reinterpret_cast<Counters*>(c->h_frame);
    SegStats* h_ss = reinterpret_cast<SegStats*>(h_cnt + 1);
    std::vector<uint8_t> h8; std::vector<float> hf;
    if (n_seg == g.n_angles) {
        // whole frame: transpose on the GPU, one D2H copy straight into the caller's row-major buffer
        const size_t npx = (size_t)g.n_cells * g.n_angles;
        if (out_u8) {
            RR_HIP(c, L.d_img_u8.ensure(npx));
            launch_assemble_u8(L.d_cols_u8.p, L.d_img_u8.p, g.n_angles, g.n_cells, g.scroll_image, c->stream);
            RR_HIP(c, hipMemcpyAsync(out_u8, L.d_img_u8.p, npx, hipMemcpyDeviceToHost, c->stream));
        }
        if (out_f32) {
            RR_HIP(c, L.d_img_f32.ensure(npx));
            launch_assemble_f32(L.d_cols_f32.p, L.d_img_f32.p, g.n_angles, g.n_cells, g.scroll_image, c->stream);
            RR_HIP(c, hipMemcpyAsync(out_f32, L.d_img_f32.p, npx * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        }
    } else {
        h8.resize((size_t)n_seg * g.n_cells);
        hf.resize(out_f32 ? (size_t)n_seg * g.n_cells : 0);
        RR_HIP(c, hipMemcpyAsync(h8.data(), L.d_cols_u8.p, h8.size(), hipMemcpyDeviceToHost, c->stream));
        if (out_f32) RR_HIP(c, hipMemcpyAsync(hf.data(), L.d_cols_f32.p, hf.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    // error bits / counters and the per-pass statistics ride home behind the image.  (Round 6 tried ONE kernel storing both into
    // the page-locked block instead -- k_frame_report, no copy engine involved: 0.151 instead of 0.139-0.141 ms per call on
    // config 2.  This is the latency path; the small copies stay.)
    RR_HIP(c, hipMemcpyAsync(h_cnt, L.d_counters.p, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    if (stats && L.d_seg_stats.p && g.n_reflections > 0)
        RR_HIP(c, hipMemcpyAsync(h_ss, L.d_seg_stats.p, n_st * sizeof(SegStats), hipMemcpyDeviceToHost, c->stream));
    RR_HIP(c, hipEventRecord(L.ev_consumed, c->stream));
    L.pending_consume = true;
    RR_HIP(c, hipStreamSynchronize(c->stream));
    if (n_seg != g.n_angles) {
        for (int s = 0; s < n_seg; s++) {
            const int col = (g.scroll_image + az_begin + s) % g.n_angles;   // RadarCPU.cpp:457
            for (int i = 0; i < g.n_cells; i++) {
                if (out_u8) out_u8[(size_t)i * g.n_angles + col] = h8[(size_t)s * g.n_cells + i];
                if (out_f32) out_f32[(size_t)i * g.n_angles + col] = hf[(size_t)s * g.n_cells + i];
            }
        }
    }
    const uint32_t overflow = h_cnt->overflow;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->nodes_visited = h_cnt->nodes; stats->tris_tested = h_cnt->tris; stats->overflow = overflow;
        if (g.n_reflections > 0)
            for (size_t k = 0; k < n_st; k++) { stats->wave_passes += h_ss[k].wave_passes; stats->hits += h_ss[k].hits; stats->signals += h_ss[k].signals; }
        if (getenv("RR_TRACE_STATS")) { rr_stats tmp; (void)rr_get_stats(c, &tmp); }     // prints the wave-level loop shape
    }
    if (overflow) RR_HIP(c, hipMemset(L.d_sticky.p, 0, sizeof(uint32_t)));   // reported here, not again by rr_synchronize
    if (overflow & 1u) return fail(c, -7, "wave/signal queue capacity exceeded; raise rr_config.max_waves_per_azimuth");
    if (overflow & 2u) return fail(c, -8, "object id or material id out of range of the tables given to rr_set_materials");
    return 0;
}

int rr_set_stats_mode(rr_ctx* c, int enable) { if (!c) return -1; c->stats_mode = enable != 0; return 0; }

int rr_get_traversal_shape(rr_ctx* c, uint64_t out[8])
{
    if (!c || !out) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    std::memset(out, 0, 8 * sizeof(uint64_t));
    Lane& L = c->lanes[c->last_lane];
    if (!L.d_counters.p) return 0;
    Counters h;
    { const int rcb = read_back(c, &h, L.d_counters.p, sizeof(h)); if (rcb) return rcb; }
    out[0] = h.n_waves; out[1] = h.it_all; out[2] = h.it_node; out[3] = h.it_leaf; out[4] = h.quad_steps; out[5] = h.max_iters;
    out[6] = h.nodes; out[7] = h.quad_steps > h.nodes ? h.quad_steps - h.nodes : 0;
    return 0;
}
int rr_set_timing_mode(rr_ctx* c, int enable) { if (!c) return -1; c->timing = enable; return 0; }

int rr_get_kernel_time(rr_ctx* c, const char* kernel, double* total_ms, uint64_t* launches, int reset)
{
    if (!c || !kernel) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    KernelTimer& t = c->timers[kernel];
    for (auto& p : t.pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { t.total_ms += ms; t.launches++; t.samples_ms.push_back(ms); }
        c->event_pool.push_back(p.first); c->event_pool.push_back(p.second);
    }
    t.pending.clear();
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    if (reset) { t.total_ms = 0.0; t.launches = 0; t.samples_ms.clear(); }
    return 0;
}

int rr_get_kernel_samples(rr_ctx* c, const char* kernel, float* out_ms, size_t capacity, size_t* n_out)
{
    if (!c || !kernel || !n_out) return -1;
    int rc = rr_get_kernel_time(c, kernel, nullptr, nullptr, 0); if (rc) return rc;
    const KernelTimer& t = c->timers[kernel];
    *n_out = t.samples_ms.size();
    if (out_ms) for (size_t i = 0; i < t.samples_ms.size() && i < capacity; i++) out_ms[i] = t.samples_ms[i];
    return 0;
}

int rr_reserve_timing_events(rr_ctx* c, size_t n)
{
    if (!c) return -1;
    RR_HIP(c, hipSetDevice(c->device));
    while (c->event_pool.size() < n) { hipEvent_t e = nullptr; RR_HIP(c, hipEventCreate(&e)); c->event_pool.push_back(e); }
    return 0;
}

int rr_get_trace_grid(rr_ctx* c, uint32_t out_rows[24], uint32_t out_hist[24], uint64_t* repaired_groups)
{
    if (!c) return -1;
    static_assert(kMaxPasses == 24, "rr_get_trace_grid's arrays");
    RR_HIP(c, hipSetDevice(c->device));
    RR_HIP(c, hipDeviceSynchronize());
    const Lane& L = c->lanes[c->last_lane];
    uint64_t rep = 0;
    for (int k = 0; k < kMaxPasses; k++) { if (out_rows) out_rows[k] = L.last_rows[k]; if (out_hist) out_hist[k] = 0; }
    for (const Lane& o : c->lanes) {
        if (!o.d_hint.p || o.hist_gen != c->hist_gen) continue;
        GridHint h;
        RR_HIP(c, hipMemcpy(&h, o.d_hint.p, sizeof(h), hipMemcpyDeviceToHost));
        rep += h.repaired;
        for (int k = 0; k < kMaxPasses && out_hist; k++) out_hist[k] = std::max(out_hist[k], h.hist[k]);
    }
    if (repaired_groups) *repaired_groups = rep;
    return 0;
}

int rr_get_graph_stats(rr_ctx* c, uint64_t* captures, uint64_t* replays)
{
    if (!c) return -1;
    if (captures) *captures = c->graph_captures;
    if (replays) *replays = c->graph_replays;
    return 0;
}

int rr_get_bvh_info(rr_ctx* c, uint64_t* n_nodes, uint64_t* n_tris, uint32_t* depth, uint32_t* stack_need)
{
    if (!c) return -1;
    if (!c->have_mesh) return fail(c, -2, "rr_set_mesh has not been called");
    if (n_nodes) *n_nodes = c->n_nodes;
    if (n_tris) *n_tris = c->n_tris;
    if (depth) *depth = c->depth;
    if (stack_need) *stack_need = c->stack_need;
    return 0;
}

int rr_debug_fresnel(rr_ctx* c, size_t n, const float* normals, const float* dirs, const double* energy, const double* v1, const float* v2,
                     float* out_refl_dir, double* out_refl_energy, float* out_refr_dir, double* out_refr_energy)
{
    if (!c) return -1;
    if (n == 0) return 0;
    if (!normals || !dirs || !energy || !v1 || !v2 || !out_refl_dir || !out_refl_energy || !out_refr_dir || !out_refr_energy)
        return fail(c, -3, "rr_debug_fresnel: null pointer");
    RR_HIP(c, hipSetDevice(c->device));
    DevBuf<float> d_n, d_d, d_v2, d_rd, d_td; DevBuf<double> d_e, d_v1, d_re, d_te;
    hipError_t e = d_n.ensure(3 * n);
    if (e == hipSuccess) e = d_d.ensure(3 * n);
    if (e == hipSuccess) e = d_v2.ensure(n);
    if (e == hipSuccess) e = d_rd.ensure(3 * n);
    if (e == hipSuccess) e = d_td.ensure(3 * n);
    if (e == hipSuccess) e = d_e.ensure(n);
    if (e == hipSuccess) e = d_v1.ensure(n);
    if (e == hipSuccess) e = d_re.ensure(n);
    if (e == hipSuccess) e = d_te.ensure(n);
    if (e == hipSuccess) e = hipMemcpy(d_n.p, normals, 3 * n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d.p, dirs, 3 * n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_v2.p, v2, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_e.p, energy, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_v1.p, v1, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_debug_fresnel(n, d_n.p, d_d.p, d_e.p, d_v1.p, d_v2.p, d_rd.p, d_re.p, d_td.p, d_te.p, c->stream);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out_refl_dir, d_rd.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_refr_dir, d_td.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_refl_energy, d_re.p, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_refr_energy, d_te.p, n * sizeof(double), hipMemcpyDeviceToHost);
    d_n.release(); d_d.release(); d_v2.release(); d_rd.release(); d_td.release(); d_e.release(); d_v1.release(); d_re.release(); d_te.release();
    if (e != hipSuccess) return fail(c, -100, std::string("rr_debug_fresnel: ") + hipGetErrorString(e));
    return 0;
}

int rr_debug_brdf(rr_ctx* c, size_t n, const float* in5, int brdf_model, float* out)
{
    if (!c) return -1;
    if (n == 0) return 0;
    if (!in5 || !out) return fail(c, -3, "rr_debug_brdf: null pointer");
    if (brdf_model != 0 && brdf_model != 1) return fail(c, -3, "rr_debug_brdf: brdf_model must be 0 or 1");
    RR_HIP(c, hipSetDevice(c->device));
    DevBuf<float> d_in, d_out;
    hipError_t e = d_in.ensure(5 * n);
    if (e == hipSuccess) e = d_out.ensure(n);
    if (e == hipSuccess) e = hipMemcpy(d_in.p, in5, 5 * n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_debug_brdf(n, d_in.p, brdf_model, d_out.p, c->stream);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out.p, n * sizeof(float), hipMemcpyDeviceToHost);
    d_in.release(); d_out.release();
    if (e != hipSuccess) return fail(c, -100, std::string("rr_debug_brdf: ") + hipGetErrorString(e));
    return 0;
}

int rr_debug_trace(rr_ctx* c, const float* origs, const float* dirs, size_t n, float* out_t, uint32_t* out_face)
{
    if (!c) return -1;
    if (!c->have_mesh) return fail(c, -2, "rr_set_mesh has not been called");
    if (n == 0) return 0;
    if (!origs || !dirs || !out_t || !out_face) return fail(c, -3, "rr_debug_trace: null pointer");
    RR_HIP(c, hipSetDevice(c->device));
    const size_t chunk = 1u << 16;
    const int stack_lds = (int)std::max<uint32_t>(1, std::min<uint32_t>(c->stack_need, (uint32_t)c->stack_lds_max));
    const int spill_depth = (int)c->stack_need - stack_lds;
    DevBuf<float> d_o, d_d, d_t; DevBuf<uint32_t> d_f, d_spill;
    RR_HIP(c, d_o.ensure(3 * chunk)); RR_HIP(c, d_d.ensure(3 * chunk)); RR_HIP(c, d_t.ensure(chunk));
    RR_HIP(c, d_f.ensure(chunk)); RR_HIP(c, d_spill.ensure(spill_depth > 0 ? (size_t)spill_depth * chunk : 1));
    Params P; std::memset(&P, 0, sizeof(P));
    P.nodes = reinterpret_cast<const Node4*>(c->d_bvh.p); P.tris = reinterpret_cast<const TriRec*>(c->d_bvh.p + c->tri_base4);
    P.tri_base4 = c->tri_base4; P.range_max = c->have_cfg ? c->cfg.range_max : 1000.0f; P.hit_pad = c->hit_pad;
    P.spill = d_spill.p; P.spill_stride = (int)chunk; P.stack_lds = stack_lds; P.spill_depth = std::max(0, spill_depth);
    P.cull_pop = c->cull_pop;
    int rc = 0;
    for (size_t b = 0; b < n && !rc; b += chunk) {
        const size_t m = std::min(chunk, n - b);
        hipError_t e = hipMemcpy(d_o.p, origs + 3 * b, 3 * m * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_d.p, dirs + 3 * b, 3 * m * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) { launch_debug_trace(P, d_o.p, d_d.p, (int)m, d_t.p, d_f.p, c->stream); e = hipStreamSynchronize(c->stream); }
        if (e == hipSuccess) e = hipMemcpy(out_t + b, d_t.p, m * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out_face + b, d_f.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, -100, std::string("rr_debug_trace: ") + hipGetErrorString(e));
    }
    d_o.release(); d_d.release(); d_t.release(); d_f.release(); d_spill.release();
    return rc;
}

}  // extern "C"
