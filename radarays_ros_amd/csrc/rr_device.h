// rr_device.h -- device-side data layout and per-hit radar math (gfx950).
//
// Arithmetic note: the reference CPU path is compiled without FMA contraction
// (CMakeLists.txt:4-5: only -std=c++17) and mixes f32 vectors with f64 wave
// energy/time.  Everything below that feeds the image keeps that op order and
// those types (this file is compiled with -ffp-contract=off); FMAs appear only
// in the BVH slab test, which only has to be conservative.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rr_bvh.h"

namespace rr {

// ---------------------------------------------------------------- vec / quat
struct V3 { float x, y, z; };
struct Quat { float x, y, z, w; };

__host__ __device__ inline V3 v_add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__host__ __device__ inline V3 v_sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__host__ __device__ inline V3 v_neg(V3 a) { return { -a.x, -a.y, -a.z }; }
__host__ __device__ inline V3 v_scale(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__host__ __device__ inline float v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline V3 v_cross(V3 a, V3 b)
{
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
__host__ __device__ inline V3 v_normalize(V3 a)
{
    const float d = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return { a.x / d, a.y / d, a.z / d };
}
// rmagine Quaternion * Quaternion (Hamilton product), term order as in rmagine
__host__ __device__ inline Quat q_mul(Quat a, Quat b)
{
    Quat r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
__host__ __device__ inline Quat q_conj(Quat a) { return { -a.x, -a.y, -a.z, a.w }; }
// rmagine Quaternion * Vector = (q (v,0) q^-1).xyz: the two Hamilton products of q_mul in its term order, with the four
// products by the zero w of (v, 0) left out -- each is +-0 for a finite q, and x + (+-0) = x: the same values (only the sign
// of an exact zero can differ, which nothing downstream can see), 8 instructions per rotation less
__host__ __device__ inline V3 q_rot(Quat q, V3 v)
{
    Quat t;                                   // t = q * (v, 0)
    t.x = q.w * v.x + q.y * v.z - q.z * v.y;
    t.y = q.w * v.y - q.x * v.z + q.z * v.x;
    t.z = q.w * v.z + q.x * v.y - q.y * v.x;
    t.w = 0.0f - q.x * v.x - q.y * v.y - q.z * v.z;
    const Quat c = q_conj(q);
    V3 r;                                     // (t * q^-1).xyz
    r.x = t.w * c.x + t.x * c.w + t.y * c.z - t.z * c.y;
    r.y = t.w * c.y - t.x * c.z + t.y * c.w + t.z * c.x;
    r.z = t.w * c.z + t.x * c.y - t.y * c.x + t.z * c.w;
    return r;
}

// ------------------------------------------------------------ device params
// Wave state, 48 B, three 16-B SoA streams (coalesced dwordx4 per lane):
//   A = (orig.x, orig.y, orig.z, dir.x)   B = (dir.y, dir.z, material_id, -)
//   C = (energy f64, time f64)            (radar_types.h:63-121; velocity and
//   polarization never change on the CPU path -- SURVEY.md §2.3 #3 -- and are
//   the constants 0.3 / 0.5 of RadarCPU.cpp:107-110)
struct WaveBuf {
    float4* A;
    float4* B;
    double2* C;
};

struct SigRec { int32_t cell; float strength; };

struct Counters {
    unsigned long long nodes, tris;   // stats mode only (atomics)
    unsigned long long wave_iters;    // sum over waves of traversal-loop iterations
    unsigned int max_iters, n_waves;  // longest wave, wave count
    unsigned int overflow, pad;       // error bits, set on rare error paths
    unsigned long long it_all, it_node, it_leaf, quad_steps;   // stats mode: wave-loop iterations (all / issuing the node path / the leaf path), quad steps
    unsigned long long pad2;          // (a multiple of 16 bytes: k_frame_report moves it as uint4)
};

// Tight later-pass trace grids (round 5).  A segment holds at most n_beam * 2^pass waves in pass `pass`, and launch_trace
// used to launch a row of 16-ray workgroups up to that bound per segment; far fewer are live (10M-triangle target, pass 3:
// 62 of 100), and a workgroup that starts only to read its segment's count and leave costs a dispatch and a wave slot.
// The host now sizes the row by what earlier batches needed: hist[p] = the largest count any segment had in pass p
// (monotone maximum, written by k_scan; copied to the host behind every batch), rows of hist * 17/16 + 32 rays.  A segment
// that exceeds its row anyway is put on ovf_list by k_scan and its remaining groups are traced by k_trace_repair, a small
// launch that follows every tightened trace launch and exits at once when the list is empty: results never depend on
// the hint.
constexpr int kMaxPasses = 24;
struct GridHint { uint32_t hist[kMaxPasses]; uint32_t ovf_n[kMaxPasses]; unsigned long long repaired; /* 16-ray groups k_trace_repair traced since the history started over */ };

// per (pass, azimuth) counters, written once by the kernel that finishes the pass
// (no atomics in the frame path: ~10 ns each, they serialise at the L2)
struct SegStats { uint32_t wave_passes, hits, signals, pad; };

struct Params {
    // scene
    const Node4* nodes;          // base of the tree allocation: nodes, then triangles
    const TriRec* tris;          // = (TriRec*)((float4*)nodes + tri_base4)
    uint32_t tri_base4;          // float4 offset of triangle 0 from `nodes`
    // per-config tables
    const float4* q_as;          // [n_angles] Tas.R (RadarCPU.cpp:202)
    const float4* beams;         // [n_beam] xyz
    const uint32_t* beam_order;  // [n_beam] trace slot -> beam index of pass 0 (rows of equal elevation: equally long rays share a wave)
    const uint32_t* beam_order2; // [n_beam] the order the LATER passes inherit (yaw-major rows)
    const float4* materials;     // [n_materials] velocity, ambient, diffuse, specular
    const double* mat_limits;    // per entry of `materials`: the angle of total reflection for a wave that meets the material coming from air (k_mat_limits)
    double limit_same;           // the same angle for v2 = v1 = 0.3 (both sides the same material)
    const int32_t* object_materials;
    const float* smear;          // [smear_w] rescaled weights (RadarCPU.cpp:48-93)
    const float* noise_rnd;      // [noise_rows][n_angles] or null; frame f of a batch reads row f % noise_rows
    int noise_rows;
    int mat_stride;              // parameter sets: frame f shades with materials[f * mat_stride + id] (0: one table)
    // parameter batch (rr_simulate_param_sets / _material_sets): every "frame" is the SAME pose under its own parameter
    // set = {material table, beam directions, number of passes}.  Frames with the same beam form a GROUP: pass 0 does not
    // depend on the materials, so it is traced once per group (for the group's first frame) and every frame of the group
    // shades those hits; a frame with fewer passes than the launch simply has no live waves in the later ones.
    int set_mode;                // 0: frames are poses; 1: frames are parameter sets
    int n_groups;                // set_mode: distinct beam tables = pass-0 trace groups (beam table g at beams + g * n_beam, same for the orders)
    unsigned char frame_passes[64];   // set_mode: ray-cast passes of frame f (<= n_passes)
    unsigned char frame_beam[64];     // set_mode: group (= beam table) of frame f
    unsigned char group_frame[64];    // set_mode: the frame whose segments pass 0 of group g is traced for
    const float* decay;          // [n_cells] expf(-energy_loss * bin range), ambient noise floor
    const float* motion_poses;   // [motion_rows][n_angles][7] per-azimuth Tsm (include_motion) or null; frame f of a batch reads table f % motion_rows
    int motion_rows;
    // frame state
    WaveBuf waves[2];            // [n_seg][2*cap] child slots, ping-pong by pass parity
    uint32_t* idx[2];            // [n_seg][cap] live slot list
    uint32_t* count[2];          // [n_seg]
    uint2* torder[2];            // [n_seg][cap] trace position -> (position in idx, child slot): the order the rays of a later pass are TRACED in (coherent order, inherited from pass 0); the slot rides along so that k_trace reaches its wave with one dependent load less
    uint32_t* refpos;            // [n_seg][2*cap] child slot -> its position in the next pass' idx
    uint8_t* cflag;              // [n_seg][2*cap] child alive flags (+ bit2 on slot 2j: hit)
    SigRec* sigtmp;              // [n_seg][2*cap] per-wave signal slots (path, air)
    uint2* hit;                  // [n_seg][cap] nearest hit of a wave: (range as float bits, or -1.0f for a miss; leaf-order triangle index): ONE 8-B store per ray
    SigRec* sig;                 // [n_seg][sigcap] ordered signal list
    uint32_t* sig_count;         // [n_seg]
    uint32_t* spill;             // traversal stack spill [depth][threads]
    Counters* counters;
    uint32_t* sticky;            // error bits OR-ed over all frames of the lane (read + cleared by rr_synchronize)
    SegStats* seg_stats;         // [n_passes][n_seg]
    uint8_t* cols_u8;            // [n_seg][n_cells]
    float* cols_f32;             // optional
    // scalars
    int az_begin, n_seg;
    // frame batch: segment s belongs to frame s / n_loc and azimuth az_begin + s % n_loc
    int n_loc, n_frames;
    // the poses of the call: [n_frames][2] float4 = (q.xyzw)(t.xyz, 0) in the lane's device table (one row for a single frame
    // and for a parameter batch, whose sets share one pose).  The pass-0 k_trace launch -- the first of every chain --
    // carries them by value (PoseArgs below), uses them for its own rays and has its first workgroup write the table that
    // every later launch of the chain reads.  It is also the ONE node whose parameters change when the chain is replayed
    // from a launch graph.  No kernel reads a pose from two places: a build that chose between a by-value pose and the
    // table made the compiler copy all of Params into scratch (2.6 KB per lane, 7 waves per SIMD; tests/test_kernel_resources.py)
    float4* pose_table;
    int n_beam, cap, sigcap;
    int n_cells, n_angles, n_materials, n_objects, material_id_air;
    int n_passes, record_multi_reflection, record_multi_path;
    int brdf_model;              // 0: A + B cos^C (reference); 1: Cook-Torrance lobe (build's own, rr_config.brdf_model)
    int signal_denoising, smear_w, smear_mode, ambient_noise, scroll;
    float thr, range_max;
    float hit_pad;               // grazing guard of the triangle test: half the builders' box padding (traverse, rr_kernels.hip)
    double resolution, multipath_threshold;
    float energy_max_f;          // (float)energy_max  (cv convertTo alpha)
    double signal_max;
    double noise_at_0, noise_at_1, noise_e_max, noise_e_min, noise_e_loss;
    int spill_stride, stack_lds, spill_depth;
    int pass0_az;                // pass 0: neighbouring segments per wave (power of two <= 16); 16 / pass0_az samples each
    // host delivery folded into the later-pass trace launches (rr_simulate_batch_host_async): row 0 of such a grid is
    // not rays -- its first copy_blocks workgroups trickle a slice of the images the lane's PREVIOUS batch left in
    // device memory to page-locked host memory, one 1-KB store per wave in flight
    const uint4* copy_src; uint4* copy_dst; unsigned long long copy_n16; int copy_blocks;
    int trace_row;           // (set by launch_trace) row length of this launch when seg_chunk > 0
    // later-pass k_trace, S > 0: the grid is (S, ceil(n_seg / S) * row) -- chunks of S neighbouring segments (azimuths of one
    // frame), and inside a chunk the SEGMENT is the fast dimension: the workgroups that run at the same time hold the same trace
    // positions of neighbouring azimuths, and a segment stays on ONE XCD (S a multiple of 8).  0: one row of the grid per segment
    int seg_chunk;
    int stackless;           // RR_STACKLESS=1: k_trace walks the tree without a stack (traverse_stackless: parent links, nodes re-fetched on the way up; no LDS) -- a measured alternative, not the default
    int cull_pop;            // later passes / rr_debug_trace: drop stack entries at pop time by their 16-bit distance bound (0: off, RR_CULL_POP=0)
    // tight later-pass trace grids (GridHint above)
    GridHint* grid_hint;     // per lane; null: off
    uint32_t* hist_host;     // page-locked host copy of grid_hint->hist (kMaxPasses words), written by the chain's k_column: the host reads it as a hint, without a fence.  A store from a kernel, not a 96-byte hipMemcpyAsync: which engine carries such a copy is the runtime's choice (the one bundled with the torch wheel runs a blit kernel per copy, 481 of them in a profiled run of the target)
    uint32_t* ovf_list;      // [n_passes][ovf_stride] segments whose count exceeds the tightened row of that pass
    int ovf_stride;
    unsigned short tight_groups[kMaxPasses];   // 16-ray workgroups per segment row of pass p; 0: the full doubling bound
};

struct PoseArgs { float p[64][7]; int n; };     // RR_MAX_BATCH poses by value: third argument of the pass-0 k_trace
struct NoPoses {};                              // ... and of the later passes
template <bool FIRST> struct PosesOf { using type = NoPoses; };
template <> struct PosesOf<true> { using type = PoseArgs; };

__host__ __device__ inline int passes_of(const Params& P, int frame) { return P.set_mode ? (int)P.frame_passes[frame] : P.n_passes; }
__host__ __device__ inline int beam_base(const Params& P, int frame) { return P.set_mode ? (int)P.frame_beam[frame] * P.n_beam : 0; }

static_assert(sizeof(Params) <= 4096, "Params is passed by value: HIP kernel arguments are limited to 4 KB");

}  // namespace rr
