// rr_lbvh.hip -- BVH build ON THE GPU (the "next" row N2 of SURVEY.md §8f: replaces
// rm::import_embree_map's Embree build, src/radar_simulator.cpp:149, for maps that must load
// in milliseconds instead of seconds).
//
//   0. early split clipping (Ernst & Greiner 2007): a face much larger than its neighbours is cut along a grid of
//      cell size L into references {clipped box, face}; L is the smallest cell that keeps the reference count
//      within +25 % (bisection over a counting kernel).  Without it one 15 m building face inflates every
//      Morton-neighbourhood of 0.2 m terrain triangles it is sorted into.                 k_split<false/true>
//   1. per-reference bounds + 63-bit Morton code of the centroid (cubic cells: one scale for all axes)   k_prim
//   2. rocprim radix sort of (code, face)                             rocprim::radix_sort_pairs
//   3. Karras 2012 binary radix tree over the sorted codes           k_karras
//   4. bottom-up bounds + subtree sizes (one atomic ticket per node)  k_refit
//   5. top-down collapse to the 4-wide node layout of rr_bvh.h, every subtree of <= 4
//      triangles becoming a leaf (contiguous in Morton order)         k_collapse (per level)
//   6. triangles in leaf order                                        k_tris
//
// The tree is lower quality than the host SAH build (rr_bvh.cpp) -- traversal is slower --
// but the nearest hit is defined order-independently (min over (t, face id)), so images are
// BIT-IDENTICAL whichever builder made the tree.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "rr_bvh.h"

namespace rr {

namespace {

struct Box6 { float lo[3], hi[3]; };

__device__ inline unsigned long long expand21(unsigned long long v)
{
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// bounds of (triangle clipped to the axis-aligned box [clo, chi]), Sutherland-Hodgman in f32; false if empty.
// The result is rounded outwards by one ulp and intersected with the cell, so the parts of a face cover it.
__device__ inline bool clip_tri_box(const float* a, const float* b, const float* c, const float* clo, const float* chi, Box6& out)
{
    float p[10][3], q[10][3];
    int n = 3;
    for (int k = 0; k < 3; k++) { p[0][k] = a[k]; p[1][k] = b[k]; p[2][k] = c[k]; }
    for (int axis = 0; axis < 3 && n > 0; axis++)
        for (int side = 0; side < 2 && n > 0; side++) {
            const float plane = side == 0 ? clo[axis] : chi[axis];
            int m = 0;
            for (int i = 0; i < n; i++) {
                const float* u = p[i]; const float* v = p[(i + 1) % n];
                const bool inu = side == 0 ? u[axis] >= plane : u[axis] <= plane;
                const bool inv = side == 0 ? v[axis] >= plane : v[axis] <= plane;
                if (inu) { q[m][0] = u[0]; q[m][1] = u[1]; q[m][2] = u[2]; m++; }
                if (inu != inv) {
                    const float t = (plane - u[axis]) / (v[axis] - u[axis]);
                    for (int k = 0; k < 3; k++) q[m][k] = u[k] + t * (v[k] - u[k]);
                    q[m][axis] = plane;
                    m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++) { p[i][0] = q[i][0]; p[i][1] = q[i][1]; p[i][2] = q[i][2]; }
        }
    if (n == 0) return false;
    for (int k = 0; k < 3; k++) { out.lo[k] = 3.0e38f; out.hi[k] = -3.0e38f; }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { out.lo[k] = fminf(out.lo[k], p[i][k]); out.hi[k] = fmaxf(out.hi[k], p[i][k]); }
    for (int k = 0; k < 3; k++) {
        const float e = 1e-6f * (fabsf(out.lo[k]) + fabsf(out.hi[k])) + 1e-30f;    // interpolation error of the cut points
        out.lo[k] = fmaxf(out.lo[k] - e, clo[k] - e); out.hi[k] = fminf(out.hi[k] + e, chi[k] + e);
    }
    return true;
}

constexpr int kMaxCellsPerAxis = 24;     // a face is cut into at most 24 cells per axis (its own coarser grid beyond that)

// EMIT = false: count[f] = references face f yields for cell size L; EMIT = true: write them at offset[f]
template <bool EMIT>
__global__ void k_split(const float* __restrict__ verts, const uint32_t* __restrict__ faces, uint32_t nf, float L, float3 slo,
                        uint32_t* __restrict__ count, const uint32_t* __restrict__ offset,
                        Box6* __restrict__ rbox, uint32_t* __restrict__ rface)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + 0];
    const float* b = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
    const float* c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
    Box6 bb;
    for (int k = 0; k < 3; k++) { bb.lo[k] = fminf(a[k], fminf(b[k], c[k])); bb.hi[k] = fmaxf(a[k], fmaxf(b[k], c[k])); }
    int i0[3], nc[3]; float cell[3];
    const float org[3] = { slo.x, slo.y, slo.z };       // the grid starts at the scene's lower corner
    long total = 1;
    for (int k = 0; k < 3; k++) {
        cell[k] = L;
        const float ext = bb.hi[k] - bb.lo[k];
        if (ext > L * (float)kMaxCellsPerAxis) cell[k] = ext / (float)kMaxCellsPerAxis;
        i0[k] = (int)floorf((bb.lo[k] - org[k]) / cell[k]);
        nc[k] = (int)floorf((bb.hi[k] - org[k]) / cell[k]) - i0[k] + 1;
        nc[k] = max(1, min(nc[k], kMaxCellsPerAxis + 1));
        total *= nc[k];
    }
    uint32_t n = 0;
    const uint32_t base = EMIT ? offset[f] : 0u;
    if (total == 1) {
        if (EMIT) { rbox[base] = bb; rface[base] = f; }
        n = 1;
    } else {
        for (int z = 0; z < nc[2]; z++) for (int y = 0; y < nc[1]; y++) for (int x = 0; x < nc[0]; x++) {
            const int id[3] = { x, y, z };
            float clo[3], chi[3];
            for (int k = 0; k < 3; k++) {
                clo[k] = id[k] == 0 ? bb.lo[k] : org[k] + (float)(i0[k] + id[k]) * cell[k];
                chi[k] = id[k] == nc[k] - 1 ? bb.hi[k] : org[k] + (float)(i0[k] + id[k] + 1) * cell[k];
            }
            Box6 part;
            if (!clip_tri_box(a, b, c, clo, chi, part)) continue;
            if (EMIT) { rbox[base + n] = part; rface[base + n] = f; }
            n++;
        }
        if (n == 0) {      // numerically empty everywhere (degenerate face): keep it whole
            if (EMIT) { rbox[base] = bb; rface[base] = f; }
            n = 1;
        }
    }
    if (!EMIT) count[f] = n;
}

__global__ void k_prim(uint32_t nf, float3 slo, float3 sinv, const Box6* __restrict__ pbox,
                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const Box6 b = pbox[f];
    const float cx = (0.5f * (b.lo[0] + b.hi[0]) - slo.x) * sinv.x;
    const float cy = (0.5f * (b.lo[1] + b.hi[1]) - slo.y) * sinv.y;
    const float cz = (0.5f * (b.lo[2] + b.hi[2]) - slo.z) * sinv.z;
    const unsigned long long ix = (unsigned long long)fminf(fmaxf(cx * 2097152.0f, 0.0f), 2097151.0f);
    const unsigned long long iy = (unsigned long long)fminf(fmaxf(cy * 2097152.0f, 0.0f), 2097151.0f);
    const unsigned long long iz = (unsigned long long)fminf(fmaxf(cz * 2097152.0f, 0.0f), 2097151.0f);
    keys[f] = expand21(ix) | (expand21(iy) << 1) | (expand21(iz) << 2);
    vals[f] = f;
}

// binary radix tree: inner nodes 0..n-2, leaves n-1+i (i = sorted position)
struct RNode { uint32_t left, right, parent; };

__device__ inline int delta(const unsigned long long* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void k_karras(const unsigned long long* __restrict__ keys, int n, RNode* __restrict__ nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t left = (lo == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    const uint32_t right = (hi == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    nodes[i].left = left; nodes[i].right = right;
    nodes[left].parent = (uint32_t)i;
    nodes[right].parent = (uint32_t)i;
    if (i == 0) nodes[0].parent = 0xFFFFFFFFu;
}

// per node (inner and leaf): box, number of triangles below, first sorted position
struct RInfo { Box6 box; uint32_t count, first; };

__global__ void k_refit(const RNode* __restrict__ nodes, const Box6* __restrict__ pbox, const uint32_t* __restrict__ sorted,
                        int n, RInfo* __restrict__ info, unsigned int* __restrict__ ticket)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t cur = (uint32_t)(n - 1 + i);
    RInfo me; me.box = pbox[sorted[i]]; me.count = 1; me.first = (uint32_t)i;
    info[cur] = me;
    if (n == 1) return;
    __threadfence();
    uint32_t p = nodes[cur].parent;
    while (p != 0xFFFFFFFFu) {
        if (atomicAdd(&ticket[p], 1u) == 0u) return;          // first arriver leaves, second one merges
        __threadfence();
        const RInfo a = info[nodes[p].left], b = info[nodes[p].right];
        RInfo m;
        for (int k = 0; k < 3; k++) { m.box.lo[k] = fminf(a.box.lo[k], b.box.lo[k]); m.box.hi[k] = fmaxf(a.box.hi[k], b.box.hi[k]); }
        m.count = a.count + b.count; m.first = min(a.first, b.first);
        info[p] = m;
        __threadfence();
        p = nodes[p].parent;
    }
}

struct Work { uint32_t rnode, node4, acc; };   // acc = stack entries needed above this node

__device__ inline float half_area(const Box6& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_collapse(const RNode* __restrict__ rn, const RInfo* __restrict__ info, int n_prims,
                           const Work* __restrict__ in, uint32_t n_in, Work* __restrict__ out, uint32_t* __restrict__ n_out,
                           Node4* __restrict__ nodes4, uint32_t* __restrict__ n_nodes4, uint32_t* __restrict__ max_need,
                           float inflate)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_in) return;
    const Work me = in[w];
    const uint32_t leaf_base = (uint32_t)(n_prims - 1);
    auto is_leaf = [&](uint32_t r) { return r >= leaf_base || info[r].count <= kMaxLeafTris; };
    uint32_t cand[4]; int nc = 0;
    if (is_leaf(me.rnode)) cand[nc++] = me.rnode;
    else { cand[nc++] = rn[me.rnode].left; cand[nc++] = rn[me.rnode].right; }
    while (nc < 4) {
        int pick = -1; float best = -1.0f;
        for (int i = 0; i < nc; i++) {
            if (is_leaf(cand[i])) continue;
            const float a = half_area(info[cand[i]].box);
            if (a > best) { best = a; pick = i; }
        }
        if (pick < 0) break;
        const uint32_t r = cand[pick];
        cand[pick] = rn[r].left; cand[nc++] = rn[r].right;
    }
    Node4 nd;
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 3; k++) nd.c[i].lo[k] = nd.c[i].hi[k] = kEmptyCoord;
        nd.c[i].ref = kEmptyRef; nd.c[i].pad = 0;
    }
    const uint32_t acc = me.acc + (uint32_t)(nc - 1);
    atomicMax(max_need, acc);
    for (int i = 0; i < nc; i++) {
        const RInfo ci = info[cand[i]];
        for (int k = 0; k < 3; k++) { nd.c[i].lo[k] = ci.box.lo[k] - inflate; nd.c[i].hi[k] = ci.box.hi[k] + inflate; }
        if (is_leaf(cand[i])) {
            nd.c[i].ref = kLeafFlag | ((ci.count - 1) << 28) | ci.first;
        } else {
            const uint32_t id = atomicAdd(n_nodes4, 1u);
            nd.c[i].ref = id;
            const uint32_t o = atomicAdd(n_out, 1u);
            out[o].rnode = cand[i]; out[o].node4 = id; out[o].acc = acc;
        }
    }
    nodes4[me.node4] = nd;
}

__global__ void k_tris(const float* __restrict__ verts, const uint32_t* __restrict__ faces, const uint32_t* __restrict__ fobj,
                       const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ rface, uint32_t n_refs, TriRec* __restrict__ tris)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_refs) return;
    const uint32_t f = rface[sorted[p]];
    const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + 0];
    const float* b = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
    const float* c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
    TriRec t;
    for (int k = 0; k < 3; k++) { t.v0[k] = a[k]; t.e1[k] = b[k] - a[k]; t.e2[k] = c[k] - a[k]; }   // same f32 ops as rr_bvh.cpp
    t.face = f; t.object = fobj ? fobj[f] : 0u; t.pad = 0;
    tris[p] = t;
}

template <typename T>
struct Tmp {
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
    ~Tmp() { if (p) (void)hipFree(p); }
};

}  // namespace

#define LB_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

// Builds into caller-owned device arrays (allocated here with hipMalloc; the caller frees them).
bool build_bvh4_gpu(const float* verts, size_t nv, const uint32_t* faces, size_t nf, const uint32_t* face_object,
                    Node4** d_nodes_out, size_t* n_nodes_out, TriRec** d_tris_out, size_t* n_tris_out,
                    uint32_t* depth_out, uint32_t* stack_need_out, float* inflate_out,
                    std::string& err, hipStream_t stream)
{
    *d_nodes_out = nullptr; *d_tris_out = nullptr; *n_nodes_out = 0; *n_tris_out = 0;
    if (nf == 0 || nf >= (1u << 28)) { err = "gpu builder: triangle count must be in [1, 2^28)"; return false; }
    for (size_t i = 0; i < 3 * nf; i++) if (faces[i] >= nv) { err = "rr_set_mesh: face index out of range"; return false; }
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    for (size_t i = 0; i < nv; i++) for (int k = 0; k < 3; k++) {
        const float x = verts[3 * i + k];
        if (!std::isfinite(x)) { err = "rr_set_mesh: non-finite vertex"; return false; }
        lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x);
    }
    float ext = 0.f, mag = 0.f;
    for (int k = 0; k < 3; k++) { ext = std::max(ext, hi[k] - lo[k]); mag = std::max(mag, std::max(std::fabs(lo[k]), std::fabs(hi[k]))); }
    const float inflate = 2e-5f * std::max(ext, mag) + 1e-6f;       // same padding rule as the host builder
    const float3 slo = make_float3(lo[0], lo[1], lo[2]);
    // ONE scale for the three axes: Morton cells are cubes.  A per-axis scale would spend every third bit on cutting a
    // 400 x 400 x 30 m map into ever flatter slabs (10M-triangle target: 0.45 -> 0.36 ms per frame, depth 19 -> 17)
    const float s1 = ext > 0.f ? 1.0f / ext : 0.f;
    const float3 sinv = make_float3(s1, s1, s1);
    const int TB = 256;
    const int nfb = (int)((nf + TB - 1) / TB);

    Tmp<float> d_verts; Tmp<uint32_t> d_faces, d_fobj, d_vals_in, d_vals, d_count, d_offset, d_rface; Tmp<Box6> d_pbox;
    Tmp<unsigned long long> d_keys_in, d_keys; Tmp<RNode> d_rn; Tmp<RInfo> d_info; Tmp<unsigned int> d_ticket;
    Tmp<Work> d_wa, d_wb; Tmp<uint32_t> d_cnt;    // cnt[0] = n_out, cnt[1] = n_nodes4, cnt[2] = max_need
    Tmp<char> d_sort_tmp, d_scan_tmp;
    LB_HIP(d_verts.alloc(3 * nv)); LB_HIP(d_faces.alloc(3 * nf));
    if (face_object) LB_HIP(d_fobj.alloc(nf));
    LB_HIP(d_count.alloc(nf)); LB_HIP(d_offset.alloc(nf));
    LB_HIP(hipMemcpyAsync(d_verts.p, verts, 3 * nv * sizeof(float), hipMemcpyHostToDevice, stream));
    LB_HIP(hipMemcpyAsync(d_faces.p, faces, 3 * nf * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    if (face_object) LB_HIP(hipMemcpyAsync(d_fobj.p, face_object, nf * sizeof(uint32_t), hipMemcpyHostToDevice, stream));

    // ---- 0. early split clipping: the smallest grid cell L that keeps the references within +25 % ----------------
    size_t scan_bytes = 0;
    LB_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_count.p, d_offset.p, 0u, nf, rocprim::plus<uint32_t>(), stream));
    LB_HIP(d_scan_tmp.alloc(scan_bytes));
    auto refs_for = [&](float L, size_t& total) -> bool {
        hipLaunchKernelGGL((k_split<false>), dim3(nfb), dim3(TB), 0, stream, d_verts.p, d_faces.p, (uint32_t)nf, L, slo, d_count.p,
                           (const uint32_t*)nullptr, (Box6*)nullptr, (uint32_t*)nullptr);
        if (rocprim::exclusive_scan(d_scan_tmp.p, scan_bytes, d_count.p, d_offset.p, 0u, nf, rocprim::plus<uint32_t>(), stream) != hipSuccess) return false;
        uint32_t last_off = 0, last_cnt = 0;
        if (hipMemcpyAsync(&last_off, d_offset.p + (nf - 1), 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
        if (hipMemcpyAsync(&last_cnt, d_count.p + (nf - 1), 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
        if (hipStreamSynchronize(stream) != hipSuccess) return false;
        total = (size_t)last_off + last_cnt;
        return true;
    };
    const size_t ref_limit = std::min<size_t>(nf + nf / 4 + 64, ((size_t)1 << 28) - 8);
    float L = std::max(ext, 1e-20f) * 2.0f;          // one cell holds the scene: no face is cut
    size_t n_refs = nf;
    if (!getenv("RR_LBVH_NO_SPLIT") && ext > 0.f) {
        float lo_l = L / 65536.0f, hi_l = L;        // refs(hi_l) <= limit always; refs grows as L shrinks
        for (int it = 0; it < 14; it++) {
            const float mid = std::sqrt(lo_l * hi_l);
            size_t tot = 0;
            if (!refs_for(mid, tot)) { err = "gpu builder: split counting failed"; return false; }
            if (tot <= ref_limit) hi_l = mid; else lo_l = mid;
        }
        L = hi_l;
    }
    if (!refs_for(L, n_refs)) { err = "gpu builder: split counting failed"; return false; }
    if (n_refs < nf || n_refs > ref_limit + nf) { err = "gpu builder: internal error (reference count)"; return false; }
    const int n = (int)n_refs;
    LB_HIP(d_pbox.alloc(n_refs)); LB_HIP(d_rface.alloc(n_refs));
    hipLaunchKernelGGL((k_split<true>), dim3(nfb), dim3(TB), 0, stream, d_verts.p, d_faces.p, (uint32_t)nf, L, slo, (uint32_t*)nullptr,
                       (const uint32_t*)d_offset.p, d_pbox.p, d_rface.p);

    LB_HIP(d_vals_in.alloc(n_refs)); LB_HIP(d_vals.alloc(n_refs));
    LB_HIP(d_keys_in.alloc(n_refs)); LB_HIP(d_keys.alloc(n_refs));
    LB_HIP(d_rn.alloc(2 * n_refs)); LB_HIP(d_info.alloc(2 * n_refs)); LB_HIP(d_ticket.alloc(n_refs));
    LB_HIP(d_wa.alloc(n_refs)); LB_HIP(d_wb.alloc(n_refs)); LB_HIP(d_cnt.alloc(4));

    hipLaunchKernelGGL(k_prim, dim3((n + TB - 1) / TB), dim3(TB), 0, stream, (uint32_t)n_refs, slo, sinv,
                       (const Box6*)d_pbox.p, d_keys_in.p, d_vals_in.p);
    size_t tmp_bytes = 0;
    LB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in.p, d_keys.p, d_vals_in.p, d_vals.p, n_refs, 0, 63, stream));
    LB_HIP(d_sort_tmp.alloc(tmp_bytes));
    LB_HIP(rocprim::radix_sort_pairs(d_sort_tmp.p, tmp_bytes, d_keys_in.p, d_keys.p, d_vals_in.p, d_vals.p, n_refs, 0, 63, stream));

    LB_HIP(hipMemsetAsync(d_ticket.p, 0, n_refs * sizeof(unsigned int), stream));
    if (n > 1) hipLaunchKernelGGL(k_karras, dim3((n - 1 + TB - 1) / TB), dim3(TB), 0, stream, d_keys.p, n, d_rn.p);
    hipLaunchKernelGGL(k_refit, dim3((n + TB - 1) / TB), dim3(TB), 0, stream, d_rn.p, d_pbox.p, d_vals.p, n, d_info.p, d_ticket.p);

    // output arrays: at most one 4-wide node per inner radix node (+ root)
    Node4* d_nodes = nullptr; TriRec* d_tris = nullptr;
    LB_HIP(hipMalloc((void**)&d_nodes, (n_refs + 1) * sizeof(Node4)));
    if (hipMalloc((void**)&d_tris, (n_refs + 4) * sizeof(TriRec)) != hipSuccess) { (void)hipFree(d_nodes); err = "hipMalloc(tris) failed"; return false; }
    hipLaunchKernelGGL(k_tris, dim3((n + TB - 1) / TB), dim3(TB), 0, stream, d_verts.p, d_faces.p, face_object ? d_fobj.p : nullptr,
                       d_vals.p, (const uint32_t*)d_rface.p, (uint32_t)n_refs, d_tris);

    const uint32_t root = (n == 1) ? 0u : 0u;   // radix root is inner node 0; a single triangle is leaf index n-1 = 0
    Work w0; w0.rnode = root; w0.node4 = 0; w0.acc = 0;
    uint32_t cnt[4] = { 0, 1, 0, 0 };
    bool ok = true;
    if (hipMemcpyAsync(d_wa.p, &w0, sizeof(Work), hipMemcpyHostToDevice, stream) != hipSuccess) ok = false;
    if (ok && hipMemcpyAsync(d_cnt.p, cnt, sizeof(cnt), hipMemcpyHostToDevice, stream) != hipSuccess) ok = false;
    uint32_t n_in = 1, depth = 0;
    Work* in = d_wa.p; Work* out = d_wb.p;
    while (ok && n_in > 0) {
        depth++;
        hipLaunchKernelGGL(k_collapse, dim3((n_in + TB - 1) / TB), dim3(TB), 0, stream, d_rn.p, d_info.p, n, in, n_in, out,
                           d_cnt.p, d_nodes, d_cnt.p + 1, d_cnt.p + 2, inflate);
        if (hipMemcpyAsync(cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { ok = false; break; }
        n_in = cnt[0];
        const uint32_t zero = 0;
        if (hipMemcpyAsync(d_cnt.p, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess) { ok = false; break; }
        std::swap(in, out);
        if (depth > 4096) { ok = false; break; }
    }
    if (ok && hipStreamSynchronize(stream) != hipSuccess) ok = false;
    if (!ok || hipGetLastError() != hipSuccess) {
        (void)hipFree(d_nodes); (void)hipFree(d_tris);
        err = "gpu BVH build failed";
        return false;
    }
    *d_nodes_out = d_nodes; *d_tris_out = d_tris; *n_nodes_out = cnt[1]; *n_tris_out = n_refs;
    *depth_out = depth; *stack_need_out = cnt[2]; *inflate_out = inflate;
    return true;
}

}  // namespace rr
