// rr_bvh.cpp -- SAH BVH2 build over triangle REFERENCES with spatial splits (task-parallel) + collapse to BVH4.
//
// The scenes of the radar path mix triangle scales by two orders of magnitude (BASELINE.json configs 3-5: a
// 0.2 m terrain grid under 4..18 m building faces that overlap each other).  An object-split-only SAH build puts
// a building face into whatever terrain node its centroid falls in and inflates that node to the size of the
// face: every ray then walks 3-4 x the nodes a clean tree needs (measured: 41 BVH4 nodes per ray at 10M
// triangles, tree depth 16).  This builder therefore works on references {clipped box, face} and may cut a
// reference at a plane ("spatial split", Stich, Friedrich, Dietrich: Spatial Splits in Bounding Volume
// Hierarchies, HPG 2009): where the children of the best object split overlap by more than `sbvh_alpha` x the
// root area, a binned spatial split is evaluated too and taken when its SAH cost is lower; a straddling
// reference is cut in two (its triangle is clipped against the plane for tight boxes), unless keeping it whole on
// one side is cheaper ("reference unsplitting").  A face may therefore sit in several leaves; the nearest hit is
// the minimum over (t, face id), so duplicates cannot change a result (rr_kernels.hip, traverse()).
// The "area" of every SAH decision carries a vertical weight (BvhOptions::vertical_weight, rr_bvh.h): radar rays
// are mostly horizontal.  Parallelism: subtrees are tasks; inside a big node binning and partitioning run over
// fixed-size chunks of the reference list (merged in chunk order, so the tree does not depend on the thread count).
#include "rr_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <vector>
#include <thread>

namespace rr {
namespace {

float g_wz = 1.0f;   // set once per build (builds are not re-entrant across different weights)

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    void grow_pt(const float* p) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    void clip_to(const Box& b) {
        for (int k = 0; k < 3; k++) { lo[k] = std::max(lo[k], b.lo[k]); hi[k] = std::min(hi[k], b.hi[k]); }
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    // "area" of the SAH = expected projection of the box for the ray distribution the tree is built for: isotropic
    // rays see dx dy + dy dz + dz dx; g_wz < 1 discounts the horizontal face (see BvhOptions::vertical_weight)
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return g_wz * (dx * dy) + dy * dz + dz * dx;
    }
};

// a reference: (part of) one face, bounded by `box`
struct Ref {
    Box box;
    uint32_t face;
    uint32_t pad;
};

// Reference lists are created and dropped at every node: 10M references pass through about twenty levels of them.
// Straight from malloc that is a page-fault storm (glibc trims the heap and unmaps the freed lists, the next level
// faults them in again: 9.0 -> 5.7 s for the 10M-triangle map just by telling malloc not to trim), so their blocks
// come from a pool that lives as long as one build: power-of-two size classes from 64 KB to 64 MB, a free list per
// class; smaller and larger requests go to malloc.  Elements are default-initialised (a resize does not zero-fill).
struct BlockPool {
    static constexpr int kMinShift = 16, kMaxShift = 26, kClasses = kMaxShift - kMinShift + 1;
    std::mutex mu[kClasses];
    std::vector<void*> free_[kClasses];
    static int class_of(size_t bytes) {
        if (bytes < ((size_t)1 << kMinShift) || bytes > ((size_t)1 << kMaxShift)) return -1;
        int c = 0;
        while (((size_t)1 << (kMinShift + c)) < bytes) c++;
        return c;
    }
    void* get(size_t bytes) {
        const int c = class_of(bytes);
        if (c < 0) return std::malloc(bytes);
        {
            std::lock_guard<std::mutex> g(mu[c]);
            if (!free_[c].empty()) { void* p = free_[c].back(); free_[c].pop_back(); return p; }
        }
        return std::malloc((size_t)1 << (kMinShift + c));
    }
    void put(void* p, size_t bytes) {
        const int c = class_of(bytes);
        if (c < 0) { std::free(p); return; }
        std::lock_guard<std::mutex> g(mu[c]);
        free_[c].push_back(p);
    }
    ~BlockPool() { for (auto& f : free_) for (void* p : f) std::free(p); }
};
BlockPool* g_pool = nullptr;     // the pool of the build in progress (builds are serialised, see build_bvh4)

template <class T> struct PoolAlloc {
    using value_type = T;
    PoolAlloc() = default;
    template <class U> PoolAlloc(const PoolAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = g_pool ? g_pool->get(n * sizeof(T)) : std::malloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) { if (g_pool) g_pool->put(p, n * sizeof(T)); else std::free(p); }
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }       // default-init: no zero fill
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const PoolAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const PoolAlloc<U>&) const { return false; }
};
using RefVec = std::vector<Ref, PoolAlloc<Ref>>;

struct Node2 {
    Box box;
    uint32_t left;    // inner: index of left child (right = left + 1); leaf: first slot in leaf_faces
    uint32_t count;   // 0 = inner
};

constexpr int kBins = 16;
constexpr uint32_t kParallelMin = 1u << 15;

// Bounds of (triangle `f` clipped to the slab lo <= x[axis] <= hi), Sutherland-Hodgman on the <= 5-gon.
// The plane coordinate of a cut point is set exactly; the other two are interpolated in double.
struct Clipper {
    const float* verts; const uint32_t* faces;

    // The boxes of (triangle `f`) in the consecutive slabs [lo + b w, lo + (b + 1) w], b = b0..b1 of one axis, for the
    // EVALUATION of a spatial split (the first and last slab are open outwards).  One sweep: every inner slab boundary
    // cuts the triangle in a segment (two edge interpolations), a slab's part is bounded by the segments of its two
    // boundaries and the vertices between them.  About 4x cheaper than clipping the triangle against every slab anew;
    // no outward rounding (the cut itself, slab() below, is what has to cover the face).
    void slab_boxes(uint32_t f, int axis, float lo, float width, int b0, int b1, Box* out /* [b1 - b0 + 1] */) const {
        double v[3][3];
        for (int k = 0; k < 3; k++) {
            const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + k];
            v[k][0] = a[0]; v[k][1] = a[1]; v[k][2] = a[2];
        }
        int ia = 0, ib = 1, ic = 2;          // vertices in ascending order along the axis
        if (v[ia][axis] > v[ib][axis]) std::swap(ia, ib);
        if (v[ib][axis] > v[ic][axis]) std::swap(ib, ic);
        if (v[ia][axis] > v[ib][axis]) std::swap(ia, ib);
        const double* A = v[ia]; const double* Bv = v[ib]; const double* C = v[ic];
        auto cut = [&](const double* u, const double* w, double plane, double* q) {
            const double t = (plane - u[axis]) / (w[axis] - u[axis]);
            for (int k = 0; k < 3; k++) q[k] = u[k] + t * (w[k] - u[k]);
            q[axis] = plane;
        };
        auto grow = [](Box& bx, const double* q) {
            for (int k = 0; k < 3; k++) { bx.lo[k] = std::min(bx.lo[k], (float)q[k]); bx.hi[k] = std::max(bx.hi[k], (float)q[k]); }
        };
        const int n = b1 - b0 + 1;
        for (int i = 0; i < n; i++) out[i].reset();
        // vertices into their slabs
        for (int k = 0; k < 3; k++) {
            int b = (int)std::floor((v[k][axis] - (double)lo) / (double)width);
            b = std::min(std::max(b, b0), b1);
            grow(out[b - b0], v[k]);
        }
        // inner boundaries: plane b separates slab b - 1 from slab b
        for (int b = b0 + 1; b <= b1; b++) {
            const double plane = (double)lo + (double)b * (double)width;
            if (!(plane > A[axis] && plane < C[axis])) continue;
            double q1[3], q2[3];
            cut(A, C, plane, q1);
            if (plane < Bv[axis]) cut(A, Bv, plane, q2);
            else if (plane > Bv[axis]) cut(Bv, C, plane, q2);
            else { q2[0] = Bv[0]; q2[1] = Bv[1]; q2[2] = Bv[2]; }
            grow(out[b - 1 - b0], q1); grow(out[b - 1 - b0], q2);
            grow(out[b - b0], q1); grow(out[b - b0], q2);
        }
    }

    Box slab(uint32_t f, int axis, float lo, float hi) const {
        double p[8][3], q[8][3];
        int n = 3;
        for (int v = 0; v < 3; v++) {
            const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + v];
            p[v][0] = a[0]; p[v][1] = a[1]; p[v][2] = a[2];
        }
        for (int side = 0; side < 2; side++) {
            const double plane = side == 0 ? (double)lo : (double)hi;
            if (!std::isfinite(plane)) continue;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double* a = p[i]; const double* b = p[(i + 1) % n];
                const bool ina = side == 0 ? a[axis] >= plane : a[axis] <= plane;
                const bool inb = side == 0 ? b[axis] >= plane : b[axis] <= plane;
                if (ina) { q[m][0] = a[0]; q[m][1] = a[1]; q[m][2] = a[2]; m++; }
                if (ina != inb) {
                    const double t = (plane - a[axis]) / (b[axis] - a[axis]);
                    for (int k = 0; k < 3; k++) q[m][k] = a[k] + t * (b[k] - a[k]);
                    q[m][axis] = plane;
                    m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++) { p[i][0] = q[i][0]; p[i][1] = q[i][1]; p[i][2] = q[i][2]; }
            if (n == 0) break;
        }
        Box b; b.reset();
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                // outward rounding of the double -> float narrowing
                const float lo_f = std::nextafter((float)p[i][k], -std::numeric_limits<float>::infinity());
                const float hi_f = std::nextafter((float)p[i][k], std::numeric_limits<float>::infinity());
                b.lo[k] = std::min(b.lo[k], lo_f); b.hi[k] = std::max(b.hi[k], hi_f);
            }
        return b;
    }
};

// Node and leaf slots are handed out per THREAD in blocks: one shared counter bumped for every node and every leaf
// (8M read-modify-writes on two cache lines that 48 threads on two sockets fight over) was what made the 10M-triangle
// build 14 x slower than the 1M one.  A thread that ends leaves the rest of its blocks unused (holes are harmless:
// leaves are addressed by {first, count}, nodes by index).
constexpr uint32_t kNodeBlock = 512;      // nodes (pairs are taken from it)
constexpr uint64_t kLeafBlock = 4096;     // leaf face slots
struct LocalAlloc { uint64_t gen = 0; uint32_t node_next = 0, node_end = 0; uint64_t leaf_next = 0, leaf_end = 0, refs_out = 0, n_spatial = 0; };
thread_local LocalAlloc tl_alloc;
std::atomic<uint64_t> g_build_gen{0};

constexpr size_t kChunk = (size_t)1 << 16;   // references per work item of the intra-node loops (fixed: the tree does not depend on the thread count)

struct Builder {
    Clipper clip;
    Node2* nodes; size_t node_cap;
    uint32_t* leaf_faces; size_t leaf_cap;
    // every shared counter on a cache line of its own: `failed`, `budget` and the read-only members above are read at
    // every node, `tasks_left` is written by every parallel loop
    alignas(64) std::atomic<uint32_t> next_node{1};
    alignas(64) std::atomic<uint64_t> next_leaf{0};
    alignas(64) std::atomic<int64_t> budget;          // extra references spatial splits may still create
    alignas(64) std::atomic<int> tasks_left;          // threads that may still be started (subtree tasks and chunk workers)
    alignas(64) std::atomic<bool> failed{false};
    alignas(64) std::atomic<uint64_t> n_spatial{0};
    alignas(64) std::atomic<uint64_t> n_refs_out{0};
    alignas(64) uint64_t gen_pad_ = 0;
    uint64_t gen = ++g_build_gen;
    LocalAlloc& local() { if (tl_alloc.gen != gen) { tl_alloc = LocalAlloc(); tl_alloc.gen = gen; } return tl_alloc; }
    // statistics of this thread -> the builder (at the end of a subtree task / of the root call)
    void flush_local() { LocalAlloc& l = local(); n_refs_out += l.refs_out; n_spatial += l.n_spatial; l.refs_out = l.n_spatial = 0; }
    uint64_t alloc_leaf(size_t n) {
        LocalAlloc& l = local();
        if (l.leaf_next + n > l.leaf_end) { const uint64_t blk = std::max<uint64_t>(kLeafBlock, n); l.leaf_next = next_leaf.fetch_add(blk); l.leaf_end = l.leaf_next + blk; }
        const uint64_t f = l.leaf_next; l.leaf_next += n; return f;
    }
    uint32_t alloc_node_pair() {
        LocalAlloc& l = local();
        if (l.node_next + 2 > l.node_end) { l.node_next = next_node.fetch_add(kNodeBlock); l.node_end = l.node_next + kNodeBlock; }
        const uint32_t f = l.node_next; l.node_next += 2; return f;
    }
    float root_area = 0.f, alpha = 1e-5f, beta = 0.f;
    uint32_t max_leaf = kMaxLeafTris;

    struct ObjSplit { float cost; int axis, bin; Box lbox, rbox; };
    struct SpaSplit { float cost; int axis, bin; float pos; uint32_t nl, nr; };
    struct Bounds { Box nb, cb; void reset() { nb.reset(); cb.reset(); } void grow(const Bounds& o) { nb.grow(o.nb); cb.grow(o.cb); } };

    int acquire(int want) {
        int got = 0;
        while (got < want) {
            int cur = tasks_left.load();
            if (cur <= 0) break;
            const int take = std::min(cur, want - got);
            if (tasks_left.compare_exchange_weak(cur, cur - take)) got += take;
        }
        return got;
    }
    void release(int k) { if (k > 0) tasks_left.fetch_add(k); }

    // f(chunk, begin, end) over fixed-size chunks of [0, n), on this thread plus as many workers as are free
    template <class F> void for_chunks(size_t n, F&& f) {
        const size_t nchunks = (n + kChunk - 1) / kChunk;
        if (nchunks <= 1) { if (n) f((size_t)0, (size_t)0, n); return; }
        const int extra = acquire((int)std::min<size_t>(nchunks - 1, 255));
        std::atomic<size_t> next{0};
        // an exception of f (bad_alloc) must not leave a joinable thread behind (std::terminate): the first one is kept,
        // the remaining chunks are skipped, and it is rethrown after the joins -- rr_set_mesh turns it into an error code
        std::exception_ptr err; std::atomic<bool> failed_here{false};
        auto work = [&] {
            try {
                for (;;) {
                    const size_t c = next.fetch_add(1);
                    if (c >= nchunks || failed_here.load(std::memory_order_relaxed)) break;
                    f(c, c * kChunk, std::min(n, (c + 1) * kChunk));
                }
            } catch (...) { if (!failed_here.exchange(true)) err = std::current_exception(); }
        };
        std::vector<std::thread> th;
        try {
            th.reserve((size_t)extra);
            for (int i = 0; i < extra; i++) th.emplace_back(work);     // no thread to be had: the chunks run on the ones that started
        } catch (...) {}
        work();
        for (auto& t : th) t.join();
        release(extra);
        if (err) std::rethrow_exception(err);
    }
    static size_t n_chunks(size_t n) { return std::max<size_t>(1, (n + kChunk - 1) / kChunk); }

    void make_leaf(uint32_t ni, const RefVec& refs) {
        const uint64_t first = alloc_leaf(refs.size());
        if (first + refs.size() > leaf_cap) { failed = true; nodes[ni].left = 0; nodes[ni].count = 1; return; }
        for (size_t i = 0; i < refs.size(); i++) leaf_faces[first + i] = refs[i].face;
        nodes[ni].left = (uint32_t)first; nodes[ni].count = (uint32_t)refs.size();
        local().refs_out += refs.size();
    }

    static inline float centroid(const Ref& r, int ax) { return 0.5f * (r.box.lo[ax] + r.box.hi[ax]); }
    static inline void grow_bounds(Bounds& b, const Ref& r) {
        b.nb.grow(r.box);
        const float c[3] = { centroid(r, 0), centroid(r, 1), centroid(r, 2) };
        b.cb.grow_pt(c);
    }

    // binned SAH object split over the three axes: ONE pass over the references (chunk-parallel)
    struct ObjBins { Box bb[3][kBins]; uint32_t bc[3][kBins]; };
    ObjSplit best_object_split(const RefVec& refs, const Box& cb) {
        float lo[3], scale[3]; bool use[3];
        for (int ax = 0; ax < 3; ax++) {
            const float ext = cb.hi[ax] - cb.lo[ax];
            use[ax] = ext > 0.f; lo[ax] = cb.lo[ax]; scale[ax] = use[ax] ? (float)kBins / ext : 0.f;
        }
        std::vector<ObjBins> part(n_chunks(refs.size()));
        for_chunks(refs.size(), [&](size_t c, size_t b0, size_t b1) {
            ObjBins& B = part[c];
            for (int ax = 0; ax < 3; ax++) for (int b = 0; b < kBins; b++) { B.bb[ax][b].reset(); B.bc[ax][b] = 0; }
            for (size_t i = b0; i < b1; i++) {
                const Ref& r = refs[i];
                for (int ax = 0; ax < 3; ax++) {
                    if (!use[ax]) continue;
                    int b = (int)((centroid(r, ax) - lo[ax]) * scale[ax]);
                    b = std::min(std::max(b, 0), kBins - 1);
                    B.bc[ax][b]++; B.bb[ax][b].grow(r.box);
                }
            }
        });
        ObjBins& T = part[0];
        for (size_t c = 1; c < part.size(); c++)
            for (int ax = 0; ax < 3; ax++) for (int b = 0; b < kBins; b++) { T.bc[ax][b] += part[c].bc[ax][b]; if (part[c].bc[ax][b]) T.bb[ax][b].grow(part[c].bb[ax][b]); }
        ObjSplit best; best.cost = std::numeric_limits<float>::infinity(); best.axis = -1; best.bin = 0;
        for (int ax = 0; ax < 3; ax++) {
            if (!use[ax]) continue;
            const Box* bb = T.bb[ax]; const uint32_t* bc = T.bc[ax];
            Box lb[kBins]; uint32_t lc[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = 0; b < kBins - 1; b++) { if (bc[b]) acc.grow(bb[b]); c += bc[b]; lb[b] = acc; lc[b] = c; }
            acc.reset(); c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                if (bc[b]) acc.grow(bb[b]);
                c += bc[b];
                if (!lc[b - 1] || !c) continue;
                const float cost = lb[b - 1].half_area() * (float)lc[b - 1] + acc.half_area() * (float)c;
                if (cost < best.cost) { best.cost = cost; best.axis = ax; best.bin = b - 1; best.lbox = lb[b - 1]; best.rbox = acc; }
            }
        }
        return best;
    }

    // binned spatial split (Stich et al. §4.2): references are chopped into the bins they span; one pass, chunk-parallel
    struct SpaBins { Box bb[3][kBins]; uint32_t en[3][kBins], ex[3][kBins]; };
    SpaSplit best_spatial_split(const RefVec& refs, const Box& nb) {
        float lo[3], scale[3], width[3]; bool use[3];
        for (int ax = 0; ax < 3; ax++) {
            const float ext = nb.hi[ax] - nb.lo[ax];
            use[ax] = ext > 0.f; lo[ax] = nb.lo[ax]; scale[ax] = use[ax] ? (float)kBins / ext : 0.f; width[ax] = ext / (float)kBins;
        }
        std::vector<SpaBins> part(n_chunks(refs.size()));
        for_chunks(refs.size(), [&](size_t c, size_t i0, size_t i1) {
            SpaBins& B = part[c];
            for (int ax = 0; ax < 3; ax++) for (int b = 0; b < kBins; b++) { B.bb[ax][b].reset(); B.en[ax][b] = B.ex[ax][b] = 0; }
            for (size_t i = i0; i < i1; i++) {
                const Ref& r = refs[i];
                for (int ax = 0; ax < 3; ax++) {
                    if (!use[ax]) continue;
                    int b0 = (int)((r.box.lo[ax] - lo[ax]) * scale[ax]), b1 = (int)((r.box.hi[ax] - lo[ax]) * scale[ax]);
                    b0 = std::min(std::max(b0, 0), kBins - 1); b1 = std::min(std::max(b1, b0), kBins - 1);
                    B.en[ax][b0]++; B.ex[ax][b1]++;
                    if (b0 == b1) { B.bb[ax][b0].grow(r.box); continue; }
                    Box parts[kBins];
                    clip.slab_boxes(r.face, ax, lo[ax], width[ax], b0, b1, parts);
                    for (int b = b0; b <= b1; b++) {
                        Box& cbx = parts[b - b0];
                        cbx.clip_to(r.box);
                        if (cbx.valid()) B.bb[ax][b].grow(cbx);
                    }
                }
            }
        });
        SpaBins& T = part[0];
        for (size_t c = 1; c < part.size(); c++)
            for (int ax = 0; ax < 3; ax++) for (int b = 0; b < kBins; b++) { T.en[ax][b] += part[c].en[ax][b]; T.ex[ax][b] += part[c].ex[ax][b]; T.bb[ax][b].grow(part[c].bb[ax][b]); }
        SpaSplit best; best.cost = std::numeric_limits<float>::infinity(); best.axis = -1; best.bin = 0; best.pos = 0.f; best.nl = best.nr = 0;
        for (int ax = 0; ax < 3; ax++) {
            if (!use[ax]) continue;
            const Box* bb = T.bb[ax];
            Box lb[kBins]; uint32_t lc[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = 0; b < kBins - 1; b++) { acc.grow(bb[b]); c += T.en[ax][b]; lb[b] = acc; lc[b] = c; }
            acc.reset(); c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bb[b]);
                c += T.ex[ax][b];
                if (!lc[b - 1] || !c) continue;
                if (lc[b - 1] >= refs.size() || c >= refs.size()) continue;   // no progress on one side
                const float cost = lb[b - 1].half_area() * (float)lc[b - 1] + acc.half_area() * (float)c;
                if (cost < best.cost) { best.cost = cost; best.axis = ax; best.bin = b - 1; best.pos = lo[ax] + (float)b * width[ax]; best.nl = lc[b - 1]; best.nr = c; }
            }
        }
        return best;
    }

    // per-chunk output of a partition pass, concatenated in chunk order afterwards
    struct Piece { RefVec L, R; Bounds bl, br; int64_t uncut = 0; };
    void concat(std::vector<Piece>& pc, RefVec& L, RefVec& R, Bounds& bl, Bounds& br) {
        std::vector<size_t> ol(pc.size() + 1, 0), orr(pc.size() + 1, 0);
        bl.reset(); br.reset();
        for (size_t c = 0; c < pc.size(); c++) { ol[c + 1] = ol[c] + pc[c].L.size(); orr[c + 1] = orr[c] + pc[c].R.size(); bl.grow(pc[c].bl); br.grow(pc[c].br); }
        L.resize(ol.back()); R.resize(orr.back());
        for_chunks(pc.size() * kChunk, [&](size_t c, size_t, size_t) {
            if (c >= pc.size()) return;
            std::copy(pc[c].L.begin(), pc[c].L.end(), L.begin() + (std::ptrdiff_t)ol[c]);
            std::copy(pc[c].R.begin(), pc[c].R.end(), R.begin() + (std::ptrdiff_t)orr[c]);
            RefVec().swap(pc[c].L); RefVec().swap(pc[c].R);
        });
    }

    void recurse(uint32_t left, RefVec& L, const Bounds& bl, RefVec& R, const Bounds& br, size_t count) {
        std::future<void> fut; bool forked = false;
        if (count >= kParallelMin && acquire(1) == 1) {
            try { fut = std::async(std::launch::async, [this, left, &L, &bl] { build(left, L, bl); flush_local(); }); forked = true; }
            catch (const std::system_error&) { release(1); }          // no thread to be had: inline below
        }
        if (forked) {
            build(left + 1, R, br);       // (if this throws, the future's destructor still waits for the other half)
            fut.get();
            release(1);
        } else {
            build(left, L, bl);
            build(left + 1, R, br);
        }
    }

    // consumes `refs`; bd = bounds of the references and of their centroids
    void build(uint32_t ni, RefVec& refs, const Bounds& bd) {
        const size_t count = refs.size();
        const Box& nb = bd.nb; const Box& cb = bd.cb;
        nodes[ni].box = nb;
        if (count <= max_leaf || failed) { make_leaf(ni, refs); RefVec().swap(refs); return; }

        const ObjSplit os = best_object_split(refs, cb);
        bool spatial = false;
        SpaSplit ss; ss.cost = std::numeric_limits<float>::infinity(); ss.axis = -1;
        if (alpha >= 0.f && budget.load(std::memory_order_relaxed) > 0) {
            bool try_spatial = os.axis < 0;
            if (os.axis >= 0) {
                Box ov = os.lbox; ov.clip_to(os.rbox);
                try_spatial = ov.valid() && ov.half_area() > alpha * root_area && ov.half_area() > beta * nb.half_area();
            }
            if (try_spatial) {
                ss = best_spatial_split(refs, nb);
                spatial = ss.axis >= 0 && ss.cost < os.cost;
            }
        }

        RefVec L, R;
        Bounds bl, br;
        int64_t reserved = 0;
        if (spatial) {
            // budget: one extra reference per straddler that is really cut (reserved for all of them, the uncut ones are returned)
            reserved = (int64_t)ss.nl + (int64_t)ss.nr - (int64_t)count;
            if (budget.fetch_sub(reserved) - reserved < 0) { budget.fetch_add(reserved); reserved = 0; spatial = false; }
        }
        if (spatial) {
            const int ax = ss.axis; const float pos = ss.pos;
            // pass 1: boxes and counts of the references that lie on one side
            struct Side { Box lb, rb; uint64_t nl = 0, nr = 0; };
            std::vector<Side> sd(n_chunks(count));
            for_chunks(count, [&](size_t c, size_t i0, size_t i1) {
                Side& S = sd[c]; S.lb.reset(); S.rb.reset();
                for (size_t i = i0; i < i1; i++) {
                    const Ref& r = refs[i];
                    if (r.box.hi[ax] <= pos) { S.lb.grow(r.box); S.nl++; }
                    else if (r.box.lo[ax] >= pos) { S.rb.grow(r.box); S.nr++; }
                }
            });
            Box lb0, rb0; lb0.reset(); rb0.reset(); uint64_t nl0 = 0, nr0 = 0;
            for (const Side& S : sd) { lb0.grow(S.lb); rb0.grow(S.rb); nl0 += S.nl; nr0 += S.nr; }
            // pass 2: distribute; a straddler is cut, or kept whole on the cheaper side (reference unsplitting,
            // Stich et al. §4.4, evaluated against the boxes of pass 1 so that the result does not depend on the order)
            std::vector<Piece> pc(n_chunks(count));
            for_chunks(count, [&](size_t c, size_t i0, size_t i1) {
                Piece& P = pc[c]; P.bl.reset(); P.br.reset();
                for (size_t i = i0; i < i1; i++) {
                    const Ref& r = refs[i];
                    if (r.box.hi[ax] <= pos) { P.L.push_back(r); grow_bounds(P.bl, r); continue; }
                    if (r.box.lo[ax] >= pos) { P.R.push_back(r); grow_bounds(P.br, r); continue; }
                    Ref a = r, b = r;
                    a.box = clip.slab(r.face, ax, -std::numeric_limits<float>::infinity(), pos); a.box.clip_to(r.box);
                    b.box = clip.slab(r.face, ax, pos, std::numeric_limits<float>::infinity()); b.box.clip_to(r.box);
                    const bool va = a.box.valid(), vb = b.box.valid();
                    int where = 0;      // 0 cut, 1 whole left, 2 whole right
                    if (va && vb) {
                        Box l1 = lb0; l1.grow(a.box); Box r1 = rb0; r1.grow(b.box);
                        Box l2 = lb0; l2.grow(r.box); Box r2 = rb0; r2.grow(r.box);
                        const float nl = (float)nl0, nr = (float)nr0;
                        const float c_split = l1.half_area() * (nl + 1) + r1.half_area() * (nr + 1);
                        const float c_left = l2.half_area() * (nl + 1) + rb0.half_area() * nr;
                        const float c_right = lb0.half_area() * nl + r2.half_area() * (nr + 1);
                        if (!(c_split <= c_left && c_split <= c_right)) where = c_left <= c_right ? 1 : 2;
                    } else where = va ? 1 : 2;
                    if (where == 0) { P.L.push_back(a); grow_bounds(P.bl, a); P.R.push_back(b); grow_bounds(P.br, b); }
                    else if (where == 1) { P.L.push_back(r); grow_bounds(P.bl, r); P.uncut++; }
                    else { P.R.push_back(r); grow_bounds(P.br, r); P.uncut++; }
                }
            });
            int64_t uncut = 0;
            for (const Piece& P : pc) uncut += P.uncut;
            concat(pc, L, R, bl, br);
            if (uncut) budget.fetch_add(uncut);
            if (L.empty() || R.empty() || L.size() >= count || R.size() >= count) {
                budget.fetch_add(reserved - uncut);
                L.clear(); R.clear(); spatial = false;
            } else local().n_spatial++;
        }
        if (!spatial) {
            bool by_bin = os.axis >= 0;
            if (by_bin) {
                const float lo = cb.lo[os.axis], ext = cb.hi[os.axis] - cb.lo[os.axis];
                const float scale = (float)kBins / ext;
                std::vector<Piece> pc(n_chunks(count));
                for_chunks(count, [&](size_t c, size_t i0, size_t i1) {
                    Piece& P = pc[c]; P.bl.reset(); P.br.reset();
                    for (size_t i = i0; i < i1; i++) {
                        const Ref& r = refs[i];
                        int b = (int)((centroid(r, os.axis) - lo) * scale);
                        b = std::min(std::max(b, 0), kBins - 1);
                        if (b <= os.bin) { P.L.push_back(r); grow_bounds(P.bl, r); }
                        else { P.R.push_back(r); grow_bounds(P.br, r); }
                    }
                });
                concat(pc, L, R, bl, br);
                if (L.empty() || R.empty()) { L.clear(); R.clear(); by_bin = false; }
            }
            if (!by_bin) {      // all centroids in one bin: halve the list
                const size_t mid = count / 2;
                L.assign(refs.begin(), refs.begin() + (std::ptrdiff_t)mid);
                R.assign(refs.begin() + (std::ptrdiff_t)mid, refs.end());
                bl.reset(); br.reset();
                for (const Ref& r : L) grow_bounds(bl, r);
                for (const Ref& r : R) grow_bounds(br, r);
            }
        }
        RefVec().swap(refs);   // free before descending

        const uint32_t left = alloc_node_pair();
        if ((size_t)left + 2 > node_cap) { failed = true; nodes[ni].left = 0; nodes[ni].count = 1; return; }
        nodes[ni].left = left; nodes[ni].count = 0;
        recurse(left, L, bl, R, br, count);
    }
};

// BVH2 -> BVH4: a 4-wide node takes the two children of a binary node and keeps expanding the inner candidate of
// largest area until it holds four.  Two passes so that the (10M-triangle) output can be written in parallel:
// measure() sizes every 4-wide subtree, emit() writes nodes and leaf triangle records at the offsets that follow
// from the sizes.  Layout: the inner children of a node are consecutive, their own descendants follow child by
// child; the triangle records of a node's leaf children come first, then those of its inner children's subtrees.
struct Collapser {
    const Node2* n2;
    const uint32_t* leaf_faces;
    Bvh4& out;
    const float* verts; const uint32_t* faces; const uint32_t* fobj;
    float inflate;
    std::atomic<int> tasks_left{0};
    std::unique_ptr<uint32_t[]> n4, nt;     // per BVH2 index, valid where a 4-wide node is rooted: nodes / triangle records of its subtree

    struct M { uint32_t n4 = 0, nt = 0, depth = 0, need = 0; double sah = 0.0; };

    int gather(uint32_t n2_idx, uint32_t cand[4]) const {
        int nc = 0;
        const Node2& root = n2[n2_idx];
        if (root.count) { cand[nc++] = n2_idx; return nc; }
        cand[nc++] = root.left; cand[nc++] = root.left + 1;
        while (nc < 4) {
            int pick = -1; float best = -1.f;
            for (int i = 0; i < nc; i++) {
                if (n2[cand[i]].count) continue;
                const float a = n2[cand[i]].box.half_area();
                if (a > best) { best = a; pick = i; }
            }
            if (pick < 0) break;
            const uint32_t l = n2[cand[pick]].left;
            cand[pick] = l; cand[nc++] = l + 1;
        }
        return nc;
    }
    bool take_thread() { int cur = tasks_left.load(); while (cur > 0) { if (tasks_left.compare_exchange_weak(cur, cur - 1)) return true; } return false; }

    M measure(uint32_t n2_idx, int level) {
        uint32_t cand[4];
        const int nc = gather(n2_idx, cand);
        M m; m.n4 = 1;
        M sub[4]; std::future<M> fut[4]; bool async_[4] = { false, false, false, false };
        for (int i = 0; i < nc; i++) {
            const Node2& c = n2[cand[i]];
            if (c.count) { m.nt += c.count; m.sah += (double)c.box.half_area() * c.count; continue; }
            m.sah += (double)c.box.half_area();
            if (level < 4 && take_thread()) {
                try { fut[i] = std::async(std::launch::async, [this, ci = cand[i], level] { return measure(ci, level + 1); }); async_[i] = true; }
                catch (const std::system_error&) { tasks_left.fetch_add(1); }
            }
            if (!async_[i]) sub[i] = measure(cand[i], level + 1);
        }
        for (int i = 0; i < nc; i++) {
            if (n2[cand[i]].count) continue;
            if (async_[i]) { sub[i] = fut[i].get(); tasks_left.fetch_add(1); }
            m.n4 += sub[i].n4; m.nt += sub[i].nt; m.sah += sub[i].sah;
            m.depth = std::max(m.depth, sub[i].depth); m.need = std::max(m.need, sub[i].need);
        }
        m.depth += 1; m.need += (uint32_t)(nc > 0 ? nc - 1 : 0);
        n4[n2_idx] = m.n4; nt[n2_idx] = m.nt;
        return m;
    }

    void emit_leaf(const Node2& n, uint32_t first) {
        for (uint32_t i = 0; i < n.count; i++) {
            const uint32_t f = leaf_faces[n.left + i];
            const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + 0];
            const float* b = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
            const float* c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
            TriRec t;
            for (int k = 0; k < 3; k++) { t.v0[k] = a[k]; t.e1[k] = b[k] - a[k]; t.e2[k] = c[k] - a[k]; }
            t.face = f; t.object = fobj ? fobj[f] : 0u; t.pad = 0;
            out.tris[first + i] = t;
        }
    }

    // writes 4-wide node `self` (rooted at BVH2 node n2_idx); its descendants go to nodes [desc_base, ...), the
    // triangle records of its subtree to [tri_base, ...)
    void emit(uint32_t self, uint32_t n2_idx, uint32_t desc_base, uint32_t tri_base, int level) {
        uint32_t cand[4];
        const int nc = gather(n2_idx, cand);
        Node4 nd;
        // empty slot: a degenerate box at +3e38 -- the slab test then yields t = +-huge
        // on every axis and can never pass (inverted +-inf boxes would: min/max reorder them)
        for (int i = 0; i < 4; i++) {
            for (int k = 0; k < 3; k++) nd.c[i].lo[k] = nd.c[i].hi[k] = kEmptyCoord;
            nd.c[i].ref = kEmptyRef; nd.c[i].pad = 0;
        }
        int n_inner = 0;
        for (int i = 0; i < nc; i++) n_inner += n2[cand[i]].count ? 0 : 1;
        uint32_t child_self = desc_base, child_desc = desc_base + (uint32_t)n_inner, tri = tri_base;
        uint32_t cs[4], cd[4], ct[4];
        for (int i = 0; i < nc; i++) {       // leaf children first in the triangle array
            const Node2& c = n2[cand[i]];
            for (int k = 0; k < 3; k++) { nd.c[i].lo[k] = c.box.lo[k] - inflate; nd.c[i].hi[k] = c.box.hi[k] + inflate; }
            if (c.count) {
                emit_leaf(c, tri);
                nd.c[i].ref = kLeafFlag | ((c.count - 1) << 28) | tri;
                tri += c.count;
            }
        }
        for (int i = 0; i < nc; i++) {
            const Node2& c = n2[cand[i]];
            if (c.count) continue;
            cs[i] = child_self++; cd[i] = child_desc; ct[i] = tri;
            nd.c[i].ref = cs[i];
            child_desc += n4[cand[i]] - 1; tri += nt[cand[i]];
        }
        out.nodes[self] = nd;
        std::future<void> fut[4]; bool async_[4] = { false, false, false, false };
        for (int i = 0; i < nc; i++) {
            if (n2[cand[i]].count) continue;
            if (level < 4 && n4[cand[i]] > 4096 && take_thread()) {
                try { fut[i] = std::async(std::launch::async, [this, a = cs[i], b = cand[i], c2 = cd[i], d = ct[i], level] { emit(a, b, c2, d, level + 1); }); async_[i] = true; }
                catch (const std::system_error&) { tasks_left.fetch_add(1); }
            }
            if (!async_[i]) emit(cs[i], cand[i], cd[i], ct[i], level + 1);
        }
        for (int i = 0; i < nc; i++) if (async_[i]) { fut[i].get(); tasks_left.fetch_add(1); }
    }
};

}  // namespace

bool build_bvh4(const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                const uint32_t* face_object, Bvh4& out, std::string& err, int n_threads, const BvhOptions* opt_in)
{
    // one build at a time per process (the SAH weight and the block pool are per-build globals; a build is parallel inside)
    static std::mutex build_mu;
    std::lock_guard<std::mutex> build_lock(build_mu);
    const auto t0 = std::chrono::steady_clock::now();
    out = Bvh4();
    BvhOptions opt; if (opt_in) opt = *opt_in;
    if (const char* e = getenv("RR_BVH_ALPHA")) opt.sbvh_alpha = (float)atof(e);      // experiments (tools/treeq)
    if (const char* e = getenv("RR_BVH_BUDGET")) opt.ref_budget = (float)atof(e);
    if (const char* e = getenv("RR_BVH_BETA")) opt.sbvh_beta = (float)atof(e);
    if (const char* e = getenv("RR_BVH_WZ")) opt.vertical_weight = (float)atof(e);
    g_wz = std::min(1.0f, std::max(0.01f, opt.vertical_weight));
    if (nf >= (1u << 28)) { err = "rr_set_mesh: more than 2^28 triangles"; return false; }
    if (nf && (!verts || !faces)) { err = "rr_set_mesh: null vertex/face pointer"; return false; }
    for (size_t i = 0; i < 3 * nf; i++) {
        if (faces[i] >= nv) { err = "rr_set_mesh: face index out of range"; return false; }
    }
    for (size_t i = 0; i < 3 * nv; i++) {
        if (!std::isfinite(verts[i])) { err = "rr_set_mesh: non-finite vertex"; return false; }
    }
    if (nf == 0) {
        Node4 nd;
        for (int i = 0; i < 4; i++) {
            for (int k = 0; k < 3; k++) nd.c[i].lo[k] = nd.c[i].hi[k] = kEmptyCoord;
            nd.c[i].ref = kEmptyRef; nd.c[i].pad = 0;
        }
        out.nodes.push_back(nd);
        out.depth = 1; out.stack_need = 0;
        return true;
    }

    BlockPool pool;
    struct PoolScope { PoolScope(BlockPool* p) { g_pool = p; } ~PoolScope() { g_pool = nullptr; } } pool_scope(&pool);
    RefVec refs(nf);
    Box scene; scene.reset();
    for (size_t f = 0; f < nf; f++) {
        Box b; b.reset();
        for (int v = 0; v < 3; v++) b.grow_pt(verts + 3 * (size_t)faces[3 * f + v]);
        refs[f].box = b; refs[f].face = (uint32_t)f; refs[f].pad = 0;
        scene.grow(b);
    }
    float ext = 0.f, mag = 0.f;
    for (int k = 0; k < 3; k++) {
        ext = std::max(ext, scene.hi[k] - scene.lo[k]);
        mag = std::max(mag, std::max(std::fabs(scene.lo[k]), std::fabs(scene.hi[k])));
        out.scene_lo[k] = scene.lo[k]; out.scene_hi[k] = scene.hi[k];
    }
    // outward padding: covers the f32 error of slab test + Moeller-Trumbore so
    // that culling never removes a triangle the exact-order brute force accepts
    out.inflate = 2e-5f * std::max(ext, mag) + 1e-6f;

    // more than ~32-64 threads lose to allocator and page-fault contention (10M triangles on a 256-thread host:
    // 36.6 s with 1 thread, 9.7 s with 8, 6.0 s with 32 or 64, 18.8 s with 256)
    if (n_threads <= 0 && getenv("RR_BVH_THREADS")) n_threads = atoi(getenv("RR_BVH_THREADS"));
    if (n_threads <= 0) n_threads = (int)std::min(48u, std::max(1u, std::thread::hardware_concurrency()));
    const bool verbose = getenv("RR_BVH_VERBOSE") != nullptr;
    auto lap = [&](const char* what) { if (verbose) fprintf(stderr, "[rr bvh] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); };
    lap("validated, references made");
    const double budget_f = std::max(0.0, (double)opt.ref_budget) * (double)nf;
    const size_t max_refs = std::min<size_t>(nf + (size_t)budget_f + 16, (size_t)1 << 28);
    // every thread that allocates (the caller + one per subtree task: a task is started for nodes of >= kParallelMin
    // references only) may leave one block of each kind unused; untouched pages cost nothing
    const size_t alloc_threads = 2 * max_refs / kParallelMin + 64;
    const size_t node_cap = 2 * max_refs + 1 + alloc_threads * kNodeBlock;
    const size_t leaf_cap = max_refs + alloc_threads * kLeafBlock;
    std::unique_ptr<Node2[]> n2(new Node2[node_cap]);                 // default-initialised: pages are touched on use only
    std::unique_ptr<uint32_t[]> leaf_faces(new uint32_t[leaf_cap]);
    uint64_t n_leaf_refs = 0; uint32_t n_nodes_used = 1;
    {
        Builder b{ Clipper{ verts, faces } };
        b.nodes = n2.get(); b.node_cap = node_cap;
        b.leaf_faces = leaf_faces.get(); b.leaf_cap = leaf_cap;
        b.budget = (int64_t)(max_refs - nf - 16 > 0 ? max_refs - nf - 16 : 0);
        b.tasks_left = n_threads - 1;
        b.root_area = scene.half_area(); b.alpha = opt.sbvh_alpha; b.beta = opt.sbvh_beta;
        b.max_leaf = kMaxLeafTris;
        Builder::Bounds bd; bd.reset();
        for (const Ref& r : refs) Builder::grow_bounds(bd, r);
        b.build(0, refs, bd);
        b.flush_local();
        if (b.failed) { err = "rr_set_mesh: internal error (reference budget exceeded)"; return false; }
        n_leaf_refs = b.n_refs_out.load();
        n_nodes_used = b.next_node.load();
        out.spatial_splits = b.n_spatial.load();
    }

    lap("binary tree built");
    Collapser c{ n2.get(), leaf_faces.get(), out, verts, faces, face_object, out.inflate };
    c.tasks_left = n_threads - 1;
    const size_t n_bin = (size_t)n_nodes_used;
    c.n4.reset(new uint32_t[n_bin]); c.nt.reset(new uint32_t[n_bin]);
    const Collapser::M m = c.measure(0, 0);
    if (m.nt != n_leaf_refs) { err = "rr_set_mesh: internal error (leaf triangle count)"; return false; }
    out.nodes.resize(m.n4);
    out.tris.resize(m.nt);
    c.emit(0, 0, 1, 0, 0);
    lap("collapsed to 4-wide");
    out.depth = m.depth;
    out.stack_need = m.need;
    const float ra = scene.half_area();
    out.sah_cost = ra > 0.f ? m.sah / ra : 0.0;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

}  // namespace rr
