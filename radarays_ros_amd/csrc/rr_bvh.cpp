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
#include "rr_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <future>
#include <limits>
#include <memory>
#include <thread>

namespace rr {
namespace {

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
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

// a reference: (part of) one face, bounded by `box`
struct Ref {
    Box box;
    uint32_t face;
    uint32_t pad;
};

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

struct Builder {
    Clipper clip;
    Node2* nodes; size_t node_cap;
    uint32_t* leaf_faces; size_t leaf_cap;
    std::atomic<uint32_t> next_node{1};
    std::atomic<uint64_t> next_leaf{0};
    std::atomic<int64_t> budget;          // extra references spatial splits may still create
    std::atomic<int> tasks_left;
    std::atomic<bool> failed{false};
    std::atomic<uint64_t> n_spatial{0}, n_refs_out{0};
    float root_area = 0.f, alpha = 1e-5f;
    uint32_t max_leaf = kMaxLeafTris;

    struct ObjSplit { float cost; int axis, bin; Box lbox, rbox; };
    struct SpaSplit { float cost; int axis, bin; float pos; uint32_t nl, nr; };

    void make_leaf(uint32_t ni, const std::vector<Ref>& refs) {
        const uint64_t first = next_leaf.fetch_add(refs.size());
        if (first + refs.size() > leaf_cap) { failed = true; nodes[ni].left = 0; nodes[ni].count = 1; return; }
        for (size_t i = 0; i < refs.size(); i++) leaf_faces[first + i] = refs[i].face;
        nodes[ni].left = (uint32_t)first; nodes[ni].count = (uint32_t)refs.size();
        n_refs_out += refs.size();
    }

    static inline float centroid(const Ref& r, int ax) { return 0.5f * (r.box.lo[ax] + r.box.hi[ax]); }

    ObjSplit best_object_split(const std::vector<Ref>& refs, const Box& cb) const {
        ObjSplit best; best.cost = std::numeric_limits<float>::infinity(); best.axis = -1; best.bin = 0;
        for (int ax = 0; ax < 3; ax++) {
            const float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.f)) continue;
            Box bb[kBins]; uint32_t bc[kBins];
            for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
            const float scale = (float)kBins / ext;
            for (const Ref& r : refs) {
                int b = (int)((centroid(r, ax) - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bc[b]++; bb[b].grow(r.box);
            }
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

    SpaSplit best_spatial_split(const std::vector<Ref>& refs, const Box& nb) const {
        SpaSplit best; best.cost = std::numeric_limits<float>::infinity(); best.axis = -1; best.bin = 0; best.pos = 0.f; best.nl = best.nr = 0;
        for (int ax = 0; ax < 3; ax++) {
            const float lo = nb.lo[ax], ext = nb.hi[ax] - nb.lo[ax];
            if (!(ext > 0.f)) continue;
            const float scale = (float)kBins / ext, width = ext / (float)kBins;
            Box bb[kBins]; uint32_t entry[kBins], exit_[kBins];
            for (int b = 0; b < kBins; b++) { bb[b].reset(); entry[b] = exit_[b] = 0; }
            for (const Ref& r : refs) {
                int b0 = (int)((r.box.lo[ax] - lo) * scale), b1 = (int)((r.box.hi[ax] - lo) * scale);
                b0 = std::min(std::max(b0, 0), kBins - 1); b1 = std::min(std::max(b1, b0), kBins - 1);
                entry[b0]++; exit_[b1]++;
                if (b0 == b1) { bb[b0].grow(r.box); continue; }
                for (int b = b0; b <= b1; b++) {
                    const float plo = b == b0 ? -std::numeric_limits<float>::infinity() : lo + (float)b * width;
                    const float phi = b == b1 ? std::numeric_limits<float>::infinity() : lo + (float)(b + 1) * width;
                    Box cb = clip.slab(r.face, ax, plo, phi);
                    cb.clip_to(r.box);
                    if (cb.valid()) bb[b].grow(cb);
                }
            }
            Box lb[kBins]; uint32_t lc[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = 0; b < kBins - 1; b++) { acc.grow(bb[b]); c += entry[b]; lb[b] = acc; lc[b] = c; }
            acc.reset(); c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bb[b]);
                c += exit_[b];
                if (!lc[b - 1] || !c) continue;
                if (lc[b - 1] >= refs.size() || c >= refs.size()) continue;   // no progress on one side
                const float cost = lb[b - 1].half_area() * (float)lc[b - 1] + acc.half_area() * (float)c;
                if (cost < best.cost) { best.cost = cost; best.axis = ax; best.bin = b - 1; best.pos = lo + (float)b * width; best.nl = lc[b - 1]; best.nr = c; }
            }
        }
        return best;
    }

    void recurse(uint32_t left, std::vector<Ref>& L, std::vector<Ref>& R, size_t count) {
        if (count >= kParallelMin && tasks_left.fetch_sub(1) > 0) {
            auto fut = std::async(std::launch::async, [this, left, &L] { build(left, L); });
            build(left + 1, R);
            fut.get();
            tasks_left.fetch_add(1);
        } else {
            if (count >= kParallelMin) tasks_left.fetch_add(1);
            build(left, L);
            build(left + 1, R);
        }
    }

    // consumes `refs`
    void build(uint32_t ni, std::vector<Ref>& refs) {
        const size_t count = refs.size();
        Box nb, cb; nb.reset(); cb.reset();
        for (const Ref& r : refs) {
            nb.grow(r.box);
            const float c[3] = { centroid(r, 0), centroid(r, 1), centroid(r, 2) };
            cb.grow_pt(c);
        }
        nodes[ni].box = nb;
        if (count <= max_leaf || failed) { make_leaf(ni, refs); std::vector<Ref>().swap(refs); return; }

        const ObjSplit os = best_object_split(refs, cb);
        bool spatial = false;
        SpaSplit ss; ss.cost = std::numeric_limits<float>::infinity(); ss.axis = -1;
        if (alpha >= 0.f && budget.load(std::memory_order_relaxed) > 0) {
            bool try_spatial = os.axis < 0;
            if (os.axis >= 0) {
                Box ov = os.lbox; ov.clip_to(os.rbox);
                try_spatial = ov.valid() && ov.half_area() > alpha * root_area;
            }
            if (try_spatial) {
                ss = best_spatial_split(refs, nb);
                spatial = ss.axis >= 0 && ss.cost < os.cost;
            }
        }

        std::vector<Ref> L, R;
        if (spatial) {
            // budget: one extra reference per straddler that is really cut
            const int64_t extra = (int64_t)ss.nl + (int64_t)ss.nr - (int64_t)count;
            if (budget.fetch_sub(extra) - extra < 0) { budget.fetch_add(extra); spatial = false; }
        }
        if (spatial) {
            const int ax = ss.axis; const float pos = ss.pos;
            L.reserve(ss.nl); R.reserve(ss.nr);
            Box lb, rb; lb.reset(); rb.reset();
            std::vector<Ref> straddle;
            for (const Ref& r : refs) {
                if (r.box.hi[ax] <= pos) { L.push_back(r); lb.grow(r.box); }
                else if (r.box.lo[ax] >= pos) { R.push_back(r); rb.grow(r.box); }
                else straddle.push_back(r);
            }
            int64_t uncut = 0;
            for (const Ref& r : straddle) {
                Ref a = r, b = r;
                a.box = clip.slab(r.face, ax, -std::numeric_limits<float>::infinity(), pos); a.box.clip_to(r.box);
                b.box = clip.slab(r.face, ax, pos, std::numeric_limits<float>::infinity()); b.box.clip_to(r.box);
                const bool va = a.box.valid(), vb = b.box.valid();
                if (va && vb) {
                    // reference unsplitting (Stich et al. §4.4): cut, or keep whole on the cheaper side
                    Box l1 = lb; l1.grow(a.box); Box r1 = rb; r1.grow(b.box);
                    Box l2 = lb; l2.grow(r.box); Box r2 = rb; r2.grow(r.box);
                    const float nl = (float)L.size(), nr = (float)R.size();
                    const float c_split = l1.half_area() * (nl + 1) + r1.half_area() * (nr + 1);
                    const float c_left = l2.half_area() * (nl + 1) + rb.half_area() * nr;
                    const float c_right = lb.half_area() * nl + r2.half_area() * (nr + 1);
                    if (c_split <= c_left && c_split <= c_right) { L.push_back(a); R.push_back(b); lb = l1; rb = r1; }
                    else if (c_left <= c_right) { L.push_back(r); lb = l2; uncut++; }
                    else { R.push_back(r); rb = r2; uncut++; }
                } else if (va) { a.box = r.box; L.push_back(a); lb.grow(a.box); uncut++; }
                else { b.box = r.box; R.push_back(b); rb.grow(b.box); uncut++; }
            }
            if (uncut) budget.fetch_add(uncut);
            if (L.empty() || R.empty() || L.size() >= count || R.size() >= count) { L.clear(); R.clear(); spatial = false; }
            else n_spatial++;
        }
        if (!spatial) {
            size_t mid = 0;
            if (os.axis >= 0) {
                const float lo = cb.lo[os.axis], ext = cb.hi[os.axis] - cb.lo[os.axis];
                const float scale = (float)kBins / ext;
                size_t i = 0, j = count;
                while (i < j) {
                    int b = (int)((centroid(refs[i], os.axis) - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    if (b <= os.bin) i++;
                    else { j--; std::swap(refs[i], refs[j]); }
                }
                mid = i;
            }
            if (mid == 0 || mid == count) mid = count / 2;
            L.assign(refs.begin(), refs.begin() + mid);
            R.assign(refs.begin() + mid, refs.end());
        }
        std::vector<Ref>().swap(refs);   // free before descending

        const uint32_t left = next_node.fetch_add(2);
        if ((size_t)left + 2 > node_cap) { failed = true; nodes[ni].left = 0; nodes[ni].count = 1; return; }
        nodes[ni].left = left; nodes[ni].count = 0;
        recurse(left, L, R, count);
    }
};

struct Collapser {
    const Node2* n2;
    const uint32_t* leaf_faces;
    Bvh4& out;
    const float* verts; const uint32_t* faces; const uint32_t* fobj;
    float inflate;
    double sah = 0.0;

    uint32_t emit_leaf(const Node2& n) {
        const uint32_t first = (uint32_t)out.tris.size();
        for (uint32_t i = 0; i < n.count; i++) {
            const uint32_t f = leaf_faces[n.left + i];
            const float* a = verts + 3 * (size_t)faces[3 * (size_t)f + 0];
            const float* b = verts + 3 * (size_t)faces[3 * (size_t)f + 1];
            const float* c = verts + 3 * (size_t)faces[3 * (size_t)f + 2];
            TriRec t;
            for (int k = 0; k < 3; k++) { t.v0[k] = a[k]; t.e1[k] = b[k] - a[k]; t.e2[k] = c[k] - a[k]; }
            t.face = f; t.object = fobj ? fobj[f] : 0u; t.pad = 0;
            out.tris.push_back(t);
        }
        return kLeafFlag | ((n.count - 1) << 28) | first;
    }

    // returns {depth, stack_need} of the subtree rooted at BVH4 node `self`
    std::pair<uint32_t, uint32_t> collapse(uint32_t self, uint32_t n2_idx) {
        // gather up to 4 BVH2 nodes: expand the inner candidate of largest area
        uint32_t cand[4]; int nc = 0;
        const Node2& root = n2[n2_idx];
        if (root.count) { cand[nc++] = n2_idx; }
        else { cand[nc++] = root.left; cand[nc++] = root.left + 1; }
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
        Node4 nd;
        // empty slot: a degenerate box at +3e38 -- the slab test then yields t = +-huge
        // on every axis and can never pass (inverted +-inf boxes would: min/max reorder them)
        for (int i = 0; i < 4; i++) {
            for (int k = 0; k < 3; k++) nd.c[i].lo[k] = nd.c[i].hi[k] = kEmptyCoord;
            nd.c[i].ref = kEmptyRef; nd.c[i].pad = 0;
        }
        uint32_t depth = 0, need = 0;
        uint32_t inner_self[4];
        for (int i = 0; i < nc; i++) {
            const Node2& c = n2[cand[i]];
            for (int k = 0; k < 3; k++) { nd.c[i].lo[k] = c.box.lo[k] - inflate; nd.c[i].hi[k] = c.box.hi[k] + inflate; }
            if (c.count) {
                nd.c[i].ref = emit_leaf(c);
                sah += (double)c.box.half_area() * c.count;
                inner_self[i] = 0xFFFFFFFFu;
            } else {
                inner_self[i] = (uint32_t)out.nodes.size();
                out.nodes.emplace_back();
                nd.c[i].ref = inner_self[i];
                sah += (double)c.box.half_area();
            }
        }
        out.nodes[self] = nd;
        for (int i = 0; i < nc; i++) {
            if (inner_self[i] == 0xFFFFFFFFu) continue;
            auto r = collapse(inner_self[i], cand[i]);
            depth = std::max(depth, r.first);
            need = std::max(need, r.second);
        }
        return { depth + 1, need + (uint32_t)(nc > 0 ? nc - 1 : 0) };
    }
};

}  // namespace

bool build_bvh4(const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                const uint32_t* face_object, Bvh4& out, std::string& err, int n_threads, const BvhOptions* opt_in)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = Bvh4();
    BvhOptions opt; if (opt_in) opt = *opt_in;
    if (const char* e = getenv("RR_BVH_ALPHA")) opt.sbvh_alpha = (float)atof(e);      // experiments (tools/treeq)
    if (const char* e = getenv("RR_BVH_BUDGET")) opt.ref_budget = (float)atof(e);
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

    std::vector<Ref> refs(nf);
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

    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    const double budget_f = std::max(0.0, (double)opt.ref_budget) * (double)nf;
    const size_t max_refs = std::min<size_t>(nf + (size_t)budget_f + 16, (size_t)1 << 28);
    const size_t node_cap = 2 * max_refs + 1;
    std::unique_ptr<Node2[]> n2(new Node2[node_cap]);                 // default-initialised: pages are touched on use only
    std::unique_ptr<uint32_t[]> leaf_faces(new uint32_t[max_refs]);
    uint64_t n_leaf_refs = 0;
    {
        Builder b{ Clipper{ verts, faces } };
        b.nodes = n2.get(); b.node_cap = node_cap;
        b.leaf_faces = leaf_faces.get(); b.leaf_cap = max_refs;
        b.budget = (int64_t)(max_refs - nf - 16 > 0 ? max_refs - nf - 16 : 0);
        b.tasks_left = n_threads - 1;
        b.root_area = scene.half_area(); b.alpha = opt.sbvh_alpha;
        b.max_leaf = kMaxLeafTris;
        b.build(0, refs);
        if (b.failed) { err = "rr_set_mesh: internal error (reference budget exceeded)"; return false; }
        n_leaf_refs = b.next_leaf.load();
        out.spatial_splits = b.n_spatial.load();
    }

    out.nodes.reserve(n_leaf_refs / 2 + 16);
    out.tris.reserve(n_leaf_refs);
    out.nodes.emplace_back();
    Collapser c{ n2.get(), leaf_faces.get(), out, verts, faces, face_object, out.inflate };
    auto r = c.collapse(0, 0);
    out.depth = r.first;
    out.stack_need = r.second;
    const float ra = scene.half_area();
    out.sah_cost = ra > 0.f ? c.sah / ra : 0.0;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out.tris.size() != n_leaf_refs) { err = "rr_set_mesh: internal error (leaf triangle count)"; return false; }
    return true;
}

}  // namespace rr
