// rr_bvh.cpp -- binned-SAH BVH2 build (task-parallel) + collapse to BVH4.
#include "rr_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <future>
#include <limits>
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
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Node2 {
    Box box;
    uint32_t left;    // inner: index of left child (right = left + 1); leaf: first prim slot
    uint32_t count;   // 0 = inner
};

constexpr int kBins = 16;
constexpr uint32_t kParallelMin = 1u << 15;

struct Builder {
    const std::vector<Box>& pbox;
    const std::vector<float>& pcen;   // 3 per prim
    std::vector<uint32_t>& prim;
    std::vector<Node2>& nodes;
    std::atomic<uint32_t> next_node{1};
    std::atomic<int> tasks_left;

    Builder(const std::vector<Box>& pb, const std::vector<float>& pc, std::vector<uint32_t>& pr,
            std::vector<Node2>& nd, int threads)
        : pbox(pb), pcen(pc), prim(pr), nodes(nd), tasks_left(threads) {}

    void build(uint32_t ni, uint32_t first, uint32_t count) {
        Box nb, cb; nb.reset(); cb.reset();
        for (uint32_t i = first; i < first + count; i++) {
            const uint32_t f = prim[i];
            nb.grow(pbox[f]);
            cb.grow_pt(&pcen[3 * (size_t)f]);
        }
        nodes[ni].box = nb;
        if (count <= kMaxLeafTris) { nodes[ni].left = first; nodes[ni].count = count; return; }

        int best_axis = -1, best_split = 0;
        float best_cost = std::numeric_limits<float>::infinity();
        for (int ax = 0; ax < 3; ax++) {
            const float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.f)) continue;
            Box bb[kBins]; uint32_t bc[kBins];
            for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
            const float scale = (float)kBins / ext;
            for (uint32_t i = first; i < first + count; i++) {
                const uint32_t f = prim[i];
                int b = (int)((pcen[3 * (size_t)f + ax] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bc[b]++; bb[b].grow(pbox[f]);
            }
            float la[kBins]; uint32_t lc[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = 0; b < kBins - 1; b++) { if (bc[b]) acc.grow(bb[b]); c += bc[b]; la[b] = acc.half_area(); lc[b] = c; }
            acc.reset(); c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                if (bc[b]) acc.grow(bb[b]);
                c += bc[b];
                if (!lc[b - 1] || !c) continue;
                const float cost = la[b - 1] * (float)lc[b - 1] + acc.half_area() * (float)c;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b - 1; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = first + count / 2;
        } else {
            const float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
            const float scale = (float)kBins / ext;
            uint32_t i = first, j = first + count;
            while (i < j) {
                const uint32_t f = prim[i];
                int b = (int)((pcen[3 * (size_t)f + best_axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                if (b <= best_split) i++;
                else { j--; std::swap(prim[i], prim[j]); }
            }
            mid = i;
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        const uint32_t left = next_node.fetch_add(2);
        nodes[ni].left = left; nodes[ni].count = 0;
        const uint32_t nl = mid - first, nr = first + count - mid;
        if (count >= kParallelMin && tasks_left.fetch_sub(1) > 0) {
            auto fut = std::async(std::launch::async, [=] { build(left, first, nl); });
            build(left + 1, mid, nr);
            fut.get();
            tasks_left.fetch_add(1);
        } else {
            if (count >= kParallelMin) tasks_left.fetch_add(1);
            build(left, first, nl);
            build(left + 1, mid, nr);
        }
    }
};

struct Collapser {
    const std::vector<Node2>& n2;
    const std::vector<uint32_t>& prim;
    const std::vector<Box>& pbox;
    Bvh4& out;
    const float* verts; const uint32_t* faces; const uint32_t* fobj;
    float inflate;
    double sah = 0.0;

    uint32_t emit_leaf(const Node2& n) {
        const uint32_t first = (uint32_t)out.tris.size();
        for (uint32_t i = 0; i < n.count; i++) {
            const uint32_t f = prim[n.left + i];
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
                const uint32_t* face_object, Bvh4& out, std::string& err, int n_threads)
{
    const auto t0 = std::chrono::steady_clock::now();
    out = Bvh4();
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

    std::vector<Box> pbox(nf);
    std::vector<float> pcen(3 * nf);
    std::vector<uint32_t> prim(nf);
    Box scene; scene.reset();
    for (size_t f = 0; f < nf; f++) {
        Box b; b.reset();
        for (int v = 0; v < 3; v++) b.grow_pt(verts + 3 * (size_t)faces[3 * f + v]);
        pbox[f] = b;
        for (int k = 0; k < 3; k++) pcen[3 * f + k] = 0.5f * (b.lo[k] + b.hi[k]);
        prim[f] = (uint32_t)f;
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

    std::vector<Node2> n2(2 * nf + 1);
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    {
        Builder b(pbox, pcen, prim, n2, n_threads - 1);
        b.build(0, 0, (uint32_t)nf);
    }

    out.nodes.reserve(nf / 2 + 16);
    out.tris.reserve(nf);
    out.nodes.emplace_back();
    Collapser c{ n2, prim, pbox, out, verts, faces, face_object, out.inflate };
    auto r = c.collapse(0, 0);
    out.depth = r.first;
    out.stack_need = r.second;
    const float ra = scene.half_area();
    out.sah_cost = ra > 0.f ? c.sah / ra : 0.0;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out.tris.size() != nf) { err = "rr_set_mesh: internal error (leaf triangle count)"; return false; }
    return true;
}

}  // namespace rr
