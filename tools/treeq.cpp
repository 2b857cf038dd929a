// tools/treeq.cpp -- BVH-quality study WITHOUT a GPU: builds the BVH4 with the product's host builder
// (csrc/rr_bvh.cpp) and walks it with the stepping rule of the HIP kernel (rr_kernels.hip traverse():
// nearest hit child first, the others pushed far-first, culled by t*(1+1e-4)+1e-3 at test time only) for a
// logged set of rays; prints node / leaf visits per ray, an estimate of the wave iterations (16 rays per
// wave, a wave runs as long as its slowest ray) and the algorithmic bytes per ray.
//
//   python tools/treeq_dump.py <config id> <passes> <out dir>     (scene + rays: the oracle's ORC_RAYLOG)
//   g++ -O2 -std=c++17 -ffp-contract=off -I radarays_ros_amd/csrc tools/treeq.cpp radarays_ros_amd/csrc/rr_bvh.cpp -o /tmp/treeq -lpthread
//   RR_BVH_ALPHA=1e-5 RR_BVH_BUDGET=1 /tmp/treeq <out dir>
//   TREEQ_STACKLESS=1: the stack-free walk north_star names (round 6) -- parent links, children visited in key order, a node
//   re-fetched and re-tested every time the walk returns to it; prints its fetch counts beside the stack walk's.
//   TREEQ_CULL_POP=1: with the later passes' cull at pop time; TREEQ_PREDICT / TREEQ_SORTED: ray-order studies; TREEQ_SLAB: an extra
//   pair of planes per child (all of them studies whose results are in DESIGN_EXPERIMENTS.md)
#include "rr_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace rr;

template <typename T>
static std::vector<T> slurp(const std::string& path)
{
    std::vector<T> v;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) { fprintf(stderr, "short read %s\n", path.c_str()); exit(2); }
    fclose(f);
    return v;
}

struct RayRec { int32_t az, pass; float o[3], d[3]; uint32_t parent, mat; };
struct Cost { unsigned merged = 0; unsigned nodes = 0, leaves = 0, tris = 0; float t = -1.f; uint32_t face = 0xFFFFFFFFu; unsigned nh[5] = {0,0,0,0,0}; unsigned leaves_after_hit = 0, nodes_after_hit = 0; };

// experiment (TREEQ_SLAB): one more pair of planes per child -- normal n (area-weighted mean normal of the subtree's
// triangles), [smin, smax] = extent of n . p over the subtree's triangles clipped to the child's box
struct Slab { float n[3]; float smin, smax; };
static std::vector<Slab> g_slab;       // [node][4]
static bool g_use_slab = false;

static Cost trace(const Bvh4& B, const float o[3], const float d[3], float range_max)
{
    Cost c;
    float inv[3], oo[3];
    for (int k = 0; k < 3; k++) {
        float dk = d[k];
        if (std::fabs(dk) < 1e-20f) dk = std::copysign(1e-20f, dk);
        inv[k] = 1.0f / dk; oo[k] = -o[k] * inv[k];
    }
    uint32_t stack[256]; float stack_t[256]; int sp = 0; static const bool cull_pop = getenv("TREEQ_CULL_POP") != nullptr;
    uint32_t cur = 0;
    float best_t = INFINITY; uint32_t best_face = 0xFFFFFFFFu;
    float tcull = range_max * 1.0001f + 1e-3f;
    bool after_leaf = false;       // TREEQ_MERGE study: a node step that directly follows a leaf step (always through a pop) could share its iteration
    while (true) {
        if (!(cur & kLeafFlag)) {
            c.nodes++;
            if (after_leaf) c.merged++;
            after_leaf = false;
            const Node4& n = B.nodes[cur];
            uint32_t key[4]; uint32_t ref[4]; int nh = 0;
            for (int q = 0; q < 4; q++) {
                const Child4& ch = n.c[q];
                float tn = 0.f, tf = INFINITY;
                for (int k = 0; k < 3; k++) {
                    const float a = std::fma(ch.lo[k], inv[k], oo[k]), b = std::fma(ch.hi[k], inv[k], oo[k]);
                    tn = std::max(tn, std::min(a, b)); tf = std::min(tf, std::max(a, b));
                }
                if (g_use_slab && ch.ref != kEmptyRef) {
                    const Slab& sl = g_slab[(size_t)cur * 4 + q];
                    const float dn = sl.n[0] * d[0] + sl.n[1] * d[1] + sl.n[2] * d[2], on = sl.n[0] * o[0] + sl.n[1] * o[1] + sl.n[2] * o[2];
                    if (std::fabs(dn) > 1e-12f) {
                        const float a = (sl.smin - on) / dn, b = (sl.smax - on) / dn;
                        tn = std::max(tn, std::min(a, b)); tf = std::min(tf, std::max(a, b));
                    } else if (on < sl.smin || on > sl.smax) tf = -1.f;
                }
                if (tn <= std::min(tf, tcull) && ch.ref != kEmptyRef) {
                    uint32_t bits; memcpy(&bits, &tn, 4);
                    key[nh] = (bits & ~3u) | (uint32_t)q; ref[nh] = ch.ref; nh++;
                }
            }
            c.nh[nh]++;
            if (best_face != 0xFFFFFFFFu) c.nodes_after_hit++;
            if (nh > 0) {
                // sort ascending by key
                for (int i = 1; i < nh; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
                for (int i = nh - 1; i >= 1; i--) { uint32_t kb = key[i] & ~3u; float tk; memcpy(&tk, &kb, 4); stack_t[sp] = tk; stack[sp++] = ref[i]; }
                cur = ref[0];
                continue;
            }
        } else {
            after_leaf = true;
            c.leaves++;
            if (best_face != 0xFFFFFFFFu) c.leaves_after_hit++;
            const uint32_t first = cur & 0x0FFFFFFFu, cnt = ((cur >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                c.tris++;
                const TriRec& T = B.tris[first + i];
                const float* v0 = T.v0; const float* e1 = T.e1; const float* e2 = T.e2;
                const float pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
                const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                const float invd = 1.0f / det;
                const float tv[3] = { o[0] - v0[0], o[1] - v0[1], o[2] - v0[2] };
                const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * invd;
                const float qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
                const float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * invd;
                const float tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * invd;
                const bool ok = det != 0.0f && u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && tt > 0.0f && tt <= range_max;
                if (ok && (tt < best_t || (tt == best_t && T.face < best_face))) {
                    best_t = tt; best_face = T.face; tcull = std::fma(tt, 1.0001f, 1e-3f);
                }
            }
        }
        // what k_trace's later passes do: an entry carries the upper 16 bits of its box's entry distance (a lower bound);
        // it is dropped at pop time when that bound lies beyond the cull distance (TREEQ_CULL_POP=exact: full precision)
        if (cull_pop) {
            static const bool exact = std::string(getenv("TREEQ_CULL_POP")) == "exact";
            auto k16 = [](float x) { uint32_t b; memcpy(&b, &x, 4); return b >> 16; };
            while (sp > 0 && (exact ? stack_t[sp - 1] > tcull : k16(stack_t[sp - 1]) > k16(tcull))) sp--;
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (best_face != 0xFFFFFFFFu) { c.t = best_t; c.face = best_face; }
    return c;
}

// The stack-free walk (north_star: "stackless"; round 6 study).  No per-ray memory at all: the state is (node, key of the
// child the walk came back from).  Children of a node are visited in increasing KEY order -- key = (bits(tmin) & ~3) | slot,
// the very key the stack walk sorts by, so the visiting order is the same nearest-first order -- and "the next child" is the
// hit child with the smallest key above the previous one; that previous key is recomputed from the node's own boxes when the
// walk returns (the slab test is a pure function of ray and box), so nothing has to be remembered per level.  Going up takes
// the parent link in the node's spare word, and the parent is fetched and tested AGAIN: an internal node costs one fetch on
// the way down and one for every child NODE the walk returns from (leaves are handled while the node's four keys are still in
// registers).  Culling is exact and late: a child is tested against the cull distance of the moment it is picked.
static Cost trace_stackless(const Bvh4& B, const std::vector<uint32_t>& parent, const float o[3], const float d[3], float range_max)
{
    Cost c;
    float inv[3], oo[3];
    for (int k = 0; k < 3; k++) {
        float dk = d[k];
        if (std::fabs(dk) < 1e-20f) dk = std::copysign(1e-20f, dk);
        inv[k] = 1.0f / dk; oo[k] = -o[k] * inv[k];
    }
    float best_t = INFINITY; uint32_t best_face = 0xFFFFFFFFu;
    float tcull = range_max * 1.0001f + 1e-3f;
    uint32_t node = 0; int from_slot = -1;
    while (true) {
        c.nodes++;                                   // a fetch of the node's four child records (first visit or return)
        if (from_slot >= 0) c.merged++;              // (re-used counter: RE-fetches)
        const Node4& n = B.nodes[node];
        float tn[4], tf[4]; uint32_t key[4];
        for (int q = 0; q < 4; q++) {
            const Child4& ch = n.c[q];
            tn[q] = 0.f; tf[q] = INFINITY;
            for (int k = 0; k < 3; k++) {
                const float a = std::fma(ch.lo[k], inv[k], oo[k]), b = std::fma(ch.hi[k], inv[k], oo[k]);
                tn[q] = std::max(tn[q], std::min(a, b)); tf[q] = std::min(tf[q], std::max(a, b));
            }
            uint32_t bits; memcpy(&bits, &tn[q], 4);
            key[q] = (bits & ~3u) | (uint32_t)q;
        }
        long prev = from_slot >= 0 ? (long)key[from_slot] : -1;
        bool descended = false;
        while (true) {
            int pick = -1;
            for (int q = 0; q < 4; q++)
                if (n.c[q].ref != kEmptyRef && tn[q] <= std::min(tf[q], tcull) && (long)key[q] > prev && (pick < 0 || key[q] < key[pick])) pick = q;
            if (pick < 0) break;
            prev = (long)key[pick];
            const uint32_t ref = n.c[pick].ref;
            if (!(ref & kLeafFlag)) { node = ref; from_slot = -1; descended = true; break; }
            c.leaves++;
            const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                c.tris++;
                const TriRec& T = B.tris[first + i];
                const float* v0 = T.v0; const float* e1 = T.e1; const float* e2 = T.e2;
                const float pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
                const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                const float invd = 1.0f / det;
                const float tv[3] = { o[0] - v0[0], o[1] - v0[1], o[2] - v0[2] };
                const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * invd;
                const float qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
                const float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * invd;
                const float tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * invd;
                const bool ok = det != 0.0f && u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && tt > 0.0f && tt <= range_max;
                if (ok && (tt < best_t || (tt == best_t && T.face < best_face))) { best_t = tt; best_face = T.face; tcull = std::fma(tt, 1.0001f, 1e-3f); }
            }
        }
        if (descended) continue;
        if (node == 0) break;
        const uint32_t pl = parent[node];
        node = pl >> 2; from_slot = (int)(pl & 3u);
    }
    if (best_face != 0xFFFFFFFFu) { c.t = best_t; c.face = best_face; }
    return c;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: treeq <dir with verts.f32 faces.u32 rays.bin> [threads]\n"); return 2; }
    const std::string dir = argv[1];
    const int threads = argc > 2 ? atoi(argv[2]) : 0;
    const auto verts = slurp<float>(dir + "/verts.f32");
    const auto faces = slurp<uint32_t>(dir + "/faces.u32");
    const auto rays = slurp<RayRec>(dir + "/rays.bin");
    printf("scene: %zu vertices, %zu faces; %zu rays\n", verts.size() / 3, faces.size() / 3, rays.size());

    Bvh4 B; std::string err;
    if (!build_bvh4(verts.data(), verts.size() / 3, faces.data(), faces.size() / 3, nullptr, B, err, threads)) {
        fprintf(stderr, "build failed: %s\n", err.c_str()); return 1;
    }
    printf("bvh4: %zu nodes, %zu leaf triangle records (x%.3f), depth %u, stack %u, spatial splits %llu, sah %.2f, build %.2f s\n",
           B.nodes.size(), B.tris.size(), (double)B.tris.size() / (faces.size() / 3), B.depth, B.stack_need,
           (unsigned long long)B.spatial_splits, B.sah_cost, B.build_seconds);

    if (getenv("TREEQ_SLAB")) {
        g_slab.assign(B.nodes.size() * 4, Slab{ { 0, 0, 1 }, -INFINITY, INFINITY });
        const float pad = B.inflate;
        // pass 1: area-weighted normal sum per child (post-order); pass 2: extents along the chosen normal
        std::vector<double> nsum(B.nodes.size() * 4 * 3, 0.0);
        struct Rec { static void sum(const Bvh4& B, uint32_t ref, double out[3], std::vector<double>& nsum) {
            out[0] = out[1] = out[2] = 0;
            if (ref & kLeafFlag) {
                const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
                for (uint32_t i = 0; i < cnt; i++) { const TriRec& T = B.tris[first + i];
                    double n[3] = { (double)T.e1[1] * T.e2[2] - (double)T.e1[2] * T.e2[1], (double)T.e1[2] * T.e2[0] - (double)T.e1[0] * T.e2[2], (double)T.e1[0] * T.e2[1] - (double)T.e1[1] * T.e2[0] };
                    // orient consistently (upper hemisphere, then +x, +y) so that the two sides of a wall do not cancel
                    if (n[2] < 0 || (n[2] == 0 && (n[0] < 0 || (n[0] == 0 && n[1] < 0)))) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
                    for (int k = 0; k < 3; k++) out[k] += n[k]; }
                return;
            }
            const Node4& n = B.nodes[ref];
            for (int q = 0; q < 4; q++) { if (n.c[q].ref == kEmptyRef) continue; double s[3]; sum(B, n.c[q].ref, s, nsum);
                for (int k = 0; k < 3; k++) { nsum[((size_t)ref * 4 + q) * 3 + k] = s[k]; out[k] += s[k]; } }
        }
        static void ext(const Bvh4& B, uint32_t ref, const float n[3], const float lo[3], const float hi[3], float& mn, float& mx) {
            if (ref & kLeafFlag) {
                const uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
                for (uint32_t i = 0; i < cnt; i++) { const TriRec& T = B.tris[first + i];
                    float a = INFINITY, b = -INFINITY;
                    for (int v = 0; v < 3; v++) { float p[3]; for (int k = 0; k < 3; k++) p[k] = T.v0[k] + (v == 1 ? T.e1[k] : v == 2 ? T.e2[k] : 0.f);
                        const float sdot = n[0] * p[0] + n[1] * p[1] + n[2] * p[2]; a = std::min(a, sdot); b = std::max(b, sdot); }
                    mn = std::min(mn, a); mx = std::max(mx, b); }
                return;
            }
            const Node4& nd = B.nodes[ref];
            for (int q = 0; q < 4; q++) if (nd.c[q].ref != kEmptyRef) ext(B, nd.c[q].ref, n, lo, hi, mn, mx);
        } };
        double tot[3]; Rec::sum(B, 0, tot, nsum);
        size_t done = 0;
        for (size_t ni = 0; ni < B.nodes.size(); ni++) for (int q = 0; q < 4; q++) {
            const Child4& ch = B.nodes[ni].c[q];
            if (ch.ref == kEmptyRef) continue;
            double* sN = &nsum[(ni * 4 + q) * 3];
            if (ch.ref & kLeafFlag) { double t3[3]; Rec::sum(B, ch.ref, t3, nsum); sN[0] = t3[0]; sN[1] = t3[1]; sN[2] = t3[2]; }
            const double len = std::sqrt(sN[0] * sN[0] + sN[1] * sN[1] + sN[2] * sN[2]);
            Slab& sl = g_slab[ni * 4 + q];
            if (len <= 0) continue;
            for (int k = 0; k < 3; k++) sl.n[k] = (float)(sN[k] / len);
            float mn = INFINITY, mx = -INFINITY;
            Rec::ext(B, ch.ref, sl.n, ch.lo, ch.hi, mn, mx);
            // the box's own extent along n bounds the clipped references
            float bmn = 0, bmx = 0; for (int k = 0; k < 3; k++) { bmn += sl.n[k] * (sl.n[k] >= 0 ? ch.lo[k] : ch.hi[k]); bmx += sl.n[k] * (sl.n[k] >= 0 ? ch.hi[k] : ch.lo[k]); }
            sl.smin = std::max(mn, bmn) - pad; sl.smax = std::min(mx, bmx) + pad;
            done++;
        }
        g_use_slab = true;
        printf("slabs: %zu children\n", done);
    }
    // optional cross-check of the hits against a reference file written by an earlier run
    std::vector<uint32_t> ref_face; std::vector<float> ref_t;
    if (FILE* f = fopen((dir + "/hits.ref").c_str(), "rb")) {
        fclose(f);
        const auto raw = slurp<uint32_t>(dir + "/hits.ref");
        ref_face.assign(raw.begin(), raw.begin() + raw.size() / 2);
        ref_t.resize(raw.size() / 2); memcpy(ref_t.data(), raw.data() + raw.size() / 2, ref_t.size() * 4);
    }

    std::vector<Cost> cost(rays.size());
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)rays.size(); i++) cost[i] = trace(B, rays[i].o, rays[i].d, 1000.0f);
    if (getenv("TREEQ_STACKLESS")) {
        std::vector<uint32_t> parent(B.nodes.size(), 0xFFFFFFFFu);
        for (size_t ni = 0; ni < B.nodes.size(); ni++) for (int q = 0; q < 4; q++) {
            const uint32_t r = B.nodes[ni].c[q].ref;
            if (r != kEmptyRef && !(r & kLeafFlag)) parent[r] = (uint32_t)(ni << 2) | (uint32_t)q;
        }
        std::vector<Cost> sl(rays.size());
#pragma omp parallel for schedule(dynamic, 256)
        for (long i = 0; i < (long)rays.size(); i++) sl[i] = trace_stackless(B, parent, rays[i].o, rays[i].d, 1000.0f);
        size_t bad = 0; double n = 0, l = 0, t = 0, re = 0, n0 = 0, l0 = 0;
        for (size_t i = 0; i < rays.size(); i++) {
            if (sl[i].face != cost[i].face || sl[i].t != cost[i].t) bad++;
            n += sl[i].nodes; l += sl[i].leaves; t += sl[i].tris; re += sl[i].merged; n0 += cost[i].nodes; l0 += cost[i].leaves;
        }
        const double m = (double)rays.size();
        // wave estimate as below: 16 consecutive rays of one (azimuth, pass)
        double ws = 0, ws0 = 0; size_t nw = 0;
        for (size_t k = 0; k < rays.size(); ) {
            size_t e = k; unsigned mx = 0, mx0 = 0;
            for (; e < std::min(rays.size(), k + 16) && rays[e].az == rays[k].az && rays[e].pass == rays[k].pass; e++) {
                mx = std::max(mx, sl[e].nodes + sl[e].leaves); mx0 = std::max(mx0, cost[e].nodes + cost[e].leaves);
            }
            ws += mx; ws0 += mx0; nw++; k = e;
        }
        printf("stackless: hits equal to the stack walk's: %s (%zu differ)\n", bad ? "NO" : "yes", bad);
        printf("stackless: node fetches/ray %.2f (%.2f of them re-fetches on the way up; stack walk %.2f)  leaves/ray %.2f (stack %.2f)  tris/ray %.2f\n",
               n / m, re / m, n0 / m, l / m, l0 / m, t / m);
        printf("stackless: steps/ray %.2f vs %.2f (x%.2f)  wave iterations (est.) %.2f vs %.2f (x%.2f)  bytes/ray %.1f\n",
               (n + l) / m, (n0 + l0) / m, (n + l) / (n0 + l0), ws / nw, ws0 / nw, ws / ws0, (n * 128 + t * 48) / m + 132);
    }

    if (getenv("TREEQ_TWO_LEVEL")) {
        // experiment: faces much larger than the median in their own tree, the rest in another; a ray walks the
        // large-face tree first and the other one with the hit distance as its range
        const size_t nf = faces.size() / 3;
        std::vector<float> area(nf);
        for (size_t f = 0; f < nf; f++) {
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
            for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) { const float x = verts[3 * (size_t)faces[3 * f + v] + k]; lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x); }
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            area[f] = dx * dy + dy * dz + dz * dx;
        }
        std::vector<float> srt(area); std::nth_element(srt.begin(), srt.begin() + nf / 2, srt.end());
        const float thr = srt[nf / 2] * (float)atof(getenv("TREEQ_TWO_LEVEL"));
        std::vector<uint32_t> fl, fs;
        for (size_t f = 0; f < nf; f++) { auto& dst = area[f] > thr ? fl : fs; for (int v = 0; v < 3; v++) dst.push_back(faces[3 * f + v]); }
        Bvh4 BL, BS;
        build_bvh4(verts.data(), verts.size() / 3, fl.data(), fl.size() / 3, nullptr, BL, err, threads);
        build_bvh4(verts.data(), verts.size() / 3, fs.data(), fs.size() / 3, nullptr, BS, err, threads);
        printf("two-level: %zu large faces (%zu nodes, %zu records), %zu small (%zu nodes, %zu records)\n", fl.size() / 3, BL.nodes.size(), BL.tris.size(), fs.size() / 3, BS.nodes.size(), BS.tris.size());
        double n = 0, l = 0, t = 0;
        std::vector<unsigned> st(rays.size());
#pragma omp parallel for schedule(dynamic, 256) reduction(+:n,l,t)
        for (long i = 0; i < (long)rays.size(); i++) {
            const Cost a = fl.empty() ? Cost() : trace(BL, rays[i].o, rays[i].d, 1000.0f);
            const float rm = a.t > 0 ? a.t * 1.0001f + 1e-3f : 1000.0f;
            const Cost b = trace(BS, rays[i].o, rays[i].d, rm);
            n += a.nodes + b.nodes; l += a.leaves + b.leaves; t += a.tris + b.tris; st[i] = a.nodes + a.leaves + b.nodes + b.leaves;
        }
        double ws = 0; size_t nw = 0;
        for (size_t k = 0; k + 16 <= st.size(); k += 16) { unsigned mx = 0; for (size_t e = k; e < k + 16; e++) mx = std::max(mx, st[e]); ws += mx; nw++; }
        const double m = (double)rays.size();
        printf("two-level: nodes/ray %.2f leaves/ray %.2f tris/ray %.2f steps/ray %.2f wave iterations (est.) %.2f bytes/ray %.1f\n", n / m, l / m, t / m, (n + l) / m, ws / nw, (n * 128 + t * 48) / m + 132);
    }

    size_t mism = 0;
    if (ref_face.size() == rays.size()) {
        for (size_t i = 0; i < rays.size(); i++) if (ref_face[i] != cost[i].face || ref_t[i] != cost[i].t) mism++;
        printf("hits vs hits.ref: %zu mismatches of %zu\n", mism, rays.size());
    } else {
        std::vector<uint32_t> raw(2 * rays.size());
        for (size_t i = 0; i < rays.size(); i++) { raw[i] = cost[i].face; memcpy(&raw[rays.size() + i], &cost[i].t, 4); }
        FILE* f = fopen((dir + "/hits.ref").c_str(), "wb"); fwrite(raw.data(), 4, raw.size(), f); fclose(f);
        printf("wrote hits.ref\n");
    }

    // per pass: means per ray; wave estimate: 16 consecutive rays of one (azimuth, pass) per wave
    std::map<int, std::vector<size_t>> by_pass;
    for (size_t i = 0; i < rays.size(); i++) by_pass[rays[i].pass].push_back(i);
    double tn = 0, tl = 0, tt = 0, tw = 0; size_t tr = 0, twv = 0;
    double merged_all = 0, wave_merged = 0;
    for (auto& kv : by_pass) {
        double n = 0, l = 0, t = 0, wave_steps = 0; size_t waves = 0;
        const auto& idx = kv.second;
        for (size_t k = 0; k < idx.size(); k += 16) {
            unsigned mx = 0; int az = rays[idx[k]].az;
            size_t e = k;
            unsigned mxm = 0;
            for (; e < std::min(idx.size(), k + 16) && rays[idx[e]].az == az; e++) { mx = std::max(mx, cost[idx[e]].nodes + cost[idx[e]].leaves); mxm = std::max(mxm, cost[idx[e]].nodes + cost[idx[e]].leaves - cost[idx[e]].merged); }
            wave_steps += mx; waves++; wave_merged += mxm;
            if (e < k + 16 && e < idx.size()) k = e - 16;   // azimuth boundary: next wave starts at e
        }
        for (size_t i : idx) { n += cost[i].nodes; l += cost[i].leaves; t += cost[i].tris; merged_all += cost[i].merged; }
        const double m = (double)idx.size();
        printf("pass %d: %8zu rays  nodes/ray %6.2f  leaves/ray %5.2f  tris/ray %6.2f  steps/ray %6.2f  wave iterations (est.) %6.2f  bytes/ray %7.1f\n",
               kv.first, idx.size(), n / m, l / m, t / m, (n + l) / m, wave_steps / waves, (n * 128 + t * 48) / m + 132);
        tn += n; tl += l; tt += t; tr += idx.size(); tw += wave_steps; twv += waves;
    }
    {
        double nh[5] = {0,0,0,0,0}, lah = 0, nah = 0; std::vector<unsigned> st;
        for (const Cost& c : cost) { for (int k = 0; k < 5; k++) nh[k] += c.nh[k]; lah += c.leaves_after_hit; nah += c.nodes_after_hit; st.push_back(c.nodes + c.leaves); }
        std::sort(st.begin(), st.end());
        printf("node visits by number of hit children 0..4 per ray: %.2f %.2f %.2f %.2f %.2f; after the first hit was found: %.2f nodes, %.2f leaves per ray\n",
               nh[0] / tr, nh[1] / tr, nh[2] / tr, nh[3] / tr, nh[4] / tr, nah / tr, lah / tr);
        printf("steps per ray percentiles: p10 %u p50 %u p90 %u p99 %u max %u\n", st[st.size() / 10], st[st.size() / 2], st[st.size() * 9 / 10], st[st.size() * 99 / 100], st.back());
    }
    if (getenv("TREEQ_PREDICT")) {
        // which ordering of the later-pass rays of one azimuth gives waves of equal cost?  keys are things the GPU knows
        // BEFORE it traces a ray: kind (reflection / refraction), material the ray travels in, cost of its parent
        std::map<std::pair<int, int>, std::vector<size_t>> grp;       // (az, pass) -> ray indices in logged order
        for (size_t i = 0; i < rays.size(); i++) grp[{ rays[i].az, rays[i].pass }].push_back(i);
        auto waves = [&](const char* name, auto keyfn) {
            double ws = 0; size_t nw = 0;
            for (auto& kv : grp) {
                if (kv.first.second == 0) continue;
                std::vector<size_t> v = kv.second;
                const auto& par = grp[{ kv.first.first, kv.first.second - 1 }];
                std::stable_sort(v.begin(), v.end(), [&](size_t a, size_t b) { return keyfn(a, par) < keyfn(b, par); });
                for (size_t k = 0; k < v.size(); k += 16) {
                    unsigned mx = 0;
                    for (size_t e = k; e < std::min(v.size(), k + 16); e++) mx = std::max(mx, cost[v[e]].nodes + cost[v[e]].leaves);
                    ws += mx; nw++;
                }
            }
            printf("  order by %-44s wave iterations %.2f\n", name, ws / nw);
        };
        auto pcost = [&](size_t i, const std::vector<size_t>& par) -> double { const size_t p = par[rays[i].parent >> 1]; return cost[p].nodes + cost[p].leaves; };
        waves("logged order (reference order)", [&](size_t, const std::vector<size_t>&) { return 0.0; });
        waves("kind (reflections first)", [&](size_t i, const std::vector<size_t>&) { return (double)(rays[i].parent & 1); });
        waves("material", [&](size_t i, const std::vector<size_t>&) { return (double)rays[i].mat; });
        waves("material, kind", [&](size_t i, const std::vector<size_t>&) { return (double)rays[i].mat * 2 + (rays[i].parent & 1); });
        waves("parent cost", [&](size_t i, const std::vector<size_t>& par) { return pcost(i, par); });
        waves("material, kind, parent cost", [&](size_t i, const std::vector<size_t>& par) { return ((double)rays[i].mat * 2 + (rays[i].parent & 1)) * 1000 + pcost(i, par); });
        waves("TRUE cost (bound)", [&](size_t i, const std::vector<size_t>&) { return (double)(cost[i].nodes + cost[i].leaves); });
    }
    if (getenv("TREEQ_SORTED")) {
        // upper bound of what ANY re-ordering of the rays inside an (azimuth, pass) group could win: waves of 16 rays of
        // equal cost (sorted by their true step count)
        double ws = 0; size_t nw = 0;
        std::map<std::pair<int, int>, std::vector<unsigned>> grp;
        for (size_t i = 0; i < rays.size(); i++) grp[{ rays[i].pass, rays[i].az }].push_back(cost[i].nodes + cost[i].leaves);
        for (auto& kv : grp) {
            auto& v = kv.second; std::sort(v.begin(), v.end());
            for (size_t k = 0; k < v.size(); k += 16) { ws += v[std::min(v.size(), k + 16) - 1]; nw++; }
        }
        printf("rays sorted by true cost inside each (azimuth, pass): wave iterations %.2f\n", ws / nw);
    }
    if (getenv("TREEQ_MERGE")) printf("leaf step + the node step popped after it in ONE iteration: %.2f merges per ray, steps/ray %.2f -> %.2f, wave iterations (est.) %.2f -> %.2f\n",
                                      merged_all / tr, (tn + tl) / tr, (tn + tl - merged_all) / tr, tw / twv, wave_merged / twv);
    printf("all   : %8zu rays  nodes/ray %6.2f  leaves/ray %5.2f  tris/ray %6.2f  steps/ray %6.2f  wave iterations (est.) %6.2f  bytes/ray %7.1f\n",
           tr, tn / tr, tl / tr, tt / tr, (tn + tl) / tr, tw / twv, (tn * 128 + tt * 48) / tr + 132);
    return mism ? 1 : 0;
}
