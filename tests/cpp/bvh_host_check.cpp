// Host-side BVH builder (radarays_ros_amd/csrc/rr_bvh.cpp) driven without a GPU, meant to be built
// with -fsanitize=address,undefined (tests/test_bvh_host.py): random soups, mixed triangle scales (spatial
// splits), degenerate triangles, duplicates, one triangle, invalid input.  Checks the structural invariants the
// traversal kernel relies on: every face in at least one leaf, leaves of 1..4 triangle records, the reference
// budget, boxes of a child contain the boxes below it, every sampled point of a face lies in the box of a leaf
// that holds the face (a face cut by spatial splits is covered by its parts), stack bound <= 3 x depth.
#include "rr_bvh.h"

#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

using namespace rr;

static bool inside(const Child4& c, const float* p, float tol)
{
    for (int a = 0; a < 3; a++) if (p[a] < c.lo[a] - tol || p[a] > c.hi[a] + tol) return false;
    return true;
}

static int check(const std::vector<float>& v, const std::vector<uint32_t>& f, int threads, const char* name,
                 bool want_spatial = false, const BvhOptions* opt = nullptr)
{
    Bvh4 b; std::string err;
    if (!build_bvh4(v.data(), v.size() / 3, f.data(), f.size() / 3, nullptr, b, err, threads, opt)) {
        std::printf("%s: build failed: %s\n", name, err.c_str()); return 1;
    }
    const size_t nf = f.size() / 3, nt = b.tris.size();
    const float budget = opt ? opt->ref_budget : BvhOptions().ref_budget;
    if (nt < nf || nt > (size_t)((1.0 + budget) * nf) + 16) { std::printf("%s: %zu triangle records for %zu faces\n", name, nt, nf); return 1; }
    if (want_spatial && (b.spatial_splits == 0 || nt == nf)) { std::printf("%s: expected spatial splits\n", name); return 1; }
    if (opt && opt->sbvh_alpha < 0.f && (b.spatial_splits != 0 || nt != nf)) { std::printf("%s: spatial splits although switched off\n", name); return 1; }
    std::vector<int> slot_seen(nt, 0);
    std::vector<std::vector<Child4>> leaf_boxes(nf);      // per face: the boxes of the leaves that hold it
    size_t leaves = 0;
    for (const Node4& n : b.nodes)
        for (int k = 0; k < 4; k++) {
            const Child4& c = n.c[k];
            if (c.ref == kEmptyRef) continue;
            if (!(c.ref & kLeafFlag)) {
                if (c.ref >= b.nodes.size()) { std::printf("%s: child index out of range\n", name); return 1; }
                // hierarchy: the child's box contains every box stored in the child node
                const Node4& m = b.nodes[c.ref];
                for (int j = 0; j < 4; j++) {
                    if (m.c[j].ref == kEmptyRef) continue;
                    const float tol = 1e-4f * (1.0f + std::fabs(m.c[j].lo[0]) + std::fabs(m.c[j].hi[0]));
                    if (!inside(c, m.c[j].lo, tol) || !inside(c, m.c[j].hi, tol)) { std::printf("%s: box not contained in its parent\n", name); return 1; }
                }
                continue;
            }
            const uint32_t first = c.ref & 0x0FFFFFFFu, cnt = ((c.ref >> 28) & 7u) + 1u;
            if (cnt > kMaxLeafTris || first + cnt > nt) { std::printf("%s: bad leaf %u+%u\n", name, first, cnt); return 1; }
            leaves++;
            for (uint32_t t = first; t < first + cnt; t++) {
                slot_seen[t]++;
                const TriRec& r = b.tris[t];
                if (r.face >= nf) { std::printf("%s: face id out of range\n", name); return 1; }
                const float* a = v.data() + 3 * (size_t)f[3 * (size_t)r.face];
                if (r.v0[0] != a[0] || r.v0[1] != a[1] || r.v0[2] != a[2]) { std::printf("%s: triangle record does not match its face\n", name); return 1; }
                for (uint32_t u = first; u < t; u++) if (b.tris[u].face == r.face) { std::printf("%s: face twice in one leaf\n", name); return 1; }
                leaf_boxes[r.face].push_back(c);
            }
        }
    for (size_t t = 0; t < nt; t++) if (slot_seen[t] != 1) { std::printf("%s: triangle slot %zu referenced %d times\n", name, t, slot_seen[t]); return 1; }
    // coverage: vertices, centroid and random interior points of every face
    std::mt19937 g(11);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (size_t face = 0; face < nf; face++) {
        if (leaf_boxes[face].empty()) { std::printf("%s: face %zu in no leaf\n", name, face); return 1; }
        const float* p0 = v.data() + 3 * (size_t)f[3 * face], *p1 = v.data() + 3 * (size_t)f[3 * face + 1], *p2 = v.data() + 3 * (size_t)f[3 * face + 2];
        for (int s = 0; s < 12; s++) {
            float u = U(g), w = U(g);
            if (u + w > 1.f) { u = 1.f - u; w = 1.f - w; }
            if (s == 0) { u = 0; w = 0; } else if (s == 1) { u = 1; w = 0; } else if (s == 2) { u = 0; w = 1; } else if (s == 3) { u = w = 1.f / 3; }
            float p[3];
            for (int a = 0; a < 3; a++) p[a] = p0[a] + u * (p1[a] - p0[a]) + w * (p2[a] - p0[a]);
            bool ok = false;
            const float tol = 1e-4f * (1.0f + std::fabs(p[0]) + std::fabs(p[1]) + std::fabs(p[2]));
            for (const Child4& c : leaf_boxes[face]) if (inside(c, p, tol)) { ok = true; break; }
            if (!ok) { std::printf("%s: a point of face %zu is covered by none of its %zu leaves\n", name, face, leaf_boxes[face].size()); return 1; }
        }
    }
    if (b.stack_need > 3 * b.depth) { std::printf("%s: stack bound %u > 3 x depth %u\n", name, b.stack_need, b.depth); return 1; }
    std::printf("%s: ok (%zu faces, %zu records, %zu nodes, %zu leaves, depth %u, %llu spatial splits)\n", name, nf, nt, b.nodes.size(), leaves, b.depth,
                (unsigned long long)b.spatial_splits);
    return 0;
}

int main(int argc, char** argv)
{
    int bad = 0;
    if (argc > 1 && std::string(argv[1]) == "big") {
        // large enough for everything that runs in parallel: subtree tasks (>= 32k references), chunk-parallel binning and
        // partition (>= 64k), the block pool and the per-thread slot blocks -- the case the ThreadSanitizer test runs.
        // Two builds: the per-thread state of the calling thread must start afresh
        std::vector<float> v; std::vector<uint32_t> f;
        auto tri = [&](float x, float y, float z, float s, float tilt) {
            const uint32_t b0 = (uint32_t)(v.size() / 3);
            const float p[9] = { x, y, z, x + s, y, z + tilt * s, x, y + s, z + 0.5f * tilt * s };
            v.insert(v.end(), p, p + 9);
            f.push_back(b0); f.push_back(b0 + 1); f.push_back(b0 + 2);
        };
        std::mt19937 g(11);
        std::uniform_real_distribution<float> U(0.f, 1.f);
        for (int i = 0; i < 400; i++) for (int j = 0; j < 400; j++) tri(0.25f * i, 0.25f * j, 0.02f * ((i * 7 + j * 13) % 11), 0.25f, 0.1f);
        for (int k = 0; k < 2000; k++) tri(100.f * U(g), 100.f * U(g), 0.01f * k, 4.f + 10.f * U(g), (k % 5) * 0.3f);
        bad += check(v, f, 8, "big mixed-scales", true);
        bad += check(v, f, 8, "big mixed-scales again", true);
        return bad ? 1 : 0;
    }
    std::mt19937 g(7);
    std::uniform_real_distribution<float> U(-20.f, 20.f), S(-0.5f, 0.5f);
    for (int n : { 1, 2, 3, 4, 5, 17, 1000, 20000 }) {
        std::vector<float> v; std::vector<uint32_t> f;
        for (int t = 0; t < n; t++) {
            const float c[3] = { U(g), U(g), U(g) };
            for (int k = 0; k < 3; k++) { v.push_back(c[0] + S(g)); v.push_back(c[1] + S(g)); v.push_back(c[2] + S(g)); }
            f.push_back(3 * t); f.push_back(3 * t + 1); f.push_back(3 * t + 2);
        }
        char nm[64]; std::snprintf(nm, sizeof nm, "soup%d", n);
        bad += check(v, f, n > 1000 ? 4 : 1, nm);
    }
    {   // mixed scales: a field of small triangles under large overlapping ones (what BASELINE configs 3-5 look like):
        // spatial splits must occur, stay inside the reference budget, and keep every face covered
        std::vector<float> v; std::vector<uint32_t> f;
        auto tri = [&](float x, float y, float z, float s, float tilt) {
            const uint32_t b0 = (uint32_t)(v.size() / 3);
            const float p[9] = { x, y, z, x + s, y, z + tilt * s, x, y + s, z + 0.5f * tilt * s };
            v.insert(v.end(), p, p + 9);
            f.push_back(b0); f.push_back(b0 + 1); f.push_back(b0 + 2);
        };
        for (int i = 0; i < 120; i++) for (int j = 0; j < 120; j++) tri(0.5f * i, 0.5f * j, 0.02f * ((i * 7 + j * 13) % 11), 0.5f, 0.1f);
        for (int k = 0; k < 150; k++) tri(U(g) + 20.f, U(g) + 20.f, 0.1f * k, 18.f + 0.1f * k, (k % 5) * 0.3f);
        bad += check(v, f, 4, "mixed-scales", true);
        BvhOptions off; off.sbvh_alpha = -1.f;
        bad += check(v, f, 2, "mixed-scales, splits off", false, &off);
        BvhOptions tight; tight.ref_budget = 0.001f;
        bad += check(v, f, 2, "mixed-scales, tiny budget", false, &tight);
    }
    {   // degenerate: zero-area triangles, all identical, all on one point
        std::vector<float> v = { 0, 0, 0, 1, 0, 0, 2, 0, 0, 5, 5, 5 };
        std::vector<uint32_t> f;
        for (int k = 0; k < 300; k++) { f.push_back(0); f.push_back(1); f.push_back(2); }
        for (int k = 0; k < 300; k++) { f.push_back(3); f.push_back(3); f.push_back(3); }
        bad += check(v, f, 2, "degenerate");
    }
    {   // invalid input must be refused, not crash
        Bvh4 b; std::string err;
        std::vector<float> v = { 0, 0, 0, 1, 0, 0, 0, 1, 0 };
        std::vector<uint32_t> f = { 0, 1, 7 };
        if (build_bvh4(v.data(), 3, f.data(), 1, nullptr, b, err, 1)) { std::printf("index out of range accepted\n"); bad++; }
        v[4] = NAN; f[2] = 2;
        if (build_bvh4(v.data(), 3, f.data(), 1, nullptr, b, err, 1)) { std::printf("NaN vertex accepted\n"); bad++; }
        if (build_bvh4(v.data(), 3, f.data(), 0, nullptr, b, err, 1)) { std::printf("empty mesh accepted\n"); bad++; }
    }
    return bad ? 1 : 0;
}
