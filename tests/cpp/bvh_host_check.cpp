// Host-side BVH builder (radarays_ros_amd/csrc/rr_bvh.cpp) driven without a GPU, meant to be built
// with -fsanitize=address,undefined (tests/test_bvh_host.py): random soups, degenerate triangles,
// duplicates, one triangle, invalid input.  Checks the structural invariants the traversal kernel
// relies on: every triangle in exactly one leaf, leaves of 1..4 triangles, child boxes contain their
// triangles, stack bound <= 3 x depth (at most three siblings wait per level).
#include "rr_bvh.h"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using namespace rr;

static int check(const std::vector<float>& v, const std::vector<uint32_t>& f, int threads, const char* name)
{
    Bvh4 b; std::string err;
    if (!build_bvh4(v.data(), v.size() / 3, f.data(), f.size() / 3, nullptr, b, err, threads)) {
        std::printf("%s: build failed: %s\n", name, err.c_str()); return 1;
    }
    const size_t nf = f.size() / 3;
    if (b.tris.size() != nf) { std::printf("%s: %zu triangles in leaves, %zu given\n", name, b.tris.size(), nf); return 1; }
    std::vector<int> seen(nf, 0);
    size_t leaves = 0;
    for (const Node4& n : b.nodes)
        for (int k = 0; k < 4; k++) {
            const Child4& c = n.c[k];
            if (c.ref == kEmptyRef) continue;
            if (!(c.ref & kLeafFlag)) { if (c.ref >= b.nodes.size()) { std::printf("%s: child index out of range\n", name); return 1; } continue; }
            const uint32_t first = c.ref & 0x0FFFFFFFu, cnt = ((c.ref >> 28) & 7u) + 1u;
            if (cnt > kMaxLeafTris || first + cnt > nf) { std::printf("%s: bad leaf %u+%u\n", name, first, cnt); return 1; }
            leaves++;
            for (uint32_t t = first; t < first + cnt; t++) {
                seen[t]++;
                const TriRec& r = b.tris[t];
                const float p[3][3] = { { r.v0[0], r.v0[1], r.v0[2] },
                                        { r.v0[0] + r.e1[0], r.v0[1] + r.e1[1], r.v0[2] + r.e1[2] },
                                        { r.v0[0] + r.e2[0], r.v0[1] + r.e2[1], r.v0[2] + r.e2[2] } };
                for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) {
                    const float tol = 1e-4f * (1.0f + std::fabs(p[i][a]));
                    if (p[i][a] < c.lo[a] - tol || p[i][a] > c.hi[a] + tol) { std::printf("%s: vertex outside its leaf box\n", name); return 1; }
                }
            }
        }
    for (size_t t = 0; t < nf; t++) if (seen[t] != 1) { std::printf("%s: triangle slot %zu referenced %d times\n", name, t, seen[t]); return 1; }
    if (b.stack_need > 3 * b.depth) { std::printf("%s: stack bound %u > 3 x depth %u\n", name, b.stack_need, b.depth); return 1; }
    std::printf("%s: ok (%zu tris, %zu nodes, %zu leaves, depth %u)\n", name, nf, b.nodes.size(), leaves, b.depth);
    return 0;
}

int main()
{
    int bad = 0;
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
