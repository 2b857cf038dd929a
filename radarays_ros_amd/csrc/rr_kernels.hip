// rr_kernels.hip -- gfx950 kernels of the radar hot path.
//
// One frame = for each ray-cast pass p (RadarCPU.cpp:220):
//     k_trace   nearest hit of every live wave against the BVH4   (RadarCPU.cpp:236)
//     k_shade   move / material select / Fresnel split / BRDF     (RadarCPU.cpp:243-371)
//     k_scan    ordered (stable) compaction of children + signals (RadarCPU.cpp:290,322,369,380)
// then
//     k_column  signals -> range-bin column, noise, scale, u8     (RadarCPU.cpp:402-542)
// and (rr_assemble) a tiled transpose into the mono8 image.
//
// Ordering: the reference appends children and signals in wave order; float
// accumulation into the slice depends on that order.  All compaction here is
// prefix-sum based (no atomics), so the signal list of an azimuth is in exactly
// the reference's order and the image does not depend on scheduling.
#include "rr_device.h"
#include <hip/hip_ext.h>

#include <algorithm>

namespace rr {

// ---------------------------------------------------------------------------
// common
// ---------------------------------------------------------------------------
__device__ inline Quat ld_quat(const float4* p) { const float4 v = *p; return { v.x, v.y, v.z, v.w }; }

// Tam = Tsm * Tas (RadarCPU.cpp:201-206); Tas.t = 0.
// ARGS: the pass-0 trace launch reads the call's poses from its own by-value argument; everything behind it from the lane's
// pose table, which that launch wrote (rr_device.h: Params::pose_table)
template <bool ARGS>
__device__ __forceinline__ void azimuth_frame(const Params& P, const PoseArgs* pa, int seg, Quat& q_am, V3& t_am)
{
    const int frame = seg / P.n_loc, az = P.az_begin + seg % P.n_loc;
    const Quat q_as = ld_quat(P.q_as + az);
    const int row = P.set_mode ? 0 : frame;      // a parameter batch: every set the same pose
    Quat q_sm; V3 t_sm;
    if constexpr (ARGS) {
        const float* ps = pa->p[row];
        q_sm = { ps[0], ps[1], ps[2], ps[3] }; t_sm = { ps[4], ps[5], ps[6] };
    } else {
        const float4 a = P.pose_table[2 * row], b = P.pose_table[2 * row + 1];
        q_sm = { a.x, a.y, a.z, a.w }; t_sm = { b.x, b.y, b.z };
    }
    if (P.motion_poses) {     // include_motion: Tsm looked up per azimuth (RadarCPU.cpp:190-196)
        const float* ps = P.motion_poses + 7 * ((size_t)(frame % P.motion_rows) * P.n_angles + az);
        q_sm = { ps[0], ps[1], ps[2], ps[3] }; t_sm = { ps[4], ps[5], ps[6] };
    }
    q_am = q_mul(q_sm, q_as);
    // Tas.t = 0: the reference forms R_sm * 0 + t_sm (rmagine T1 * T2).  Rotating the zero vector gives (+-0, +-0, +-0) for
    // any finite quaternion and x + (+-0) = x, so t_am IS t_sm (only the sign of a zero component of t_sm could differ, which
    // no ray can see): 59 instructions per wave less in k_trace than the two quaternion products of the literal form
    t_am = t_sm;
}

struct Hit { float t; uint32_t tri; uint32_t face; };

// ---------------------------------------------------------------------------
// BVH4 traversal: ONE RAY PER QUAD of lanes (16 rays per wave64).
//   inner node : lane q slab-tests child q; the four keys are exchanged with DPP
//                quad_perm broadcasts (no LDS), every lane sorts them identically
//   leaf       : lane q intersects triangle q (<= 4 per leaf); nearest by a DPP
//                xor-butterfly on (t, face)
// Control state (cur, sp, best) is replicated in the four lanes, so a quad never
// diverges; the stack lives in LDS (one entry per ray, written by lane 0 of the
// quad) and spills to global memory beyond `stack_lds` entries.
// ---------------------------------------------------------------------------
constexpr int kRaysPerWave = 16;
constexpr int kTraceThreads = 64;                        // ONE wave = 16 rays per workgroup: a wave gives its LDS and its slot back the moment it is done, no barrier waits for a sibling (128 threads: 461 vs 447 us per later-pass launch alone, 4,170-4,190 vs 4,270-4,290 images/s on the target; 256 / 512 lose more)
constexpr int kRaysPerBlock = kTraceThreads / 4;
#ifndef RR_CULL_POP
#define RR_CULL_POP 1
#endif
constexpr bool kCullPop = RR_CULL_POP != 0;   // later passes: stack entries carry a 16-bit lower bound of their entry distance (6 B per entry)

#define RR_DPP_I(x, ctrl) __builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xF, 0xF, true)
#define RR_DPP_F(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), 0xF, 0xF, true))
// quad_perm encodings: broadcast lane k -> k*0x55; xor 1 -> 0xB1; xor 2 -> 0x4E
#define RR_QBCAST0 0x00
#define RR_QBCAST1 0x55
#define RR_QBCAST2 0xAA
#define RR_QBCAST3 0xFF
// rotations: lane q reads lane (q + k) & 3
#define RR_QROT1 0x39
#define RR_QROT2 0x4E
#define RR_QROT3 0x93
#define RR_QXOR1 0xB1
#define RR_QXOR2 0x4E

// A ray prepared for the slab test: 1/d with tiny components clamped so it stays finite, and
// oo = -o/d, so that (plane - o)/d is one fma.
struct RaySetup { V3 o, d; float idx, idy, idz, oox, ooy, ooz; };
__device__ inline RaySetup ray_setup(V3 o, V3 d)
{
    RaySetup R; R.o = o; R.d = d;
    const float eps = 1e-20f;
    const float dx = fabsf(d.x) < eps ? copysignf(eps, d.x) : d.x;
    const float dy = fabsf(d.y) < eps ? copysignf(eps, d.y) : d.y;
    const float dz = fabsf(d.z) < eps ? copysignf(eps, d.z) : d.z;
    // v_rcp_f32 (1 ulp) instead of three IEEE divisions (~30 instructions): 1 / d only feeds the slab test, which merely
    // has to be conservative -- an ulp of 1 / d moves a box plane by 6e-8 x t, the boxes are padded by 2e-5 x the scene's
    // extent; the triangle test does not use it, so hits stay bit-exact (tests/fuzz/fuzz_trace.py)
    R.idx = __builtin_amdgcn_rcpf(dx); R.idy = __builtin_amdgcn_rcpf(dy); R.idz = __builtin_amdgcn_rcpf(dz);
    R.oox = -o.x * R.idx; R.ooy = -o.y * R.idy; R.ooz = -o.z * R.idz;
    return R;
}

template <bool STATS, bool SPILL, bool CULL>
__device__ inline Hit traverse(const float4* __restrict__ base4, const uint32_t tri_base4,
                               const RaySetup& R, float range_max, float hit_pad,
                               uint32_t* lds_stack, int stack_lds, uint32_t* spill, int spill_stride, int gray,
                               unsigned& n_nodes, unsigned& n_tris, unsigned* wstat = nullptr)
{
    const int q = threadIdx.x & 3;
    const int wave = threadIdx.x >> 6;
    const int rw = (threadIdx.x & 63) >> 2;                         // ray within the wave
    uint32_t* my = lds_stack + (size_t)wave * stack_lds * kRaysPerWave + rw;   // entry e at my[e*16]
    // beside every reference: the upper 16 bits of the entry distance of its box (truncated, i.e. a lower bound).  An
    // entry whose bound lies beyond the cull distance by the time it is popped is dropped without fetching the node
    uint16_t* myk = reinterpret_cast<uint16_t*>(lds_stack + (size_t)(kTraceThreads / 64) * stack_lds * kRaysPerWave) +
                    (size_t)wave * stack_lds * kRaysPerWave + rw;
    const V3 o = R.o, d = R.d;
    const float idx = R.idx, idy = R.idy, idz = R.idz, oox = R.oox, ooy = R.ooy, ooz = R.ooz;

    unsigned long long bestkey = 0x7F80000000000000ull;   // (+inf : 0): only a finite t can beat it
    uint32_t best_first = 0;                              // first triangle of the leaf that holds the best hit
    float tcull = range_max * 1.0001f + 1e-3f;
    int sp = 0;
    uint32_t cur = 0;   // root

    // child references are float4 offsets from base4 (nodes: 8 float4, 2 per child; triangles: 3 float4):
    // lane q adds 2q (its child record) or 3q (its triangle)
    const uint32_t q2 = 2u * (uint32_t)q, q3 = 3u * (uint32_t)q;
    const int qsh = (threadIdx.x & 63) & ~3;                        // bit position of this quad in a ballot
    uint32_t misskey = 0x7F800000u | (uint32_t)q;                   // key of a missed child
    asm volatile("" : "+v"(misskey));                              // opaque: keeps (tmin & ~3) | q one v_and_or_b32
    while (true) {
        // ---- ONE batch of loads per step, whatever the step is (single s_waitcnt) ----
        const bool leaf = (cur & kLeafFlag) != 0;
        if (STATS) n_tris += 0x10000u;   // high half: loop iterations of this lane = its traversal steps, node steps that descend included (advisor, round 4: counted at the pop only, a tree with more interior descents was scored as cheaper)
        if (STATS && wstat) {   // wave-level shape of this iteration (same value in every live lane)
            wstat[0]++; wstat[1] += __ballot(!leaf) != 0ull; wstat[2] += __ballot(leaf) != 0ull;
            wstat[3] += (unsigned)__builtin_popcountll(__ballot(true)) >> 2;
        }
        const uint32_t first = cur & 0x0FFFFFFFu;                   // float4 offset of the node / of the leaf's first triangle
        const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
        const float4* p = base4 + (first + (leaf ? q3 : q2));
        const float4 A = p[0], B = p[1];
        float4 C = make_float4(0.f, 0.f, 0.f, 0.f);
        if (leaf) C = p[2];          // triangle arrays are padded: q >= cnt reads stay in bounds
        if (!leaf) {
            // child record: A = (lo.x lo.y lo.z hi.x)  B = (hi.y hi.z ref pad)
            if (STATS) { n_nodes += (q == 0); }
            const float ax = __builtin_fmaf(A.x, idx, oox), bx = __builtin_fmaf(A.w, idx, oox);
            const float ay = __builtin_fmaf(A.y, idy, ooy), by = __builtin_fmaf(B.x, idy, ooy);
            const float az = __builtin_fmaf(A.z, idz, ooz), bz = __builtin_fmaf(B.y, idz, ooz);
            const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            // (no safety factor on tmax: every box is padded by 2e-5 x the scene extent on each side, orders of magnitude
            // more than the rounding of the six FMAs, so a flat box still has tmin < tmax for any ray that touches it)
            const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            // tmin <= min(tmax, tcull) as ONE compare (its mask is the ballot below).  tcull >= 0, so the
            // signed-integer minimum of the two bit patterns is the float minimum whenever tmax >= 0 and
            // a negative value (-> no hit, tmin >= 0) whenever tmax < 0
            const float lim = __int_as_float(min(__float_as_int(tmax), __float_as_int(tcull)));
            const bool h = tmin <= lim;
            // distinct keys (lane id in the low bits); a miss sorts last
            const uint32_t mykey = h ? ((__float_as_uint(tmin) & ~3u) | (uint32_t)q) : misskey;
            const uint32_t myref = __float_as_uint(B.z);
            // rank of my child among the four (0 = nearest): the three other keys arrive by quad
            // rotations; keys are < 2^31, so the sign of (other - mine) says "other is nearer"
            const uint32_t d1 = (uint32_t)RR_DPP_I(mykey, RR_QROT1) - mykey;
            const uint32_t d2 = (uint32_t)RR_DPP_I(mykey, RR_QROT2) - mykey;
            const uint32_t d3 = (uint32_t)RR_DPP_I(mykey, RR_QROT3) - mykey;
            const int rank = (int)((d1 >> 31) + (d2 >> 31) + (d3 >> 31));
            // number of hit children of the quad
            const int nhit = __builtin_popcount((unsigned)(__builtin_amdgcn_ballot_w64(h) >> qsh) & 0xFu);
            // nearest child -> cur (OR-reduce over the quad); the others go on the stack far-first,
            // each lane storing its OWN reference: no sorted copies of the refs are needed
            uint32_t nxt = (h && rank == 0) ? myref : 0u;
            nxt |= (uint32_t)RR_DPP_I(nxt, RR_QXOR1);
            nxt |= (uint32_t)RR_DPP_I(nxt, RR_QXOR2);
            if (h && rank > 0) {
                const int e = sp + (nhit - 1 - rank);
                if (SPILL && e >= stack_lds) spill[(size_t)(e - stack_lds) * spill_stride + gray] = myref;
                else { my[e * kRaysPerWave] = myref; if (CULL) myk[e * kRaysPerWave] = (uint16_t)(__float_as_uint(tmin) >> 16); }
            }
            if (nhit > 0) { sp += nhit - 1; cur = nxt; continue; }
        } else {
            // Moeller-Trumbore, f32, un-fused: bit-identical to the CPU restatement.  Computed by every
            // lane (padded triangle arrays keep q >= cnt in bounds), selected at the end: no branches
            if (STATS) n_tris += ((uint32_t)q < cnt);
            const V3 v0 = { A.x, A.y, A.z }, e1 = { B.x, B.y, B.z }, e2 = { C.x, C.y, C.z };
            const V3 pvec = v_cross(d, e2);
            const float det = v_dot(e1, pvec);
            const float inv = 1.0f / det;
            const V3 tvec = v_sub(o, v0);
            const float u = v_dot(tvec, pvec) * inv;
            const V3 qvec = v_cross(tvec, e1);
            const float v = v_dot(d, qvec) * inv;
            const float tt = v_dot(e2, qvec) * inv;
            bool ok = ((uint32_t)q < cnt) && (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) &&
                      (tt > 0.0f && tt <= range_max);
            // Grazing guard (round 5): a ray within 0.3 degrees of the triangle's plane -- det^2 < 2.5e-5 |e1 x e2|^2, the
            // threshold rides in the record's spare word (k_tri_graze) -- makes Moeller-Trumbore ill-conditioned: its
            // barycentric test can accept a point centimetres outside the triangle, outside the padded box any hierarchy
            // holds the triangle in (seed 307 of round 4's nearest-hit fuzz).  Such a hit counts only if its point lies in
            // the triangle's box padded by hit_pad (= half the builders' padding), which makes the nearest hit a property
            // of the mesh alone, not of the tree; the oracle applies the same rule with the same un-fused arithmetic.
            // Wave-uniform branch, taken by ~1 leaf step in 50: 4 instructions on the path everybody runs
            const bool graze = ok && (det * det < C.w);
            if (__builtin_amdgcn_ballot_w64(graze) != 0ull) {
                const V3 ph = v_add(o, v_scale(d, tt));
                const V3 v1 = v_add(v0, e1), v2 = v_add(v0, e2);
                const bool inside =
                    ph.x >= fminf(v0.x, fminf(v1.x, v2.x)) - hit_pad && ph.x <= fmaxf(v0.x, fmaxf(v1.x, v2.x)) + hit_pad &&
                    ph.y >= fminf(v0.y, fminf(v1.y, v2.y)) - hit_pad && ph.y <= fmaxf(v0.y, fmaxf(v1.y, v2.y)) + hit_pad &&
                    ph.z >= fminf(v0.z, fminf(v1.z, v2.z)) - hit_pad && ph.z <= fmaxf(v0.z, fmaxf(v1.z, v2.z)) + hit_pad;
                if (graze && !inside) ok = false;
            }
            // quad-wide nearest (t, then lower face index) in two 32-bit rounds: t is positive or +inf, so its
            // order is the unsigned order of its bits -> minimum over the quad with two DPP mins; then, among
            // the lanes that hold that minimum, the lowest (face << 2 | lane): the low bits name the winning
            // lane, i.e. the triangle's slot in the leaf (face < 2^28).  A leaf without a hit yields
            // (+inf : something), which never beats the initial (+inf : 0).
            const uint32_t tb = ok ? __float_as_uint(tt) : 0x7F800000u;
            uint32_t tm = min(tb, (uint32_t)RR_DPP_I(tb, RR_QXOR1));
            tm = min(tm, (uint32_t)RR_DPP_I(tm, RR_QXOR2));
            const uint32_t fk = (tb == tm) ? ((__float_as_uint(A.w) << 2) | (uint32_t)q) : 0xFFFFFFFFu;
            uint32_t fm = min(fk, (uint32_t)RR_DPP_I(fk, RR_QXOR1));
            fm = min(fm, (uint32_t)RR_DPP_I(fm, RR_QXOR2));
            const unsigned long long key = ((unsigned long long)tm << 32) | fm;
            if (key < bestkey) {
                bestkey = key; best_first = first;
                tcull = __builtin_fmaf(__uint_as_float(tm), 1.0001f, 1e-3f);   // a bound only: may be fused
            }
        }
        // pop
        if (!CULL) {
            if (sp == 0) break;
            sp--;
            cur = (SPILL && sp >= stack_lds) ? spill[(size_t)(sp - stack_lds) * spill_stride + gray] : my[sp * kRaysPerWave];
        } else {
            // truncation is monotone: key(tmin) > key(tcull) implies tmin > tcull, the very test the node's own visit
            // would fail for every child (their entry distances are not below the parent's)
            const uint32_t ck = __float_as_uint(tcull) >> 16;
            bool more = false;
            while (sp > 0) {
                sp--;
                uint32_t k = 0;
                if (SPILL && sp >= stack_lds) cur = spill[(size_t)(sp - stack_lds) * spill_stride + gray];
                else { cur = my[sp * kRaysPerWave]; k = myk[sp * kRaysPerWave]; }
                asm volatile("" : "+v"(cur), "+v"(k));      // both reads in ONE LDS round trip (the reference is not fetched after the test)
                if (k <= ck) { more = true; break; }
            }
            if (!more) break;
        }
    }
    Hit best;
    best.t = __uint_as_float((uint32_t)(bestkey >> 32));
    const bool hit = (uint32_t)(bestkey >> 32) < 0x7F800000u;
    best.face = hit ? ((uint32_t)bestkey >> 2) : 0xFFFFFFFFu;
    best.tri = hit ? (best_first - tri_base4) / 3u + ((uint32_t)bestkey & 3u) : 0xFFFFFFFFu;   // leaf-order triangle index
    return best;
}

// ---------------------------------------------------------------------------
// The STACK-FREE walk of the same tree (north_star: "stackless"; RR_STACKLESS=1, round 6 -- measured and not the default, see
// DESIGN.md §3).  No per-ray memory at all, no LDS: the state is (the node whose four child records the quad's registers hold,
// the key of the child the walk last went into).  Children are visited in increasing KEY order -- key = (bits(tmin) & ~3) |
// slot, the key the stack walk ranks by, so the order is the same nearest-first order -- "the next child" being the hit child
// with the smallest key above the previous one.  Going up follows the parent link in the node's spare word (child record 0's
// pad: parent offset | slot, written by k_parent_links), and the parent is FETCHED AND TESTED AGAIN: its keys are a pure
// function of ray and boxes, so the key of the child the walk came back from is recomputed, not remembered -- nothing is kept
// per level.  Leaves are handled while their node's keys are still in registers.  Culling is exact and late (a child is
// tested against the cull distance of the moment it is picked).  Price: an internal node is fetched once on the way down and
// once for every child NODE the walk returns from -- 1.74x the steps of the stack walk on config 3 (tools/treeq.cpp
// TREEQ_STACKLESS, same hits).  Nearest hit = min over (t, face id) as everywhere: bit-identical results.
// ---------------------------------------------------------------------------
__device__ inline Hit traverse_stackless(const float4* __restrict__ base4, const uint32_t tri_base4,
                                         const RaySetup& R, float range_max, float hit_pad)
{
    const int q = threadIdx.x & 3;
    const V3 o = R.o, d = R.d;
    const float idx = R.idx, idy = R.idy, idz = R.idz, oox = R.oox, ooy = R.ooy, ooz = R.ooz;
    unsigned long long bestkey = 0x7F80000000000000ull;
    uint32_t best_first = 0;
    float tcull = range_max * 1.0001f + 1e-3f;
    const uint32_t q2 = 2u * (uint32_t)q, q3 = 3u * (uint32_t)q;
    uint32_t cur = 0;                    // what this iteration fetches: a node (float4 offset) or a leaf reference
    int from = -1;                       // >= 0: the node is one the walk RETURNS to, from the child in this slot
    // the node the walk is in, as this lane sees it: entry / exit distance, key and reference of child q; the node's parent link
    float tmin = 0.0f, tmax = -1.0f;
    uint32_t mykey = 0x7FFFFFFFu, myref = 0u, plink = 0xFFFFFFFFu, leafkey = 0u;
    while (true) {
        const bool leaf = (cur & kLeafFlag) != 0;
        const uint32_t first = cur & 0x0FFFFFFFu;
        const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
        const float4* p = base4 + (first + (leaf ? q3 : q2));
        const float4 A = p[0], B = p[1];
        float4 C = make_float4(0.f, 0.f, 0.f, 0.f);
        if (leaf) C = p[2];
        int prev;
        if (!leaf) {
            const float ax = __builtin_fmaf(A.x, idx, oox), bx = __builtin_fmaf(A.w, idx, oox);
            const float ay = __builtin_fmaf(A.y, idy, ooy), by = __builtin_fmaf(B.x, idy, ooy);
            const float az = __builtin_fmaf(A.z, idz, ooz), bz = __builtin_fmaf(B.y, idz, ooz);
            tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            mykey = (__float_as_uint(tmin) & ~3u) | (uint32_t)q;
            myref = __float_as_uint(B.z);
            plink = (uint32_t)RR_DPP_I(__float_as_uint(B.w), RR_QBCAST0);
            // returning: everything up to and including the child the walk came from is done (its key, recomputed)
            uint32_t pk = (q == from) ? mykey : 0u;
            pk |= (uint32_t)RR_DPP_I(pk, RR_QXOR1);
            pk |= (uint32_t)RR_DPP_I(pk, RR_QXOR2);
            prev = from >= 0 ? (int)pk : -1;
        } else {
            // (the leaf step of traverse(): Moeller-Trumbore, grazing guard, quad-wide nearest)
            const V3 v0 = { A.x, A.y, A.z }, e1 = { B.x, B.y, B.z }, e2 = { C.x, C.y, C.z };
            const V3 pvec = v_cross(d, e2);
            const float det = v_dot(e1, pvec);
            const float inv = 1.0f / det;
            const V3 tvec = v_sub(o, v0);
            const float u = v_dot(tvec, pvec) * inv;
            const V3 qvec = v_cross(tvec, e1);
            const float v = v_dot(d, qvec) * inv;
            const float tt = v_dot(e2, qvec) * inv;
            bool ok = ((uint32_t)q < cnt) && (det != 0.0f) && (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && u + v <= 1.0f) &&
                      (tt > 0.0f && tt <= range_max);
            const bool graze = ok && (det * det < C.w);
            if (__builtin_amdgcn_ballot_w64(graze) != 0ull) {
                const V3 ph = v_add(o, v_scale(d, tt));
                const V3 v1 = v_add(v0, e1), v2 = v_add(v0, e2);
                const bool inside =
                    ph.x >= fminf(v0.x, fminf(v1.x, v2.x)) - hit_pad && ph.x <= fmaxf(v0.x, fmaxf(v1.x, v2.x)) + hit_pad &&
                    ph.y >= fminf(v0.y, fminf(v1.y, v2.y)) - hit_pad && ph.y <= fmaxf(v0.y, fmaxf(v1.y, v2.y)) + hit_pad &&
                    ph.z >= fminf(v0.z, fminf(v1.z, v2.z)) - hit_pad && ph.z <= fmaxf(v0.z, fmaxf(v1.z, v2.z)) + hit_pad;
                if (graze && !inside) ok = false;
            }
            const uint32_t tb = ok ? __float_as_uint(tt) : 0x7F800000u;
            uint32_t tm = min(tb, (uint32_t)RR_DPP_I(tb, RR_QXOR1));
            tm = min(tm, (uint32_t)RR_DPP_I(tm, RR_QXOR2));
            const uint32_t fk = (tb == tm) ? ((__float_as_uint(A.w) << 2) | (uint32_t)q) : 0xFFFFFFFFu;
            uint32_t fm = min(fk, (uint32_t)RR_DPP_I(fk, RR_QXOR1));
            fm = min(fm, (uint32_t)RR_DPP_I(fm, RR_QXOR2));
            const unsigned long long key = ((unsigned long long)tm << 32) | fm;
            if (key < bestkey) {
                bestkey = key; best_first = first;
                tcull = __builtin_fmaf(__uint_as_float(tm), 1.0001f, 1e-3f);
            }
            prev = (int)leafkey;
        }
        // the next child of the node in registers: the hit child (against the cull distance of NOW) with the smallest key above prev
        const float lim = __int_as_float(min(__float_as_int(tmax), __float_as_int(tcull)));
        const bool h = (tmin <= lim) && ((int)mykey > prev);
        const uint32_t cand = h ? mykey : 0x7FFFFFFFu;
        uint32_t m = min(cand, (uint32_t)RR_DPP_I(cand, RR_QXOR1));
        m = min(m, (uint32_t)RR_DPP_I(m, RR_QXOR2));
        if (m != 0x7FFFFFFFu) {
            uint32_t nxt = (cand == m) ? myref : 0u;
            nxt |= (uint32_t)RR_DPP_I(nxt, RR_QXOR1);
            nxt |= (uint32_t)RR_DPP_I(nxt, RR_QXOR2);
            cur = nxt; leafkey = m; from = -1;          // (a leaf: the node's registers stay; a node: they are replaced)
        } else {
            if (plink == 0xFFFFFFFFu) break;            // the root has no more children: done
            cur = plink & ~7u; from = (int)(plink & 7u);
        }
    }
    Hit best;
    best.t = __uint_as_float((uint32_t)(bestkey >> 32));
    const bool hit = (uint32_t)(bestkey >> 32) < 0x7F800000u;
    best.face = hit ? ((uint32_t)bestkey >> 2) : 0xFFFFFFFFu;
    best.tri = hit ? (best_first - tri_base4) / 3u + ((uint32_t)bestkey & 3u) : 0xFFFFFFFFu;
    return best;
}

// one 16-ray group (= one wave) of a trace launch: group gx of segment row seg_y (pass 0: tile gx of the flat sequence)
template <bool FIRST, bool STATS, bool SPILL, bool CULL, bool SL = false>
__device__ __forceinline__ void trace_group(const Params& P, const int pass, const int seg_y, const int gx, const int count,
                                            uint32_t* lds_stack, const int gdim_x, const PoseArgs* poses = nullptr)
{
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int cur = pass & 1;
    RaySetup R; int j, seg;
    {
        const int rr = r;          // every lane of the quad prepares its ray itself: the same instructions as one lane per ray and an LDS hand-off (round 2), without the hand-off and its two barriers
        int k = gx * kRaysPerBlock + rr;           // trace slot
        seg = seg_y;
        bool live = k < count;
        if (FIRST) {
            // wave w = tile (sample block sb, azimuth block ab) of (16 / A) samples x A neighbouring segments
            const int ns = P.set_mode ? P.n_groups * P.n_loc : P.n_seg;     // parameter sets: one pass 0 per beam group
            const int A = P.pass0_az, lgA = 31 - __builtin_clz(A), Sw = kRaysPerWave >> lgA;
            const int n_ab = (ns + A - 1) >> lgA;
            const int w = gx * (kRaysPerBlock / kRaysPerWave) + (rr >> 4), r16 = rr & 15;
            const int sb = w / n_ab, ab = w - sb * n_ab;
            k = sb * Sw + (r16 >> lgA); seg = (ab << lgA) + (r16 & (A - 1));
            live = k < P.n_beam && seg < ns;
            if (!live) { seg = 0; k = 0; }
            else if (P.set_mode) { const int g = seg / P.n_loc; seg = (int)P.group_frame[g] * P.n_loc + (seg - g * P.n_loc); }   // the group's first frame
        }
        // pass 0 is traced in a sorted order of the beam samples (rows of equal elevation); results are
        // stored under the wave's own index j, so the reference order is untouched
        j = -1;
        V3 orig = { 0.0f, 0.0f, 0.0f }, dir = { 1.0f, 0.0f, 0.0f };
        if (live) {
            const int bb = FIRST ? beam_base(P, seg / P.n_loc) : 0;
            uint2 tj = make_uint2(0u, 0u);
            if (!FIRST) tj = P.torder[cur][(size_t)seg * P.cap + k];
            j = FIRST ? (int)P.beam_order[bb + k] : (int)tj.x;
            if (FIRST) {
                const float4 b = P.beams[bb + j];
                dir = { b.x, b.y, b.z };
            } else {
                const uint32_t slot = tj.y;           // = idx[cur][j], carried by the trace order (k_scan)
                const size_t w = (size_t)seg * 2 * P.cap + slot;
                const float4 A = P.waves[cur].A[w], B = P.waves[cur].B[w];
                orig = { A.x, A.y, A.z };
                dir = { A.w, B.x, B.y };
            }
        }
        Quat q_am; V3 t_am;
        azimuth_frame<FIRST>(P, poses, seg, q_am, t_am);
        // (pass 0: every ray starts in the sensor's origin -- rotating the zero vector is +-0, see azimuth_frame)
        R = ray_setup(FIRST ? t_am : v_add(q_rot(q_am, orig), t_am), q_rot(q_am, dir));
    }
    const bool active = j >= 0;

    unsigned nn = 0, nt = 0;
    unsigned ws[4] = { 0, 0, 0, 0 };
    if (active) {
        const int gray = ((FIRST ? 0 : seg_y) * gdim_x + (int)gx) * kRaysPerBlock + r;   // spill column of this ray slot
        Hit h;
        if constexpr (SL) h = traverse_stackless(reinterpret_cast<const float4*>(P.nodes), P.tri_base4, R, P.range_max, P.hit_pad);
        else h = traverse<STATS, SPILL, CULL>(reinterpret_cast<const float4*>(P.nodes), P.tri_base4, R, P.range_max, P.hit_pad, lds_stack, P.stack_lds,
                                              P.spill, P.spill_stride, gray, nn, nt, STATS ? ws : nullptr);
        if (q == 0) {
            const size_t hk = (size_t)seg * P.cap + j;
            P.hit[hk] = make_uint2(__float_as_uint((h.tri != 0xFFFFFFFFu) ? h.t : -1.0f), h.tri);
        }
    }
    if (STATS) {
        unsigned it = nt >> 16; nt &= 0xFFFFu;
        for (int off = 32; off > 0; off >>= 1) { nn += __shfl_down(nn, off); nt += __shfl_down(nt, off); it = max(it, (unsigned)__shfl_down(it, off)); }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.counters->nodes, (unsigned long long)nn);
            atomicAdd(&P.counters->tris, (unsigned long long)nt);
            atomicAdd(&P.counters->wave_iters, (unsigned long long)it);
            atomicMax(&P.counters->max_iters, it);
            atomicAdd(&P.counters->n_waves, 1u);
        }
        // the lane that stayed longest in the loop saw every iteration of the wave
        unsigned a = ws[0], b = ws[1], c2 = ws[2], d2 = ws[3];
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned a2 = __shfl_down(a, off), b2 = __shfl_down(b, off), c3 = __shfl_down(c2, off), d3 = __shfl_down(d2, off);
            if (a2 > a) { a = a2; b = b2; c2 = c3; d2 = d3; }
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.counters->it_all, (unsigned long long)a);
            atomicAdd(&P.counters->it_node, (unsigned long long)b);
            atomicAdd(&P.counters->it_leaf, (unsigned long long)c2);
            atomicAdd(&P.counters->quad_steps, (unsigned long long)d2);
        }
    }
}

// grid: (ceil(bound/16) [+ a copy row], n_seg), block 64 (= one wave = 16 rays), dynamic LDS = stack_lds * 16 * (4 | 6) B
template <bool FIRST, bool STATS, bool SPILL, bool CULL, bool SL = false>
__global__ __launch_bounds__(kTraceThreads) void k_trace(const Params P, const int pass, const typename PosesOf<FIRST>::type poses)
{
    // k_trace is the long pole of a step: its waves go first when they share a SIMD with the
    // column / shade waves of the other steps in flight (config 2: live launch 135 -> 122 us)
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ uint32_t lds_stack[];
    const int cur = pass & 1;
    if (FIRST && blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) P.counters->overflow = 0;   // error bits of this frame
        if (P.grid_hint && threadIdx.x < kMaxPasses) P.grid_hint->ovf_n[threadIdx.x] = 0;   // overflow lists of the tightened rows
        if constexpr (FIRST) {      // the poses, for every launch behind this one (rows of 8 floats: q.xyzw, t.xyz, 0)
            float* table = reinterpret_cast<float*>(P.pose_table);
            for (int i = threadIdx.x; i < 8 * poses.n; i += kTraceThreads) table[i] = (i & 7) < 7 ? poses.p[i >> 3][i & 7] : 0.0f;
        }
    }
    // pass 0: the rays of ALL segments of the launch are tiled into waves of (16 / A) beam samples x A
    // neighbouring azimuths (A = pass0_az = 16: ONE sample in 16 azimuths, i.e. 16 rays of identical
    // elevation, 0.9 degrees apart, that walk almost the same nodes in lockstep -- few iterations pay both
    // the node and the leaf path, no slow ray to wait for), and the tiles are launched sample-block-major,
    // so the waves that run at the same time are neighbours in azimuth: k_trace 128 us (round 1: Morton
    // order of the samples inside one azimuth, azimuth after azimuth) -> 80 us per 640k rays at config 2.
    // Every wave is full, too (200 rays per azimuth would otherwise leave a 13th wave with 8 rays)
    // later passes may carry a host copy in row 0 of the grid (see Params::copy_src)
    const int row0 = FIRST ? 0 : (P.copy_blocks > 0 ? 1 : 0);
    if (!FIRST && row0 && blockIdx.y == 0) {
        if ((int)blockIdx.x < P.copy_blocks) {
            __builtin_amdgcn_s_setprio(0);
            const size_t nthreads = (size_t)P.copy_blocks * kTraceThreads;
            for (size_t i = (size_t)blockIdx.x * kTraceThreads + threadIdx.x; i < P.copy_n16; i += nthreads) {
                const uint4 v = P.copy_src[i];
                P.copy_dst[i] = v;
                // ONE store per wave in flight: the writes leave at the pace PCIe takes them instead of filling the
                // memory pipeline's write queues, where the stores of every other kernel would wait behind them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        return;
    }
    int seg_y = (int)blockIdx.y - row0, gx = (int)blockIdx.x, gdim = (int)gridDim.x;
    if (!FIRST && P.seg_chunk > 0) {
        // chunks of S neighbouring segments; inside a chunk the segment is the fast dimension: y - row0 = chunk * row + group
        gdim = P.trace_row;
        const int yy = (int)blockIdx.y - row0, ch = yy / gdim;
        gx = yy - ch * gdim;
        seg_y = ch * P.seg_chunk + (int)blockIdx.x;
        if (seg_y >= P.n_seg) return;
    }
    const int count = FIRST ? 0 : (int)P.count[cur][seg_y];
    if (!FIRST && gx * kRaysPerBlock >= count) return;
    const PoseArgs* pa = nullptr;
    if constexpr (FIRST) pa = &poses;
    trace_group<FIRST, STATS, SPILL, CULL, SL>(P, pass, seg_y, gx, count, lds_stack, gdim, pa);
}

// The remainder of the segments whose count exceeds the tightened row of this pass (GridHint, rr_device.h): launched
// behind every tightened k_trace launch with a small fixed grid; every workgroup exits at once when the list is empty.
// Off the fast path on purpose: the loops here are what the main kernel must not contain (a loop around the ray set-up
// makes the compiler hoist the kernel-argument loads: 36 -> 94-102 SGPRs, 53 -> 58 VGPRs, +5 % per launch, round 4)
template <bool CULL>
__global__ __launch_bounds__(kTraceThreads) void k_trace_repair(const Params P, const int pass)
{
    extern __shared__ uint32_t lds_stack[];
    const uint32_t n = P.grid_hint->ovf_n[pass];
    if (n == 0) return;
    __builtin_amdgcn_s_setprio(2);
    const int tight = (int)P.tight_groups[pass];
    for (uint32_t i = 0; i < n; i++) {
        const int seg = (int)P.ovf_list[(size_t)pass * P.ovf_stride + i];
        const int count = (int)P.count[pass & 1][seg];
        for (int g = tight + (int)blockIdx.x; g * kRaysPerBlock < count; g += (int)gridDim.x) {
            trace_group<false, false, false, CULL>(P, pass, seg, g, count, lds_stack, 0);
            if (threadIdx.x == 0) atomicAdd(&P.grid_hint->repaired, 1ull);
        }
    }
}

// generic rays (tests): one quad per ray
// steps != null: the statistics build of the traversal; *steps += the traversal steps (node + leaf) of every ray -- what
// rr_set_mesh compares candidate trees by
template <bool CULL, bool STATS>
__global__ __launch_bounds__(kTraceThreads) void k_debug_trace(const Params P, const float* origs, const float* dirs, int n,
                                                               float* out_t, uint32_t* out_face, unsigned long long* steps)
{
    extern __shared__ uint32_t lds_stack[];
    const int i = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 2);
    if (i >= n) return;
    const V3 o = { origs[3 * i], origs[3 * i + 1], origs[3 * i + 2] };
    const V3 d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
    unsigned nn = 0, nt = 0;
    const RaySetup R = ray_setup(o, d);
    const Hit h = traverse<STATS, true, CULL>(reinterpret_cast<const float4*>(P.nodes), P.tri_base4, R, P.range_max, P.hit_pad, lds_stack, P.stack_lds,
                                  P.spill, P.spill_stride, i, nn, nt);
    if ((threadIdx.x & 3) == 0) {
        if (out_t) out_t[i] = (h.tri != 0xFFFFFFFFu) ? h.t : -1.0f;
        if (out_face) out_face[i] = h.face;
        if (STATS) atomicAdd(steps, (unsigned long long)(nt >> 16));     // high half: loop iterations of this ray
    }
}

// ---------------------------------------------------------------------------
// per-hit radar math
// ---------------------------------------------------------------------------
// acos(float) in the reference is the C++ float overload (acosf): the angle is
// an f32 value widened to f64 (this is what makes R_eff > 1 for v2 = 0, see
// SURVEY.md §8c).  Computed as the f32 rounding of the f64 acos.
__device__ inline float acosf_ref(float x) { return (float)acos((double)x); }

// radar_algorithms.h:55-139 (only dir + energy of both results are used,
// RadarCPU.cpp:285-286,364-365)
// angle of total reflection, radar_algorithms.h:77-83: asin(n2 / n1) where defined, else 100.  It depends on the pair of
// velocities only, i.e. on the material entry: tabulated once per material table instead of once per wave-pass
__device__ inline double fresnel_angle_limit(double v1, double v2)
{
    const double n1 = v2, n2 = v1;
    double angle_limit = 100.0;
    if (n1 > 0.0) { const double n21 = n2 / n1; if (fabs(n21) <= 1.0) angle_limit = asin(n21); }
    return angle_limit;
}
__global__ void k_mat_limits(const float4* materials, size_t n, double* limits)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) limits[i] = fresnel_angle_limit(0.3, (double)materials[i].x);
}
void launch_mat_limits(const float4* materials, size_t n, double* limits, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_mat_limits, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, materials, n, limits);
}

__device__ inline void fresnel_split(V3 n, const V3 d, const float incidence_f, const double energy, const double v1, const double v2,
                                     const double angle_limit, V3& rdir, double& renergy, V3& tdir, double& tenergy)
{
    const double polarization = 0.5;   // RadarCPU.cpp:108
    const double n1 = v2, n2 = v1;     // radar_algorithms.h:62-63
    // acosf(-d . n): the caller computed it (the BRDF angle of RadarCPU.cpp:308 is the same expression on the same values)
    const double incidence_angle = (double)incidence_f;
    rdir = v_add(d, v_scale(v_scale(n, 2.0f), v_dot(v_neg(n), d)));   // :73
    tdir = { 0.0f, 0.0f, 0.0f };
    bool transmitted = false;          // false: tdir stays the zero vector of radar_algorithms.h:66
    if (n1 > 0.0) {
        if (incidence_angle <= angle_limit) {
            if (v_dot(n, d) > 0.0f) n = v_neg(n);   // :92
            if (n2 > 0.0) {
                const double n12 = n1 / n2;
                const double c = cos(incidence_angle);
                tdir = v_add(v_scale(d, (float)n12),
                             v_scale(n, (float)(n12 * c - sqrt(1 - n12 * n12 * (1 - c * c)))));   // :100
                transmitted = true;
            }
        }
    }
    double rs, rp;
    const double eps = 0.0001;
    if (!transmitted) {
        // :106 for the zero vector: the dot product is (+-)0 and acosf(+-0) is the f32 value of pi/2 -- every wave that meets
        // an opaque material (v = 0: the KAIST wall) or is totally reflected.  h = (double)(float)(pi/2) is a CONSTANT, so
        // sin / cos of (incidence -+ h) come from ONE sincos(incidence) and the angle-sum identities with sin(h), cos(h)
        // as correctly rounded f64 constants, instead of two sincos with two argument reductions.  cos(h) = -4.37e-8 (h is
        // not pi/2: that is what makes R_eff exceed 1 by 1e-7 for v2 = 0) -- each sum below has one term of
        // that size and one >= 1e-4 (the s > pi - eps branch takes everything nearer to grazing), so nothing cancels: the
        // results are within 2 ulp(f64) of the two-sincos form, the freedom the GPU's libm has against the host's anyway.
        const double h = (double)1.57079637050628662109375f;
        const double ch = -0x1.777a5cf72ceccp-25, sh = 0x1.ffffffffffff7p-1;
        const double s = incidence_angle + h;      // > eps always
        if (s > M_PI - eps) { rs = 1.0; rp = 1.0; }
        else {
            double si, ci;
            sincos(incidence_angle, &si, &ci);
            const double a = si * ch, b = ci * sh, e = ci * ch, f = si * sh;
            const double sd = a - b, ss = a + b, cd = e + f, cs = e - f;     // sin / cos of (incidence - h), (incidence + h)
            rs = -sd / ss;
            rp = (sd / cd) / (ss / cs);
        }
    } else {
        const double refraction_angle = (double)acosf_ref(v_dot(tdir, v_neg(n)));   // :106
        const double s = incidence_angle + refraction_angle;
        if (s < eps) {
            rs = (n1 - n2) / (n1 + n2);
            rp = rs;
        } else if (s > M_PI - eps) {
            rs = 1.0; rp = 1.0;
        } else {
            // rs = -sin(df) / sin(s), rp = tan(df) / tan(s)  (radar_algorithms.h:116-121).  One sincos per angle
            // (one argument reduction) instead of sin + tan; tan = sin / cos is within an ulp of libm's tan --
            // the same freedom the GPU's libm already has against the host's
            const double df = incidence_angle - refraction_angle;
            double sd, cd, ss, cs;
            sincos(df, &sd, &cd);
            sincos(s, &ss, &cs);
            rs = -sd / ss;
            rp = (sd / cd) / (ss / cs);
        }
    }
    const double Rs = rs * rs, Rp = rp * rp;
    const double Reff = polarization * Rs + (1.0 - polarization) * Rp;
    const double Teff = 1.0 - Reff;
    renergy = Reff * energy;
    tenergy = Teff * energy;
}

// test hook (rr_debug_fresnel): fresnel_split exactly as k_shade calls it -- the incidence angle by acosf of the f32 dot product
// (one value for Fresnel and BRDF), the angle of total reflection by the function that fills the material table
__global__ void k_debug_fresnel(size_t n, const float* normals, const float* dirs, const double* energy, const double* v1, const float* v2,
                                float* out_rdir, double* out_re, float* out_tdir, double* out_te)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 nrm = { normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] }, d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
    const float inc = acosf_ref(v_dot(v_neg(d), nrm));
    const double v2d = (double)v2[i];
    V3 rdir, tdir; double re, te;
    fresnel_split(nrm, d, inc, energy[i], v1[i], v2d, fresnel_angle_limit(v1[i], v2d), rdir, re, tdir, te);
    out_rdir[3 * i] = rdir.x; out_rdir[3 * i + 1] = rdir.y; out_rdir[3 * i + 2] = rdir.z;
    out_tdir[3 * i] = tdir.x; out_tdir[3 * i + 1] = tdir.y; out_tdir[3 * i + 2] = tdir.z;
    out_re[i] = re; out_te[i] = te;
}
void launch_debug_fresnel(size_t n, const float* normals, const float* dirs, const double* energy, const double* v1, const float* v2,
                          float* out_rdir, double* out_re, float* out_tdir, double* out_te, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_debug_fresnel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, normals, dirs, energy, v1, v2, out_rdir, out_re, out_tdir, out_te);
}

// NOT in the reference checkout (its Cook-Torrance model lives on the dev/flex branch, README.md:83-85):
// this build's own specification for BASELINE.json configs[4], PARITY UNPINNED.  The cos^C lobe of the
// shader below is replaced by the microfacet backscatter lobe D_GGX * G_Smith, normalised to 1 at normal
// incidence (monostatic radar: light, view and half vector coincide, so the Fresnel term is the one already
// in the reflected energy), with alpha^2 = 2 / (C + 2) (Blinn-Phong exponent -> GGX roughness).  f32, un-fused,
// the same operation order as the CPU twin the parity tests compare it with.
__device__ inline float ct_lobe(float angle, float specular_exp)
{
    float sn, c;
    sincosf(angle, &sn, &c);
    if (!(c > 0.0f)) return 0.0f;
    float a2 = 2.0f / (specular_exp + 2.0f);
    a2 = fminf(fmaxf(a2, 1e-4f), 1.0f);
    const float cc = c * c;
    const float d = a2 * cc + sn * sn;              // = cc (a2 - 1) + 1 without the cancellation near normal incidence
    const float D = a2 / (d * d);
    const float g1 = (2.0f * c) / (c + sqrtf(a2 + (1.0f - a2) * cc));
    return a2 * D * (g1 * g1);
}

// radar_algorithms.h:168-187 (model 0); model 1: the lobe above in place of cos^C
__device__ inline float back_reflection_shader(float incidence_angle, float energy,
                                               float diffuse, float specular_fac, float specular_exp, int model)
{
    float I_specular;
    if (model == 1) I_specular = ct_lobe(incidence_angle, specular_exp);
    else {
        const float IdotR = cosf(incidence_angle);
        I_specular = powf(IdotR, specular_exp);
    }
    const float I_total = diffuse * 1.0f + specular_fac * I_specular;
    return I_total * energy;
}

// test hook (rr_debug_brdf): the shader exactly as k_shade calls it; in = [n][5] (angle, energy, ambient, diffuse, specular)
__global__ void k_debug_brdf(size_t n, const float* __restrict__ in, int model, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = back_reflection_shader(in[5 * i], in[5 * i + 1], in[5 * i + 2], in[5 * i + 3], in[5 * i + 4], model);
}
void launch_debug_brdf(size_t n, const float* in, int model, float* out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_debug_brdf, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, in, model, out);
}

// RadarCPU.cpp:410-413: time -> range bin
__device__ inline int signal_cell(double time, double resolution)
{
    const float half_time = (float)(time / 2.0);
    const float signal_dist = (float)(0.3 * (double)half_time);
    return (int)((double)signal_dist / resolution);
}

// grid: (ceil(cap/64), n_seg), block 64 (one wave: no workgroup waits for its slowest wave)
template <bool FIRST>
__global__ __launch_bounds__(64) void k_shade(const Params P, const int pass)
{
    const int seg = blockIdx.y;
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int cur = pass & 1, nxt = cur ^ 1;
    const int count = FIRST ? P.n_beam : (int)P.count[cur][seg];
    if (j >= count) return;
    const int frame = seg / P.n_loc;
    const int np_f = passes_of(P, frame);          // parameter sets: this frame's own number of passes
    const bool last = (pass == np_f - 1);

    const size_t base2 = (size_t)seg * 2 * P.cap;
    const size_t s0 = base2 + 2 * (size_t)j, s1 = s0 + 1;
    if (FIRST && pass >= np_f) {                    // a set with n_reflections = 0: no ray is cast, the image stays empty
        P.cflag[s0] = 0; P.cflag[s1] = 0;
        const SigRec none = { -1, 0.0f };
        P.sigtmp[s0] = none; P.sigtmp[s1] = none;
        return;
    }

    V3 orig = { 0.0f, 0.0f, 0.0f }, dir;
    double energy = 1.0, time = 0.0;   // RadarCPU.cpp:107,112
    uint32_t mat = 0;                   // RadarCPU.cpp:111
    if (FIRST) {
        const float4 b = P.beams[beam_base(P, frame) + j];
        dir = { b.x, b.y, b.z };
    } else {
        const uint32_t slot = P.idx[cur][(size_t)seg * P.cap + j];
        const size_t w = base2 + slot;
        const float4 A = P.waves[cur].A[w], B = P.waves[cur].B[w];
        const double2 C = P.waves[cur].C[w];
        orig = { A.x, A.y, A.z };
        dir = { A.w, B.x, B.y };
        mat = __float_as_uint(B.z);
        energy = C.x; time = C.y;
    }

    uint8_t f0 = 0, f1 = 0;
    SigRec sg0 = { -1, 0.0f }, sg1 = { -1, 0.0f };

    // parameter sets: the frames of a beam group share the rays of pass 0, traced once for the group's first frame
    const size_t hk = (size_t)((FIRST && P.set_mode) ? (int)P.group_frame[P.frame_beam[frame]] * P.n_loc + (seg - frame * P.n_loc) : seg) * P.cap + j;
    const uint2 hitrec = P.hit[hk];
    const float range = __uint_as_float(hitrec.x);
    if (range >= 0.0f)   // miss => the wave dies silently (RadarCPU.cpp:252-255)
    {
        f0 |= 4;
        const uint32_t tri = hitrec.y;
        const float4* tp = reinterpret_cast<const float4*>(P.tris + tri);
        const float4 tb = tp[1], tc = tp[2];
        const uint32_t obj_id = __float_as_uint(tb.w);
        const V3 e1 = { tb.x, tb.y, tb.z }, e2 = { tc.x, tc.y, tc.z };

        Quat q_am; V3 t_am;
        azimuth_frame<false>(P, nullptr, seg, q_am, t_am);
        // rmagine: geometric normal, normalised, rotated into the sensor frame,
        // flipped to oppose the ray; RadarCPU.cpp:248 normalises once more
        V3 nint = v_normalize(v_cross(e1, e2));
        nint = q_rot(q_conj(q_am), nint);
        if (v_dot(dir, nint) > 0.0f) nint = v_neg(nint);
        const V3 normal = v_normalize(nint);

        // incidence = wave.move(range)  (radar_types.h:108-120)
        const V3 dir_in = dir;
        orig = v_add(orig, v_scale(dir, range));
        time += (double)range / 0.3;

        // material select (RadarCPU.cpp:266-280)
        bool ok = true;
        uint32_t mat_refr;
        if ((int)mat == P.material_id_air) {
            if (obj_id >= (uint32_t)P.n_objects) { ok = false; mat_refr = 0; }
            else mat_refr = (uint32_t)P.object_materials[obj_id];
        } else {
            mat_refr = (uint32_t)P.material_id_air;
        }
        if (ok && mat_refr >= (uint32_t)P.n_materials) ok = false;
        if (!ok) {
            atomicOr(&P.counters->overflow, 2u); atomicOr(P.sticky, 2u);
        } else {
            const float4 m = P.materials[(size_t)frame * P.mat_stride + mat_refr];
            const float v_refraction = (mat != mat_refr) ? m.x : (float)0.3;
            const double angle_limit = (mat != mat_refr) ? P.mat_limits[(size_t)frame * P.mat_stride + mat_refr] : P.limit_same;

            V3 rdir, tdir; double renergy, tenergy;
            const float incidence_angle = acosf_ref(v_dot(v_neg(dir_in), normal));   // radar_algorithms.h:69 and RadarCPU.cpp:308: one value
            fresnel_split(normal, dir_in, incidence_angle, energy, 0.3, (double)v_refraction, angle_limit, rdir, renergy, tdir, tenergy);

            const float skip_dist = 0.001f;   // RadarCPU.cpp:374
            if (renergy > (double)P.thr)      // :288
            {
                if (!last) {
                    f0 |= 1;
                    const V3 o2 = v_add(orig, v_scale(rdir, skip_dist));
                    const double t2 = time + (double)skip_dist / 0.3;
                    P.waves[nxt].A[s0] = make_float4(o2.x, o2.y, o2.z, rdir.x);
                    P.waves[nxt].B[s0] = make_float4(rdir.y, rdir.z, __uint_as_float(mat), 0.0f);
                    P.waves[nxt].C[s0] = make_double2(renergy, t2);
                }
                if ((int)mat == P.material_id_air)   // :302
                {
                    const float ret = back_reflection_shader(incidence_angle, (float)renergy, m.y, m.z, m.w, P.brdf_model);
                    if (pass == 0 || P.record_multi_reflection) {   // :319
                        const float time_back = (float)(time * 2.0);
                        sg0.cell = signal_cell((double)time_back, P.resolution);
                        sg0.strength = ret;
                        if (sg0.cell < 0) sg0.cell = -1;
                    }
                    if (pass > 0 && P.record_multi_path) {   // :325-360
                        const float dist = sqrtf(orig.x * orig.x + orig.y * orig.y + orig.z * orig.z);
                        const V3 dsh = { orig.x / dist, orig.y / dist, orig.z / dist };
                        const double time_to_sensor = (double)dist / 0.3;
                        const double sensor_view_scalar = (double)v_dot(dir_in, dsh);
                        const float ang = acosf_ref(v_dot(v_neg(rdir), dsh));
                        if (sensor_view_scalar > P.multipath_threshold) {
                            sg1.strength = back_reflection_shader(ang, (float)renergy, m.y, m.z, m.w, P.brdf_model);
                            sg1.cell = signal_cell(time + time_to_sensor, P.resolution);
                            if (sg1.cell < 0) sg1.cell = -1;
                        }
                    }
                }
            }
            if (!last && tenergy > (double)P.thr)   // :367
            {
                f1 |= 1;
                const V3 o2 = v_add(orig, v_scale(tdir, skip_dist));
                const double t2 = time + (double)skip_dist / 0.3;
                P.waves[nxt].A[s1] = make_float4(o2.x, o2.y, o2.z, tdir.x);
                P.waves[nxt].B[s1] = make_float4(tdir.y, tdir.z, __uint_as_float(mat_refr), 0.0f);
                P.waves[nxt].C[s1] = make_double2(tenergy, t2);
            }
        }
    }
    P.cflag[s0] = f0; P.cflag[s1] = f1;
    P.sigtmp[s0] = sg0; P.sigtmp[s1] = sg1;
}

// ---------------------------------------------------------------------------
// ordered compaction: children -> idx[nxt], signals -> sig list.  grid n_seg, block 256
// ---------------------------------------------------------------------------
__device__ inline int block_excl_scan(int v, int& total, int* lds /*[8]*/)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int w = 0; w < 4; w++) { const int s = lds[w]; if (w < wid) pre += s; tot += s; }
    __syncthreads();
    total = tot;
    return pre + x - v;
}

template <bool FIRST>
__global__ __launch_bounds__(256) void k_scan(const Params P, const int pass)
{
    __shared__ int lds[8];
    const int seg = blockIdx.x;
    const int cur = pass & 1, nxt = cur ^ 1;
    const int count = FIRST ? P.n_beam : (int)P.count[cur][seg];
    const int n_slots = 2 * count;
    const size_t base2 = (size_t)seg * 2 * P.cap;

    int n_child = 0;
    int n_sig = FIRST ? 0 : (int)P.sig_count[seg];
    const int sig_before = n_sig;
    int n_hit = 0, n_refl = 0, my_hits = 0;
    unsigned ovf = 0;
    // the loads of block b + 1 are in flight while block b is scanned (a block is two barriers and a dependent scatter)
    uint8_t f_nx = 0; SigRec sr_nx = { -1, 0.0f };
    if ((int)threadIdx.x < n_slots) { f_nx = P.cflag[base2 + threadIdx.x]; sr_nx = P.sigtmp[base2 + threadIdx.x]; }
    for (int b = 0; b < n_slots; b += 256) {
        const int s = b + threadIdx.x;
        const uint8_t f = f_nx; const SigRec sr = sr_nx;
        f_nx = 0; sr_nx = { -1, 0.0f };
        if (s + 256 < n_slots) { f_nx = P.cflag[base2 + s + 256]; sr_nx = P.sigtmp[base2 + s + 256]; }
        const int hit = (f >> 2) & 1;
        // the counts of a 256-slot block ride in ONE scan, 10 bits apart: children (even slots = reflections and odd slots
        // = refractions apart: the trace order below needs the number of reflections), signals; hits are only summed
        const int c = f & 1;
        const int g = sr.cell >= 0 ? 1 : 0;
        int tot;
        const int pre = block_excl_scan(((s & 1) ? (c << 10) : c) | (g << 20), tot, lds);
        const int pos = n_child + (pre & 1023) + ((pre >> 10) & 1023);
        if (c) { if (pos < P.cap) { P.idx[nxt][(size_t)seg * P.cap + pos] = (uint32_t)s; P.refpos[base2 + s] = (uint32_t)pos; } else { ovf = 1; P.refpos[base2 + s] = 0xFFFFFFFFu; } }
        n_child += (tot & 1023) + ((tot >> 10) & 1023);
        n_refl += tot & 1023;
        const int spos = n_sig + ((pre >> 20) & 1023);
        if (g) { if (spos < P.sigcap) P.sig[(size_t)seg * P.sigcap + spos] = sr; else ovf = 1; }
        n_sig += (tot >> 20) & 1023;
        my_hits += hit;
    }
    {   // hits of the pass: one reduction instead of a scan per block
        int tot;
        block_excl_scan(my_hits, tot, lds);
        n_hit = tot;
    }
    // trace order of the next pass: children in the (spatially sorted) trace order of their
    // parents, all reflections first, then all refractions -> neighbouring quads stay coherent
    __syncthreads();
    {
        // both children of a parent in one sweep: reflections fill [0, R), refractions [R, ...), R = the number of PLACED
        // reflections (a child beyond the capacity has no reference position and no place in the trace order)
        int n_refl_placed = 0;
        {
            // reflections sit at even slots; the ordered list holds slots in increasing order, so the placed ones are a
            // prefix: count those among the first min(n_child, cap) entries = reflections whose position is below cap
            n_refl_placed = n_refl;
            if (n_child > P.cap) {       // overflow (reported): recount exactly
                int mine = 0;
                for (int k = threadIdx.x; k < count; k += 256) {
                    const size_t sl = base2 + 2 * (size_t)k;
                    if ((P.cflag[sl] & 1) && P.refpos[sl] != 0xFFFFFFFFu) mine++;
                }
                int tot; block_excl_scan(mine, tot, lds); n_refl_placed = tot;
            }
        }
        int placed0 = 0, placed1 = 0;
        auto fetch = [&](int k, int& c0, int& c1, uint32_t& rp0, uint32_t& rp1, uint32_t& jp) {
            c0 = c1 = 0; rp0 = rp1 = 0xFFFFFFFFu; jp = 0;
            if (k < count) {
                const uint32_t j = FIRST ? P.beam_order2[beam_base(P, seg / P.n_loc) + k] : P.torder[cur][(size_t)seg * P.cap + k].x;
                const size_t sl = base2 + 2 * (size_t)j;
                const uint2 rp = *reinterpret_cast<const uint2*>(P.refpos + sl);      // both children: one 8-B load (sl is even)
                const uint8_t f0 = P.cflag[sl], f1 = P.cflag[sl + 1];
                if (f0 & 1) { rp0 = rp.x; c0 = rp0 != 0xFFFFFFFFu; }
                if (f1 & 1) { rp1 = rp.y; c1 = rp1 != 0xFFFFFFFFu; }
                jp = j;
            }
        };
        int c0n, c1n; uint32_t rp0n, rp1n, jpn;
        fetch((int)threadIdx.x, c0n, c1n, rp0n, rp1n, jpn);
        for (int b = 0; b < count; b += 256) {
            const int c0 = c0n, c1 = c1n; const uint32_t rp0 = rp0n, rp1 = rp1n, jp = jpn;
            fetch(b + 256 + (int)threadIdx.x, c0n, c1n, rp0n, rp1n, jpn);
            int tot;
            const int pre = block_excl_scan(c0 | (c1 << 10), tot, lds);
            // (position in the next pass' list, child slot = 2 x the parent's position [+ 1]): the slot saves k_trace the idx lookup
            if (c0) P.torder[nxt][(size_t)seg * P.cap + placed0 + (pre & 1023)] = make_uint2(rp0, 2u * jp);
            if (c1) P.torder[nxt][(size_t)seg * P.cap + n_refl_placed + placed1 + ((pre >> 10) & 1023)] = make_uint2(rp1, 2u * jp + 1u);
            placed0 += tot & 1023; placed1 += (tot >> 10) & 1023;
        }
    }
    if (__syncthreads_or((int)ovf) && threadIdx.x == 0) { atomicOr(&P.counters->overflow, 1u); atomicOr(P.sticky, 1u); }
    if (threadIdx.x == 0) {
        P.count[nxt][seg] = (uint32_t)min(n_child, P.cap);
        if (P.grid_hint && pass + 1 < kMaxPasses) {     // tight trace grids (rr_device.h: GridHint)
            const uint32_t cn = (uint32_t)min(n_child, P.cap);
            // a monotone maximum: once it has settled nobody issues the atomic any more (a stale read only costs one)
            if (cn > P.grid_hint->hist[pass + 1]) atomicMax(&P.grid_hint->hist[pass + 1], cn);
            const uint32_t tg = P.tight_groups[pass + 1];
            if (tg && cn > tg * (uint32_t)kRaysPerBlock) {
                const uint32_t at = atomicAdd(&P.grid_hint->ovf_n[pass + 1], 1u);
                P.ovf_list[(size_t)(pass + 1) * P.ovf_stride + at] = (uint32_t)seg;
            }
        }
        P.sig_count[seg] = (uint32_t)min(n_sig, P.sigcap);
        SegStats st; st.wave_passes = (uint32_t)count; st.hits = (uint32_t)n_hit;
        st.signals = (uint32_t)(n_sig - sig_before); st.pad = 0;
        P.seg_stats[(size_t)pass * P.n_seg + seg] = st;
    }
}

// ---------------------------------------------------------------------------
// Perlin noise (image_algorithms.h:14-106), f64, table in constant memory
// ---------------------------------------------------------------------------
__constant__ unsigned char c_perm[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30,
    69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62,
    94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136,
    171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122,
    60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161,
    1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86,
    164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126,
    255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253,
    19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193,
    238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31,
    181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93,
    222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180
};
__device__ inline double p_fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
__device__ inline double p_lerp(double t, double a, double b) { return a + t * (b - a); }
// grad(hash, x, y, z = 0) of image_algorithms.h:57-67 is (+-u) + (+-v) with u, v picked from
// {x, y, 0} by the low four hash bits: a*x + b*y with a, b in {-1, 0, +1}.  Products with +-1
// and 0 are exact and the sum has the same two operands, so this is the same f64 value (only
// the sign of an exact zero can differ, which no later sum can see).  The (a, b) pair of
// pt[k] & 15 is tabulated per k so one LDS read replaces the hash decode.
__device__ inline double2 p_grad_coef(int hash)
{
    const int h = hash & 15;
    const double su = (h & 1) == 0 ? 1.0 : -1.0, sv = (h & 2) == 0 ? 1.0 : -1.0;
    double a = 0.0, b = 0.0;
    if (h < 8) a = su; else b = su;                 // u = h < 8 ? x : y
    if (h < 4) b = sv;                              // v = h < 4 ? y : (h == 12 || h == 14 ? x : z)
    else if (h == 12 || h == 14) a = sv;
    return make_double2(a, b);
}
__device__ inline double p_grad2(const double2 g, double x, double y) { return g.x * x + g.y * y; }
// perlin_noise(x, y, z = 0) of image_algorithms.h:69-106 along ONE image column.
// For z = 0: Z = 0, w = fade(0) = 0, and lerp(0, a, b) = a + 0*(b - a) = a exactly (b - a is
// finite), so the z-1 half of the lattice is not evaluated -- bit-identical result.
// Within a column sy is fixed, so Y, y and v = fade(y) are column constants and the two hash
// chains pt[(pt[X] + Y) & 255], pt[(pt[X] + Y + 1) & 255] depend on the lattice column X only:
// they are tabulated once per column (257 entries: X + 1 needs no wrap) together with the
// y-half of the gradient, entry = (a, b * y) resp. (a, b * (y - 1)) -- the same f64 product the
// reference forms per pixel -- so one evaluation is four 16-B LDS reads off ONE address,
// four (mul, add) gradients and three lerps.
constexpr int kPerlinRow = 257;
struct PerlinCol { double v; const double2* t0; const double2* t1; };   // t0: row y, t1: row y - 1
__device__ inline void perlin_col_table(const unsigned char* pt, double sy, double2* t0, double2* t1,
                                        int tid, int n_threads)
{
    const int Y = (int)floor(sy) & 255;
    const double y = sy - floor(sy);
    for (int e = tid; e < 2 * kPerlinRow; e += n_threads) {
        const int row = e >= kPerlinRow ? 1 : 0;
        const int X = (e - row * kPerlinRow) & 255;
        // hash of the lattice corner (X, Y + row, Z = 0): pt[pt[pt[X] + Y + row] + 0]  (image_algorithms.h:80-94); its gradient
        // is decoded here -- four entries per thread -- instead of through a 4-KB table of the 256 decodes
        const double2 g = p_grad_coef(pt[pt[(pt[X] + Y + row) & 255]]);
        (row ? t1 : t0)[e - row * kPerlinRow] = make_double2(g.x, g.y * (row ? y - 1 : y));
    }
}
__device__ inline double perlin_col(const PerlinCol& pc, double sx)
{
    const double fl = floor(sx);
    const int X = (int)fl & 255;
    const double x = sx - fl, x1 = x - 1;
    const double u = p_fade(x);
    const double2 e00 = pc.t0[X], e10 = pc.t0[X + 1], e01 = pc.t1[X], e11 = pc.t1[X + 1];
    return p_lerp(pc.v, p_lerp(u, e00.x * x + e00.y, e10.x * x1 + e10.y),
                        p_lerp(u, e01.x * x + e01.y, e11.x * x1 + e11.y));
}

// defined variate stream for ambient_noise == 1 (the reference draws from
// std::random_device, RadarCPU.cpp:461-482: unreproducible by construction)
__device__ inline float uniform01(uint32_t seed, uint32_t col, uint32_t i)
{
    unsigned long long z = ((unsigned long long)seed << 40) ^ ((unsigned long long)col << 20) ^ (unsigned long long)i;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// cv::saturate_cast<uchar>(float): cvRound (round-half-even); NaN / out of int range -> 0
__device__ inline uint8_t saturate_u8(float x)
{
    const float r = __builtin_amdgcn_fmed3f(__builtin_rintf(x), 0.0f, 255.0f);
    return (__builtin_fabsf(x) < 2147483648.0f) ? (uint8_t)(int)r : (uint8_t)0;
}

// range decay of the ambient noise floor, one value per bin (RadarCPU.cpp:519-521)
__global__ void k_decay_table(float* decay, int n_cells, double resolution, double energy_loss_d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const float energy_loss = (float)energy_loss_d;
    const float x = (float)(((double)(float)i + 0.5) * resolution);
    decay[i] = expf(-energy_loss * x);
}
void launch_decay_table(float* decay, int n_cells, double resolution, double energy_loss, hipStream_t s)
{
    hipLaunchKernelGGL(k_decay_table, dim3((n_cells + 255) / 256), dim3(256), 0, s, decay, n_cells, resolution, energy_loss);
}

// ---------------------------------------------------------------------------
// signals -> slice -> column  (RadarCPU.cpp:402-542).  grid n_seg, block 256.
// Each lane owns one range bin of a 64-bin tile in a register and replays the
// azimuth's signals IN ORDER (f64 add, f32 store like `slice.at<float>() +=`),
// so the column is bit-reproducible and equal to the sequential CPU loop.
// ---------------------------------------------------------------------------
#ifndef RR_SIG_CHUNK
#define RR_SIG_CHUNK 1024
#endif
constexpr int kSigChunk = RR_SIG_CHUNK;          // signals staged per round (8 B each)
// threads per azimuth column: 4 waves when a launch brings thousands of columns (frame batches: +0.5..1 % against 8 waves;
// 16 waves lose 12-26 %), 8 waves for the few hundred columns of a single frame (its k_column then takes 80 instead of 91 us)

template <int kColThreads>
__global__ __launch_bounds__(kColThreads) void k_column(const Params P)
{
    constexpr int kColWaves = kColThreads / 64;
    extern __shared__ float lds_col[];              // [n_cells] slice
    // One LDS region serves the two phases of a column (k_column's workgroups per CU are bounded by LDS: 40 KB each
    // gave 4, i.e. 4 waves per SIMD for a kernel that waits as long as it issues):
    //   replay phase: the current chunk of signals, the widened smear weights, the waves' replay lists
    //   noise phase : the permutation + gradient tables, then the four Perlin column tables built from them
    constexpr int kWPad = 64;
    constexpr size_t kReplayBytes = kSigChunk * sizeof(SigRec) + (256 + 2 * kWPad) * sizeof(double) + kColWaves * 64 * sizeof(int2);
    constexpr size_t kNoiseBytes = 4 * kPerlinRow * sizeof(double2) + 256;
    constexpr size_t kUnionBytes = kReplayBytes > kNoiseBytes ? kReplayBytes : kNoiseBytes;
    __shared__ __align__(16) unsigned char s_union[kUnionBytes];
    SigRec* s_sig = reinterpret_cast<SigRec*>(s_union);
    // smear weights, widened once (the replay multiplies in f64), with 64 entries of padding on either side: the 64
    // lanes of a tile that overlaps a signal's window then read 64 CONSECUTIVE doubles (bin - first lies in
    // [-63, W + 62]), conflict-free, and the lanes outside the window are dropped by the select, not by a clamped
    // address (the clamp sent them all to one address on a busy bank: 0.26 conflict cycles per LDS instruction)
    double* s_w = reinterpret_cast<double*>(s_union + kSigChunk * sizeof(SigRec));
    int2 (*s_list)[64] = reinterpret_cast<int2 (*)[64]>(s_union + kSigChunk * sizeof(SigRec) + (256 + 2 * kWPad) * sizeof(double));   // per wave: overlapping signals of one 64-signal batch
    double2* s_tab = reinterpret_cast<double2*>(s_union);                                   // noise phase
    unsigned char* s_perm = reinterpret_cast<unsigned char*>(s_tab + 4 * kPerlinRow);
    __shared__ unsigned long long s_tiles[2];
    __shared__ int s_odd;                             // the chunk holds an echo that is negative or not finite
    __shared__ float s_red[kColWaves];
    __shared__ int s_wcnt[kColWaves];

    const int seg = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n_cells = P.n_cells;
    const int W = P.signal_denoising > 0 ? P.smear_w : 1;
    const int mode = P.signal_denoising > 0 ? P.smear_mode : 0;
    const int n_tiles = (n_cells + 63) >> 6;

    // signal stream of this azimuth, in the reference's order: the compacted list of
    // passes 0..P-2 followed by the per-wave slots of the last pass (invalid = cell < 0)
    const int n_list = (P.n_passes > 1) ? (int)P.sig_count[seg] : 0;
    int count_last = 0;
    if (P.n_passes == 1) count_last = P.n_beam;
    else if (P.n_passes > 1) count_last = (int)P.count[(P.n_passes - 1) & 1][seg];
    const int n_slots = 2 * count_last;
    const int S = n_list + n_slots;
    const size_t base2 = (size_t)seg * 2 * P.cap;

    for (int i = tid; i < n_cells; i += kColThreads) lds_col[i] = 0.0f;
    for (int i = tid; i < 256 + 2 * kWPad; i += kColThreads) {
        const int k = i - kWPad;
        s_w[i] = (P.signal_denoising > 0 && k >= 0 && k < W) ? (double)P.smear[k] : 0.0;
    }
    __syncthreads();

    int n_valid_last = 0, n_hit_last = 0;
    float rmax = 0.0f;          // running maximum of the bins this lane accumulates (max_val, RadarCPU.cpp:404)
    for (int c0 = 0; c0 < S; c0 += kSigChunk) {
        const int n_in = min(kSigChunk, S - c0);
        if (tid < 2) s_tiles[tid] = 0ull;
        if (tid == 2) s_odd = 0;
        // stage the chunk, keeping only the signals that land in the image (RadarCPU.cpp:414), IN ORDER:
        // ballot rank inside the wave + the counts of the waves before it (the per-wave slots of the last
        // pass are half empty -- no multipath echo -- so the replay scans half as many entries)
        int n = 0;
        // 64-bin tiles touched by the kept signals of this chunk: collected per lane in registers and OR-ed into LDS
        // once per wave (one LDS atomic per signal and tile put every lane of a wave on the same two addresses:
        // 10.4 M bank-conflict cycles per launch on the 10M-triangle target, profiles/r02c_t_pmc_summary.json)
        unsigned long long tm0 = 0ull, tm1 = 0ull;
        bool odd = false;
        for (int r0 = 0; r0 < n_in; r0 += kColThreads) {
            const int i = r0 + tid;
            const int v = c0 + i;
            SigRec r; r.cell = -1; r.strength = 0.0f;
            if (i < n_in) {
                if (v < n_list) r = P.sig[(size_t)seg * P.sigcap + v];
                else {
                    r = P.sigtmp[base2 + (v - n_list)];
                    n_valid_last += r.cell >= 0;
                    n_hit_last += (P.cflag[base2 + (v - n_list)] >> 2) & 1;
                }
            }
            const bool keep = r.cell >= 0 && r.cell < n_cells;
            const unsigned long long km = __ballot(keep);
            if (lane == 0) s_wcnt[wid] = __builtin_popcountll(km);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < kColWaves; w++) { const int cw = s_wcnt[w]; before += w < wid ? cw : 0; total += cw; }
            if (keep) {
                const int pos = n + before + __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                s_sig[pos] = r;
                odd = odd || !(r.strength >= 0.0f && r.strength < __builtin_inff());
                int lo = r.cell - mode, hi = r.cell - mode + W - 1;
                lo = max(lo, 0); hi = min(hi, n_cells - 1);
                for (int t = lo >> 6; t <= (hi >> 6); t++) { if (t < 64) tm0 |= 1ull << t; else tm1 |= 1ull << (t - 64); }
            }
            n += total;
            __syncthreads();
        }
        for (int off = 32; off > 0; off >>= 1) { tm0 |= __shfl_xor(tm0, off); tm1 |= __shfl_xor(tm1, off); }
        const bool wave_odd = __ballot(odd) != 0ull;
        if (lane == 0) { if (tm0) atomicOr(&s_tiles[0], tm0); if (tm1) atomicOr(&s_tiles[1], tm1); if (wave_odd) atomicOr(&s_odd, 1); }
        __syncthreads();
        // Nearly every chunk holds only finite, non-negative echoes (a negative one needs cos^C with an odd C on a multipath
        // echo, a NaN a negative cosine under a fractional exponent).  Then (a) a bin never falls, so its running maximum is its
        // final value and (b) s * 0 is +0, so a lane OUTSIDE a signal's window may add its zero weight (the padded table) and
        // keeps its value bit for bit: the replay needs neither the window select nor the per-echo fmax -- 7 instead of 11
        // instructions on the chain that dominates this kernel.  Any other chunk takes the general form.
        const bool simple = s_odd == 0;
        for (int t = wid; t < n_tiles; t += kColWaves) {
            if (!((s_tiles[t >> 6] >> (t & 63)) & 1ull)) continue;
            const int g = t * 64 + lane;
            const int tlo = t * 64, thi = tlo + 63;
            const bool g_ok = g > 0 && g < n_cells;      // RadarCPU.cpp:424 (bin 0 is never written)
            float acc = (g < n_cells) ? lds_col[g] : 0.0f;
            const float acc_in = acc;
            // scan 64 signals per step; replay the overlapping ones in order.  A lane outside 0 < g < C
            // gets a bin index that fails every range test (RadarCPU.cpp:424: bin 0 is never written)
            const int gb8 = 8 * g;                           // byte offset of bin g in the f64 weight table
            const unsigned W8 = g_ok ? 8u * (unsigned)W : 0u;   // a lane outside 0 < g < C accepts no offset at all
            const char* wbytes = reinterpret_cast<const char*>(s_w);
            int2* list = s_list[wid];
            // one replay: acc = (float)((double)acc + (double)strength * w[g - first])   (RadarCPU.cpp:426)
#define RR_REPLAY(E)                                                                                     \
            {                                                                                            \
                const unsigned off = (unsigned)(gb8 - (E).x);                                            \
                const double wv = *reinterpret_cast<const double*>(wbytes + 8 * kWPad + (int)off);       \
                const float nv = (float)((double)acc + (double)__int_as_float((E).y) * wv);              \
                acc = (off < W8) ? nv : acc;                                                             \
                rmax = fmaxf(rmax, acc);       /* `if (slice > max_val) max_val = slice` (NaN never wins) */  \
            }
#define RR_REPLAY_S(E)                                                                                   \
            {                                                                                            \
                const double wv = *reinterpret_cast<const double*>(wbytes + 8 * kWPad + (gb8 - (E).x));  \
                acc = (float)((double)acc + (double)__int_as_float((E).y) * wv);                         \
            }
            for (int b0 = 0; b0 < n; b0 += 64) {
                const int i = b0 + lane;
                SigRec r; r.cell = 0x40000000; r.strength = 0.0f;
                if (i < n) r = s_sig[i];
                const int first = r.cell - mode;
                const bool ov = !(first > thi || first + W - 1 < tlo);
                unsigned long long m = __ballot(ov);
                if (m == 0ull) continue;
                if (P.signal_denoising > 0) {
                    // the overlapping signals of this batch, compacted in order into a wave-private list:
                    // the replay then reads them back with uniform (broadcast) LDS reads -- no lane
                    // extraction -- and the reads of several replays are in flight together; the
                    // f64-add / f32-round chain itself stays in signal order
                    const int cnt = __builtin_popcountll(m);
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (ov) list[pos] = make_int2(8 * first, __float_as_int(r.strength));
                    // same wave, in-order LDS: make the writes of the other lanes visible to the reads below
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int k = 0;
                    if (simple) {
                        for (; k + 4 <= cnt; k += 4) {
                            const int2 e0 = list[k], e1 = list[k + 1], e2 = list[k + 2], e3 = list[k + 3];
                            RR_REPLAY_S(e0) RR_REPLAY_S(e1) RR_REPLAY_S(e2) RR_REPLAY_S(e3)
                        }
                        for (; k < cnt; k++) { const int2 e = list[k]; RR_REPLAY_S(e) }
                    } else {
                        for (; k + 4 <= cnt; k += 4) {
                            const int2 e0 = list[k], e1 = list[k + 1], e2 = list[k + 2], e3 = list[k + 3];
                            RR_REPLAY(e0) RR_REPLAY(e1) RR_REPLAY(e2) RR_REPLAY(e3)
                        }
                        for (; k < cnt; k++) { const int2 e = list[k]; RR_REPLAY(e) }
                    }
                } else {
                    while (m) {
                        const int b = __builtin_ctzll(m); m &= m - 1;
                        const int f_b = __builtin_amdgcn_readlane(first, b);
                        const float s_b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(r.strength), b));
                        if (g - f_b == 0 && g < n_cells) { acc = fmaxf(acc, s_b); rmax = fmaxf(rmax, acc); }   // :439-448
                    }
                }
            }
#undef RR_REPLAY
#undef RR_REPLAY_S
            if (simple && P.signal_denoising > 0) {
                acc = g_ok ? acc : acc_in;            // bin 0 (RadarCPU.cpp:424) and the lanes beyond the column took part blindly
                rmax = fmaxf(rmax, acc);              // non-negative echoes: the running maximum of a bin is where it ends
            }
            if (g < n_cells) lds_col[g] = acc;
        }
        __syncthreads();
    }
    // counters of the last pass (earlier passes are written by k_scan)
    {
        __shared__ int s_cnt[2];
        if (tid < 2) s_cnt[tid] = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) { n_valid_last += __shfl_down(n_valid_last, off); n_hit_last += __shfl_down(n_hit_last, off); }
        if (lane == 0) { atomicAdd(&s_cnt[0], n_valid_last); atomicAdd(&s_cnt[1], n_hit_last); }
        __syncthreads();
        if (tid == 0 && P.n_passes > 0) {
            SegStats st; st.wave_passes = (uint32_t)count_last; st.hits = (uint32_t)s_cnt[1];
            st.signals = (uint32_t)s_cnt[0]; st.pad = 0;
            P.seg_stats[(size_t)(P.n_passes - 1) * P.n_seg + seg] = st;
        }
    }

    // max_val is the RUNNING maximum of RadarCPU.cpp:428-431 / :445-448: the largest value any bin held at
    // any time.  A multipath echo can be negative (cos^C with an odd C, RadarCPU.cpp:347-353), so a bin may
    // fall again: the lanes tracked their own running maximum while replaying (rmax), reduce that
    float m = rmax;
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    if (lane == 0) s_red[wid] = m;
    __syncthreads();
    float max_val = 0.0f;
    for (int k = 0; k < kColWaves; k++) max_val = fmaxf(max_val, s_red[k]);

    const int angle_id = P.az_begin + seg % P.n_loc;
    const int col = (P.scroll + angle_id) % P.n_angles;   // :457 (placement is done by the assemble step)
    const float final_scale = (float)(P.signal_max / (double)max_val);   // :533
    const float rnd = (P.ambient_noise && P.noise_rnd) ? P.noise_rnd[(size_t)((seg / P.n_loc) % P.noise_rows) * P.n_angles + angle_id] : 0.0f;
    PerlinCol pc1 = { 0.0, nullptr, nullptr }, pc2 = pc1;
    if (P.ambient_noise == 2) {   // the replay region is dead (barriers above): its LDS holds the noise tables now
        for (int i = tid; i < 256; i += kColThreads) s_perm[i] = c_perm[i];
        __syncthreads();
        double2* tab = s_tab;
        const double sy1 = (double)col * 0.05, sy2 = (double)col * 0.2;
        perlin_col_table(s_perm, sy1, tab, tab + kPerlinRow, tid, kColThreads);
        perlin_col_table(s_perm, sy2, tab + 2 * kPerlinRow, tab + 3 * kPerlinRow, tid, kColThreads);
        pc1 = { p_fade(sy1 - floor(sy1)), tab, tab + kPerlinRow };
        pc2 = { p_fade(sy2 - floor(sy2)), tab + 2 * kPerlinRow, tab + 3 * kPerlinRow };
        __syncthreads();
    }

    // noise amplitude of one bin as a function of its signal (RadarCPU.cpp:497-512)
    const float signal_max = max_val;
    const float signal_amp = signal_max - 0.0f;
    const float noise_at_0 = (float)((double)signal_amp * P.noise_at_0);
    const float noise_at_1 = (float)((double)signal_amp * P.noise_at_1);
    auto noise_amp_of = [&](float signal) -> float {
        const float signal_ = (float)(1.0 - (double)((signal - 0.0f) / signal_amp));
        // std::pow(signal_, 4.0) (RadarCPU.cpp:509): two exact-order squarings in f64 differ from a
        // correctly rounded pow by < 1.5 ulp(f64), invisible after the narrowing to f32
        const double sg2 = (double)signal_ * (double)signal_;
        const float signal__ = (float)(sg2 * sg2);
        return (float)((double)(signal__ * noise_at_0) + (1.0 - (double)signal__) * (double)noise_at_1);
    };
    // most bins hold no echo: their amplitude is the SAME expression evaluated at signal = 0 * energy_max,
    // computed once per column; a wave whose 64 bins are all empty takes it instead of the division chain
    const float amp_empty = P.ambient_noise ? noise_amp_of(0.0f * P.energy_max_f) : 0.0f;
    for (int i = tid; i < n_cells; i += kColThreads) {
        const float raw = lds_col[i];
        float v = raw * P.energy_max_f;   // :453
        if (P.ambient_noise) {   // :459-528
            const float signal = v;
            double p = 0.0;
            if (P.ambient_noise == 1) {
                p = (double)uniform01((uint32_t)(int)rnd, (uint32_t)col, (uint32_t)i);
            } else if (P.ambient_noise == 2) {
                const double random_begin = (double)rnd;
                const double p1 = perlin_col(pc1, random_begin + (double)i * 0.05);
                const double p2 = perlin_col(pc2, random_begin + (double)i * 0.2);
                p = 0.9 * p1 + 0.1 * p2;
            }
            // (raw != 0 is also true for NaN: such bins take the general expression)
            const float noise_amp = (__ballot(raw != 0.0f) == 0ull) ? amp_empty : noise_amp_of(signal);
            const float noise_energy_max = (float)((double)signal_max * P.noise_e_max);
            const float noise_energy_min = (float)((double)signal_max * P.noise_e_min);
            float y_noise = (float)((double)noise_amp * p);
            // expf(-energy_loss * x) with x = (float)((i + 0.5) * resolution) (RadarCPU.cpp:519-521)
            // depends on the bin only: tabulated by k_decay_table with the same arithmetic
            y_noise = y_noise + (noise_energy_max - noise_energy_min) * P.decay[i] + noise_energy_min;
            y_noise = fabsf(y_noise);
            v = signal + y_noise;
        }
        v = v * final_scale;
        P.cols_u8[(size_t)seg * n_cells + i] = saturate_u8(v);   // :542
        if (P.cols_f32) P.cols_f32[(size_t)seg * n_cells + i] = v;
    }
    // the history of per-pass wave counts this batch leaves behind (GridHint, written by the k_scan launches before this
    // one) goes to the host's page-locked copy: 96 bytes, one workgroup
    if (blockIdx.x == 0 && P.hist_host && P.grid_hint && tid < kMaxPasses) P.hist_host[tid] = P.grid_hint->hist[tid];
}

// ---------------------------------------------------------------------------
// columns [n_angles][n_cells] -> image [n_cells][n_angles], col = (scroll + a) % n_angles
// grid (ceil(n_cells/64), ceil(n_angles/64)), block 256 (64x4)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_assemble(const T* __restrict__ cols, T* __restrict__ img,
                                                  int n_angles, int n_cells, int scroll,
                                                  int n_loc, size_t block_stride, size_t frame_stride)
{
    // blockIdx.z = frame of a multi-frame step: its columns start frame_stride elements further,
    // its image is the next [n_cells][n_angles] slab
    cols += (size_t)blockIdx.z * frame_stride;
    img += (size_t)blockIdx.z * (size_t)n_cells * n_angles;
    // azimuth a lives at cols + (a / n_loc) * block_stride + (a % n_loc) * n_cells
    // (contiguous [n_angles][n_cells] when block_stride == n_loc * n_cells)
    __shared__ T tile[64][65];
    const int c0 = blockIdx.x * 64, a0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int a = a0 + r, c = c0 + tx;
        if (a < n_angles && c < n_cells) tile[r][tx] = cols[(size_t)(a / n_loc) * block_stride + (size_t)(a % n_loc) * n_cells + c];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, a = a0 + tx;
        if (a < n_angles && c < n_cells) {
            const int col = (scroll + a) % n_angles;
            img[(size_t)c * n_angles + col] = tile[tx][r];
        }
    }
}

// mono8 fast path: four bins per load, four azimuths per store.  Needs n_cells % 4 == 0,
// n_angles % 4 == 0, scroll % 4 == 0 (a group of four azimuths then never straddles the
// wrap), n_loc % 4 == 0 is NOT needed (rows are addressed one by one).  Divisions by n_loc
// are multiplications by `magic` = floor(2^32 / n_loc) + 1, exact for a < 65536.
// grid (ceil(n_cells/64), ceil(n_angles/64), n_frames), block 256
__global__ __launch_bounds__(256) void k_assemble_u8x4(const uint8_t* __restrict__ cols, uint8_t* __restrict__ img,
                                                       int n_angles, int n_cells, int scroll,
                                                       int n_loc, uint32_t magic, size_t block_stride, size_t frame_stride)
{
    cols += (size_t)blockIdx.z * frame_stride;
    img += (size_t)blockIdx.z * (size_t)n_cells * n_angles;
    __shared__ uint32_t tile[64][17];              // [azimuth][16 words of 4 bins] + 1 pad
    const int c0 = blockIdx.x * 64, a0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    {
        const int w = t & 15, c = c0 + 4 * w;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = (t >> 4) + 16 * k, a = a0 + r;
            uint32_t v = 0;
            if (a < n_angles && c < n_cells) {
                const uint32_t blk = __umulhi((uint32_t)a, magic);
                const uint32_t loc = (uint32_t)a - blk * (uint32_t)n_loc;
                v = *reinterpret_cast<const uint32_t*>(cols + (size_t)blk * block_stride + (size_t)loc * n_cells + c);
            }
            tile[r][w] = v;
        }
    }
    __syncthreads();
    {
        const int g = t & 15, a = a0 + 4 * g;      // group of four azimuths
        int col = scroll + a;
        if (col >= n_angles) col -= n_angles;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = (t >> 4) + 16 * k, c = c0 + r;   // bin row of the tile
            if (a < n_angles && c < n_cells) {
                const int w = r >> 2, sh = 8 * (r & 3);
                const uint32_t b0 = (tile[4 * g + 0][w] >> sh) & 0xFFu, b1 = (tile[4 * g + 1][w] >> sh) & 0xFFu;
                const uint32_t b2 = (tile[4 * g + 2][w] >> sh) & 0xFFu, b3 = (tile[4 * g + 3][w] >> sh) & 0xFFu;
                *reinterpret_cast<uint32_t*>(img + (size_t)c * n_angles + col) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// objective of the parameter optimisation (scripts/radaray_opti.py:170-211): PSNR of a simulated mono8 image against
// ONE real image.  The device part is the exact integer sum of squared differences per image; the host finishes
// 10 log10(255^2 / (sse / n)) in f64 like skimage.metrics.peak_signal_noise_ratio.
// grid (blocks, n_images), block 256; sse[img] must be zero before the launch
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_score(const uint8_t* __restrict__ imgs, const uint8_t* __restrict__ ref, size_t npx,
                                               unsigned long long* sse)
{
    const uint8_t* img = imgs + (size_t)blockIdx.y * npx;
    unsigned long long acc = 0;
    const size_t n16 = npx / 16, stride = (size_t)gridDim.x * blockDim.x;
    const bool aligned = ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(ref)) & 15u) == 0;
    size_t done = 0;
    if (aligned) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const uint4 a = reinterpret_cast<const uint4*>(img)[i], b = reinterpret_cast<const uint4*>(ref)[i];
            const uint32_t aw[4] = { a.x, a.y, a.z, a.w }, bw[4] = { b.x, b.y, b.z, b.w };
            uint32_t part = 0;                       // 16 x 255^2 fits easily
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = (int)((aw[w] >> (8 * k)) & 0xFFu) - (int)((bw[w] >> (8 * k)) & 0xFFu);
                    part += (uint32_t)(d * d);
                }
            acc += part;
        }
        done = n16 * 16;
    }
    for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const int d = (int)img[i] - (int)ref[i];
        acc += (unsigned long long)(d * d);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ unsigned long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sse[blockIdx.y], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
// ---------------------------------------------------------------------------
// Host delivery without the runtime's copy engines: device memory -> page-locked host memory, stored by a kernel.
// Which engine a hipMemcpyAsync to page-locked memory runs on is the HIP runtime's choice -- the runtime bundled with the
// torch wheel (ROCm 7.0.2) launches a blit kernel (`__amd_rocclr_copyBuffer`) that competes with the frame kernels and reads
// 27-36k images/s on config 2, the image's own runtime uses SDMA (39.4k = the link) -- so a caller's process decided how fast
// the library delivers.  This kernel is the library's own: one-wave workgroups, 16 B per lane (1 KB per wave and store),
// at most `inflight` stores per wave outstanding (0: no limit).  The limit is what keeps the stores of OTHER kernels from
// queueing behind PCIe-paced writes in the memory pipeline (the 7 % of DESIGN.md §5); a flush at the end of a run, with
// nothing else on the chip, takes none.
// XCD confinement (xcd >= 0): the hardware deals the workgroups of a launch out to the 8 XCDs round robin in flat order, and
// each XCD has its own L2 and its own path into the fabric; PCIe-paced stores fill the write queues of the XCD they come
// from, and every other kernel's stores on THAT XCD wait behind them.  A launch of 8 x blocks workgroups of which only those
// with blockIdx % 8 == xcd work keeps the damage to one eighth of the chip.
// grid `blocks` (8 x blocks when confined), block 64
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_copy_host(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int inflight, int xcd)
{
    unsigned b = blockIdx.x, nb = gridDim.x;
    if (xcd >= 0) {
        if ((int)(b & 7u) != xcd) return;
        b >>= 3; nb >>= 3;
    }
    // (more loads in flight per lane -- the loop unrolled 2 / 4 / 8 times -- were measured and change nothing: 24-25k images/s
    // on config 2 in every shape; the runtime's own blit kernel is this very loop)
    const size_t nthreads = (size_t)nb * blockDim.x;
    int k = 0;
    for (size_t i = (size_t)b * blockDim.x + threadIdx.x; i < n16; i += nthreads) {
        const uint4 v = src[i];
        dst[i] = v;
        if (inflight > 0 && ++k >= inflight) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k = 0; }
    }
}
void launch_copy_host(const void* src, void* dst, size_t bytes, int blocks, int inflight, int xcd, hipStream_t s, int threads)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0) return;
    threads = std::max(64, std::min(1024, threads)) & ~63;
    unsigned g = (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)blocks, (n16 + threads - 1) / threads));
    if (xcd >= 0) g *= 8u;
    hipLaunchKernelGGL(k_copy_host, dim3(g), dim3(threads), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16, inflight, xcd);
}

// rr_peek_error_bits_async: one word into a page-locked host word, by a kernel's store (no copy engine involved)
__global__ void k_store_u32(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) { *dst = *src; }
void launch_store_u32(const uint32_t* src, uint32_t* h_dst, hipStream_t s) { hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, s, src, h_dst); }

// word-wise copy of a small table (either side may be page-locked host memory): the set-up uploads of rr_set_params
__global__ __launch_bounds__(256) void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_words(const void* src, void* dst, size_t bytes, hipStream_t s)
{
    const size_t n = bytes / 4;
    if (!n) return;
    const int g = (int)std::min<size_t>(256, (n + 255) / 256);
    hipLaunchKernelGGL(k_copy_words, dim3(g), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), n);
}

void launch_score(const uint8_t* imgs, const uint8_t* ref, size_t npx, int n_images, unsigned long long* sse, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(64, (npx / 16 + 255) / 256));
    hipLaunchKernelGGL(k_score, dim3(blocks, (unsigned)n_images), dim3(256), 0, s, imgs, ref, npx, sse);
}

// ---------------------------------------------------------------------------
// launchers (called from rr_api.cpp through plain C++ prototypes)
// ---------------------------------------------------------------------------
// once per tree upload: builder references (node index / first triangle) -> float4 offsets from the
// base of the tree allocation (rr_bvh.h)
__global__ void k_encode_refs(Node4* nodes, size_t n_children, uint32_t tri_base4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_children) return;
    uint32_t& ref = nodes[i >> 2].c[i & 3].ref;
    const uint32_t r = ref;
    if (r == kEmptyRef) return;
    if (r & kLeafFlag) ref = (r & 0xF0000000u) | (tri_base4 + 3u * (r & 0x0FFFFFFFu));
    else ref = r * 8u;
}
// ... and the grazing threshold of every triangle record into its spare word: 2.5e-5 |e1 x e2|^2 (see traverse)
__global__ void k_tri_graze(TriRec* tris, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 e1 = { tris[i].e1[0], tris[i].e1[1], tris[i].e1[2] }, e2 = { tris[i].e2[0], tris[i].e2[1], tris[i].e2[2] };
    const V3 c = v_cross(e1, e2);
    tris[i].pad = __float_as_uint(2.5e-5f * v_dot(c, c));
}
// ... and every internal node's way up, for the stack-free walk (traverse_stackless): child record 0's spare word of node c =
// (offset of c's parent) | (c's slot in it); the root keeps 0xFFFFFFFF.  Runs behind k_encode_refs (references are offsets by then)
__global__ void k_parent_links(Node4* nodes, size_t n_children)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) nodes[0].c[0].pad = 0xFFFFFFFFu;
    if (i >= n_children) return;
    const uint32_t r = nodes[i >> 2].c[i & 3].ref;
    if (r == kEmptyRef || (r & kLeafFlag)) return;
    nodes[r >> 3].c[0].pad = (uint32_t)((i >> 2) << 3) | (uint32_t)(i & 3);
}
void launch_encode_refs(Node4* nodes, size_t n_nodes, uint32_t tri_base4, hipStream_t s, size_t n_tris)
{
    const size_t n = n_nodes * 4;
    if (n == 0) return;
    hipLaunchKernelGGL(k_encode_refs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nodes, n, tri_base4);
    hipLaunchKernelGGL(k_parent_links, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nodes, n);
    if (n_tris) hipLaunchKernelGGL(k_tri_graze, dim3((unsigned)((n_tris + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<TriRec*>(reinterpret_cast<float4*>(nodes) + tri_base4), n_tris);
}

void launch_trace(const Params& P, int pass, const PoseArgs* poses, bool stats, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop,
                  hipEvent_t ev_rep_start, hipEvent_t ev_rep_stop, bool* repair_launched)
{
    if (repair_launched) *repair_launched = false;
    const int n_seg = (pass == 0 && P.set_mode) ? P.n_groups * P.n_loc : P.n_seg;
    // pass 0: one flat sequence of n_seg x n_beam rays; later passes: a row of blocks per segment
    const int A0 = P.pass0_az, Sw0 = kRaysPerWave / A0;
    const size_t waves0 = (size_t)((n_seg + A0 - 1) / A0) * (size_t)((P.n_beam + Sw0 - 1) / Sw0);
    // later passes: a segment holds at most n_beam * 2^pass waves (each wave has <= 2 children) -- no blocks are
    // launched beyond that bound (pass 1 of the KAIST preset: 13 instead of 50 blocks per segment; the rest would
    // read their segment's count and exit)
    const long bound = std::min<long>((long)P.cap, pass < 20 ? (long)P.n_beam << pass : (long)P.cap);
    // ... and since round 5 the row is as long as earlier batches needed (GridHint): tight_groups[pass], 0 = the bound
    const unsigned row = (pass > 0 && pass < kMaxPasses && P.tight_groups[pass]) ? (unsigned)P.tight_groups[pass]
                                                                               : (unsigned)((bound + kRaysPerBlock - 1) / kRaysPerBlock);
    // (an odd row length deals the rows' workgroups evenly over the 8 XCDs, see run_frame; the extra workgroup of a full row
    // exits on its segment's count.  Not on the spill path, whose columns are laid out for the bound)
    const bool tightened = pass > 0 && pass < kMaxPasses && P.tight_groups[pass];     // (the host keeps tightened rows odd itself)
    const unsigned row_odd = (pass > 0 && P.spill_depth == 0 && !tightened) ? (row | 1u) : row;
    dim3 grid = pass == 0 ? dim3((unsigned)((waves0 + (kTraceThreads / 64) - 1) / (kTraceThreads / 64)))
                          : dim3(row_odd, n_seg + (P.copy_blocks > 0 ? 1 : 0));
    Params Pl = P;
    if (pass > 0 && P.seg_chunk > 0) {
        const unsigned S = (unsigned)P.seg_chunk, n_chunks = ((unsigned)n_seg + S - 1) / S;
        if ((size_t)n_chunks * row + 1 <= 65535) { grid = dim3(S, n_chunks * row + (P.copy_blocks > 0 ? 1 : 0)); Pl.trace_row = (int)row; }
        else Pl.seg_chunk = 0;        // (beyond the grid's y limit: the plain layout)
    }
    if (pass == 0) Pl.copy_blocks = 0;
    else Pl.copy_blocks = std::min<int>(P.copy_blocks, (int)grid.x);      // the copy's workgroups are the first of row 0
    dim3 block(kTraceThreads);
    // later passes cull stack entries at pop time (6-B entries) as long as 32 one-wave workgroups still fit a CU's 160 KB of LDS
    // (5 KB each: up to 53 entries); a deeper tree keeps the 4-B entries -- the lost occupancy would cost more than the
    // cull returns (GPU-built tree of the 10M-triangle target, 56 entries: 0.465 vs 0.442 ms per frame)
    const bool cull = kCullPop && P.cull_pop && pass > 0 && (size_t)P.stack_lds * kRaysPerBlock * 6 <= 10240 / (128 / kTraceThreads);
    const size_t lds = (size_t)P.stack_lds * kRaysPerBlock * (cull ? 6 : 4);
    const bool spill = P.spill_depth > 0;
// hipExtLaunchKernelGGL: the optional events take the dispatch's own begin/end timestamps (what
    // rocprofv3 reports), not the time the launch spent waiting for CUs held by other streams
#define RR_LAUNCH_TRACE0(S, X) hipExtLaunchKernelGGL((k_trace<true, S, X, false>), grid, block, lds, s, ev_start, ev_stop, 0, Pl, pass, *poses)
#define RR_LAUNCH_TRACE(F, S, X, C) hipExtLaunchKernelGGL((k_trace<F, S, X, C>), grid, block, lds, s, ev_start, ev_stop, 0, Pl, pass, NoPoses{})
    if (P.stackless && !stats) {       // the stack-free walk: no LDS at all (RR_STACKLESS=1; the statistics build keeps the stack walk)
        if (pass == 0) hipExtLaunchKernelGGL((k_trace<true, false, false, false, true>), grid, block, 0, s, ev_start, ev_stop, 0, Pl, pass, *poses);
        else hipExtLaunchKernelGGL((k_trace<false, false, false, false, true>), grid, block, 0, s, ev_start, ev_stop, 0, Pl, pass, NoPoses{});
    } else if (pass == 0) {
        if (stats) { if (spill) RR_LAUNCH_TRACE0(true, true); else RR_LAUNCH_TRACE0(true, false); }
        else       { if (spill) RR_LAUNCH_TRACE0(false, true); else RR_LAUNCH_TRACE0(false, false); }
    } else if (cull) {
        if (stats) { if (spill) RR_LAUNCH_TRACE(false, true, true, true); else RR_LAUNCH_TRACE(false, true, false, true); }
        else       { if (spill) RR_LAUNCH_TRACE(false, false, true, true); else RR_LAUNCH_TRACE(false, false, false, true); }
    } else {
        if (stats) { if (spill) RR_LAUNCH_TRACE(false, true, true, false); else RR_LAUNCH_TRACE(false, true, false, false); }
        else       { if (spill) RR_LAUNCH_TRACE(false, false, true, false); else RR_LAUNCH_TRACE(false, false, false, false); }
    }
#undef RR_LAUNCH_TRACE
#undef RR_LAUNCH_TRACE0
    if (pass > 0 && pass < kMaxPasses && P.tight_groups[pass]) {     // host guarantees: no statistics build, no spill path
        const dim3 rgrid(128);
        // (its own pair of events in timing mode: the repair is not inside the trace launch's begin / end)
        if (!ev_rep_start) {
            if (cull) hipLaunchKernelGGL((k_trace_repair<true>), rgrid, block, lds, s, Pl, pass);
            else      hipLaunchKernelGGL((k_trace_repair<false>), rgrid, block, lds, s, Pl, pass);
        } else if (cull) hipExtLaunchKernelGGL((k_trace_repair<true>), rgrid, block, lds, s, ev_rep_start, ev_rep_stop, 0, Pl, pass);
        else             hipExtLaunchKernelGGL((k_trace_repair<false>), rgrid, block, lds, s, ev_rep_start, ev_rep_stop, 0, Pl, pass);
        if (repair_launched) *repair_launched = true;
    }
}

// the pass-0 trace kernel of the plain build (no statistics): the node of a replayed launch graph whose parameters change from
// replay to replay -- (Params, pass, PoseArgs) -- see rr_api.hip: run_frame
void* trace0_kernel(bool spill, bool stackless)
{
    if (stackless) return (void*)k_trace<true, false, false, false, true>;
    return spill ? (void*)k_trace<true, false, true, false> : (void*)k_trace<true, false, false, false>;
}
// ... and the Params bytes launch_trace hands that kernel
Params trace0_params(const Params& P) { Params Pl = P; Pl.copy_blocks = 0; return Pl; }

// (ev_start / ev_stop, timing mode: the dispatch's own begin / end timestamps -- what rocprofv3 reports as the kernel's
// duration -- not the time the launch spent waiting for the kernels of other streams)
void launch_shade(const Params& P, int pass, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const int cap_p = (int)std::min<long>((long)P.cap, pass < 20 ? (long)P.n_beam << pass : (long)P.cap);   // see launch_trace
    dim3 grid((cap_p + 63) / 64, P.n_seg), block(64);
    if (!ev_start) {
        if (pass == 0) hipLaunchKernelGGL((k_shade<true>), grid, block, 0, s, P, pass);
        else           hipLaunchKernelGGL((k_shade<false>), grid, block, 0, s, P, pass);
    } else if (pass == 0) hipExtLaunchKernelGGL((k_shade<true>), grid, block, 0, s, ev_start, ev_stop, 0, P, pass);
    else                  hipExtLaunchKernelGGL((k_shade<false>), grid, block, 0, s, ev_start, ev_stop, 0, P, pass);
}

void launch_scan(const Params& P, int pass, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    dim3 grid(P.n_seg), block(256);
    if (!ev_start) {
        if (pass == 0) hipLaunchKernelGGL((k_scan<true>), grid, block, 0, s, P, pass);
        else           hipLaunchKernelGGL((k_scan<false>), grid, block, 0, s, P, pass);
    } else if (pass == 0) hipExtLaunchKernelGGL((k_scan<true>), grid, block, 0, s, ev_start, ev_stop, 0, P, pass);
    else                  hipExtLaunchKernelGGL((k_scan<false>), grid, block, 0, s, ev_start, ev_stop, 0, P, pass);
}

void launch_column(const Params& P, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    dim3 grid(P.n_seg);
    const size_t lds = (((size_t)P.n_cells * sizeof(float)) + 15) & ~(size_t)15;
    if (!ev_start) {
        if (P.n_seg >= 1024) hipLaunchKernelGGL(k_column<256>, grid, dim3(256), lds, s, P);
        else hipLaunchKernelGGL(k_column<512>, grid, dim3(512), lds, s, P);
    } else if (P.n_seg >= 1024) hipExtLaunchKernelGGL(k_column<256>, grid, dim3(256), lds, s, ev_start, ev_stop, 0, P);
    else hipExtLaunchKernelGGL(k_column<512>, grid, dim3(512), lds, s, ev_start, ev_stop, 0, P);
}

void launch_assemble_u8(const uint8_t* cols, uint8_t* img, int n_angles, int n_cells, int scroll, hipStream_t s,
                        int n_loc, size_t block_stride, int n_frames, size_t frame_stride)
{
    dim3 grid((n_cells + 63) / 64, (n_angles + 63) / 64, n_frames > 0 ? n_frames : 1), block(256);
    if (n_loc <= 0) { n_loc = n_angles; block_stride = (size_t)n_angles * n_cells; }
    const int sc = ((scroll % n_angles) + n_angles) % n_angles;
    const bool aligned = ((reinterpret_cast<uintptr_t>(cols) | reinterpret_cast<uintptr_t>(img) | block_stride | frame_stride) & 3u) == 0;
    if (aligned && n_cells % 4 == 0 && n_angles % 4 == 0 && sc % 4 == 0 && n_angles < 65536 && n_loc > 1) {
        const uint32_t magic = (uint32_t)((1ull << 32) / (uint32_t)n_loc) + 1u;
        hipLaunchKernelGGL(k_assemble_u8x4, grid, block, 0, s, cols, img, n_angles, n_cells, sc, n_loc, magic,
                           block_stride, frame_stride);
        return;
    }
    hipLaunchKernelGGL((k_assemble<uint8_t>), grid, block, 0, s, cols, img, n_angles, n_cells, scroll, n_loc, block_stride,
                       frame_stride);
}

void launch_assemble_f32(const float* cols, float* img, int n_angles, int n_cells, int scroll, hipStream_t s)
{
    dim3 grid((n_cells + 63) / 64, (n_angles + 63) / 64), block(256);
    hipLaunchKernelGGL((k_assemble<float>), grid, block, 0, s, cols, img, n_angles, n_cells, scroll, n_angles, (size_t)n_angles * n_cells, (size_t)0);
}

void launch_debug_trace(const Params& P, const float* origs, const float* dirs, int n,
                        float* out_t, uint32_t* out_face, hipStream_t s, unsigned long long* steps)
{
    dim3 grid((n + kRaysPerBlock - 1) / kRaysPerBlock), block(kTraceThreads);
    const bool cull = kCullPop && P.cull_pop;
    const size_t lds = (size_t)P.stack_lds * kRaysPerBlock * (cull ? 6 : 4);
    if (steps) {
        if (cull) hipLaunchKernelGGL((k_debug_trace<true, true>), grid, block, lds, s, P, origs, dirs, n, out_t, out_face, steps);
        else hipLaunchKernelGGL((k_debug_trace<false, true>), grid, block, lds, s, P, origs, dirs, n, out_t, out_face, steps);
    } else {
        if (cull) hipLaunchKernelGGL((k_debug_trace<true, false>), grid, block, lds, s, P, origs, dirs, n, out_t, out_face, steps);
        else hipLaunchKernelGGL((k_debug_trace<false, false>), grid, block, lds, s, P, origs, dirs, n, out_t, out_face, steps);
    }
}

}  // namespace rr
