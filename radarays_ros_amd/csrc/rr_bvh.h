// rr_bvh.h -- host-side builder of the 4-wide BVH the HIP traversal kernel walks.
//
// Replaces rm::import_embree_map / Embree's BVH (reference
// src/radar_simulator.cpp:149; rmagine+Embree are not part of the reference
// tree).  Layout is chosen for MI355X: one node = 128 B = one L2 cache line,
// child boxes stored SoA so a lane fetches it with 8 dwordx4 loads; triangles
// are 48 B (3 x float4) in leaf order so a leaf is one contiguous run.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rr {

// child reference encoding AS BUILT (host and GPU builders):
//   inner : node index (< 0x80000000)
//   leaf  : 0x80000000 | (count-1) << 28 | first_triangle   (count 1..8, first < 2^28)
//   empty : box = degenerate point at kEmptyCoord (never hit), ref = kEmptyRef
// ON THE DEVICE the low 28 bits are re-encoded once per upload (k_encode_refs) as the float4 offset of the
// child from the base of the ONE allocation that holds the nodes and then the triangles:
//   inner : node index * 8                         leaf : tri_base4 + first_triangle * 3   (flag and count kept)
// so a traversal step forms `base + offset + lane part` the same way whatever the child is.
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kEmptyRef = 0x7FFFFFFFu;
constexpr uint32_t kMaxLeafTris = 4;
constexpr float kEmptyCoord = 3.0e38f;

// One RAY is traced by a QUAD of lanes; lane q owns child q.  A node is four 32-byte
// child records, so lane q fetches everything it needs (box + child reference) with
// two 16-byte loads from one address -- the same shape as a triangle fetch, which
// lets the kernel issue ONE batch of loads per traversal step whatever the step is.
struct alignas(16) Child4 {
    float lo[3];
    float hi[3];
    uint32_t ref;       // see encoding above
    uint32_t pad;
};
struct alignas(16) Node4 {
    Child4 c[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-byte line");

struct alignas(16) TriRec {
    float v0[3]; uint32_t face;    // original face index (tie-break + normal lookup)
    float e1[3]; uint32_t object;  // rmagine object/geometry id -> object_materials[]
    float e2[3]; uint32_t pad;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

struct Bvh4 {
    std::vector<Node4> nodes;   // nodes[0] is the root
    std::vector<TriRec> tris;   // leaf order; a face cut by spatial splits has one record per leaf that holds a part of it
    uint32_t depth = 0;         // node levels on the longest root->leaf path
    uint32_t stack_need = 0;    // upper bound of traversal stack entries
    float inflate = 0.f;        // outward padding applied to every box
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};
    double sah_cost = 0.0;
    double build_seconds = 0.0;
    uint64_t spatial_splits = 0;   // nodes split by a plane instead of by object partition
};

struct BvhOptions {
    // spatial splits (SBVH) are evaluated where the children of the best object split overlap by more than
    // sbvh_alpha x the root's surface area; < 0 switches them off (plain binned SAH)
    float sbvh_alpha = 1e-5f;
    // ... and by more than sbvh_beta x the node's own surface area (see rr_bvh.cpp)
    float sbvh_beta = 0.05f;
    // weight of the horizontal (xy) face of a box in the SAH's area: 1 = isotropic rays (the textbook SAH); < 1 says the
    // rays are mostly horizontal, as a radar's are (map z up; +-5 degrees of beam, reflections off walls stay level):
    // a box is then hit in proportion to its vertical cross-sections, and growing a node upwards is what costs.
    // 0.5 measured against 1.0 with tools/treeq (traversal steps per ray; wave iterations): 100k terrain over a floor
    // 12.3 vs 16.2; 26.4 vs 28.2 (a spatial split otherwise files a part of the floor into every terrain node and makes
    // it as tall as the floor is deep), 1M 19.1 vs 18.5; 29.6 vs 30.6, 10M 22.8 vs 23.2; 36.0 vs 37.5; flat between 0.4 and 0.6
    float vertical_weight = 0.5f;
    // spatial splits may add at most ref_budget x (number of faces) references (leaf triangle records)
    float ref_budget = 1.0f;
};

// Builds a binned-SAH binary BVH with spatial splits (multi-threaded), collapses it to 4-wide.
// Returns false and sets err on invalid input.
bool build_bvh4(const float* verts, size_t nv, const uint32_t* faces, size_t nf,
                const uint32_t* face_object, Bvh4& out, std::string& err, int n_threads = 0,
                const BvhOptions* options = nullptr);

}  // namespace rr
