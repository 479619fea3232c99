// Device-side scene layout (what crt_upload_scene builds in HBM).
//
// Everything a ray touches while it walks the scene is one of two record
// arrays, so a lane's whole traversal state is a 32-bit record index:
//
//  * box-pair records (64 B, one cache line, fetched as 4 x dwordx4 per lane):
//    two boxes (min / max interleaved per axis) and two 32-bit references.  Records [0, node_count) describe the
//    scene's node list: record i holds the world boxes of node i and node i+1,
//    so the linear node scan of the reference (renderer/trace.cu:51-68,
//    166-193) advances two nodes per fetch.  The records after that are the
//    interior nodes of every mesh BVH: the boxes of the left and the right
//    child, which the reference keeps in two different 72-byte nodes
//    (renderer/bvh.cuh:13-35) and therefore reads through two more lines.
//  * triangle records (48 B, 3 x dwordx4): the three vertices of one BVH
//    element, gathered in leaf order at upload, so a leaf's triangles are
//    consecutive and the element -> face -> vertex indirection of
//    MeshObject::IntersectTriangle (scene/objects.cu:273-276) is gone from the
//    loop.  The w lane of the first vertex carries the original face index.
#pragma once
#include <stdint.h>

#include "../crt_vec.h"

namespace crt {

struct alignas(16) BoxPair {
    float a[6];                 // box A as (min, max) pairs per axis: x, y, z - the layout the packed
    float b[6];                 // two-float vector instructions of the slab test read without a shuffle
    uint32_t ref_a, ref_b;
    uint32_t aux_a, aux_b;
};
static_assert(sizeof(BoxPair) == 64, "BoxPair must be one 64-byte line");

// BVH child reference.  Leaf: bit 31 set, bits 30..27 = triangle count (0..8),
// bits 26..0 = first triangle record.  Interior: absolute box-pair record index.
// host-side helper: fill one half of a pair from min / max triples
inline void set_box(float* half, const float* bmin, const float* bmax) {
    for (int k = 0; k < 3; k++) { half[2 * k] = bmin[k]; half[2 * k + 1] = bmax[k]; }
}

constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefNone = 0xffffffffu;     // only as "stack empty" marker
CRT_HD uint32_t make_leaf_ref(uint32_t start, uint32_t count) { return kRefLeaf | (count << 27) | start; }
CRT_HD bool ref_is_leaf(uint32_t r) { return (r & kRefLeaf) != 0; }
CRT_HD uint32_t leaf_start(uint32_t r) { return r & 0x07ffffffu; }
CRT_HD uint32_t leaf_count(uint32_t r) { return (r >> 27) & 0xfu; }
constexpr uint32_t kLeafMost = 13u;                       // triangles in a BVH leaf, at most; the two counts above it mark
constexpr uint32_t kTlasLeave = 14u, kTlasEnter = 15u;    // references of the BVH over scene nodes (DevScene::tlas_root)

// Scene-node half of a box pair: ref = object kind (2 bits) | (index of the
// first node after this node's subtree << 2); aux unused.
CRT_HD uint32_t node_ref_kind(uint32_t r) { return r & 3u; }
CRT_HD uint32_t node_ref_after(uint32_t r) { return r >> 2; }

struct alignas(16) TriRec {
    float p0[3]; uint32_t face;
    float p1[3]; uint32_t pad1;
    float p2[3]; uint32_t pad2;
};
static_assert(sizeof(TriRec) == 48, "TriRec is three 16-byte loads");

// Shading attributes of one triangle record (same index as its TriRec): the three vertex normals
// and the three texture vertices, gathered at upload so that a shaded hit needs one fetch instead
// of the mesh -> index triple -> vertex chain (MeshObject::IntersectTriangle, scene/objects.cu:308-320).
struct alignas(16) TriAttr {
    float n[9];
    float t[9];
    float pad[2];
};
static_assert(sizeof(TriAttr) == 80, "TriAttr is five 16-byte loads");

// Per scene node, read only when a ray enters the node.
struct alignas(16) NodeXf {
    float itm[12];      // world -> local
    float tm[12];       // local -> world
    uint32_t root_ref;  // mesh: BVH root reference
    int32_t mesh_id;
    int32_t material_id;
    uint32_t flags;     // mesh: kXfHasNormals | kXfHasTex; bits 2..3: the node's object kind
};
static_assert(sizeof(NodeXf) == 112, "NodeXf layout");
constexpr uint32_t kXfHasNormals = 1u, kXfHasTex = 2u, kXfKindShift = 2u;

struct DevTexture {
    uint32_t width, height;
    uint32_t offset;        // byte offset of texel (0,0) in the texel pool
    uint32_t first;         // texel (0,0) as r | g << 8 | b << 16: all there is to a 1x1 texture
};

struct DevMaterial {
    int32_t diffuse, specular, refraction, emission;
    float glossiness, ior;
    float absorption[3];
    float pad;
};

struct DevLight {
    int32_t kind;
    float intensity[3];
    float pos[3];
    float size;
    float radiance[3];      // intensity * float(1 / (pi * size^2)) for point lights
    float pad;
};

struct DevScene {
    // camera / render info
    uint32_t width, height;
    float cam_pos[3]; float dof;
    float pixel_size;
    float cX[3], cY[3];
    float top_left[3];
    // scene
    uint32_t node_count, light_count;
    int32_t env_texture;
    uint32_t stack_entries;          // LDS stack slots per lane (deeper entries overflow into the side buffer)
    uint32_t stack_overflow;         // entries per lane beyond those, kept in device memory behind the slot records
    uint32_t sched_box_min;          // box steps run unconditionally while this many lanes of a wave want one
    uint32_t sched_special_min;      // a triangle / transform turn runs once this many lanes wait for it
    uint32_t sched_exchange_min;     // an exchange turn (post results, take new queries) runs once this many lanes want one
    // block shape of the trace kernel: the first trav_waves waves of a block run ray queries, the others shade
    uint32_t trav_waves;
    uint32_t slots;                  // pixel slots per block (<= 1024)
    uint32_t serve_batch_min;        // a shading wave waits for this many slots ...
    uint32_t serve_low_water;        // ... while at least this many queries are ready for the traversal lanes ...
    uint32_t serve_patience;         // ... but for no more than this many polls
    uint32_t serve_priority;         // shading waves run at raised instruction-issue priority
    uint32_t first_step;             // the shading wave that sets a query up also takes the first step of its node scan while at least this many queries wait for traversal lanes
    uint32_t pair_live;              // ... or while fewer than this many of the block's slots still hold a pixel
    // regrouping kernel: query contexts of a block, and how a traversal wave sticks to the operation it runs
    uint32_t contexts;               // query contexts (stack column + parked state) per block, 0: the kernel without regrouping
    uint32_t keep_box, keep_leaf, keep_node, keep_exchange;   // a wave stays with its operation, without a vote, while it can fill this many lanes for it
    uint32_t exchange_first;         // an exchange (results out, new queries in) goes before any other operation once it can fill this many lanes
    uint32_t first_step_idle;        // ... or (queries started straight in contexts) while fewer than this many of the contexts within the lane's reach are free
    uint32_t direct_start;           // shading lanes put a new query straight into a free query context when one is within reach
    uint32_t swing_waves, swing_fill;   // this many of the traversal waves take a shading turn when at least swing_fill slots wait for one
    uint32_t burst_max;              // turns of one operation between two votes, at most
    uint32_t tail_live;              // once the pixel queue is nearly empty (DevPartition::tail_window), a block keeps only this many of its slots supplied
    uint32_t pair_below;             // a light sample's shadow query and the next segment's query start together while fewer than this many queries wait for traversal lanes
    uint32_t fault_inject;           // tests: block 0 drops one ray query on the floor (the watchdog then ends the launch with an error)
    unsigned long long watchdog_ticks;   // 100 MHz ticks a wave may find nothing to do before it reports a protocol error
    // Candidate mask of the scene-node scan (see scan_candidates in kernels.hip).  drift_rel / drift_abs bound how
    // far a ray can move over one whole query through the ToLocal / FromLocal round trips of the nodes it enters:
    // |pos' - pos| <= drift_rel * max|pos| + drift_abs, |dir' - dir| <= drift_rel * max|dir| (per component).
    // Scenes of more than 128 nodes: bit g of the mask stands for the group of 2^group_shift consecutive nodes
    // starting at node g << group_shift (group_pairs: the boxes of the groups, laid out like the node records).
    uint32_t scan_mask;              // bit 0: a candidate mask is in use (small drift); bit 1: the counting build follows it too (profiling); bit 2: its bits are groups of nodes
    uint32_t group_shift, group_count;
    float drift_rel, drift_abs;
    // Opt-in mode that does not reproduce the reference's node order: a BVH over the world boxes of the object nodes
    // takes the place of the scan.  Its interior records lie among the box pairs; a child reference with leaf count
    // kTlasEnter names a scene node to enter (in the start field), one with kTlasLeave on the stack marks where a
    // query that walks a mesh returns to world space.
    uint32_t tlas_root;
    const BoxPair* group_pairs;
    const BoxPair* pairs;
    const TriRec* tris;
    const NodeXf* xf;
    const TriAttr* attrs;
    const DevMaterial* materials;
    const DevTexture* textures;
    const uint8_t* texels;
    const DevLight* lights;
    // accumulators (W*H float3 each)
    float* results;
    float* normals;
    float* albedos;
};

// Work partition: the image is cut into strips of strip_rows rows; this context
// owns strips part, part + parts, ...  Inside a strip pixels are queued in 8x8
// tiles so that the 64 lanes of a wave start on one tile.
struct DevPartition {
    uint32_t part, parts, strip_rows;
    uint32_t tiles_x;            // ceil(width / 8)
    uint32_t items_per_strip;    // tiles_x * (strip_rows / 8) * 64
    uint32_t total_items;        // owned strips * items_per_strip
    // Tiles are handed out in the order of this table (costliest first, as measured by the previous
    // pass) when it is set, in storage order otherwise; every pass adds the ray queries of each
    // finished pixel to its tile's entry of tile_cost.  Pixels are independent, so the order of the
    // queue does not change any result.
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    uint32_t self_reset;         // the launch leaves the pixel queue's counter at zero for the next one (no memset between launches)
    uint32_t tail_window;        // queue items from the end of the queue within which blocks stop refilling all of their slots (0: never)
};

struct DevStats {
    unsigned long long closest_rays, shadow_rays, paths;
    unsigned long long node_box_tests, node_enters, bvh_interior, bvh_leaves;
    unsigned long long tri_tests, tri_hits, sphere_tests, plane_tests, texel_quads;
    unsigned long long sched_turns[12], sched_lanes[12], sched_cycles[12];
    unsigned long long path_cycles[8];
    unsigned long long scan_checks, scan_violations;   // counted build: box hits of the scan checked against the candidate mask
    unsigned long long errors;                         // waves that gave up waiting on the block's queues (must stay 0)
};

}  // namespace crt
