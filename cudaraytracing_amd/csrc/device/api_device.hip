// Device half of the C ABI: context, scene upload (BVH build + re-layout) and
// kernel launches.  Plays the part of Renderer::DoRendering
// (renderer/renderer.cu:16-73) with the pass loop left to the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../host/host.h"
#include "kernels.hip"

using namespace crt;

struct crt_ctx {
    int device = 0;
    std::string error;
    bool have_scene = false;
    DevScene S{};
    DevPartition P{};
    std::vector<void*> allocs;                 // everything owned by the current scene
    uint8_t* rgb8 = nullptr;
    uint32_t* queue = nullptr;
    uint32_t* cold = nullptr;                  // pixel-slot records (and stack overflow columns) of the trace kernel
    uint32_t* tile_cost = nullptr;             // ray queries per tile, measured by the pass in flight
    uint32_t* tile_order = nullptr;            // tiles sorted by the previous pass's cost
    uint32_t tile_capacity = 0;
    bool order_valid = false;
    bool deep = false;                         // BVH deeper than the LDS stack: kDeep kernels
    bool regroup = false;                      // the kernels whose waves regroup queries by operation (kRegroup)
    uint32_t mesh_nodes = 0;                   // scene nodes that hold a mesh
    int64_t tail_window_knob = -1;             // CRT_TAIL_WINDOW, or -1: the usual window for the partition (set_partition)
    bool want_tlas = false;                    // crt_set_tlas: asked for by the caller, takes effect at the next upload
    bool tlas = false;                         // the uploaded scene is walked through a BVH over its object nodes (kTlas)
    uint32_t stack_overflow = 0;               // stack entries per thread kept in the side buffer
    bool interleave = false;                   // tile order: expensive and cheap tiles alternate
    bool no_balance = false;                   // CRT_NO_BALANCE=1: always hand tiles out in storage order
    DevStats* stats = nullptr;
    std::map<int, uint32_t*> jumps;            // pass -> 160 x 5 jump matrix on the device
    static constexpr int kRing = 256;
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};   // event pair per launch, ring buffer
    uint64_t launches = 0;                          // launches recorded so far
    uint64_t reported = 0;                          // launches already handed out by crt_kernel_times
    unsigned char* scratch = nullptr;          // this device's copy of the static workspace (crt_render_pass_noalloc)
    int num_cus = 0;
    size_t lds_bytes = 0;
    int blocks_per_cu = 0, blocks_per_cu_counted = 0;
};

// Workspace of crt_render_pass_noalloc: the pixel-queue counter, one sequence-jump matrix and the per-lane path
// state of the trace kernel, in a fixed array of the code object instead of memory from hipMalloc.  Every GPU the
// library runs on holds its own copy at its own address until the process ends; host code never uses the array's
// name as an address, only the pointer hipGetSymbolAddress returned for the device in use (scratch_for_device).
constexpr size_t kScratchBytes = 32u << 20;
constexpr int kScratchPasses = 64;      // sequence-jump matrices of passes 1 .. kScratchPasses - 1 are kept in the workspace
constexpr size_t kJumpBytes = 160 * 5 * 4;
constexpr size_t kScratchQueueOff = 0, kScratchJumpOff = 256, kScratchStateOff = (kScratchJumpOff + kScratchPasses * kJumpBytes + 4095) / 4096 * 4096;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (ctx)->error = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return CRT_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

int fail(crt_ctx* ctx, int code, const std::string& msg) { ctx->error = msg; return code; }

// A trace kernel whose hand-off protocol failed has counted its waves that gave up in DevStats::errors and left
// the image incomplete: every call that hands results to the caller looks here first.  (Work queued on other
// streams than the default one is the caller's to wait for, as for the copy itself.)
int check_kernel_errors(crt_ctx* ctx) {
    unsigned long long errors = 0;
    HIP_TRY(ctx, hipMemcpy(&errors, &ctx->stats->errors, sizeof errors, hipMemcpyDeviceToHost));
    if (errors) return fail(ctx, CRT_E_HIP, "the trace kernel reported " + std::to_string(errors) + " queue protocol error(s): the image is incomplete");
    return CRT_OK;
}

// Device address of `scratch` on the current device, looked up once per device.  The first look-up on a device also
// fills the table of sequence-jump matrices (the same for every context and every scene), so that the call that
// allocates nothing copies nothing either.
hipError_t scratch_for_device(int device, unsigned char** out);

void free_scene(crt_ctx* ctx) {
    for (void* p : ctx->allocs) (void)hipFree(p);
    ctx->allocs.clear();
    for (auto& kv : ctx->jumps) (void)hipFree(kv.second);
    ctx->jumps.clear();
    ctx->rgb8 = nullptr;
    ctx->have_scene = false;
}

template <typename T>
int upload(crt_ctx* ctx, const std::vector<T>& host, const T** dev) {
    void* p = nullptr;
    const size_t bytes = host.size() * sizeof(T);
    HIP_TRY(ctx, hipMalloc(&p, bytes ? bytes : 16));
    ctx->allocs.push_back(p);
    if (bytes) HIP_TRY(ctx, hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return CRT_OK;
}

// ---- XORWOW sequence jump: the 160x160 GF(2) matrix of "advance the xorshift part by
// pass * 2^67 steps" (what curand_init does with its subsequence argument).
struct BitMat { uint32_t row[160][5]; };

void mat_apply(const BitMat& m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; w++)
        for (int b = 0; b < 32; b++)
            if (in[w] & (1u << b))
                for (int k = 0; k < 5; k++) r[k] ^= m.row[w * 32 + b][k];
    for (int k = 0; k < 5; k++) out[k] = r[k];
}

// c = "a then b"
void mat_mul(const BitMat& a, const BitMat& b, BitMat& c) {
    for (int r = 0; r < 160; r++) mat_apply(b, a.row[r], c.row[r]);
}

const BitMat& subsequence_step() {
    static BitMat base;
    static bool ready = false;
    if (!ready) {
        BitMat a, t;
        for (int r = 0; r < 160; r++) {
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[r / 32] = 1u << (r % 32);
            const uint32_t x = v[0] ^ (v[0] >> 2);
            uint32_t o[5] = {v[1], v[2], v[3], v[4], (v[4] ^ (v[4] << 4)) ^ (x ^ (x << 1))};
            memcpy(a.row[r], o, sizeof o);
        }
        for (int i = 0; i < 67; i++) { mat_mul(a, a, t); a = t; }
        base = a;
        ready = true;
    }
    return base;
}

void jump_for_pass(unsigned pass, BitMat& out) {
    BitMat result, sq = subsequence_step(), t;
    bool have = false;
    for (unsigned p = pass; p; p >>= 1) {
        if (p & 1u) {
            if (!have) { result = sq; have = true; }
            else { mat_mul(result, sq, t); result = t; }
        }
        if (p >> 1) { mat_mul(sq, sq, t); sq = t; }
    }
    out = result;
}

int get_jump(crt_ctx* ctx, int pass, const uint32_t** dev) {
    *dev = nullptr;
    if (pass == 0) return CRT_OK;
    auto it = ctx->jumps.find(pass);
    if (it == ctx->jumps.end()) {
        BitMat m;
        jump_for_pass((unsigned)pass, m);
        uint32_t* p = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&p, sizeof m.row));
        HIP_TRY(ctx, hipMemcpy(p, m.row, sizeof m.row, hipMemcpyHostToDevice));
        it = ctx->jumps.emplace(pass, p).first;
    }
    *dev = it->second;
    return CRT_OK;
}

hipError_t scratch_for_device(int device, unsigned char** out) {
    static std::mutex lock;
    static std::map<int, unsigned char*> known;
    std::lock_guard<std::mutex> guard(lock);
    auto it = known.find(device);
    if (it == known.end()) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        static std::vector<uint32_t> table;      // passes 1 .. kScratchPasses - 1, computed once per process
        if (table.empty()) {
            table.resize((size_t)kScratchPasses * 800, 0u);
            for (int pass = 1; pass < kScratchPasses; pass++) {
                BitMat m;
                jump_for_pass((unsigned)pass, m);
                memcpy(&table[(size_t)pass * 800], m.row, kJumpBytes);
            }
        }
        e = hipMemcpy((unsigned char*)p + kScratchJumpOff, table.data(), table.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        e = hipMemset((unsigned char*)p + kScratchQueueOff, 0, 256);      // the pixel queue's counter: every launch leaves it at zero again
        if (e != hipSuccess) return e;
        it = known.emplace(device, (unsigned char*)p).first;
    }
    *out = it->second;
    return hipSuccess;
}

int validate_desc(crt_ctx* ctx, const crt_scene_desc* d) {
    if (!d) return fail(ctx, CRT_E_INVALID, "null scene");
    if (d->camera.width == 0 || d->camera.height == 0) return fail(ctx, CRT_E_INVALID, "image size is zero");
    if ((uint64_t)d->camera.width * d->camera.height > 0x7fffffffull / 3) return fail(ctx, CRT_E_INVALID, "image too large");
    if (d->node_count == 0 || !d->nodes) return fail(ctx, CRT_E_INVALID, "scene has no nodes");
    if (d->material_count == 0 || !d->materials) return fail(ctx, CRT_E_INVALID, "scene has no materials");
    if (d->texture_count == 0 || !d->textures) return fail(ctx, CRT_E_INVALID, "scene has no textures");
    if (d->env_texture < 0 || (uint32_t)d->env_texture >= d->texture_count) return fail(ctx, CRT_E_INVALID, "environment texture index out of range");
    for (uint32_t i = 0; i < d->texture_count; i++) {
        const crt_texture& t = d->textures[i];
        if ((t.width == 0) != (t.height == 0)) return fail(ctx, CRT_E_INVALID, "texture with exactly one zero dimension");
        if ((uint64_t)t.width * t.height > 0 && !t.rgb) return fail(ctx, CRT_E_INVALID, "texture without texels");
    }
    for (uint32_t i = 0; i < d->material_count; i++) {
        const crt_material& m = d->materials[i];
        for (int id : {m.diffuse, m.specular, m.refraction, m.emission})
            if (id < 0 || (uint32_t)id >= d->texture_count) return fail(ctx, CRT_E_INVALID, "material texture index out of range");
    }
    if (d->light_count && !d->lights) return fail(ctx, CRT_E_INVALID, "light count without a light array");
    if (d->mesh_count && !d->meshes) return fail(ctx, CRT_E_INVALID, "mesh count without a mesh array");
    for (uint32_t i = 0; i < d->light_count; i++)
        if (d->lights[i].kind < CRT_LIGHT_AMBIENT || d->lights[i].kind > CRT_LIGHT_POINT) return fail(ctx, CRT_E_INVALID, "bad light kind");
    for (uint32_t i = 0; i < d->mesh_count; i++) {
        const crt_mesh& m = d->meshes[i];
        if ((m.nv && !m.v) || (m.nf && !m.f)) return fail(ctx, CRT_E_INVALID, "mesh without vertex / face arrays");
        if (m.nvn && (!m.vn || !m.fn)) return fail(ctx, CRT_E_INVALID, "mesh normals without index triples");
        if (m.nvt && (!m.vt || !m.ft)) return fail(ctx, CRT_E_INVALID, "mesh texture vertices without index triples");
        if (m.nf > 0x07ffffffu) return fail(ctx, CRT_E_LIMIT, "mesh has more than 2^27 faces");
        for (uint32_t k = 0; k < 3 * m.nf; k++) {
            if (m.f[k] >= m.nv) return fail(ctx, CRT_E_INVALID, "mesh vertex index out of range");
            if (m.nvn && m.fn[k] >= m.nvn) return fail(ctx, CRT_E_INVALID, "mesh normal index out of range");
            if (m.nvt && m.ft[k] >= m.nvt) return fail(ctx, CRT_E_INVALID, "mesh texture index out of range");
        }
    }
    // child counts must describe a depth-first forest that ends exactly at node_count: walking the list with a
    // stack of "children still to come" must never run past the end and must leave no child outstanding
    {
        std::vector<uint32_t> open;
        for (uint32_t i = 0; i < d->node_count; i++) {
            while (!open.empty() && open.back() == 0) open.pop_back();
            if (!open.empty()) open.back()--;
            if (d->nodes[i].child_count) open.push_back(d->nodes[i].child_count);
        }
        while (!open.empty() && open.back() == 0) open.pop_back();
        if (!open.empty()) return fail(ctx, CRT_E_INVALID, "node child counts run past the end of the node list");
    }
    for (uint32_t i = 0; i < d->node_count; i++) {
        const crt_node& n = d->nodes[i];
        if (n.object_kind < CRT_OBJ_NONE || n.object_kind > CRT_OBJ_MESH) return fail(ctx, CRT_E_INVALID, "bad object kind");
        if (n.object_kind == CRT_OBJ_MESH && (n.mesh_id < 0 || (uint32_t)n.mesh_id >= d->mesh_count)) return fail(ctx, CRT_E_INVALID, "node mesh index out of range");
        if (n.material_id < 0 || (uint32_t)n.material_id >= d->material_count) return fail(ctx, CRT_E_INVALID, "node material index out of range");
    }
    return CRT_OK;
}

// index of the first node after node i's subtree, from the depth-first child counts
void subtree_ends(const crt_scene_desc* d, std::vector<uint32_t>& after) {
    const uint32_t n = d->node_count;
    after.assign(n, n);
    // iterative: process nodes in reverse so children are known
    std::vector<uint32_t> size(n, 1);
    for (int i = (int)n - 1; i >= 0; i--) {
        uint32_t j = (uint32_t)i + 1, total = 1;
        for (uint32_t c = 0; c < d->nodes[i].child_count && j < n; c++) { total += size[j]; j += size[j]; }
        size[i] = total;
        after[i] = (uint32_t)i + total > n ? n : (uint32_t)i + total;
    }
}

// A BVH over the object nodes of the scene (no counterpart in the reference, which scans its node list from end to
// end for every ray: renderer/trace.cu:51, 166).  Each object's box is the box of its own geometry in world space -
// the unit sphere, the unit plane or the mesh's bounds carried through the node's matrix - not the node's box of the
// scene file, which also spans the node's children.  Binary tree by a sweep of the surface-area cost over the centres
// sorted along each axis; interior records are appended to `pairs`, a child that is an object is a reference with
// count kTlasEnter and the node's index.  root: the reference to start from (an empty leaf when the scene has no
// object); depth: interior levels above the deepest object.
void tlas_build(const crt_scene_desc* d, std::vector<BoxPair>& pairs, uint32_t& root, uint32_t& depth) {
    struct Item { float lo[3], hi[3]; uint32_t node; };
    std::vector<Item> items;
    for (uint32_t i = 0; i < d->node_count; i++) {
        const crt_node& n = d->nodes[i];
        double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
        if (n.object_kind == CRT_OBJ_NONE) continue;
        if (n.object_kind == CRT_OBJ_PLANE) lo[2] = hi[2] = 0;
        if (n.object_kind == CRT_OBJ_MESH) {
            const crt_mesh& M = d->meshes[n.mesh_id];
            if (M.nf == 0) continue;
            for (int k = 0; k < 3; k++) { lo[k] = M.box_min[k]; hi[k] = M.box_max[k]; }
        }
        double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
        bool finite = true;
        for (int corner = 0; corner < 8; corner++) {
            const double p[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
            for (int row = 0; row < 3; row++) {
                const double w = (double)n.tm[row] * p[0] + (double)n.tm[3 + row] * p[1] + (double)n.tm[6 + row] * p[2] + (double)n.tm[9 + row];
                if (!(w == w) || fabs(w) > 1e30) finite = false;
                wlo[row] = fmin(wlo[row], w); whi[row] = fmax(whi[row], w);
            }
        }
        Item it;
        it.node = i;
        // The object is tested in its own space with a ray that went through the node's inverse matrix in single
        // precision: the box is widened by far more than that rounding, so that no hit of the object lies outside it.
        double reach = 0;
        for (int k = 0; k < 3; k++) reach = fmax(reach, fmax(fabs(wlo[k]), fabs(whi[k])));
        const double pad = 1e-4 * reach + 1e-6;
        for (int k = 0; k < 3; k++) {
            it.lo[k] = finite ? (float)(wlo[k] - pad) : -CRT_BIGFLOAT;
            it.hi[k] = finite ? (float)(whi[k] + pad) : CRT_BIGFLOAT;
        }
        items.push_back(it);
    }
    depth = 0;
    root = make_leaf_ref(0, 0);
    if (items.empty()) return;
    auto grow = [](float* lo, float* hi, const Item& it) { for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], it.lo[k]); hi[k] = fmaxf(hi[k], it.hi[k]); } };
    auto half_area = [](const float* lo, const float* hi) -> double {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    };
    std::vector<double> right_area;
    // items [b, e) -> the reference of their subtree and its box
    std::function<uint32_t(uint32_t, uint32_t, uint32_t, float*, float*)> build = [&](uint32_t b, uint32_t e, uint32_t level, float* lo, float* hi) -> uint32_t {
        for (int k = 0; k < 3; k++) { lo[k] = CRT_BIGFLOAT; hi[k] = -CRT_BIGFLOAT; }
        for (uint32_t i = b; i < e; i++) grow(lo, hi, items[i]);
        if (e - b == 1) return make_leaf_ref(items[b].node, kTlasEnter);
        if (level + 1 > depth) depth = level + 1;
        int best_axis = 0;
        uint32_t best_cut = (b + e) / 2;
        double best_cost = 1e300;
        for (int axis = 0; axis < 3 && level < 48; axis++) {      // (beyond 48 levels: halve, whatever the cost says)
            std::sort(items.begin() + b, items.begin() + e, [axis](const Item& x, const Item& y) {
                const float cx = x.lo[axis] + x.hi[axis], cy = y.lo[axis] + y.hi[axis];
                return cx < cy || (cx == cy && x.node < y.node);
            });
            right_area.assign(e - b + 1, 0.0);
            float rl[3] = {CRT_BIGFLOAT, CRT_BIGFLOAT, CRT_BIGFLOAT}, rh[3] = {-CRT_BIGFLOAT, -CRT_BIGFLOAT, -CRT_BIGFLOAT};
            for (uint32_t i = e; i-- > b + 1;) { grow(rl, rh, items[i]); right_area[i - b] = half_area(rl, rh); }
            float ll[3] = {CRT_BIGFLOAT, CRT_BIGFLOAT, CRT_BIGFLOAT}, lh[3] = {-CRT_BIGFLOAT, -CRT_BIGFLOAT, -CRT_BIGFLOAT};
            for (uint32_t cut = b + 1; cut < e; cut++) {
                grow(ll, lh, items[cut - 1]);
                const double cost = half_area(ll, lh) * (cut - b) + right_area[cut - b] * (e - cut);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_cut = cut; }
            }
        }
        std::sort(items.begin() + b, items.begin() + e, [best_axis](const Item& x, const Item& y) {
            const float cx = x.lo[best_axis] + x.hi[best_axis], cy = y.lo[best_axis] + y.hi[best_axis];
            return cx < cy || (cx == cy && x.node < y.node);
        });
        const size_t rec = pairs.size();
        pairs.push_back(BoxPair{});
        float la[3], ha[3], lb[3], hb[3];
        const uint32_t ref_a = build(b, best_cut, level + 1, la, ha);
        const uint32_t ref_b = build(best_cut, e, level + 1, lb, hb);
        BoxPair& p = pairs[rec];
        set_box(p.a, la, ha); set_box(p.b, lb, hb);
        p.ref_a = ref_a; p.ref_b = ref_b; p.aux_a = p.aux_b = 0;
        return (uint32_t)rec;
    };
    float lo[3], hi[3];
    root = build(0, (uint32_t)items.size(), 0, lo, hi);
}

// queue items of the whole image cut into strips of strip_rows rows (padded to whole 8x8 tiles)
uint64_t partition_items(const DevScene& S, uint32_t strip_rows) {
    const uint64_t strips = (S.height + strip_rows - 1) / strip_rows;
    return strips * ((S.width + 7) / 8) * (strip_rows / 8) * 64;
}

void set_partition(crt_ctx* ctx, uint32_t part, uint32_t parts, uint32_t strip_rows) {
    DevPartition& P = ctx->P;
    P.part = part; P.parts = parts; P.strip_rows = strip_rows;
    P.tiles_x = (ctx->S.width + 7) / 8;
    P.items_per_strip = P.tiles_x * (strip_rows / 8) * 64;
    const uint32_t strips = (ctx->S.height + strip_rows - 1) / strip_rows;
    const uint32_t owned = strips > part ? (strips - part + parts - 1) / parts : 0;
    P.total_items = owned * P.items_per_strip;
    P.tile_order = nullptr;
    P.tile_cost = nullptr;
    P.self_reset = 0;
    {
        const uint64_t in_flight = (uint64_t)ctx->num_cus * ctx->S.slots, fifth = P.total_items / 5;
        const uint32_t usual = ctx->regroup && ctx->mesh_nodes > 0 ? (uint32_t)(in_flight < fifth ? in_flight : fifth) : 0u;
        P.tail_window = ctx->tail_window_knob >= 0 ? (uint32_t)ctx->tail_window_knob : usual;
    }
    ctx->order_valid = false;                  // costs of another partition say nothing about this one
}

// global memory one block of the trace kernel needs: its pixel-slot records and, for deep trees, the stack overflow
// columns of its traversal threads
size_t block_state_bytes(const crt_ctx* ctx) {
    const size_t columns = ctx->regroup ? ctx->S.contexts : ctx->S.trav_waves * 64;
    return (size_t)ctx->S.slots * kSlotWords * 4 + columns * 2 * ctx->stack_overflow * 4;
}

// blocks to launch: what stays resident, but no more than gives every traversal lane a pixel
uint32_t trace_grid(const crt_ctx* ctx, int per_cu) {
    uint32_t grid = (uint32_t)(ctx->num_cus * per_cu);
    const uint32_t lanes = ctx->S.trav_waves * 64;
    const uint32_t need = (ctx->P.total_items + lanes - 1) / lanes;
    return grid > need ? need : grid;
}

// The build of the trace kernel for this scene: counting or plain, primary hits only or whole paths, deep tree (stack
// overflow in the side buffer) or not, and one or two words of candidate mask.
using TraceKernel = void (*)(DevScene, DevPartition, int, const uint32_t*, uint32_t*, DevStats*, int32_t*, int32_t*, float*, uint32_t*);

template <bool kCount, bool kPrimary>
TraceKernel trace_variant(bool deep, bool wide, bool regroup, bool tlas) {
    if (tlas) return trace_kernel<kCount, kPrimary, false, false, true, true>;
    if (regroup) return wide ? trace_kernel<kCount, kPrimary, false, true, true> : trace_kernel<kCount, kPrimary, false, false, true>;
    if (deep) return wide ? trace_kernel<kCount, kPrimary, true, true, false> : trace_kernel<kCount, kPrimary, true, false, false>;
    return wide ? trace_kernel<kCount, kPrimary, false, true, false> : trace_kernel<kCount, kPrimary, false, false, false>;
}

TraceKernel trace_variant(const crt_ctx* ctx, bool counted, bool primary) {
    const bool wide = ctx->S.node_count > 64;      // two mask words: 65 to 128 nodes, or groups of nodes beyond that
    if (primary) return trace_variant<false, true>(ctx->deep, wide, ctx->regroup, ctx->tlas);
    return counted ? trace_variant<true, false>(ctx->deep, wide, ctx->regroup, ctx->tlas) : trace_variant<false, false>(ctx->deep, wide, ctx->regroup, ctx->tlas);
}

void launch_trace(crt_ctx* ctx, bool counted, bool primary, uint32_t grid, hipStream_t stream, const DevPartition& P, int samples, const uint32_t* jump,
                  uint32_t* queue, int32_t* out_node, int32_t* out_face, float* out_t, uint32_t* slot_mem) {
    hipLaunchKernelGGL(trace_variant(ctx, counted, primary), dim3(grid), dim3(CRT_BLOCK), ctx->lds_bytes, stream, ctx->S, P, samples, jump,
                       queue, ctx->stats, out_node, out_face, out_t, slot_mem);
}

}  // namespace

extern "C" {

int crt_create(int device, crt_ctx** out) {
    if (!out) return CRT_E_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { crt::set_host_error("no HIP device available"); return CRT_E_HIP; }
    if (device < 0 || device >= count) { crt::set_host_error("device index out of range"); return CRT_E_INVALID; }
    crt_ctx* ctx = new crt_ctx();
    ctx->device = device;
    // one exit for every failure: crt_destroy releases whatever exists by then
    auto give_up = [&](const char* msg) { crt_destroy(ctx); crt::set_host_error(msg); return CRT_E_HIP; };
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return give_up("cannot select HIP device");
    ctx->num_cus = prop.multiProcessorCount;
    if (scratch_for_device(device, &ctx->scratch) != hipSuccess) return give_up("cannot resolve the static workspace on this device");
    if (hipMalloc((void**)&ctx->queue, 256) != hipSuccess || hipMalloc((void**)&ctx->stats, sizeof(DevStats)) != hipSuccess ||
        hipMemset(ctx->stats, 0, sizeof(DevStats)) != hipSuccess)
        return give_up("cannot allocate context buffers");
    for (int i = 0; i < crt_ctx::kRing; i++)
        if (hipEventCreate(&ctx->ev0[i]) != hipSuccess || hipEventCreate(&ctx->ev1[i]) != hipSuccess) return give_up("cannot create timing events");
    *out = ctx;
    return CRT_OK;
}

void crt_destroy(crt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    if (ctx->queue) (void)hipFree(ctx->queue);
    if (ctx->stats) (void)hipFree(ctx->stats);
    for (int i = 0; i < crt_ctx::kRing; i++) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
    }
    delete ctx;
}

int crt_set_tlas(crt_ctx* ctx, int enable) {
    if (!ctx) return CRT_E_INVALID;
    ctx->want_tlas = enable != 0;
    return CRT_OK;
}

int crt_device_count(int* count) {
    if (!count) return CRT_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { crt::set_host_error("no HIP device available"); *count = 0; return CRT_E_HIP; }
    *count = n;
    return CRT_OK;
}

const char* crt_last_error(const crt_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int crt_upload_scene(crt_ctx* ctx, const crt_scene_desc* d) {
    if (!ctx) return CRT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = validate_desc(ctx, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    free_scene(ctx);

    const uint32_t N = d->node_count;
    std::vector<uint32_t> after;
    subtree_ends(d, after);

    // ---- BVH per mesh, then the global record arrays
    std::vector<HostBvh> bvhs(d->mesh_count);
    uint32_t max_depth = 1;
    for (uint32_t m = 0; m < d->mesh_count; m++) {
        const crt_mesh& M = d->meshes[m];
        std::string err;
        if (!bvh_build(M.v, M.nv, M.f, M.nf, M.box_min, M.box_max, bvhs[m], err))
            return fail(ctx, err.find("depth") != std::string::npos ? CRT_E_LIMIT : CRT_E_INVALID, "mesh " + std::to_string(m) + ": " + err);
        if (bvhs[m].depth > max_depth) max_depth = bvhs[m].depth;
    }

    std::vector<BoxPair> pairs(N);
    std::vector<TriRec> tris;
    std::vector<TriAttr> attrs;
    std::vector<uint32_t> mesh_root(d->mesh_count, 0);
    auto node_half = [&](uint32_t i, float* bmin, float* bmax, uint32_t& ref) {
        if (i < N) {
            memcpy(bmin, d->nodes[i].box_min, 12); memcpy(bmax, d->nodes[i].box_max, 12);
            ref = (uint32_t)d->nodes[i].object_kind | (after[i] << 2);
        } else {
            for (int k = 0; k < 3; k++) { bmin[k] = CRT_BIGFLOAT; bmax[k] = -CRT_BIGFLOAT; }
            ref = CRT_OBJ_NONE | (N << 2);
        }
    };
    for (uint32_t i = 0; i < N; i++) {
        BoxPair& p = pairs[i];
        float lo[3], hi[3];
        node_half(i, lo, hi, p.ref_a); set_box(p.a, lo, hi);
        node_half(i + 1, lo, hi, p.ref_b); set_box(p.b, lo, hi);
        p.aux_a = p.aux_b = 0;
    }
    for (uint32_t m = 0; m < d->mesh_count; m++) {
        const crt_mesh& M = d->meshes[m];
        const HostBvh& B = bvhs[m];
        const uint32_t tri_base = (uint32_t)tris.size();
        if ((uint64_t)tri_base + M.nf > 0x07ffffffull) return fail(ctx, CRT_E_LIMIT, "more than 2^27 triangles in the scene");
        for (uint32_t s = 0; s < M.nf; s++) {
            const uint32_t face = B.elements[s];
            TriRec t{};
            memcpy(t.p0, M.v + 3 * M.f[3 * face], 12);
            memcpy(t.p1, M.v + 3 * M.f[3 * face + 1], 12);
            memcpy(t.p2, M.v + 3 * M.f[3 * face + 2], 12);
            t.face = face;
            tris.push_back(t);
            TriAttr a{};
            if (M.nvn > 0) for (int k = 0; k < 3; k++) memcpy(a.n + 3 * k, M.vn + 3 * M.fn[3 * face + k], 12);
            if (M.nvt > 0) for (int k = 0; k < 3; k++) memcpy(a.t + 3 * k, M.vt + 3 * M.ft[3 * face + k], 12);
            attrs.push_back(a);
        }
        // interior nodes get consecutive records in node order
        const uint32_t nn = B.node_count();
        std::vector<uint32_t> rec_of(nn, 0);
        uint32_t next_rec = (uint32_t)pairs.size();
        for (uint32_t i = 0; i < nn; i++) if (!B.leaf[i]) rec_of[i] = next_rec++;
        auto ref_of = [&](uint32_t i) -> uint32_t {
            if (B.leaf[i]) return make_leaf_ref(tri_base + B.elem_start[i], B.elem_count[i]);
            return rec_of[i];
        };
        for (uint32_t i = 0; i < nn; i++) {
            if (B.leaf[i]) { if (B.elem_count[i] > kLeafMost) return fail(ctx, CRT_E_LIMIT, "BVH leaf with more than 13 elements"); continue; }
            BoxPair p{};
            const uint32_t l = B.left[i], r = B.right[i];
            set_box(p.a, &B.box[6 * l], &B.box[6 * l + 3]);
            set_box(p.b, &B.box[6 * r], &B.box[6 * r + 3]);
            p.ref_a = ref_of(l); p.ref_b = ref_of(r);
            pairs.push_back(p);
        }
        mesh_root[m] = nn ? ref_of(0) : make_leaf_ref(tri_base, 0);
    }
    // ---- opt-in: a BVH over the object nodes in place of the reference's node scans (trace.cu:51, 166)
    ctx->tlas = crt::knob_value(getenv("CRT_TLAS"), ctx->want_tlas ? 1 : 0, 0, 1) != 0;
    uint32_t tlas_root = make_leaf_ref(0, 0), tlas_depth = 0;
    const uint64_t tlas_first = pairs.size();      // (records from here on belong to the BVH over the nodes)
    if (ctx->tlas) tlas_build(d, pairs, tlas_root, tlas_depth);
    if (pairs.size() >= 0x80000000ull) return fail(ctx, CRT_E_LIMIT, "too many BVH nodes");
    // Every reference the kernels will follow, checked against the arrays before anything is uploaded: a kernel that
    // walks out of an array faults the GPU, so a slip of the builders above must end here, as an error.
    {
        const uint64_t n_pairs = pairs.size(), n_tris = tris.size();
        auto ref_ok = [&](uint32_t ref, bool top_level) -> bool {
            if (!ref_is_leaf(ref)) return ref >= N && ref < n_pairs;
            const uint32_t count = leaf_count(ref);
            if (count == kTlasEnter) return top_level && leaf_start(ref) < N;
            if (count == kTlasLeave) return false;      // (only ever on a stack)
            return count <= kLeafMost && (uint64_t)leaf_start(ref) + count <= n_tris;
        };
        bool ok = true;
        for (uint32_t i = 0; i < N && ok; i++) ok = node_ref_after(pairs[i].ref_a) <= N && node_ref_after(pairs[i].ref_b) <= N;
        for (uint64_t i = N; i < n_pairs && ok; i++) ok = ref_ok(pairs[i].ref_a, i >= tlas_first) && ref_ok(pairs[i].ref_b, i >= tlas_first);
        for (uint32_t m = 0; m < d->mesh_count && ok; m++) ok = ref_ok(mesh_root[m], false);
        ok = ok && ref_ok(tlas_root, true);
        if (!ok) return fail(ctx, CRT_E_INVALID, "internal error: a BVH record refers outside the scene's arrays");
    }

    std::vector<NodeXf> xf(N);
    for (uint32_t i = 0; i < N; i++) {
        const crt_node& n = d->nodes[i];
        memcpy(xf[i].itm, n.itm, 48); memcpy(xf[i].tm, n.tm, 48);
        xf[i].mesh_id = n.object_kind == CRT_OBJ_MESH ? n.mesh_id : -1;
        xf[i].root_ref = n.object_kind == CRT_OBJ_MESH ? mesh_root[n.mesh_id] : 0;
        xf[i].material_id = n.material_id;
        xf[i].flags = (uint32_t)n.object_kind << kXfKindShift;
        if (n.object_kind == CRT_OBJ_MESH) xf[i].flags |= (d->meshes[n.mesh_id].nvn > 0 ? kXfHasNormals : 0u) | (d->meshes[n.mesh_id].nvt > 0 ? kXfHasTex : 0u);
    }

    std::vector<DevTexture> textures(d->texture_count);
    std::vector<uint8_t> texels;
    for (uint32_t i = 0; i < d->texture_count; i++) {
        const crt_texture& t = d->textures[i];
        textures[i].width = t.width; textures[i].height = t.height; textures[i].offset = (uint32_t)texels.size();
        textures[i].first = (t.width && t.height) ? ((uint32_t)t.rgb[0] | ((uint32_t)t.rgb[1] << 8) | ((uint32_t)t.rgb[2] << 16)) : 0u;
        const size_t bytes = (size_t)t.width * t.height * 3;
        if (texels.size() + bytes > 0xffffffffull) return fail(ctx, CRT_E_LIMIT, "more than 4 GiB of texels");
        if (bytes) texels.insert(texels.end(), t.rgb, t.rgb + bytes);
    }
    texels.resize(texels.size() + 16, 0);

    std::vector<DevMaterial> materials(d->material_count);
    for (uint32_t i = 0; i < d->material_count; i++) {
        const crt_material& m = d->materials[i];
        DevMaterial dm{};
        dm.diffuse = m.diffuse; dm.specular = m.specular; dm.refraction = m.refraction; dm.emission = m.emission;
        dm.glossiness = m.glossiness; dm.ior = m.ior;
        memcpy(dm.absorption, m.absorption, 12);
        materials[i] = dm;
    }

    std::vector<DevLight> lights(d->light_count ? d->light_count : 1);
    for (uint32_t i = 0; i < d->light_count; i++) {
        const crt_light& l = d->lights[i];
        DevLight dl{};
        dl.kind = l.kind; dl.size = l.size;
        memcpy(dl.intensity, l.intensity, 12); memcpy(dl.pos, l.vec, 12);
        // PointLight::Radiance (scene/lights.cuh:62): the factor is a double expression narrowed to float
        const float k = l.kind == CRT_LIGHT_POINT ? (float)(1.0f / (3.14159265358979323846 * l.size * l.size)) : 1.0f;
        for (int c = 0; c < 3; c++) dl.radiance[c] = l.kind == CRT_LIGHT_POINT ? l.intensity[c] * k : l.intensity[c];
        lights[i] = dl;
    }

    DevScene& S = ctx->S;
    memset(&S, 0, sizeof S);
    std::vector<BoxPair> group_pairs(1);
    const crt_camera& cam = d->camera;
    S.width = cam.width; S.height = cam.height;
    memcpy(S.cam_pos, cam.position, 12); S.dof = cam.dof; S.pixel_size = cam.pixel_size;
    memcpy(S.cX, cam.cX, 12); memcpy(S.cY, cam.cY, 12); memcpy(S.top_left, cam.top_left_pixel, 12);
    S.node_count = N; S.light_count = d->light_count; S.env_texture = d->env_texture;
    // Block shape first: the LDS left for the traversal stack depends on it.  12 of the 16 waves walk rays and 4 shade;
    // a tree too deep for the LDS stack (bound by memory latency, little shading) does better with 13 and 3.  Scenes
    // with little to walk through keep the shading waves busier than the traversal waves: with fewer than 8 mesh
    // nodes 11 + 5 is the better split, without any mesh (spheres and planes only: a query is a few NODE turns) 9 + 7
    // (measured over the fixture scenes; any split computes the same image).
    const uint32_t kLdsLimit = 141 * 1024, kLdsLimitDeep = 150 * 1024;
    // (BVH over the nodes: its own levels, the marker under a mesh walk, then the mesh's levels)
    const uint32_t stack_total = max_depth + 2 + (ctx->tlas ? tlas_depth + 2 : 0);
    S.tlas_root = tlas_root;
    {
        const uint32_t usual = CRT_BLOCK / 64 * 12 / 16;
        uint32_t mesh_nodes = 0;
        for (uint32_t i = 0; i < N; i++) mesh_nodes += d->nodes[i].object_kind == CRT_OBJ_MESH ? 1u : 0u;
        ctx->mesh_nodes = mesh_nodes;
        uint32_t split = usual;
        if (stack_total * usual * 64 * 8 > kLdsLimit) split = usual + 1;
        else if (mesh_nodes == 0) split = usual - 3;
        else if (mesh_nodes < 8) split = usual - 1;
        S.trav_waves = crt::knob_value(getenv("CRT_TRAV_WAVES"), split, 1, CRT_BLOCK / 64 - 1);
    }
    // The traversal stack lives in LDS, depth + 2 entries of 8 bytes per traversal lane, beside the block's queues
    // (18 KiB).  A tree too deep for that (more than 21 entries at 832 lanes) runs the kernel variant for deep trees:
    // queues of half the size (10 KiB), the LDS holds the top of the stack (23 entries) and older entries move to
    // device memory.  CRT_STACK_ENTRIES overrides the number of LDS entries (tests: the same scene with the whole
    // stack in LDS, or with almost all of it in memory).
    const uint32_t lane_bytes = S.trav_waves * 64 * 8;
    const uint32_t fit = kLdsLimit / lane_bytes, fit_deep = kLdsLimitDeep / lane_bytes;
    // Trees that the plain LDS stack holds run the regrouping kernels; a tree too deep for it is bound by memory
    // latency, and there the kernels for deep trees (more queries in flight, 23 stack entries in LDS) are faster.
    ctx->regroup = ctx->tlas || crt::knob_value(getenv("CRT_REGROUP"), stack_total <= fit ? 1 : 0, 0, 1) != 0;
    if (ctx->regroup) {
        // Regrouping kernels: 13 of the 16 waves traverse, and the last S.swing_waves of those take shading turns when the
        // shading queues run long, so one split serves scenes with much and with little to walk through.
        S.trav_waves = crt::knob_value(getenv("CRT_TRAV_WAVES"), 13, 1, CRT_BLOCK / 64 - 1);
        // a block has S.contexts query contexts, each with a stack column and 112 bytes of parked
        // state in LDS; the LDS left beside the rings decides how many stack entries a column holds (the top of the
        // stack; older entries move to device memory).
        S.contexts = crt::knob_value(getenv("CRT_CONTEXTS"), 832, 64, kMaxContexts) & ~63u;
        ctx->deep = false;
        size_t fixed = 0;
        for (int variant = 0; variant < 3; variant++) {
            hipFuncAttributes fa;
            HIP_TRY(ctx, hipFuncGetAttributes(&fa, (const void*)trace_variant(ctx, variant == 1, variant == 2)));
            if (fa.sharedSizeBytes > fixed) fixed = fa.sharedSizeBytes;
        }
        const size_t per_context = (160u * 1024 - fixed) / S.contexts;
        if (per_context < kContextBytes + 2 * 8) return fail(ctx, CRT_E_LIMIT, "query contexts do not fit in LDS");
        const uint32_t room = (uint32_t)((per_context - kContextBytes) / 8);
        const uint32_t entries = crt::knob_value(getenv("CRT_STACK_ENTRIES"), room, 2, room);
        S.stack_entries = stack_total < entries ? stack_total : entries;
        ctx->stack_overflow = stack_total - S.stack_entries;
    } else {
        const uint32_t lds_entries_max = crt::knob_value(getenv("CRT_STACK_ENTRIES"), stack_total <= fit ? fit : fit_deep, 2, 66);
        S.stack_entries = stack_total < lds_entries_max ? stack_total : lds_entries_max;
        ctx->stack_overflow = stack_total - S.stack_entries;
        ctx->deep = ctx->stack_overflow > 0 || S.stack_entries * lane_bytes > kLdsLimit;
    }
    S.stack_overflow = ctx->stack_overflow;
    S.keep_box = crt::knob_value(getenv("CRT_KEEP_BOX"), 8, 1, 65);
    S.keep_leaf = crt::knob_value(getenv("CRT_KEEP_LEAF"), 8, 1, 65);
    S.keep_node = crt::knob_value(getenv("CRT_KEEP_NODE"), 8, 1, 65);
    S.keep_exchange = crt::knob_value(getenv("CRT_KEEP_EXCHANGE"), 40, 1, 65);
    S.exchange_first = crt::knob_value(getenv("CRT_EXCHANGE_FIRST"), 32, 1, 65);
    S.first_step_idle = crt::knob_value(getenv("CRT_FIRST_STEP_IDLE"), 6, 0, 65);
    S.direct_start = crt::knob_value(getenv("CRT_DIRECT_START"), 1, 0, 1);
    S.swing_waves = crt::knob_value(getenv("CRT_SWING_WAVES"), 6, 0, 16);
    S.swing_fill = crt::knob_value(getenv("CRT_SWING_FILL"), 128, 1, 1u << 20);
    S.burst_max = crt::knob_value(getenv("CRT_BURST_MAX"), 64, 1, 1u << 20);
    {
        // Drift bound of the scan's candidate mask.  One round trip through node j maps p to fl(tm * fl(itm * p)): its
        // distance from p is at most |R p + r| (R = tm3 * itm3 - I, r = tm3 * t_itm + t_tm: the stored matrices are
        // not exact inverses) plus the rounding of the two affine maps, 8.2 u |tm3| (|itm3| |p| + |t_itm|) + 4.1 u |t_tm|
        // (u = 2^-24, four roundings per output).  A query enters every node at most once, so the sums over all
        // object nodes bound the total drift; 5 % on top covers the growth of |p| along the way.
        const double u = 1.0 / 16777216.0;
        double rel = 0, abs_ = 0;
        for (uint32_t i = 0; i < N; i++) {
            const crt_node& n = d->nodes[i];
            if (n.object_kind == CRT_OBJ_NONE) continue;
            double nr = 0, nabs = 0, nrt = 0, nat = 0, nt = 0;
            for (int row = 0; row < 3; row++) {
                double r_row = 0, a_row = 0, rt = n.tm[9 + row], at = 0;
                for (int col = 0; col < 3; col++) {
                    double rc = 0, ac = 0;
                    for (int k = 0; k < 3; k++) {
                        rc += (double)n.tm[3 * k + row] * (double)n.itm[3 * col + k];
                        ac += fabs((double)n.tm[3 * k + row]) * fabs((double)n.itm[3 * col + k]);
                    }
                    r_row += fabs(rc - (row == col ? 1.0 : 0.0));
                    a_row += ac;
                }
                for (int k = 0; k < 3; k++) {
                    rt += (double)n.tm[3 * k + row] * (double)n.itm[9 + k];
                    at += fabs((double)n.tm[3 * k + row]) * fabs((double)n.itm[9 + k]);
                }
                nr = fmax(nr, r_row); nabs = fmax(nabs, a_row); nrt = fmax(nrt, fabs(rt)); nat = fmax(nat, at);
                nt = fmax(nt, fabs((double)n.tm[9 + row]));
            }
            rel += nr + 9.0 * u * nabs;
            abs_ += nrt + 9.0 * u * (nat + nt);
        }
        rel *= 1.05; abs_ *= 1.05;
        // (the 5 % assume that |p| grows by less than that along a query: exp(rel) - 1 does, far below either limit)
        const bool grouped = N > 128;
        const bool usable = rel == rel && abs_ == abs_ && rel < (grouped ? 1e-2 : 1e-3) && abs_ < 1e30;
        S.scan_mask = usable ? (grouped ? 5u : 1u) : 0u;
        if (grouped) {
            // One mask bit per group of 2^shift consecutive nodes, at most 128 groups; a group's box is the union of
            // the boxes of its object nodes (a group without one is never a candidate).
            uint32_t shift = 1;
            while (((N + (1u << shift) - 1) >> shift) > 128) shift++;
            const uint32_t G = (N + (1u << shift) - 1) >> shift;
            S.group_shift = shift; S.group_count = G;
            group_pairs.assign(G, BoxPair{});
            auto group_half = [&](uint32_t g, float* bmin, float* bmax, uint32_t& ref) {
                for (int k = 0; k < 3; k++) { bmin[k] = CRT_BIGFLOAT; bmax[k] = -CRT_BIGFLOAT; }
                bool any = false;
                for (uint32_t i = g << shift; g < G && i < N && i < ((g + 1) << shift); i++) {
                    if (d->nodes[i].object_kind == CRT_OBJ_NONE) continue;
                    any = true;
                    for (int k = 0; k < 3; k++) { bmin[k] = fminf(bmin[k], d->nodes[i].box_min[k]); bmax[k] = fmaxf(bmax[k], d->nodes[i].box_max[k]); }
                }
                ref = (any ? (uint32_t)CRT_OBJ_SPHERE : (uint32_t)CRT_OBJ_NONE) | ((g + 1) << 2);
            };
            for (uint32_t g = 0; g < G; g++) {
                BoxPair& p = group_pairs[g];
                float lo[3], hi[3];
                group_half(g, lo, hi, p.ref_a); set_box(p.a, lo, hi);
                group_half(g + 1, lo, hi, p.ref_b); set_box(p.b, lo, hi);
                p.aux_a = p.aux_b = 0;
            }
        }
        S.drift_rel = (float)rel * 1.000001f; S.drift_abs = (float)abs_ * 1.000001f;
        if (const char* e = getenv("CRT_NO_SCAN_MASK")) if (atoi(e) != 0) S.scan_mask = 0;
        if (ctx->tlas) S.scan_mask = 0;      // nothing is scanned
        // profiling only: the counted build then scans by the mask too (its node counters no longer match the reference's)
        if (const char* e = getenv("CRT_COUNTED_BY_MASK")) if (atoi(e) != 0 && S.scan_mask) S.scan_mask = 3;
    }
    // Scheduler and block-shape parameters: defaults measured on the benchmark frame (cache resident, bound by
    // instruction issue) and, for a BVH deeper than the LDS stack, on the ten-million-triangle scene (bound by memory
    // latency: longer box bursts, later exchanges, and two queries per pixel in flight whenever there are two).
    // Environment overrides outside the stated range are ignored.
    S.sched_box_min = crt::knob_value(getenv("CRT_SCHED_BOX_MIN"), ctx->deep ? 32 : 20, 1, 64);
    S.sched_special_min = crt::knob_value(getenv("CRT_SCHED_SPECIAL_MIN"), 6, 1, 64);
    S.sched_exchange_min = crt::knob_value(getenv("CRT_SCHED_EXCHANGE_MIN"), ctx->deep ? 16 : 12, 1, 64);
    S.first_step = crt::knob_value(getenv("CRT_FIRST_STEP"), 96, 0, 1u << 20);
    {
        const uint32_t most = ctx->deep ? kMaxSlotsDeep : kMaxSlots, usual = ctx->deep ? 72 * S.trav_waves : (ctx->regroup ? 108 : 96) * S.trav_waves;
        S.slots = crt::knob_value(getenv("CRT_SLOTS"), usual < most ? usual : most, 64, most);
    }
    S.serve_batch_min = crt::knob_value(getenv("CRT_SERVE_BATCH_MIN"), 64, 1, 64);
    S.serve_low_water = crt::knob_value(getenv("CRT_SERVE_LOW_WATER"), 64, 0, kQueueSize);
    S.serve_patience = crt::knob_value(getenv("CRT_SERVE_PATIENCE"), 64, 0, 1u << 20);
    S.serve_priority = crt::knob_value(getenv("CRT_SERVE_PRIORITY"), 1, 0, 1);
    S.pair_live = crt::knob_value(getenv("CRT_PAIR_LIVE"), ctx->deep ? (1u << 20) : S.trav_waves * 80, 0, 1u << 20);
    S.pair_below = crt::knob_value(getenv("CRT_PAIR_BELOW"), 0, 0, 1u << 20);
    // End of the pixel queue (regrouping kernels): within the last "as many pixels as the GPU holds at a time" (but no more
    // than a fifth of the partition) a block keeps only contexts - 128 of its slots supplied, so that less is in
    // flight when the queue runs dry and the blocks finish closer together.  Measured on the benchmark frame: 139.9 ->
    // 137.9 ms, 2 to 3 % on the other scenes with meshes; a window of a quarter of the frame or a floor below 450 slots
    // lose it again.  Scenes without a mesh (queries of a few steps: the shading turns are what a block waits for, and they
    // want every slot) were 4 to 5 % slower with it and keep all slots supplied to the end.
    ctx->tail_window_knob = getenv("CRT_TAIL_WINDOW") ? (int64_t)crt::knob_value(getenv("CRT_TAIL_WINDOW"), 0, 0, 0x7fffffff) : -1;
    S.tail_live = crt::knob_value(getenv("CRT_TAIL_LIVE"), ctx->regroup && S.contexts > 256 ? S.contexts - 128 : S.slots, 64, 1u << 20);
    S.fault_inject = crt::knob_value(getenv("CRT_FAULT_INJECT"), 0, 0, 1);      // tests only: the kernel loses one query, so that its watchdog has something to find
    S.watchdog_ticks = 100000000ull * crt::knob_value(getenv("CRT_WATCHDOG_SECONDS"), 30, 1, 3600);
    ctx->interleave = crt::knob_value(getenv("CRT_TILE_INTERLEAVE"), 0, 0, 1) != 0;
    { const char* e = getenv("CRT_NO_BALANCE"); ctx->no_balance = e && atoi(e) != 0; }
    if ((rc = upload(ctx, pairs, &S.pairs)) || (rc = upload(ctx, tris, &S.tris)) || (rc = upload(ctx, xf, &S.xf)) ||
        (rc = upload(ctx, attrs, &S.attrs)) || (rc = upload(ctx, materials, &S.materials)) ||
        (rc = upload(ctx, textures, &S.textures)) || (rc = upload(ctx, texels, &S.texels)) || (rc = upload(ctx, lights, &S.lights)) ||
        (rc = upload(ctx, group_pairs, &S.group_pairs)))
        return rc;

    const size_t px = (size_t)cam.width * cam.height;
    float* acc[3];
    for (int k = 0; k < 3; k++) {
        HIP_TRY(ctx, hipMalloc((void**)&acc[k], px * 12));
        ctx->allocs.push_back(acc[k]);
        HIP_TRY(ctx, hipMemset(acc[k], 0, px * 12));
    }
    S.results = acc[0]; S.normals = acc[1]; S.albedos = acc[2];
    HIP_TRY(ctx, hipMalloc((void**)&ctx->rgb8, px * 3));
    ctx->allocs.push_back(ctx->rgb8);
    HIP_TRY(ctx, hipMemset(ctx->rgb8, 0, px * 3));
    HIP_TRY(ctx, hipMemset(ctx->stats, 0, sizeof(DevStats)));

    // two LDS arrays (node reference, entry distance) of stack_entries words per traversal thread
    const uint32_t trav_threads = S.trav_waves * 64;
    ctx->lds_bytes = ctx->regroup ? (size_t)S.contexts * (S.stack_entries * 8 + kContextBytes) : (size_t)S.stack_entries * trav_threads * 4 * 2;
    if (!ctx->regroup && ctx->lds_bytes > (ctx->deep ? 150u : 141u) * 1024) return fail(ctx, CRT_E_LIMIT, "traversal stack does not fit in LDS");      // 160 KiB less the queues and counters of the block
    for (int variant = 0; variant < 3; variant++)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)trace_variant(ctx, variant == 1, variant == 2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes));
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->blocks_per_cu, trace_variant(ctx, false, false), CRT_BLOCK, ctx->lds_bytes));
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->blocks_per_cu_counted, trace_variant(ctx, true, false), CRT_BLOCK, ctx->lds_bytes));
    if (ctx->blocks_per_cu < 1) ctx->blocks_per_cu = 1;
    if (ctx->blocks_per_cu_counted < 1) ctx->blocks_per_cu_counted = 1;
    {
        const int per_cu = ctx->blocks_per_cu > ctx->blocks_per_cu_counted ? ctx->blocks_per_cu : ctx->blocks_per_cu_counted;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->cold, (size_t)ctx->num_cus * per_cu * block_state_bytes(ctx)));
        ctx->allocs.push_back(ctx->cold);
    }
    {
        // one entry per 8x8 tile of the image, with room for the ragged last strip of any strip height up to 512 rows
        ctx->tile_capacity = ((S.width + 7) / 8) * ((S.height + 7) / 8 + 64);
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tile_cost, (size_t)ctx->tile_capacity * 4));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tile_order, (size_t)ctx->tile_capacity * 4));
        ctx->allocs.push_back(ctx->tile_cost);
        ctx->allocs.push_back(ctx->tile_order);
    }

    if (partition_items(S, 16) > 0x7fffffffull) return fail(ctx, CRT_E_LIMIT, "image too large for the pixel queue");
    ctx->have_scene = true;
    set_partition(ctx, 0, 1, 16);
    return CRT_OK;
}

int crt_set_partition(crt_ctx* ctx, uint32_t part, uint32_t parts, uint32_t strip_rows) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    if (parts == 0 || part >= parts || strip_rows == 0 || strip_rows % 8 != 0) return fail(ctx, CRT_E_INVALID, "bad partition (strip_rows must be a positive multiple of 8)");
    if (partition_items(ctx->S, strip_rows) > 0x7fffffffull) return fail(ctx, CRT_E_LIMIT, "image too large for the pixel queue");
    set_partition(ctx, part, parts, strip_rows);
    return CRT_OK;
}

int crt_clear_async(crt_ctx* ctx, void* stream) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)ctx->S.width * ctx->S.height;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->S.results, 0, px * 12, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->S.normals, 0, px * 12, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->S.albedos, 0, px * 12, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->stats, 0, sizeof(DevStats), st));
    return CRT_OK;
}

int crt_clear(crt_ctx* ctx) {
    const int rc = crt_clear_async(ctx, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return CRT_OK;
}

static int render_impl(crt_ctx* ctx, int pass_id, int samples, hipStream_t stream, bool async, bool counted) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    if (pass_id < 0 || samples <= 0) return fail(ctx, CRT_E_INVALID, "pass_id must be >= 0 and samples > 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t* jump = nullptr;
    int rc = get_jump(ctx, pass_id, &jump);
    if (rc) return rc;
    if (ctx->P.total_items == 0) return CRT_OK;
    HIP_TRY(ctx, hipMemsetAsync(ctx->queue, 0, 4, stream));
    const uint32_t grid = trace_grid(ctx, counted ? ctx->blocks_per_cu_counted : ctx->blocks_per_cu);
    // Load balance: this pass measures a cost per tile and the next one hands tiles out costliest first.
    // (Passes of one context are expected on one stream, or ordered by the caller.)
    const uint32_t tiles = ctx->P.total_items >> 6;
    DevPartition P = ctx->P;
    const bool balance = tiles <= ctx->tile_capacity && !ctx->no_balance;
    if (balance) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->tile_cost, 0, (size_t)tiles * 4, stream));
        P.tile_cost = ctx->tile_cost;
        P.tile_order = ctx->order_valid ? ctx->tile_order : nullptr;
    }
    const int slot = (int)(ctx->launches % crt_ctx::kRing);
    HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], stream));
    launch_trace(ctx, counted, false, grid, stream, P, samples, jump, ctx->queue, nullptr, nullptr, nullptr, ctx->cold);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], stream));
    ctx->launches++;
    if (balance) {
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, ctx->tile_cost, ctx->tile_order, tiles, ctx->interleave ? 1 : 0);
        HIP_TRY(ctx, hipGetLastError());
        ctx->order_valid = true;
    }
    if (!async) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        return check_kernel_errors(ctx);
    }
    return CRT_OK;
}

int crt_render_pass(crt_ctx* ctx, int pass_id, int samples, void* stream, int async) {
    return render_impl(ctx, pass_id, samples, (hipStream_t)stream, async != 0, false);
}

int crt_render_pass_counted(crt_ctx* ctx, int pass_id, int samples) {
    return render_impl(ctx, pass_id, samples, nullptr, false, true);
}

int crt_render_pass_noalloc(crt_ctx* ctx, int pass_id, int samples, void* stream_, int async) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    if (pass_id < 0 || samples <= 0) return fail(ctx, CRT_E_INVALID, "pass_id must be >= 0 and samples > 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->P.total_items == 0) return CRT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    // the grid is cut down to the blocks whose pixel slots fit the workspace
    const size_t block_bytes = block_state_bytes(ctx);
    const size_t blocks_fit = (kScratchBytes - kScratchStateOff) / block_bytes;
    if (blocks_fit < 1)
        return fail(ctx, CRT_E_LIMIT, "scene needs " + std::to_string(block_bytes) + " bytes of path state per block, the static workspace holds " +
                                      std::to_string(kScratchBytes - kScratchStateOff));
    uint32_t grid = trace_grid(ctx, ctx->blocks_per_cu);
    if (grid > blocks_fit) grid = (uint32_t)blocks_fit;
    unsigned char* ws = ctx->scratch;
    const uint32_t* jump = nullptr;
    if (pass_id >= kScratchPasses)
        return fail(ctx, CRT_E_LIMIT, "pass_id " + std::to_string(pass_id) + " is beyond the " + std::to_string(kScratchPasses) + " passes whose sequence jumps the static workspace holds");
    if (pass_id > 0) jump = (const uint32_t*)(ws + kScratchJumpOff + (size_t)pass_id * kJumpBytes);
    // The workspace keeps what the previous call left.  The path state is written before it is read, pixel by pixel; the
    // pixel queue's counter has to start at zero, and every launch leaves it there for the next one (a memset node
    // between two kernels of a recorded graph was seen to run before the first kernel had finished).
    DevPartition P = ctx->P;
    P.tile_cost = nullptr; P.tile_order = nullptr;      // storage order: the balance tables belong to crt_render_pass
    P.self_reset = 1;
    uint32_t* queue = (uint32_t*)(ws + kScratchQueueOff);
    uint32_t* state = (uint32_t*)(ws + kScratchStateOff);
    launch_trace(ctx, false, false, grid, stream, P, samples, jump, queue, nullptr, nullptr, nullptr, state);
    HIP_TRY(ctx, hipGetLastError());
    if (!async) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        return check_kernel_errors(ctx);
    }
    return CRT_OK;
}

int crt_last_kernel_ms(crt_ctx* ctx, float* ms) {
    if (!ctx || !ms) return CRT_E_INVALID;
    if (!ctx->launches) return fail(ctx, CRT_E_INVALID, "no render pass has been launched");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)((ctx->launches - 1) % crt_ctx::kRing);
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1[slot]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0[slot], ctx->ev1[slot]));
    return CRT_OK;
}

int crt_kernel_times(crt_ctx* ctx, float* ms, uint32_t cap, uint32_t* count) {
    if (!ctx || !count || (cap && !ms)) return CRT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint64_t first = ctx->reported;
    if (ctx->launches - first > (uint64_t)crt_ctx::kRing) first = ctx->launches - crt_ctx::kRing;
    uint32_t n = 0;
    for (uint64_t i = first; i < ctx->launches; i++, n++) {
        const int slot = (int)(i % crt_ctx::kRing);
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1[slot]));
        float t = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev0[slot], ctx->ev1[slot]));
        if (n < cap) ms[n] = t;
    }
    ctx->reported = ctx->launches;
    *count = n;
    return CRT_OK;
}

int crt_convert(crt_ctx* ctx, float pass_mul, int srgb, void* rgb_out_device, void* stream, int async) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->S.width * ctx->S.height;
    uint8_t* out = rgb_out_device ? (uint8_t*)rgb_out_device : ctx->rgb8;
    hipLaunchKernelGGL(convert_kernel, dim3((n + CRT_BLOCK - 1) / CRT_BLOCK), dim3(CRT_BLOCK), 0, (hipStream_t)stream,
                       ctx->S.results, out, n, srgb, pass_mul);
    HIP_TRY(ctx, hipGetLastError());
    if (!async) HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CRT_OK;
}

int crt_prepare_denoise(crt_ctx* ctx, float pass_mul, float* beauty_device, float* normal_device, float* albedo_device,
                        void* stream, int async) {
    if (!ctx || !beauty_device || !normal_device || !albedo_device) return ctx ? fail(ctx, CRT_E_INVALID, "null output buffer") : CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n3 = ctx->S.width * ctx->S.height * 3;
    hipLaunchKernelGGL(prepare_denoise_kernel, dim3((n3 + CRT_BLOCK - 1) / CRT_BLOCK), dim3(CRT_BLOCK), 0, (hipStream_t)stream,
                       ctx->S.results, ctx->S.normals, ctx->S.albedos, beauty_device, normal_device, albedo_device, n3, pass_mul);
    HIP_TRY(ctx, hipGetLastError());
    if (!async) HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CRT_OK;
}

int crt_convert_floats(crt_ctx* ctx, const float* rgb_in_device, int srgb, void* rgb_out_device, void* stream, int async) {
    if (!ctx || !rgb_in_device) return ctx ? fail(ctx, CRT_E_INVALID, "null input buffer") : CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->S.width * ctx->S.height;
    uint8_t* out = rgb_out_device ? (uint8_t*)rgb_out_device : ctx->rgb8;
    hipLaunchKernelGGL(convert_floats_kernel, dim3((n + CRT_BLOCK - 1) / CRT_BLOCK), dim3(CRT_BLOCK), 0, (hipStream_t)stream,
                       rgb_in_device, out, n, srgb);
    HIP_TRY(ctx, hipGetLastError());
    if (!async) HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CRT_OK;
}

int crt_read_image(crt_ctx* ctx, uint8_t* rgb_host) {
    if (!ctx || !rgb_host) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(rgb_host, ctx->rgb8, (size_t)ctx->S.width * ctx->S.height * 3, hipMemcpyDeviceToHost));
    return check_kernel_errors(ctx);
}

int crt_read_accumulators(crt_ctx* ctx, float* results, float* normals, float* albedos) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->S.width * ctx->S.height * 12;
    if (results) HIP_TRY(ctx, hipMemcpy(results, ctx->S.results, bytes, hipMemcpyDeviceToHost));
    if (normals) HIP_TRY(ctx, hipMemcpy(normals, ctx->S.normals, bytes, hipMemcpyDeviceToHost));
    if (albedos) HIP_TRY(ctx, hipMemcpy(albedos, ctx->S.albedos, bytes, hipMemcpyDeviceToHost));
    return check_kernel_errors(ctx);
}

int crt_device_buffers(crt_ctx* ctx, void** results, void** normals, void** albedos, void** rgb8) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    if (results) *results = ctx->S.results;
    if (normals) *normals = ctx->S.normals;
    if (albedos) *albedos = ctx->S.albedos;
    if (rgb8) *rgb8 = ctx->rgb8;
    return CRT_OK;
}

int crt_primary_hits(crt_ctx* ctx, int pass_id, int32_t* node, int32_t* face, float* t) {
    if (!ctx) return CRT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, CRT_E_NOSCENE, "no scene uploaded");
    if (pass_id < 0) return fail(ctx, CRT_E_INVALID, "pass_id must be >= 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t* jump = nullptr;
    int rc = get_jump(ctx, pass_id, &jump);
    if (rc) return rc;
    const size_t px = (size_t)ctx->S.width * ctx->S.height;
    // released on every return path
    struct Buffers {
        int32_t *node = nullptr, *face = nullptr;
        float* t = nullptr;
        ~Buffers() { if (node) (void)hipFree(node); if (face) (void)hipFree(face); if (t) (void)hipFree(t); }
    } d;
    HIP_TRY(ctx, hipMalloc((void**)&d.node, px * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d.face, px * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d.t, px * 4));
    // pixels this context does not own read as "not traced"
    HIP_TRY(ctx, hipMemset(d.node, 0xff, px * 4));
    HIP_TRY(ctx, hipMemset(d.face, 0xff, px * 4));
    HIP_TRY(ctx, hipMemset(d.t, 0, px * 4));
    if (ctx->P.total_items) {
        HIP_TRY(ctx, hipMemset(ctx->queue, 0, 4));
        const uint32_t grid = trace_grid(ctx, ctx->blocks_per_cu);
        launch_trace(ctx, false, true, grid, nullptr, ctx->P, 1, jump, ctx->queue, d.node, d.face, d.t, ctx->cold);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (node) HIP_TRY(ctx, hipMemcpy(node, d.node, px * 4, hipMemcpyDeviceToHost));
    if (face) HIP_TRY(ctx, hipMemcpy(face, d.face, px * 4, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(ctx, hipMemcpy(t, d.t, px * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_get_stats(crt_ctx* ctx, crt_stats* out) {
    if (!ctx || !out) return CRT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevStats s;
    HIP_TRY(ctx, hipMemcpy(&s, ctx->stats, sizeof s, hipMemcpyDeviceToHost));
    out->closest_rays = s.closest_rays; out->shadow_rays = s.shadow_rays; out->paths = s.paths;
    out->node_box_tests = s.node_box_tests; out->node_enters = s.node_enters; out->bvh_interior = s.bvh_interior;
    out->bvh_leaves = s.bvh_leaves; out->tri_tests = s.tri_tests; out->tri_hits = s.tri_hits;
    out->sphere_tests = s.sphere_tests; out->plane_tests = s.plane_tests; out->texel_quads = s.texel_quads;
    for (int k = 0; k < 12; k++) { out->sched_turns[k] = s.sched_turns[k]; out->sched_lanes[k] = s.sched_lanes[k]; out->sched_cycles[k] = s.sched_cycles[k]; }
    for (int k = 0; k < 8; k++) out->path_cycles[k] = s.path_cycles[k];
    out->scan_checks = s.scan_checks; out->scan_violations = s.scan_violations;
    out->errors = s.errors;
    if (s.errors) return fail(ctx, CRT_E_HIP, "the trace kernel reported " + std::to_string(s.errors) + " queue protocol error(s): the image is incomplete");
    return CRT_OK;
}

int crt_rng_selftest(crt_ctx* ctx, const uint32_t* seeds, const uint32_t* subsequences, uint32_t n, uint32_t draws, uint32_t* out) {
    if (!ctx || !seeds || !subsequences || !out || n == 0) return CRT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one jump matrix per distinct non-zero subsequence
    std::map<uint32_t, uint32_t> slot;
    std::vector<uint32_t> mats, index(n);
    for (uint32_t i = 0; i < n; i++) {
        if (subsequences[i] == 0) { index[i] = 0xffffffffu; continue; }
        auto it = slot.find(subsequences[i]);
        if (it == slot.end()) {
            BitMat m;
            jump_for_pass(subsequences[i], m);
            it = slot.emplace(subsequences[i], (uint32_t)(mats.size() / 800)).first;
            mats.insert(mats.end(), &m.row[0][0], &m.row[0][0] + 800);
        }
        index[i] = it->second;
    }
    uint32_t *d_seeds = nullptr, *d_index = nullptr, *d_mats = nullptr, *d_out = nullptr;
    const size_t out_words = (size_t)n * (6 + draws);
    HIP_TRY(ctx, hipMalloc((void**)&d_seeds, n * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d_index, n * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d_mats, mats.size() * 4 + 16));
    HIP_TRY(ctx, hipMalloc((void**)&d_out, out_words * 4));
    HIP_TRY(ctx, hipMemcpy(d_seeds, seeds, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_index, index.data(), n * 4, hipMemcpyHostToDevice));
    if (!mats.empty()) HIP_TRY(ctx, hipMemcpy(d_mats, mats.data(), mats.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rng_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, d_seeds, d_index, d_mats, n, draws, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(out, d_out, out_words * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_seeds); (void)hipFree(d_index); (void)hipFree(d_mats); (void)hipFree(d_out);
    return CRT_OK;
}

int crt_math_selftest(crt_ctx* ctx, int fn, const float* x, const float* y, uint32_t n, float* out) {
    if (!ctx || !out || fn < 0 || (fn > 8 && fn != 100)) return CRT_E_INVALID;
    if (fn != 100 && (!x || n == 0)) return CRT_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (fn == 100) {
        // every float bit pattern through the reciprocal the kernels use and through the division it stands for:
        // out[0] = number of patterns on which the two differ (x is not read)
        unsigned long long* d_bad = nullptr;
        unsigned long long bad = 0;
        HIP_TRY(ctx, hipMalloc((void**)&d_bad, sizeof bad));
        int rc = CRT_OK;
        if (hipMemset(d_bad, 0, sizeof bad) != hipSuccess) rc = CRT_E_HIP;
        if (rc == CRT_OK) {
            hipLaunchKernelGGL(rcp_exhaustive_kernel, dim3(4096), dim3(256), 0, nullptr, d_bad);
            if (hipGetLastError() != hipSuccess || hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) rc = CRT_E_HIP;
        }
        (void)hipFree(d_bad);
        if (rc != CRT_OK) return fail(ctx, rc, "reciprocal self-test failed to run");
        out[0] = (float)bad;
        return CRT_OK;
    }
    float *d_x = nullptr, *d_y = nullptr, *d_o = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_x, n * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d_y, n * 4));
    HIP_TRY(ctx, hipMalloc((void**)&d_o, n * 4));
    HIP_TRY(ctx, hipMemcpy(d_x, x, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_y, y ? y : x, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, fn, d_x, d_y, n, d_o);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpy(out, d_o, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_o);
    return CRT_OK;
}

}  // extern "C"
