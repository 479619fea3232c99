// TEST INFRASTRUCTURE ONLY.  CPU oracle for the per-pixel trace path.
//
// A plain CPU restatement of the reference algorithm (smotired/CUDARaytracing),
// kept deliberately close to the reference's own shape: array-of-struct nodes,
// 64-bit indices, element -> face -> vertex indirection, one object visit per
// scene node, recursion-free stacks.  Each function names the reference
// file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product never does.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and
// it cannot be built in this image (it needs the CUDA toolkit headers
// cuda_runtime.h, curand_kernel.h and libcu++'s cuda/std/variant).  The only
// outputs of the reference available are the counts recorded when the project
// was surveyed (BVH node counts per mesh, primary-hit counts per scene); the
// tests check those.  The cuRAND XORWOW seeding constants below are restated
// from the published algorithm and are not verifiable here; the sequence jump
// is cross-checked against rocRAND's tables (same recurrence).
//
// Arithmetic: every float expression keeps the reference's association and
// promotion (including its double sub-expressions); build with
// -ffp-contract=off.  Transcendentals come from oracle_detmath.h, or glibc
// with -DORACLE_LIBM_GLIBC.  float -> int follows the device conversion rule
// (saturating, NaN -> 0) on the path and the host rule when building textures.
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "oracle_detmath.h"

// ---------------------------------------------------------------- math layer
#ifdef ORACLE_LIBM_GLIBC
static inline float m_sinf(float x) { return sinf(x); }
static inline float m_cosf(float x) { return cosf(x); }
static inline float m_tanf(float x) { return tanf(x); }
static inline float m_expf(float x) { return expf(x); }
static inline float m_powf(float x, float y) { return powf(x, y); }
static inline float m_atan2f(float y, float x) { return atan2f(y, x); }
static inline float m_asinf(float x) { return asinf(x); }
static inline float m_acosf(float x) { return acosf(x); }
#else
static inline float m_sinf(float x) { return odm::det_sinf(x); }
static inline float m_cosf(float x) { return odm::det_cosf(x); }
static inline float m_tanf(float x) { return odm::det_tanf(x); }
static inline float m_expf(float x) { return odm::det_expf(x); }
static inline float m_powf(float x, float y) { return odm::det_powf(x, y); }
static inline float m_atan2f(float y, float x) { return odm::det_atan2f(y, x); }
static inline float m_asinf(float x) { return odm::det_asinf(x); }
static inline float m_acosf(float x) { return odm::det_acosf(x); }
#endif

static inline int dev_f2i(float f) { return odm::f2i_sat(f); }

// ------------------------------------------------ math/float3.cuh:5-194
#define BIGFLOAT 3.402823466e+38f
struct float3 { float x, y, z; };
typedef float3 color;
static inline float3 mk(float x, float y, float z) { float3 r = {x, y, z}; return r; }
#define F3_ZERO mk(0.0f, 0.0f, 0.0f)
#define F3_RIGHT mk(1.0f, 0.0f, 0.0f)
#define F3_UP mk(0.0f, 1.0f, 0.0f)
#define F3_FORWARD mk(0.0f, 0.0f, 1.0f)
#define BLACK mk(0.0f, 0.0f, 0.0f)
#define WHITE mk(1.0f, 1.0f, 1.0f)
static inline float3 operator-(float3 a) { return mk(-a.x, -a.y, -a.z); }
static inline float3 operator*(float3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline float3 operator*(float s, float3 a) { return mk(a.x * s, a.y * s, a.z * s); }
static inline void operator*=(float3& a, float s) { a.x *= s; a.y *= s; a.z *= s; }
static inline float3 operator+(float3 a, float3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline void operator+=(float3& a, float3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
static inline float3 operator-(float3 a, float3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float3 operator*(float3 a, float3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline void operator*=(float3& a, float3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
static inline float operator%(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// float3.cuh:154-156: three scaled basis vectors, kept literally
static inline float3 cross(float3 a, float3 b) {
    return F3_RIGHT * (a.y * b.z - b.y * a.z) - F3_UP * (a.x * b.z - b.x * a.z) + F3_FORWARD * (a.x * b.y - b.x * a.y);
}
static inline float lengthsq(float3 a) { return a % a; }
static inline float length(float3 a) { return sqrtf(a % a); }
static inline float3 asNorm(float3 a) { float scale = 1.0f / sqrtf(a % a); return scale * a; }
static inline void doNorm(float3& a) { float scale = 1.0f / sqrtf(a % a); a *= scale; }
static inline float ref(const float3& a, unsigned i) { switch (i) { case 0: return a.x; case 1: return a.y; case 2: return a.z; default: return a.x; } }

// ------------------------------------------------ renderer/settings.cuh:6-21
#define BIAS 0.02f
#define MAX_SURVIVAL 0.95f
#define F_EPS 1.1920929e-7f

// ------------------------------------------------ math/matrix.cuh:72-97
struct Matrix {
    float cells[12];
    void TransformPosition(float3& v) const {
        v = mk(cells[0] * v.x + cells[3] * v.y + cells[6] * v.z + cells[9],
               cells[1] * v.x + cells[4] * v.y + cells[7] * v.z + cells[10],
               cells[2] * v.x + cells[5] * v.y + cells[8] * v.z + cells[11]);
    }
    void TransformDirection(float3& v) const {
        v = mk(cells[0] * v.x + cells[3] * v.y + cells[6] * v.z,
               cells[1] * v.x + cells[4] * v.y + cells[7] * v.z,
               cells[2] * v.x + cells[5] * v.y + cells[8] * v.z);
    }
    void TransformNormal(float3& v) const {
        v = mk(cells[0] * v.x + cells[1] * v.y + cells[2] * v.z,
               cells[3] * v.x + cells[4] * v.y + cells[5] * v.z,
               cells[6] * v.x + cells[7] * v.y + cells[8] * v.z);
        doNorm(v);
    }
};

// ------------------------------------------------ cuRAND XORWOW (third party, not in the reference tree)
// Recurrence and output: Marsaglia xorwow, 5-word xorshift plus Weyl counter.
// Seeding as curand_init(seed, subsequence, offset = 0) (trace.cu:22-23): the
// seed words are salted and scrambled, then the xorshift part is advanced by
// subsequence * 2^67 steps (the Weyl word is unchanged because 2^67 * 362437
// is a multiple of 2^32).
struct XorwowState { uint32_t d; uint32_t v[5]; };

static inline uint32_t xorwow_next(XorwowState* s) {
    uint32_t t = (s->v[0] ^ (s->v[0] >> 2));
    s->v[0] = s->v[1]; s->v[1] = s->v[2]; s->v[2] = s->v[3]; s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437;
    return s->v[4] + s->d;
}

// 160 x 160 GF(2) matrices, row r = image of state bit r (word r / 32, bit r % 32).
struct JumpMatrix { uint32_t row[160][5]; };

static void jump_apply(const JumpMatrix& m, uint32_t v[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (v[i] & (1u << j))
                for (int k = 0; k < 5; k++) r[k] ^= m.row[i * 32 + j][k];
    for (int k = 0; k < 5; k++) v[k] = r[k];
}

static void jump_square(const JumpMatrix& a, JumpMatrix& out) {
    for (int r = 0; r < 160; r++) {
        uint32_t v[5];
        for (int k = 0; k < 5; k++) v[k] = a.row[r][k];
        jump_apply(a, v);
        for (int k = 0; k < 5; k++) out.row[r][k] = v[k];
    }
}

// g_seqjump[k] advances by 2^67 * 4^k steps, the table cuRAND walks with the
// base-4 digits of the subsequence number.
static JumpMatrix g_seqjump[16];
static std::atomic<int> g_seqjump_ready(0);
static void seqjump_init() {
    static std::atomic<int> lock(0);
    if (g_seqjump_ready.load()) return;
    int expected = 0;
    while (!lock.compare_exchange_weak(expected, 1)) { expected = 0; std::this_thread::yield(); }
    if (!g_seqjump_ready.load()) {
        JumpMatrix* a = new JumpMatrix;
        JumpMatrix* b = new JumpMatrix;
        for (int r = 0; r < 160; r++) {           // one step of the xorshift part
            XorwowState s; s.d = 0;
            for (int k = 0; k < 5; k++) s.v[k] = 0;
            s.v[r / 32] = 1u << (r % 32);
            xorwow_next(&s);
            for (int k = 0; k < 5; k++) a->row[r][k] = s.v[k];
        }
        for (int i = 0; i < 67; i++) { jump_square(*a, *b); JumpMatrix* t = a; a = b; b = t; }
        for (int k = 0; k < 16; k++) {
            g_seqjump[k] = *a;
            jump_square(*a, *b); jump_square(*b, *a);
        }
        delete a; delete b;
        g_seqjump_ready.store(1);
    }
    lock.store(0);
}

static void xorwow_init(uint64_t seed, uint64_t subsequence, XorwowState* state) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    state->d = 6615241u + t1 + t0;
    state->v[0] = 123456789u + t0;
    state->v[1] = 362436069u ^ t0;
    state->v[2] = 521288629u + t1;
    state->v[3] = 88675123u ^ t1;
    state->v[4] = 5783321u + t0;
    uint64_t p = subsequence;
    int k = 0;
    while (p && k < 16) {
        for (unsigned t = 0; t < (p & 3); t++) jump_apply(g_seqjump[k], state->v);
        p >>= 2; k++;
    }
}

// renderer/rng.cuh:8-12
static inline float RandomFloat(XorwowState* rng) {
    constexpr float rmax = 0x1.fffffep-1;
    const float r = static_cast<float>(xorwow_next(rng)) * 0x1p-32f;
    return r < rmax ? r : rmax;
}

// ------------------------------------------------ scene model (scene/scene.cuh, rays.cuh, lights.cuh, texture.cuh)
struct Counters {
    uint64_t closest_rays = 0, shadow_rays = 0, paths = 0;
    uint64_t node_box_tests = 0, node_enters = 0, bvh_interior = 0, bvh_leaves = 0;
    uint64_t tri_tests = 0, tri_hits = 0, sphere_tests = 0, plane_tests = 0, texel_quads = 0;
    void add(const Counters& o) {
        closest_rays += o.closest_rays; shadow_rays += o.shadow_rays; paths += o.paths;
        node_box_tests += o.node_box_tests; node_enters += o.node_enters; bvh_interior += o.bvh_interior;
        bvh_leaves += o.bvh_leaves; tri_tests += o.tri_tests; tri_hits += o.tri_hits;
        sphere_tests += o.sphere_tests; plane_tests += o.plane_tests; texel_quads += o.texel_quads;
    }
};
static thread_local Counters* tl_cnt = nullptr;
#define COUNT(field) do { if (tl_cnt) tl_cnt->field++; } while (0)

struct Color24 { uint8_t r, g, b; };
static inline color ToColor(Color24 c) { constexpr float over256 = 1.0f / 256.0f; return mk(c.r, c.g, c.b) * over256; }
static inline uint8_t ClampToInt(float f) { int v = dev_f2i(f); return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }

struct Texture {
    const Color24* data; unsigned width, height;
    color Eval(float3 const& uvw) const;
    color EvalEnvironment(const float3& dir) const;
};

#define HIT_NONE 0
#define HIT_FRONT 1
#define HIT_BACK 2
#define HIT_FRONT_AND_BACK (HIT_FRONT | HIT_BACK)

struct Node;
struct Light;
struct Hit {
    float3 pos; float z; float3 n; float3 uvw; bool front; const Node* node; bool hitLight; const Light* light;
    int face;   // oracle-only: face index of the last accepted triangle (-1 otherwise)
    Hit() : pos(F3_ZERO), z(BIGFLOAT), n(F3_UP), uvw(F3_ZERO), front(true), node(nullptr), hitLight(false), light(nullptr), face(-1) {}
    color Eval(const Texture* texture) const { return texture->Eval(uvw); }
};
struct Ray { float3 pos, dir; color contribution; };
struct ShadowRay { float3 pos, dir; };
struct SampleInfo { color mult; float prob; float dist; SampleInfo() : mult(BLACK), prob(0.0f), dist(BIGFLOAT) {} };

// scene/rays.cuh:108-187
struct Box {
    float3 pmin, pmax;
    void Init() { pmin = mk(BIGFLOAT, BIGFLOAT, BIGFLOAT); pmax = mk(-BIGFLOAT, -BIGFLOAT, -BIGFLOAT); }
    void operator+=(const float3 pos) {
        if (pmin.x > pos.x) pmin.x = pos.x; if (pmin.y > pos.y) pmin.y = pos.y; if (pmin.z > pos.z) pmin.z = pos.z;
        if (pmax.x < pos.x) pmax.x = pos.x; if (pmax.y < pos.y) pmax.y = pos.y; if (pmax.z < pos.z) pmax.z = pos.z;
    }
    void operator+=(const Box box) { *this += box.pmin; *this += box.pmax; }
    // the Ray and ShadowRay overloads are the same arithmetic (rays.cuh:138-187)
    bool Intersect(const float3& rpos, const float3& rdir, float& dist, const float t_max, float* enter = nullptr) const {
        const float3 inv = mk(1.0f / rdir.x, 1.0f / rdir.y, 1.0f / rdir.z);
        const float3 tLow = (pmin - rpos) * inv;
        const float3 tHigh = (pmax - rpos) * inv;
        const float3 tClose = mk(fminf(tLow.x, tHigh.x), fminf(tLow.y, tHigh.y), fminf(tLow.z, tHigh.z));
        const float3 tFar = mk(fmaxf(tLow.x, tHigh.x), fmaxf(tLow.y, tHigh.y), fmaxf(tLow.z, tHigh.z));
        const float tEnter = fmaxf(tClose.x, fmaxf(tClose.y, tClose.z));
        const float tExit = fminf(tFar.x, fminf(tFar.y, tFar.z));
        if (enter) *enter = tEnter;
        if ((tEnter >= -FLT_EPSILON || tExit >= -FLT_EPSILON) && tEnter <= tExit && tEnter <= t_max) {
            dist = tEnter >= -FLT_EPSILON ? tEnter : tExit;
            return true;
        }
        return false;
    }
};

// renderer/bvh.cuh:13-35
struct BVHNode {
    bool leaf; size_t elemStartIndex, elemCount, leftChild, rightChild, nextNode; Box box;
};

struct Mesh {
    std::vector<float3> v, vn, vt;
    std::vector<uint32_t> f, fn, ft;    // index triples
    size_t nv = 0, nf = 0, nvn = 0, nvt = 0;
    Box box;
    // BVH (renderer/bvh.cuh:37-41)
    std::vector<size_t> elements;
    std::vector<BVHNode> nodes;
    size_t depth = 0;
    bool HasNormals() const { return nvn > 0; }
    bool HasTextureVertices() const { return nvt > 0; }
    float GetElementCenter(size_t elementId, unsigned axis) const;
    Box GetElementBounds(size_t elementId) const;
    void SplitOrFinish(size_t nodeId);
    void ConstructBVH();
    size_t CalculateDepth(size_t nodeId) const;
    bool Traverse(Ray const& ray, Hit& hit, int hitSide) const;
    bool TraverseShadow(ShadowRay const& ray, float tMax, int hitSide) const;
    bool IntersectTriangle(Ray const& ray, Hit& hit, int hitSide, unsigned faceID) const;
    bool IntersectShadowTriangle(ShadowRay const& ray, float tMax, int hitSide, unsigned faceID) const;
};

struct Material {
    const Texture* diffuse; const Texture* specular; float glossiness;
    const Texture* refraction; const Texture* emission; color absorption; float ior;
    bool GenerateSample(float3 const& v, Hit const& hit, float3& dir, XorwowState* rng, SampleInfo& info) const;
    void GetSampleInfo(float3 const& v, Hit const& hit, float3 const& dir, SampleInfo& info) const;
};

enum { OBJ_NONE = 0, OBJ_SPHERE = 1, OBJ_PLANE = 2, OBJ_MESH = 3 };
enum { LIGHT_AMBIENT = 0, LIGHT_DIRECT = 1, LIGHT_POINT = 2 };

struct Light {
    int kind; color intensity; float3 position; float size;
    // scene/lights.cuh:62 - the scale factor is evaluated in double, then narrowed
    color Radiance() const {
        if (kind == LIGHT_POINT) return intensity * (float)(1.0f / (M_PI * size * size));
        return intensity;
    }
    bool Intersect(Ray const& ray, Hit& hit) const;
    bool GenerateSample(float3 const& v, Hit const& hit, float3& dir, XorwowState* rng, SampleInfo& info) const;
};

struct Node {
    size_t childCount; int objectKind; const Mesh* mesh; const Material* material; Box boundingBox; Matrix tm, itm;
    void ToLocal(float3& pos, float3& dir) const { itm.TransformPosition(pos); itm.TransformDirection(dir); }
    void FromLocal(float3& pos, float3& dir) const { tm.TransformPosition(pos); tm.TransformDirection(dir); }
    void FromLocal(Hit& hit) const { tm.TransformPosition(hit.pos); itm.TransformNormal(hit.n); hit.node = this; }
};

struct Camera {
    unsigned width, height; float3 position; float dof; int sRGB; float pixelSize; float3 cX, cY, cZ, topLeftPixel;
    float3 target, up; float fov;
};

struct Scene {
    Camera cam;
    std::vector<Node> nodes;
    std::vector<Light> lights;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<std::vector<Color24>> texels;
    std::vector<Mesh*> meshes;
    const Texture* env = nullptr;
    ~Scene() { for (Mesh* m : meshes) delete m; }
};

// ------------------------------------------------ renderer/bvh.cu:7-167
static inline float SurfaceArea(const Box& box) {
    const float l = box.pmax.x - box.pmin.x; const float w = box.pmax.y - box.pmin.y; const float h = box.pmax.z - box.pmin.z;
    return 2 * l * w + 2 * l * h + 2 * w * h;
}

float Mesh::GetElementCenter(const size_t elementId, const unsigned axis) const {
    float sum = 0.0f;
    for (int i = 0; i < 3; i++) sum += ref(v[f[3 * elements[elementId] + i]], axis);
    return sum * 0.3333333333333333f;
}

Box Mesh::GetElementBounds(const size_t elementId) const {
    Box b; b.Init();
    for (int i = 0; i < 3; i++) b += v[f[3 * elements[elementId] + i]];
    return b;
}

void Mesh::SplitOrFinish(const size_t nodeId) {
    constexpr int binCount = 32;
    constexpr float binSize = 1.0f / static_cast<float>(binCount);
    struct AreaSplitResult { unsigned axis = 0; unsigned bin = 0; float cost = BIGFLOAT; Box lbox{}; Box rbox{}; };

    if (nodes[nodeId].elemCount == 1) { nodes[nodeId].leaf = true; nodes[nodeId].nextNode = nodes.size(); return; }
    const Box box = nodes[nodeId].box;
    const size_t elemStartIndex = nodes[nodeId].elemStartIndex, elemCount = nodes[nodeId].elemCount;
    const float outer = 1.0f / SurfaceArea(box);
    const float3 binDim = (box.pmax - box.pmin) * binSize;
    AreaSplitResult best;
    memset(&best.lbox, 0, sizeof(Box)); memset(&best.rbox, 0, sizeof(Box));
    for (unsigned axis = 0; axis < 3; axis++) {
        for (int bin = 1; bin < binCount; bin++) {
            float splitPoint = ref(box.pmin, axis) + static_cast<float>(bin) * ref(binDim, axis);
            AreaSplitResult result;
            result.axis = axis; result.bin = bin; result.lbox.Init(); result.rbox.Init();
            size_t lcount = 0, rcount = 0;
            for (size_t i = 0; i < elemCount; i++) {
                const float center = GetElementCenter(elemStartIndex + i, axis);
                const Box bounds = GetElementBounds(elemStartIndex + i);
                if (center <= splitPoint) { result.lbox += bounds; lcount++; }
                else { result.rbox += bounds; rcount++; }
            }
            const float larea = SurfaceArea(result.lbox);
            const float rarea = SurfaceArea(result.rbox);
            const float lcost = (larea * outer) * lcount;
            const float rcost = (rarea * outer) * rcount;
            result.cost = 1 + lcost + rcost;
            if (result.cost < best.cost) best = result;
        }
    }
    if (best.cost > elemCount && elemCount <= 8) { nodes[nodeId].leaf = true; nodes[nodeId].nextNode = nodes.size(); return; }
    if (nodes.size() > 4 * nf + 64) { nodes[nodeId].leaf = true; nodes[nodeId].nextNode = nodes.size(); depth = 1000; return; }   // runaway guard (degenerate input)

    nodes[nodeId].leaf = false;
    size_t i = 0, j = elemCount;
    const float splitPoint = ref(box.pmin, best.axis) + static_cast<float>(best.bin) * ref(binDim, best.axis);
    while (i < j) {
        const float center = GetElementCenter(elemStartIndex + i, best.axis);
        if (center <= splitPoint) i++;
        else { j--; const size_t t = elements[elemStartIndex + i]; elements[elemStartIndex + i] = elements[elemStartIndex + j]; elements[elemStartIndex + j] = t; }
    }
    const size_t left = nodes.size();
    nodes[nodeId].leftChild = left;
    nodes.push_back(BVHNode{false, elemStartIndex, i, 0, 0, 0, best.lbox});
    SplitOrFinish(left);
    const size_t right = nodes.size();
    nodes[nodeId].rightChild = right;
    nodes.push_back(BVHNode{false, elemStartIndex + i, elemCount - i, 0, 0, 0, best.rbox});
    SplitOrFinish(right);
    nodes[nodeId].nextNode = nodes.size();
}

size_t Mesh::CalculateDepth(const size_t nodeId) const {
    const BVHNode& node = nodes[nodeId];
    if (node.leaf) return 1;
    const size_t a = CalculateDepth(node.leftChild), b = CalculateDepth(node.rightChild);
    return 1 + (a > b ? a : b);
}

void Mesh::ConstructBVH() {
    elements.resize(nf);
    for (size_t i = 0; i < nf; i++) elements[i] = i;
    nodes.clear();
    nodes.reserve(2 * nf + 1);
    nodes.push_back(BVHNode{false, 0, nf, 0, 0, 0, box});
    SplitOrFinish(0);
    if (depth != 1000) depth = CalculateDepth(0);
}

// ------------------------------------------------ scene/texture.cu:14-63
static float3 TileClamp(float3 const& uvw) {
    float3 u;
    u.x = uvw.x - dev_f2i(uvw.x);
    u.y = uvw.y - dev_f2i(uvw.y);
    u.z = uvw.z - dev_f2i(uvw.z);
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}

color Texture::Eval(float3 const& uvw) const {
    if (width + height == 0) return BLACK;
    COUNT(texel_quads);
    float3 u = TileClamp(uvw);
    const float x = width * u.x;
    const float y = height * u.y;
    int ix = dev_f2i(x);
    int iy = dev_f2i(y);
    const float fx = x - ix;
    const float fy = y - iy;
    // the comparisons and divisions below mix int with unsigned, as in the reference
    if (ix < 0) ix -= (ix / width - 1) * width;
    if ((unsigned)ix >= width) ix -= ((unsigned)ix / width) * width;
    int ixp = ix + 1;
    if ((unsigned)ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if ((unsigned)iy >= height) iy -= ((unsigned)iy / height) * height;
    int iyp = iy + 1;
    if ((unsigned)iyp >= height) iyp -= height;
    return ToColor(data[iy * width + ix]) * ((1 - fx) * (1 - fy)) +
           ToColor(data[iy * width + ixp]) * (fx * (1 - fy)) +
           ToColor(data[iyp * width + ix]) * ((1 - fx) * fy) +
           ToColor(data[iyp * width + ixp]) * (fx * fy);
}

color Texture::EvalEnvironment(const float3& dir) const {
    const float3 d = asNorm(dir);
    const float phi = m_atan2f(d.x, d.z);
    const float theta = m_acosf(fminf(fmaxf(d.y, -1.0f), 1.0f));
    const float u = (phi + M_PI) * (0.5f / M_PI);     // double arithmetic, narrowed on assignment
    const float v = theta * (1.0f / M_PI);
    return Eval(mk(u, v, 0.0f));
}

// ------------------------------------------------ math/vector.cu:3-52
static void orthonormals(const float3 n, float3& x, float3& y) {
    if (fabsf(n.x) > fabsf(n.z)) x = mk(-n.y, n.x, 0.0f);
    else x = mk(0.0f, -n.z, n.y);
    doNorm(x);
    y = cross(n, x);
}

static float3 reflect(const float3 incident, const float3 normal) { return 2 * (incident % normal) * normal - incident; }

static float3 transmit(const float3 incident, const float3 normal, const float outerIor, const float innerIor, bool& totalInnerReflection) {
    const float cos_theta = incident % normal;
    const float eta = outerIor / innerIor;
    const float cos_phi_squared = 1 - eta * eta * (1 - cos_theta * cos_theta);
    if (cos_phi_squared < 0) { totalInnerReflection = true; return reflect(incident, normal); }
    totalInnerReflection = false;
    return -normal * sqrtf(cos_phi_squared) + eta * (cos_theta * normal - incident);
}

static float3 glossyNormal(const float3 normal, const float glossiness, XorwowState* rng) {
    const float x = RandomFloat(rng);
    const float phi = RandomFloat(rng) * M_PI * 2.0f;      // double product, narrowed
    const float cos_theta = m_powf(1 - x, 1.0f / (glossiness + 1.0f));
    float3 u, v;
    orthonormals(normal, u, v);
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    const float cos_phi = m_cosf(phi);
    const float sin_phi = m_sinf(phi);
    return normal * cos_theta + u * sin_theta * cos_phi + v * sin_theta * sin_phi;
}

// ------------------------------------------------ scene/objects.cu:4-366
#define EPS 0.000001f
#define OBJ_OVERPI 0.318309f

static bool SphereIntersectRay(Ray const& ray, Hit& hit, const int hitSide) {
    if (hitSide == HIT_NONE) return false;
    COUNT(sphere_tests);
    if (ray.dir % ray.pos >= 0) return false;
    const float a = ray.dir % ray.dir;
    const float b = ray.dir % ray.pos * 2;
    const float c = ray.pos % ray.pos - 1;
    const float determinant = b * b - 4 * a * c;
    if (determinant < 0) return false;
    const float sqrtDeterminant = sqrtf(determinant);
    const float front = (-b - sqrtDeterminant) / (2 * a);
    const float back = (-b + sqrtDeterminant) / (2 * a);
    if (hitSide & HIT_FRONT) {
        if (front >= 0) {
            if (front >= hit.z) return false;
            hit.z = front;
            hit.pos = ray.pos + front * ray.dir;
            hit.n = hit.pos;
            hit.uvw = mk(0.5f * OBJ_OVERPI * m_atan2f(hit.pos.y, hit.pos.x) + 0.5f,
                         OBJ_OVERPI * m_asinf(fminf(fmaxf(hit.pos.z, -1.0f), 1.0f)) + 0.5f, 0);
            hit.front = true; hit.hitLight = false; hit.face = -1;
            return true;
        }
    }
    if (hitSide & HIT_BACK) {
        if (back < 0) return false;
        if (back >= hit.z) return false;
        hit.z = back;
        hit.pos = ray.pos + back * ray.dir;
        hit.n = hit.pos;
        hit.uvw = mk(0.5f * OBJ_OVERPI * m_atan2f(hit.pos.y, hit.pos.x) + 0.5f,
                     OBJ_OVERPI * m_asinf(fminf(fmaxf(hit.pos.z, -1.0f), 1.0f)) + 0.5f, 0);
        hit.front = false; hit.hitLight = false; hit.face = -1;
        return true;
    }
    return false;
}

static bool SphereIntersectShadowRay(const ShadowRay& ray, const float tMax, const int hitSide) {
    if (hitSide == HIT_NONE) return false;
    COUNT(sphere_tests);
    if (ray.dir % ray.pos >= 0) return false;
    const float a = ray.dir % ray.dir;
    const float b = ray.dir % ray.pos * 2;
    const float c = ray.pos % ray.pos - 1;
    const float determinant = b * b - 4 * a * c;
    if (determinant < 0) return false;
    const float sqrtDeterminant = sqrtf(determinant);
    const float front = (-b - sqrtDeterminant) / (2 * a);
    const float back = (-b + sqrtDeterminant) / (2 * a);
    if (hitSide & HIT_FRONT && front >= 0 && front <= tMax) return true;
    if (hitSide & HIT_BACK && back >= 0 && back <= tMax) return true;
    return false;
}

static bool PlaneIntersectRay(Ray const& ray, Hit& hit, const int hitSide) {
    if (hitSide == HIT_NONE) return false;
    COUNT(plane_tests);
    if (ray.dir.z == 0) return false;
    const float t = -ray.pos.z / ray.dir.z;
    if (t < 0 || t >= hit.z) return false;
    if (!(ray.dir.z < 0 && (hitSide & HIT_FRONT)) && !(ray.dir.z > 0 && (hitSide & HIT_BACK))) return false;
    const float3 hitPos = ray.pos + t * ray.dir;
    if (fabsf(hitPos.x) > 1 || fabsf(hitPos.y) > 1) return false;
    hit.z = t;
    hit.pos = hitPos;
    hit.n = F3_FORWARD;
    hit.uvw = 0.5f * (hitPos + mk(1, 1, 0));
    hit.front = ray.dir.z < 0;
    hit.hitLight = false; hit.face = -1;
    return true;
}

static bool PlaneIntersectShadowRay(const ShadowRay& ray, const float tMax, const int hitSide) {
    if (hitSide == HIT_NONE) return false;
    COUNT(plane_tests);
    if (ray.dir.z == 0) return false;
    const float t = -ray.pos.z / ray.dir.z;
    if (t < 0 || t >= tMax) return false;
    if (!(ray.dir.z < 0 && (hitSide & HIT_FRONT)) && !(ray.dir.z > 0 && (hitSide & HIT_BACK))) return false;
    const float3 hitPos = ray.pos + t * ray.dir;
    if (fabsf(hitPos.x) > 1 || fabsf(hitPos.y) > 1) return false;
    return true;
}

// NOT the reference's rule, and off unless a test switches it on (oracle_set_node_cull): the node scan culls a node
// by the distance at which the ray ENTERS the node's box.  The reference culls by the distance its box test reports,
// which for a ray that starts inside the box is the way out (rays.cuh:156, trace.cu:59): a node around the ray's origin
// is skipped whenever an earlier node of the list was hit nearer than that way out, whatever the node itself holds,
// and the BVH walk inside a mesh drops boxes around the origin by the same measure (objects.cu:190-192), so what a
// ray finds depends on the order in which the nodes are visited.
// The switch states the one rule by which the product's opt-in BVH over the nodes departs from the reference on
// purpose, so that a test can tell that departure from a defect.
static std::atomic<int> g_cull_by_entry(0);

#define BVH_MAX_DEPTH 64

bool Mesh::Traverse(Ray const& ray, Hit& hit, const int hitSide) const {
    struct StackEntry { size_t node; float minT; };
    StackEntry stack[BVH_MAX_DEPTH + 2];
    size_t stackTop = 0;
    stack[stackTop++] = StackEntry{0, 0};
    bool hitAnything = false;
    while (stackTop > 0) {
        const StackEntry top = stack[--stackTop];
        if (hit.z < top.minT) continue;
        if (nodes[top.node].leaf) {
            COUNT(bvh_leaves);
            const size_t* elems = elements.data() + nodes[top.node].elemStartIndex;
            const size_t triCount = nodes[top.node].elemCount;
            for (size_t i = 0; i < triCount; i++)
                if (IntersectTriangle(ray, hit, hitSide, (unsigned)elems[i])) hitAnything = true;
            continue;
        }
        COUNT(bvh_interior);
        const size_t leftID = nodes[top.node].leftChild;
        const size_t rightID = nodes[top.node].rightChild;
        float tLeft = BIGFLOAT, tRight = BIGFLOAT;
        float eLeft = 0, eRight = 0;
        const bool hitLeft = nodes[leftID].box.Intersect(ray.pos, ray.dir, tLeft, BIGFLOAT, &eLeft);
        const bool hitRight = nodes[rightID].box.Intersect(ray.pos, ray.dir, tRight, BIGFLOAT, &eRight);
        if (g_cull_by_entry.load(std::memory_order_relaxed)) {      // (not the reference's rule: see g_cull_by_entry)
            if (hitLeft) tLeft = eLeft;
            if (hitRight) tRight = eRight;
        }
        if (!hitLeft && !hitRight) continue;
        const size_t closeID = tLeft <= tRight ? leftID : rightID;
        const size_t farID = tLeft <= tRight ? rightID : leftID;
        const float tClose = fminf(tLeft, tRight);
        const float tFar = fmaxf(tLeft, tRight);
        stack[stackTop++] = StackEntry{farID, tFar};
        stack[stackTop++] = StackEntry{closeID, tClose};
    }
    return hitAnything;
}

bool Mesh::TraverseShadow(ShadowRay const& ray, const float tMax, const int hitSide) const {
    size_t stack[BVH_MAX_DEPTH + 2];
    size_t stackTop = 0;
    stack[stackTop++] = 0;
    while (stackTop > 0) {
        const size_t top = stack[--stackTop];
        if (nodes[top].leaf) {
            COUNT(bvh_leaves);
            const size_t* elems = elements.data() + nodes[top].elemStartIndex;
            const size_t triCount = nodes[top].elemCount;
            for (size_t i = 0; i < triCount; i++)
                if (IntersectShadowTriangle(ray, tMax, hitSide, (unsigned)elems[i])) return true;
            continue;
        }
        COUNT(bvh_interior);
        const size_t leftID = nodes[top].leftChild;
        const size_t rightID = nodes[top].rightChild;
        float tLeft = BIGFLOAT, tRight = BIGFLOAT;
        const bool hitLeft = nodes[leftID].box.Intersect(ray.pos, ray.dir, tLeft, tMax);
        const bool hitRight = nodes[rightID].box.Intersect(ray.pos, ray.dir, tRight, tMax);
        if (!hitLeft && !hitRight) continue;
        const size_t closeID = tLeft <= tRight ? leftID : rightID;
        const size_t farID = tLeft <= tRight ? rightID : leftID;
        stack[stackTop++] = farID;
        stack[stackTop++] = closeID;
    }
    return false;
}

bool Mesh::IntersectTriangle(Ray const& ray, Hit& hit, const int hitSide, const unsigned faceID) const {
    COUNT(tri_tests);
    const float3 p0 = v[f[3 * faceID]];
    const float3 p1 = v[f[3 * faceID + 1]];
    const float3 p2 = v[f[3 * faceID + 2]];
    const float3 edge1 = p1 - p0;
    const float3 edge2 = p2 - p0;
    const float3 rayXEdge2 = cross(ray.dir, edge2);
    const float determinant = edge1 % rayXEdge2;
    if (determinant > -EPS && determinant < EPS) return false;
    const float inverseDeterminant = 1 / determinant;
    const float3 offset = ray.pos - p0;
    const float b1 = inverseDeterminant * (offset % rayXEdge2);
    if (b1 <= EPS || b1 >= 1 + EPS) return false;
    const float3 offsetXEdge1 = cross(offset, edge1);
    const float b2 = inverseDeterminant * (ray.dir % offsetXEdge1);
    if (b2 <= -EPS || (b1 + b2) >= 1 + EPS) return false;
    const float t = inverseDeterminant * (edge2 % offsetXEdge1);
    if (t <= EPS || t >= hit.z + EPS) return false;
    const float ndot = cross(edge1, edge2) % ray.dir;
    if (!(ndot < EPS && (hitSide & HIT_FRONT)) && !(ndot > -EPS && (hitSide & HIT_BACK))) return false;
    COUNT(tri_hits);
    const float b0 = 1 - b2 - b1;
    const float3 barycentric = mk(b0, b1, b2);
    const float3 gn = asNorm(cross(edge1, edge2));
    const float3 n = !HasNormals() ? gn : vn[fn[3 * faceID]] * barycentric.x + vn[fn[3 * faceID + 1]] * barycentric.y + vn[fn[3 * faceID + 2]] * barycentric.z;
    const float3 uvw = !HasTextureVertices() ? F3_ZERO : vt[ft[3 * faceID]] * barycentric.x + vt[ft[3 * faceID + 1]] * barycentric.y + vt[ft[3 * faceID + 2]] * barycentric.z;
    hit.z = t;
    hit.n = n;
    hit.pos = ray.pos + t * ray.dir;
    hit.uvw = uvw;
    hit.front = ndot < 0;
    hit.hitLight = false;
    hit.face = (int)faceID;
    return true;
}

bool Mesh::IntersectShadowTriangle(ShadowRay const& ray, const float tMax, const int hitSide, const unsigned faceID) const {
    COUNT(tri_tests);
    const float3 p0 = v[f[3 * faceID]];
    const float3 p1 = v[f[3 * faceID + 1]];
    const float3 p2 = v[f[3 * faceID + 2]];
    const float3 edge1 = p1 - p0;
    const float3 edge2 = p2 - p0;
    const float3 rayXEdge2 = cross(ray.dir, edge2);
    const float determinant = edge1 % rayXEdge2;
    if (determinant > -EPS && determinant < EPS) return false;
    const float inverseDeterminant = 1 / determinant;
    const float3 offset = ray.pos - p0;
    const float b1 = inverseDeterminant * (offset % rayXEdge2);
    if (b1 <= EPS || b1 >= 1 + EPS) return false;
    const float3 offsetXEdge1 = cross(offset, edge1);
    const float b2 = inverseDeterminant * (ray.dir % offsetXEdge1);
    if (b2 <= -EPS || (b1 + b2) >= 1 + EPS) return false;
    const float t = inverseDeterminant * (edge2 % offsetXEdge1);
    if (t <= EPS || t >= tMax + EPS) return false;
    const float ndot = cross(edge1, edge2) % ray.dir;
    if (!(ndot < EPS && (hitSide & HIT_FRONT)) && !(ndot > -EPS && (hitSide & HIT_BACK))) return false;
    return true;
}

static bool ObjIntersect(const Node* node, Ray const& ray, Hit& hit, int hitSide) {
    switch (node->objectKind) {
        case OBJ_SPHERE: return SphereIntersectRay(ray, hit, hitSide);
        case OBJ_PLANE: return PlaneIntersectRay(ray, hit, hitSide);
        case OBJ_MESH: if (hitSide == HIT_NONE) return false; return node->mesh->Traverse(ray, hit, hitSide);
        default: return false;
    }
}

static bool ObjIntersectShadow(const Node* node, ShadowRay const& ray, float tMax, int hitSide) {
    switch (node->objectKind) {
        case OBJ_SPHERE: return SphereIntersectShadowRay(ray, tMax, hitSide);
        case OBJ_PLANE: return PlaneIntersectShadowRay(ray, tMax, hitSide);
        case OBJ_MESH: if (hitSide == HIT_NONE) return false; return node->mesh->TraverseShadow(ray, tMax, hitSide);
        default: return false;
    }
}

// ------------------------------------------------ scene/lights.cu:7-89
static const float L_OVERPI = 1.0f / M_PI;
static const float F_PI = static_cast<float>(M_PI);

bool Light::GenerateSample(float3 const& v, Hit const& hit, float3& dir, XorwowState* rng, SampleInfo& info) const {
    (void)v;
    if (kind != LIGHT_POINT) return false;    // ambient / directional: no draws, outputs untouched (lights.cuh:36,51)
    const float3 L = asNorm(hit.pos - position);
    const float x = RandomFloat(rng);
    const float phi = RandomFloat(rng) * 2 * F_PI;
    const float cos_theta = 1 - x;
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    float3 a, b;
    orthonormals(L, a, b);
    const float3 n = L * cos_theta + a * sin_theta * m_cosf(phi) + b * sin_theta * m_sinf(phi);
    const float3 target = position + n * size;
    dir = asNorm(target - hit.pos);
    const float d2 = lengthsq(target - hit.pos);
    const float cos_theta_l = dir % -asNorm(target - position);
    info.prob = d2 / (2 * F_PI * size * size * cos_theta_l);
    info.mult = Radiance();
    info.dist = sqrtf(d2);
    return true;
}

bool Light::Intersect(Ray const& ray, Hit& hit) const {
    if (kind != LIGHT_POINT) return false;
    const float3 o = ray.pos - position;
    const float a = ray.dir % ray.dir;
    const float b = (2 * ray.dir) % o;
    const float c = o % o - size * size;
    const float determinant = b * b - 4 * a * c;
    if (determinant < 0) return false;
    const float sqrtDeterminant = sqrtf(determinant);
    const float front = (-b - sqrtDeterminant) / (2 * a);
    const float back = (-b + sqrtDeterminant) / (2 * a);
    if (front >= 0) {
        if (front >= hit.z) return false;
        hit.z = front;
        const float3 normPos = asNorm(o);
        hit.pos = ray.pos + front * ray.dir;
        hit.n = normPos;
        hit.uvw = mk(0.5f * L_OVERPI * m_atan2f(normPos.y, normPos.x) + 0.5f,
                     L_OVERPI * m_asinf(fminf(fmaxf(normPos.z, -1.0f), 1.0f)) + 0.5f, 0);
        hit.front = true; hit.hitLight = true; hit.face = -1;
        return true;
    }
    if (back < 0) return false;
    if (back >= hit.z) return false;
    hit.z = back;
    const float3 normPos = asNorm(hit.pos);      // reads the previous hit position (lights.cu:78)
    hit.pos = ray.pos + back * ray.dir;
    hit.uvw = mk(0.5f * L_OVERPI * m_atan2f(normPos.y, normPos.x) + 0.5f,
                 L_OVERPI * m_asinf(fminf(fmaxf(normPos.z, -1.0f), 1.0f)) + 0.5f, 0);
    hit.n = normPos;
    hit.front = false; hit.hitLight = true; hit.face = -1;
    return true;
}

// ------------------------------------------------ renderer/shading.cu:8-150
static const float S_OVERPI = 1.0f / M_PI;
static inline color cexp(color c) { return mk(m_expf(c.x), m_expf(c.y), m_expf(c.z)); }

bool Material::GenerateSample(float3 const& v, Hit const& hit, float3& dir, XorwowState* rng, SampleInfo& info) const {
    const color kD = hit.Eval(diffuse);
    const color kS = hit.Eval(specular);
    const color kT = hit.Eval(refraction);
    constexpr float oneThird = 1.0f / 3.0f;
    float pD = (kD.x + kD.y + kD.z) * oneThird;
    float pS = (kS.x + kS.y + kS.z) * oneThird;
    float pT = (kT.x + kT.y + kT.z) * oneThird;
    if (pD + pS + pT > MAX_SURVIVAL) {
        const float normalization = MAX_SURVIVAL / (pD + pS + pT);
        pD *= normalization; pS *= normalization; pT *= normalization;
    }
    color kA = WHITE;
    if (!hit.front) kA = cexp(absorption * -hit.z);
    const float p = RandomFloat(rng);
    if (p < pD) {
        const float x = RandomFloat(rng);
        const float phi = RandomFloat(rng) * 2 * F_PI;
        const float cos_theta = 1 - x;
        const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
        float3 u, w;
        orthonormals(hit.n, u, w);
        dir = hit.n * cos_theta + u * sin_theta * m_cosf(phi) + w * sin_theta * m_sinf(phi);
        info.prob = pD * S_OVERPI;
        info.mult = kA * kD * S_OVERPI * (hit.n % dir);
        return true;
    }
    const float3 gln = glossyNormal(hit.n, glossiness, rng);
    if (p < pD + pS) {
        dir = reflect(v, gln);
        info.prob = pS * (glossiness + 1) * 0.125f * S_OVERPI * m_powf(gln % hit.n, glossiness);
        info.mult = kA * kS * (glossiness + 2) * 0.125f * S_OVERPI * m_powf(gln % hit.n, glossiness);
        return true;
    }
    if (p < pD + pS + pT) {
        const float eta_i = hit.front ? 1 : ior;
        const float eta_o = hit.front ? ior : 1;
        const float3 h = hit.front ? gln : -gln;
        bool tir = false;
        dir = transmit(v, h, eta_i, eta_o, tir);
        const float f_0 = m_powf((eta_i - eta_o) / (eta_i + eta_o), 2);
        const float cos_theta = v % h;
        float f_theta = f_0 + (1 - f_0) * m_powf(1 - cos_theta, 5);
        f_theta = fminf(fmaxf(f_theta, 0.0f), 1.0f);
        if (!tir) {
            if (RandomFloat(rng) >= f_theta) {
                info.prob = pT * (1 - f_theta);
                info.mult = kA * kT * (1 - f_theta);
            } else {
                dir = reflect(v, h);
                info.prob = pT * f_theta;
                info.mult = kA * kT * f_theta;
            }
        } else {
            info.prob = pT;
            info.mult = kA * kT;
        }
        return true;
    }
    info.prob = 1 - pT - pS - pD;
    info.mult = BLACK;
    return false;
}

void Material::GetSampleInfo(float3 const& v, Hit const& hit, float3 const& dir, SampleInfo& info) const {
    const color kD = hit.Eval(diffuse);
    const color kS = hit.Eval(specular);
    const color kT = hit.Eval(refraction);
    constexpr float oneThird = 1.0f / 3.0f;
    float pD = (kD.x + kD.y + kD.z) * oneThird;
    float pS = (kS.x + kS.y + kS.z) * oneThird;
    float pT = (kT.x + kT.y + kT.z) * oneThird;
    if (pD + pS + pT > MAX_SURVIVAL) {
        const float normalization = MAX_SURVIVAL / (pD + pS + pT);
        pD *= normalization; pS *= normalization; pT *= normalization;
    }
    if (hit.n % dir >= 0) {
        info.prob += pD * (hit.n % dir > 0 ? S_OVERPI : 0);
        info.mult += kD * S_OVERPI * (hit.n % dir);
        const float3 rn = asNorm(v + dir);
        info.prob += pS * (glossiness + 1) * 0.5f * S_OVERPI * m_powf(hit.n % rn, glossiness + 1);
        info.mult += kS * (glossiness + 2) * 0.125f * S_OVERPI * m_powf(hit.n % rn, glossiness);
    }
}

// ------------------------------------------------ renderer/trace.cu:44-196
static bool TraceThroughScene(const Scene& sc, Ray& ray, Hit& hit) {
    COUNT(closest_rays);
    ray.pos += ray.dir * BIAS;
    const int hitSide = HIT_FRONT_AND_BACK;     // IsPrimary() is never true: bounce stays 0 (rays.cuh:87, trace.cu:38)
    bool hitAnything = false;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        const Node* node = &sc.nodes[i];
        if (node->objectKind != OBJ_NONE) {
            float boxZ = BIGFLOAT;
            COUNT(node_box_tests);
            float enterZ = 0;
            const bool hitBox = node->boundingBox.Intersect(ray.pos, ray.dir, boxZ, BIGFLOAT, &enterZ);
            if (hitBox && g_cull_by_entry.load(std::memory_order_relaxed)) boxZ = enterZ;
            if (hitBox && boxZ < hit.z) {
                COUNT(node_enters);
                node->ToLocal(ray.pos, ray.dir);
                if (ObjIntersect(node, ray, hit, hitSide)) {
                    hitAnything = true;
                    node->FromLocal(hit);
                }
                node->FromLocal(ray.pos, ray.dir);
            }
        }
    }
    for (size_t i = 0; i < sc.lights.size(); i++) {
        const Light* thisLight = &sc.lights[i];
        if (thisLight->Intersect(ray, hit)) { hitAnything = true; hit.light = thisLight; }
    }
    return hitAnything;
}

static bool TraceShadowRay(const Scene& sc, ShadowRay& ray, const float3 n, const float tMax, const int hitSide = HIT_FRONT_AND_BACK) {
    COUNT(shadow_rays);
    ray.pos += ray.dir * BIAS + n * BIAS;
    unsigned skip = 0;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        const Node* node = &sc.nodes[i];
        if (skip > 0) { skip--; skip += node->childCount; continue; }
        if (node->objectKind != OBJ_NONE) {
            float dist;
            COUNT(node_box_tests);
            const bool hitBox = node->boundingBox.Intersect(ray.pos, ray.dir, dist, tMax);
            if (!hitBox) { skip = node->childCount; continue; }
            COUNT(node_enters);
            node->ToLocal(ray.pos, ray.dir);
            if (ObjIntersectShadow(node, ray, tMax, hitSide)) return true;
            node->FromLocal(ray.pos, ray.dir);
        }
    }
    return false;
}

static void TracePath(const Scene& sc, Ray const& origin, color* target, float3* normal, color* albedo, XorwowState* rng) {
    COUNT(paths);
    Ray ray = origin;
    for (;;) {                                   // bounce is never incremented (trace.cu:85)
        const float3 v = -asNorm(ray.dir);
        Hit hit;
        const bool hitAnything = TraceThroughScene(sc, ray, hit);
        if (!hitAnything) {
            *target += ray.contribution * sc.env->EvalEnvironment(ray.dir);
            *normal += ray.contribution * asNorm(ray.dir);
            *albedo += ray.contribution * sc.env->EvalEnvironment(ray.dir);
            break;
        }
        if (hit.hitLight) {
            const Light* light = hit.light;
            const color rad = light->Radiance();
            *target += ray.contribution * rad;
            *normal += ray.contribution * hit.n;
            *albedo += ray.contribution * rad;
            break;
        }
        const float lightPick = RandomFloat(rng);
        const Material* mtl = hit.node->material;
        *normal += ray.contribution * hit.n;
        *albedo += ray.contribution * hit.Eval(mtl->diffuse) + hit.Eval(mtl->emission);
        *target += ray.contribution * hit.Eval(mtl->emission);

        // Next-event estimation.  With no lights the reference dereferences a null
        // light (trace.cu:113-114); defined here as "no light sample".  Ambient and
        // directional lights generate no sample and weigh zero (lights.cuh:36,51), so
        // their shadow query is result-neutral and is not traced or counted.
        if (!sc.lights.empty()) {
            const int lightId = dev_f2i(lightPick * static_cast<float>(sc.lights.size()));
            const Light* light = &sc.lights[lightId];
            float3 lDir = F3_ZERO;
            SampleInfo lInfo;
            if (light->GenerateSample(v, hit, lDir, rng, lInfo)) {
                SampleInfo brdf;
                mtl->GetSampleInfo(v, hit, lDir, brdf);
                ShadowRay shadowRay = {hit.pos, lDir};
                if (!TraceShadowRay(sc, shadowRay, hit.n, lInfo.dist)) {
                    const float probDenom = lInfo.prob * lInfo.prob + brdf.prob * brdf.prob;
                    if (probDenom > F_EPS && lInfo.prob > F_EPS) {
                        const float powerOverProb = lInfo.prob / probDenom;
                        *target += ray.contribution * lInfo.mult * brdf.mult * powerOverProb;
                    }
                }
            }
        }

        float3 nDir;
        SampleInfo nInfo;
        if (!mtl->GenerateSample(v, hit, nDir, rng, nInfo)) break;
        ray.pos = hit.pos;
        ray.dir = nDir;
        if (nInfo.prob <= F_EPS) break;
        ray.contribution *= nInfo.mult * (1.0f / nInfo.prob);
    }
}

// renderer/trace.cu:7-41 for one pixel
static void camera_ray(const Scene& sc, unsigned pX, unsigned pY, XorwowState* rng, int samples, Ray& ray) {
    const Camera& c = sc.cam;
    const float3 pixelCoords = c.topLeftPixel + c.pixelSize * (pX * c.cX - pY * c.cY);
    const float invertedSampleCount = 1.0f / static_cast<float>(samples);
    const float dof_r = sqrtf(RandomFloat(rng));
    const float dof_theta = RandomFloat(rng) * 2 * F_PI;
    const float dof_x = dof_r * m_cosf(dof_theta);
    const float dof_y = dof_r * m_sinf(dof_theta);
    const float3 rayStartPos = c.position + c.dof * (dof_x * c.cX - dof_y * c.cY);
    const float aa_x = RandomFloat(rng) - 0.5f;
    const float aa_y = RandomFloat(rng) - 0.5f;
    const float3 rayEndPos = pixelCoords + c.pixelSize * (aa_x * c.cX - aa_y * c.cY);
    ray.pos = rayStartPos;
    ray.dir = rayEndPos - rayStartPos;
    ray.contribution = WHITE * invertedSampleCount;
}

static void trace_pixel(const Scene& sc, unsigned pX, unsigned pY, int passId, int samples, float* results, float* normals, float* albedos) {
    const unsigned pI = pY * sc.cam.width + pX;
    XorwowState rng;
    xorwow_init(pI, (uint64_t)passId, &rng);
    color target = mk(results[3 * pI], results[3 * pI + 1], results[3 * pI + 2]);
    float3 normal = mk(normals[3 * pI], normals[3 * pI + 1], normals[3 * pI + 2]);
    color albedo = mk(albedos[3 * pI], albedos[3 * pI + 1], albedos[3 * pI + 2]);
    for (int i = 0; i < samples; i++) {
        Ray ray;
        camera_ray(sc, pX, pY, &rng, samples, ray);
        TracePath(sc, ray, &target, &normal, &albedo, &rng);
    }
    results[3 * pI] = target.x; results[3 * pI + 1] = target.y; results[3 * pI + 2] = target.z;
    normals[3 * pI] = normal.x; normals[3 * pI + 1] = normal.y; normals[3 * pI + 2] = normal.z;
    albedos[3 * pI] = albedo.x; albedos[3 * pI + 1] = albedo.y; albedos[3 * pI + 2] = albedo.z;
}

// ------------------------------------------------ math/color.cu:7-23
static inline float LinearTosRGB(const float ch) { return ch < 0.0031308f ? ch * 12.92f : m_powf(ch, 0.41666f) * 1.055f - 0.055f; }

// ------------------------------------------------ camera frame (lib/xmlload.cu:324-354)
static void camera_derive(Camera& c) {
    const float DEG2RAD = 0.017453292519943296f;
    const float aspectRatio = (float)c.width / (float)c.height;
    const float fovRad = c.fov * DEG2RAD;
    const float focaldist = length(c.target - c.position);
    c.cZ = asNorm(c.target - c.position);
    c.cX = cross(c.cZ, asNorm(c.up));
    c.cY = cross(c.cX, c.cZ);
    const float planeHeight = 2 * focaldist * m_tanf(fovRad * 0.5f);
    const float3 plane = mk(planeHeight * aspectRatio, planeHeight, focaldist);
    c.pixelSize = planeHeight / (float)c.height;
    const float3 planeCenter = c.position + focaldist * c.cZ;
    const float3 topLeftCorner = planeCenter - (plane.x * 0.5f * c.cX) + (plane.y * 0.5f * c.cY);
    c.topLeftPixel = topLeftCorner + c.pixelSize * 0.5f * (c.cX - c.cY);
}

// ------------------------------------------------ flat scene file ("CRTSCN01") reader
namespace {

struct FileCamera {
    uint32_t width, height; float position[3]; float dof; int32_t srgb; float pixel_size;
    float cX[3], cY[3], cZ[3]; float top_left_pixel[3]; float target[3], up[3], fov;
};
struct FileNode { uint32_t child_count; int32_t object_kind, mesh_id, material_id; float box_min[3], box_max[3]; float tm[12], itm[12]; };
struct FileMaterial { int32_t diffuse, specular, refraction, emission; float glossiness, ior; float absorption[3]; };
struct FileLight { int32_t kind; float intensity[3]; float vec[3]; float size; };

bool read_blob(FILE* fp, std::vector<uint8_t>& out) {
    uint64_t hdr[2];
    if (fread(hdr, sizeof hdr, 1, fp) != 1) return false;
    if (hdr[0] > (1ull << 36) || hdr[1] > (1ull << 36)) return false;
    std::vector<uint8_t> comp(hdr[1]);
    if (hdr[1] && fread(comp.data(), 1, hdr[1], fp) != hdr[1]) return false;
    out.resize(hdr[0]);
    if (hdr[0]) {
        uLongf rawlen = (uLongf)hdr[0];
        if (uncompress(out.data(), &rawlen, comp.data(), (uLong)hdr[1]) != Z_OK || rawlen != hdr[0]) return false;
    }
    return true;
}

float3 f3of(const float* p) { return mk(p[0], p[1], p[2]); }

}  // namespace

static Scene* load_scene_file(const char* path) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return nullptr;
    Scene* sc = new Scene;
    auto fail = [&]() { fclose(fp); delete sc; return (Scene*)nullptr; };
    char magic[8];
    uint32_t counts[5]; int32_t env;
    if (fread(magic, 8, 1, fp) != 1 || memcmp(magic, "CRTSCN01", 8) != 0) return fail();
    if (fread(counts, sizeof counts, 1, fp) != 1 || fread(&env, 4, 1, fp) != 1) return fail();
    std::vector<uint8_t> blob;
    if (!read_blob(fp, blob) || blob.size() != sizeof(FileCamera)) return fail();
    FileCamera fc; memcpy(&fc, blob.data(), sizeof fc);
    Camera& c = sc->cam;
    c.width = fc.width; c.height = fc.height; c.position = f3of(fc.position); c.dof = fc.dof; c.sRGB = fc.srgb;
    c.pixelSize = fc.pixel_size; c.cX = f3of(fc.cX); c.cY = f3of(fc.cY); c.cZ = f3of(fc.cZ);
    c.topLeftPixel = f3of(fc.top_left_pixel); c.target = f3of(fc.target); c.up = f3of(fc.up); c.fov = fc.fov;
    std::vector<FileNode> fnodes(counts[0]);
    if (!read_blob(fp, blob) || blob.size() != counts[0] * sizeof(FileNode)) return fail();
    if (counts[0]) memcpy(fnodes.data(), blob.data(), blob.size());
    std::vector<FileMaterial> fmats(counts[1]);
    if (!read_blob(fp, blob) || blob.size() != counts[1] * sizeof(FileMaterial)) return fail();
    if (counts[1]) memcpy(fmats.data(), blob.data(), blob.size());
    std::vector<FileLight> flights(counts[2]);
    if (!read_blob(fp, blob) || blob.size() != counts[2] * sizeof(FileLight)) return fail();
    if (counts[2]) memcpy(flights.data(), blob.data(), blob.size());
    sc->textures.resize(counts[3]); sc->texels.resize(counts[3]);
    for (uint32_t t = 0; t < counts[3]; t++) {
        uint32_t wh[2];
        if (fread(wh, sizeof wh, 1, fp) != 1 || !read_blob(fp, blob) || blob.size() != (size_t)wh[0] * wh[1] * 3) return fail();
        sc->texels[t].resize((size_t)wh[0] * wh[1]);
        if (!blob.empty()) memcpy(sc->texels[t].data(), blob.data(), blob.size());
        sc->textures[t].data = sc->texels[t].data(); sc->textures[t].width = wh[0]; sc->textures[t].height = wh[1];
    }
    for (uint32_t m = 0; m < counts[4]; m++) {
        Mesh* mesh = new Mesh;
        sc->meshes.push_back(mesh);
        uint32_t n[6]; float bmin[3], bmax[3];
        if (fread(n, sizeof n, 1, fp) != 1 || fread(bmin, 12, 1, fp) != 1 || fread(bmax, 12, 1, fp) != 1) return fail();
        mesh->nv = n[0]; mesh->nf = n[1]; mesh->nvn = n[2]; mesh->nvt = n[3];
        mesh->box.pmin = f3of(bmin); mesh->box.pmax = f3of(bmax);
        auto rd_f3 = [&](std::vector<float3>& dst, size_t cnt) { if (!read_blob(fp, blob) || blob.size() != cnt * 12) return false; dst.resize(cnt); if (cnt) memcpy(dst.data(), blob.data(), blob.size()); return true; };
        auto rd_u3 = [&](std::vector<uint32_t>& dst, size_t cnt) { if (!read_blob(fp, blob) || blob.size() != cnt * 12) return false; dst.resize(cnt * 3); if (cnt) memcpy(dst.data(), blob.data(), blob.size()); return true; };
        if (!rd_f3(mesh->v, n[0]) || !rd_u3(mesh->f, n[1]) || !rd_f3(mesh->vn, n[2]) || !rd_u3(mesh->fn, n[4]) ||
            !rd_f3(mesh->vt, n[3]) || !rd_u3(mesh->ft, n[5])) return fail();
        mesh->ConstructBVH();
    }
    fclose(fp);
    sc->materials.resize(counts[1]);
    for (uint32_t i = 0; i < counts[1]; i++) {
        Material& m = sc->materials[i];
        m.diffuse = &sc->textures[fmats[i].diffuse]; m.specular = &sc->textures[fmats[i].specular];
        m.refraction = &sc->textures[fmats[i].refraction]; m.emission = &sc->textures[fmats[i].emission];
        m.glossiness = fmats[i].glossiness; m.ior = fmats[i].ior; m.absorption = f3of(fmats[i].absorption);
    }
    sc->lights.resize(counts[2]);
    for (uint32_t i = 0; i < counts[2]; i++) {
        Light& l = sc->lights[i];
        l.kind = flights[i].kind; l.intensity = f3of(flights[i].intensity); l.position = f3of(flights[i].vec); l.size = flights[i].size;
    }
    sc->nodes.resize(counts[0]);
    for (uint32_t i = 0; i < counts[0]; i++) {
        Node& nd = sc->nodes[i];
        nd.childCount = fnodes[i].child_count; nd.objectKind = fnodes[i].object_kind;
        nd.mesh = nd.objectKind == OBJ_MESH ? sc->meshes[fnodes[i].mesh_id] : nullptr;
        nd.material = &sc->materials[fnodes[i].material_id];
        nd.boundingBox.pmin = f3of(fnodes[i].box_min); nd.boundingBox.pmax = f3of(fnodes[i].box_max);
        memcpy(nd.tm.cells, fnodes[i].tm, 48); memcpy(nd.itm.cells, fnodes[i].itm, 48);
    }
    sc->env = &sc->textures[env];
    return sc;
}

// ------------------------------------------------ C entry points (ctypes)
extern "C" {

void oracle_set_node_cull(int by_entry) { g_cull_by_entry.store(by_entry ? 1 : 0); }

void* oracle_scene_load(const char* path) { seqjump_init(); return load_scene_file(path); }
void oracle_scene_free(void* h) { delete (Scene*)h; }

void oracle_scene_info(void* h, uint32_t* out) {   // width, height, nodes, lights, materials, meshes, srgb
    Scene* sc = (Scene*)h;
    out[0] = sc->cam.width; out[1] = sc->cam.height; out[2] = (uint32_t)sc->nodes.size(); out[3] = (uint32_t)sc->lights.size();
    out[4] = (uint32_t)sc->materials.size(); out[5] = (uint32_t)sc->meshes.size(); out[6] = (uint32_t)sc->cam.sRGB;
}

void oracle_set_resolution(void* h, uint32_t w, uint32_t hgt) {
    Scene* sc = (Scene*)h; sc->cam.width = w; sc->cam.height = hgt; camera_derive(sc->cam);
}

// camera as 22 floats: pixel_size, cX, cY, cZ, top_left, position, dof + 3 pad
void oracle_get_camera(void* h, float* out) {
    const Camera& c = ((Scene*)h)->cam;
    out[0] = c.pixelSize;
    const float3 v[5] = {c.cX, c.cY, c.cZ, c.topLeftPixel, c.position};
    for (int i = 0; i < 5; i++) { out[1 + 3 * i] = v[i].x; out[2 + 3 * i] = v[i].y; out[3 + 3 * i] = v[i].z; }
    out[16] = c.dof;
}

uint32_t oracle_bvh_node_count(void* h, uint32_t mesh) { return (uint32_t)((Scene*)h)->meshes[mesh]->nodes.size(); }
uint32_t oracle_bvh_depth(void* h, uint32_t mesh) { return (uint32_t)((Scene*)h)->meshes[mesh]->depth; }
uint32_t oracle_mesh_face_count(void* h, uint32_t mesh) { return (uint32_t)((Scene*)h)->meshes[mesh]->nf; }

// Copies the BVH of one mesh out field by field (u32 / float), elements last.
void oracle_bvh_arrays(void* h, uint32_t mesh, uint8_t* leaf, uint32_t* start, uint32_t* count, uint32_t* left,
                       uint32_t* right, uint32_t* next, float* box, uint32_t* elements) {
    const Mesh* m = ((Scene*)h)->meshes[mesh];
    for (size_t i = 0; i < m->nodes.size(); i++) {
        const BVHNode& n = m->nodes[i];
        leaf[i] = n.leaf; start[i] = (uint32_t)n.elemStartIndex; count[i] = (uint32_t)n.elemCount;
        left[i] = (uint32_t)n.leftChild; right[i] = (uint32_t)n.rightChild; next[i] = (uint32_t)n.nextNode;
        box[6 * i] = n.box.pmin.x; box[6 * i + 1] = n.box.pmin.y; box[6 * i + 2] = n.box.pmin.z;
        box[6 * i + 3] = n.box.pmax.x; box[6 * i + 4] = n.box.pmax.y; box[6 * i + 5] = n.box.pmax.z;
    }
    for (size_t i = 0; i < m->elements.size(); i++) elements[i] = (uint32_t)m->elements[i];
}

// One pass over rows [y0, y1): accumulators are read, added to and written back
// (trace.cu:39; the caller zeroes them before pass 0).  counters may be NULL;
// otherwise 12 u64 in the order of struct Counters.
void oracle_render(void* h, int pass, int samples, float* results, float* normals, float* albedos,
                   uint32_t y0, uint32_t y1, int nthreads, uint64_t* counters) {
    const Scene& sc = *(Scene*)h;
    if (nthreads < 1) nthreads = 1;
    if (y1 > sc.cam.height) y1 = sc.cam.height;
    std::atomic<uint32_t> nextRow(y0);
    std::vector<Counters> cnt(nthreads);
    auto work = [&](int tid) {
        tl_cnt = counters ? &cnt[tid] : nullptr;
        for (;;) {
            const uint32_t y = nextRow.fetch_add(1);
            if (y >= y1) break;
            for (uint32_t x = 0; x < sc.cam.width; x++) trace_pixel(sc, x, y, pass, samples, results, normals, albedos);
        }
        tl_cnt = nullptr;
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters) {
        Counters total;
        for (auto& c : cnt) total.add(c);
        memcpy(counters, &total, sizeof total);
    }
}

// First camera ray of the pass for every pixel of rows [y0, y1), closest hit only.
// node: hit node index, -1 miss, -2 - light index for a light.  face: face index
// for mesh hits else -1.  t: hit.z.
void oracle_primary_hits(void* h, int pass, uint32_t y0, uint32_t y1, int32_t* node, int32_t* face, float* t) {
    const Scene& sc = *(Scene*)h;
    if (y1 > sc.cam.height) y1 = sc.cam.height;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = 0; x < sc.cam.width; x++) {
            const unsigned pI = y * sc.cam.width + x;
            XorwowState rng;
            xorwow_init(pI, (uint64_t)pass, &rng);
            Ray ray;
            camera_ray(sc, x, y, &rng, 1, ray);
            Hit hit;
            const bool any = TraceThroughScene(sc, ray, hit);
            int32_t id = -1;
            if (any) id = hit.hitLight ? -2 - (int32_t)(hit.light - sc.lights.data()) : (int32_t)(hit.node - sc.nodes.data());
            if (node) node[pI] = id;
            if (face) face[pI] = (any && !hit.hitLight) ? hit.face : -1;
            if (t) t[pI] = hit.z;
        }
}

// Pixel-centre primary rays without any random offset (the survey's anchor
// configuration): returns the number of pixels whose ray hits a node or light.
uint64_t oracle_count_center_hits(void* h) {
    const Scene& sc = *(Scene*)h;
    const Camera& c = sc.cam;
    uint64_t hits = 0;
    for (unsigned y = 0; y < c.height; y++)
        for (unsigned x = 0; x < c.width; x++) {
            const float3 pixelCoords = c.topLeftPixel + c.pixelSize * (x * c.cX - y * c.cY);
            Ray ray; ray.pos = c.position; ray.dir = pixelCoords - c.position; ray.contribution = WHITE;
            Hit hit;
            if (TraceThroughScene(sc, ray, hit)) hits++;
        }
    return hits;
}

// math/color.cu:19-23
void oracle_convert(const float* in, uint8_t* out, uint64_t n, int sRGB, float passMultiplier) {
    for (uint64_t i = 0; i < n; i++) {
        color c = mk(in[3 * i], in[3 * i + 1], in[3 * i + 2]) * passMultiplier;
        if (sRGB) c = mk(LinearTosRGB(c.x), LinearTosRGB(c.y), LinearTosRGB(c.z));
        out[3 * i] = ClampToInt(c.x * 256); out[3 * i + 1] = ClampToInt(c.y * 256); out[3 * i + 2] = ClampToInt(c.z * 256);
    }
}

// State words (d, v0..v4) after init, then `draws` raw outputs, per (seed, subsequence).
void oracle_xorwow(const uint32_t* seeds, const uint32_t* subseq, uint32_t n, uint32_t draws, uint32_t* out) {
    seqjump_init();
    for (uint32_t i = 0; i < n; i++) {
        XorwowState s;
        xorwow_init(seeds[i], subseq[i], &s);
        uint32_t* o = out + (size_t)i * (6 + draws);
        o[0] = s.d; for (int k = 0; k < 5; k++) o[1 + k] = s.v[k];
        for (uint32_t d = 0; d < draws; d++) o[6 + d] = xorwow_next(&s);
    }
}

// The 2^67-step jump as 160 rows of 5 words (row r = image of state bit r).
void oracle_seqjump_matrix(uint32_t k, uint32_t* out) {
    seqjump_init();
    memcpy(out, g_seqjump[k].row, sizeof g_seqjump[k].row);
}

// fn: 0 sin, 1 cos, 2 exp, 3 pow(x, y), 4 atan2(x, y), 5 asin, 6 acos, 7 tan
void oracle_math(int fn, const float* x, const float* y, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = m_sinf(x[i]); break;
            case 1: out[i] = m_cosf(x[i]); break;
            case 2: out[i] = m_expf(x[i]); break;
            case 3: out[i] = m_powf(x[i], y[i]); break;
            case 4: out[i] = m_atan2f(x[i], y[i]); break;
            case 5: out[i] = m_asinf(x[i]); break;
            case 6: out[i] = m_acosf(x[i]); break;
            default: out[i] = m_tanf(x[i]); break;
        }
    }
}

int oracle_libm_is_glibc(void) {
#ifdef ORACLE_LIBM_GLIBC
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
