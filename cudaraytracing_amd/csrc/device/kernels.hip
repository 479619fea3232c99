// gfx950 kernels of the per-pixel trace path.
//
// render_kernel replaces TracePrimaryRays (renderer/trace.cu:7-41) and every
// device function it reaches.  It is a persistent kernel: each lane owns one
// pixel at a time, runs that pixel's samples one after another (they share one
// random stream, renderer/trace.cu:22-25) and, when the pixel is finished,
// takes the next unowned pixel from a global queue with one wave-aggregated
// atomic (ballot + mbcnt prefix), so lanes do not idle on short pixels.
//
// Inside a ray query the lanes of a wave are generally at different scene
// nodes and in different mesh BVHs.  Both the linear scene-node scan and the
// BVH descent are therefore expressed as the same step - "fetch one 64-byte
// box-pair record, slab-test two boxes against my current ray" - so that all
// 64 lanes execute the expensive part together whatever phase each one is in.
// The per-lane traversal stack lives in LDS, laid out [entry][thread] so that
// any mix of stack depths is bank-conflict free.
//
// Arithmetic keeps the reference's operation order (see crt_vec.h); the file is
// compiled with -ffp-contract=off and uses the pinned transcendentals of
// crt_detmath.h, so results equal the CPU oracle bit for bit.
#include <hip/hip_runtime.h>

#include <float.h>

#include "dev_scene.h"

namespace crt {

#define CRT_BLOCK 256
#define CRT_BIAS 0.02f
#define CRT_MAX_SURVIVAL 0.95f
#define CRT_F_EPS 1.1920929e-7f
#define CRT_TRI_EPS 0.000001f
#define CRT_MAX_SEGMENTS 100000      // safety bound on one path; survival <= 0.95 per segment

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------ random numbers
// XORWOW with cuRAND's seeding (curand_init(seed, subsequence, 0), renderer/trace.cu:22-23).
struct Rng {
    uint32_t d, v0, v1, v2, v3, v4;
};

DEV uint32_t rng_next(Rng& s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// RandomFloat (renderer/rng.cuh:8-12)
DEV float rng_float(Rng& s) {
    const float r = (float)rng_next(s) * 0x1p-32f;
    return r < 0x1.fffffep-1f ? r : 0x1.fffffep-1f;
}

// jump: 160 rows x 5 words, row r = image of state bit r under "advance by
// subsequence * 2^67 steps"; nullptr for subsequence 0.
DEV void rng_init(Rng& s, uint32_t seed_lo, const uint32_t* __restrict__ jump) {
    const uint32_t s0 = seed_lo ^ 0xaad26b49u;
    const uint32_t s1 = 0u ^ 0xf7dcefddu;          // high seed word is zero: seeds are pixel indices
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v0 = 123456789u + t0;
    s.v1 = 362436069u ^ t0;
    s.v2 = 521288629u + t1;
    s.v3 = 88675123u ^ t1;
    s.v4 = 5783321u + t0;
    if (jump) {
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
        auto fold = [&](uint32_t word, const uint32_t* __restrict__ rows) {
#pragma unroll 1
            for (int b = 0; b < 32; b++) {
                const uint32_t m = 0u - ((word >> b) & 1u);
                const uint32_t* row = rows + b * 5;
                r0 ^= row[0] & m; r1 ^= row[1] & m; r2 ^= row[2] & m; r3 ^= row[3] & m; r4 ^= row[4] & m;
            }
        };
        fold(s.v0, jump); fold(s.v1, jump + 160); fold(s.v2, jump + 320); fold(s.v3, jump + 480); fold(s.v4, jump + 640);
        s.v0 = r0; s.v1 = r1; s.v2 = r2; s.v3 = r3; s.v4 = r4;
    }
}

// ------------------------------------------------------------------ counters
struct Tally {
    uint32_t closest_rays, shadow_rays, paths;
    uint32_t node_box_tests, node_enters, bvh_interior, bvh_leaves, tri_tests, tri_hits, sphere_tests, plane_tests, texel_quads;
};
#define TALLY(field) do { if (kCount) tally.field++; } while (0)

// ------------------------------------------------------------------ hit record (scene/rays.cuh:20-61)
struct HitRec {
    f3 pos; float z; f3 n; f3 uvw;
    bool front;
    int node;        // hit scene node
    int light;       // >= 0: the hit is on this light
    int face;        // face index for mesh hits
};

DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// Box slab test (Box::IntersectRay / IntersectShadowRay, scene/rays.cuh:138-187) with the
// reciprocal direction hoisted; dist is meaningful only when the test passes.
DEV bool slab(const float* bmin, const float* bmax, f3 pos, f3 inv, float tmax, float& dist) {
    const float lx = (bmin[0] - pos.x) * inv.x, ly = (bmin[1] - pos.y) * inv.y, lz = (bmin[2] - pos.z) * inv.z;
    const float hx = (bmax[0] - pos.x) * inv.x, hy = (bmax[1] - pos.y) * inv.y, hz = (bmax[2] - pos.z) * inv.z;
    const float cx = fminf(lx, hx), cy = fminf(ly, hy), cz = fminf(lz, hz);
    const float fx = fmaxf(lx, hx), fy = fmaxf(ly, hy), fz = fmaxf(lz, hz);
    const float tEnter = fmaxf(cx, fmaxf(cy, cz));
    const float tExit = fminf(fx, fminf(fy, fz));
    dist = tEnter >= -FLT_EPSILON ? tEnter : tExit;
    return (tEnter >= -FLT_EPSILON || tExit >= -FLT_EPSILON) && tEnter <= tExit && tEnter <= tmax;
}

DEV f3 recip3(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// ------------------------------------------------------------------ textures (scene/texture.cu:14-63)
DEV f3 texel(const uint8_t* __restrict__ p) {
    return mk3((float)p[0], (float)p[1], (float)p[2]) * (1.0f / 256.0f);
}

template <bool kCount>
DEV f3 tex_eval(const DevScene& S, int tex, f3 uvw, Tally& tally) {
    const DevTexture t = S.textures[tex];
    if (t.width + t.height == 0) return mk3(0, 0, 0);
    TALLY(texel_quads);
    f3 u;
    u.x = uvw.x - (float)crtm::f2i_sat(uvw.x);
    u.y = uvw.y - (float)crtm::f2i_sat(uvw.y);
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    const float x = (float)t.width * u.x;
    const float y = (float)t.height * u.y;
    int ix = crtm::f2i_sat(x);
    int iy = crtm::f2i_sat(y);
    const float fx = x - (float)ix;
    const float fy = y - (float)iy;
    const uint32_t w = t.width, h = t.height;
    // int / unsigned mixing as in the reference (texture.cu:38-46)
    if (ix < 0) ix -= (int)(((uint32_t)ix / w - 1u) * w);
    if ((uint32_t)ix >= w) ix -= (int)(((uint32_t)ix / w) * w);
    int ixp = ix + 1;
    if ((uint32_t)ixp >= w) ixp -= (int)w;
    if (iy < 0) iy -= (int)(((uint32_t)iy / h - 1u) * h);
    if ((uint32_t)iy >= h) iy -= (int)(((uint32_t)iy / h) * h);
    int iyp = iy + 1;
    if ((uint32_t)iyp >= h) iyp -= (int)h;
    const uint8_t* base = S.texels + t.offset;
    const f3 c00 = texel(base + 3u * ((uint32_t)iy * w + (uint32_t)ix));
    const f3 c01 = texel(base + 3u * ((uint32_t)iy * w + (uint32_t)ixp));
    const f3 c10 = texel(base + 3u * ((uint32_t)iyp * w + (uint32_t)ix));
    const f3 c11 = texel(base + 3u * ((uint32_t)iyp * w + (uint32_t)ixp));
    return c00 * ((1 - fx) * (1 - fy)) + c01 * (fx * (1 - fy)) + c10 * ((1 - fx) * fy) + c11 * (fx * fy);
}

template <bool kCount>
DEV f3 tex_env(const DevScene& S, f3 dir, Tally& tally) {
    const f3 d = normalized(dir);
    const float phi = crtm::det_atan2f(d.x, d.z);
    const float theta = crtm::det_acosf(fminf(fmaxf(d.y, -1.0f), 1.0f));
    // double arithmetic narrowed on assignment (texture.cu:60-61)
    const float u = (float)(((double)phi + 3.14159265358979323846) * (0.5f / 3.14159265358979323846));
    const float v = (float)((double)theta * (1.0f / 3.14159265358979323846));
    return tex_eval<kCount>(S, S.env_texture, mk3(u, v, 0.0f), tally);
}

// ------------------------------------------------------------------ math/vector.cu:3-52
DEV void orthonormals(f3 n, f3& x, f3& y) {
    if (fabsf(n.x) > fabsf(n.z)) x = mk3(-n.y, n.x, 0.0f);
    else x = mk3(0.0f, -n.z, n.y);
    x = normalized(x);
    y = cross(n, x);
}

DEV f3 reflect(f3 incident, f3 normal) { return 2 * dot(incident, normal) * normal - incident; }

DEV f3 transmit(f3 incident, f3 normal, float outerIor, float innerIor, bool& tir) {
    const float cos_theta = dot(incident, normal);
    const float eta = outerIor / innerIor;
    const float cos_phi_squared = 1 - eta * eta * (1 - cos_theta * cos_theta);
    if (cos_phi_squared < 0) { tir = true; return reflect(incident, normal); }
    tir = false;
    return -normal * sqrtf(cos_phi_squared) + eta * (cos_theta * normal - incident);
}

DEV f3 glossy_normal(f3 normal, float glossiness, Rng& rng) {
    const float x = rng_float(rng);
    const float phi = (float)((double)rng_float(rng) * 3.14159265358979323846 * 2.0);
    const float cos_theta = crtm::det_powf(1 - x, 1.0f / (glossiness + 1.0f));
    f3 u, v;
    orthonormals(normal, u, v);
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    const float cos_phi = crtm::det_cosf(phi);
    const float sin_phi = crtm::det_sinf(phi);
    return normal * cos_theta + u * sin_theta * cos_phi + v * sin_theta * sin_phi;
}

// ------------------------------------------------------------------ primitives (scene/objects.cu:7-159)
DEV void sphere_uvw(f3 p, f3& uvw) {
    uvw = mk3(0.5f * 0.318309f * crtm::det_atan2f(p.y, p.x) + 0.5f,
              0.318309f * crtm::det_asinf(fminf(fmaxf(p.z, -1.0f), 1.0f)) + 0.5f, 0);
}

// unit sphere, local space; both sides accepted (the reference always passes FRONT_AND_BACK)
DEV bool sphere_closest(f3 pos, f3 dir, HitRec& hit) {
    if (dot(dir, pos) >= 0) return false;
    const float a = dot(dir, dir);
    const float b = dot(dir, pos) * 2;
    const float c = dot(pos, pos) - 1;
    const float det = b * b - 4 * a * c;
    if (det < 0) return false;
    const float sq = sqrtf(det);
    const float front = (-b - sq) / (2 * a);
    const float back = (-b + sq) / (2 * a);
    if (front >= 0) {
        if (front >= hit.z) return false;
        hit.z = front;
        hit.pos = pos + front * dir;
        hit.n = hit.pos;
        sphere_uvw(hit.pos, hit.uvw);
        hit.front = true; hit.light = -1; hit.face = -1;
        return true;
    }
    if (back < 0) return false;
    if (back >= hit.z) return false;
    hit.z = back;
    hit.pos = pos + back * dir;
    hit.n = hit.pos;
    sphere_uvw(hit.pos, hit.uvw);
    hit.front = false; hit.light = -1; hit.face = -1;
    return true;
}

DEV bool sphere_shadow(f3 pos, f3 dir, float tMax) {
    if (dot(dir, pos) >= 0) return false;
    const float a = dot(dir, dir);
    const float b = dot(dir, pos) * 2;
    const float c = dot(pos, pos) - 1;
    const float det = b * b - 4 * a * c;
    if (det < 0) return false;
    const float sq = sqrtf(det);
    const float front = (-b - sq) / (2 * a);
    const float back = (-b + sq) / (2 * a);
    if (front >= 0 && front <= tMax) return true;
    if (back >= 0 && back <= tMax) return true;
    return false;
}

DEV bool plane_closest(f3 pos, f3 dir, HitRec& hit) {
    if (dir.z == 0) return false;
    const float t = -pos.z / dir.z;
    if (t < 0 || t >= hit.z) return false;
    if (!(dir.z < 0) && !(dir.z > 0)) return false;
    const f3 p = pos + t * dir;
    if (fabsf(p.x) > 1 || fabsf(p.y) > 1) return false;
    hit.z = t;
    hit.pos = p;
    hit.n = mk3(0, 0, 1);
    hit.uvw = 0.5f * (p + mk3(1, 1, 0));
    hit.front = dir.z < 0;
    hit.light = -1; hit.face = -1;
    return true;
}

DEV bool plane_shadow(f3 pos, f3 dir, float tMax) {
    if (dir.z == 0) return false;
    const float t = -pos.z / dir.z;
    if (t < 0 || t >= tMax) return false;
    if (!(dir.z < 0) && !(dir.z > 0)) return false;
    const f3 p = pos + t * dir;
    if (fabsf(p.x) > 1 || fabsf(p.y) > 1) return false;
    return true;
}

// PointLight::Intersect (scene/lights.cu:36-89), world space
DEV bool light_intersect(const DevLight& L, f3 rpos, f3 rdir, HitRec& hit) {
    const f3 lp = ld3(L.pos);
    const f3 o = rpos - lp;
    const float a = dot(rdir, rdir);
    const float b = dot(2 * rdir, o);
    const float c = dot(o, o) - L.size * L.size;
    const float det = b * b - 4 * a * c;
    if (det < 0) return false;
    const float sq = sqrtf(det);
    const float front = (-b - sq) / (2 * a);
    const float back = (-b + sq) / (2 * a);
    const float overpi = (float)(1.0f / 3.14159265358979323846);
    if (front >= 0) {
        if (front >= hit.z) return false;
        hit.z = front;
        const f3 np = normalized(o);
        hit.pos = rpos + front * rdir;
        hit.n = np;
        hit.uvw = mk3(0.5f * overpi * crtm::det_atan2f(np.y, np.x) + 0.5f,
                      overpi * crtm::det_asinf(fminf(fmaxf(np.z, -1.0f), 1.0f)) + 0.5f, 0);
        hit.front = true;
        return true;
    }
    if (back < 0) return false;
    if (back >= hit.z) return false;
    hit.z = back;
    const f3 np = normalized(hit.pos);      // the previous hit position, as the reference reads it (lights.cu:78)
    hit.pos = rpos + back * rdir;
    hit.uvw = mk3(0.5f * overpi * crtm::det_atan2f(np.y, np.x) + 0.5f,
                  overpi * crtm::det_asinf(fminf(fmaxf(np.z, -1.0f), 1.0f)) + 0.5f, 0);
    hit.n = np;
    hit.front = false;
    return true;
}

// ------------------------------------------------------------------ scene traversal
// One function for both query kinds:
//   kShadow == false : TraceThroughScene (renderer/trace.cu:44-81) + MeshObject::Traverse
//                      (scene/objects.cu:173-222) + IntersectTriangle (270-324)
//   kShadow == true  : TraceShadowRay (renderer/trace.cu:160-196) + TraverseShadow (224-268)
//                      + IntersectShadowTriangle (326-366); returns true when occluded.
// rpos / rdir are the ray in world space and are transformed in place into and
// out of every entered node, as the reference does.  The caller has applied the bias.
template <bool kShadow, bool kCount>
DEV bool trace_scene(const DevScene& S, f3& rpos, f3& rdir, float tMax, HitRec& hit,
                     uint32_t* __restrict__ stack_ref, float* __restrict__ stack_t, Tally& tally) {
    const BoxPair* __restrict__ pairs = S.pairs;
    const TriRec* __restrict__ tris = S.tris;
    const uint32_t node_count = S.node_count;
    const uint32_t tid = threadIdx.x;

    bool any = false;
    uint32_t node = 0;            // next scene node to look at (scan mode)
    bool trav = false;            // inside a mesh BVH
    uint32_t cur = 0;             // interior BVH record to expand (trav mode)
    uint32_t sp = 0;              // LDS stack height
    uint32_t entered = 0;         // scene node being traversed
    f3 inv = recip3(rdir);
    // pending closest hit on the mesh being traversed
    bool mesh_hit = false, mesh_front = false;
    uint32_t mesh_slot = 0;
    float mesh_b1 = 0, mesh_b2 = 0;

    while (node < node_count || trav) {
        // ---- the common step: one record, two slab tests
        const uint32_t rec_idx = trav ? cur : node;
        const float4* rp = reinterpret_cast<const float4*>(pairs + rec_idx);
        const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
        const float a_min[3] = {q0.x, q0.y, q0.z}, a_max[3] = {q0.w, q1.x, q1.y};
        const float b_min[3] = {q1.z, q1.w, q2.x}, b_max[3] = {q2.y, q2.z, q2.w};
        const uint32_t ref_a = __float_as_uint(q3.x), ref_b = __float_as_uint(q3.y);
        const float limit = kShadow ? tMax : CRT_BIGFLOAT;
        float tA = CRT_BIGFLOAT, tB = CRT_BIGFLOAT;
        float dA, dB;
        const bool hitA = slab(a_min, a_max, rpos, inv, limit, dA);
        const bool hitB = slab(b_min, b_max, rpos, inv, limit, dB);
        if (hitA) tA = dA;
        if (hitB) tB = dB;

        bool resolve = false;           // run the stack-resolve loop below
        bool have_next = false;
        uint32_t next_ref = 0;
        float next_t = 0;

        if (trav) {
            TALLY(bvh_interior);
            if (hitA || hitB) {
                const bool a_near = tA <= tB;
                const uint32_t far_ref = a_near ? ref_b : ref_a;
                stack_ref[sp * CRT_BLOCK + tid] = far_ref;
                if (!kShadow) stack_t[sp * CRT_BLOCK + tid] = fmaxf(tA, tB);
                sp++;
                next_ref = a_near ? ref_a : ref_b;
                next_t = fminf(tA, tB);
                have_next = true;
            }
            resolve = true;
        } else {
            // ---- scan mode: node `node` is half A, node + 1 is half B
            uint32_t enter_node = kRefNone;
            uint32_t enter_kind = 0;
            uint32_t nxt;
            const uint32_t kind_a = node_ref_kind(ref_a);
            bool a_entered = false;
            nxt = node + 1;
            if (kind_a != CRT_OBJ_NONE) {
                TALLY(node_box_tests);
                if (kShadow) {
                    if (hitA) { a_entered = true; }
                    else nxt = node_ref_after(ref_a);           // skip the subtree (trace.cu:171-185)
                } else if (hitA && tA < hit.z) a_entered = true;
            }
            if (a_entered) { enter_node = node; enter_kind = kind_a; }
            else if (nxt == node + 1 && nxt < node_count) {
                const uint32_t kind_b = node_ref_kind(ref_b);
                nxt = node + 2;
                if (kind_b != CRT_OBJ_NONE) {
                    TALLY(node_box_tests);
                    if (kShadow) {
                        if (hitB) { enter_node = node + 1; enter_kind = kind_b; }
                        else nxt = node_ref_after(ref_b);
                    } else if (hitB && tB < hit.z) { enter_node = node + 1; enter_kind = kind_b; }
                }
            }
            if (enter_node == kRefNone) {
                node = nxt;
            } else {
                TALLY(node_enters);
                const NodeXf* __restrict__ xf = S.xf + enter_node;
                rpos = xform_point(xf->itm, rpos);
                rdir = xform_dir(xf->itm, rdir);
                if (enter_kind == CRT_OBJ_MESH) {
                    inv = recip3(rdir);
                    trav = true; entered = enter_node; sp = 0; mesh_hit = false;
                    next_ref = xf->root_ref; next_t = 0; have_next = true;
                    resolve = true;
                } else {
                    bool got;
                    if (enter_kind == CRT_OBJ_SPHERE) {
                        TALLY(sphere_tests);
                        if (kShadow) got = sphere_shadow(rpos, rdir, tMax);
                        else got = sphere_closest(rpos, rdir, hit);
                    } else {
                        TALLY(plane_tests);
                        if (kShadow) got = plane_shadow(rpos, rdir, tMax);
                        else got = plane_closest(rpos, rdir, hit);
                    }
                    if (got) {
                        if (kShadow) return true;
                        any = true;
                        hit.pos = xform_point(xf->tm, hit.pos);
                        hit.n = xform_normal(xf->itm, hit.n);
                        hit.node = (int)enter_node;
                    }
                    rpos = xform_point(xf->tm, rpos);
                    rdir = xform_dir(xf->tm, rdir);
                    inv = recip3(rdir);
                    node = enter_node + 1;
                }
            }
        }

        if (resolve) {
            // pop until an interior node that survives the distance cull is found
            bool finished = false;
            for (;;) {
                if (!have_next) {
                    if (sp == 0) { finished = true; break; }
                    sp--;
                    next_ref = stack_ref[sp * CRT_BLOCK + tid];
                    if (!kShadow) next_t = stack_t[sp * CRT_BLOCK + tid];
                }
                have_next = false;
                if (!kShadow && hit.z < next_t) continue;
                if (!ref_is_leaf(next_ref)) { cur = next_ref; break; }
                TALLY(bvh_leaves);
                const uint32_t t0 = leaf_start(next_ref), tn = leaf_count(next_ref);
                for (uint32_t k = 0; k < tn; k++) {
                    TALLY(tri_tests);
                    const float4* tp = reinterpret_cast<const float4*>(tris + t0 + k);
                    const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                    const f3 p0 = mk3(v0.x, v0.y, v0.z);
                    const f3 edge1 = mk3(v1.x, v1.y, v1.z) - p0;
                    const f3 edge2 = mk3(v2.x, v2.y, v2.z) - p0;
                    const f3 rXe2 = cross(rdir, edge2);
                    const float det = dot(edge1, rXe2);
                    if (det > -CRT_TRI_EPS && det < CRT_TRI_EPS) continue;
                    const float idet = 1 / det;
                    const f3 offset = rpos - p0;
                    const float b1 = idet * dot(offset, rXe2);
                    if (b1 <= CRT_TRI_EPS || b1 >= 1 + CRT_TRI_EPS) continue;
                    const f3 oXe1 = cross(offset, edge1);
                    const float b2 = idet * dot(rdir, oXe1);
                    if (b2 <= -CRT_TRI_EPS || (b1 + b2) >= 1 + CRT_TRI_EPS) continue;
                    const float t = idet * dot(edge2, oXe1);
                    const float far_limit = (kShadow ? tMax : hit.z) + CRT_TRI_EPS;
                    if (t <= CRT_TRI_EPS || t >= far_limit) continue;
                    const float ndot = dot(cross(edge1, edge2), rdir);
                    if (!(ndot < CRT_TRI_EPS) && !(ndot > -CRT_TRI_EPS)) continue;
                    if (kShadow) return true;
                    TALLY(tri_hits);
                    hit.z = t;
                    mesh_hit = true; mesh_front = ndot < 0; mesh_slot = t0 + k; mesh_b1 = b1; mesh_b2 = b2;
                }
            }
            if (finished) {
                // leave the mesh: finish the hit (objects.cu:308-323), then Node::FromLocal for hit and ray
                const NodeXf* __restrict__ xf = S.xf + entered;
                if (!kShadow && mesh_hit) {
                    const float4* tp = reinterpret_cast<const float4*>(tris + mesh_slot);
                    const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                    const f3 p0 = mk3(v0.x, v0.y, v0.z);
                    const f3 edge1 = mk3(v1.x, v1.y, v1.z) - p0;
                    const f3 edge2 = mk3(v2.x, v2.y, v2.z) - p0;
                    const uint32_t face = __float_as_uint(v0.w);
                    const float b0 = 1 - mesh_b2 - mesh_b1;
                    const DevMesh M = S.meshes[xf->mesh_id];
                    f3 n, uvw;
                    if (!M.has_normals) n = normalized(cross(edge1, edge2));
                    else {
                        const uint32_t i0 = M.fn[3 * face], i1 = M.fn[3 * face + 1], i2 = M.fn[3 * face + 2];
                        n = ld3(M.vn + 3 * i0) * b0 + ld3(M.vn + 3 * i1) * mesh_b1 + ld3(M.vn + 3 * i2) * mesh_b2;
                    }
                    if (!M.has_tex) uvw = mk3(0, 0, 0);
                    else {
                        const uint32_t i0 = M.ft[3 * face], i1 = M.ft[3 * face + 1], i2 = M.ft[3 * face + 2];
                        uvw = ld3(M.vt + 3 * i0) * b0 + ld3(M.vt + 3 * i1) * mesh_b1 + ld3(M.vt + 3 * i2) * mesh_b2;
                    }
                    any = true;
                    hit.pos = xform_point(xf->tm, rpos + hit.z * rdir);
                    hit.n = xform_normal(xf->itm, n);
                    hit.uvw = uvw;
                    hit.front = mesh_front;
                    hit.light = -1;
                    hit.face = (int)face;
                    hit.node = (int)entered;
                }
                rpos = xform_point(xf->tm, rpos);
                rdir = xform_dir(xf->tm, rdir);
                inv = recip3(rdir);
                trav = false;
                node = entered + 1;
            }
        }
    }

    if (kShadow) return false;

    // lights are tested last, against the ray as the node loop left it (trace.cu:72-78)
    for (uint32_t i = 0; i < S.light_count; i++) {
        const DevLight L = S.lights[i];
        if (L.kind == CRT_LIGHT_POINT && light_intersect(L, rpos, rdir, hit)) { any = true; hit.light = (int)i; hit.face = -1; }
    }
    return any;
}

DEV void hit_init(HitRec& h) {
    h.pos = mk3(0, 0, 0); h.z = CRT_BIGFLOAT; h.n = mk3(0, 1, 0); h.uvw = mk3(0, 0, 0);
    h.front = true; h.node = -1; h.light = -1; h.face = -1;
}

// ------------------------------------------------------------------ camera (renderer/trace.cu:14-38)
DEV void camera_ray(const DevScene& S, uint32_t pX, uint32_t pY, Rng& rng, f3& pos, f3& dir) {
    const f3 cX = ld3(S.cX), cY = ld3(S.cY);
    const f3 pixel = ld3(S.top_left) + S.pixel_size * ((float)pX * cX - (float)pY * cY);
    const float F_PI = 3.14159265358979323846f;
    const float dof_r = sqrtf(rng_float(rng));
    const float dof_theta = rng_float(rng) * 2 * F_PI;
    const float dof_x = dof_r * crtm::det_cosf(dof_theta);
    const float dof_y = dof_r * crtm::det_sinf(dof_theta);
    const f3 start = ld3(S.cam_pos) + S.dof * (dof_x * cX - dof_y * cY);
    const float aa_x = rng_float(rng) - 0.5f;
    const float aa_y = rng_float(rng) - 0.5f;
    const f3 end = pixel + S.pixel_size * (aa_x * cX - aa_y * cY);
    pos = start;
    dir = end - start;
}

// queue item -> pixel; false when the item falls outside the image
DEV bool item_to_pixel(const DevScene& S, const DevPartition& P, uint32_t item, uint32_t& pX, uint32_t& pY) {
    const uint32_t strip_local = item / P.items_per_strip;
    const uint32_t in_strip = item - strip_local * P.items_per_strip;
    const uint32_t tile = in_strip >> 6, lane = in_strip & 63u;
    const uint32_t tile_row = tile / P.tiles_x, tile_col = tile - tile_row * P.tiles_x;
    const uint32_t strip = strip_local * P.parts + P.part;
    pX = tile_col * 8u + (lane & 7u);
    pY = strip * P.strip_rows + tile_row * 8u + (lane >> 3);
    return pX < S.width && pY < S.height;
}

DEV uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------ the path tracer
template <bool kCount>
__global__ __launch_bounds__(CRT_BLOCK) void render_kernel(DevScene S, DevPartition P, int samples,
                                                           const uint32_t* __restrict__ jump,
                                                           uint32_t* __restrict__ queue, DevStats* __restrict__ stats) {
    extern __shared__ uint32_t lds[];
    uint32_t* stack_ref = lds;
    float* stack_t = reinterpret_cast<float*>(lds + S.stack_entries * CRT_BLOCK);

    Tally tally = {};
    const float F_PI = 3.14159265358979323846f;
    const float OVERPI = (float)(1.0f / 3.14159265358979323846);
    const float inv_samples = 1.0f / (float)samples;

    bool have_pixel = false, queue_empty = false;
    uint32_t pI = 0, pX = 0, pY = 0;
    int sample = 0;
    bool new_path = true;
    Rng rng = {};
    f3 target = mk3(0, 0, 0), normal = mk3(0, 0, 0), albedo = mk3(0, 0, 0);
    f3 rpos = mk3(0, 0, 0), rdir = mk3(0, 0, 1), contrib = mk3(0, 0, 0);
    uint32_t segments = 0;

    for (;;) {
        // ---- refill idle lanes from the pixel queue
        if (!queue_empty) {
            const bool need = !have_pixel;
            const unsigned long long mask = __ballot(need);
            if (mask) {
                uint32_t base = 0;
                const uint32_t cnt = (uint32_t)__popcll(mask);
                if (lane_rank(mask) == 0 && need) base = atomicAdd(queue, cnt);
                base = __shfl(base, __ffsll((long long)mask) - 1, 64);
                if (base >= P.total_items) queue_empty = true;      // uniform: every later item is past the end too
                if (need) {
                    const uint32_t item = base + lane_rank(mask);
                    if (item < P.total_items && item_to_pixel(S, P, item, pX, pY)) {
                        have_pixel = true;
                        pI = pY * S.width + pX;
                        rng_init(rng, pI, jump);
                        target = ld3(S.results + 3ull * pI);
                        normal = ld3(S.normals + 3ull * pI);
                        albedo = ld3(S.albedos + 3ull * pI);
                        sample = 0; new_path = true;
                    }
                }
            }
        }
        if (__ballot(have_pixel) == 0ull) {
            if (queue_empty) break;
            continue;
        }
        if (!have_pixel) continue;

        // ---- one path segment (renderer/trace.cu:83-158)
        if (new_path) {
            camera_ray(S, pX, pY, rng, rpos, rdir);
            contrib = mk3(1, 1, 1) * inv_samples;
            new_path = false; segments = 0;
            TALLY(paths);
        }
        bool path_done = false;
        const f3 v = -normalized(rdir);
        HitRec hit;
        hit_init(hit);
        tally.closest_rays++;
        rpos += rdir * CRT_BIAS;
        const bool any = trace_scene<false, kCount>(S, rpos, rdir, CRT_BIGFLOAT, hit, stack_ref, stack_t, tally);
        if (!any) {
            const f3 env = tex_env<kCount>(S, rdir, tally);
            target += contrib * env;
            normal += contrib * normalized(rdir);
            albedo += contrib * env;
            path_done = true;
        } else if (hit.light >= 0) {
            const f3 rad = ld3(S.lights[hit.light].radiance);
            target += contrib * rad;
            normal += contrib * hit.n;
            albedo += contrib * rad;
            path_done = true;
        } else {
            const float light_pick = rng_float(rng);
            const DevMaterial mtl = S.materials[S.xf[hit.node].material_id];
            const f3 kD = tex_eval<kCount>(S, mtl.diffuse, hit.uvw, tally);
            const f3 kS = tex_eval<kCount>(S, mtl.specular, hit.uvw, tally);
            const f3 kT = tex_eval<kCount>(S, mtl.refraction, hit.uvw, tally);
            const f3 kE = tex_eval<kCount>(S, mtl.emission, hit.uvw, tally);
            normal += contrib * hit.n;
            albedo += contrib * kD + kE;
            target += contrib * kE;

            // lobe probabilities (renderer/shading.cu:17-29 and 122-134)
            const float third = 1.0f / 3.0f;
            float pD = (kD.x + kD.y + kD.z) * third;
            float pS = (kS.x + kS.y + kS.z) * third;
            float pT = (kT.x + kT.y + kT.z) * third;
            if (pD + pS + pT > CRT_MAX_SURVIVAL) {
                const float norm = CRT_MAX_SURVIVAL / (pD + pS + pT);
                pD *= norm; pS *= norm; pT *= norm;
            }
            const float g = mtl.glossiness;

            // next-event estimation toward one light (trace.cu:112-143).  Ambient and
            // directional lights produce no sample and zero weight, so nothing is traced.
            if (S.light_count > 0) {
                const int light_id = crtm::f2i_sat(light_pick * (float)S.light_count);
                const DevLight L = S.lights[light_id];
                if (L.kind == CRT_LIGHT_POINT) {
                    // PointLight::GenerateSample (scene/lights.cu:10-34)
                    const f3 lp = ld3(L.pos);
                    const f3 Ldir = normalized(hit.pos - lp);
                    const float x = rng_float(rng);
                    const float phi = rng_float(rng) * 2 * F_PI;
                    const float cos_theta = 1 - x;
                    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
                    f3 a, b;
                    orthonormals(Ldir, a, b);
                    const f3 n = Ldir * cos_theta + a * sin_theta * crtm::det_cosf(phi) + b * sin_theta * crtm::det_sinf(phi);
                    const f3 tgt = lp + n * L.size;
                    const f3 ldir = normalized(tgt - hit.pos);
                    const float d2 = lengthsq(tgt - hit.pos);
                    const float cos_theta_l = dot(ldir, -normalized(tgt - lp));
                    const float l_prob = d2 / (2 * F_PI * L.size * L.size * cos_theta_l);
                    const f3 l_mult = ld3(L.radiance);
                    const float l_dist = sqrtf(d2);
                    // Material::GetSampleInfo (renderer/shading.cu:116-150)
                    float b_prob = 0.0f;
                    f3 b_mult = mk3(0, 0, 0);
                    const float ndl = dot(hit.n, ldir);
                    if (ndl >= 0) {
                        b_prob += pD * (ndl > 0 ? OVERPI : 0);
                        b_mult += kD * OVERPI * ndl;
                        const f3 rn = normalized(v + ldir);
                        const float ndh = dot(hit.n, rn);
                        b_prob += pS * (g + 1) * 0.5f * OVERPI * crtm::det_powf(ndh, g + 1);
                        b_mult += kS * (g + 2) * 0.125f * OVERPI * crtm::det_powf(ndh, g);
                    }
                    // TraceShadowRay (trace.cu:160-196)
                    f3 spos = hit.pos, sdir = ldir;
                    spos += sdir * CRT_BIAS + hit.n * CRT_BIAS;
                    tally.shadow_rays++;
                    HitRec dummy;
                    dummy.z = CRT_BIGFLOAT;
                    const bool occluded = trace_scene<true, kCount>(S, spos, sdir, l_dist, dummy, stack_ref, stack_t, tally);
                    if (!occluded) {
                        const float denom = l_prob * l_prob + b_prob * b_prob;
                        if (denom > CRT_F_EPS && l_prob > CRT_F_EPS) {
                            const float pop = l_prob / denom;
                            target += contrib * l_mult * b_mult * pop;
                        }
                    }
                }
            }

            // Material::GenerateSample (renderer/shading.cu:11-113)
            f3 kA = mk3(1, 1, 1);
            if (!hit.front) {
                const f3 e = ld3(mtl.absorption) * -hit.z;
                kA = mk3(crtm::det_expf(e.x), crtm::det_expf(e.y), crtm::det_expf(e.z));
            }
            const float p = rng_float(rng);
            f3 ndir = mk3(0, 0, 0), n_mult = mk3(0, 0, 0);
            float n_prob = 0;
            bool alive = true;
            if (p < pD) {
                const float x = rng_float(rng);
                const float phi = rng_float(rng) * 2 * F_PI;
                const float cos_theta = 1 - x;
                const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
                f3 u, w;
                orthonormals(hit.n, u, w);
                ndir = hit.n * cos_theta + u * sin_theta * crtm::det_cosf(phi) + w * sin_theta * crtm::det_sinf(phi);
                n_prob = pD * OVERPI;
                n_mult = kA * kD * OVERPI * dot(hit.n, ndir);
            } else {
                const f3 gln = glossy_normal(hit.n, g, rng);
                if (p < pD + pS) {
                    ndir = reflect(v, gln);
                    const float pw = crtm::det_powf(dot(gln, hit.n), g);
                    n_prob = pS * (g + 1) * 0.125f * OVERPI * pw;
                    n_mult = kA * kS * (g + 2) * 0.125f * OVERPI * pw;
                } else if (p < pD + pS + pT) {
                    const float eta_i = hit.front ? 1 : mtl.ior;
                    const float eta_o = hit.front ? mtl.ior : 1;
                    const f3 h = hit.front ? gln : -gln;
                    bool tir = false;
                    ndir = transmit(v, h, eta_i, eta_o, tir);
                    const float f_0 = crtm::det_powf((eta_i - eta_o) / (eta_i + eta_o), 2);
                    const float cos_theta = dot(v, h);
                    float f_theta = f_0 + (1 - f_0) * crtm::det_powf(1 - cos_theta, 5);
                    f_theta = fminf(fmaxf(f_theta, 0.0f), 1.0f);
                    if (!tir) {
                        if (rng_float(rng) >= f_theta) {
                            n_prob = pT * (1 - f_theta);
                            n_mult = kA * kT * (1 - f_theta);
                        } else {
                            ndir = reflect(v, h);
                            n_prob = pT * f_theta;
                            n_mult = kA * kT * f_theta;
                        }
                    } else {
                        n_prob = pT;
                        n_mult = kA * kT;
                    }
                } else {
                    alive = false;
                }
            }
            if (!alive) path_done = true;
            else {
                rpos = hit.pos;
                rdir = ndir;
                if (n_prob <= CRT_F_EPS) path_done = true;
                else {
                    contrib *= n_mult * (1.0f / n_prob);
                    if (++segments >= CRT_MAX_SEGMENTS) path_done = true;
                }
            }
        }

        if (path_done) {
            new_path = true;
            if (++sample >= samples) {
                float* r = S.results + 3ull * pI;
                float* nn = S.normals + 3ull * pI;
                float* al = S.albedos + 3ull * pI;
                r[0] = target.x; r[1] = target.y; r[2] = target.z;
                nn[0] = normal.x; nn[1] = normal.y; nn[2] = normal.z;
                al[0] = albedo.x; al[1] = albedo.y; al[2] = albedo.z;
                have_pixel = false;
            }
        }
    }

    // ---- counters: one atomic per wave and field
    const unsigned long long c_closest = wave_sum(tally.closest_rays), c_shadow = wave_sum(tally.shadow_rays);
    unsigned long long c_paths = 0, c_nb = 0, c_ne = 0, c_bi = 0, c_bl = 0, c_tt = 0, c_th = 0, c_sp = 0, c_pl = 0, c_tx = 0;
    if (kCount) {
        c_paths = wave_sum(tally.paths); c_nb = wave_sum(tally.node_box_tests); c_ne = wave_sum(tally.node_enters);
        c_bi = wave_sum(tally.bvh_interior); c_bl = wave_sum(tally.bvh_leaves); c_tt = wave_sum(tally.tri_tests);
        c_th = wave_sum(tally.tri_hits); c_sp = wave_sum(tally.sphere_tests); c_pl = wave_sum(tally.plane_tests);
        c_tx = wave_sum(tally.texel_quads);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&stats->closest_rays, c_closest);
        atomicAdd(&stats->shadow_rays, c_shadow);
        if (kCount) {
            atomicAdd(&stats->paths, c_paths); atomicAdd(&stats->node_box_tests, c_nb); atomicAdd(&stats->node_enters, c_ne);
            atomicAdd(&stats->bvh_interior, c_bi); atomicAdd(&stats->bvh_leaves, c_bl); atomicAdd(&stats->tri_tests, c_tt);
            atomicAdd(&stats->tri_hits, c_th); atomicAdd(&stats->sphere_tests, c_sp); atomicAdd(&stats->plane_tests, c_pl);
            atomicAdd(&stats->texel_quads, c_tx);
        }
    }
}

// ------------------------------------------------------------------ parity hook: first camera ray per pixel
__global__ __launch_bounds__(CRT_BLOCK) void primary_kernel(DevScene S, DevPartition P, const uint32_t* __restrict__ jump,
                                                            int32_t* __restrict__ out_node, int32_t* __restrict__ out_face,
                                                            float* __restrict__ out_t) {
    extern __shared__ uint32_t lds[];
    uint32_t* stack_ref = lds;
    float* stack_t = reinterpret_cast<float*>(lds + S.stack_entries * CRT_BLOCK);
    const uint32_t item = blockIdx.x * CRT_BLOCK + threadIdx.x;
    uint32_t pX, pY;
    if (item >= P.total_items || !item_to_pixel(S, P, item, pX, pY)) return;
    const uint32_t pI = pY * S.width + pX;
    Rng rng;
    rng_init(rng, pI, jump);
    f3 rpos, rdir;
    camera_ray(S, pX, pY, rng, rpos, rdir);
    HitRec hit;
    hit_init(hit);
    Tally tally = {};
    rpos += rdir * CRT_BIAS;
    const bool any = trace_scene<false, false>(S, rpos, rdir, CRT_BIGFLOAT, hit, stack_ref, stack_t, tally);
    int32_t id = -1;
    if (any) id = hit.light >= 0 ? -2 - hit.light : hit.node;
    out_node[pI] = id;
    out_face[pI] = (any && hit.light < 0) ? hit.face : -1;
    out_t[pI] = hit.z;
}

// ------------------------------------------------------------------ ConvertColors (math/color.cu:7-23)
DEV float linear_to_srgb(float ch) { return ch < 0.0031308f ? ch * 12.92f : crtm::det_powf(ch, 0.41666f) * 1.055f - 0.055f; }
DEV uint8_t clamp_to_byte(float f) {
    const int v = crtm::f2i_sat(f);
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

__global__ __launch_bounds__(CRT_BLOCK) void convert_kernel(const float* __restrict__ in, uint8_t* __restrict__ out,
                                                            uint32_t n, int srgb, float pass_mul) {
    const uint32_t i = blockIdx.x * CRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 c = ld3(in + 3ull * i) * pass_mul;
    if (srgb) c = mk3(linear_to_srgb(c.x), linear_to_srgb(c.y), linear_to_srgb(c.z));
    out[3ull * i] = clamp_to_byte(c.x * 256);
    out[3ull * i + 1] = clamp_to_byte(c.y * 256);
    out[3ull * i + 2] = clamp_to_byte(c.z * 256);
}

// ------------------------------------------------------------------ self tests
__global__ void rng_selftest_kernel(const uint32_t* __restrict__ seeds, const uint32_t* __restrict__ jump_index,
                                    const uint32_t* __restrict__ jumps, uint32_t n, uint32_t draws, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r;
    const uint32_t ji = jump_index[i];
    rng_init(r, seeds[i], ji == 0xffffffffu ? nullptr : jumps + (size_t)ji * 800);
    uint32_t* o = out + (size_t)i * (6 + draws);
    o[0] = r.d; o[1] = r.v0; o[2] = r.v1; o[3] = r.v2; o[4] = r.v3; o[5] = r.v4;
    for (uint32_t d = 0; d < draws; d++) o[6 + d] = rng_next(r);
}

__global__ void math_selftest_kernel(int fn, const float* __restrict__ x, const float* __restrict__ y, uint32_t n,
                                     float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = crtm::det_sinf(x[i]); break;
        case 1: r = crtm::det_cosf(x[i]); break;
        case 2: r = crtm::det_expf(x[i]); break;
        case 3: r = crtm::det_powf(x[i], y[i]); break;
        case 4: r = crtm::det_atan2f(x[i], y[i]); break;
        case 5: r = crtm::det_asinf(x[i]); break;
        case 6: r = crtm::det_acosf(x[i]); break;
        default: r = crtm::det_tanf(x[i]); break;
    }
    out[i] = r;
}

}  // namespace crt
