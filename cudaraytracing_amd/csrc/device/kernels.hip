// gfx950 kernels of the per-pixel trace path.
//
// trace_kernel replaces TracePrimaryRays (renderer/trace.cu:7-41) and every
// device function it reaches.  It is a persistent kernel, one 1024-thread block
// per CU.  A block owns a pool of pixel slots: a slot holds one pixel at a time
// and runs that pixel's samples one after another (they share one random stream,
// renderer/trace.cu:22-25); a finished slot takes the next pixel of a global
// queue with one wave-aggregated atomic (ballot + mbcnt prefix).
//
// Slots, not lanes, own pixels: everything a pixel carries from one ray query to
// the next lives in the slot's 320-byte record in global memory.  And queries,
// not lanes, own traversal state: a query in flight has a context in LDS (112
// bytes of state and a column of the traversal stack).  The work of a query is
// cut into three operations - BOX: slab-test the two children of a BVH node (one
// 64-byte record); LEAF: up to two triangles of a leaf; NODE: one step of the
// scene-node scan (leave the mesh just walked, test the next node's box, enter
// it: transform, sphere / plane test or BVH root) - shared by closest-hit and
// shadow queries.
//   * A traversal wave runs ONE operation per turn with as many lanes as it can
//     fill.  A lane loads what the operation reads from its query's context,
//     takes the step and stores what changed; a lane whose query now waits for
//     another operation parks the context in that operation's pool (a bit per
//     context, one word per lane position, one LDS atomic) and takes a parked
//     context that waits for the wave's operation.  Contexts stay with their lane
//     position (and its partner in the other half of the wave), so context
//     records and stack columns are spread over the LDS banks like the lanes.
//     Nothing of a query stays in a lane's registers from turn to turn.
//   * Shading waves run everything between two queries of a pixel (hit
//     attributes, textures, light and BSDF sampling, Russian roulette, the camera
//     ray of a new path, the next pixel of a finished slot) for up to 64 slots
//     at a time, in turns that are all of one kind (closest-hit query ended /
//     shadow query ended or pixel needed).  A new query goes straight into a free
//     context and into the pool of its first operation (the slot's RAY record
//     and the ray ring are the fall-back when no context is within reach); a
//     query that has ended is written into the slot's HIT record by a traversal
//     wave's exchange turn, and the slot goes into one of the two shading queues
//     (rings in LDS: positions reserved with one atomic per wave, entries
//     compacted with ballot + mbcnt).
//   * Because a traversal wave holds nothing between turns, the last few of them
//     take shading turns whenever the shading queues run long: the split between
//     the two kinds of work follows the frame.
// When a block runs short of pixels, a pixel's shadow query and the closest-hit
// query of its next path segment run side by side (they are independent; the
// results are joined in the reference's order), which shortens the chain of
// dependent queries that bounds the end of a frame.
// The order of operations of any one pixel is exactly the reference's; only their
// interleaving across pixels is scheduled.  (Trees deeper than the LDS stack run
// the kernel variants without contexts - kRegroup false - in which a traversal
// lane keeps its query in registers from start to end and the wave votes on the
// operation of each turn.)
//
// Arithmetic keeps the reference's operation order (see crt_vec.h); the file is
// compiled with -ffp-contract=off and uses the pinned transcendentals of
// crt_detmath.h, so results equal the CPU oracle bit for bit.
//
// One variant does not: kTlas, chosen by the caller for scenes of very many nodes
// (crt_set_tlas).  Its queries walk a BVH over the scene's object nodes - BOX
// steps over box pairs as inside a mesh, a NODE step in and out of every object
// the walk reaches - instead of scanning the node list from end to end
// (renderer/trace.cu:51, 166), and cull boxes by the distance at which the ray
// enters them.  Objects are met in another order than the reference meets them,
// so ties and last bits differ.
#include <hip/hip_runtime.h>

#include <float.h>

#include <type_traits>

#include "dev_scene.h"

namespace crt {

#ifndef CRT_BLOCK
#define CRT_BLOCK 1024               // threads of a trace-kernel block: one block per CU, so that all traversal and shading waves of a CU share one pool of pixel slots (the element-wise kernels use it too)
#endif
#ifndef CRT_WAVES_PER_SIMD
#define CRT_WAVES_PER_SIMD 4        // register budget 128: the best measured balance of occupancy and spills
#endif
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
    uint32_t turns[12], lanes[12];   // per operation: 0 box, 1 triangle, 2 transform, 3 shading, 4 exchange, 5 pixel fetch, 8 park, 9 unpark
    uint32_t scan_checks, scan_violations;
    unsigned long long cycles[12];
};
#define TALLY(field) do { if (kCount) tally.field++; } while (0)

DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// Box slab test (Box::IntersectRay / IntersectShadowRay, scene/rays.cuh:138-187) with the
// reciprocal direction hoisted; dist is meaningful only when the test passes.  Each argument is the
// (min, max) pair of one axis: subtraction and multiplication run as two-float packed instructions
// (v_pk_add_f32 / v_pk_mul_f32), which round each half exactly like the scalar forms.
typedef float f2 __attribute__((ext_vector_type(2)));
// (enter: where the ray enters the box, negative for a ray that starts inside it - the reference reports the way OUT
// as that ray's distance, and culls by it)
DEV bool slab(f2 bx, f2 by, f2 bz, f3 pos, f3 inv, float tmax, float& dist, float& enter) {
    const f2 tx = (bx - pos.x) * inv.x, ty = (by - pos.y) * inv.y, tz = (bz - pos.z) * inv.z;
    const float cx = fminf(tx.x, tx.y), cy = fminf(ty.x, ty.y), cz = fminf(tz.x, tz.y);
    const float fx = fmaxf(tx.x, tx.y), fy = fmaxf(ty.x, ty.y), fz = fmaxf(tz.x, tz.y);
    const float tEnter = fmaxf(cx, fmaxf(cy, cz));
    const float tExit = fminf(fx, fminf(fy, fz));
    dist = tEnter >= -FLT_EPSILON ? tEnter : tExit;
    enter = tEnter;
    return (tEnter >= -FLT_EPSILON || tExit >= -FLT_EPSILON) && tEnter <= tExit && tEnter <= tmax;
}
DEV bool slab(f2 bx, f2 by, f2 bz, f3 pos, f3 inv, float tmax, float& dist) {
    float enter;
    return slab(bx, by, bz, pos, inv, tmax, dist, enter);
}

// Node::ToLocal / FromLocal for a whole ray (scene/scene.cuh:150-189): x, y, z hold (position, direction) pairs;
// both are multiplied by the 3x3 part with the operation order of xform_point / xform_dir, then the position
// half gets the translation.
DEV void xform_ray(const float* c, f2& x, f2& y, f2& z) {
    f2 nx = c[0] * x + c[3] * y + c[6] * z;
    f2 ny = c[1] * x + c[4] * y + c[7] * z;
    f2 nz = c[2] * x + c[5] * y + c[8] * z;
    nx.x = nx.x + c[9]; ny.x = ny.x + c[10]; nz.x = nz.x + c[11];
    x = nx; y = ny; z = nz;
}

DEV f3 recip3(f3 d) {
    // three reciprocals side by side, and one common way out for the rare operand the short form does not cover
    f3 r = mk3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));
    if (!(rcp_fast_ok(r.x) && rcp_fast_ok(r.y) && rcp_fast_ok(r.z))) {
        asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
        r = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    }
    return r;
}

// Unit sphere / unit plane against a ray in the object's own space (Sphere::IntersectRay / IntersectShadowRay,
// Plane::IntersectRay / IntersectShadowRay: scene/objects.cu:7-159).  hz: the closest hit so far.
DEV bool unit_object_hit(bool sphere, f3 pos, f3 dir, bool shadow, float limit, float hz, float& t, bool& front) {
    bool got = false;
    front = true; t = 0;
    if (sphere) {
        if (!(dot(dir, pos) >= 0)) {
            const float a = dot(dir, dir);
            const float b = dot(dir, pos) * 2;
            const float c = dot(pos, pos) - 1;
            const float det = b * b - 4 * a * c;
            if (!(det < 0)) {
                const float sq = sqrtf(det);
                const float tf = (-b - sq) / (2 * a);
                const float tb = (-b + sq) / (2 * a);
                if (shadow) got = (tf >= 0 && tf <= limit) || (tb >= 0 && tb <= limit);
                else if (tf >= 0) { if (!(tf >= hz)) { got = true; t = tf; front = true; } }
                else if (!(tb < 0) && !(tb >= hz)) { got = true; t = tb; front = false; }
            }
        }
    } else {
        if (!(dir.z == 0)) {
            t = -pos.z / dir.z;
            const bool in_range = shadow ? !(t < 0 || t >= limit) : !(t < 0 || t >= hz);
            if (in_range && ((dir.z < 0) || (dir.z > 0))) {
                const f3 p = pos + t * dir;
                if (!(fabsf(p.x) > 1 || fabsf(p.y) > 1)) { got = true; front = dir.z < 0; }
            }
        }
    }
    return got;
}

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
    f3 c00, c01, c10, c11;
    if (w * h == 1u) {
        // a constant colour (what every colour attribute of a material becomes): its one texel travels in the
        // descriptor; the four taps and their weights are still combined as for any texture
        c00 = mk3((float)(t.first & 255u), (float)((t.first >> 8) & 255u), (float)((t.first >> 16) & 255u)) * (1.0f / 256.0f);
        c01 = c00; c10 = c00; c11 = c00;
    } else {
        const uint8_t* base = S.texels + t.offset;
        c00 = texel(base + 3u * ((uint32_t)iy * w + (uint32_t)ix));
        c01 = texel(base + 3u * ((uint32_t)iy * w + (uint32_t)ixp));
        c10 = texel(base + 3u * ((uint32_t)iyp * w + (uint32_t)ix));
        c11 = texel(base + 3u * ((uint32_t)iyp * w + (uint32_t)ixp));
    }
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
    const float cos_theta = crtm::det_powf(1 - x, rcp_ieee(glossiness + 1.0f));
    f3 u, v;
    orthonormals(normal, u, v);
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    float sin_phi, cos_phi;
    crtm::det_sincosf(phi, sin_phi, cos_phi);
    return normal * cos_theta + u * sin_theta * cos_phi + v * sin_theta * sin_phi;
}

// ------------------------------------------------------------------ camera (renderer/trace.cu:14-38)
DEV void camera_ray(const DevScene& S, uint32_t pX, uint32_t pY, Rng& rng, f3& pos, f3& dir) {
    const f3 cX = ld3(S.cX), cY = ld3(S.cY);
    const f3 pixel = ld3(S.top_left) + S.pixel_size * ((float)pX * cX - (float)pY * cY);
    const float F_PI = 3.14159265358979323846f;
    const float dof_r = sqrtf(rng_float(rng));
    const float dof_theta = rng_float(rng) * 2 * F_PI;
    float dof_sin, dof_cos;
    crtm::det_sincosf(dof_theta, dof_sin, dof_cos);
    const float dof_x = dof_r * dof_cos;
    const float dof_y = dof_r * dof_sin;
    const f3 start = ld3(S.cam_pos) + S.dof * (dof_x * cX - dof_y * cY);
    const float aa_x = rng_float(rng) - 0.5f;
    const float aa_y = rng_float(rng) - 0.5f;
    const f3 end = pixel + S.pixel_size * (aa_x * cX - aa_y * cY);
    pos = start;
    dir = end - start;
}

// queue item -> pixel; false when the item falls outside the image.  tile is the tile's index in
// storage order (strip-major over the strips this context owns).
DEV bool item_to_pixel(const DevScene& S, const DevPartition& P, uint32_t item, uint32_t& pX, uint32_t& pY, uint32_t& tile) {
    const uint32_t lane = item & 63u;
    tile = item >> 6;
    if (P.tile_order) tile = P.tile_order[tile];
    const uint32_t tiles_per_strip = P.items_per_strip >> 6;
    const uint32_t strip_local = tile / tiles_per_strip;
    const uint32_t in_strip = tile - strip_local * tiles_per_strip;
    const uint32_t tile_row = in_strip / P.tiles_x, tile_col = in_strip - tile_row * P.tiles_x;
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

// ------------------------------------------------------------------ candidate mask of the scene-node scan
// The reference tests a ray against the box of every scene node in list order (renderer/trace.cu:51-68,
// 166-193).  A node whose box is missed has no effect on the query, so the scan may skip any node that is
// CERTAIN to be missed.  "Certain" has to hold for the ray as it will be when the scan reaches the node: every
// node the ray enters sends it through ToLocal and FromLocal, which moves it by rounding errors.  S.drift_rel /
// S.drift_abs bound that movement for a whole query, so a node is dropped only if the slab test fails for every
// ray within that distance of the ray the query starts with:
//   * the box is grown by the position bound;
//   * an axis whose direction component is within 64 x the direction bound of zero puts no constraint on t
//     (its sign, and with it the order of the two slab planes, could change);
//   * on the other axes 1 / dir moves by less than 1 / 63 relative; entry and exit distance are widened by 1 / 32
//     of their magnitude, which also covers the rounding of the test itself (x -> x +- |x| / 32 is monotone, so
//     widening the maximum / minimum of the per-axis values is widening each of them);
//   * the three acceptance conditions of Box::IntersectRay are then applied to the widened interval.
// A shadow ray also leaves out the subtree of an object node that is certain to be missed, exactly as the scan
// does for every object node it misses (trace.cu:171-185).  Bit i of the result: node i has to be tested.
// The counted build does not scan by the mask; it checks every box hit of its full scan against it.
// Scenes of more than 128 nodes run the same test over the boxes of groups of consecutive nodes (`pairs` is then the
// table of group boxes, `n` their number, no subtree rule): the scan walks the nodes of candidate groups one by one,
// as the reference does, and passes over the other groups.
struct NodeMask {
    unsigned long long lo, hi;      // bit i of lo: node i; bit i of hi: node 64 + i (scenes of 65 to 128 nodes)
};

DEV NodeMask scan_candidates(const DevScene& S, const BoxPair* __restrict__ pairs, uint32_t n, f3 pos, f3 dir, float tmax, bool shadow) {
    const float pmax = fmaxf(fabsf(pos.x), fmaxf(fabsf(pos.y), fabsf(pos.z)));
    const float dmax = fmaxf(fabsf(dir.x), fmaxf(fabsf(dir.y), fabsf(dir.z)));
    const float dp = S.drift_rel * pmax + S.drift_abs;
    const float dd = 64.0f * (S.drift_rel * dmax);
    // anything not finite: test every node
    if (!(pmax < 1e30f) || !(dmax < 1e30f) || !(dmax > 0.0f)) return NodeMask{~0ull, ~0ull};
    const bool cx = fabsf(dir.x) >= dd && fabsf(dir.x) > 1e-30f, cy = fabsf(dir.y) >= dd && fabsf(dir.y) > 1e-30f, cz = fabsf(dir.z) >= dd && fabsf(dir.z) > 1e-30f;
    // unconstrained axis: (min - BIG) * 1 and (max + BIG) * 1 give -BIG and +BIG
    const float big = 1e37f;
    const f2 ox = cx ? f2{pos.x + dp, pos.x - dp} : f2{big, -big};
    const f2 oy = cy ? f2{pos.y + dp, pos.y - dp} : f2{big, -big};
    const f2 oz = cz ? f2{pos.z + dp, pos.z - dp} : f2{big, -big};
    // (a filter: the hardware's reciprocal estimate, 1 ulp, is well inside the 1 / 32 by which the interval is widened)
    const float ix = cx ? __builtin_amdgcn_rcpf(dir.x) : 1.0f, iy = cy ? __builtin_amdgcn_rcpf(dir.y) : 1.0f, iz = cz ? __builtin_amdgcn_rcpf(dir.z) : 1.0f;
    const float wide = 1.0f / 32.0f;
    // plane distances as one fused multiply-add per pair, box * (1 / dir) - origin * (1 / dir): the product on the right
    // is rounded once more than in (box - origin) * (1 / dir), by at most 2^-24 of its magnitude, so 2^-22 of the
    // largest such product (over the constrained axes) goes into the margin
    const f2 oix = ox * ix, oiy = oy * iy, oiz = oz * iz;
    const float reach = fmaxf(cx ? fmaxf(fabsf(oix.x), fabsf(oix.y)) : 0.0f, fmaxf(cy ? fmaxf(fabsf(oiy.x), fabsf(oiy.y)) : 0.0f, cz ? fmaxf(fabsf(oiz.x), fabsf(oiz.y)) : 0.0f));
    const float margin = __builtin_fmaf(reach, 0x1p-22f, 1e-6f);
    uint32_t skip_until = 0;
    // The node records never change during a launch and the index is the same in every lane: they are read through
    // the constant address space, i.e. with scalar loads, a whole 64-byte record (nodes i and i + 1) at a time.
    typedef uint32_t u16v __attribute__((ext_vector_type(16)));
    const __attribute__((address_space(4))) u16v* recs = reinterpret_cast<const __attribute__((address_space(4))) u16v*>(reinterpret_cast<uintptr_t>(pairs));
    auto one = [&](uint32_t word, uint32_t i, uint32_t ref, f2 bx, f2 by, f2 bz) -> uint32_t {
        const f2 tx = __builtin_elementwise_fma(bx, f2{ix, ix}, -oix), ty = __builtin_elementwise_fma(by, f2{iy, iy}, -oiy), tz = __builtin_elementwise_fma(bz, f2{iz, iz}, -oiz);
        float t_in = fmaxf(fminf(tx.x, tx.y), fmaxf(fminf(ty.x, ty.y), fminf(tz.x, tz.y)));
        float t_out = fminf(fmaxf(tx.x, tx.y), fminf(fmaxf(ty.x, ty.y), fmaxf(tz.x, tz.y)));
        // (this is a filter, not path arithmetic: fused multiply-add is fine here)
        t_in = __builtin_fmaf(-wide, fabsf(t_in), t_in - margin);
        t_out = __builtin_fmaf(wide, fabsf(t_out), t_out + margin);
        // a NaN bound (cannot arise from finite inputs) fails all three comparisons and so keeps the node
        const bool cand = !((t_in > t_out) | (t_out < -2.0f * FLT_EPSILON) | (t_in > tmax));
        const bool live = i >= skip_until;
        if (shadow & live & !cand) skip_until = node_ref_after(ref);
        return word | ((live & cand) ? 1u << (i & 31u) : 0u);
    };
    auto as_f = [](uint32_t v) { return __uint_as_float(v); };
    auto half = [&](uint32_t first, uint32_t last) -> uint32_t {
        uint32_t word = 0;
        for (uint32_t i = first; i < last; i += 2) {
            const u16v r = recs[i];
            const uint32_t ref_a = r[12], ref_b = r[13];
            if (node_ref_kind(ref_a) != CRT_OBJ_NONE) word = one(word, i, ref_a, f2{as_f(r[0]), as_f(r[1])}, f2{as_f(r[2]), as_f(r[3])}, f2{as_f(r[4]), as_f(r[5])});
            if (node_ref_kind(ref_b) != CRT_OBJ_NONE) word = one(word, i + 1, ref_b, f2{as_f(r[6]), as_f(r[7])}, f2{as_f(r[8]), as_f(r[9])}, f2{as_f(r[10]), as_f(r[11])});
        }
        return word;
    };
    auto upto = [&](uint32_t limit) -> uint32_t { return n < limit ? n : limit; };
    const uint32_t w0 = half(0, upto(32));
    const uint32_t w1 = n > 32u ? half(32, upto(64)) : 0u;
    const uint32_t w2 = n > 64u ? half(64, upto(96)) : 0u;
    const uint32_t w3 = n > 96u ? half(96, upto(128)) : 0u;
    return NodeMask{((unsigned long long)w1 << 32) | w0, ((unsigned long long)w3 << 32) | w2};
}

DEV void acc_add(float* __restrict__ p, f3 v) {
    p[0] += v.x; p[1] += v.y; p[2] += v.z;
}

// ------------------------------------------------------------------ per-lane state
// Operation a traversal lane needs next.
//   BOX    slab-test the two children of a BVH interior node (one box-pair record)
//   LEAF   test the next one or two triangles of the pending leaf
//   NODE   one step of the scene-node scan: leave the mesh just walked, test the next node's box, enter the node
//          (world -> local, sphere / plane test or BVH root)
//   POST   the query has ended: its result has to be handed to the shading waves
//   IDLE   the lane has no query: it takes the next one from the block's ray queue
//   EMPTY  (regrouping kernel) the lane holds no query context; an IDLE lane holds a free one
enum : uint32_t { S_BOX = 0, S_LEAF = 1, S_NODE = 2, S_POST = 3, S_IDLE = 4, S_EMPTY = 5 };
enum : uint32_t { H_NONE = 0, H_SPHERE = 1, H_PLANE = 2, H_MESH = 3 };
// why a slot is in the shading queue
enum : uint32_t { PK_NEW_PATH = 0, PK_CLOSEST_DONE = 1, PK_SHADOW_CLEAR = 2, PK_SHADOW_BLOCKED = 3, PK_FETCH = 4 };
// what a shading turn decided for a slot
enum : uint32_t { R_NONE = 0, R_RAY = 1, R_REFETCH = 2, R_RETIRE = 3, R_TWO_RAYS = 4 };
constexpr uint32_t kPendExit = 0x20000000u;     // NODE: leave the mesh of node `entered` first

struct Lane {
    uint32_t col;                   // stack column: the lane's own, or (regrouping kernel) the query context the lane holds
    uint32_t slot;                  // pixel slot whose query this lane is running
    uint32_t pk;                    // how the query ended
    // current ray query (closest hit or shadow)
    // Ray in the space it is currently in, kept as (position, direction) pairs per axis so that transforms and
    // the triangle test run on two-float packed instructions; and 1 / direction.
    f2 pdx, pdy, pdz;
    f3 qinv;
    DEV f3 qpos() const { return mk3(pdx.x, pdy.x, pdz.x); }
    DEV f3 qdir() const { return mk3(pdx.y, pdy.y, pdz.y); }
    DEV void set_qpos(f3 p) { pdx.x = p.x; pdy.x = p.y; pdz.x = p.z; }
    DEV void set_qdir(f3 d) { pdx.y = d.x; pdy.y = d.y; pdz.y = d.z; }
    float limit;                    // slab t_max: FLT_MAX for closest hit, the light distance for shadow rays
    uint32_t node;                  // next scene node to scan
    unsigned long long mask;        // scene nodes the scan still has to test (bit i = node i), when it runs by candidates
    unsigned long long mask_hi;     // the same for nodes 64..127 (scenes of more than 64 nodes)
    uint32_t cur;                   // BVH record to expand
    uint32_t sp;                    // stack height
    uint32_t lo, top;               // deep trees only: lowest stack index still in LDS, ring position of index sp
    uint32_t entered;               // scene node whose mesh is being traversed
    uint32_t pend;                  // LEAF: remaining leaf range; NODE: kPendExit when a mesh has to be left first
    bool shadow, mesh_hit;
    bool more;                      // regrouping kernel: the scan has a further node to look at after the one the query is in
    // closest hit so far; normal / texture coordinates are derived from (h_*) when the query ends
    float hz;
    f3 hpos;                        // world space
    uint32_t h_kind, h_node;
    float h_a, h_b;                 // mesh: barycentrics b1, b2; sphere / plane: local hit x, y
    uint32_t h_c;                   // mesh: triangle record; sphere / plane: bits of local hit z
    bool h_front;
};

// ------------------------------------------------------------------ pixel slots and the block's two queues
// A block owns S.slots pixel slots.  A slot holds one pixel at a time and everything that pixel carries from one
// ray query to the next, as one 320-byte record in global memory:
//   words  0..15  RAY   the query to run: written by a shading lane, read by the traversal lane that takes it
//   words 16..31  HIT   how the query ended: written by that traversal lane, read by a shading lane
//   words 32..59  PATH  random stream (6), throughput (3), view vector (3), sample, segment, cost, pixel; then what is
//                       parked while a shadow query runs (continuation direction, flags, light sample, hit
//                       position) and the pixel's tile: only the shading waves touch it
//   word      60  how the shadow query of a pair of queries ended
//   words 64..79  RAY2  the shadow query that runs beside the closest-hit query of RAY (below)
// Slots, not lanes, own pixels: a traversal lane that finishes a query posts the slot to the block's shading queue
// and takes whatever query is ready next, a shading lane does the opposite, so both kinds of wave run with their
// lanes full while any one pixel still sees its operations in the reference's order.  Both queues are rings of
// slot numbers in LDS; an entry carries the lap number of its position, so a consumer that has claimed a position
// knows when the producer's write has landed without a second counter.
constexpr uint32_t kQueueLog = 11, kQueueSize = 1u << kQueueLog;      // ring positions; slots per block <= kQueueSize
constexpr uint32_t kDeepQueueLog = 10;      // the kernel variant for deep trees: half the slots at most, and the LDS that saves goes to the traversal stack
constexpr uint32_t kSlotWords = 80;
constexpr uint32_t kRecRay = 0, kRecHit = 16, kRecPath = 32, kRecShadowEnd = 60, kRecRay2 = 64;
// A slot may have two queries under way at once: the shadow query of a light sample and the closest-hit query of the
// next path segment do not depend on each other (the reference runs them one after the other, trace.cu:137-156).
// The shadow query of such a pair is read from RAY2 and announced in the ray queue as slot | (ring size of the slot queues), its
// result goes to word kRecShadowEnd; whichever of the two ends last queues the slot for shading (join_bits).
constexpr uint32_t kMaxSlots = kQueueSize, kMaxSlotsDeep = 1u << kDeepQueueLog;
constexpr uint32_t kLanePaired = 0x80000000u;        // in Lane::slot: the query is one of a pair
constexpr uint32_t kPkPaired = 0x100u;               // in the HIT record's first word, beside PK_CLOSEST_DONE: a shadow query ran beside this one
static_assert(kRecRay2 + 16 <= kSlotWords && kRecShadowEnd >= kRecPath + 28 && kRecShadowEnd < kRecRay2, "slot record layout");
constexpr uint32_t kMaxWaves = 16;
constexpr uint32_t kMaxContexts = 32 * 64;  // query contexts of a block (regrouping kernel): a multiple of 64, one 32-bit pool word per lane position
constexpr uint32_t kContextBytes = 112;      // the parked part of a context: 7 x 16 bytes

// The rings of a block, as indices into BlockCtl::head / tail (one array each, so that a wave reads all fills with
// two LDS loads).  kQBox .. kQIdle: query contexts parked by the operation they wait for (the values equal the lane
// states S_BOX .. S_IDLE); kQRay: slots whose next query waits for a traversal lane; kQShade0 / kQShade1: slots
// waiting for a shading lane, in two queues so that a shading turn is all of one kind - [0] a closest-hit query has
// ended (the full shading code), [1] a shadow query has ended or the slot needs a pixel (little code).
enum : uint32_t { kQBox = 0, kQLeaf = 1, kQNode = 2, kQPost = 3, kQIdle = 4, kQRay = 5, kQShade0 = 6, kQShade1 = 7 };

struct BlockCtl {
    uint32_t head[8], tail[8];
    uint32_t live;                      // slots that still have a pixel or may get one
    uint32_t abort;                     // a wave gave up waiting (a protocol error): every wave leaves
};

DEV uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Ring entries are 16 bits wide (LDS is what limits the traversal stack): the low kLog bits are the value, the bits
// above them the lap number of the position, truncated.  A position always holds what the previous lap wrote there
// until this lap's producer overwrites it, so the truncated lap number tells the two apart.
typedef uint16_t ring_t;
DEV uint32_t ring_ld(const ring_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void ring_st(ring_t* p, uint32_t v) { __hip_atomic_store(p, (ring_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Every lane with `want` appends its slot.  The caller has made the slot's record visible to the block
// (release fence) before the call.
template <uint32_t kLog>
DEV void queue_push(uint32_t* tail, ring_t* ring, bool want, uint32_t slot) {
    constexpr uint32_t kSize = 1u << kLog;
    const unsigned long long m = __ballot(want);
    if (m == 0) return;
    uint32_t base = 0;
    if (want && lane_rank(m) == 0) base = atomicAdd(tail, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (want) {
        const uint32_t idx = base + lane_rank(m);
        ring_st(&ring[idx & (kSize - 1u)], (((idx >> kLog) + 1u) << kLog) | slot);
    }
}

// The lanes with `want` claim up to that many entries, but none unless at least min_take are there; a lane that
// got one returns true and its slot.  One compare-and-swap per wave; when another wave wins it the call
// takes nothing and the caller comes back later.
template <uint32_t kLog>
DEV bool queue_pop(uint32_t* head, const uint32_t* tail, const ring_t* ring, bool want, uint32_t min_take, uint32_t slot_limit,
                   uint32_t& slot, uint32_t* abort_flag) {
    constexpr uint32_t kSize = 1u << kLog;
    const unsigned long long m = __ballot(want);
    if (m == 0) return false;
    const uint32_t nw = (uint32_t)__popcll(m);
    uint32_t h = 0, k = 0;
    if (want && lane_rank(m) == 0) {
        h = lds_ld(head);
        const uint32_t avail = lds_ld(tail) - h;
        k = avail < nw ? avail : nw;
        if (k < min_take || avail > kSize) k = 0;
        if (k && atomicCAS(head, h, h + k) != h) k = 0;
    }
    const int leader = __ffsll((long long)m) - 1;
    h = __shfl(h, leader, 64);
    k = __shfl(k, leader, 64);
    const uint32_t r = lane_rank(m);
    const bool got = want && r < k;
    if (got) {
        const uint32_t idx = h + r;
        const uint32_t tag = ((idx >> kLog) + 1u) & ((1u << (16 - kLog)) - 1u);
        uint32_t e = ring_ld(&ring[idx & (kSize - 1u)]), spins = 0;
        // the producer reserved this position before the count said so; its write is a few instructions away
        while ((e >> kLog) != tag) {
            if (++spins > (1u << 22)) { lds_st(abort_flag, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
            e = ring_ld(&ring[idx & (kSize - 1u)]);
        }
        slot = e & (kSize - 1u);
        if (slot >= slot_limit) slot = slot_limit - 1u;      // only after an abort: stay inside the block's records
    }
    return got;
}

// ------------------------------------------------------------------ the path tracer
// One launch runs every pixel of the partition.  The waves of a block have fixed roles: the first S.trav_waves
// run ray queries (BOX / LEAF / NODE operations, voted per turn), the others run everything between two
// queries of a pixel (shading, light and BSDF sampling, new camera rays, taking the next pixel from the global
// queue).  kDeep: the BVH is deeper than the LDS stack; entries above S.stack_entries live in a per-thread
// overflow area of the side buffer (rarely reached: a ray's stack is usually a fraction of the tree depth).  kWide: the
// scene has 65 to 128 nodes, so the candidate mask of the scene-node scan has a second word.
// kRegroup: queries, not lanes, own traversal state.  A query in flight has a context in LDS (its stack column and a
// 112-byte record); a wave keeps the queries that wait for the operation it is running in its lanes' registers and
// parks the others in the ring of the operation they wait for, where the waves that run that operation pick them up.
// kTlas (an opt-in mode of the regrouping kernel whose images are not the reference's bit for bit): queries walk a BVH
// over the scene's object nodes instead of scanning the node list; objects are met in the order of the walk.
template <bool kCount, bool kPrimary, bool kDeep, bool kWide, bool kRegroup, bool kTlas = false>
__global__ __launch_bounds__(CRT_BLOCK, CRT_WAVES_PER_SIMD) void trace_kernel(DevScene S, DevPartition P, int samples,
                                                             const uint32_t* __restrict__ jump,
                                                             uint32_t* __restrict__ queue, DevStats* __restrict__ stats,
                                                             int32_t* __restrict__ out_node, int32_t* __restrict__ out_face,
                                                             float* __restrict__ out_t, uint32_t* slot_mem) {
    static_assert(!(kDeep && kRegroup), "the regrouping kernel keeps the top of every stack in LDS already");
    static_assert(!kTlas || (kRegroup && !kWide), "the walk over a BVH of the nodes is a mode of the regrouping kernel and uses no candidate mask");
    extern __shared__ uint32_t lds[];
    __shared__ BlockCtl ctl;
    constexpr uint32_t kQLog = kDeep ? kDeepQueueLog : kQueueLog, kQSize = 1u << kQLog;      // ring of slots; slots per block <= kQSize
    constexpr uint32_t kRLog = kQLog + 1u, kRSize = 1u << kRLog, kSecond = kQSize;            // ring of ray queries: up to two per slot
    __shared__ ring_t shade_ring[2][kQSize], ray_ring[kRSize];
    // Parked query contexts (regrouping kernel).  Context c belongs to lane position c % 64 of every traversal wave: only
    // that lane of a wave parks it or takes it over, so stack columns and context records stay spread over the LDS
    // banks like the lanes.  pool[q][p]: bit j set = context j * 64 + p is parked and waits for operation q.
    __shared__ uint32_t pool[5][64];
    __shared__ uint32_t join_bits[kQSize / 32];      // bit = slot: one query of the slot's pair has ended
    __shared__ unsigned long long sect_clock[kMaxWaves], path_sections[kMaxWaves * 8], wave_clock[kMaxWaves * 2];      // counted build only
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    const uint32_t trav_threads = S.trav_waves * 64u;
    const uint32_t n_slots = S.slots;
    const uint32_t n_ctx = kRegroup ? S.contexts : trav_threads;       // stack columns
    uint32_t* const block_slots = slot_mem + (size_t)blockIdx.x * n_slots * kSlotWords;
    auto slot_rec = [&](uint32_t slot) -> uint32_t* { return block_slots + (size_t)slot * kSlotWords; };

    // ---- block set-up: every slot starts in the shading queue asking for a pixel, every context in the ring of free ones
    if (kCount) for (uint32_t i = tid; i < kMaxWaves * 8; i += CRT_BLOCK) path_sections[i] = 0;
    for (uint32_t i = tid; i < kRSize; i += CRT_BLOCK) ray_ring[i] = 0;
    for (uint32_t i = tid; i < kQSize; i += CRT_BLOCK) {
        shade_ring[0][i] = 0;
        shade_ring[1][i] = (ring_t)(i < n_slots ? ((1u << kQLog) | i) : 0u);
    }
    if (kRegroup && tid < 5 * 64) pool[tid >> 6][tid & 63u] = (tid >> 6) == kQIdle ? (uint32_t)((1ull << (n_ctx >> 6)) - 1ull) : 0u;
    for (uint32_t i = tid; i < n_slots; i += CRT_BLOCK) slot_rec(i)[kRecHit] = PK_FETCH;
    if (tid < kQSize / 32) join_bits[tid] = 0;
    if (tid < 8) { ctl.head[tid] = 0; ctl.tail[tid] = tid == kQShade1 ? n_slots : 0u; }
    if (tid == 0) { ctl.live = n_slots; ctl.abort = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const BoxPair* __restrict__ pairs = S.pairs;
    const TriRec* __restrict__ tris = S.tris;
    const uint32_t node_count = S.node_count;
    // the plain kernels scan the candidate nodes only; the counted one scans every node, as the reference does, and
    // checks its box hits against the mask
    const bool follow_mask = S.scan_mask != 0 && (!kCount || (S.scan_mask & 2u) != 0);      // bit 1: profiling, the counted build follows the mask too
    // mask bits are groups of nodes (more than 128 nodes); such scenes always run the kernel variant with two mask
    // words, so the one-word variant carries none of the code for it
    const bool grouped = kWide && (S.scan_mask & 4u) != 0;
    const bool by_mask = follow_mask && !grouped;
    const bool by_group = follow_mask && grouped;
    // the first step of a query's scan is taken by the shading wave that sets the query up (write_query)
    constexpr bool kFirstStepHere = !kCount && !kWide && !kTlas;
    const uint32_t group_shift = S.group_shift;

    Tally tally = {};
    // counted build: when the wave started / first found the pixel queue dry (per wave, in LDS)
    if (kCount && (tid & 63u) == 0) { wave_clock[wave * 2] = __builtin_amdgcn_s_memtime(); wave_clock[wave * 2 + 1] = 0; }
    // A wave that finds nothing to do waits; when NOTHING in the block has moved for S.watchdog_ticks (100 MHz ticks: no
    // ring has received an entry and no slot has retired) it reports a protocol error instead of waiting on.  Waves
    // that are merely out of work while a few costly pixels finish their chains of queries see those queries move.
    auto now = []() -> unsigned long long { return __builtin_amdgcn_s_memrealtime(); };
    auto block_progress = [&]() -> uint32_t {
        uint32_t sum = lds_ld(&ctl.live);
        for (int k = 0; k < 8; k++) sum += lds_ld(&ctl.tail[k]);
        return sum;
    };
    unsigned long long idle_since = 0;
    uint32_t idle_polls = 0, idle_seen = 0;
    auto idle_wait = [&]() -> bool {       // false: leave the kernel
        if (lds_ld(&ctl.live) == 0 || lds_ld(&ctl.abort) != 0) return false;
        if (idle_polls == 0) { idle_since = now(); idle_seen = block_progress(); }
        if ((++idle_polls & 255u) == 0) {
            const uint32_t seen = block_progress();
            if (seen != idle_seen) { idle_seen = seen; idle_since = now(); }
            else if (now() - idle_since > S.watchdog_ticks) {
                lds_st(&ctl.abort, 1u);
                if ((tid & 63u) == 0) atomicAdd(&stats->errors, 1ull);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(8);
        return true;
    };

    // ---- what a lane knows of the query it works on, and what both kinds of wave need to start a query --------
    Lane L = {};
    L.col = kRegroup ? (tid & 63u) : tid;
    uint32_t state = kRegroup ? (uint32_t)S_EMPTY : (uint32_t)S_IDLE;

    // the parked part of the query contexts follows the stack (regrouping kernel)
    unsigned char* const ctx_mem = reinterpret_cast<unsigned char*>(lds + 2u * S.stack_entries * n_ctx);
    // the query has run out of scene nodes
    auto query_done = [&]() -> uint32_t { L.pk = L.shadow ? PK_SHADOW_CLEAR : PK_CLOSEST_DONE; return S_POST; };
    auto shadow_blocked = [&]() -> uint32_t { L.pk = PK_SHADOW_BLOCKED; return S_POST; };
    constexpr bool wide_mask = kWide;             // scenes of 65 to 128 nodes: the second mask word is in use
    auto first_candidate = [&]() -> uint32_t {
        if (L.mask) return (uint32_t)__ffsll((long long)L.mask) - 1u;
        if (wide_mask && L.mask_hi) return 63u + (uint32_t)__ffsll((long long)L.mask_hi);
        return node_count;
    };
    // bits [0, k) of a 64-bit word, for any k (k <= 0: none, k >= 64: all)
    auto low_bits = [](int k) -> unsigned long long { return k <= 0 ? 0ull : (k >= 64 ? ~0ull : (1ull << k) - 1ull); };
    auto after_scan = [&]() -> uint32_t { return L.node < node_count ? (uint32_t)S_NODE : query_done(); };
    // scan by groups: node n if its group is a candidate, else the first node of the next candidate group
    auto next_from = [&](uint32_t n) -> uint32_t {
        if (!by_group || n >= node_count) return n;
        const uint32_t g = n >> group_shift;
        const unsigned long long lo = L.mask & ~low_bits((int)g);
        const unsigned long long hi = wide_mask ? (L.mask_hi & ~low_bits((int)g - 64)) : 0ull;
        const uint32_t first = lo ? (uint32_t)__ffsll((long long)lo) - 1u : (hi ? 63u + (uint32_t)__ffsll((long long)hi) : 0xffffffffu);
        if (first == g) return n;
        return first == 0xffffffffu ? node_count : first << group_shift;
    };

    // The query is through with the object of node `entered` and its ray is still in that node's space.  When the
    // scan has a further node to look at, the next NODE turn takes the ray back to world space first.  When it has
    // none, the query is over: a shadow query's ray is of no further use, and a closest-hit query posts its result
    // with the ray as it is; the shading wave that takes the result does the way back (same operations), with all
    // of its lanes busy instead of the few a traversal turn would have for it.
    auto walk_ended = [&]() -> uint32_t {
        L.pend = kPendExit;
        // (regrouping kernel: worked out when the node was entered, so that the operations inside a node need not carry the mask)
        const uint32_t next = kRegroup ? (L.more ? 0u : node_count) : (by_mask ? first_candidate() : next_from(L.entered + 1u));
        if (next < node_count) return S_NODE;
        if (L.shadow) { L.pend = 0; return query_done(); }
        L.pk = PK_CLOSEST_DONE;
        return S_POST;
    };

    // A query begins: the lane's state from the four words of a RAY record (see write_query), and the operation the
    // query waits for first.
    auto query_start = [&](uint32_t slot, const uint4 a, const uint4 b, const uint4 c, const uint4 d) -> uint32_t {
        L.slot = slot | ((b.w & 2u) ? kLanePaired : 0u);
        L.pdx = f2{__uint_as_float(a.x), __uint_as_float(b.x)};
        L.pdy = f2{__uint_as_float(a.y), __uint_as_float(b.y)};
        L.pdz = f2{__uint_as_float(a.z), __uint_as_float(b.z)};
        L.limit = __uint_as_float(a.w);
        L.shadow = (b.w & 1u) != 0;
        L.qinv = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
        L.node = c.w;
        L.mask = (unsigned long long)d.x | ((unsigned long long)d.y << 32);
        if (kWide) L.mask_hi = (unsigned long long)d.z | ((unsigned long long)d.w << 32);
        L.pend = 0;
        L.hz = CRT_BIGFLOAT; L.hpos = mk3(0, 0, 0); L.h_kind = H_NONE;
        // no node of the scene can be hit at all: the query is over before its first box step
        uint32_t begin = after_scan();
        if (kRegroup) L.more = (by_mask ? first_candidate() : next_from(c.w + 1u)) < node_count;      // (for a query that starts inside node c.w, below)
        // the shading wave has taken the query into the mesh of node c.w already (see write_query): d.z is the root
        if (kFirstStepHere && (b.w & 4u)) {
            L.entered = c.w; L.sp = 0; L.lo = 0; L.top = 0; L.mesh_hit = false;
            L.cur = d.z;
            begin = S_BOX;
        }
        // ... or through a sphere / plane: b.w has the outcome, the hit (if any) is in c and d.z
        if (kFirstStepHere && (b.w & 8u)) {
            L.entered = c.w;
            L.mesh_hit = (b.w & 16u) != 0;
            if (L.mesh_hit) {
                L.hz = __uint_as_float(d.z);
                L.h_kind = (b.w & 64u) ? H_SPHERE : H_PLANE;
                L.h_node = c.w; L.h_front = (b.w & 32u) != 0;
                L.h_a = __uint_as_float(c.x); L.h_b = __uint_as_float(c.y); L.h_c = c.z;
            }
            begin = walk_ended();
        }
        // the walk over the BVH of the nodes starts at its root: an interior record, the only object of the scene, or nothing
        if (kTlas) {
            L.entered = 0; L.sp = 0; L.lo = 0; L.top = 0; L.mesh_hit = false; L.more = false;
            const uint32_t r = S.tlas_root;
            if (!ref_is_leaf(r)) { L.cur = r; begin = S_BOX; }
            else if (leaf_count(r) == kTlasEnter) { L.pend = r; begin = S_NODE; }
            else begin = query_done();
        }
        return begin;
    };

    // ---- query contexts (regrouping kernel) --------------------------------------------------------
    // The state of a query lives in its context in LDS, as seven 16-byte words:
    //   c0 position, limit            c1 direction, node the query is inside      c2 1 / direction, slot
    //   c3 closest hit distance, BVH record, packed small fields, pending leaf range / exit marker
    //   c4 candidate mask, next node to scan, (first word of the second mask word)
    //   c5 hit: barycentrics or local point, triangle record, node and kind
    //   c6 world position of the hit, (second word of the second mask word)
    // A lane's registers hold a query for one turn only: every turn loads what its operation reads and stores what it
    // may have changed, so a lane carries nothing from turn to turn but the context's number.  All lanes load (a lane
    // without a context reads some context's words and ignores them): no branch, and the registers are defined at one
    // place for every lane.
    static_assert(kContextBytes == 112, "the context address below is col * 128 - col * 16");
    auto ctx_words = [&]() -> uint4* { return reinterpret_cast<uint4*>(ctx_mem + ((L.col << 7) - (L.col << 4))); };
    auto put_small = [&]() -> uint32_t {
        return L.sp | (L.lo << 8) | (L.top << 16) | (L.pk << 24) | (L.shadow ? 1u << 27 : 0u) | (L.more ? 1u << 28 : 0u) | (L.mesh_hit ? 1u << 29 : 0u) | (L.h_front ? 1u << 30 : 0u);
    };
    auto get_c0 = [&](const uint4 c) { L.pdx.x = __uint_as_float(c.x); L.pdy.x = __uint_as_float(c.y); L.pdz.x = __uint_as_float(c.z); L.limit = __uint_as_float(c.w); };
    auto get_c1 = [&](const uint4 c) { L.pdx.y = __uint_as_float(c.x); L.pdy.y = __uint_as_float(c.y); L.pdz.y = __uint_as_float(c.z); L.entered = c.w; };
    auto get_c2 = [&](const uint4 c) { L.qinv = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)); L.slot = (c.w & (kQSize - 1u)) | ((c.w & kQSize) ? kLanePaired : 0u); };
    auto get_c3 = [&](const uint4 c) {
        L.hz = __uint_as_float(c.x); L.cur = c.y; L.pend = c.w;
        const uint32_t w = c.z;
        L.sp = w & 255u; L.lo = (w >> 8) & 255u; L.top = (w >> 16) & 255u; L.pk = (w >> 24) & 7u;
        L.shadow = (w & (1u << 27)) != 0; L.more = (w & (1u << 28)) != 0; L.mesh_hit = (w & (1u << 29)) != 0; L.h_front = (w & (1u << 30)) != 0;
    };
    auto get_c4 = [&](const uint4 c) { L.mask = (unsigned long long)c.x | ((unsigned long long)c.y << 32); L.node = c.z; if (kWide) L.mask_hi = (L.mask_hi & 0xffffffff00000000ull) | c.w; };
    auto get_c5 = [&](const uint4 c) { L.h_a = __uint_as_float(c.x); L.h_b = __uint_as_float(c.y); L.h_c = c.z; L.h_node = c.w & 0x0fffffffu; L.h_kind = c.w >> 28; };
    auto get_c6 = [&](const uint4 c) { L.hpos = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)); if (kWide) L.mask_hi = (L.mask_hi & 0xffffffffull) | ((unsigned long long)c.w << 32); };
    auto put_c0 = [&]() -> uint4 { return uint4{__float_as_uint(L.pdx.x), __float_as_uint(L.pdy.x), __float_as_uint(L.pdz.x), __float_as_uint(L.limit)}; };
    auto put_c1 = [&]() -> uint4 { return uint4{__float_as_uint(L.pdx.y), __float_as_uint(L.pdy.y), __float_as_uint(L.pdz.y), L.entered}; };
    auto put_c2 = [&]() -> uint4 { return uint4{__float_as_uint(L.qinv.x), __float_as_uint(L.qinv.y), __float_as_uint(L.qinv.z), (L.slot & (kQSize - 1u)) | ((L.slot & kLanePaired) ? kQSize : 0u)}; };
    auto put_c3 = [&]() -> uint4 { return uint4{__float_as_uint(L.hz), L.cur, put_small(), L.pend}; };
    auto put_c4 = [&]() -> uint4 { return uint4{(uint32_t)L.mask, (uint32_t)(L.mask >> 32), L.node, kWide ? (uint32_t)L.mask_hi : 0u}; };
    auto put_c5 = [&]() -> uint4 { return uint4{__float_as_uint(L.h_a), __float_as_uint(L.h_b), L.h_c, (L.h_node & 0x0fffffffu) | (L.h_kind << 28)}; };
    auto put_c6 = [&]() -> uint4 { return uint4{__float_as_uint(L.hpos.x), __float_as_uint(L.hpos.y), __float_as_uint(L.hpos.z), kWide ? (uint32_t)(L.mask_hi >> 32) : 0u}; };

    // The selected lanes give their contexts up: the context's bit goes into the pool of the operation it waits for
    // (its state is in LDS already).
    const uint32_t lane = tid & 63u;
    bool spilled = false;      // the lane has moved a stack entry to device memory since the last park
    auto park = [&](bool sel) {
        if (kCount) { tally.turns[8]++; tally.lanes[8] += sel; }
        // The context's words were written to LDS by this wave before the bit is set, and the LDS serves a wave's
        // operations in order: the wave that takes the bit reads those words.  Only device memory (a stack entry moved out
        // of the LDS ring) needs the wait of a release fence; otherwise the fence just keeps the compiler from reordering.
        if (__ballot(spilled) != 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); spilled = false; }
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (sel) {
            __hip_atomic_fetch_or(&pool[state][L.col & 63u], 1u << (L.col >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            state = S_EMPTY;
        }
    };
    // What lane position p may take: contexts of position p and of position p ^ 32.  The two positions map to the same
    // LDS banks and lie in different halves of the wave, which the LDS serves in different cycles: a lane that works
    // on a context of its partner position conflicts with nobody.
    auto parked = [&](uint32_t q, uint32_t& own, uint32_t& partner) {
        const uint32_t* pair = &pool[q][lane & 31u];
        const uint32_t lo = lds_ld(pair), hi = lds_ld(pair + 32);
        own = lane < 32u ? lo : hi; partner = lane < 32u ? hi : lo;
    };
    // Lanes with `want` (empty ones) take a context that waits for operation q, the first `most` of the lanes that
    // have one to take.  One attempt: a lane that loses its pick to another wave stays empty for this turn.
    auto claim_from = [&](uint32_t q, uint32_t own, uint32_t partner, bool want, uint32_t most, uint32_t& col) -> bool {
        const uint32_t pos = own ? lane : lane ^ 32u;
        uint32_t m = want ? (own ? own : partner) : 0u;
        if (most < 64u && lane_rank(__ballot(m != 0)) >= most) m = 0;
        bool got = false;
        uint32_t j = 0;
        if (m != 0) {
            j = (uint32_t)__ffs((int)m) - 1u;
            const uint32_t old = __hip_atomic_fetch_and(&pool[q][pos], ~(1u << j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            got = (old >> j) & 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (kCount) { tally.turns[9]++; tally.lanes[9] += got; }
        col = (j << 6) | pos;
        return got;
    };
    auto claim = [&](uint32_t q, bool want, uint32_t most, uint32_t& col) -> bool {
        uint32_t own, partner;
        parked(q, own, partner);
        return claim_from(q, own, partner, want, most, col);
    };
    auto unpark = [&](uint32_t q, bool want, uint32_t most) {
        uint32_t col = 0;
        if (claim(q, want, most, col)) { L.col = col; state = q; }
    };

    // ---- traversal: stack, operations, exchange (used by the waves whose turn is a traversal turn) ----------------
    // =====================================================================================================
    // traversal waves
    // =====================================================================================================
    // The traversal stack lives in LDS, one column per traversal thread (per query context in the regrouping
    // kernel), laid out [entry][column]; the plain layout keeps node references and distances in two arrays, the
    // regrouping one keeps an entry's two words side by side (one 8-byte access).
    uint32_t* __restrict__ stack_ref = lds;
    float* __restrict__ stack_t = reinterpret_cast<float*>(lds + S.stack_entries * trav_threads);
    uint2* __restrict__ stack_rt = reinterpret_cast<uint2*>(lds);
    const uint32_t lds_entries = S.stack_entries;
    // overflow of the traversal stack (deep trees): behind the slot records, one column per stack column
    // (laid out [block][entry][column], reference and distance of an entry side by side: one 8-byte access per lane)
    uint2* const block_overflow = reinterpret_cast<uint2*>(slot_mem + (size_t)gridDim.x * n_slots * kSlotWords) +
                                  (size_t)blockIdx.x * S.stack_overflow * n_ctx;

    // Push / pop.  (slot * columns as a 24-bit product: the full 32-bit multiply issues at a quarter of the rate.)
    // kDeep and kRegroup: LDS holds the TOP of the stack, as a ring of lds_entries positions: entries [L.lo, L.sp)
    // live at position index % lds_entries (L.top is that position for L.sp), and a push into a full ring first
    // moves the oldest entry, the one the new one is about to overwrite, to its place in device memory.  Pops meet
    // memory only when a walk has come back below everything the ring holds.
    constexpr bool kRingStack = kDeep || kRegroup;
    auto stack_push = [&](uint32_t ref, float t, bool keep) {
        if (!kRingStack) {
            // written even when it is not kept (the slot above the top, left unused): no branch around the write
            stack_ref[__umul24(L.sp, trav_threads) + tid] = ref; stack_t[__umul24(L.sp, trav_threads) + tid] = t;
            L.sp += keep ? 1u : 0u;
        } else if (keep) {
            const uint32_t at = __umul24(L.top, n_ctx) + L.col;
            if (L.sp - L.lo == lds_entries) {
                block_overflow[__umul24(L.lo, n_ctx) + L.col] = kRegroup ? stack_rt[at] : uint2{stack_ref[at], __float_as_uint(stack_t[at])};
                L.lo++;
                spilled = true;
            }
            if (kRegroup) stack_rt[at] = uint2{ref, __float_as_uint(t)};
            else { stack_ref[at] = ref; stack_t[at] = t; }
            L.sp++;
            L.top = L.top + 1u == lds_entries ? 0u : L.top + 1u;
        }
    };
    auto stack_pop = [&](uint32_t& ref, float& t) {
        L.sp--;
        if (!kRingStack) { ref = stack_ref[__umul24(L.sp, trav_threads) + tid]; t = stack_t[__umul24(L.sp, trav_threads) + tid]; return; }
        L.top = L.top == 0u ? lds_entries - 1u : L.top - 1u;
        if (L.sp >= L.lo) {
            const uint32_t at = __umul24(L.top, n_ctx) + L.col;
            if (kRegroup) { const uint2 e = stack_rt[at]; ref = e.x; t = __uint_as_float(e.y); }
            else { ref = stack_ref[at]; t = stack_t[at]; }
        } else { const uint2 e = block_overflow[__umul24(L.sp, n_ctx) + L.col]; ref = e.x; t = __uint_as_float(e.y); L.lo = L.sp; }
    };

    // pop until an interior node that survives the distance cull, a non-empty leaf, or an empty stack
    // (MeshObject::Traverse / TraverseShadow loop heads, scene/objects.cu:184-196, 231-242)
    auto resolve = [&](bool have_next, uint32_t next_ref, float next_t) -> uint32_t {
        for (;;) {
            if (!have_next) {
                if (L.sp == 0) return kTlas ? query_done() : walk_ended();
                stack_pop(next_ref, next_t);
            }
            have_next = false;
            if (!L.shadow && L.hz < next_t) continue;
            if (!ref_is_leaf(next_ref)) { L.cur = next_ref; return S_BOX; }
            TALLY(bvh_leaves);
            if (leaf_count(next_ref) == 0) continue;
            // (BVH of the nodes: an object to enter, or the end of a mesh walk - the ray goes back to world space)
            if (kTlas && leaf_count(next_ref) >= kTlasLeave) {
                L.pend = leaf_count(next_ref) == kTlasLeave ? kPendExit : next_ref;
                return S_NODE;
            }
            L.pend = next_ref;
            return S_LEAF;
        }
    };

    // ---- BOX: one BVH interior node, i.e. the slab tests of its two children (objects.cu:198-218, 244-264)
    auto op_box = [&]() -> uint32_t {
        const float4* rp = reinterpret_cast<const float4*>(pairs + L.cur);
        const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
        const uint32_t ref_a = __float_as_uint(q3.x), ref_b = __float_as_uint(q3.y);
        float tA = CRT_BIGFLOAT, tB = CRT_BIGFLOAT;
        float dA, dB, eA, eB;
        const bool hitA = slab(f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, L.qpos(), L.qinv, L.limit, dA, eA);
        const bool hitB = slab(f2{q1.z, q1.w}, f2{q2.x, q2.y}, f2{q2.z, q2.w}, L.qpos(), L.qinv, L.limit, dB, eB);
        // This mode goes by the distance at which a ray ENTERS a box, for the boxes over the nodes and for those inside a
        // mesh alike.  (The reference's box test reports the way OUT for a ray that starts inside a box; its node scan skips
        // a node, and its BVH walk a subtree, when that distance lies beyond the closest hit so far - trace.cu:57-59,
        // objects.cu:190-192 - so what it finds depends on the order of the nodes, which this mode does not keep.)
        if (kTlas) { dA = eA; dB = eB; }
        if (hitA) tA = dA;
        if (hitB) tB = dB;
        TALLY(bvh_interior);
        // both children go on the stack, far one first (objects.cu:216-218); when neither box is hit the
        // slot above the top is written and left unused, so one copy of the pop loop serves both cases
        const bool any = hitA || hitB;
        const bool a_near = tA <= tB;
        // The reference puts the far child on the stack even when its own box was missed (objects.cu:216-218); popped
        // later, such an INTERIOR node is expanded to no effect: the boxes of its children lie inside its own box
        // (unions of vertex bounds, renderer/bvh.cu:82-98, 160-166) and every step of the slab test is monotone in
        // the box planes, so both are missed and nothing follows.  Where the stack is the scarce resource (deep
        // trees) the plain kernels leave such a node out; the counting build keeps it, to count what the reference
        // counts.  A missed LEAF stays in either case: its triangles are tested, as the reference tests them.
        // A closest-hit query pops nothing that lies beyond its hit (objects.cu:190-192), and the hit only comes
        // closer: a far child that is beyond it already need not go on the stack at all.
        const float far_t = fmaxf(tA, tB);
        bool keep = any && (L.shadow || !(L.hz < far_t));
        if (!kCount || follow_mask) keep = keep && ((hitA && hitB) || ref_is_leaf(a_near ? ref_b : ref_a));      // (counting build: only when profiling the plain kernels' walk)
        if (kTlas) keep = keep && hitA && hitB;      // (a leaf whose box is missed is left out too: an object, or triangles that cannot be hit)
        stack_push(a_near ? ref_b : ref_a, far_t, keep);
        return resolve(any, a_near ? ref_a : ref_b, fminf(tA, tB));
    };

    // ---- LEAF: the next two triangles of the pending leaf, one after the other (IntersectTriangle /
    // IntersectShadowTriangle, objects.cu:270-366); both records are fetched up front so that the
    // pair costs one memory round trip
    bool leaf_hit = false;      // set by tri_test: the lane's closest hit has changed
    auto tri_test = [&](const float4 v0, const float4 v1, const float4 v2, uint32_t slot) -> bool {
        TALLY(tri_tests);
        // Same operations and order as the scalar form; independent products and sums of the two cross
        // products and of the dot products are paired into packed instructions.
        const f2 ex = f2{v1.x, v2.x} - v0.x, ey = f2{v1.y, v2.y} - v0.y, ez = f2{v1.z, v2.z} - v0.z;   // (edge1, edge2)
        f2 odx = L.pdx, ody = L.pdy, odz = L.pdz;                                                        // (offset, dir)
        odx.x -= v0.x; ody.x -= v0.y; odz.x -= v0.z;
        // (oXe1, rXe2) = (cross(offset, edge1), cross(dir, edge2)), component-wise as in cross()
        const f2 cx = ody * ez - ey * odz;
        const f2 cy = -(odx * ez - ex * odz);
        const f2 cz = odx * ey - ex * ody;
        // (det, t numerator) = (dot(edge1, rXe2), dot(edge2, oXe1)); (b1, b2 numerators) = (dot(offset, rXe2), dot(dir, oXe1)):
        // each pair multiplies by the cross products with their halves swapped
        const f2 u = ex * cx.yx + ey * cy.yx + ez * cz.yx;
        const f2 w = odx * cx.yx + ody * cy.yx + odz * cz.yx;
        const f3 edge1 = mk3(ex.x, ey.x, ez.x), edge2 = mk3(ex.y, ey.y, ez.y), dir = mk3(odx.y, ody.y, odz.y);
        const float det = u.x;
        bool ok = !(det > -CRT_TRI_EPS && det < CRT_TRI_EPS);
        const float idet = rcp_ieee(det);
        const float b1 = idet * w.x;
        ok = ok && !(b1 <= CRT_TRI_EPS || b1 >= 1 + CRT_TRI_EPS);
        const float b2 = idet * w.y;
        ok = ok && !(b2 <= -CRT_TRI_EPS || (b1 + b2) >= 1 + CRT_TRI_EPS);
        const float t = idet * u.y;
        const float far_limit = (L.shadow ? L.limit : L.hz) + CRT_TRI_EPS;
        ok = ok && !(t <= CRT_TRI_EPS || t >= far_limit);
        const float ndot = dot(cross(edge1, edge2), dir);
        ok = ok && ((ndot < CRT_TRI_EPS) || (ndot > -CRT_TRI_EPS));
        if (ok && !L.shadow) {
            leaf_hit = true;
            TALLY(tri_hits);
            L.hz = t;
            L.mesh_hit = true;
            L.h_kind = H_MESH; L.h_node = L.entered; L.h_a = b1; L.h_b = b2; L.h_c = slot; L.h_front = ndot < 0;
        }
        return ok;
    };
    auto op_leaf = [&]() -> uint32_t {
        const uint32_t slot = leaf_start(L.pend), count = leaf_count(L.pend);
        const bool two = count >= 2;
        const uint32_t second = two ? slot + 1 : slot;      // a single triangle is fetched twice rather than reading past the leaf
        const uint32_t left = count - (two ? 2u : 1u);
        L.pend = make_leaf_ref(slot + (two ? 2u : 1u), left);
        const float4* tp = reinterpret_cast<const float4*>(tris + slot);
        const float4* tq = reinterpret_cast<const float4*>(tris + second);
        const float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
        const float4 b0 = tq[0], b1 = tq[1], b2 = tq[2];
        if (tri_test(a0, a1, a2, slot) && L.shadow) return shadow_blocked();
        if (two && tri_test(b0, b1, b2, second) && L.shadow) return shadow_blocked();
        if (left) return S_LEAF;
        return resolve(false, 0, 0);
    };

    // ---- NODE: one step of the scene-node scan (trace.cu:51-68, 166-193).  A lane whose BVH walk has just ended
    // first leaves that mesh (world position of its hit, Node::FromLocal for the ray: trace.cu:64-66, 191); then the
    // box of the next scene node is tested, and if the query has to enter the node it does so in the same turn:
    // Node::ToLocal, then the sphere / plane test in place or the start of a BVH walk (trace.cu:61-66, 188-191;
    // objects.cu:7-159).
    auto op_node = [&]() -> uint32_t {
        if (kTlas) {
            // Walk over the BVH of the nodes: the turn either takes the query into the object of one node (Node::ToLocal,
            // then the sphere / plane test in place or the start of the mesh's BVH walk under a marker on the stack) or,
            // when that marker comes off the stack, back out of the mesh (Node::FromLocal, world position of a hit).
            const bool leaving = L.pend == kPendExit;
            const uint32_t idx = leaving ? L.entered : (leaf_start(L.pend) < node_count ? leaf_start(L.pend) : 0u);
            const float4* __restrict__ xp = reinterpret_cast<const float4*>(S.xf + idx);
            L.pend = 0;
            // (one copy of the pop loop behind all three cases: what differs is whether there is a node to go on with)
            bool have_next = false;
            uint32_t next_ref = 0;
            if (leaving) {
                const float4 r3 = xp[3], r4 = xp[4], r5 = xp[5];
                const float tm[12] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w};
                if (L.mesh_hit) L.hpos = xform_point(tm, L.qpos() + L.hz * L.qdir());
                L.mesh_hit = false;
                xform_ray(tm, L.pdx, L.pdy, L.pdz);
                L.qinv = recip3(L.qdir());
            } else {
                const float4 r0 = xp[0], r1 = xp[1], r2 = xp[2], r6 = xp[6];
                const float itm[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                const uint32_t kind = (__float_as_uint(r6.w) >> kXfKindShift) & 3u;
                TALLY(node_enters);
                f2 px = L.pdx, py = L.pdy, pz = L.pdz;
                xform_ray(itm, px, py, pz);
                if (kind == CRT_OBJ_MESH) {
                    L.pdx = px; L.pdy = py; L.pdz = pz;
                    L.qinv = recip3(L.qdir());
                    L.entered = idx; L.mesh_hit = false;
                    stack_push(make_leaf_ref(0, kTlasLeave), -CRT_BIGFLOAT, true);
                    have_next = true; next_ref = __float_as_uint(r6.x);
                } else {
                    // sphere / plane: tested with a local copy of the ray, the query's own ray stays in world space
                    const f3 pos = mk3(px.x, py.x, pz.x), dir = mk3(px.y, py.y, pz.y);
                    if (kind == CRT_OBJ_SPHERE) TALLY(sphere_tests); else TALLY(plane_tests);
                    bool front;
                    float t;
                    const bool got = (kind == CRT_OBJ_SPHERE || kind == CRT_OBJ_PLANE) && unit_object_hit(kind == CRT_OBJ_SPHERE, pos, dir, L.shadow, L.limit, L.hz, t, front);
                    if (got) {
                        if (L.shadow) return shadow_blocked();
                        const f3 p = pos + t * dir;
                        // (the forward matrix is fetched only now: few of the objects a walk looks at are hit)
                        const float4 r3 = xp[3], r4 = xp[4], r5 = xp[5];
                        const float tm[12] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w};
                        L.hz = t;
                        L.hpos = xform_point(tm, p);
                        L.h_kind = kind == CRT_OBJ_SPHERE ? H_SPHERE : H_PLANE;
                        L.h_node = idx; L.h_front = front;
                        L.h_a = p.x; L.h_b = p.y; L.h_c = __float_as_uint(p.z);
                    }
                }
            }
            return resolve(have_next, next_ref, 0.0f);
        }
        // Which node is looked at does not depend on the way back, so the forward matrix of the node being left and
        // the box record of the next node are fetched together (lanes that leave nothing read the record of node 0,
        // one address for all of them).
        const bool leaving = L.pend == kPendExit;
        if (leaving) L.node = by_mask ? first_candidate() : next_from(L.entered + 1u);      // (a further node exists: walk_ended)
        const float4* __restrict__ lp = reinterpret_cast<const float4*>(S.xf + (leaving ? L.entered : 0u));
        const float4* rp = reinterpret_cast<const float4*>(pairs + L.node);
        const float4 r3 = lp[3], r4 = lp[4], r5 = lp[5];
        const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
        if (leaving) {
            const float tm[12] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w};
            if (L.mesh_hit) L.hpos = xform_point(tm, L.qpos() + L.hz * L.qdir());
            L.pend = 0;
            xform_ray(tm, L.pdx, L.pdy, L.pdz);
            L.qinv = recip3(L.qdir());
        }
        // node `node` is half A of its record, node + 1 is half B.  Closest hit enters a node whose box is
        // hit nearer than the current hit; a shadow ray enters on any box hit and skips the whole
        // subtree of a node with an object whose box it misses (trace.cu:171-185).
        // (The half after the last node of an odd list is a dummy without an object and with an empty box.)
        const uint32_t ref_a = __float_as_uint(q3.x), ref_b = __float_as_uint(q3.y);
        float tA = CRT_BIGFLOAT, tB = CRT_BIGFLOAT;
        float dA, dB;
        const bool hitA = slab(f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, L.qpos(), L.qinv, L.limit, dA);
        const bool hitB = slab(f2{q1.z, q1.w}, f2{q2.x, q2.y}, f2{q2.z, q2.w}, L.qpos(), L.qinv, L.limit, dB);
        if (hitA) tA = dA;
        if (hitB) tB = dB;
        const uint32_t kind_a = node_ref_kind(ref_a), kind_b = node_ref_kind(ref_b);
        const uint32_t after_a = node_ref_after(ref_a), after_b = node_ref_after(ref_b);
        const bool sh = L.shadow;
        bool enter_a, enter_b;
        if (by_mask) {
            // Scan by candidates: half A is the lowest candidate, half B is looked at when it is one too.  Nodes in
            // between were certain misses; a candidate object node that a shadow ray misses takes its subtree along.
            const uint32_t i = L.node;
            enter_a = hitA & (sh | (tA < L.hz));
            if (!wide_mask) {
                unsigned long long m = L.mask & ~(1ull << i);
                if (sh & !hitA) m &= ~(((1ull << (after_a & 63u)) - 1ull) | (after_a >= 64u ? ~0ull : 0ull)) | ((2ull << i) - 1ull);
                const bool cand_b = !enter_a & (i < 63u) & (((m >> ((i + 1u) & 63u)) & 1ull) != 0);
                enter_b = cand_b & hitB & (sh | (tB < L.hz));
                if (cand_b) {
                    m &= ~(2ull << i);
                    if (sh & !hitB) m &= ~(((1ull << (after_b & 63u)) - 1ull) | (after_b >= 64u ? ~0ull : 0ull)) | ((4ull << i) - 1ull);
                }
                L.mask = m;
            } else {
                // the same with two words: `drop(from, to)` clears the nodes after `from` and before `to`
                unsigned long long m0 = L.mask, m1 = L.mask_hi;
                auto clear_bit = [&](uint32_t b) { if (b < 64u) m0 &= ~(1ull << b); else m1 &= ~(1ull << (b - 64u)); };
                auto drop = [&](uint32_t from, uint32_t to) {
                    m0 &= low_bits((int)from + 1) | ~low_bits((int)to);
                    m1 &= low_bits((int)from + 1 - 64) | ~low_bits((int)to - 64);
                };
                clear_bit(i);
                if (sh & !hitA) drop(i, after_a);
                const uint32_t j = i + 1u;
                const bool cand_b = !enter_a & (j < 128u) & ((((j < 64u ? m0 : m1) >> (j & 63u)) & 1ull) != 0);
                enter_b = cand_b & hitB & (sh | (tB < L.hz));
                if (cand_b) {
                    clear_bit(j);
                    if (sh & !hitB) drop(j, after_b);
                }
                L.mask = m0; L.mask_hi = m1;
            }
            if (!(enter_a | enter_b)) {
                L.node = first_candidate();
                return after_scan();
            }
        } else {
            const bool obj_a = kind_a != CRT_OBJ_NONE, obj_b = kind_b != CRT_OBJ_NONE;
            enter_a = obj_a & hitA & (sh | (tA < L.hz));
            const bool jump_a = sh & obj_a & (after_a != L.node + 1);            // shadow ray past a missed subtree
            const bool look_b = !(enter_a | jump_a);
            enter_b = look_b & obj_b & hitB & (sh | (tB < L.hz));
            if (kCount) {
                tally.node_box_tests += (obj_a ? 1u : 0u) + ((look_b & obj_b) ? 1u : 0u);
                // every box the full scan finds hit must be a candidate of the mask the plain kernels scan by
                if (S.scan_mask) {
                    const bool chk_a = obj_a & hitA, chk_b = look_b & obj_b & hitB;
                    tally.scan_checks += (chk_a ? 1u : 0u) + (chk_b ? 1u : 0u);
                    auto is_candidate = [&](uint32_t node) -> bool { const uint32_t b = grouped ? node >> group_shift : node; return (((b < 64u ? L.mask : L.mask_hi) >> (b & 63u)) & 1ull) != 0; };
                    tally.scan_violations += ((chk_a && !is_candidate(L.node)) ? 1u : 0u) + ((chk_b && !is_candidate(L.node + 1u)) ? 1u : 0u);
                }
            }
            if (!(enter_a | enter_b)) {
                L.node = next_from(jump_a ? after_a : ((sh & obj_b) ? after_b : L.node + 2));
                return after_scan();
            }
        }
        const uint32_t idx = enter_a ? L.node : L.node + 1u;
        if (kRegroup) L.more = (by_mask ? first_candidate() : next_from(idx + 1u)) < node_count;
        const uint32_t kind = enter_a ? kind_a : kind_b;
        // the inverse matrix and the tags of the node's record; the forward matrix is fetched when the node is left
        const float4* __restrict__ xp = reinterpret_cast<const float4*>(S.xf + idx);
        const float4 r0 = xp[0], r1 = xp[1], r2 = xp[2], r6 = xp[6];
        const float itm[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        TALLY(node_enters);
        xform_ray(itm, L.pdx, L.pdy, L.pdz);
        if (kind == CRT_OBJ_MESH) {
            L.qinv = recip3(L.qdir());
            L.entered = idx; L.sp = 0; L.lo = 0; L.top = 0; L.mesh_hit = false;
            return resolve(true, __float_as_uint(r6.x), 0.0f);
        }
        const f3 pos = L.qpos(), dir = L.qdir();
        if (kind == CRT_OBJ_SPHERE) TALLY(sphere_tests); else TALLY(plane_tests);
        bool front;
        float t;
        const bool got = unit_object_hit(kind == CRT_OBJ_SPHERE, pos, dir, L.shadow, L.limit, L.hz, t, front);
        L.entered = idx;
        L.mesh_hit = false;
        if (got) {
            if (L.shadow) return shadow_blocked();
            const f3 p = pos + t * dir;
            L.hz = t;
            L.mesh_hit = true;            // the world position follows when the node is left, as after a BVH walk
            L.h_kind = kind == CRT_OBJ_SPHERE ? H_SPHERE : H_PLANE;
            L.h_node = idx; L.h_front = front;
            L.h_a = p.x; L.h_b = p.y; L.h_c = __float_as_uint(p.z);
        }
        return walk_ended();
    };

    // ---- EXCHANGE: lanes whose query has ended write its result into their slot and queue the slot for
    // shading; they and the lanes that were idle already then take the next ready queries, whichever pixels
    // those belong to
    bool took = false;      // op_exchange: the lane has taken a new query
    auto op_exchange = [&]() {
        const bool post = state == S_POST;
        const uint32_t my_slot = L.slot & (kQSize - 1u);
        const bool paired = (L.slot & kLanePaired) != 0;
        if (post) {
            uint32_t* rec = slot_rec(my_slot) + kRecHit;
            if (L.pk == PK_CLOSEST_DONE) {
                uint4* r4 = reinterpret_cast<uint4*>(rec);
                const f3 qp = L.qpos(), qd = L.qdir();
                // last word: 0, or the node whose space the ray (and a hit found in it) is still in: node + 1, bit 31 = hit found there
                const uint32_t inside = !kTlas && L.pend == kPendExit ? ((L.entered + 1u) | (L.mesh_hit ? 0x80000000u : 0u)) : 0u;
                r4[0] = uint4{L.pk | (paired ? kPkPaired : 0u), __float_as_uint(L.hz), L.h_kind | (L.h_front ? 4u : 0u) | (L.h_node << 3), L.h_c};
                r4[1] = uint4{__float_as_uint(L.hpos.x), __float_as_uint(L.hpos.y), __float_as_uint(L.hpos.z), __float_as_uint(L.h_a)};
                r4[2] = uint4{__float_as_uint(qp.x), __float_as_uint(qp.y), __float_as_uint(qp.z), __float_as_uint(L.h_b)};
                r4[3] = uint4{__float_as_uint(qd.x), __float_as_uint(qd.y), __float_as_uint(qd.z), inside};
            } else if (paired) {
                slot_rec(my_slot)[kRecShadowEnd] = L.pk;
            } else {
                rec[0] = L.pk;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // of a pair, the query that ends second hands the slot on (both results are in the record by then)
        bool hand_on = post;
        if (post && paired) {
            const uint32_t bit = 1u << (my_slot & 31u);
            hand_on = (__hip_atomic_fetch_xor(&join_bits[my_slot >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) != 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        queue_push<kQLog>(&ctl.tail[kQShade0], shade_ring[0], hand_on && (paired || L.pk == PK_CLOSEST_DONE), my_slot);
        queue_push<kQLog>(&ctl.tail[kQShade1], shade_ring[1], hand_on && !paired && L.pk != PK_CLOSEST_DONE, my_slot);
        if (post) state = S_IDLE;
        uint32_t slot = 0;
        const bool got = queue_pop<kRLog>(&ctl.head[kQRay], &ctl.tail[kQRay], ray_ring, state == S_IDLE, 1u, kRSize, slot, &ctl.abort);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (kCount) { const uint32_t n_got = (uint32_t)__popcll(__ballot(got)); if ((tid & 63u) == 0) tally.cycles[5] += n_got; }
        took = got;
        if (got) {
            const uint32_t second = slot & kSecond;
            slot &= kQSize - 1u;
            if (slot >= n_slots) slot = n_slots - 1u;      // only after an abort: stay inside the block's records
            const uint4* r4 = reinterpret_cast<const uint4*>(slot_rec(slot) + (second ? kRecRay2 : kRecRay));
            const uint4 a = r4[0], b = r4[1], c = r4[2], d = r4[3];
            state = query_start(slot, a, b, c, d);
        }
    };

    // ---- shading: everything between two ray queries of a pixel (TracePath, trace.cu:83-158), the camera ray of a
    // new path (trace.cu:14-38) and the next pixel of a finished slot -----------------------------------------------
    const float F_PI = 3.14159265358979323846f;
    const float OVERPI = (float)(1.0f / 3.14159265358979323846);
    const float inv_samples = 1.0f / (float)samples;
    bool queue_empty = false;
    uint32_t queue_seen = 0;      // where the pixel queue stood when this wave last took pixels from it
    bool fault_done = false, lose_first = false;
    bool direct_first = false, direct_second = false;      // the turn's queries went straight into query contexts (write_query)
    // The traversal lanes of the whole block wait on what these few waves produce: they go first whenever they
    // have work (they sleep when they have none).

    // Section clock of the counted build: one stamp per wave in LDS, advanced by the first active
    // lane, so that a section's cycles are counted once per turn whichever lanes run it.
#define SECT(i) do { if (kCount && lane_rank(__ballot(true)) == 0) { \
        const unsigned long long t_n = __builtin_amdgcn_s_memtime(); \
        path_sections[wave * 8 + i] += t_n - sect_clock[wave]; sect_clock[wave] = t_n; } } while (0)

    // one shading turn for the lane's slot; returns what becomes of the slot
    // (kind: std::true_type for slots of the closest-hit queue, std::false_type for the other queue)
    auto serve = [&](auto kind, bool have, uint32_t slot) -> uint32_t {
        constexpr bool kClosest = decltype(kind)::value;
        uint32_t result = R_NONE;
        uint32_t* const rec = slot_rec(have ? slot : 0u);
        // Two queries of a pixel at once shorten its chain of queries, at the price of shading turns that are less
        // uniform (the second query's set-up runs only for the lanes that have one).  While the block has pixels
        // enough to keep its traversal lanes busy the kernel is bound by throughput and the trade does not pay;
        // once slots have retired (no pixels left to take: the end of a frame, or a small share of one) and fewer
        // than pair_live of them are left, the frame lasts as long as the longest chain, and it does.
        // A frame shared among several GPUs (P.parts > 1) is short of pixels on every one of them sooner or later and
        // pairs from the start: measured, that is the better trade from two GPUs on.  (pair_below: pair also while
        // few queries are ready.)
        // The first step of a query's node scan (write_query) is taken here only while the traversal lanes have queries in
        // reserve: when they run dry the shading waves are what the block waits for, and they keep their turns short.
        // (Regrouping kernel with queries started straight in contexts: the sign that the traversal side has work in
        // reserve is that few of the contexts within this lane's reach are free.)
        bool first_now = kFirstStepHere && lds_ld(&ctl.tail[kQRay]) - lds_ld(&ctl.head[kQRay]) >= S.first_step;
        if (kFirstStepHere && kRegroup && S.direct_start) {
            uint32_t own, partner;
            parked(kQIdle, own, partner);
            first_now = first_now || (uint32_t)(__popc(own) + __popc(partner)) < S.first_step_idle;
        }
        const bool pair_now = kClosest && (P.parts > 1u || lds_ld(&ctl.live) < S.pair_live || lds_ld(&ctl.tail[kQRay]) - lds_ld(&ctl.head[kQRay]) < S.pair_below);
        uint32_t pk = have ? rec[kRecHit] : (uint32_t)PK_NEW_PATH;
        const bool was_paired = kClosest && (pk & kPkPaired) != 0;
        pk &= ~kPkPaired;
        Rng rng = {};
        uint32_t pI = 0, tile = 0, cost = 0, segments = 0;
        int sample = 0;
        bool fresh = false;                                 // pixel taken this turn: nothing to load
        f3 contrib = mk3(0, 0, 0), v = mk3(0, 0, 0);

        // ---- a slot without a pixel takes the next one of the global queue: consecutive items, one atomic per wave
        {
            const bool need = !kClosest && have && pk == PK_FETCH;
            const unsigned long long mask = __ballot(need);
            if (mask != 0) {
                if (kCount) { tally.turns[5]++; tally.lanes[5] += need; }
                // Towards the end of the queue a block stops refilling all of its slots: when the last pixels are handed
                // out, every block still has to finish what its slots hold (whole pixels, each a serial chain of
                // samples), and the less that is, the closer together the blocks finish.  Slots beyond S.tail_live
                // retire instead of taking a pixel once fewer than S.tail_window pixels are left; the pixels they leave
                // go to whichever block asks next.  (Which slot renders a pixel changes nothing about the pixel.)
                unsigned long long fetchers = mask;
                if (!queue_empty && P.total_items - queue_seen < P.tail_window) {
                    const uint32_t live_now = lds_ld(&ctl.live);
                    const uint32_t spare = live_now > S.tail_live ? live_now - S.tail_live : 0u;
                    if (need && lane_rank(mask) < spare) result = R_RETIRE;
                    fetchers = __ballot(need && result == R_NONE);
                }
                if (queue_empty) { if (need) result = R_RETIRE; }
                else if (fetchers != 0) {
                    const unsigned long long mask = fetchers;
                    const bool need = (fetchers >> lane) & 1ull;
                    uint32_t base = 0;
                    const uint32_t cnt = (uint32_t)__popcll(mask);
                    if (need && lane_rank(mask) == 0) base = atomicAdd(queue, cnt);
                    base = __shfl(base, __ffsll((long long)mask) - 1, 64);
                    queue_seen = base + cnt < P.total_items ? base + cnt : P.total_items;
                    if (base + cnt >= P.total_items) {      // uniform: the queue has run dry
                        queue_empty = true;
                        if (kCount && lane_rank(__ballot(true)) == 0 && wave_clock[wave * 2 + 1] == 0) wave_clock[wave * 2 + 1] = __builtin_amdgcn_s_memtime();
                    }
                    if (need) {
                        const uint32_t item = base + lane_rank(mask);
                        uint32_t pX, pY;
                        if (item >= P.total_items) result = R_RETIRE;
                        else if (item_to_pixel(S, P, item, pX, pY, tile)) {
                            pI = pY * S.width + pX;
                            rng_init(rng, pI, jump);
                            pk = PK_NEW_PATH;
                            fresh = true;
                        } else result = R_REFETCH;           // item outside the image: ask again
                    }
                }
            }
        }
        if (!have || result != R_NONE) return result;

        uint4* const path4 = reinterpret_cast<uint4*>(rec + kRecPath);
        if (!fresh) {
            const uint4 p0 = path4[0], p1 = path4[1], p2 = path4[2], p3 = path4[3];
            rng.d = p0.x; rng.v0 = p0.y; rng.v1 = p0.z; rng.v2 = p0.w; rng.v3 = p1.x; rng.v4 = p1.y;
            contrib = mk3(__uint_as_float(p1.z), __uint_as_float(p1.w), __uint_as_float(p2.x));
            v = mk3(__uint_as_float(p2.y), __uint_as_float(p2.z), __uint_as_float(p2.w));
            sample = (int)p3.x; segments = p3.y; cost = p3.z; pI = p3.w;
        }

        bool want_shadow = false, alive = false;
        f3 spos = mk3(0, 0, 0), sdir = mk3(0, 0, 1);
        float sdist = 0;
        // the pixel's three accumulators are updated in place (renderer/trace.cu:39, 92-141)
        float* __restrict__ acc_target = S.results + 3ull * pI;
        float* __restrict__ acc_normal = S.normals + 3ull * pI;
        float* __restrict__ acc_albedo = S.albedos + 3ull * pI;
        f3 next_dir = mk3(0, 0, 1);
        // the query's ray as the node loop left it, and its hit
        f3 qpos = mk3(0, 0, 0), qdir = mk3(0, 0, 1), hpos = mk3(0, 0, 0);
        float hz = CRT_BIGFLOAT;

        if (!kClosest && (pk == PK_SHADOW_CLEAR || pk == PK_SHADOW_BLOCKED)) {
            // end of TraceShadowRay: add the light sample if nothing was in the way (trace.cu:137-143)
            const uint4 p4 = path4[4], p5 = path4[5];
            const uint32_t flags = p4.w;
            if (pk == PK_SHADOW_CLEAR && (flags & 1u)) acc_add(acc_target, mk3(__uint_as_float(p5.x), __uint_as_float(p5.y), __uint_as_float(p5.z)));
            alive = (flags & 2u) != 0;
            next_dir = mk3(__uint_as_float(p4.x), __uint_as_float(p4.y), __uint_as_float(p4.z));
            hpos = mk3(__uint_as_float(p5.w), __uint_as_float(path4[6].x), __uint_as_float(path4[6].y));
        } else if (kClosest) {
            // the shadow query that ran beside this closest-hit query: its light sample comes first (trace.cu:137-143)
            if (was_paired) {
                const uint4 p5 = path4[5];
                if (rec[kRecShadowEnd] == PK_SHADOW_CLEAR && (rec[kRecPath + 19] & 1u))
                    acc_add(acc_target, mk3(__uint_as_float(p5.x), __uint_as_float(p5.y), __uint_as_float(p5.z)));
            }
            const uint4* hit4 = reinterpret_cast<const uint4*>(rec + kRecHit);
            const uint4 h0 = hit4[0], h1 = hit4[1], h2 = hit4[2], h3 = hit4[3];
            hz = __uint_as_float(h0.y);
            uint32_t h_kind = h0.z & 3u;
            const bool h_front = (h0.z & 4u) != 0;
            const uint32_t h_node = h0.z >> 3, h_c = h0.w;
            hpos = mk3(__uint_as_float(h1.x), __uint_as_float(h1.y), __uint_as_float(h1.z));
            const float h_a = __uint_as_float(h1.w), h_b = __uint_as_float(h2.w);
            qpos = mk3(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(h2.z));
            qdir = mk3(__uint_as_float(h3.x), __uint_as_float(h3.y), __uint_as_float(h3.z));
            if (h3.w != 0) {
                // the query ended inside a node: world position of a hit found there and Node::FromLocal for the
                // ray (trace.cu:64-66), exactly as a NODE turn of the traversal waves does it
                const float4* __restrict__ xp = reinterpret_cast<const float4*>(S.xf + ((h3.w & 0x7fffffffu) - 1u));
                const float4 r3 = xp[3], r4 = xp[4], r5 = xp[5];
                const float tm[12] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w};
                if (h3.w & 0x80000000u) hpos = xform_point(tm, qpos + hz * qdir);
                f2 px = f2{qpos.x, qdir.x}, py = f2{qpos.y, qdir.y}, pz = f2{qpos.z, qdir.z};
                xform_ray(tm, px, py, pz);
                qpos = mk3(px.x, py.x, pz.x); qdir = mk3(px.y, py.y, pz.y);
            }
            if (kCount && lane_rank(__ballot(true)) == 0) sect_clock[wave] = __builtin_amdgcn_s_memtime();

            // lights are intersected last, with the ray as the node loop left it (trace.cu:72-78)
            int hit_light = -1;
            f3 light_n = mk3(0, 0, 0);
            for (uint32_t i = 0; i < S.light_count; i++) {
                const DevLight Li = S.lights[i];
                if (Li.kind != CRT_LIGHT_POINT) continue;
                // PointLight::Intersect (scene/lights.cu:36-89)
                const f3 lp = ld3(Li.pos);
                const f3 o = qpos - lp;
                const float a = dot(qdir, qdir);
                const float b = dot(2 * qdir, o);
                const float c = dot(o, o) - Li.size * Li.size;
                const float det = b * b - 4 * a * c;
                if (det < 0) continue;
                const float sq = sqrtf(det);
                const float tf = (-b - sq) / (2 * a);
                const float tb = (-b + sq) / (2 * a);
                if (tf >= 0) {
                    if (tf >= hz) continue;
                    hz = tf;
                    light_n = normalized(o);
                    hpos = qpos + tf * qdir;
                    hit_light = (int)i;
                    continue;
                }
                if (tb < 0) continue;
                if (tb >= hz) continue;
                hz = tb;
                light_n = normalized(hpos);      // the previous hit position, as the reference reads it (lights.cu:78)
                hpos = qpos + tb * qdir;
                hit_light = (int)i;
            }
            const bool any = hit_light >= 0 || h_kind != H_NONE;
            SECT(0);

            if (kPrimary) {
                int32_t id = -1, face = -1;
                if (hit_light >= 0) id = -2 - hit_light;
                else if (any) { id = (int32_t)h_node; if (h_kind == H_MESH) face = (int32_t)tris[h_c].face; }
                out_node[pI] = id; out_face[pI] = face; out_t[pI] = hz;
            } else if (!any) {
                const f3 env = tex_env<kCount>(S, qdir, tally);
                acc_add(acc_target, contrib * env);
                acc_add(acc_normal, contrib * normalized(qdir));
                acc_add(acc_albedo, contrib * env);
            } else if (hit_light >= 0) {
                const f3 rad = ld3(S.lights[hit_light].radiance);
                acc_add(acc_target, contrib * rad);
                acc_add(acc_normal, contrib * light_n);
                acc_add(acc_albedo, contrib * rad);
            } else {
                // normal and texture coordinates of the hit (objects.cu:44-50, 126-131, 308-320) and
                // Node::FromLocal for the normal (scene/scene.cuh:185-189)
                const NodeXf* __restrict__ xf = S.xf + h_node;
                f3 n, uvw;
                if (h_kind == H_MESH) {
                    // interpolated vertex normal and texture vertex of the hit triangle (objects.cu:308-320)
                    const float b1 = h_a, b2 = h_b;
                    const float b0 = 1 - b2 - b1;
                    const float4* ap = reinterpret_cast<const float4*>(S.attrs + h_c);
                    const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4];
                    const uint32_t mesh_flags = xf->flags;
                    if (!(mesh_flags & kXfHasNormals)) {
                        const float4* tp = reinterpret_cast<const float4*>(tris + h_c);
                        const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                        const f3 p0 = mk3(v0.x, v0.y, v0.z);
                        n = normalized(cross(mk3(v1.x, v1.y, v1.z) - p0, mk3(v2.x, v2.y, v2.z) - p0));
                    } else {
                        n = mk3(a0.x, a0.y, a0.z) * b0 + mk3(a0.w, a1.x, a1.y) * b1 + mk3(a1.z, a1.w, a2.x) * b2;
                    }
                    if (!(mesh_flags & kXfHasTex)) uvw = mk3(0, 0, 0);
                    else uvw = mk3(a2.y, a2.z, a2.w) * b0 + mk3(a3.x, a3.y, a3.z) * b1 + mk3(a3.w, a4.x, a4.y) * b2;
                } else {
                    const f3 p = mk3(h_a, h_b, __uint_as_float(h_c));
                    if (h_kind == H_SPHERE) {
                        n = p;
                        uvw = mk3(0.5f * 0.318309f * crtm::det_atan2f(p.y, p.x) + 0.5f,
                                  0.318309f * crtm::det_asinf(fminf(fmaxf(p.z, -1.0f), 1.0f)) + 0.5f, 0);
                    } else {
                        n = mk3(0, 0, 1);
                        uvw = 0.5f * (p + mk3(1, 1, 0));
                    }
                }
                n = xform_normal(xf->itm, n);
                SECT(1);

                const float light_pick = rng_float(rng);
                const DevMaterial mtl = S.materials[xf->material_id];
                const f3 kD = tex_eval<kCount>(S, mtl.diffuse, uvw, tally);
                const f3 kS = tex_eval<kCount>(S, mtl.specular, uvw, tally);
                const f3 kT = tex_eval<kCount>(S, mtl.refraction, uvw, tally);
                const f3 kE = tex_eval<kCount>(S, mtl.emission, uvw, tally);
                SECT(2);
                acc_add(acc_normal, contrib * n);
                acc_add(acc_albedo, contrib * kD + kE);
                acc_add(acc_target, contrib * kE);

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

                // next-event estimation toward one light (trace.cu:112-143).  Ambient and directional
                // lights produce no sample and zero weight, so nothing is traced for them; with no
                // light at all (where the reference reads a null light) no sample is taken either.
                bool nee_valid = false;
                f3 nee = mk3(0, 0, 0);
                if (S.light_count > 0) {
                    const int light_id = crtm::f2i_sat(light_pick * (float)S.light_count);
                    const DevLight Ls = S.lights[light_id];
                    if (Ls.kind == CRT_LIGHT_POINT) {
                        // PointLight::GenerateSample (scene/lights.cu:10-34)
                        const f3 lp = ld3(Ls.pos);
                        const f3 Ldir = normalized(hpos - lp);
                        const float x = rng_float(rng);
                        const float phi = rng_float(rng) * 2 * F_PI;
                        const float cos_theta = 1 - x;
                        const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
                        f3 a, b;
                        orthonormals(Ldir, a, b);
                        float sin_phi, cos_phi;
                        crtm::det_sincosf(phi, sin_phi, cos_phi);
                        const f3 ln = Ldir * cos_theta + a * sin_theta * cos_phi + b * sin_theta * sin_phi;
                        const f3 tgt = lp + ln * Ls.size;
                        const f3 ldir = normalized(tgt - hpos);
                        const float d2 = lengthsq(tgt - hpos);
                        const float cos_theta_l = dot(ldir, -normalized(tgt - lp));
                        const float l_prob = d2 / (2 * F_PI * Ls.size * Ls.size * cos_theta_l);
                        const f3 l_mult = ld3(Ls.radiance);
                        sdist = sqrtf(d2);
                        // Material::GetSampleInfo (renderer/shading.cu:116-150)
                        float b_prob = 0.0f;
                        f3 b_mult = mk3(0, 0, 0);
                        const float ndl = dot(n, ldir);
                        if (ndl >= 0) {
                            b_prob += pD * (ndl > 0 ? OVERPI : 0);
                            b_mult += kD * OVERPI * ndl;
                            const f3 rn = normalized(v + ldir);
                            const float ndh = dot(n, rn);
                            b_prob += pS * (g + 1) * 0.5f * OVERPI * crtm::det_powf(ndh, g + 1);
                            b_mult += kS * (g + 2) * 0.125f * OVERPI * crtm::det_powf(ndh, g);
                        }
                        const float denom = l_prob * l_prob + b_prob * b_prob;
                        if (denom > CRT_F_EPS && l_prob > CRT_F_EPS) {
                            const float pop = l_prob / denom;
                            nee = contrib * l_mult * b_mult * pop;
                            nee_valid = true;
                        }
                        // TraceShadowRay prologue (trace.cu:160-162)
                        spos = hpos; sdir = ldir;
                        spos += sdir * CRT_BIAS + n * CRT_BIAS;
                        want_shadow = true;
                    }
                }

                SECT(3);
                // Material::GenerateSample (renderer/shading.cu:11-113)
                f3 kA = mk3(1, 1, 1);
                if (!h_front) {
                    const f3 e = ld3(mtl.absorption) * -hz;
                    kA = mk3(crtm::det_expf(e.x), crtm::det_expf(e.y), crtm::det_expf(e.z));
                }
                const float p = rng_float(rng);
                f3 ndir = mk3(0, 0, 0), n_mult = mk3(0, 0, 0);
                float n_prob = 0;
                bool sampled = true;
                if (p < pD) {
                    const float x = rng_float(rng);
                    const float phi = rng_float(rng) * 2 * F_PI;
                    const float cos_theta = 1 - x;
                    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
                    f3 u, w;
                    orthonormals(n, u, w);
                    float sin_phi, cos_phi;
                    crtm::det_sincosf(phi, sin_phi, cos_phi);
                    ndir = n * cos_theta + u * sin_theta * cos_phi + w * sin_theta * sin_phi;
                    n_prob = pD * OVERPI;
                    n_mult = kA * kD * OVERPI * dot(n, ndir);
                } else {
                    const f3 gln = glossy_normal(n, g, rng);
                    if (p < pD + pS) {
                        ndir = reflect(v, gln);
                        const float pw = crtm::det_powf(dot(gln, n), g);
                        n_prob = pS * (g + 1) * 0.125f * OVERPI * pw;
                        n_mult = kA * kS * (g + 2) * 0.125f * OVERPI * pw;
                    } else if (p < pD + pS + pT) {
                        const float eta_i = h_front ? 1 : mtl.ior;
                        const float eta_o = h_front ? mtl.ior : 1;
                        const f3 h = h_front ? gln : -gln;
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
                        sampled = false;
                    }
                }
                // the path continues unless the sample died, its probability vanished or the safety bound hit
                if (sampled) {
                    next_dir = ndir;
                    if (!(n_prob <= CRT_F_EPS)) {
                        contrib *= n_mult * rcp_ieee(n_prob);
                        alive = ++segments < CRT_MAX_SEGMENTS;
                    }
                }
                if (want_shadow) {
                    // parked until the shadow query comes back: continuation direction, light sample, hit position
                    // (flag 4: the continuation is under way beside the shadow query)
                    path4[4] = uint4{__float_as_uint(next_dir.x), __float_as_uint(next_dir.y), __float_as_uint(next_dir.z), (nee_valid ? 1u : 0u) | (alive ? (pair_now ? 6u : 2u) : 0u)};
                    path4[5] = uint4{__float_as_uint(nee.x), __float_as_uint(nee.y), __float_as_uint(nee.z), __float_as_uint(hpos.x)};
                    rec[kRecPath + 24] = __float_as_uint(hpos.y); rec[kRecPath + 25] = __float_as_uint(hpos.z);
                }
            }
        }

        SECT(4);
        // ---- what comes next for this pixel
        f3 rpos, rdir;
        float rlimit;
        bool rshadow;
        // A light sample whose path goes on: the shadow query and the next segment's closest-hit query may start
        // together (pair_now, the same for the whole turn).
        const bool two = want_shadow && alive && pair_now;
        if (want_shadow) tally.shadow_rays++;
        if (want_shadow && !two) {
            rpos = spos; rdir = sdir; rlimit = sdist; rshadow = true;
            result = R_RAY;
        } else {
            bool have_ray = true;
            rpos = hpos; rdir = next_dir;
            if (alive) {
                // continue from the hit position (trace.cu:150-156)
            } else if (pk != PK_NEW_PATH && ++sample >= samples) {
                have_ray = false;
                if (!kPrimary && P.tile_cost) atomicMax(P.tile_cost + (fresh ? tile : rec[kRecPath + 26]), cost);
            } else {
                // a new path: camera ray (trace.cu:26-38)
                const uint32_t pY = pI / S.width, pX = pI - pY * S.width;
                camera_ray(S, pX, pY, rng, rpos, rdir);
                contrib = mk3(1, 1, 1) * inv_samples;
                segments = 0;
                TALLY(paths);
            }
            if (have_ray) {
                // TraceThroughScene prologue (renderer/trace.cu:44-50)
                v = -normalized(rdir);
                rpos = rpos + rdir * CRT_BIAS;
                rlimit = CRT_BIGFLOAT;
                rshadow = false;
                tally.closest_rays++;
                result = two ? R_TWO_RAYS : R_RAY;
            } else {
                rlimit = 0; rshadow = false;
                result = R_REFETCH;
            }
        }
        path4[0] = uint4{rng.d, rng.v0, rng.v1, rng.v2};
        path4[1] = uint4{rng.v3, rng.v4, __float_as_uint(contrib.x), __float_as_uint(contrib.y)};
        path4[2] = uint4{__float_as_uint(contrib.z), __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
        path4[3] = uint4{(uint32_t)sample, segments, cost + (two ? 2u : 1u), pI};
        if (fresh) rec[kRecPath + 26] = tile;
        // a query's record: the ray, and one pass over the node boxes for the scan's candidate mask
        auto write_query = [&](uint32_t* dst, f3 qpos_, f3 qdir_, float qlimit, bool qshadow, bool& direct) {
            NodeMask mask = {~0ull, ~0ull};
            uint32_t node = 0;
            if (S.scan_mask) {
                mask = grouped ? scan_candidates(S, S.group_pairs, S.group_count, qpos_, qdir_, qlimit, false)
                               : scan_candidates(S, pairs, node_count, qpos_, qdir_, qlimit, qshadow);
                const uint32_t first = mask.lo ? (uint32_t)__ffsll((long long)mask.lo) - 1u : (mask.hi ? 63u + (uint32_t)__ffsll((long long)mask.hi) : 0xffffffffu);
                if (by_mask) node = first < node_count ? first : node_count;
                if (by_group) node = first < S.group_count ? first << group_shift : node_count;
            }
            f3 rinv = recip3(qdir_);
            // The first step of the scan is taken here, where every lane has a query, rather than by the few
            // traversal lanes of a NODE turn: the box of the lowest candidate (trace.cu:55, 170) and, when the query
            // has to enter a mesh through it, Node::ToLocal (trace.cu:61, 188).  Same operations on the same values as
            // op_node's, so nothing changes but who carries them out.  A query that finds the box missed starts at
            // its next candidate; one that would enter anything but a mesh with an interior root starts as before.
            uint32_t entered = 0, root = 0;
            if (first_now && by_mask && node < node_count) {
                const float4* rp = reinterpret_cast<const float4*>(pairs + node);
                const float4 q0 = rp[0], q1 = rp[1], q3 = rp[3];
                const uint32_t ref_a = __float_as_uint(q3.x), after_a = node_ref_after(ref_a);
                float dA;
                const bool hitA = slab(f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, qpos_, rinv, qlimit, dA);
                const bool enter = hitA & (qshadow | (dA < CRT_BIGFLOAT));
                unsigned long long m = mask.lo & ~(1ull << node);
                if (qshadow & !hitA) m &= ~(((1ull << (after_a & 63u)) - 1ull) | (after_a >= 64u ? ~0ull : 0ull)) | ((2ull << node) - 1ull);
                if (!enter) {
                    mask.lo = m;
                    node = m ? (uint32_t)__ffsll((long long)m) - 1u : node_count;
                } else {
                    const uint32_t kind = node_ref_kind(ref_a);
                    const float4* __restrict__ xp = reinterpret_cast<const float4*>(S.xf + node);
                    const float4 r0 = xp[0], r1 = xp[1], r2 = xp[2], r6 = xp[6];
                    const float itm[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                    f2 px = f2{qpos_.x, qdir_.x}, py = f2{qpos_.y, qdir_.y}, pz = f2{qpos_.z, qdir_.z};
                    xform_ray(itm, px, py, pz);
                    const f3 lpos = mk3(px.x, py.x, pz.x), ldir = mk3(px.y, py.y, pz.y);
                    if (kind == CRT_OBJ_MESH) {
                        if (!ref_is_leaf(__float_as_uint(r6.x))) {
                            qpos_ = lpos; qdir_ = ldir;
                            rinv = recip3(qdir_);
                            mask.lo = m;
                            entered = 4u; root = __float_as_uint(r6.x);
                        }
                    } else {
                        // sphere / plane: the test itself as well; the query goes on from the point where it leaves the node
                        // (in place of 1 / direction, which that step computes afresh, the record carries the local hit point)
                        bool front;
                        float t;
                        const bool got = unit_object_hit(kind == CRT_OBJ_SPHERE, lpos, ldir, qshadow, qlimit, CRT_BIGFLOAT, t, front);
                        if (!(qshadow && got)) {      // (a blocked shadow query: left to the traversal lane, which ends it)
                            qpos_ = lpos; qdir_ = ldir;
                            rinv = lpos + t * ldir;
                            mask.lo = m;
                            entered = 8u | (got ? 16u : 0u) | (front ? 32u : 0u) | (kind == CRT_OBJ_SPHERE ? 64u : 0u);
                            root = __float_as_uint(t);
                        }
                    }
                }
            }
            const uint4 w0 = uint4{__float_as_uint(qpos_.x), __float_as_uint(qpos_.y), __float_as_uint(qpos_.z), __float_as_uint(qlimit)};
            const uint4 w1 = uint4{__float_as_uint(qdir_.x), __float_as_uint(qdir_.y), __float_as_uint(qdir_.z), (qshadow ? 1u : 0u) | (two ? 2u : 0u) | entered};
            const uint4 w2 = uint4{__float_as_uint(rinv.x), __float_as_uint(rinv.y), __float_as_uint(rinv.z), node};
            const uint4 w3 = uint4{(uint32_t)mask.lo, (uint32_t)(mask.lo >> 32), kWide ? (uint32_t)mask.hi : root, (uint32_t)(mask.hi >> 32)};
            // Regrouping kernel: when a free query context is within this lane's reach the query goes straight into
            // it and into the pool of the operation it starts with - no RAY record, no ray ring, no traversal lane
            // spent on picking it up.  Otherwise (every context of the lane's positions is busy) it waits in the
            // slot's RAY record and the ray ring for a traversal lane with a free context.
            direct = false;
            if (kRegroup && S.direct_start) {
                uint32_t col = 0;
                if (claim(kQIdle, true, 64u, col)) {
                    L.col = col;
                    const uint32_t begin = query_start(slot, w0, w1, w2, w3);
                    uint4* const c = ctx_words();
                    c[0] = put_c0(); c[1] = put_c1(); c[2] = put_c2(); c[3] = put_c3(); if (!kTlas) c[4] = put_c4(); c[5] = put_c5(); c[6] = put_c6();
                    state = begin;
                    direct = true;
                }
            }
            if (!direct) {
                uint4* ray4 = reinterpret_cast<uint4*>(dst);
                ray4[0] = w0; ray4[1] = w1; ray4[2] = w2; ray4[3] = w3;
            }
        };
        if (result == R_RAY || result == R_TWO_RAYS) {
            // (a query that went straight into a context is parked right away: the lane's context registers serve the second query next)
            write_query(rec + kRecRay, rpos, rdir, rlimit, rshadow, direct_first);
            if (direct_first) { if (!lose_first) park(true); state = S_EMPTY; }
            if (kClosest && two) {
                write_query(rec + kRecRay2, spos, sdir, sdist, true, direct_second);
                if (direct_second) { park(true); state = S_EMPTY; }
            }
        } else {
            rec[kRecHit] = PK_FETCH;
        }
        SECT(5);
        return result;
    };

#undef SECT

    // ---- one step of a traversal wave of the regrouping kernel -----------------------------------------------------
    // A wave runs ONE operation with as many of its lanes as it can fill: the lanes that wait for something else park
    // their queries, free lanes take parked queries that wait for the wave's operation.  The wave stays with its
    // operation while enough lanes stay with it; then it votes: the operation with the most queries within reach of its
    // lanes is next.  Returns 0: leave the kernel, 1: an operation ran, 2: nothing within reach.
    uint32_t mode = S_POST;
    uint32_t fruitless = 0;      // votes in a row after which no lane of the wave moved
    // turns of one traversal operation, until too few lanes can be filled for it
    auto run = [&](auto op_tag, uint32_t keep) {
        constexpr uint32_t kOp = decltype(op_tag)::value;
        unsigned long long t_mark = 0;
        if (kCount) t_mark = __builtin_amdgcn_s_memtime();
        // what is parked for this operation at the lane's positions: read at the end of a turn for the next one, so
        // that the wait for it overlaps the parking of the lanes that leave
        uint32_t own, partner;
        parked(kOp, own, partner);
        for (uint32_t turns = 0;;) {
            const bool other = state != S_EMPTY && state != kOp;
            if (__ballot(other) != 0) park(other);
            { uint32_t col = 0; if (claim_from(kOp, own, partner, state == S_EMPTY, 64u, col)) { L.col = col; state = kOp; } }
            const bool active = state == kOp;
            if (__ballot(active) == 0) break;
            fruitless = 0;
            if (kCount) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tally.cycles[8] += t - t_mark; t_mark = t; tally.turns[kOp]++; tally.lanes[kOp] += active; }
            uint4* const c = ctx_words();
            if (kOp == S_BOX) {
                const uint4 c0 = c[0], c2 = c[2], c3 = c[3];
                get_c0(c0); get_c2(c2); get_c3(c3);
                if (active) { state = op_box(); c[3] = put_c3(); }
            } else if (kOp == S_LEAF) {
                const uint4 c0 = c[0], c1 = c[1], c3 = c[3];
                get_c0(c0); get_c1(c1); get_c3(c3);
                if (active) {
                    leaf_hit = false;
                    state = op_leaf();
                    c[3] = put_c3();
                    if (leaf_hit) c[5] = put_c5();
                }
            } else {
                const uint4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6];
                get_c0(c0); get_c1(c1); get_c2(c2); get_c3(c3); if (!kTlas) get_c4(c4); get_c5(c5); get_c6(c6);
                if (active) {
                    state = op_node();
                    c[0] = put_c0(); c[1] = put_c1(); c[2] = put_c2(); c[3] = put_c3(); if (!kTlas) c[4] = put_c4(); c[5] = put_c5(); c[6] = put_c6();
                }
            }
            if (kCount) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tally.cycles[kOp] += t - t_mark; t_mark = t; }
            // another turn without a vote while enough lanes stay with the operation (free lanes take what is parked for it)
            parked(kOp, own, partner);
            const uint32_t stay = (uint32_t)__popcll(__ballot(state == kOp));
            if (stay < keep || ++turns >= S.burst_max) break;
        }
        if (kCount) tally.cycles[8] += __builtin_amdgcn_s_memtime() - t_mark;
    };
    // (Nothing of a query outlives a step in the lane's registers; saying so keeps their old contents from being carried
    // through the other kind of step.)
    auto forget_query = [&]() { const uint32_t col = L.col; L = Lane{}; L.col = col; };
    auto trav_step = [&]() -> int {
        forget_query();
        unsigned long long t_vote = 0;
        if (kCount) t_vote = __builtin_amdgcn_s_memtime();
        // A wave that keeps voting without moving any query has met a case the protocol does not cover: it
        // reports the error and leaves, as does every wave once one has.
        bool give_up = false;
        if ((++fruitless & 255u) == 0) {
            if (lds_ld(&ctl.abort) != 0) return 0;
            give_up = fruitless > (1u << 20);
        }
        // lanes this wave could run each operation with: its own that wait for it, and parked ones of every other lane's position
        uint32_t pB, pL, pN, pP, pI;
        { uint32_t o, q; parked(kQBox, o, q); pB = o | q; parked(kQLeaf, o, q); pL = o | q; parked(kQNode, o, q); pN = o | q; parked(kQPost, o, q); pP = o | q; parked(kQIdle, o, q); pI = o | q; }
        const uint32_t h_ray = lds_ld(&ctl.head[kQRay]), fR = lds_ld(&ctl.tail[kQRay]) - h_ray;
        const uint32_t cB = (uint32_t)__popcll(__ballot(state == S_BOX || pB != 0)), cL = (uint32_t)__popcll(__ballot(state == S_LEAF || pL != 0));
        const uint32_t cN = (uint32_t)__popcll(__ballot(state == S_NODE || pN != 0));
        const bool can_post = state == S_POST || pP != 0;
        const uint32_t posts = (uint32_t)__popcll(__ballot(can_post));
        const uint32_t idles = (uint32_t)__popcll(__ballot(!can_post && (state == S_IDLE || pI != 0)));
        // an exchange serves the lanes with a result to post (they take a new query like the idle ones) and as many
        // free contexts as the ray ring has queries for
        const uint32_t takers = fR > posts ? (fR - posts < idles ? fR - posts : idles) : 0u;
        const uint32_t cX = posts + takers;
        if (kCount) {      // census at every vote
            tally.turns[6]++; tally.lanes[6] += (state == S_IDLE || state == S_EMPTY); tally.lanes[7] += state == S_POST;
            if (lane == 0) { tally.cycles[6] += fR; tally.cycles[7] += (lds_ld(&ctl.tail[kQShade0]) - lds_ld(&ctl.head[kQShade0])) + (lds_ld(&ctl.tail[kQShade1]) - lds_ld(&ctl.head[kQShade1])); }
        }
        if (give_up) {
            lds_st(&ctl.abort, 1u);
            if (lane == 0) atomicAdd(&stats->errors, 1ull);
            return 0;
        }
        const uint32_t c_mode = mode == S_BOX ? cB : (mode == S_LEAF ? cL : (mode == S_NODE ? cN : cX));
        const uint32_t keep_mode = mode == S_BOX ? S.keep_box : (mode == S_LEAF ? S.keep_leaf : (mode == S_NODE ? S.keep_node : S.keep_exchange));
        if (cX >= S.exchange_first) mode = S_POST;      // queries in, results out: before anything else once there are enough of them
        else if (c_mode < keep_mode) {
            uint32_t best = S_POST, best_c = cX;
            if (cN > best_c) { best = S_NODE; best_c = cN; }
            if (cL > best_c) { best = S_LEAF; best_c = cL; }
            if (cB > best_c) { best = S_BOX; best_c = cB; }
            if (best_c == 0) {
                // nothing within reach: no query anywhere in the block that this wave's lanes could move
                fruitless = 0;
                return 2;
            }
            mode = best;
        }
        idle_polls = 0;
        if (kCount) tally.cycles[8] += __builtin_amdgcn_s_memtime() - t_vote;
        if (mode == S_BOX) run(std::integral_constant<uint32_t, S_BOX>{}, S.keep_box);
        else if (mode == S_LEAF) run(std::integral_constant<uint32_t, S_LEAF>{}, S.keep_leaf);
        else if (mode == S_NODE) run(std::integral_constant<uint32_t, S_NODE>{}, S.keep_node);
        else {
            // lanes that wait for a traversal operation park, and so do free contexts beyond the queries the ray ring
            // has for them (a lane that holds one could not take a result to post); free lanes take results to
            // post, then free contexts for the queries that the lanes at hand cannot take
            unsigned long long t_start = 0;
            if (kCount) t_start = __builtin_amdgcn_s_memtime();
            const bool leave = state <= S_NODE || (state == S_IDLE && lane_rank(__ballot(state == S_IDLE)) >= fR);
            if (__ballot(leave) != 0) park(leave);
            unpark(kQPost, state == S_EMPTY, 64u);
            const uint32_t at_hand = (uint32_t)__popcll(__ballot(state == S_POST || state == S_IDLE));
            if (fR > at_hand) unpark(kQIdle, state == S_EMPTY, fR - at_hand);
            if (kCount) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tally.cycles[8] += t - t_start; t_start = t; tally.turns[4]++; tally.lanes[4] += (state == S_POST || state == S_IDLE); }
            const unsigned long long waiting = __ballot(state == S_IDLE);
            if (__ballot(state == S_POST) != 0) fruitless = 0;
            uint4* const c = ctx_words();
            {
                const uint4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c5 = c[5], c6 = c[6];
                get_c0(c0); get_c1(c1); get_c2(c2); get_c3(c3); get_c5(c5); get_c6(c6);
            }
            op_exchange();
            if (took) { c[0] = put_c0(); c[1] = put_c1(); c[2] = put_c2(); c[3] = put_c3(); if (!kTlas) c[4] = put_c4(); c[5] = put_c5(); c[6] = put_c6(); }
            if (__ballot(state == S_IDLE) != waiting) fruitless = 0;
            if (kCount) tally.cycles[4] += __builtin_amdgcn_s_memtime() - t_start;
        }
        return 1;
    };

    // ---- one step of a shading wave: take up to 64 slots, shade them, hand each one on -------------------------------
    // Returns 1: a turn was run, 2: nothing to shade.
    uint32_t patience = 0;
    auto shade_step = [&]() -> int {
        if (kRegroup) forget_query();
        // Full turns when the traversal lanes have queries in reserve; while their queue runs low a partial turn
        // is better than letting them idle.
        uint32_t min_take = 1;
        {
            const uint32_t ready = lds_ld(&ctl.tail[kQRay]) - lds_ld(&ctl.head[kQRay]);
            if (ready >= S.serve_low_water && patience < S.serve_patience) min_take = S.serve_batch_min;
        }
        // the fuller of the two shading queues
        const uint32_t fill0 = lds_ld(&ctl.tail[kQShade0]) - lds_ld(&ctl.head[kQShade0]);
        const uint32_t fill1 = lds_ld(&ctl.tail[kQShade1]) - lds_ld(&ctl.head[kQShade1]);
        const uint32_t which = fill1 > fill0 ? 1u : 0u;
        uint32_t slot = 0;
        const bool have = queue_pop<kQLog>(&ctl.head[kQShade0 + which], &ctl.tail[kQShade0 + which], shade_ring[which], true, min_take, n_slots, slot, &ctl.abort);
        if (__ballot(have) == 0) {
            patience++;
            return 2;
        }
        patience = 0; idle_polls = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        unsigned long long t_start = 0;
        if (kCount) { t_start = __builtin_amdgcn_s_memtime(); tally.turns[3]++; tally.lanes[3] += have; }
        // fault injection (tests of the error path): the first query the first lane of a wave of block 0 sets up is never
        // handed on, so its slot never comes back
        lose_first = S.fault_inject && !fault_done && blockIdx.x == 0 && have && lane_rank(__ballot(have)) == 0;
        fault_done = fault_done || S.fault_inject;
        direct_first = false; direct_second = false;
        const uint32_t result = which == 0 ? serve(std::true_type{}, have, slot) : serve(std::false_type{}, have, slot);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        queue_push<kRLog>(&ctl.tail[kQRay], ray_ring, (result == R_RAY || result == R_TWO_RAYS) && !direct_first && !lose_first, slot);
        queue_push<kRLog>(&ctl.tail[kQRay], ray_ring, result == R_TWO_RAYS && !direct_second, slot | kSecond);
        queue_push<kQLog>(&ctl.tail[kQShade1], shade_ring[1], result == R_REFETCH, slot);
        const unsigned long long retired = __ballot(result == R_RETIRE);
        if (retired != 0 && lane_rank(__ballot(true)) == 0) atomicSub(&ctl.live, (uint32_t)__popcll(retired));
        if (kCount) tally.cycles[3] += __builtin_amdgcn_s_memtime() - t_start;
        return 1;
    };

    // ---- the waves' loops ----------------------------------------------------------------------------------------
    if constexpr (kRegroup) {
        // The first S.trav_waves waves traverse, the others shade (at raised priority: the traversal lanes of the whole
        // block wait on what they produce; they sleep when they have nothing).  A traversal wave keeps nothing of a query
        // in its registers between two turns, so it can take a shading turn when the shading queues have run long
        // (S.swing_fill slots waiting): the split between the two kinds of work then follows the frame, not a constant.
        const bool shader = wave >= S.trav_waves;
        if (shader && S.serve_priority) __builtin_amdgcn_s_setprio(3);
        for (;;) {
            int r;
            if (shader) r = shade_step();
            else {
                bool swing = false;
                if (wave + S.swing_waves >= S.trav_waves) {
                    const uint32_t waiting = (lds_ld(&ctl.tail[kQShade0]) - lds_ld(&ctl.head[kQShade0])) + (lds_ld(&ctl.tail[kQShade1]) - lds_ld(&ctl.head[kQShade1]));
                    swing = waiting >= S.swing_fill && waiting <= kQSize;
                }
                if (swing) {
                    // the wave's lanes give up the contexts they hold: a shading turn uses the lane's context registers
                    if (__ballot(state != S_EMPTY) != 0) park(state != S_EMPTY);
                    if (S.serve_priority) __builtin_amdgcn_s_setprio(3);
                    r = shade_step();
                    if (S.serve_priority) __builtin_amdgcn_s_setprio(0);
                    if (r == 2) r = trav_step();
                } else r = trav_step();
            }
            if (r == 0) break;
            if (r == 2 && !idle_wait()) break;
        }
    } else if (wave < S.trav_waves) {
        for (;;) {
            // Which operation runs this turn.  Box steps are the bulk of the work and run whenever
            // sched_box_min lanes want one.  Otherwise a cheap traversal operation (triangle, transform)
            // runs once sched_special_min lanes wait for it, an exchange once sched_exchange_min lanes have a
            // result to post or nothing to do; failing that whatever has lanes.
            uint32_t n_box = (uint32_t)__popcll(__ballot(state == S_BOX));
            // box burst: this inner loop carries only the few values a box step changes.  It ends as soon as enough
            // lanes have a result to post: an exchange is cheap and gives them their next query.
            while (n_box >= S.sched_box_min && (uint32_t)__popcll(__ballot(state == S_POST)) < S.sched_exchange_min) {
                unsigned long long t_box = 0;
                if (kCount) { t_box = __builtin_amdgcn_s_memtime(); tally.turns[0]++; tally.lanes[0] += state == S_BOX; }
                if (state == S_BOX) state = op_box();
                if (kCount) tally.cycles[0] += __builtin_amdgcn_s_memtime() - t_box;
                n_box = (uint32_t)__popcll(__ballot(state == S_BOX));
            }
            uint32_t best;
            {
                const uint32_t n_leaf = (uint32_t)__popcll(__ballot(state == S_LEAF));
                const uint32_t n_xf = (uint32_t)__popcll(__ballot(state == S_NODE));
                const uint32_t n_post = (uint32_t)__popcll(__ballot(state == S_POST));
                const uint32_t n_idle = 64u - n_box - n_leaf - n_xf - n_post;
                const uint32_t special = n_leaf >= n_xf ? (uint32_t)S_LEAF : (uint32_t)S_NODE;
                const uint32_t n_special = n_leaf >= n_xf ? n_leaf : n_xf;
                // idle lanes count toward an exchange only while the ray queue has something for them
                uint32_t n_take = 0;
                if (n_idle) { const uint32_t avail = lds_ld(&ctl.tail[kQRay]) - lds_ld(&ctl.head[kQRay]); n_take = avail < n_idle ? avail : n_idle; }
                const uint32_t n_ex = n_post + n_take;
                if (kCount) {      // census at every vote: lanes without a query, and (once per wave) the fill of the two queues
                    tally.turns[6]++; tally.lanes[6] += state == S_IDLE; tally.lanes[7] += state == S_POST;
                    if ((tid & 63u) == 0) { tally.cycles[6] += lds_ld(&ctl.tail[kQRay]) - lds_ld(&ctl.head[kQRay]); tally.cycles[7] += (lds_ld(&ctl.tail[kQShade0]) - lds_ld(&ctl.head[kQShade0])) + (lds_ld(&ctl.tail[kQShade1]) - lds_ld(&ctl.head[kQShade1])); }
                }
                if (n_ex >= S.sched_exchange_min) best = S_POST;
                else if (n_special >= S.sched_special_min) best = special;
                else if (n_box > 0) best = S_BOX;
                else if (n_special > 0) best = special;
                else if (n_ex > 0) best = S_POST;
                else {
                    // every lane idle and no query ready
                    if (!idle_wait()) break;
                    continue;
                }
            }
            idle_polls = 0;
            unsigned long long t_start = 0;
            if (kCount) t_start = __builtin_amdgcn_s_memtime();
            switch (best) {
                case S_BOX: {
                    // below the burst threshold: keep stepping while every lane that took the step wants another one
                    // (the lone lanes at the end of a frame are the common case), then vote again
                    uint32_t n_now;
                    do {
                        if (kCount) { tally.turns[0]++; tally.lanes[0] += state == S_BOX; }
                        if (state == S_BOX) state = op_box();
                        n_now = (uint32_t)__popcll(__ballot(state == S_BOX));
                    } while (n_now == n_box);
                    break;
                }
                case S_LEAF: if (kCount) { tally.turns[1]++; tally.lanes[1] += state == S_LEAF; } if (state == S_LEAF) state = op_leaf(); break;
                case S_NODE: if (kCount) { tally.turns[2]++; tally.lanes[2] += state == S_NODE; } if (state == S_NODE) state = op_node(); break;
                default:
                    if (kCount) { tally.turns[4]++; tally.lanes[4] += (state == S_POST || state == S_IDLE); }
                    op_exchange();
                    break;
            }
            if (kCount) {
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
                switch (best) { case S_BOX: tally.cycles[0] += dt; break; case S_LEAF: tally.cycles[1] += dt; break; case S_NODE: tally.cycles[2] += dt; break; default: tally.cycles[4] += dt; break; }
            }
        }
    } else {
        if (S.serve_priority) __builtin_amdgcn_s_setprio(3);
        for (;;) {
            const int r = shade_step();
            if (r == 2 && !idle_wait()) break;
        }
    }

    // ---- counters: one atomic per wave and field
    const unsigned long long c_closest = wave_sum(tally.closest_rays), c_shadow = wave_sum(tally.shadow_rays);
    unsigned long long c_sc = 0, c_sv = 0;
    unsigned long long c_paths = 0, c_nb = 0, c_ne = 0, c_bi = 0, c_bl = 0, c_tt = 0, c_th = 0, c_sp = 0, c_pl = 0, c_tx = 0;
    if (kCount) {
        c_paths = wave_sum(tally.paths); c_nb = wave_sum(tally.node_box_tests); c_ne = wave_sum(tally.node_enters);
        c_bi = wave_sum(tally.bvh_interior); c_bl = wave_sum(tally.bvh_leaves); c_tt = wave_sum(tally.tri_tests);
        c_th = wave_sum(tally.tri_hits); c_sp = wave_sum(tally.sphere_tests); c_pl = wave_sum(tally.plane_tests);
        c_tx = wave_sum(tally.texel_quads);
        c_sc = wave_sum(tally.scan_checks); c_sv = wave_sum(tally.scan_violations);
    }
    if (!kPrimary && (tid & 63u) == 0) {
        if (c_closest) atomicAdd(&stats->closest_rays, c_closest);
        if (c_shadow) atomicAdd(&stats->shadow_rays, c_shadow);
        if (kCount) {
            atomicAdd(&stats->paths, c_paths); atomicAdd(&stats->node_box_tests, c_nb); atomicAdd(&stats->node_enters, c_ne);
            atomicAdd(&stats->bvh_interior, c_bi); atomicAdd(&stats->bvh_leaves, c_bl); atomicAdd(&stats->tri_tests, c_tt);
            atomicAdd(&stats->tri_hits, c_th); atomicAdd(&stats->sphere_tests, c_sp); atomicAdd(&stats->plane_tests, c_pl);
            atomicAdd(&stats->texel_quads, c_tx);
            atomicAdd(&stats->scan_checks, c_sc); atomicAdd(&stats->scan_violations, c_sv);
        }
    }
    if (lds_ld(&ctl.abort) != 0 && tid == 0) atomicAdd(&stats->errors, 1ull);
    // The pixel queue's counter resets itself when the launch is told to (the entry point that must not even enqueue a
    // memset): queue[1] counts the blocks that are through, and the last one clears both words for the next launch.
    if (P.self_reset) {
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            if (atomicAdd(&queue[1], 1u) == gridDim.x - 1u) { queue[0] = 0; queue[1] = 0; __threadfence(); }
        }
    }
    if (kCount && !kPrimary) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned long long t = wave_sum(tally.turns[k]), l = wave_sum(tally.lanes[k]);
            if ((tid & 63u) == 0) { atomicAdd(&stats->sched_turns[k], t / 64); atomicAdd(&stats->sched_lanes[k], l); atomicAdd(&stats->sched_cycles[k], tally.cycles[k]); }
        }
        for (int k = 0; k < 6; k++)
            if ((tid & 63u) == 0) atomicAdd(&stats->path_cycles[k], path_sections[wave * 8 + k]);
        if ((tid & 63u) == 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            const unsigned long long t_kernel = wave_clock[wave * 2], t_dry = wave_clock[wave * 2 + 1];
            atomicAdd(&stats->path_cycles[6], t_end - t_kernel);                       // wave lifetime
            atomicAdd(&stats->path_cycles[7], t_dry ? t_end - t_dry : 0ull);           // of which after the queue ran dry
        }
    }
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

// Tile order for the next pass: tiles sorted by the cost the pass just finished measured, costliest
// first, so that the lanes' last pixels are cheap ones and the persistent grid drains evenly.  One
// block: counting sort over 2048 cost classes (order inside a class is arbitrary).
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n, int interleave) {
    __shared__ uint32_t bins[2048];
    __shared__ uint32_t max_cost;
    const uint32_t t = threadIdx.x;
    if (t == 0) max_cost = 1;
    bins[t] = 0; bins[t + 1024] = 0;
    __syncthreads();
    uint32_t m = 0;
    for (uint32_t i = t; i < n; i += 1024) m = cost[i] > m ? cost[i] : m;
    atomicMax(&max_cost, m);
    __syncthreads();
    const unsigned long long top = max_cost;
    for (uint32_t i = t; i < n; i += 1024) atomicAdd(&bins[2047u - (uint32_t)((unsigned long long)cost[i] * 2047ull / top)], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < 2048; b++) { const uint32_t c = bins[b]; bins[b] = run; run += c; }
    }
    __syncthreads();
    // Costliest first, but with the cheap half of the list interleaved from its cheap end: expensive pixels are
    // traversal-heavy and cheap ones shading-heavy, and a block's fixed split of traversal and shading waves wants a
    // steady mix of both; the tiles handed out last are then the middle of the list.
    for (uint32_t i = t; i < n; i += 1024) {
        const uint32_t p = atomicAdd(&bins[2047u - (uint32_t)((unsigned long long)cost[i] * 2047ull / top)], 1u);
        order[interleave && p > (n - 1u) / 2u ? 2u * (n - 1u - p) + 1u : (interleave ? 2u * p : p)] = i;
    }
}

// PrepareForDenoise (math/color.cu:32-51): the three accumulators scaled by 1 / passes into the float3
// images a denoiser takes (beauty, normal, albedo).
__global__ __launch_bounds__(CRT_BLOCK) void prepare_denoise_kernel(const float* __restrict__ results, const float* __restrict__ normals,
                                                                    const float* __restrict__ albedos, float* __restrict__ beauty,
                                                                    float* __restrict__ normal, float* __restrict__ albedo,
                                                                    uint32_t n3, float pass_mul) {
    const uint32_t i = blockIdx.x * CRT_BLOCK + threadIdx.x;      // one float per thread: the repack is element-wise
    if (i >= n3) return;
    beauty[i] = results[i] * pass_mul;
    normal[i] = normals[i] * pass_mul;
    albedo[i] = albedos[i] * pass_mul;
}

// ConvertColors, float overload (math/color.cu:25-30): a float3 image (a denoiser's output) to 8 bits.
__global__ __launch_bounds__(CRT_BLOCK) void convert_floats_kernel(const float* __restrict__ in, uint8_t* __restrict__ out,
                                                                   uint32_t n, int srgb) {
    const uint32_t i = blockIdx.x * CRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 c = ld3(in + 3ull * i);
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
        case 8: r = rcp_ieee(x[i]); break;
        default: r = crtm::det_tanf(x[i]); break;
    }
    out[i] = r;
}

// Every one of the 2^32 float bit patterns through rcp_ieee and through the division it stands for; the number of
// patterns on which they differ (NaN results count as equal) is added to *differences.
__global__ void rcp_exhaustive_kernel(unsigned long long* differences) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        float x = __uint_as_float((uint32_t)i);
        const float got = rcp_ieee(x);
        asm volatile("" : "+v"(x));
        const float want = 1.0f / x;
        if (!(want != want && got != got) && __float_as_uint(got) != __float_as_uint(want)) bad++;
    }
    if (bad) atomicAdd(differences, bad);
}

}  // namespace crt
