// float3 / 3x4 matrix arithmetic with the operation order fixed.
//
// Every expression is written so that the sequence of roundings equals the
// reference's (math/float3.cuh:25-194, math/matrix.cuh:72-97): sums associate
// left to right, nothing is fused, and normalisation is "1/sqrt then multiply".
// Compile with -ffp-contract=off.
#pragma once
#include "crt_detmath.h"

namespace crt {

#define CRT_BIGFLOAT 3.402823466e+38f

struct f3 {
    float x, y, z;
};

CRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
CRT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
CRT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
CRT_HD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
CRT_HD void operator+=(f3& a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
CRT_HD void operator*=(f3& a, f3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
CRT_HD void operator*=(f3& a, float s) { a.x *= s; a.y *= s; a.z *= s; }

CRT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// The reference builds the cross product from three scaled basis vectors
// (float3.cuh:154-156).  For finite operands that differs from the direct form
// below only in the sign of exact zeros, which no comparison on the path can
// observe; non-finite operands already mean a NaN pixel.
CRT_HD f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - b.y * a.z, -(a.x * b.z - b.x * a.z), a.x * b.y - b.x * a.y);
}

// Literal three-basis-vector form, used by host code that derives camera and
// transform data so those arrays equal the reference's value for value.
CRT_HD f3 cross_literal(f3 a, f3 b) {
    const f3 r = mk3(1.0f, 0.0f, 0.0f), u = mk3(0.0f, 1.0f, 0.0f), f = mk3(0.0f, 0.0f, 1.0f);
    return r * (a.y * b.z - b.y * a.z) - u * (a.x * b.z - b.x * a.z) + f * (a.x * b.y - b.x * a.y);
}

// 1.0f / x, correctly rounded as IEEE division gives it.  On the GPU: the hardware estimate and one Newton step in
// fused arithmetic, which equals the division whenever the result is a normal number; a result that is not (x zero,
// denormal, infinite, NaN or beyond 2^126) is replaced by the division itself.  Equality holds for every one of the
// 2^32 inputs (crt_math_selftest, function 100, compares them all on the device).
#if defined(__HIP_DEVICE_COMPILE__)
CRT_HD float rcp_fast(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
CRT_HD bool rcp_fast_ok(float r) { return __builtin_amdgcn_class(r, 0x108); }      // 0x108: positive or negative normal
#else
CRT_HD float rcp_fast(float x) { return 1.0f / x; }      // (host pass: the division is the definition)
CRT_HD bool rcp_fast_ok(float) { return true; }
#endif
CRT_HD float rcp_ieee(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = rcp_fast(x);
    if (!rcp_fast_ok(r)) {
        asm volatile("" : "+v"(x));      // keeps the division out of the common path
        r = 1.0f / x;
    }
    return r;
#else
    return 1.0f / x;
#endif
}

CRT_HD float lengthsq(f3 a) { return dot(a, a); }
CRT_HD float length(f3 a) { return sqrtf(dot(a, a)); }
CRT_HD f3 normalized(f3 a) {
    const float s = rcp_ieee(sqrtf(dot(a, a)));
    return s * a;
}

CRT_HD float comp(const f3& a, unsigned i) { return i == 1 ? a.y : (i == 2 ? a.z : a.x); }

// 3x4 affine matrix, column major, implicit last row [0 0 0 1] (matrix.cuh:10-20).
struct mat34 {
    float c[12];
};

CRT_HD mat34 mat_identity() {
    mat34 m;
    for (int i = 0; i < 12; i++) m.c[i] = 0.0f;
    m.c[0] = 1.0f; m.c[4] = 1.0f; m.c[8] = 1.0f;
    return m;
}

CRT_HD f3 xform_point(const float* c, f3 v) {
    return mk3(c[0] * v.x + c[3] * v.y + c[6] * v.z + c[9],
               c[1] * v.x + c[4] * v.y + c[7] * v.z + c[10],
               c[2] * v.x + c[5] * v.y + c[8] * v.z + c[11]);
}

CRT_HD f3 xform_dir(const float* c, f3 v) {
    return mk3(c[0] * v.x + c[3] * v.y + c[6] * v.z,
               c[1] * v.x + c[4] * v.y + c[7] * v.z,
               c[2] * v.x + c[5] * v.y + c[8] * v.z);
}

// transpose multiply then normalise (matrix.cuh:92-97)
CRT_HD f3 xform_normal(const float* c, f3 v) {
    return normalized(mk3(c[0] * v.x + c[1] * v.y + c[2] * v.z,
                          c[3] * v.x + c[4] * v.y + c[5] * v.z,
                          c[6] * v.x + c[7] * v.y + c[8] * v.z));
}

}  // namespace crt
