// Deterministic single-precision transcendentals.
//
// Every function takes and returns float but evaluates in IEEE-754 double with
// only +, -, *, /, sqrt, rint and bit operations.  With FP contraction disabled
// the result is the same bit pattern on x86-64 hosts and on gfx950, which is
// what lets the renderer's accumulators be compared bit for bit against a CPU
// run.  Error is well below 1 ULP of float (the double result carries ~1e-13
// relative error before the final rounding), so values agree with a correctly
// rounded libm except for rare last-bit ties.
//
// The reference calls sinf/cosf/powf/expf/atan2f/asinf/acosf/tanf from the CUDA
// math library (trace.cu:27-30, shading.cu:34-66, vector.cu:37-48,
// objects.cu:48-49, texture.cu:57-58, color.cu:9, lights.cu:65-66); those
// results are implementation-defined to a few ULP, these are the pinned
// replacements.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
// always inlined: the kernels are built without device function calls
#define CRT_HD __host__ __device__ __attribute__((always_inline)) inline
#else
#define CRT_HD inline
#endif

namespace crtm {

CRT_HD uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
CRT_HD double   u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
CRT_HD uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
CRT_HD float    u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

CRT_HD bool is_nan(float x) { return x != x; }
CRT_HD bool is_inf(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
CRT_HD float make_nan() { return u2f(0x7fc00000u); }
CRT_HD float make_inf() { return u2f(0x7f800000u); }

// 2^k for -1022 <= k <= 1023
CRT_HD double pow2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }

// e^t for finite t in [-110, 90]; relative error ~1e-15.
CRT_HD double exp_core(double t) {
    const double kd = rint(t * 1.4426950408889634);
    const int k = (int)kd;
    // ln2 split: hi has 32 significant bits so kd*hi is exact for |kd| < 2^20.
    const double r = (t - kd * 0.6931471803691238) - kd * 1.9082149292705877e-10;
    // Taylor series to r^12, |r| <= 0.3466 -> truncation < 1e-16.
    double p = 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * pow2i(k);
}

// ln(x) for finite x > 0 that is a normal double (every positive float is).
CRT_HD double log_core(double x) {
    const uint64_t b = d2u(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    // 2*atanh(s), |s| <= 0.1716 -> z^10/21 < 3e-17
    double p = 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double lm = 2.0 * s * p;
    const double ed = (double)e;
    return ed * 0.6931471803691238 + (ed * 1.9082149292705877e-10 + lm);
}

CRT_HD float det_expf(float x) {
    if (is_nan(x)) return x;
    if (x > 89.0f) return make_inf();
    if (x < -104.0f) return 0.0f;
    return (float)exp_core((double)x);
}

// C99 pow() case analysis, float in / float out.
CRT_HD float det_powf(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (is_nan(x) || is_nan(y)) return make_nan();
    // integer / odd classification of y
    bool yint = false, yodd = false;
    {
        const float ay = fabsf(y);
        if (ay >= 16777216.0f) { yint = true; yodd = false; }          // >= 2^24: even integer
        else {
            const float fl = floorf(ay);
            if (fl == ay) { yint = true; yodd = (((int)fl) & 1) != 0; }
        }
    }
    if (is_inf(y)) {
        const float ax = fabsf(x);
        if (ax == 1.0f) return 1.0f;
        return ((ax > 1.0f) == (y > 0.0f)) ? make_inf() : 0.0f;
    }
    if (x == 0.0f) {
        const bool neg = (f2u(x) >> 31) != 0;
        if (y < 0.0f) return (yodd && neg) ? -make_inf() : make_inf();
        return (yodd && neg) ? -0.0f : 0.0f;
    }
    if (is_inf(x)) {
        if (x > 0.0f) return y > 0.0f ? make_inf() : 0.0f;
        if (y > 0.0f) return yodd ? -make_inf() : make_inf();
        return yodd ? -0.0f : 0.0f;
    }
    if (yint && fabsf(y) <= 2048.0f) {
        // integer exponent: binary powering in double (at most 22 roundings, ~2e-15 relative)
        double b = (double)x, r = 1.0;
        for (unsigned n = (unsigned)fabsf(y); n; n >>= 1) {
            if (n & 1u) r = r * b;
            b = b * b;
        }
        return (float)(y < 0.0f ? 1.0 / r : r);
    }
    double sign = 1.0;
    double ax = (double)x;
    if (x < 0.0f) {
        if (!yint) return make_nan();
        if (yodd) sign = -1.0;
        ax = -ax;
    }
    double t = (double)y * log_core(ax);
    if (t > 90.0) return (float)(sign * (double)make_inf());
    if (t < -110.0) return (float)(sign * 0.0);
    return (float)(sign * exp_core(t));
}

// Reduce x to r in [-pi/4, pi/4] with x = k*pi/2 + r; returns k mod 4.
CRT_HD int rem_pio2(double x, double& r) {
    const double kd = rint(x * 0.6366197723675814);
    r = ((x - kd * 1.57079632673412561417e+00) - kd * 6.07710050630396597660e-11)
        - kd * 2.02226624879595063154e-21;
    // kd can exceed int range for huge |x|; take it modulo 4 in double first.
    const double k4 = kd - 4.0 * floor(kd * 0.25);
    return ((int)k4) & 3;
}

CRT_HD double sin_poly(double r) {
    const double z = r * r;
    double p = 1.58969099521155010221e-10;
    p = p * z + -2.50507602534068634195e-08;
    p = p * z + 2.75573137070700676789e-06;
    p = p * z + -1.98412698298579493134e-04;
    p = p * z + 8.33333333332248946124e-03;
    p = p * z + -1.66666666666666324348e-01;
    return r + r * (z * p);
}

CRT_HD double cos_poly(double r) {
    const double z = r * r;
    double p = -1.13596475577881948265e-11;
    p = p * z + 2.08757232129817482790e-09;
    p = p * z + -2.75573143513906633035e-07;
    p = p * z + 2.48015872894767294178e-05;
    p = p * z + -1.38888888888741095749e-03;
    p = p * z + 4.16666666666666019037e-02;
    return (1.0 - 0.5 * z) + (z * z) * p;
}

CRT_HD double sin_core(double x) {
    double r;
    const int n = rem_pio2(x, r);
    switch (n) {
        case 0: return sin_poly(r);
        case 1: return cos_poly(r);
        case 2: return -sin_poly(r);
        default: return -cos_poly(r);
    }
}

CRT_HD double cos_core(double x) {
    double r;
    const int n = rem_pio2(x, r);
    switch (n) {
        case 0: return cos_poly(r);
        case 1: return -sin_poly(r);
        case 2: return -cos_poly(r);
        default: return sin_poly(r);
    }
}

CRT_HD void det_sincosf(float x, float& s, float& c) {
    if (is_nan(x) || is_inf(x)) { s = make_nan(); c = make_nan(); return; }
    double r;
    const int n = rem_pio2((double)x, r);
    const double sp = sin_poly(r), cp = cos_poly(r);
    double sv, cv;
    switch (n) {
        case 0: sv = sp; cv = cp; break;
        case 1: sv = cp; cv = -sp; break;
        case 2: sv = -sp; cv = -cp; break;
        default: sv = -cp; cv = sp; break;
    }
    s = x == 0.0f ? x : (float)sv;
    c = (float)cv;
}

CRT_HD float det_sinf(float x) {
    if (is_nan(x) || is_inf(x)) return make_nan();
    if (x == 0.0f) return x;
    return (float)sin_core((double)x);
}

CRT_HD float det_cosf(float x) {
    if (is_nan(x) || is_inf(x)) return make_nan();
    return (float)cos_core((double)x);
}

CRT_HD float det_tanf(float x) {
    if (is_nan(x) || is_inf(x)) return make_nan();
    if (x == 0.0f) return x;
    return (float)(sin_core((double)x) / cos_core((double)x));
}

// atan(t) for 0 <= t <= 1.
CRT_HD double atan01(double t) {
    double c, base;
    if (t < 0.125)      { c = 0.0;  base = 0.0; }
    else if (t < 0.375) { c = 0.25; base = 0.24497866312686414; }
    else if (t < 0.625) { c = 0.5;  base = 0.4636476090008061; }
    else if (t < 0.875) { c = 0.75; base = 0.6435011087932844; }
    else                { c = 1.0;  base = 0.7853981633974483; }
    const double u = (t - c) / (1.0 + t * c);     // |u| <= 0.125
    const double z = u * u;
    double p = -1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + -1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + -1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + -1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + -1.0 / 3.0;
    p = p * z + 1.0;
    return base + u * p;
}

// atan2 in double for finite-or-infinite, non-NaN arguments.
CRT_HD double atan2_core(double y, double x) {
    const double PI = 3.141592653589793;
    const double PIO2 = 1.5707963267948966;
    const bool xneg = (d2u(x) >> 63) != 0;
    const bool yneg = (d2u(y) >> 63) != 0;
    const double ax = xneg ? -x : x;
    const double ay = yneg ? -y : y;
    double a;                                   // angle of (ax, ay) in [0, pi/2]
    const bool xinf = ax > 1.7976931348623157e308;
    const bool yinf = ay > 1.7976931348623157e308;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (xinf && yinf) a = 0.7853981633974483;
    else if (xinf) a = 0.0;
    else if (yinf) a = PIO2;
    else if (ay <= ax) a = atan01(ay / ax);
    else a = PIO2 - atan01(ax / ay);
    if (xneg) a = PI - a;
    return yneg ? -a : a;
}

CRT_HD float det_atan2f(float y, float x) {
    if (is_nan(x) || is_nan(y)) return make_nan();
    return (float)atan2_core((double)y, (double)x);
}

CRT_HD float det_asinf(float x) {
    if (is_nan(x) || x > 1.0f || x < -1.0f) return make_nan();
    if (x == 0.0f) return x;
    const double d = (double)x;
    return (float)atan2_core(d, sqrt((1.0 - d) * (1.0 + d)));
}

CRT_HD float det_acosf(float x) {
    if (is_nan(x) || x > 1.0f || x < -1.0f) return make_nan();
    const double d = (double)x;
    return (float)atan2_core(sqrt((1.0 - d) * (1.0 + d)), d);
}

// float -> int with the device conversion rule the reference runs under
// (round toward zero, saturate, NaN -> 0).
CRT_HD int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}

}  // namespace crtm
