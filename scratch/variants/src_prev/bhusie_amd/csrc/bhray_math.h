// bhray_math.h — binary32 vector helpers shared by the HIP kernels and the host code that
// derives per-frame constants.  Implements the numerics contract of DESIGN.md §Numerics: every
// WGSL operator of /root/reference/src/renderer/shaders/ray.wgsl is one IEEE binary32 operation
// in source order (the translation unit is compiled with -ffp-contract=off), with
//   N1 dot = (x*x + y*y) + z*z, N2 vector/scalar = vector * (1/scalar), N3 small integer powers
//   by multiplication, N5 mix(a,b,t) = a*(1-t) + b*t and compare-select min/max, and N7: the integrator
//   alone uses explicit fused multiply-adds (fdot, fcross, fmadd3) — the only FMAs in the product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BH_HD __host__ __device__ __forceinline__

namespace bhray {

struct F3 { float x, y, z; };

BH_HD F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
BH_HD F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
BH_HD F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
BH_HD F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
BH_HD F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
BH_HD F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
BH_HD float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BH_HD float length(F3 a) { return sqrtf(dot(a, a)); }
BH_HD F3 div_s(F3 a, float s) { float r = 1.0f / s; return a * r; }
BH_HD F3 normalize(F3 a) { return div_s(a, length(a)); }
BH_HD float distance(F3 a, F3 b) { return length(a - b); }
BH_HD F3 cross(F3 a, F3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
BH_HD float min_(float a, float b) { return b < a ? b : a; }
BH_HD float max_(float a, float b) { return a < b ? b : a; }
BH_HD float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
BH_HD float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }
BH_HD F3 mix3(F3 a, F3 b, float t) { return f3(mix_(a.x, b.x, t), mix_(a.y, b.y, t), mix_(a.z, b.z, t)); }
BH_HD F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
// N7: fused forms, used ONLY by the integrator (f, next_ray_euler, next_ray_rk, exit distance)
BH_HD float fdot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
BH_HD float flength(F3 a) { return sqrtf(fdot(a, a)); }
BH_HD F3 fnormalize(F3 a) { return div_s(a, flength(a)); }
BH_HD float fdistance(F3 a, F3 b) { return flength(a - b); }
BH_HD F3 fcross(F3 a, F3 b) { return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
BH_HD F3 fmadd3(F3 w, float s, F3 v) { return f3(fmaf(w.x, s, v.x), fmaf(w.y, s, v.y), fmaf(w.z, s, v.z)); }   // v + w*s
BH_HD F3 lin2(F3 a, float ca, F3 b, float cb) { return fmadd3(b, cb, a * ca); }                                 // a*ca + b*cb
BH_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
BH_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ------------------------------------------------------------------------------------------
// portable transcendental forms (numerics contract N4) — same operation sequence as the oracle
// ------------------------------------------------------------------------------------------
BH_HD float bh_pow_m001(float x) {          // x^(-0.001), ray.wgsl:459
    if (!(x == x) || x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0x7f800000u);
    if (x == u2f(0x7f800000u)) return 0.0f;
    uint32_t u = f2u(x);
    int e = (int)(u >> 23) - 127;
    if ((u >> 23) == 0) { x = x * 8388608.0f; u = f2u(x); e = (int)(u >> 23) - 127 - 23; }
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.111111112f;
    p = p * s2 + 0.142857149f;
    p = p * s2 + 0.2f;
    p = p * s2 + 0.333333343f;
    p = p * s2 + 1.0f;
    float lnm = (2.0f * s) * p;
    float lnx = (float)e * 0.693147182f + lnm;
    float t = -0.001f * lnx;
    float q = 0.00138888892f;
    q = q * t + 0.00833333377f;
    q = q * t + 0.0416666679f;
    q = q * t + 0.166666672f;
    q = q * t + 0.5f;
    q = q * t + 1.0f;
    q = q * t + 1.0f;
    return q;
}

BH_HD float bh_asin_kernel(float z) {
    float z2 = z * z;
    float p = 4.2163199048e-2f;
    p = p * z2 + 2.4181311049e-2f;
    p = p * z2 + 4.5470025998e-2f;
    p = p * z2 + 7.4953002686e-2f;
    p = p * z2 + 1.6666752422e-1f;
    return z + (z * z2) * p;
}
BH_HD float bh_acos(float x) {              // ray.wgsl:266
    if (!(x == x) || x > 1.0f || x < -1.0f) return u2f(0x7fc00000u);
    if (x > 0.5f) { float z = sqrtf((1.0f - x) * 0.5f); return 2.0f * bh_asin_kernel(z); }
    if (x < -0.5f) { float z = sqrtf((1.0f + x) * 0.5f); return 3.14159274f - 2.0f * bh_asin_kernel(z); }
    return 1.57079637f - bh_asin_kernel(x);
}


BH_HD float bh_atan2(float y, float x) {      // ray.wgsl:257-258, 632
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax < ay ? ay : ax, mn = ax < ay ? ax : ay;
    float a = mx == 0.0f ? 0.0f : mn / mx;
    float t = a, base = 0.0f;
    if (a > 0.414213568f) { t = (a - 1.0f) / (a + 1.0f); base = 0.785398185f; }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = base + ((p * z) * t + t);
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return (f2u(y) >> 31) ? -r : r;
}

template <int KIND>   // 0 sin, 1 cos (ray.wgsl:634)
BH_HD float bh_sincos(float xin) {
    float x = fabsf(xin);
    bool sign = (KIND == 0) ? ((f2u(xin) >> 31) != 0u) : false;
    if (!(x <= 3.0e9f)) return u2f(0x7fc00000u);
    uint32_t j = (uint32_t)(x * 1.27323954f);
    j = j + (j & 1u);
    float y = (float)j;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    j = j & 7u;
    if (j > 3u) { sign = !sign; j = j - 4u; }
    if (KIND == 1 && j > 1u) sign = !sign;
    float z = x * x;
    const bool mid = (j == 1u || j == 2u);
    const bool use_cos = (KIND == 0) ? mid : !mid;
    float r;
    if (use_cos) {
        float p = 2.443315711809948e-5f;
        p = p * z - 1.388731625493765e-3f;
        p = p * z + 4.166664568298827e-2f;
        r = ((p * z) * z - 0.5f * z) + 1.0f;
    } else {
        float p = -1.9515295891e-4f;
        p = p * z + 8.3321608736e-3f;
        p = p * z - 1.6666654611e-1f;
        r = (p * z) * x + x;
    }
    return sign ? -r : r;
}


BH_HD float bh_tan(float x) { return bh_sincos<0>(x) / bh_sincos<1>(x); }     // ray.wgsl:279 (N4)

}  // namespace bhray
