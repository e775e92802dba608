// bhray_math.h — binary32 vector helpers shared by the HIP kernels and the host code that
// derives per-frame constants.  Implements the numerics contract of DESIGN.md §Numerics: every
// WGSL operator of /root/reference/src/renderer/shaders/ray.wgsl is one IEEE binary32 operation
// in source order (the translation unit is compiled with -ffp-contract=off), with
//   N1 dot = (x*x + y*y) + z*z, N2 vector/scalar = vector * (1/scalar), N3 small integer powers
//   by multiplication, N5 mix(a,b,t) = a*(1-t) + b*t and compare-select min/max.
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

}  // namespace bhray
