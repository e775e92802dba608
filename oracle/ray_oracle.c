/*
 * ray_oracle.c — CPU restatement of the reference ray pass.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows, function by function, /root/reference/src/renderer/shaders/ray.wgsl:1-847 (each
 * function below cites the WGSL lines it restates), with the texture semantics of
 * src/renderer/texture.rs:16-69 (RGBA8 unorm, bilinear, clamp-to-edge, one mip, no sRGB) and
 * the dispatch of src/renderer/pipelines/ray_pipeline.rs:301-309 (one invocation per pixel).
 *
 * PINNED TO THE REFERENCE'S SHADER TEXT, not to a driver run: the reference has no tests, golden vectors or benchmarks
 *  and cannot be built or run here (Rust + WGSL, no toolchain, missing assets: F1/F2), but its shader text is
 * EXECUTED by oracle/wgsl_exec.py (an interpreter; nothing restated) and this file's literal mode reproduces those frames
 * word for word (tests/golden/wgsl_exec.npz, tests/test_wgsl_pin.py).  Also pinned by (1) an independently written NumPy
 * restatement (oracle/np_ray.py) that must agree with it, (2) analytic known-answer tests (tests/test_oracle_kat.py),
 * (3) the committed fixtures under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libbhray) never links, loads or calls it.
 *
 * Numerics contract: every WGSL arithmetic operator is ONE IEEE-754
 * binary32 operation, evaluated in source order with WGSL precedence; no FMA contraction
 * (compile with -ffp-contract=off); and
 *   N1  dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z ; length(v) = sqrt(dot(v,v))
 *   N2  vector / scalar  = vector * (1.0f / scalar)   (one correctly rounded reciprocal)
 *       normalize(v)     = v / length(v) under N2
 *   N3  pow(x, 2.0) = x*x ; pow(x, 4.0) = (x*x)*(x*x) ; pow(x, 5.0) = ((x*x)*(x*x))*x ;
 *       pow(length(v), 2.0) = dot(v, v) (ray.wgsl:419,470: the squared length without the round trip through sqrt)
 *   N4  pow(e_max, -0.001), acos, atan2, sin, cos and tan (= sin/cos) use the portable forms bh_*
 *       below, specified to the bit: they steer the trajectory (pow), the copy/interpolate/
 *       trace classification (acos) or texture coordinates, where an ulp is amplified by the
 *       texture gradient (atan2, sin, cos).  Only pow(.,1.3) (optical depth, ray.wgsl:623)
 *       uses libm; its 1-2 ulp spread is not amplified.
 *   N5  mix(a,b,t) = a*(1-t) + b*t ; clamp(x,lo,hi) = min(max(x,lo),hi) with
 *       max(a,b) = a<b ? b : a and min(a,b) = b<a ? b : a ; smoothstep per WGSL spec.
 *   N6  untyped WGSL `const` expressions (the Cash–Karp tableau, b_i - b*_i) are evaluated
 *       in binary64 and rounded once to binary32 (naga AbstractFloat const-evaluation).
 *   N7  the integrator — f, next_ray_euler, next_ray_rk and the exit distance (ray.wgsl:401-480, 533) — is evaluated
 *       with fused multiply-add, as WGSL permits and GPU shader compilers do (a*b+c contracts to one rounding):
 *         fdot(a,b)    = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *         fcross(a,b)  = (fma(a.y,b.z, -(a.z*b.y)), fma(a.z,b.x, -(a.x*b.z)), fma(a.x,b.y, -(a.y*b.x)))
 *         v + w*s      = fma(w, s, v) per component            (stage arguments, direction and position updates)
 *         sum k_i*c_i  = fma(k_n,c_n, ... fma(k_2,c_2, k_1*c_1))  (left to right; first product rounded)
 *       length / normalize / distance inside the integrator are built on fdot.  Everything else (intersections,
 *       shading, grid classification, create_ray) keeps N0: no contraction.
 *   N9  reassociation in the integrator (WGSL: "an implementation may reassociate operations"): the scalar factors of f are
 *       combined once per step, s = (-1.5*h2) * (1/dist^5), positions are taken relative to the hole once per step,
 *       q0 = p0 - bh, and a stage evaluates f(p0 + h*sum) as fma(sum, h, q0) * s  (6 instead of 12 operations per stage).
 *   N10 the step size is distributed into the RK stages: K_i = h*k_i = (q0 + sum_j a_ij K_j) * (s*h), so that
 *       e = sum (b_i - b*_i) K_i and direction + (sum b*_i K_i) need no further multiplication by h; terms whose tableau
 *       coefficient is exactly zero (b_2 = b*_2 = 0) are dropped (WGSL lets an implementation assume no NaN/inf, so
 *       0*k + x = x).  Euler: direction + q0 * (s*step).
 *   oracle_set_literal(1) switches the integrator to the operator-by-operator reading (N0-N2 only) so that the distance
 *   between the contract and the literal evaluation can be measured (tests/test_oracle_kat.py).
 * Deviations from the reference, both unobservable in it:
 *   D1  the RK retry loop (ray.wgsl:425-451) cannot change h, so it never terminates when
 *       e_max > 1 (or NaN); it is executed exactly once here.
 *   D2  the BVH stack holds node indices (BVH_STACK deep) rather than 19 whole nodes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BVH_STACK 64

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

/* ---- byte layouts (must equal include/bhray.h; restated here so the oracle is standalone) */
typedef struct { int32_t material_count, model_count; float time; int32_t integration_method;
                 float step_size; int32_t max_iterations; float angle_division_threshold;
                 int32_t highlight_interpolation; } o_details;
typedef struct { float position[3]; uint32_t pad; float forward[3]; float fov; } o_camera;
typedef struct { float inner_radius, outer_radius, rotation_speed, relativity_radius;
                 float position[3]; int32_t show_disk_texture; float normal[3];
                 int32_t show_red_shift; float rotation_matrix[12]; float feather_amount;
                 int32_t pad[8]; } o_black_hole;
typedef struct { float min_corner[3]; int32_t left_child; float max_corner[3];
                 int32_t obj_count; } o_node;
typedef struct { int32_t p1, p2, p3, n1, n2, n3; } o_tri;
typedef struct { float position[3]; int32_t visible;
                 const float* points; const float* normals; const o_tri* triangles;
                 const o_node* nodes; const int32_t* bvh_lookup;
                 int32_t point_count, normal_count, triangle_count, node_count; } o_model;
typedef struct { const uint8_t* rgba; int32_t w, h; } o_tex;

typedef struct {
    uint64_t pixels, copied, interpolated, traced, steps, flat_iters, node_pairs, triangles,
             disk_hits, sky_samples;
} o_counters;

typedef struct {
    const o_camera* camera; const o_details* details; const o_black_hole* bh;
    const o_model* models;           /* details->model_count entries */
    o_tex t_temp, t_disk, t_sky;
    o_counters* cnt;                 /* per-thread, may be NULL */
    float* aux;                      /* per-ray diagnostics of trace_ray (oracle_render_aux), may be NULL */
} scene;

/* ---- vector helpers under the numerics contract ---------------------------------------- */
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   /* N1 */
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 divs(v3 a, float s) { float r = 1.0f / s; return muls(a, r); }       /* N2 */
static inline v3 normalize(v3 a) { return divs(a, length(a)); }
static inline float distance(v3 a, v3 b) { return length(sub(a, b)); }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float fmin_(float a, float b) { return b < a ? b : a; }                 /* N5 */
static inline float fmax_(float a, float b) { return a < b ? b : a; }
static inline float clampf(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline v3 mix3(v3 a, v3 b, float t) { return V(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
static inline v3 fromp(const float* p) { return V(p[0], p[1], p[2]); }
/* N7: fused forms used only by the integrator */
static inline float fdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float flength(v3 a) { return sqrtf(fdot(a, a)); }
static inline v3 fnormalize(v3 a) { return divs(a, flength(a)); }
static inline float fdistance(v3 a, v3 b) { return flength(sub(a, b)); }
static inline v3 fcross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 fmadd3(v3 w, float s, v3 v) { return V(fmaf(w.x, s, v.x), fmaf(w.y, s, v.y), fmaf(w.z, s, v.z)); }   /* v + w*s */
static inline float smoothstep(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* ---- N4: portable transcendental forms -------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* x^(-0.001) for x > 0 (ray.wgsl:459).  ln x = e*ln2 + 2*atanh(s), s=(m-1)/(m+1),
 * m in [sqrt(.5),sqrt(2)); then exp(t), t = -0.001*ln x, by a degree-6 Taylor polynomial
 * (|t| <= 0.09 over all finite binary32).  powf special cases: 0 -> +inf, inf -> 0, NaN/neg -> NaN. */
float bh_pow_m001(float x) {
    if (!(x == x) || x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0x7f800000u);
    if (x == u2f(0x7f800000u)) return 0.0f;
    uint32_t u = f2u(x);
    int32_t e = (int32_t)(u >> 23) - 127;
    if ((u >> 23) == 0) {                    /* subnormal: scale by 2^23 exactly */
        x = x * 8388608.0f; u = f2u(x); e = (int32_t)(u >> 23) - 127 - 23;
    }
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);     /* [1,2) */
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }   /* [0.7071,1.4142) */
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.111111112f;                  /* 1/9 */
    p = p * s2 + 0.142857149f;               /* 1/7 */
    p = p * s2 + 0.2f;
    p = p * s2 + 0.333333343f;
    p = p * s2 + 1.0f;
    float lnm = (2.0f * s) * p;
    float lnx = (float)e * 0.693147182f + lnm;
    float t = -0.001f * lnx;
    float q = 0.00138888892f;                /* 1/720 */
    q = q * t + 0.00833333377f;              /* 1/120 */
    q = q * t + 0.0416666679f;               /* 1/24 */
    q = q * t + 0.166666672f;                /* 1/6 */
    q = q * t + 0.5f;
    q = q * t + 1.0f;
    q = q * t + 1.0f;
    return q;
}

/* acos(x) (ray.wgsl:266): |x|<=0.5: pi/2 - asin(x); x>0.5: 2*asin(sqrt((1-x)/2));
 * x<-0.5: pi - 2*asin(sqrt((1+x)/2)); asin(z) = z + z*z2*P(z2), P = degree-5 minimax-style
 * polynomial (coefficients of the classic single-precision asin kernel).  |x|>1 or NaN -> NaN. */
static inline float bh_asin_kernel(float z) {
    float z2 = z * z;
    float p = 4.2163199048e-2f;
    p = p * z2 + 2.4181311049e-2f;
    p = p * z2 + 4.5470025998e-2f;
    p = p * z2 + 7.4953002686e-2f;
    p = p * z2 + 1.6666752422e-1f;
    return z + (z * z2) * p;
}
float bh_acos(float x) {
    if (!(x == x) || x > 1.0f || x < -1.0f) return u2f(0x7fc00000u);
    if (x > 0.5f) {
        float z = sqrtf((1.0f - x) * 0.5f);
        return 2.0f * bh_asin_kernel(z);
    }
    if (x < -0.5f) {
        float z = sqrtf((1.0f + x) * 0.5f);
        return 3.14159274f - 2.0f * bh_asin_kernel(z);
    }
    return 1.57079637f - bh_asin_kernel(x);
}


/* atan2(y,x) (ray.wgsl:257-258, 632): a = min/max of |x|,|y|; atan(a) on [0,1] with one reduction at
 * tan(pi/8) and the classic degree-4 odd single-precision kernel; then octant / sign fix-ups. */
float bh_atan2(float y, float x) {
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

/* sin / cos (ray.wgsl:634): Cody–Waite reduction by pi/4 octants (three-part pi/4, exact products for
 * |x| < 8192) and the classic single-precision kernels on [-pi/4, pi/4].  kind 0 = sin, 1 = cos. */
static inline float bh_sincos(float xin, int kind) {
    float x = fabsf(xin);
    int sign = (kind == 0) ? (int)(f2u(xin) >> 31) : 0;
    if (!(x <= 3.0e9f)) return u2f(0x7fc00000u);         /* inf / NaN (and beyond int range) */
    uint32_t j = (uint32_t)(x * 1.27323954f);            /* x * 4/pi, truncated */
    j = j + (j & 1u);
    float y = (float)j;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    j = j & 7u;
    if (j > 3u) { sign = !sign; j = j - 4u; }
    if (kind == 1 && j > 1u) sign = !sign;
    float z = x * x;
    int use_cos = (kind == 0) ? (j == 1u || j == 2u) : !(j == 1u || j == 2u);
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
float bh_sin(float x) { return bh_sincos(x, 0); }
float bh_cos(float x) { return bh_sincos(x, 1); }
float bh_tan(float x) { return bh_sincos(x, 0) / bh_sincos(x, 1); }   /* ray.wgsl:279 */

/* ---- textures: texture.rs:16-69 + textureSampleLevel(.., 0.0) --------------------------- */
static inline v4 texel(const o_tex* t, int x, int y) {
    const uint8_t* p = t->rgba + 4 * ((size_t)y * (size_t)t->w + (size_t)x);
    v4 r = { (float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f };
    return r;
}
static inline float unit_coord(float u, int n, int* i0, int* i1) {
    float x = u * (float)n - 0.5f;
    if (!(x >= -1.0f)) x = -1.0f;            /* also catches NaN */
    if (x > (float)n) x = (float)n;
    float fl = floorf(x);
    int a = (int)fl, b = a + 1;
    if (a < 0) a = 0; if (a > n - 1) a = n - 1;
    if (b < 0) b = 0; if (b > n - 1) b = n - 1;
    *i0 = a; *i1 = b;
    return x - fl;
}
static v4 sample_bilinear(const o_tex* t, float u, float v) {
    int x0, x1, y0, y1;
    float fx = unit_coord(u, t->w, &x0, &x1);
    float fy = unit_coord(v, t->h, &y0, &y1);
    v4 a = texel(t, x0, y0), b = texel(t, x1, y0), c = texel(t, x0, y1), d = texel(t, x1, y1);
    v4 r;
    r.x = mixf(mixf(a.x, b.x, fx), mixf(c.x, d.x, fx), fy);
    r.y = mixf(mixf(a.y, b.y, fx), mixf(c.y, d.y, fx), fy);
    r.z = mixf(mixf(a.z, b.z, fx), mixf(c.z, d.z, fx), fy);
    r.w = mixf(mixf(a.w, b.w, fx), mixf(c.w, d.w, fx), fy);
    return r;
}

/* ---- records: ray.wgsl:36-39, 92-98, 395-399 --------------------------------------------- */
typedef struct { v3 position, direction; } Ray;
typedef struct { v3 color; float opacity; float t; v3 normal; int hit; } RenderState;
typedef struct { float h, e_max; Ray ray; } RKState;

static inline RenderState rs_zero(void) { RenderState r; memset(&r, 0, sizeof r); return r; }

#define PI_F 3.1415926f                        /* ray.wgsl:131 */

/* Cash–Karp tableau, ray.wgsl:133-165 (N6) */
#define K(x) ((float)(x))
static const float a_21 = K(1.0 / 5.0);
static const float a_31 = K(3.0 / 40.0), a_32 = K(9.0 / 40.0);
static const float a_41 = K(3.0 / 10.0), a_42 = K(-9.0 / 10.0), a_43 = K(6.0 / 5.0);
static const float a_51 = K(-11.0 / 54.0), a_52 = K(5.0 / 2.0), a_53 = K(-70.0 / 27.0), a_54 = K(35.0 / 27.0);
static const float a_61 = K(1631.0 / 55296.0), a_62 = K(175.0 / 512.0), a_63 = K(575.0 / 13824.0),
                   a_64 = K(44275.0 / 110592.0), a_65 = K(253.0 / 4096.0);
static const float b_a_1 = K(2825.0 / 27648.0), b_a_2 = K(0.0), b_a_3 = K(18575.0 / 48384.0),
                   b_a_4 = K(13525.0 / 55296.0), b_a_5 = K(277.0 / 14336.0), b_a_6 = K(1.0 / 4.0);
static const float db_1 = K(37.0 / 378.0 - 2825.0 / 27648.0), db_2 = K(0.0 - 0.0),
                   db_3 = K(250.0 / 621.0 - 18575.0 / 48384.0), db_4 = K(125.0 / 594.0 - 13525.0 / 55296.0),
                   db_5 = K(0.0 - 277.0 / 14336.0), db_6 = K(512.0 / 1771.0 - 1.0 / 4.0);

/* ---- ray.wgsl:245-267 helpers ------------------------------------------------------------ */
static v3 cartesian_to_spherical(v3 c) {                       /* ray.wgsl:255-261 */
    float rho = length(c);
    float theta = bh_atan2(sqrtf(c.x * c.x + c.y * c.y), c.z);
    float phi = bh_atan2(c.y, c.x);
    return V(rho, theta, phi);
}
static float angle_between(v3 v1, v3 v2) {                     /* ray.wgsl:263-267 */
    float d = dot(v1, v2);
    float c = d / (length(v1) * length(v2));
    return bh_acos(c);
}

/* ---- ray.wgsl:269-285 create_ray ---------------------------------------------------------- */
static Ray create_ray(const scene* S, int px, int py, int sw, int sh) {
    int sm = (sw - 1) < (sh - 1) ? (sw - 1) : (sh - 1);
    float increment = 1.0f / (float)sm;
    float posx = (2.0f * ((float)px - (float)(sw - 1) * 0.5f)) * increment;
    float posy = (2.0f * ((float)py - (float)(sh - 1) * 0.5f)) * increment;
    v3 fwd = fromp(S->camera->forward);
    v3 plane_up = V(0.0f, -1.0f, 0.0f);
    v3 right = normalize(cross(fwd, plane_up));
    v3 up = normalize(cross(fwd, right));
    float fov_factor = 1.0f / bh_tan(S->camera->fov / 2.0f);
    v3 d = normalize(add(add(muls(right, posx), muls(up, posy)), muls(fwd, fov_factor)));
    Ray r = { fromp(S->camera->position), d };
    return r;
}

/* ---- ray.wgsl:725-766 hit_sphere ----------------------------------------------------------- */
static RenderState hit_sphere(Ray ray, float radius, v3 center, v3 color, float t_min, float t_max) {
    RenderState rs = rs_zero();
    rs.hit = 0; rs.t = t_max;
    v3 oc = sub(ray.position, center);
    float a = dot(ray.direction, ray.direction);
    float b = 2.0f * dot(oc, ray.direction);
    float c = dot(oc, oc) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float sq = sqrtf(disc);
        float t1 = (-b - sq) / (2.0f * a);
        float t2 = (-b + sq) / (2.0f * a);
        float t_closest = t_max;
        if (t1 > t_min && t1 < t_max) t_closest = t1;
        if (t2 > t_min && t2 < t_max && t2 < t_closest) t_closest = t2;
        if (t_closest < t_max && t_closest > t_min) {
            v3 ip = add(ray.position, muls(ray.direction, t_closest));
            rs.color = color; rs.opacity = 1.0f; rs.t = t_closest;
            rs.normal = normalize(sub(ip, center)); rs.hit = 1;
            return rs;
        }
    }
    return rs;
}

/* ---- ray.wgsl:668-701 hit_torus2d ---------------------------------------------------------- */
static RenderState hit_torus2d(Ray ray, float inner, float outer, v3 tpos, v3 normal, float t_min, float t_max) {
    float denom = dot(normal, ray.direction);
    RenderState rs = rs_zero();
    rs.hit = 0; rs.t = t_max;
    v3 dist = sub(tpos, ray.position);
    float t = dot(dist, normal) / denom;
    if (t < t_max && t > t_min) {
        rs.normal = denom < 0.0f ? neg(normal) : normal;
        v3 ip = add(ray.position, muls(ray.direction, t));
        float dc = distance(tpos, ip);
        if (dc >= inner && dc <= outer) {
            rs.color = V(1.0f, 1.0f, 1.0f); rs.opacity = 1.0f; rs.t = t; rs.hit = 1;
            return rs;
        }
    }
    return rs;
}

/* ---- ray.wgsl:598-666 hit_black_hole ------------------------------------------------------- */
static RenderState hit_black_hole(const scene* S, Ray ray, float t_min, float t_max, float total_distance) {
    const o_black_hole* bh = S->bh;
    v3 bpos = fromp(bh->position), bnormal = fromp(bh->normal);
    float inner = bh->inner_radius, outer = bh->outer_radius;

    RenderState rs = hit_sphere(ray, 1.0f, bpos, V(0, 0, 0), t_min, t_max);
    RenderState dh = hit_torus2d(ray, inner, outer, bpos, bnormal, t_min, t_max);

    if (dh.hit && dh.t < rs.t) {
        rs = dh;
        if (S->cnt) S->cnt->disk_hits++;
        v3 ip = add(ray.position, muls(ray.direction, rs.t));
        float dist = distance(bpos, ip);
        /* disk_displacement (ray.wgsl:618) is dead */
        float density = 1.0f - length(divs(ip, outer));
        density *= smoothstep(inner, inner + 1.0f, dist);
        density *= 1.0f / sqrtf(dist);                                   /* inverseSqrt */
        float od = powf(30.0f * density, 1.3f);
        rs.opacity = clampf(od * 0.2f, 0.0f, 1.0f);
        rs.color = V(od, od, od);

        if (bh->show_disk_texture != 0) {
            float r = (dist - inner) / (outer - inner);
            v3 rel = divs(sub(ip, bpos), outer);
            const float* M = bh->rotation_matrix;
            v3 c0 = V(M[0], M[1], M[2]), c1 = V(M[4], M[5], M[6]), c2 = V(M[8], M[9], M[10]);
            v3 rot = add(add(muls(c0, rel.x), muls(c1, rel.y)), muls(c2, rel.z));
            float angle = -bh_atan2(rot.z, rot.x);
            float ph = angle + S->details->time * bh->rotation_speed;
            float u = bh_sin(ph) * r, v = bh_cos(ph) * r;
            u = (u + 1.0f) * 0.5f; v = (v + 1.0f) * 0.5f;              /* (uv+1)/2 under N2 */
            v4 dc = sample_bilinear(&S->t_disk, u, v);
            rs.opacity *= clampf(0.7f + dc.w * 0.5f, 0.0f, 1.0f);
            rs.color = mulv(rs.color, muls(V(dc.x, dc.y, dc.z), dc.w));
        }

        if (bh->show_red_shift != 0) {
            float temp_max = 100000.0f, temp_min = 10000.0f, temp = 15000.0f;
            float y = 1.0f - (temp - temp_min) / (temp_max - temp_min);
            v3 sv = muls(cross(normalize(ip), normalize(V(0.0f, -1.0f, 0.0f))), 0.6f);
            float velocity = dot(ray.direction, sv);
            float doppler = sqrtf((1.0f - velocity) / (1.0f + velocity));
            float grav = sqrtf((1.0f - 2.0f / dist) / (1.0f - 2.0f / total_distance));
            float sh = clampf(grav * doppler, 0.0f, 1.0f);
            float shift = sh * sh;
            v4 sc = sample_bilinear(&S->t_temp, shift, y);
            rs.color = mulv(rs.color, V(sc.x, sc.y, sc.z));
        }
    }
    return rs;
}

/* ---- ray.wgsl:703-723 hit_aabb -------------------------------------------------------------- */
static float hit_aabb(Ray ray, const o_node* n, v3 offset) {
    v3 inv = V(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    v3 mn = add(fromp(n->min_corner), offset), mx = add(fromp(n->max_corner), offset);
    v3 t1 = mulv(sub(mn, ray.position), inv), t2 = mulv(sub(mx, ray.position), inv);
    v3 tmn = V(fmin_(t1.x, t2.x), fmin_(t1.y, t2.y), fmin_(t1.z, t2.z));
    v3 tmx = V(fmax_(t1.x, t2.x), fmax_(t1.y, t2.y), fmax_(t1.z, t2.z));
    float tmin_axis = fmax_(fmax_(tmn.x, tmn.y), tmn.z);
    float tmax_axis = fmin_(fmin_(tmx.x, tmx.y), tmx.z);
    if (tmin_axis > tmax_axis || tmax_axis < 0.0f) return 1e8f;
    return tmin_axis;
}

/* determinant(mat3x3(c0,c1,c2)) — cofactor expansion along the first column */
static inline float det3(v3 c0, v3 c1, v3 c2) {
    return (c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z))
           + c2.x * (c0.y * c1.z - c1.y * c0.z);
}

/* ---- ray.wgsl:768-847 hit_triangle ---------------------------------------------------------- */
static RenderState hit_triangle(Ray ray, float t_min, float t_max, v3 A, v3 B, v3 C, v3 n1, v3 n2, v3 n3) {
    RenderState rs = rs_zero();
    rs.hit = 0; rs.t = t_max;
    v3 ab = sub(B, A), ac = sub(C, A);
    v3 n = normalize(cross(ab, ac));
    float rdt = dot(ray.direction, n);
    if (rdt > 0.0f) { rdt = rdt * -1.0f; n = muls(n, -1.0f); }
    if (fabsf(rdt) < 0.00001f) return rs;
    float den = det3(ray.direction, sub(A, B), sub(A, C));
    if (fabsf(den) < 0.00001f) return rs;
    float u = det3(ray.direction, sub(A, ray.position), sub(A, C)) / den;
    if (u < 0.0f || u > 1.0f) return rs;
    float v = det3(ray.direction, sub(A, B), sub(A, ray.position)) / den;
    if (v < 0.0f || u + v > 1.0f) return rs;
    float t = det3(sub(A, ray.position), sub(A, B), sub(A, C)) / den;
    if (t > t_min && t < t_max) {
        v3 nm = add(add(muls(n1, (1.0f - u) - v), muls(n2, u)), muls(n3, v));
        v3 color = V(-nm.x * 0.5f + 0.5f, -nm.y * 0.5f + 0.5f, -nm.z * 0.5f + 0.5f);
        rs.normal = n; rs.color = color; rs.opacity = 1.0f; rs.t = t; rs.hit = 1;
        return rs;
    }
    return rs;
}

/* ---- ray.wgsl:287-363 trace_ray_model (D2: index stack) -------------------------------------- */
static RenderState trace_ray_model(const scene* S, Ray ray, int mi, float t_min, float t_max) {
    const o_model* M = &S->models[mi];
    v3 mpos = fromp(M->position);
    RenderState closest = rs_zero();
    closest.t = t_max;
    int node = 0;
    int stack[BVH_STACK]; int sp = 0;
    for (;;) {
        const o_node* N = &M->nodes[node];
        int obj_count = N->obj_count, contents = N->left_child;
        if (obj_count == 0) {
            int c1 = contents, c2 = contents + 1;
            if (S->cnt) S->cnt->node_pairs++;
            float d1 = hit_aabb(ray, &M->nodes[c1], mpos);
            float d2 = hit_aabb(ray, &M->nodes[c2], mpos);
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; int tc = c1; c1 = c2; c2 = tc; }
            if (d1 > closest.t) {
                if (sp == 0) break;
                node = stack[--sp];
            } else {
                node = c1;
                if (d2 < closest.t) { if (sp < BVH_STACK) stack[sp++] = c2; }
            }
        } else {
            for (int i = 0; i < obj_count; i++) {
                int idx = M->bvh_lookup[contents + i];
                o_tri ti = M->triangles[idx];
                if (S->cnt) S->cnt->triangles++;
                RenderState r = hit_triangle(ray, t_min, t_max,
                    add(fromp(M->points + 4 * ti.p1), mpos), add(fromp(M->points + 4 * ti.p2), mpos),
                    add(fromp(M->points + 4 * ti.p3), mpos),
                    fromp(M->normals + 4 * ti.n1), fromp(M->normals + 4 * ti.n2), fromp(M->normals + 4 * ti.n3));
                if (r.hit && r.t < closest.t) closest = r;
            }
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    return closest;
}

/* ---- ray.wgsl:365-393 hit_ray --------------------------------------------------------------- */
static RenderState hit_ray(const scene* S, Ray ray, float t_min, float t_max, float ray_distance,
                           int render_triangles, int render_black_hole) {
    RenderState closest = rs_zero();
    closest.t = t_max;
    /* The reference evaluates hit_black_hole unconditionally and discards the result when
     * !render_black_hole (ray.wgsl:369-374); it has no side effects, so it is skipped here
     * (keeps disk_hits = shading events that reach the image). */
    if (render_black_hole) {
        RenderState r = hit_black_hole(S, ray, t_min, t_max, ray_distance);
        if (r.hit && r.t < closest.t) closest = r;
    }
    if (render_triangles) {
        for (int i = 0; i < S->details->model_count; i++) {
            if (S->models[i].visible != 0) {
                RenderState r = trace_ray_model(S, ray, i, t_min, t_max);
                if (r.hit && r.t < closest.t) {
                    closest = r;
                    v3 light = normalize(V(0.2f, 0.2f, -1.0f));
                    float diffuse = dot(closest.normal, light);
                    closest.color = muls(closest.color, diffuse);
                }
            }
        }
    }
    return closest;
}

/* ---- ray.wgsl:401-403 f ---------------------------------------------------------------------- */
/* Two evaluations of the integrator.  The CONTRACT one (default; what the HIP kernel computes, bit for bit) uses the
 * evaluation freedoms WGSL grants - fused multiply-add (N7), pow of small integers by multiplication (N3), and
 * reassociation (N9): f(p) = (p - bh) * s with the per-step scalar s = (-1.5*h2) * (1/dist^5), and stage positions kept
 * relative to the hole, q_i = fma(sum_i, h, p0 - bh).  The LITERAL one (oracle_set_literal(1)) evaluates the shader text
 * operator by operator under N0-N2 (+ d*d*d*d*d for pow(d,5), l*l for pow(l,2)); tests/test_oracle_kat.py measures the
 * distance between the two, which is what separates any two conforming WGSL implementations. */
/* A THIRD evaluation (oracle_set_eval(2), kernel flag BHRAY_F_EVAL_FMA): the literal expression tree with fused multiply-add
 * contraction ONLY - every `x*y + z` of the text whose product is a direct operand of the addition becomes one fma, the first
 * product of a sum of products stays rounded - and none of the reassociations N9/N10 (no per-step scalar, no step size folded into
 * the stages, zero-coefficient terms kept).  It is what a shader compiler's default contraction does to ray.wgsl:401-480 and nothing
 * more.  It exists to show that the pixels on which the contract differs from the literal text by more than 1e-4 are the pixels on
 * which ANY two legal evaluations differ (tests/test_gpu_literal.py). */
static int g_eval = 0;                /* 0 contract (N3/N7/N9/N10), 1 literal (N0-N2), 2 fma contraction only */
#define g_literal (g_eval == 1)
void oracle_set_literal(int on) { g_eval = on != 0 ? 1 : 0; }
int oracle_get_literal(void) { return g_eval == 1; }
void oracle_set_eval(int mode) { g_eval = (mode == 1 || mode == 2) ? mode : 0; }
int oracle_get_eval(void) { return g_eval; }

static inline float pow5(float d) { return ((d * d) * (d * d)) * d; }                  /* N3 */
static inline v3 f_literal(const scene* S, v3 p, float h2, float dist) {
    v3 num = muls(sub(p, fromp(S->bh->position)), -1.5f * h2);
    return divs(num, pow5(dist));
}
/* N9: the scalar of f for one step */
static inline float f_scale(float h2, float dist) { return (-1.5f * h2) * (1.0f / pow5(dist)); }

/* ---- ray.wgsl:405-465 next_ray_rk (D1) -------------------------------------------------------- */
static RKState next_ray_rk_literal(const scene* S, RKState st) {
    Ray ray = st.ray;
    float dist = length(sub(ray.position, fromp(S->bh->position)));
    float lc = length(cross(ray.position, ray.direction));
    float h2 = lc * lc;
    v3 dydx = f_literal(S, ray.position, h2, dist);

    float h = st.h;
    v3 k1 = dydx;
    v3 k2 = f_literal(S, add(ray.position, muls(muls(k1, a_21), h)), h2, dist);
    v3 k3 = f_literal(S, add(ray.position, muls(add(muls(k1, a_31), muls(k2, a_32)), h)), h2, dist);
    v3 k4 = f_literal(S, add(ray.position, muls(add(add(muls(k1, a_41), muls(k2, a_42)), muls(k2, a_43)), h)), h2, dist);
    v3 k5 = f_literal(S, add(ray.position, muls(add(add(add(muls(k1, a_51), muls(k2, a_52)), muls(k3, a_53)), muls(k4, a_54)), h)), h2, dist);
    v3 k6 = f_literal(S, add(ray.position, muls(add(add(add(add(muls(k1, a_61), muls(k2, a_62)), muls(k3, a_63)), muls(k4, a_64)), muls(k5, a_65)), h)), h2, dist);

    v3 es = add(add(add(add(add(muls(k1, db_1), muls(k2, db_2)), muls(k3, db_3)), muls(k4, db_4)), muls(k5, db_5)), muls(k6, db_6));
    v3 e = muls(es, h);
    st.e_max = fmax_(fmax_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));

    v3 ds = add(add(add(add(add(muls(k1, b_a_1), muls(k2, b_a_2)), muls(k3, b_a_3)), muls(k4, b_a_4)), muls(k5, b_a_5)), muls(k6, b_a_6));
    st.ray.direction = normalize(add(st.ray.direction, muls(ds, st.h)));
    st.ray.position = add(st.ray.position, muls(ray.direction, st.h));   /* old direction */

    if (st.e_max > 0.00002f) st.h = st.h * (0.9f * bh_pow_m001(st.e_max));
    else st.h = st.h * 1.0001f;
    return st;
}

/* fma contraction only (oracle_set_eval(2)): the literal tree, `x*y + z` fused where the text has it */
static inline v3 f_fma(const scene* S, v3 p, float h2, float dist) {       /* fn f: no x*y + z in it */
    v3 num = muls(sub(p, fromp(S->bh->position)), -1.5f * h2);
    return divs(num, pow5(dist));
}
static RKState next_ray_rk_fma(const scene* S, RKState st) {
    Ray ray = st.ray;
    v3 p0 = ray.position, d0 = ray.direction;
    float dist = flength(sub(p0, fromp(S->bh->position)));
    float lc = flength(fcross(p0, d0));
    float h2 = lc * lc;
    float h = st.h;
    v3 k1 = f_fma(S, p0, h2, dist);
    v3 k2 = f_fma(S, fmadd3(muls(k1, a_21), h, p0), h2, dist);
    v3 k3 = f_fma(S, fmadd3(fmadd3(k2, a_32, muls(k1, a_31)), h, p0), h2, dist);
    v3 k4 = f_fma(S, fmadd3(fmadd3(k2, a_43, fmadd3(k2, a_42, muls(k1, a_41))), h, p0), h2, dist);                    /* a_43*k_2 (sic) */
    v3 k5 = f_fma(S, fmadd3(fmadd3(k4, a_54, fmadd3(k3, a_53, fmadd3(k2, a_52, muls(k1, a_51)))), h, p0), h2, dist);
    v3 k6 = f_fma(S, fmadd3(fmadd3(k5, a_65, fmadd3(k4, a_64, fmadd3(k3, a_63, fmadd3(k2, a_62, muls(k1, a_61))))), h, p0), h2, dist);
    v3 es = fmadd3(k6, db_6, fmadd3(k5, db_5, fmadd3(k4, db_4, fmadd3(k3, db_3, fmadd3(k2, db_2, muls(k1, db_1))))));
    v3 e = muls(es, h);
    st.e_max = fmax_(fmax_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));
    v3 ds = fmadd3(k6, b_a_6, fmadd3(k5, b_a_5, fmadd3(k4, b_a_4, fmadd3(k3, b_a_3, fmadd3(k2, b_a_2, muls(k1, b_a_1))))));
    st.ray.direction = fnormalize(fmadd3(ds, h, d0));
    st.ray.position = fmadd3(d0, h, p0);                                      /* old direction */
    if (st.e_max > 0.00002f) st.h = st.h * (0.9f * bh_pow_m001(st.e_max));
    else st.h = st.h * 1.0001f;
    return st;
}
static Ray next_ray_euler_fma(const scene* S, Ray ray, float step) {
    float lc = flength(fcross(ray.position, ray.direction));
    float h2 = lc * lc;
    float dist = flength(sub(ray.position, fromp(S->bh->position)));
    ray.direction = fnormalize(fmadd3(f_fma(S, ray.position, h2, dist), step, ray.direction));
    ray.position = fmadd3(ray.direction, step, ray.position);
    return ray;
}

/* contract: N3, N7, N9 */
/* sum of scaled vectors, left to right: first product rounded, the rest fused */
static inline v3 lin2(v3 a, float ca, v3 b, float cb) { return fmadd3(b, cb, muls(a, ca)); }
static RKState next_ray_rk(const scene* S, RKState st) {
    if (g_eval == 1) return next_ray_rk_literal(S, st);
    if (g_eval == 2) return next_ray_rk_fma(S, st);
    Ray ray = st.ray;
    v3 p0 = ray.position;
    v3 q0 = sub(p0, fromp(S->bh->position));                     /* N9: position relative to the hole, once per step */
    float dist = flength(q0);
    v3 cr = fcross(p0, ray.direction);
    float h2 = fdot(cr, cr);                                     /* N3: pow(length(v), 2.0) = dot(v, v) */
    float s = f_scale(h2, dist);                                 /* N9 */

    /* N10: the step size is distributed into the stages, K_i = h*k_i = (q0 + sum a_ij K_j) * (s*h); terms whose tableau
       coefficient is exactly 0 (b_2, b*_2) are dropped */
    float h = st.h;
    float sh = s * h;
    v3 K1 = muls(q0, sh);
    v3 K2 = muls(fmadd3(K1, a_21, q0), sh);
    v3 K3 = muls(fmadd3(K2, a_32, fmadd3(K1, a_31, q0)), sh);
    v3 K4 = muls(fmadd3(K2, a_43, fmadd3(K2, a_42, fmadd3(K1, a_41, q0))), sh);                        /* a_43*k_2 (sic) */
    v3 K5 = muls(fmadd3(K4, a_54, fmadd3(K3, a_53, fmadd3(K2, a_52, fmadd3(K1, a_51, q0)))), sh);
    v3 K6 = muls(fmadd3(K5, a_65, fmadd3(K4, a_64, fmadd3(K3, a_63, fmadd3(K2, a_62, fmadd3(K1, a_61, q0))))), sh);

    v3 e = fmadd3(K6, db_6, fmadd3(K5, db_5, fmadd3(K4, db_4, fmadd3(K3, db_3, muls(K1, db_1)))));
    /* yscal = 1, eps = 1: e/yscal and e_max/eps are exact */
    st.e_max = fmax_(fmax_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));
    /* D1: retry loop body cannot change h (h_temp < h for e_max > 1); executed once. */

    /* the small terms are summed first and added to the unit-length direction once (one rounding at magnitude 1) */
    v3 ds = fmadd3(K6, b_a_6, fmadd3(K5, b_a_5, fmadd3(K4, b_a_4, fmadd3(K3, b_a_3, muls(K1, b_a_1)))));
    st.ray.direction = fnormalize(add(st.ray.direction, ds));
    st.ray.position = fmadd3(ray.direction, st.h, st.ray.position);      /* old direction */

    if (st.e_max > 0.00002f) st.h = st.h * (0.9f * bh_pow_m001(st.e_max));
    else st.h = st.h * 1.0001f;
    return st;
}

/* ---- ray.wgsl:467-480 next_ray_euler ------------------------------------------------------------ */
static Ray next_ray_euler_literal(const scene* S, Ray ray, float step) {
    float lc = length(cross(ray.position, ray.direction));
    float h2 = lc * lc;
    float dist = length(sub(ray.position, fromp(S->bh->position)));
    ray.direction = normalize(add(ray.direction, muls(f_literal(S, ray.position, h2, dist), step)));
    ray.position = add(ray.position, muls(ray.direction, step));
    return ray;
}
static Ray next_ray_euler(const scene* S, Ray ray, float step) {
    if (g_eval == 1) return next_ray_euler_literal(S, ray, step);
    if (g_eval == 2) return next_ray_euler_fma(S, ray, step);
    v3 cr = fcross(ray.position, ray.direction);
    float h2 = fdot(cr, cr);                                     /* N3: pow(length(v), 2.0) = dot(v, v) */
    v3 q0 = sub(ray.position, fromp(S->bh->position));
    float dist = flength(q0);
    ray.direction = fnormalize(fmadd3(q0, f_scale(h2, dist) * step, ray.direction));          /* N9, N10 */
    ray.position = fmadd3(ray.direction, step, ray.position);
    return ray;
}

/* ---- ray.wgsl:482-596 trace_ray ------------------------------------------------------------------ */
static v4 trace_ray(const scene* S, Ray ray) {
    const o_black_hole* bh = S->bh;
    v3 bpos = fromp(bh->position);
    float bh_radius = bh->relativity_radius;
    int relativity = distance(ray.position, bpos) < bh_radius;
    const float t_max = 1e5f, t_min = 1e-8f;
    Ray curr = ray, prev = ray;
    float color_amount = 1.0f;
    v3 color = V(0, 0, 0);
    float step_size = S->details->step_size;
    RKState rk = { step_size, 0.0f, curr };
    float ray_distance = distance(ray.position, bpos);
    int hit = 0;
    int i = 0;
    float closest_to_bh = distance(curr.position, bpos);
    float aux_edge = 1e30f, aux_hits = 0.0f;
    if (S->cnt) S->cnt->traced++;

    for (; i < S->details->max_iterations; i++) {
        RenderState crs = rs_zero();
        crs.t = t_max;
        if (relativity) {
            if (S->cnt) S->cnt->steps++;
            prev = curr;
            if (S->details->integration_method == 0) {
                curr = next_ray_euler(S, curr, step_size);
            } else {
                rk = next_ray_rk(S, rk);
                curr = rk.ray;
                step_size = rk.h;
            }
            float cd = g_eval == 1 ? distance(curr.position, bpos) : fdistance(curr.position, bpos);   /* N7: the integrator's distance */
            if (cd < closest_to_bh) closest_to_bh = cd;
            prev.direction = curr.direction;
            crs = hit_ray(S, prev, t_min, step_size, ray_distance, 0, 1);
            if (S->aux) {                                   /* diagnostics only: how close to the disk's rims did this step cross its plane? */
                v3 n = fromp(bh->normal);
                float t = dot(sub(bpos, prev.position), n) / dot(n, prev.direction);
                if (t < step_size && t > t_min) {
                    float dc = distance(bpos, add(prev.position, muls(prev.direction, t)));
                    float e0 = fabsf(dc - bh->inner_radius), e1 = fabsf(dc - bh->outer_radius);
                    float e = e0 < e1 ? e0 : e1;
                    if (e < aux_edge) aux_edge = e;
                    if (dc >= bh->inner_radius && dc <= bh->outer_radius) aux_hits += 1.0f;
                }
            }
            if (cd > bh_radius) {
                relativity = 0;
                float fw = bh_radius * bh->feather_amount;
                float fs = bh_radius - fw;
                float lin = clampf((closest_to_bh - fs) / fw, 0.0f, 1.0f);
                float m = lin * lin;
                curr.direction = mix3(curr.direction, ray.direction, m);
            }
        } else {
            if (S->cnt) S->cnt->flat_iters++;
            float rd = distance(ray.position, bpos);
            RenderState rs = hit_ray(S, curr, t_min, t_max, rd, 1, 0);
            RenderState hs = hit_sphere(prev, bh_radius, bpos, V(0, 0, 0), t_min, t_max);
            if (!hs.hit && !rs.hit) break;
            if (hs.hit && hs.t < rs.t) {
                curr.position = add(curr.position, muls(curr.direction, hs.t));
                relativity = 1;
                /* The RK state carries its own ray (ray.wgsl:528-529): it is NOT refreshed
                 * from curr_ray here — the reference steps rk_state.ray, see note below. */
            } else {
                crs = rs;
            }
        }
        if (crs.hit) {
            curr.position = add(curr.position, muls(prev.direction, crs.t));
            v3 cc = V(clampf(crs.color.x, 0, 1), clampf(crs.color.y, 0, 1), clampf(crs.color.z, 0, 1));
            color = add(color, muls(cc, color_amount * crs.opacity));
            color_amount *= 1.0f - crs.opacity;
            hit = 1;
        }
        if (color_amount < 0.005f) break;
    }

    v4 out;
    if (S->aux) { S->aux[0] = closest_to_bh; S->aux[1] = (float)i; S->aux[2] = aux_edge; S->aux[3] = aux_hits; }
    if (hit || i <= 5) {
        if (color_amount > 0.001f) {
            if (S->cnt) S->cnt->sky_samples++;
            v3 sp = cartesian_to_spherical(V(curr.direction.x, curr.direction.z, curr.direction.y));
            float u = (sp.z + 2.6f * PI_F) / (2.0f * PI_F);
            float v = (PI_F - sp.y) / PI_F;
            u = u - truncf(u); v = v - truncf(v);                    /* WGSL % 1.0 = truncated */
            v4 sc = sample_bilinear(&S->t_sky, u, v);
            v3 miss = V((sc.x * sc.x) * (sc.x * sc.x), (sc.y * sc.y) * (sc.y * sc.y), (sc.z * sc.z) * (sc.z * sc.z));
            color = add(color, muls(miss, color_amount));
        }
        out.x = color.x; out.y = color.y; out.z = color.z; out.w = 1.0f;
        return out;
    }
    out.x = curr.direction.x; out.y = curr.direction.y; out.z = curr.direction.z; out.w = 0.0f;
    return out;
}

/* ---- ray.wgsl:167-243 main: one invocation per pixel ------------------------------------------ */
static inline v4 load4(const float* img, int w, int h, int x, int y) {
    /* textureLoad; out-of-range coordinates are clamped (only reachable on ladders that do
     * not follow the reference rule r <- 3r-2; documented in DESIGN.md) */
    if (x < 0) x = 0; if (x > w - 1) x = w - 1;
    if (y < 0) y = 0; if (y > h - 1) y = h - 1;
    const float* p = img + 4 * ((size_t)y * (size_t)w + (size_t)x);
    v4 r = { p[0], p[1], p[2], p[3] };
    return r;
}
static inline v3 xyz(v4 a) { return V(a.x, a.y, a.z); }

static v4 pixel_main(const scene* S, int px, int py, int sw, int sh, const float* prev, int pw, int ph) {
    if (pw == 1 && ph == 1) {                                   /* base case, ray.wgsl:178-182 */
        return trace_ray(S, create_ray(S, px, py, sw, sh));
    }
    int sfx = (sw - 1) / (pw - 1), sfy = (sh - 1) / (ph - 1);   /* ray.wgsl:185 (integer) */
    float rx = (float)pw / (float)(sw + (sfx - 1)), ry = (float)ph / (float)(sh + (sfy - 1));
    float ppx = (float)px * rx, ppy = (float)py * ry;
    float tlx = floorf(ppx), tly = floorf(ppy);
    v4 c_tl = load4(prev, pw, ph, (int)tlx, (int)tly);
    if (fabsf(tlx - ppx) < 0.001f && fabsf(tly - ppy) < 0.001f) {   /* ray.wgsl:193-194 */
        if (S->cnt) S->cnt->copied++;
        return c_tl;
    }
    v4 c_bl = load4(prev, pw, ph, (int)tlx, (int)(tly + 1.0f));
    v4 c_tr = load4(prev, pw, ph, (int)(tlx + 1.0f), (int)tly);
    v4 c_br = load4(prev, pw, ph, (int)(tlx + 1.0f), (int)(tly + 1.0f));
    /* tl_cart..br_cart (ray.wgsl:203-206) are dead */
    float a0 = angle_between(xyz(c_bl), xyz(c_tl));
    float a1 = angle_between(xyz(c_br), xyz(c_tr));
    float a2 = angle_between(xyz(c_tl), xyz(c_tr));
    float a3 = angle_between(xyz(c_bl), xyz(c_br));
    float thr = S->details->angle_division_threshold;
    int alphas0 = c_tl.w == 0.0f && c_tr.w == 0.0f && c_bl.w == 0.0f && c_br.w == 0.0f;
    if (alphas0 && a0 < thr && a1 < thr && a2 < thr && a3 < thr) {     /* ray.wgsl:217-234 */
        float tx = ppx - tlx, ty = ppy - tly;
        v3 top = mix3(xyz(c_tl), xyz(c_tr), tx);
        v3 bot = mix3(xyz(c_bl), xyz(c_br), tx);
        v3 p = mix3(top, bot, ty);
        if (S->cnt) S->cnt->interpolated++;
        v4 o = { p.x, p.y, p.z, 0.0f };
        return o;
    }
    return trace_ray(S, create_ray(S, px, py, sw, sh));             /* ray.wgsl:236-238 */
}


/* ---- sky resolve pass: /root/reference/src/renderer/shaders/sky.wgsl:1-38 ----------------------------------
 * (the consumer of the ray pass output.)  alpha == 0 pixels carry an escape direction:
 * direction -> equirect uv -> sky^4, alpha 1; other pixels pass through.  The target is rgba16float
 * (sky.wgsl:1): binary32 -> binary16 with round-to-nearest-even. */
static uint16_t f32_to_f16_rne(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((em > 0x7f800000u) ? 0x0200u | ((em >> 13) & 0x03ffu) : 0u));   /* inf / NaN */
    if (em >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 /* rounds to >= 65520 -> inf */
    if (em < 0x33000001u) return (uint16_t)sign;                             /* < 2^-25 (or == 2^-25 tie to even 0) -> 0 */
    int32_t e = (int32_t)(em >> 23) - 127;
    uint32_t m = (em & 0x007fffffu) | 0x00800000u;
    uint32_t shift, half;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); }                     /* subnormal half */
    else { shift = 13; }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u);
    half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out;
    if (e < -14) out = q;                                                    /* may carry into the smallest normal */
    else out = ((uint32_t)(e + 15) << 10) + (q - 0x400u);                    /* q in [0x400, 0x800]; carry bumps the exponent */
    return (uint16_t)(sign | out);
}
uint16_t oracle_f32_to_f16(float f) { return f32_to_f16_rne(f); }

int oracle_sky_resolve(const float* prev, int w, int h, const uint8_t* sky_rgba, int sw, int sh, uint16_t* out) {
    if (!prev || !sky_rgba || !out) return -1;
    o_tex t = { sky_rgba, sw, sh };
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const float* p = prev + 4 * ((size_t)y * w + x);
            uint16_t* o = out + 4 * ((size_t)y * w + x);
            float r = p[0], g = p[1], b = p[2], a = p[3];
            if (a == 0.0f) {                                                  /* sky.wgsl:19 */
                v3 sp = cartesian_to_spherical(V(p[0], p[2], p[1]));          /* p.xzy */
                float u = (sp.z + 2.6f * PI_F) / (2.0f * PI_F);
                float v = (PI_F - sp.y) / PI_F;
                u = u - truncf(u); v = v - truncf(v);
                v4 sc = sample_bilinear(&t, u, v);
                r = (sc.x * sc.x) * (sc.x * sc.x); g = (sc.y * sc.y) * (sc.y * sc.y); b = (sc.z * sc.z) * (sc.z * sc.z); a = 1.0f;
            }
            o[0] = f32_to_f16_rne(r); o[1] = f32_to_f16_rne(g); o[2] = f32_to_f16_rne(b); o[3] = f32_to_f16_rne(a);
        }
    }
    return 0;
}

/* ---- exported entry points (ctypes) --------------------------------------------------------------- */
typedef struct {
    const o_camera* camera; const o_details* details; const o_black_hole* bh;
    const o_model* models;
    o_tex t_temp, t_disk, t_sky;
} oracle_scene;

/* Renders the pixels (x in [x0,x1), y in [y0,y1)) of one ladder level of size sw×sh into
 * out[sh][sw][4] (other pixels untouched).  prev/pw/ph = previous level (pw=ph=1 ⇒ base).
 * mask (optional, sw*sh bytes): only pixels with mask!=0 are computed.  Row-parallel (OpenMP). */
int oracle_render_level(const oracle_scene* os, int sw, int sh, const float* prev, int pw, int ph,
                        float* out, int x0, int y0, int x1, int y1, const uint8_t* mask,
                        o_counters* counters) {
    if (!os || !out || sw < 1 || sh < 1) return -1;
    o_counters total; memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        o_counters local; memset(&local, 0, sizeof local);
        scene S; S.camera = os->camera; S.details = os->details; S.bh = os->bh; S.models = os->models;
        S.t_temp = os->t_temp; S.t_disk = os->t_disk; S.t_sky = os->t_sky; S.cnt = counters ? &local : NULL; S.aux = NULL;
#pragma omp for schedule(dynamic, 1)
        for (int y = y0; y < y1; y++) {
            for (int x = x0; x < x1; x++) {
                if (mask && !mask[(size_t)y * sw + x]) continue;
                v4 c = pixel_main(&S, x, y, sw, sh, prev, pw, ph);
                float* o = out + 4 * ((size_t)y * (size_t)sw + (size_t)x);
                o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
                local.pixels++;
            }
        }
#pragma omp critical
        {
            total.pixels += local.pixels; total.copied += local.copied; total.interpolated += local.interpolated;
            total.traced += local.traced; total.steps += local.steps; total.flat_iters += local.flat_iters;
            total.node_pairs += local.node_pairs; total.triangles += local.triangles;
            total.disk_hits += local.disk_hits; total.sky_samples += local.sky_samples;
        }
    }
    if (counters) {
        counters->pixels += total.pixels; counters->copied += total.copied; counters->interpolated += total.interpolated;
        counters->traced += total.traced; counters->steps += total.steps; counters->flat_iters += total.flat_iters;
        counters->node_pairs += total.node_pairs; counters->triangles += total.triangles;
        counters->disk_hits += total.disk_hits; counters->sky_samples += total.sky_samples;
    }
    return 0;
}

/* Diagnostics for the parity tests: EVERY pixel of a sw x sh level traced from the camera (no grid), aux[sh][sw][4] =
 * (closest approach to the hole, iterations, smallest distance between a disk-plane crossing and the disk's inner / outer rim
 * (1e30: the ray never crossed the plane), number of disk hits).  These are the quantities that say WHY a ray amplifies rounding
 * differences (it wound around the photon sphere; it crossed the plane at a rim, where an ulp decides hit / no hit). */
int oracle_render_aux(const oracle_scene* os, int sw, int sh, float* aux) {
    if (!os || !aux || sw < 1 || sh < 1) return -1;
#pragma omp parallel
    {
        scene S; S.camera = os->camera; S.details = os->details; S.bh = os->bh; S.models = os->models;
        S.t_temp = os->t_temp; S.t_disk = os->t_disk; S.t_sky = os->t_sky; S.cnt = NULL;
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < sh; y++) {
            for (int x = 0; x < sw; x++) {
                S.aux = aux + 4 * ((size_t)y * (size_t)sw + (size_t)x);
                (void)trace_ray(&S, create_ray(&S, x, y, sw, sh));
            }
        }
    }
    return 0;
}

/* Diagnostics: the grid decision of ray.wgsl:167-243 alone, for every pixel of a level - 0 copy, 1 interpolate, 2 trace - without
 * tracing anything (the parity analysis and the queue-order simulation of profiles/queue_order_sim.py want to know WHICH pixels a
 * level traces). */
int oracle_classify_level(const oracle_scene* os, int sw, int sh, const float* prev, int pw, int ph, uint8_t* kind) {
    if (!os || !kind || !prev || sw < 1 || sh < 1 || pw < 2 || ph < 2) return -1;
    const float thr = os->details->angle_division_threshold;
    const int sfx = (sw - 1) / (pw - 1), sfy = (sh - 1) / (ph - 1);
    const float rx = (float)pw / (float)(sw + (sfx - 1)), ry = (float)ph / (float)(sh + (sfy - 1));
#pragma omp parallel for schedule(static)
    for (int py = 0; py < sh; py++) {
        for (int px = 0; px < sw; px++) {
            const float ppx = (float)px * rx, ppy = (float)py * ry;
            const float tlx = floorf(ppx), tly = floorf(ppy);
            uint8_t k = 2;
            if (fabsf(tlx - ppx) < 0.001f && fabsf(tly - ppy) < 0.001f) k = 0;
            else {
                v4 c_tl = load4(prev, pw, ph, (int)tlx, (int)tly), c_bl = load4(prev, pw, ph, (int)tlx, (int)(tly + 1.0f));
                v4 c_tr = load4(prev, pw, ph, (int)(tlx + 1.0f), (int)tly), c_br = load4(prev, pw, ph, (int)(tlx + 1.0f), (int)(tly + 1.0f));
                const int alphas0 = c_tl.w == 0.0f && c_tr.w == 0.0f && c_bl.w == 0.0f && c_br.w == 0.0f;
                if (alphas0 && angle_between(xyz(c_bl), xyz(c_tl)) < thr && angle_between(xyz(c_br), xyz(c_tr)) < thr &&
                    angle_between(xyz(c_tl), xyz(c_tr)) < thr && angle_between(xyz(c_bl), xyz(c_br)) < thr) k = 1;
            }
            kind[(size_t)py * (size_t)sw + (size_t)px] = k;
        }
    }
    return 0;
}

/* Per-function probes for known-answer tests and cross-checks (tests/ only). */
void oracle_create_ray(const oracle_scene* os, int px, int py, int sw, int sh, float out6[6]) {
    scene S; memset(&S, 0, sizeof S); S.camera = os->camera; S.details = os->details; S.bh = os->bh;
    Ray r = create_ray(&S, px, py, sw, sh);
    out6[0] = r.position.x; out6[1] = r.position.y; out6[2] = r.position.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
}
void oracle_trace_ray(const oracle_scene* os, const float ray6[6], float out4[4], o_counters* c) {
    scene S; S.camera = os->camera; S.details = os->details; S.bh = os->bh; S.models = os->models;
    S.t_temp = os->t_temp; S.t_disk = os->t_disk; S.t_sky = os->t_sky; S.cnt = c; S.aux = NULL;
    Ray r = { V(ray6[0], ray6[1], ray6[2]), V(ray6[3], ray6[4], ray6[5]) };
    v4 o = trace_ray(&S, r);
    out4[0] = o.x; out4[1] = o.y; out4[2] = o.z; out4[3] = o.w;
}
/* n integrator steps from ray6 with step h; method 0 Euler / 1 RK; out: pos, dir, h, e_max per step (8 floats) */
void oracle_integrate(const oracle_scene* os, const float ray6[6], float h, int method, int n, float* out8) {
    scene S; memset(&S, 0, sizeof S); S.camera = os->camera; S.details = os->details; S.bh = os->bh;
    Ray r = { V(ray6[0], ray6[1], ray6[2]), V(ray6[3], ray6[4], ray6[5]) };
    RKState st = { h, 0.0f, r };
    for (int i = 0; i < n; i++) {
        if (method == 0) { st.ray = next_ray_euler(&S, st.ray, h); }
        else st = next_ray_rk(&S, st);
        float* o = out8 + 8 * i;
        o[0] = st.ray.position.x; o[1] = st.ray.position.y; o[2] = st.ray.position.z;
        o[3] = st.ray.direction.x; o[4] = st.ray.direction.y; o[5] = st.ray.direction.z;
        o[6] = st.h; o[7] = st.e_max;
    }
}
/* kind: 0 sphere(radius=p[0], center=p[1..3]); 1 torus2d(inner=p[0], outer=p[1], pos=p[2..4], normal=p[5..7]);
 * 2 aabb(min=p[0..2], max=p[3..5], offset=p[6..8]) -> out[0]=t ; 3 triangle(A,B,C,n1,n2,n3 = p[0..17]).
 * out9: color(3), opacity, t, normal(3), hit */
void oracle_hit(int kind, const float ray6[6], const float* p, float t_min, float t_max, float out9[9]) {
    Ray r = { V(ray6[0], ray6[1], ray6[2]), V(ray6[3], ray6[4], ray6[5]) };
    RenderState rs = rs_zero();
    if (kind == 0) rs = hit_sphere(r, p[0], V(p[1], p[2], p[3]), V(0, 0, 0), t_min, t_max);
    else if (kind == 1) rs = hit_torus2d(r, p[0], p[1], V(p[2], p[3], p[4]), V(p[5], p[6], p[7]), t_min, t_max);
    else if (kind == 2) {
        o_node n; memcpy(n.min_corner, p, 12); memcpy(n.max_corner, p + 3, 12); n.left_child = 0; n.obj_count = 0;
        rs.t = hit_aabb(r, &n, V(p[6], p[7], p[8]));
    } else if (kind == 3) {
        rs = hit_triangle(r, t_min, t_max, V(p[0], p[1], p[2]), V(p[3], p[4], p[5]), V(p[6], p[7], p[8]),
                          V(p[9], p[10], p[11]), V(p[12], p[13], p[14]), V(p[15], p[16], p[17]));
    }
    out9[0] = rs.color.x; out9[1] = rs.color.y; out9[2] = rs.color.z; out9[3] = rs.opacity; out9[4] = rs.t;
    out9[5] = rs.normal.x; out9[6] = rs.normal.y; out9[7] = rs.normal.z; out9[8] = (float)rs.hit;
}
void oracle_sample(const uint8_t* rgba, int w, int h, float u, float v, float out4[4]) {
    o_tex t = { rgba, w, h };
    v4 c = sample_bilinear(&t, u, v);
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = c.w;
}
float oracle_angle_between(const float a[3], const float b[3]) { return angle_between(fromp(a), fromp(b)); }
int oracle_num_threads(void);
#ifdef _OPENMP
#include <omp.h>
int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int oracle_num_threads(void) { return 1; }
void oracle_set_threads(int n) { (void)n; }
#endif
