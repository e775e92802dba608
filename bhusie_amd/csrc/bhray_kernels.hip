// bhray_kernels.hip — gfx950 (CDNA4) kernels of the geodesic ray-trace pass.
//
// Replaces the WGSL compute shader /root/reference/src/renderer/shaders/ray.wgsl:1-847
// (one invocation per pixel, @workgroup_size(8,8,1)) with two kernels per ladder level:
//
//   classify_kernel  ray.wgsl:167-243 (`main`): per pixel copy / interpolate from the coarser
//                    level and store, or — when the pixel must be traced — append it to a work
//                    queue.  One wave = one 8x8 pixel tile; the append is a wave ballot + one
//                    atomic per wave, so queue order keeps tile locality.
//   trace_kernel     ray.wgsl:269-285, 365-393, 401-666, 725-766 (`create_ray`, `trace_ray`
//                    and everything it calls): persistent waves pull rays from the queue.  A
//                    lane whose ray has finished is refilled from the queue (ballot + prefix
//                    popcount), so step-count divergence between rays does not idle lanes; the
//                    rare flat-space / BVH iterations and the epilogue are executed in their
//                    own wave-uniform phases between batches of integrator steps, so that the
//                    hot loop contains only the integrator and the per-step horizon/disk test.
//
// No MFMA: the path is an f32 ODE march (VALU) plus byte/int texture and BVH reads (HBM/L2).
// Numerics: DESIGN.md §2 — compiled with -ffp-contract=off; operation order follows the shader, the integrator uses the
// explicit fused forms of N7, so results are bit-identical to the CPU oracle except for the shading-only optical-depth
// powf(.,1.3) (device libm, 1-2 ulp, not amplified).
#include <hip/hip_fp16.h>
#include <type_traits>

#include "bhray_internal.h"
#include "bhray_math.h"

namespace bhray {

// ------------------------------------------------------------------------------------------
// textures: RGBA8 unorm, bilinear, clamp-to-edge (texture.rs:32,61-69; textureSampleLevel 0)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 texel(const TexDev& t, int x, int y) {
    const uchar4 p = *reinterpret_cast<const uchar4*>(t.rgba + 4 * ((size_t)y * (size_t)t.w + (size_t)x));
    return make_float4((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f, (float)p.w / 255.0f);
}
__device__ __forceinline__ float unit_coord(float u, int n, int& i0, int& i1) {
    float x = u * (float)n - 0.5f;
    if (!(x >= -1.0f)) x = -1.0f;
    if (x > (float)n) x = (float)n;
    float fl = floorf(x);
    int a = (int)fl, b = a + 1;
    a = a < 0 ? 0 : a; a = a > n - 1 ? n - 1 : a;
    b = b < 0 ? 0 : b; b = b > n - 1 ? n - 1 : b;
    i0 = a; i1 = b;
    return x - fl;
}
__device__ __forceinline__ float4 sample_bilinear(const TexDev& t, float u, float v) {
    int x0, x1, y0, y1;
    float fx = unit_coord(u, t.w, x0, x1);
    float fy = unit_coord(v, t.h, y0, y1);
    float4 a = texel(t, x0, y0), b = texel(t, x1, y0), c = texel(t, x0, y1), d = texel(t, x1, y1);
    float4 r;
    r.x = mix_(mix_(a.x, b.x, fx), mix_(c.x, d.x, fx), fy);
    r.y = mix_(mix_(a.y, b.y, fx), mix_(c.y, d.y, fx), fy);
    r.z = mix_(mix_(a.z, b.z, fx), mix_(c.z, d.z, fx), fy);
    r.w = mix_(mix_(a.w, b.w, fx), mix_(c.w, d.w, fx), fy);
    return r;
}

// The same sample in two halves - the four texel loads issued, then (later) converted and blended - so that a caller with two independent samples has both
// in flight at once (disk shading; a wave that runs alone waits a memory round trip for each: one frame at a time -0.5 to -1 %, profiles/EXPERIMENTS.md R6.13).
struct BilinearTaps { uchar4 a, b, c, d; float fx, fy; };
__device__ __forceinline__ BilinearTaps fetch_bilinear(const TexDev& t, float u, float v) {
    int x0, x1, y0, y1;
    BilinearTaps r;
    r.fx = unit_coord(u, t.w, x0, x1);
    r.fy = unit_coord(v, t.h, y0, y1);
    r.a = *reinterpret_cast<const uchar4*>(t.rgba + 4 * ((size_t)y0 * (size_t)t.w + (size_t)x0));
    r.b = *reinterpret_cast<const uchar4*>(t.rgba + 4 * ((size_t)y0 * (size_t)t.w + (size_t)x1));
    r.c = *reinterpret_cast<const uchar4*>(t.rgba + 4 * ((size_t)y1 * (size_t)t.w + (size_t)x0));
    r.d = *reinterpret_cast<const uchar4*>(t.rgba + 4 * ((size_t)y1 * (size_t)t.w + (size_t)x1));
    return r;
}
__device__ __forceinline__ float4 unorm4(uchar4 p) { return make_float4((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f, (float)p.w / 255.0f); }
__device__ __forceinline__ float4 blend_bilinear(const BilinearTaps& t) {
    const float4 a = unorm4(t.a), b = unorm4(t.b), c = unorm4(t.c), d = unorm4(t.d);
    float4 r;
    r.x = mix_(mix_(a.x, b.x, t.fx), mix_(c.x, d.x, t.fx), t.fy);
    r.y = mix_(mix_(a.y, b.y, t.fx), mix_(c.y, d.y, t.fx), t.fy);
    r.z = mix_(mix_(a.z, b.z, t.fx), mix_(c.z, d.z, t.fx), t.fy);
    r.w = mix_(mix_(a.w, b.w, t.fx), mix_(c.w, d.w, t.fx), t.fy);
    return r;
}

// ------------------------------------------------------------------------------------------
// intersections
// ------------------------------------------------------------------------------------------
struct Hit {            // RenderState (ray.wgsl:92-98) without the members nothing reads
    F3 color;
    float opacity;
    float t;
    bool hit;
};

// hit_sphere, ray.wgsl:725-766.  Returns hit and t (the normal is never consumed on this path).
__device__ __forceinline__ bool hit_sphere(F3 pos, F3 dir, float radius, F3 center, float t_min, float t_max, float& t_out) {
    F3 oc = pos - center;
    float a = dot(dir, dir);
    float b = 2.0f * dot(oc, dir);
    float c = dot(oc, oc) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float sq = sqrtf(disc);
        float t1 = (-b - sq) / (2.0f * a);
        float t2 = (-b + sq) / (2.0f * a);
        float tc = t_max;
        if (t1 > t_min && t1 < t_max) tc = t1;
        if (t2 > t_min && t2 < t_max && t2 < tc) tc = t2;
        if (tc < t_max && tc > t_min) { t_out = tc; return true; }
    }
    return false;
}

// hit_torus2d, ray.wgsl:668-701.
__device__ __forceinline__ bool hit_torus2d(F3 pos, F3 dir, float inner, float outer, F3 tpos, F3 normal,
                                            float t_min, float t_max, float& t_out) {
    float denom = dot(normal, dir);
    F3 dist = tpos - pos;
    float t = dot(dist, normal) / denom;
    if (t < t_max && t > t_min) {
        F3 ip = pos + dir * t;
        float dc = distance(tpos, ip);
        if (dc >= inner && dc <= outer) { t_out = t; return true; }
    }
    return false;
}

// Frame uniforms read inside the integrator step loop.  FrameParams lives in HBM and is read with scalar loads; a load that
// sits behind a branch (disk shading, sphere exit) cannot be hoisted by the compiler - the pointer is not known to be
// dereferenceable - and an s_load + s_waitcnt inside the loop is fully exposed when a wave runs alone on its SIMD (+10-20 %
// per coarse-level launch).  The loop's operands are therefore loaded once per frame and pinned in SGPRs.
struct HotParams {
    F3 bh, bn;
    float bn_len, inner, outer, R, ray_distance, feather, time_rot, step_size;
    float outer_pad, plane_c1, plane_c2;      // black_hole_culls: outer + 0.05, 1.01 |n|, 1e-4 |n| (slightly more than the unfolded form's margins: rounded up)
    int max_iter, show_tex, show_shift;
    float M[9];
    TexDev disk, temp;
};
__device__ __forceinline__ float pin_sgpr(float v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ int pin_sgpr(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ const uint8_t* pin_sgpr(const uint8_t* v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ TexDev pin_sgpr(const TexDev& t) { TexDev r; r.rgba = pin_sgpr(t.rgba); r.w = pin_sgpr(t.w); r.h = pin_sgpr(t.h); return r; }
template <bool FOLDED_CULLS = false>
__device__ __forceinline__ HotParams load_hot(const FrameParams& P) {
    HotParams H;
    H.bh = f3(pin_sgpr(P.bh[0]), pin_sgpr(P.bh[1]), pin_sgpr(P.bh[2]));
    H.bn = f3(pin_sgpr(P.bn[0]), pin_sgpr(P.bn[1]), pin_sgpr(P.bn[2]));
    H.bn_len = pin_sgpr(P.bn_len);
    H.inner = pin_sgpr(P.inner); H.outer = pin_sgpr(P.outer); H.R = pin_sgpr(P.R);
    if (FOLDED_CULLS) { H.outer_pad = pin_sgpr(P.cull_outer_pad); H.plane_c1 = pin_sgpr(P.cull_plane_c1); H.plane_c2 = P.cull_plane_c2; }      // formed on the host (gfx950 has no scalar float arithmetic)
    else { H.outer_pad = 0.0f; H.plane_c1 = 0.0f; H.plane_c2 = 0.0f; }
    H.ray_distance = pin_sgpr(P.ray_distance); H.feather = pin_sgpr(P.feather);
    H.time_rot = pin_sgpr(P.time_rot); H.step_size = pin_sgpr(P.step_size);
    H.max_iter = pin_sgpr(P.max_iter); H.show_tex = pin_sgpr(P.show_tex); H.show_shift = pin_sgpr(P.show_shift);
#pragma unroll
    for (int k = 0; k < 9; k++) H.M[k] = pin_sgpr(P.M[k]);
    H.disk = pin_sgpr(P.disk); H.temp = pin_sgpr(P.temp);
    return H;
}

// Disk shading, ray.wgsl:612-663 (the part of hit_black_hole after the disk won).
template <bool COUNT>
__device__ __forceinline__ void shade_disk(const HotParams& P, F3 pos, F3 dir, float t, float total_distance, Hit& rs,
                                        unsigned long long* cnt) {
    F3 bpos = P.bh;
    F3 ip = pos + dir * t;
    float dist = distance(bpos, ip);
    float density = 1.0f - length(div_s(ip, P.outer));
    {
        float e0 = P.inner, e1 = P.inner + 1.0f;
        float s = clamp_((dist - e0) / (e1 - e0), 0.0f, 1.0f);
        density *= s * s * (3.0f - 2.0f * s);
    }
    density *= 1.0f / sqrtf(dist);
    float od = powf(30.0f * density, 1.3f);
    rs.opacity = clamp_(od * 0.2f, 0.0f, 1.0f);
    rs.color = f3(od, od, od);
    if (COUNT) cnt[8]++;
    // The two texture samples do not depend on each other: both sets of texel loads are issued before either is used (the arithmetic and its order are the shader's).
    BilinearTaps tap_disk, tap_temp;
    if (P.show_tex != 0) {
        float r = (dist - P.inner) / (P.outer - P.inner);
        F3 rel = div_s(ip - bpos, P.outer);
        F3 c0 = f3(P.M[0], P.M[1], P.M[2]), c1 = f3(P.M[3], P.M[4], P.M[5]), c2 = f3(P.M[6], P.M[7], P.M[8]);
        F3 rot = (c0 * rel.x + c1 * rel.y) + c2 * rel.z;
        float angle = -bh_atan2(rot.z, rot.x);
        float ph = angle + P.time_rot;        // time * rotation_speed (ray.wgsl:633), one binary32 product, formed on the host
        float u = bh_sincos<0>(ph) * r, v = bh_sincos<1>(ph) * r;
        u = (u + 1.0f) * 0.5f; v = (v + 1.0f) * 0.5f;
        tap_disk = fetch_bilinear(P.disk, u, v);
    }
    if (P.show_shift != 0) {
        float temp_max = 100000.0f, temp_min = 10000.0f, temp = 15000.0f;
        float y = 1.0f - (temp - temp_min) / (temp_max - temp_min);
        F3 sv = cross(normalize(ip), normalize(f3(0.0f, -1.0f, 0.0f))) * 0.6f;
        float velocity = dot(dir, sv);
        float doppler = sqrtf((1.0f - velocity) / (1.0f + velocity));
        float grav = sqrtf((1.0f - 2.0f / dist) / (1.0f - 2.0f / total_distance));
        float sh = clamp_(grav * doppler, 0.0f, 1.0f);
        tap_temp = fetch_bilinear(P.temp, sh * sh, y);
    }
    if (P.show_tex != 0) {
        const float4 dc = blend_bilinear(tap_disk);
        rs.opacity *= clamp_(0.7f + dc.w * 0.5f, 0.0f, 1.0f);
        rs.color = rs.color * (f3(dc.x, dc.y, dc.z) * dc.w);
    }
    if (P.show_shift != 0) {
        const float4 sc = blend_bilinear(tap_temp);
        rs.color = rs.color * f3(sc.x, sc.y, sc.z);
    }
}

// hit_black_hole, ray.wgsl:598-666: horizon sphere (radius 1, colour 0) vs. disk.
//
// The shader evaluates both intersections on every integrator step.  Here a step first checks, from
// quantities the literal tests compute anyway (oc.oc and the signed plane distance), whether the
// segment (t_min, t_max = step] can reach the horizon or the disk at all; if it cannot, the quadratic
// (sqrt + 2 divisions) and the plane division are skipped.  The culls are conservative by >= 1 %:
//   horizon: a hit needs |oc| <= 1 + t|d| with t < step, |d| <= 1 + 2e-7   -> skip when |oc| > 1.05 + 1.05 step
//   disk:    t = fl(numer/denom) < step with |denom| <= |n||d|(1 + 4e-7)  -> skip when |numer| > 1.01 |n| step,
//            and the hit point must lie within `outer` of the centre       -> skip when |oc| > outer + 0.05 + 1.05 step
// so a skipped test is one whose literal evaluation returns "no hit": results are unchanged bit for bit
// (checked against the oracle, which always evaluates the literal tests).  The cull quantities are not part of the
// shader, so they are the cheapest sufficient ones: |oc| is the distance the integrator already carries for this
// position (`pos_dist`, within a few ulp of the literal |oc|, far inside the margins), and the plane distance is
// n.(b - pos), from the hole-relative vector.
// Geometry only.  Returns true when the disk is the nearest hit (its parameter in td_out); the shading of that hit
// (shade_disk) is run by the caller - the trace kernel defers it to a wave-uniform phase of its own.  rs holds the horizon
// result (hit, t, colour 0, opacity 1) or "no hit".
// The two cull predicates of a step (see above): can the segment reach the horizon / the disk at all?
// FOLDED: the same three predicates with the frame's constants folded (the culls are not the shader's arithmetic: any form that stays
// conservative by more than its own rounding will do - the margins are 1-5 %, the rounding 1e-7): pos_dist <= c + 1.05 t  <=>
// pos_dist - 1.05 t <= c, one fused operation for both radial tests; the plane test's bound one fused operation - 4 vector instructions
// per step less.  Measured (profiles/r05_ab_culls_folded.txt, three rounds): the mesh variant +2.4 % (20- and 400-frame blocks), the
// no-mesh RK kernel -0.7 % / -1.0 %, Euler -1.3 %: fewer instructions, a worse schedule.  So the mesh variant folds, the others do not.
template <bool FOLDED>
__device__ __forceinline__ void black_hole_culls(const HotParams& H, F3 pos, float pos_dist, float t_max, bool& near_horizon, bool& near_disk) {
    // signed plane distance from the hole-RELATIVE position: its rounding error (a few ulp of |pos - bh| <= outer + reach) does not
    // grow with |bh|, unlike n.bh - n.pos for a hole far from the origin
    const float numer = fdot(H.bh - pos, H.bn);
    if (FOLDED) {
        const float e = __builtin_fmaf(t_max, -1.05f, pos_dist);
        near_horizon = e <= 1.05f;
        near_disk = (e <= H.outer_pad) & (fabsf(numer) <= __builtin_fmaf(t_max, H.plane_c1, H.plane_c2));     // & : no branch
    } else {
        const float reach = 1.05f * t_max + 0.05f;
        near_horizon = pos_dist <= 1.0f + reach;
        near_disk = (pos_dist <= H.outer + reach) & (fabsf(numer) <= (1.01f * t_max) * H.bn_len + 1e-4f * H.bn_len);     // & : no branch
    }
}
// The same predicates with the plane distance handed in (the unified march has the hole-relative position of the segment's start already: n.(pos - b) is the
// negative of n.(b - pos) bit for bit - a - b = -(b - a) and fdot's products and sums change sign with their operands - and only its magnitude is used).
template <bool FOLDED>
__device__ __forceinline__ void black_hole_culls_rel(const HotParams& H, float numer, float pos_dist, float t_max, bool& near_horizon, bool& near_disk) {
    if (FOLDED) {
        const float e = __builtin_fmaf(t_max, -1.05f, pos_dist);
        near_horizon = e <= 1.05f;
        near_disk = (e <= H.outer_pad) & (fabsf(numer) <= __builtin_fmaf(t_max, H.plane_c1, H.plane_c2));
    } else {
        const float reach = 1.05f * t_max + 0.05f;
        near_horizon = pos_dist <= 1.0f + reach;
        near_disk = (pos_dist <= H.outer + reach) & (fabsf(numer) <= (1.01f * t_max) * H.bn_len + 1e-4f * H.bn_len);
    }
}
__device__ __forceinline__ bool hit_black_hole_geom(const HotParams& H, F3 pos, F3 dir, bool near_horizon, bool near_disk, float t_min, float t_max, Hit& rs, float& td_out) {
    const F3 bpos = H.bh;
    float ts = t_max, td = t_max;
    bool hs = false, hd = false;
    if (near_horizon) hs = hit_sphere(pos, dir, 1.0f, bpos, t_min, t_max, ts);
    if (near_disk) hd = hit_torus2d(pos, dir, H.inner, H.outer, bpos, H.bn, t_min, t_max, td);
    rs.hit = hs; rs.t = hs ? ts : t_max; rs.color = f3(0.0f, 0.0f, 0.0f); rs.opacity = hs ? 1.0f : 0.0f;
    td_out = td;
    return hd && td < rs.t;
}

// hit_aabb, ray.wgsl:703-723 (node = two float4 halves).
__device__ __forceinline__ float hit_aabb(F3 pos, F3 inv, float4 lo, float4 hi, F3 offset) {
    F3 mn = f3(lo.x, lo.y, lo.z) + offset, mx = f3(hi.x, hi.y, hi.z) + offset;
    F3 t1 = (mn - pos) * inv, t2 = (mx - pos) * inv;
    F3 tmn = f3(min_(t1.x, t2.x), min_(t1.y, t2.y), min_(t1.z, t2.z));
    F3 tmx = f3(max_(t1.x, t2.x), max_(t1.y, t2.y), max_(t1.z, t2.z));
    float tmin_axis = max_(max_(tmn.x, tmn.y), tmn.z);
    float tmax_axis = min_(min_(tmx.x, tmx.y), tmx.z);
    if (tmin_axis > tmax_axis || tmax_axis < 0.0f) return 1e8f;
    return tmin_axis;
}

__device__ __forceinline__ float det3(F3 c0, F3 c1, F3 c2) {
    return (c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z))
           + c2.x * (c0.y * c1.z - c1.y * c0.z);
}

// hit_triangle, ray.wgsl:768-847.  normal_out = face normal (consumed by the diffuse term).
__device__ __forceinline__ bool hit_triangle(F3 pos, F3 dir, float t_min, float t_max, F3 A, F3 B, F3 C,
                                             F3 n1, F3 n2, F3 n3, float& t_out, F3& color_out, F3& normal_out) {
    F3 ab = B - A, ac = C - A;
    F3 n = normalize(cross(ab, ac));
    float rdt = dot(dir, n);
    if (rdt > 0.0f) { rdt = rdt * -1.0f; n = n * -1.0f; }
    if (fabsf(rdt) < 0.00001f) return false;
    float den = det3(dir, A - B, A - C);
    if (fabsf(den) < 0.00001f) return false;
    float u = det3(dir, A - pos, A - C) / den;
    if (u < 0.0f || u > 1.0f) return false;
    float v = det3(dir, A - B, A - pos) / den;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = det3(A - pos, A - B, A - C) / den;
    if (t > t_min && t < t_max) {
        F3 nm = (n1 * ((1.0f - u) - v) + n2 * u) + n3 * v;
        color_out = f3(-nm.x * 0.5f + 0.5f, -nm.y * 0.5f + 0.5f, -nm.z * 0.5f + 0.5f);
        normal_out = n; t_out = t;
        return true;
    }
    return false;
}

// trace_ray_model, ray.wgsl:287-363.
// Latency is what matters here (a traversal is a chain of dependent loads executed for a few lanes of a wave), so
// the data is laid out for ONE round trip per tree level and per leaf:
//  - an inner node is never re-read: when a child pair is fetched (one 64 B segment — the builder allocates siblings
//    adjacently, triangle.rs:239-243), the pair's own (left_child, obj_count) words are kept for the child that is
//    descended into or pushed, because a node's bounds are not needed again once its parent has tested them;
//  - leaf geometry is pre-gathered at upload into `leaf` (3 points + 3 normals per bvh_lookup slot), removing the
//    bvh_lookup -> triangle -> point/normal index chain (ray.wgsl:333-343) from the traversal.
// LDS (north_star: "LDS-staged triangle/node tiles"):
//   BHRAY_BVH_LDS_STACK  entries of the short traversal stack (entry k of lane t at [k * threads + t]: conflict-free 8-byte accesses);
//                        see "The traversal stack" below.
//   (The first N nodes of the breadth-first order staged in LDS once per block were built and measured in round 2: no gain, the top
//   of the tree is L1/L2-resident anyway - profiles/EXPERIMENTS.md; removed in round 4.)
#ifndef BHRAY_TRACE_THREADS
#define BHRAY_TRACE_THREADS 256  // threads per persistent trace block (a multiple of 64; 64 / 128 / 512 measured slower: DESIGN.md §4)
#endif
#ifndef BHRAY_BVH_LDS_STACK
#define BHRAY_BVH_LDS_STACK 8      // entries of the short traversal stack in LDS (16 KB per 256-thread block: 8 blocks per CU fit in 160 KB)
#endif
// (The build flags of rounds 3-5's measured experiments - BHRAY_EXPERIMENT_*, BHRAY_BVH_PREFETCH, BHRAY_MESH_PARK / BHRAY_FLAT_COLD / BHRAY_BVH_WHILE_WHILE for the
// latency build - are gone from the source; what each measured is in profiles/EXPERIMENTS.md R5.5, R5.6, R5.8, R5.9.  The mesh variant's build for a saturated
// device, trace_kernel<.., MODELS = true, .., DENSE = true>, is the one that parks its marching state, marks the flat phase unlikely and traverses while-while.)
#ifndef BHRAY_THIN_STRIDED_BELOW
#define BHRAY_THIN_STRIDED_BELOW 32   // thin dealing of a whole frame: a wave's share is taken STRIDED (every waves-th entry) when it is below this many rays (see trace_kernel)
#endif
#ifndef BHRAY_MODEL_INLINE
#define BHRAY_MODEL_INLINE __forceinline__   // the traversal inline, in a kernel budgeted for 5 waves per SIMD (96 VGPRs): the step loop stays free of spills and what is
                                             // parked around a flat phase is 116-156 bytes per lane once per phase.  As a __noinline__ call (rounds 2-3, 8 waves, 64 VGPRs)
                                             // the calling convention saved 360 bytes per lane around EVERY call: 43 MB of scratch writes per 1080p launch.  Inline at 6 or 8
                                             // waves the spills move into the step loop (2 819 / 1 610 Mrays/s, R3.6).  profiles/EXPERIMENTS.md R4.3
#endif
struct BvhLds { int2* stack; };     // stack: this lane's column (entry k at stack[k * BHRAY_TRACE_THREADS])

// The traversal stack.  The reference stacks 19 whole nodes and has no overflow check (ray.wgsl:292,327); round 2 kept 64 two-word entries
// per lane in scratch - 512 bytes per lane, 42 MB of scratch writes per 1080p launch.  Now: a SHORT stack of BHRAY_BVH_LDS_STACK entries
// in LDS (a ring: a push onto a full ring overwrites the oldest - shallowest - entry) plus a RESTART TRAIL: two 64-bit masks, one bit per
// tree level - `pend` (the far child of the node visited at that level was pushed and has not been visited yet) and `wentfar` (the
// current path took the far child at that level).  A pop that finds the ring empty while `pend` has a bit set re-descends from the
// root along `wentfar` to the deepest pending level and takes its far child.  The re-descent repeats no DECISION: near / far order is a
// function of the ray and the boxes alone, and whether a far child is visited was decided when it was pushed (with the closest hit of
// that moment, as the reference decides it) and is recorded in `pend` - so the nodes are visited in the same order, with the same
// pruning, and equal-t ties between triangles resolve as in the reference.  Node pairs re-read on the way down are not counted.
// A tree deeper than 64 levels raises BHRAY_E_BVH_DEPTH (D2).
template <bool COUNT, bool WW>
__device__ BHRAY_MODEL_INLINE void trace_ray_model(const ModelDev& M, const BvhLds lds, F3 pos, F3 dir, float t_min, float t_max,
                                             Hit& closest, F3& normal_out, unsigned long long* cnt, int* err) {
    static_assert(BHRAY_BVH_LDS_STACK >= 2 && BHRAY_BVH_STACK <= 64, "short stack / trail sizes");
    constexpr int D = BHRAY_BVH_LDS_STACK;
    F3 mpos = ld3(M.pos);
    F3 inv = f3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    closest.hit = false; closest.t = t_max; closest.color = f3(0, 0, 0); closest.opacity = 0.0f;
    normal_out = f3(0, 0, 0);
    const int root_contents = __float_as_int(M.nodes[0].w), root_count = __float_as_int(M.nodes[1].w);     // nodes[0], untested root (ray.wgsl:291)
    int contents = root_contents, obj_count = root_count;
    unsigned long long pend = 0ull, wentfar = 0ull;
    int lev = 0;                      // tree level of the decision the current node's children are (root's children: 0)
    int sp = 0, held = 0;             // ring position; valid entries in the ring (<= D)
    int target = -1;                  // >= 0: re-descending to take the pending far child of that level
    if (WW) {
    // "while-while" form: every lane first walks inner nodes until it stands at a leaf (or is done), then the lanes that hold a leaf test
    // their triangles together.  In the one-loop form below a lane at a leaf (<= 2 triangles, ~300 instructions) and a lane at an inner node
    // (two box tests, ~80) take turns within every iteration; rays of one tile reach their leaves at different iterations, so most
    // iterations paid for both.  Every lane still visits its own nodes in its own order: same hits, same counters, same equal-t ties.
    bool alive = true;
    // the pop of the one-loop form: the next node for this lane, or the end of its traversal
    auto do_pop = [&]() {
        if (pend == 0ull) { alive = false; return; }
        const int k = 63 - __builtin_clzll(pend);               // the deepest level with a pending far child: the top of the stack, if it is still there
        if (held > 0) {
            sp--; held--;
            const int2 e = lds.stack[(sp % D) * BHRAY_TRACE_THREADS];
            contents = e.x; obj_count = e.y;
            pend &= ~(1ull << k); wentfar |= 1ull << k;
            lev = k + 1;
        } else {                                                // its entry was overwritten: find it again from the root
            target = k; contents = root_contents; obj_count = root_count; lev = 0;
        }
    };
    while (alive) {
        while (alive && obj_count == 0) {
            if (lev >= BHRAY_BVH_STACK) { *err = BHRAY_E_BVH_DEPTH; alive = false; break; }
            const float4* pair = M.nodes + 2 * (size_t)contents;
            const float4 a_lo = pair[0], a_hi = pair[1], b_lo = pair[2], b_hi = pair[3];
            float d1 = hit_aabb(pos, inv, a_lo, a_hi, mpos);
            float d2 = hit_aabb(pos, inv, b_lo, b_hi, mpos);
            int2 n1 = make_int2(__float_as_int(a_lo.w), __float_as_int(a_hi.w));
            int2 n2 = make_int2(__float_as_int(b_lo.w), __float_as_int(b_hi.w));
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; int2 tn = n1; n1 = n2; n2 = tn; }
            const unsigned long long bit = 1ull << lev;
            if (target >= 0) {                                      // re-descent: follow the recorded path, decide nothing
                if (lev < target) {
                    const int2 n = (wentfar & bit) ? n2 : n1;
                    contents = n.x; obj_count = n.y; lev++;
                } else {                                            // the pending far child itself
                    pend &= ~bit; wentfar |= bit;
                    contents = n2.x; obj_count = n2.y; lev++;
                    target = -1;
                }
                continue;
            }
            if (COUNT) cnt[6]++;
            if (d1 > closest.t) {
                do_pop();
            } else {
                contents = n1.x; obj_count = n1.y;
                wentfar &= ~bit;
                if (d2 < closest.t) {                               // the far child will be visited, whatever closest.t becomes (as in the reference)
                    pend |= bit;
                    lds.stack[(sp % D) * BHRAY_TRACE_THREADS] = n2;
                    sp++; held = held < D ? held + 1 : D;
                }
                lev++;
            }
        }
        if (alive) {
            for (int i = 0; i < obj_count; i++) {
                const float4* g = M.leaf + 6 * (size_t)(contents + i);
                const float4 A = g[0], B = g[1], Cc = g[2], N1 = g[3], N2 = g[4], N3 = g[5];
                if (COUNT) cnt[7]++;
                float t; F3 col, nrm;
                if (hit_triangle(pos, dir, t_min, t_max, f3(A.x, A.y, A.z) + mpos, f3(B.x, B.y, B.z) + mpos,
                                 f3(Cc.x, Cc.y, Cc.z) + mpos, f3(N1.x, N1.y, N1.z), f3(N2.x, N2.y, N2.z),
                                 f3(N3.x, N3.y, N3.z), t, col, nrm)) {
                    if (t < closest.t) { closest.hit = true; closest.t = t; closest.color = col; closest.opacity = 1.0f; normal_out = nrm; }
                }
            }
            do_pop();
        }
    }
    } else {
    for (;;) {
        bool pop = false;
        if (obj_count == 0) {
            if (lev >= BHRAY_BVH_STACK) { *err = BHRAY_E_BVH_DEPTH; break; }
            const float4* pair = M.nodes + 2 * (size_t)contents;
            const float4 a_lo = pair[0], a_hi = pair[1], b_lo = pair[2], b_hi = pair[3];
            float d1 = hit_aabb(pos, inv, a_lo, a_hi, mpos);
            float d2 = hit_aabb(pos, inv, b_lo, b_hi, mpos);
            int2 n1 = make_int2(__float_as_int(a_lo.w), __float_as_int(a_hi.w));
            int2 n2 = make_int2(__float_as_int(b_lo.w), __float_as_int(b_hi.w));
            if (d1 > d2) { float td = d1; d1 = d2; d2 = td; int2 tn = n1; n1 = n2; n2 = tn; }
            const unsigned long long bit = 1ull << lev;
            if (target >= 0) {                                      // re-descent: follow the recorded path, decide nothing
                if (lev < target) {
                    const int2 n = (wentfar & bit) ? n2 : n1;
                    contents = n.x; obj_count = n.y; lev++;
                } else {                                            // the pending far child itself
                    pend &= ~bit; wentfar |= bit;
                    contents = n2.x; obj_count = n2.y; lev++;
                    target = -1;
                }
                continue;
            }
            if (COUNT) cnt[6]++;
            if (d1 > closest.t) {
                pop = true;
            } else {
                contents = n1.x; obj_count = n1.y;
                wentfar &= ~bit;
                if (d2 < closest.t) {                               // the far child will be visited, whatever closest.t becomes (as in the reference)
                    pend |= bit;
                    lds.stack[(sp % D) * BHRAY_TRACE_THREADS] = n2;
                    sp++; held = held < D ? held + 1 : D;
                }
                lev++;
            }
        } else {
            for (int i = 0; i < obj_count; i++) {
                const float4* g = M.leaf + 6 * (size_t)(contents + i);
                const float4 A = g[0], B = g[1], Cc = g[2], N1 = g[3], N2 = g[4], N3 = g[5];
                if (COUNT) cnt[7]++;
                float t; F3 col, nrm;
                if (hit_triangle(pos, dir, t_min, t_max, f3(A.x, A.y, A.z) + mpos, f3(B.x, B.y, B.z) + mpos,
                                 f3(Cc.x, Cc.y, Cc.z) + mpos, f3(N1.x, N1.y, N1.z), f3(N2.x, N2.y, N2.z),
                                 f3(N3.x, N3.y, N3.z), t, col, nrm)) {
                    if (t < closest.t) { closest.hit = true; closest.t = t; closest.color = col; closest.opacity = 1.0f; normal_out = nrm; }
                }
            }
            pop = true;
        }
        if (pop) {
            if (pend == 0ull) break;
            const int k = 63 - __builtin_clzll(pend);               // the deepest level with a pending far child: the top of the stack, if it is still there
            if (held > 0) {
                sp--; held--;
                const int2 e = lds.stack[(sp % D) * BHRAY_TRACE_THREADS];
                contents = e.x; obj_count = e.y;
                pend &= ~(1ull << k); wentfar |= 1ull << k;
                lev = k + 1;
            } else {                                                // its entry was overwritten: find it again from the root
                target = k; contents = root_contents; obj_count = root_count; lev = 0;
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------
// integrators
// ------------------------------------------------------------------------------------------
// Correctly rounded 1/x and sqrt(x) for the step loop.  The compiler's IEEE lowering spends 11 (division) and 16 (sqrt)
// instructions, mostly on scaling for denormal inputs/results.  On gfx950 the hardware approximations are good enough that
//   1/x     = v_rcp_f32 + one Newton step in FMA                 for 2^-125 <= |x| < 2^126
//   sqrt(x) = v_rsq_f32 + one residual correction in FMA           for 2^-95 <= x <= 2^95
// equal the IEEE result for EVERY input in those ranges (verified exhaustively, all 2^32 bit patterns: bhray_selftest,
// tests/test_gpu_parity.py, profiles/ubench/exact_math.hip, exact_norm.hip).  Outside the range - decided per wave, so the branch is uniform -
// the IEEE lowering runs.  Same bits as `1.0f / x` and `sqrtf(x)`, 7-11 instructions less per use.
__device__ __forceinline__ float rcp_newton(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_corrected(float x) {
    // v_rsq_f32 (1 ulp) and one residual correction in FMA (Markstein): s0 = x*y, s = s0 + (x - s0*s0) * y/2
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float res = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(res, 0.5f * y, s0);
}
#ifndef BHRAY_INT_GUARDS
#define BHRAY_INT_GUARDS 1     // (A/B against the two-compare form, profiles/r06_ab_int_guards.txt: Euler +1.1 %, RK -0.2 %, one frame at a time 0 to -2 %)
#endif
#if BHRAY_INT_GUARDS
// The same ranges as ONE unsigned compare of the bit pattern (a subtraction and a compare instead of two compares and a scalar OR; no scalar operands): a negative
// operand or a NaN lands above the range and takes the IEEE lowering, which is exact for every input - the short sequences' results never are used outside
// [2^-125, 2^126) / [2^-95, 2^95] (bhray_selftest runs rcp_rn / sqrt_rn through these tests over all 2^32 patterns).
__device__ __forceinline__ bool rcp_in_range(float x) { return (__float_as_uint(x) - 0x01000000u) < 0x7d800000u; }      // 2^-125 <= x < 2^126 (positive x)
__device__ __forceinline__ bool sqrt_in_range(float x) { return (__float_as_uint(x) - 0x10000000u) <= 0x5f000000u; }    // 2^-95 <= x <= 2^95
#else
__device__ __forceinline__ bool rcp_in_range(float x) { return fabsf(x) >= 0x1p-125f && fabsf(x) < 0x1p126f; }
__device__ __forceinline__ bool sqrt_in_range(float x) { return x >= 0x1p-95f && x <= 0x1p95f; }
#endif
// The short sequence is computed unconditionally and REPLACED behind a wave-uniform, never-taken-in-practice branch: the result does
// not wait for the range test and the step has one not-taken branch per use instead of a diamond (two) - a wave that runs
// alone on its SIMD pays for every branch (profiles/ubench/lone_wave.hip).
__device__ __forceinline__ float rcp_rn(float x) {                 // == 1.0f / x
    float r = rcp_newton(x);
    if (__builtin_expect(__ballot(!rcp_in_range(x)) != 0ull, 0)) r = 1.0f / x;
    return r;
}
__device__ __forceinline__ float sqrt_rn(float x) {                // == sqrtf(x)
    float r = sqrt_corrected(x);
    if (__builtin_expect(__ballot(!sqrt_in_range(x)) != 0ull, 0)) r = sqrtf(x);
    return r;
}
// == fnormalize(a) (bhray_math.h): a * (1 / sqrt(fdot(a, a))); one guard covers both (sqrt of an in-range x is in rcp's range)
__device__ __forceinline__ F3 fnormalize_rn(F3 a) {
    const float d = fdot(a, a);
    float r = rcp_newton(sqrt_corrected(d));
    if (__builtin_expect(__ballot(!sqrt_in_range(d)) != 0ull, 0)) r = 1.0f / sqrtf(d);
    return a * r;
}
// == bh_pow_m001(x) (bhray_math.h) for every x > 0.00002f, the only values next_ray_rk passes (all of them, +inf included, checked
// by bhray_selftest).  The portable form's NaN / negative / zero / denormal arms are dead there, +inf becomes a select, and the
// quotient (m - 1) / (m + 1), m + 1 in [1.70, 2.42], is the short correctly rounded sequence: reciprocal, product, one residual
// correction: 63 -> 46 instructions.  (The arm is rare - 0.3 % of the wave-steps of a 1080p frame, counted in round 2 - so what
// its length buys is code layout: +0.6 % at saturation, measured.)
__device__ __forceinline__ float pow_m001_step(float x) {
    const uint32_t u = f2u(x);
    int e = (int)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }
    const float num = m - 1.0f, den = m + 1.0f;
    const float r = rcp_newton(den);
    const float q0 = num * r;
    const float s = __builtin_fmaf(__builtin_fmaf(-den, q0, num), r, q0);
    const float s2 = s * s;
    float p = 0.111111112f;
    p = p * s2 + 0.142857149f;
    p = p * s2 + 0.2f;
    p = p * s2 + 0.333333343f;
    p = p * s2 + 1.0f;
    const float lnm = (2.0f * s) * p;
    const float lnx = (float)e * 0.693147182f + lnm;
    const float t = -0.001f * lnx;
    float q = 0.00138888892f;
    q = q * t + 0.00833333377f;
    q = q * t + 0.0416666679f;
    q = q * t + 0.166666672f;
    q = q * t + 0.5f;
    q = q * t + 1.0f;
    q = q * t + 1.0f;
    return x == u2f(0x7f800000u) ? 0.0f : q;
}
__device__ __forceinline__ float fdistance_rn(F3 a, F3 b) { const F3 v = a - b; return sqrt_rn(fdot(v, v)); }   // == fdistance(a, b)

__device__ __forceinline__ float pow5(float d) { return ((d * d) * (d * d)) * d; }

// max_(max_(|x|, |y|), |z|) under N5 (compare-select: a < b ? b : a) in three instructions instead of five: t = the IEEE maximum of |y| and |z|
// (v_max_f32: the operand that is not a NaN, a NaN only when both are), then |x| < t ? t : |x|.  Equal to the compare-select nest for EVERY input:
// x NaN -> both compares of the nest are false -> |x| (here: the compare is false -> |x|); y or z NaN with x a number -> the nest skips the NaN
// (so does the maximum); no NaN -> the maximum.  (bhray_selftest: 16^3 triples of special values and 2^20 random bit patterns, bit for bit, on the device.)
__device__ __forceinline__ float max3_abs(float x, float y, float z) {
    float t; asm("v_max_f32 %0, |%1|, |%2|" : "=v"(t) : "v"(y), "v"(z));
    const float ax = fabsf(x);
    return ax < t ? t : ax;
}
// cd < closest ? cd : closest in one instruction: v_min_f32(cd, closest) - the IEEE minimum returns the operand that is not a NaN, which is what the select
// does when cd is a NaN; `closest` never is one (it starts at the sphere's radius and only ever takes numbers); distances carry no negative zero.
__device__ __forceinline__ float closest_min(float cd, float closest) {
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(cd), "v"(closest));
    return r;
}

// f, ray.wgsl:401-403, under N9: f(p) = (p - bh) * s with the per-step scalar s = (-1.5*h2) * (1/dist^5); positions are kept
// relative to the hole (q = p - bh), so a stage is fma(sum, h, q0) * s.
// Cash–Karp tableau, ray.wgsl:133-165: untyped consts are evaluated in binary64 and rounded once.
#define KF(x) ((float)(x))
__device__ constexpr float A21 = KF(1.0 / 5.0);
__device__ constexpr float A31 = KF(3.0 / 40.0), A32 = KF(9.0 / 40.0);
__device__ constexpr float A41 = KF(3.0 / 10.0), A42 = KF(-9.0 / 10.0), A43 = KF(6.0 / 5.0);
__device__ constexpr float A51 = KF(-11.0 / 54.0), A52 = KF(5.0 / 2.0), A53 = KF(-70.0 / 27.0), A54 = KF(35.0 / 27.0);
__device__ constexpr float A61 = KF(1631.0 / 55296.0), A62 = KF(175.0 / 512.0), A63 = KF(575.0 / 13824.0),
                           A64 = KF(44275.0 / 110592.0), A65 = KF(253.0 / 4096.0);
__device__ constexpr float BA1 = KF(2825.0 / 27648.0), BA3 = KF(18575.0 / 48384.0),
                           BA4 = KF(13525.0 / 55296.0), BA5 = KF(277.0 / 14336.0), BA6 = KF(1.0 / 4.0);
__device__ constexpr float DB1 = KF(37.0 / 378.0 - 2825.0 / 27648.0),
                           DB3 = KF(250.0 / 621.0 - 18575.0 / 48384.0), DB4 = KF(125.0 / 594.0 - 13525.0 / 55296.0),
                           DB5 = KF(0.0 - 277.0 / 14336.0), DB6 = KF(512.0 / 1771.0 - 1.0 / 4.0);

// next_ray_rk, ray.wgsl:405-465.  The retry loop (425-451) cannot change h and is run once.  N7: fused arithmetic.
// `dist` = flength(pos - bpos), carried from the previous step's exit test (same operands, same value).
// (A form with the (x, y) components of every 3-vector operation in one packed instruction was built and measured: faster in isolation,
// slower in this kernel - a packed instruction takes no literal, and the kernel has no SGPRs left for the 25 coefficients: EXPERIMENTS.md R3.13.)
// (`_to`: the new position goes to a register set of its own - the unified march of bhray_step_u.inc alternates two, so that "previous = current" is renaming)
// (FAST: the short 1/x and sqrt sequences WITHOUT their range tests and the step size's common arm only; the return value says for which lanes that is
// not the exact step - an operand outside a sequence's range, or an error estimate above the threshold: the one-test step of bhray_step_u.inc runs the exact
// form again for those, behind the test every step has anyway.  A wave that runs alone pays ~35 cycles for every test of its own - profiles/EXPERIMENTS.md R6.12.)
template <bool FAST>
__device__ __forceinline__ bool next_ray_rk_t(const F3 q0, const F3 p0, F3& pos, F3& dir, float& h_io, float dist) {
    const F3 d0 = dir;                     // q0 = p0 - bpos (N9), carried by the caller together with dist = flength(q0)
    const F3 cr = fcross(p0, d0);
    const float h2 = fdot(cr, cr);                   // N3: pow(length(v), 2.0) = dot(v, v)
    const float d5 = pow5(dist);
    const float s = (-1.5f * h2) * (FAST ? rcp_newton(d5) : rcp_rn(d5));   // N9
    const float h = h_io;
    // N10: K_i = h*k_i = (q0 + sum a_ij K_j) * (s*h); zero-coefficient terms (b_2, b*_2) dropped
    const float sh = s * h;
    const F3 K1 = q0 * sh;
    const F3 K2 = fmadd3(K1, A21, q0) * sh;
    const F3 K3 = fmadd3(K2, A32, fmadd3(K1, A31, q0)) * sh;
    const F3 K4 = fmadd3(K2, A43, fmadd3(K2, A42, fmadd3(K1, A41, q0))) * sh;
    const F3 K5 = fmadd3(K4, A54, fmadd3(K3, A53, fmadd3(K2, A52, fmadd3(K1, A51, q0)))) * sh;
    const F3 K6 = fmadd3(K5, A65, fmadd3(K4, A64, fmadd3(K3, A63, fmadd3(K2, A62, fmadd3(K1, A61, q0))))) * sh;
    const F3 e = fmadd3(K6, DB6, fmadd3(K5, DB5, fmadd3(K4, DB4, fmadd3(K3, DB3, K1 * DB1))));
    const float e_max = max3_abs(e.x, e.y, e.z);
    // the small terms are summed first and added to the unit-length direction once (one rounding at magnitude 1)
    const F3 ds = fmadd3(K6, BA6, fmadd3(K5, BA5, fmadd3(K4, BA4, fmadd3(K3, BA3, K1 * BA1))));
    if constexpr (FAST) {
        const F3 v = d0 + ds;
        const float nn = fdot(v, v);
        dir = v * rcp_newton(sqrt_corrected(nn));
        pos = fmadd3(d0, h, p0);
        h_io = h * 1.0001f;
        return !rcp_in_range(d5) | !sqrt_in_range(nn) | (e_max > 0.00002f);
    } else {
        dir = fnormalize_rn(d0 + ds);
        pos = fmadd3(d0, h, p0);
        if (e_max > 0.00002f) h_io = h * (0.9f * pow_m001_step(e_max));
        else h_io = h * 1.0001f;
        return false;
    }
}
__device__ __forceinline__ void next_ray_rk_to(const F3 q0, const F3 p0, F3& pos, F3& dir, float& h_io, float dist) { (void)next_ray_rk_t<false>(q0, p0, pos, dir, h_io, dist); }
__device__ __forceinline__ void next_ray_rk(F3 q0, F3& pos, F3& dir, float& h_io, float dist) {
    const F3 p0 = pos;
    next_ray_rk_to(q0, p0, pos, dir, h_io, dist);
}

// next_ray_euler, ray.wgsl:467-480 (N7, N9).
template <bool FAST>
__device__ __forceinline__ bool next_ray_euler_t(const F3 q0, const F3 p0, F3& pos, F3& dir, float step, float dist) {
    const F3 cr = fcross(p0, dir);
    const float h2 = fdot(cr, cr);                   // N3: pow(length(v), 2.0) = dot(v, v)
    const float d5 = pow5(dist);
    const float s = (-1.5f * h2) * (FAST ? rcp_newton(d5) : rcp_rn(d5));
    bool bad = false;
    if constexpr (FAST) {
        const F3 v = fmadd3(q0, s * step, dir);
        const float nn = fdot(v, v);
        dir = v * rcp_newton(sqrt_corrected(nn));
        bad = !rcp_in_range(d5) | !sqrt_in_range(nn);
    } else {
        dir = fnormalize_rn(fmadd3(q0, s * step, dir));   // N9, N10
    }
    pos = fmadd3(dir, step, p0);
    return bad;
}
__device__ __forceinline__ void next_ray_euler_to(const F3 q0, const F3 p0, F3& pos, F3& dir, float step, float dist) { (void)next_ray_euler_t<false>(q0, p0, pos, dir, step, dist); }
__device__ __forceinline__ void next_ray_euler(F3 q0, F3& pos, F3& dir, float step, float dist) {
    const F3 p0 = pos;
    next_ray_euler_to(q0, p0, pos, dir, step, dist);
}

// The LITERAL reading of the integrator (BHRAY_F_LITERAL, EVAL == 1): ray.wgsl:401-480 operator by operator under N0-N2 — one
// binary32 operation per WGSL operator in source order, no fused multiply-add, no reassociation; pow(d, 5) = ((d*d)*(d*d))*d,
// pow(l, 2) = l*l, pow(e, -0.001) the portable form.  Bit-identical to oracle_set_literal(1) and to tests/golden/frames_literal.npz:
// the variant exists so that "the contract (N3/N7/N9/N10) is a permitted evaluation of the shader text" is a measured distance
// between two kernels, not prose.  Tuned this round without changing a bit: the correctly rounded 1/x and sqrt(x) are the short
// gfx950 sequences of N8 (rcp_rn / sqrt_rn: equal to the IEEE lowering on every input, bhray_selftest), the functions are inlined,
// the reciprocal of dist^5 - the same value in all six evaluations of f - is formed once, and the step-size power is the
// domain-specialised form (pow_m001_step == bh_pow_m001 on its whole domain).
__device__ __forceinline__ float length_rn(F3 v) { return sqrt_rn(dot(v, v)); }                       // == length(v)
__device__ __forceinline__ F3 normalize_rn(F3 v) {                                                    // == normalize(v) = v * (1 / length(v))
    const float d = dot(v, v);
    float r = rcp_newton(sqrt_corrected(d));
    if (__builtin_expect(__ballot(!sqrt_in_range(d)) != 0ull, 0)) r = 1.0f / sqrtf(d);
    return v * r;
}
__device__ __forceinline__ F3 f_literal(F3 p, F3 bpos, float h2, float inv_d5) {          // fn f, ray.wgsl:401-403: num / pow(dist, 5) = num * (1 / dist^5) (N2)
    const F3 num = (p - bpos) * (-1.5f * h2);
    return num * inv_d5;
}
__device__ constexpr float DB2 = KF(0.0 - 0.0), BA2 = KF(0.0);
__device__ __forceinline__ void next_ray_rk_literal(F3 bpos, F3& pos, F3& dir, float& h_io) {   // ray.wgsl:405-465 (D1: loop once)
    const F3 p0 = pos, d0 = dir;
    const float dist = length_rn(p0 - bpos);
    const float lc = length_rn(cross(p0, d0));
    const float h2 = lc * lc;
    const float h = h_io;
    const float i5 = rcp_rn(pow5(dist));
    const F3 k1 = f_literal(p0, bpos, h2, i5);
    const F3 k2 = f_literal(p0 + (k1 * A21) * h, bpos, h2, i5);
    const F3 k3 = f_literal(p0 + (k1 * A31 + k2 * A32) * h, bpos, h2, i5);
    const F3 k4 = f_literal(p0 + ((k1 * A41 + k2 * A42) + k2 * A43) * h, bpos, h2, i5);                         // a_43*k_2 (sic)
    const F3 k5 = f_literal(p0 + (((k1 * A51 + k2 * A52) + k3 * A53) + k4 * A54) * h, bpos, h2, i5);
    const F3 k6 = f_literal(p0 + ((((k1 * A61 + k2 * A62) + k3 * A63) + k4 * A64) + k5 * A65) * h, bpos, h2, i5);
    const F3 es = ((((k1 * DB1 + k2 * DB2) + k3 * DB3) + k4 * DB4) + k5 * DB5) + k6 * DB6;
    const F3 e = es * h;
    const float e_max = max_(max_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));
    const F3 ds = ((((k1 * BA1 + k2 * BA2) + k3 * BA3) + k4 * BA4) + k5 * BA5) + k6 * BA6;
    dir = normalize_rn(d0 + ds * h);
    pos = p0 + d0 * h;                                                                                           // old direction
    if (e_max > 0.00002f) h_io = h * (0.9f * pow_m001_step(e_max));
    else h_io = h * 1.0001f;
}
__device__ __forceinline__ void next_ray_euler_literal(F3 bpos, F3& pos, F3& dir, float step) {   // ray.wgsl:467-480
    const float lc = length_rn(cross(pos, dir));
    const float h2 = lc * lc;
    const float dist = length_rn(pos - bpos);
    dir = normalize_rn(dir + f_literal(pos, bpos, h2, rcp_rn(pow5(dist))) * step);
    pos = pos + dir * step;
}

// A THIRD evaluation (BHRAY_F_EVAL_FMA, EVAL == 2): the literal expression tree with fused multiply-add contraction ONLY - every
// `x*y + z` of the text whose product is a direct operand of the addition is one fma, the first product of a sum of products stays
// rounded - and none of the reassociations N9 / N10 (no per-step scalar, no step size folded into the stages, zero-coefficient
// terms kept).  What a shader compiler's default contraction does to ray.wgsl:401-480, and nothing more.  Bit-identical to
// oracle_set_eval(2).  It exists to show that the pixels on which the contract differs from the literal text by more than 1e-4
// are the pixels on which ANY two legal evaluations differ (tests/test_gpu_literal.py).
__device__ __forceinline__ void next_ray_rk_fma(F3 bpos, F3& pos, F3& dir, float& h_io) {
    const F3 p0 = pos, d0 = dir;
    const float dist = sqrt_rn(fdot(p0 - bpos, p0 - bpos));
    const F3 cr = fcross(p0, d0);
    const float lc = sqrt_rn(fdot(cr, cr));
    const float h2 = lc * lc;
    const float h = h_io;
    const float i5 = rcp_rn(pow5(dist));
    const F3 k1 = f_literal(p0, bpos, h2, i5);
    const F3 k2 = f_literal(fmadd3(k1 * A21, h, p0), bpos, h2, i5);
    const F3 k3 = f_literal(fmadd3(fmadd3(k2, A32, k1 * A31), h, p0), bpos, h2, i5);
    const F3 k4 = f_literal(fmadd3(fmadd3(k2, A43, fmadd3(k2, A42, k1 * A41)), h, p0), bpos, h2, i5);
    const F3 k5 = f_literal(fmadd3(fmadd3(k4, A54, fmadd3(k3, A53, fmadd3(k2, A52, k1 * A51))), h, p0), bpos, h2, i5);
    const F3 k6 = f_literal(fmadd3(fmadd3(k5, A65, fmadd3(k4, A64, fmadd3(k3, A63, fmadd3(k2, A62, k1 * A61)))), h, p0), bpos, h2, i5);
    const F3 es = fmadd3(k6, DB6, fmadd3(k5, DB5, fmadd3(k4, DB4, fmadd3(k3, DB3, fmadd3(k2, DB2, k1 * DB1)))));
    const F3 e = es * h;
    const float e_max = max_(max_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));
    const F3 ds = fmadd3(k6, BA6, fmadd3(k5, BA5, fmadd3(k4, BA4, fmadd3(k3, BA3, fmadd3(k2, BA2, k1 * BA1)))));
    dir = fnormalize_rn(fmadd3(ds, h, d0));
    pos = fmadd3(d0, h, p0);
    if (e_max > 0.00002f) h_io = h * (0.9f * pow_m001_step(e_max));
    else h_io = h * 1.0001f;
}
__device__ __forceinline__ void next_ray_euler_fma(F3 bpos, F3& pos, F3& dir, float step) {
    const F3 cr = fcross(pos, dir);
    const float lc = sqrt_rn(fdot(cr, cr));
    const float h2 = lc * lc;
    const float dist = sqrt_rn(fdot(pos - bpos, pos - bpos));
    dir = fnormalize_rn(fmadd3(f_literal(pos, bpos, h2, rcp_rn(pow5(dist))), step, dir));
    pos = fmadd3(dir, step, pos);
}

// number of set bits of a wave mask below this lane (prefix popcount): v_mbcnt_lo + v_mbcnt_hi, no per-lane mask registers
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Append to a work queue from a whole block with ONE atomic: every wave counts its entries (ballot), the counts are summed through
// LDS, wave 0 reserves the block's range, every wave writes its entries at its offset.  An atomic on one word costs ~12 ns whoever
// issues it, so one per wave made the classification of a 1080p level (32 k waves) an atomic-bound 75 us for 25 us of memory work -
// and, with 16-20 frames in flight, held the whole device back: one atomic per 4 waves is worth +8 % frames per second (4 900 ->
// 5 300 Mrays/s); 16 waves per block (1 024 threads) classify fastest alone but find no room beside the persistent trace waves.
// Geometry of a classify / predict block: 4 waves, each handling BHRAY_CLASSIFY_TPW 8x8-pixel tiles one after the other; the
// block's BX x BY tiles form a rectangle of the level, and their queue entries are appended together, tile after tile in row-major
// order of the rectangle.  Rays that are neighbours in the frame are then neighbours in the queue over a 2-D neighbourhood, and a
// trace wave (64 consecutive entries, refills of a few consecutive entries) holds rays of similar length that take the rare paths
// at the same steps.
#define BHRAY_CLASSIFY_WAVES 4
#define BHRAY_CLASSIFY_THREADS (64 * BHRAY_CLASSIFY_WAVES)
#define BHRAY_CLASSIFY_TILES (BHRAY_CLASSIFY_BX * BHRAY_CLASSIFY_BY)
#define BHRAY_CLASSIFY_TPW (BHRAY_CLASSIFY_TILES / BHRAY_CLASSIFY_WAVES)
static_assert(BHRAY_CLASSIFY_TILES % BHRAY_CLASSIFY_WAVES == 0 && BHRAY_CLASSIFY_TILES <= 64, "classify block geometry");
// pixel of lane `lane` in the t-th tile of wave `wave` of block `block`: level column x, index j into L.rows; false outside the level's region
__device__ __forceinline__ bool classify_pixel(const LevelParams& L, int block, int wave, int t, int lane, int& x, int& j) {
    const int tiles_x = (L.x1 - L.x0 + 7) >> 3;
    const int blocks_x = (tiles_x + BHRAY_CLASSIFY_BX - 1) / BHRAY_CLASSIFY_BX;
    const int q = t * BHRAY_CLASSIFY_WAVES + wave;                       // tile of the block, row-major in its BX x BY rectangle
    const int tx = (block % blocks_x) * BHRAY_CLASSIFY_BX + q % BHRAY_CLASSIFY_BX;
    const int ty = (block / blocks_x) * BHRAY_CLASSIFY_BY + q / BHRAY_CLASSIFY_BX;
    x = L.x0 + tx * 8 + (lane & 7);
    j = ty * 8 + (lane >> 3);
    return tx < tiles_x && x < L.x1 && j < L.nrows;
}
// Append the block's entries with ONE atomic: every wave counts the entries of each of its tiles (ballot), the counts are summed
// through LDS, wave 0 reserves the block's range, every wave writes its entries at its tiles' offsets.  An atomic on one word
// costs ~12 ns whoever issues it: one per wave made the classification of a 1080p level (32 k waves) an atomic-bound 75 us for 25 us
// of memory work and, with 16-20 frames in flight, held the whole device back.
__device__ __forceinline__ void block_append(const bool (&want)[BHRAY_CLASSIFY_TPW], const uint32_t (&entry)[BHRAY_CLASSIFY_TPW],
                                             uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount, uint32_t* lds /* [TILES] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m[BHRAY_CLASSIFY_TPW];
#pragma unroll
    for (int t = 0; t < BHRAY_CLASSIFY_TPW; t++) {
        m[t] = __ballot(want[t]);
        if (lane == 0) lds[t * BHRAY_CLASSIFY_WAVES + wave] = (uint32_t)__popcll(m[t]);
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t v = lane < BHRAY_CLASSIFY_TILES ? lds[lane] : 0u, incl = v;
#pragma unroll
        for (int off = 1; off < BHRAY_CLASSIFY_TILES; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += o; }
        const uint32_t total = (uint32_t)__shfl((int)incl, BHRAY_CLASSIFY_TILES - 1);
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(qcount, total);
        base = (uint32_t)__shfl((int)base, 0);
        if (lane < BHRAY_CLASSIFY_TILES) lds[lane] = base + incl - v;                 // first slot of tile `lane`
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < BHRAY_CLASSIFY_TPW; t++)
        if (want[t]) queue[lds[t * BHRAY_CLASSIFY_WAVES + wave] + lanes_below(m[t])] = entry[t];
}

// ------------------------------------------------------------------------------------------
// classify: ray.wgsl:167-243
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 load_prev(const LevelParams& L, int x, int y) {
    x = x < 0 ? 0 : (x > L.pw - 1 ? L.pw - 1 : x);
    y = y < 0 ? 0 : (y > L.ph - 1 ? L.ph - 1 : y);
    return L.prev[(size_t)y * (size_t)L.pw + (size_t)x];
}
// angle_between(a, b) < threshold, ray.wgsl:263-267 + 222-226.  bh_acos is monotone non-increasing over all binary32 values of
// [-1, 1] (checked exhaustively: bhray_selftest), so  bh_acos(c) < thr  <=>  c > cstar  for the host-computed
// cstar = largest c with bh_acos(c) >= thr (FrameParams::acos_cstar); c outside [-1, 1] or NaN gives acos = NaN = "not smaller",
// as before.  Same decisions, 4 x ~60 instructions less per classified pixel; the square roots use the exact short sequence (N8).
__device__ __forceinline__ float angle_cosine(float4 a, float4 b) {
    F3 v1 = f3(a.x, a.y, a.z), v2 = f3(b.x, b.y, b.z);
    float d = dot(v1, v2);
    return d / (sqrt_rn(dot(v1, v1)) * sqrt_rn(dot(v2, v2)));
}
__device__ __forceinline__ bool cosine_below_threshold(float c, float cstar) { return c > cstar && c <= 1.0f; }
__device__ __forceinline__ size_t out_index(const LevelParams& L, int x, int y) {
    const int oy = L.rowmap ? L.rowmap[y] : y;
    return (size_t)oy * (size_t)L.out_pitch + (size_t)(x - L.out_x0);
}

// FIXUP: the exact per-level classification of the temporal mode (LevelParams::pass == CLASSIFY_FIXUP), its own instantiation so that
// the other passes keep their short-circuited angle tests (the merged kernel cost the saturated pass 2 %).
template <bool COUNT, bool FIXUP>
__global__ __launch_bounds__(BHRAY_CLASSIFY_THREADS) void classify_kernel(const FrameParams* __restrict__ Pb, const FrameLaunch* __restrict__ Fb) {
    __shared__ uint32_t append_lds[BHRAY_CLASSIFY_TILES];
    const FrameParams& P = Pb[blockIdx.y];
    const FrameLaunch& F = Fb[blockIdx.y];
    const LevelParams& L = F.L;
    uint32_t* __restrict__ queue = F.queue;
    uint32_t* __restrict__ qcount = F.qctl;
    Counters64* __restrict__ counters = F.counters;
    // one wave = one 8x8 tile at a time, BHRAY_CLASSIFY_TPW tiles per wave (classify_pixel)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    bool want[BHRAY_CLASSIFY_TPW];
    uint32_t entry[BHRAY_CLASSIFY_TPW];
    unsigned long long cnt_valid = 0, cnt_copy = 0, cnt_interp = 0;
#pragma unroll
    for (int t = 0; t < BHRAY_CLASSIFY_TPW; t++) {
    int x, j;
    const bool valid = classify_pixel(L, (int)blockIdx.x, wave, t, lane, x, j);
    const int y = valid ? L.rows[j] : 0;
    bool need_trace = false;
    bool near_trace = false;                         // interpolated, but within the temporal margin of being traced
    int kind = -1;                                   // 0 copy, 1 interpolate, 2 trace
    if (valid) {
        if (L.pw == 1 && L.ph == 1) {
            need_trace = true; kind = 2;
        } else {
            const float ppx = (float)x * L.rx, ppy = (float)y * L.ry;
            const float tlx = floorf(ppx), tly = floorf(ppy);
            const float4 c_tl = load_prev(L, (int)tlx, (int)tly);
            const bool tentative = L.pass == CLASSIFY_TENTATIVE;
            const float4 pending = make_float4(0.0f, 0.0f, 0.0f, BHRAY_PENDING_ALPHA);
            if (fabsf(tlx - ppx) < 0.001f && fabsf(tly - ppy) < 0.001f) {
                if (!L.no_store) L.out[out_index(L, x, y)] = c_tl;       // tentative: a PENDING source stays PENDING; the KEEP pass copies the final value
                kind = 0;
            } else {
                const float4 c_bl = load_prev(L, (int)tlx, (int)(tly + 1.0f));
                const float4 c_tr = load_prev(L, (int)(tlx + 1.0f), (int)tly);
                const float4 c_br = load_prev(L, (int)(tlx + 1.0f), (int)(tly + 1.0f));
                const bool alphas0 = c_tl.w == 0.0f && c_tr.w == 0.0f && c_bl.w == 0.0f && c_br.w == 0.0f;
                bool interp = false;
                if (alphas0) {                       // a PENDING neighbour (alpha 2) fails this test: the pixel is queued, conservatively
                    const float cs = P.acos_cstar;
                    if (FIXUP) {      // temporal mode also wants to know how close an interpolated pixel came to being traced
                        const float cn = P.acos_cstar_near;
                        const float c0 = angle_cosine(c_bl, c_tl), c1 = angle_cosine(c_br, c_tr), c2 = angle_cosine(c_tl, c_tr), c3 = angle_cosine(c_bl, c_br);
                        interp = cosine_below_threshold(c0, cs) && cosine_below_threshold(c1, cs) && cosine_below_threshold(c2, cs) && cosine_below_threshold(c3, cs);
                        near_trace = interp && !(cosine_below_threshold(c0, cn) && cosine_below_threshold(c1, cn) && cosine_below_threshold(c2, cn) && cosine_below_threshold(c3, cn));
                    } else {
                        interp = cosine_below_threshold(angle_cosine(c_bl, c_tl), cs) && cosine_below_threshold(angle_cosine(c_br, c_tr), cs) &&
                                 cosine_below_threshold(angle_cosine(c_tl, c_tr), cs) && cosine_below_threshold(angle_cosine(c_bl, c_br), cs);
                    }
                }
                if (interp) {
                    const float tx_ = ppx - tlx, ty_ = ppy - tly;
                    F3 top = mix3(f3(c_tl.x, c_tl.y, c_tl.z), f3(c_tr.x, c_tr.y, c_tr.z), tx_);
                    F3 bot = mix3(f3(c_bl.x, c_bl.y, c_bl.z), f3(c_br.x, c_br.y, c_br.z), tx_);
                    F3 p = mix3(top, bot, ty_);
                    if (!L.no_store) L.out[out_index(L, x, y)] = make_float4(p.x, p.y, p.z, 0.0f);
                    kind = 1;
                } else {
                    need_trace = true; kind = 2;
                    if (tentative && !L.no_store) L.out[out_index(L, x, y)] = pending;
                }
            }
        }
        if (need_trace && L.pass == CLASSIFY_KEEP) need_trace = false;      // traced by the superset launch: leave the pixel as it is
        if (need_trace && L.spec) {                  // speculative mode: the traced value already exists
            L.out[out_index(L, x, y)] = L.spec[(size_t)y * (size_t)L.w + (size_t)x];
            need_trace = false;
        }
    }
    if (FIXUP) {
        // temporal speculation: remember, per pixel, whether the shader traces it (the next frame's prediction is built from these
        // marks, predict_kernel), and send to this frame's own queue only what the predicted launch has not delivered (stamp)
        if (valid) F.need[(size_t)y * (size_t)L.w + (size_t)x] = (need_trace || near_trace) ? 1 : 0;
        if (need_trace && F.stamp[(size_t)y * (size_t)L.w + (size_t)x] == F.stamp_value) need_trace = false;
    }
    want[t] = need_trace;
    entry[t] = ((uint32_t)L.tag << 30) | ((uint32_t)y << 15) | (uint32_t)x;
    if (COUNT) { cnt_valid += __popcll(__ballot(valid)); cnt_copy += __popcll(__ballot(kind == 0)); cnt_interp += __popcll(__ballot(kind == 1)); }
    }   // tiles of this wave
    // block-wide compaction: one atomic per block, tiles, waves and lanes keep their order (a pass that queues nothing has no queue)
    if (queue) block_append(want, entry, queue, qcount, append_lds);
    if (COUNT && lane == 0) {
        atomicAdd(&counters->v[0], cnt_valid);
        atomicAdd(&counters->v[1], cnt_copy);
        atomicAdd(&counters->v[2], cnt_interp);
    }
}


// ------------------------------------------------------------------------------------------
// trace: ray.wgsl:269-285 + 482-596
// ------------------------------------------------------------------------------------------
enum : int { M_EMPTY = 0, M_REL = 1, M_FLAT = 2, M_FINISH = 3, M_SHADE_REL = 4, M_SHADE_FLAT = 5 };   // M_SHADE_x: a disk hit waits for its shading, then continues in mode x
#ifndef BHRAY_THIN_WAVES
#define BHRAY_THIN_WAVES 1024    // latency build: a short queue is dealt out evenly over this many waves (MI355X: 256 CUs x 4 SIMDs); 0 = off
#endif
#ifndef BHRAY_REL_BATCH
#define BHRAY_REL_BATCH 16       // integrator steps between refill / flat / epilogue phases
#endif
#ifndef BHRAY_REL_BATCH_EULER_DENSE
#define BHRAY_REL_BATCH_EULER_DENSE BHRAY_REL_BATCH   // ... of the dense Euler kernel (its step is less than half the RK step's instructions: the phases between batches weigh more)
#endif
#ifndef BHRAY_FLAT_MIN_LANES
#define BHRAY_FLAT_MIN_LANES 48    // mesh variant: run the flat/BVH phase when this many lanes wait for it ...
#endif
#ifndef BHRAY_FLAT_DEFER
#define BHRAY_FLAT_DEFER 64        // ... or after this many rounds at the latest (measured 12/4: 2735, 48/64: 2890 Mrays/s)
#endif
#ifndef BHRAY_TRACE_WAVES_MESH
#define BHRAY_TRACE_WAVES_MESH 5 // mesh variant (BVH traversal inline + short stack in LDS): 96 VGPRs.  Round 4, mesh workload, 20- / 200-frame blocks / one frame at a
                                 // time: call + 8 waves 3 747 / 4 323 Mrays/s / 3.05 ms; inline + 5 waves 3 937 / 4 359 / 2.60; inline + 4 waves 3 660 / 3 995 / 2.58
#endif
#ifndef BHRAY_TRACE_WAVES_MESH_DENSE
#define BHRAY_TRACE_WAVES_MESH_DENSE 6   // the mesh variant for a saturated device (round 5): 80 VGPRs like the dense no-mesh kernel - possible because the traversal runs in a region of
                                         // its own with the marching state stored to scratch around it (MESH_PARK below), the flat phase is marked unlikely (the allocator then keeps
                                         // the step loop free of spills) and the traversal is the while-while form (fewer registers).  400-frame blocks 4 434 -> 4 646 Mrays/s, 20-frame
                                         // blocks 3 941 -> 3 915; one frame at a time keeps the 5-wave build (profiles/EXPERIMENTS.md R5.8)
#endif
#ifndef BHRAY_TRACE_WAVES
#define BHRAY_TRACE_WAVES 4      // waves per SIMD the trace kernel is register-budgeted for (<=128 VGPRs): the latency build
#endif
// The no-mesh kernel is built twice.  Budgeted for 4 waves per SIMD the compiler uses ~99 VGPRs and keeps more of the step's
// independent chains in flight: a wave that runs alone on its SIMD (coarse ladder levels, one frame in flight, row tiles of a
// multi-GPU frame) steps 7-15 % faster.  Budgeted for 6 waves (80 VGPRs, no spills) a saturated device gains 4 % (more
// co-resident launches).  The host picks per ctx (bhray_api.hip).
#ifndef BHRAY_TRACE_WAVES_DENSE
#define BHRAY_TRACE_WAVES_DENSE 6
#endif
#ifndef BHRAY_UNIFIED
#define BHRAY_UNIFIED 1         // the no-mesh contract kernels march in pairs of steps over two position register sets (bhray_step_u.inc): 16 -> 3 register moves per step
#endif
#ifndef BHRAY_ONE_TEST
#define BHRAY_ONE_TEST 1        // bit 0: the latency builds, bit 1: the dense builds - the unified march's step with ONE test: the range tests of its short 1/x and sqrt sequences and the
#endif                          // step-size power's arm folded into the rare-path test every step has (a lane for which they matter runs the exact step again behind it); bit 2 (tests): lanes flagged at random
#ifndef BHRAY_PNUMER_EARLY
#define BHRAY_PNUMER_EARLY 1
#endif
#ifndef BHRAY_ORIGIN_PATH
#define BHRAY_ORIGIN_PATH 1      // bit 0: Euler, bit 1: RK - a second copy of the unified pairs for a hole at the scene's origin (no position - bpos per step): Euler +1.7 %; RK -1.8 % (the doubled loop costs its kernel 48 bytes of scratch in the phases): Euler only
#endif
#ifndef BHRAY_UNIFIED_MESH
#define BHRAY_UNIFIED_MESH 1    // ... and the mesh variant's kernels too (RK +1.5-2.2 %, Euler +3.5 % on configs[2]; profiles/EXPERIMENTS.md R6.10)
#endif
#ifndef BHRAY_REFILL_MIN
#define BHRAY_REFILL_MIN 16      // refill from the queue (one atomic on its head + a dependent load) only when this many lanes are empty, or nobody is
                                 // stepping: measured 1 / 8 / 16 / 24 / 32 / 48 -> 5 357 / 5 428 / 5 435 / 5 414 / 5 387 / 5 254 Mrays/s (Euler 8 009 -> 8 148 at 16)
#endif
#ifndef BHRAY_REFILL_MIN_EULER_DENSE
#define BHRAY_REFILL_MIN_EULER_DENSE BHRAY_REFILL_MIN
#endif
#ifndef BHRAY_WAVE_PRIO
#define BHRAY_WAVE_PRIO 1        // bit 0: the latency builds, bit 1: the dense builds - a wave raises its issue priority (s_setprio) while it holds rays predicted to be long: a launch
                                 // lasts as long as its longest ray, and beside three other waves on its SIMD that ray steps at 1.0 us per iteration instead of 0.72.  One frame at a
                                 // time: RK -3 % (S = 2: 1.16 -> 1.13 ms) / -2 % (S = 3: 0.97 -> 0.95; -10 % with timing events in the stream), Euler -2 %; the dense builds at saturation LOSE 3.5 % (the arbiter serves the preferred
                                 // wave's dependent chain where another wave had an instruction ready), and so does the drop-in shim with two frames in flight (+7 %): bit 1 stays off and the
                                 // host enables bit 0's launches (FrameLaunch::probe_empty bit 5) for a ctx with ONE frame slot only.  profiles/EXPERIMENTS.md R6.5
#endif
// predicted length class of a ray from its impact parameter b (b^2 = |(cam - hole) x dir|^2, horizon radius 1: the photon sphere's critical value is 27/4; rays just outside it wind
// round the hole and are the longest of a frame, rays far outside cross the sphere on a chord): class 3 for BHRAY_PRIO_3_LO < b^2 < BHRAY_PRIO_3_HI, 2 / 1 for the wider bands
#ifndef BHRAY_PRIO_3_LO
#define BHRAY_PRIO_3_LO 5.0f     // (bands measured: (4, 12) / (2, 25) / 60, (5, 9) / (3, 16) / 40 - shipped -, (4, 12) / (1, 30) / 100, (6, 8) / (4, 12) / 25, (0, 16) / (0, 36) / 80: within 1.5 % of each other)
#define BHRAY_PRIO_3_HI 9.0f
#define BHRAY_PRIO_2_LO 3.0f
#define BHRAY_PRIO_2_HI 16.0f
#define BHRAY_PRIO_1_HI 40.0f
#endif
#ifndef BHRAY_MESH_COLD_LDS
#define BHRAY_MESH_COLD_LDS 0    // mesh variant: the cold per-lane state in LDS as in the dense build
#endif
#ifndef BHRAY_HIT_LDS
#define BHRAY_HIT_LDS 1         // dense build: "a hit happened" in the cold LDS state rather than an SGPR pair merged at every join of the step loop (+0.3 %, A/B in two sessions: profiles/EXPERIMENTS.md R3.9)
#endif

// Cold per-lane ray state: values the integrator step loop reads or writes only on its rare paths (sphere exit, an actual hit)
// or not at all (the pixel id) — 8 words per lane.  The dense build (6 waves per SIMD = 80 VGPRs) keeps them in LDS, one word
// per lane per plane (lane-consecutive addresses: conflict-free ds_read/ds_write_b32), which takes them out of the register
// budget of the hot loop: the compiler otherwise spills around the phases between the step batches (scratch traffic through
// HBM: 7.5 MB written per 1080p launch against 2.9 MB for the 4-wave build).  The latency build keeps them in registers.
template <bool IN_LDS> struct ColdState;
template <> struct ColdState<false> {
    uint32_t pix_ = 0; F3 color_ = {0, 0, 0}, rdir_ = {0, 0, 1}; float pend_t_ = 0.0f;
    __device__ __forceinline__ explicit ColdState(float*) {}
    __device__ __forceinline__ uint32_t pix() const { return pix_; }
    __device__ __forceinline__ void set_pix(uint32_t v) { pix_ = v; }
    __device__ __forceinline__ F3 color() const { return color_; }
    __device__ __forceinline__ void set_color(F3 v) { color_ = v; }
    __device__ __forceinline__ F3 rdir() const { return rdir_; }
    __device__ __forceinline__ void set_rdir(F3 v) { rdir_ = v; }
    __device__ __forceinline__ float pend_t() const { return pend_t_; }
    __device__ __forceinline__ void set_pend_t(float v) { pend_t_ = v; }
    __device__ __forceinline__ bool hit() const { return false; }
    __device__ __forceinline__ void set_hit(bool) {}
    int urg_ = 0;
    __device__ __forceinline__ int urg() const { return urg_; }
    __device__ __forceinline__ void set_urg(int v) { urg_ = v; }
};
template <> struct ColdState<true> {
    float* b;                                                                     // this lane's column: plane k at b[S * k]
    __device__ __forceinline__ explicit ColdState(float* lds) : b(lds + threadIdx.x) {}
    __device__ __forceinline__ uint32_t pix() const { return __float_as_uint(b[0]); }
    __device__ __forceinline__ void set_pix(uint32_t v) { b[0] = __uint_as_float(v); }
    static constexpr int S = BHRAY_TRACE_THREADS;                                 // plane stride
    __device__ __forceinline__ F3 color() const { return f3(b[S], b[2 * S], b[3 * S]); }
    __device__ __forceinline__ void set_color(F3 v) { b[S] = v.x; b[2 * S] = v.y; b[3 * S] = v.z; }
    __device__ __forceinline__ F3 rdir() const { return f3(b[4 * S], b[5 * S], b[6 * S]); }
    __device__ __forceinline__ void set_rdir(F3 v) { b[4 * S] = v.x; b[5 * S] = v.y; b[6 * S] = v.z; }
    __device__ __forceinline__ float pend_t() const { return b[7 * S]; }
    __device__ __forceinline__ void set_pend_t(float v) { b[7 * S] = v; }
    __device__ __forceinline__ bool hit() const { return b[8 * S] != 0.0f; }                      // (BHRAY_HIT_LDS)
    __device__ __forceinline__ void set_hit(bool v) { b[8 * S] = v ? 1.0f : 0.0f; }
    __device__ __forceinline__ int urg() const { return __float_as_int(b[(8 + BHRAY_HIT_LDS) * S]); }      // (BHRAY_WAVE_PRIO)
    __device__ __forceinline__ void set_urg(int v) { b[(8 + BHRAY_HIT_LDS) * S] = __int_as_float(v); }
};

#include "bhray_quad.inc"

#ifdef BHRAY_WAVE_LOG      // a measurement build (profiles/jobs/r6_wave_log.py): every trace wave leaves a record of when and where it was resident
__device__ unsigned long long* g_wave_log = nullptr;
__device__ unsigned g_wave_log_n = 0, g_wave_log_cap = 0;
}  // namespace bhray
extern "C" __attribute__((visibility("default"))) int bhray_debug_wave_log(unsigned long long* buf, unsigned cap, unsigned* n_out) {
    if (n_out) { if (hipMemcpyFromSymbol(n_out, HIP_SYMBOL(bhray::g_wave_log_n), sizeof(unsigned)) != hipSuccess) return -1; }
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(bhray::g_wave_log), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(bhray::g_wave_log_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(bhray::g_wave_log_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
namespace bhray {
#endif
#ifndef BHRAY_TRACE_KERNEL_ATTR
#define BHRAY_TRACE_KERNEL_ATTR      // (experiments: e.g. __attribute__((amdgpu_num_sgpr(88))))
#endif
template <int METHOD, bool MODELS, bool COUNT, bool DENSE, int EVAL = 0>
__global__ BHRAY_TRACE_KERNEL_ATTR __launch_bounds__(BHRAY_TRACE_THREADS, MODELS ? (DENSE ? BHRAY_TRACE_WAVES_MESH_DENSE : BHRAY_TRACE_WAVES_MESH) : (DENSE ? BHRAY_TRACE_WAVES_DENSE : BHRAY_TRACE_WAVES)) void trace_kernel(const FrameParams* __restrict__ Pb, const FrameLaunch* __restrict__ Fb, const int nb, int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    int err = 0;
    // Execution span of this launch (timed batches only: Fb[0].span != nullptr): first block's start and last block's end on the
    // device's constant-rate clock (s_memrealtime).  Unlike HIP events around the launch it excludes the time the launch waits in
    // its queue behind the persistent kernels of other frames: it is what rocprofv3 --kernel-trace reports for the kernel.
    // span[0] = max(~start) (zero-initialised), span[1] = max(end).
#ifndef BHRAY_NO_SPAN
    if (Fb[0].span && threadIdx.x == 0) atomicMax(&Fb[0].span[0], ~(unsigned long long)wall_clock64());
#endif
#ifdef BHRAY_WAVE_LOG
    const unsigned long long wl_t0 = wall_clock64();
#endif
    constexpr int REL_BATCH = (METHOD == 0 && DENSE && !MODELS) ? BHRAY_REL_BATCH_EULER_DENSE : BHRAY_REL_BATCH;
    constexpr int REFILL_MIN = (METHOD == 0 && DENSE && !MODELS) ? BHRAY_REFILL_MIN_EULER_DENSE : BHRAY_REFILL_MIN;
    constexpr bool UNIFIED = BHRAY_UNIFIED != 0 && EVAL == 0 && (!MODELS || BHRAY_UNIFIED_MESH != 0) && !COUNT;           // the unified march (bhray_step_u.inc) in the contract kernels that do not count
    constexpr bool ONE_TEST = UNIFIED && !MODELS && (((BHRAY_ONE_TEST & 1) != 0 && !DENSE) || ((BHRAY_ONE_TEST & 2) != 0 && DENSE));
    constexpr bool COLD_LDS = (DENSE && !MODELS) || (MODELS && BHRAY_MESH_COLD_LDS != 0);
    constexpr bool MESH_DENSE = MODELS && DENSE;                                              // the mesh variant's build for a saturated device
    constexpr bool MESH_PARK = MESH_DENSE && !COLD_LDS;   // its traversal in a region of its own (see the flat phase)
    constexpr bool FLAT_COLD = MESH_DENSE;                // ... the flat phase marked unlikely
    constexpr bool BVH_WW = MESH_DENSE;                   // ... and the while-while traversal
    constexpr bool WAVE_PRIO = ((BHRAY_WAVE_PRIO & 1) != 0 && !DENSE && !MODELS) || ((BHRAY_WAVE_PRIO & 2) != 0 && DENSE);   // (the mesh variant's lone launches wait for their traversals: nothing, measured)
    __shared__ float cold_lds[COLD_LDS ? (8 + BHRAY_HIT_LDS + (WAVE_PRIO ? 1 : 0)) * BHRAY_TRACE_THREADS : 1];
    // Integrator steps this wave issues for the frames of the batch -> Fb[0].work at the kernel's end.  Wave-uniform: a scalar register.
    // Counted in whole batches of steps, where the step loop is entered (a batch cut short by its last ray counts in full), and for the
    // whole launch rather than per frame: the count of a batch's steps kept live across the loop cost the Euler kernel 1-2 %, a flush
    // per frame the mesh variant 11-18 vector spills (profiles/EXPERIMENTS.md R5.2).
    unsigned work_steps = 0;
    // mesh variant: the short traversal stacks (trace_ray_model) live in LDS
    // (dynamic LDS: with a static array the compiler assumes 64 KB of LDS per CU - gfx950 has 160 KB -, concludes that occupancy is
    // LDS-limited and gives up the 64-VGPR budget of 8 waves per SIMD: 142-152 VGPRs, 3 waves)
    extern __shared__ float4 bvh_dyn_lds[];
    BvhLds bvh_lds; bvh_lds.stack = reinterpret_cast<int2*>(bvh_dyn_lds) + threadIdx.x;
    // the frames of the batch, starting with this block's own: a block whose frame has run dry helps with the others
    for (int fi = 0; fi < nb; fi++) {
    const int fb = (int)((blockIdx.x + (unsigned)fi) % (unsigned)nb);
    const FrameParams& P = Pb[fb];
    const FrameLaunch& F = Fb[fb];
    const LevelParams& L = F.L;
    const SpecLevels& SL = F.SL;
    const uint32_t* __restrict__ queue = F.queue;
    uint32_t* __restrict__ qhead = F.qctl + 1;
    Counters64* __restrict__ counters = F.counters;
    const uint32_t qcount = F.qctl[0];
    // A queue that is (nearly) used up when this wave arrives is seen with a plain load, before any atomic: 4 096 waves hitting one
    // word with failing atomics cost ~50 us per launch (11-13 ns each) - what an empty queue (a fix-up launch of the temporal mode)
    // used to take.  Only here (a load in front of every refill doubles the refill's round trips: -6 % throughput, measured) and only
    // in the latency build and in launches the host expects to be nearly empty (probe_empty): in the dense build even the untaken
    // branch costs a saturated device 2 % (measured: 4 930 -> 4 835 Mrays/s).
    if (!DENSE && (F.probe_empty & 1) && __hip_atomic_load(qhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= qcount) continue;
    // Latency build, a queue with fewer rays than one wave per SIMD has lanes (coarse ladder levels, fix-up launches): the rays are dealt
    // out evenly over the first BHRAY_THIN_WAVES waves (one per SIMD: the first blocks of a grid land on different CUs) - wave w takes
    // entries [w * share, (w + 1) * share) once, without an atomic - instead of 64 to each of the first waves.  (Over ALL waves of the
    // grid, four per SIMD, the waves hold each other up: level 0+1 0.333 -> 0.374 ms.)
    // Such a launch lasts as long as its longest ray, and what that ray pays per iteration is what its wave executes: with 63
    // lane-mates nearly every iteration includes the step-size power and the hit tests (one lane of 64 suffices), with a few
    // lane-mates mostly only what the ray itself needs.  Scheduling only: every ray is the same sequence of operations.
    // A batch (nb > 1): the blocks whose OWN frame this is (blockIdx % nb == fb: every block starts with its own frame) deal the frame's
    // rays out among themselves the same way, and the blocks that come by later to help (fi > 0) leave such a frame alone - it has no
    // queue head to pull from.  (A rank of an 8-way partition renders its coarse levels in launches of ten frames x a few hundred rays.)
    // A WHOLE frame, one frame per launch (the host says so: bit 1 of probe_empty): a share of fewer than BHRAY_THIN_STRIDED_BELOW rays is taken STRIDED - wave w takes
    // entries w, w + waves, w + 2 waves, ... - so that every wave holds an even sample of the frame instead of neighbouring pixels: no wave is left with nothing but the photon
    // ring's rays.  One frame at a time, levels 0+1 of 1920x1080: RK 0.285 -> 0.247 ms, Euler 0.194 -> 0.162, with the mesh 0.66 -> 0.57 (the frame 1.21 -> 1.18, 0.805 -> 0.762 ms).
    // Not for a rank of a partition (the sky slab of an 8-way partition: level 2 0.187 -> 0.205) nor for batches (0.077 -> 0.081 ms per frame): neighbouring rays finish
    // together, and an even sample only helps the launch that holds the ring and nothing else (profiles/EXPERIMENTS.md R5.9).
    uint32_t thin_share = 0, thin_block = blockIdx.x, thin_stride = 0;
    if (MODELS) {                                         // the mesh variant (96-VGPR budget: -2 % with the batch form below) keeps the single-frame form
        if (!DENSE && nb == 1 && BHRAY_THIN_WAVES > 0) {
            const uint32_t total = gridDim.x * (BHRAY_TRACE_THREADS / 64);
            const uint32_t waves = total < (uint32_t)BHRAY_THIN_WAVES ? total : (uint32_t)BHRAY_THIN_WAVES;
            const uint32_t share = (qcount + waves - 1) / waves;
            if (share < 64u) thin_share = share > 0u ? share : 1u;
            if ((F.probe_empty & 2) && share < (uint32_t)BHRAY_THIN_STRIDED_BELOW) thin_stride = waves;
        }
    } else if (!DENSE && BHRAY_THIN_WAVES > 0 && (nb == 1 || gridDim.x >= 4u * (uint32_t)nb)) {   // (every frame of a batch needs blocks of its own)
        uint32_t own_blocks = gridDim.x;
        if (nb > 1) { own_blocks = (gridDim.x - (uint32_t)fb + (uint32_t)nb - 1u) / (uint32_t)nb; thin_block = blockIdx.x / (uint32_t)nb; }
        const uint32_t total = own_blocks * (BHRAY_TRACE_THREADS / 64);
        const uint32_t cap = nb > 1 ? ((uint32_t)BHRAY_THIN_WAVES + (uint32_t)nb - 1u) / (uint32_t)nb : (uint32_t)BHRAY_THIN_WAVES;
        const uint32_t waves = total < cap ? total : cap;
        if (waves > 0u) {
            const uint32_t share = (qcount + waves - 1) / waves;
            if (share < 64u) thin_share = share > 0u ? share : 1u;
            if ((F.probe_empty & 2) && share < (uint32_t)BHRAY_THIN_STRIDED_BELOW) thin_stride = waves;
        }
        if (thin_share != 0u && fi != 0) continue;
    }
    const HotParams H = load_hot<MODELS>(P);
    // The quad march (bhray_quad.inc): a queue that fits 16 rays per wave on the waves the host allows it (bits 2-4 of probe_empty: waves per
    // SIMD; 0 = off) is marched with one ray per QUAD of lanes - x, y, z on three lanes - instead of one per lane: fewer instructions per
    // iteration of the launch's longest ray, the same operations per ray.  Dealt out once like a scalar thin share; whole rounds of one wave per SIMD.
    if constexpr (!DENSE && !MODELS && !COUNT && EVAL == 0) {
        const uint32_t q_wps = ((uint32_t)F.probe_empty >> 2) & 7u;
        if (q_wps != 0u && BHRAY_THIN_WAVES > 0 && (nb == 1 || gridDim.x >= 4u * (uint32_t)nb)) {
            uint32_t own_blocks = gridDim.x, my_block = blockIdx.x;
            if (nb > 1) { own_blocks = (gridDim.x - (uint32_t)fb + (uint32_t)nb - 1u) / (uint32_t)nb; my_block = blockIdx.x / (uint32_t)nb; }
            const uint32_t total = own_blocks * (BHRAY_TRACE_THREADS / 64);
            const uint32_t round = nb > 1 ? ((uint32_t)BHRAY_THIN_WAVES + (uint32_t)nb - 1u) / (uint32_t)nb : (uint32_t)BHRAY_THIN_WAVES;     // one wave per SIMD
            const uint32_t rounds = ((qcount + 15u) / 16u + round - 1u) / round;
            uint32_t waves = rounds * round;
            if (waves > total) waves = total;
            if (rounds <= q_wps && rounds <= (uint32_t)BHRAY_QUAD_MAX_WPS && (unsigned long long)waves * 16ull >= (unsigned long long)qcount) {
                if (fi != 0) continue;                                    // (a block that comes by to help leaves such a frame alone, as with the scalar thin shares)
                const uint32_t w = my_block * (BHRAY_TRACE_THREADS / 64) + (threadIdx.x >> 6);
                if (w < waves && qcount != 0u) {
                    const uint32_t share = (qcount + waves - 1u) / waves;          // <= 16
                    const bool strided = (F.probe_empty & 2) != 0;                 // a whole frame, one frame per launch: entries w, w + waves, ... (see the thin shares below)
                    quad_march<METHOD>(P, F, H, qcount, strided ? w : w * share, share, strided ? waves : 0u, work_steps);
                }
                continue;
            }
        }
    }
    const F3 bpos = H.bh;
    const float t_max = 1e5f, t_min = 1e-8f;

    // per-lane ray state (trace_ray locals, ray.wgsl:486-516)
    int mode = M_EMPTY;
    ColdState<COLD_LDS> cold(cold_lds);   // pix, color, rdir (the ray's original direction), pend_t
    F3 cpos = f3(0, 0, 0), cdir = f3(0, 0, 1), ppos = f3(0, 0, 0), pdir = f3(0, 0, 1);      // set when a lane takes a ray
    F3 rkpos = f3(0, 0, 0), rkdir = f3(0, 0, 1);
    float rkh = 0.0f;
    float amount = 1.0f, closest = H.ray_distance;
    // the segment length of a step's hit test (ray.wgsl:530,541): Euler - the uniform step size; RK - the post-step h, i.e. rkh
    float dist_c = P.ray_distance_f;      // flength(integrator position - bpos) (N7), carried between steps
    float cpos_dist = P.ray_distance_f;   // flength(cpos - bpos): equals dist_c except in RK mode after a hit moved cpos
    F3 qrel = f3(0, 0, 0);                // integrator position - bpos (the operand of dist_c), carried with it: the next step's q0
    int it = 0;
    // "a hit happened" (0 / 1).  As a bool it lives in an SGPR pair and costs three scalar operations at every control-flow join of the
    // step loop, taken or not - the latency build keeps it in a VGPR; the dense build has no VGPR to spare (80: 6 waves per SIMD).
    typename std::conditional<COLD_LDS, bool, int>::type hit = 0;
    constexpr bool HIT_IN_LDS = COLD_LDS && BHRAY_HIT_LDS != 0;      // dense build: the flag in the cold LDS state instead (no mask merging at the step loop's joins)
#define HIT_SET(v) do { if (HIT_IN_LDS) cold.set_hit(v); else hit = (v); } while (0)
#define HIT_GET() (HIT_IN_LDS ? cold.hit() : (bool)hit)
    bool exhausted = false;
    int flat_round = 0;
    unsigned long long cnt[13];           // [0..9] = bhray_counters' frame counters, [10] wave steps (lane 0), [11] unused (rays adopted by the drain merging of round 2), [12] longest ray
    if (COUNT) { for (int k = 0; k < 13; k++) cnt[k] = 0; }

    for (;;) {
        // ---- refill finished lanes from the queue (wave ballot + prefix popcount)
        {
            const unsigned long long need = __ballot(mode == M_EMPTY);
            if (need != 0ull && !exhausted && (REFILL_MIN <= 1 || __popcll(need) >= REFILL_MIN || !__any(mode == M_REL))) {
                uint32_t n = (uint32_t)__popcll(need);
                uint32_t base = 0;
                if (!DENSE && thin_share != 0u) {                 // this wave's share, once (all lanes are empty: n == 64 > share)
                    n = thin_share;
                    base = thin_block * (BHRAY_TRACE_THREADS / 64) + (threadIdx.x >> 6);
                    if (thin_stride != 0u) { if (base >= thin_stride) n = 0u; }      // strided: entries base, base + waves, ... (the waves beyond `waves` take nothing)
                    else base *= thin_share;
                    exhausted = true;
                } else {
                    // (the lane id recomputed here - two instructions - instead of held in a VGPR across the step loop: the dense build has none to spare)
                    if ((int)lanes_below(~0ull) == (int)__builtin_ctzll(need)) base = atomicAdd(qhead, n);
                    base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(need));
                    if (base + n >= qcount) exhausted = true;
                }
                const uint32_t rank = lanes_below(need);
                uint32_t idx = base + rank;
                if (!DENSE && thin_stride != 0u) idx = base + rank * thin_stride;
                if (mode == M_EMPTY && (DENSE || rank < n) && idx < qcount) {
                    const uint32_t pix = queue[idx];
                    cold.set_pix(pix);
                    const int px = (int)(pix & 0x7fffu), py = (int)((pix >> 15) & 0x7fffu);
                    // level geometry: the launch's level, or (speculative multi-level launch) the entry's tagged level
                    int lw = L.w, lh = L.h;
                    if (SL.n > 0) {
                        const int lv = (int)(pix >> 30);
                        lw = SL.l[0].w; lh = SL.l[0].h;
#pragma unroll
                        for (int q = 1; q < BHRAY_MAX_SPEC_LEVELS; q++) if (lv == q) { lw = SL.l[q].w; lh = SL.l[q].h; }
                    }
                    // create_ray, ray.wgsl:269-285 (right/up/fwd_ff hoisted to the host, bit-identical)
                    const int sm = (lw - 1) < (lh - 1) ? (lw - 1) : (lh - 1);
                    const float increment = 1.0f / (float)sm;
                    const float posx = (2.0f * ((float)px - (float)(lw - 1) * 0.5f)) * increment;
                    const float posy = (2.0f * ((float)py - (float)(lh - 1) * 0.5f)) * increment;
                    const F3 cam = ld3(P.cam);               // uniform: scalar loads here, not three VGPRs held across the step loop
                    const F3 rdir = normalize((ld3(P.right) * posx + ld3(P.up) * posy) + ld3(P.fwd_ff));
                    cold.set_rdir(rdir);
                    cpos = cam; cdir = rdir; ppos = cam; pdir = rdir;
                    rkpos = cam; rkdir = rdir; rkh = P.step_size;
                    cold.set_color(f3(0, 0, 0)); amount = 1.0f; closest = H.ray_distance;
                    dist_c = P.ray_distance_f; cpos_dist = P.ray_distance_f; qrel = cam - bpos;
                    it = 0; HIT_SET(0);
                    mode = P.relativity0 ? M_REL : M_FLAT;
                    if (COUNT) cnt[3]++;
                    if (WAVE_PRIO && (F.probe_empty & 32)) {   // scheduling only: plain arithmetic, no pixel depends on it
                        const F3 cb = cross(qrel, rdir);
                        const float b2 = dot(cb, cb);
                        cold.set_urg((b2 > BHRAY_PRIO_3_LO && b2 < BHRAY_PRIO_3_HI) ? 3 : ((b2 > BHRAY_PRIO_2_LO && b2 < BHRAY_PRIO_2_HI) ? 2 : (b2 < BHRAY_PRIO_1_HI ? 1 : 0)));
                    }
                }
                if (WAVE_PRIO && (F.probe_empty & 32)) {       // the wave's priority: its longest ray's class (until the next refill)
                    const int u = mode != M_EMPTY ? cold.urg() : -1;
                    if (__any(u == 3)) __builtin_amdgcn_s_setprio(3);
                    else if (__any(u == 2)) __builtin_amdgcn_s_setprio(2);
                    else if (__any(u == 1)) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
            if (!__any(mode != M_EMPTY)) break;
        }
        // ---- deferred disk shading (ray.wgsl:612-663 and the hit bookkeeping of 537-552) for lanes that paused on a disk hit
        if (__any(mode >= M_SHADE_REL)) {
            if (mode >= M_SHADE_REL) {
                const float pend_t = cold.pend_t();
                Hit crs; crs.hit = true; crs.t = pend_t; crs.color = f3(0.0f, 0.0f, 0.0f); crs.opacity = 0.0f;
                shade_disk<COUNT>(H, ppos, pdir, pend_t, H.ray_distance, crs, cnt);
                cpos = cpos + pdir * crs.t;
                cpos_dist = fdistance(cpos, bpos);
                if (METHOD == 0) { dist_c = cpos_dist; qrel = cpos - bpos; }
                const F3 cc = f3(clamp_(crs.color.x, 0.0f, 1.0f), clamp_(crs.color.y, 0.0f, 1.0f), clamp_(crs.color.z, 0.0f, 1.0f));
                cold.set_color(cold.color() + cc * (amount * crs.opacity));
                amount *= 1.0f - crs.opacity;
                HIT_SET(1);
                if (amount < 0.005f) mode = M_FINISH;
                else { it++; mode = (mode == M_SHADE_FLAT) ? M_FLAT : M_REL; }
            }
        }

        // ---- flat-space iterations (ray.wgsl:554-569), one per lane that is in flat space.
        // With meshes a flat iteration is a BVH traversal executed by the whole wave for the few lanes that need it,
        // so those lanes are batched: the phase runs when enough of them wait, when nobody is integrating, or at the
        // latest every BHRAY_FLAT_DEFER-th round.  Each ray still sees its own iterations in order: results are unchanged.
        bool run_flat = true;
        if (MODELS) {
            const unsigned long long mf = __ballot(mode == M_FLAT);
            flat_round++;
            run_flat = mf != 0ull && (__popcll(mf) >= BHRAY_FLAT_MIN_LANES || !__any(mode == M_REL) || flat_round >= BHRAY_FLAT_DEFER);
            if (run_flat) flat_round = 0;
        }
        // BHRAY_MESH_PARK (mesh variant): the traversal runs in a region of its own in which NOTHING of the march is live - every per-lane
        // variable of the ray state machine is stored to scratch in front of it and loaded back behind it, once per flat phase in which a
        // lane's ray passes the root-box test at all.  The compiler would otherwise keep part of the marching state in registers across the
        // traversal (it parks 37-47 dwords of ~70) and, at the register budget of six waves per SIMD, spill inside the traversal's loops.
        Hit mesh_r; F3 mesh_nrm = f3(0, 0, 0);
        mesh_r.hit = false; mesh_r.t = t_max; mesh_r.color = f3(0, 0, 0); mesh_r.opacity = 0.0f;
        if (MESH_PARK) {
            static_assert(BHRAY_MAX_MODELS == 1, "the parked traversal handles the reference's one model");
            bool want = false;
            if (__builtin_expect(run_flat && __any(mode == M_FLAT), 0)) {
                if (mode == M_FLAT && it < H.max_iter && P.model_count > 0 && P.models[0].visible != 0) {
                    bool skip = false;
                    if (P.models[0].root_cull != 0) {
                        const F3 inv = f3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
                        if (fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY) {
                            const ModelDev& Md = P.models[0];
                            const float d0 = hit_aabb(cpos, inv, make_float4(Md.root_lo[0], Md.root_lo[1], Md.root_lo[2], 0.0f),
                                                      make_float4(Md.root_hi[0], Md.root_hi[1], Md.root_hi[2], 0.0f), ld3(Md.pos));
                            skip = d0 > t_max;
                            if (COUNT && skip) cnt[6]++;
                        }
                    }
                    want = !skip;
                }
                if (__any(want)) {
                    float pk[40];
                    float* pkp = pk;
                    const F3 tpos = cpos, tdir = cdir;
                    pk[0] = cpos.x; pk[1] = cpos.y; pk[2] = cpos.z; pk[3] = cdir.x; pk[4] = cdir.y; pk[5] = cdir.z;
                    pk[6] = ppos.x; pk[7] = ppos.y; pk[8] = ppos.z; pk[9] = pdir.x; pk[10] = pdir.y; pk[11] = pdir.z;
                    pk[12] = rkpos.x; pk[13] = rkpos.y; pk[14] = rkpos.z; pk[15] = rkdir.x; pk[16] = rkdir.y; pk[17] = rkdir.z;
                    pk[18] = rkh; pk[19] = amount; pk[20] = closest; pk[21] = dist_c; pk[22] = cpos_dist;
                    pk[23] = qrel.x; pk[24] = qrel.y; pk[25] = qrel.z;
                    pk[26] = __int_as_float(it); pk[27] = __int_as_float((int)hit); pk[28] = __int_as_float(mode);
                    { const F3 cc_ = cold.color(), rd_ = cold.rdir();
                      pk[29] = __uint_as_float(cold.pix()); pk[30] = cc_.x; pk[31] = cc_.y; pk[32] = cc_.z;
                      pk[33] = rd_.x; pk[34] = rd_.y; pk[35] = rd_.z; pk[36] = cold.pend_t(); }
                    asm volatile("" : "+v"(pkp) : : "memory");         // what was stored may be read and changed behind the compiler's back: nothing above stays in a register
                    if (want) trace_ray_model<COUNT, BVH_WW>(P.models[0], bvh_lds, tpos, tdir, t_min, t_max, mesh_r, mesh_nrm, cnt, &err);
                    asm volatile("" : "+v"(pkp) : : "memory");
                    cpos = f3(pkp[0], pkp[1], pkp[2]); cdir = f3(pkp[3], pkp[4], pkp[5]);
                    ppos = f3(pkp[6], pkp[7], pkp[8]); pdir = f3(pkp[9], pkp[10], pkp[11]);
                    rkpos = f3(pkp[12], pkp[13], pkp[14]); rkdir = f3(pkp[15], pkp[16], pkp[17]);
                    rkh = pkp[18]; amount = pkp[19]; closest = pkp[20]; dist_c = pkp[21]; cpos_dist = pkp[22];
                    qrel = f3(pkp[23], pkp[24], pkp[25]);
                    it = __float_as_int(pkp[26]); hit = __float_as_int(pkp[27]); mode = __float_as_int(pkp[28]);
                    cold.set_pix(__float_as_uint(pkp[29])); cold.set_color(f3(pkp[30], pkp[31], pkp[32]));
                    cold.set_rdir(f3(pkp[33], pkp[34], pkp[35])); cold.set_pend_t(pkp[36]);
                }
            }
        }
        // (FLAT_COLD: the flat phase marked unlikely, so that the register allocator weighs the step loop above the traversal's loops)
        const bool flat_now = run_flat && __any(mode == M_FLAT);
        if (FLAT_COLD ? __builtin_expect(flat_now, 0) : flat_now) {
            if (mode == M_FLAT) {
                if (it >= H.max_iter) {
                    mode = M_FINISH;
                } else {
                    if (COUNT) cnt[5]++;
                    Hit rs; rs.hit = false; rs.t = t_max; rs.color = f3(0, 0, 0); rs.opacity = 0.0f;
                    if (MESH_PARK) {      // the traversal has run in its own region above
                        if (mesh_r.hit && mesh_r.t < rs.t) {
                            rs = mesh_r;
                            const F3 light = normalize(f3(0.2f, 0.2f, -1.0f));
                            rs.color = rs.color * dot(mesh_nrm, light);
                        }
                    } else if (MODELS) {
                        for (int mi = 0; mi < P.model_count; mi++) {
                            if (P.models[mi].visible != 0) {
                                Hit r; F3 nrm;
                                r.hit = false;
                                // The traversal's first visit tests the two children of the root.  A ray that misses the union of
                                // their boxes misses both (the slab test is monotone in the box when 1/dir is finite), so that visit
                                // - a function call and a dependent 64-byte load for the whole wave - is skipped; most rays that
                                // leave the sphere point away from the mesh.  Counted as the visit it replaces.
                                bool skip = false;
                                if (P.models[mi].root_cull != 0) {
                                    const F3 inv = f3(1.0f / cdir.x, 1.0f / cdir.y, 1.0f / cdir.z);
                                    if (fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY) {
                                        const ModelDev& Md = P.models[mi];
                                        const float d0 = hit_aabb(cpos, inv, make_float4(Md.root_lo[0], Md.root_lo[1], Md.root_lo[2], 0.0f),
                                                                  make_float4(Md.root_hi[0], Md.root_hi[1], Md.root_hi[2], 0.0f), ld3(Md.pos));
                                        skip = d0 > t_max;
                                        if (COUNT && skip) cnt[6]++;
                                    }
                                }
                                if (!skip) trace_ray_model<COUNT, BVH_WW>(P.models[mi], bvh_lds, cpos, cdir, t_min, t_max, r, nrm, cnt, &err);
                                if (r.hit && r.t < rs.t) {
                                    rs = r;
                                    const F3 light = normalize(f3(0.2f, 0.2f, -1.0f));
                                    rs.color = rs.color * dot(nrm, light);
                                }
                            }
                        }
                    }
                    float ths = t_max;
                    // (the radius through an opaque copy: R*R is loop-invariant, and hoisted out of the frame loop it costs the dense build a
                    // VGPR it does not have - it was the kernel's one spill, 8 bytes of scratch per lane)
                    float Rflat = H.R; asm volatile("" : "+v"(Rflat));
                    const bool hs = hit_sphere(ppos, pdir, Rflat, bpos, t_min, t_max, ths);
                    if (!hs && !rs.hit) {
                        mode = M_FINISH;                                   // break (no increment)
                    } else {
                        bool chit = false; Hit crs = rs;
                        if (hs && ths < rs.t) { cpos = cpos + cdir * ths; mode = M_REL; cpos_dist = fdistance(cpos, bpos); if (METHOD == 0) { dist_c = cpos_dist; qrel = cpos - bpos; } }
                        else { chit = rs.hit; }
                        if (chit) {
                            cpos = cpos + pdir * crs.t;
                            cpos_dist = fdistance(cpos, bpos);
                            if (METHOD == 0) { dist_c = cpos_dist; qrel = cpos - bpos; }
                            const F3 cc = f3(clamp_(crs.color.x, 0.0f, 1.0f), clamp_(crs.color.y, 0.0f, 1.0f), clamp_(crs.color.z, 0.0f, 1.0f));
                            cold.set_color(cold.color() + cc * (amount * crs.opacity));
                            amount *= 1.0f - crs.opacity;
                            HIT_SET(1);
                        }
                        if (amount < 0.005f) mode = M_FINISH; else it++;
                    }
                }
            }
        }

        // ---- epilogue (ray.wgsl:583-595) for lanes whose loop ended
        if (__any(mode == M_FINISH)) {
            if (mode == M_FINISH) {
                float4 o;
                F3 color = cold.color();
                const uint32_t pix = cold.pix();
                if (COUNT && (unsigned long long)it > cnt[12]) cnt[12] = (unsigned long long)it;
                if (HIT_GET() || it <= 5) {
                    if (amount > 0.001f) {
                        if (COUNT) cnt[9]++;
                        // cartesian_to_spherical(dir.xzy), ray.wgsl:255-261, 585-586
                        const float theta = bh_atan2(sqrtf(cdir.x * cdir.x + cdir.z * cdir.z), cdir.y);
                        const float phi = bh_atan2(cdir.z, cdir.x);
                        const float PI_F = 3.1415926f;
                        float u = (phi + 2.6f * PI_F) / (2.0f * PI_F);
                        float v = (PI_F - theta) / PI_F;
                        u = u - truncf(u); v = v - truncf(v);
                        const float4 sc = sample_bilinear(P.sky, u, v);
                        const F3 miss = f3((sc.x * sc.x) * (sc.x * sc.x), (sc.y * sc.y) * (sc.y * sc.y), (sc.z * sc.z) * (sc.z * sc.z));
                        color = color + miss * amount;
                    }
                    o = make_float4(color.x, color.y, color.z, 1.0f);
                } else {
                    o = make_float4(cdir.x, cdir.y, cdir.z, 0.0f);
                }
                {
                    const int ox = (int)(pix & 0x7fffu), oy = (int)((pix >> 15) & 0x7fffu);
                    if (SL.n > 0) {
                        const int lv = (int)(pix >> 30);
                        float4* dst = SL.l[0].out; int pitch = SL.l[0].out_pitch, x0 = SL.l[0].out_x0; const int32_t* rowmap = SL.l[0].rowmap;
#pragma unroll
                        for (int q = 1; q < BHRAY_MAX_SPEC_LEVELS; q++) if (lv == q) { dst = SL.l[q].out; pitch = SL.l[q].out_pitch; x0 = SL.l[q].out_x0; rowmap = SL.l[q].rowmap; }
                        const int orow = rowmap ? rowmap[oy] : oy;
                        dst[(size_t)orow * (size_t)pitch + (size_t)(ox - x0)] = o;
                        if (COUNT) {                                       // where the work lies (bhray_get_row_work)
                            unsigned long long* rw = SL.l[0].row_work;
#pragma unroll
                            for (int q = 1; q < BHRAY_MAX_SPEC_LEVELS; q++) if (lv == q) rw = SL.l[q].row_work;
                            if (rw) atomicAdd(&rw[oy], (unsigned long long)it);
                        }
                        if (F.stamp_value != 0u) {                       // temporal speculation: this pixel of this level is done for this frame
                            uint32_t* st = SL.l[0].stamp; int sw = SL.l[0].w;
#pragma unroll
                            for (int q = 1; q < BHRAY_MAX_SPEC_LEVELS; q++) if (lv == q) { st = SL.l[q].stamp; sw = SL.l[q].w; }
                            st[(size_t)oy * (size_t)sw + (size_t)ox] = F.stamp_value;
                        }
                    } else {
                        L.out[out_index(L, ox, oy)] = o;
                        if (COUNT && F.row_work) atomicAdd(&F.row_work[oy], (unsigned long long)it);
                    }
                }
                mode = M_EMPTY;
            }
        }

        // ---- a batch of integrator steps (ray.wgsl:522-553) for lanes inside the sphere
        if (__any(mode == M_REL)) work_steps += (unsigned)REL_BATCH;
        if constexpr (UNIFIED) {
            // The unified march (bhray_step_u.inc): pairs of steps over two position register sets, the state that only the rare paths and the other
            // phases read (cpos / cdir / ppos / pdir apart from the integrator's own) written when a lane LEAVES the march, not on every step.
            // RK: the integrator runs on rkpos / rkdir and the hit test's segment starts at cpos, which differs from rkpos for one step after a
            // disk hit or a sphere entry moved it (ray.wgsl keeps two rays) - a wave that holds such a lane takes one step of the general form first.
            if (METHOD == 1) {
                const bool odd = (mode == M_REL) & ((cpos.x != rkpos.x) | (cpos.y != rkpos.y) | (cpos.z != rkpos.z) | (cpos_dist != dist_c));
                if (__any(odd)) {
                    work_steps += 1u;
                    {                                        // (the lean text in both builds: the dense text leaves a step with `continue`)
#define BHRAY_STEP_LEAN 1
#include "bhray_step.inc"
#undef BHRAY_STEP_LEAN
                    }
                }
            }
            if (mode == M_REL && it >= H.max_iter) mode = M_FINISH;      // the iteration limit (ray.wgsl:522) in front of the pairs; inside them it is tested where `it` changes
            F3& upos = METHOD == 0 ? cpos : rkpos;      // the integrator's position and direction
            F3& udir = METHOD == 0 ? cdir : rkdir;
            // (the hole at the origin - all three words +0, wave-uniform: a scalar test - marches without forming position - bpos: see bhray_step_u.inc)
            if (!MODELS && (BHRAY_ORIGIN_PATH & (1 << METHOD)) != 0 && ((__float_as_uint(H.bh.x) | __float_as_uint(H.bh.y) | __float_as_uint(H.bh.z)) == 0u)) {
#define BHRAY_U_ORIGIN 1
                for (int k = 0; k < REL_BATCH; k += 2) {
                    if (!__any(mode == M_REL)) break;
#define BHRAY_U_FIRST 1
#include "bhray_step_u.inc"
#undef BHRAY_U_FIRST
#define BHRAY_U_FIRST 0
#include "bhray_step_u.inc"
#undef BHRAY_U_FIRST
                }
                if (mode == M_REL) qrel = upos;
#undef BHRAY_U_ORIGIN
            } else {
#define BHRAY_U_ORIGIN 0
            for (int k = 0; k < REL_BATCH; k += 2) {
                if (!__any(mode == M_REL)) break;
#define BHRAY_U_FIRST 1
#include "bhray_step_u.inc"
#undef BHRAY_U_FIRST
#define BHRAY_U_FIRST 0
#include "bhray_step_u.inc"
#undef BHRAY_U_FIRST
            }
#undef BHRAY_U_ORIGIN
            }
            if (mode == M_REL) {                         // between batches every lane's state is exactly the general step's (after the second step of a pair ppos
                if (METHOD == 1) { cpos = rkpos; cdir = rkdir; }   // is the previous position already): a general step, the iteration limit inside it, the other phases read it
                pdir = udir;
                cpos_dist = dist_c;
            }
        } else {
        for (int k = 0; k < REL_BATCH; k++) {       // (unrolled by 2 / 4 to let prev = curr become renaming: -1 % / 0 %, measured)
            if (!__any(mode == M_REL)) break;
            if (COUNT && lane == 0) cnt[10]++;
            if (DENSE || MODELS) {                    // see bhray_step.inc
#define BHRAY_STEP_LEAN 0
#include "bhray_step.inc"
#undef BHRAY_STEP_LEAN
            } else {
#define BHRAY_STEP_LEAN 1
#include "bhray_step.inc"
#undef BHRAY_STEP_LEAN
            }
        }
        }
    }

    if (COUNT) {
        for (int k = 3; k < 12; k++) {
            unsigned long long v = cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&counters->v[k], v);
        }
        {
            unsigned long long v = cnt[12];
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(v, off); v = o > v ? o : v; }
            if (lane == 0 && v) atomicMax(&counters->v[12], v);
        }
    }
    }   // frames of the batch
    if (work_steps != 0u && Fb[0].work && lanes_below(~0ull) == 0u) atomicAdd(&Fb[0].work[blockIdx.x & (BHRAY_WORK_WORDS - 1)], (unsigned long long)work_steps);
#ifndef BHRAY_NO_SPAN
    if (Fb[0].span && threadIdx.x == 0) atomicMax(&Fb[0].span[1], (unsigned long long)wall_clock64());
#endif
    if (err) *err_flag = err;
#ifdef BHRAY_WAVE_LOG
    if (g_wave_log && lanes_below(~0ull) == 0u) {      // one record per wave: start, end (100 MHz), where it ran, how many integrator steps it issued
        const unsigned i = atomicAdd(&g_wave_log_n, 1u);
        if (i < g_wave_log_cap) {
            unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_wave_log[4 * i] = wl_t0; g_wave_log[4 * i + 1] = wall_clock64();
            g_wave_log[4 * i + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            g_wave_log[4 * i + 3] = (unsigned long long)work_steps | ((unsigned long long)(DENSE ? 1 : 0) << 24) | ((((unsigned long long)(size_t)Fb[0].queue) >> 8) << 32);   // the queue's address: which launch
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// sky resolve: sky.wgsl:1-38 (the pass after the ray levels, mod.rs:419)
// ------------------------------------------------------------------------------------------
// HBM-bound: 16 B read + 8 B written per pixel (+ 4 sky texels, mostly cache hits); one thread per pixel,
// consecutive lanes on consecutive pixels, so loads are 1 KiB and stores 512 B per wave instruction.
__global__ __launch_bounds__(256) void sky_kernel(const TexDev sky, const float4* __restrict__ src, uint2* __restrict__ dst, size_t npix) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    float4 p = src[i];
    if (p.w == 0.0f) {                                                   // sky.wgsl:19
        // cartesian_to_spherical(p.xzy), sky.wgsl:20, 32-38
        const float theta = bh_atan2(sqrtf(p.x * p.x + p.z * p.z), p.y);
        const float phi = bh_atan2(p.z, p.x);
        const float PI_F = 3.1415926f;
        float u = (phi + 2.6f * PI_F) / (2.0f * PI_F);
        float v = (PI_F - theta) / PI_F;
        u = u - truncf(u); v = v - truncf(v);
        const float4 sc = sample_bilinear(sky, u, v);
        p = make_float4((sc.x * sc.x) * (sc.x * sc.x), (sc.y * sc.y) * (sc.y * sc.y), (sc.z * sc.z) * (sc.z * sc.z), 1.0f);
    }
    // rgba16float target (sky.wgsl:1): round to nearest even
    const uint32_t lo = (uint32_t)__half_as_ushort(__float2half_rn(p.x)) | ((uint32_t)__half_as_ushort(__float2half_rn(p.y)) << 16);
    const uint32_t hi = (uint32_t)__half_as_ushort(__float2half_rn(p.z)) | ((uint32_t)__half_as_ushort(__float2half_rn(p.w)) << 16);
    dst[i] = make_uint2(lo, hi);
}

// Launchers report the error of THEIR launch: HIP's last-error state is sticky across unrelated calls of the process (e.g. a
// refused hipSetDevice of another ctx), so it is cleared first.
hipError_t launch_sky(const TexDev& sky, const float4* src, uint2* dst, size_t npix, hipStream_t s) {
    if (npix == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(sky_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, sky, src, dst, npix);
    return hipGetLastError();
}

// bhray_selftest: the step loop's rcp_rn / sqrt_rn against the compiler's IEEE `1.0f / x` and `sqrtf(x)` on all 2^32 bit
// patterns (a wave tests 64 consecutive patterns, so in-range waves take the short sequences, the others the IEEE lowering).
__global__ __launch_bounds__(256) void selftest_kernel(unsigned long long* __restrict__ bad) {
    unsigned long long nr = 0, ns = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = u2f((uint32_t)i);
        const float a = rcp_rn(x), b = 1.0f / x;
        if (f2u(a) != f2u(b) && !(a != a && b != b)) nr++;
        const float c = sqrt_rn(x), d = sqrtf(x);
        if (f2u(c) != f2u(d) && !(c != c && d != d)) ns++;
        if (x > 0.00002f && f2u(pow_m001_step(x)) != f2u(bh_pow_m001(x))) nr++;      // the step-size power on its whole domain
    }
    // max3_abs / closest_min against the compare-select forms they replace (N5), bit for bit: every triple / pair of 16 special values (quiet NaNs of
    // both signs with payloads, infinities, zeros, denormals, numbers) and 2^20 pseudo-random bit patterns
    {
        const uint32_t sp[16] = {0x7fc00000u, 0xffc00001u, 0x7fe12345u, 0x7f800000u, 0xff800000u, 0x00000000u, 0x80000000u, 0x00000001u,
                                 0x807fffffu, 0x00800000u, 0x3f800000u, 0xbf800000u, 0x37a7c5acu, 0x7f7fffffu, 0xff7fffffu, 0x3727c5acu};
        const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
        if (gid < (1ull << 20)) {
            uint32_t a, b, c;
            if (gid < 4096) { a = sp[gid & 15]; b = sp[(gid >> 4) & 15]; c = sp[(gid >> 8) & 15]; }
            else { a = (uint32_t)gid * 2654435761u; b = a * 2246822519u + 374761393u; c = b * 3266489917u + 668265263u; }
            // (no SIGNALLING NaNs: arithmetic never produces one, and the operands here are results of arithmetic - e of fused multiply-adds, cd of a square root;
            // the hardware maximum / minimum would quiet and return one where the compare-select skips it)
            if ((a & 0x7fc00000u) == 0x7f800000u && (a & 0x003fffffu)) a |= 0x00400000u;
            if ((b & 0x7fc00000u) == 0x7f800000u && (b & 0x003fffffu)) b |= 0x00400000u;
            if ((c & 0x7fc00000u) == 0x7f800000u && (c & 0x003fffffu)) c |= 0x00400000u;
            const float x = u2f(a), y = u2f(b), z = u2f(c);
            const float want = max_(max_(fabsf(x), fabsf(y)), fabsf(z)), got = max3_abs(x, y, z);
            if (f2u(want) != f2u(got)) nr++;
            const float cd = fabsf(x), cl = fabsf(y);                          // distances: no sign; `closest` is never a NaN
            if (!(cl != cl)) { const float w2 = cd < cl ? cd : cl, g2 = closest_min(cd, cl); if (f2u(w2) != f2u(g2)) nr++; }
        }
    }
    if (nr) atomicAdd(&bad[0], nr);
    if (ns) atomicAdd(&bad[1], ns);
    // bh_acos non-increasing: over [-1, -0] (bit patterns 0xbf800000 down to 0x80000001 against their successor toward zero)
    // and over [+0, 1] (0x00000000 .. 0x3f7fffff against their successor)
    unsigned long long na = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < 0x3f800000ull; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t u = (uint32_t)i;
        if (bh_acos(u2f(u + 1u)) > bh_acos(u2f(u))) na++;                                   // x in [0, 1): next value up
        if (u > 0u && bh_acos(u2f(0x80000000u | (u - 1u))) > bh_acos(u2f(0x80000000u | u))) na++;   // x = -|u|: next value toward zero
    }
    if (!(bh_acos(u2f(0x80000000u)) == bh_acos(0.0f))) na++;
    if (na) atomicAdd(&bad[2], na);
}

hipError_t launch_selftest(unsigned long long* bad2, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(selftest_kernel, dim3(8192), dim3(256), 0, s, bad2);
    return hipGetLastError();
}

// temporal speculation: the prediction of a level = every pixel that is not a copy position and has, within `radius` pixels, a pixel
// the previous frame's exact classification had to trace (F.need).  radius 0 is last frame's traced set itself; radius 1 also
// covers the pixels that region boundaries reach when the camera moves by up to a pixel per frame at that level - those are the
// long rays (photon ring, disk edge), and a single one of them missing costs a whole dependent launch.  Same tiling as classify.
template <int DUMMY>
__global__ __launch_bounds__(BHRAY_CLASSIFY_THREADS) void predict_kernel(const FrameParams* __restrict__ Pb, const FrameLaunch* __restrict__ Fb) {
    __shared__ uint32_t append_lds[BHRAY_CLASSIFY_TILES];
    // one launch for all levels: blockIdx.z = level (its FrameLaunch entries follow the previous level's), blockIdx.y = frame; the
    // grid is as wide as the largest level needs, the blocks beyond a level's own count leave at once
    const FrameLaunch& F = Fb[blockIdx.z * gridDim.y + blockIdx.y];
    const LevelParams& L = F.L;
    if ((int)blockIdx.x >= F.blocks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool want[BHRAY_CLASSIFY_TPW];
    uint32_t entry[BHRAY_CLASSIFY_TPW];
#pragma unroll
    for (int t = 0; t < BHRAY_CLASSIFY_TPW; t++) {
        int x, j;
        const bool valid = classify_pixel(L, (int)blockIdx.x, wave, t, lane, x, j);
        const int y = valid ? L.rows[j] : 0;
        bool predict = false;
        if (valid) {
            bool copy = false;
            if (!(L.pw == 1 && L.ph == 1)) {
                const float ppx = (float)x * L.rx, ppy = (float)y * L.ry;
                copy = fabsf(floorf(ppx) - ppx) < 0.001f && fabsf(floorf(ppy) - ppy) < 0.001f;      // ray.wgsl:193: copied, never traced
                // A pixel of the level's last column / row interpolates between a coarse pixel and ITSELF (the neighbour index clamps,
                // ray.wgsl:196-199): the shader's test is then acos(v.v / (|v| |v|)) < threshold, and whether that quotient rounds to 1
                // or to 1 + ulp (acos = NaN: "trace") depends on the last bits of v - it flips with any camera motion.  Always predicted.
                if (!copy && ((int)floorf(ppx) + 1 > L.pw - 1 || (int)floorf(ppy) + 1 > L.ph - 1)) predict = true;
            }
            if (!copy && !predict) {
                const int r = F.radius;
                const int xa = x - r < 0 ? 0 : x - r, xb = x + r > L.w - 1 ? L.w - 1 : x + r;
                const int ya = y - r < 0 ? 0 : y - r, yb = y + r > L.h - 1 ? L.h - 1 : y + r;
                for (int yy = ya; yy <= yb && !predict; yy++)
                    for (int xx = xa; xx <= xb; xx++) if (F.need[(size_t)yy * (size_t)L.w + (size_t)xx]) { predict = true; break; }
            }
        }
        want[t] = predict;
        entry[t] = ((uint32_t)L.tag << 30) | ((uint32_t)y << 15) | (uint32_t)x;
    }
    block_append(want, entry, F.queue, F.qctl, append_lds);
}
hipError_t launch_predict(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int levels, int blocks, hipStream_t s) {
    if (blocks <= 0 || nb <= 0 || levels <= 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(predict_kernel<0>, dim3(blocks, nb, levels), dim3(BHRAY_CLASSIFY_THREADS), 0, s, Pb, Fb);
    return hipGetLastError();
}

// argument-block upload (pinned host memory -> HBM) + reset of the batch's queue control words, see launch_upload
__global__ __launch_bounds__(256) void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                     uint32_t* __restrict__ zero, size_t nzero) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i < nzero) zero[i] = 0u;
}

hipError_t launch_upload(const void* pinned_src, void* dst, size_t n16, uint32_t* zero, size_t nzero, hipStream_t s) {
    const size_t n = n16 > nzero ? n16 : nzero;
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint4*)pinned_src, (uint4*)dst, n16, zero, nzero);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_classify(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int blocks, bool count, bool fixup, hipStream_t s) {
    if (blocks <= 0 || nb <= 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 g(blocks, nb), b(BHRAY_CLASSIFY_THREADS);
    if (fixup) {
        if (count) hipLaunchKernelGGL((classify_kernel<true, true>), g, b, 0, s, Pb, Fb);
        else hipLaunchKernelGGL((classify_kernel<false, true>), g, b, 0, s, Pb, Fb);
    } else {
        if (count) hipLaunchKernelGGL((classify_kernel<true, false>), g, b, 0, s, Pb, Fb);
        else hipLaunchKernelGGL((classify_kernel<false, false>), g, b, 0, s, Pb, Fb);
    }
    return hipGetLastError();
}

// Which builds of trace_kernel<METHOD, MODELS, COUNT, DENSE, EVAL> exist (x 2 integrators = 32 instantiations; 48 in round 5, 72 in round 3).
// eval: 0 the numerics contract, 1 BHRAY_F_LITERAL, 2 BHRAY_F_EVAL_FMA.  Every variant has a build for lone launches (the latency build);
// the build for a saturated device (dense) exists where throughput is reported: not for counting kernels (BHRAY_F_COUNTERS is a diagnosis
// mode: the same counts whichever build marches) and, of the two measurement-only evaluations, not for the mesh variant.  A launch that
// asks for a build that does not exist gets the latency build of the same variant: the same pixels, bit for bit (every build is).
constexpr bool trace_variant_exists(int eval, bool models, bool dense, bool count) {
    return !(count && dense) && !(eval != 0 && models && dense);
}
template <int METHOD, bool MODELS, bool COUNT, bool DENSE, int EVAL>
static const void* trace_kernel_ptr_t() {
    if constexpr (trace_variant_exists(EVAL, MODELS, DENSE, COUNT)) return (const void*)trace_kernel<METHOD, MODELS, COUNT, DENSE, EVAL>;
    else return trace_kernel_ptr_t<METHOD, MODELS, COUNT, false, EVAL>();
}
template <int METHOD, int EVAL>
static const void* trace_kernel_ptr_me(bool models, bool count, bool dense) {
    if (models) {
        if (dense) return count ? trace_kernel_ptr_t<METHOD, true, true, true, EVAL>() : trace_kernel_ptr_t<METHOD, true, false, true, EVAL>();
        return count ? trace_kernel_ptr_t<METHOD, true, true, false, EVAL>() : trace_kernel_ptr_t<METHOD, true, false, false, EVAL>();
    }
    if (dense) return count ? trace_kernel_ptr_t<METHOD, false, true, true, EVAL>() : trace_kernel_ptr_t<METHOD, false, false, true, EVAL>();
    return count ? trace_kernel_ptr_t<METHOD, false, true, false, EVAL>() : trace_kernel_ptr_t<METHOD, false, false, false, EVAL>();
}
static const void* trace_kernel_ptr(int method, bool models, bool count, bool dense, int eval) {
    if (method == 0) {
        if (eval == 1) return trace_kernel_ptr_me<0, 1>(models, count, dense);
        if (eval == 2) return trace_kernel_ptr_me<0, 2>(models, count, dense);
        return trace_kernel_ptr_me<0, 0>(models, count, dense);
    }
    if (eval == 1) return trace_kernel_ptr_me<1, 1>(models, count, dense);
    if (eval == 2) return trace_kernel_ptr_me<1, 2>(models, count, dense);
    return trace_kernel_ptr_me<1, 0>(models, count, dense);
}
static size_t trace_dyn_lds(bool models) { return models ? (size_t)BHRAY_BVH_LDS_STACK * BHRAY_TRACE_THREADS * 8 : 0; }   // trace_ray_model's traversal ring

hipError_t launch_trace(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int method, bool models, bool count, bool dense, int eval, int* err_flag,
                        int grid_blocks, hipStream_t s) {
    if (nb <= 0) return hipSuccess;
    (void)hipGetLastError();
    void* args[] = {(void*)&Pb, (void*)&Fb, (void*)&nb, (void*)&err_flag};
    return hipLaunchKernel(trace_kernel_ptr(method, models, count, dense, eval), dim3((grid_blocks * 256 + BHRAY_TRACE_THREADS - 1) / BHRAY_TRACE_THREADS),
                           dim3(BHRAY_TRACE_THREADS), args, trace_dyn_lds(models), s);
}

int trace_blocks_per_cu(int method, int has_models, int count, int dense, int eval) {
    int n = 0;
    const void* f = trace_kernel_ptr(method, has_models != 0, count != 0, dense != 0, eval);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, BHRAY_TRACE_THREADS, trace_dyn_lds(has_models != 0)) != hipSuccess || n < 1) n = 2;
    n = n * BHRAY_TRACE_THREADS / 256;            // in units of 256 threads (the grid is sized in those)
    return n < 1 ? 1 : n;
}

}  // namespace bhray
