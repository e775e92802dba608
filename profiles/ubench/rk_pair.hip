// rk_pair.hip (MI355X): TWO rays per lane with the integrator written on 2-vectors, so that every multiply / add / fma is ONE packed
// instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) carrying both rays.  Not for FMA throughput - a packed FMA issues at half the
// rate (pk_fma.hip) - but because a SIMD cannot hide the latency of a dependent VALU instruction behind other waves (valu_latency.hip:
// one chain per wave costs the SIMD 4.5 cycles per instruction with 1 or 8 waves; only independent instructions of the SAME wave reach
// 2.2), and the RK step is full of one-chain sections (1/x, sqrt and normalisation fix-ups, the stage-to-stage dependency).  A packed
// instruction puts two independent operations into every issue slot of such a chain.  Same arithmetic per ray, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value rk_pair.hip -o rk_pair
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
using namespace bhray;
typedef float v2 __attribute__((ext_vector_type(2)));
struct P3 { v2 x, y, z; };
__device__ __forceinline__ v2 pfma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 sp(float s) { v2 r = {s, s}; return r; }
__device__ __forceinline__ P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ P3 operator*(P3 a, v2 s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ P3 pmadd3(P3 w, v2 s, P3 v) { return {pfma(w.x, s, v.x), pfma(w.y, s, v.y), pfma(w.z, s, v.z)}; }   // v + w*s
__device__ __forceinline__ v2 pdot(P3 a, P3 b) { return pfma(a.z, b.z, pfma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ P3 pcross(P3 a, P3 b) { return {pfma(a.y, b.z, -(a.z * b.y)), pfma(a.z, b.x, -(a.x * b.z)), pfma(a.x, b.y, -(a.y * b.x))}; }
__device__ __forceinline__ v2 prcp(v2 x) { v2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; const v2 e = pfma(-x, r, sp(1.0f)); return pfma(e, r, r); }
__device__ __forceinline__ v2 psqrt(v2 x) {
    v2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const v2 s0 = x * y; const v2 res = pfma(-s0, s0, x); return pfma(res, sp(0.5f) * y, s0);
}
__device__ __forceinline__ void step_pair(P3 q0, P3& pos, P3& dir, v2& h_io, v2 dist) {
    const P3 p0 = pos, d0 = dir;
    const P3 cr = pcross(p0, d0);
    const v2 h2 = pdot(cr, cr);
    const v2 d2 = dist * dist;
    const v2 s = (sp(-1.5f) * h2) * prcp((d2 * d2) * dist);
    const v2 h = h_io, sh = s * h;
    const P3 K1 = q0 * sh;
    const P3 K2 = pmadd3(K1, sp(A21), q0) * sh;
    const P3 K3 = pmadd3(K2, sp(A32), pmadd3(K1, sp(A31), q0)) * sh;
    const P3 K4 = pmadd3(K2, sp(A43), pmadd3(K2, sp(A42), pmadd3(K1, sp(A41), q0))) * sh;
    const P3 K5 = pmadd3(K4, sp(A54), pmadd3(K3, sp(A53), pmadd3(K2, sp(A52), pmadd3(K1, sp(A51), q0)))) * sh;
    const P3 K6 = pmadd3(K5, sp(A65), pmadd3(K4, sp(A64), pmadd3(K3, sp(A63), pmadd3(K2, sp(A62), pmadd3(K1, sp(A61), q0))))) * sh;
    const P3 e = pmadd3(K6, sp(DB6), pmadd3(K5, sp(DB5), pmadd3(K4, sp(DB4), pmadd3(K3, sp(DB3), K1 * sp(DB1)))));
    const float em0 = max_(max_(fabsf(e.x.x), fabsf(e.y.x)), fabsf(e.z.x)), em1 = max_(max_(fabsf(e.x.y), fabsf(e.y.y)), fabsf(e.z.y));
    const P3 ds = pmadd3(K6, sp(BA6), pmadd3(K5, sp(BA5), pmadd3(K4, sp(BA4), pmadd3(K3, sp(BA3), K1 * sp(BA1)))));
    const P3 a = d0 + ds;
    const v2 r = prcp(psqrt(pdot(a, a)));
    dir = a * r;
    pos = pmadd3(d0, h, p0);
    v2 f = {1.0001f, 1.0001f};
    if (em0 > 0.00002f) f.x = 0.9f * pow_m001_step(em0);
    if (em1 > 0.00002f) f.y = 0.9f * pow_m001_step(em1);
    h_io = h * f;
}

__device__ __forceinline__ HotParams mkhot(float outer, float R, float ny) {
    HotParams H; H.bh = f3(pin_sgpr(0.0f), pin_sgpr(0.0f), pin_sgpr(0.0f)); H.bn = f3(pin_sgpr(0.1f), pin_sgpr(ny), pin_sgpr(0.05f)); H.bn_len = pin_sgpr(1.0f);
    H.inner = pin_sgpr(2.0f); H.outer = pin_sgpr(outer); H.R = pin_sgpr(R); return H;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int steps, float x0, float outer, float R, float ny) {
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    if (MODE == 2) {                     // the scalar step with the kernel's per-step tail: previous position, culls, the rare-path branch, counters
        const HotParams H = mkhot(outer, R, ny);
        F3 pos = f3(x0 + threadIdx.x * 0.01f, 6.5f, -19.0f), dir = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
        F3 q = pos - bpos; float h = 0.15f, dist = length(q), closest = dist; int it = 0, rare_n = 0;
        for (int i = 0; i < steps; i++) {
            const F3 ppos = pos; const float ppos_dist = dist;
            next_ray_rk(q, pos, dir, h, dist);
            q = pos - bpos;
            const float cd = sqrt_rn(fdot(q, q));
            dist = cd; if (cd < closest) closest = cd;
            bool nh, nd; black_hole_culls(H, ppos, ppos_dist, h, nh, nd);
            it++;
            if (nh || nd || cd > H.R) { Hit crs; float td; if (hit_black_hole_geom(H, ppos, dir, nh, nd, 1e-8f, h, crs, td)) rare_n++; if (crs.hit) rare_n += 2; }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x + dir.y + h + closest + it + rare_n;
    } else if (MODE == 3) {              // the packed pair step with the same tail per slot
        const HotParams H = mkhot(outer, R, ny);
        F3 pa = f3(x0 + threadIdx.x * 0.01f, 6.5f, -19.0f), da = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
        F3 pb = f3(x0 + threadIdx.x * 0.01f + 0.37f, 7.5f, -19.0f), db = normalize(f3(0.01f * threadIdx.x, 0.03f, 1.0f));
        P3 pos = {{pa.x, pb.x}, {pa.y, pb.y}, {pa.z, pb.z}}, dir = {{da.x, db.x}, {da.y, db.y}, {da.z, db.z}};
        const P3 bp = {sp(0.0f), sp(0.0f), sp(0.0f)};
        P3 q = pos - bp; v2 h = sp(0.15f), dist = psqrt(pdot(q, q)), closest = dist; int it0 = 0, it1 = 0, rare_n = 0;
        for (int i = 0; i < steps; i++) {
            const P3 ppos = pos; const v2 ppos_dist = dist;
            step_pair(q, pos, dir, h, dist);
            q = pos - bp;
            const v2 cd = psqrt(pdot(q, q));
            dist = cd;
            closest.x = cd.x < closest.x ? cd.x : closest.x; closest.y = cd.y < closest.y ? cd.y : closest.y;
            bool nh0, nd0, nh1, nd1;
            black_hole_culls(H, f3(ppos.x.x, ppos.y.x, ppos.z.x), ppos_dist.x, h.x, nh0, nd0);
            black_hole_culls(H, f3(ppos.x.y, ppos.y.y, ppos.z.y), ppos_dist.y, h.y, nh1, nd1);
            it0++; it1++;
            const bool r0 = nh0 || nd0 || cd.x > H.R, r1 = nh1 || nd1 || cd.y > H.R;
            if (r0 || r1) {
                if (r0) { Hit crs; float td; if (hit_black_hole_geom(H, f3(ppos.x.x, ppos.y.x, ppos.z.x), f3(dir.x.x, dir.y.x, dir.z.x), nh0, nd0, 1e-8f, h.x, crs, td)) rare_n++; if (crs.hit) rare_n += 2; }
                if (r1) { Hit crs; float td; if (hit_black_hole_geom(H, f3(ppos.x.y, ppos.y.y, ppos.z.y), f3(dir.x.y, dir.y.y, dir.z.y), nh1, nd1, 1e-8f, h.y, crs, td)) rare_n++; if (crs.hit) rare_n += 2; }
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x.x + dir.y.y + h.x + h.y + closest.x + closest.y + pos.x.y + it0 + it1 + rare_n;
    } else if (MODE == 0) {
        F3 pos = f3(x0 + threadIdx.x * 0.01f, 2.5f, -19.0f), dir = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
        F3 q = pos - bpos; float h = 0.15f, dist = length(q), closest = dist;
        for (int i = 0; i < steps; i++) {
            next_ray_rk(q, pos, dir, h, dist);
            q = pos - bpos;
            const float cd = sqrt_rn(fdot(q, q));
            dist = cd; if (cd < closest) closest = cd;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x + dir.y + h + closest;
    } else {
        F3 pa = f3(x0 + threadIdx.x * 0.01f, 2.5f, -19.0f), da = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
        F3 pb = f3(x0 + threadIdx.x * 0.01f + 0.37f, 3.5f, -19.0f), db = normalize(f3(0.01f * threadIdx.x, 0.03f, 1.0f));
        P3 pos = {{pa.x, pb.x}, {pa.y, pb.y}, {pa.z, pb.z}}, dir = {{da.x, db.x}, {da.y, db.y}, {da.z, db.z}};
        const P3 bp = {sp(0.0f), sp(0.0f), sp(0.0f)};
        P3 q = pos - bp; v2 h = sp(0.15f), dist = psqrt(pdot(q, q)), closest = dist;
        for (int i = 0; i < steps; i++) {
            step_pair(q, pos, dir, h, dist);
            q = pos - bp;
            const v2 cd = psqrt(pdot(q, q));
            dist = cd;
            closest.x = cd.x < closest.x ? cd.x : closest.x; closest.y = cd.y < closest.y ? cd.y : closest.y;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x.x + dir.y.y + h.x + h.y + closest.x + closest.y + pos.x.y;
    }
}

template <int MODE>
void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, threads = 256, steps = 250, R = (MODE & 1) ? 2 : 1;
    float* out; (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, steps, 0.5f, 10.0f, 20.0f, -0.95f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, steps, 0.5f, 10.0f, 20.0f, -0.95f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double raysteps = (double)blocks * threads * R * steps;
    printf("%s%s, %d waves/SIMD: kernel %.3f ms, %.1f G ray-steps/s\n", (MODE & 1) ? "2 rays/lane packed" : "1 ray/lane scalar ", MODE >= 2 ? " + tail" : "", waves_per_simd, ms, raysteps / (ms * 1e-3) / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 6, 8}) run<0>(w);
    for (int w : {1, 2, 3, 4, 6}) run<1>(w);
    for (int w : {1, 4, 6, 8}) run<2>(w);
    for (int w : {1, 2, 3, 4, 6}) run<3>(w);
    return 0;
}
