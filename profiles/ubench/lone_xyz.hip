// lone_xyz.hip (MI355X): the product's RK step for a wave that runs ALONE, with the (x, y) components of every 3-vector operation in
// one packed instruction and z in a scalar one (pk_lone.hip: 12.65 against 16.5 ticks per vec3 FMA for a lone wave).  Same operations,
// same order, same bits; the state stays F3 at the interface (does the register allocator pair x and y without copies?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -I../../include lone_xyz.hip -o lone_xyz
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
using namespace bhray;
typedef float v2 __attribute__((ext_vector_type(2)));
struct Q3 { v2 xy; float z; };
__device__ __forceinline__ v2 pfma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 sp(float s) { v2 r = {s, s}; return r; }
__device__ __forceinline__ Q3 q3(F3 a) { Q3 r; r.xy.x = a.x; r.xy.y = a.y; r.z = a.z; return r; }
__device__ __forceinline__ F3 un(Q3 a) { return f3(a.xy.x, a.xy.y, a.z); }
__device__ __forceinline__ Q3 qmadd(Q3 w, float s, Q3 v) { Q3 r; r.xy = pfma(w.xy, sp(s), v.xy); r.z = fmaf(w.z, s, v.z); return r; }   // v + w*s
__device__ __forceinline__ Q3 qmul(Q3 w, float s) { Q3 r; r.xy = w.xy * sp(s); r.z = w.z * s; return r; }
__device__ __forceinline__ Q3 qadd(Q3 a, Q3 b) { Q3 r; r.xy = a.xy + b.xy; r.z = a.z + b.z; return r; }

template <bool PK>
__device__ __forceinline__ void step(F3 q0f, F3& pos, F3& dir, float& h_io, float dist) {
    if (!PK) { next_ray_rk(q0f, pos, dir, h_io, dist); return; }
    const F3 p0 = pos, d0 = dir;
    const F3 cr = fcross(p0, d0);
    const float h2 = fdot(cr, cr);
    const float s = (-1.5f * h2) * rcp_rn(pow5(dist));
    const float h = h_io, sh = s * h;
    const Q3 q0 = q3(q0f);
    const Q3 K1 = qmul(q0, sh);
    const Q3 K2 = qmul(qmadd(K1, A21, q0), sh);
    const Q3 K3 = qmul(qmadd(K2, A32, qmadd(K1, A31, q0)), sh);
    const Q3 K4 = qmul(qmadd(K2, A43, qmadd(K2, A42, qmadd(K1, A41, q0))), sh);
    const Q3 K5 = qmul(qmadd(K4, A54, qmadd(K3, A53, qmadd(K2, A52, qmadd(K1, A51, q0)))), sh);
    const Q3 K6 = qmul(qmadd(K5, A65, qmadd(K4, A64, qmadd(K3, A63, qmadd(K2, A62, qmadd(K1, A61, q0))))), sh);
    const Q3 e = qmadd(K6, DB6, qmadd(K5, DB5, qmadd(K4, DB4, qmadd(K3, DB3, qmul(K1, DB1)))));
    const float e_max = max_(max_(fabsf(e.xy.x), fabsf(e.xy.y)), fabsf(e.z));
    const Q3 ds = qmadd(K6, BA6, qmadd(K5, BA5, qmadd(K4, BA4, qmadd(K3, BA3, qmul(K1, BA1)))));
    const Q3 a = qadd(q3(d0), ds);
    const F3 af = un(a);
    const float d = fdot(af, af);
    float r = rcp_newton(sqrt_corrected(d));
    if (__builtin_expect(__ballot(!sqrt_in_range(d)) != 0ull, 0)) r = 1.0f / sqrtf(d);
    dir = un(qmul(a, r));
    pos = un(qmadd(q3(d0), h, q3(p0)));
    if (e_max > 0.00002f) h_io = h * (0.9f * pow_m001_step(e_max));
    else h_io = h * 1.0001f;
}

template <bool PK>
__global__ void k(float* out, long long* cyc, int steps, float x0) {
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    F3 pos = f3(x0 + threadIdx.x * 0.01f, 6.5f, -19.0f), dir = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
    F3 q = pos - bpos;
    float h = 0.15f, dist = length(q), closest = dist;
    long long t0 = clock64();
    for (int i = 0; i < steps; i++) {
        step<PK>(q, pos, dir, h, dist);
        if (PK) { const Q3 qq = qadd(q3(pos), q3(f3(-bpos.x, -bpos.y, -bpos.z))); q = un(qq); } else q = pos - bpos;
        const float cd = sqrt_rn(fdot(q, q)); dist = cd; if (cd < closest) closest = cd;
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x + dir.y + h + closest + pos.z + dir.x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool PK>
void run(const char* what, int blocks, int threads) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float)); (void)hipMalloc(&cyc, blocks * sizeof(long long));
    const int steps = 300;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PK>), dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k<PK>), dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    long long h; (void)hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    float o; (void)hipMemcpy(&o, out, sizeof o, hipMemcpyDeviceToHost);
    printf("%-26s %5d waves: %6.0f ticks per step (wave 0), %.1f G ray-steps/s, out[0] = %.9g\n", what, blocks * threads / 64, (double)h / steps,
           (double)blocks * threads * steps / (ms * 1e-3) / 1e9, o);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<false>("scalar x, y, z", 1, 64); run<true>("packed (x, y) + z", 1, 64);
    run<false>("scalar x, y, z", 1024, 256); run<true>("packed (x, y) + z", 1024, 256);
    run<false>("scalar x, y, z", 1536, 256); run<true>("packed (x, y) + z", 1536, 256);
    return 0;
}
