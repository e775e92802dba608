// Can the normalisation's second transcendental go?  r = RN(1 / s), s = RN(sqrt(x)), is formed as v_rsq_f32 -> sqrt_corrected -> v_rcp_f32 + one Newton step (rcp_newton).
// Candidates that seed the reciprocal with the v_rsq_f32 result y (~1/sqrt(x) ~ 1/s) instead of a second transcendental, compared with 1.0f / s for EVERY x in [2^-95, 2^95]:
//   C1  r = fma(e, y, y),          e  = fma(-s, y, 1)                     (one Newton step from y)
//   C2  r2 = fma(e2, r, r),        e2 = fma(-s, r, 1)                     (a second step)
//   C3  C2's form with the residual taken against the first iterate in higher precision: r2 = fma(e2, r, r) where r = y + y*e computed as above (same as C2; kept for the count)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize exact_norm2.hip -o exact_norm2
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
using namespace bhray;
__global__ void k(unsigned long long* bad) {
    unsigned long long b1 = 0, b2 = 0, b3 = 0, n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = u2f((uint32_t)i);
        if (!(x >= 0x1p-95f && x <= 0x1p95f)) continue;
        n++;
        const float y = __builtin_amdgcn_rsqf(x);
        const float s0 = x * y;
        const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y, s0);       // sqrt_corrected
        const float want = 1.0f / s;
        const float e = __builtin_fmaf(-s, y, 1.0f);
        const float r = __builtin_fmaf(e, y, y);
        if (f2u(r) != f2u(want)) b1++;
        const float e2 = __builtin_fmaf(-s, r, 1.0f);
        const float r2 = __builtin_fmaf(e2, r, r);
        if (f2u(r2) != f2u(want)) b2++;
        const float e3 = __builtin_fmaf(-s, r2, 1.0f);
        const float r3 = __builtin_fmaf(e3, r2, r2);
        if (f2u(r3) != f2u(want)) b3++;
    }
    atomicAdd(&bad[0], b1); atomicAdd(&bad[1], b2); atomicAdd(&bad[2], b3); atomicAdd(&bad[3], n);
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 32); (void)hipMemset(d, 0, 32);
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, d);
    unsigned long long h[4]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("inputs %llu: one Newton step from v_rsq differs from 1/s in %llu, two steps in %llu, three in %llu\n", h[3], h[0], h[1], h[2]);
    return 0;
}
