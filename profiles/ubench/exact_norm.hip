// Exhaustive check: RN(1/RN(sqrt(x))) from v_rsq_f32 + FMA corrections against the IEEE lowering, all positive binary32 x.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_norm.hip -o exact_norm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ void cand(float x, int variant, float& s, float& r) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s0 = x * y;
    float res = __builtin_fmaf(-s0, s0, x);
    s = __builtin_fmaf(res, 0.5f * y, s0);
    if (variant >= 2) { res = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(res, 0.5f * y, s); }
    float e = __builtin_fmaf(-s, y, 1.0f);
    r = __builtin_fmaf(e, y, y);
    if (variant >= 1) { e = __builtin_fmaf(-s, r, 1.0f); r = __builtin_fmaf(e, r, r); }
}
__global__ void check(unsigned long long* bad, int variant, uint32_t lo, uint32_t hi) {
    unsigned long long bs = 0, br = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + lo; i < hi; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = u2f((uint32_t)i);
        float s, r; cand(x, variant, s, r);
        const float s_ref = __builtin_sqrtf(x), r_ref = 1.0f / s_ref;
        if (f2u(s) != f2u(s_ref)) bs++;
        if (f2u(r) != f2u(r_ref)) br++;
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (br) atomicAdd(&bad[1], br);
}
int main() {
    unsigned long long* bad; (void)hipMalloc(&bad, 16);
    const uint32_t lo = (uint32_t)(127 - 95) << 23, hi = ((uint32_t)(127 + 95) << 23) + 1;     // x in [2^-95, 2^95]
    for (int v = 0; v < 3; v++) {
        (void)hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, v, lo, hi);
        unsigned long long h[2]; (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("variant %d: sqrt mismatches %llu, 1/sqrt mismatches %llu of %u inputs\n", v, h[0], h[1], hi - lo);
    }
    return 0;
}
