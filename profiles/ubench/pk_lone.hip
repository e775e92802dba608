// pk_lone.hip (MI355X): packed FP32 for a wave that runs ALONE.  pk_fma.hip (round 2) showed that v_pk_fma_f32 issues at half the rate on
// a saturated SIMD (no more FMAs per clock).  A lone wave is not limited by the SIMD's FMA rate but by its own issue interval (~5.5 ticks
// per instruction at 3-way ILP, ifetch.hip) - does a packed instruction cost it the same interval as a scalar one?
//   scalar3  x, y, z as three v_fmac chains (what the integrator does today): 3 instructions per vec3 operation
//   xy_z     v_pk_fma_f32 on (x, y) + v_fmac on z: 2 instructions per vec3 operation
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value pk_lone.hip -o pk_lone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, long long* cyc, int iters, float kc) {
    float x = 1.0f + threadIdx.x * 1e-3f, y = 2.0f + threadIdx.x * 1e-3f, z = 3.0f;
    v2f xy = {x, y}, bxy = {1e-9f, 2e-9f}, kk = {kc, kc};
    float b0 = 1e-9f, b1 = 2e-9f, b2 = 3e-9f, ks = kc;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (KIND == 0) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(ks), "v"(b0));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(y) : "v"(ks), "v"(b1));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(z) : "v"(ks), "v"(b2));
            } else if (KIND == 1) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(xy) : "v"(kk), "v"(bxy));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(z) : "v"(ks), "v"(b2));
            } else if (KIND == 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(xy) : "v"(kk), "v"(bxy));
            } else {
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(xy) : "v"(kk));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(z) : "v"(ks));
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + xy.x + xy.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* what, int per_op, int blocks, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 500;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.2f); hipDeviceSynchronize(); }
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    const double ops = (double)iters * 32;
    printf("%-22s %4d waves: %.2f ticks per vec-op (%d instructions, %.2f ticks each)\n", what, blocks * threads / 64, (double)h[0] / ops, per_op, (double)h[0] / ops / per_op);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("scalar x,y,z fma", 3, 1, 64);
    run<1>("pk(x,y) + z fma", 2, 1, 64);
    run<2>("pk fma alone (1 chain)", 1, 1, 64);
    run<3>("pk(x,y) + z mul", 2, 1, 64);
    run<0>("scalar x,y,z fma", 3, 1024, 256);
    run<1>("pk(x,y) + z fma", 2, 1024, 256);
    run<0>("scalar x,y,z fma", 3, 1536, 256);     // 6 waves per SIMD
    run<1>("pk(x,y) + z fma", 2, 1536, 256);
    return 0;
}
