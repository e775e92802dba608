// ifetch.hip (MI355X): does a wave that runs ALONE on its SIMD pay for the BYTES of its instructions?  The Cash-Karp stages are
// ~80 v_fmac / v_fmamk with a 32-bit literal each (8-byte encodings); the same operations with the constant in an SGPR are 4 bytes.
// CHAINS independent accumulators, one wave, clock64 per instruction:
//   sgpr    v_fmac_f32 v, s, v          (VOP2, 4 bytes)
//   literal v_fmac_f32 v, 0x3e4ccccd, v (VOP2 + literal, 8 bytes)
//   vop3    v_fma_f32 v, v, s, v        (VOP3, 8 bytes, no literal)
//   hipcc --offload-arch=gfx950 -O3 ifetch.hip -o ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int CHAINS>
__global__ void k(float* out, long long* cyc, int iters, float kc) {
    float a[CHAINS], b[CHAINS];
    for (int c = 0; c < CHAINS; c++) { a[c] = 1.0f + threadIdx.x * 1e-3f + c; b[c] = 1e-9f * (c + 1); }
    float ks = kc; asm volatile("" : "+s"(ks));
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "s"(ks), "v"(b[c]));
                else if (KIND == 1) asm volatile("v_fmac_f32 %0, 0x3e4ccccd, %1" : "+v"(a[c]) : "v"(b[c]));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(b[c]), "s"(ks));
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int CHAINS>
void run(const char* what, int blocks, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 500;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.2f); hipDeviceSynchronize(); }
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    printf("%-8s chains %d, %4d waves: %.2f clock64 ticks per instruction\n", what, CHAINS, blocks * threads / 64, (double)h[0] / ((double)iters * 32 * CHAINS));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 1>("sgpr", 1, 64); run<1, 1>("literal", 1, 64); run<2, 1>("vop3", 1, 64);
    run<0, 3>("sgpr", 1, 64); run<1, 3>("literal", 1, 64); run<2, 3>("vop3", 1, 64);
    run<0, 6>("sgpr", 1, 64); run<1, 6>("literal", 1, 64); run<2, 6>("vop3", 1, 64);
    run<0, 3>("sgpr", 1, 256); run<1, 3>("literal", 1, 256);          // one wave per SIMD of a CU (shared instruction fetch)
    run<0, 3>("sgpr", 1024, 256); run<1, 3>("literal", 1024, 256);    // 4 waves per SIMD, whole chip
    return 0;
}
