// pk_sgpr.hip (MI355X): what does a v_pk_fma_f32 cost when its coefficient is an SGPR pair read with op_sel_hi:[1,0,1] (the form the pair
// march's Cash-Karp stages compile to: the tableau in SGPRs, broadcast to both halves) against a VGPR pair, and against the scalar kernel's
// v_fmac_f32 with a 32-bit literal?  THREE independent chains per wave (the x / y / z components of a stage), whole chip, 4 and 6 waves per
// SIMD.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value pk_sgpr.hip -o pk_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float m) {
    const int lane = threadIdx.x & 63;
    if (MODE == 0) {                       // scalar, literal coefficient (VOP2 v_fmac_f32 with a 32-bit literal), 3 chains
        float a0 = 1.0f + lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2; const float c = 1e-9f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                asm volatile("v_fmac_f32 %0, 0x3f800001, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_fmac_f32 %0, 0x3f800001, %1" : "+v"(a1) : "v"(c));
                asm volatile("v_fmac_f32 %0, 0x3f800001, %1" : "+v"(a2) : "v"(c));
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2;
    } else {
        v2 a0 = {1.0f + lane * 1e-3f, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f;
        const v2 mm = {m, m}, cc = {1e-9f, 1e-9f};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (MODE == 1) {           // VGPR pair coefficient
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(mm), "v"(cc));
                } else if (MODE == 2) {    // SGPR pair, low word broadcast
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a0) : "s"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a1) : "s"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a2) : "s"(mm), "v"(cc));
                } else if (MODE == 3) {    // the stage's shape: acc = fma(K, coeff_sgpr, acc) with K another register (no self-dependence on src0)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(a1), "s"(mm));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1) : "v"(a2), "s"(mm));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a2) : "v"(a0), "s"(mm));
                } else {                   // ONE dependent chain of packed FMAs (the one-chain sections)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(mm), "v"(cc));
                }
            }
        }
        const v2 s = (a0 + a1) + a2;
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    }
}
template <int MODE>
static void run(const char* name, int blocks) {
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 3000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 48 * (blocks * 4.0 / 1024.0);
    printf("%-64s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks / 256, ms * 1e-3 * 2.4e9 / insts_per_simd);
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 1024, 1536, 2048}) {
        run<0>("v_fma_f32, literal coefficient, 3 chains", blocks);
        run<1>("v_pk_fma_f32, VGPR-pair coefficient, 3 chains", blocks);
        run<2>("v_pk_fma_f32, SGPR coefficient op_sel_hi:[1,0,1], 3 chains", blocks);
        run<3>("v_pk_fma_f32, SGPR coefficient, stage-shaped (rotating), 3 chains", blocks);
        run<4>("v_pk_fma_f32, ONE dependent chain", blocks);
    }
    return 0;
}
