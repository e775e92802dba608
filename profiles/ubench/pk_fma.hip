// pk_fma.hip (MI355X): does packed FP32 (v_pk_fma_f32: two FMAs per lane per instruction, operands in 64-bit register pairs) issue
// at the rate of v_fma_f32?  If it does, the x/y components of the integrator's 3-vectors can share instructions (2 instead of 3 per
// vector FMA, a third off the Cash-Karp stages).  Same yardstick as exec_half.hip: independent chains, the whole chip, cycles per
// wave-instruction per SIMD at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value pk_fma.hip -o pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: v_fma_f32 x4 chains; 1: v_pk_fma_f32 x4 chains (8 FMAs); 2: v_pk_mul_f32 + v_pk_add_f32 (unfused)
__global__ __launch_bounds__(256) void k(float* out, int iters, float m) {
    const int lane = threadIdx.x & 63;
    if (MODE == 0) {
        float a0 = 1.0f + lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                a0 = __builtin_fmaf(a0, m, 1e-9f); a1 = __builtin_fmaf(a1, m, 1e-9f);
                a2 = __builtin_fmaf(a2, m, 1e-9f); a3 = __builtin_fmaf(a3, m, 1e-9f);
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = (a0 + a1) + (a2 + a3);
    } else {
        float2v a0 = {1.0f + lane * 1e-3f, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
        const float2v mm = {m, m}, cc = {1e-9f, 1e-9f};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (MODE == 1) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(mm), "v"(cc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(mm), "v"(cc));
                } else {
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(mm));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(cc));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(mm));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a3) : "v"(cc));
                }
            }
        }
        const float2v s = (a0 + a1) + (a2 + a3);
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    }
}

template <int MODE>
static void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * (blocks * 4.0 / 1024.0);
    printf("%-44s blocks %5d: %.3f ms, %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 2048}) {
        run<0>("v_fma_f32 (1 FMA per lane per instruction)", blocks);
        run<1>("v_pk_fma_f32 (2 FMAs per lane per instruction)", blocks);
        run<2>("v_pk_mul_f32 / v_pk_add_f32", blocks);
    }
    return 0;
}
