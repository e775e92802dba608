// exec_half.hip (MI355X): does a wave64 VALU instruction whose upper (or lower) 32 lanes are all inactive cost one SIMD-32 pass
// instead of two?  If it does, compacting the live rays of a draining wave into one half halves its issue cost.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exec_half.hip -o exec_half
// Four independent v_fma chains per lane (the issue-rate regime, profiles/ubench/README.md), full chip, 8 waves per SIMD and 1.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(float* out, int iters, float m, unsigned long long mask) {
    const int lane = threadIdx.x & 63;
    float a0 = 1.0f + lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                a0 = __builtin_fmaf(a0, m, 1e-9f); a1 = __builtin_fmaf(a1, m, 1e-9f);
                a2 = __builtin_fmaf(a2, m, 1e-9f); a3 = __builtin_fmaf(a3, m, 1e-9f);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (a0 + a1) + (a2 + a3);
}

static void run(const char* name, unsigned long long mask, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, mask);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * (blocks * 4.0 / 1024.0);      // wave-instructions issued per SIMD
    printf("%-28s blocks %5d: %.3f ms, %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 2048}) {
        run("all 64 lanes", ~0ull, blocks);
        run("lanes 0-31 (low half)", 0x00000000ffffffffull, blocks);
        run("lanes 32-63 (high half)", 0xffffffff00000000ull, blocks);
        run("even lanes (32, both halves)", 0x5555555555555555ull, blocks);
        run("lanes 0-15", 0x000000000000ffffull, blocks);
        run("lanes 0-15 + 32-47", 0x0000ffff0000ffffull, blocks);
    }
    return 0;
}
