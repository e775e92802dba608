// Micro-benchmark (MI355X): single-wave VALU issue behaviour, to size the latency-bound coarse levels.
//   chains = number of independent v_mul_f32 dependency chains interleaved in one wave.
// Prints cycles per instruction for 1 wave on one SIMD, and for W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void k(float* out, long long* cyc, int iters, float m) {
    float a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = 1.0f + threadIdx.x * 1e-3f + c;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) a[c] = a[c] * m;
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(int blocks, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 2000;
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double n = (double)iters * 16 * CHAINS;
    printf("chains %d blocks %d threads %d: clock64 ticks/instr %.2f, wall ns/instr %.3f (%.2f cycles @2.4GHz)\n", CHAINS, blocks, threads,
           (double)h[0] / n, ms * 1e6 / n, ms * 1e6 / n * 2.4);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(1, 64); run<2>(1, 64); run<4>(1, 64); run<8>(1, 64);
    run<1>(1, 256); run<4>(1, 256);            // 4 waves = 1 per SIMD
    run<1>(1, 1024); run<4>(1, 1024);          // 16 waves = 4 per SIMD
    run<1>(256 * 8, 256); run<4>(256 * 8, 256);   // full chip, 8 waves/SIMD
    return 0;
}
