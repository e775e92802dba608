// How the shader clock behaves over a sequence of ~8 ms VALU-bound kernels separated by host synchronisations (the shape of bench.py's
// timed blocks), and after idle gaps.  s_memtime (shader clock) against s_memrealtime (100 MHz) inside the kernel.
//   hipcc --offload-arch=gfx950 -O2 clockramp.hip -o issue_rules_clockramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <thread>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f, c = 1.0f, m = 0.9990f, d = 1e-3f;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++)
            asm volatile("v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "+v"(a), "+v"(b), "+v"(c) : "v"(m), "v"(d));
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
    const int blocks = 256 * 6;
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipHostMalloc(&t, 16);
    auto one = [&](int iters) {
        auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, t, iters);
        (void)hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        printf(" %.0f(%.2fms)", (double)t[0] / (double)t[1] * 100.0, ms);
    };
    printf("MHz (host ms) of successive kernels, back to back with a device synchronisation between them:\n");
    for (int i = 0; i < 80; i++) { one(9000); if (i % 10 == 9) printf("\n"); }
    for (int gap_ms : {1, 5, 20, 100, 500}) {
        printf("after %d ms idle:", gap_ms);
        std::this_thread::sleep_for(std::chrono::milliseconds(gap_ms));
        for (int i = 0; i < 6; i++) one(9000);
        printf("\n");
    }
    printf("short kernels (~0.4 ms) back to back:\n");
    for (int i = 0; i < 40; i++) { one(450); if (i % 10 == 9) printf("\n"); }
    return 0;
}
