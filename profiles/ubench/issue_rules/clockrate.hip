// What clock does the chip hold under a VALU-bound loop?  s_memtime (shader clock counter) against s_memrealtime (100 MHz) inside the kernel,
// per opcode mix and waves per SIMD.   hipcc --offload-arch=gfx950 -O2 clockrate.hip -o clockrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f, c = 1.0f, m = 0.9990f, d = 1e-3f;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "+v"(a), "+v"(b), "+v"(c) : "v"(m), "v"(d)); }
            if (OP == 1) { asm volatile("v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_mul_f32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(m), "v"(d)); }
            if (OP == 2) { asm volatile("v_mov_b32 %0, %3\n v_mov_b32 %1, %3\n v_mov_b32 %2, %4" : "+v"(a), "+v"(b), "+v"(c) : "v"(m), "v"(d)); }
            if (OP == 3) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0" : "+v"(a), "+v"(b), "+v"(c) : "v"(m), "v"(d)); }
        }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int OP> void run(const char* name, int w, int iters) {
    const int blocks = 256 * w;
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&t, (size_t)blocks * 16);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, t, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, t, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), t, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0; for (int i = 0; i < blocks; i++) { cs += h[2 * i]; rs += h[2 * i + 1]; }
    const double n = (double)iters * 96;
    printf("%-6s %dw iters %7d: kernel %8.3f ms | memtime ticks/instr %.3f, realtime(100MHz) ns/instr %.3f | memtime ticks per us %.1f | wall ns per wave-instr per SIMD %.4f\n",
           name, w, iters, ms, cs / blocks / n, rs / blocks * 10.0 / n, (cs / rs) * 100.0, ms * 1e6 / (n * w));
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    for (int w : {1, 4, 8}) { run<0>("fma", w, 4000); run<1>("mul", w, 4000); run<2>("mov", w, 4000); run<3>("nop", w, 4000); }
    // a long run: does the clock sag with time (power)?
    run<0>("fma", 8, 4000); run<0>("fma", 8, 40000); run<0>("fma", 8, 400000);
    run<1>("mul", 8, 400000);
    return 0;
}
