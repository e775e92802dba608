// How many waves per SIMD are really resident for a kernel of N VGPRs?  8 waves per SIMD requested (2 048 blocks of 256 threads), every wave spins for 1 ms:
// elapsed / 1 ms = rounds = ceil(8 / resident).  Also: concurrently resident waves counted directly (an atomic up / down around the spin, the maximum kept).
//   hipcc --offload-arch=gfx950 -O2 census.hip -o census
#include <hip/hip_runtime.h>
#include <cstdio>
template <int REG>
__global__ __launch_bounds__(256) void k(int* cnt, float* sink) {
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    if (REG == 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
    if (REG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (REG == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (REG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (REG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (REG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if ((threadIdx.x & 63) == 0) { const int c = atomicAdd(&cnt[0], 1) + 1; atomicMax(&cnt[1], c); }
    while (wall_clock64() - t0 < 100000ull) { for (int i = 0; i < 64; i++) a = a * 1.0001f + 0.5f; }
    if ((threadIdx.x & 63) == 0) atomicSub(&cnt[0], 1);
    if (a == 12345.678f) sink[0] = a;
}
template <int REG> void run() {
    int* cnt; float* sink; (void)hipMalloc(&cnt, 8); (void)hipMalloc(&sink, 4); (void)hipMemset(cnt, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<REG>, dim3(2048), dim3(256), 0, 0, cnt, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int h[2]; (void)hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
    printf("%3d VGPRs: elapsed %.2f ms (rounds of 1 ms), most waves resident at once %d = %.2f per SIMD\n", REG, ms, h[1], h[1] / 1024.0);
}
int main() { run<32>(); run<64>(); run<72>(); run<80>(); run<96>(); run<128>(); return 0; }
