// How well does the dispatcher keep wave slots filled when the waves of a workgroup live for very different times?  One launch with many more workgroups than fit
// (80 VGPRs: 6 waves per SIMD), every WAVE spins for its own pseudo-random time; fill = sum of the waves' lifetimes / (elapsed x 1024 SIMDs x 6 slots).
// Block sizes 64 / 128 / 256 / 512; lifetimes uniform in [t, t] (no variance), [t/2, 3t/2], [t/8, 15t/8], and "mostly short, a few long" (the trace kernels' shape).
//   hipcc --offload-arch=gfx950 -O2 wg_fill.hip -o wg_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int T>
__global__ __launch_bounds__(T) void k(unsigned long long* life, int mode, unsigned base_ticks, float* sink, int* cnt) {
    const unsigned wave = (blockIdx.x * T + threadIdx.x) >> 6;
    unsigned h = wave * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float u = (h & 0xffffff) / 16777216.0f;
    float f = 1.0f;
    if (mode == 1) f = 0.5f + u;
    if (mode == 2) f = 0.125f + 1.75f * u;
    if (mode == 3) f = (u < 0.85f) ? 0.5f : (u < 0.97f ? 1.5f : 6.0f);          // median well below the mean, a few stragglers
    const unsigned long long want = (unsigned long long)(base_ticks * f);
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    if ((threadIdx.x & 63) == 0) { const int c = atomicAdd(&cnt[0], 1) + 1; atomicMax(&cnt[1], c); }
    asm volatile("v_mov_b32 v79, 0" ::: "v79");                                     // 80 VGPRs: six waves per SIMD
    while (wall_clock64() - t0 < want) { for (int i = 0; i < 64; i++) a = a * 1.0001f + 0.5f; }
    if ((threadIdx.x & 63) == 0) { life[wave] = wall_clock64() - t0; atomicSub(&cnt[0], 1); }
    if (a == 12345.678f) sink[0] = a;
}
template <int T> void run(int mode, int blocks_per_cu_total, unsigned base_us) {
    const int waves = 256 * blocks_per_cu_total * 4;                                // the same number of waves for every block size
    const int blocks = waves * 64 / T;
    unsigned long long* life; float* sink; int* cnt;
    (void)hipMalloc(&life, (size_t)waves * 8); (void)hipMalloc(&sink, 4); (void)hipMalloc(&cnt, 8); (void)hipMemset(cnt, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(T), 0, 0, life, 0, 100u, sink, cnt); (void)hipDeviceSynchronize(); (void)hipMemset(cnt, 0, 8);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(T), 0, 0, life, mode, base_us * 100u, sink, cnt);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves); (void)hipMemcpy(h.data(), life, (size_t)waves * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double fill = (sum / 100.0) / (ms * 1e3 * 1024.0 * 6.0);
    int hc[2]; (void)hipMemcpy(hc, cnt, 8, hipMemcpyDeviceToHost);
    printf("block %3d threads, lifetimes mode %d, %d waves (%.1f per slot), base %u us: elapsed %.3f ms, fill %.3f, most resident at once %.2f per SIMD\n", T, mode, waves, waves / (1024.0 * 6.0), base_us, ms, fill, hc[1] / 1024.0);
    (void)hipFree(life); (void)hipFree(sink);
}
int main() {
    for (unsigned us : {25u, 100u, 400u, 1600u}) {
        const int per = us >= 400 ? 12 : 60;
        run<64>(0, per, us); run<256>(0, per, us); run<64>(3, per, us); run<256>(3, per, us);
    }
    return 0;
}
