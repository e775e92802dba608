// Micro-benchmark (MI355X): cycles per Cash–Karp step of the product's own next_ray_rk (+ the per-step
// distance and culled hit test), without the ray state machine around it.  Compares with the trace
// kernel's measured ~4000 cycles per step for a lone wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value rk_step.hip -o rk_step
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
#include <vector>
using namespace bhray;

template <int MODE>
__global__ void k(float* out, long long* cyc, int steps, float x0) {
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    F3 pos = f3(x0 + threadIdx.x * 0.01f, 2.5f, -19.0f), dir = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
    F3 q = pos - bpos;
    float h = 0.15f, dist = length(q), closest = dist;
    int hits = 0;
    long long t0 = clock64();
    for (int i = 0; i < steps; i++) {
        const F3 ppos = pos;
        next_ray_rk(q, pos, dir, h, dist);
        q = pos - bpos;
        if (MODE >= 1) {
            const float cd = sqrt_rn(fdot(q, q));
            dist = cd;
            if (cd < closest) closest = cd;
        } else {
            dist = dist + 0.0f;
        }
        if (MODE >= 2) {
            const F3 oc = ppos - bpos;
            const float oc2 = dot(oc, oc);
            const float reach = 1.05f * h + 0.05f;
            const float hr = 1.0f + reach;
            float ts = h;
            if (oc2 <= hr * hr) hits += hit_sphere(ppos, dir, 1.0f, bpos, 1e-8f, h, ts) ? 1 : 0;
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x + dir.y + h + closest + hits;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(int blocks, int threads, const char* what) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    const int steps = 250;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d threads %4d: %.0f ns per step (%.0f cycles @2.4GHz), kernel %.3f ms\n", what, blocks, threads,
           ms * 1e6 / steps, ms * 1e6 / steps * 2.4, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>(1, 64, "rk only, 1 wave");
    run<1>(1, 64, "rk + distance, 1 wave");
    run<2>(1, 64, "rk + distance + cull, 1 wave");
    run<2>(256, 256, "same, 1 wave/SIMD full chip");
    run<2>(256 * 2, 256, "same, 2 waves/SIMD");
    run<2>(256 * 4, 256, "same, 4 waves/SIMD");
    run<2>(256 * 8, 256, "same, 8 waves/SIMD");
    return 0;
}
