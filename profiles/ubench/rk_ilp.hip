// Micro-benchmark (MI355X): does instruction-level parallelism from TWO independent rays per lane (plain scalar code, no
// packing) beat more waves per SIMD for the product's own RK step?  Reports ns per ray-step at saturation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value rk_ilp.hip -o rk_ilp
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
using namespace bhray;

template <int R>
__global__ void k(float* out, int steps, float x0) {
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    F3 pos[R], dir[R], q[R]; float h[R], dist[R], closest[R]; int hits[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        pos[r] = f3(x0 + threadIdx.x * 0.01f + r * 0.37f, 2.5f + r, -19.0f); dir[r] = normalize(f3(0.01f * threadIdx.x, 0.02f + 0.01f * r, 1.0f));
        h[r] = 0.15f; q[r] = pos[r] - bpos; dist[r] = length(q[r]); closest[r] = dist[r]; hits[r] = 0;
    }
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const F3 ppos = pos[r];
            next_ray_rk(q[r], pos[r], dir[r], h[r], dist[r]);
            q[r] = pos[r] - bpos;
            const float cd = sqrt_rn(fdot(q[r], q[r]));
            dist[r] = cd;
            if (cd < closest[r]) closest[r] = cd;
            const F3 oc = ppos - bpos;
            const float oc2 = dot(oc, oc);
            const float reach = 1.05f * h[r] + 0.05f, hr = 1.0f + reach;
            float ts = h[r];
            if (oc2 <= hr * hr) hits[r] += hit_sphere(ppos, dir[r], 1.0f, bpos, 1e-8f, h[r], ts) ? 1 : 0;
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < R; r++) s += pos[r].x + dir[r].y + h[r] + closest[r] + hits[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, threads = 256, steps = 250;
    float* out; (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(threads), 0, 0, out, steps, 0.5f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(threads), 0, 0, out, steps, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double raysteps = (double)blocks * threads * R * steps;
    printf("rays/lane %d, %d waves/SIMD: kernel %.3f ms, %.1f G ray-steps/s\n", R, waves_per_simd, ms, raysteps / (ms * 1e-3) / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 6, 8}) run<1>(w);
    for (int w : {1, 2, 3, 4}) run<2>(w);
    for (int w : {1, 2}) run<3>(w);
    return 0;
}
