// lone_wave.hip (MI355X): where do the ~2 400 cycles per iteration of a wave that runs alone on its SIMD go?  (The tail of every
// latency-bound launch - coarse ladder levels, fix-up launches, row tiles - is such waves.)  The product's RK step with pieces
// switched off, one wave, clock64 per step:
//   GUARD   the wave-uniform range checks in front of the short 1/x and sqrt sequences (3 ballots + branches per step)
//   ADAPT   the data-dependent step-size branch (e_max > 2e-5 -> portable pow)
// Result (clock64 ticks per step, one wave): product 928; without the three range guards 730; without the step-size branch 883;
// without both 666; and without the distance 626 - the branches are 28 % of a bare step.  A product variant with ONE speculative
// branch per step (short sequences unconditionally, rare conditions OR-ed, IEEE redo behind one wave-uniform branch) was built
// on this: bit-identical, but the trace kernel's launches got no shorter (level 0+1 0.344 -> 0.319 ms, level 3 0.616 -> 0.635) and a
// saturated device lost 1.5 % - in the kernel an iteration is ~2 400 cycles of which the step is a third; reverted.
// (the step<> below restates the product's next_ray_rk with the pieces switchable; it needs the product's helper functions)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value lone_wave.hip -o lone_wave
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
using namespace bhray;

template <bool GUARD, bool ADAPT>
__device__ __forceinline__ void step(F3 q0, F3& pos, F3& dir, float& h_io, float dist) {
    const F3 p0 = pos, d0 = dir;
    const F3 cr = fcross(p0, d0);
    const float h2 = fdot(cr, cr);
    const float s = (-1.5f * h2) * (GUARD ? rcp_rn(pow5(dist)) : rcp_newton(pow5(dist)));
    const float h = h_io, sh = s * h;
    const F3 K1 = q0 * sh;
    const F3 K2 = fmadd3(K1, A21, q0) * sh;
    const F3 K3 = fmadd3(K2, A32, fmadd3(K1, A31, q0)) * sh;
    const F3 K4 = fmadd3(K2, A43, fmadd3(K2, A42, fmadd3(K1, A41, q0))) * sh;
    const F3 K5 = fmadd3(K4, A54, fmadd3(K3, A53, fmadd3(K2, A52, fmadd3(K1, A51, q0)))) * sh;
    const F3 K6 = fmadd3(K5, A65, fmadd3(K4, A64, fmadd3(K3, A63, fmadd3(K2, A62, fmadd3(K1, A61, q0))))) * sh;
    const F3 e = fmadd3(K6, DB6, fmadd3(K5, DB5, fmadd3(K4, DB4, fmadd3(K3, DB3, K1 * DB1))));
    const float e_max = max_(max_(fabsf(e.x), fabsf(e.y)), fabsf(e.z));
    const F3 ds = fmadd3(K6, BA6, fmadd3(K5, BA5, fmadd3(K4, BA4, fmadd3(K3, BA3, K1 * BA1))));
    const F3 a = d0 + ds;
    const float d = fdot(a, a);
    dir = GUARD ? fnormalize_rn(a) : a * rcp_newton(sqrt_corrected(d));
    pos = fmadd3(d0, h, p0);
    if (ADAPT) { if (e_max > 0.00002f) h_io = h * (0.9f * bh_pow_m001(e_max)); else h_io = h * 1.0001f; }
    else h_io = h * 1.0001f + e_max * 1e-30f;
}

template <bool GUARD, bool ADAPT, bool DIST>
__global__ void k(float* out, long long* cyc, int steps, float x0) {
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    F3 pos = f3(x0 + threadIdx.x * 0.01f, 6.5f, -19.0f), dir = normalize(f3(0.01f * threadIdx.x, 0.02f, 1.0f));
    F3 q = pos - bpos;
    float h = 0.15f, dist = length(q), closest = dist;
    long long t0 = clock64();
    for (int i = 0; i < steps; i++) {
        step<GUARD, ADAPT>(q, pos, dir, h, dist);
        q = pos - bpos;
        if (DIST) { const float cd = GUARD ? sqrt_rn(fdot(q, q)) : sqrt_corrected(fdot(q, q)); dist = cd; if (cd < closest) closest = cd; }
        else dist = dist + 1e-7f;
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = pos.x + dir.y + h + closest;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool GUARD, bool ADAPT, bool DIST>
void run(const char* what) {
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&cyc, sizeof(long long));
    const int steps = 300;
    const int steps_long = 30000;                    // long enough for the wall clock to mean something
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<GUARD, ADAPT, DIST>), dim3(1), dim3(64), 0, 0, out, cyc, steps, 0.5f); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<GUARD, ADAPT, DIST>), dim3(1), dim3(64), 0, 0, out, cyc, steps_long, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long hl; hipMemcpy(&hl, cyc, sizeof hl, hipMemcpyDeviceToHost);
    printf("%-44s %6.0f clock64 ticks per step; long run: %.0f ns per step, %.0f ticks per step -> tick = %.2f ns\n", what, (double)h / steps,
           ms * 1e6 / steps_long, (double)hl / steps_long, ms * 1e6 / (double)hl);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<true, true, true>("guards + adaptive h + distance (product)");
    run<false, true, true>("no range guards");
    run<true, false, true>("no adaptive-h branch");
    run<false, false, true>("no guards, no adaptive-h branch");
    run<false, false, false>("... and no distance");
    return 0;
}
