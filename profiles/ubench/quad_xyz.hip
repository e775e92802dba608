// quad_xyz.hip (MI355X): the product's RK step for a wave that runs ALONE, with the x, y, z components of ONE ray on three lanes of a
// quad (lane 4r + c holds component c of ray r; lane 4r + 3 idles along) instead of three registers of one lane - VERDICT r5 item 1.
// Every 3-vector multiply / add / fma of the Cash-Karp stages is then ONE instruction instead of three; what reduces across the
// components (the fused dots of h2, of the normalisation and of the exit distance, the cross product, e_max) fetches its operands
// from the neighbouring lanes with quad_perm DPP moves, in the product's operation order - the same bits (out[] is compared).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -I../../include quad_xyz.hip -o quad_xyz
#include "../../bhusie_amd/csrc/bhray_kernels.hip"
#include <cstdio>
#include <vector>
using namespace bhray;

// quad_perm control words: lane i of a quad reads lane sel[i]
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL> __device__ __forceinline__ float qperm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float bx(float v) { return qperm<QP(0, 0, 0, 0)>(v); }     // component x of the lane's ray, in every lane of the quad
__device__ __forceinline__ float by(float v) { return qperm<QP(1, 1, 1, 1)>(v); }
__device__ __forceinline__ float bz(float v) { return qperm<QP(2, 2, 2, 2)>(v); }
__device__ __forceinline__ float r1(float v) { return qperm<QP(1, 2, 0, 0)>(v); }     // component c + 1 (cyclic); lane 3 of a quad follows lane 2 (it carries a copy of z: no garbage in the range guards' ballots)
__device__ __forceinline__ float r2(float v) { return qperm<QP(2, 0, 1, 1)>(v); }     // component c + 2
// fdot(a, b) = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)) in every lane of the quad
__device__ __forceinline__ float qdot(float a, float b) {
    const float p = a * b;
    return fmaf(bz(a), bz(b), fmaf(by(a), by(b), bx(p)));
}
__device__ __forceinline__ float qdot_self(float a) {
    const float p = a * a;
    const float ay = by(a), az = bz(a);
    return fmaf(az, az, fmaf(ay, ay, bx(p)));
}
// fcross(a, b).c = fma(a[c+1], b[c+2], -(a[c+2] * b[c+1]))
__device__ __forceinline__ float qcross(float a, float b) { return fmaf(r1(a), r2(b), -(r2(a) * r1(b))); }

// the product's next_ray_rk (bhray_kernels.hip) on one component per lane; q0, pos, dir: this lane's component; h, dist: the ray's, in every lane
__device__ __forceinline__ void quad_rk(float q0, float& pos, float& dir, float& h_io, float dist) {
    const float p0 = pos, d0 = dir;
    const float cr = qcross(p0, d0);
    const float h2 = qdot_self(cr);
    const float s = (-1.5f * h2) * rcp_rn(pow5(dist));
    const float h = h_io, sh = s * h;
    const float K1 = q0 * sh;
    const float K2 = fmaf(K1, A21, q0) * sh;
    const float K3 = fmaf(K2, A32, fmaf(K1, A31, q0)) * sh;
    const float K4 = fmaf(K2, A43, fmaf(K2, A42, fmaf(K1, A41, q0))) * sh;
    const float K5 = fmaf(K4, A54, fmaf(K3, A53, fmaf(K2, A52, fmaf(K1, A51, q0)))) * sh;
    const float K6 = fmaf(K5, A65, fmaf(K4, A64, fmaf(K3, A63, fmaf(K2, A62, fmaf(K1, A61, q0))))) * sh;
    const float e = fmaf(K6, DB6, fmaf(K5, DB5, fmaf(K4, DB4, fmaf(K3, DB3, K1 * DB1))));
    const float ae = fabsf(e);
    const float e_max = max_(max_(bx(ae), by(ae)), bz(ae));
    const float ds = fmaf(K6, BA6, fmaf(K5, BA5, fmaf(K4, BA4, fmaf(K3, BA3, K1 * BA1))));
    const float a = d0 + ds;
    const float d = qdot_self(a);
    float r = rcp_newton(sqrt_corrected(d));
    if (__builtin_expect(__ballot(!sqrt_in_range(d)) != 0ull, 0)) r = 1.0f / sqrtf(d);
    dir = a * r;
    pos = fmaf(d0, h, p0);
    if (e_max > 0.00002f) h_io = h * (0.9f * pow_m001_step(e_max));
    else h_io = h * 1.0001f;
}

// MODE 0: the product's scalar step, one ray per lane; MODE 1: one ray per quad
template <int MODE>
__global__ void k(float* out, long long* cyc, int steps, float x0, long long* wall = nullptr) {
    const long long w0 = wall_clock64();
    const F3 bpos = f3(0.0f, 0.0f, 0.0f);
    const int ray = MODE == 0 ? (int)threadIdx.x : (int)(threadIdx.x >> 2);          // the quad build marches rays 0 .. 15 of the scalar build's 64
    const int c = threadIdx.x & 3;
    F3 pos = f3(x0 + ray * 0.01f, 6.5f, -19.0f), dir = normalize(f3(0.01f * ray, 0.02f, 1.0f));
    F3 q = pos - bpos;
    float h = 0.15f, dist = length(q), closest = dist;
    float o = 0.0f;
    long long t0, t1;
    if (MODE == 0) {
        t0 = clock64();
        for (int i = 0; i < steps; i++) {
            next_ray_rk(q, pos, dir, h, dist);
            q = pos - bpos;
            const float cd = sqrt_rn(fdot(q, q)); dist = cd; if (cd < closest) closest = cd;
        }
        t1 = clock64();
        o = pos.x + dir.y + h + closest + pos.z + dir.x;
    } else {
        float pc = c == 0 ? pos.x : c == 1 ? pos.y : pos.z, dc = c == 0 ? dir.x : c == 1 ? dir.y : dir.z, qc = c == 0 ? q.x : c == 1 ? q.y : q.z;
        const float bc = c == 0 ? bpos.x : c == 1 ? bpos.y : bpos.z;
        t0 = clock64();
        for (int i = 0; i < steps; i++) {
            quad_rk(qc, pc, dc, h, dist);
            qc = pc - bc;
            const float cd = sqrt_rn(qdot_self(qc)); dist = cd; if (cd < closest) closest = cd;
        }
        t1 = clock64();
        o = bx(pc) + by(dc) + h + closest + bz(pc) + bx(dc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (wall && (threadIdx.x & 63) == 0) wall[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = wall_clock64() - w0;      // 100 MHz, constant rate
}

template <int MODE>
void run(const char* what, int blocks, int threads, float* first16) {
    float* out; long long* cyc; long long* wall;
    const int nw = blocks * threads / 64;
    (void)hipMalloc(&out, (size_t)(blocks > 1024 ? blocks : 1024) * 256 * sizeof(float)); (void)hipMalloc(&cyc, (blocks > 1024 ? blocks : 1024) * sizeof(long long)); (void)hipMalloc(&wall, nw * sizeof(long long));
    const int steps = 3000;
    for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL((k<MODE>), dim3(1024), dim3(256), 0, 0, out, cyc, 300, 0.5f, (long long*)nullptr);     // the whole chip busy first: the clocks are up when the measured launches run
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f, wall); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, steps, 0.5f, wall);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<long long> hw(nw); (void)hipMemcpy(hw.data(), wall, nw * sizeof(long long), hipMemcpyDeviceToHost);
    long long wmax = 0, wsum = 0; for (long long v : hw) { wmax = v > wmax ? v : wmax; wsum += v; }
    long long h; (void)hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    float o[64]; (void)hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    for (int r = 0; r < 16; r++) first16[r] = MODE == 0 ? o[r] : o[4 * r];
    const double rays = (double)blocks * threads / (MODE == 0 ? 1 : 4);
    printf("%-34s %5d waves: %6.0f clock64 ticks per step (wave 0); wall clock per step: mean %.1f ns, slowest wave %.1f ns; launch %.1f ns per step; %.1f G ray-steps/s, out[0] = %.9g\n", what, nw, (double)h / steps,
           (double)wsum / nw * 10.0 / steps, (double)wmax * 10.0 / steps, ms * 1e6 / steps, rays * steps / (ms * 1e-3) / 1e9, o[0]);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(wall);
}

int main() {
    float a[16], b[16];
    run<0>("scalar: one ray per lane", 1, 64, a); run<1>("quad: x, y, z on three lanes", 1, 64, b);
    int same = 0; for (int r = 0; r < 16; r++) same += __builtin_bit_cast(unsigned, a[r]) == __builtin_bit_cast(unsigned, b[r]);
    printf("rays 0..15 after the steps: %d of 16 bit-identical between the two layouts\n", same);
    // the whole chip (256 CUs), 1 / 2 / 3 / 4 waves per SIMD: blocks of 256 threads, one / two / three / four blocks per CU
    for (int bpc = 1; bpc <= 4; bpc++) { run<0>("scalar: one ray per lane", 256 * bpc, 256, a); run<1>("quad: x, y, z on three lanes", 256 * bpc, 256, b); }
    return 0;
}
