// pcie_handoff.hip - what moves a 1920x1080 RGBA32F frame (33.2 MB) from HBM to pinned host memory fastest on this box?
//   (a) hipMemcpyAsync D2H on one stream (one SDMA engine)            (b) two halves on two streams
//   (c) a copy kernel storing straight into the mapped pinned buffer   (d) the same with nontemporal stores / fewer blocks
// Prints GB/s of each, 20 frames back to back.  Build: hipcc --offload-arch=gfx950 -O3 pcie_handoff.hip -o pcie_handoff
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_k(const f4v* __restrict__ s, f4v* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy_nt(const f4v* __restrict__ s, f4v* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
__global__ __launch_bounds__(256) void scatter_k(const f4v* __restrict__ s, f4v* __restrict__ d, size_t n, size_t stride) {   // 16-B stores, lanes far apart
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const size_t j = (i * stride) % n; d[j] = s[j]; }
}
int main() {
    const size_t n = (size_t)1920 * 1080, bytes = n * 16; const int F = 20;
    void *dev, *host[2];
    CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
    for (auto& h : host) { CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); memset(h, 0, bytes); }
    hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto run = [&](const char* name, auto&& body) {
        for (int w = 0; w < 3; w++) body(w);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < F; f++) body(f);
        (void)hipDeviceSynchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-58s %7.2f GB/s  %6.3f ms per frame\n", name, bytes * F / dt / 1e9, dt / F * 1e3);
    };
    run("(a) hipMemcpyAsync D2H, one stream", [&](int f) { (void)hipMemcpyAsync(host[f & 1], dev, bytes, hipMemcpyDeviceToHost, s[0]); });
    run("(b) two halves on two streams", [&](int f) { for (int k = 0; k < 2; k++) (void)hipMemcpyAsync((char*)host[f & 1] + k * bytes / 2, (char*)dev + k * bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s[k]); });
    run("(b4) four quarters on four streams", [&](int f) { for (int k = 0; k < 4; k++) (void)hipMemcpyAsync((char*)host[f & 1] + k * bytes / 4, (char*)dev + k * bytes / 4, bytes / 4, hipMemcpyDeviceToHost, s[k]); });
    for (int blocks : {16, 64, 256, 1024}) {
        char nm[96]; snprintf(nm, sizeof nm, "(c) copy kernel into mapped pinned memory, %4d blocks", blocks);
        run(nm, [&](int f) { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, s[0], (const f4v*)dev, (f4v*)host[f & 1], n); });
        snprintf(nm, sizeof nm, "(d) ... nontemporal, %4d blocks", blocks);
        run(nm, [&](int f) { hipLaunchKernelGGL(copy_nt, dim3(blocks), dim3(256), 0, s[0], (const f4v*)dev, (f4v*)host[f & 1], n); });
    }
    run("(e) scattered 16-B stores into mapped pinned memory, 256 blk", [&](int f) { hipLaunchKernelGGL(scatter_k, dim3(256), dim3(256), 0, s[0], (const f4v*)dev, (f4v*)host[f & 1], n, (size_t)4099); });
    // (f) host -> device for completeness (the consumer's queue.write_texture leg)
    run("(f) hipMemcpyAsync H2D, one stream", [&](int f) { (void)hipMemcpyAsync(dev, host[f & 1], bytes, hipMemcpyHostToDevice, s[0]); });
    return 0;
}
