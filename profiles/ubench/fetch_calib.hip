// fetch_calib.hip (MI355X): what does rocprofv3's FETCH_SIZE count for the trace kernel's access pattern?
// MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The trace kernel reads 4-byte texels and
// queue entries, scattered.  Three kernels over a cold 2 GiB buffer (> 256 MiB Infinity Cache), every byte touched at most once:
//   wide      16 B per lane, consecutive lanes consecutive            -> bytes requested = N * 16
//   narrow64  4 B per lane, one dword per 64-B line (stride 64 B)      -> lines touched * 64 B is what HBM must deliver at 64-B granularity
//   narrow128 4 B per lane, one dword per 128-B line (stride 128 B)
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void wide(const uint4* p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int STRIDE_DW>
__global__ void narrow(const unsigned* p, size_t nlines, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; unsigned acc = 0;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * STRIDE_DW];
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 2ull << 30;
    void* buf; unsigned* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes); hipDeviceSynchronize();
    const int blocks = 4096, threads = 256;
    // each kernel on its own quarter of the buffer so that nothing is cached from a previous kernel
    const size_t q = bytes / 4;
    hipLaunchKernelGGL(wide, dim3(blocks), dim3(threads), 0, 0, (const uint4*)buf, q / 16, out);
    hipLaunchKernelGGL(narrow<16>, dim3(blocks), dim3(threads), 0, 0, (const unsigned*)((char*)buf + q), q / 64, out);
    hipLaunchKernelGGL(narrow<32>, dim3(blocks), dim3(threads), 0, 0, (const unsigned*)((char*)buf + 2 * q), q / 128, out);
    hipDeviceSynchronize();
    printf("wide: %zu bytes requested; narrow64: %zu lines x 64 B = %zu; narrow128: %zu lines x 128 B = %zu (x 64 B = %zu)\n",
           q, q / 64, q, q / 128, q, q / 2);
    return 0;
}
