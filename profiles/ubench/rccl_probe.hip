// rccl_probe.hip — what RCCL allows on a ONE-GPU box (decides how libbhray's in-library gather is tested there):
//  (1) ncclCommInitAll over one device; (2) grouped ncclSend/ncclRecv to SELF, several per group, on a user stream;
//  (3) ncclCommInitAll over a duplicated device list {0,0} (expected: refused).
// build: hipcc --offload-arch=gfx950 rccl_probe.hip -o rccl_probe -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <vector>
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define NC(x) do { ncclResult_t r = (x); if (r != ncclSuccess) { printf("NCCL %s: %s\n", #x, ncclGetErrorString(r)); return 1; } } while (0)
int main() {
    int ndev = 0; HC(hipGetDeviceCount(&ndev)); printf("devices %d\n", ndev);
    int v = 0; NC(ncclGetVersion(&v)); printf("rccl version %d\n", v);
    int devs[1] = {0}; ncclComm_t comm;
    NC(ncclCommInitAll(&comm, 1, devs));
    printf("ncclCommInitAll(1) ok\n");
    hipStream_t st; HC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t n = 1 << 20; const int K = 3;
    std::vector<float*> src(K), dst(K);
    std::vector<float> h(n);
    for (int k = 0; k < K; k++) {
        HC(hipMalloc(&src[k], n * 4)); HC(hipMalloc(&dst[k], n * 4));
        for (size_t i = 0; i < n; i++) h[i] = (float)(i % 977) + 1000.0f * k;
        HC(hipMemcpy(src[k], h.data(), n * 4, hipMemcpyHostToDevice)); HC(hipMemset(dst[k], 0, n * 4));
    }
    for (int rep = 0; rep < 3; rep++) {
        NC(ncclGroupStart());
        for (int k = 0; k < K; k++) NC(ncclSend(src[k], n, ncclFloat, 0, comm, st));
        for (int k = 0; k < K; k++) NC(ncclRecv(dst[k], n, ncclFloat, 0, comm, st));
        NC(ncclGroupEnd());
    }
    HC(hipStreamSynchronize(st));
    int bad = 0;
    for (int k = 0; k < K; k++) {
        HC(hipMemcpy(h.data(), dst[k], n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) if (h[i] != (float)(i % 977) + 1000.0f * k) { bad++; break; }
    }
    printf("self send/recv x%d per group: %s\n", K, bad ? "MISMATCH" : "ok (in-order matching)");
    hipEvent_t a, b; HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    HC(hipEventRecord(a, st));
    for (int rep = 0; rep < 20; rep++) {
        NC(ncclGroupStart());
        for (int k = 0; k < K; k++) NC(ncclSend(src[k], n, ncclFloat, 0, comm, st));
        for (int k = 0; k < K; k++) NC(ncclRecv(dst[k], n, ncclFloat, 0, comm, st));
        NC(ncclGroupEnd());
    }
    HC(hipEventRecord(b, st)); HC(hipStreamSynchronize(st));
    float ms = 0; HC(hipEventElapsedTime(&ms, a, b));
    printf("self copy: %.3f ms per group of %d x 4 MiB (%.1f GB/s)\n", ms / 20, K, K * n * 4.0 / (ms / 20 * 1e-3) / 1e9);
    NC(ncclCommDestroy(comm));
    int d2[2] = {0, 0}; ncclComm_t c2[2];
    ncclResult_t r = ncclCommInitAll(c2, 2, d2);
    printf("ncclCommInitAll({0,0}) -> %s\n", ncclGetErrorString(r));
    return bad;
}
