#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// does global_load_lds_dword reach LDS addresses beyond 64 KB?  writes 64 floats at byte offset
// `off` of the dynamic LDS through the DMA path, reads them back with ds_read.
__global__ void k(const float* __restrict__ g, float* out, int noff, const int* offs) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 256; i += 64) lds[i] = -1.f;
    __syncthreads();
    for (int t = 0; t < noff; ++t) {
        float* dst = lds + offs[t] / 4;
        __builtin_amdgcn_global_load_lds(g + t * 64 + lane, dst, 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < noff; ++t) out[t * 64 + lane] = lds[offs[t] / 4 + lane];
}
int main() {
    std::vector<int> offs = {0, 4096, 32768, 65280, 65536, 70000 / 4 * 4, 98304, 110000 / 4 * 4, 131072, 150000 / 4 * 4};
    int n = offs.size();
    std::vector<float> h(n * 64);
    for (int i = 0; i < n * 64; ++i) h[i] = 1000.f + i;
    float *g, *o; int* d_off;
    hipMalloc(&g, h.size() * 4); hipMalloc(&o, h.size() * 4); hipMalloc(&d_off, n * 4);
    hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_off, offs.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, g, o, n, d_off);
    std::vector<float> r(h.size());
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += r[t * 64 + l] != h[t * 64 + l];
        printf("offset %7d: %s (first %.0f want %.0f)\n", offs[t], bad ? "WRONG" : "ok", r[t * 64], h[t * 64]);
    }
    return 0;
}
