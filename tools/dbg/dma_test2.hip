#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// does an exec-masked global_load_lds_dword leave the LDS words of inactive lanes alone?
__global__ void k(const float* __restrict__ g, float* out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * 64; i += 64) lds[i] = -1.f;
    __syncthreads();
    if (lane < 32) __builtin_amdgcn_global_load_lds(g + lane, lds, 4, 0, 0);             // row 0: lower half
    if (lane >= 32) __builtin_amdgcn_global_load_lds(g + 64 + lane, lds + 64, 4, 0, 0);   // row 1: upper half
    if ((lane & 31) < 30) __builtin_amdgcn_global_load_lds(g + 128 + lane, lds + 128, 4, 0, 16);  // row 2: k mask
    if (lane < 32) __builtin_amdgcn_global_load_lds(g + 192 + lane, lds + 192, 4, 0, 0);  // row 3: lower then upper
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane >= 32) __builtin_amdgcn_global_load_lds(g + 192 + lane, lds + 192, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < 4; ++t) out[t * 64 + lane] = lds[t * 64 + lane];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
    float *g, *o;
    (void)hipMalloc(&g, 1024); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o);
    std::vector<float> r(256);
    (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    for (int t = 0; t < 4; ++t) {
        printf("row %d:", t);
        for (int l = 0; l < 64; l += 1) if (l % 8 == 0 || l == 30 || l == 31 || l == 62 || l==63) printf(" [%d]=%.0f", l, r[t * 64 + l]);
        printf("\n");
    }
    return 0;
}
