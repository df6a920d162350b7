// spfm_ingest.hip -- device-side CSR -> CSC transposition (replaces the
// reference's X.tocsc(), dataset.py:119-123, for CSR input).  A separate translation unit: the
// only user of rocPRIM (one stable radix sort) in the library.
//
// The CSR image is what the caller hands over; uploaded once, it is also the engine's row-major
// image.  The CSC image is the stable sort of the entries by column id: CSR order is (row, col)
// ascending, a stable sort by column keeps the rows of a column ascending -- the canonical CSC
// the kernels and the host-side schedule builders expect.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace spfm {

// flag[0] |= 1 if the column ids of some row are not strictly ascending or out of [0, d)
__global__ void ingest_check_kernel(int64_t n, int32_t d, const int64_t* __restrict__ rptr,
                                    const int32_t* __restrict__ ridx, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = 0;
    int32_t prev = -1;
    for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii) {
        const int32_t j = ridx[ii];
        bad |= (j <= prev) | (j >= d);
        prev = j;
    }
    if (bad) atomicOr(flag, 1);
}

// pos[ii] = ii; row[ii] = the row of CSR entry ii (binary search in rptr)
__global__ void ingest_rows_kernel(int64_t n, int64_t nnz, const int64_t* __restrict__ rptr,
                                   int32_t* __restrict__ pos, int32_t* __restrict__ row) {
    const int64_t ii = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= nnz) return;
    int64_t lo = 0, hi = n;  // largest i with rptr[i] <= ii
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rptr[mid] <= ii) lo = mid;
        else hi = mid;
    }
    pos[ii] = (int32_t)ii;
    row[ii] = (int32_t)lo;
}

// CSC arrays from the sorted permutation: cidx[e] = row of CSR entry perm[e], cval likewise
template <typename T>
__global__ void ingest_gather_kernel(int64_t nnz, const int32_t* __restrict__ perm,
                                     const int32_t* __restrict__ row, const T* __restrict__ rval,
                                     int32_t* __restrict__ cidx, T* __restrict__ cval) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t p = perm[e];
    cidx[e] = row[p];
    cval[e] = rval[p];
}

// cptr[j] = first position of the sorted keys with key >= j (j = 0..d)
__global__ void ingest_cptr_kernel(int32_t d, int64_t nnz, const int32_t* __restrict__ keys,
                                   int64_t* __restrict__ cptr) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > d) return;
    int64_t lo = 0, hi = nnz;  // first index with keys[idx] >= j
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < j) lo = mid + 1;
        else hi = mid;
    }
    cptr[j] = lo;
}

// Device pointers throughout; scratch is allocated and freed inside.  Returns hipSuccess, or an
// error; *invalid = 1 when the CSR structure is not canonical (nothing else is then written).
template <typename T>
hipError_t device_csr_to_csc(int64_t n, int32_t d, int64_t nnz, const int64_t* rptr,
                             const int32_t* ridx, const T* rval, int64_t* cptr, int32_t* cidx,
                             T* cval, int* invalid, hipStream_t stream) {
    *invalid = 0;
    hipError_t e;
    int* d_flag = nullptr;
    int32_t *pos = nullptr, *row = nullptr, *keys_out = nullptr, *perm = nullptr;
    void* temp = nullptr;
    auto done = [&](hipError_t rc) {
        (void)hipFree(d_flag);
        (void)hipFree(pos);
        (void)hipFree(row);
        (void)hipFree(keys_out);
        (void)hipFree(perm);
        (void)hipFree(temp);
        return rc;
    };
    if ((e = hipMalloc(&d_flag, sizeof(int))) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), stream)) != hipSuccess) return done(e);
    if (n > 0)
        hipLaunchKernelGGL(ingest_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, n, d, rptr, ridx, d_flag);
    int h_flag = 0;
    if ((e = hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream)) !=
            hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess)
        return done(e);
    if (h_flag) {
        *invalid = 1;
        return done(hipSuccess);
    }
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
    if ((e = hipMalloc(&pos, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&row, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&keys_out, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&perm, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if (nnz > 0) {
        hipLaunchKernelGGL(ingest_rows_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0,
                           stream, n, nnz, rptr, pos, row);
        int end_bit = 1;
        while (end_bit < 31 && (1 << end_bit) < d) ++end_bit;
        size_t temp_bytes = 0;
        if ((e = rocprim::radix_sort_pairs(nullptr, temp_bytes, ridx, keys_out, pos, perm,
                                           (size_t)nnz, 0, end_bit, stream)) != hipSuccess)
            return done(e);
        if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
        if ((e = rocprim::radix_sort_pairs(temp, temp_bytes, ridx, keys_out, pos, perm, (size_t)nnz,
                                           0, end_bit, stream)) != hipSuccess)
            return done(e);
        hipLaunchKernelGGL((ingest_gather_kernel<T>), dim3((unsigned)((nnz + 255) / 256)), dim3(256),
                           0, stream, nnz, perm, row, rval, cidx, cval);
    }
    hipLaunchKernelGGL(ingest_cptr_kernel, dim3((unsigned)((d + 1 + 255) / 256)), dim3(256), 0, stream,
                       d, nnz, keys_out, cptr);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    return done(hipSuccess);
}

template hipError_t device_csr_to_csc<float>(int64_t, int32_t, int64_t, const int64_t*,
                                             const int32_t*, const float*, int64_t*, int32_t*,
                                             float*, int*, hipStream_t);
template hipError_t device_csr_to_csc<double>(int64_t, int32_t, int64_t, const int64_t*,
                                              const int32_t*, const double*, int64_t*, int32_t*,
                                              double*, int*, hipStream_t);

}  // namespace spfm
