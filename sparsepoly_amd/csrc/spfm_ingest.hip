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

// ---------------------------------------------------------------------------------------------
// Entry stream of the persistent row-block passes on the device (host form:
// spfm_schedule.cpp build_rowblock_stream): entries sorted by (row block g, step b, slot q, row).
// A column's rows ascend, so its entries in row block g are one contiguous range of the CSC
// arrays -- two binary searches per (column, row block) give every count without an atomic and
// every position deterministically: the same sp / src / lmask as the host builder.
// ---------------------------------------------------------------------------------------------
#include <rocprim/device/device_scan.hpp>

namespace spfm {

__device__ __forceinline__ int64_t rbs_lower_bound(const int32_t* __restrict__ a, int64_t lo,
                                                   int64_t hi, int64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// thread <-> (visiting position pos, row block g), g fastest
__global__ void rbs_count_kernel(int32_t d, int G, int nb, int64_t rows_per,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ bptr,
                                 const int64_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                 const uint8_t* __restrict__ skip, int32_t* __restrict__ cnt,
                                 int32_t* __restrict__ first) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;  // the step of position pos: largest b with bptr[b] <= pos
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    const int b = blo, q = pos - bptr[b];
    const int32_t j = order[pos];
    const int64_t cb = cptr[j], ce = cptr[j + 1];
    const int64_t lo = rbs_lower_bound(cidx, cb, ce, (int64_t)g * rows_per);
    const int64_t hi = rbs_lower_bound(cidx, lo, ce, (int64_t)(g + 1) * rows_per);
    first[t] = (int32_t)lo;
    int32_t m = (int32_t)(hi - lo);
    if (skip)  // entries left out of the stream (relaxed runs: those on conflict rows)
        for (int64_t ii = lo; ii < hi; ++ii) m -= skip[ii] ? 1 : 0;
    cnt[((size_t)g * nb + b) * 65 + q] = m;
}

__global__ void rbs_fill_kernel(int32_t d, int G, int nb, const int32_t* __restrict__ bptr,
                                const int64_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                const int32_t* __restrict__ order, int64_t rows_per,
                                const uint8_t* __restrict__ skip,
                                const int32_t* __restrict__ cnt, const int32_t* __restrict__ sp,
                                const int32_t* __restrict__ first, int32_t* __restrict__ src) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    const size_t at = ((size_t)g * nb + blo) * 65 + (size_t)(pos - bptr[blo]);
    const int32_t m = cnt[at], dst = sp[at], lo = first[t];
    if (!skip) {
        for (int32_t u = 0; u < m; ++u) src[(size_t)dst + u] = lo + u;
    } else {  // the range's end again, then its entries that stay
        const int32_t j = order[pos];
        const int64_t hi = rbs_lower_bound(cidx, lo, cptr[j + 1], (int64_t)(g + 1) * rows_per);
        int32_t u = 0;
        for (int64_t ii = lo; ii < hi; ++ii)
            if (!skip[ii]) src[(size_t)dst + u++] = (int32_t)ii;
    }
}

// thread <-> (g, b): slots with more than long_thresh entries
__global__ void rbs_lmask_kernel(int G, int nb, int long_thresh, const int32_t* __restrict__ bptr,
                                 const int32_t* __restrict__ cnt, uint32_t* __restrict__ lmask,
                                 int* __restrict__ any_long) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * nb) return;
    const int b = (int)(t % nb);
    const int ncols = bptr[b + 1] - bptr[b];
    uint32_t m0 = 0u, m1 = 0u;
    for (int q = 0; q < ncols && q < 64; ++q)
        if (cnt[(size_t)t * 65 + q] > long_thresh) {
            if (q < 32) m0 |= 1u << q;
            else m1 |= 1u << (q - 32);
        }
    lmask[(size_t)t * 2] = m0;
    lmask[(size_t)t * 2 + 1] = m1;
    if (m0 | m1) atomicOr(any_long, 1);
}

// Device pointers: order[d] (column of every visiting position), bptr[nb + 1], cptr / cidx,
// skip[nnz] or nullptr (CSC entries that are not part of the stream).
// Outputs (device, allocated by the caller): sp[G * nb * 65 + 1], src[nnz], lmask[G * nb * 2];
// *has_long = any long slot, *n_entries = entries in the stream.  Scratch is allocated and freed
// inside.
hipError_t device_rowblock_stream(int64_t n, int32_t d, int64_t nnz, int G, int nb, int long_thresh,
                                  const int32_t* order, const int32_t* bptr, const int64_t* cptr,
                                  const int32_t* cidx, const uint8_t* skip, int32_t* sp, int32_t* src,
                                  uint32_t* lmask, int* has_long, int64_t* n_entries,
                                  hipStream_t stream) {
    *has_long = 0;
    hipError_t e;
    int32_t *cnt = nullptr, *first = nullptr;
    int* flag = nullptr;
    void* temp = nullptr;
    auto done = [&](hipError_t rc) {
        (void)hipFree(cnt);
        (void)hipFree(first);
        (void)hipFree(flag);
        (void)hipFree(temp);
        return rc;
    };
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    const size_t nsp = (size_t)G * nb * 65 + 1;
    const int64_t pairs = (int64_t)d * G;
    if ((e = hipMalloc(&cnt, sizeof(int32_t) * nsp)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&first, sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1))) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&flag, sizeof(int))) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * nsp, stream)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(flag, 0, sizeof(int), stream)) != hipSuccess) return done(e);
    const unsigned blocks = (unsigned)((pairs + 255) / 256);
    hipLaunchKernelGGL(rbs_count_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, rows_per,
                       order, bptr, cptr, cidx, skip, cnt, first);
    size_t temp_bytes = 0;
    if ((e = rocprim::exclusive_scan(nullptr, temp_bytes, cnt, sp, (int32_t)0, nsp,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
    if ((e = rocprim::exclusive_scan(temp, temp_bytes, cnt, sp, (int32_t)0, nsp,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    hipLaunchKernelGGL(rbs_fill_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, bptr, cptr, cidx,
                       order, rows_per, skip, cnt, sp, first, src);
    hipLaunchKernelGGL(rbs_lmask_kernel, dim3((unsigned)(((int64_t)G * nb + 255) / 256)), dim3(256),
                       0, stream, G, nb, long_thresh, bptr, cnt, lmask, flag);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    int32_t total = 0;  // entries in the stream = the scan's last element
    if ((e = hipMemcpyAsync(has_long, flag, sizeof(int), hipMemcpyDeviceToHost, stream)) !=
            hipSuccess ||
        (e = hipMemcpyAsync(&total, sp + (nsp - 1), sizeof(int32_t), hipMemcpyDeviceToHost,
                            stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess)
        return done(e);
    if (n_entries) *n_entries = total;
    (void)nnz;
    return done(hipSuccess);
}

}  // namespace spfm
