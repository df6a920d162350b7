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


// ---------------------------------------------------------------------------------------------
// Entry streams of the persistent pbcd pass and of the wide passes on the device (round 4; host
// forms: spfm_schedule.cpp build_pb_stream / build_wide_stream; bitwise the same tables).
// Both sort a (row block, step)'s entries by (slot order, row) and flag the entries whose row
// the previous step touched.  Counts and positions come from the two binary searches per
// (column, row block) of the row-block stream above.  "Touched by the previous step" is a
// property of the ROW: some other column of the row sits in step b-1 -- read off the CSR image
// (a row's ~50 columns) with step_of[column].
// ---------------------------------------------------------------------------------------------

// step_of[column] = its step in the schedule
__global__ void sb_step_of_kernel(int32_t d, int nb, const int32_t* __restrict__ order,
                                  const int32_t* __restrict__ bptr, int32_t* __restrict__ step_of) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= d) return;
    int blo = 0, bhi = nb;
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    step_of[order[pos]] = blo;
}

__device__ __forceinline__ int sb_row_in_step(const int64_t* __restrict__ rptr,
                                              const int32_t* __restrict__ ridx,
                                              const int32_t* __restrict__ step_of, int32_t row,
                                              int step) {
    if (step < 0) return 0;
    int hit = 0;
    for (int64_t c = rptr[row]; c < rptr[row + 1]; ++c) hit |= (step_of[ridx[c]] == step);
    return hit;
}

// ---- pbcd stream.  thread <-> (visiting position pos, row block g), g fastest: entries of the
// column in the block
__global__ void pbs_count_kernel(int32_t d, int G, int nb, int64_t rows_per,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ bptr,
                                 const int64_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                 int32_t* __restrict__ cnt /* [G][nb][64] */,
                                 int32_t* __restrict__ first) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;
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
    cnt[((size_t)g * nb + b) * 64 + q] = (int32_t)(hi - lo);
}

// thread <-> (g, b): the slot -> (group, t) map (build_pb_stream's rule: columns in descending
// order of their entries, ties by slot, each to the group with the fewest entries so far that
// still has room, ties by group), the groups' entry counts, every slot's offset inside its group
__global__ void pbs_assign_kernel(int G, int nb, int NG, int balance,
                                  const int32_t* __restrict__ bptr, const int32_t* __restrict__ cnt,
                                  uint8_t* __restrict__ tab /* [G][nb][64] */,
                                  int32_t* __restrict__ gcnt /* [G][nb][NG+1] */,
                                  int32_t* __restrict__ soff /* [G][nb][64] */,
                                  uint8_t* __restrict__ where /* [G][nb][64] */) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)G * nb) return;
    const int b = (int)(t % nb);
    const int QM = 64 / NG;
    const int nc = bptr[b + 1] - bptr[b];
    const int nc2 = b + 2 < nb ? bptr[b + 3] - bptr[b + 2] : 0;
    const int nw = min(64, max(nc, nc2));
    const int32_t* c = cnt + (size_t)t * 64;
    uint8_t* tb = tab + (size_t)t * 64;
    int32_t* gc = gcnt + (size_t)t * (NG + 1);
    int32_t* so = soff + (size_t)t * 64;
    uint8_t* wh = where + (size_t)t * 64;
    int load[16], used[16];
    for (int r = 0; r < NG; ++r) load[r] = used[r] = 0;
    for (int z = 0; z < 64; ++z) tb[z] = 0xFF;
    if (!balance) {
        for (int q = 0; q < nw; ++q) {
            const int r = q % NG, tt = q / NG;
            tb[r * QM + tt] = (uint8_t)q;
            wh[q] = (uint8_t)((r << 3) | tt);
            load[r] += q < nc ? c[q] : 0;
        }
    } else {
        uint8_t idx[64];
        for (int q = 0; q < nw; ++q) {  // stable insertion sort, descending by count
            const int cq = q < nc ? c[q] : 0;
            int z = q;
            while (z > 0) {
                const int pz = idx[z - 1];
                if ((pz < nc ? c[pz] : 0) >= cq) break;
                idx[z] = idx[z - 1];
                --z;
            }
            idx[z] = (uint8_t)q;
        }
        for (int z = 0; z < nw; ++z) {
            const int q = idx[z];
            int best = -1;
            for (int r = 0; r < NG; ++r)
                if (used[r] < QM && (best < 0 || load[r] < load[best])) best = r;
            tb[best * QM + used[best]] = (uint8_t)q;
            wh[q] = (uint8_t)((best << 3) | used[best]);
            used[best]++;
            load[best] += q < nc ? c[q] : 0;
        }
    }
    for (int r = 0; r < NG; ++r) {
        gc[r] = load[r];
        int at = 0;
        for (int tt = 0; tt < QM; ++tt) {
            const int q = tb[r * QM + tt];
            if (q == 0xFF) continue;
            so[q] = at;
            at += q < nc ? c[q] : 0;
        }
    }
    gc[NG] = 0;
}

// thread <-> (pos, g): the column's entries in the block go to their slot's place; meta = t |
// 0x80 (row touched by the previous step)
__global__ void pbs_fill_kernel(int32_t d, int G, int nb, int NG, const int32_t* __restrict__ bptr,
                                const int32_t* __restrict__ cidx, const int32_t* __restrict__ cnt,
                                const int32_t* __restrict__ gsp, const int32_t* __restrict__ soff,
                                const uint8_t* __restrict__ where,
                                const int32_t* __restrict__ first, const int64_t* __restrict__ rptr,
                                const int32_t* __restrict__ ridx,
                                const int32_t* __restrict__ step_of, int32_t* __restrict__ src,
                                uint8_t* __restrict__ meta) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    const int b = blo, q = pos - bptr[b];
    const size_t gb = (size_t)g * nb + b;
    const int32_t m = cnt[gb * 64 + q], lo = first[t];
    const int w = where[gb * 64 + q];
    const int32_t dst = gsp[gb * (NG + 1) + (w >> 3)] + soff[gb * 64 + q];
    for (int32_t u = 0; u < m; ++u) {
        src[(size_t)dst + u] = lo + u;
        const int hz = sb_row_in_step(rptr, ridx, step_of, cidx[lo + u], b - 1);
        meta[(size_t)dst + u] = (uint8_t)((w & 7) | (hz ? 0x80 : 0));
    }
}

// Device pointers in: order[d], bptr[nb+1], cptr / cidx (CSC), rptr / ridx (CSR of the same
// matrix).  Out (device, allocated by the caller): gsp[G*nb*(NG+1) + 1], src[nnz], meta[nnz],
// tab[G*nb*64].  Scratch (step_of, counts, offsets) is allocated and freed inside.
hipError_t device_pb_stream(int64_t n, int32_t d, int64_t nnz, int G, int nb, int NG, int balance,
                            const int32_t* order, const int32_t* bptr, const int64_t* cptr,
                            const int32_t* cidx, const int64_t* rptr, const int32_t* ridx,
                            int32_t* gsp, int32_t* src, uint8_t* meta, uint8_t* tab,
                            hipStream_t stream) {
    hipError_t e;
    int32_t *cnt = nullptr, *first = nullptr, *gcnt = nullptr, *soff = nullptr, *step_of = nullptr;
    uint8_t* where = nullptr;
    void* temp = nullptr;
    auto done = [&](hipError_t rc) {
        (void)hipFree(cnt);
        (void)hipFree(first);
        (void)hipFree(gcnt);
        (void)hipFree(soff);
        (void)hipFree(step_of);
        (void)hipFree(where);
        (void)hipFree(temp);
        return rc;
    };
    if (NG > 16 || NG < 1 || nb < 1) return hipErrorInvalidValue;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    const size_t gbn = (size_t)G * nb;
    const size_t ngsp = gbn * (size_t)(NG + 1) + 1;
    const int64_t pairs = (int64_t)d * G;
    if ((e = hipMalloc(&cnt, sizeof(int32_t) * gbn * 64)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&first, sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1))) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&gcnt, sizeof(int32_t) * ngsp)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&soff, sizeof(int32_t) * gbn * 64)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&where, gbn * 64)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&step_of, sizeof(int32_t) * (size_t)d)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * gbn * 64, stream)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(gcnt, 0, sizeof(int32_t) * ngsp, stream)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(sb_step_of_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream, d,
                       nb, order, bptr, step_of);
    const unsigned blocks = (unsigned)((pairs + 255) / 256);
    hipLaunchKernelGGL(pbs_count_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, rows_per, order,
                       bptr, cptr, cidx, cnt, first);
    hipLaunchKernelGGL(pbs_assign_kernel, dim3((unsigned)((gbn + 127) / 128)), dim3(128), 0, stream, G,
                       nb, NG, balance, bptr, cnt, tab, gcnt, soff, where);
    size_t temp_bytes = 0;
    if ((e = rocprim::exclusive_scan(nullptr, temp_bytes, gcnt, gsp, (int32_t)0, ngsp,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
    if ((e = rocprim::exclusive_scan(temp, temp_bytes, gcnt, gsp, (int32_t)0, ngsp,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    hipLaunchKernelGGL(pbs_fill_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, NG, bptr, cidx,
                       cnt, gsp, soff, where, first, rptr, ridx, step_of, src, meta);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    (void)nnz;
    return done(hipSuccess);
}

// ---- wide stream: entries sorted by (row block g, step b, slot q, row); wsp[g*tot + wbase[b] + q]
// (ncols + 1 boundaries per step, tot = d + nb); hz[e] = row touched by the previous step
__global__ void wds_count_kernel(int32_t d, int G, int nb, int64_t rows_per, int tot,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ bptr,
                                 const int64_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                 int32_t* __restrict__ cnt, int32_t* __restrict__ first) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    const int32_t j = order[pos];
    const int64_t cb = cptr[j], ce = cptr[j + 1];
    const int64_t lo = rbs_lower_bound(cidx, cb, ce, (int64_t)g * rows_per);
    const int64_t hi = rbs_lower_bound(cidx, lo, ce, (int64_t)(g + 1) * rows_per);
    first[t] = (int32_t)lo;
    // wbase[b] + q = bptr[b] + b + (pos - bptr[b]) = pos + b
    cnt[(size_t)g * tot + (size_t)pos + (size_t)blo] = (int32_t)(hi - lo);
}

__global__ void wds_fill_kernel(int32_t d, int G, int nb, int tot, const int32_t* __restrict__ bptr,
                                const int32_t* __restrict__ cidx, const int32_t* __restrict__ cnt,
                                const int32_t* __restrict__ wsp, const int32_t* __restrict__ first,
                                const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
                                const int32_t* __restrict__ step_of, int32_t* __restrict__ src,
                                uint8_t* __restrict__ hz) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)d * G) return;
    const int pos = (int)(t / G), g = (int)(t % G);
    int blo = 0, bhi = nb;
    while (bhi - blo > 1) {
        const int mid = (blo + bhi) >> 1;
        if (bptr[mid] <= pos) blo = mid;
        else bhi = mid;
    }
    const size_t at = (size_t)g * tot + (size_t)pos + (size_t)blo;
    const int32_t m = cnt[at], dst = wsp[at], lo = first[t];
    for (int32_t u = 0; u < m; ++u) {
        src[(size_t)dst + u] = lo + u;
        hz[(size_t)dst + u] = (uint8_t)sb_row_in_step(rptr, ridx, step_of, cidx[lo + u], blo - 1);
    }
}

// Out (device, allocated by the caller): wsp[G*tot + 1], src[nnz], hz[nnz]; tot = d + nb.
hipError_t device_wide_stream(int64_t n, int32_t d, int64_t nnz, int G, int nb,
                              const int32_t* order, const int32_t* bptr, const int64_t* cptr,
                              const int32_t* cidx, const int64_t* rptr, const int32_t* ridx,
                              int32_t* wsp, int32_t* src, uint8_t* hz, hipStream_t stream) {
    hipError_t e;
    int32_t *cnt = nullptr, *first = nullptr, *step_of = nullptr;
    void* temp = nullptr;
    auto done = [&](hipError_t rc) {
        (void)hipFree(cnt);
        (void)hipFree(first);
        (void)hipFree(step_of);
        (void)hipFree(temp);
        return rc;
    };
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    const int tot = d + nb;
    const size_t nw = (size_t)G * (size_t)tot + 1;
    const int64_t pairs = (int64_t)d * G;
    if ((e = hipMalloc(&cnt, sizeof(int32_t) * nw)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&first, sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1))) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&step_of, sizeof(int32_t) * (size_t)d)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * nw, stream)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(sb_step_of_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream, d,
                       nb, order, bptr, step_of);
    const unsigned blocks = (unsigned)((pairs + 255) / 256);
    hipLaunchKernelGGL(wds_count_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, rows_per, tot,
                       order, bptr, cptr, cidx, cnt, first);
    size_t temp_bytes = 0;
    if ((e = rocprim::exclusive_scan(nullptr, temp_bytes, cnt, wsp, (int32_t)0, nw,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
    if ((e = rocprim::exclusive_scan(temp, temp_bytes, cnt, wsp, (int32_t)0, nw,
                                     rocprim::plus<int32_t>(), stream)) != hipSuccess)
        return done(e);
    hipLaunchKernelGGL(wds_fill_kernel, dim3(blocks), dim3(256), 0, stream, d, G, nb, tot, bptr, cidx,
                       cnt, wsp, first, rptr, ridx, step_of, src, hz);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    (void)nnz;
    return done(hipSuccess);
}

}  // namespace spfm
