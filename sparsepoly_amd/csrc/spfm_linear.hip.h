// spfm_linear.hip.h -- cd_linear step kernels (multi-kernel engine) and column norms
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
#pragma once
#include "spfm_common.hip.h"

namespace spfm {

// ------------------------------------------------------------------ cd_linear

// cd_linear._cd_linear_epoch (optimizer/cd_linear.py:8-33), gradient half:
// part[q] = sum_i dloss(yhat_i, y_i) * x_ij
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_grad_kernel(
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval,
    const typename Vec2<T>::type* __restrict__ yy, int loss, double* __restrict__ part) {
    __shared__ double red[16];
    const int q = blockIdx.x;
    const int j = cols[q];
    const int64_t b = cptr[j], e = cptr[j + 1];
    double g = 0.0, h = 0.0;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
        const int i = cidx[ii];
        const typename Vec2<T>::type yv = yy[i];
        g += dloss_dev(loss, (double)yv.x, (double)yv.y) * (double)cval[ii];
    }
    block_sum2(g, h, red);
    if (threadIdx.x == 0) part[q] = g;
}

// cd_linear.py:19-31: step, w update, sum_viol, prediction update
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_sync_kernel(
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ yy,
    const double* __restrict__ part, double* __restrict__ w,
    const double* __restrict__ col_norm_sq, double alpha, double mu,
    double* __restrict__ viol_col) {
    const int q = blockIdx.x;
    const int j = cols[q];
    const double wj = w[j];
    double upd = part[q];
    upd += alpha * wj;
    const double inv = mu * col_norm_sq[j] + alpha;
    upd /= inv;
    __syncthreads();  // every thread has read w[j] before thread 0 rewrites it
    if (threadIdx.x == 0) {
        w[j] = wj - upd;
        viol_col[j] += fabs(upd);
    }
    if (upd == 0.0) return;
    const int64_t b = cptr[j], e = cptr[j + 1];
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
        const size_t i = (size_t)cidx[ii];
        yy[2 * i] = (T)((double)yy[2 * i] - upd * (double)cval[ii]);
    }
}

// Single-GPU fused form of the two kernels above (no exchange between the gradient
// and the update): one launch per step, column found through its descriptor, the
// first two entries per thread stay in registers between the two halves.
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_fused_kernel(
    const ColDesc* __restrict__ desc, const int32_t* __restrict__ cidx,
    const T* __restrict__ cval, T* __restrict__ yy, int loss, double* __restrict__ w,
    const double* __restrict__ col_norm_sq, double alpha, double mu,
    double* __restrict__ viol_col) {
    __shared__ double red[16];
    constexpr int PF = 2;
    const ColDesc cd = desc[blockIdx.x];
    const int j = cd.j;
    const int tid = threadIdx.x;
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const double wj = w[j];
    const double cn = col_norm_sq[j];
    int ri[PF];
    double rx[PF], ryh[PF];
    bool rv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int off = tid + u * kBlock;
        rv[u] = off < cd.len;
        ri[u] = rv[u] ? cidx[cd.start + off] : 0;
        rx[u] = rv[u] ? (double)cval[cd.start + off] : 0.0;
    }
    double g = 0.0, h = 0.0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const typename Vec2<T>::type yv = yy2[ri[u]];
        ryh[u] = (double)yv.x;
        g += rv[u] ? dloss_dev(loss, (double)yv.x, (double)yv.y) * rx[u] : 0.0;
    }
    for (int64_t ii = cd.start + tid + PF * kBlock; ii < cd.start + cd.len; ii += kBlock) {
        const typename Vec2<T>::type yv = yy2[cidx[ii]];
        g += dloss_dev(loss, (double)yv.x, (double)yv.y) * (double)cval[ii];
    }
    block_sum2(g, h, red);
    double upd = g;
    upd += alpha * wj;
    const double inv = mu * cn + alpha;
    upd /= inv;
    if (tid == 0) {
        w[j] = wj - upd;
        viol_col[j] += fabs(upd);
    }
    if (upd == 0.0) return;
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (rv[u]) yy[2 * (size_t)ri[u]] = (T)(ryh[u] - upd * rx[u]);
    for (int64_t ii = cd.start + tid + PF * kBlock; ii < cd.start + cd.len; ii += kBlock) {
        const size_t i = (size_t)cidx[ii];
        yy[2 * i] = (T)((double)yy[2 * i] - upd * (double)cval[ii]);
    }
}

// col_norm_sq = row_norms(X.T, squared=True) (sparse_factorization_machines.py:409)
template <typename T>
__global__ __launch_bounds__(kBlock) void col_norm_kernel(int d, const int64_t* __restrict__ cptr,
                                                          const T* __restrict__ cval,
                                                          double* __restrict__ out) {
    // one wave per column
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= d) return;
    double a = 0.0;
    for (int64_t ii = cptr[wave] + lane; ii < cptr[wave + 1]; ii += kWave) {
        const double x = (double)cval[ii];
        a += x * x;
    }
    a = wave_sum(a);
    if (lane == 0) out[wave] = a;
}


}  // namespace spfm
