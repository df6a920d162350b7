// spfm_predict.hip.h -- ANOVA / all-subsets prediction, loss sums, small utilities
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
#pragma once
#include "spfm_common.hip.h"

namespace spfm {

// ------------------------------------------------------------------- predict

// (k,d) -> (d,k)
static __global__ void transpose_kernel(const double* __restrict__ in, int rows, int cols,
                                 double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * cols) return;
    const int r = (int)(t / cols), c = (int)(t % cols);
    out[(size_t)c * rows + r] = in[t];
}

// _get_output (sparse_factorization_machines.py:437-451): one wavefront per row,
// lanes over components; the order-M ANOVA kernel of (p_s, x_i) is evaluated by
// the same DP as pcd.py:23-30 (kernels.py:71-115 computes the identical value
// through closed forms on dense (n,k) intermediates).  Pt is (d,k).
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void anova_predict_kernel(
    int64_t n, int k, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
    const T* __restrict__ rval, const double* __restrict__ Pt, const double* __restrict__ lams,
    double* __restrict__ out /* accumulated */) {
    const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    double acc = 0.0;
    const int64_t b = rptr[row], e = rptr[row + 1];
    for (int s = lane; s < k; s += kWave) {
        if constexpr (M == 0) {  // all-subsets kernel, kernels.py:117-137
            double a = 1.0;
            for (int64_t ii = b; ii < e; ++ii)
                a *= 1 + (double)rval[ii] * Pt[(size_t)ridx[ii] * k + s];
            acc += a * lams[s];
        } else {
            double a[M + 1];
            a[0] = 1.0;
#pragma unroll
            for (int t = 1; t <= M; ++t) a[t] = 0.0;
            for (int64_t ii = b; ii < e; ++ii) {
                const double px = Pt[(size_t)ridx[ii] * k + s] * (double)rval[ii];
#pragma unroll
                for (int t = M; t >= 1; --t) a[t] += a[t - 1] * px;
            }
            acc += a[M] * lams[s];
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] += acc;
}

// out[i] += sum_j x_ij w_j   (safe_sparse_dot(X, w_), :442-443), thread per row
template <typename T>
__global__ __launch_bounds__(kBlock) void linear_predict_kernel(
    int64_t n, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
    const T* __restrict__ rval, const double* __restrict__ w, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii) a += (double)rval[ii] * w[ridx[ii]];
    out[i] += a;
}

template <typename T>
__global__ void store_pred_kernel(int64_t n, const double* __restrict__ pred, T* __restrict__ yy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) yy[2 * i] = (T)pred[i];
}

// (yhat, y) of a handle that shares another's data image: predictions zero, targets copied
template <typename T>
__global__ void copy_targets_kernel(int64_t n, const T* __restrict__ src_yy, T* __restrict__ yy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        yy[2 * i] = (T)0;
        yy[2 * i + 1] = src_yy[2 * i + 1];
    }
}

template <typename T>
__global__ void load_pred_kernel(int64_t n, const T* __restrict__ yy, double* __restrict__ pred) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pred[i] = (double)yy[2 * i];
}

// per-block partial sums of loss(yhat_i, y_i); finished by reduce_sum_kernel
template <typename T>
__global__ __launch_bounds__(kBlock) void loss_partial_kernel(
    int64_t n, const typename Vec2<T>::type* __restrict__ yy, int loss,
    double* __restrict__ partial) {
    __shared__ double red[16];
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock) {
        const typename Vec2<T>::type yv = yy[i];
        a += loss_dev(loss, (double)yv.x, (double)yv.y);
    }
    block_sum2(a, b, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
}


}  // namespace spfm
