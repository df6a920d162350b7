// spfm_kernels.hip.h -- gfx950 device code of the sparse-FM proximal CD core.
//
// Execution model: the coordinate order is partitioned into
// batches of columns that share no row.  One batch = one dependent step =
//   grad kernel   (one 256-thread workgroup per column: gather A[i], (yhat,y)[i]
//                  over the column's rows, f64 wave-shuffle + LDS reduction)
//   chain kernel  (one wavefront: step size, gradient step, prox and the
//                  regularizer's cache recurrence, serial in batch order)
//   sync kernel   (one workgroup per column: scatter-update of A[i], yhat[i])
// Kernel boundaries on one stream are the only inter-workgroup synchronisation
// (cheaper on MI355X than an in-kernel grid barrier, MI355X_MICROARCH price list).
//
// Storage type T (float|double): X values, A caches, (yhat,y).  Everything that
// is reduced or fed to the prox is float64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spfm {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kMaxDegree = 6;  // SPFM_MAX_DEGREE

enum { LOSS_SQUARED = 0, LOSS_SQUARED_HINGE = 1, LOSS_LOGISTIC = 2 };
enum { REG_L1 = 0, REG_L21 = 1, REG_SQL12 = 2, REG_SQL21 = 3, REG_OMEGATI = 4, REG_OMEGACS = 5 };

// Device control block: values that change between graph replays live here, not
// in kernel arguments.
struct Ctl {
    int s;          // component of the current pcd pass (pcd.py:92)
    int pass;       // index into comp_order
    double lam;     // lams[s]
    int pad[2];
};

// One column of the schedule: where its entries live in the CSC arrays.  Built
// per schedule in visiting order, so a workgroup finds its column with one load.
struct ColDesc {
    int64_t start;
    int32_t len;
    int32_t j;
};

// Regularizer state on the device (regularizer/*.py jitclass members)
struct RegState {
    double* abs_p;   // (d)      SquaredL12/OmegaTI _abs_p
    double* norms;   // (d)      SquaredL21/OmegaCS _norms
    double* cache;   // (kMaxDegree+2) _cache ; SquaredL12/SquaredL21: cache[0]
    double* dcache;  // (kMaxDegree+2) OmegaCS _dcache (persists between calls)
};

// ------------------------------------------------------------------ helpers

// loss.py:23-24, :44-51, :67-71
__device__ __forceinline__ double dloss_dev(int loss, double p, double y) {
    if (loss == LOSS_SQUARED) return p - y;
    if (loss == LOSS_LOGISTIC) {
        const double z = p * y;
        if (z > 18.0) return -y * exp(-z);
        if (z < -18.0) return -y;
        return -y / (exp(z) + 1.0);
    }
    const double z = 1 - p * y;
    return (z > 0) ? -2 * y * z : 0.0;
}

// loss.py:20-21, :34-42, :61-65
__device__ __forceinline__ double loss_dev(int loss, double p, double y) {
    if (loss == LOSS_SQUARED) return 0.5 * ((p - y) * (p - y));
    if (loss == LOSS_LOGISTIC) {
        const double z = p * y;
        if (z > 18) return exp(-z);
        if (z < -18) return -z;
        return log(1.0 + exp(-z));
    }
    const double z = 1 - p * y;
    return (z > 0) ? z * z : 0.0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// broadcast lane `src` (wave-uniform index) of a double through SGPRs
__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// sum over `width` consecutive lanes (width = power of two <= 64)
__device__ __forceinline__ double group_sum(double v, int width) {
    for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// Deterministic block reduction of two values; result valid in every thread.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red /*>= 2*4+2*/) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
        red[2 * wave] = a;
        red[2 * wave + 1] = b;
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    double sa = 0, sb = 0;
    for (int w = 0; w < nw; ++w) {
        sa += red[2 * w];
        sb += red[2 * w + 1];
    }
    a = sa;
    b = sb;
}

template <typename T>
struct Vec2;
template <>
struct Vec2<float> {
    using type = float2;
};
template <>
struct Vec2<double> {
    using type = double2;
};

// ------------------------------------------------------------ control kernels

__global__ void begin_pass_kernel(Ctl* ctl, const int32_t* comp_order, const double* lams) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int s = comp_order[ctl->pass];
        ctl->s = s;
        ctl->lam = lams[s];
        ctl->pass += 1;
    }
}

// sum viol_col[0..d) -> out[0]  (one workgroup, fixed order => deterministic)
__global__ __launch_bounds__(kBlock) void reduce_sum_kernel(const double* __restrict__ v, int n,
                                                             double* __restrict__ out) {
    __shared__ double red[16];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += v[i];
    block_sum2(a, b, red);
    if (threadIdx.x == 0) out[0] = a;
}

// ---------------------------------------------------------- pcd: precompute

// pcd._precompute_A_all_degree (optimizer/pcd.py:15-30) for component ctl->s,
// row-parallel over the CSR image (the reference sweeps columns; the per-row
// recurrence visits the row's entries in the same ascending-column order).
// A[i, M] is never read during training (pcd.py:11-12) and is not stored.
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pcd_precompute_kernel(
    const Ctl* __restrict__ ctl, int64_t n, const int64_t* __restrict__ rptr,
    const int32_t* __restrict__ ridx, const T* __restrict__ rval, const double* __restrict__ P,
    int d, T* __restrict__ A) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double* ps = P + (size_t)ctl->s * d;
    double a[M];  // a[0] = 1 implicit at index 0
    a[0] = 1.0;
#pragma unroll
    for (int t = 1; t < M; ++t) a[t] = 0.0;
    const int64_t b = rptr[i], e = rptr[i + 1];
    for (int64_t ii = b; ii < e; ++ii) {
        const double p = ps[ridx[ii]];
        const double x = (double)rval[ii];
#pragma unroll
        for (int t = M - 1; t >= 1; --t) a[t] += a[t - 1] * p * x;
    }
#pragma unroll
    for (int t = 1; t < M; ++t) A[(size_t)i * (M - 1) + (t - 1)] = (T)a[t];
}

// ------------------------------------------------- pcd: regularizer cache

// regularizer.compute_cache_pcd(P, degree, s): squaredl12.py:42-45 (|P[s]| and
// its sum), omegati.py:62-74 (|P[s]| and the elementary symmetric polynomials
// e_0..e_M of |P[s,:]|).  One workgroup; e_t by per-thread DP over a strided
// subset, then a tree of truncated polynomial products (e_t is symmetric, so any
// partition of the features gives the same value up to rounding).
template <int M>
__global__ __launch_bounds__(kBlock) void pcd_compute_cache_kernel(const Ctl* __restrict__ ctl,
                                                                    const double* __restrict__ P,
                                                                    int d, int reg,
                                                                    double* __restrict__ cache) {
    __shared__ double sh[kBlock * (M + 1)];
    const double* ps = P + (size_t)ctl->s * d;
    const int tid = threadIdx.x;
    if (reg == REG_SQL12) {
        double a = 0, b = 0;
        for (int j = tid; j < d; j += kBlock) {
            const double v = fabs(ps[j]);
            a += v;
        }
        block_sum2(a, b, sh);
        if (tid == 0) cache[0] = a;
        return;
    }
    if (reg != REG_OMEGATI) return;
    double c[M + 1];
    c[0] = 1.0;
#pragma unroll
    for (int t = 1; t <= M; ++t) c[t] = 0.0;
    for (int j = tid; j < d; j += kBlock) {
        const double v = fabs(ps[j]);
#pragma unroll
        for (int t = M; t >= 1; --t) c[t] += c[t - 1] * v;
    }
#pragma unroll
    for (int t = 0; t <= M; ++t) sh[tid * (M + 1) + t] = c[t];
    __syncthreads();
    for (int half = kBlock / 2; half >= 1; half >>= 1) {
        if (tid < half) {
            double o[M + 1];
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u <= t; ++u)
                    acc += sh[tid * (M + 1) + u] * sh[(tid + half) * (M + 1) + (t - u)];
                o[t] = acc;
            }
#pragma unroll
            for (int t = 0; t <= M; ++t) sh[tid * (M + 1) + t] = o[t];
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int t = 0; t <= M; ++t) cache[t] = sh[t];
    }
}

// ------------------------------------------------------------- pcd: gradient

// First pass of pcd._update (optimizer/pcd.py:52-59) for every column of one
// batch: part[2q] = sum_i dloss(yhat_i, y_i) * dA_i[M-1], part[2q+1] = sum_i
// dA_i[M-1]^2 with dA from _grad_anova (pcd.py:8-12); pold[q] = P[s, j] (the
// snapshot every workgroup of the following chain reads).  One workgroup per column.
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pcd_grad_kernel(
    const Ctl* __restrict__ ctl, const ColDesc* __restrict__ desc,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, const T* __restrict__ A,
    const typename Vec2<T>::type* __restrict__ yy, const double* __restrict__ P, int d, int loss,
    double* __restrict__ part, double* __restrict__ pold) {
    __shared__ double red[16];
    const int q = blockIdx.x;
    const ColDesc cd = desc[q];
    const double p = P[(size_t)ctl->s * d + cd.j];
    const int64_t b = cd.start, e = cd.start + cd.len;
    double g = 0.0, h = 0.0;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
        const int i = cidx[ii];
        const double x = (double)cval[ii];
        const typename Vec2<T>::type yv = yy[i];
        double dprev = x;  // dA[0]
#pragma unroll
        for (int t = 1; t < M; ++t) {
            const double a = (double)A[(size_t)i * (M - 1) + (t - 1)];
            dprev = x * (a - p * dprev);
        }
        g += dloss_dev(loss, (double)yv.x, (double)yv.y) * dprev;
        h += dprev * dprev;
    }
    block_sum2(g, h, red);
    if (threadIdx.x == 0) {
        part[2 * q] = g;
        part[2 * q + 1] = h;
        pold[q] = p;
    }
}

// ---------------------------------------------------------------- pcd: chain

// Second half of pcd._update (optimizer/pcd.py:61-68) for up to 64 columns held one
// per lane: step size and gradient step are lane-parallel, then the prox and the
// regularizer's cache recurrence run as a wave-uniform serial loop over columns
// 0..last in batch order (prox_cd: l1.py:32-33, squaredl12.py:52-57,
// omegati.py:82-99,104; update_cache_pcd: squaredl12.py:47-50, omegati.py:76-80).
// Returns this lane's new coordinate.  _abs_p[j] of the reference equals |p_old|
// here because a sweep visits every j exactly once per pass.
// Rounding note: squaredl12's 2*st*dcache/(1+2*st) is evaluated as
// (2*st/(1+2*st))*dcache so that the division leaves the serial loop.
template <int M>
__device__ __forceinline__ double pcd_chain_lanes(int reg, int lane, int last, bool valid,
                                                  double p_old, double g, double h, double lam,
                                                  double mu, double beta, double gamma,
                                                  double eta, double (&cache)[M + 1]) {
    double pin = 0.0, st = 0.0;
    if (valid) {
        double inv = h * mu;
        inv += beta;
        double upd = g * lam;
        upd += beta * p_old;
        upd /= inv;
        pin = p_old - eta * upd;
        st = eta * gamma / inv;
    }
    if (reg == REG_L1) {
        const double sg = (pin > 0) ? 1.0 : ((pin < 0) ? -1.0 : 0.0);
        const double m = fabs(pin) - st;
        return sg * (m > 0.0 ? m : 0.0);
    }
    const double ab = fabs(p_old);
    double mine = 0.0;
    if (reg == REG_SQL12) {
        const double den = 1 + 2 * st;
        const double pp = pin / den;
        const double app = fabs(pp);
        const double tt = 2 * st / den;
        const double sg = (pp > 0) ? 1.0 : -1.0;
        double c0 = cache[0];
        for (int i = 0; i <= last; ++i) {
            const double ai = readlane_d(ab, i), ti = readlane_d(tt, i), pi = readlane_d(app, i);
            const double dc = c0 - ai;
            const double m = pi - ti * dc;
            const double r = (m > 0) ? m : 0.0;
            c0 = dc + r;
            if (lane == i) mine = r;
        }
        cache[0] = c0;
        return sg * mine;
    }
    // REG_OMEGATI
    {
        const double apin = fabs(pin);
        const double sg = (pin > 0) ? 1.0 : -1.0;
        for (int i = 0; i <= last; ++i) {
            const double ai = readlane_d(ab, i), si = readlane_d(st, i), pi = readlane_d(apin, i);
            double dc[M + 2];
            dc[1] = 1.0;
#pragma unroll
            for (int deg = 2; deg <= M; ++deg) {
                double v = cache[deg - 1];
                v -= dc[deg - 1] * ai;
                dc[deg] = (v < 0) ? 0.0 : v;
            }
            const double m = pi - si * dc[M];
            const double r = (m > 0) ? m : 0.0;
#pragma unroll
            for (int deg = 1; deg < M; ++deg) cache[deg] = dc[deg + 1] + dc[deg] * r;
            if (lane == i) mine = r;
        }
        return sg * mine;
    }
}

// Stand-alone chain for batches of more than 64 columns (and for the multi-kernel
// path): one wavefront, 64 columns at a time; writes P[s,j], sum_viol
// (pcd.py:119-121) and delta = p_old - p_new for the sync kernel.
template <int M>
__global__ __launch_bounds__(kWave) void pcd_chain_kernel(
    const Ctl* __restrict__ ctl, const ColDesc* __restrict__ desc, int ncols,
    double* __restrict__ P, int d, const double* __restrict__ part,
    const double* __restrict__ pold, int reg, const double* __restrict__ cache_in,
    double* __restrict__ cache_out, double mu, double beta, double gamma, double eta,
    double* __restrict__ delta, double* __restrict__ viol_col) {
    const int lane = threadIdx.x;
    const double lam = ctl->lam;
    double* ps = P + (size_t)ctl->s * d;
    double cache[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) cache[t] = cache_in[t];
    for (int base = 0; base < ncols; base += kWave) {
        const int q = base + lane;
        const bool valid = q < ncols;
        const int cnt = min(kWave, ncols - base);
        double p_old = 0.0, g = 0.0, h = 0.0;
        int j = 0;
        if (valid) {
            j = desc[q].j;
            p_old = pold[q];
            g = part[2 * q];
            h = part[2 * q + 1];
        }
        const double res =
            pcd_chain_lanes<M>(reg, lane, cnt - 1, valid, p_old, g, h, lam, mu, beta, gamma, eta,
                               cache);
        if (valid) {
            const double dl = p_old - res;
            ps[j] = res;
            delta[q] = dl;
            viol_col[j] += fabs(dl);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t <= M; ++t) cache_out[t] = cache[t];
    }
}

// ----------------------------------------------------------------- pcd: sync

// "synchronize predictions and caches" (optimizer/pcd.py:124-133) for every
// column of one batch.  A column whose coordinate did not move is skipped (the
// reference's loop is an exact no-op for update == 0).
template <typename T, int M>
__device__ __forceinline__ void pcd_sync_entry(size_t i, double x, double p_old, double upd,
                                               double lam, T* __restrict__ A,
                                               T* __restrict__ yy) {
    double dprev = x;
#pragma unroll
    for (int t = 1; t < M; ++t) {
        const size_t at = i * (M - 1) + (t - 1);
        const double a = (double)A[at];
        const double dcur = x * (a - p_old * dprev);
        A[at] = (T)(a - upd * dprev);
        dprev = dcur;
    }
    const double yh = (double)yy[2 * i];
    yy[2 * i] = (T)(yh - lam * upd * dprev);
}

template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pcd_sync_kernel(
    const Ctl* __restrict__ ctl, const ColDesc* __restrict__ desc,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ A,
    T* __restrict__ yy /* (yhat,y) pairs */, const double* __restrict__ delta,
    const double* __restrict__ pold) {
    const int q = blockIdx.x;
    const double upd = delta[q];
    if (upd == 0.0) return;
    const double p_old = pold[q];
    const double lam = ctl->lam;
    const ColDesc cd = desc[q];
    const int64_t b = cd.start, e = cd.start + cd.len;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock)
        pcd_sync_entry<T, M>((size_t)cidx[ii], (double)cval[ii], p_old, upd, lam, A, yy);
}

// Fused chain + sync for batches of at most 64 columns: every workgroup runs the
// (cheap, scalar) chain redundantly up to its own column while its other waves
// already have the column's entries and their A / yhat values in flight; only the
// last workgroup publishes the regularizer cache (double-buffered: cache_in is
// never written in this launch).  Saves one dependent kernel boundary per step.
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pcd_chain_sync_kernel(
    const Ctl* __restrict__ ctl, const ColDesc* __restrict__ desc, int ncols,
    double* __restrict__ P, int d, const double* __restrict__ part,
    const double* __restrict__ pold, int reg, const double* __restrict__ cache_in,
    double* __restrict__ cache_out, double mu, double beta, double gamma, double eta,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ A,
    T* __restrict__ yy, double* __restrict__ viol_col) {
    __shared__ double sh[2];
    constexpr int PF = 2;  // entries per thread fetched before the chain result is known
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ColDesc cd = desc[q];
    const double lam = ctl->lam;
    int ri[PF];
    double rx[PF];
    bool rv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int off = tid + u * kBlock;
        rv[u] = off < cd.len;
        ri[u] = rv[u] ? cidx[cd.start + off] : 0;
        rx[u] = rv[u] ? (double)cval[cd.start + off] : 0.0;
    }
    if (wave == 0) {
        const bool valid = lane < ncols;
        double p_old = 0.0, g = 0.0, h = 0.0;
        if (valid) {
            p_old = pold[lane];
            g = part[2 * lane];
            h = part[2 * lane + 1];
        }
        double cache[M + 1];
#pragma unroll
        for (int t = 0; t <= M; ++t) cache[t] = cache_in[t];
        const double res = pcd_chain_lanes<M>(reg, lane, q, valid, p_old, g, h, lam, mu, beta,
                                              gamma, eta, cache);
        if (lane == q) {
            const double dl = p_old - res;
            P[(size_t)ctl->s * d + cd.j] = res;
            viol_col[cd.j] += fabs(dl);
            sh[0] = dl;
            sh[1] = p_old;
        }
        if (q == ncols - 1 && lane == 0) {
#pragma unroll
            for (int t = 0; t <= M; ++t) cache_out[t] = cache[t];
        }
    }
    __syncthreads();
    const double upd = sh[0];
    if (upd == 0.0) return;
    const double p_old = sh[1];
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (rv[u]) pcd_sync_entry<T, M>((size_t)ri[u], rx[u], p_old, upd, lam, A, yy);
    for (int64_t ii = cd.start + tid + PF * kBlock; ii < cd.start + cd.len; ii += kBlock)
        pcd_sync_entry<T, M>((size_t)cidx[ii], (double)cval[ii], p_old, upd, lam, A, yy);
}

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

// Single-GPU fused form of the two kernels above (no exchange between the
// gradient and the update): one launch per step.
template <typename T>
__global__ __launch_bounds__(kBlock) void lin_fused_kernel(
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ yy, int loss,
    double* __restrict__ w, const double* __restrict__ col_norm_sq, double alpha, double mu,
    double* __restrict__ viol_col) {
    __shared__ double red[16];
    const int q = blockIdx.x;
    const int j = cols[q];
    const int64_t b = cptr[j], e = cptr[j + 1];
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    double g = 0.0, h = 0.0;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
        const int i = cidx[ii];
        const typename Vec2<T>::type yv = yy2[i];
        g += dloss_dev(loss, (double)yv.x, (double)yv.y) * (double)cval[ii];
    }
    block_sum2(g, h, red);
    const double wj = w[j];
    double upd = g;
    upd += alpha * wj;
    const double inv = mu * col_norm_sq[j] + alpha;
    upd /= inv;
    __syncthreads();
    if (threadIdx.x == 0) {
        w[j] = wj - upd;
        viol_col[j] += fabs(upd);
    }
    if (upd == 0.0) return;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
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

// -------------------------------------------------------------- pbcd kernels
// Layouts: P (d, k) f64 (the transposed copy of sparse_factorization_machines.py
// :285); A (n, (M-1)*k) storage T with A[i][(t-1)*k + s] = reference A[i, t, s];
// A[i, 0, :] = 1 and A[i, M, :] (never read: pbcd.py:12-15) are not stored.
// Thread mapping: a group of L lanes (L = power of two >= min(k, 64)) owns one
// column entry at a time; lane l handles components l, l+L, ... (C of them).

// pbcd._precompute_A_all_degree (optimizer/pbcd.py:18-33), thread per (row, s)
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pbcd_precompute_kernel(
    int64_t n, int k, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
    const T* __restrict__ rval, const double* __restrict__ P /* (d,k) */, T* __restrict__ A) {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (tid >= n * k) return;
    const int64_t i = tid / k;
    const int s = (int)(tid - i * k);
    double a[M];
    a[0] = 1.0;
#pragma unroll
    for (int t = 1; t < M; ++t) a[t] = 0.0;
    for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii) {
        const double p = P[(size_t)ridx[ii] * k + s];
        const double x = (double)rval[ii];
#pragma unroll
        for (int t = M - 1; t >= 1; --t) a[t] += a[t - 1] * p * x;
    }
#pragma unroll
    for (int t = 1; t < M; ++t) A[(size_t)i * (M - 1) * k + (size_t)(t - 1) * k + s] = (T)a[t];
}

// norms[j] = ||P[j,:]||_2 for all j (squaredl21.py:36-38, omegacs.py:64-66):
// one wave per feature.
__global__ __launch_bounds__(kBlock) void pbcd_norms_kernel(int d, int k,
                                                            const double* __restrict__ P,
                                                            double* __restrict__ norms) {
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= d) return;
    double a = 0.0;
    for (int s = lane; s < k; s += kWave) {
        const double v = P[(size_t)wave * k + s];
        a += v * v;
    }
    a = wave_sum(a);
    if (lane == 0) norms[wave] = sqrt(a);
}

// squaredl21: cache = sum(norms); omegacs: __recompute_cache_bcd(degree)
// (omegacs.py:52-62) -- e_t(norms) by the same product tree as the pcd variant.
template <int M>
__global__ __launch_bounds__(kBlock) void pbcd_compute_cache_kernel(int d, int reg, RegState rs) {
    __shared__ double sh[kBlock * (M + 1)];
    const int tid = threadIdx.x;
    if (reg == REG_SQL21) {
        double a = 0, b = 0;
        for (int j = tid; j < d; j += kBlock) a += rs.norms[j];
        block_sum2(a, b, sh);
        if (tid == 0) rs.cache[0] = a;
        return;
    }
    if (reg != REG_OMEGACS) return;
    double c[M + 1];
    c[0] = 1.0;
#pragma unroll
    for (int t = 1; t <= M; ++t) c[t] = 0.0;
    for (int j = tid; j < d; j += kBlock) {
        const double v = rs.norms[j];
#pragma unroll
        for (int t = M; t >= 1; --t) c[t] += c[t - 1] * v;
    }
#pragma unroll
    for (int t = 0; t <= M; ++t) sh[tid * (M + 1) + t] = c[t];
    __syncthreads();
    for (int half = kBlock / 2; half >= 1; half >>= 1) {
        if (tid < half) {
            double o[M + 1];
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u <= t; ++u)
                    acc += sh[tid * (M + 1) + u] * sh[(tid + half) * (M + 1) + (t - u)];
                o[t] = acc;
            }
#pragma unroll
            for (int t = 0; t <= M; ++t) sh[tid * (M + 1) + t] = o[t];
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int t = 0; t <= M; ++t) rs.cache[t] = sh[t];
    }
}

// First pass of pbcd._update (optimizer/pbcd.py:56-67): part[q*(k+1) + s] =
// sum_i dloss_i * dA[i, M-1, s]; part[q*(k+1) + k] = sum_s sum_i dA[i, M-1, s]^2.
template <typename T, int M, int L, int C>
__global__ __launch_bounds__(kBlock) void pbcd_grad_kernel(
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, const T* __restrict__ A,
    const typename Vec2<T>::type* __restrict__ yy, const double* __restrict__ P /* (d,k) */,
    int k, int loss, double* __restrict__ part) {
    constexpr int G = kBlock / L;  // entry groups per workgroup
    extern __shared__ double shm[];  // G * k + 16
    double* red = shm + (size_t)G * k;
    const int q = blockIdx.x;
    const int j = cols[q];
    const int grp = threadIdx.x / L, lane = threadIdx.x % L;
    double p[C], grad[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        p[c] = (s < k) ? P[(size_t)j * k + s] : 0.0;
        grad[c] = 0.0;
    }
    double hs = 0.0, dummy = 0.0;
    const int64_t b = cptr[j], e = cptr[j + 1];
    const size_t slab = (size_t)(M - 1) * k;
    for (int64_t ii = b + grp; ii < e; ii += G) {
        const int i = cidx[ii];
        const double x = (double)cval[ii];
        const typename Vec2<T>::type yv = yy[i];
        const double dl = dloss_dev(loss, (double)yv.x, (double)yv.y);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = lane + c * L;
            if (s < k) {
                double dprev = x;
#pragma unroll
                for (int t = 1; t < M; ++t) {
                    const double a = (double)A[(size_t)i * slab + (size_t)(t - 1) * k + s];
                    dprev = x * (a - p[c] * dprev);
                }
                grad[c] += dl * dprev;
                hs += dprev * dprev;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        if (s < k) shm[(size_t)grp * k + s] = grad[c];
    }
    block_sum2(hs, dummy, red);  // contains the __syncthreads that publishes shm
    for (int s = threadIdx.x; s < k; s += kBlock) {
        double acc = 0.0;
        for (int g2 = 0; g2 < G; ++g2) acc += shm[(size_t)g2 * k + s];
        part[(size_t)q * (k + 1) + s] = acc;
    }
    if (threadIdx.x == 0) part[(size_t)q * (k + 1) + k] = hs;
}

// Second half of pbcd._update (optimizer/pbcd.py:68-79), P[j] write-back,
// sum_viol (pbcd.py:146) and regularizer.update_cache_pbcd (pbcd.py:145) for all
// columns of a batch in batch order; one wavefront, lanes over components.
// prox_bcd: l1.py:44-45, l21.py:33-38, squaredl21.py:45-55, omegacs.py:78-106;
// update_cache_pbcd: squaredl21.py:40-43, omegacs.py:68-76.
template <int M, int C>
__global__ __launch_bounds__(kWave) void pbcd_chain_kernel(
    const int32_t* __restrict__ cols, int ncols, double* __restrict__ P /* (d,k) */, int k, int d,
    const double* __restrict__ part, const double* __restrict__ lams, int reg, RegState rs,
    int top_ncache, double mu, double beta, double gamma, double eta, double* __restrict__ delta,
    double* __restrict__ pold, double* __restrict__ viol_col) {
    const int lane = threadIdx.x;
    double lam[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * kWave;
        lam[c] = (s < k) ? lams[s] : 0.0;
    }
    // regularizer scalars, uniform across lanes
    double cache[kMaxDegree + 2], dcache[kMaxDegree + 2];
#pragma unroll
    for (int t = 0; t < kMaxDegree + 2; ++t) {
        cache[t] = (t < top_ncache) ? rs.cache[t] : 0.0;
        dcache[t] = (t < top_ncache) ? rs.dcache[t] : 0.0;
    }
    for (int q = 0; q < ncols; ++q) {
        const int j = cols[q];
        double p[C], po[C];
        double inv = part[(size_t)q * (k + 1) + k];
        inv *= mu;
        inv += beta;
        const double st0 = eta * gamma / inv;
        double sq = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = lane + c * kWave;
            if (s < k) {
                po[c] = P[(size_t)j * k + s];
                double g = part[(size_t)q * (k + 1) + s];
                g *= lam[c];
                g += beta * po[c];
                g /= inv;
                p[c] = po[c] - eta * g;
            } else {
                po[c] = 0.0;
                p[c] = 0.0;
            }
        }
        if (reg == REG_L1) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double v = p[c];
                const double sg = (v > 0) ? 1.0 : ((v < 0) ? -1.0 : 0.0);
                const double m = fabs(v) - st0;
                p[c] = sg * (m > 0.0 ? m : 0.0);
            }
        } else {
            double strength = st0;
            if (reg == REG_SQL21) {
                const double den = 1 + 2 * st0;
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] /= den;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) sq += p[c] * p[c];
            const double l2 = sqrt(wave_sum(sq));
            const double nj = (reg == REG_L21) ? 0.0 : rs.norms[j];
            if (reg == REG_SQL21) {
                if (cache[0] < nj) {  // squaredl21.py:48-49 "to avoid numerical error"
                    double a = 0.0;
                    for (int jj = lane; jj < d; jj += kWave) a += rs.norms[jj];
                    cache[0] = wave_sum(a);
                }
                const double dc = cache[0] - nj;
                strength = 2 * dc * st0 / (1.0 + 2 * st0);
            } else if (reg == REG_OMEGACS) {
#pragma unroll
                for (int deg = 2; deg <= M; ++deg) {
                    dcache[deg] = cache[deg - 1];
                    dcache[deg] -= dcache[deg - 1] * nj;
                }
                double mn = dcache[0];
#pragma unroll
                for (int t = 1; t < kMaxDegree + 2; ++t)
                    if (t < top_ncache && dcache[t] < mn) mn = dcache[t];
                if (mn < 0) {  // omegacs.py:90-96 fallback
                    if (lane == 0) rs.norms[j] = 0.0;
                    __threadfence_block();
                    // __recompute_cache_bcd(degree - 1): serial in the reference; here
                    // lane-strided DP + product across lanes via shuffles
                    double cc[kMaxDegree + 2];
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                    for (int jj = lane; jj < d; jj += kWave) {
                        const double v = (jj == j) ? 0.0 : rs.norms[jj];
#pragma unroll
                        for (int t = M - 1; t >= 1; --t) cc[t] += cc[t - 1] * v;
                    }
                    for (int m2 = 32; m2 >= 1; m2 >>= 1) {
                        double oth[kMaxDegree + 2], o[kMaxDegree + 2];
#pragma unroll
                        for (int t = 0; t < M; ++t) oth[t] = __shfl_xor(cc[t], m2, kWave);
#pragma unroll
                        for (int t = 0; t < M; ++t) {
                            double acc = 0.0;
#pragma unroll
                            for (int u = 0; u <= t; ++u) acc += cc[u] * oth[t - u];
                            o[t] = acc;
                        }
#pragma unroll
                        for (int t = 0; t < M; ++t) cc[t] = o[t];
                    }
                    // cache[1:] = 0 then DP up to degree-1 (omegacs.py:54-59)
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t)
                        cache[t] = (t < M) ? cc[t] : 0.0;
                    dcache[0] = 0.0;
                    dcache[1] = 1.0;
#pragma unroll
                    for (int deg = 2; deg <= M; ++deg) dcache[deg] = cache[M - 1];
                }
                strength = st0 * dcache[M];
            }
            if (l2 > strength) {
                const double f = 1.0 - strength / l2;
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] *= f;
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) p[c] = 0.0;
            }
        }
        // write back, violation, cache update
        double va = 0.0, sq2 = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = lane + c * kWave;
            if (s < k) {
                const double dl = po[c] - p[c];
                P[(size_t)j * k + s] = p[c];
                delta[(size_t)q * k + s] = dl;
                pold[(size_t)q * k + s] = po[c];
                va += fabs(dl);
                sq2 += p[c] * p[c];
            }
        }
        va = wave_sum(va);
        if (lane == 0) viol_col[j] += va;
        if (reg == REG_SQL21 || reg == REG_OMEGACS) {
            const double l2n = sqrt(wave_sum(sq2));
            const double nj = rs.norms[j];
            if (reg == REG_SQL21) {
                cache[0] -= nj;
                cache[0] += l2n;
            } else {
#pragma unroll
                for (int deg = 1; deg <= M; ++deg) {
                    cache[deg] += dcache[deg] * l2n;
                    cache[deg] -= dcache[deg] * nj;
                }
            }
            if (lane == 0) rs.norms[j] = l2n;
            __threadfence_block();
            if (reg == REG_OMEGACS) {
                double mn = cache[0];
#pragma unroll
                for (int t = 1; t < kMaxDegree + 2; ++t)
                    if (t < top_ncache && cache[t] < mn) mn = cache[t];
                if (mn < 0) {  // omegacs.py:75-76: __recompute_cache_bcd(degree)
                    double cc[kMaxDegree + 2];
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                    for (int jj = lane; jj < d; jj += kWave) {
                        const double v = (jj == j) ? l2n : rs.norms[jj];
#pragma unroll
                        for (int t = M; t >= 1; --t) cc[t] += cc[t - 1] * v;
                    }
                    for (int m2 = 32; m2 >= 1; m2 >>= 1) {
                        double oth[kMaxDegree + 2], o[kMaxDegree + 2];
#pragma unroll
                        for (int t = 0; t <= M; ++t) oth[t] = __shfl_xor(cc[t], m2, kWave);
#pragma unroll
                        for (int t = 0; t <= M; ++t) {
                            double acc = 0.0;
#pragma unroll
                            for (int u = 0; u <= t; ++u) acc += cc[u] * oth[t - u];
                            o[t] = acc;
                        }
#pragma unroll
                        for (int t = 0; t <= M; ++t) cc[t] = o[t];
                    }
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t)
                        cache[t] = (t <= M) ? cc[t] : 0.0;
                }
            }
        }
    }
    if (lane == 0 && (reg == REG_SQL21 || reg == REG_OMEGACS)) {
#pragma unroll
        for (int t = 0; t < kMaxDegree + 2; ++t)
            if (t < top_ncache) {
                rs.cache[t] = cache[t];
                rs.dcache[t] = dcache[t];
            }
    }
}

// "synchronize predictions and caches" (optimizer/pbcd.py:135-144)
template <typename T, int M, int L, int C>
__global__ __launch_bounds__(kBlock) void pbcd_sync_kernel(
    const int32_t* __restrict__ cols, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ A,
    T* __restrict__ yy, const double* __restrict__ lams, int k,
    const double* __restrict__ delta, const double* __restrict__ pold) {
    constexpr int G = kBlock / L;
    const int q = blockIdx.x;
    const int j = cols[q];
    const int grp = threadIdx.x / L, lane = threadIdx.x % L;
    double po[C], up[C], lu[C];
    bool any = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        po[c] = (s < k) ? pold[(size_t)q * k + s] : 0.0;
        up[c] = (s < k) ? delta[(size_t)q * k + s] : 0.0;
        lu[c] = (s < k) ? lams[s] * up[c] : 0.0;
        any |= (up[c] != 0.0);
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;  // block did not move: exact no-op
    const int64_t b = cptr[j], e = cptr[j + 1];
    const size_t slab = (size_t)(M - 1) * k;
    for (int64_t ii = b + grp; ii < e; ii += G) {
        const size_t i = (size_t)cidx[ii];
        const double x = (double)cval[ii];
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = lane + c * L;
            if (s < k) {
                double dprev = x;
#pragma unroll
                for (int t = 1; t < M; ++t) {
                    const size_t at = i * slab + (size_t)(t - 1) * k + s;
                    const double a = (double)A[at];
                    const double dcur = x * (a - po[c] * dprev);
                    A[at] = (T)(a - up[c] * dprev);
                    dprev = dcur;
                }
                acc += lu[c] * dprev;
            }
        }
        acc = group_sum(acc, L);
        if (lane == 0) yy[2 * i] = (T)((double)yy[2 * i] - acc);
    }
}

// ------------------------------------------------------------------- predict

// (k,d) -> (d,k)
__global__ void transpose_kernel(const double* __restrict__ in, int rows, int cols,
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
