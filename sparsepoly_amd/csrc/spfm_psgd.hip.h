// spfm_psgd.hip.h -- minibatch proximal SGD (reference optimizer/psgd.py:125-199)
// Part of the gfx950 device code of the sparse-FM solvers; see spfm_kernels.hip.h.
//
// Unlike pcd/pbcd this solver is throughput-bound: one minibatch is
//   psgd_grad_kernel    rows of the batch in parallel (one L-lane group per row, lane =
//                       component): ANOVA DP, prediction, dloss, gradient scatter with
//                       hardware f64 atomics into grad_P (n_orders, d, k) / grad_w (d)
//   psgd_update_kernel  one dense HBM pass over P: SGD step, shrink, grad reset and -- for
//                       l1 / l21 -- the prox, fused (psgd.py:94-122)
// and, for squaredl12 / squaredl21, whose prox needs the support of a whole vector
// (regularizer/utils.py:27-70), a few psgd_mich_* passes: the reference's randomised
// pivot search is replaced by the monotone fixed-point iteration
//   G <- {i : |p_i| >= tau(G)},  tau(G) = 2 c S_G / (1 + 2 c |G|)
// which is monotone from any start and ends at the same support the reference finds; each
// sweep is one coalesced read of P, and consecutive minibatches warm-start each other.
#pragma once
#include "spfm_common.hip.h"

namespace spfm {

// One minibatch as the kernels see it when a whole run of minibatches is replayed from a
// hipGraph (identical kernel arguments for every batch): the host tabulates the epoch
// (psgd.py:9-22 step sizes included) and two device counters walk through the table --
// the gradient kernel reads idx[0] and leaves idx[1] = that batch for the update kernel,
// which leaves idx[0] = batch + 1 when it is done.  Kernels on one stream run one after
// the other, so a counter is never written while a kernel that reads it is running.
struct PsgdBatch {
    long long pos;  // first position of the batch in the sample order
    int B;          // rows in the batch
    int pad;
    double cp, denp, strength, cw, denw;  // eta_P/B, 1+eta_P*beta, prox strength, eta_w/B, 1+eta_w*alpha
};

constexpr int kPsgdNB = 1024;  // workgroups of the dense passes (partials per vector)
constexpr int kPsgdMaxC = 4;   // component chunks per lane (k <= 4 * 64)
constexpr int kPsgdRC = 64;    // row entries per chunk: all their P loads are in flight together

// One chunk (<= kPsgdRC entries) of a row, spread over the L lanes of its group: lane ln
// holds entry tb + t*L + ln in slot t.  Entries are broadcast with group shuffles, so the
// dependent P loads of the whole chunk are issued back to back (one round trip) and stay
// in registers for the gradient pass.  With one wave per SIMD resident (a minibatch is only
// ~1000 waves) registers are free and memory round trips are the whole cost.
template <typename T, int L>
struct PsgdChunk {
    static constexpr int NT = kPsgdRC / L;
    int jl[NT];
    double xl[NT];
    double p[kPsgdRC];
    int cnt;

    __device__ __forceinline__ void load_entries(int64_t tb, int64_t hi, int ln,
                                                 const int32_t* __restrict__ ridx,
                                                 const T* __restrict__ rval) {
        cnt = (int)((hi - tb < kPsgdRC) ? (hi - tb) : kPsgdRC);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t e = tb + t * L + ln;
            const bool in = e < hi;
            jl[t] = in ? ridx[e] : 0;
            xl[t] = in ? (double)rval[e] : 0.0;  // 0 beyond the row: every update is a no-op
        }
    }
    __device__ __forceinline__ int col(int u) const { return __shfl(jl[u / L], u % L, L); }
    __device__ __forceinline__ double val(int u) const { return __shfl(xl[u / L], u % L, L); }

    __device__ __forceinline__ void load_p(const double* __restrict__ Pcol, int k, bool act) {
#pragma unroll
        for (int u = 0; u < kPsgdRC; ++u) {
            const int j = col(u);
            p[u] = (act && u < cnt) ? Pcol[(size_t)j * k] : 0.0;
        }
    }
    // _anova (psgd.py:34-44) over this chunk, continuing a[]
    __device__ __forceinline__ void dp(double* a, int deg) const {
#pragma unroll
        for (int u = 0; u < kPsgdRC; ++u) {
            const double x = val(u);
#pragma unroll
            for (int t = kMaxDegree; t >= 1; --t)
                if (t <= deg) a[t] += a[t - 1] * x * p[u];
        }
    }
    // _grad_anova + the scatter of _update_grads (psgd.py:25-31, :86-91)
    __device__ __forceinline__ void grad(const double* a, int deg, double dl,
                                         double* __restrict__ Gcol, int k, bool act) const {
#pragma unroll
        for (int u = 0; u < kPsgdRC; ++u) {
            const int j = col(u);
            const double x = val(u);
            double dprev = x;
#pragma unroll
            for (int t = 1; t < kMaxDegree; ++t)
                if (t < deg) dprev = x * (a[t] - p[u] * dprev);
            if (act && u < cnt) unsafeAtomicAdd(&Gcol[(size_t)j * k], dl * dprev);
        }
    }
};

__device__ __forceinline__ void psgd_a_init(double* a) {
    a[0] = 1.0;
#pragma unroll
    for (int t = 1; t <= kMaxDegree; ++t) a[t] = 0.0;
}

// One L-lane group per row of the minibatch.
template <typename T, int L>
__global__ __launch_bounds__(kBlock) void psgd_grad_kernel(
    const int32_t* __restrict__ samples, int B, const int64_t* __restrict__ rptr,
    const int32_t* __restrict__ ridx, const T* __restrict__ rval, const T* __restrict__ yy,
    const double* __restrict__ Pt, const double* __restrict__ w,
    const double* __restrict__ lams, int n_orders, int k, int d, int degree, int loss,
    int fit_linear, double* __restrict__ grad_P, double* __restrict__ grad_w,
    double* __restrict__ loss_row, const PsgdBatch* __restrict__ sched, int* __restrict__ idx) {
    constexpr int gpb = kBlock / L;
    const int grp = threadIdx.x / L, ln = threadIdx.x % L;
    const int r = blockIdx.x * gpb + grp;
    if (sched) {  // table-driven (graph replay): this launch handles batch idx[0]
        const int bi = idx[0];
        if (blockIdx.x == 0 && threadIdx.x == 0) idx[1] = bi;
        B = sched[bi].B;
        samples += sched[bi].pos;
        loss_row += sched[bi].pos;
    }
    if (r >= B) return;  // whole groups leave; no block-level synchronisation below
    const int i = samples[r];
    const int64_t lo = rptr[i], hi = rptr[i + 1];
    const int C = (k + L - 1) / L;
    // the common shape (one order, k <= L, row fits one chunk) keeps its chunk -- entries
    // and P values -- in registers between the prediction and the gradient pass
    const bool keep = (n_orders == 1 && C == 1 && hi - lo <= kPsgdRC);
    PsgdChunk<T, L> ch;
    double a[kMaxDegree + 1];
    const double yi = (double)yy[2 * (size_t)i + 1];
    // _pred (psgd.py:47-57)
    double yp = 0.0;
    for (int64_t e = lo + ln; e < hi; e += L) yp += (double)rval[e] * w[ridx[e]];
    for (int o = 0; o < n_orders; ++o) {
        const int deg = degree - o;
        for (int c = 0; c < C; ++c) {
            const int s = c * L + ln;
            const bool act = s < k;
            const double* Pcol = Pt + (size_t)o * d * k + (act ? s : 0);
            psgd_a_init(a);
            for (int64_t tb = lo; tb < hi; tb += kPsgdRC) {
                ch.load_entries(tb, hi, ln, ridx, rval);
                ch.load_p(Pcol, k, act);
                ch.dp(a, deg);
            }
            double top = a[1];
#pragma unroll
            for (int t = 2; t <= kMaxDegree; ++t)
                if (t == deg) top = a[t];
            if (act) yp += lams[s] * top;
        }
    }
    yp = group_sum(yp, L);
    const double dL = dloss_dev(loss, yp, yi);
    if (ln == 0) loss_row[r] = loss_dev(loss, yp, yi);
    // _update_grads (psgd.py:60-91)
    if (fit_linear)
        for (int64_t e = lo + ln; e < hi; e += L)
            unsafeAtomicAdd(&grad_w[ridx[e]], dL * (double)rval[e]);
    if (keep) {
        if (hi == lo) return;  // empty row: the chunk was never loaded
        const bool act = ln < k;
        ch.grad(a, degree, act ? dL * lams[ln] : 0.0, grad_P + (act ? ln : 0), k, act);
        return;
    }
    for (int o = 0; o < n_orders; ++o) {
        const int deg = degree - o;
        for (int c = 0; c < C; ++c) {
            const int s = c * L + ln;
            const bool act = s < k;
            const double* Pcol = Pt + (size_t)o * d * k + (act ? s : 0);
            double* Gcol = grad_P + (size_t)o * d * k + (act ? s : 0);
            const double dl = act ? dL * lams[s] : 0.0;
            psgd_a_init(a);
            for (int64_t tb = lo; tb < hi; tb += kPsgdRC) {
                ch.load_entries(tb, hi, ln, ridx, rval);
                ch.load_p(Pcol, k, act);
                ch.dp(a, deg);
            }
            for (int64_t tb = lo; tb < hi; tb += kPsgdRC) {
                ch.load_entries(tb, hi, ln, ridx, rval);
                ch.load_p(Pcol, k, act);
                ch.grad(a, deg, dl, Gcol, k, act);
            }
        }
    }
}

// soft_thresholding (regularizer/utils.py:8-9)
__device__ __forceinline__ double soft_thr(double v, double y) {
    const double m = fabs(v) - y;
    const double sg = (double)((v > 0) - (v < 0));
    return sg * (m > 0.0 ? m : 0.0);
}

// Michelot state of one prox (all on the device):
//   part   [V][NB][2]  per-workgroup partial (sum |.|, count) of the entries >= cond[v]
//   cond   [V]         current membership threshold 2 c S / (1 + 2 c theta)  (utils.py:54)
//   thr    [V]         final soft threshold 2 c (S / (1 + 2 c theta))        (utils.py:69-70)
//   theta  [V]         support size of the previous sweep (-1 = none yet)
//   conv   [V]         1 once the support of vector v stopped changing
// cond persists from one minibatch to the next as a warm start: for ANY guess g the set
// {|p| >= g} yields tau <= tau*, after which the iteration is monotone again.
struct MichState {
    double* part;
    double* cond;
    double* thr;
    double* theta;
    int* conv;
    int* done;
    int V;
    int NB;
};

__device__ __forceinline__ bool mich_all_converged(const MichState& ms) {
    int all = 1;
    for (int v = threadIdx.x; v < ms.V; v += kBlock)
        if (!ms.conv[v]) all = 0;
    return __syncthreads_and(all) != 0;
}

// _update_params (psgd.py:94-122): SGD step + shrink + grad reset on every row of
// P (n_orders, d, k), fused with the prox for l1 / l21.  For squaredl12 / squaredl21 the
// pass also emits the first Michelot partials with v = o*k + s (squaredl12) or v = o
// (squaredl21, over the row norms it stores), measured against the warm-start cond.
template <int L>
__global__ __launch_bounds__(kBlock) void psgd_update_kernel(
    double* __restrict__ Pt, double* __restrict__ grad_P, double* __restrict__ w,
    double* __restrict__ grad_w, int n_orders, int k, int d, int reg, double cp, double denp,
    double strength, int fit_linear, double cw, double denw, double* __restrict__ norms,
    MichState ms, const PsgdBatch* __restrict__ sched, int* __restrict__ idx) {
    constexpr int gpb = kBlock / L;
    __shared__ double red[2][kBlock];
    if (sched) {  // table-driven (graph replay): batch idx[1], then hand idx[0] on
        const int bi = idx[1];
        cp = sched[bi].cp;
        denp = sched[bi].denp;
        strength = sched[bi].strength;
        cw = sched[bi].cw;
        denw = sched[bi].denw;
        if (blockIdx.x == 0 && threadIdx.x == 0) idx[0] = bi + 1;
    }
    const int grp = threadIdx.x / L, ln = threadIdx.x % L;
    const int C = (k + L - 1) / L;
    if (fit_linear) {
        for (int j = blockIdx.x * kBlock + threadIdx.x; j < d; j += gridDim.x * kBlock) {
            const double g = grad_w[j] * cw;
            w[j] = (w[j] - g) / denw;
            grad_w[j] = 0.0;
        }
    }
    const bool mich = (reg == REG_SQL12 || reg == REG_SQL21);
    if (mich && blockIdx.x == 0) {
        for (int v = threadIdx.x; v < ms.V; v += kBlock) {
            ms.theta[v] = -1.0;
            ms.conv[v] = 0;
        }
        if (threadIdx.x == 0) *ms.done = 0;
    }
    for (int o = 0; o < n_orders; ++o) {
        double acc[kPsgdMaxC] = {0, 0, 0, 0}, cnt[kPsgdMaxC] = {0, 0, 0, 0};
        double cv[kPsgdMaxC] = {0, 0, 0, 0};
        if (reg == REG_SQL12) {
#pragma unroll
            for (int c = 0; c < kPsgdMaxC; ++c)
                if (c < C && c * L + ln < k) cv[c] = ms.cond[(size_t)o * k + c * L + ln];
        } else if (reg == REG_SQL21) {
            cv[0] = ms.cond[o];
        }
        for (int j = blockIdx.x * gpb + grp; j < d; j += gridDim.x * gpb) {
            const size_t base = ((size_t)o * d + j) * k;
            double p[kPsgdMaxC];
            double q = 0.0;
#pragma unroll
            for (int c = 0; c < kPsgdMaxC; ++c) {
                const int s = c * L + ln;
                p[c] = 0.0;
                if (c < C && s < k) {
                    const double g = grad_P[base + s] * cp;
                    grad_P[base + s] = 0.0;
                    p[c] = (Pt[base + s] - g) / denp;
                    q += fabs(p[c]) * fabs(p[c]);
                }
            }
            if (reg == REG_L1) {
#pragma unroll
                for (int c = 0; c < kPsgdMaxC; ++c) p[c] = soft_thr(p[c], strength);
            } else if (reg == REG_L21) {  // l21.py:43-48
                double nr = sqrt(group_sum(q, L));
                if (nr <= strength) nr = INFINITY;
                const double f = 1.0 - strength / nr;
#pragma unroll
                for (int c = 0; c < kPsgdMaxC; ++c) p[c] *= f;
            } else if (reg == REG_SQL12) {
#pragma unroll
                for (int c = 0; c < kPsgdMaxC; ++c) {
                    const double a = fabs(p[c]);
                    if (a >= cv[c]) {
                        acc[c] += a;
                        cnt[c] += 1.0;
                    }
                }
            } else {  // REG_SQL21: squaredl21.py:66
                const double nr = sqrt(group_sum(q, L));
                if (ln == 0) norms[(size_t)o * d + j] = nr;
                if (nr >= cv[0]) {
                    acc[0] += nr;
                    cnt[0] += 1.0;
                }
            }
#pragma unroll
            for (int c = 0; c < kPsgdMaxC; ++c) {
                const int s = c * L + ln;
                if (c < C && s < k) Pt[base + s] = p[c];
            }
        }
        if (reg == REG_SQL12) {
            for (int c = 0; c < C; ++c) {
                __syncthreads();
                red[0][threadIdx.x] = acc[c];
                red[1][threadIdx.x] = cnt[c];
                __syncthreads();
                const int s = c * L + ln;
                if (grp == 0 && s < k) {
                    double sa = 0, sc = 0;
                    for (int g = 0; g < gpb; ++g) {
                        sa += red[0][g * L + ln];
                        sc += red[1][g * L + ln];
                    }
                    double* dst = ms.part + (((size_t)o * k + s) * ms.NB + blockIdx.x) * 2;
                    dst[0] = sa;
                    dst[1] = sc;
                }
            }
        } else if (reg == REG_SQL21) {
            __syncthreads();
            red[0][threadIdx.x] = acc[0];
            red[1][threadIdx.x] = cnt[0];
            __syncthreads();
            if (threadIdx.x == 0) {
                double sa = 0, sc = 0;
                for (int g = 0; g < gpb; ++g) {
                    sa += red[0][g * L];
                    sc += red[1][g * L];
                }
                double* dst = ms.part + ((size_t)o * ms.NB + blockIdx.x) * 2;
                dst[0] = sa;
                dst[1] = sc;
            }
        }
    }
}

// One wave per vector: fold its NB partials in fixed order, derive the next thresholds,
// mark the vector converged when its support size did not change.
static __global__ __launch_bounds__(kBlock) void psgd_mich_finish_kernel(
    MichState ms, double strength, const PsgdBatch* __restrict__ sched, const int* __restrict__ idx) {
    if (sched) strength = sched[idx[1]].strength;  // table-driven (graph replay)
    const int v = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= ms.V || ms.conv[v]) return;
    double S = 0, th = 0;
    for (int b = lane; b < ms.NB; b += kWave) {
        S += ms.part[((size_t)v * ms.NB + b) * 2];
        th += ms.part[((size_t)v * ms.NB + b) * 2 + 1];
    }
    S = wave_sum(S);
    th = wave_sum(th);
    if (lane == 0) {
        const double den = 1.0 + 2.0 * strength * th;
        ms.cond[v] = 2 * strength * S / den;
        ms.thr[v] = 2 * strength * (S / den);
        if (th == ms.theta[v]) ms.conv[v] = 1;
        ms.theta[v] = th;
    }
}

// One Michelot sweep: partial (sum, count) of the entries with |.| >= cond[v].
template <int L>
__global__ __launch_bounds__(kBlock) void psgd_mich_reduce_kernel(
    const double* __restrict__ Pt, const double* __restrict__ norms, int n_orders, int k, int d,
    int reg, MichState ms) {
    if (mich_all_converged(ms)) return;
    constexpr int gpb = kBlock / L;
    __shared__ double red[2][kBlock];
    const int grp = threadIdx.x / L, ln = threadIdx.x % L;
    const int C = (k + L - 1) / L;
    for (int o = 0; o < n_orders; ++o) {
        if (reg == REG_SQL12) {
            for (int c = 0; c < C; ++c) {
                const int s = c * L + ln;
                const bool act = s < k && !ms.conv[(size_t)o * k + s];
                const double cv = act ? ms.cond[(size_t)o * k + s] : 0.0;
                double sa = 0, sc = 0;
                if (act)
                    for (int j = blockIdx.x * gpb + grp; j < d; j += gridDim.x * gpb) {
                        const double a = fabs(Pt[((size_t)o * d + j) * k + s]);
                        if (a >= cv) {
                            sa += a;
                            sc += 1.0;
                        }
                    }
                __syncthreads();
                red[0][threadIdx.x] = sa;
                red[1][threadIdx.x] = sc;
                __syncthreads();
                if (grp == 0 && act) {
                    double ta = 0, tc = 0;
                    for (int g = 0; g < gpb; ++g) {
                        ta += red[0][g * L + ln];
                        tc += red[1][g * L + ln];
                    }
                    double* dst = ms.part + (((size_t)o * k + s) * ms.NB + blockIdx.x) * 2;
                    dst[0] = ta;
                    dst[1] = tc;
                }
            }
        } else {
            if (ms.conv[o]) continue;
            const double cv = ms.cond[o];
            double sa = 0, sc = 0;
            for (int j = blockIdx.x * kBlock + threadIdx.x; j < d; j += gridDim.x * kBlock) {
                const double a = norms[(size_t)o * d + j];
                if (a >= cv) {
                    sa += a;
                    sc += 1.0;
                }
            }
            __shared__ double r2[16];
            block_sum2(sa, sc, r2);
            if (threadIdx.x == 0) {
                double* dst = ms.part + ((size_t)o * ms.NB + blockIdx.x) * 2;
                dst[0] = sa;
                dst[1] = sc;
            }
        }
    }
}

// done <- every vector converged (read by the host between chunks of sweeps)
static __global__ __launch_bounds__(kBlock) void psgd_mich_check_kernel(MichState ms) {
    const bool all = mich_all_converged(ms);
    if (threadIdx.x == 0) *ms.done = all ? 1 : 0;
}

// graph replay: a fixed number of sweeps was recorded; if that was not enough for some
// minibatch, say so (sticky) -- the host then redoes the epoch from its snapshot, eagerly
static __global__ __launch_bounds__(kBlock) void psgd_mich_verify_kernel(MichState ms, int* failed) {
    const bool all = mich_all_converged(ms);
    if (threadIdx.x == 0 && !all) *failed = 1;
}

// Final soft threshold (squaredl12.py:73-75 / squaredl21.py:67-74).
template <int L>
__global__ __launch_bounds__(kBlock) void psgd_mich_apply_kernel(
    double* __restrict__ Pt, const double* __restrict__ norms, int n_orders, int k, int d,
    int reg, const double* __restrict__ thr) {
    constexpr int gpb = kBlock / L;
    const int grp = threadIdx.x / L, ln = threadIdx.x % L;
    const int C = (k + L - 1) / L;
    for (int o = 0; o < n_orders; ++o)
        for (int j = blockIdx.x * gpb + grp; j < d; j += gridDim.x * gpb) {
            const size_t base = ((size_t)o * d + j) * k;
            double nr = 0, nn = 0;
            if (reg == REG_SQL21) {
                nr = norms[(size_t)o * d + j];
                nn = soft_thr(nr, thr[o]);
            }
            for (int c = 0; c < C; ++c) {
                const int s = c * L + ln;
                if (s >= k) continue;
                double p = Pt[base + s];
                if (reg == REG_SQL12) {
                    p = soft_thr(p, thr[(size_t)o * k + s]);
                } else {
                    if (nr > 0) p /= nr;
                    p *= nn;
                }
                Pt[base + s] = p;
            }
        }
}

}  // namespace spfm
