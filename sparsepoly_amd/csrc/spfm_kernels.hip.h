// spfm_kernels.hip.h -- gfx950 device code of the sparse-FM proximal CD core.
//
// Execution model: the coordinate order is partitioned into
// batches of columns that share no row; one batch = one dependent step.  Two engines:
//   persistent row-block pass (pcd_prb_kernel / lin_prb_kernel): one launch per
//     component pass, each workgroup owns a block of rows, the per-step column partial
//     sums are exchanged as tagged 8-byte granules (agent-scope stores / loads);
//   multi-kernel (multi-GPU, pbcd): per step a gather kernel (workgroup per column, f64
//     wave-shuffle + LDS reduction), [RCCL all-reduce], and a fused chain + scatter
//     kernel; kernel boundaries on one stream are the only inter-workgroup
//     synchronisation and the launch sequence is replayed from a hipGraph.
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

// Model kind by template parameter M: M >= 2 = factorization machine of degree M (ANOVA
// kernel, caches A[i, 1..M-1]); M == 0 = all-subsets model (kernel prod_j (1 + p_j x_j),
// one cache value A[i] per component; reference optimizer/pcd_all.py, pbcd_all.py,
// regularizers called with degree = -1).
template <int M>
struct Kind {
    static constexpr int AS = (M == 0) ? 1 : (M - 1);  // cache values per (row, component)
};

// dA_{M-1} of pcd._grad_anova (pcd.py:8-12) or the all-subsets derivative
// x A / (1 + x p) (pcd_all.py:28) from the cache values a[0..AS)
template <int M>
__device__ __forceinline__ double grad_factor(const double* a, double x, double p) {
    if constexpr (M == 0) {
        return x * a[0] / (1.0 + x * p);
    } else {
        double dprev = x;
#pragma unroll
        for (int t = 1; t < M; ++t) dprev = x * (a[t - 1] - p * dprev);
        return dprev;
    }
}

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

// pcd._precompute_A_all_degree (optimizer/pcd.py:15-30): per row the reference's column
// sweep visits the row's entries in ascending column order; the kernel below keeps that
// order inside every row.  A[i, M] is never read during training (pcd.py:11-12) and is
// not stored.
// All components in one pass over the CSR image (the "one precompute pass for all s"
// of the roofline model): A_all[s][i][t-1] = A^{(s)}[i, t].  Valid because
// P[s,:] changes only during pass s, so A^{(s)} computed from the epoch-start P equals
// what the reference recomputes at the start of pass s (pcd.py:94).  One wavefront per
// row at a time, lanes over components (P^T rows are coalesced 8k-byte reads), 8 P^T
// loads in flight; results are staged through LDS so that the per-component slabs
// are written in contiguous runs.  Pt is (d, k).
template <typename T, int M>
__global__ __launch_bounds__(kBlock) void pcd_precompute_all_kernel(
    int64_t n, int k, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
    const T* __restrict__ rval, const double* __restrict__ Pt, T* __restrict__ A_all) {
    constexpr int R = 32;       // rows per tile
    constexpr int RP = R + 1;   // padded row stride in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char pre_lds[];
    T* tile = reinterpret_cast<T*>(pre_lds);  // [AS][64][RP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t tile0 = (int64_t)blockIdx.x * R; tile0 < n; tile0 += (int64_t)gridDim.x * R) {
        for (int s0 = 0; s0 < k; s0 += kWave) {
            const int s = s0 + lane;
            const bool sv = s < k;
            for (int r = wave; r < R; r += kBlock / kWave) {
                const int64_t i = tile0 + r;
                if (i >= n) break;
                constexpr int NA = (M == 0) ? 2 : M;
                double a[NA];
                a[0] = 1.0;
#pragma unroll
                for (int t = 1; t < NA; ++t) a[t] = (M == 0) ? 1.0 : 0.0;  // M==0: a[1] = product
                const int64_t b = rptr[i], e = rptr[i + 1];
                for (int64_t c = b; c < e; c += kWave) {
                    const int cnt = (int)((e - c < kWave) ? (e - c) : kWave);
                    const int my_col = (lane < cnt) ? ridx[c + lane] : 0;
                    const float my_xf = (lane < cnt) ? (float)rval[c + lane] : 0.f;
                    const double my_xd = (lane < cnt) ? (double)rval[c + lane] : 0.0;
                    for (int q = 0; q < cnt; q += 8) {
                        double pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int src = (q + u < cnt) ? (q + u) : q;
                            const int col = __builtin_amdgcn_readlane(my_col, src);
                            pv[u] = sv ? Pt[(size_t)col * k + s] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (q + u < cnt) {
                                double x;
                                if (sizeof(T) == 4)
                                    x = (double)__int_as_float(__builtin_amdgcn_readlane(
                                        __float_as_int(my_xf), q + u));
                                else
                                    x = readlane_d(my_xd, q + u);
                                if constexpr (M == 0) {
                                    a[1] *= 1.0 + pv[u] * x;  // pcd_all.py:18
                                } else {
#pragma unroll
                                    for (int t = M - 1; t >= 1; --t) a[t] += a[t - 1] * pv[u] * x;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 1; t <= Kind<M>::AS; ++t)
                    tile[((t - 1) * kWave + lane) * RP + r] = (T)a[t];
            }
            __syncthreads();
            constexpr int AS = Kind<M>::AS;
            const int per_s = R * AS;
            for (int idx = tid; idx < kWave * per_s; idx += kBlock) {
                const int sl = idx / per_s, rem = idx - sl * per_s;
                const int r = rem / AS, t1 = rem - r * AS;
                const int64_t i = tile0 + r;
                if (s0 + sl < k && i < n)
                    A_all[((size_t)(s0 + sl) * n + i) * AS + t1] = tile[(t1 * kWave + sl) * RP + r];
            }
            __syncthreads();
        }
    }
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
    if constexpr (M == 0) {  // omegati.py:75-80: _cache_all_subsets = prod_j (1 + |p_sj|)
        double pr = 1.0;
        for (int j = tid; j < d; j += kBlock) pr *= 1.0 + fabs(ps[j]);
        sh[tid] = pr;
        __syncthreads();
        for (int half = kBlock / 2; half >= 1; half >>= 1) {
            if (tid < half) sh[tid] *= sh[tid + half];
            __syncthreads();
        }
        if (tid == 0) cache[0] = sh[0];
        return;
    }
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
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, const T* __restrict__ A_all,
    size_t a_stride, const typename Vec2<T>::type* __restrict__ yy, const double* __restrict__ P,
    int d, int loss, double* __restrict__ part, double* __restrict__ pold) {
    __shared__ double red[16];
    const int q = blockIdx.x;
    const ColDesc cd = desc[q];
    const T* __restrict__ A = A_all + (size_t)ctl->s * a_stride;
    const double p = P[(size_t)ctl->s * d + cd.j];
    const int64_t b = cd.start, e = cd.start + cd.len;
    constexpr int AS = Kind<M>::AS;
    double g = 0.0, h = 0.0;
    for (int64_t ii = b + threadIdx.x; ii < e; ii += kBlock) {
        const int i = cidx[ii];
        const double x = (double)cval[ii];
        const typename Vec2<T>::type yv = yy[i];
        double a[AS];
#pragma unroll
        for (int t = 0; t < AS; ++t) a[t] = (double)A[(size_t)i * AS + t];
        const double dprev = grad_factor<M>(a, x, p);
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

// ---- speculative affine scan for the degree-2 cache recurrences --------------
// For degree 2 the regularizer cache is one scalar c and column i maps it through a
// piecewise-affine f_i (squaredl12.py:47-57: c' = (c - a) + max(p - t (c - a), 0);
// omegati.py:76-99 at degree 2: u = max(c - a, 0), c' = u + max(p - s u, 0)).  Given
// the branch each column takes, f_i is affine, and the values seen by all 64 columns
// follow from ONE wave-parallel prefix composition of affine maps (6 shuffle steps)
// instead of a 64-long dependent loop.  The branches are guessed (from the previous
// round's values, initially from c at the start of the batch), the scan is evaluated,
// and every lane re-checks its own branch with the value it actually receives; all
// lanes before the first mismatch are then provably right, so each round fixes at
// least one more column and the fixed point is exactly the sequential result (up to
// the rounding of composed vs. step-by-step affine evaluation, ~1e-16 relative).
__device__ __forceinline__ void affine_scan_inclusive(double& al, double& be, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double oa = __shfl_up(al, o, kWave);
        const double ob = __shfl_up(be, o, kWave);
        if (lane >= o) {  // mine after other: x -> al*(oa*x + ob) + be
            be = al * ob + be;
            al = al * oa;
        }
    }
}

// value of the cache in front of column `lane` given c0 and the inclusive scan
__device__ __forceinline__ double affine_before(double al_inc, double be_inc, double c0, int lane) {
    const double pa = __shfl_up(al_inc, 1, kWave);
    const double pb = __shfl_up(be_inc, 1, kWave);
    return (lane == 0) ? c0 : (pa * c0 + pb);
}

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
        const double c0 = cache[0];
        const bool act = valid && lane <= last;
        // branch guess: evaluate every column at c0
        bool nz = (app - tt * (c0 - ab)) > 0;
        double cb = c0, m = 0.0, al = 1.0, be = 0.0;
        for (int round = 0; round <= kWave; ++round) {
            al = act ? (nz ? (1.0 - tt) : 1.0) : 1.0;
            be = act ? (nz ? (app - (1.0 - tt) * ab) : -ab) : 0.0;
            affine_scan_inclusive(al, be, lane);
            cb = affine_before(al, be, c0, lane);
            m = fma(-tt, cb - ab, app);
            const bool nz2 = m > 0;
            const unsigned long long bad = __ballot(act && (nz2 != nz));
            nz = nz2;
            if (bad == 0ull) break;
        }
        const double r = (act && nz) ? m : 0.0;
        cache[0] = readlane_d(al, last) * c0 + readlane_d(be, last);
        return sg * r;
    }
    // REG_OMEGATI
    {
        const double apin = fabs(pin);
        const double sg = (pin > 0) ? 1.0 : -1.0;
        if constexpr (M == 0) {
            // all-subsets (omegati.py:100-102, 87-88): c /= 1 + |p_old|; strength *= c;
            // soft-threshold; c *= 1 + |p_new| -- multiplicative, so a plain serial loop
            double c = cache[0];
            for (int i = 0; i <= last; ++i) {
                const double ai = readlane_d(ab, i), si = readlane_d(st, i),
                             pi = readlane_d(apin, i);
                c /= 1.0 + ai;
                const double m = pi - si * c;
                const double r = (m > 0) ? m : 0.0;
                c *= 1.0 + r;
                if (lane == i) mine = r;
            }
            cache[0] = c;
            return sg * mine;
        }
        if constexpr (M == 2) {
            // degree 2: u = max(c - a, 0); r = max(p - s u, 0); c' = u + r  (dcache[1] = 1)
            const double c0 = cache[1];
            const bool act = valid && lane <= last;
            bool pos = (c0 - ab) >= 0;                       // clip of omegati.py:97-98 inactive
            bool nz = (apin - st * (pos ? (c0 - ab) : 0.0)) > 0;
            double cb = c0, u = 0.0, m = 0.0, al = 1.0, be = 0.0;
            for (int round = 0; round <= kWave; ++round) {
                if (!act) {
                    al = 1.0;
                    be = 0.0;
                } else if (!pos) {   // u = 0, r = p
                    al = 0.0;
                    be = apin;
                } else if (nz) {     // c' = (1 - s)(c - a) + p
                    al = 1.0 - st;
                    be = apin - al * ab;
                } else {             // c' = c - a
                    al = 1.0;
                    be = -ab;
                }
                affine_scan_inclusive(al, be, lane);
                cb = affine_before(al, be, c0, lane);
                const double v = cb - ab;
                const bool pos2 = !(v < 0);
                u = pos2 ? v : 0.0;
                m = apin - st * u;
                const bool nz2 = m > 0;
                const unsigned long long bad =
                    __ballot(act && ((pos2 != pos) || (pos2 && (nz2 != nz))));
                pos = pos2;
                nz = nz2;
                if (bad == 0ull) break;
            }
            const double r = act ? ((m > 0) ? m : 0.0) : 0.0;
            cache[1] = readlane_d(al, last) * c0 + readlane_d(be, last);
            return sg * r;
        }
        if constexpr (M > 2)
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
    if constexpr (M == 0) {  // pcd_all.py:92-98
        const double a0 = (double)A[i];
        double yh = (double)yy[2 * i];
        yh -= lam * a0;
        double a1 = a0 / (1.0 + x * p_old);
        a1 *= 1.0 + x * (p_old - upd);
        yh += lam * a1;
        A[i] = (T)a1;
        yy[2 * i] = (T)yh;
        return;
    }
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
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ A_all,
    size_t a_stride, T* __restrict__ yy /* (yhat,y) pairs */, const double* __restrict__ delta,
    const double* __restrict__ pold) {
    const int q = blockIdx.x;
    const double upd = delta[q];
    if (upd == 0.0) return;
    T* __restrict__ A = A_all + (size_t)ctl->s * a_stride;
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
    const int32_t* __restrict__ cidx, const T* __restrict__ cval, T* __restrict__ A_all,
    size_t a_stride, T* __restrict__ yy, double* __restrict__ viol_col) {
    __shared__ double sh[2];
    T* __restrict__ A = A_all + (size_t)ctl->s * a_stride;
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

// ------------------------------------------- persistent row-block pass (PRB)
//
// One launch sweeps ALL batches of a component pass.  G workgroups (one per CU),
// workgroup g owns the contiguous row block R_g; A and (yhat,y) rows of R_g are
// read and written by that workgroup only, so they need no inter-workgroup
// coherence and stay warm in its XCD's L2.  The entries of every batch are
// pre-sorted on the host per (workgroup, batch, slot) (`erow/eval/sp`), so all
// entry loads are plain streaming loads at known addresses.
//
// Per dependent step the only exchange is the all-gather of the per-slot partial
// sums: workgroup g stores its (sum dloss*dA, sum dA^2) pairs write-through
// (agent-scope relaxed atomic stores = global_store sc1) into slab[parity][g][slot],
// drains them (s_waitcnt vmcnt(0)), and one lane adds 1 to the step's arrival
// counter; one lane polls that counter with sc1 loads until G arrivals, after
// which the same wave reads all G slabs with sc1 loads and sums them in fixed
// order g = 0..G-1 (bitwise identical in every workgroup).  This is the hand-off
// form "one signalling lane per storing workgroup, counter add / sc1 poll, all
// stores and loads sc1, one workgroup per CU" of MI355X_MICROARCH.md (Valid forms,
// first table row).  Every workgroup then runs the scalar chain redundantly and
// scatter-updates its own rows.  Slabs are double-buffered by step parity: a
// workgroup can only be two publishes ahead of a reader if it passed the
// intermediate all-gather, which needs that reader's arrival.
// Every spin is bounded; on time-out the abort word is set and all workgroups
// leave the loop (the host reports the failure).

struct PrbArgs {
    int G;                 // workgroups
    int nb;                // batches in the sweep
    const int32_t* bptr;   // [nb+1] batch boundaries into desc
    const ColDesc* desc;   // columns in visiting order
    const int32_t* sp;     // [G][nb][65] slot boundaries into erow/eval
    const uint32_t* lmask; // [G][nb][2] bit q: slot q is "long" in this row block
    int has_long;          // 0: no long slot anywhere in the schedule (masks not even read)
    const int32_t* erow;   // entry row ids, sorted by (workgroup, batch, slot, row)
    double* slab;          // [2][G][64][2]
    unsigned* abort_flag;  // [1]
    long long* stamps;     // diagnostic: [G][16] accumulated cycles per phase (8 control-wave,
                           // 8 worker-wave values), or nullptr
};

__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p),
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    const unsigned long long u =
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}

// ---- tagged-granule exchange ----------------------------------------------------
// A partial sum travels as ONE naturally aligned 8-byte word: the double with its two
// lowest mantissa bits replaced by a step tag (relative perturbation <= 2^-51, applied
// before the value is used anywhere, so every workgroup sums identical numbers).  The
// data is the flag (MI355X_MICROARCH.md "R2's granule"): one sc1 store publishes, sc1
// loads poll the word itself; no drain, no counter, no fence.  Slabs are double-buffered
// by step parity and zeroed before every launch; tag(b) = ((b >> 1) % 3) + 1 is never 0
// and differs from the tag of the slab's previous occupant (step b - 2).
__device__ __forceinline__ unsigned long long prb_tag(int b) {
    return (unsigned long long)(((b >> 1) % 3) + 1);
}
__device__ __forceinline__ void prb_store_granule(double* p, double v, unsigned long long tag) {
    const unsigned long long u =
        ((unsigned long long)__double_as_longlong(v) & ~3ull) | tag;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long prb_load_granule(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}

// Worker wave `w` (0..3) sums the granules of workgroups [w*G/4, (w+1)*G/4) for slot
// `lane` (all loads in flight together, re-swept until every tag matches); the control
// wave later adds the four quarter sums in order w = 0..3, so the total is the same bit
// pattern in every workgroup.  Returns false after a bounded number of sweeps.
template <int NV>
__device__ __forceinline__ bool prb_collect_quarter(const PrbArgs& a, int b, int w, int lane,
                                                    int ncols, double* out /* [4][64][2] LDS */) {
    const double* slab = a.slab + (size_t)(b & 1) * a.G * 64 * 2;
    const unsigned long long tag = prb_tag(b);
    const int g0 = (a.G * w) / 4, g1 = (a.G * (w + 1)) / 4;
    double tot[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) tot[v] = 0.0;
    bool ok = true;
    if (lane < ncols) {
        const double* sl = slab + (size_t)lane * 2;
        constexpr int GU = 8;  // granule pairs polled together per lane
        for (int gg = g0; gg < g1; gg += GU) {
            unsigned long long t[GU][NV];
            unsigned spins = 0;
            for (;;) {
                bool all = true;
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const bool in = gg + u < g1;
                        t[u][v] = in ? prb_load_granule(sl + (size_t)(gg + u) * 128 + v) : tag;
                        all = all && ((t[u][v] & 3ull) == tag);
                    }
                if (all) break;
                if ((++spins & 63u) == 0) {
                    if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) ||
                        spins > (1u << 21)) {
                        __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                }
            }
            if (!ok) break;
#pragma unroll
            for (int u = 0; u < GU; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (gg + u < g1) tot[v] += __longlong_as_double((long long)(t[u][v] & ~3ull));
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) out[((size_t)w * 64 + lane) * 2 + v] = tot[v];
    return ok;
}

// The entries one thread owns in one step: 4 lanes share a slot, each keeps up to
// PRB_PF entries (row, value) in registers; a slot with more than 4*PRB_PF entries in
// this row block falls back to a reload loop for the rest.
constexpr int PRB_PF = 4;
// A (workgroup, step, slot) segment longer than this is a "long slot" (a very frequent
// feature): the 4 lanes of the slot skip it and all 256 worker threads stride over it.
constexpr int kPrbLong = 48;
template <typename T>
struct PrbEntries {
    int e0, e1;
    int row[PRB_PF];
    T x[PRB_PF];
};

__device__ __forceinline__ void prb_load_sp(const PrbArgs& a, int g, int b, int slot, int ncols,
                                            int& e0, int& e1, unsigned long long& lmask) {
    e0 = 0;
    e1 = 0;
    lmask = 0ull;
    if (a.has_long) {
        const uint32_t* lm = a.lmask + ((size_t)g * a.nb + b) * 2;
        lmask = ((unsigned long long)lm[1] << 32) | (unsigned long long)lm[0];
    }
    if (slot < ncols && !((lmask >> slot) & 1ull)) {  // long slots: no per-lane entries
        const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
        e0 = spb[slot];
        e1 = spb[slot + 1];
    }
}

// q-th set bit of m (q < popcount(m))
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int q) {
    for (int t = 0; t < q; ++t) m &= m - 1;
    return __builtin_ctzll(m);
}

template <typename T>
__device__ __forceinline__ void prb_load_entries(const PrbArgs& a, const T* __restrict__ eval,
                                                 int e0, int e1, int sub, PrbEntries<T>& en) {
    en.e0 = e0;
    en.e1 = e1;
#pragma unroll
    for (int u = 0; u < PRB_PF; ++u) {
        const int e = e0 + sub + 4 * u;
        const bool v = e < e1;
        en.row[u] = v ? a.erow[e] : 0;
        en.x[u] = v ? eval[e] : (T)0;
    }
}

// Workgroup = 5 wavefronts: wave 0 is the CONTROL wave (publish, poll, chain), waves
// 1..4 are WORKERS (256 threads = 64 slots x 4 lanes) that own the entries.  Software
// pipeline of step b: its entries are already in worker registers (loaded during step
// b-1 from slot bounds loaded during step b-2), so phase 1 starts with the row gathers;
// the workers issue the next step's streaming loads while the control wave waits for
// the other workgroups.
constexpr int kPrbThreads = 320;

template <typename T, int M, int LOSS>
__global__ __launch_bounds__(kPrbThreads) void pcd_prb_kernel(
    const Ctl* __restrict__ ctl, PrbArgs a, const T* __restrict__ eval, T* __restrict__ A_all,
    size_t a_stride, T* __restrict__ yy, const double* __restrict__ pold_sched,
    double* __restrict__ P, int d, int reg, const double* __restrict__ cache_in, double mu,
    double beta, double gamma, double eta, double* __restrict__ viol_pos) {
    T* __restrict__ A = A_all + (size_t)ctl->s * a_stride;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];  // sized to pin 1 WG / CU
    double* sh_delta = dyn_lds + 128;  // [64]
    double* sh_pold = dyn_lds + 192;   // [64]
    double* sh_quart = dyn_lds + 256;  // [4][64][2] quarter sums over workgroups
    int* sh_ok = reinterpret_cast<int*>(dyn_lds + 768);
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool control = wave == 0;
    const int wt = tid - 64;  // worker thread id (negative on the control wave)
    const int slot = control ? 64 : (wt >> 2), sub = wt & 3;
    const int s = ctl->s;
    const double lam = ctl->lam;
    double* ps = P + (size_t)s * d;
    double cache[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) cache[t] = cache_in[t];

    double* sh_long = dyn_lds + 1024;  // [64][4][2] wave partials of long slots
    PrbEntries<T> cur, nxt;
    int c0 = a.bptr[0], c1 = a.bptr[1];
    int c2 = (a.nb > 1) ? a.bptr[2] : c1;
    int c3 = (a.nb > 2) ? a.bptr[3] : c2;
    unsigned long long lm0 = 0ull, lm1 = 0ull;  // long-slot masks of steps b, b+1
    {
        int e0, e1;
        prb_load_sp(a, g, 0, slot, c1 - c0, e0, e1, lm0);
        prb_load_entries<T>(a, eval, e0, e1, sub, cur);
    }
    int ne0 = 0, ne1 = 0;  // slot bounds of step b+1
    if (a.nb > 1) prb_load_sp(a, g, 1, slot, c2 - c1, ne0, ne1, lm1);
    double p_slot = (slot < c1 - c0) ? pold_sched[c0 + slot] : 0.0;
    if (tid == 0) *sh_ok = 1;
    // diagnostic stamps (only when a.stamps != nullptr): cycles per phase, thread 0
    const bool stamp = a.stamps != nullptr;
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = stamp ? clock64() : 0;
#define PRB_STAMP(k)                        \
    if (stamp && tid == 0) {                \
        const long long tn = clock64();     \
        acc[k] += tn - tprev;               \
        tprev = tn;                         \
    }
#define PRB_WSTAMP(k)                       \
    if (stamp && tid == 64) {               \
        const long long tn = clock64();     \
        acc[k] += tn - tprev;               \
        tprev = tn;                         \
    }

    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = (b + 4 <= a.nb) ? a.bptr[b + 4] : c3;  // used two steps from now
        // ---- phase 1 (workers): gather the rows of the prefetched entries, partial sums
        // (pcd.py:52-59); A / yhat values stay in registers for phase 3
        constexpr int AS = Kind<M>::AS;
        double av[PRB_PF][AS];
        double yh[PRB_PF], yt[PRB_PF], dlast[PRB_PF];
        double pl = 0.0;
        int jl = 0;
        if (control) {
            if (lane < ncols) {
                pl = pold_sched[c0 + lane];
                if (g == 0) jl = a.desc[c0 + lane].j;
            }
        } else {
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {  // all gathers in flight before any use
                const size_t i = (size_t)cur.row[u];
                const typename Vec2<T>::type yv = yy2[i];
                yh[u] = (double)yv.x;
                yt[u] = (double)yv.y;
#pragma unroll
                for (int t = 0; t < AS; ++t) av[u][t] = (double)A[i * AS + t];
            }
            double ag = 0.0, ah = 0.0;
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {
                const double dprev = grad_factor<M>(av[u], (double)cur.x[u], p_slot);
                dlast[u] = dprev;
                const double dl = dloss_dev(LOSS, yh[u], yt[u]);
                const bool v = cur.e0 + sub + 4 * u < cur.e1;
                ag += v ? dl * dprev : 0.0;
                ah += v ? dprev * dprev : 0.0;
            }
            for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {  // rare: long slot
                const int i = a.erow[e];
                const double x = (double)eval[e];
                const typename Vec2<T>::type yv = yy2[i];
                double a1[AS];
#pragma unroll
                for (int t = 0; t < AS; ++t) a1[t] = (double)A[(size_t)i * AS + t];
                const double dprev = grad_factor<M>(a1, x, p_slot);
                ag += dloss_dev(LOSS, (double)yv.x, (double)yv.y) * dprev;
                ah += dprev * dprev;
            }
            ag += __shfl_xor(ag, 1, kWave);
            ah += __shfl_xor(ah, 1, kWave);
            ag += __shfl_xor(ag, 2, kWave);
            ah += __shfl_xor(ah, 2, kWave);
            PRB_WSTAMP(0)  // gather + partial sums
            // publish this row block's partial sums of the slot (tagged granules).  Slots
            // beyond the batch are published too (as zeros): every word of a slab is then
            // rewritten at every use of the buffer, so a reader can never meet a stale
            // word that happens to carry the current tag.
            if (sub == 0 && !((lm0 >> slot) & 1ull)) {
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + slot) * 2;
                const unsigned long long tag = prb_tag(b);
                prb_store_granule(sl, ag, tag);
                prb_store_granule(sl + 1, ah, tag);
            }
        }
        // ---- long slots of this row block (rare: very frequent features): the whole
        // workgroup strides over the slot's entries; one extra barrier, taken by all waves
        const unsigned long long lmu =
            ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(lm0 >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)lm0);
        if (lmu != 0ull) {
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            int qi = 0;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1, ++qi) {
                const int q = __builtin_ctzll(mm);
                if (!control) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    const double pq = pold_sched[c0 + q];
                    double lg = 0.0, lh = 0.0;
                    for (int e = le0 + wt; e < le1; e += 256) {
                        const int i = a.erow[e];
                        const double x = (double)eval[e];
                        const typename Vec2<T>::type yv = yy2[i];
                        double a1[Kind<M>::AS];
#pragma unroll
                        for (int t = 0; t < Kind<M>::AS; ++t)
                            a1[t] = (double)A[(size_t)i * Kind<M>::AS + t];
                        const double dprev = grad_factor<M>(a1, x, pq);
                        lg += dloss_dev(LOSS, (double)yv.x, (double)yv.y) * dprev;
                        lh += dprev * dprev;
                    }
                    lg = wave_sum(lg);
                    lh = wave_sum(lh);
                    if (lane == 0) {
                        sh_long[(qi * 4 + (wave - 1)) * 2] = lg;
                        sh_long[(qi * 4 + (wave - 1)) * 2 + 1] = lh;
                    }
                }
            }
            __syncthreads();
            if (!control && wt < qi) {
                const int q = nth_set_bit(lmu, wt);
                double tg = 0.0, th = 0.0;
                for (int w4 = 0; w4 < 4; ++w4) {
                    tg += sh_long[(wt * 4 + w4) * 2];
                    th += sh_long[(wt * 4 + w4) * 2 + 1];
                }
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + q) * 2;
                const unsigned long long tag = prb_tag(b);
                prb_store_granule(sl, tg, tag);
                prb_store_granule(sl + 1, th, tag);
            }
        }
        PRB_STAMP(0)
        double p_next = 0.0;
        int n2e0 = 0, n2e1 = 0;
        unsigned long long lm2 = 0ull;
        // slot bounds + long-slot mask of step b+2: issued before the sweep so that the
        // (scalar) mask load has landed long before the barrier's lgkmcnt(0)
        if (b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
        if (!control) {
            PRB_WSTAMP(1)  // publish issue
            const bool ok = prb_collect_quarter<2>(a, b, wave - 1, lane, ncols, sh_quart);
            if (!ok) *sh_ok = 0;
            PRB_WSTAMP(2)  // granule sweep until every workgroup's partials are in
            if (b + 1 < a.nb) {
                // prefetch (after the exchange: vmcnt retires in order, so streaming loads
                // issued earlier would delay every granule check): entries of step b+1
                // (bounds already in registers), bounds of b+2
                prb_load_entries<T>(a, eval, ne0, ne1, sub, nxt);
                if (slot < c2 - c1) p_next = pold_sched[c1 + slot];
            }
        }
        // B3: quarter sums in LDS.  Raw barrier: only LDS traffic must have landed; the
        // prefetch loads just issued stay in flight across it (a __syncthreads() would
        // add s_waitcnt vmcnt(0) and expose their HBM latency on every step).
        PRB_WSTAMP(3)  // prefetch issue
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PRB_STAMP(3)
        PRB_WSTAMP(4)  // B3
        if (!*sh_ok) break;
        if (control) {
            double tot[2];
#pragma unroll
            for (int v = 0; v < 2; ++v)
                tot[v] = ((sh_quart[(0 * 64 + lane) * 2 + v] + sh_quart[(1 * 64 + lane) * 2 + v]) +
                          sh_quart[(2 * 64 + lane) * 2 + v]) +
                         sh_quart[(3 * 64 + lane) * 2 + v];
            const bool valid = lane < ncols;
            const double res = pcd_chain_lanes<M>(reg, lane, ncols - 1, valid, pl, tot[0], tot[1],
                                                  lam, mu, beta, gamma, eta, cache);
            const double dl = valid ? (pl - res) : 0.0;
            sh_delta[lane] = dl;
            sh_pold[lane] = pl;
            if (g == 0 && valid) {
                ps[jl] = res;
                viol_pos[c0 + lane] = fabs(dl);  // by position; folded into viol_col later
            }
            PRB_STAMP(4)
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B4: deltas in LDS
        PRB_STAMP(5)
        PRB_WSTAMP(5)  // waiting for the control wave's chain
        // ---- phase 3 (workers): scatter-update of the own rows (pcd.py:124-133)
        if (slot < ncols) {
            const double upd = sh_delta[slot];
            if (upd != 0.0) {
                const double p_old = sh_pold[slot];
#pragma unroll
                for (int u = 0; u < PRB_PF; ++u) {
                    if (cur.e0 + sub + 4 * u < cur.e1) {
                        const size_t i = (size_t)cur.row[u];
                        const double x = (double)cur.x[u];
                        if constexpr (M == 0) {  // pcd_all.py:92-98
                            double yn = yh[u] - lam * av[u][0];
                            double an = av[u][0] / (1.0 + x * p_old);
                            an *= 1.0 + x * (p_old - upd);
                            yn += lam * an;
                            A[i] = (T)an;
                            yy[2 * i] = (T)yn;
                        } else {
                            double dprev = x;
#pragma unroll
                            for (int t = 1; t < M; ++t) {
                                const double a1 = av[u][t - 1];
                                const double dcur = x * (a1 - p_old * dprev);
                                A[i * (M - 1) + (t - 1)] = (T)(a1 - upd * dprev);
                                dprev = dcur;
                            }
                            yy[2 * i] = (T)(yh[u] - lam * upd * dlast[u]);
                        }
                    }
                }
                for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4)
                    pcd_sync_entry<T, M>((size_t)a.erow[e], (double)eval[e], p_old, upd, lam, A,
                                         yy);
            }
        }
        if (lmu != 0ull && !control) {  // long slots: every worker thread scatters
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1) {
                const int q = __builtin_ctzll(mm);
                const double upd = sh_delta[q];
                if (upd != 0.0) {
                    const double p_old = sh_pold[q];
                    const int le0 = spb[q], le1 = spb[q + 1];
                    for (int e = le0 + wt; e < le1; e += 256)
                        pcd_sync_entry<T, M>((size_t)a.erow[e], (double)eval[e], p_old, upd, lam,
                                             A, yy);
                }
            }
        }
        cur = nxt;
        p_slot = p_next;
        ne0 = n2e0;
        ne1 = n2e1;
        lm0 = lm1;
        lm1 = lm2;
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        PRB_WSTAMP(6)  // scatter issue
        __syncthreads();  // B5: rows move between slots from step to step
        PRB_STAMP(6)
        PRB_WSTAMP(7)  // B5 (stores acknowledged)
    }
#undef PRB_STAMP
#undef PRB_WSTAMP
    if (stamp && (tid == 0 || tid == 64)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a.stamps[(size_t)g * 16 + (tid == 64 ? 8 : 0) + q] = acc[q];
    }
}

// cd_linear._cd_linear_epoch (optimizer/cd_linear.py:8-33) as one persistent launch:
// same row-block ownership, entry stream and tagged-granule exchange as pcd_prb_kernel,
// one value per slot; the update has no regularizer, so the control wave's "chain" is
// lane-parallel.  w_sched / cn_sched are w and col_norm_sq in visiting order (w as of the
// epoch start: workgroup 0 writes the new w[j] while others may still read the old one).
template <typename T, int LOSS>
__global__ __launch_bounds__(kPrbThreads) void lin_prb_kernel(
    PrbArgs a, const T* __restrict__ eval, T* __restrict__ yy,
    const double* __restrict__ w_sched, const double* __restrict__ cn_sched,
    double* __restrict__ w, double alpha, double mu, double* __restrict__ viol_pos) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double* sh_delta = dyn_lds + 128;  // [64]
    double* sh_quart = dyn_lds + 256;  // [4][64][2]
    int* sh_ok = reinterpret_cast<int*>(dyn_lds + 768);
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool control = wave == 0;
    const int wt = tid - 64;
    const int slot = control ? 64 : (wt >> 2), sub = wt & 3;
    double* sh_long = dyn_lds + 1024;  // [64][4][2]
    PrbEntries<T> cur, nxt;
    int c0 = a.bptr[0], c1 = a.bptr[1];
    int c2 = (a.nb > 1) ? a.bptr[2] : c1;
    int c3 = (a.nb > 2) ? a.bptr[3] : c2;
    unsigned long long lm0 = 0ull, lm1 = 0ull;
    {
        int e0, e1;
        prb_load_sp(a, g, 0, slot, c1 - c0, e0, e1, lm0);
        prb_load_entries<T>(a, eval, e0, e1, sub, cur);
    }
    int ne0 = 0, ne1 = 0;
    if (a.nb > 1) prb_load_sp(a, g, 1, slot, c2 - c1, ne0, ne1, lm1);
    if (tid == 0) *sh_ok = 1;
    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = (b + 4 <= a.nb) ? a.bptr[b + 4] : c3;
        double yh[PRB_PF];
        double wl = 0.0, cnl = 0.0;
        int jl = 0;
        int n2e0 = 0, n2e1 = 0;
        unsigned long long lm2 = 0ull;
        const unsigned long long lmu =
            ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(lm0 >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)lm0);
        if (lmu != 0ull) {  // long slots first: whole-workgroup partial sums (see pcd_prb_kernel)
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            int qi = 0;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1, ++qi) {
                const int q = __builtin_ctzll(mm);
                if (!control) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    double lg = 0.0;
                    for (int e = le0 + wt; e < le1; e += 256) {
                        const typename Vec2<T>::type yv = yy2[(size_t)a.erow[e]];
                        lg += dloss_dev(LOSS, (double)yv.x, (double)yv.y) * (double)eval[e];
                    }
                    lg = wave_sum(lg);
                    if (lane == 0) sh_long[(qi * 4 + (wave - 1)) * 2] = lg;
                }
            }
            __syncthreads();
            if (!control && wt < qi) {
                const int q = nth_set_bit(lmu, wt);
                double tg = 0.0;
                for (int w4 = 0; w4 < 4; ++w4) tg += sh_long[(wt * 4 + w4) * 2];
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + q) * 2;
                prb_store_granule(sl, tg, prb_tag(b));
            }
        }
        if (control) {
            if (lane < ncols) {
                wl = w_sched[c0 + lane];
                cnl = cn_sched[c0 + lane];
                if (g == 0) jl = a.desc[c0 + lane].j;
            }
        } else {
            double yt[PRB_PF];
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {
                const typename Vec2<T>::type yv = yy2[(size_t)cur.row[u]];
                yh[u] = (double)yv.x;
                yt[u] = (double)yv.y;
            }
            double ag = 0.0;
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {
                const bool v = cur.e0 + sub + 4 * u < cur.e1;
                ag += v ? dloss_dev(LOSS, yh[u], yt[u]) * (double)cur.x[u] : 0.0;
            }
            for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {
                const typename Vec2<T>::type yv = yy2[(size_t)a.erow[e]];
                ag += dloss_dev(LOSS, (double)yv.x, (double)yv.y) * (double)eval[e];
            }
            ag += __shfl_xor(ag, 1, kWave);
            ag += __shfl_xor(ag, 2, kWave);
            if (sub == 0 && !((lm0 >> slot) & 1ull)) {
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + slot) * 2;
                prb_store_granule(sl, ag, prb_tag(b));
            }
            if (b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
            const bool ok = prb_collect_quarter<1>(a, b, wave - 1, lane, ncols, sh_quart);
            if (!ok) *sh_ok = 0;
            if (b + 1 < a.nb) {
                prb_load_entries<T>(a, eval, ne0, ne1, sub, nxt);
            }
        }
        if (control && b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // quarter sums in LDS
        if (!*sh_ok) break;
        if (control) {
            const double tot = ((sh_quart[(0 * 64 + lane) * 2] + sh_quart[(1 * 64 + lane) * 2]) +
                                sh_quart[(2 * 64 + lane) * 2]) +
                               sh_quart[(3 * 64 + lane) * 2];
            const bool valid = lane < ncols;
            double upd = tot;           // cd_linear.py:19-24
            upd += alpha * wl;
            const double inv = mu * cnl + alpha;
            upd /= inv;
            if (!valid) upd = 0.0;
            sh_delta[lane] = upd;
            if (g == 0 && valid) {
                w[jl] = wl - upd;
                viol_pos[c0 + lane] = fabs(upd);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // updates in LDS
        if (slot < ncols) {
            const double upd = sh_delta[slot];
            if (upd != 0.0) {
#pragma unroll
                for (int u = 0; u < PRB_PF; ++u)
                    if (cur.e0 + sub + 4 * u < cur.e1)
                        yy[2 * (size_t)cur.row[u]] = (T)(yh[u] - upd * (double)cur.x[u]);
                for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {
                    const size_t i = (size_t)a.erow[e];
                    yy[2 * i] = (T)((double)yy[2 * i] - upd * (double)eval[e]);
                }
            }
        }
        if (lmu != 0ull && !control) {
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1) {
                const int q = __builtin_ctzll(mm);
                const double upd = sh_delta[q];
                if (upd != 0.0) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    for (int e = le0 + wt; e < le1; e += 256) {
                        const size_t i = (size_t)a.erow[e];
                        yy[2 * i] = (T)((double)yy[2 * i] - upd * (double)eval[e]);
                    }
                }
            }
        }
        cur = nxt;
        ne0 = n2e0;
        ne1 = n2e1;
        lm0 = lm1;
        lm1 = lm2;
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        __syncthreads();
    }
}

// out[pos] = v[desc[pos].j]
__global__ void gather_sched_kernel(int d, const ColDesc* __restrict__ desc,
                                    const double* __restrict__ v, double* __restrict__ out) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < d) out[pos] = v[desc[pos].j];
}

// viol_col[desc[pos].j] += viol_pos[pos]   (sum_viol bookkeeping of the persistent pass)
__global__ void fold_viol_kernel(int d, const ColDesc* __restrict__ desc,
                                 const double* __restrict__ viol_pos,
                                 double* __restrict__ viol_col) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < d) viol_col[desc[pos].j] += viol_pos[pos];
}

// in visiting order: out[pos] = P[s, desc[pos].j]
__global__ void snapshot_row_kernel(const Ctl* __restrict__ ctl, const double* __restrict__ P,
                                    int d, const ColDesc* __restrict__ desc,
                                    double* __restrict__ out) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < d) out[pos] = P[(size_t)ctl->s * d + desc[pos].j];
}

// erow/eval = cidx/cval gathered through the host-built entry permutation
template <typename T>
__global__ void prb_gather_kernel(int64_t nnz, const int32_t* __restrict__ src,
                                  const int32_t* __restrict__ cidx, const T* __restrict__ cval,
                                  int32_t* __restrict__ erow, T* __restrict__ eval) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) {
        const int32_t q = src[e];
        erow[e] = cidx[q];
        eval[e] = cval[q];
    }
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
    if constexpr (M == 0) {  // pbcd_all.py:9-20
        double a = 1.0;
        for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii)
            a *= 1.0 + P[(size_t)ridx[ii] * k + s] * (double)rval[ii];
        A[(size_t)i * k + s] = (T)a;
    } else {
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
        for (int t = 1; t < M; ++t)
            A[(size_t)i * (M - 1) * k + (size_t)(t - 1) * k + s] = (T)a[t];
    }
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
    if constexpr (M == 0) {  // omegacs.py:60-62: _cache_all_subsets = prod_j (1 + norm_j)
        double pr = 1.0;
        for (int j = tid; j < d; j += kBlock) pr *= 1.0 + rs.norms[j];
        sh[tid] = pr;
        __syncthreads();
        for (int half = kBlock / 2; half >= 1; half >>= 1) {
            if (tid < half) sh[tid] *= sh[tid + half];
            __syncthreads();
        }
        if (tid == 0) rs.cache[0] = sh[0];
        return;
    }
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

// One pbcd step = four launches:
//   pbcd_grad_kernel   kPbW workgroups per column: partial sums of the first pass of
//                      pbcd._update (optimizer/pbcd.py:56-67)
//   pbcd_prep_kernel   one wave per column, lanes over components: step size, gradient
//                      step (pbcd.py:68-78) and everything of prox_bcd that does not depend
//                      on the regularizer's running cache (block norm, L1 / L21 prox)
//   pbcd_chain_kernel  one wave, lanes over columns: the scalar cache recurrences of
//                      SquaredL21 / OmegaCS in batch order (squaredl21.py:40-55,
//                      omegacs.py:68-106) -> one shrink factor per column
//   pbcd_sync_kernel   kPbW workgroups per column: p_j = f * p_j', P[j] write-back and
//                      "synchronize predictions and caches" (pbcd.py:135-144)
// Rounding note: update_cache_pbcd's l2 = ||P[j]|| after the prox is taken as f * ||p_j'||
// (equal up to ~2 ulp) so that the chain needs no vector work.
constexpr int kPbW = 8;  // workgroups per column in the gather / scatter kernels

template <typename T, int M, int L, int C>
__global__ __launch_bounds__(kBlock) void pbcd_grad_kernel(
    const ColDesc* __restrict__ desc, const int32_t* __restrict__ cidx,
    const T* __restrict__ cval, const T* __restrict__ A,
    const typename Vec2<T>::type* __restrict__ yy, const double* __restrict__ P /* (d,k) */,
    int k, int loss, double* __restrict__ part /* [ncols][kPbW][k+1] */) {
    constexpr int G = kBlock / L;  // entry groups per workgroup
    constexpr int U = 4;           // entries per group in flight
    extern __shared__ double shm[];  // G * k + 16
    double* red = shm + (size_t)G * k;
    const int q = blockIdx.x / kPbW, w = blockIdx.x % kPbW;
    const ColDesc cd = desc[q];
    const int grp = threadIdx.x / L, lane = threadIdx.x % L;
    double p[C], grad[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        p[c] = (s < k) ? P[(size_t)cd.j * k + s] : 0.0;
        grad[c] = 0.0;
    }
    double hs = 0.0, dummy = 0.0;
    const int64_t e = cd.start + cd.len;
    constexpr int AS = Kind<M>::AS;
    const size_t slab = (size_t)AS * k;
    for (int64_t ii0 = cd.start + (int64_t)w * G + grp; ii0 < e; ii0 += (int64_t)U * G * kPbW) {
        int iu[U];
        double xu[U], dlu[U];
        double au[U][C][AS];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // all loads of U entries in flight together
            const int64_t ii = ii0 + (int64_t)u * G * kPbW;
            const bool v = ii < e;
            iu[u] = v ? cidx[ii] : -1;
            xu[u] = v ? (double)cval[ii] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (iu[u] >= 0) {
                const typename Vec2<T>::type yv = yy[iu[u]];
                dlu[u] = dloss_dev(loss, (double)yv.x, (double)yv.y);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int s = lane + c * L;
#pragma unroll
                    for (int t = 0; t < AS; ++t)
                        au[u][c][t] =
                            (s < k) ? (double)A[(size_t)iu[u] * slab + (size_t)t * k + s] : 0.0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (iu[u] >= 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int s = lane + c * L;
                    if (s < k) {
                        const double dprev = grad_factor<M>(au[u][c], xu[u], p[c]);
                        grad[c] += dlu[u] * dprev;
                        hs += dprev * dprev;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        if (s < k) shm[(size_t)grp * k + s] = grad[c];
    }
    block_sum2(hs, dummy, red);  // contains the __syncthreads that publishes shm
    double* out = part + ((size_t)q * kPbW + w) * (k + 1);
    for (int s = threadIdx.x; s < k; s += kBlock) {
        double acc = 0.0;
        for (int g2 = 0; g2 < G; ++g2) acc += shm[(size_t)g2 * k + s];
        out[s] = acc;
    }
    if (threadIdx.x == 0) out[k] = hs;
}

// per column: scal[q] = {l2 of p_j', st0 = eta*gamma/inv, f (L1/L21: final), unused}
template <int C>
__global__ __launch_bounds__(kWave) void pbcd_prep_kernel(
    const ColDesc* __restrict__ desc, const double* __restrict__ P /* (d,k) */, int k,
    const double* __restrict__ part, const double* __restrict__ lams, int reg, double mu,
    double beta, double gamma, double eta, double* __restrict__ pin /* [ncols][k] p_j' */,
    double* __restrict__ pold /* [ncols][k] */, double* __restrict__ scal /* [ncols][4] */) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int j = desc[q].j;
    const double* pq = part + (size_t)q * kPbW * (k + 1);
    double inv = 0.0;
#pragma unroll
    for (int w = 0; w < kPbW; ++w) inv += pq[(size_t)w * (k + 1) + k];
    inv *= mu;
    inv += beta;
    const double st0 = eta * gamma / inv;
    double p[C];
    double sq = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * kWave;
        p[c] = 0.0;
        if (s < k) {
            const double po = P[(size_t)j * k + s];
            double g = 0.0;
#pragma unroll
            for (int w = 0; w < kPbW; ++w) g += pq[(size_t)w * (k + 1) + s];
            g *= lams[s];
            g += beta * po;
            g /= inv;
            double v = po - eta * g;
            if (reg == REG_L1) {  // l1.py:44-45, element-wise
                const double sg = (v > 0) ? 1.0 : ((v < 0) ? -1.0 : 0.0);
                const double m = fabs(v) - st0;
                v = sg * (m > 0.0 ? m : 0.0);
            } else if (reg == REG_SQL21) {
                v /= 1 + 2 * st0;  // squaredl21.py:46
            }
            p[c] = v;
            pold[(size_t)q * k + s] = po;
            pin[(size_t)q * k + s] = v;
            sq += v * v;
        }
    }
    const double l2 = sqrt(wave_sum(sq));
    if (lane == 0) {
        double f = 1.0;
        if (reg == REG_L21) f = (l2 > st0) ? (1.0 - st0 / l2) : 0.0;  // l21.py:33-38
        scal[4 * q + 0] = l2;
        scal[4 * q + 1] = st0;
        scal[4 * q + 2] = f;
        scal[4 * q + 3] = 0.0;
    }
}

// Scalar chain for SquaredL21 / OmegaCS: lanes = columns (64 at a time), wave-uniform
// serial loop; writes the shrink factor f into scal[q][2] and the new block norm into
// norms[j].  Fallback branches ("numerical error": squaredl21.py:48-49,
// omegacs.py:75-76,90-96) recompute from all d norms with the whole wave.
template <int M>
__global__ __launch_bounds__(kWave) void pbcd_chain_kernel(
    const ColDesc* __restrict__ desc, int ncols, int d, int reg, RegState rs, int top_ncache,
    double* __restrict__ scal) {
    const int lane = threadIdx.x;
    double cache[kMaxDegree + 2], dcache[kMaxDegree + 2];
    {   // one vector load each, then broadcast (the state is wave-uniform)
        const double cv = (lane < top_ncache) ? rs.cache[lane] : 0.0;
        const double dv = (lane < top_ncache) ? rs.dcache[lane] : 0.0;
#pragma unroll
        for (int t = 0; t < kMaxDegree + 2; ++t) {
            cache[t] = readlane_d(cv, t);
            dcache[t] = readlane_d(dv, t);
        }
    }
    for (int base = 0; base < ncols; base += kWave) {
        const int q = base + lane;
        const bool valid = q < ncols;
        const int cnt = min(kWave, ncols - base);
        int j = 0;
        double l2 = 0.0, st0 = 0.0, njl = 0.0;
        if (valid) {
            j = desc[q].j;
            l2 = scal[4 * q + 0];
            st0 = scal[4 * q + 1];
            njl = rs.norms[j];
        }
        if constexpr (M == 2) {
            // Degree 2: the cache is one scalar c (= sum of block norms) and column i maps it
            // through c' = (c - n_i) + max(l2_i - t_i (c - n_i), 0), t_i = st0 (omegacs) or
            // 2 st0 / (1 + 2 st0) (squaredl21): the same piecewise-affine recurrence as
            // pcd's squaredl12, solved by the speculative affine scan.  If any column would
            // take one of the reference's "numerical error" branches the chunk is redone by
            // the serial loop below, which restates them.
            const double c0 = (reg == REG_SQL21) ? cache[0] : cache[1];
            const double tt = (reg == REG_SQL21) ? (2 * st0 / (1.0 + 2 * st0)) : st0;
            bool nz = valid && (l2 - tt * (c0 - njl)) > 0;
            double cb = c0, m = 0.0, al = 1.0, be = 0.0;
            for (int round = 0; round <= kWave; ++round) {
                al = valid ? (nz ? (1.0 - tt) : 1.0) : 1.0;
                be = valid ? (nz ? (l2 - (1.0 - tt) * njl) : -njl) : 0.0;
                affine_scan_inclusive(al, be, lane);
                cb = affine_before(al, be, c0, lane);
                m = l2 - tt * (cb - njl);
                const bool nz2 = m > 0;
                const unsigned long long bad = __ballot(valid && (nz2 != nz));
                nz = nz2;
                if (bad == 0ull) break;
            }
            const double l2n = (valid && nz) ? m : 0.0;
            const double dc2 = cb - njl;  // dcache[2] (omegacs) / dcache (squaredl21)
            // cache[2] += dcache[2] * l2n - dcache[2] * n_j per column (omegacs.py:71-73)
            double c2term = (valid && reg == REG_OMEGACS) ? (dc2 * l2n - dc2 * njl) : 0.0;
            double c2pre = c2term;  // inclusive prefix sum
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const double v = __shfl_up(c2pre, o, kWave);
                if (lane >= o) c2pre += v;
            }
            const double c_after = al * c0 + be;  // cache after this column
            const bool trouble = valid && ((dc2 < 0) || (c_after < 0) ||
                                           (reg == REG_OMEGACS && cache[2] + c2pre < 0));
            if (__ballot(trouble) == 0ull) {
                const double f = (valid && nz) ? (1.0 - (tt * dc2) / l2) : 0.0;
                if (valid) {
                    scal[4 * q + 2] = f;
                    rs.norms[j] = l2n;  // = l2 - strength, the value the scan propagated
                }
                const double c_end = readlane_d(c_after, cnt - 1);
                if (reg == REG_SQL21) {
                    cache[0] = c_end;
                } else {
                    cache[1] = c_end;
                    cache[2] += readlane_d(c2pre, cnt - 1);
                    dcache[2] = readlane_d(dc2, cnt - 1);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                continue;
            }
        }
        if constexpr (M == 0) {
            // all-subsets OmegaCS (omegacs.py:99-106, 77-81): c /= 1 + n_j; strength = st0 c;
            // shrink; c *= 1 + new norm.  Multiplicative: serial loop over the chunk.
            double c = cache[0];
            double f_m = 0.0, l2n_m = 0.0;
            for (int i = 0; i < cnt; ++i) {
                const double l2i = readlane_d(l2, i), si = readlane_d(st0, i),
                             nj = readlane_d(njl, i);
                c /= 1.0 + nj;
                const double strength = si * c;
                const double f = (l2i > strength) ? (1.0 - strength / l2i) : 0.0;
                const double l2n = f * l2i;
                c *= 1.0 + l2n;
                if (lane == i) {
                    f_m = f;
                    l2n_m = l2n;
                }
            }
            cache[0] = c;
            if (valid) {
                scal[4 * q + 2] = f_m;
                rs.norms[j] = l2n_m;
            }
            continue;
        }
        double f_mine = 0.0, l2n_mine = 0.0;
// rare fallback paths re-read all d norms from memory: first store the norms of the
// columns of this chunk that were already processed (they live in registers)
#define PBCD_FLUSH_NORMS                                            \
    {                                                               \
        if (valid && lane < i) rs.norms[j] = l2n_mine;              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      \
    }
        for (int i = 0; i < cnt; ++i) {
            const double l2i = readlane_d(l2, i), si = readlane_d(st0, i);
            double nj = readlane_d(njl, i);
            const int ji = __builtin_amdgcn_readlane(j, i);
            double strength;
            if (reg == REG_SQL21) {
                if (cache[0] < nj) {  // squaredl21.py:48-49
                    PBCD_FLUSH_NORMS
                    double a = 0.0;
                    for (int jj = lane; jj < d; jj += kWave) a += rs.norms[jj];
                    cache[0] = wave_sum(a);
                }
                const double dc = cache[0] - nj;
                strength = 2 * dc * si / (1.0 + 2 * si);
            } else {  // REG_OMEGACS
#pragma unroll
                for (int deg = 2; deg <= M; ++deg) {
                    dcache[deg] = cache[deg - 1];
                    dcache[deg] -= dcache[deg - 1] * nj;
                }
                double mn = dcache[0];
#pragma unroll
                for (int t = 1; t < kMaxDegree + 2; ++t)
                    if (t < top_ncache && dcache[t] < mn) mn = dcache[t];
                if (mn < 0) {  // omegacs.py:90-96
                    PBCD_FLUSH_NORMS
                    double cc[kMaxDegree + 2];
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                    for (int jj = lane; jj < d; jj += kWave) {
                        const double v = (jj == ji) ? 0.0 : rs.norms[jj];
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
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t) cache[t] = (t < M) ? cc[t] : 0.0;
                    dcache[0] = 0.0;
                    dcache[1] = 1.0;
#pragma unroll
                    for (int deg = 2; deg <= M; ++deg) dcache[deg] = cache[M - 1];
                    nj = 0.0;  // self._norms[j] = 0.0
                }
                strength = si * dcache[M];
            }
            const double f = (l2i > strength) ? (1.0 - strength / l2i) : 0.0;
            const double l2n = f * l2i;
            if (reg == REG_SQL21) {  // squaredl21.py:40-43
                cache[0] -= nj;
                cache[0] += l2n;
            } else {  // omegacs.py:68-76
#pragma unroll
                for (int deg = 1; deg <= M; ++deg) {
                    cache[deg] += dcache[deg] * l2n;
                    cache[deg] -= dcache[deg] * nj;
                }
                double mn = cache[0];
#pragma unroll
                for (int t = 1; t < kMaxDegree + 2; ++t)
                    if (t < top_ncache && cache[t] < mn) mn = cache[t];
                if (mn < 0) {  // __recompute_cache_bcd(degree)
                    PBCD_FLUSH_NORMS
                    double cc[kMaxDegree + 2];
#pragma unroll
                    for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                    for (int jj = lane; jj < d; jj += kWave) {
                        double v = rs.norms[jj];
                        if (jj == ji) v = l2n;
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
                    for (int t = 0; t < kMaxDegree + 2; ++t) cache[t] = (t <= M) ? cc[t] : 0.0;
                }
            }
            if (lane == i) {
                f_mine = f;
                l2n_mine = l2n;
            }
        }
        if (valid) {
            scal[4 * q + 2] = f_mine;
            rs.norms[j] = l2n_mine;
        }
        // the next chunk (and its fallback paths) read rs.norms of this chunk's columns
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#undef PBCD_FLUSH_NORMS
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < kMaxDegree + 2; ++t)
            if (t < top_ncache) {
                rs.cache[t] = cache[t];
                rs.dcache[t] = dcache[t];
            }
    }
}

// p_j = f * p_j' (prox_bcd's shrink), P[j] write-back, sum_viol (pbcd.py:146) and
// "synchronize predictions and caches" (optimizer/pbcd.py:135-144)
template <typename T, int M, int L, int C>
__global__ __launch_bounds__(kBlock) void pbcd_sync_kernel(
    const ColDesc* __restrict__ desc, const int32_t* __restrict__ cidx,
    const T* __restrict__ cval, T* __restrict__ A, T* __restrict__ yy,
    const double* __restrict__ lams, int k, double* __restrict__ P /* (d,k) */,
    const double* __restrict__ pin, const double* __restrict__ pold,
    const double* __restrict__ scal, double* __restrict__ viol_col) {
    constexpr int G = kBlock / L;
    const int q = blockIdx.x / kPbW, w = blockIdx.x % kPbW;
    const ColDesc cd = desc[q];
    const int grp = threadIdx.x / L, lane = threadIdx.x % L;
    const double f = scal[4 * q + 2];
    double po[C], up[C], lu[C];
    bool any = false;
    double va = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        po[c] = 0.0;
        up[c] = 0.0;
        lu[c] = 0.0;
        if (s < k) {
            po[c] = pold[(size_t)q * k + s];
            const double pn = pin[(size_t)q * k + s] * f;
            up[c] = po[c] - pn;
            lu[c] = lams[s] * up[c];
            any |= (up[c] != 0.0);
            if (w == 0 && grp == 0) {
                P[(size_t)cd.j * k + s] = pn;
                va += fabs(up[c]);
            }
        }
    }
    if (w == 0 && grp == 0) {
        va = group_sum(va, L);
        if (lane == 0) viol_col[cd.j] += va;
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;  // block did not move: exact no-op
    const int64_t e = cd.start + cd.len;
    const size_t slab = (size_t)Kind<M>::AS * k;
    constexpr int U = 4;  // entries per group in flight
    constexpr int AS = Kind<M>::AS;
    double lamc[C], pn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = lane + c * L;
        lamc[c] = (s < k) ? lams[s] : 0.0;
        pn[c] = po[c] - up[c];
    }
    for (int64_t ii0 = cd.start + (int64_t)w * G + grp; ii0 < e; ii0 += (int64_t)U * G * kPbW) {
        int iu[U];
        double xu[U];
        double au[U][C][AS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ii = ii0 + (int64_t)u * G * kPbW;
            const bool v = ii < e;
            iu[u] = v ? cidx[ii] : -1;
            xu[u] = v ? (double)cval[ii] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (iu[u] >= 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int s = lane + c * L;
#pragma unroll
                    for (int t = 0; t < AS; ++t)
                        au[u][c][t] =
                            (s < k) ? (double)A[(size_t)iu[u] * slab + (size_t)t * k + s] : 0.0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (iu[u] >= 0) {
                const size_t i = (size_t)iu[u];
                if constexpr (M == 0) {  // pbcd_all.py:121-127
                    double d_old = 0.0, d_new = 0.0;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int s = lane + c * L;
                        if (s < k) {
                            const double a0 = au[u][c][0];
                            double a1 = a0 / (1.0 + xu[u] * po[c]);
                            a1 *= 1.0 + xu[u] * pn[c];
                            A[i * slab + s] = (T)a1;
                            d_old += lamc[c] * a0;
                            d_new += lamc[c] * a1;
                        }
                    }
                    d_old = group_sum(d_old, L);
                    d_new = group_sum(d_new, L);
                    if (lane == 0) yy[2 * i] = (T)(((double)yy[2 * i] - d_old) + d_new);
                } else {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int s = lane + c * L;
                        if (s < k) {
                            double dprev = xu[u];
#pragma unroll
                            for (int t = 1; t < M; ++t) {
                                const double a = au[u][c][t - 1];
                                const double dcur = xu[u] * (a - po[c] * dprev);
                                A[i * slab + (size_t)(t - 1) * k + s] = (T)(a - up[c] * dprev);
                                dprev = dcur;
                            }
                            acc += lu[c] * dprev;
                        }
                    }
                    acc = group_sum(acc, L);
                    if (lane == 0) yy[2 * i] = (T)((double)yy[2 * i] - acc);
                }
            }
        }
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
