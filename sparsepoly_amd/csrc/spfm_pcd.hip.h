// spfm_pcd.hip.h -- pcd: precompute, regularizer cache, gradient, chain (affine scan), scatter
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
#pragma once
#include "spfm_common.hip.h"

namespace spfm {

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
// e_0..e_M of |P[s,:]|).  Workgroup b reduces its slice of the features (one workgroup
// alone is latency-starved: 150 us for 100k features) into part[b][0..M]: e_t by
// per-thread DP over a strided subset, then a tree of truncated polynomial products (e_t
// is symmetric, so any partition of the features gives the same value up to rounding);
// pcd_cache_combine_kernel folds the partials in fixed order.
constexpr int kCacheBlocks = 64;
template <int M>
__global__ __launch_bounds__(kBlock) void pcd_compute_cache_kernel(const Ctl* __restrict__ ctl,
                                                                    const double* __restrict__ P,
                                                                    int d_all, int reg,
                                                                    double* __restrict__ part) {
    __shared__ double sh[kBlock * (M + 1)];
    const int per = (d_all + gridDim.x - 1) / gridDim.x;
    const int lo = min(per * (int)blockIdx.x, d_all);
    const int d = min(per, d_all - lo);
    const double* ps = P + (size_t)ctl->s * d_all + lo;
    double* cache = part + (size_t)blockIdx.x * (M + 1);
    const int tid = threadIdx.x;
    if (reg == REG_SQL12) {
        // 8 independent loads in flight per thread (same summation order as a plain loop:
        // the padding adds exact zeros)
        double a = 0, b = 0;
        for (int j0 = tid; j0 < d; j0 += kBlock * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * kBlock;
                v[u] = (j < d) ? fabs(ps[j]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        block_sum2(a, b, sh);
        if (tid == 0) cache[0] = a;
        return;
    }
    if (reg != REG_OMEGATI) return;
    if constexpr (M == 0) {  // omegati.py:75-80: _cache_all_subsets = prod_j (1 + |p_sj|)
        double pr = 1.0;
        for (int j0 = tid; j0 < d; j0 += kBlock * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * kBlock;
                v[u] = (j < d) ? fabs(ps[j]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) pr *= 1.0 + v[u];
        }
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
    for (int j0 = tid; j0 < d; j0 += kBlock * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * kBlock;
            v[u] = (j < d) ? fabs(ps[j]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int t = M; t >= 1; --t) c[t] += c[t - 1] * v[u];
        }
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

// cache <- combination of the per-workgroup partials, in workgroup order (one thread)
template <int M>
__global__ void pcd_cache_combine_kernel(int reg, int nblk, const double* __restrict__ part,
                                         double* __restrict__ cache) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (reg == REG_SQL12) {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += part[(size_t)b * (M + 1)];
        cache[0] = a;
        return;
    }
    if (reg != REG_OMEGATI) return;
    if constexpr (M == 0) {
        double pr = 1.0;
        for (int b = 0; b < nblk; ++b) pr *= part[b];
        cache[0] = pr;
    } else {
        double c[M + 1];
#pragma unroll
        for (int t = 0; t <= M; ++t) c[t] = part[t];
        for (int b = 1; b < nblk; ++b) {
            double o[M + 1];
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u <= t; ++u) acc += c[u] * part[(size_t)b * (M + 1) + (t - u)];
                o[t] = acc;
            }
#pragma unroll
            for (int t = 0; t <= M; ++t) c[t] = o[t];
        }
#pragma unroll
        for (int t = 0; t <= M; ++t) cache[t] = c[t];
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
// Lane movement for the scan uses DPP (data-parallel primitives: the operand of a VALU
// move is taken from another lane, ~one issue cycle) instead of ds_bpermute-based
// __shfl_up (an LDS-crossbar round trip per step): the chain is on the critical path of
// every dependent step.  gfx9/CDNA controls: row_shr:n (within rows of 16 lanes),
// row_bcast:15 / row_bcast:31 (last lane of a row / of the lower half to the following
// rows), wave_shr:1.  A lane without a valid source keeps `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_d(double old, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// one scan step: compose the map of the lanes in front (identity where there is none)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void affine_scan_step(double& al, double& be) {
    const double oa = dpp_move_d<CTRL, ROW_MASK>(1.0, al);
    const double ob = dpp_move_d<CTRL, ROW_MASK>(0.0, be);
    be = al * ob + be;  // mine after other: x -> al*(oa*x + ob) + be
    al = al * oa;
}

__device__ __forceinline__ void affine_scan_inclusive(double& al, double& be, int /*lane*/) {
    affine_scan_step<0x111, 0xf>(al, be);  // row_shr:1
    affine_scan_step<0x112, 0xf>(al, be);  // row_shr:2
    affine_scan_step<0x114, 0xf>(al, be);  // row_shr:4
    affine_scan_step<0x118, 0xf>(al, be);  // row_shr:8   -> every row of 16 is scanned
    affine_scan_step<0x142, 0xa>(al, be);  // row_bcast:15 into rows 1 and 3
    affine_scan_step<0x143, 0xc>(al, be);  // row_bcast:31 into rows 2 and 3
}

// inclusive prefix sum over the wave with the same DPP moves (a lane without a source adds 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void sum_scan_step(double& v) {
    v += dpp_move_d<CTRL, ROW_MASK>(0.0, v);
}
__device__ __forceinline__ double dpp_prefix_sum_inclusive(double v) {
    sum_scan_step<0x111, 0xf>(v);  // row_shr:1
    sum_scan_step<0x112, 0xf>(v);  // row_shr:2
    sum_scan_step<0x114, 0xf>(v);  // row_shr:4
    sum_scan_step<0x118, 0xf>(v);  // row_shr:8
    sum_scan_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    sum_scan_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// value of the cache in front of column `lane` given c0 and the inclusive scan
__device__ __forceinline__ double affine_before(double al_inc, double be_inc, double c0, int /*lane*/) {
    const double pa = dpp_move_d<0x138, 0xf>(0.0, al_inc);  // wave_shr:1; lane 0: x -> c0
    const double pb = dpp_move_d<0x138, 0xf>(c0, be_inc);
    return pa * c0 + pb;
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
                                                  double eta, double (&cache)[M + 1],
                                                  double* sc /* LDS [68][4]; used for M > 2 */) {
    double pin = 0.0, st = 0.0;
    if (valid) {
        double inv = h * mu;
        inv += beta;
        double upd = g * lam;
        upd += beta * p_old;
        const double rinv = recip_nr(inv);  // one reciprocal for both quotients (see recip_nr)
        upd *= rinv;
        pin = p_old - eta * upd;
        st = (eta * gamma) * rinv;
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
        const double rden = recip_nr(den);
        const double pp = pin * rden;
        const double app = fabs(pp);
        const double tt = (2 * st) * rden;
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
            if (__ballot(act && !pos) != 0ull) count_branch(BR_OMEGATI_CLIP, lane);
            cache[1] = readlane_d(al, last) * c0 + readlane_d(be, last);
            return sg * r;
        }
        if constexpr (M == 3) {
            // Degree 3 in parallel (round 3).  The column map is not affine in the cache -- but
            // GIVEN the columns' results r_i the two cache values in front of every column are
            // plain prefix sums (omegati.py:82-99 at degree 3, no clip):
            //     c1 in front of i = c1_0 + sum_{k<i} (r_k - a_k),         dc2_i = c1 - a_i
            //     c2 in front of i = c2_0 + sum_{k<i} dc2_k (r_k - a_k),   dc3_i = c2 - dc2_i a_i
            // and r_i = max(|p_i| - s_i dc3_i, 0) depends on them only through the strength s_i
            // (~1e-7): guess r, two DPP scans, recompute r, until no r changes a bit -- two or
            // three rounds instead of a 37-column dependent loop (2.2 us of a 5.8 us step on
            // config 3).  The fixed point is the sequential result up to the association of the
            // sums.  A negative dc (the clips of omegati.py:97-98, rounding only) or no fixed
            // point in 8 rounds: the serial form below takes the step.
            const bool act = valid && lane <= last;
            const double a_i = act ? ab : 0.0, s_i = act ? st : 0.0, p_i = act ? apin : 0.0;
            const double c10 = cache[1], c20 = cache[2];
            double r;
            {
                const double dc2 = c10 - a_i;
                const double m0 = p_i - s_i * (c20 - dc2 * a_i);
                r = (m0 > 0) ? m0 : 0.0;
            }
            bool fixed = false;
            unsigned long long negl = 0ull;
            double be1 = 0.0, be2 = 0.0;
            for (int round = 0; round < 8 && !fixed; ++round) {
                const double d1 = r - a_i;
                double al1 = 1.0;
                be1 = d1;
                affine_scan_inclusive(al1, be1, lane);
                const double c1b = affine_before(al1, be1, c10, lane);
                const double dc2 = c1b - a_i;
                double al2 = 1.0;
                be2 = dc2 * d1;
                affine_scan_inclusive(al2, be2, lane);
                const double c2b = affine_before(al2, be2, c20, lane);
                const double dc3 = c2b - dc2 * a_i;
                const double m = p_i - s_i * dc3;
                const double rn = (m > 0) ? m : 0.0;
                negl = __ballot(act && ((__double2hiint(dc2) | __double2hiint(dc3)) < 0));
                fixed = __ballot(__double_as_longlong(rn) != __double_as_longlong(r)) == 0ull;
                r = rn;
            }
            if (fixed && negl == 0ull) {
                cache[1] = c10 + readlane_d(be1, 63);
                cache[2] = c20 + readlane_d(be2, 63);
                return sg * (act ? r : 0.0);
            }
        }
        if constexpr (M > 2) {
            // Serial over the columns (the map of a nonzero column is not affine in the cache).
            // A lone wave issues one instruction every four cycles (eight for an f64 operation),
            // so this loop -- the critical path of every step of a degree >= 3 pass, 3.6 us of a
            // 7.6 us step at 37 columns when written with v_readlane broadcasts, a clip and a
            // conditional move per column -- is bound by its instruction COUNT.  Hence:
            //  * the per-column operands (|p_old|, strength, |p_in|) go to LDS once and come back
            //    as broadcast reads (2 LDS instructions instead of 6 v_readlane);
            //  * the loop runs in lane 0 alone, which makes "store this column's result" one LDS
            //    write instead of a compare and two conditional moves;
            //  * the clips of omegati.py:97-98 only ever fire through rounding (the cache values
            //    are elementary symmetric sums of |p| with one term removed: >= 0 in exact
            //    arithmetic), so the first attempt leaves them out -- v used as it is, the sign
            //    bits OR-ed into one word -- and a step that saw a negative v (or a -0.0) is redone
            //    with the clips in place.  Where no clip fires both forms compute the same bits.
            //  * columns go in blocks of four, the next block's operands fetched while the current
            //    one is computed; lanes behind `last` hold (0, 0, 0), for which a column is an
            //    exact no-op, so there is no tail loop (sc has one block of slack: [68][4]).
            {
                const bool in = lane <= last;
                double* my = sc + lane * 4;
                my[0] = in ? ab : 0.0;
                my[1] = in ? st : 0.0;
                my[2] = in ? apin : 0.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS ops are in order
            double c[M + 1];
#pragma unroll
            for (int t = 0; t <= M; ++t) c[t] = cache[t];
            unsigned neg = 0u;
            const int nblk = (last + 4) >> 2;
            auto sweep = [&](auto clip_tag) __attribute__((always_inline)) {
                constexpr bool CLIP = decltype(clip_tag)::value;
                double na[4], ns[4], np[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    na[u] = sc[u * 4];
                    ns[u] = sc[u * 4 + 1];
                    np[u] = sc[u * 4 + 2];
                }
                for (int blk = 0; blk < nblk; ++blk) {
                    double ca[4], cs[4], cp[4];
                    double* nx = sc + (blk + 1) * 16;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ca[u] = na[u];
                        cs[u] = ns[u];
                        cp[u] = np[u];
                        na[u] = nx[u * 4];
                        ns[u] = nx[u * 4 + 1];
                        np[u] = nx[u * 4 + 2];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double ai = ca[u], si = cs[u], pi = cp[u];
                        double dc[M + 2];
                        dc[1] = 1.0;
#pragma unroll
                        for (int deg = 2; deg <= M; ++deg) {
                            double v = c[deg - 1];
                            v -= dc[deg - 1] * ai;
                            if constexpr (CLIP) {
                                dc[deg] = (v < 0) ? 0.0 : v;
                            } else {
                                neg |= (unsigned)__double2hiint(v);
                                dc[deg] = v;
                            }
                        }
                        const double m = pi - si * dc[M];
                        const double r = (m > 0) ? m : 0.0;
#pragma unroll
                        for (int deg = 1; deg < M; ++deg) c[deg] = dc[deg + 1] + dc[deg] * r;
                        sc[(blk * 4 + u) * 4 + 3] = r;
                    }
                }
            };
            if (lane == 0) sweep(std::false_type{});
            if (__builtin_amdgcn_readfirstlane((int)neg) < 0) {
                count_branch(BR_OMEGATI_CLIP, lane);
#pragma unroll
                for (int t = 0; t <= M; ++t) c[t] = cache[t];
                if (lane == 0) sweep(std::true_type{});
            }
#pragma unroll
            for (int t = 1; t < M; ++t) cache[t] = readlane_d(c[t], 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mine = (lane <= last) ? sc[lane * 4 + 3] : 0.0;
        }
        return sg * mine;
    }
}


// The same chain for a WIDE step: up to 8 x 64 columns, one per thread of a 512-thread workgroup,
// wave w holding columns 64 w .. 64 w + 63 (degree 2; pcdw_kernel).  pcd_chain_lanes called once
// per 64 columns by one wave costs a full scan per call, six in a row for a step of 364 columns
// while seven waves wait.  Here all waves scan their columns at once:
//   1. every wave guesses its columns' branches at the step's input cache c0 (the cache moves by
//      ~1e-7 per column), scans, and publishes its total map c -> A c + B;
//   2. barrier; a wave's true input is the maps of the waves in front of it applied to c0 in
//      order -- the expression the sequential rounds evaluate (cache = A * cache + B);
//   3. the wave checks its branches against that input and re-scans until they agree; if any wave
//      changed a branch (hence its map), the changed maps are published again and 2-3 repeat.
// The inclusive scans depend on the branches only, so with the same (unique, sequentially
// consistent) branches every value is the bit pattern the one-wave form produces.  All eight
// waves must call (barriers inside).  sh_map: LDS [8][2] doubles, sh_bad: LDS 3 ints.
template <int M>
__device__ __forceinline__ double pcd_chain_waves(int reg, int lane, int wave, int ncols, bool valid,
                                                  double p_old, double g, double h, double lam,
                                                  double mu, double beta, double gamma, double eta,
                                                  double (&cache)[M + 1], double* sh_map,
                                                  int* sh_bad) {
    static_assert(M == 2, "wide steps: degree 2");
    double pin = 0.0, st = 0.0;
    if (valid) {
        double inv = h * mu;
        inv += beta;
        double upd = g * lam;
        upd += beta * p_old;
        const double rinv = recip_nr(inv);
        upd *= rinv;
        pin = p_old - eta * upd;
        st = (eta * gamma) * rinv;
    }
    if (reg == REG_L1) {
        const double sg = (pin > 0) ? 1.0 : ((pin < 0) ? -1.0 : 0.0);
        const double m = fabs(pin) - st;
        return sg * (m > 0.0 ? m : 0.0);
    }
    const bool sq = reg == REG_SQL12;
    const int nw = (ncols + 63) >> 6;                   // waves that hold columns
    const int last = min(64, ncols - 64 * wave) - 1;    // this wave's last column (< 0: none)
    const bool act = valid && lane <= last;
    const double ab = fabs(p_old);
    // lane constants: the column's result is m = app - tt * u with u = c - ab (squaredl12.py:52-57,
    // pre-scaled as in pcd_chain_lanes) or u = max(c - ab, 0) (omegati.py:82-99 at degree 2)
    double app, tt, sg;
    if (sq) {
        const double rden = recip_nr(1 + 2 * st);
        const double pp = pin * rden;
        app = fabs(pp);
        tt = (2 * st) * rden;
        sg = (pp > 0) ? 1.0 : -1.0;
    } else {
        app = fabs(pin);
        tt = st;
        sg = (pin > 0) ? 1.0 : -1.0;
    }
    const int ci = sq ? 0 : 1;
    const double c0 = cache[ci];
    bool pos = sq ? true : ((c0 - ab) >= 0);
    bool nz = (app - tt * (pos ? (c0 - ab) : 0.0)) > 0;
    double al = 1.0, be = 0.0, m = 0.0;
    auto scan = [&]() __attribute__((always_inline)) {
        if (!act) {
            al = 1.0;
            be = 0.0;
        } else if (!pos) {  // omegati clip: u = 0, r = p
            al = 0.0;
            be = app;
        } else if (nz) {    // c' = (1 - tt)(c - a) + app
            al = 1.0 - tt;
            be = app - al * ab;
        } else {            // c' = c - a
            al = 1.0;
            be = -ab;
        }
        affine_scan_inclusive(al, be, lane);
    };
    scan();
    bool dirty = true;
    double mA[8], mB[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        mA[w] = 1.0;
        mB[w] = 0.0;
    }
    for (int gr = 0; gr < 3 * kWave; ++gr) {
        if (dirty && lane == 0) {
            sh_map[wave * 2] = last >= 0 ? readlane_d(al, last >= 0 ? last : 0) : 1.0;
            sh_map[wave * 2 + 1] = last >= 0 ? readlane_d(be, last >= 0 ? last : 0) : 0.0;
        }
        if (wave == 0 && lane == 0) {
            if (gr == 0) sh_bad[0] = sh_bad[1] = sh_bad[2] = 0;
            else sh_bad[(gr + 1) % 3] = 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // all eight maps in flight at once (a dependent LDS round trip per wave in front would
        // cost more than the scan); maps of waves without columns are never applied
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            mA[w] = sh_map[w * 2];
            mB[w] = sh_map[w * 2 + 1];
        }
        double c_in = c0;
#pragma unroll
        for (int w = 0; w < 7; ++w)
            if (w < wave) c_in = mA[w] * c_in + mB[w];
        bool changed = false;
        for (int lr = 0; lr <= kWave; ++lr) {
            const double cb = affine_before(al, be, c_in, lane);
            const double v = cb - ab;
            const bool pos2 = sq ? true : !(v < 0);
            const double u = pos2 ? v : 0.0;
            m = sq ? fma(-tt, u, app) : (app - tt * u);
            const bool nz2 = m > 0;
            const unsigned long long bad =
                __ballot(act && ((pos2 != pos) || (pos2 && (nz2 != nz))));
            pos = pos2;
            nz = nz2;
            if (bad == 0ull) break;
            changed = true;
            scan();
        }
        dirty = changed;
        if (changed && lane == 0) sh_bad[gr % 3] = 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (sh_bad[gr % 3] == 0) break;
    }
    if (!sq && __ballot(act && !pos) != 0ull) count_branch(BR_OMEGATI_CLIP, lane);
    double c_end = c0;  // the maps read in the last round are the final ones (nobody changed)
#pragma unroll
    for (int w = 0; w < 8; ++w)
        if (w < nw) c_end = mA[w] * c_end + mB[w];
    cache[ci] = c_end;
    const double r = sq ? ((act && nz) ? m : 0.0) : (act ? ((m > 0) ? m : 0.0) : 0.0);
    return sg * r;
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
    __shared__ double sh_chain[(kWave + 4) * 4];
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
                               cache, sh_chain);
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

// the same update on a row block held in LDS as (A[i, 1..AS], r_i = yhat_i - y_i)
template <typename T, int M>
__device__ __forceinline__ void pcd_sync_entry_lds(int il, double x, double p_old, double upd,
                                                   double lam, T* lds_a, T* lds_r) {
    constexpr int AS = Kind<M>::AS;
    double yh = (double)lds_r[il];
    if constexpr (M == 0) {
        const double a0 = (double)lds_a[il];
        yh -= lam * a0;
        double a1 = a0 / (1.0 + x * p_old);
        a1 *= 1.0 + x * (p_old - upd);
        yh += lam * a1;
        lds_a[il] = (T)a1;
        lds_r[il] = (T)yh;
    } else {
        double dprev = x;
#pragma unroll
        for (int t = 0; t < AS; ++t) {
            const double a = (double)lds_a[il * AS + t];
            const double dcur = x * (a - p_old * dprev);
            lds_a[il * AS + t] = (T)(a - upd * dprev);
            dprev = dcur;
        }
        lds_r[il] = (T)(yh - lam * upd * dprev);
    }
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
    __shared__ double sh_chain[(kWave + 4) * 4];
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
                                              gamma, eta, cache, sh_chain);
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


}  // namespace spfm
