// spfm_pbcd.hip.h -- pbcd: precompute, norms / cache, gradient, prep, chain, scatter
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
#pragma once
#include "spfm_common.hip.h"
#include "spfm_pcd.hip.h"

namespace spfm {

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
static __global__ __launch_bounds__(kBlock) void pbcd_norms_kernel(int d, int k,
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
// COH = true: the value crosses workgroups INSIDE one launch (prep + chain fused): written
// through and read past the caches with agent-scope (sc1) accesses, as in the persistent
// pass -- fences would flush the whole L2 of the XCD.
template <bool COH>
__device__ __forceinline__ double pb_ld(const double* p) {
    if constexpr (COH) return ld_agent(p);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void pb_st(double* p, double v) {
    if constexpr (COH) st_agent(p, v);
    else *p = v;
}

constexpr int kPbW = 32;  // workgroups per column in the gather / scatter kernels

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
template <int C, bool COH>
__device__ __forceinline__ void pbcd_prep_col(
    int q, int lane, const ColDesc* __restrict__ desc, const double* __restrict__ P /* (d,k) */,
    int k, const double* __restrict__ part, const double* __restrict__ lams, int reg, double mu,
    double beta, double gamma, double eta, double* __restrict__ pin /* [ncols][k] p_j' */,
    double* __restrict__ pold /* [ncols][k] */, double* __restrict__ scal /* [ncols][4] */) {
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
        pb_st<COH>(&scal[4 * q + 0], l2);
        pb_st<COH>(&scal[4 * q + 1], st0);
        scal[4 * q + 2] = f;
        scal[4 * q + 3] = 0.0;
    }
}

// Scalar chain for SquaredL21 / OmegaCS: lanes = columns (64 at a time), wave-uniform
// serial loop; writes the shrink factor f into scal[q][2] and the new block norm into
// norms[j].  Fallback branches ("numerical error": squaredl21.py:48-49,
// omegacs.py:75-76,90-96) recompute from all d norms with the whole wave.
// The wave-uniform regularizer state (cache, dcache) lives in registers: loaded by
// pbcd_chain_load, advanced by pbcd_chain_core (one call per dependent step), written back by
// pbcd_chain_store.  The per-step kernels do all three per launch; the persistent pass keeps
// the state in the control wave's registers for the whole sweep.
__device__ __forceinline__ void pbcd_chain_load(int lane, RegState rs, int top_ncache,
                                                double* cache, double* dcache) {
    // one vector load each, then broadcast (the state is wave-uniform)
    const double cv = (lane < top_ncache) ? rs.cache[lane] : 0.0;
    const double dv = (lane < top_ncache) ? rs.dcache[lane] : 0.0;
#pragma unroll
    for (int t = 0; t < kMaxDegree + 2; ++t) {
        cache[t] = readlane_d(cv, t);
        dcache[t] = readlane_d(dv, t);
    }
}
__device__ __forceinline__ void pbcd_chain_store(int lane, RegState rs, int top_ncache,
                                                 const double* cache, const double* dcache) {
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < kMaxDegree + 2; ++t)
            if (t < top_ncache) {
                rs.cache[t] = cache[t];
                rs.dcache[t] = dcache[t];
            }
    }
}

// Degree 2, one chunk of <= 64 columns (lane = column): the cache is one scalar c (= sum of
// block norms; csum) and column i maps it through c' = (c - n_i) + max(l2_i - t_i (c - n_i), 0),
// t_i = st0 (omegacs) or 2 st0 / (1 + 2 st0) (squaredl21): the same piecewise-affine recurrence
// as pcd's squaredl12, solved by the speculative affine scan.  csum = cache[0] (squaredl21) or
// cache[1] (omegacs), c2acc = cache[2], dc2last = dcache[2] (omegacs).  Returns false --
// nothing written -- if any column would take one of the reference's "numerical error"
// branches: the caller then runs pbcd_chain_serial_chunk, which restates them.
// PRIV = true (persistent pass, where EVERY workgroup runs the chain redundantly): the new block
// norms are returned in l2n_out instead of being stored -- the caller stores them one step
// later, when no workgroup can still be reading this step's old norms (see
// pbcd_chain_serial_chunk).
template <bool COH, bool PRIV = false>
__device__ __forceinline__ bool pbcd_chain_fast2_chunk(int lane, int cnt, bool valid, int q, int j,
                                                       double l2, double st0, double njl, int reg,
                                                       RegState rs, double* __restrict__ scal,
                                                       double& csum, double& c2acc,
                                                       double& dc2last, double* l2n_out = nullptr) {
    const double c0 = csum;
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
    // inclusive prefix sum (DPP moves: __shfl_up is an LDS-crossbar round trip per step, and this
    // chain runs in every workgroup on every step's critical path)
    const double c2pre = dpp_prefix_sum_inclusive(c2term);
    const double c_after = al * c0 + be;  // cache after this column
    const bool trouble = valid && ((dc2 < 0) || (c_after < 0) ||
                                   (reg == REG_OMEGACS && c2acc + c2pre < 0));
    if (__ballot(trouble) == 0ull) {
        const double f = (valid && nz) ? (1.0 - (tt * dc2) / l2) : 0.0;
        if (valid) {
            scal[4 * q + 2] = f;
            if constexpr (!PRIV) rs.norms[j] = l2n;  // = l2 - strength, the value the scan propagated
        }
        if constexpr (PRIV) *l2n_out = l2n;
        const double c_end = readlane_d(c_after, cnt - 1);
        csum = c_end;
        if (reg != REG_SQL21) {
            c2acc += readlane_d(c2pre, cnt - 1);
            dc2last = readlane_d(dc2, cnt - 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    }
    return false;
}

// One chunk of <= 64 columns (lane = column) by the serial loop: the all-subsets recurrence
// (multiplicative) or the degree-M cache recurrences with the reference's "numerical error"
// branches (squaredl21.py:48-49, omegacs.py:75-76,90-96), which recompute from all d norms with
// the whole wave.
// PRIV = true: nothing is stored to rs.norms here (every workgroup of the persistent pass runs
// this redundantly, and a slower workgroup's fallback must still find this step's OLD norms in
// memory); the fallback sums read memory and substitute the values of the step's columns
// already processed from registers; the new norms are returned in l2n_out.
template <int M, bool COH, bool PRIV = false>
__device__ __forceinline__ void pbcd_chain_serial_chunk(int lane, int cnt, bool valid, int q, int j,
                                                        double l2, double st0, double njl, int d,
                                                        int reg, RegState rs, int top_ncache,
                                                        double* __restrict__ scal, double* cache,
                                                        double* dcache, double* l2n_out = nullptr) {
    // block norm of column jj as the sequential sweep sees it at position i of this chunk
    auto norm_at = [&](int jj, int i, double l2n_mine_) __attribute__((always_inline)) -> double {
        double v = rs.norms[jj];
        if constexpr (PRIV) {
            for (int ii = 0; ii < i; ++ii) {
                const int jx = __builtin_amdgcn_readlane(j, ii);
                const double vx = readlane_d(l2n_mine_, ii);
                v = (jj == jx) ? vx : v;
            }
        }
        return v;
    };
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
            if constexpr (!PRIV) rs.norms[j] = l2n_m;
        }
        if constexpr (PRIV) *l2n_out = l2n_m;
        return;
    }
    double f_mine = 0.0, l2n_mine = 0.0;
// rare fallback paths re-read all d norms from memory: first store the norms of the
// columns of this chunk that were already processed (they live in registers)
#define PBCD_FLUSH_NORMS                                            \
{                                                               \
    if constexpr (!PRIV) {                                          \
        if (valid && lane < i) rs.norms[j] = l2n_mine;              \
    }                                                               \
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
                count_branch<PRIV>(BR_SQL21_RESUM, lane);
                PBCD_FLUSH_NORMS
                double a = 0.0;
                for (int jj = lane; jj < d; jj += kWave) a += norm_at(jj, i, l2n_mine);
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
                count_branch<PRIV>(BR_OMEGACS_DCACHE, lane);
                PBCD_FLUSH_NORMS
                double cc[kMaxDegree + 2];
#pragma unroll
                for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                for (int jj = lane; jj < d; jj += kWave) {
                    const double v = (jj == ji) ? 0.0 : norm_at(jj, i, l2n_mine);
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
                count_branch<PRIV>(BR_OMEGACS_CACHE, lane);
                PBCD_FLUSH_NORMS
                double cc[kMaxDegree + 2];
#pragma unroll
                for (int t = 0; t < kMaxDegree + 2; ++t) cc[t] = (t == 0) ? 1.0 : 0.0;
                for (int jj = lane; jj < d; jj += kWave) {
                    double v = norm_at(jj, i, l2n_mine);
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
        if constexpr (!PRIV) rs.norms[j] = l2n_mine;
    }
    if constexpr (PRIV) *l2n_out = l2n_mine;
    // the next chunk (and its fallback paths) read rs.norms of this chunk's columns
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#undef PBCD_FLUSH_NORMS
}

// PRE = true: the column ids and their old block norms were loaded ahead of time by the
// caller (lane = column, one chunk of <= 64 columns); desc is then unused.
template <int M, bool COH, bool PRE = false>
__device__ __forceinline__ void pbcd_chain_core(
    int lane, const ColDesc* __restrict__ desc, int ncols, int d, int reg, RegState rs,
    int top_ncache, double* __restrict__ scal, double* cache, double* dcache, int j_pre = 0,
    double njl_pre = 0.0) {
    for (int base = 0; base < ncols; base += kWave) {
        const int q = base + lane;
        const bool valid = q < ncols;
        const int cnt = min(kWave, ncols - base);
        int j = 0;
        double l2 = 0.0, st0 = 0.0, njl = 0.0;
        if (valid) {
            j = PRE ? j_pre : desc[q].j;
            l2 = pb_ld<COH>(&scal[4 * q + 0]);
            st0 = pb_ld<COH>(&scal[4 * q + 1]);
            njl = PRE ? njl_pre : rs.norms[j];
        }
        if constexpr (M == 2) {
            const double ca = cache[0], cb1 = cache[1];
            double csum = (reg == REG_SQL21) ? ca : cb1, c2acc = cache[2], dc2last = dcache[2];
            if (pbcd_chain_fast2_chunk<COH>(lane, cnt, valid, q, j, l2, st0, njl, reg, rs, scal,
                                            csum, c2acc, dc2last)) {
                if (reg == REG_SQL21) cache[0] = csum;
                else cache[1] = csum;
                cache[2] = c2acc;
                dcache[2] = dc2last;
                continue;
            }
        }
        pbcd_chain_serial_chunk<M, COH>(lane, cnt, valid, q, j, l2, st0, njl, d, reg, rs,
                                        top_ncache, scal, cache, dcache);
    }
}

template <int M, bool COH>
__device__ __forceinline__ void pbcd_chain_body(
    int lane, const ColDesc* __restrict__ desc, int ncols, int d, int reg, RegState rs,
    int top_ncache, double* __restrict__ scal) {
    double cache[kMaxDegree + 2], dcache[kMaxDegree + 2];
    pbcd_chain_load(lane, rs, top_ncache, cache, dcache);
    pbcd_chain_core<M, COH>(lane, desc, ncols, d, reg, rs, top_ncache, scal, cache, dcache);
    pbcd_chain_store(lane, rs, top_ncache, cache, dcache);
}

// stand-alone forms
template <int C>
__global__ __launch_bounds__(kWave) void pbcd_prep_kernel(
    const ColDesc* __restrict__ desc, const double* __restrict__ P, int k,
    const double* __restrict__ part, const double* __restrict__ lams, int reg, double mu,
    double beta, double gamma, double eta, double* __restrict__ pin, double* __restrict__ pold,
    double* __restrict__ scal) {
    pbcd_prep_col<C, false>(blockIdx.x, threadIdx.x, desc, P, k, part, lams, reg, mu, beta, gamma,
                            eta, pin, pold, scal);
}
template <int M>
__global__ __launch_bounds__(kWave) void pbcd_chain_kernel(
    const ColDesc* __restrict__ desc, int ncols, int d, int reg, RegState rs, int top_ncache,
    double* __restrict__ scal) {
    pbcd_chain_body<M, false>(threadIdx.x, desc, ncols, d, reg, rs, top_ncache, scal);
}

// prep + chain in one launch: one wave per column prepares it, writes the two scalars the
// chain needs (block norm, strength) through to memory and takes a ticket; the wave that
// takes the last ticket runs the chain, reading those scalars past the caches.  One kernel
// boundary less per dependent step; the ticket word resets itself.
template <int M, int C>
__global__ __launch_bounds__(kWave) void pbcd_prep_chain_kernel(
    const ColDesc* __restrict__ desc, int ncols, const double* __restrict__ P, int k,
    const double* __restrict__ part, const double* __restrict__ lams, int reg, double mu,
    double beta, double gamma, double eta, double* __restrict__ pin, double* __restrict__ pold,
    double* __restrict__ scal, int d, RegState rs, int top_ncache, int* __restrict__ ticket) {
    pbcd_prep_col<C, true>(blockIdx.x, threadIdx.x, desc, P, k, part, lams, reg, mu, beta, gamma,
                           eta, pin, pold, scal);
    // the write-through scalars of this column are acknowledged before the ticket is taken
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    int last = 0;
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == ncols - 1) ? 1 : 0;
        if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    pbcd_chain_body<M, true>(threadIdx.x, desc, ncols, d, reg, rs, top_ncache, scal);
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


}  // namespace spfm
