// spfm_pcdw.hip.h -- WIDE persistent passes for pcd (degree 2) and cd_linear: dependent steps of
// up to 512 columns (pcdw_kernel).  Part of the gfx950 device code of the sparse-FM proximal CD
// core; see spfm_kernels.hip.h for the execution model and DESIGN.md section 3d.
#pragma once
#include "spfm_common.hip.h"
#include "spfm_pcd.hip.h"
#include "spfm_prb.hip.h"

namespace spfm {

// ----------------------------------------------------------- wide persistent pass (PCDW)
//
// pcd_prb_kernel's all-to-all sweep reads G x 64 x 2 granules per workgroup and step; with the
// colour classes of a big, wide matrix (BASELINE configs[4], 10M x 1M: ~364 columns per class)
// that exchange would be 6-8 x larger, or the class is cut into 64-column steps (6 x the
// dependent steps).  This pass takes a class of up to 512 columns as ONE step:
//   * one THREAD per column slot (512 threads): thread q walks the (few) entries of column q
//     that fall into the workgroup's row block -- no cross-lane reduction at all, fixed order;
//     its (sum dloss dA, sum dA^2) go out as two tagged granules
//   * exchange as in the persistent pbcd pass (spfm_pbprb.hip.h): 16 columns x 2 values = one
//     32-granule vector ("vslot"); the vslot's OWNER workgroup sums the G partial vectors in
//     fixed order (and, with several GPUs, adds the ranks' vectors through the peer-mapped
//     slabs), publishes the totals; every workgroup collects the <= 32 total vectors
//   * every workgroup runs the chain redundantly, its eight waves in parallel on 64 columns each
//     (pcd_chain_waves: branch guess at the step's input cache, per-wave affine scan, the waves'
//     total maps composed in order, re-checked until consistent; round 2 ran pcd_chain_lanes
//     once per 64 columns on one wave); the regularizer cache is carried through the steps in
//     every thread; cd_linear has no chain -- each thread forms its column's update itself
//   * thread q scatter-updates its column's rows.
// Rows (A[i], yhat_i, y_i) stay in global memory, owned by the workgroup (LR = 0), or -- float
// storage, squared loss, block small enough -- live in LDS for the whole pass as (A[i],
// residual) (LR = 1, as in pcd_prb_kernel).  Pipeline for LR = 0: a thread's entries (row, x)
// are loaded two steps ahead, the row state of the next step's entries one step ahead, except
// rows the current step updates (host flag), which are read after the end-of-step barrier.
// LR = 2 (round 4; a block too large for LDS, float storage, squared loss -- BASELINE configs[4]
// on one GPU): the first `lds_rows` rows of the block live in LDS in the residual form, the rest
// in global memory; the pipeline is LR = 0's, an entry's row state comes from wherever its row
// lives (the pass is bound by the CU's outstanding misses: every row in LDS is a miss less).

struct PcdwArgs {
    int G;                  // workgroups
    int nb;                 // steps in the sweep
    const int32_t* bptr;    // [nb+1]
    const int32_t* jsched;  // [d] column ids in visiting order
    const int32_t* wbase;   // [nb+1] offset of a step's slot boundaries inside a row block's table
    const int32_t* wsp;     // [G][tot] slot boundaries into the entry stream
    int tot;                // d + nb
    const int32_t* erow;    // entry rows, sorted by (workgroup, step, slot, row); bit 31: hazard
    double* slabA;          // [2][32][G][32] partial vectors
    double* slabB;          // [2][32][32]    totals
    int rows_per, n_rows;
    int lds_rows;           // LR = 2: rows of a block kept in LDS (its first ones)
    unsigned* abort_flag;
    unsigned spin_max;     // polls of one wait before the pass gives up (default 2^21)
    int n_ranks, rank;
    double* const* slabC;   // [n_ranks] peer-mapped [2][32][n_ranks][32]
    long long* stamps;      // diagnostic (STAMP instantiation): [G][16] cycles per phase, thread 0
};

struct PcdwParams {
    const Ctl* ctl;          // pcd: component of the pass
    size_t a_stride;         // pcd: elements between the components' cache slices
    double* P;               // pcd: (k, d)
    int d, reg, loss;
    const double* cache_in;  // pcd: regularizer cache of the pass
    double mu, beta, gamma, eta, alpha;
    const double* sched0;    // pcd: P[s, order] snapshot; cd_linear: w in visiting order
    const double* sched1;    // cd_linear: col_norm_sq in visiting order
    double* wout;            // cd_linear: w
    double* viol_pos;
};

constexpr int kPcdwThreads = 512;
// entries of a slot kept in registers per thread.  A thread with more entries reads the rest
// through two dependent global loads, in the gradient phase and again in the scatter -- and every
// step waits for its slowest workgroup.  With ~2 entries per thread (Poisson) 3 registers left
// 14 % of the threads on that path: 2M x 200k (rows in LDS) 7.88 us per step with 3, 7.16 with 6,
// 7.02 with 8; 6M x 600k (rows in global memory) 15.6 / 13.8 / 13.45.
constexpr int kPcdwEPT = 8;

// dynamic LDS of the fixed part (bytes); LR adds (rows in LDS) * (KIND == 0 ? 8 : 4)
constexpr size_t kPcdwLdsFixed = sizeof(double) * (2 * 16 * 32 + 1024 + 512 + 512) + 16;

__device__ __forceinline__ bool pcdw_poll_fail(const PcdwArgs& a, unsigned& spins) {
    if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            spins > a.spin_max) {
            __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

// owner vslot of workgroup g in owner round r (-1: none): 32 vslots spread over the workgroups
__device__ __forceinline__ int pcdw_owned_vslot(int G, int g, int r) {
    if (G >= 32) {
        const int stride = G / 32;
        return (g % stride == 0 && g / stride < 32) ? g / stride : -1;
    }
    const int v = g + r * G;
    return v < 32 ? v : -1;
}

// Row record of the wide pcd pass with its rows in global memory: (yhat_i, y_i, A_s[i]) side by
// side, so that an entry's row state is ONE line instead of one of `yy` and one of the
// component's cache slice.  At 10M rows the pass is bound by the CU's outstanding misses (430
// random rows per workgroup and step): half the lines, half the misses.  Packed from yy / A at
// the start of a component pass and unpacked at its end (two streaming kernels, 0.1 ms).
template <typename T>
struct __attribute__((aligned(4 * sizeof(T)))) PcdwRec {
    T yh, y, a, pad;
};

template <typename T>
__global__ void pcdw_pack_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                 const T* __restrict__ yy, const T* __restrict__ A_all,
                                 PcdwRec<T>* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const T* A = A_all + (size_t)ctl->s * a_stride;
        PcdwRec<T> r;
        r.yh = yy[2 * i];
        r.y = yy[2 * i + 1];
        r.a = A[i];
        r.pad = (T)0;
        rec[i] = r;
    }
}

template <typename T>
__global__ void pcdw_unpack_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                   const PcdwRec<T>* __restrict__ rec, T* __restrict__ yy,
                                   T* __restrict__ A_all) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        T* A = A_all + (size_t)ctl->s * a_stride;
        const PcdwRec<T> r = rec[i];
        yy[2 * i] = r.yh;
        A[i] = r.a;
    }
}

// Squared loss, float storage: the record of a row in global memory can be the 8 bytes the LDS
// rows are -- (A_s[i], r_i = yhat_i - y_i): dloss IS the residual -- half the bytes per random
// access, twice the rows per cache line and per megabyte of L2 (round 4, pcdwe_kernel R8).
static __global__ void pcdw_pack8_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                         const float* __restrict__ yy, const float* __restrict__ A_all,
                                         float2* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float* A = A_all + (size_t)ctl->s * a_stride;
        rec[i] = make_float2(A[i], (float)((double)yy[2 * i] - (double)yy[2 * i + 1]));
    }
}
static __global__ void pcdw_unpack8_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                           const float2* __restrict__ rec, float* __restrict__ yy,
                                           float* __restrict__ A_all) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float* A = A_all + (size_t)ctl->s * a_stride;
        const float2 r = rec[i];
        A[i] = r.x;
        yy[2 * i] = (float)((double)r.y + (double)yy[2 * i + 1]);
    }
}

template <typename T>
struct PcdwSet {  // a thread's entries of one step: [e0, e0 + cnt), the first kPcdwEPT loaded
    int e0, cnt;
    int row[kPcdwEPT];  // bit 31: the row was touched by the previous step
    T x[kPcdwEPT];
};

// KIND 0: pcd component pass (degree 2), 1: cd_linear epoch.  LR 0: rows in global memory,
// 1: (A[i], residual) in LDS (float storage, squared loss), 2: the first lds_rows rows of the
// block in LDS like 1, the others in global memory like 0.
template <typename T, int KIND, int LR, bool STAMP = false>
__global__ __launch_bounds__(kPcdwThreads) void pcdw_kernel(PcdwArgs a, PcdwParams pp,
                                                            const T* __restrict__ eval,
                                                            T* __restrict__ A_all,
                                                            T* __restrict__ yy,
                                                            PcdwRec<T>* __restrict__ rec) {
    // KIND 0, rows (also) in global memory: they live in `rec` (packed records)
    constexpr bool PACKED = (KIND == 0 && LR != 1);
    constexpr bool HYB = (LR == 2);
    static_assert(LR == 0 || sizeof(T) == 4, "LDS-resident rows: float storage");
    constexpr int NG = 16, L = 32, EPT = kPcdwEPT;
    using Set = PcdwSet<T>;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double* sh_red = dyn_lds;                // [2][NG][L] owner part sums
    double* sh_tot = sh_red + 2 * NG * L;    // [512][2] totals of the step's columns
    double* sh_delta = sh_tot + 1024;        // [512]: [0..16) the waves' chain maps, then 3 flags
    double* sh_map = sh_delta;
    int* sh_bad = reinterpret_cast<int*>(sh_delta + 16);
    double* sh_pold = sh_delta + 512;        // [512]
    int* sh_ok = reinterpret_cast<int*>(sh_pold + 512);
    const int lds_n = HYB ? a.lds_rows : a.rows_per;  // rows of the block that live in LDS
    T* lds_a = reinterpret_cast<T*>(sh_ok + 4);  // LR: [lds_n] A[i] (pcd)
    T* lds_r = lds_a + (KIND == 0 ? lds_n : 0);  // LR: [lds_n] residual
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lane = tid % L, grp = tid / L, wlane = tid & 63, wave = tid >> 6;
    const int q = tid;  // the column slot this thread works for
    T* __restrict__ A = (KIND == 0) ? A_all + (size_t)pp.ctl->s * pp.a_stride : A_all;
    const double lam = (KIND == 0) ? pp.ctl->lam : 0.0;
    const int s_comp = (KIND == 0) ? pp.ctl->s : 0;
    const int row0 = LR ? g * a.rows_per : 0;
    double cache[3] = {0.0, 0.0, 0.0};
    if constexpr (KIND == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t) cache[t] = pp.cache_in[t];
    }
    if (tid == 0) *sh_ok = 1;
    long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = STAMP ? clock64() : 0;
#define PW_STAMP(kk)                        \
    if constexpr (STAMP) {                  \
        if (tid == 0) {                     \
            const long long tn = clock64(); \
            acc[kk] += tn - tprev;          \
            tprev = tn;                     \
        }                                   \
    }
    if constexpr (LR != 0) {  // row block -> LDS (residual form: dloss = yhat - y)
        const int nr = min(lds_n, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPcdwThreads) {
            if constexpr (HYB && KIND == 0) {  // (the pass's rows were packed into records)
                const PcdwRec<T> r = rec[(size_t)(row0 + il)];
                lds_a[il] = r.a;
                lds_r[il] = (T)((double)r.yh - (double)r.y);
            } else {
                const typename Vec2<T>::type yv = yy2[(size_t)(row0 + il)];
                if constexpr (KIND == 0) lds_a[il] = A[(size_t)(row0 + il)];
                lds_r[il] = (T)((double)yv.x - (double)yv.y);
            }
        }
    }
    // (LR = 2: the prologue below already reads row state, also LDS rows)
    if constexpr (HYB) __syncthreads();
    // LR = 2: does row i of this workgroup's block live in LDS?
    auto in_lds = [&](int i) __attribute__((always_inline)) -> bool {
        return (unsigned)(i - row0) < (unsigned)lds_n;
    };

    auto bounds = [&](int b, int qq, int ncols_b, int& e0, int& e1) __attribute__((always_inline)) {
        e0 = 0;
        e1 = 0;
        if (b < a.nb && qq < ncols_b) {
            const int32_t* p = a.wsp + (size_t)g * a.tot + a.wbase[b] + qq;
            e0 = p[0];
            e1 = p[1];
        }
    };
    auto load_entries = [&](Set& s, int e0, int e1) __attribute__((always_inline)) {
        s.e0 = e0;
        s.cnt = e1 - e0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const bool v = u < s.cnt;
            s.row[u] = v ? a.erow[e0 + u] : 0;
            s.x[u] = v ? eval[e0 + u] : (T)0;
        }
    };
    // row state of a set's register-resident entries (LR = 0): hz = 0 the entries not flagged,
    // 1 the flagged ones
    auto load_rows = [&](const Set& s, T* av, T* yh, T* yt, int hz) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (u < s.cnt && (int)((unsigned)s.row[u] >> 31) == hz) {
                const size_t i = (size_t)(s.row[u] & 0x7fffffff);
                if constexpr (HYB) {
                    if (in_lds((int)i)) {  // (r, 0): dloss = r
                        yh[u] = lds_r[(int)i - row0];
                        yt[u] = (T)0;
                        if constexpr (KIND == 0) av[u] = lds_a[(int)i - row0];
                        continue;
                    }
                }
                if constexpr (PACKED) {
                    const PcdwRec<T> r = rec[i];
                    yh[u] = r.yh;
                    yt[u] = r.y;
                    av[u] = r.a;
                } else {
                    const typename Vec2<T>::type yv = yy2[i];
                    yh[u] = yv.x;
                    yt[u] = yv.y;
                    if constexpr (KIND == 0) av[u] = A[i];
                }
            }
        }
    };

    int c0 = a.bptr[0], c1 = a.bptr[min(1, a.nb)];
    int c2 = a.bptr[min(2, a.nb)], c3 = a.bptr[min(3, a.nb)];
    Set cur, nxt, nn;
    T av[EPT], yh[EPT], yt[EPT], avn[EPT], yhn[EPT], ytn[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) av[u] = yh[u] = yt[u] = avn[u] = yhn[u] = ytn[u] = (T)0;
    int b2e0, b2e1;
    {
        int e0, e1;
        bounds(0, q, c1 - c0, e0, e1);
        load_entries(cur, e0, e1);
        bounds(1, q, c2 - c1, e0, e1);
        load_entries(nxt, e0, e1);
        bounds(2, q, c3 - c2, b2e0, b2e1);
        nn = nxt;
        if constexpr (LR != 1) load_rows(cur, av, yh, yt, 0);
    }
    double s0 = (q < c1 - c0) ? pp.sched0[c0 + q] : 0.0, s0n = 0.0;  // p_old / w of the slot
    double s1 = (KIND == 1 && q < c1 - c0) ? pp.sched1[c0 + q] : 0.0, s1n = 0.0;
    __syncthreads();

    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = a.bptr[min(b + 4, a.nb)];
        const int nv = (ncols + 15) >> 4;                       // vslots of this step
        const int nwv = max(nv, ((c3 - c2) + 15) >> 4);         // written: also the next use's
        const unsigned long long tag = prb_tag(b);
        const int par = b & 1;
        double* slabA = a.slabA + (size_t)par * 32 * a.G * L;
        double* slabB = a.slabB + (size_t)par * 32 * L;

        // Everything loaded in the previous step has landed by now; a real S_WAITCNT tells the
        // compiler so (see pbcd_prb_kernel), else the prefetch block below stalls on its own loads
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        // ---- phase 0 (LR = 0 / 2): rows this step shares with the previous one
        if constexpr (LR != 1) load_rows(cur, av, yh, yt, 1);
        PW_STAMP(0)  // hazard rows
        // ---- phase 1: the thread's column: partial sums over the block's rows (pcd.py:52-59,
        // cd_linear.py:15-18)
        double ag = 0.0, ah = 0.0;
        const int nfast = min(cur.cnt, EPT);
        if (q < ncols) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                if (u < nfast) {
                    double y0, y1, a1 = 0.0;
                    if constexpr (LR == 1) {
                        const int il = (cur.row[u] & 0x7fffffff) - row0;
                        y0 = (double)lds_r[il];
                        y1 = 0.0;
                        if constexpr (KIND == 0) a1 = (double)lds_a[il];
                    } else {
                        y0 = (double)yh[u];
                        y1 = (double)yt[u];
                        a1 = (double)av[u];
                    }
                    const double x = (double)cur.x[u];
                    const double dl = dloss_dev(pp.loss, y0, y1);
                    if constexpr (KIND == 0) {
                        const double dprev = x * (a1 - s0 * x);
                        ag += dl * dprev;
                        ah += dprev * dprev;
                    } else {
                        ag += dl * x;
                    }
                }
            }
            for (int u = EPT; u < cur.cnt; ++u) {  // more entries than registers: from memory
                const int i = a.erow[cur.e0 + u] & 0x7fffffff;
                const double x = (double)eval[cur.e0 + u];
                double y0, y1, a1 = 0.0;
                if (LR == 1 || (HYB && in_lds(i))) {
                    y0 = (double)lds_r[i - row0];
                    y1 = 0.0;
                    if constexpr (KIND == 0) a1 = (double)lds_a[i - row0];
                } else if constexpr (PACKED) {
                    const PcdwRec<T> r = rec[i];
                    y0 = (double)r.yh;
                    y1 = (double)r.y;
                    a1 = (double)r.a;
                } else {
                    const typename Vec2<T>::type yv = yy2[i];
                    y0 = (double)yv.x;
                    y1 = (double)yv.y;
                    if constexpr (KIND == 0) a1 = (double)A[i];
                }
                const double dl = dloss_dev(pp.loss, y0, y1);
                if constexpr (KIND == 0) {
                    const double dprev = x * (a1 - s0 * x);
                    ag += dl * dprev;
                    ah += dprev * dprev;
                } else {
                    ag += dl * x;
                }
            }
        }
        if (q < 16 * nwv) {  // (slots beyond the step publish zeros: tags stay fresh)
            double* dst = slabA + ((size_t)(q >> 4) * a.G + g) * L + (q & 15);
            prb_store_granule(dst, ag, tag);
            prb_store_granule(dst + 16, ah, tag);
        }
        sh_pold[q] = s0;
        PW_STAMP(1)  // sums + publish

        // ---- phase 2: owners reduce their vslot over the workgroups (+ over the GPUs)
        const int n_rounds = (a.G >= 32) ? 1 : (nwv + a.G - 1) / a.G;
        for (int r = 0; r < n_rounds; ++r) {
            const int v = pcdw_owned_vslot(a.G, g, r);
            const bool own = v >= 0 && v < nv;
            double* red = sh_red + (size_t)(r & 1) * NG * L;
            if (own) {
                double tot = 0.0;
                constexpr int GU = sizeof(T) == 4 ? 16 : 8;  // sources polled together per lane (float: all 256 in one round; double storage is short of registers)
                for (int src0 = grp; src0 < a.G; src0 += NG * GU) {
                    unsigned long long t[GU];
                    unsigned spins = 0;
                    bool ok = true;
                    for (;;) {
                        bool all = true;
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            const int src = src0 + u * NG;
                            t[u] = (src < a.G)
                                       ? prb_load_granule(slabA + ((size_t)v * a.G + src) * L + lane)
                                       : tag;
                            all = all && ((t[u] & 3ull) == tag);
                        }
                        if (all) break;
                        if (pcdw_poll_fail(a, spins)) {
                            ok = false;
                            break;
                        }
                    }
                    if (!ok) {
                        *sh_ok = 0;
                        break;
                    }
#pragma unroll
                    for (int u = 0; u < GU; ++u)
                        if (src0 + u * NG < a.G)
                            tot += __longlong_as_double((long long)(t[u] & ~3ull));
                }
                red[grp * L + lane] = tot;
            }
            PW_STAMP(2)  // owner poll
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (v >= nv && v < nwv && v >= 0 && grp == 0) {  // keep the unused vslot's tags fresh
                prb_store_granule(slabB + (size_t)v * L + lane, 0.0, tag);
                if (a.n_ranks > 1) {
                    const size_t off = ((size_t)par * 32 + v) * a.n_ranks * L;
                    for (int rr = 0; rr < a.n_ranks; ++rr)
                        prb_store_granule_sys(a.slabC[rr] + off + (size_t)a.rank * L + lane, 0.0, tag);
                }
            }
            if (own && grp == 0) {
                double tot = red[lane];
#pragma unroll
                for (int w = 1; w < NG; ++w) tot += red[w * L + lane];
                if (a.n_ranks > 1) {
                    const size_t off = ((size_t)par * 32 + v) * a.n_ranks * L;
                    for (int rr = 0; rr < a.n_ranks; ++rr)
                        prb_store_granule_sys(a.slabC[rr] + off + (size_t)a.rank * L + lane, tot, tag);
                    double gt = 0.0;
                    bool ok = true;
                    const double* mine = a.slabC[a.rank];
                    for (int rr = 0; rr < a.n_ranks && ok; ++rr) {
                        unsigned long long t;
                        unsigned spins = 0;
                        for (;;) {
                            t = prb_load_granule_sys(mine + off + (size_t)rr * L + lane);
                            if ((t & 3ull) == tag) break;
                            if (pcdw_poll_fail(a, spins)) {
                                ok = false;
                                break;
                            }
                        }
                        gt += __longlong_as_double((long long)(t & ~3ull));
                    }
                    if (!ok) *sh_ok = 0;
                    tot = gt;
                }
                prb_store_granule(slabB + (size_t)v * L + lane, tot, tag);
            }
        }

        PW_STAMP(3)  // owner barrier + total + publish
        // ---- prefetch (in front of the collect poll, as in the pbcd pass): entries of step
        // b+2, row state of step b+1 that this step does not touch, slot data of step b+1
        // Order matters: the row gathers use the entries loaded a step ago, and the compiler --
        // conservative about loads that are pending across the loop's back edge -- puts a full
        // vmcnt(0) in front of that use; issued behind this step's (cold, streaming) entry loads
        // it would wait for those too.  So: rows first, the entry stream last.
        if constexpr (LR != 1) load_rows(nxt, avn, yhn, ytn, 0);
        s0n = (q < c2 - c1) ? pp.sched0[c1 + q] : 0.0;
        if constexpr (KIND == 1) s1n = (q < c2 - c1) ? pp.sched1[c1 + q] : 0.0;
        int jmine = 0;  // workgroup 0 writes the parameters
        if (g == 0 && q < ncols) jmine = a.jsched[c0 + q];
        int b3e0, b3e1;
        bounds(b + 3, q, c4 - c3, b3e0, b3e1);
        load_entries(nn, b2e0, b2e1);

        PW_STAMP(4)  // prefetch issue
        // ---- phase 3: every workgroup collects the totals of all vslots
        {
            const int total = nv * L;
            constexpr int RU = 32 * L / kPcdwThreads;  // 2
            unsigned long long t[RU];
            unsigned spins = 0;
            bool ok = true;
            for (;;) {
                bool all = true;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int idx = tid + u * kPcdwThreads;
                    t[u] = (idx < total) ? prb_load_granule(slabB + idx) : tag;
                    all = all && ((t[u] & 3ull) == tag);
                }
                if (all) break;
                if (pcdw_poll_fail(a, spins)) {
                    ok = false;
                    break;
                }
            }
            if (!ok) *sh_ok = 0;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int idx = tid + u * kPcdwThreads;
                if (idx < total) {
                    const int v = idx / L, l = idx % L;
                    sh_tot[(size_t)(v * 16 + (l & 15)) * 2 + (l >> 4)] =
                        __longlong_as_double((long long)(t[u] & ~3ull));
                }
            }
        }
        PW_STAMP(5)  // collect poll
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // totals, p_old in LDS
        if (!*sh_ok) break;
        PW_STAMP(6)  // barrier

        // ---- phase 4: the update.  pcd: the chain over the step's columns (pcd.py:61-68 + the
        // regularizer's cache recurrence), every workgroup redundantly; cd_linear: no chain
        // (cd_linear.py:19-24)
        double delta = 0.0;
        if constexpr (KIND == 0) {
            // all eight waves, 64 columns each, in parallel (pcd_chain_waves)
            const bool valid = q < ncols;
            const double t0 = valid ? sh_tot[(size_t)q * 2] : 0.0;
            const double t1 = valid ? sh_tot[(size_t)q * 2 + 1] : 0.0;
            const double res = pcd_chain_waves<2>(pp.reg, wlane, wave, ncols, valid, s0, t0, t1, lam,
                                                  pp.mu, pp.beta, pp.gamma, pp.eta, cache, sh_map,
                                                  sh_bad);
            if (valid) {
                delta = s0 - res;
                if (g == 0) {
                    pp.P[(size_t)s_comp * pp.d + jmine] = s0 - delta;
                    pp.viol_pos[c0 + q] = fabs(delta);
                }
            }
        } else {
            if (q < ncols) {
                double upd = sh_tot[(size_t)q * 2];
                upd += pp.alpha * s0;
                upd /= pp.mu * s1 + pp.alpha;
                delta = upd;
                if (g == 0) {
                    pp.wout[jmine] = s0 - upd;
                    pp.viol_pos[c0 + q] = fabs(upd);
                }
            }
        }

        PW_STAMP(7)  // chain rounds + barrier
        // ---- phase 5: scatter over the thread's entries (pcd.py:124-133, cd_linear.py:28-31)
        if (q < ncols && delta != 0.0) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                if (u < nfast) {
                    const int i = cur.row[u] & 0x7fffffff;
                    const double x = (double)cur.x[u];
                    if constexpr (LR == 1) {
                        const int il = i - row0;
                        if constexpr (KIND == 0) {
                            const double a1 = (double)lds_a[il];
                            const double dprev = x * (a1 - s0 * x);
                            lds_a[il] = (T)(a1 - delta * x);
                            lds_r[il] = (T)((double)lds_r[il] - lam * delta * dprev);
                        } else {
                            lds_r[il] = (T)((double)lds_r[il] - delta * x);
                        }
                    } else if (HYB && in_lds(i)) {  // the row's state is in registers (r, 0, A)
                        const int il = i - row0;
                        if constexpr (KIND == 0) {
                            const double a1 = (double)av[u];
                            const double dprev = x * (a1 - s0 * x);
                            lds_a[il] = (T)(a1 - delta * x);
                            lds_r[il] = (T)((double)yh[u] - lam * delta * dprev);
                        } else {
                            lds_r[il] = (T)((double)yh[u] - delta * x);
                        }
                    } else {
                        if constexpr (PACKED) {
                            const double a1 = (double)av[u];
                            const double dprev = x * (a1 - s0 * x);
                            PcdwRec<T> r;
                            r.yh = (T)((double)yh[u] - lam * delta * dprev);
                            r.y = yt[u];
                            r.a = (T)(a1 - delta * x);
                            r.pad = (T)0;
                            rec[(size_t)i] = r;
                        } else if constexpr (KIND == 0) {
                            const double a1 = (double)av[u];
                            const double dprev = x * (a1 - s0 * x);
                            A[(size_t)i] = (T)(a1 - delta * x);
                            yy[2 * (size_t)i] = (T)((double)yh[u] - lam * delta * dprev);
                        } else {
                            yy[2 * (size_t)i] = (T)((double)yh[u] - delta * x);
                        }
                    }
                }
            }
            for (int u = EPT; u < cur.cnt; ++u) {
                const int i = a.erow[cur.e0 + u] & 0x7fffffff;
                const double x = (double)eval[cur.e0 + u];
                if (LR == 1 || (HYB && in_lds(i))) {
                    const int il = i - row0;
                    if constexpr (KIND == 0) {
                        const double a1 = (double)lds_a[il];
                        const double dprev = x * (a1 - s0 * x);
                        lds_a[il] = (T)(a1 - delta * x);
                        lds_r[il] = (T)((double)lds_r[il] - lam * delta * dprev);
                    } else {
                        lds_r[il] = (T)((double)lds_r[il] - delta * x);
                    }
                } else if constexpr (PACKED) {
                    PcdwRec<T> r = rec[(size_t)i];
                    const double a1 = (double)r.a;
                    const double dprev = x * (a1 - s0 * x);
                    r.a = (T)(a1 - delta * x);
                    r.yh = (T)((double)r.yh - lam * delta * dprev);
                    rec[(size_t)i] = r;
                } else {
                    const double y0 = (double)yy[2 * (size_t)i];
                    if constexpr (KIND == 0) {
                        const double a1 = (double)A[(size_t)i];
                        const double dprev = x * (a1 - s0 * x);
                        A[(size_t)i] = (T)(a1 - delta * x);
                        yy[2 * (size_t)i] = (T)(y0 - lam * delta * dprev);
                    } else {
                        yy[2 * (size_t)i] = (T)(y0 - delta * x);
                    }
                }
            }
        }
        PW_STAMP(8)  // scatter
        // ---- rotate the pipeline
        cur = nxt;
        nxt = nn;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            av[u] = avn[u];
            yh[u] = yhn[u];
            yt[u] = ytn[u];
        }
        b2e0 = b3e0;
        b2e1 = b3e1;
        s0 = s0n;
        s1 = s1n;
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        if constexpr (LR == 1)  // all rows in LDS: only LDS traffic has to land
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            __syncthreads();  // rows move between threads from step to step (stores drained)
        PW_STAMP(9)  // rotate + end barrier
    }
#undef PW_STAMP
    if (STAMP && a.stamps != nullptr && tid == 0) {
#pragma unroll
        for (int t = 0; t < 10; ++t) a.stamps[(size_t)g * 16 + t] = acc[t];
    }
    if constexpr (LR != 0) {  // write the row block back (yhat = r + y)
        __syncthreads();
        const int nr = min(lds_n, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPcdwThreads) {
            const size_t i = (size_t)(row0 + il);
            if constexpr (HYB && KIND == 0) {  // into the records (unpacked behind the pass)
                PcdwRec<T> r = rec[i];
                r.yh = (T)((double)lds_r[il] + (double)r.y);
                r.a = lds_a[il];
                rec[i] = r;
            } else {
                if constexpr (KIND == 0) A[i] = lds_a[il];
                yy[2 * i] = (T)((double)lds_r[il] + (double)yy[2 * i + 1]);
            }
        }
    }
}

// ------------------------------------------------ entry-parallel form (round 4): pcdwe_kernel
//
// pcdw_kernel walks a column's entries with the column's thread: with ~2 entries per thread and
// step (Poisson) that is eight predicated row gathers and sixteen predicated entry-stream loads
// per wave, most of them with a handful of live lanes -- and the in-kernel stamps at config-5
// size show the step bound by that instruction count through the CU's one address path (prefetch
// issue 11.9 k of 33 k cycles; one register slot less: -0.8 k).  Here the memory side of a step
// is ENTRY-parallel: thread t handles entries E0 + t and E0 + 512 + t of the workgroup's
// contiguous entry range of the step (sorted by slot, row): the entry stream is read with full
// coalesced wave loads, every row gather has all its lanes live, the scatter likewise.  The
// column side stays thread-per-slot: an entry's contribution (dloss * dA, dA^2) goes to LDS,
// the slot's thread adds its entries' contributions in order (same order as pcdw_kernel: same
// bits), publishes, runs the chain, leaves its Delta in LDS for the entry threads' scatter.
// Which slot an entry belongs to: the slot threads write their slot id over their entry range
// into an LDS table one step ahead.  Entries beyond kPcdweCap of a (workgroup, step) are walked
// by their slot's thread as in pcdw_kernel.  Rows: global memory (LR = 0) or the block's first
// rows in LDS (LR = 2), as pcdw_kernel.  Two LDS barriers more per step than pcdw_kernel.
constexpr int kPcdweEPR = 2;                          // register entries per thread
constexpr int kPcdweCap = kPcdweEPR * kPcdwThreads;   // entries of a (workgroup, step) handled entry-parallel
constexpr size_t kPcdweLdsFixed =
    kPcdwLdsFixed + sizeof(double) * (2 * kPcdweCap + 512) + sizeof(unsigned short) * 2 * kPcdweCap + 64;

template <typename T>
struct PcdweSet {  // a thread's share of a (workgroup, step)'s entries: E0 + tid + 512 r
    int E0, cnt;             // the workgroup's entry range of the step
    int row[kPcdweEPR];      // bit 31: the row was touched by the previous step
    T x[kPcdweEPR];
};

template <typename T, int KIND, int LR, bool STAMP = false, bool R8 = false>
__global__ __launch_bounds__(kPcdwThreads) void pcdwe_kernel(PcdwArgs a, PcdwParams pp,
                                                             const T* __restrict__ eval,
                                                             T* __restrict__ A_all,
                                                             T* __restrict__ yy,
                                                             PcdwRec<T>* __restrict__ rec) {
    static_assert(LR >= 0 && LR <= 2, "rows: 0 global memory, 1 LDS, 2 the block's first rows in LDS");
    static_assert(LR == 0 || sizeof(T) == 4, "LDS-resident rows: float storage");
    constexpr bool PACKED = (KIND == 0 && LR != 1);  // rows (also) in global memory: packed records
    static_assert(!R8 || (PACKED && sizeof(T) == 4), "8-byte (A, residual) records: pcd, float");
    float2* __restrict__ rec8 = reinterpret_cast<float2*>(rec);  // R8: (A, residual) per row
    constexpr bool HYB = (LR != 0);                  // rows may live in LDS
    constexpr bool ALL = (LR == 1);                  // ... all of them
    constexpr int NG = 16, L = 32, EPR = kPcdweEPR, CAP = kPcdweCap;
    using Set = PcdweSet<T>;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double* sh_red = dyn_lds;                // [2][NG][L] owner part sums
    double* sh_tot = sh_red + 2 * NG * L;    // [512][2] totals of the step's columns
    double* sh_delta = sh_tot + 1024;        // [512]: [0..16) the waves' chain maps, then 3 flags
    double* sh_map = sh_delta;
    int* sh_bad = reinterpret_cast<int*>(sh_delta + 16);
    double* sh_pold = sh_delta + 512;        // [512] p_old / w of the step's slots
    int* sh_ok = reinterpret_cast<int*>(sh_pold + 512);
    double2* sh_c = reinterpret_cast<double2*>(sh_pold + 512 + 2);  // [CAP] an entry's contribution
    double* sh_dl = reinterpret_cast<double*>(sh_c + CAP);          // [512] the slots' Deltas
    unsigned short* sh_slot = reinterpret_cast<unsigned short*>(sh_dl + 512);  // [2][CAP]
    const int lds_n = ALL ? a.rows_per : (HYB ? a.lds_rows : 0);  // rows of the block that live in LDS
    T* lds_a = reinterpret_cast<T*>(sh_slot + 2 * CAP + 8);  // [lds_n] A[i] (pcd)
    T* lds_r = lds_a + (KIND == 0 ? lds_n : 0);              // [lds_n] residual
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lane = tid % L, grp = tid / L, wlane = tid & 63, wave = tid >> 6;
    const int q = tid;  // the column slot this thread works for
    const double lam = (KIND == 0) ? pp.ctl->lam : 0.0;
    const int s_comp = (KIND == 0) ? pp.ctl->s : 0;
    const int row0 = HYB ? g * a.rows_per : 0;
    double cache[3] = {0.0, 0.0, 0.0};
    if constexpr (KIND == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t) cache[t] = pp.cache_in[t];
    }
    if (tid == 0) *sh_ok = 1;
    long long acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = STAMP ? clock64() : 0;
#define PW_STAMP(kk)                        \
    if constexpr (STAMP) {                  \
        if (tid == 0) {                     \
            const long long tn = clock64(); \
            acc[kk] += tn - tprev;          \
            tprev = tn;                     \
        }                                   \
    }
    if constexpr (HYB) {  // the block's (first) rows -> LDS (residual form: dloss = yhat - y)
        const int nr = min(lds_n, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPcdwThreads) {
            if constexpr (R8) {
                const float2 r = rec8[(size_t)(row0 + il)];
                lds_a[il] = r.x;
                lds_r[il] = r.y;
            } else if constexpr (PACKED) {  // (the pass's rows were packed into records)
                const PcdwRec<T> r = rec[(size_t)(row0 + il)];
                lds_a[il] = r.a;
                lds_r[il] = (T)((double)r.yh - (double)r.y);
            } else {
                const typename Vec2<T>::type yv = yy2[(size_t)(row0 + il)];
                if constexpr (KIND == 0) lds_a[il] = A_all[(size_t)pp.ctl->s * pp.a_stride + (size_t)(row0 + il)];
                lds_r[il] = (T)((double)yv.x - (double)yv.y);
            }
        }
    }
    if constexpr (HYB) __syncthreads();  // the prologue below already reads row state from LDS
    auto in_lds = [&](int i) __attribute__((always_inline)) -> bool {
        return ALL || (HYB && (unsigned)(i - row0) < (unsigned)lds_n);
    };
    // the slot's entry range (slot role) and the workgroup's entry range of a step (uniform)
    // (wb = a.wbase[b], loaded an iteration ahead: the boundary loads must not wait for it)
    auto bounds = [&](int b, int wb, int qq, int ncols_b, int& e0, int& e1) __attribute__((always_inline)) {
        e0 = 0;
        e1 = 0;
        if (b < a.nb && qq < ncols_b) {
            const int32_t* p = a.wsp + (size_t)g * a.tot + wb + qq;
            e0 = p[0];
            e1 = p[1];
        }
    };
    auto wg_range = [&](int b, int wb, int ncols_b, int& E0, int& E1) __attribute__((always_inline)) {
        E0 = 0;
        E1 = 0;
        if (b < a.nb) {
            const int32_t* p = a.wsp + (size_t)g * a.tot + wb;
            E0 = p[0];
            E1 = p[ncols_b];
        }
    };
    auto load_entries = [&](Set& s, int E0, int E1) __attribute__((always_inline)) {
        s.E0 = E0;
        s.cnt = E1 - E0;
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            const int idx = tid + r * kPcdwThreads;
            const bool v = idx < s.cnt;
            // (streamed once per pass: non-temporal, so that the stream does not push the row
            // records out of the Infinity Cache)
            s.row[r] = v ? a.erow[E0 + idx] : 0;
            s.x[r] = v ? eval[E0 + idx] : (T)0;
        }
    };
    // row state of a set's entries: hz = 0 the entries not flagged, 1 the flagged ones
    auto load_rows = [&](const Set& s, T* av, T* yh, T* yt, int hz) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            if (tid + r * kPcdwThreads < s.cnt && (int)((unsigned)s.row[r] >> 31) == hz) {
                const int i = s.row[r] & 0x7fffffff;
                if (in_lds(i)) {  // (r, 0): dloss = r
                    yh[r] = lds_r[i - row0];
                    yt[r] = (T)0;
                    if constexpr (KIND == 0) av[r] = lds_a[i - row0];
                } else if constexpr (R8) {
                    const float2 rr = rec8[(size_t)i];
                    yh[r] = rr.y;
                    yt[r] = (T)0;
                    av[r] = rr.x;
                } else if constexpr (PACKED) {
                    const PcdwRec<T> rr = rec[(size_t)i];
                    yh[r] = rr.yh;
                    yt[r] = rr.y;
                    av[r] = rr.a;
                } else {
                    const typename Vec2<T>::type yv = yy2[(size_t)i];
                    yh[r] = yv.x;
                    yt[r] = yv.y;
                }
            }
        }
    };
    // slot role: this slot's id over its entries of a step -> the table the entry threads read
    auto write_slots = [&](int buf, int E0, int e0, int e1) __attribute__((always_inline)) {
        unsigned short* tab = sh_slot + buf * CAP;
        const int hi = min(e1 - E0, CAP);
        for (int idx = e0 - E0; idx < hi; ++idx) tab[idx] = (unsigned short)q;
    };

    int c0 = a.bptr[0], c1 = a.bptr[min(1, a.nb)];
    int c2 = a.bptr[min(2, a.nb)], c3 = a.bptr[min(3, a.nb)];
    Set cur, nxt, nn;
    T av[EPR], yh[EPR], yt[EPR], avn[EPR], yhn[EPR], ytn[EPR];
    int sl[EPR];     // slots of the current step's entries ...
    double pv[EPR];  // ... and their p_old (read once: sh_pold is rewritten at the end of the step)
#pragma unroll
    for (int r = 0; r < EPR; ++r) {
        av[r] = yh[r] = yt[r] = avn[r] = yhn[r] = ytn[r] = (T)0;
        sl[r] = 0;
        pv[r] = 0.0;
    }
    int se0, se1, sn0, sn1, s2e0, s2e1;  // the slot's entries of steps b, b+1, b+2
    int R2E0, R2E1;                      // the workgroup's entries of step b+2
    {
        int E0, E1;
        const int w0 = a.wbase[0], w1 = a.wbase[min(1, a.nb)], w2 = a.wbase[min(2, a.nb)];
        wg_range(0, w0, c1 - c0, E0, E1);
        load_entries(cur, E0, E1);
        wg_range(1, w1, c2 - c1, E0, E1);
        load_entries(nxt, E0, E1);
        wg_range(2, w2, c3 - c2, R2E0, R2E1);
        nn = nxt;
        bounds(0, w0, q, c1 - c0, se0, se1);
        bounds(1, w1, q, c2 - c1, sn0, sn1);
        bounds(2, w2, q, c3 - c2, s2e0, s2e1);
        write_slots(0, cur.E0, se0, se1);
        load_rows(cur, av, yh, yt, 0);
    }
    double s0 = (q < c1 - c0) ? pp.sched0[c0 + q] : 0.0, s0n = 0.0;  // p_old / w of the slot
    double s1 = (KIND == 1 && q < c1 - c0) ? pp.sched1[c0 + q] : 0.0, s1n = 0.0;
    sh_pold[q] = s0;
    int wb3 = a.wbase[min(3, a.nb)];  // table offset of step b+3
    __syncthreads();

    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = a.bptr[min(b + 4, a.nb)];
        const int wb4 = a.wbase[min(b + 4, a.nb)];  // used from the next iteration on
        const int nv = (ncols + 15) >> 4;                       // vslots of this step
        const int nwv = max(nv, ((c3 - c2) + 15) >> 4);         // written: also the next use's
        const unsigned long long tag = prb_tag(b);
        const int par = b & 1;
        double* slabA = a.slabA + (size_t)par * 32 * a.G * L;
        double* slabB = a.slabB + (size_t)par * 32 * L;
        const int ncap = min(cur.cnt, CAP);  // entries handled entry-parallel

        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see pcdw_kernel
        // the slots of this step's entries (written one step ahead, behind a barrier)
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            const int idx = tid + r * kPcdwThreads;
            sl[r] = (idx < ncap) ? (int)sh_slot[par * CAP + idx] : 0;
        }
        // ---- phase 0: rows this step shares with the previous one
        load_rows(cur, av, yh, yt, 1);
        PW_STAMP(0)  // hazard rows
        // ---- phase 1: the entries' contributions (pcd.py:52-59, cd_linear.py:15-18) -> LDS
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            const int idx = tid + r * kPcdwThreads;
            if (idx < ncap) {
                const double x = (double)cur.x[r];
                const double dl = dloss_dev(pp.loss, (double)yh[r], (double)yt[r]);
                double2 c;
                if constexpr (KIND == 0) {
                    pv[r] = sh_pold[sl[r]];
                    const double dprev = x * ((double)av[r] - pv[r] * x);
                    c.x = dl * dprev;
                    c.y = dprev * dprev;
                } else {
                    c.x = dl * x;
                    c.y = 0.0;
                }
                sh_c[idx] = c;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // contributions in LDS
        // the slot's sums, its entries in order
        double ag = 0.0, ah = 0.0;
        if (q < ncols) {
            const int lo = se0 - cur.E0, hi = min(se1 - cur.E0, CAP);
            for (int idx = lo; idx < hi; ++idx) {
                const double2 c = sh_c[idx];
                ag += c.x;
                ah += c.y;
            }
            for (int e = max(se0, cur.E0 + CAP); e < se1; ++e) {  // beyond the table: from memory
                const int i = a.erow[e] & 0x7fffffff;
                const double x = (double)eval[e];
                double y0, y1, a1 = 0.0;
                if (in_lds(i)) {
                    y0 = (double)lds_r[i - row0];
                    y1 = 0.0;
                    if constexpr (KIND == 0) a1 = (double)lds_a[i - row0];
                } else if constexpr (R8) {
                    const float2 rr = rec8[i];
                    y0 = (double)rr.y;
                    y1 = 0.0;
                    a1 = (double)rr.x;
                } else if constexpr (PACKED) {
                    const PcdwRec<T> rr = rec[i];
                    y0 = (double)rr.yh;
                    y1 = (double)rr.y;
                    a1 = (double)rr.a;
                } else {
                    const typename Vec2<T>::type yv = yy2[i];
                    y0 = (double)yv.x;
                    y1 = (double)yv.y;
                }
                const double dl = dloss_dev(pp.loss, y0, y1);
                if constexpr (KIND == 0) {
                    const double dprev = x * (a1 - s0 * x);
                    ag += dl * dprev;
                    ah += dprev * dprev;
                } else {
                    ag += dl * x;
                }
            }
        }
        if (q < 16 * nwv) {  // (slots beyond the step publish zeros: tags stay fresh)
            double* dst = slabA + ((size_t)(q >> 4) * a.G + g) * L + (q & 15);
            prb_store_granule(dst, ag, tag);
            prb_store_granule(dst + 16, ah, tag);
        }
        PW_STAMP(1)  // sums + publish

        // ---- phase 2: owners reduce their vslot over the workgroups (+ over the GPUs)
        const int n_rounds = (a.G >= 32) ? 1 : (nwv + a.G - 1) / a.G;
        for (int r = 0; r < n_rounds; ++r) {
            const int v = pcdw_owned_vslot(a.G, g, r);
            const bool own = v >= 0 && v < nv;
            double* red = sh_red + (size_t)(r & 1) * NG * L;
            if (own) {
                double tot = 0.0;
                constexpr int GU = sizeof(T) == 4 ? 16 : 8;
                for (int src0 = grp; src0 < a.G; src0 += NG * GU) {
                    unsigned long long t[GU];
                    unsigned spins = 0;
                    bool ok = true;
                    for (;;) {
                        bool all = true;
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            const int src = src0 + u * NG;
                            t[u] = (src < a.G)
                                       ? prb_load_granule(slabA + ((size_t)v * a.G + src) * L + lane)
                                       : tag;
                            all = all && ((t[u] & 3ull) == tag);
                        }
                        if (all) break;
                        if (pcdw_poll_fail(a, spins)) {
                            ok = false;
                            break;
                        }
                    }
                    if (!ok) {
                        *sh_ok = 0;
                        break;
                    }
#pragma unroll
                    for (int u = 0; u < GU; ++u)
                        if (src0 + u * NG < a.G)
                            tot += __longlong_as_double((long long)(t[u] & ~3ull));
                }
                red[grp * L + lane] = tot;
            }
            PW_STAMP(2)  // owner poll
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (v >= nv && v < nwv && v >= 0 && grp == 0) {  // keep the unused vslot's tags fresh
                prb_store_granule(slabB + (size_t)v * L + lane, 0.0, tag);
                if (a.n_ranks > 1) {
                    const size_t off = ((size_t)par * 32 + v) * a.n_ranks * L;
                    for (int rr = 0; rr < a.n_ranks; ++rr)
                        prb_store_granule_sys(a.slabC[rr] + off + (size_t)a.rank * L + lane, 0.0, tag);
                }
            }
            if (own && grp == 0) {
                double tot = red[lane];
#pragma unroll
                for (int w = 1; w < NG; ++w) tot += red[w * L + lane];
                if (a.n_ranks > 1) {
                    const size_t off = ((size_t)par * 32 + v) * a.n_ranks * L;
                    for (int rr = 0; rr < a.n_ranks; ++rr)
                        prb_store_granule_sys(a.slabC[rr] + off + (size_t)a.rank * L + lane, tot, tag);
                    double gt = 0.0;
                    bool ok = true;
                    const double* mine = a.slabC[a.rank];
                    for (int rr = 0; rr < a.n_ranks && ok; ++rr) {
                        unsigned long long t;
                        unsigned spins = 0;
                        for (;;) {
                            t = prb_load_granule_sys(mine + off + (size_t)rr * L + lane);
                            if ((t & 3ull) == tag) break;
                            if (pcdw_poll_fail(a, spins)) {
                                ok = false;
                                break;
                            }
                        }
                        gt += __longlong_as_double((long long)(t & ~3ull));
                    }
                    if (!ok) *sh_ok = 0;
                    tot = gt;
                }
                prb_store_granule(slabB + (size_t)v * L + lane, tot, tag);
            }
        }

        PW_STAMP(3)  // owner barrier + total + publish
        // ---- prefetch: row state of step b+1 that this step does not touch, slot data of step
        // b+1, the slot table of step b+1, entries of step b+2 (rows first, the entry stream
        // last: see pcdw_kernel)
        load_rows(nxt, avn, yhn, ytn, 0);
        PW_STAMP(10)  // (prefetch: row gathers)
        s0n = (q < c2 - c1) ? pp.sched0[c1 + q] : 0.0;
        if constexpr (KIND == 1) s1n = (q < c2 - c1) ? pp.sched1[c1 + q] : 0.0;
        int jmine = 0;  // workgroup 0 writes the parameters
        if (g == 0 && q < ncols) jmine = a.jsched[c0 + q];
        int s3e0, s3e1, R3E0, R3E1;
        bounds(b + 3, wb3, q, c4 - c3, s3e0, s3e1);
        wg_range(b + 3, wb3, c4 - c3, R3E0, R3E1);
        PW_STAMP(11)  // (prefetch: slot data, bounds)
        load_entries(nn, R2E0, R2E1);
        PW_STAMP(12)  // (prefetch: entry stream)
        write_slots(par ^ 1, nxt.E0, sn0, sn1);  // read at the top of step b+1 (end barrier between)

        PW_STAMP(4)  // prefetch issue
        // ---- phase 3: every workgroup collects the totals of all vslots
        {
            const int total = nv * L;
            constexpr int RU = 32 * L / kPcdwThreads;  // 2
            unsigned long long t[RU];
            unsigned spins = 0;
            bool ok = true;
            for (;;) {
                bool all = true;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int idx = tid + u * kPcdwThreads;
                    t[u] = (idx < total) ? prb_load_granule(slabB + idx) : tag;
                    all = all && ((t[u] & 3ull) == tag);
                }
                if (all) break;
                if (pcdw_poll_fail(a, spins)) {
                    ok = false;
                    break;
                }
            }
            if (!ok) *sh_ok = 0;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int idx = tid + u * kPcdwThreads;
                if (idx < total) {
                    const int v = idx / L, l = idx % L;
                    sh_tot[(size_t)(v * 16 + (l & 15)) * 2 + (l >> 4)] =
                        __longlong_as_double((long long)(t[u] & ~3ull));
                }
            }
        }
        PW_STAMP(5)  // collect poll
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // totals in LDS
        if (!*sh_ok) break;
        PW_STAMP(6)  // barrier

        // ---- phase 4: the update (see pcdw_kernel); the slot's Delta -> LDS
        double delta = 0.0;
        if constexpr (KIND == 0) {
            const bool valid = q < ncols;
            const double t0 = valid ? sh_tot[(size_t)q * 2] : 0.0;
            const double t1 = valid ? sh_tot[(size_t)q * 2 + 1] : 0.0;
            const double res = pcd_chain_waves<2>(pp.reg, wlane, wave, ncols, valid, s0, t0, t1, lam,
                                                  pp.mu, pp.beta, pp.gamma, pp.eta, cache, sh_map,
                                                  sh_bad);
            if (valid) {
                delta = s0 - res;
                if (g == 0) {
                    pp.P[(size_t)s_comp * pp.d + jmine] = s0 - delta;
                    pp.viol_pos[c0 + q] = fabs(delta);
                }
            }
        } else {
            if (q < ncols) {
                double upd = sh_tot[(size_t)q * 2];
                upd += pp.alpha * s0;
                upd /= pp.mu * s1 + pp.alpha;
                delta = upd;
                if (g == 0) {
                    pp.wout[jmine] = s0 - upd;
                    pp.viol_pos[c0 + q] = fabs(upd);
                }
            }
        }
        sh_dl[q] = delta;
        PW_STAMP(7)  // chain rounds + barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // Deltas in LDS
        // ---- phase 5: scatter, entry-parallel (pcd.py:124-133, cd_linear.py:28-31)
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            const int idx = tid + r * kPcdwThreads;
            if (idx < ncap) {
                const double dlt = sh_dl[sl[r]];
                if (dlt != 0.0) {
                    const int i = cur.row[r] & 0x7fffffff;
                    const double x = (double)cur.x[r];
                    if (in_lds(i)) {  // the row's state is in registers as (r, 0, A)
                        const int il = i - row0;
                        if constexpr (KIND == 0) {
                            const double a1 = (double)av[r];
                            const double dprev = x * (a1 - pv[r] * x);
                            lds_a[il] = (T)(a1 - dlt * x);
                            lds_r[il] = (T)((double)yh[r] - lam * dlt * dprev);
                        } else {
                            lds_r[il] = (T)((double)yh[r] - dlt * x);
                        }
                    } else if constexpr (R8) {
                        const double a1 = (double)av[r];
                        const double dprev = x * (a1 - pv[r] * x);
                        rec8[(size_t)i] = make_float2((float)(a1 - dlt * x),
                                                      (float)((double)yh[r] - lam * dlt * dprev));
                    } else if constexpr (PACKED) {
                        const double a1 = (double)av[r];
                        const double dprev = x * (a1 - pv[r] * x);
                        PcdwRec<T> rr;
                        rr.yh = (T)((double)yh[r] - lam * dlt * dprev);
                        rr.y = yt[r];
                        rr.a = (T)(a1 - dlt * x);
                        rr.pad = (T)0;
                        rec[(size_t)i] = rr;
                    } else {
                        yy[2 * (size_t)i] = (T)((double)yh[r] - dlt * x);
                    }
                }
            }
        }
        if (q < ncols && delta != 0.0) {  // beyond the table: the slot's thread, from memory
            for (int e = max(se0, cur.E0 + CAP); e < se1; ++e) {
                const int i = a.erow[e] & 0x7fffffff;
                const double x = (double)eval[e];
                if (in_lds(i)) {
                    const int il = i - row0;
                    if constexpr (KIND == 0) {
                        const double a1 = (double)lds_a[il];
                        const double dprev = x * (a1 - s0 * x);
                        lds_a[il] = (T)(a1 - delta * x);
                        lds_r[il] = (T)((double)lds_r[il] - lam * delta * dprev);
                    } else {
                        lds_r[il] = (T)((double)lds_r[il] - delta * x);
                    }
                } else if constexpr (R8) {
                    const float2 rr = rec8[(size_t)i];
                    const double a1 = (double)rr.x;
                    const double dprev = x * (a1 - s0 * x);
                    rec8[(size_t)i] = make_float2((float)(a1 - delta * x),
                                                  (float)((double)rr.y - lam * delta * dprev));
                } else if constexpr (PACKED) {
                    PcdwRec<T> rr = rec[(size_t)i];
                    const double a1 = (double)rr.a;
                    const double dprev = x * (a1 - s0 * x);
                    rr.a = (T)(a1 - delta * x);
                    rr.yh = (T)((double)rr.yh - lam * delta * dprev);
                    rec[(size_t)i] = rr;
                } else {
                    const double y0 = (double)yy[2 * (size_t)i];
                    yy[2 * (size_t)i] = (T)(y0 - delta * x);
                }
            }
        }
        PW_STAMP(8)  // scatter
        // ---- rotate the pipeline
        cur = nxt;
        nxt = nn;
#pragma unroll
        for (int r = 0; r < EPR; ++r) {
            av[r] = avn[r];
            yh[r] = yhn[r];
            yt[r] = ytn[r];
        }
        se0 = sn0;
        se1 = sn1;
        sn0 = s2e0;
        sn1 = s2e1;
        s2e0 = s3e0;
        s2e1 = s3e1;
        R2E0 = R3E0;
        R2E1 = R3E1;
        s0 = s0n;
        s1 = s1n;
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        wb3 = wb4;
        sh_pold[q] = s0;  // (this step's readers are behind the Delta barrier)
        if constexpr (ALL)  // all rows in LDS: only LDS traffic has to land
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            __syncthreads();  // rows move between threads from step to step (stores drained)
        PW_STAMP(9)  // rotate + end barrier
    }
#undef PW_STAMP
    if (STAMP && a.stamps != nullptr && tid == 0) {
#pragma unroll
        for (int t = 0; t < 14; ++t) a.stamps[(size_t)g * 16 + t] = acc[t];
    }
    if constexpr (HYB) {  // write the LDS rows back (yhat = r + y)
        __syncthreads();
        const int nr = min(lds_n, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPcdwThreads) {
            const size_t i = (size_t)(row0 + il);
            if constexpr (R8) {
                rec8[i] = make_float2(lds_a[il], lds_r[il]);
            } else if constexpr (PACKED) {  // into the records (unpacked behind the pass)
                PcdwRec<T> r = rec[i];
                r.yh = (T)((double)lds_r[il] + (double)r.y);
                r.a = lds_a[il];
                rec[i] = r;
            } else {
                if constexpr (KIND == 0) A_all[(size_t)pp.ctl->s * pp.a_stride + i] = lds_a[il];
                yy[2 * i] = (T)((double)lds_r[il] + (double)yy[2 * i + 1]);
            }
        }
    }
}

// erow/eval = cidx/cval gathered through the host-built entry permutation; bit 31 of erow =
// the row was touched by the previous step
template <typename T>
__global__ void pcdw_gather_kernel(int64_t nnz, const int32_t* __restrict__ src,
                                   const uint8_t* __restrict__ hz, const int32_t* __restrict__ cidx,
                                   const T* __restrict__ cval, int32_t* __restrict__ erow,
                                   T* __restrict__ eval) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) {
        const int32_t qq = src[e];
        erow[e] = cidx[qq] | (hz[e] ? (int32_t)0x80000000 : 0);
        eval[e] = cval[qq];
    }
}


}  // namespace spfm
