// spfm_pbprb.hip.h -- persistent row-block pass for pbcd (pbcd_prb_kernel)
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3c.
#pragma once
#include "spfm_common.hip.h"
#include "spfm_pcd.hip.h"
#include "spfm_prb.hip.h"
#include "spfm_pbcd.hip.h"

#ifndef PBPRB_SLOW_ATTR
#define PBPRB_SLOW_ATTR __forceinline__
#endif

namespace spfm {

// ---------------------------------------------------- persistent pbcd pass (PBPRB)
//
// pbcd.pbcd_epoch (optimizer/pbcd.py:82-148) as ONE launch.  Workgroup g owns the rows
// [g*rows_per, (g+1)*rows_per): their state stays in global memory (no LDS residency) and is read
// and written by the owner only.
//
// Row records (round 3).  A row's state is ONE packed record of AS slices of L elements (L = the
// lanes of a slot group, k <= L - 2): slice t holds A[i, t+1, 0..k) and zero padding; slice 0
// carries yhat_i at lane L-2 and y_i at lane L-1.  float storage, k <= 30, degree 2: exactly
// one 128-byte line per row -- the gather of a row (cache values AND prediction AND target) is
// one line instead of a 120-byte run straddling two lines plus an 8-byte (yhat, y) word
// elsewhere, and the scatter rewrites the whole line with one store instruction per group
// (every lane its own element: the cache values, the padding, lane L-2 the updated yhat, lane
// L-1 the target) instead of partial-line write-through stores of A and a 4-byte store of
// yhat.  Packed by pbprb_pack_kernel (which is pbcd._precompute_A_all_degree, pbcd.py:18-33,
// writing this layout) at the start of the epoch; pbprb_unpack_kernel returns yhat to `yy`.
//
// Exchange.  A step's payload is k + 1 sums per column (pbcd.py:60-67: grad[s] and
// sum_s inv_step_sizes[s]), too much for the flat all-to-all of the pcd pass, so it is a
// REDUCE-SCATTER followed by an ALL-GATHER, both with tagged granules (spfm_prb.hip.h):
//   1. every workgroup publishes, per slot, L granules (lanes = components; lane L-2 = its
//      rows' sum of dA^2) into slabA[parity][slot][workgroup][L]                (sc1 stores)
//   2. the slot's OWNER workgroup sums the G partial vectors in fixed order, takes
//      pbcd._update's step (pbcd.py:68-78) with the cache-independent part of prox_bcd and
//      publishes p_j' (lanes), ||p_j'|| (lane L-2), eta*gamma/inv (lane L-1) into
//      slabB[parity][slot][L]
//   3. every workgroup reads the vectors of all slots (ncols * L granules); its control wave
//      runs the scalar cache recurrence of SquaredL21 / OmegaCS (pbcd_chain_core; L1 / L21
//      need none) redundantly; the workgroup scatter-updates its rows (pbcd.py:135-144).
// A slab word is always rewritten at the use of its buffer that precedes a read (slots of this
// step AND of the step after next), so a reader can never meet a stale word with its tag.
// Multi-GPU (n_ranks > 1): between 2's sum and step the owner writes its GPU's vector into
// slabC[slot][rank] of EVERY GPU (peer-mapped stores over xGMI) and adds the n_ranks vectors
// of its own slabC in rank order: every GPU's owner of a slot forms the identical global sum
// and the identical update; nothing else crosses GPUs.
//
// Threads.  512 = NG groups of L lanes (L = 32 for k <= 30, 64 for k <= 62; lane =
// component).  Which group handles which slot is the host's choice per (workgroup, step)
// (`gtab`, round 4: columns dealt to the groups by their entry counts, so that no group has
// many more entries than the others -- every step waits for the busiest group of the busiest
// workgroup: 14.9 entries on BASELINE config 4 with the fixed map "slot q -> group q % NG", 6.5
// with the balanced one, against a mean of 4.4); the host sorts a (workgroup, step)'s entries
// by (group, slot position in the group, row), so a group's entries are one contiguous run
// (`gsp`).
//
// Software pipeline (everything a step needs is in registers when it starts):
//   entries    lane-parallel (lane u <-> entry u of the group: row, x, slot, yhat_i, y_i),
//              loaded two steps ahead; broadcast to the lanes through LDS
//   row state  A[i, :, lane] of the group's first ER entries in LDS (double-buffered by step
//              parity), fetched one step ahead right after the step's last poll by LDS-DMA
//              (global_load_lds_dword: no staging registers) -- except rows the current step
//              itself updates (host flag 0x80), fetched after the end-of-step barrier
//   slot data  column ids two steps, P[j, :] / old block norms one step ahead
// Entries beyond ER of a group take a slow path with loads at the point of use.
// A group's entries are sorted by slot, so phases 1 and 5 run one static loop over the slots
// (the slot's P row, Delta, lam*Delta are plain registers) with a dynamic loop over the slot's
// entries inside; the per-row prediction decrement sum_s lam_s Delta_s dA_s is one DPP
// all-reduce over the group's lanes, applied by the lane that holds the entry's yhat.  The
// per-slot scalar sums (sum dA^2, did-the-block-move, ||Delta||_1) use a transposing butterfly.

struct PbPrbArgs {
    int G;                 // row workgroups (each owns a block of rows)
    int nb;                // steps in the sweep
    const int32_t* bptr;   // [nb+1]
    const int32_t* jsched; // [d] column ids in visiting order
    const int32_t* gsp;    // [G][nb][NG+1] group boundaries into the entry stream
    const int32_t* erow;   // entry rows, sorted by (workgroup, step, group, slot, row)
    const uint8_t* emeta;  // slot position t inside the group | 0x80 (row touched by previous step)
    const uint8_t* gtab;   // [G][nb][64] slot of (group, t) = gtab[.. + group * QM + t], 0xFF: none --
                           // the host's balanced map of a (workgroup, step)'s columns to groups
    double* slabA;         // [2][64][G][L] partial vectors
    double* slabB;         // [2][64][L]    published block updates
    int rows_per, n_rows;
    unsigned* abort_flag;
    unsigned spin_max;     // polls of one wait before the pass gives up (default 2^21)
    int n_ranks, rank;     // multi-GPU: ranks sharing the sweep
    double* const* slabC;  // [n_ranks] slabC[r] = GPU r's [2][64][n_ranks][L] (peer-mapped)
    // relaxed runs (CR instantiation; spfm_schedule.cpp schedule_relax): the conflict rows of a
    // step -- rows shared by two of its columns -- are not in the entry stream; their owners
    // publish the row records, every workgroup replays their two updates in order
    const int32_t* cf_ptr;  // [nb+1] conflict rows of a step
    const void* cf;         // PrbConf<T>[]: row, slots of the two columns, their two x values
    const int16_t* clist;   // [d][8] per column position: conflict index | role << 8, -1 none
    double* slabR;          // [2][64][L] tagged granules: the conflict rows' packed records
    long long* stamps;     // [G][16] diagnostic phase timers or nullptr
    int dbg;               // diagnostic switches (bit 0: stage rows through registers, no LDS-DMA)
    unsigned* dbg_out;     // [16] diagnostic counters
};

constexpr int kPbPrbThreads = 512;

// entries of a slot group whose rows are staged in LDS (64 KB for the two row buffers)
template <typename T, int M>
constexpr int pbprb_er() {
    constexpr int ER0 = (int)(16 * 4 / sizeof(T)) / Kind<M>::AS;
    return ER0 >= 8 ? (ER0 / 8) * 8 : (ER0 >= 4 ? 4 : (ER0 >= 2 ? 2 : 1));
}

// `site` / `step`: which wait gave up first, and where in the sweep (diagnostic: option
// "pbprb_dbg" bit 3 zeroes the counters, spfm_debug_prb_stamps returns them)
__device__ __forceinline__ bool pbprb_poll_fail(const PbPrbArgs& a, unsigned& spins, int site = 0,
                                                int step = 0) {
    if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            spins > a.spin_max) {
            if ((a.dbg & 8) && spins > a.spin_max) {
                atomicAdd(&a.dbg_out[4 + site], 1u);
                atomicMin(&a.dbg_out[3], (unsigned)step);
                atomicMax(&a.dbg_out[8 + site], (unsigned)blockIdx.x + 1u);
            }
            __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

// owner slot of workgroup g in owner round r (-1: none).  G >= 64: one round, the 64 slots
// spread evenly over the workgroups; G < 64: slot = g + r * G.
__device__ __forceinline__ int pbprb_owned_slot(int G, int g, int r) {
    if (G >= 64) {
        const int stride = G / 64;
        return (g % stride == 0 && g / stride < 64) ? g / stride : -1;
    }
    const int q = g + r * G;
    return q < 64 ? q : -1;
}

// lane ^ XM inside 32-lane halves, no address register (ds_swizzle bit-mask mode)
template <int XM>
__device__ __forceinline__ double pb_swz_xor(double v) {
    static_assert(XM >= 1 && XM < 32, "swizzle works inside 32 lanes");
    int lo = __double2loint(v), hi = __double2hiint(v);
    constexpr int pat = (XM << 10) | 0x1F;
    lo = __builtin_amdgcn_ds_swizzle(lo, pat);
    hi = __builtin_amdgcn_ds_swizzle(hi, pat);
    return __hiloint2double(hi, lo);
}

// v(lane) + v(lane ^ 16) in every lane: gfx950's v_permlane16_swap exchanges the odd rows (of
// 16 lanes) of one register with the even rows of another -- fed two copies of v it leaves
// (row0,row0,row2,row2) and (row1,row1,row3,row3), whose sum is the pair sum in all four rows.
// Two VALU instructions per dword instead of a ds_swizzle trip through the LDS pipe.
__device__ __forceinline__ double pb_pairsum16(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h2[0], (int)l2[0]) + __hiloint2double((int)h2[1], (int)l2[1]);
}

// Sum over the L lanes of a group, result in every lane: quad_perm / row_ror moves inside the
// rows of 16 lanes (DPP: one issue slot each, no LDS crossbar round trip), a row swap across
// the two rows of a 32-lane half, one cross-half shuffle for L = 64.  Fixed order.
template <int L>
__device__ __forceinline__ double pb_group_allsum(double v) {
    v += dpp_move_d<0xB1, 0xf>(0.0, v);   // quad_perm [1,0,3,2]
    v += dpp_move_d<0x4E, 0xf>(0.0, v);   // quad_perm [2,3,0,1]
    v += dpp_move_d<0x124, 0xf>(0.0, v);  // row_ror:4
    v += dpp_move_d<0x128, 0xf>(0.0, v);  // row_ror:8
    v = pb_pairsum16(v);
    if constexpr (L == 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// value of lane `src` of the caller's group in every lane of the group: v_readlane (scalar
// path, a few cycles) instead of a ds_bpermute round trip; `src` must be wave-uniform
template <int L>
__device__ __forceinline__ double pb_bcast(double v, int src, int grp) {
    if constexpr (L == 64) {
        return readlane_d(v, src);
    } else {
        const double lo = readlane_d(v, src), hi = readlane_d(v, 32 + src);
        return (grp & 1) ? hi : lo;
    }
}

// Transposing butterfly: v[0..N) per lane -> lane l ends with the sum over the group's lanes
// of v[l & (N-1)].  Stage MASK halves the number of live values: the lane whose bit is set
// keeps the odd entries and hands the even ones to its partner.  Fixed order.
template <int N, int MASK>
__device__ __forceinline__ void pb_mr_stage(double* v, int lane) {
    if constexpr (N > 1) {
        const bool hi = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const double keep = hi ? v[2 * i + 1] : v[2 * i];
            const double send = hi ? v[2 * i] : v[2 * i + 1];
            if constexpr (MASK == 1)
                v[i] = keep + dpp_move_d<0xB1, 0xf>(0.0, send);  // quad_perm [1,0,3,2]
            else if constexpr (MASK == 2)
                v[i] = keep + dpp_move_d<0x4E, 0xf>(0.0, send);  // quad_perm [2,3,0,1]
            else
                v[i] = keep + pb_swz_xor<MASK>(send);
        }
        pb_mr_stage<N / 2, MASK * 2>(v, lane);
    }
}
template <int N, int L>
__device__ __forceinline__ double pb_multi_reduce(double* v, int lane) {
    static_assert(N <= 16 && N >= 1 && (N & (N - 1)) == 0, "N: power of two <= 16");
    pb_mr_stage<N, 1>(v, lane);
    double r = v[0];
    if constexpr (N <= 1) r += dpp_move_d<0xB1, 0xf>(0.0, r);   // lane ^ 1
    if constexpr (N <= 2) r += dpp_move_d<0x4E, 0xf>(0.0, r);   // lane ^ 2
    if constexpr (N <= 4) r += dpp_move_d<0x124, 0xf>(0.0, r);  // row_ror:4 (the 4-orbit ...
    if constexpr (N <= 8) r += dpp_move_d<0x128, 0xf>(0.0, r);  // row_ror:8  ... of lane & 3)
    r = pb_pairsum16(r);
    if constexpr (L == 64) r += __shfl_xor(r, 32, kWave);
    return r;
}

// v[qi] without a dynamically indexed array (which would live in scratch): a select tree on
// the bits of qi, all indices compile-time constants
template <int QM>
__device__ __forceinline__ double pb_sel(const double (&v)[QM], int qi) {
    static_assert(QM == 4 || QM == 8, "4 or 8 slots per group");
    // values first: `c ? v[1] : v[0]` on lvalues selects the ADDRESS and loads once, which
    // pins the array in scratch
    const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    const bool b0 = (qi & 1) != 0, b1 = (qi & 2) != 0;
    const double lo01 = b0 ? v1 : v0, lo23 = b0 ? v3 : v2;
    const double lo = b1 ? lo23 : lo01;
    if constexpr (QM == 4) {
        return lo;
    } else {
        const double v4 = v[4], v5 = v[5], v6 = v[6], v7 = v[7];
        const double hi45 = b0 ? v5 : v4, hi67 = b0 ? v7 : v6;
        const double hi = b1 ? hi67 : hi45;
        return (qi & 4) ? hi : lo;
    }
}
// v[qi] += x, same constraint
template <int QM, int I = 0>
__device__ __forceinline__ void pb_acc(double (&v)[QM], int qi, double x) {
    if constexpr (I < QM) {
        v[I] += (qi == I) ? x : 0.0;
        pb_acc<QM, I + 1>(v, qi, x);
    }
}

// One step of the SquaredL21 / OmegaCS cache recurrence for the persistent pass (<= 64
// columns, lane = column; column ids and old norms preloaded).  The regularizer state lives in
// LDS (`state`: cache[kMaxDegree+2], dcache[kMaxDegree+2]).  Degree 2 takes the affine-scan
// path on three scalars; everything else -- and a degree-2 step that would hit one of the
// reference's "numerical error" branches -- goes through the serial loop, kept out of line so
// that its register needs (product trees over all d norms) do not weigh on every step.
template <int M>
__device__ PBPRB_SLOW_ATTR void pbprb_chain_slow(int lane, int ncols, int j, double njl,
                                                           double l2, double st0, int d, int reg,
                                                           RegState rs, int top_ncache,
                                                           double* scal, double* state,
                                                           double* l2n_out) {
    double cache[kMaxDegree + 2], dcache[kMaxDegree + 2];
#pragma unroll
    for (int t = 0; t < kMaxDegree + 2; ++t) {
        cache[t] = state[t];
        dcache[t] = state[kMaxDegree + 2 + t];
    }
    pbcd_chain_serial_chunk<M, false, true>(lane, ncols, lane < ncols, lane, j, l2, st0, njl, d,
                                            reg, rs, top_ncache, scal, cache, dcache, l2n_out);
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < kMaxDegree + 2; ++t) {
            state[t] = cache[t];
            state[kMaxDegree + 2 + t] = dcache[t];
        }
    }
}
template <int M>
__device__ __forceinline__ void pbprb_chain_step(int lane, int ncols, int j, double njl, int d,
                                                 int reg, RegState rs, int top_ncache, double* scal,
                                                 double* state, double* l2n_out) {
    const bool valid = lane < ncols;
    const double l2 = valid ? scal[4 * lane + 0] : 0.0, st0 = valid ? scal[4 * lane + 1] : 0.0;
    if constexpr (M == 2) {
        const int ci = (reg == REG_SQL21) ? 0 : 1;
        double csum = state[ci], c2acc = state[2], dc2last = state[kMaxDegree + 2 + 2];
        if (pbcd_chain_fast2_chunk<false, true>(lane, ncols, valid, lane, j, l2, st0, njl, reg, rs,
                                                scal, csum, c2acc, dc2last, l2n_out)) {
            if (lane == 0) {
                state[ci] = csum;
                state[2] = c2acc;
                state[kMaxDegree + 2 + 2] = dc2last;
            }
            return;
        }
    }
    pbprb_chain_slow<M>(lane, ncols, j, njl, l2, st0, d, reg, rs, top_ncache, scal, state, l2n_out);
}

// pbcd._precompute_A_all_degree (pbcd.py:18-33; all-subsets: pbcd_all.py:9-20) writing the
// packed row records of the persistent pass (see the header): thread per (row, lane of the
// record); lanes < k run the row's DP for their component, lane L-2 / L-1 copy (yhat, y).
template <typename T, int M, int L>
__global__ __launch_bounds__(kBlock) void pbprb_pack_kernel(
    int64_t n, int k, const int64_t* __restrict__ rptr, const int32_t* __restrict__ ridx,
    const T* __restrict__ rval, const double* __restrict__ P /* (d,k) */,
    const T* __restrict__ yy, T* __restrict__ R) {
    constexpr int AS = Kind<M>::AS;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (tid >= n * L) return;
    const int64_t i = tid / L;
    const int l = (int)(tid - i * L);
    T* ri = R + (size_t)i * AS * L;
    if (l >= k) {
#pragma unroll
        for (int t = 0; t < AS; ++t) ri[(size_t)t * L + l] = (T)0;
        if (l == L - 2) ri[l] = yy[2 * (size_t)i];
        if (l == L - 1) ri[l] = yy[2 * (size_t)i + 1];
        return;
    }
    if constexpr (M == 0) {
        double a = 1.0;
        for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii)
            a *= 1.0 + P[(size_t)ridx[ii] * k + l] * (double)rval[ii];
        ri[l] = (T)a;
    } else {
        double a[M];
        a[0] = 1.0;
#pragma unroll
        for (int t = 1; t < M; ++t) a[t] = 0.0;
        for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii) {
            const double p = P[(size_t)ridx[ii] * k + l];
            const double x = (double)rval[ii];
#pragma unroll
            for (int t = M - 1; t >= 1; --t) a[t] += a[t - 1] * p * x;
        }
#pragma unroll
        for (int t = 1; t < M; ++t) ri[(size_t)(t - 1) * L + l] = (T)a[t];
    }
}

// the epoch's predictions back into `yy` (the other passes' home of (yhat, y))
template <typename T, int AS, int L>
__global__ void pbprb_unpack_kernel(int64_t n, const T* __restrict__ R, T* __restrict__ yy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) yy[2 * (size_t)i] = R[(size_t)i * AS * L + (L - 2)];
}

// (Round 4, measured and not kept -- docs/HISTORY.md: an "early phase" that formed the next
// step's sums for the unshared rows during the collect wait, 35.0 vs 27.6 ms per epoch: the
// slot owners have no wait to fill, and they set the pace; the new records of rows shared by
// consecutive steps forwarded through LDS instead of a global round trip: phase 0 0.77 -> 0.48
// us, but the kernel is bound by its instruction count and the table look-ups cost more in the
// scatter and the prefetch than the round trip; a scatter that took four entries at a time with
// one butterfly for their prediction decrements; round 3's dedicated owner workgroups.)
// pbcd._update (pbcd.py:68-79) up to the cache-dependent part of prox_bcd, for the block of one
// column: `tot` = the column's sums (lanes < k: sum dloss * dA per component, lane L-2: sum dA^2),
// `pold` the block.  Returns p' in the component lanes; l2 = ||p'||, st0 = eta * gamma / inv.
template <int L>
__device__ __forceinline__ double pb_block_step(double tot, double pold, double lam, bool kl,
                                                int grp, int reg, double mu, double beta,
                                                double gamma, double eta, double& l2, double& st0) {
    const double hsum = pb_bcast<L>(tot, L - 2, grp);
    double inv = hsum * mu;
    inv += beta;
    st0 = eta * gamma / inv;
    double v = 0.0;
    if (kl) {
        double gr = tot * lam;
        gr += beta * pold;
        gr /= inv;
        v = pold - eta * gr;
        if (reg == REG_L1) {  // l1.py:44-45, element-wise
            const double sg = (v > 0) ? 1.0 : ((v < 0) ? -1.0 : 0.0);
            const double m = fabs(v) - st0;
            v = sg * (m > 0.0 ? m : 0.0);
        } else if (reg == REG_SQL21) {
            v /= 1 + 2 * st0;  // squaredl21.py:46
        }
    }
    l2 = sqrt(pb_group_allsum<L>(v * v));
    return v;
}

template <typename T>
struct PbESet {  // lane u <-> entry e0 + u of the group (u < min(cnt, L))
    int e0, cnt;
    int row, meta;
    T x;
};

// CR = true: relaxed runs for pbcd, degree 2, k <= 30.  A step is a run of consecutive
// columns of the reference's own order that may share rows (schedule_relax: a row in at most two
// columns of the run, <= 64 such conflict rows).  The conflict rows are not in the entry stream:
// their owner publishes the row's record, the slot owners publish the columns' TOTALS (without
// those rows), and every workgroup redundantly replays the run: a column's sums = its totals +
// its conflict rows' terms -- for a row's earlier column from the published record, for its
// later column from the record after the earlier column's update (rounded to T as the scatter
// would), which depends on that column's Delta, which depends (regularizer chain) on everything
// in front of it.  Evaluated in rounds -- all terms from the current Deltas, all blocks' steps,
// the chain from the step's input cache -- until no Delta changes a bit: column q depends on
// columns < q only, so the fixed point is the sequential sweep's result (pbcd.py:110-146).
template <typename T, int M, int L, bool STAMP = false, bool CR = false>
__global__ __launch_bounds__(kPbPrbThreads) void pbcd_prb_kernel(
    PbPrbArgs a, const T* __restrict__ eval, T* __restrict__ R /* packed row records */,
    double* __restrict__ P /* (d,k) */, int k, int d, const double* __restrict__ lams, int loss,
    int reg, RegState rs, int top_ncache, double mu, double beta, double gamma, double eta,
    double* __restrict__ viol_pos) {
    constexpr int NG = kPbPrbThreads / L;  // slot groups per workgroup
    constexpr int QM = 64 / NG;            // slots per group (<= 64 slots per step)
    constexpr int AS = Kind<M>::AS;
    constexpr int NW = kPbPrbThreads / 64;  // waves
    static_assert(!CR || (M == 2 && L == 32), "relaxed pbcd runs: degree 2, k <= 30");
    // (relaxed runs hold ~35 entries per workgroup and step: 4 LDS row slots per group leave room
    // for the replay's tables)
    constexpr int ER = (CR && pbprb_er<T, M>() > 4) ? 4 : pbprb_er<T, M>();
    using ESet = PbESet<T>;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double* sh_red = dyn_lds;                    // [2][NG][L] owner part sums
    double* sh_pt = dyn_lds + 2 * NG * L;        // [64][L] published vectors of the step
    double* sh_scal = sh_pt + 64 * L;            // [64][4] l2, st0, f, -
    double2* sh_xd = reinterpret_cast<double2*>(sh_scal + 256);  // [NG][L] (x, dloss) per entry
    double* sh_cache = sh_scal + 256 + 2 * NG * L;  // [2][kMaxDegree+2] regularizer cache, dcache
    int2* sh_rm = reinterpret_cast<int2*>(sh_cache + 2 * (kMaxDegree + 2));  // [2][NG][L] (row, meta)
    int* sh_ok = reinterpret_cast<int*>(sh_rm + 2 * NG * L);
    // row buffers [2][NW][ER][AS][64]: a wave's slice is written lane-linearly by LDS-DMA
    T* sh_rows = reinterpret_cast<T*>(sh_ok + 4);
    // CR: old blocks, Deltas, conflict terms (earlier / later column), conflict records [64][L];
    // the saved regularizer state; per conflict (qa, qb, xa, xb); per column its conflict list
    double* sh_po = reinterpret_cast<double*>(sh_rows + (size_t)2 * NW * ER * AS * 64);
    double* sh_dl = sh_po + 64 * L;
    double* sh_ta = sh_dl + 64 * L;
    double* sh_tb = sh_ta + 64 * L;
    double* sh_cr = sh_tb + 64 * L;
    double* sh_csave = sh_cr + 64 * L;
    double2* sh_cx = reinterpret_cast<double2*>(sh_csave + 2 * (kMaxDegree + 2));  // [64] (xa, xb)
    int2* sh_cq = reinterpret_cast<int2*>(sh_cx + 64);                             // [64] (qa, qb)
    short* sh_cl = reinterpret_cast<short*>(sh_cq + 64);                           // [64][8]
    int* sh_crow = reinterpret_cast<int*>(sh_cl + 64 * 8);                         // [64] row ids
    int* sh_chg = sh_crow + 64;
    const int g = (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid % L, grp = tid / L;
    const int wlane = tid & 63, wave = tid >> 6;
    const int gb = grp * L;  // the group's slice of the per-entry LDS arrays
    constexpr size_t rowlen = (size_t)AS * L;  // elements of one packed row record
    const bool kl = lane < k;
    const double lam = kl ? lams[lane] : 0.0;
    const bool chained = (reg == REG_SQL21 || reg == REG_OMEGACS);
    const bool fixed_owner = a.G >= 64;
    const int oq = fixed_owner ? pbprb_owned_slot(a.G, g, 0) : -1;  // the slot this WG owns
    if (wave == 0 && wlane < 2 * (kMaxDegree + 2)) {
        const int t = wlane % (kMaxDegree + 2);
        const double* srcp = (wlane < kMaxDegree + 2) ? rs.cache : rs.dcache;
        sh_cache[wlane] = (t < top_ncache) ? srcp[t] : 0.0;
    }
    if (tid == 0) *sh_ok = 1;

    long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = STAMP ? clock64() : 0;
#define PB_STAMP(kk)                        \
    if constexpr (STAMP) {                  \
        if (tid == 0) {                     \
            const long long tn = clock64(); \
            acc[kk] += tn - tprev;          \
            tprev = tn;                     \
        }                                   \
    }

    // ---- pipeline helpers ----------------------------------------------------------------
    auto bounds = [&](int b, int& e0, int& e1) __attribute__((always_inline)) {
        e0 = 0;
        e1 = 0;
        if (b < a.nb) {
            const int32_t* p = a.gsp + ((size_t)g * a.nb + b) * (NG + 1) + grp;
            e0 = p[0];
            e1 = p[1];
        }
    };
    auto load_entries = [&](ESet& s, int e0, int e1) __attribute__((always_inline)) {
        s.e0 = e0;
        s.cnt = e1 - e0;
        const bool v = lane < s.cnt;
        s.row = v ? a.erow[e0 + lane] : 0;
        s.meta = v ? (int)a.emeta[e0 + lane] : 0;
        s.x = v ? eval[e0 + lane] : (T)0;
    };
    auto rows_at = [&](int par, int u, int t) __attribute__((always_inline)) -> T* {
        return sh_rows + ((((size_t)par * NW + wave) * ER + u) * AS + t) * 64;
    };
    // packed rows of a set (-> LDS buffer `par`); hz = 0: the entries not flagged, 1: the flagged
    // ones (after the barrier that ends the step which updated them).  Every lane fetches its own
    // element of the record: the group's L lanes take one contiguous, aligned slice.
    auto fetch_rows = [&](ESet& s, int par, int hz) __attribute__((always_inline)) {
        const int2* rm_ = sh_rm + par * NG * L + gb;
        const int nf = min(s.cnt, ER);
        // wave-uniform trip count (the two groups of a wave differ in nf): the LDS-DMA's
        // destination goes through M0, i.e. it is taken from ONE lane -- with per-lane trip
        // counts the compiler's unrolled remainder loop runs the halves at different u
        const int nfw = (L == 64) ? __builtin_amdgcn_readfirstlane(nf)
                                  : max(__builtin_amdgcn_readlane(nf, 0),
                                        __builtin_amdgcn_readlane(nf, 32));
        for (int ub = 0; ub < nfw; ub += 4) {
            int2 rm[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) rm[uu] = rm_[min(ub + uu, ER - 1)];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int u = ub + uu;
                if (u < nf && ((rm[uu].y >> 7) & 1) == hz) {
                    const size_t base = (size_t)rm[uu].x * rowlen + lane;
#pragma unroll
                    for (int t = 0; t < AS; ++t) {
                        if constexpr (sizeof(T) == 4) {
                            // sc1: served by L2, past the CU's L1 (rows are read once)
                            __builtin_amdgcn_global_load_lds(R + base + (size_t)t * L,
                                                             rows_at(par, min(u, ER - 1), t), 4, 0,
                                                             16);
                        } else {
                            rows_at(par, min(u, ER - 1), t)[wlane] = R[base + (size_t)t * L];
                        }
                    }
                }
            }
        }
    };
    auto publish_meta = [&](const ESet& s, int par) __attribute__((always_inline)) {
        sh_rm[par * NG * L + gb + lane] = make_int2(s.row, s.meta);  // lane-parallel -> LDS
    };
    auto wave_lds_sync = [&]() __attribute__((always_inline)) {
        // LDS writes of this wave visible to its own later reads (the LDS queue is in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto col_id = [&](int c, int ncols_of, int q) __attribute__((always_inline)) -> int {
        return (q >= 0 && q < ncols_of) ? a.jsched[c + q] : -1;
    };
    // the slots of the group's QM positions at a step: QM bytes, 0xFF = none (PbPrbArgs::gtab)
    struct SlotIds {
        unsigned w[QM / 4];
    };
    auto load_slots = [&](int b) __attribute__((always_inline)) -> SlotIds {
        SlotIds r;
#pragma unroll
        for (int i = 0; i < QM / 4; ++i) r.w[i] = 0xFFFFFFFFu;
        if (b < a.nb) {
            const unsigned* p =
                reinterpret_cast<const unsigned*>(a.gtab + ((size_t)g * a.nb + b) * 64 + grp * QM);
#pragma unroll
            for (int i = 0; i < QM / 4; ++i) r.w[i] = p[i];
        }
        return r;
    };
    auto slot_of = [&](const SlotIds& sl, int t) __attribute__((always_inline)) -> int {
        return (int)((sl.w[t >> 2] >> (8 * (t & 3))) & 0xFFu);
    };

    int c0 = a.bptr[0], c1 = a.bptr[min(1, a.nb)];
    int c2 = a.bptr[min(2, a.nb)], c3 = a.bptr[min(3, a.nb)];
    ESet cur, nxt, nn;
    int b2e0, b2e1;  // group bounds of step b+2
    {
        int e0, e1;
        bounds(0, e0, e1);
        load_entries(cur, e0, e1);
        bounds(1, e0, e1);
        load_entries(nxt, e0, e1);
        bounds(2, b2e0, b2e1);
        nn = nxt;
        publish_meta(cur, 0);
        publish_meta(nxt, 1);
        wave_lds_sync();
        fetch_rows(cur, 0, 0);  // step 0: nothing is flagged
    }
    // slot data: the group's slots of steps b .. b+2 (b+3 loaded inside the loop), their column
    // ids of steps b, b+1 (b+2 loaded inside the loop); P rows of step b
    SlotIds qs0 = load_slots(0), qs1 = load_slots(1), qs2 = load_slots(2);
    int j0[QM], j1[QM];
    double po[QM], pon[QM];
#pragma unroll
    for (int t = 0; t < QM; ++t) {
        j0[t] = col_id(c0, c1 - c0, slot_of(qs0, t));
        j1[t] = col_id(c1, c2 - c1, slot_of(qs1, t));
        po[t] = (j0[t] >= 0 && kl) ? P[(size_t)j0[t] * k + lane] : 0.0;
        pon[t] = 0.0;
    }
    int oj1 = col_id(c1, c2 - c1, oq);  // owner's column of step b+1
    double opo = 0.0, opon = 0.0;       // its P row (group 0's lanes)
    {
        const int oj0 = col_id(c0, c1 - c0, oq);
        opo = (oj0 >= 0 && grp == 0 && kl) ? P[(size_t)oj0 * k + lane] : 0.0;
    }
    int cj0 = (wave == 0) ? col_id(c0, c1 - c0, wlane) : -1;  // chain: lane = slot
    int cj1 = (wave == 0) ? col_id(c1, c2 - c1, wlane) : -1;
    double cn0 = (cj0 >= 0 && chained) ? rs.norms[cj0] : 0.0, cn1 = 0.0;
    int cjp = -1;           // chain: column of the previous step held by this lane ...
    double l2n_prev = 0.0;  // ... and its new block norm, stored one step late
    // CR: conflict-row boundaries of steps b .. b+3 and the tables of step 0
    int cfp0 = 0, cfp1 = 0, cfp2 = 0, cfp3 = 0;
    if constexpr (CR) {
        cfp0 = a.cf_ptr[0];
        cfp1 = a.cf_ptr[min(1, a.nb)];
        cfp2 = a.cf_ptr[min(2, a.nb)];
        cfp3 = a.cf_ptr[min(3, a.nb)];
    }
    __syncthreads();

    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = a.bptr[min(b + 4, a.nb)];  // used from the next iteration on
        // slots written this step: this step's and those of the buffer's next use, so that a
        // word read at step b+2 was rewritten at step b (its tag is never tag(b+2))
        const int nw = max(ncols, c3 - c2);
        const unsigned long long tag = prb_tag(b);
        const int par = b & 1;
        double* slabA = a.slabA + (size_t)par * 64 * a.G * L;
        double* slabB = a.slabB + (size_t)par * 64 * L;
        const int2* srm = sh_rm + par * NG * L + gb;  // (row, meta) of the group's entries
        const int nfast = min(cur.cnt, ER);
        const int nconf = CR ? cfp1 - cfp0 : 0;  // this step's conflict rows

        // ---- phase 0: rows this step shares with the previous one (after its barrier)
        fetch_rows(cur, par, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have landed in LDS
        wave_lds_sync();
        {   // lane u <-> entry u: dloss from the record's (yhat, y) (lanes L-2, L-1 of slice 0),
            // with x -> LDS for the group's broadcast reads
            double dl = 0.0;
            if (lane < nfast) {
                const T* r0 = rows_at(par, lane, 0) + (wlane - lane);
                dl = dloss_dev(loss, (double)r0[L - 2], (double)r0[L - 1]);
            }
            sh_xd[gb + lane] = make_double2((double)cur.x, dl);
        }
        wave_lds_sync();
        PB_STAMP(0)
        // ---- phase 1: partial sums of the own rows (pbcd.py:56-67), published per slot
        double gs[QM], hs[QM];
#pragma unroll
        for (int t = 0; t < QM; ++t) {
            gs[t] = 0.0;
            hs[t] = 0.0;
        }
        // the group's fast entries are sorted by slot: segment t = [seg[t], seg[t+1]) (one ballot
        // per slot on the lane-parallel slot indices), so the slot's p_j is a static register
        int seg[QM + 1];
        seg[0] = 0;
#pragma unroll
        for (int t = 0; t < QM; ++t) {
            const unsigned long long bal = __ballot(lane < nfast && (cur.meta & 7) <= t);
            seg[t + 1] = (L == 64) ? __builtin_popcountll(bal)
                                   : __builtin_popcount((unsigned)(bal >> (32 * (grp & 1))));
        }
#pragma unroll
        for (int t = 0; t < QM; ++t) {
            double gacc = 0.0, hacc = 0.0;
            const double p = po[t];
            for (int u = seg[t]; u < seg[t + 1]; ++u) {
                const double2 xd = sh_xd[gb + u];
                double ad[AS];
#pragma unroll
                for (int tt = 0; tt < AS; ++tt) ad[tt] = (double)rows_at(par, u, tt)[wlane];
                const double dprev = kl ? grad_factor<M>(ad, xd.x, p) : 0.0;
                gacc += xd.y * dprev;
                hacc += dprev * dprev;
            }
            gs[t] = gacc;
            hs[t] = hacc;
        }
        for (int u = ER; u < cur.cnt; ++u) {  // slow path: beyond the LDS-staged rows
            const int e = cur.e0 + u;
            const int i = a.erow[e];
            const int qi = (int)a.emeta[e] & 7;
            const double x = (double)eval[e];
            const T* ri = R + (size_t)i * rowlen;
            const double dl = dloss_dev(loss, (double)ri[L - 2], (double)ri[L - 1]);
            double ad[AS];
#pragma unroll
            for (int t = 0; t < AS; ++t) ad[t] = kl ? (double)ri[(size_t)t * L + lane] : 0.0;
            const double p = pb_sel(po, qi);
            const double dprev = kl ? grad_factor<M>(ad, x, p) : 0.0;
            pb_acc<QM>(gs, qi, dl * dprev);
            pb_acc<QM>(hs, qi, dprev * dprev);
        }
        {
            // sum_s inv_step_sizes[s] (pbcd.py:68-70) of the QM slots by one butterfly: lane l
            // ends with the sum of slot (l & (QM-1)); lane L-2 of slot t's vector needs slot t
            double hv[QM];
#pragma unroll
            for (int t = 0; t < QM; ++t) hv[t] = hs[t];
            const double hr = pb_multi_reduce<QM, L>(hv, lane);
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                const int q = slot_of(qs0, t);
                const double hsum = pb_bcast<L>(hr, t, grp);
                if (q < nw) {
                    const double v = (lane == L - 2) ? hsum : (kl ? gs[t] : 0.0);
                    prb_store_granule(slabA + ((size_t)q * a.G + g) * L + lane, v, tag);
                }
            }
        }
        PB_STAMP(1)
        if constexpr (CR) {
            // ---- relaxed runs: this step's conflict rows.  Tables -> LDS; the records of the
            // rows this workgroup owns -> slabR (as the previous step left them: its end barrier
            // drained the scatter's stores).  Slots unused now but read at the buffer's next use
            // are rewritten with zeros by workgroup 0 (stale-tag rule).
            const PrbConf<T>* cfa = reinterpret_cast<const PrbConf<T>*>(a.cf);
            double* slabR = a.slabR + (size_t)par * 64 * L;
            if (tid < nconf) {
                const PrbConf<T> cf = cfa[cfp0 + tid];
                sh_cq[tid] = make_int2(cf.qq & 0xff, cf.qq >> 8);
                sh_cx[tid] = make_double2((double)cf.xa, (double)cf.xb);
                sh_crow[tid] = cf.row;
            }
            if (tid < ncols)
                reinterpret_cast<uint4*>(sh_cl)[tid] = reinterpret_cast<const uint4*>(a.clist)[c0 + tid];
            const int ncw = max(nconf, cfp3 - cfp2);
            for (int c = grp; c < ncw; c += NG) {
                if (c < nconf) {
                    const int row = cfa[cfp0 + c].row;
                    if (row / a.rows_per == g)
                        prb_store_granule(slabR + (size_t)c * L + lane,
                                          (double)R[(size_t)row * rowlen + lane], tag);
                } else if (g == 0) {
                    prb_store_granule(slabR + (size_t)c * L + lane, 0.0, tag);
                }
            }
        }

        // ---- phase 2: owners reduce their slot over the workgroups and take the step
        const int n_rounds = fixed_owner ? 1 : (nw + a.G - 1) / a.G;
        for (int r = 0; r < n_rounds; ++r) {
            const int q = fixed_owner ? oq : pbprb_owned_slot(a.G, g, r);
            const bool own = q >= 0 && q < ncols;
            double* red = sh_red + (size_t)(r & 1) * NG * L;
            if (own) {
                // group `grp` sums the source workgroups grp, grp + NG, ... in that order
                double tot = 0.0;
                constexpr int GU = (L == 32) ? 16 : 8;  // sources per lane and round (L = 32: all 256 in one round)
                for (int s0 = grp; s0 < a.G; s0 += NG * GU) {
                    unsigned long long t[GU];
                    unsigned spins = 0;
                    bool ok = true;
                    for (;;) {
                        bool all = true;
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            const int src = s0 + u * NG;
                            t[u] = (src < a.G)
                                       ? prb_load_granule(slabA + ((size_t)q * a.G + src) * L + lane)
                                       : tag;
                            all = all && ((t[u] & 3ull) == tag);
                        }
                        if (all) break;
                        if (pbprb_poll_fail(a, spins, 1, b)) {
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
                        if (s0 + u * NG < a.G)
                            tot += __longlong_as_double((long long)(t[u] & ~3ull));
                }
                red[grp * L + lane] = tot;
            }
            PB_STAMP(2)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // part sums in LDS
            if (q >= ncols && q < nw && grp == 0) {
                // slot unused in this step but read at the buffer's next use: rewritten now
                prb_store_granule(slabB + (size_t)q * L + lane, 0.0, tag);
                if (a.n_ranks > 1) {
                    const size_t off = ((size_t)par * 64 + q) * a.n_ranks * L;
                    for (int rr = 0; rr < a.n_ranks; ++rr)
                        prb_store_granule_sys(a.slabC[rr] + off + (size_t)a.rank * L + lane, 0.0, tag);
                }
            }
            if (own && grp == 0) {
                double tot = red[lane];
#pragma unroll
                for (int w = 1; w < NG; ++w) tot += red[w * L + lane];
                if (a.n_ranks > 1) {
                    // one xGMI hop: this GPU's vector into every GPU's slabC[slot][rank], then
                    // the n_ranks vectors of the own slabC summed in rank order
                    const size_t off = ((size_t)par * 64 + q) * a.n_ranks * L;
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
                            if (pbprb_poll_fail(a, spins, 2, b)) {
                                ok = false;
                                break;
                            }
                        }
                        gt += __longlong_as_double((long long)(t & ~3ull));
                    }
                    if (!ok) *sh_ok = 0;
                    tot = gt;
                }
                // pbcd._update (pbcd.py:68-79) up to the cache-dependent part of prox_bcd
                double pold = opo;
                if (!fixed_owner) {
                    const int j = a.jsched[c0 + q];
                    pold = kl ? P[(size_t)j * k + lane] : 0.0;
                }
                if constexpr (CR) {
                    // relaxed runs: the columns' sums still lack their conflict rows -- the
                    // totals go out as they are, every workgroup takes the steps itself
                    prb_store_granule(slabB + (size_t)q * L + lane, tot, tag);
                } else {
                    double l2, st0;
                    const double v = pb_block_step<L>(tot, pold, lam, kl, grp /* == 0 */, reg, mu,
                                                      beta, gamma, eta, l2, st0);
                    const double outv = (lane == L - 2) ? l2 : ((lane == L - 1) ? st0 : v);
                    prb_store_granule(slabB + (size_t)q * L + lane, outv, tag);
                }
            }
        }
        PB_STAMP(3)

        // ---- prefetch: entries of step b+2, rows of step b+1 that this step does not touch,
        // column ids / P rows / old norms of the coming steps.  Issued behind the owners' work,
        // in front of the collect poll: vmcnt retires in order, so the first tag check waits for
        // these loads too -- but everybody waits about that long for the owners anyway.  (In
        // front of the owner poll it delays the owners themselves: 13.1 vs 12.4 us per step;
        // behind the collect poll its issue time sits on the critical path: 14.1.  Round 3:
        // waves 1..7 polling at once with an empty load queue and prefetching during wave 0's
        // chain instead: 12.1 vs 11.0 -- the totals arrive ~5 us after the publish whoever polls
        // and however early; the exchange is bound by the burst of 2 MB of partial vectors that
        // all workgroups write, and the owners read, at the same instant.)
        int b3e0, b3e1;
        bounds(b + 3, b3e0, b3e1);
        const SlotIds qs3 = load_slots(b + 3);
        load_entries(nn, b2e0, b2e1);
        fetch_rows(nxt, par ^ 1, 0);
        int j2[QM];
#pragma unroll
        for (int t = 0; t < QM; ++t) {
            j2[t] = col_id(c2, c3 - c2, slot_of(qs2, t));
            pon[t] = (j1[t] >= 0 && kl) ? P[(size_t)j1[t] * k + lane] : 0.0;
        }
        const int oj2 = col_id(c2, c3 - c2, oq);
        opon = (oj1 >= 0 && grp == 0 && kl) ? P[(size_t)oj1 * k + lane] : 0.0;
        const int cj2 = (wave == 0) ? col_id(c2, c3 - c2, wlane) : -1;
        cn1 = (cj1 >= 0 && chained) ? rs.norms[cj1] : 0.0;
        int cfp4 = 0;
        if constexpr (CR) cfp4 = a.cf_ptr[min(b + 4, a.nb)];
        PB_STAMP(4)
        // ---- phase 3: every workgroup collects the published vectors of all slots
        {
            const int total = ncols * L;
            constexpr int RU = 64 * L / kPbPrbThreads;  // granules per thread at most
            unsigned long long t[RU];
            unsigned spins = 0;
            bool ok = true;
            for (;;) {
                bool all = true;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int idx = tid + u * kPbPrbThreads;
                    t[u] = (idx < total) ? prb_load_granule(slabB + idx) : tag;
                    all = all && ((t[u] & 3ull) == tag);
                }
                if (all) break;
                if (pbprb_poll_fail(a, spins, 3, b)) {
                    ok = false;
                    break;
                }
            }
            if (!ok) *sh_ok = 0;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int idx = tid + u * kPbPrbThreads;
                if (idx < total) {
                    const double v = __longlong_as_double((long long)(t[u] & ~3ull));
                    sh_pt[idx] = v;
                    if constexpr (!CR) {
                        const int q = idx / L, l = idx % L;
                        if (l == L - 2) sh_scal[4 * q + 0] = v;
                        if (l == L - 1) sh_scal[4 * q + 1] = v;
                    }
                }
            }
            if constexpr (CR) {  // ... and the conflict rows' records
                const int total2 = nconf * L;
                const double* slabR = a.slabR + (size_t)par * 64 * L;
                spins = 0;
                for (;;) {
                    bool all = true;
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int idx = tid + u * kPbPrbThreads;
                        t[u] = (idx < total2) ? prb_load_granule(slabR + idx) : tag;
                        all = all && ((t[u] & 3ull) == tag);
                    }
                    if (all) break;
                    if (pbprb_poll_fail(a, spins, 4, b)) {
                        ok = false;
                        break;
                    }
                }
                if (!ok) *sh_ok = 0;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int idx = tid + u * kPbPrbThreads;
                    if (idx < total2) sh_cr[idx] = __longlong_as_double((long long)(t[u] & ~3ull));
                }
            }
        }
        PB_STAMP(5)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // sh_pt / sh_scal
        if (!*sh_ok) break;
        PB_STAMP(9)

        // ---- phase 4: shrink factors.  L1: none; L21: from (l2, st0); SquaredL21 / OmegaCS:
        // the scalar cache recurrence in step order (every workgroup redundantly; every
        // workgroup also writes the new block norms -- one step late -- so that its own later
        // reads are consistent)
        double l2n_step = 0.0;
        const double* vfin = sh_pt;  // the blocks' p' of the step (CR: the last round's buffer)
        // the previous step's new block norms go to memory now: every workgroup has finished
        // that step's chain (it published this step's partial sums since), so none can still
        // need the old values
        if (wave == 0 && chained && cjp >= 0) rs.norms[cjp] = l2n_prev;
        auto chain_phase = [&]() __attribute__((always_inline)) {
            if (wave == 0) {
                if (chained) {
                    double l2n_new = 0.0;
                    pbprb_chain_step<M>(wlane, ncols, cj0, cn0, d, reg, rs, top_ncache, sh_scal,
                                        sh_cache, &l2n_new);
                    l2n_step = l2n_new;
                } else if (wlane < ncols) {
                    double f = 1.0;
                    if (reg == REG_L21) {  // l21.py:33-38
                        const double l2 = sh_scal[4 * wlane], st0 = sh_scal[4 * wlane + 1];
                        f = (l2 > st0) ? (1.0 - st0 / l2) : 0.0;
                    }
                    sh_scal[4 * wlane + 2] = f;
                }
            }
        };
        if constexpr (!CR) {
            chain_phase();
        } else {
            // ---- relaxed runs: replay the run (see the header).  My slots' old blocks -> LDS,
            // Deltas start at zero, the regularizer state is saved for the rounds
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                const int q = slot_of(qs0, t);
                if (q < ncols) {
                    sh_po[q * L + lane] = po[t];
                    sh_dl[q * L + lane] = 0.0;
                }
            }
            if (wave == 0 && wlane < 2 * (kMaxDegree + 2)) sh_csave[wlane] = sh_cache[wlane];
            if (tid == 0) *sh_chg = 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the earlier column's term of my conflict rows: from the published record
            for (int c = grp; c < nconf; c += NG) {
                const int2 cq = sh_cq[c];
                const double2 cx = sh_cx[c];
                const double av = kl ? sh_cr[c * L + lane] : 0.0;
                const double pa = sh_po[cq.x * L + lane];
                const double dAa = kl ? cx.x * (av - pa * cx.x) : 0.0;
                const double dl0 = dloss_dev(loss, sh_cr[c * L + L - 2], sh_cr[c * L + L - 1]);
                const double hs = pb_group_allsum<L>(dAa * dAa);
                sh_ta[c * L + lane] = (lane == L - 2) ? hs : dl0 * dAa;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // my columns' sums without the later-column terms: totals + earlier-column terms.
            // Parked in LDS for the rounds (sh_dl: free until the rounds are over)
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                const int q = slot_of(qs0, t);
                if (q < ncols) {
                    double b0 = sh_pt[q * L + lane];
                    for (int e = 0; e < 8; ++e) {
                        const int ce = sh_cl[q * 8 + e];
                        if (ce >= 0 && (ce >> 8) == 0) b0 += sh_ta[(ce & 0xff) * L + lane];
                    }
                    sh_dl[q * L + lane] = b0;
                }
            }
            PB_STAMP(10)
            // Rounds.  Delta of a column = its old block - p' * f, read straight from LDS (first
            // round: Delta = 0).  A round: [the conflict rows' later-column terms, spread evenly
            // over the groups] barrier [my columns' sums and steps] barrier [chain] barrier; it
            // was the last one when no p' and no f changed a bit (then no Delta did).
            for (int round = 0;; ++round) {
                // the later column's term of my conflict rows: the record after the earlier
                // column's update for its current Delta (pbcd.py:135-144), rounded to T as the
                // scatter rounds
                for (int c = grp; c < nconf; c += NG) {
                    const int2 cq = sh_cq[c];
                    const double2 cx = sh_cx[c];
                    const double av = kl ? sh_cr[c * L + lane] : 0.0;
                    const double pa = sh_po[cq.x * L + lane], pbv = sh_po[cq.y * L + lane];
                    const double Da = (round == 0 || !kl)
                                          ? 0.0
                                          : pa - sh_pt[cq.x * L + lane] * sh_scal[4 * cq.x + 2];
                    const double dAa = kl ? cx.x * (av - pa * cx.x) : 0.0;
                    const double a1 = kl ? (double)(T)(av - Da * cx.x) : 0.0;
                    const double dec = pb_group_allsum<L>(kl ? (lam * Da) * dAa : 0.0);
                    const double y1 = (double)(T)(sh_cr[c * L + L - 2] - dec);
                    const double dAb = kl ? cx.y * (a1 - pbv * cx.y) : 0.0;
                    const double dl1 = dloss_dev(loss, y1, sh_cr[c * L + L - 1]);
                    const double hs = pb_group_allsum<L>(dAb * dAb);
                    sh_tb[c * L + lane] = (lane == L - 2) ? hs : dl1 * dAb;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                bool changed = false;
#pragma unroll
                for (int t = 0; t < QM; ++t) {
                    const int q = slot_of(qs0, t);
                    if (q < ncols) {  // (group-uniform): sums = totals + conflict terms; the step
                        double tot = sh_dl[q * L + lane];
                        for (int e = 0; e < 8; ++e) {
                            const int ce = sh_cl[q * 8 + e];
                            if (ce >= 0 && (ce >> 8) != 0) tot += sh_tb[(ce & 0xff) * L + lane];
                        }
                        double l2, st0;
                        const double v = pb_block_step<L>(tot, po[t], lam, kl, grp, reg, mu, beta,
                                                          gamma, eta, l2, st0);
                        changed = changed || round == 0 ||
                                  __double_as_longlong(v) != __double_as_longlong(sh_pt[q * L + lane]);
                        sh_pt[q * L + lane] = v;
                        if (lane == 0) {
                            sh_scal[4 * q + 3] = sh_scal[4 * q + 2];  // f of the previous round
                            sh_scal[4 * q + 0] = l2;
                            sh_scal[4 * q + 1] = st0;
                        }
                    }
                }
                if (__ballot(changed) != 0ull && wlane == 0) *sh_chg = round + 1;
                if (wave == 0 && wlane < 2 * (kMaxDegree + 2)) sh_cache[wlane] = sh_csave[wlane];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                chain_phase();
                if (wave == 0 && wlane < ncols && round > 0 &&
                    __double_as_longlong(sh_scal[4 * wlane + 2]) !=
                        __double_as_longlong(sh_scal[4 * wlane + 3]))
                    *sh_chg = round + 1;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (*sh_chg != round + 1 || round > ncols + 1) {
                    if (tid == 0 && g == a.G - 1) {
                        atomicAdd(&g_branch_count[BR_RELAX_STEPS], 1u);
                        atomicAdd(&g_branch_count[BR_RELAX_ROUNDS], (unsigned)(round + 1));
                    }
                    break;
                }
            }
            PB_STAMP(11)
            // the columns' Deltas for the conflict rows' final records
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                const int q = slot_of(qs0, t);
                if (q < ncols)
                    sh_dl[q * L + lane] = kl ? po[t] - vfin[q * L + lane] * sh_scal[4 * q + 2] : 0.0;
            }
        }
        if (wave == 0 && chained) {
            cjp = cj0;
            l2n_prev = l2n_step;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // factors in LDS
        PB_STAMP(6)

        // ---- phase 5: p_j = f * p_j', write-back, scatter over the own rows (pbcd.py:135-146)
        double pn[QM], up[QM], lu[QM], mv[QM];
        {
            double vav[QM];
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                const int qq = slot_of(qs0, t);
                const int q = min(qq, 63);
                const bool vq = qq < ncols;
                const double f = sh_scal[4 * q + 2];
                const double pt = vfin[q * L + lane];
                pn[t] = (vq && kl) ? pt * f : 0.0;
                up[t] = (vq && kl) ? po[t] - pn[t] : 0.0;
                lu[t] = lam * up[t];
                {   // did the block move?  one ballot per slot (exact no-op otherwise)
                    const unsigned long long bal = __ballot(up[t] != 0.0);
                    const unsigned long long mine =
                        (L == 64) ? bal : ((bal >> (32 * (grp & 1))) & 0xffffffffull);
                    mv[t] = (mine != 0ull) ? 1.0 : 0.0;
                }
                vav[t] = fabs(up[t]);
                if (g == 0 && vq && kl) P[(size_t)j0[t] * k + lane] = pn[t];
            }
            // ||Delta||_1 per slot by one butterfly: lane t of the group ends with slot t's sum
            if (g == 0) {
                const double vr = pb_multi_reduce<QM, L>(vav, lane);
                const int ql = slot_of(qs0, lane & (QM - 1));
                if (lane < QM && ql < ncols) viol_pos[c0 + ql] = vr;
            }
        }
        {
            // per slot (static registers po / up / lu), per entry: the new record -- cache values
            // in the component lanes, the prediction minus sum_s lam_s Delta_s dA_s (one DPP
            // all-reduce) in lane L-2, the target in lane L-1, zeros in the padding -- written
            // back as whole slices (slice 0 of float storage, k <= 30: one full 128-byte line)
#pragma unroll
            for (int t = 0; t < QM; ++t) {
                if (mv[t] == 0.0) continue;  // block did not move: exact no-op (group-uniform)
                const double pol = po[t], upl = up[t], lul = lu[t], pnl = pn[t];
                for (int u = seg[t]; u < seg[t + 1]; ++u) {
                    const double x = sh_xd[gb + u].x;
                    const size_t base = (size_t)srm[u].x * rowlen + lane;
                    double ad[AS];
#pragma unroll
                    for (int tt = 0; tt < AS; ++tt) ad[tt] = (double)rows_at(par, u, tt)[wlane];
                    // lane L-2 holds yhat_old in ad[0], lane L-1 the target
                    if constexpr (M == 0) {  // pbcd_all.py:121-127
                        const double a0 = kl ? ad[0] : 0.0;
                        double a1 = a0 / (1.0 + x * pol);
                        a1 *= 1.0 + x * pnl;
                        const double d_old = pb_group_allsum<L>(kl ? lam * a0 : 0.0);
                        const double d_new = pb_group_allsum<L>(kl ? lam * a1 : 0.0);
                        double outv = kl ? a1 : 0.0;
                        if (lane == L - 2) outv = (ad[0] - d_old) + d_new;
                        if (lane == L - 1) outv = ad[0];
                        R[base] = (T)outv;
                    } else {
                        double nv[AS];
                        double dprev = x;
#pragma unroll
                        for (int tt = 1; tt < M; ++tt) {
                            const double avv = kl ? ad[tt - 1] : 0.0;
                            const double dcur = x * (avv - pol * dprev);
                            nv[tt - 1] = kl ? avv - upl * dprev : 0.0;
                            dprev = dcur;
                        }
                        const double dec = pb_group_allsum<L>(kl ? lul * dprev : 0.0);
                        if (lane == L - 2) nv[0] = ad[0] - dec;
                        if (lane == L - 1) nv[0] = ad[0];
#pragma unroll
                        for (int tt = 0; tt < AS; ++tt) R[base + (size_t)tt * L] = (T)nv[tt];
                    }
                }
            }
        }
        for (int u = ER; u < cur.cnt; ++u) {  // slow path
            const int e = cur.e0 + u;
            const int i = a.erow[e];
            const int qi = (int)a.emeta[e] & 7;
            if (pb_sel(mv, qi) == 0.0) continue;
            const double x = (double)eval[e];
            T* ri = R + (size_t)i * rowlen;
            const double y0 = (double)ri[L - 2];
            const double pol = pb_sel(po, qi), upl = pb_sel(up, qi);
            if constexpr (M == 0) {
                const double pnl = pb_sel(pn, qi);
                double d_old = 0.0, d_new = 0.0;
                if (kl) {
                    const double a0 = (double)ri[lane];
                    double a1 = a0 / (1.0 + x * pol);
                    a1 *= 1.0 + x * pnl;
                    ri[lane] = (T)a1;
                    d_old = lam * a0;
                    d_new = lam * a1;
                }
                d_old = group_sum(d_old, L);
                d_new = group_sum(d_new, L);
                if (lane == 0) ri[L - 2] = (T)((y0 - d_old) + d_new);
            } else {
                const double lul = pb_sel(lu, qi);
                double accv = 0.0;
                if (kl) {
                    double dprev = x;
#pragma unroll
                    for (int t = 1; t < M; ++t) {
                        const double avv = (double)ri[(size_t)(t - 1) * L + lane];
                        const double dcur = x * (avv - pol * dprev);
                        ri[(size_t)(t - 1) * L + lane] = (T)(avv - upl * dprev);
                        dprev = dcur;
                    }
                    accv = lul * dprev;
                }
                accv = group_sum(accv, L);
                if (lane == 0) ri[L - 2] = (T)(y0 - accv);
            }
        }
        if constexpr (CR) {
            // the conflict rows' final records, by their owner: both updates in order
            // (pbcd.py:135-144 twice), each rounded to T as the scatter rounds
            for (int c = grp; c < nconf; c += NG) {
                const int row = sh_crow[c];
                if (row / a.rows_per != g) continue;
                const int2 cq = sh_cq[c];
                const double2 cx = sh_cx[c];
                const double av = kl ? sh_cr[c * L + lane] : 0.0;
                const double pa = sh_po[cq.x * L + lane], pbv = sh_po[cq.y * L + lane];
                const double Da = sh_dl[cq.x * L + lane], Db = sh_dl[cq.y * L + lane];
                const double dAa = kl ? cx.x * (av - pa * cx.x) : 0.0;
                const double a1 = kl ? (double)(T)(av - Da * cx.x) : 0.0;
                const double dec_a = pb_group_allsum<L>(kl ? (lam * Da) * dAa : 0.0);
                const double y1 = (double)(T)(sh_cr[c * L + L - 2] - dec_a);
                const double dAb = kl ? cx.y * (a1 - pbv * cx.y) : 0.0;
                const double a2 = kl ? a1 - Db * cx.y : 0.0;
                const double dec_b = pb_group_allsum<L>(kl ? (lam * Db) * dAb : 0.0);
                double outv = a2;
                if (lane == L - 2) outv = y1 - dec_b;
                if (lane == L - 1) outv = sh_cr[c * L + L - 1];
                R[(size_t)row * rowlen + lane] = (T)outv;
            }
        }
        PB_STAMP(7)
        // ---- rotate the pipeline
        cur = nxt;
        nxt = nn;
        publish_meta(nxt, par);  // step b+2's slice (parity b&1): step b is done with it
        b2e0 = b3e0;
        b2e1 = b3e1;
        qs0 = qs1;
        qs1 = qs2;
        qs2 = qs3;
        if constexpr (CR) {
            cfp0 = cfp1;
            cfp1 = cfp2;
            cfp2 = cfp3;
            cfp3 = cfp4;
        }
#pragma unroll
        for (int t = 0; t < QM; ++t) {
            j0[t] = j1[t];
            j1[t] = j2[t];
            po[t] = pon[t];
        }
        oj1 = oj2;
        opo = opon;
        cj0 = cj1;
        cj1 = cj2;
        cn0 = cn1;
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        __syncthreads();  // rows move between groups from step to step (stores drained)
        PB_STAMP(8)
    }
#undef PB_STAMP
    if (wave == 0 && chained && cjp >= 0) rs.norms[cjp] = l2n_prev;  // the last step's norms
    if (wave == 0 && g == 0 && wlane < 2 * (kMaxDegree + 2)) {
        const int t = wlane % (kMaxDegree + 2);
        double* dstp = (wlane < kMaxDegree + 2) ? rs.cache : rs.dcache;
        if (t < top_ncache) dstp[t] = sh_cache[wlane];
    }
    if (STAMP && a.stamps != nullptr && tid == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) a.stamps[(size_t)g * 16 + q] = acc[q];
    }
}

// dynamic LDS the kernel needs (bytes)
template <typename T, int M, int L, bool CR = false>
constexpr size_t pbcd_prb_lds_bytes() {
    constexpr int NG = kPbPrbThreads / L;
    constexpr int AS = Kind<M>::AS;
    constexpr int ER = (CR && pbprb_er<T, M>() > 4) ? 4 : pbprb_er<T, M>();
    return (CR ? sizeof(double) * (5 * 64 * L + 2 * (kMaxDegree + 2) + 2 * 64) + sizeof(int) * (3 * 64 + 4) +
                     sizeof(short) * 64 * 8 : 0) +
           sizeof(double) * (2 * NG * L + 64 * L + 256 + 2 * NG * L + 2 * (kMaxDegree + 2)) +
           sizeof(int) * (4 * NG * L + 4) +
           sizeof(T) * (size_t)2 * (kPbPrbThreads / 64) * ER * AS * 64;
}

}  // namespace spfm
