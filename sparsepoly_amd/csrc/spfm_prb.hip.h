// spfm_prb.hip.h -- persistent row-block passes (pcd_prb_kernel, lin_prb_kernel) and their exchange
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
#pragma once
#include "spfm_common.hip.h"
#include "spfm_pcd.hip.h"

namespace spfm {

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
// (agent-scope relaxed atomic stores = global_store sc1) into slab[parity][g][slot]
// as TAGGED GRANULES (below: the data is the flag, no counter, no drain), and every
// workgroup reads all G slabs with sc1 loads, re-reading until every word carries the
// step's tag, and sums them in fixed order g = 0..G-1 (bitwise identical in every
// workgroup).  All stores and loads of the hand-off are sc1, one workgroup per CU
// (MI355X_MICROARCH.md, valid hand-off forms).  Every workgroup then runs the scalar
// chain redundantly and scatter-updates its own rows.  Slabs are double-buffered by
// step parity: a workgroup can only be two publishes ahead of a reader if it passed
// the intermediate all-gather, which needs that reader's publish.
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
    double* slab;          // [2][G][64][2] partial sums (parity double-buffered)
    int rows_per;          // rows per workgroup (row block g = [g*rows_per, (g+1)*rows_per))
    int n_rows;            // n_samples
    unsigned* abort_flag;  // [1]
    unsigned spin_max;     // polls of one wait before the pass gives up (default 2^21)
    long long* stamps;     // diagnostic: [G][16] accumulated cycles per phase (8 control-wave,
                           // 8 worker-wave values), or nullptr
    // multi-GPU (n_ranks > 1): after the local sweep every workgroup holds this GPU's per-slot
    // totals; workgroup 0 writes them into EVERY GPU's exchange slab (peer-mapped stores over
    // xGMI, system scope), every workgroup then adds the n_ranks vectors found in its own GPU's
    // slab in rank order -- one more hop, no collective, bit-identical totals on every GPU
    int n_ranks, rank;
    double* const* xslab;  // [n_ranks] peer pointers; xslab[r] = GPU r's [2][n_ranks][64][2]
    // relaxed runs (CR instantiation of pcd_prb_kernel; spfm_schedule.cpp schedule_relax): the
    // conflict rows of a step -- rows shared by two of its columns -- are not in the entry
    // stream; their owners publish the row state, every workgroup's chain replays them
    const int32_t* cf_ptr;  // [nb+1] conflict rows of a step
    const void* cf;         // PrbConf<T>[]: row, slots of the two columns, their two x values
    const int16_t* clist;   // [d][8] per column position: conflict index | role << 8, -1 none
    double* cslab;          // [2][64][4] tagged granules: (yhat or residual, y, A[i,1]) of a row
    void* rec;              // LR = 3: packed row records float4[n] of the pass's component
};

template <typename T>
struct __attribute__((aligned(8))) PrbConf {
    int32_t row;
    int32_t qq;  // slot of the earlier column | slot of the later column << 8
    T xa, xb;    // the row's entries in the two columns
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

// plain store: the line stays in the storing XCD's L2 (same-XCD readers hit it there)
__device__ __forceinline__ void prb_store_granule_l2(double* p, double v, unsigned long long tag) {
    const unsigned long long u =
        ((unsigned long long)__double_as_longlong(v) & ~3ull) | tag;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(u) : "memory");
}

// system-scope forms of the tagged granule (cross-GPU exchange)
__device__ __forceinline__ void prb_store_granule_sys(double* p, double v, unsigned long long tag) {
    const unsigned long long u =
        ((unsigned long long)__double_as_longlong(v) & ~3ull) | tag;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long prb_load_granule_sys(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cross-GPU stage of the exchange, run by a control wave (lane = slot): tot[0..NV) are this
// GPU's totals (identical in every workgroup); on return they are the sums over all GPUs.
template <int NV>
__device__ __forceinline__ bool prb_cross_gpu(const PrbArgs& a, int g, int b, int lane,
                                              double* tot) {
    const unsigned long long tag = prb_tag(b);
    const size_t par_off = (size_t)(b & 1) * a.n_ranks * 64 * 2;
    if (g == 0) {
        for (int rr = 0; rr < a.n_ranks; ++rr) {
            double* dst = a.xslab[rr] + par_off + ((size_t)a.rank * 64 + lane) * 2;
#pragma unroll
            for (int v = 0; v < NV; ++v) prb_store_granule_sys(dst + v, tot[v], tag);
        }
    }
    const double* mine = a.xslab[a.rank] + par_off + (size_t)lane * 2;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    for (int rr = 0; rr < a.n_ranks; ++rr) {
        unsigned long long t[NV];
        unsigned spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                t[v] = prb_load_granule_sys(mine + (size_t)rr * 64 * 2 + v);
                all = all && ((t[v] & 3ull) == tag);
            }
            if (all) break;
            if ((++spins & 63u) == 0) {
                if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                    spins > a.spin_max) {
                    __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += __longlong_as_double((long long)(t[v] & ~3ull));
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) tot[v] = acc[v];
    return true;
}

// Sweeping wave `w` (0..nparts-1) sums the granules of workgroups [w*G/nparts,
// (w+1)*G/nparts) for slot `lane` (all loads in flight together, re-swept until every
// tag matches); the control wave later adds the part sums in order w = 0..nparts-1, so
// the total is the same bit pattern in every workgroup.  Returns false after a bounded
// number of sweeps.  (prb_collect_quarter: the name dates from four sweeping waves.)
typedef unsigned __attribute__((ext_vector_type(4))) prb_u4;

// Eight 16-byte agent-scope loads (one granule PAIR each) issued back to back, then one
// wait.  A pair is two independently tagged 8-byte words, so a torn 16-byte read is
// harmless; the wide load halves the number of cache-line requests of the sweep, which is
// bound by the CU's request rate (every granule line is fetched uncached through sc1).
__device__ __forceinline__ void prb_load_pairs8(const double* const* p, prb_u4* r) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]),
          "=&v"(r[6]), "=&v"(r[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}

template <int NV>
__device__ __forceinline__ bool prb_collect_quarter(const PrbArgs& a, int b, int w, int lane,
                                                    int ncols, double* out /* [nparts][64][2] LDS */,
                                                    int nparts = 4) {
    const double* slab = a.slab + (size_t)(b & 1) * a.G * 64 * 2;
    const unsigned long long tag = prb_tag(b);
    const int g0 = (a.G * w) / nparts, g1 = (a.G * (w + 1)) / nparts;
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
                if constexpr (NV == 2) {
                    const double* ptr[GU];
                    prb_u4 r[GU];
#pragma unroll
                    for (int u = 0; u < GU; ++u)  // beyond the part: re-read its last pair
                        ptr[u] = sl + (size_t)((gg + u < g1) ? gg + u : g1 - 1) * 128;
                    prb_load_pairs8(ptr, r);
#pragma unroll
                    for (int u = 0; u < GU; ++u) {
                        t[u][0] = ((unsigned long long)r[u].y << 32) | r[u].x;
                        t[u][1] = ((unsigned long long)r[u].w << 32) | r[u].z;
                        all = all && ((t[u][0] & 3ull) == tag) && ((t[u][1] & 3ull) == tag);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < GU; ++u)
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const bool in = gg + u < g1;
                            t[u][v] = in ? prb_load_granule(sl + (size_t)(gg + u) * 128 + v) : tag;
                            all = all && ((t[u][v] & 3ull) == tag);
                        }
                }
                if (all) break;
                if ((++spins & 63u) == 0) {
                    if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) ||
                        spins > a.spin_max) {
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
    const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
    // Two separate paths on purpose: the mask is consumed only inside the has_long branch, so
    // the common (no long slot anywhere) path carries no wait for it.  Merged, the compiler
    // put an s_waitcnt vmcnt(0) at the join, which -- vmcnt retiring in order -- also
    // drained the granule stores published just before (0.5 us per step).
    if (a.has_long) {
        const uint32_t* lm = a.lmask + ((size_t)g * a.nb + b) * 2;
        lmask = ((unsigned long long)lm[1] << 32) | (unsigned long long)lm[0];
        if (slot < ncols && !((lmask >> slot) & 1ull)) {  // long slots: no per-lane entries
            e0 = spb[slot];
            e1 = spb[slot + 1];
        }
    } else if (slot < ncols) {
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
                                                 int e0, int e1, int sub, PrbEntries<T>& en,
                                                 int pad_row = 0) {
    en.e0 = e0;
    en.e1 = e1;
#pragma unroll
    for (int u = 0; u < PRB_PF; ++u) {
        const int e = e0 + sub + 4 * u;
        const bool v = e < e1;
        en.row[u] = v ? a.erow[e] : pad_row;  // padding reads a valid row of this block
        en.x[u] = v ? eval[e] : (T)0;
    }
}

// Workgroup = 8 wavefronts: wave 0 is the CONTROL wave (chain), waves 1..4 are WORKERS
// (256 threads = 64 slots x 4 lanes) that own the entries, waves 5..7 are HELPERS.  All
// eight take part in the granule sweep, an eighth of the workgroups each, so that at
// G = 64 every lane needs ONE round of 8 pair loads instead of two back-to-back rounds (the
// bare exchange, spfm_debug_exchange_cost: 2.10 us instead of 2.60 us per step).  The
// control wave and the helpers start polling only when the own workers have published (an
// LDS word): polling from the start of the step floods the fabric.  Software
// pipeline of step b: its entries are already in worker registers (loaded during step
// b-1 from slot bounds loaded during step b-2), so phase 1 starts with the row gathers;
// the workers issue the next step's streaming loads while the control wave waits for
// the other workgroups.
constexpr int kPrbThreads = 512;
constexpr int kPrbParts = 8;  // waves that sweep = parts of the workgroup range
constexpr int kPrbLdsFixed = 2560;  // doubles of fixed LDS (control data, part sums, long slots)

// LR != 0: row state resident in LDS (float storage; one cache value per row: M == 2 or
// the all-subsets model, or two: M == 3).  The workgroup keeps A[i, 1..AS] and a 4-byte
// prediction word of its rows in LDS for the whole pass (8-9 bytes per row, 125 KB at
// 15 625 rows; 12-13 bytes for M == 3, which needs more, smaller row blocks), so the
// per-step gather and scatter are LDS accesses instead of L2 round trips and the
// end-of-step barrier no longer waits for store acknowledgements.
//   LR == 1 (squared loss): the word is the residual r_i = yhat_i - y_i; dloss is the
//            residual itself, so the kernel runs unchanged with (yhat, y) := (r, 0);
//            written back as yhat = r + y.
//   LR == 2 (targets are +-1, any loss): the word is yhat_i, the label's sign is one byte.
// The block is loaded at the start and written back at the end of the launch.
// STAMP = true compiles the in-kernel phase timers in (diagnostic build of one
// configuration); as a runtime switch they cost 5 % of every step.
// REGC >= 0: the regularizer as a compile-time constant (the chain then carries only that
// regularizer's code: -3.6 % per step on config 2); REGC = -1: taken from the argument.
// MG = true: several GPUs share the sweep (cross-GPU stage behind the local sweep); a separate
// instantiation, because even a never-taken branch in the control wave cost 2 % per step.
template <int NV>
__device__ __forceinline__ bool prb_poll(const PrbArgs& a, const double* p, unsigned long long tag,
                                         double* out);

// pcd.py:124-133 at degree 3 on one packed row record (yhat, y, A[i,1], A[i,2])
__device__ __forceinline__ void pcd_sync_entry_rec(size_t i, double x, double p_old, double upd,
                                                   double lam, float4* __restrict__ rec) {
    const float4 r = rec[i];
    const double a1 = (double)r.z, a2 = (double)r.w;
    const double d1 = x * (a1 - p_old * x);
    const double d2 = x * (a2 - p_old * d1);
    float4 o;
    o.x = (float)((double)r.x - lam * upd * d2);
    o.y = r.y;
    o.z = (float)(a1 - upd * x);
    o.w = (float)(a2 - upd * d1);
    rec[i] = o;
}

// rows of component ctl->s as packed records, and back (degree 3, float; pcd_prb_kernel LR = 3)
static __global__ void prb_pack3_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                 const float* __restrict__ yy, const float* __restrict__ A_all,
                                 float4* __restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float* A = A_all + (size_t)ctl->s * a_stride;
        float4 r;
        r.x = yy[2 * i];
        r.y = yy[2 * i + 1];
        r.z = A[2 * i];
        r.w = A[2 * i + 1];
        rec[i] = r;
    }
}
static __global__ void prb_unpack3_kernel(const Ctl* __restrict__ ctl, int64_t n, size_t a_stride,
                                   const float4* __restrict__ rec, float* __restrict__ yy,
                                   float* __restrict__ A_all) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float* A = A_all + (size_t)ctl->s * a_stride;
        const float4 r = rec[i];
        yy[2 * i] = r.x;
        A[2 * i] = r.z;
        A[2 * i + 1] = r.w;
    }
}

// CR = true: relaxed runs (steps whose columns may share rows, see PrbArgs / schedule_relax);
// degree 2, single GPU.
constexpr int kPrbLdsCR = 256;  // extra doubles of fixed LDS of the CR instantiation
template <typename T, int M, int LOSS, int LR, bool STAMP = false, int REGC = -1, bool MG = false,
          bool CR = false>
__global__ __launch_bounds__(kPrbThreads) void pcd_prb_kernel(
    const Ctl* __restrict__ ctl, PrbArgs a, const T* __restrict__ eval, T* __restrict__ A_all,
    size_t a_stride, T* __restrict__ yy, const double* __restrict__ pold_sched,
    double* __restrict__ P, int d, int reg, const double* __restrict__ cache_in, double mu,
    double beta, double gamma, double eta, double* __restrict__ viol_pos) {
    T* __restrict__ A = A_all + (size_t)ctl->s * a_stride;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];  // sized to pin 1 WG / CU
    double* sh_delta = dyn_lds + 128;  // [64]
    double* sh_pold = dyn_lds + 192;   // [64]
    double* sh_chain = dyn_lds + 256;  // [68][4] operands / results of the serial prox loop
    double* sh_quart = dyn_lds + 1536;  // [8][64][2] part sums over workgroups
    int* sh_ok = reinterpret_cast<int*>(dyn_lds + 768);
    int* sh_go = reinterpret_cast<int*>(dyn_lds + 770);  // last step this workgroup published
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool control = wave == 0;
    const bool worker = wave >= 1 && wave <= 4;
    const int part = worker ? wave - 1 : (control ? 4 : wave);  // swept part, 0..7
    const int wt = tid - 64;  // worker thread id (0..255 on the worker waves)
    const int slot = worker ? (wt >> 2) : 64, sub = wt & 3;
    const int s = ctl->s;
    const double lam = ctl->lam;
    double* ps = P + (size_t)s * d;
    double cache[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) cache[t] = cache_in[t];

    double* sh_long = dyn_lds + 1024;  // [64][4][2] wave partials of long slots
    // LR 1 / 2: rows resident in LDS; 3: rows in global memory as packed 16-byte records
    // (yhat, y, A[i,1], A[i,2]) -- degree 3, float: one line per gather, one store per scatter
    // instead of two lines and three 4-byte stores
    constexpr bool LDSR = (LR == 1 || LR == 2);
    constexpr bool PK = (LR == 3);
    static_assert(!LDSR || (Kind<M>::AS <= 2 && sizeof(T) == 4 &&
                            (LR == 2 || LOSS == LOSS_SQUARED)),
                  "LDS-resident rows: float storage, one or two cache values per row");
    static_assert(!PK || (sizeof(T) == 4 && Kind<M>::AS == 2),
                  "packed row records: float storage, two cache values per row");
    float4* __restrict__ rec = reinterpret_cast<float4*>(a.rec);
    constexpr int AS = Kind<M>::AS;
    static_assert(!CR || (M >= 2 && !MG && !STAMP), "relaxed runs: degree >= 2, single GPU");
    constexpr int CSN = 2 + Kind<M>::AS;  // CR: published state of a conflict row (yhat, y, A[i,1..])
    constexpr int CSS = 8;                // ... stride of a conflict row's granules in the slab
    const int row0 = LDSR ? g * a.rows_per : 0;
    // CR: per-conflict contributions of the step, behind the fixed block
    double* sh_ce = dyn_lds + kPrbLdsFixed;  // [64][2] of a row's EARLIER column (constant)
    double* sh_cv = sh_ce + 128;             // [64][2] of its LATER column (per round)
    T* lds_a = reinterpret_cast<T*>(dyn_lds + kPrbLdsFixed + (CR ? kPrbLdsCR : 0));  // [rows_per][AS] A[i, 1..AS]
    T* lds_r = lds_a + (size_t)a.rows_per * AS;                // [rows_per] residual or yhat
    unsigned char* lds_s = reinterpret_cast<unsigned char*>(lds_r + a.rows_per);  // y > 0
    if constexpr (LDSR) {
        const int nr = min(a.rows_per, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPrbThreads) {
            const typename Vec2<T>::type yv = yy2[(size_t)(row0 + il)];
#pragma unroll
            for (int t = 0; t < AS; ++t) lds_a[il * AS + t] = A[(size_t)(row0 + il) * AS + t];
            if constexpr (LR == 1) {
                lds_r[il] = (T)((double)yv.x - (double)yv.y);
            } else {
                lds_r[il] = yv.x;
                lds_s[il] = yv.y > (T)0 ? 1 : 0;
            }
        }
        __syncthreads();
    }
    PrbEntries<T> cur, nxt;
    int c0 = a.bptr[0], c1 = a.bptr[1];
    int c2 = (a.nb > 1) ? a.bptr[2] : c1;
    int c3 = (a.nb > 2) ? a.bptr[3] : c2;
    unsigned long long lm0 = 0ull, lm1 = 0ull;  // long-slot masks of steps b, b+1
    {
        int e0, e1;
        prb_load_sp(a, g, 0, slot, c1 - c0, e0, e1, lm0);
        prb_load_entries<T>(a, eval, e0, e1, sub, cur, row0);
    }
    int ne0 = 0, ne1 = 0;  // slot bounds of step b+1
    if (a.nb > 1) prb_load_sp(a, g, 1, slot, c2 - c1, ne0, ne1, lm1);
    double p_slot = (slot < c1 - c0) ? pold_sched[c0 + slot] : 0.0;
    if (tid == 0) {
        *sh_ok = 1;
        *sh_go = 0;
    }
    // CR: the control wave's view of the conflict rows: lane c <-> conflict c of step b (cfc)
    // and b+1 (cfn), lane q <-> conflict list of column q (clq / clqn); cp0..cp3 = cf_ptr[b..b+3]
    int cp0 = 0, cp1 = 0, cp2 = 0, cp3 = 0;
    PrbConf<T> cfc, cfn;
    cfc.row = cfn.row = 0;
    cfc.qq = cfn.qq = 0;
    cfc.xa = cfc.xb = cfn.xa = cfn.xb = (T)0;
    prb_u4 clq = {~0u, ~0u, ~0u, ~0u}, clqn = {~0u, ~0u, ~0u, ~0u};
    if constexpr (CR) {
        if (control) {
            const PrbConf<T>* cfa = reinterpret_cast<const PrbConf<T>*>(a.cf);
            cp0 = a.cf_ptr[0];
            cp1 = a.cf_ptr[min(1, a.nb)];
            cp2 = a.cf_ptr[min(2, a.nb)];
            cp3 = a.cf_ptr[min(3, a.nb)];
            if (lane < cp1 - cp0) cfc = cfa[cp0 + lane];
            if (lane < c1 - c0)
                clq = reinterpret_cast<const prb_u4*>(a.clist)[c0 + lane];
        }
    }
    // diagnostic stamps (only when a.stamps != nullptr): cycles per phase, thread 0
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = STAMP ? clock64() : 0;
#define PRB_STAMP(k)                        \
    if constexpr (STAMP) {                  \
        if (tid == 0) {                     \
            const long long tn = clock64(); \
            acc[k] += tn - tprev;           \
            tprev = tn;                     \
        }                                   \
    }
#define PRB_HSTAMP(k)                       \
    if constexpr (STAMP) {                  \
        if (tid == 320) {                   \
            const long long tn = clock64(); \
            acc[k] += tn - tprev;           \
            tprev = tn;                     \
        }                                   \
    }
#define PRB_WSTAMP(k)                       \
    if constexpr (STAMP) {                  \
        if (tid == 64) {                    \
            const long long tn = clock64(); \
            acc[k] += tn - tprev;           \
            tprev = tn;                     \
        }                                   \
    }

    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = (b + 4 <= a.nb) ? a.bptr[b + 4] : c3;  // used two steps from now
        // Drain vmcnt here, where it is free (this step's entries are needed at once and
        // everything else was issued a step ago).  It is a real S_WAITCNT (not inline asm), so
        // the compiler's wait-count bookkeeping knows that nothing is pending from the previous
        // iteration; without it, loop-carried destinations (slot bounds, prefetched entries)
        // made it insert vmcnt(0) right behind the granule stores further down, i.e. wait
        // for their acknowledgement (0.5 us per step).  vmcnt(0), expcnt/lgkmcnt untouched.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // ---- phase 1 (workers): gather the rows of the prefetched entries, partial sums
        // (pcd.py:52-59); A / yhat values stay in registers for phase 3
        double av[PRB_PF][AS];
        double yh[PRB_PF], yt[PRB_PF], dlast[PRB_PF];
        double pl = 0.0;
        int jl = 0;
        double cst[CSN];  // CR, control wave: state of conflict row `lane` (yhat, y, A[i,1..AS])
#pragma unroll
        for (int t = 0; t < CSN; ++t) cst[t] = 0.0;
        if (control) {
            if (lane < ncols) {
                pl = pold_sched[c0 + lane];
                if (g == 0) jl = a.desc[c0 + lane].j;
            }
            if constexpr (CR) {
                // conflict rows of this step: the owner publishes the row's state (as the
                // workers would gather it); slots the buffer's next use will read are rewritten
                // with zeros (stale tags, as for the column slots)
                const int nc = cp1 - cp0, ncw = max(nc, cp3 - cp2);
                const unsigned long long ctag = prb_tag(b);
                double* cs = a.cslab + ((size_t)(b & 1) * 64 + lane) * CSS;
                if (lane < nc) {
                    const int i = cfc.row;
                    if (i / a.rows_per == g) {
                        double st[CSN];
                        if constexpr (LDSR) {
                            st[0] = (double)lds_r[i - row0];
                            st[1] = (LR == 1) ? 0.0 : (lds_s[i - row0] ? 1.0 : -1.0);
#pragma unroll
                            for (int t = 0; t < AS; ++t) st[2 + t] = (double)lds_a[(i - row0) * AS + t];
                        } else if constexpr (PK) {
                            const float4 r = rec[(size_t)i];
                            st[0] = (double)r.x;
                            st[1] = (double)r.y;
                            st[2] = (double)r.z;
                            st[2 + AS - 1] = (double)r.w;
                        } else {
                            const typename Vec2<T>::type yv = yy2[(size_t)i];
                            st[0] = (double)yv.x;
                            st[1] = (double)yv.y;
#pragma unroll
                            for (int t = 0; t < AS; ++t) st[2 + t] = (double)A[(size_t)i * AS + t];
                        }
#pragma unroll
                        for (int t = 0; t < CSN; ++t) prb_store_granule(cs + t, st[t], ctag);
                    }
                } else if (lane < ncw && (lane % a.G) == g) {
#pragma unroll
                    for (int t = 0; t < CSN; ++t) prb_store_granule(cs + t, 0.0, ctag);
                }
            }
        } else if (worker) {
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {  // all gathers in flight before any use
                if constexpr (LDSR) {
                    const int il = cur.row[u] - row0;
                    yh[u] = (double)lds_r[il];
                    yt[u] = (LR == 1) ? 0.0 : (lds_s[il] ? 1.0 : -1.0);
#pragma unroll
                    for (int t = 0; t < AS; ++t) av[u][t] = (double)lds_a[il * AS + t];
                } else if constexpr (PK) {
                    const float4 r = rec[(size_t)cur.row[u]];
                    yh[u] = (double)r.x;
                    yt[u] = (double)r.y;
                    av[u][0] = (double)r.z;
                    av[u][AS - 1] = (double)r.w;
                } else {
                    const size_t i = (size_t)cur.row[u];
                    const typename Vec2<T>::type yv = yy2[i];
                    yh[u] = (double)yv.x;
                    yt[u] = (double)yv.y;
#pragma unroll
                    for (int t = 0; t < AS; ++t) av[u][t] = (double)A[i * AS + t];
                }
            }
            double ag = 0.0, ah = 0.0;
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {
                const double dprev = grad_factor<M>(av[u], (double)cur.x[u], p_slot);
                dlast[u] = dprev;
                const double dl = dloss_dev(LOSS, yh[u], yt[u]);
                // (the mask is needed although a padding entry has x = 0: its row state may be
                // uninitialised LDS of a workgroup without rows, and NaN * 0 is NaN)
                const bool v = cur.e0 + sub + 4 * u < cur.e1;
                ag += v ? dl * dprev : 0.0;
                ah += v ? dprev * dprev : 0.0;
            }
            for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {  // rare: long slot
                const int i = a.erow[e];
                const double x = (double)eval[e];
                double a1[AS], y0, y1;
                if constexpr (LDSR) {
                    y0 = (double)lds_r[i - row0];
                    y1 = (LR == 1) ? 0.0 : (lds_s[i - row0] ? 1.0 : -1.0);
#pragma unroll
                    for (int t = 0; t < AS; ++t) a1[t] = (double)lds_a[(i - row0) * AS + t];
                } else if constexpr (PK) {
                    const float4 r = rec[(size_t)i];
                    y0 = (double)r.x;
                    y1 = (double)r.y;
                    a1[0] = (double)r.z;
                    a1[AS - 1] = (double)r.w;
                } else {
                    const typename Vec2<T>::type yv = yy2[i];
                    y0 = (double)yv.x;
                    y1 = (double)yv.y;
#pragma unroll
                    for (int t = 0; t < AS; ++t) a1[t] = (double)A[(size_t)i * AS + t];
                }
                const double dprev = grad_factor<M>(a1, x, p_slot);
                ag += dloss_dev(LOSS, y0, y1) * dprev;
                ah += dprev * dprev;
            }
            // the slot's 4 lanes are one DPP quad: quad_perm moves (a VALU operand modifier)
            // instead of ds_bpermute shuffles (two LDS-crossbar round trips on the step's
            // critical path); same pairing, same sums
            ag += dpp_move_d<0xB1, 0xf>(0.0, ag);  // quad_perm:[1,0,3,2]
            ah += dpp_move_d<0xB1, 0xf>(0.0, ah);
            ag += dpp_move_d<0x4E, 0xf>(0.0, ag);  // quad_perm:[2,3,0,1]
            ah += dpp_move_d<0x4E, 0xf>(0.0, ah);
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
                if (worker) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    const double pq = pold_sched[c0 + q];
                    double lg = 0.0, lh = 0.0;
                    for (int e = le0 + wt; e < le1; e += 256) {
                        const int i = a.erow[e];
                        const double x = (double)eval[e];
                        double a1[Kind<M>::AS], y0, y1;
                        if constexpr (LDSR) {
                            y0 = (double)lds_r[i - row0];
                            y1 = (LR == 1) ? 0.0 : (lds_s[i - row0] ? 1.0 : -1.0);
#pragma unroll
                            for (int t = 0; t < Kind<M>::AS; ++t)
                                a1[t] = (double)lds_a[(i - row0) * Kind<M>::AS + t];
                        } else if constexpr (PK) {
                            const float4 r = rec[(size_t)i];
                            y0 = (double)r.x;
                            y1 = (double)r.y;
                            a1[0] = (double)r.z;
                            a1[Kind<M>::AS - 1] = (double)r.w;
                        } else {
                            const typename Vec2<T>::type yv = yy2[i];
                            y0 = (double)yv.x;
                            y1 = (double)yv.y;
#pragma unroll
                            for (int t = 0; t < Kind<M>::AS; ++t)
                                a1[t] = (double)A[(size_t)i * Kind<M>::AS + t];
                        }
                        const double dprev = grad_factor<M>(a1, x, pq);
                        lg += dloss_dev(LOSS, y0, y1) * dprev;
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
            if (worker && wt < qi) {
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
        if (worker) {
            PRB_WSTAMP(1)  // publish issue
            // (measured: raising the flag earlier -- right behind the granule stores, ~150 cycles
            // sooner -- costs 3 %: 271 -> 279 ms per epoch on config 2)
            if (tid == 64 + 255)  // last worker wave: this workgroup's partials are on their way
                __hip_atomic_store(sh_go, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (LDSR && b + 1 < a.nb) {
                // rows in LDS: the end-of-step barrier no longer drains vmcnt, so the
                // streaming prefetch of step b+1 is issued BEFORE the exchange -- it has the
                // whole sweep (which waits for the slowest workgroup anyway) to land
                prb_load_entries<T>(a, eval, ne0, ne1, sub, nxt, row0);
                if (slot < c2 - c1) p_next = pold_sched[c1 + slot];
            }
            {
                const bool ok =
                    prb_collect_quarter<2>(a, b, part, lane, ncols, sh_quart, kPrbParts);
                if (!ok) *sh_ok = 0;
            }
            PRB_WSTAMP(2)  // granule sweep until every workgroup's partials are in
            if (!LDSR && b + 1 < a.nb) {
                // prefetch (after the exchange: vmcnt retires in order, so streaming loads
                // issued earlier would delay every granule check): entries of step b+1
                // (bounds already in registers), bounds of b+2
                prb_load_entries<T>(a, eval, ne0, ne1, sub, nxt, row0);
                if (slot < c2 - c1) p_next = pold_sched[c1 + slot];
            }
        } else {
            // control wave and helpers: their parts of the sweep, once the own workers have
            // published (the other workgroups are at the same point of the step)
            unsigned spins = 0;
            while (__hip_atomic_load(sh_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                       b + 1 &&
                   ++spins < (1u << 24))
                __builtin_amdgcn_s_sleep(1);
            PRB_HSTAMP(1)  // helper wave 5: from the end of the previous step to "go"
            const bool ok = prb_collect_quarter<2>(a, b, part, lane, ncols, sh_quart, kPrbParts);
            if (!ok) *sh_ok = 0;
            PRB_HSTAMP(2)  // its part of the sweep
            if constexpr (CR) {
                if (control) {
                    if (lane < cp1 - cp0) {  // the published state of "my" conflict row
                        if (!prb_poll<CSN>(a, a.cslab + ((size_t)(b & 1) * 64 + lane) * CSS,
                                           prb_tag(b), cst))
                            *sh_ok = 0;
                    }
                    // tables of step b+1 (consumed at its start)
                    const PrbConf<T>* cfa = reinterpret_cast<const PrbConf<T>*>(a.cf);
                    if (lane < cp2 - cp1) cfn = cfa[cp1 + lane];
                    clqn = prb_u4{~0u, ~0u, ~0u, ~0u};
                    if (lane < c2 - c1) clqn = reinterpret_cast<const prb_u4*>(a.clist)[c1 + lane];
                }
            }
        }
        // B3: part sums in LDS.  Raw barrier: only LDS traffic must have landed; the
        // prefetch loads just issued stay in flight across it (a __syncthreads() would
        // add s_waitcnt vmcnt(0) and expose their HBM latency on every step).
        PRB_WSTAMP(3)  // prefetch issue
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!*sh_ok) break;
        PRB_STAMP(3)
        PRB_HSTAMP(7)  // helper: B3
        PRB_WSTAMP(4)  // B3
        if (control) {
            double tot[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                tot[v] = sh_quart[lane * 2 + v];
#pragma unroll
                for (int w = 1; w < kPrbParts; ++w) tot[v] += sh_quart[(w * 64 + lane) * 2 + v];
            }
            if constexpr (MG) {
                if (!prb_cross_gpu<2>(a, g, b, lane, tot)) *sh_ok = 0;
            }
            const bool valid = lane < ncols;
            bool serial = false;
            if constexpr (CR) serial = (cp1 - cp0) > 0;
            if (!serial) {
                const double res = pcd_chain_lanes<M>(REGC >= 0 ? REGC : reg, lane, ncols - 1, valid,
                                                      pl, tot[0], tot[1], lam, mu, beta, gamma, eta,
                                                      cache, sh_chain);
                const double dl = valid ? (pl - res) : 0.0;
                sh_delta[lane] = dl;
                sh_pold[lane] = pl;
                if (g == 0 && valid) {
                    ps[jl] = res;
                    viol_pos[c0 + lane] = fabs(dl);  // by position; folded into viol_col later
                }
            }
            if constexpr (CR) {
                if (serial) {
                    // Relaxed run.  A column's sums = the totals of the row blocks (conflict rows
                    // left out) + its conflict rows with the state the sequential sweep finds
                    // (pcd.py:97-135): the published state for the row's EARLIER column, that state
                    // after the earlier column's update for the LATER one.  Lane c works for
                    // conflict row c, lane q for column q.  The later columns' terms depend on the
                    // earlier columns' deltas, which depend (regularizer cache) on everything in
                    // front of them: evaluated in rounds -- all terms from the current deltas, then
                    // the whole chain (pcd_chain_lanes: exact for the sums it is given) -- until no
                    // delta changes a bit.  Column q only depends on columns < q, so round r fixes
                    // column r at the latest and the fixed point is the sequential result; the
                    // couplings are weak (one row in ~500, a cache term of 1e-7), 3-5 rounds.
                    const int nc = cp1 - cp0;
                    const int qa = cfc.qq & 0xff, qb = cfc.qq >> 8;
                    const double xa = (double)cfc.xa, xb = (double)cfc.xb;
                    sh_pold[lane] = pl;
                    sh_delta[lane] = 0.0;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    // of "my" conflict row: the earlier column's dA chain on the published state
                    // (da[t] = dA_t of pcd._grad_anova, da[0] = x) -- its term of that column's
                    // sums, and what its update will subtract from the row (pcd.py:124-133)
                    double da[AS + 1], pb = 0.0;
#pragma unroll
                    for (int t = 0; t <= AS; ++t) da[t] = 0.0;
                    if (lane < nc) {
                        const double pa = sh_pold[qa];
                        pb = sh_pold[qb];
                        da[0] = xa;
#pragma unroll
                        for (int t = 1; t <= AS; ++t) da[t] = xa * (cst[1 + t] - pa * da[t - 1]);
                        sh_ce[lane * 2] = dloss_dev(LOSS, cst[0], cst[1]) * da[AS];
                        sh_ce[lane * 2 + 1] = da[AS] * da[AS];
                    }
                    // the row as the earlier column leaves it for a given delta (stored as T)
                    auto after_a = [&](double Da, double* a1, double& y0n) __attribute__((always_inline)) {
#pragma unroll
                        for (int t = 0; t < AS; ++t) a1[t] = (double)(T)(cst[2 + t] - Da * da[t]);
                        y0n = (double)(T)(cst[0] - lam * Da * da[AS]);
                    };
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    // my column's conflict list (8 entries: index | role << 8, -1 = none)
                    int le[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const unsigned w = clq[t >> 1];
                        le[t] = (int)(short)((t & 1) ? (w >> 16) : (w & 0xffffu));
                    }
                    double g0 = tot[0], h0 = tot[1];
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (le[t] >= 0 && (le[t] >> 8) == 0) {
                            g0 += sh_ce[(le[t] & 0xff) * 2];
                            h0 += sh_ce[(le[t] & 0xff) * 2 + 1];
                        }
                    double cc0[M + 1];
#pragma unroll
                    for (int t = 0; t <= M; ++t) cc0[t] = cache[t];
                    double dl = 0.0, res = pl;
                    for (int round = 0; round <= ncols + 1; ++round) {
                        if (lane < nc) {  // the later column's term of "my" row
                            double a1[AS], y0n;
                            after_a(sh_delta[qa], a1, y0n);
                            const double dAb = grad_factor<M>(a1, xb, pb);
                            sh_cv[lane * 2] = dloss_dev(LOSS, y0n, cst[1]) * dAb;
                            sh_cv[lane * 2 + 1] = dAb * dAb;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        double gq = g0, hq = h0;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (le[t] >= 0 && (le[t] >> 8) != 0) {
                                gq += sh_cv[(le[t] & 0xff) * 2];
                                hq += sh_cv[(le[t] & 0xff) * 2 + 1];
                            }
#pragma unroll
                        for (int t = 0; t <= M; ++t) cache[t] = cc0[t];
                        res = pcd_chain_lanes<M>(reg, lane, ncols - 1, valid, pl, gq, hq, lam, mu,
                                                 beta, gamma, eta, cache, sh_chain);
                        const double dn = valid ? (pl - res) : 0.0;
                        // (a relative threshold instead of the bitwise test buys nothing: the
                        // rounds end with the DEPTH of the dependencies -- 2^-36 and bitwise both
                        // take 5.2 rounds on config 2, 2^-20 takes 4.2; tools/relax_probe.py)
                        const bool same = __double_as_longlong(dn) == __double_as_longlong(dl);
                        dl = dn;
                        sh_delta[lane] = dl;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        if (__ballot(!same) == 0ull) {
                            if (lane == 0 && g == a.G - 1) {
                                atomicAdd(&g_branch_count[BR_RELAX_STEPS], 1u);
                                atomicAdd(&g_branch_count[BR_RELAX_ROUNDS], (unsigned)(round + 1));
                            }
                            break;
                        }
                    }
                    if (g == 0 && valid) {
                        ps[jl] = res;
                        viol_pos[c0 + lane] = fabs(dl);
                    }
                    // the conflict rows' final state, by their owner: both updates in order
                    if (lane < nc && cfc.row / a.rows_per == g) {
                        const int i = cfc.row;
                        const double Db = sh_delta[qb];
                        double a1[AS], y0n;
                        after_a(sh_delta[qa], a1, y0n);
                        T a2[AS];
                        double db = xb;  // dA chain of the later column on the updated row
#pragma unroll
                        for (int t = 0; t < AS; ++t) {
                            const double dn = xb * (a1[t] - pb * db);
                            a2[t] = (T)(a1[t] - Db * db);
                            db = dn;
                        }
                        const T y0f = (T)(y0n - lam * Db * db);
                        if constexpr (LDSR) {
#pragma unroll
                            for (int t = 0; t < AS; ++t) lds_a[(i - row0) * AS + t] = a2[t];
                            lds_r[i - row0] = y0f;
                        } else if constexpr (PK) {
                            float4 o;
                            o.x = (float)y0f;
                            o.y = (float)cst[1];
                            o.z = (float)a2[0];
                            o.w = (float)a2[AS - 1];
                            rec[(size_t)i] = o;
                        } else {
#pragma unroll
                            for (int t = 0; t < AS; ++t) A[(size_t)i * AS + t] = a2[t];
                            yy[2 * (size_t)i] = y0f;
                        }
                    }
                }
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
                            if constexpr (LDSR) {
                                lds_a[i - row0] = (T)an;
                                lds_r[i - row0] = (T)yn;
                            } else {
                                A[i] = (T)an;
                                yy[2 * i] = (T)yn;
                            }
                        } else if constexpr (LDSR) {
                            const int il = (int)i - row0;
                            double dprev = x;
#pragma unroll
                            for (int t = 0; t < AS; ++t) {
                                const double a1 = av[u][t];
                                const double dcur = x * (a1 - p_old * dprev);
                                lds_a[il * AS + t] = (T)(a1 - upd * dprev);
                                dprev = dcur;
                            }
                            lds_r[il] = (T)(yh[u] - lam * upd * dlast[u]);
                        } else if constexpr (PK) {
                            const double d1 = x * (av[u][0] - p_old * x);
                            float4 o;
                            o.x = (float)(yh[u] - lam * upd * dlast[u]);
                            o.y = (float)yt[u];
                            o.z = (float)(av[u][0] - upd * x);
                            o.w = (float)(av[u][AS - 1] - upd * d1);
                            rec[i] = o;
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
                for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {
                    if constexpr (LDSR)
                        pcd_sync_entry_lds<T, M>(a.erow[e] - row0, (double)eval[e], p_old, upd,
                                                 lam, lds_a, lds_r);
                    else if constexpr (PK)
                        pcd_sync_entry_rec((size_t)a.erow[e], (double)eval[e], p_old, upd, lam, rec);
                    else
                        pcd_sync_entry<T, M>((size_t)a.erow[e], (double)eval[e], p_old, upd, lam,
                                             A, yy);
                }
            }
        }
        if (lmu != 0ull && worker) {  // long slots: every worker thread scatters
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1) {
                const int q = __builtin_ctzll(mm);
                const double upd = sh_delta[q];
                if (upd != 0.0) {
                    const double p_old = sh_pold[q];
                    const int le0 = spb[q], le1 = spb[q + 1];
                    for (int e = le0 + wt; e < le1; e += 256) {
                        if constexpr (LDSR)
                            pcd_sync_entry_lds<T, M>(a.erow[e] - row0, (double)eval[e], p_old,
                                                     upd, lam, lds_a, lds_r);
                        else if constexpr (PK)
                            pcd_sync_entry_rec((size_t)a.erow[e], (double)eval[e], p_old, upd, lam,
                                               rec);
                        else
                            pcd_sync_entry<T, M>((size_t)a.erow[e], (double)eval[e], p_old, upd,
                                                 lam, A, yy);
                    }
                }
            }
        }
        cur = nxt;
        p_slot = p_next;
        ne0 = n2e0;
        ne1 = n2e1;
        lm0 = lm1;
        lm1 = lm2;
        if constexpr (CR) {
            if (control) {
                cp0 = cp1;
                cp1 = cp2;
                cp2 = cp3;
                cp3 = a.cf_ptr[min(b + 4, a.nb)];
                cfc = cfn;
                clq = clqn;
            }
        }
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        PRB_WSTAMP(6)  // scatter issue
        // B5: rows move between slots from step to step.  With the rows in LDS only LDS
        // traffic has to land (the prefetch loads of the next step stay in flight).
        if constexpr (LDSR)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            __syncthreads();
        PRB_STAMP(6)
        if constexpr (STAMP) {
            if (tid == 320) tprev = clock64();  // helper: the rest of the step is not its time
        }
        PRB_WSTAMP(7)  // B5 (stores acknowledged)
    }
#undef PRB_STAMP
#undef PRB_HSTAMP
#undef PRB_WSTAMP
    if constexpr (LDSR) {  // write the row block back (LR == 1: yhat = r + y)
        __syncthreads();
        const int nr = min(a.rows_per, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPrbThreads) {
            const size_t i = (size_t)(row0 + il);
#pragma unroll
            for (int t = 0; t < AS; ++t) A[i * AS + t] = lds_a[il * AS + t];
            if constexpr (LR == 1)
                yy[2 * i] = (T)((double)lds_r[il] + (double)yy[2 * i + 1]);
            else
                yy[2 * i] = lds_r[il];
        }
    }
    if (STAMP && a.stamps != nullptr && (tid == 0 || tid == 64)) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (tid == 64 || (q != 1 && q != 2 && q != 7))
                a.stamps[(size_t)g * 16 + (tid == 64 ? 8 : 0) + q] = acc[q];
    }
    if (STAMP && a.stamps != nullptr && tid == 320) {  // helper wave 5 in the control block's gaps
        a.stamps[(size_t)g * 16 + 1] = acc[1];
        a.stamps[(size_t)g * 16 + 2] = acc[2];
        a.stamps[(size_t)g * 16 + 7] = acc[7];
    }
}

// cd_linear._cd_linear_epoch (optimizer/cd_linear.py:8-33) as one persistent launch:
// same row-block ownership, entry stream and tagged-granule exchange as pcd_prb_kernel,
// one value per slot; the update has no regularizer, so the control wave's "chain" is
// lane-parallel.  w_sched / cn_sched are w and col_norm_sq in visiting order (w as of the
// epoch start: workgroup 0 writes the new w[j] while others may still read the old one).
// LR as in pcd_prb_kernel: 1 = residual word per row in LDS (squared loss), 2 = prediction
// word + label sign (+-1 targets); float storage.  4-5 bytes per row: it always fits.
// CR = true: relaxed runs as in pcd_prb_kernel -- the conflict rows' state is (yhat or residual, y),
// a row's later column sees yhat - u_a x_a (cd_linear.py:28-31); there is no regularizer chain, so
// the rounds stop after the depth of the row dependencies.
template <typename T, int LOSS, int LR, bool MG = false, bool CR = false>
__global__ __launch_bounds__(kPrbThreads) void lin_prb_kernel(
    PrbArgs a, const T* __restrict__ eval, T* __restrict__ yy,
    const double* __restrict__ w_sched, const double* __restrict__ cn_sched,
    double* __restrict__ w, double alpha, double mu, double* __restrict__ viol_pos) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double* sh_delta = dyn_lds + 128;  // [64]
    double* sh_quart = dyn_lds + 1536;  // [8][64][2]
    int* sh_ok = reinterpret_cast<int*>(dyn_lds + 768);
    int* sh_go = reinterpret_cast<int*>(dyn_lds + 770);
    const typename Vec2<T>::type* yy2 = reinterpret_cast<const typename Vec2<T>::type*>(yy);
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool control = wave == 0;
    const bool worker = wave >= 1 && wave <= 4;
    const int part = worker ? wave - 1 : (control ? 4 : wave);
    const int wt = tid - 64;
    const int slot = worker ? (wt >> 2) : 64, sub = wt & 3;
    double* sh_long = dyn_lds + 1024;  // [64][4][2]
    static_assert(LR == 0 || (sizeof(T) == 4 && (LR == 2 || LOSS == LOSS_SQUARED)),
                  "LDS-resident rows: float storage");
    const int row0 = LR ? g * a.rows_per : 0;
    static_assert(!CR || !MG, "relaxed runs: single GPU");
    double* sh_ce = dyn_lds + kPrbLdsFixed;  // CR: [64] term of a conflict row's EARLIER column
    double* sh_cv = sh_ce + 64;              // CR: [64] ... of its LATER column (per round)
    T* lds_r = reinterpret_cast<T*>(dyn_lds + kPrbLdsFixed + (CR ? kPrbLdsCR : 0));  // [rows_per] residual or yhat
    unsigned char* lds_s = reinterpret_cast<unsigned char*>(lds_r + a.rows_per);  // y > 0
    if constexpr (LR != 0) {
        const int nr = min(a.rows_per, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPrbThreads) {
            const typename Vec2<T>::type yv = yy2[(size_t)(row0 + il)];
            if constexpr (LR == 1) {
                lds_r[il] = (T)((double)yv.x - (double)yv.y);
            } else {
                lds_r[il] = yv.x;
                lds_s[il] = yv.y > (T)0 ? 1 : 0;
            }
        }
        __syncthreads();
    }
    // (yhat, y) of a row as the loss sees them
    auto row_state = [&](int i, double& y0, double& y1) {
        if constexpr (LR != 0) {
            y0 = (double)lds_r[i - row0];
            y1 = (LR == 1) ? 0.0 : (lds_s[i - row0] ? 1.0 : -1.0);
        } else {
            const typename Vec2<T>::type yv = yy2[(size_t)i];
            y0 = (double)yv.x;
            y1 = (double)yv.y;
        }
    };
    auto row_update = [&](int i, double yold, double dec) {  // yhat_i (or r_i) -= dec
        if constexpr (LR != 0) lds_r[i - row0] = (T)(yold - dec);
        else yy[2 * (size_t)i] = (T)(yold - dec);
    };
    PrbEntries<T> cur, nxt;
    int c0 = a.bptr[0], c1 = a.bptr[1];
    int c2 = (a.nb > 1) ? a.bptr[2] : c1;
    int c3 = (a.nb > 2) ? a.bptr[3] : c2;
    unsigned long long lm0 = 0ull, lm1 = 0ull;
    {
        int e0, e1;
        prb_load_sp(a, g, 0, slot, c1 - c0, e0, e1, lm0);
        prb_load_entries<T>(a, eval, e0, e1, sub, cur, row0);
    }
    int ne0 = 0, ne1 = 0;
    if (a.nb > 1) prb_load_sp(a, g, 1, slot, c2 - c1, ne0, ne1, lm1);
    if (tid == 0) {
        *sh_ok = 1;
        *sh_go = 0;
    }
    // CR: see pcd_prb_kernel
    int cp0 = 0, cp1 = 0, cp2 = 0, cp3 = 0;
    PrbConf<T> cfc, cfn;
    cfc.row = cfn.row = 0;
    cfc.qq = cfn.qq = 0;
    cfc.xa = cfc.xb = cfn.xa = cfn.xb = (T)0;
    prb_u4 clq = {~0u, ~0u, ~0u, ~0u}, clqn = {~0u, ~0u, ~0u, ~0u};
    if constexpr (CR) {
        if (control) {
            const PrbConf<T>* cfa = reinterpret_cast<const PrbConf<T>*>(a.cf);
            cp0 = a.cf_ptr[0];
            cp1 = a.cf_ptr[min(1, a.nb)];
            cp2 = a.cf_ptr[min(2, a.nb)];
            cp3 = a.cf_ptr[min(3, a.nb)];
            if (lane < cp1 - cp0) cfc = cfa[cp0 + lane];
            if (lane < c1 - c0) clq = reinterpret_cast<const prb_u4*>(a.clist)[c0 + lane];
        }
    }
    for (int b = 0; b < a.nb; ++b) {
        const int ncols = c1 - c0;
        const int c4 = (b + 4 <= a.nb) ? a.bptr[b + 4] : c3;
        double yh[PRB_PF];
        double wl = 0.0, cnl = 0.0;
        int jl = 0;
        double cst[2] = {0.0, 0.0};  // CR, control wave: (yhat or residual, y) of conflict row `lane`
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
                if (worker) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    double lg = 0.0;
                    for (int e = le0 + wt; e < le1; e += 256) {
                        double y0, y1;
                        row_state(a.erow[e], y0, y1);
                        lg += dloss_dev(LOSS, y0, y1) * (double)eval[e];
                    }
                    lg = wave_sum(lg);
                    if (lane == 0) sh_long[(qi * 4 + (wave - 1)) * 2] = lg;
                }
            }
            __syncthreads();
            if (worker && wt < qi) {
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
            if constexpr (CR) {  // the owners publish the conflict rows' state
                const int nc = cp1 - cp0, ncw = max(nc, cp3 - cp2);
                const unsigned long long ctag = prb_tag(b);
                double* cs = a.cslab + ((size_t)(b & 1) * 64 + lane) * 8;
                if (lane < nc) {
                    if (cfc.row / a.rows_per == g) {
                        double y0, y1;
                        row_state(cfc.row, y0, y1);
                        prb_store_granule(cs, y0, ctag);
                        prb_store_granule(cs + 1, y1, ctag);
                    }
                } else if (lane < ncw && (lane % a.G) == g) {
                    prb_store_granule(cs, 0.0, ctag);
                    prb_store_granule(cs + 1, 0.0, ctag);
                }
            }
        } else if (worker) {
            double yt[PRB_PF];
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) row_state(cur.row[u], yh[u], yt[u]);
            double ag = 0.0;
#pragma unroll
            for (int u = 0; u < PRB_PF; ++u) {
                const bool v = cur.e0 + sub + 4 * u < cur.e1;
                ag += v ? dloss_dev(LOSS, yh[u], yt[u]) * (double)cur.x[u] : 0.0;
            }
            for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {
                double y0, y1;
                row_state(a.erow[e], y0, y1);
                ag += dloss_dev(LOSS, y0, y1) * (double)eval[e];
            }
            ag += dpp_move_d<0xB1, 0xf>(0.0, ag);  // quad_perm:[1,0,3,2] (see pcd_prb_kernel)
            ag += dpp_move_d<0x4E, 0xf>(0.0, ag);  // quad_perm:[2,3,0,1]
            if (sub == 0 && !((lm0 >> slot) & 1ull)) {
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + slot) * 2;
                prb_store_granule(sl, ag, prb_tag(b));
            }
            if (tid == 64 + 255)
                __hip_atomic_store(sh_go, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
            {
                const bool ok =
                    prb_collect_quarter<1>(a, b, part, lane, ncols, sh_quart, kPrbParts);
                if (!ok) *sh_ok = 0;
            }
            if (b + 1 < a.nb) {
                prb_load_entries<T>(a, eval, ne0, ne1, sub, nxt, row0);
            }
        } else if (!control) {  // helpers: their parts, once the workers have published
            if (b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
            unsigned spins = 0;
            while (__hip_atomic_load(sh_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                       b + 1 &&
                   ++spins < (1u << 24))
                __builtin_amdgcn_s_sleep(1);
            const bool ok = prb_collect_quarter<1>(a, b, part, lane, ncols, sh_quart, kPrbParts);
            if (!ok) *sh_ok = 0;
        }
        if (control) {
            if (b + 2 < a.nb) prb_load_sp(a, g, b + 2, slot, c3 - c2, n2e0, n2e1, lm2);
            {
                unsigned spins = 0;
                while (__hip_atomic_load(sh_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                           b + 1 &&
                       ++spins < (1u << 24))
                    __builtin_amdgcn_s_sleep(1);
                const bool ok =
                    prb_collect_quarter<1>(a, b, part, lane, ncols, sh_quart, kPrbParts);
                if (!ok) *sh_ok = 0;
            }
            if constexpr (CR) {
                if (lane < cp1 - cp0) {
                    if (!prb_poll<2>(a, a.cslab + ((size_t)(b & 1) * 64 + lane) * 8, prb_tag(b), cst))
                        *sh_ok = 0;
                }
                const PrbConf<T>* cfa = reinterpret_cast<const PrbConf<T>*>(a.cf);
                if (lane < cp2 - cp1) cfn = cfa[cp1 + lane];
                clqn = prb_u4{~0u, ~0u, ~0u, ~0u};
                if (lane < c2 - c1) clqn = reinterpret_cast<const prb_u4*>(a.clist)[c1 + lane];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // part sums in LDS
        if (!*sh_ok) break;
        if (control) {
            double tot = sh_quart[lane * 2];
#pragma unroll
            for (int w = 1; w < kPrbParts; ++w) tot += sh_quart[(w * 64 + lane) * 2];
            if constexpr (MG) {
                if (!prb_cross_gpu<1>(a, g, b, lane, &tot)) *sh_ok = 0;
            }
            const bool valid = lane < ncols;
            bool relaxed_step = false;
            if constexpr (CR) relaxed_step = (cp1 - cp0) > 0;
            const double inv = mu * cnl + alpha;
            double upd = 0.0;
            if (!relaxed_step) {
                upd = tot;  // cd_linear.py:19-24
                upd += alpha * wl;
                upd /= inv;
                if (!valid) upd = 0.0;
                sh_delta[lane] = upd;
            }
            if constexpr (CR) {
                if (relaxed_step) {
                    // the conflict rows' terms in rounds (lane c <-> conflict row c, lane q <->
                    // column q): a row's later column sees the prediction after the earlier
                    // column's update (stored as T); no chain couples the columns, so the rounds
                    // end with the depth of the row dependencies
                    const int nc = cp1 - cp0;
                    const int qa = cfc.qq & 0xff, qb = cfc.qq >> 8;
                    const double xa = (double)cfc.xa, xb = (double)cfc.xb;
                    sh_delta[lane] = 0.0;
                    if (lane < nc) sh_ce[lane] = dloss_dev(LOSS, cst[0], cst[1]) * xa;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    int le[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const unsigned wv = clq[t >> 1];
                        le[t] = (int)(short)((t & 1) ? (wv >> 16) : (wv & 0xffffu));
                    }
                    double g0 = tot;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (le[t] >= 0 && (le[t] >> 8) == 0) g0 += sh_ce[le[t] & 0xff];
                    for (int round = 0; round <= ncols + 1; ++round) {
                        if (lane < nc) {
                            const double y0n = (double)(T)(cst[0] - sh_delta[qa] * xa);
                            sh_cv[lane] = dloss_dev(LOSS, y0n, cst[1]) * xb;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        double gq = g0;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (le[t] >= 0 && (le[t] >> 8) != 0) gq += sh_cv[le[t] & 0xff];
                        double un = gq;
                        un += alpha * wl;
                        un /= inv;
                        if (!valid) un = 0.0;
                        const bool same = __double_as_longlong(un) == __double_as_longlong(upd);
                        upd = un;
                        sh_delta[lane] = upd;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        if (__ballot(!same) == 0ull) break;
                    }
                    if (lane < nc && cfc.row / a.rows_per == g) {  // the row's final prediction
                        const double y0n = (double)(T)(cst[0] - sh_delta[qa] * xa);
                        row_update(cfc.row, y0n, sh_delta[qb] * xb);
                    }
                }
            }
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
                        row_update(cur.row[u], yh[u], upd * (double)cur.x[u]);
                for (int e = cur.e0 + sub + 4 * PRB_PF; e < cur.e1; e += 4) {
                    double y0, y1;
                    row_state(a.erow[e], y0, y1);
                    row_update(a.erow[e], y0, upd * (double)eval[e]);
                }
            }
        }
        if (lmu != 0ull && worker) {
            const int32_t* spb = a.sp + ((size_t)g * a.nb + b) * 65;
            for (unsigned long long mm = lmu; mm != 0ull; mm &= mm - 1) {
                const int q = __builtin_ctzll(mm);
                const double upd = sh_delta[q];
                if (upd != 0.0) {
                    const int le0 = spb[q], le1 = spb[q + 1];
                    for (int e = le0 + wt; e < le1; e += 256) {
                        double y0, y1;
                        row_state(a.erow[e], y0, y1);
                        row_update(a.erow[e], y0, upd * (double)eval[e]);
                    }
                }
            }
        }
        cur = nxt;
        ne0 = n2e0;
        ne1 = n2e1;
        lm0 = lm1;
        lm1 = lm2;
        if constexpr (CR) {
            if (control) {
                cp0 = cp1;
                cp1 = cp2;
                cp2 = cp3;
                cp3 = a.cf_ptr[min(b + 4, a.nb)];
                cfc = cfn;
                clq = clqn;
            }
        }
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = c4;
        if constexpr (LR != 0)  // rows in LDS: only LDS traffic has to land
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            __syncthreads();
    }
    if constexpr (LR != 0) {  // write the row block back (LR == 1: yhat = r + y)
        __syncthreads();
        const int nr = min(a.rows_per, a.n_rows - row0);
        for (int il = tid; il < nr; il += kPrbThreads) {
            const size_t i = (size_t)(row0 + il);
            if constexpr (LR == 1)
                yy[2 * i] = (T)((double)lds_r[il] + (double)yy[2 * i + 1]);
            else
                yy[2 * i] = lds_r[il];
        }
    }
}

// ---- diagnostic: latency of one cross-workgroup hand-off -----------------------------
// Workgroups 0 and `partner` (all others leave at once) play ping-pong on two 8-byte words
// with the same agent-scope (sc1) stores and loads as the granule exchange: a round trip is
// two hops "store becomes visible to the other CU's load".  Workgroups are dealt to the 8
// XCDs round-robin, so partner = 1 crosses XCDs and partner = 8 stays inside XCD 0; the
// XCC_ID register of both players is returned for confirmation.  Bounded spins.
static __global__ __launch_bounds__(kWave) void hop_pingpong_kernel(unsigned long long* words,
                                                             int rounds, int partner,
                                                             int* info /* [4] */) {
    const int b = blockIdx.x;
    if ((b != 0 && b != partner) || threadIdx.x != 0) return;
    const int me = (b == 0) ? 0 : 1;
    info[me] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID, 4 bits
    unsigned long long* mine = words + (me ? 16 : 0);   // separate cache lines
    unsigned long long* theirs = words + (me ? 0 : 16);
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0)
            __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
               (unsigned long long)r) {
            if (++spins > (1u << 22)) {
                info[2] = r;  // gave up (partner never ran): reported as a failure
                return;
            }
        }
        if (me == 1)
            __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Poll NV adjacent granules until all carry `tag`; bounded, sets the abort word on time-out.
template <int NV>
__device__ __forceinline__ bool prb_poll(const PrbArgs& a, const double* p,
                                         unsigned long long tag, double* out) {
    unsigned long long t[NV];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            t[v] = prb_load_granule(p + v);
            all = all && ((t[v] & 3ull) == tag);
        }
        if (all) break;
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                spins > a.spin_max) {
                __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) out[v] = __longlong_as_double((long long)(t[v] & ~3ull));
    return true;
}

// ---- diagnostic: cost of the bare all-to-all exchange ---------------------------------
// G workgroups repeat `rounds` times exactly what one step of the persistent pass does for
// its exchange and nothing else: the 4 worker waves publish 64 tagged granule pairs into the
// workgroup's slab (parity double-buffered), the slabs of all G workgroups are swept until
// every tag is current, then a workgroup barrier.  readers_mod = 1: the 4 worker waves sweep
// a quarter each (two rounds of loads per lane at G = 64; the engine before v6);
// readers_mod > 1: only workgroups with g % readers_mod == 0 sweep, the others read one pair
// per slot from that leader, which republishes its totals (two-level variant);
// readers_mod = -8: eight waves sweep an eighth each, control wave and helpers gated on the
// publish flag (the engine since v6); readers_mod = -12: twelve waves, the eight sweepers
// all store-free.
// xcd_mode = 1: the grid is 8x oversized, only workgroups that landed on XCD `xcd_pick` stay and
// number themselves through a counter (abort_flag[1]); their publishes are PLAIN stores (the
// line stays in that XCD's L2, MI355X_MICROARCH.md store table) and the sc1 loads of the
// sweep are served by that L2.
static __global__ __launch_bounds__(768) void exchange_probe_kernel(PrbArgs a, int rounds,
                                                                     int ncols, int readers_mod,
                                                                     int xcd_mode) {
    __shared__ double quart[8 * 64 * 2];
    __shared__ int ok_flag;
    __shared__ int go_flag;
    __shared__ int my_rank;
    int g = blockIdx.x;
    if (xcd_mode) {
        if (threadIdx.x == 0) {
            const int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);
            int r = -1;
            if (xcc == xcd_mode - 1) r = (int)atomicAdd(a.abort_flag + 1, 1u);
            my_rank = (r >= 0 && r < a.G) ? r : -1;
        }
        __syncthreads();
        g = my_rank;
        if (g < 0) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool control = wave == 0;
    const int wt = tid - 64, slot = (wave >= 1 && wave <= 4) ? (wt >> 2) : 64, sub = wt & 3;
    const bool eight = readers_mod < 0;  // 8 sweeping waves, one round each
    if (eight) readers_mod = 1;
    const bool sweeps = (g % readers_mod) == 0;
    if (tid == 0) {
        ok_flag = 1;
        go_flag = 0;
    }
    __syncthreads();
    for (int b = 0; b < rounds; ++b) {
        const unsigned long long tag = prb_tag(b);
        if (eight) {
            const bool worker = wave >= 1 && wave <= 4;
            if (worker) {
                if (sub == 0) {
                    double* sl =
                        a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + slot) * 2;
                    if (xcd_mode) {
                        prb_store_granule_l2(sl, 1.0 + b, tag);
                        prb_store_granule_l2(sl + 1, 2.0 + g, tag);
                    } else {
                        prb_store_granule(sl, 1.0 + b, tag);
                        prb_store_granule(sl + 1, 2.0 + g, tag);
                    }
                }
                if (tid == 64 + 255)
                    __hip_atomic_store(&go_flag, b + 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                unsigned spins = 0;
                while (__hip_atomic_load(&go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                           b + 1 &&
                       ++spins < (1u << 24))
                    __builtin_amdgcn_s_sleep(1);
            }
            if (blockDim.x == 768) {  // 12 waves: the 8 sweepers are all store-free
                if (!worker) {
                    const int part12 = control ? 0 : wave - 4;  // waves 5..11 -> 1..7
                    if (!prb_collect_quarter<2>(a, b, part12, lane, ncols, quart, 8)) ok_flag = 0;
                }
            } else {
                const int part = worker ? wave - 1 : (control ? 4 : wave);  // 0..7
                if (!prb_collect_quarter<2>(a, b, part, lane, ncols, quart, 8)) ok_flag = 0;
            }
            __syncthreads();
            if (!ok_flag) break;
            continue;
        }
        if (!control && wave <= 4) {
            if (sub == 0) {
                double* sl = a.slab + (size_t)(b & 1) * a.G * 64 * 2 + ((size_t)g * 64 + slot) * 2;
                prb_store_granule(sl, 1.0 + b, tag);
                prb_store_granule(sl + 1, 2.0 + g, tag);
            }
            if (sweeps && !prb_collect_quarter<2>(a, b, wave - 1, lane, ncols, quart)) ok_flag = 0;
        }
        __syncthreads();
        if (!ok_flag) break;
        if (readers_mod > 1 && control) {
            double* tb = a.slab + (size_t)2 * a.G * 64 * 2 +
                         ((size_t)(b & 1) * a.G + (g / readers_mod)) * 64 * 2 + (size_t)lane * 2;
            if (sweeps) {
                prb_store_granule(tb, quart[lane * 2], tag);
                prb_store_granule(tb + 1, quart[lane * 2 + 1], tag);
            } else {
                double t2[2];
                bool okp = true;
                if (lane < ncols) okp = prb_poll<2>(a, tb, tag, t2);
                if (!__all(okp)) ok_flag = 0;
            }
        }
        __syncthreads();
        if (!ok_flag) break;
    }
}

// ---- connect-time handshake of the in-kernel cross-GPU exchange ------------------------
// What the persistent passes rely on, checked once per spfm_peer_connect: a system-scope store
// into a peer-mapped slab becomes visible to a kernel that is ALREADY RUNNING on the owning GPU
// and polls with system-scope loads (fine-grained memory; coarse-grained memory only promises
// visibility at kernel boundaries).  Lane r stores `word` into rank r's probe region at this
// rank's position and polls the own region at position r until rank r's word arrives; bounded
// by wall-clock ticks (wall_clock64: the 100 MHz constant counter), since the ranks enter this kernel milliseconds
// apart.  ok[0] = 1 when every rank's word arrived.
static __global__ __launch_bounds__(kWave) void peer_probe_kernel(double* const* slabs, size_t off,
                                                           int n_ranks, int rank,
                                                           unsigned long long word,
                                                           unsigned long long max_ticks, int* ok) {
    const int r = threadIdx.x;
    bool good = true;
    if (r < n_ranks) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(slabs[r] + off) + rank, word,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* mine =
            reinterpret_cast<const unsigned long long*>(slabs[rank] + off) + r;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != word) {
            if (wall_clock64() - t0 > max_ticks) {
                good = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const bool all = __all(good);
    if (threadIdx.x == 0) ok[0] = all ? 1 : 0;
}

// ---- diagnostic: the entry stream of the persistent pass and nothing else --------------
// What a worker thread of pcd_prb_kernel loads per step (slot bounds, then its 4-byte row ids
// and values: lane = 4 lanes per slot, PRB_PF entries each, the reload loop for longer slots),
// folded into a checksum so that nothing is optimised away.  Counter calibration: its
// FETCH_SIZE under rocprofv3 against the known byte count.
template <typename T>
__global__ __launch_bounds__(kPrbThreads) void prb_stream_probe_kernel(PrbArgs a,
                                                                      const T* __restrict__ eval,
                                                                      double* __restrict__ sink) {
    const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const bool worker = wave >= 1 && wave <= 4;
    const int wt = tid - 64, slot = worker ? (wt >> 2) : 64, sub = wt & 3;
    double acc = 0.0;
    for (int b = 0; b < a.nb; ++b) {
        const int ncols = a.bptr[b + 1] - a.bptr[b];
        int e0, e1;
        unsigned long long lm;
        prb_load_sp(a, g, b, slot, ncols, e0, e1, lm);
        for (int e = e0 + sub; e < e1; e += 4) acc += (double)a.erow[e] + (double)eval[e];
    }
    sink[(size_t)g * kPrbThreads + tid] = acc;
}

// Diagnostic: the SCATTER side of a persistent pass with its rows in global memory -- every
// worker thread stores one BYTES-wide record per entry at its row (records[row]), the store
// pattern of pcd_prb_kernel's packed-record scatter (degree 3: 16 bytes per row; plain arrays: 4
// or 8), and nothing else is written.  Counter calibration: its WRITE_SIZE under rocprofv3
// against nnz * BYTES.
template <int BYTES>
__global__ __launch_bounds__(kPrbThreads) void prb_write_probe_kernel(PrbArgs a,
                                                                     float* __restrict__ records) {
    static_assert(BYTES == 4 || BYTES == 8 || BYTES == 16, "record width");
    const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const bool worker = wave >= 1 && wave <= 4;
    const int wt = tid - 64, slot = worker ? (wt >> 2) : 64, sub = wt & 3;
    for (int b = 0; b < a.nb; ++b) {
        const int ncols = a.bptr[b + 1] - a.bptr[b];
        int e0, e1;
        unsigned long long lm;
        prb_load_sp(a, g, b, slot, ncols, e0, e1, lm);
        for (int e = e0 + sub; e < e1; e += 4) {
            const size_t row = (size_t)a.erow[e];
            const float v = (float)(b + e);
            if constexpr (BYTES == 4) records[row] = v;
            else if constexpr (BYTES == 8) reinterpret_cast<float2*>(records)[row] = make_float2(v, v);
            else reinterpret_cast<float4*>(records)[row] = make_float4(v, v, v, v);
        }
    }
}

// out[pos] = v[desc[pos].j]
static __global__ void gather_sched_kernel(int d, const ColDesc* __restrict__ desc,
                                    const double* __restrict__ v, double* __restrict__ out) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < d) out[pos] = v[desc[pos].j];
}

// viol_col[desc[pos].j] += viol_pos[pos]   (sum_viol bookkeeping of the persistent pass)
static __global__ void fold_viol_kernel(int d, const ColDesc* __restrict__ desc,
                                 const double* __restrict__ viol_pos,
                                 double* __restrict__ viol_col) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < d) viol_col[desc[pos].j] += viol_pos[pos];
}

// in visiting order: out[pos] = P[s, desc[pos].j]
static __global__ void snapshot_row_kernel(const Ctl* __restrict__ ctl, const double* __restrict__ P,
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


}  // namespace spfm
