// spfm_common.hip.h -- constants, device control structures, losses, reductions
// Part of the gfx950 device code of the sparse-FM proximal CD core; see
// spfm_kernels.hip.h for the execution model and DESIGN.md section 3.
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

// Diagnostic: how often the device chains took one of the reference's "numerical error"
// branches (a handful of atomic adds on paths that are rare by construction).  Read and reset
// through spfm_debug_branch_counts; used by the tests that force those branches.
enum {
    BR_OMEGATI_CLIP = 0,     // omegati.py:97-98   _dcache[t] clipped at 0
    BR_OMEGACS_DCACHE = 1,   // omegacs.py:90-96   negative _dcache: recompute with degree-1
    BR_OMEGACS_CACHE = 2,    // omegacs.py:75-76   negative _cache after the update: recompute
    BR_SQL21_RESUM = 3,      // squaredl21.py:48-49 _cache < _norms[j]: re-sum the norms
    BR_RELAX_STEPS = 4,      // relaxed runs: merged steps that had conflict rows ...
    BR_RELAX_ROUNDS = 5,     // ... and the rounds their chains took (diagnostic, not a branch)
    BR_COUNT = 8
};
// (one copy per translation unit: spfm_engine.hip.h SPFM_DEFINE_BRANCH_COUNTS)
static __device__ unsigned g_branch_count[BR_COUNT];
// Counted once per step that took the branch.  REDUNDANT = true: the chain runs in every
// workgroup of the launch (the persistent passes; pcd_chain_sync_kernel, up to the workgroup's own
// column) -- the LAST workgroup of the grid runs the step's whole chain exactly once and counts;
// false: one workgroup runs the chain (whichever it is), it counts.
template <bool REDUNDANT = true>
__device__ __forceinline__ void count_branch(int which, int lane) {
    if (lane == 0 && (!REDUNDANT || blockIdx.x == gridDim.x - 1))
        atomicAdd(&g_branch_count[which], 1u);
}

// ------------------------------------------------------------------ helpers

// 1/d to about one ulp: v_rcp_f64 and two Newton steps -- the core of the compiler's IEEE
// division without its operand scaling and final fix-up (11-13 instructions per quotient).
// Used where a lone wave's instruction count is the critical path of a dependent step and two
// quotients share a denominator (the step-size block of the chains): n * recip_nr(d) differs
// from n / d by at most ~1.5 ulp; d must be a normal number of moderate exponent (here
// mu*h + beta and 1 + 2*strength).
__device__ __forceinline__ double recip_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}

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

static __global__ void begin_pass_kernel(Ctl* ctl, const int32_t* comp_order, const double* lams) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int s = comp_order[ctl->pass];
        ctl->s = s;
        ctl->lam = lams[s];
        ctl->pass += 1;
    }
}

// host-stepped epochs (user-defined regularizer objects): the caller computed the step's new
// coordinates; pnew_delta[q] holds p_new on entry and p_old - p_new on return (what
// pcd_sync_kernel consumes), P[s, j] and sum_viol are updated (pcd.py:119-121)
static __global__ void host_apply_pcd_kernel(const Ctl* __restrict__ ctl, const ColDesc* __restrict__ desc,
                                      int ncols, double* __restrict__ P, int d,
                                      const double* __restrict__ pold,
                                      double* __restrict__ pnew_delta,
                                      double* __restrict__ viol_col) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    const double pn = pnew_delta[q], dl = pold[q] - pn;
    const int j = desc[q].j;
    P[(size_t)ctl->s * d + j] = pn;
    pnew_delta[q] = dl;
    viol_col[j] += fabs(dl);
}

// undo of begin_pass_kernel's counter step for a pass that was announced but not launched
static __global__ void unbegin_pass_kernel(Ctl* ctl) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctl->pass -= 1;
}

// sum viol_col[0..d) -> out[0]  (one workgroup, fixed order => deterministic)
static __global__ __launch_bounds__(kBlock) void reduce_sum_kernel(const double* __restrict__ v, int n,
                                                             double* __restrict__ out) {
    __shared__ double red[16];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += v[i];
    block_sum2(a, b, red);
    if (threadIdx.x == 0) out[0] = a;
}

// first stage of a long deterministic sum: out[block] = sum of the block's contiguous
// slice of v (fixed slicing and fixed order inside => reproducible)
static __global__ __launch_bounds__(kBlock) void reduce_partial_kernel(const double* __restrict__ v,
                                                                 int64_t n,
                                                                 double* __restrict__ out) {
    __shared__ double red[16];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    double a = 0, b = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) a += v[i];
    block_sum2(a, b, red);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

}  // namespace spfm
