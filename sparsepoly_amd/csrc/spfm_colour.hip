// spfm_colour.hip -- the first-fit colouring of the column conflict graph on the device
// (the schedule product; host form: spfm_schedule.cpp
// schedule_colored_parallel, whose rounds this file restates for the GPU).
//
// Two columns conflict when they share a row.  The sequential first fit visits the columns in
// the given order and gives each the lowest colour that none of its rows holds yet and whose class
// is not full (max_batch columns).  Rounds of kRound columns, exactly as on the host:
//   (1) snapshot  one workgroup per column of the round collects the colours present in its rows
//                 as of the START of the round (per-row colour lists, LDS bitmap of 4096 colours);
//   (2) conflicts which columns of the round share a row: (1) also enters every column into its
//                 rows' lists of round members; a second pass over the same rows pairs up the
//                 members of every row (a row is touched by more than one column of a round in
//                 under 1 % of the cases; merging the 32 k pairs' row lists would cost more than
//                 the whole colouring);
//   (3) resolve   ONE workgroup walks the round in order: column q takes the lowest colour outside
//                 its snapshot set, not full, and not held by an earlier column of the round that
//                 shares a row with it -- what the sequential loop would have chosen;
//   (4) commit    the columns append their colour to their rows' lists.
// All launches of all rounds are queued on the stream without a host round trip; the result is
// the colour of every visiting position.  More than 4096 colours: *overflow, the host form takes
// over.  Identical to the host colouring for every input (tests/test_hip_colour.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace spfm {

constexpr int kRound = 256;        // columns per round = threads of the resolving workgroup
constexpr int kColourWords = 64;   // 4096 colours

__global__ __launch_bounds__(256) void colour_snapshot_kernel(
    int nq, int head, const int32_t* __restrict__ order, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const int64_t* __restrict__ rptr,
    const int32_t* __restrict__ rcnt, const int32_t* __restrict__ rcol,
    int32_t* __restrict__ rmcnt, int32_t* __restrict__ rmem,
    const unsigned long long* __restrict__ full, unsigned long long* __restrict__ F) {
    __shared__ unsigned long long u[kColourWords];
    const int q = blockIdx.x;
    if (q >= nq) return;
    if (threadIdx.x < kColourWords) u[threadIdx.x] = full[threadIdx.x];
    __syncthreads();
    const int32_t j = order[head + q];
    const int64_t cb = cptr[j], ce = cptr[j + 1];
    for (int64_t ii = cb + threadIdx.x; ii < ce; ii += blockDim.x) {
        const int32_t i = cidx[ii];
        const int32_t cnt = rcnt[i];
        const int64_t rb = rptr[i];
        rmem[rb + atomicAdd(&rmcnt[i], 1)] = q;  // member of the round on this row
        const int32_t* rc = rcol + rb;
        for (int32_t t = 0; t < cnt; ++t) {
            const int32_t c = rc[t];
            atomicOr(&u[c >> 6], 1ull << (c & 63));
        }
    }
    __syncthreads();
    if (threadIdx.x < kColourWords) F[(size_t)q * kColourWords + threadIdx.x] = u[threadIdx.x];
}

// conf[q][p >> 6] bit (p & 63), p < q: the round's columns q and p share a row
__global__ __launch_bounds__(256) void colour_conflict_kernel(
    int nq, int head, const int32_t* __restrict__ order, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const int64_t* __restrict__ rptr,
    const int32_t* __restrict__ rmcnt, const int32_t* __restrict__ rmem,
    unsigned long long* __restrict__ conf) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int32_t j = order[head + q];
    const int64_t cb = cptr[j], ce = cptr[j + 1];
    for (int64_t ii = cb + threadIdx.x; ii < ce; ii += blockDim.x) {
        const int32_t i = cidx[ii];
        const int32_t cnt = rmcnt[i];
        if (cnt < 2) continue;
        const int32_t* rm = rmem + rptr[i];
        for (int32_t t = 0; t < cnt; ++t) {
            const int32_t p = rm[t];
            if (p < q) atomicOr(&conf[(size_t)q * (kRound / 64) + (p >> 6)], 1ull << (p & 63));
        }
    }
}

// state[0] = overflow flag, state[1] = classes in use.  The walk over the round is ONE WAVE: lane w
// holds word w of the 4096-colour bitmaps, so a column costs a few LDS round trips and a ballot
// and no workgroup barrier (a first version with 256 threads and three barriers per column spent
// 1.9 us per column, 0.48 ms per round -- 90 % of the whole colouring).
__global__ __launch_bounds__(kRound) void colour_resolve_kernel(
    int nq, int head, int max_batch, const unsigned long long* __restrict__ F_in,
    unsigned long long* __restrict__ conf, unsigned long long* __restrict__ full,
    int32_t* __restrict__ class_size, int32_t* __restrict__ colour_out,
    int32_t* __restrict__ round_colour, int* __restrict__ state) {
    extern __shared__ unsigned long long lds[];
    unsigned long long* F = lds;                                    // [kRound][kColourWords]
    unsigned long long* cf = F + (size_t)kRound * kColourWords;     // [kRound][kRound / 64]
    unsigned long long* extra = cf + (size_t)kRound * (kRound / 64);  // [kColourWords] scratch
    int* col = reinterpret_cast<int*>(extra + kColourWords);          // [kRound]
    int* cs = col + kRound;                                           // [4096] class sizes
    const int tid = threadIdx.x;
    for (int x = tid; x < kColourWords * 64; x += kRound) cs[x] = class_size[x];
    for (int x = tid; x < nq * kColourWords; x += kRound) F[x] = F_in[x];
    for (int x = tid; x < nq * (kRound / 64); x += kRound) {
        cf[x] = conf[x];
        conf[x] = 0ull;  // clean for the next round
    }
    if (tid < kColourWords) extra[tid] = 0ull;
    __syncthreads();
    if (state[0]) return;  // an earlier round overflowed: nothing more to do (uniform)
    if (tid >= 64) return;  // the walk is wave 0's
    const int lane = tid;
    unsigned long long fl = full[lane];  // word `lane` of the full-class bitmap
    int n_used = state[1];
    bool overflow = false;
    for (int q = 0; q < nq; ++q) {
        // colours of the earlier columns of the round that share a row with q -> extra
        const unsigned long long* cq = cf + (size_t)q * (kRound / 64);
#pragma unroll
        for (int t = 0; t < kRound / 64; ++t) {
            const int p = t * 64 + lane;
            if (p < q && ((cq[t] >> lane) & 1ull)) {
                const int c = col[p];
                atomicOr(&extra[c >> 6], 1ull << (c & 63));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long taken = F[(size_t)q * kColourWords + lane] | fl | extra[lane];
        extra[lane] = 0ull;
        const unsigned long long fr = ~taken;
        const unsigned long long have = __ballot(fr != 0ull);
        int c;
        if (have == 0ull) {
            overflow = true;
            c = 0;
        } else {
            const int w = __builtin_amdgcn_readfirstlane(__builtin_ctzll(have));  // lowest free word
            const int bit = (fr != 0ull) ? __builtin_ctzll(fr) : 0;
            c = w * 64 + __builtin_amdgcn_readlane(bit, w);
        }
        if (lane == 0) {
            col[q] = c;
            cs[c] += 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // the class is full once it holds max_batch columns (lane c / 64 keeps that word)
        if (cs[c] >= max_batch && lane == (c >> 6)) fl |= 1ull << (c & 63);
        if (c + 1 > n_used) n_used = c + 1;
    }
    if (overflow && lane == 0) state[0] = 1;
    if (lane == 0) state[1] = n_used;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int x = lane; x < nq; x += 64) {
        colour_out[head + x] = col[x];
        round_colour[x] = col[x];
    }
    for (int x = lane; x < kColourWords * 64; x += 64) class_size[x] = cs[x];
    full[lane] = fl;
}

__global__ __launch_bounds__(256) void colour_commit_kernel(
    int nq, int head, const int32_t* __restrict__ order, const int64_t* __restrict__ cptr,
    const int32_t* __restrict__ cidx, const int64_t* __restrict__ rptr, int32_t* __restrict__ rcnt,
    int32_t* __restrict__ rcol, int32_t* __restrict__ rmcnt,
    const int32_t* __restrict__ round_colour) {
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int32_t j = order[head + q];
    const int32_t c = round_colour[q];
    const int64_t cb = cptr[j], ce = cptr[j + 1];
    for (int64_t ii = cb + threadIdx.x; ii < ce; ii += blockDim.x) {
        const int32_t i = cidx[ii];
        const int32_t slot = atomicAdd(&rcnt[i], 1);
        rcol[rptr[i] + slot] = c;
        rmcnt[i] = 0;  // the next round starts with empty member lists
    }
}

// Device pointers: cptr / cidx (CSC structure), rptr (CSR row pointers of the same matrix).
// order_host: the visiting order.  colour_host[q] = colour of the column at visiting position q.
// Returns hipSuccess with *overflow = 1 when 4096 colours do not suffice (colour_host unspecified).
hipError_t device_first_fit(int64_t n, int32_t d, int64_t nnz, const int64_t* cptr,
                            const int32_t* cidx, const int64_t* rptr, const int32_t* order_host,
                            int max_batch, int32_t* colour_host, int* n_colours, int* overflow,
                            hipStream_t stream) {
    *overflow = 0;
    *n_colours = 0;
    hipError_t e;
    int32_t *d_order = nullptr, *rcnt = nullptr, *rcol = nullptr, *class_size = nullptr,
            *d_colour = nullptr, *round_colour = nullptr, *rmcnt = nullptr, *rmem = nullptr;
    unsigned long long *F = nullptr, *conf = nullptr, *full = nullptr;
    int* state = nullptr;
    auto done = [&](hipError_t rc) {
        (void)hipFree(d_order);
        (void)hipFree(rcnt);
        (void)hipFree(rcol);
        (void)hipFree(rmcnt);
        (void)hipFree(rmem);
        (void)hipFree(class_size);
        (void)hipFree(d_colour);
        (void)hipFree(round_colour);
        (void)hipFree(F);
        (void)hipFree(conf);
        (void)hipFree(full);
        (void)hipFree(state);
        return rc;
    };
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1), nr = (size_t)(n > 0 ? n : 1);
    const size_t conf_words = (size_t)kRound * (kRound / 64);
    if ((e = hipMalloc(&d_order, sizeof(int32_t) * (size_t)d)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&rcnt, sizeof(int32_t) * nr)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&rcol, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&rmcnt, sizeof(int32_t) * nr)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&rmem, sizeof(int32_t) * nz)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&class_size, sizeof(int32_t) * kColourWords * 64)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_colour, sizeof(int32_t) * (size_t)d)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&round_colour, sizeof(int32_t) * kRound)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&F, sizeof(unsigned long long) * kRound * kColourWords)) != hipSuccess)
        return done(e);
    if ((e = hipMalloc(&conf, sizeof(unsigned long long) * conf_words)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&full, sizeof(unsigned long long) * kColourWords)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&state, sizeof(int) * 2)) != hipSuccess) return done(e);
    if ((e = hipMemcpyAsync(d_order, order_host, sizeof(int32_t) * (size_t)d, hipMemcpyHostToDevice,
                            stream)) != hipSuccess)
        return done(e);
    if ((e = hipMemsetAsync(rcnt, 0, sizeof(int32_t) * nr, stream)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(rmcnt, 0, sizeof(int32_t) * nr, stream)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(class_size, 0, sizeof(int32_t) * kColourWords * 64, stream)) != hipSuccess)
        return done(e);
    if ((e = hipMemsetAsync(conf, 0, sizeof(unsigned long long) * conf_words, stream)) != hipSuccess)
        return done(e);
    if ((e = hipMemsetAsync(full, 0, sizeof(unsigned long long) * kColourWords, stream)) != hipSuccess)
        return done(e);
    if ((e = hipMemsetAsync(state, 0, sizeof(int) * 2, stream)) != hipSuccess) return done(e);
    const size_t lds = sizeof(unsigned long long) *
                           ((size_t)kRound * kColourWords + conf_words + kColourWords) +
                       sizeof(int) * (kRound + kColourWords * 64);
    if ((e = hipFuncSetAttribute((const void*)colour_resolve_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
        return done(e);
    for (int head = 0; head < d; head += kRound) {
        const int nq = d - head < kRound ? d - head : kRound;
        hipLaunchKernelGGL(colour_snapshot_kernel, dim3(nq), dim3(256), 0, stream, nq, head, d_order,
                           cptr, cidx, rptr, rcnt, rcol, rmcnt, rmem, full, F);
        hipLaunchKernelGGL(colour_conflict_kernel, dim3(nq), dim3(256), 0, stream, nq, head, d_order,
                           cptr, cidx, rptr, rmcnt, rmem, conf);
        hipLaunchKernelGGL(colour_resolve_kernel, dim3(1), dim3(kRound), lds, stream, nq, head,
                           max_batch, F, conf, full, class_size, d_colour, round_colour, state);
        hipLaunchKernelGGL(colour_commit_kernel, dim3(nq), dim3(256), 0, stream, nq, head, d_order,
                           cptr, cidx, rptr, rcnt, rcol, rmcnt, round_colour);
    }
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    int h_state[2] = {0, 0};
    if ((e = hipMemcpyAsync(h_state, state, sizeof(int) * 2, hipMemcpyDeviceToHost, stream)) !=
            hipSuccess ||
        (e = hipMemcpyAsync(colour_host, d_colour, sizeof(int32_t) * (size_t)d, hipMemcpyDeviceToHost,
                            stream)) != hipSuccess ||
        (e = hipStreamSynchronize(stream)) != hipSuccess)
        return done(e);
    *overflow = h_state[0];
    *n_colours = h_state[1];
    return done(hipSuccess);
}

}  // namespace spfm
