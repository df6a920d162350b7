// spfm_engine.hip -- host engine + C ABI (include/spfm.h) of the gfx950 sparse-FM
// proximal coordinate-descent core.  See DESIGN.md for the execution model.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "../../include/spfm.h"
#include "spfm_kernels.hip.h"

namespace spfm {
void schedule_exact(int64_t, int32_t, const int64_t*, const int32_t*, const int32_t*, int,
                    std::vector<int32_t>&);
void schedule_colored(int64_t, int32_t, const int64_t*, const int32_t*, const int32_t*, int,
                      std::vector<int32_t>&, std::vector<int32_t>&);
bool csr_to_csc(int64_t, int32_t, const int64_t*, const int32_t*, std::vector<int64_t>&,
                std::vector<int32_t>&, std::vector<int64_t>&);
int schedule_threads();
void csc_to_csr(int64_t, int32_t, const int64_t*, const int32_t*, std::vector<int64_t>&,
                std::vector<int32_t>&, std::vector<int64_t>&);
void build_wide_stream(int64_t, const int64_t*, const int32_t*, const std::vector<int32_t>&,
                       const std::vector<int32_t>&, int, std::vector<int32_t>&,
                       std::vector<int32_t>&, std::vector<int32_t>&, std::vector<uint8_t>&);
void build_pb_stream(int64_t, const int64_t*, const int32_t*, const std::vector<int32_t>&,
                     const std::vector<int32_t>&, int, int, std::vector<int32_t>&,
                     std::vector<int32_t>&, std::vector<uint8_t>&);
void build_rowblock_stream(int64_t, const int64_t*, const int32_t*, const std::vector<int32_t>&,
                           const std::vector<int32_t>&, int, int, std::vector<int32_t>&,
                           std::vector<int32_t>&, std::vector<uint32_t>&, const uint8_t*);
void schedule_relax(int64_t, int32_t, const int64_t*, const int32_t*, const int32_t*, int, int,
                    std::vector<int32_t>&, std::vector<int32_t>&, std::vector<int32_t>&,
                    std::vector<int32_t>&, std::vector<int64_t>&, std::vector<int64_t>&,
                    std::vector<int16_t>&, std::vector<uint8_t>&);
// spfm_ingest.hip: device-side CSR -> CSC (one stable radix sort by column id)
template <typename T>
hipError_t device_csr_to_csc(int64_t, int32_t, int64_t, const int64_t*, const int32_t*, const T*,
                             int64_t*, int32_t*, T*, int*, hipStream_t);
extern template hipError_t device_csr_to_csc<float>(int64_t, int32_t, int64_t, const int64_t*,
                                                    const int32_t*, const float*, int64_t*,
                                                    int32_t*, float*, int*, hipStream_t);
extern template hipError_t device_csr_to_csc<double>(int64_t, int32_t, int64_t, const int64_t*,
                                                     const int32_t*, const double*, int64_t*,
                                                     int32_t*, double*, int*, hipStream_t);
// spfm_ingest.hip: the row-block entry stream on the device (same result as build_rowblock_stream)
hipError_t device_rowblock_stream(int64_t, int32_t, int64_t, int, int, int, const int32_t*,
                                  const int32_t*, const int64_t*, const int32_t*, const uint8_t*,
                                  int32_t*, int32_t*, uint32_t*, int*, int64_t*, hipStream_t);
// spfm_colour.hip: the first-fit colouring on the device (same result as schedule_colored)
hipError_t device_first_fit(int64_t, int32_t, int64_t, const int64_t*, const int32_t*, const int64_t*,
                            const int32_t*, int, int32_t*, int*, int*, hipStream_t);
}  // namespace spfm

using namespace spfm;

static thread_local std::string g_create_error;

// ------------------------------------------------------------------ RCCL (lazy)
// RCCL is loaded with dlopen so that the single-GPU path has no link dependency
// and shares whichever librccl the process already holds (PyTorch ships one).
namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ {
    char internal[128];
};
enum { ncclSum_ = 0 };
enum { ncclFloat64_ = 8 };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        // 1) a copy the process already holds (PyTorch maps its own librccl.so): share it;
        // 2) otherwise load ROCm's, with local scope so that it never interposes on a
        //    copy another library may bring later.
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* nm : names) {
            lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            const char* fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
            for (const char* nm : fresh) {
                lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        }
        if (!lib) {
            err = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
}  // namespace

// --------------------------------------------------------------------- buffers
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t b) {
        if (b <= bytes && p) return hipSuccess;
        release();
        if (b == 0) b = 16;
        hipError_t e = hipMalloc(&p, b);
        if (e == hipSuccess) bytes = b;
        return e;
    }
    template <typename U>
    U* as() const {
        return reinterpret_cast<U*>(p);
    }
};

#define HIPC(expr)                                                                       \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            err = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return SPFM_ERR_RUNTIME;                                                     \
        }                                                                                \
    } while (0)

#define FAIL(code, msg) \
    do {                \
        err = (msg);    \
        return (code);  \
    } while (0)

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

struct ProfSlot {
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
    double ms = 0.0;
    int64_t launches = 0, nnz = 0;
};

struct spfm_engine {
    int device = 0, dtype = SPFM_F32;
    hipStream_t stream = nullptr;
    std::string err;
    std::string devname;

    // data
    int64_t n = 0, nnz = 0;
    int d = 0;
    bool have_data = false;
    DevBuf cptr, cidx, cval, rptr, ridx, rval, yy, A, col_norm;
    std::vector<int64_t> h_cptr;
    std::vector<int32_t> h_cidx;
    bool col_norm_reduced = false;

    // params
    int n_orders = 0, k = 0;
    bool have_params = false;
    DevBuf P, Pt, w, lams;
    bool p_valid = true, pt_valid = false;  // which of P (k,d) / Pt (d,k) is current
    std::vector<double> h_lams;

    // config
    int solver = -1, loss = 0, reg = 0, top_degree = 0;
    bool configured = false;
    DevBuf norms, cache, dcache;

    // schedule
    std::vector<int32_t> order, batch_ptr;
    DevBuf d_order, d_desc;
    int max_batch_cols = 0;
    bool have_schedule = false;
    int64_t sched_version = 0;

    // work
    DevBuf part, delta, pold, viol_col, scalar, ctl, comp_order, pred_tmp, partial, pb_scal,
        pb_ticket;
    bool pbcd_fuse = true;  // prep + chain in one launch (ticket hand-off)
    double* h_scalar = nullptr;  // pinned

    // psgd (minibatch solver): gradient accumulators, sample order, Michelot state
    DevBuf sg_gradP, sg_gradw, sg_samples, sg_part, sg_cond, sg_thr, sg_theta, sg_done,
        sg_norms, sg_conv, sg_sched, sg_idx, sg_snapP, sg_snapw, sg_snapc;
    bool psgd_force_eager = false;
    int psgd_graph_sweeps = 4;
    bool psgd_warm = false;  // sg_cond holds thresholds of a previous minibatch
    int psgd_redone = 0;  // epochs that fell back from graph replay to eager launches
    std::vector<PsgdBatch> h_sched;

    // graphs
    bool use_graph = true;
    bool fuse_chain = true;  // fused chain+sync kernel for batches of <= 64 columns
    int max_batch_opt = 4096;

    // Recovery from a persistent pass that could not run to its end (a workgroup not resident,
    // a peer that never answers): the epoch is all-or-nothing like the reference's
    // (pcd.py:71-137).  Parameters and regularizer state are snapshot before the launches; after
    // a time-out they are restored, y_pred is recomputed from them (the arguments of the last
    // spfm_init_pred) and the epoch is redone on the multi-kernel engine, which this handle then
    // keeps using.  `pers_fallbacks` counts the events (option "persistent_fallbacks").
    bool pers_failed = false;
    int pers_fallbacks = 0;
    std::string pers_reason;
    unsigned spin_max = 1u << 21;  // polls of one in-kernel wait before the pass gives up
    int debug_drop = 0;            // test hook: the next N persistent launches lack a workgroup
    bool have_pred_args = false;
    int pa_degree = 0, pa_lin = 0, pa_lower = 0;
    DevBuf snapP, snapW, snapC;
    std::map<const void*, int> resident_cache;  // kernel -> workgroups that can be resident

    // persistent row-block pass (single GPU, pcd): one launch per component pass
    bool persistent = true;
    int prb_G = 64;
    bool prb_lds = true;  // keep the row block (A, residual) in LDS when it fits (f32, squared)
    int prb_lds_active = 0;  // what the last pcd pass actually used (0 / 1 residual / 2 sign)
    bool y_pm1 = false;      // every target is +1 or -1
    bool prb_ready = false;
    int prb_has_long = 0;
    int prb_long = kPrbLong;  // entries per (workgroup, step, slot) above which a slot is "long"
    DevBuf prb_sp, prb_erow, prb_eval, prb_slab, prb_abort, prow_old, d_bptr, prb_stamps,
        prb_viol, prb_cn, prb_lmask, prb_rec;
    bool prb_pack = true;    // degree-3 passes with rows in global memory: packed row records
    int prb_pack_active = 0;  // what the last pcd pass used
    bool prb_stamp_on = false;
    int wide_min_cols = 110;     // mean class width below which 64-column steps are used instead
    bool wide_stamp_on = false;  // pcdw_stamps: phase timers of the wide pcd pass (float storage)
    DevBuf wide_stamps;
    DevBuf w_rec;  // packed row records of the wide pcd pass (rows in global memory)
    static constexpr size_t kPrbLds = 84 * 1024;  // > half of the CU's 160 KiB: 1 WG per CU
    // relaxed runs: for a schedule of tiny steps (the reference order: 2.6 columns
    // per step) the degree-2 pcd pass merges consecutive steps into runs of ~20 columns whose few
    // shared rows the chains replay; its own boundaries, entry stream and conflict tables
    bool relax_on = true;
    int relax_state = 0;  // 0 not tried for this schedule, 1 in use, -1 not worth it
    std::vector<int32_t> r_batch_ptr;
    int relax_has_long = 0;
    DevBuf r_bptr, r_sp, r_erow, r_eval, r_lmask, r_cfptr, r_cf, r_clist, r_cslab;
    // wide persistent passes (spfm_pcdw.hip.h): steps of up to 512 columns, degree-2 pcd and
    // cd_linear; chosen when the schedule has a step of more than 64 columns
    bool wide_on = true;
    int pcdw_G = 0;  // workgroups of the wide pass; 0 = chosen from the rows (wide_groups)
    bool wide_ready = false;
    int wide_G = 0, wide_tot = 0;
    int wide_lr_active = 0;
    DevBuf w_wbase, w_wsp, w_erow, w_eval, w_slabA, w_slabB;
    // persistent pbcd pass (spfm_pbprb.hip.h): its own workgroup count, hence its own entry
    // stream when that differs from the pcd / cd_linear pass's
    bool pb_persistent = true;
    int pbprb_G = 256;
    int pbprb_owners = 0;   // dedicated owner workgroups (measured, no gain; off)
    int pb_GO = 0;          // what the installed stream was built for
    int probe_xcd = 0, probe_lds = 60 * 1024;  // diagnostics (spfm_debug_exchange_cost)
    bool pb_stream_ready = false;
    int pb_stream_G = 0, pb_stream_NG = 0;
    DevBuf pb_sp, pb_erow, pb_eval, pb_meta, pb_slabA, pb_slabB, pb_slabC, pb_stamps, pb_rec;
    bool pb_stamp_on = false;
    int pbprb_active = 0;  // what the last pbcd epoch used
    int pb_dbg = 0;
    DevBuf pb_dbgbuf;
    std::map<std::string, hipGraphExec_t> graphs;

    // comm
    ncclComm_t comm = nullptr;
    // Host shared-memory communicator (spfm_comm_init_shm): the same sharded protocol with
    // the all-reduce done through a POSIX shm segment, for ranks that share ONE GPU (RCCL
    // refuses two ranks per device) -- exercises the multi-GPU path on a single-GPU box.
    struct ShmComm {
        static constexpr size_t kMaxDoubles = 1 << 16;
        struct Hdr {
            volatile int arrive;
            volatile int sense;
            int pad[14];
        };
        Hdr* hdr = nullptr;
        double* slots = nullptr;  // [n_ranks][kMaxDoubles]
        size_t bytes = 0;
        int local_sense = 0;
    } shm;
    std::vector<double> shm_host;
    bool dist() const { return comm != nullptr || shm.hdr != nullptr; }
    int n_ranks = 1, rank = 0;
    // In-kernel cross-GPU exchange of the persistent passes (spfm_peer_alloc / _connect): one
    // exchange slab per GPU, mapped into every rank (hipIpc); the kernels write their GPU's
    // per-step totals into every GPU's slab and poll their own -- no per-step collective.
    // Layout (doubles): [0, 16K) pcd / cd_linear [2][n_ranks][64][2]; [16K, ...) pbcd
    // [2][64][n_ranks][64].
    static constexpr size_t kPeerPcdOff = 0, kPeerPbOff = 16 * 1024;
    static constexpr size_t kPeerProbeOff = kPeerPbOff + (size_t)2 * 64 * 8 * 64;  // [8] handshake
    static constexpr size_t kPeerDoubles = kPeerProbeOff + 64;
    int peer_generation = 0;  // connects so far (the handshake word differs per connect)
    void* peer_own = nullptr;
    std::vector<void*> peer_ptr;     // [n_ranks] mapped bases ([rank] = own)
    DevBuf peer_tab_pcd, peer_tab_pb;  // device tables of the per-kernel region pointers
    bool peer_ready = false;

    // profile
    bool prof_on = false;
    ProfSlot prof[5];

    ~spfm_engine() {
        clear_graphs();
        for (auto& ps : prof)
            for (auto e : ps.ev) (void)hipEventDestroy(e);
        if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
        for (size_t r = 0; r < peer_ptr.size(); ++r)
            if (peer_ptr[r] && peer_ptr[r] != peer_own) (void)hipIpcCloseMemHandle(peer_ptr[r]);
        if (peer_own) (void)hipFree(peer_own);
        if (shm.hdr) munmap((void*)shm.hdr, shm.bytes);
        if (h_scalar) (void)hipHostFree(h_scalar);
        if (stream) (void)hipStreamDestroy(stream);
    }

    void clear_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
    }

    RegState regstate() {
        RegState rs;
        rs.norms = norms.as<double>();
        rs.cache = cache.as<double>();
        rs.dcache = dcache.as<double>();
        return rs;
    }

    size_t tsize() const { return dtype == SPFM_F32 ? 4 : 8; }

    int sync() {
        HIPC(hipStreamSynchronize(stream));
        return SPFM_OK;
    }

    // ---------------------------------------------------------------- profiling
    // One event pair per recorded launch (bounded pool); launches beyond the pool
    // are not counted, so ms / launches / nnz always describe the same set.
    static constexpr size_t kProfPool = 32768;
    bool prof_armed = false;
    void prof_begin(int which, int64_t nnz_launch) {
        prof_armed = false;
        if (!prof_on) return;
        ProfSlot& ps = prof[which];
        if (ps.used + 2 > kProfPool) return;
        if (ps.used + 2 > ps.ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess) return;
            if (hipEventCreate(&b) != hipSuccess) {
                (void)hipEventDestroy(a);
                return;
            }
            ps.ev.push_back(a);
            ps.ev.push_back(b);
        }
        ps.launches++;
        ps.nnz += nnz_launch;
        (void)hipEventRecord(ps.ev[ps.used], stream);
        prof_armed = true;
    }
    void prof_cancel(int which, int64_t nnz_launch) {  // the launch announced by prof_begin was not made
        if (!prof_armed) return;
        prof[which].launches--;
        prof[which].nnz -= nnz_launch;
        prof_armed = false;
    }
    void prof_end(int which) {
        if (!prof_armed) return;
        ProfSlot& ps = prof[which];
        (void)hipEventRecord(ps.ev[ps.used + 1], stream);
        ps.used += 2;
        prof_armed = false;
    }
    void prof_collect() {
        if (!prof_on) return;
        (void)hipStreamSynchronize(stream);
        for (auto& ps : prof) {
            for (size_t i = 0; i + 1 < ps.used; i += 2) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ps.ev[i], ps.ev[i + 1]) == hipSuccess) ps.ms += ms;
            }
            ps.used = 0;
        }
    }

    int64_t batch_nnz(int b) const {
        int64_t s = 0;
        for (int q = batch_ptr[b]; q < batch_ptr[b + 1]; ++q)
            s += h_cptr[order[q] + 1] - h_cptr[order[q]];
        return s;
    }

    // ------------------------------------------------------------------- comm
    // sense-reversing barrier over the shm header; bounded (30 s) so that a dead peer
    // becomes an error instead of a hang
    int shm_barrier() {
        shm.local_sense ^= 1;
        if (__atomic_add_fetch(&shm.hdr->arrive, 1, __ATOMIC_ACQ_REL) == n_ranks) {
            __atomic_store_n(&shm.hdr->arrive, 0, __ATOMIC_RELAXED);
            __atomic_store_n(&shm.hdr->sense, shm.local_sense, __ATOMIC_RELEASE);
            return SPFM_OK;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(&shm.hdr->sense, __ATOMIC_ACQUIRE) != shm.local_sense) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                FAIL(SPFM_ERR_RUNTIME, "shm communicator: peer did not arrive within 30 s");
        }
        return SPFM_OK;
    }

    int allreduce_shm(double* buf, size_t count) {
        for (size_t off = 0; off < count; off += ShmComm::kMaxDoubles) {  // long vectors in pieces
            int rc = allreduce_shm_piece(buf + off, std::min(ShmComm::kMaxDoubles, count - off));
            if (rc) return rc;
        }
        return SPFM_OK;
    }
    int allreduce_shm_piece(double* buf, size_t count) {
        shm_host.resize(count);
        HIPC(hipMemcpyAsync(shm_host.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost,
                            stream));
        HIPC(hipStreamSynchronize(stream));
        std::memcpy(shm.slots + (size_t)rank * ShmComm::kMaxDoubles, shm_host.data(),
                    sizeof(double) * count);
        int rc = shm_barrier();
        if (rc) return rc;
        for (size_t i = 0; i < count; ++i) {  // fixed rank order: identical on every rank
            double a = 0.0;
            for (int r = 0; r < n_ranks; ++r) a += shm.slots[(size_t)r * ShmComm::kMaxDoubles + i];
            shm_host[i] = a;
        }
        rc = shm_barrier();  // nobody overwrites a slot before everyone has read it
        if (rc) return rc;
        HIPC(hipMemcpyAsync(buf, shm_host.data(), sizeof(double) * count, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipStreamSynchronize(stream));
        return SPFM_OK;
    }

    int allreduce(double* buf, size_t count) {
        if (shm.hdr) return allreduce_shm(buf, count);
        if (!comm) return SPFM_OK;
        int rc = g_rccl.AllReduce(buf, buf, count, ncclFloat64_, ncclSum_, comm, stream);
        if (rc != 0) {
            err = std::string("ncclAllReduce: ") +
                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            return SPFM_ERR_RUNTIME;
        }
        return SPFM_OK;
    }

    int ensure_col_norm() {
        if (col_norm_reduced || !dist()) return SPFM_OK;
        int rc = allreduce(col_norm.as<double>(), (size_t)d);
        if (rc) return rc;
        col_norm_reduced = true;
        return SPFM_OK;
    }

    // --------------------------------------------------------- P <-> Pt images
    int ensure_p() {
        if (p_valid) return SPFM_OK;
        for (int o = 0; o < n_orders; ++o) {
            const size_t off = (size_t)o * k * d;
            hipLaunchKernelGGL(transpose_kernel, dim3(cdiv((int64_t)k * d, 256)), dim3(256), 0,
                               stream, Pt.as<double>() + off, d, k, P.as<double>() + off);
        }
        HIPC(hipGetLastError());
        p_valid = true;
        return SPFM_OK;
    }
    int ensure_pt() {
        if (pt_valid) return SPFM_OK;
        HIPC(Pt.alloc(sizeof(double) * (size_t)n_orders * k * d));
        for (int o = 0; o < n_orders; ++o) {
            const size_t off = (size_t)o * k * d;
            hipLaunchKernelGGL(transpose_kernel, dim3(cdiv((int64_t)k * d, 256)), dim3(256), 0,
                               stream, P.as<double>() + off, k, d, Pt.as<double>() + off);
        }
        HIPC(hipGetLastError());
        pt_valid = true;
        return SPFM_OK;
    }

    // ---------------------------------------------------------------- graph util
    // Runs `body` (which only enqueues work on `stream`) either directly or, when
    // graphs are enabled, captured once under `key` and replayed.
    template <typename F>
    int run_cached(const std::string& key, F&& body) {
        const bool graph_ok = use_graph && !prof_on && !dist();
        if (!graph_ok) return body();
        auto it = graphs.find(key);
        if (it == graphs.end()) {
            hipGraph_t g = nullptr;
            HIPC(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            int rc = body();
            hipError_t e = hipStreamEndCapture(stream, &g);
            if (rc != SPFM_OK) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            if (e != hipSuccess) {
                err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
                return SPFM_ERR_RUNTIME;
            }
            hipGraphExec_t ge = nullptr;
            e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) {
                err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
                return SPFM_ERR_RUNTIME;
            }
            it = graphs.emplace(key, ge).first;
        }
        HIPC(hipGraphLaunch(it->second, stream));
        return SPFM_OK;
    }

    // =================================================================== data
    // uploads both images; values in CSC order (`data_csc`) or in CSR order (`data_csr`) --
    // the other order goes through `perm` (position in the wanted order -> position in the
    // given one); conversions to the storage type run on host threads
    template <typename T>
    int upload_images(const int64_t* h_cp, const int32_t* h_ci, const int64_t* h_rp,
                      const int32_t* h_ri, const double* data_csc, const double* data_csr,
                      const int64_t* perm, const double* y) {
        std::vector<T> cv((size_t)nnz), rv((size_t)nnz);
        const int T_ = (nnz >= (1 << 20)) ? schedule_threads() : 1;
        {
            std::vector<std::thread> pool;
            auto work = [&](int tid) {
                const int64_t per = (nnz + T_ - 1) / T_;
                const int64_t lo = per * tid, hi = std::min<int64_t>(nnz, lo + per);
                if (data_csc) {
                    for (int64_t ii = lo; ii < hi; ++ii) cv[(size_t)ii] = (T)data_csc[ii];
                    for (int64_t ii = lo; ii < hi; ++ii) rv[(size_t)ii] = (T)data_csc[perm[ii]];
                } else {
                    for (int64_t ii = lo; ii < hi; ++ii) rv[(size_t)ii] = (T)data_csr[ii];
                    for (int64_t ii = lo; ii < hi; ++ii) cv[(size_t)ii] = (T)data_csr[perm[ii]];
                }
            };
            for (int t = 1; t < T_; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto& th : pool) th.join();
        }
        std::vector<T> hy((size_t)n * 2);
        for (int64_t i = 0; i < n; ++i) {
            hy[(size_t)2 * i] = (T)0;
            hy[(size_t)2 * i + 1] = (T)y[i];
        }
        HIPC(cptr.alloc(sizeof(int64_t) * ((size_t)d + 1)));
        HIPC(cidx.alloc(sizeof(int32_t) * (size_t)nnz));
        HIPC(cval.alloc(sizeof(T) * (size_t)nnz));
        HIPC(rptr.alloc(sizeof(int64_t) * ((size_t)n + 1)));
        HIPC(ridx.alloc(sizeof(int32_t) * (size_t)nnz));
        HIPC(rval.alloc(sizeof(T) * (size_t)nnz));
        HIPC(yy.alloc(sizeof(T) * 2 * (size_t)n));
        HIPC(col_norm.alloc(sizeof(double) * (size_t)d));
        HIPC(hipMemcpyAsync(cptr.p, h_cp, sizeof(int64_t) * ((size_t)d + 1), hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(cidx.p, h_ci, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(cval.p, cv.data(), sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(rptr.p, h_rp, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(ridx.p, h_ri, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(rval.p, rv.data(), sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(yy.p, hy.data(), sizeof(T) * 2 * (size_t)n, hipMemcpyHostToDevice,
                            stream));
        hipLaunchKernelGGL((col_norm_kernel<T>), dim3(cdiv((int64_t)d * 64, kBlock)),
                           dim3(kBlock), 0, stream, d, cptr.as<int64_t>(), cval.as<T>(),
                           col_norm.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(stream));  // host staging vectors die here
        return SPFM_OK;
    }

    template <typename T>
    int set_data_t(const int64_t* indptr, const int32_t* indices, const double* data,
                   const double* y) {
        std::vector<int64_t> h_rptr, perm;
        std::vector<int32_t> h_ridx;
        csc_to_csr(n, d, indptr, indices, h_rptr, h_ridx, perm);
        return upload_images<T>(indptr, indices, h_rptr.data(), h_ridx.data(), data, nullptr,
                                perm.data(), y);
    }

    // after the images are on the device: state shared by both ingest forms
    int data_installed(const double* y) {
        y_pm1 = true;
        for (int64_t i = 0; i < n; ++i)
            if (std::fabs(y[i]) != 1.0) {
                y_pm1 = false;
                break;
            }
        have_data = true;
        have_schedule = false;
        configured = false;
        col_norm_reduced = false;
        clear_graphs();
        HIPC(viol_col.alloc(sizeof(double) * (size_t)d));
        HIPC(pred_tmp.alloc(sizeof(double) * (size_t)(n > 0 ? n : 1)));
        HIPC(partial.alloc(sizeof(double) * 1024));
        return SPFM_OK;
    }

    // CSR ingest on the DEVICE (round 3; SURVEY.md 8f N4 as written): the CSR arrays go up once
    // (they are the engine's row-major image anyway), the CSC image is their stable radix sort
    // by column id (spfm_ingest.hip); the host keeps only the CSC *structure* (indptr, row ids),
    // copied back for the schedule and stream builders.  Returns kIngestFallback when the device
    // path cannot be used (the host-thread transposition then takes over).
    bool ingest_device = true;
    int ingest_device_used = 0;
    int co_tenants = 1;  // persistent passes of other handles expected on the device at the same time
    static constexpr int kIngestFallback = 2;
    template <typename T>
    int set_data_csr_device(const int64_t* indptr, const int32_t* indices, const double* data,
                            const double* y) {
        std::vector<T> rv((size_t)(nnz > 0 ? nnz : 1));
        const int T_ = (nnz >= (1 << 20)) ? schedule_threads() : 1;
        {
            std::vector<std::thread> pool;
            auto work = [&](int tid) {
                const int64_t per = (nnz + T_ - 1) / T_;
                const int64_t lo = per * tid, hi = std::min<int64_t>(nnz, lo + per);
                for (int64_t ii = lo; ii < hi; ++ii) rv[(size_t)ii] = (T)data[ii];
            };
            for (int t = 1; t < T_; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto& th : pool) th.join();
        }
        std::vector<T> hy((size_t)n * 2 + 2);
        for (int64_t i = 0; i < n; ++i) {
            hy[(size_t)2 * i] = (T)0;
            hy[(size_t)2 * i + 1] = (T)y[i];
        }
        const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
        HIPC(cptr.alloc(sizeof(int64_t) * ((size_t)d + 1)));
        HIPC(cidx.alloc(sizeof(int32_t) * nz));
        HIPC(cval.alloc(sizeof(T) * nz));
        HIPC(rptr.alloc(sizeof(int64_t) * ((size_t)n + 1)));
        HIPC(ridx.alloc(sizeof(int32_t) * nz));
        HIPC(rval.alloc(sizeof(T) * nz));
        HIPC(yy.alloc(sizeof(T) * 2 * (size_t)(n > 0 ? n : 1)));
        HIPC(col_norm.alloc(sizeof(double) * (size_t)d));
        HIPC(hipMemcpyAsync(rptr.p, indptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice,
                            stream));
        if (nnz > 0) {
            HIPC(hipMemcpyAsync(ridx.p, indices, sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemcpyAsync(rval.p, rv.data(), sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice,
                                stream));
        }
        if (n > 0)
            HIPC(hipMemcpyAsync(yy.p, hy.data(), sizeof(T) * 2 * (size_t)n, hipMemcpyHostToDevice,
                                stream));
        HIPC(hipStreamSynchronize(stream));
        int invalid = 0;
        hipError_t e = device_csr_to_csc<T>(n, d, nnz, rptr.as<int64_t>(), ridx.as<int32_t>(),
                                            rval.as<T>(), cptr.as<int64_t>(), cidx.as<int32_t>(),
                                            cval.as<T>(), &invalid, stream);
        if (e != hipSuccess) {  // e.g. no room for the sort's scratch: the host path takes over
            (void)hipGetLastError();
            return kIngestFallback;
        }
        if (invalid)
            FAIL(SPFM_ERR_INVALID,
                 "set_data: CSR must have sorted, duplicate-free column indices in [0, d)");
        h_cptr.resize((size_t)d + 1);
        h_cidx.resize((size_t)nnz);
        HIPC(hipMemcpyAsync(h_cptr.data(), cptr.p, sizeof(int64_t) * ((size_t)d + 1),
                            hipMemcpyDeviceToHost, stream));
        if (nnz > 0)
            HIPC(hipMemcpyAsync(h_cidx.data(), cidx.p, sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyDeviceToHost, stream));
        hipLaunchKernelGGL((col_norm_kernel<T>), dim3(cdiv((int64_t)d * 64, kBlock)), dim3(kBlock), 0,
                           stream, d, cptr.as<int64_t>(), cval.as<T>(), col_norm.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(stream));
        return SPFM_OK;
    }

    // CSR ingest (replaces get_dataset's X.tocsc(), dataset.py:119-123, too): the CSC image
    // is built on the device (above) or by host threads; the CSR image is the input itself
    int set_data_csr(int64_t n_, int32_t d_, const int64_t* indptr, const int32_t* indices,
                     const double* data, const double* y) {
        if (n_ < 0 || d_ <= 0 || !indptr || !y) FAIL(SPFM_ERR_INVALID, "set_data: bad arguments");
        if (n_ >= (int64_t)1 << 31) FAIL(SPFM_ERR_UNSUPPORTED, "n_samples must be < 2^31");
        if (indptr[0] != 0) FAIL(SPFM_ERR_INVALID, "set_data: indptr[0] != 0");
        for (int64_t i = 0; i < n_; ++i)
            if (indptr[i + 1] < indptr[i]) FAIL(SPFM_ERR_INVALID, "set_data: indptr not monotone");
        ingest_device_used = 0;
        if (ingest_device && indptr[n_] < ((int64_t)1 << 31)) {
            if (have_params && d_ != d) have_params = false;
            n = n_;
            d = d_;
            nnz = indptr[n_];
            int rc = dtype == SPFM_F32 ? set_data_csr_device<float>(indptr, indices, data, y)
                                       : set_data_csr_device<double>(indptr, indices, data, y);
            if (rc == SPFM_OK) {
                ingest_device_used = 1;
                return data_installed(y);
            }
            if (rc != kIngestFallback) {
                have_data = false;
                return rc;
            }
        }
        std::vector<int64_t> cp, perm;
        std::vector<int32_t> ci;
        if (!csr_to_csc(n_, d_, indptr, indices, cp, ci, perm))
            FAIL(SPFM_ERR_INVALID,
                 "set_data: CSR must have sorted, duplicate-free column indices in [0, d)");
        if (have_params && d_ != d) have_params = false;
        n = n_;
        d = d_;
        nnz = indptr[n_];
        h_cptr.swap(cp);
        h_cidx.swap(ci);
        int rc = (dtype == SPFM_F32)
                     ? upload_images<float>(h_cptr.data(), h_cidx.data(), indptr, indices, nullptr,
                                            data, perm.data(), y)
                     : upload_images<double>(h_cptr.data(), h_cidx.data(), indptr, indices, nullptr,
                                             data, perm.data(), y);
        if (rc) return rc;
        return data_installed(y);
    }

    int set_data(int64_t n_, int32_t d_, const int64_t* indptr, const int32_t* indices,
                 const double* data, const double* y) {
        if (n_ < 0 || d_ <= 0 || !indptr || !y) FAIL(SPFM_ERR_INVALID, "set_data: bad arguments");
        if (n_ >= (int64_t)1 << 31) FAIL(SPFM_ERR_UNSUPPORTED, "n_samples must be < 2^31");
        if (indptr[0] != 0) FAIL(SPFM_ERR_INVALID, "set_data: indptr[0] != 0");
        for (int j = 0; j < d_; ++j)
            if (indptr[j + 1] < indptr[j]) FAIL(SPFM_ERR_INVALID, "set_data: indptr not monotone");
        const int64_t nz = indptr[d_];
        for (int64_t ii = 0; ii < nz; ++ii)
            if (indices[ii] < 0 || indices[ii] >= n_)
                FAIL(SPFM_ERR_INVALID, "set_data: row index out of range");
        // canonical CSC required: ascending, duplicate-free rows inside each column
        for (int j = 0; j < d_; ++j)
            for (int64_t ii = indptr[j] + 1; ii < indptr[j + 1]; ++ii)
                if (indices[ii] <= indices[ii - 1])
                    FAIL(SPFM_ERR_INVALID,
                         "set_data: CSC must have sorted, duplicate-free row indices");
        if (have_params && d_ != d) have_params = false;
        n = n_;
        d = d_;
        nnz = nz;
        h_cptr.assign(indptr, indptr + d + 1);
        h_cidx.assign(indices, indices + nnz);
        int rc = (dtype == SPFM_F32) ? set_data_t<float>(indptr, indices, data, y)
                                     : set_data_t<double>(indptr, indices, data, y);
        if (rc) return rc;
        return data_installed(y);
    }

    // ================================================================= params
    int set_params(int n_orders_, int k_, int32_t d_, const double* P_, const double* w_,
                   const double* lams_) {
        if (n_orders_ <= 0 || k_ <= 0 || d_ <= 0 || !P_ || !w_ || !lams_)
            FAIL(SPFM_ERR_INVALID, "set_params: bad arguments");
        if (have_data && d_ != d)
            FAIL(SPFM_ERR_INVALID, "set_params: n_features differs from the data");
        if (!have_data) d = d_;
        for (int s = 0; s < k_; ++s)
            if (std::fabs(lams_[s]) != 1.0) FAIL(SPFM_ERR_INVALID, "Lambdas must be +1 or -1.");
        if (n_orders_ != n_orders || k_ != k) {
            configured = false;
            clear_graphs();
        }
        n_orders = n_orders_;
        k = k_;
        h_lams.assign(lams_, lams_ + k);
        HIPC(P.alloc(sizeof(double) * (size_t)n_orders * k * d));
        HIPC(w.alloc(sizeof(double) * (size_t)d));
        HIPC(lams.alloc(sizeof(double) * (size_t)k));
        HIPC(hipMemcpyAsync(P.p, P_, sizeof(double) * (size_t)n_orders * k * d,
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(w.p, w_, sizeof(double) * (size_t)d, hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(lams.p, lams_, sizeof(double) * (size_t)k, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipStreamSynchronize(stream));
        p_valid = true;
        pt_valid = false;
        have_params = true;
        return SPFM_OK;
    }

    int get_params(double* P_, double* w_) {
        if (!have_params) FAIL(SPFM_ERR_INVALID, "get_params: no parameters set");
        int rc = ensure_p();
        if (rc) return rc;
        if (P_)
            HIPC(hipMemcpyAsync(P_, P.p, sizeof(double) * (size_t)n_orders * k * d,
                                hipMemcpyDeviceToHost, stream));
        if (w_)
            HIPC(hipMemcpyAsync(w_, w.p, sizeof(double) * (size_t)d, hipMemcpyDeviceToHost,
                                stream));
        HIPC(hipStreamSynchronize(stream));
        return SPFM_OK;
    }

    // ============================================================== configure
    int configure(int solver_, int loss_, int reg_, int top_degree_) {
        if (!have_data || !have_params)
            FAIL(SPFM_ERR_INVALID, "configure: set data and parameters first");
        if (loss_ < 0 || loss_ > 2) FAIL(SPFM_ERR_INVALID, "Loss function not supported.");
        if (reg_ < 0 || reg_ > 5) FAIL(SPFM_ERR_INVALID, "Regularizer not supported.");
        if (solver_ != SPFM_SOLVER_PCD && solver_ != SPFM_SOLVER_PBCD && solver_ != SPFM_SOLVER_PSGD)
            FAIL(SPFM_ERR_INVALID, "Solver is not supported.");
        if (solver_ == SPFM_SOLVER_PSGD) return configure_psgd(loss_, reg_, top_degree_);
        const bool all_subsets = top_degree_ == -1;  // regularizers are called with degree = -1
        if (!all_subsets && top_degree_ < 2)
            FAIL(SPFM_ERR_UNSUPPORTED, "degree must be >= 2 (factorization machine) or -1 (all-subsets)");
        if (top_degree_ > SPFM_MAX_DEGREE)
            FAIL(SPFM_ERR_UNSUPPORTED, "degree > 6 is not supported by the HIP engine");
        if (all_subsets) {
            // sparse_all_subsets.py:33-38: l1 / l21 / omegacs / omegati
            const bool ok_pcd = (reg_ == SPFM_REG_L1 || reg_ == SPFM_REG_OMEGATI);
            const bool ok_pbcd = (reg_ == SPFM_REG_L1 || reg_ == SPFM_REG_L21 || reg_ == SPFM_REG_OMEGACS);
            if ((solver_ == SPFM_SOLVER_PCD && !ok_pcd) || (solver_ == SPFM_SOLVER_PBCD && !ok_pbcd))
                FAIL(SPFM_ERR_INVALID, "this regularizer cannot be used with this solver (all-subsets)");
            if (solver_ == SPFM_SOLVER_PBCD && k > 256)
                FAIL(SPFM_ERR_UNSUPPORTED, "pbcd: n_components > 256 not supported");
        } else if (solver_ == SPFM_SOLVER_PCD) {
            // init_cache_pcd exists only for l1 / squaredl12 / omegati (README.md:28-32)
            if (reg_ != SPFM_REG_L1 && reg_ != SPFM_REG_SQUAREDL12 && reg_ != SPFM_REG_OMEGATI)
                FAIL(SPFM_ERR_INVALID, "this regularizer cannot be used with solver='pcd'");
            if (reg_ == SPFM_REG_SQUAREDL12 && top_degree_ > 2)
                FAIL(SPFM_ERR_INVALID, "SquaredL12 supports only degree=2.");
        } else {
            if (reg_ != SPFM_REG_L1 && reg_ != SPFM_REG_L21 && reg_ != SPFM_REG_SQUAREDL21 &&
                reg_ != SPFM_REG_OMEGACS)
                FAIL(SPFM_ERR_INVALID, "this regularizer cannot be used with solver='pbcd'");
            if (reg_ == SPFM_REG_SQUAREDL21 && top_degree_ != 2)
                FAIL(SPFM_ERR_INVALID, "SquaredL21 supports only degree=2.");
            if (k > 256) FAIL(SPFM_ERR_UNSUPPORTED, "pbcd: n_components > 256 not supported");
        }
        solver = solver_;
        loss = loss_;
        reg = reg_;
        top_degree = top_degree_;
        clear_graphs();
        const size_t ncache = kMaxDegree + 2;
        HIPC(norms.alloc(sizeof(double) * (size_t)d));
        HIPC(cache.alloc(sizeof(double) * ncache * 2));  // pcd: double-buffered per batch
        HIPC(dcache.alloc(sizeof(double) * ncache));
        HIPC(hipMemsetAsync(norms.p, 0, sizeof(double) * (size_t)d, stream));
        HIPC(hipMemsetAsync(cache.p, 0, sizeof(double) * ncache * 2, stream));
        double hd[kMaxDegree + 2] = {0};
        hd[1] = 1.0;  // omegacs.py:46 ; omegati sets it in compute_cache_pcd
        HIPC(hipMemcpyAsync(dcache.p, hd, sizeof(double) * ncache, hipMemcpyHostToDevice, stream));
        // pcd keeps the caches of ALL components (one precompute pass per epoch): same
        // footprint as pbcd's (n, (m-1), k) tensor
        const size_t arow = (size_t)(top_degree > 0 ? top_degree - 1 : 1) * k;
        HIPC(A.alloc(tsize() * (size_t)(n > 0 ? n : 1) * arow));
        HIPC(ctl.alloc(sizeof(Ctl)));
        HIPC(hipMemsetAsync(ctl.p, 0, sizeof(Ctl), stream));
        HIPC(comp_order.alloc(sizeof(int32_t) * (size_t)k));
        HIPC(scalar.alloc(sizeof(double) * 8));
        HIPC(pb_ticket.alloc(sizeof(int) * 4));
        if (!h_scalar) HIPC(hipHostMalloc((void**)&h_scalar, sizeof(double) * 8));
        HIPC(hipStreamSynchronize(stream));
        configured = true;
        return alloc_work();
    }

    int alloc_work() {
        if (!configured || !have_schedule) return SPFM_OK;
        const size_t per_part = (solver == SPFM_SOLVER_PBCD) ? ((size_t)k + 1) * kPbW : 2;
        const size_t per_delta = (solver == SPFM_SOLVER_PBCD) ? (size_t)k : 1;
        HIPC(part.alloc(sizeof(double) * per_part * (size_t)max_batch_cols));
        HIPC(delta.alloc(sizeof(double) * per_delta * (size_t)max_batch_cols));
        HIPC(pold.alloc(sizeof(double) * per_delta * (size_t)max_batch_cols));
        HIPC(pb_scal.alloc(sizeof(double) * 4 * (size_t)max_batch_cols));
        return SPFM_OK;
    }

    // =============================================================== schedule
    // First-fit colouring of the conflict graph in the visiting order `jf` into order / batch_ptr:
    // on the device when the conflict structure is the handle's own matrix (spfm_colour.hip; the
    // same classes as the host form, tests/test_hip_colour.py), else -- global structure of a
    // sharded run, small problems, more than 4096 colours -- by the host threads.
    bool colour_device = true;
    int colour_device_used = 0;
    bool stream_device = true;   // the 64-column pass's entry stream built on the device
    int stream_device_used = 0;
    int colour_columns(int64_t rows, const int64_t* cp, const int32_t* ci, bool own,
                       const int32_t* jf, int max_batch) {
        colour_device_used = 0;
        if (own && colour_device && d >= 4096 && nnz >= (1 << 20) && nnz < ((int64_t)1 << 31) &&
            (int64_t)d / std::max(1, max_batch) < 3500 && rptr.p && cptr.p) {
            std::vector<int32_t> col((size_t)d);
            int nc = 0, ovf = 0;
            const hipError_t e = device_first_fit(n, d, nnz, cptr.as<int64_t>(), cidx.as<int32_t>(),
                                                  rptr.as<int64_t>(), jf, max_batch, col.data(), &nc,
                                                  &ovf, stream);
            if (e == hipSuccess && !ovf && nc > 0) {
                // classes in colour order, their columns in visiting order (stable counting sort)
                std::vector<int32_t> bp((size_t)nc + 1, 0);
                for (int q = 0; q < d; ++q) bp[(size_t)col[(size_t)q] + 1]++;
                for (int c = 0; c < nc; ++c) bp[(size_t)c + 1] += bp[(size_t)c];
                std::vector<int32_t> pos(bp.begin(), bp.end() - 1);
                order.assign((size_t)d, 0);
                for (int q = 0; q < d; ++q) order[(size_t)pos[(size_t)col[(size_t)q]]++] = jf[q];
                batch_ptr = std::move(bp);
                colour_device_used = 1;
                return SPFM_OK;
            }
            (void)hipGetLastError();
        }
        schedule_colored(rows, d, cp, ci, jf, max_batch, order, batch_ptr);
        return SPFM_OK;
    }

    int set_schedule(int mode, const int32_t* indices_feature, const int64_t* cf_indptr,
                     const int32_t* cf_indices, int64_t cf_rows, int32_t* order_out,
                     int32_t* n_batches_out) {
        if (!have_data) FAIL(SPFM_ERR_INVALID, "set_schedule: no data");
        if (!indices_feature) FAIL(SPFM_ERR_INVALID, "set_schedule: indices_feature is NULL");
        std::vector<char> seen((size_t)d, 0);
        for (int q = 0; q < d; ++q) {
            const int j = indices_feature[q];
            if (j < 0 || j >= d || seen[(size_t)j])
                FAIL(SPFM_ERR_INVALID, "set_schedule: indices_feature is not a permutation");
            seen[(size_t)j] = 1;
        }
        const int64_t* cp = cf_indptr ? cf_indptr : h_cptr.data();
        const int32_t* ci = cf_indptr ? cf_indices : h_cidx.data();
        const int64_t rows = cf_indptr ? cf_rows : n;
        if (cf_indptr && (!cf_indices || cf_rows <= 0))
            FAIL(SPFM_ERR_INVALID, "set_schedule: bad conflict structure");
        // persistent passes: 64 columns per step; the wide passes (degree-2 pcd, cd_linear) 512
        const bool pers = persistent && (!dist() || peer_ready);
        const bool wide_cfg = wide_on && configured && solver == SPFM_SOLVER_PCD && top_degree == 2;
        const int max_batch =
            pers ? std::min(max_batch_opt, wide_cfg ? 512 : 64) : max_batch_opt;
        if (mode == SPFM_SCHED_EXACT) {
            order.assign(indices_feature, indices_feature + d);
            schedule_exact(rows, d, cp, ci, indices_feature, max_batch, batch_ptr);
        } else if (mode == SPFM_SCHED_COLORED) {
            colour_columns(rows, cp, ci, !cf_indptr, indices_feature, max_batch);
            if (pers && max_batch > 64 && batch_ptr.size() > 1) {
                // A wide step costs about twice a 64-column step (two fabric hops, 7.0 vs 3.2 us
                // on one GPU): classes of moderate width are cheaper as more, narrower steps.
                // 110 columns per class is where d/64 steps of the 64-column pass equal the
                // classes' count of wide steps; below it, colour again with 64.
                int32_t widest = 0;
                for (size_t b = 0; b + 1 < batch_ptr.size(); ++b)
                    widest = std::max(widest, batch_ptr[b + 1] - batch_ptr[b]);
                const double mean_cols = (double)d / (double)(batch_ptr.size() - 1);
                // ... with the 64-column pass's rows in LDS; when the row blocks of 64 workgroups
                // do not fit (float storage: > ~1.1 M rows per GPU; double storage: never) its
                // step costs 5.7 us and the break-even is 80 columns
                // Decided from GLOBAL inputs only (the conflict structure's row count over the
                // ranks, the loss, the options): every rank of a sharded run must
                // cut its sweep into the same steps, or the exchange of the replicated chain
                // mismatches.  Same LDS formula as pcd_pass_prb (degree 2: one cache value).
                int lds_max = 0, ncu = 0;
                HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock,
                                           device));
                HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
                const int nr_eff = dist() ? n_ranks : 1;
                const int64_t rows_rank = (rows + nr_eff - 1) / nr_eff;  // largest row shard
                const int Gp = std::max(1, std::min(prb_G, ncu));
                const size_t lds_lr = sizeof(double) * kPrbLdsFixed +
                                      (size_t)((rows_rank + Gp - 1) / Gp) *
                                          (4 + (loss == SPFM_LOSS_SQUARED ? 4 : 5)) + 16;
                // (a non-squared loss keeps its rows in LDS only when all targets are +-1 --
                // the classifiers' case; that rank-local fact is deliberately NOT part of the
                // decision, which must come out the same on every rank without a collective:
                // rank 0 alone may be colouring, bench.py / a cached Schedule)
                const bool rows_fit = dtype == SPFM_F32 && prb_lds && lds_lr <= (size_t)lds_max;
                const double limit = rows_fit ? (double)wide_min_cols : 0.72 * (double)wide_min_cols;
                if (widest > 64 && mean_cols < limit)
                    colour_columns(rows, cp, ci, !cf_indptr, indices_feature, 64);
            }
        } else {
            FAIL(SPFM_ERR_INVALID, "set_schedule: unknown mode");
        }
        int rc = install_schedule();
        if (rc) return rc;
        if (order_out) std::memcpy(order_out, order.data(), sizeof(int32_t) * (size_t)d);
        if (n_batches_out) *n_batches_out = (int32_t)batch_ptr.size() - 1;
        return SPFM_OK;
    }

    // upload order / descriptors / batch boundaries of the schedule in `order`, `batch_ptr`
    int install_schedule() {
        max_batch_cols = 1;
        for (size_t b = 0; b + 1 < batch_ptr.size(); ++b)
            max_batch_cols = std::max(max_batch_cols, batch_ptr[b + 1] - batch_ptr[b]);
        std::vector<ColDesc> hdesc((size_t)d);
        for (int q = 0; q < d; ++q) {
            const int j = order[(size_t)q];
            hdesc[(size_t)q].start = h_cptr[(size_t)j];
            hdesc[(size_t)q].len = (int32_t)(h_cptr[(size_t)j + 1] - h_cptr[(size_t)j]);
            hdesc[(size_t)q].j = j;
        }
        HIPC(d_order.alloc(sizeof(int32_t) * (size_t)d));
        HIPC(d_desc.alloc(sizeof(ColDesc) * (size_t)d));
        HIPC(hipMemcpyAsync(d_order.p, order.data(), sizeof(int32_t) * (size_t)d,
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(d_desc.p, hdesc.data(), sizeof(ColDesc) * (size_t)d,
                            hipMemcpyHostToDevice, stream));
        std::vector<int32_t> hb(batch_ptr.begin(), batch_ptr.end());
        HIPC(d_bptr.alloc(sizeof(int32_t) * hb.size()));
        HIPC(hipMemcpyAsync(d_bptr.p, hb.data(), sizeof(int32_t) * hb.size(),
                            hipMemcpyHostToDevice, stream));
        HIPC(hipStreamSynchronize(stream));
        have_schedule = true;
        prb_ready = false;
        pb_stream_ready = false;
        wide_ready = false;
        relax_state = 0;
        ++sched_version;
        clear_graphs();
        return alloc_work();
    }

    int set_schedule_raw(const int32_t* order_in, const int32_t* bptr_in, int32_t nb,
                         const int64_t* cf_indptr, const int32_t* cf_indices, int64_t cf_rows) {
        if (!have_data) FAIL(SPFM_ERR_INVALID, "set_schedule_raw: no data");
        if (!order_in || !bptr_in || nb < 1) FAIL(SPFM_ERR_INVALID, "set_schedule_raw: bad arguments");
        if (bptr_in[0] != 0 || bptr_in[nb] != d)
            FAIL(SPFM_ERR_INVALID, "set_schedule_raw: batch_ptr must run from 0 to n_features");
        std::vector<char> seen((size_t)d, 0);
        for (int q = 0; q < d; ++q) {
            const int j = order_in[q];
            if (j < 0 || j >= d || seen[(size_t)j])
                FAIL(SPFM_ERR_INVALID, "set_schedule_raw: order is not a permutation");
            seen[(size_t)j] = 1;
        }
        const int64_t* cp = cf_indptr ? cf_indptr : h_cptr.data();
        const int32_t* ci = cf_indptr ? cf_indices : h_cidx.data();
        const int64_t rows = cf_indptr ? cf_rows : n;
        if (cf_indptr && (!cf_indices || cf_rows <= 0))
            FAIL(SPFM_ERR_INVALID, "set_schedule_raw: bad conflict structure");
        std::vector<int32_t> stamp((size_t)rows, -1);
        for (int b = 0; b < nb; ++b) {
            if (bptr_in[b + 1] < bptr_in[b])
                FAIL(SPFM_ERR_INVALID, "set_schedule_raw: batch_ptr is not monotone");
            for (int q = bptr_in[b]; q < bptr_in[b + 1]; ++q) {
                const int j = order_in[q];
                for (int64_t ii = cp[j]; ii < cp[j + 1]; ++ii) {
                    if (stamp[(size_t)ci[ii]] == b)
                        FAIL(SPFM_ERR_INVALID,
                             "set_schedule_raw: two columns of one batch share a row");
                    stamp[(size_t)ci[ii]] = b;
                }
            }
        }
        order.assign(order_in, order_in + d);
        batch_ptr.assign(bptr_in, bptr_in + nb + 1);
        return install_schedule();
    }

    int n_batches() const { return (int)batch_ptr.size() - 1; }

    // ================================================================ predict
    template <typename T, int M>
    void launch_anova(int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv,
                      const double* Pt_o, double* out) {
        hipLaunchKernelGGL((anova_predict_kernel<T, M>), dim3(cdiv(rows * 64, kBlock)),
                           dim3(kBlock), 0, stream, rows, k, rp, ri, rv, Pt_o, lams.as<double>(),
                           out);
    }
    template <typename T>
    int anova_dispatch(int M, int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv,
                       const double* Pt_o, double* out) {
        switch (M) {
            case 0: launch_anova<T, 0>(rows, rp, ri, rv, Pt_o, out); break;
            case 2: launch_anova<T, 2>(rows, rp, ri, rv, Pt_o, out); break;
            case 3: launch_anova<T, 3>(rows, rp, ri, rv, Pt_o, out); break;
            case 4: launch_anova<T, 4>(rows, rp, ri, rv, Pt_o, out); break;
            case 5: launch_anova<T, 5>(rows, rp, ri, rv, Pt_o, out); break;
            case 6: launch_anova<T, 6>(rows, rp, ri, rv, Pt_o, out); break;
            default: FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
        }
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    // out (device, f64, length rows) = _get_output on the given CSR image
    template <typename T>
    int output_t(int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv, int degree,
                 int fit_linear, int add_lower, double* out) {
        if (rows == 0) return SPFM_OK;
        int rc = ensure_p();
        if (rc) return rc;
        pt_valid = false;  // P is the source of truth here
        rc = ensure_pt();
        if (rc) return rc;
        HIPC(hipMemsetAsync(out, 0, sizeof(double) * (size_t)rows, stream));
        rc = anova_dispatch<T>(kind_of(degree), rows, rp, ri, rv, Pt.as<double>(), out);
        if (rc) return rc;
        if (add_lower) {
            if (n_orders < 2) FAIL(SPFM_ERR_INVALID, "add_lower_deg2 needs P_[1]");
            rc = anova_dispatch<T>(2, rows, rp, ri, rv, Pt.as<double>() + (size_t)k * d, out);
            if (rc) return rc;
        }
        if (fit_linear) {
            hipLaunchKernelGGL((linear_predict_kernel<T>), dim3(cdiv(rows, kBlock)), dim3(kBlock),
                               0, stream, rows, rp, ri, rv, w.as<double>(), out);
            HIPC(hipGetLastError());
        }
        return SPFM_OK;
    }

    template <typename T>
    int init_pred_t(int degree, int fit_linear, int add_lower) {
        int rc = output_t<T>(n, rptr.as<int64_t>(), ridx.as<int32_t>(), rval.as<T>(), degree,
                             fit_linear, add_lower, pred_tmp.as<double>());
        if (rc) return rc;
        if (n > 0) {
            hipLaunchKernelGGL((store_pred_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream, n,
                               pred_tmp.as<double>(), yy.as<T>());
            HIPC(hipGetLastError());
        }
        return sync();
    }

    int init_pred(int degree, int fit_linear, int add_lower) {
        if (!have_data || !have_params) FAIL(SPFM_ERR_INVALID, "init_pred: no data/params");
        have_pred_args = true;
        pa_degree = degree;
        pa_lin = fit_linear;
        pa_lower = add_lower;
        return dtype == SPFM_F32 ? init_pred_t<float>(degree, fit_linear, add_lower)
                                 : init_pred_t<double>(degree, fit_linear, add_lower);
    }

    template <typename T>
    int get_y_pred_t(double* out) {
        if (n == 0) return SPFM_OK;
        hipLaunchKernelGGL((load_pred_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream, n,
                           yy.as<T>(), pred_tmp.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(out, pred_tmp.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost,
                            stream));
        return sync();
    }

    template <typename T>
    int loss_sum_t(double* out) {
        const int nb = 512;
        hipLaunchKernelGGL((loss_partial_kernel<T>), dim3(nb), dim3(kBlock), 0, stream, n,
                           yy.as<typename Vec2<T>::type>(), loss, partial.as<double>());
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(kBlock), 0, stream,
                           partial.as<double>(), nb, scalar.as<double>());
        HIPC(hipGetLastError());
        int rc = allreduce(scalar.as<double>(), 1);
        if (rc) return rc;
        HIPC(hipMemcpyAsync(h_scalar, scalar.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        rc = sync();
        if (rc) return rc;
        *out = h_scalar[0];
        return SPFM_OK;
    }

    template <typename T>
    int predict_csr_t(int64_t rows, const int64_t* indptr, const int32_t* indices,
                      const double* data, int degree, int fit_linear, int add_lower,
                      double* out) {
        if (rows == 0) return SPFM_OK;
        const int64_t nz = indptr[rows];
        for (int64_t ii = 0; ii < nz; ++ii)
            if (indices[ii] < 0 || indices[ii] >= d)
                FAIL(SPFM_ERR_INVALID, "predict: column index out of range");
        std::vector<T> hv((size_t)nz);
        for (int64_t ii = 0; ii < nz; ++ii) hv[(size_t)ii] = (T)data[ii];
        DevBuf rp, ri, rv, o;
        HIPC(rp.alloc(sizeof(int64_t) * ((size_t)rows + 1)));
        HIPC(ri.alloc(sizeof(int32_t) * (size_t)nz));
        HIPC(rv.alloc(sizeof(T) * (size_t)nz));
        HIPC(o.alloc(sizeof(double) * (size_t)rows));
        HIPC(hipMemcpyAsync(rp.p, indptr, sizeof(int64_t) * ((size_t)rows + 1),
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(ri.p, indices, sizeof(int32_t) * (size_t)nz, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(rv.p, hv.data(), sizeof(T) * (size_t)nz, hipMemcpyHostToDevice,
                            stream));
        int rc = output_t<T>(rows, rp.as<int64_t>(), ri.as<int32_t>(), rv.as<T>(), degree,
                             fit_linear, add_lower, o.as<double>());
        if (rc) return rc;
        HIPC(hipMemcpyAsync(out, o.p, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost,
                            stream));
        return sync();
    }

    // ================================================================== epochs
    int epoch_prologue() {
        if (!have_data || !have_params || !configured)
            FAIL(SPFM_ERR_INVALID, "epoch: data, parameters and configuration are required");
        if (!have_schedule) FAIL(SPFM_ERR_INVALID, "epoch: call spfm_set_schedule first");
        int rc = ensure_col_norm();
        if (rc) return rc;
        HIPC(hipMemsetAsync(viol_col.p, 0, sizeof(double) * (size_t)d, stream));
        return SPFM_OK;
    }

    int epoch_epilogue(double* viol) {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(kBlock), 0, stream,
                           viol_col.as<double>(), d, scalar.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(h_scalar, scalar.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        prof_collect();
        if (viol) *viol = h_scalar[0];
        return SPFM_OK;
    }

    static std::string fkey(const char* tag, std::initializer_list<double> v,
                            std::initializer_list<int64_t> iv) {
        std::string s(tag);
        char buf[64];
        for (double x : v) {
            snprintf(buf, sizeof buf, "|%a", x);
            s += buf;
        }
        for (int64_t x : iv) {
            snprintf(buf, sizeof buf, "|%lld", (long long)x);
            s += buf;
        }
        return s;
    }

    // ------------------------------------------------------------- cd_linear
    template <typename T>
    int lin_body(double alpha) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        const int nb = n_batches();
        for (int b = 0; b < nb; ++b) {
            const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
            if (nc == 0) continue;
            const int32_t* cols = d_order.as<int32_t>() + c0;
            prof_begin(4, prof_on ? batch_nnz(b) : 0);
            if (!dist()) {
                hipLaunchKernelGGL((lin_fused_kernel<T>), dim3(nc), dim3(kBlock), 0, stream,
                                   d_desc.as<ColDesc>() + c0, cidx.as<int32_t>(), cval.as<T>(),
                                   yy.as<T>(), loss, w.as<double>(), col_norm.as<double>(), alpha,
                                   mu, viol_col.as<double>());
            } else {
                hipLaunchKernelGGL((lin_grad_kernel<T>), dim3(nc), dim3(kBlock), 0, stream, cols,
                                   cptr.as<int64_t>(), cidx.as<int32_t>(), cval.as<T>(),
                                   yy.as<typename Vec2<T>::type>(), loss, part.as<double>());
                int rc = allreduce(part.as<double>(), (size_t)nc);
                if (rc) return rc;
                hipLaunchKernelGGL((lin_sync_kernel<T>), dim3(nc), dim3(kBlock), 0, stream, cols,
                                   cptr.as<int64_t>(), cidx.as<int32_t>(), cval.as<T>(),
                                   yy.as<T>(), part.as<double>(), w.as<double>(),
                                   col_norm.as<double>(), alpha, mu, viol_col.as<double>());
            }
            prof_end(4);
        }
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    template <typename T, int LOSS>
    int lin_prb(double alpha) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        int rc = ensure_prb<T>();
        if (rc) return rc;
        // row block resident in LDS (4-5 bytes per row: residual, or prediction + label sign)
        constexpr bool can_lr = std::is_same<T, float>::value;
        constexpr int LRV = (LOSS == LOSS_SQUARED) ? 1 : 2;
        // relaxed runs: the merged steps of the reference order, as the pcd passes
        bool relaxed = false;
        if (relax_candidate()) {
            rc = ensure_relax<T>();
            if (rc) return rc;
            relaxed = relax_state == 1;
        }
        const PrbArgs pa = relaxed ? relax_args() : prb_args();
        int lds_max = 0;
        HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        const size_t lds_lr = sizeof(double) * (kPrbLdsFixed + (relaxed ? kPrbLdsCR : 0)) +
                              (size_t)pa.rows_per * 5 + 16;
        const bool use_lr = can_lr && prb_lds && lds_lr <= (size_t)lds_max && (LRV == 1 || y_pm1);
        const size_t lds_bytes = use_lr ? std::max(lds_lr, kPrbLds) : kPrbLds;
        if constexpr (can_lr) {
            if (use_lr)
                HIPC(hipFuncSetAttribute((const void*)lin_prb_kernel<T, LOSS, LRV>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
        }
        HIPC(hipFuncSetAttribute((const void*)lin_prb_kernel<T, LOSS, 0>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPrbLds));
        hipLaunchKernelGGL(gather_sched_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), w.as<double>(), prow_old.as<double>());
        HIPC(hipMemsetAsync(prb_slab.p, 0, prb_slab.bytes, stream));
        if (relaxed) HIPC(hipMemsetAsync(r_cslab.p, 0, r_cslab.bytes, stream));
        {
            int prc = peer_clear(kPeerPcdOff, kPeerPbOff);
            if (prc) return prc;
        }
        prof_begin(4, nnz);
        bool launched = false;
        auto launch_cr = [&](auto lr_tag, size_t lds) -> int {
            constexpr int LRc = decltype(lr_tag)::value;
            auto* fn = lin_prb_kernel<T, LOSS, LRc, false, true>;
            HIPC(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
            if (!resident_ok((const void*)fn, kPrbThreads, lds, prb_G)) return kNotResident;
            hipLaunchKernelGGL(fn, dim3(launch_groups(prb_G)), dim3(kPrbThreads), lds, stream, pa,
                               r_eval.as<T>(), yy.as<T>(), prow_old.as<double>(),
                               prb_cn.as<double>(), w.as<double>(), alpha, mu,
                               prb_viol.as<double>());
            return SPFM_OK;
        };
        auto launch = [&](auto lr_tag, auto mg_tag, size_t lds) -> int {
            constexpr int LRc = decltype(lr_tag)::value;
            constexpr bool MGc = decltype(mg_tag)::value;
            HIPC(hipFuncSetAttribute((const void*)lin_prb_kernel<T, LOSS, LRc, MGc>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (!resident_ok((const void*)lin_prb_kernel<T, LOSS, LRc, MGc>, kPrbThreads, lds, prb_G))
                return kNotResident;
            hipLaunchKernelGGL((lin_prb_kernel<T, LOSS, LRc, MGc>), dim3(launch_groups(prb_G)),
                               dim3(kPrbThreads),
                               lds, stream, pa, prb_eval.as<T>(), yy.as<T>(),
                               prow_old.as<double>(), prb_cn.as<double>(), w.as<double>(), alpha,
                               mu, prb_viol.as<double>());
            return SPFM_OK;
        };
        int lrc = SPFM_OK;
        if (relaxed) {
            launched = true;
            bool done = false;
            if constexpr (can_lr) {
                if (use_lr) {
                    lrc = launch_cr(std::integral_constant<int, LRV>{}, lds_bytes);
                    done = true;
                }
            }
            if (!done) lrc = launch_cr(std::integral_constant<int, 0>{}, kPrbLds);
        }
        if constexpr (can_lr) {
            if (use_lr && !launched) {
                lrc = pa.n_ranks > 1
                          ? launch(std::integral_constant<int, LRV>{}, std::true_type{}, lds_bytes)
                          : launch(std::integral_constant<int, LRV>{}, std::false_type{}, lds_bytes);
                launched = true;
            }
        }
        if (!launched)
            lrc = pa.n_ranks > 1
                      ? launch(std::integral_constant<int, 0>{}, std::true_type{}, kPrbLds)
                      : launch(std::integral_constant<int, 0>{}, std::false_type{}, kPrbLds);
        if (lrc == kNotResident) prof_cancel(4, nnz);
        if (lrc) return lrc;
        prof_end(4);
        hipLaunchKernelGGL(fold_viol_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), prb_viol.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }
    template <typename T>
    int lin_prb_loss(double alpha) {
        switch (loss) {
            case SPFM_LOSS_SQUARED: return lin_prb<T, LOSS_SQUARED>(alpha);
            case SPFM_LOSS_SQUARED_HINGE: return lin_prb<T, LOSS_SQUARED_HINGE>(alpha);
            default: return lin_prb<T, LOSS_LOGISTIC>(alpha);
        }
    }

    void mark_not_resident(const char* what) {
        pers_failed = true;
        pers_fallbacks += 1;
        pers_reason = std::string(what) + ": its workgroups cannot all be resident on this device";
    }

    int cd_linear_epoch(double alpha, double* viol) {
        int rc = epoch_prologue();
        if (rc) return rc;
        const bool wide = wide_usable();
        if (wide || prb_usable()) {
            const char* what = wide ? "wide persistent cd_linear pass" : "persistent cd_linear pass";
            rc = snapshot_state(w.as<double>(), (size_t)d, snapW);
            if (rc) return rc;
            if (wide) rc = dtype == SPFM_F32 ? lin_wide<float>(alpha) : lin_wide<double>(alpha);
            else rc = dtype == SPFM_F32 ? lin_prb_loss<float>(alpha) : lin_prb_loss<double>(alpha);
            if (rc == kNotResident) {  // nothing was launched: the multi-kernel engine takes over
                mark_not_resident(what);
                return cd_linear_epoch(alpha, viol);
            }
            if (rc) return rc;
            rc = epoch_epilogue(viol);
            if (rc) return rc;
            bool aborted = false;
            rc = persistent_aborted(&aborted);
            if (rc) return rc;
            if (aborted) {  // all-or-nothing (cd_linear.py:8-33): back to the epoch's start, redo
                rc = recover_from_abort(w.as<double>(), (size_t)d, snapW, what);
                if (rc) return rc;
                return cd_linear_epoch(alpha, viol);
            }
            return SPFM_OK;
        }
        const std::string key = fkey("lin", {alpha}, {loss, sched_version});
        rc = run_cached(key, [&]() {
            return dtype == SPFM_F32 ? lin_body<float>(alpha) : lin_body<double>(alpha);
        });
        if (rc) return rc;
        return epoch_epilogue(viol);
    }

    // -------------------------------------------------------------------- pcd
    template <typename T, int M>
    int pcd_pass_body(int order_idx, double beta, double gamma, double eta) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        double* Po = P.as<double>() + (size_t)order_idx * k * d;
        Ctl* c = ctl.as<Ctl>();
        double* cbuf[2] = {cache.as<double>(), cache.as<double>() + (kMaxDegree + 2)};
        hipLaunchKernelGGL(begin_pass_kernel, dim3(1), dim3(64), 0, stream, c,
                           comp_order.as<int32_t>(), lams.as<double>());
        const size_t a_stride = (size_t)n * Kind<M>::AS;
        if (reg != SPFM_REG_L1) {
            hipLaunchKernelGGL((pcd_compute_cache_kernel<M>), dim3(kCacheBlocks), dim3(kBlock), 0,
                               stream, c, Po, d, reg, partial.as<double>());
            hipLaunchKernelGGL((pcd_cache_combine_kernel<M>), dim3(1), dim3(64), 0, stream, reg,
                               kCacheBlocks, partial.as<double>(), cbuf[0]);
        }
        const int nb = n_batches();
        int par = 0;  // batch b reads cbuf[par], writes cbuf[par ^ 1]
        for (int b = 0; b < nb; ++b) {
            const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
            if (nc == 0) continue;
            const ColDesc* desc = d_desc.as<ColDesc>() + c0;
            const int64_t bn = prof_on ? batch_nnz(b) : 0;
            prof_begin(0, bn);
            hipLaunchKernelGGL((pcd_grad_kernel<T, M>), dim3(nc), dim3(kBlock), 0, stream, c, desc,
                               cidx.as<int32_t>(), cval.as<T>(), A.as<T>(), a_stride,
                               yy.as<typename Vec2<T>::type>(), Po, d, loss, part.as<double>(),
                               pold.as<double>());
            prof_end(0);
            int rc = allreduce(part.as<double>(), (size_t)2 * nc);
            if (rc) return rc;
            if (nc <= kWave && fuse_chain) {
                prof_begin(1, bn);
                hipLaunchKernelGGL((pcd_chain_sync_kernel<T, M>), dim3(nc), dim3(kBlock), 0, stream,
                                   c, desc, nc, Po, d, part.as<double>(), pold.as<double>(), reg,
                                   cbuf[par], cbuf[par ^ 1], mu, beta, gamma, eta,
                                   cidx.as<int32_t>(), cval.as<T>(), A.as<T>(), a_stride,
                                   yy.as<T>(), viol_col.as<double>());
                prof_end(1);
            } else {
                hipLaunchKernelGGL((pcd_chain_kernel<M>), dim3(1), dim3(kWave), 0, stream, c, desc,
                                   nc, Po, d, part.as<double>(), pold.as<double>(), reg, cbuf[par],
                                   cbuf[par ^ 1], mu, beta, gamma, eta, delta.as<double>(),
                                   viol_col.as<double>());
                prof_begin(1, bn);
                hipLaunchKernelGGL((pcd_sync_kernel<T, M>), dim3(nc), dim3(kBlock), 0, stream, c,
                                   desc, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(), a_stride,
                                   yy.as<T>(), delta.as<double>(), pold.as<double>());
                prof_end(1);
            }
            par ^= 1;
        }
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    // ---------------------------------------------------- persistent row-block pass
    // (several ranks: the persistent passes need the peer-mapped exchange slabs)
    bool prb_usable() const {
        return persistent && !pers_failed && (!dist() || peer_ready) && max_batch_cols <= 64 &&
               nnz < ((int64_t)1 << 31) && n > 0;
    }

    // ---- residency and recovery of the persistent passes
    // All workgroups of a persistent launch must be resident at once (they wait for each other
    // inside the kernel).  `fn` with `threads` threads and `lds` bytes of dynamic LDS: do G
    // workgroups fit the device (times the ranks that share it in a one-GPU rehearsal)?
    bool resident_ok(const void* fn, int threads, size_t lds, int G) {
        int per_cu = 0;
        auto it = resident_cache.find(fn);
        if (it != resident_cache.end()) {
            per_cu = it->second;
        } else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) !=
                hipSuccess) {
                (void)hipGetLastError();
                per_cu = 1;  // unknown: the in-kernel time-out stays the safety net
            }
            resident_cache[fn] = per_cu;
        }
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
            return true;
        // host-shm communicator: ranks on one GPU; co_tenants: handles of this process whose
        // persistent passes run side by side (concurrent fits, one stream each)
        const int64_t sharers = (int64_t)(shm.hdr ? n_ranks : 1) * co_tenants;
        return (int64_t)per_cu * ncu >= (int64_t)G * sharers;
    }
    static constexpr int kNotResident = 1;  // internal: the launch was not made, nothing changed
    int launch_groups(int G) {  // test hook: a launch that lacks its last workgroup times out
        if (debug_drop > 0 && G > 1) {
            --debug_drop;
            return G - 1;
        }
        return G;
    }
    int snapshot_state(const double* params, size_t count, DevBuf& dst) {
        HIPC(dst.alloc(sizeof(double) * count));
        HIPC(hipMemcpyAsync(dst.p, params, sizeof(double) * count, hipMemcpyDeviceToDevice, stream));
        const size_t nc = kMaxDegree + 2;
        HIPC(snapC.alloc(sizeof(double) * nc * 3));
        HIPC(hipMemcpyAsync(snapC.p, cache.p, sizeof(double) * nc * 2, hipMemcpyDeviceToDevice,
                            stream));
        HIPC(hipMemcpyAsync(snapC.as<double>() + nc * 2, dcache.p, sizeof(double) * nc,
                            hipMemcpyDeviceToDevice, stream));
        return SPFM_OK;
    }
    // did the persistent launches of this epoch time out?  With several ranks the answer is
    // agreed on (sum of the flags), so that all of them redo the epoch together.
    int persistent_aborted(bool* out) {
        *out = false;
        if (!prb_abort.p) return SPFM_OK;
        unsigned flag = 0;
        HIPC(hipMemcpyAsync(&flag, prb_abort.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        double any = flag ? 1.0 : 0.0;
        if (dist()) {
            HIPC(hipMemcpyAsync(scalar.as<double>() + 6, &any, sizeof(double),
                                hipMemcpyHostToDevice, stream));
            HIPC(hipStreamSynchronize(stream));
            int rc = allreduce(scalar.as<double>() + 6, 1);
            if (rc) return rc;
            HIPC(hipMemcpyAsync(&any, scalar.as<double>() + 6, sizeof(double),
                                hipMemcpyDeviceToHost, stream));
            HIPC(hipStreamSynchronize(stream));
        }
        if (flag) HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
        *out = any != 0.0;
        return SPFM_OK;
    }
    // after a time-out: parameters and regularizer state back to the epoch's start, y_pred
    // recomputed from them, the persistent passes switched off for this handle
    int recover_from_abort(double* params, size_t count, const DevBuf& src, const char* what) {
        pers_failed = true;
        pers_fallbacks += 1;
        pers_reason = std::string(what) +
                      " timed out waiting for its workgroups (not all resident, or a peer GPU "
                      "did not answer)";
        if (!have_pred_args)
            FAIL(SPFM_ERR_RUNTIME, pers_reason + "; the model is half-updated (no spfm_init_pred "
                                                 "call to recompute y_pred from)");
        const size_t nc = kMaxDegree + 2;
        HIPC(hipMemcpyAsync(params, src.p, sizeof(double) * count, hipMemcpyDeviceToDevice, stream));
        HIPC(hipMemcpyAsync(cache.p, snapC.p, sizeof(double) * nc * 2, hipMemcpyDeviceToDevice,
                            stream));
        HIPC(hipMemcpyAsync(dcache.p, snapC.as<double>() + nc * 2, sizeof(double) * nc,
                            hipMemcpyDeviceToDevice, stream));
        clear_graphs();
        return init_pred(pa_degree, pa_lin, pa_lower);
    }

    template <typename T>
    int ensure_prb() {
        if (prb_ready) return SPFM_OK;
        int ncu = 0;
        HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        if (prb_G > ncu) prb_G = ncu;
        if (prb_G < 1) prb_G = 1;
        // the entry stream: on the device (spfm_ingest.hip device_rowblock_stream: two binary
        // searches per (column, row block), a scan, a fill -- the host builder's sp / src / lmask
        // exactly, tests/test_hip_stream.py), or by the host threads (stream_device=0, no room)
        const int nb_ = n_batches();
        const size_t nsp = (size_t)prb_G * nb_ * 65 + 1;
        DevBuf d_src;
        HIPC(d_src.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        HIPC(prb_sp.alloc(sizeof(int32_t) * nsp));
        HIPC(prb_lmask.alloc(sizeof(uint32_t) * (size_t)prb_G * nb_ * 2));
        stream_device_used = 0;
        if (stream_device && nnz >= (1 << 20)) {
            int hl = 0;
            const hipError_t e = device_rowblock_stream(
                n, d, nnz, prb_G, nb_, prb_long, d_order.as<int32_t>(), d_bptr.as<int32_t>(),
                cptr.as<int64_t>(), cidx.as<int32_t>(), nullptr, prb_sp.as<int32_t>(),
                d_src.as<int32_t>(), prb_lmask.as<uint32_t>(), &hl, nullptr, stream);
            if (e == hipSuccess) {
                prb_has_long = hl;
                stream_device_used = 1;
            } else {
                (void)hipGetLastError();
            }
        }
        std::vector<int32_t> sp, src;
        std::vector<uint32_t> lmask;
        if (!stream_device_used) {
            build_rowblock_stream(n, h_cptr.data(), h_cidx.data(), order, batch_ptr, prb_G,
                                  prb_long, sp, src, lmask, nullptr);
            prb_has_long = 0;
            for (uint32_t m : lmask) prb_has_long |= (m != 0u);
            HIPC(hipMemcpyAsync(prb_lmask.p, lmask.data(), sizeof(uint32_t) * lmask.size(),
                                hipMemcpyHostToDevice, stream));
        }
        HIPC(prb_erow.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        HIPC(prb_eval.alloc(sizeof(T) * (size_t)(nnz > 0 ? nnz : 1)));
        HIPC(prb_slab.alloc(sizeof(double) * 2 * ((size_t)prb_G + 1) * 64 * 2));
        HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
        HIPC(prow_old.alloc(sizeof(double) * (size_t)d));
        HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
        HIPC(prb_cn.alloc(sizeof(double) * (size_t)d));
        HIPC(prb_stamps.alloc(sizeof(long long) * 16 * (size_t)prb_G));
        HIPC(hipMemsetAsync(prb_stamps.p, 0, prb_stamps.bytes, stream));
        HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
        HIPC(hipMemsetAsync(prb_slab.p, 0, prb_slab.bytes, stream));
        if (!stream_device_used)
            HIPC(hipMemcpyAsync(prb_sp.p, sp.data(), sizeof(int32_t) * sp.size(),
                                hipMemcpyHostToDevice, stream));
        if (nnz > 0) {
            if (!stream_device_used)
                HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * (size_t)nnz,
                                    hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL((prb_gather_kernel<T>), dim3(cdiv(nnz, 256)), dim3(256), 0, stream,
                               nnz, d_src.as<int32_t>(), cidx.as<int32_t>(), cval.as<T>(),
                               prb_erow.as<int32_t>(), prb_eval.as<T>());
            HIPC(hipGetLastError());
        }
        hipLaunchKernelGGL(gather_sched_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), col_norm.as<double>(), prb_cn.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(stream));
        prb_ready = true;
        return SPFM_OK;
    }

    PrbArgs prb_args() {
        PrbArgs a;
        a.G = prb_G;
        a.nb = n_batches();
        a.bptr = d_bptr.as<int32_t>();
        a.desc = d_desc.as<ColDesc>();
        a.sp = prb_sp.as<int32_t>();
        a.lmask = prb_lmask.as<uint32_t>();
        a.has_long = prb_has_long;
        a.erow = prb_erow.as<int32_t>();
        a.slab = prb_slab.as<double>();
        a.rows_per = (int)std::max<int64_t>((n + prb_G - 1) / prb_G, 1);
        a.n_rows = (int)n;
        a.abort_flag = prb_abort.as<unsigned>();
        a.spin_max = spin_max;
        a.stamps = prb_stamp_on ? prb_stamps.as<long long>() : nullptr;
        a.n_ranks = peer_ready ? n_ranks : 1;
        a.rank = rank;
        a.xslab = peer_ready ? peer_tab_pcd.as<double*>() : nullptr;
        a.cf_ptr = nullptr;
        a.cf = nullptr;
        a.clist = nullptr;
        a.cslab = nullptr;
        a.rec = nullptr;
        return a;
    }

    // ---- relaxed runs for schedules of tiny steps (degree-2 pcd pass, one GPU)
    // worth trying: the persistent 64-column pass is in use and the strict steps are narrow
    bool relax_candidate() const {
        return relax_on && prb_usable() && !dist() && n_batches() > 0 &&
               (double)d / (double)n_batches() < 12.0;
    }
    template <typename T>
    int ensure_relax() {
        if (relax_state != 0) return SPFM_OK;
        relax_state = -1;
        int rc = ensure_prb<T>();  // workgroup count, col_norm / viol buffers, abort word
        if (rc) return rc;
        std::vector<int32_t> cf_ptr, cf_row, cf_qq, sp, src;
        std::vector<int64_t> cf_ia, cf_ib;
        std::vector<int16_t> clist;
        std::vector<uint8_t> skip;
        std::vector<uint32_t> lmask;
        schedule_relax(n, d, h_cptr.data(), h_cidx.data(), order.data(), 64, 64, r_batch_ptr, cf_ptr,
                       cf_row, cf_qq, cf_ia, cf_ib, clist, skip);
        const int nbr = (int)r_batch_ptr.size() - 1;
        if ((double)nbr > 0.6 * (double)n_batches()) return SPFM_OK;  // not worth a second stream
        // the merged steps' entry stream (without the entries on conflict rows): on the device like
        // the strict one (ensure_prb), or by the host builder
        DevBuf d_src;
        HIPC(r_bptr.alloc(sizeof(int32_t) * r_batch_ptr.size()));
        HIPC(hipMemcpyAsync(r_bptr.p, r_batch_ptr.data(), sizeof(int32_t) * r_batch_ptr.size(),
                            hipMemcpyHostToDevice, stream));
        const size_t nsp_r = (size_t)prb_G * nbr * 65 + 1;
        bool dev_stream = false;
        size_t ne = 0;
        if (stream_device && nnz >= (1 << 20)) {
            DevBuf d_skip;
            HIPC(d_skip.alloc((size_t)nnz));
            HIPC(hipMemcpyAsync(d_skip.p, skip.data(), (size_t)nnz, hipMemcpyHostToDevice, stream));
            HIPC(d_src.alloc(sizeof(int32_t) * (size_t)nnz));
            HIPC(r_sp.alloc(sizeof(int32_t) * nsp_r));
            HIPC(r_lmask.alloc(sizeof(uint32_t) * (size_t)prb_G * nbr * 2));
            int hl = 0;
            int64_t tot = 0;
            const hipError_t e = device_rowblock_stream(
                n, d, nnz, prb_G, nbr, prb_long, d_order.as<int32_t>(), r_bptr.as<int32_t>(),
                cptr.as<int64_t>(), cidx.as<int32_t>(), d_skip.as<uint8_t>(), r_sp.as<int32_t>(),
                d_src.as<int32_t>(), r_lmask.as<uint32_t>(), &hl, &tot, stream);
            if (e == hipSuccess) {
                relax_has_long = hl;
                ne = (size_t)tot;
                dev_stream = true;
            } else {
                (void)hipGetLastError();
            }
        }
        if (!dev_stream) {
            build_rowblock_stream(n, h_cptr.data(), h_cidx.data(), order, r_batch_ptr, prb_G,
                                  prb_long, sp, src, lmask, skip.data());
            relax_has_long = 0;
            for (uint32_t m : lmask) relax_has_long |= (m != 0u);
            ne = src.size();
        }
        const size_t ncf = cf_row.size();
        // the conflict rows' x values, in the storage type
        std::vector<PrbConf<T>> hcf(ncf ? ncf : 1);
        {
            std::vector<T> hv((size_t)(nnz > 0 ? nnz : 1));
            // (on the handle's own stream: a copy on the null stream would create that stream,
            // which then holds one of the process's few hardware queues for good -- and
            // concurrent fits, one stream each, end up two to a queue)
            HIPC(hipMemcpyAsync(hv.data(), cval.p, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost,
                                stream));
            HIPC(hipStreamSynchronize(stream));
            for (size_t c = 0; c < ncf; ++c) {
                hcf[c].row = cf_row[c];
                hcf[c].qq = cf_qq[c];
                hcf[c].xa = hv[(size_t)cf_ia[c]];
                hcf[c].xb = hv[(size_t)cf_ib[c]];
            }
        }
        if (!dev_stream) {
            HIPC(d_src.alloc(sizeof(int32_t) * (ne ? ne : 1)));
            HIPC(r_sp.alloc(sizeof(int32_t) * sp.size()));
            HIPC(r_lmask.alloc(sizeof(uint32_t) * lmask.size()));
        }
        HIPC(r_erow.alloc(sizeof(int32_t) * (ne ? ne : 1)));
        HIPC(r_eval.alloc(sizeof(T) * (ne ? ne : 1)));
        HIPC(r_cfptr.alloc(sizeof(int32_t) * cf_ptr.size()));
        HIPC(r_cf.alloc(sizeof(PrbConf<T>) * hcf.size()));
        HIPC(r_clist.alloc(sizeof(int16_t) * clist.size() + 16));
        HIPC(r_cslab.alloc(sizeof(double) * 2 * 64 * 8));
        if (!dev_stream) {
            HIPC(hipMemcpyAsync(r_sp.p, sp.data(), sizeof(int32_t) * sp.size(), hipMemcpyHostToDevice,
                                stream));
            HIPC(hipMemcpyAsync(r_lmask.p, lmask.data(), sizeof(uint32_t) * lmask.size(),
                                hipMemcpyHostToDevice, stream));
        }
        HIPC(hipMemcpyAsync(r_cfptr.p, cf_ptr.data(), sizeof(int32_t) * cf_ptr.size(),
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(r_cf.p, hcf.data(), sizeof(PrbConf<T>) * hcf.size(),
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(r_clist.p, clist.data(), sizeof(int16_t) * clist.size(),
                            hipMemcpyHostToDevice, stream));
        if (ne > 0) {
            if (!dev_stream)
                HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * ne, hipMemcpyHostToDevice,
                                    stream));
            hipLaunchKernelGGL((prb_gather_kernel<T>), dim3(cdiv((int64_t)ne, 256)), dim3(256), 0,
                               stream, (int64_t)ne, d_src.as<int32_t>(), cidx.as<int32_t>(),
                               cval.as<T>(), r_erow.as<int32_t>(), r_eval.as<T>());
            HIPC(hipGetLastError());
        }
        HIPC(hipStreamSynchronize(stream));
        relax_state = 1;
        return SPFM_OK;
    }
    PrbArgs relax_args() {
        PrbArgs a = prb_args();
        a.nb = (int)r_batch_ptr.size() - 1;
        a.bptr = r_bptr.as<int32_t>();
        a.sp = r_sp.as<int32_t>();
        a.lmask = r_lmask.as<uint32_t>();
        a.has_long = relax_has_long;
        a.erow = r_erow.as<int32_t>();
        a.stamps = nullptr;
        a.cf_ptr = r_cfptr.as<int32_t>();
        a.cf = r_cf.p;
        a.clist = r_clist.as<int16_t>();
        a.cslab = r_cslab.as<double>();
        return a;
    }

    template <typename T, int M, int LOSS>
    int pcd_pass_prb(int order_idx, double beta, double gamma, double eta) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        int rc = ensure_prb<T>();
        if (rc) return rc;
        double* Po = P.as<double>() + (size_t)order_idx * k * d;
        Ctl* c = ctl.as<Ctl>();
        double* cb = cache.as<double>();
        // row block resident in LDS when the variant exists and fits (squared loss: A[i,1..AS] +
        // residual, 8 / 12 bytes per row; +-1 targets: A + yhat + sign, 9 / 13 bytes)
        constexpr bool can_lr = std::is_same<T, float>::value && Kind<M>::AS <= 2;
        constexpr int LRV = (LOSS == LOSS_SQUARED) ? 1 : 2;
        // the in-kernel phase timers exist as a separate instantiation of ONE configuration
        // (float, degree 2, squared loss, rows in LDS): tools/prb_stamp_probe.py
        constexpr bool can_stamp = std::is_same<T, float>::value && M == 2 && LOSS == LOSS_SQUARED;
        // ... and of (float, degree 3, squared loss, rows in global memory)
        constexpr bool can_stamp3 = std::is_same<T, float>::value && M == 3 && LOSS == LOSS_SQUARED;
        if (prb_stamp_on && !((can_stamp && prb_lds) || (can_stamp3 && !prb_lds)))
            FAIL(SPFM_ERR_UNSUPPORTED,
                 "prb_stamps: built for float storage, squared loss and degree 2 with prb_lds=1 "
                 "or degree 3 with prb_lds=0 only");
        // relaxed runs: a schedule of tiny steps -- the reference order -- is run
        // as merged steps of ~20 columns by the CR instantiation (degree 2, one GPU)
        bool relaxed = false;
        if constexpr (M == 2 || M == 3) {
            if (relax_candidate() && !prb_stamp_on) {
                rc = ensure_relax<T>();
                if (rc) return rc;
                relaxed = relax_state == 1;
            }
        }
        PrbArgs pa = relaxed ? relax_args() : prb_args();
        // degree 3, float storage, rows in global memory: packed 16-byte row records
        // (yhat, y, A[i,1], A[i,2]) for the pass's component (pcd_prb_kernel LR = 3)
        constexpr bool can_pk = std::is_same<T, float>::value && M == 3;
        bool packed = false;
        if (prb_stamp_on && pa.n_ranks > 1)
            FAIL(SPFM_ERR_UNSUPPORTED,
                 "prb_stamps: the timer instantiation has no cross-GPU stage (single rank only)");
        int lds_max = 0;
        HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        const size_t lds_lr = sizeof(double) * (kPrbLdsFixed + (relaxed ? kPrbLdsCR : 0)) +
                              (size_t)pa.rows_per * (4 * Kind<M>::AS + (LRV == 1 ? 4 : 5)) + 16;
        const bool use_lr = can_lr && prb_lds && lds_lr <= (size_t)lds_max &&
                            (LRV == 1 || y_pm1);
        const size_t lds_bytes = use_lr ? std::max(lds_lr, kPrbLds) : kPrbLds;
        prb_lds_active = use_lr ? LRV : 0;
        hipLaunchKernelGGL(begin_pass_kernel, dim3(1), dim3(64), 0, stream, c,
                           comp_order.as<int32_t>(), lams.as<double>());
        if (reg != SPFM_REG_L1) {
            hipLaunchKernelGGL((pcd_compute_cache_kernel<M>), dim3(kCacheBlocks), dim3(kBlock), 0,
                               stream, c, Po, d, reg, partial.as<double>());
            hipLaunchKernelGGL((pcd_cache_combine_kernel<M>), dim3(1), dim3(64), 0, stream, reg,
                               kCacheBlocks, partial.as<double>(), cb);
        }
        hipLaunchKernelGGL(snapshot_row_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, c, Po, d,
                           d_desc.as<ColDesc>(), prow_old.as<double>());
        HIPC(hipMemsetAsync(prb_slab.p, 0, prb_slab.bytes, stream));  // tag 0 = "not yet"
        if (relaxed) HIPC(hipMemsetAsync(r_cslab.p, 0, r_cslab.bytes, stream));
        if constexpr (can_pk) {
            if (!use_lr && !prb_stamp_on && prb_pack) {
                HIPC(prb_rec.alloc(sizeof(float) * 4 * (size_t)n));
                pa.rec = prb_rec.p;
                hipLaunchKernelGGL(prb_pack3_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, c, n,
                                   (size_t)n * 2, yy.as<float>(), A.as<float>(),
                                   prb_rec.as<float4>());
                packed = true;
            }
        }
        {
            int prc = peer_clear(kPeerPcdOff, kPeerPbOff);
            if (prc) return prc;
        }
        prof_begin(0, nnz);
        // one instantiation per (rows in LDS?, timers?, regularizer): the fast configuration
        // (float, one cache value per row) gets the regularizer as a compile-time constant
        auto go = [&](auto lr_tag, auto stamp_tag, auto reg_tag) -> int {
            constexpr int LRc = decltype(lr_tag)::value;
            constexpr bool STc = decltype(stamp_tag)::value;
            constexpr int RGc = decltype(reg_tag)::value;
            const size_t lds = (LRc == 1 || LRc == 2) ? lds_bytes : kPrbLds;
            auto launch = [&](auto mg_tag) -> int {
                constexpr bool MGc = decltype(mg_tag)::value;
                HIPC(hipFuncSetAttribute(
                    (const void*)pcd_prb_kernel<T, M, LOSS, LRc, STc, RGc, MGc>,
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (!resident_ok((const void*)pcd_prb_kernel<T, M, LOSS, LRc, STc, RGc, MGc>,
                                 kPrbThreads, lds, prb_G))
                    return kNotResident;
                hipLaunchKernelGGL((pcd_prb_kernel<T, M, LOSS, LRc, STc, RGc, MGc>),
                                   dim3(launch_groups(prb_G)), dim3(kPrbThreads), lds, stream, c, pa, prb_eval.as<T>(),
                                   A.as<T>(), (size_t)n * Kind<M>::AS, yy.as<T>(),
                                   prow_old.as<double>(), Po, d, reg, cb, mu, beta, gamma, eta,
                                   prb_viol.as<double>());
                return SPFM_OK;
            };
            if constexpr (STc) {  // timers: single GPU only (refused above for several ranks)
                return launch(std::false_type{});
            } else {
                return pa.n_ranks > 1 ? launch(std::true_type{}) : launch(std::false_type{});
            }
        };
        using std::integral_constant;
        int lrc = SPFM_OK;
        bool launched = false;
        if constexpr (M == 2 || M == 3) {
            if (relaxed) {
                auto launch_cr = [&](auto lr_tag) -> int {
                    constexpr int LRc = decltype(lr_tag)::value;
                    const size_t lds = (LRc == 1 || LRc == 2) ? lds_bytes : kPrbLds;
                    auto* fn = pcd_prb_kernel<T, M, LOSS, LRc, false, -1, false, true>;
                    HIPC(hipFuncSetAttribute((const void*)fn,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    if (!resident_ok((const void*)fn, kPrbThreads, lds, prb_G)) return kNotResident;
                    hipLaunchKernelGGL(fn, dim3(launch_groups(prb_G)), dim3(kPrbThreads), lds, stream,
                                       c, pa, r_eval.as<T>(), A.as<T>(), (size_t)n * Kind<M>::AS,
                                       yy.as<T>(), prow_old.as<double>(), Po, d, reg, cb, mu, beta,
                                       gamma, eta, prb_viol.as<double>());
                    return SPFM_OK;
                };
                launched = true;
                bool done = false;
                if constexpr (can_lr) {
                    if (use_lr) {
                        lrc = launch_cr(integral_constant<int, LRV>{});
                        done = true;
                    }
                }
                if constexpr (can_pk) {
                    if (!done && packed) {
                        lrc = launch_cr(integral_constant<int, 3>{});
                        done = true;
                    }
                }
                if (!done) lrc = launch_cr(integral_constant<int, 0>{});
            }
        }
        if constexpr (can_lr) {
            if (use_lr && !launched) {
                launched = true;
                if constexpr (can_stamp) {
                    if (prb_stamp_on)
                        lrc = go(integral_constant<int, LRV>{}, std::true_type{},
                                 integral_constant<int, -1>{});
                }
                if (!prb_stamp_on) {
                    if (reg == SPFM_REG_L1)
                        lrc = go(integral_constant<int, LRV>{}, std::false_type{},
                                 integral_constant<int, REG_L1>{});
                    else if (reg == SPFM_REG_OMEGATI)
                        lrc = go(integral_constant<int, LRV>{}, std::false_type{},
                                 integral_constant<int, REG_OMEGATI>{});
                    else if constexpr (M == 2)
                        lrc = go(integral_constant<int, LRV>{}, std::false_type{},
                                 integral_constant<int, REG_SQL12>{});
                    else
                        lrc = go(integral_constant<int, LRV>{}, std::false_type{},
                                 integral_constant<int, -1>{});
                }
            }
        }
        if constexpr (can_pk) {
            if (!launched && packed) {
                launched = true;
                lrc = go(integral_constant<int, 3>{}, std::false_type{}, integral_constant<int, -1>{});
            }
        }
        if constexpr (can_stamp3) {
            if (!launched && prb_stamp_on) {
                launched = true;
                lrc = go(integral_constant<int, 0>{}, std::true_type{}, integral_constant<int, -1>{});
            }
        }
        if (!launched)
            lrc = go(integral_constant<int, 0>{}, std::false_type{}, integral_constant<int, -1>{});
        prb_pack_active = packed ? 1 : 0;
        if (lrc == kNotResident) prof_cancel(0, nnz);
        if (lrc) return lrc;
        prof_end(0);
        if constexpr (can_pk) {
            if (packed)
                hipLaunchKernelGGL(prb_unpack3_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, c, n,
                                   (size_t)n * 2, prb_rec.as<float4>(), yy.as<float>(),
                                   A.as<float>());
        }
        hipLaunchKernelGGL(fold_viol_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), prb_viol.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    template <typename T, int M>
    int pcd_prb_loss(int order_idx, double beta, double gamma, double eta) {
        switch (loss) {
            case SPFM_LOSS_SQUARED:
                return pcd_pass_prb<T, M, LOSS_SQUARED>(order_idx, beta, gamma, eta);
            case SPFM_LOSS_SQUARED_HINGE:
                return pcd_pass_prb<T, M, LOSS_SQUARED_HINGE>(order_idx, beta, gamma, eta);
            default:
                return pcd_pass_prb<T, M, LOSS_LOGISTIC>(order_idx, beta, gamma, eta);
        }
    }

    template <typename T>
    int pcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
        switch (M) {
            case 0: return pcd_prb_loss<T, 0>(order_idx, beta, gamma, eta);
            case 2: return pcd_prb_loss<T, 2>(order_idx, beta, gamma, eta);
            case 3: return pcd_prb_loss<T, 3>(order_idx, beta, gamma, eta);
            case 4: return pcd_prb_loss<T, 4>(order_idx, beta, gamma, eta);
            case 5: return pcd_prb_loss<T, 5>(order_idx, beta, gamma, eta);
            case 6: return pcd_prb_loss<T, 6>(order_idx, beta, gamma, eta);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
    }

    // tag 0 = "not yet" in the own exchange slab; all ranks must have passed the previous launch
    // before anybody clears (the caller's epochs are collective: one clear per launch, and a
    // rank only starts writing to a peer after that peer's clear because the first remote store
    // of a launch follows a local sweep that needs ... nothing remote).  To be safe the clear is
    // followed by a barrier over the host communicator.
    int peer_clear(size_t off_doubles, size_t n_doubles) {
        if (!peer_ready) return SPFM_OK;
        HIPC(hipMemsetAsync(reinterpret_cast<double*>(peer_own) + off_doubles, 0,
                            sizeof(double) * n_doubles, stream));
        HIPC(hipStreamSynchronize(stream));
        return host_barrier();
    }
    int host_barrier() {
        if (shm.hdr) return shm_barrier();
        if (comm) {  // a 1-element all-reduce doubles as the barrier
            int rc = allreduce(scalar.as<double>() + 4, 1);
            if (rc) return rc;
            HIPC(hipStreamSynchronize(stream));
        }
        return SPFM_OK;
    }

    // ------------------------------------------------------ wide persistent passes
    bool wide_usable() const {
        return persistent && !pers_failed && wide_on && (!dist() || peer_ready) &&
               max_batch_cols > 64 && max_batch_cols <= 512 && nnz < ((int64_t)1 << 31) && n > 0;
    }

    // Workgroups of the wide pass.  Its exchange (reduce-scatter to slot owners + all-gather)
    // grows with the workgroup count, the entry loops shrink with it; measured optimum: about 160
    // entries per workgroup and step (the shard of a multi-GPU run -- 1.25 M rows of the
    // 10M x 1M problem, 22.7 k entries per step: 23.9 ms per component pass at 128 workgroups,
    // 27.3 at 256; 2M x 200k, 36.5 k entries per step: 7.5 us per step at 256, 8.7 at 128).  More
    // workgroups than that when the rows fit LDS only then.  An explicit "pcdw_groups" wins;
    // concurrent tenants keep to their share of the CUs.
    int wide_groups(int ncu, size_t lds_max) const {
        int g = pcdw_G;
        if (g <= 0) {
            const int64_t steps = std::max<int64_t>(1, (int64_t)batch_ptr.size() - 1);
            const int64_t per_step = nnz / steps;
            g = (int)std::min<int64_t>(ncu, std::max<int64_t>(64, ((per_step / 160 + 15) / 16) * 16));
            auto fits = [&](int gg) {
                const size_t rows_per = ((size_t)n + (size_t)gg - 1) / (size_t)gg;
                return kPcdwLdsFixed + rows_per * 8 + 16 <= lds_max;
            };
            if (!fits(g) && fits(ncu))
                while (g < ncu && !fits(g)) g = std::min(ncu, g + 16);
        }
        g = std::min(g, std::max(1, ncu / co_tenants));
        return std::max(1, std::min(g, ncu));
    }

    template <typename T>
    int ensure_wide() {
        int ncu = 0, lds_max = 0;
        HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        const int G = wide_groups(ncu, (size_t)lds_max);
        if (wide_ready && wide_G == G) return SPFM_OK;
        std::vector<int32_t> wbase, wsp, src;
        std::vector<uint8_t> hz;
        build_wide_stream(n, h_cptr.data(), h_cidx.data(), order, batch_ptr, G, wbase, wsp, src, hz);
        DevBuf d_src, d_hz;
        HIPC(d_src.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        HIPC(d_hz.alloc((size_t)(nnz > 0 ? nnz : 1)));
        HIPC(w_wbase.alloc(sizeof(int32_t) * wbase.size()));
        HIPC(w_wsp.alloc(sizeof(int32_t) * wsp.size()));
        HIPC(w_erow.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1) + 64));
        HIPC(w_eval.alloc(sizeof(T) * (size_t)(nnz > 0 ? nnz : 1) + 64));
        HIPC(w_slabA.alloc(sizeof(double) * 2 * 32 * (size_t)G * 32));
        HIPC(w_slabB.alloc(sizeof(double) * 2 * 32 * 32));
        HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
        HIPC(prow_old.alloc(sizeof(double) * (size_t)d));
        HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
        HIPC(prb_cn.alloc(sizeof(double) * (size_t)d));
        HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
        HIPC(hipMemcpyAsync(w_wbase.p, wbase.data(), sizeof(int32_t) * wbase.size(),
                            hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(w_wsp.p, wsp.data(), sizeof(int32_t) * wsp.size(),
                            hipMemcpyHostToDevice, stream));
        if (nnz > 0) {
            HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemcpyAsync(d_hz.p, hz.data(), (size_t)nnz, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL((pcdw_gather_kernel<T>), dim3(cdiv(nnz, 256)), dim3(256), 0, stream,
                               nnz, d_src.as<int32_t>(), d_hz.as<uint8_t>(), cidx.as<int32_t>(),
                               cval.as<T>(), w_erow.as<int32_t>(), w_eval.as<T>());
            HIPC(hipGetLastError());
        }
        hipLaunchKernelGGL(gather_sched_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), col_norm.as<double>(), prb_cn.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(stream));
        wide_G = G;
        wide_tot = (int)wbase.back();
        wide_ready = true;
        return SPFM_OK;
    }

    PcdwArgs wide_args() {
        PcdwArgs a;
        a.G = wide_G;
        a.nb = n_batches();
        a.bptr = d_bptr.as<int32_t>();
        a.jsched = d_order.as<int32_t>();
        a.wbase = w_wbase.as<int32_t>();
        a.wsp = w_wsp.as<int32_t>();
        a.tot = wide_tot;
        a.erow = w_erow.as<int32_t>();
        a.slabA = w_slabA.as<double>();
        a.slabB = w_slabB.as<double>();
        a.rows_per = (int)std::max<int64_t>((n + wide_G - 1) / wide_G, 1);
        a.n_rows = (int)n;
        a.abort_flag = prb_abort.as<unsigned>();
        a.spin_max = spin_max;
        a.n_ranks = peer_ready ? n_ranks : 1;
        a.rank = rank;
        a.slabC = peer_ready ? peer_tab_pb.as<double*>() : nullptr;
        return a;
    }

    // one launch: KIND 0 = a pcd component pass (degree 2), 1 = the cd_linear epoch
    template <typename T, int KIND>
    int wide_launch(PcdwArgs& a, PcdwParams& pp, T* Aptr) {
        int lds_max = 0;
        HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        constexpr bool can_lr = std::is_same<T, float>::value;
        const size_t lds_lr = kPcdwLdsFixed + (size_t)a.rows_per * (KIND == 0 ? 8 : 4) + 16;
        const bool use_lr = can_lr && prb_lds && loss == SPFM_LOSS_SQUARED && lds_lr <= (size_t)lds_max;
        wide_lr_active = use_lr ? 1 : 0;
        HIPC(hipMemsetAsync(w_slabA.p, 0, w_slabA.bytes, stream));
        HIPC(hipMemsetAsync(w_slabB.p, 0, w_slabB.bytes, stream));
        {
            int prc = peer_clear(kPeerPbOff, kPeerProbeOff - kPeerPbOff);
            if (prc) return prc;
        }
        a.stamps = nullptr;
        // pcd with the rows in global memory: packed row records (see PcdwRec)
        PcdwRec<T>* rec = nullptr;
        const bool packed = KIND == 0 && !use_lr;
        if (packed) {
            HIPC(w_rec.alloc(sizeof(PcdwRec<T>) * (size_t)n));
            rec = w_rec.as<PcdwRec<T>>();
            hipLaunchKernelGGL((pcdw_pack_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream,
                               pp.ctl, n, pp.a_stride, yy.as<T>(), Aptr, rec);
            HIPC(hipGetLastError());
        }
        auto unpack = [&]() -> int {
            if (packed) {
                hipLaunchKernelGGL((pcdw_unpack_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0,
                                   stream, pp.ctl, n, pp.a_stride, rec, yy.as<T>(), Aptr);
                HIPC(hipGetLastError());
            }
            return SPFM_OK;
        };
        // one launch site: residency check, test hook, launch
        auto fire = [&](auto* fn, size_t lds) -> int {
            HIPC(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
            if (!resident_ok((const void*)fn, kPcdwThreads, lds, a.G)) return kNotResident;
            hipLaunchKernelGGL(fn, dim3(launch_groups(a.G)), dim3(kPcdwThreads), lds, stream, a, pp,
                               w_eval.as<T>(), Aptr, yy.as<T>(), rec);
            HIPC(hipGetLastError());
            return SPFM_OK;
        };
        if constexpr (can_lr && KIND == 0) {
            if (wide_stamp_on) {  // diagnostic instantiations (tools/pcdw_stamp_probe.py)
                HIPC(wide_stamps.alloc(sizeof(long long) * 16 * (size_t)a.G));
                HIPC(hipMemsetAsync(wide_stamps.p, 0, wide_stamps.bytes, stream));
                a.stamps = wide_stamps.as<long long>();
                const size_t lds = use_lr ? std::max(lds_lr, kPrbLds) : kPrbLds;
                int frc = use_lr ? fire(&pcdw_kernel<T, KIND, 1, true>, lds)
                                 : fire(&pcdw_kernel<T, KIND, 0, true>, lds);
                if (frc) return frc;
                return unpack();
            }
        }
        if constexpr (can_lr) {
            if (use_lr) return fire(&pcdw_kernel<T, KIND, 1>, std::max(lds_lr, kPrbLds));
        }
        int frc = fire(&pcdw_kernel<T, KIND, 0>, kPrbLds);
        if (frc) return frc;
        return unpack();
    }

    template <typename T>
    int pcd_pass_wide(int order_idx, double beta, double gamma, double eta) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        int rc = ensure_wide<T>();
        if (rc) return rc;
        double* Po = P.as<double>() + (size_t)order_idx * k * d;
        Ctl* c = ctl.as<Ctl>();
        hipLaunchKernelGGL(begin_pass_kernel, dim3(1), dim3(64), 0, stream, c,
                           comp_order.as<int32_t>(), lams.as<double>());
        if (reg != SPFM_REG_L1) {
            hipLaunchKernelGGL((pcd_compute_cache_kernel<2>), dim3(kCacheBlocks), dim3(kBlock), 0,
                               stream, c, Po, d, reg, partial.as<double>());
            hipLaunchKernelGGL((pcd_cache_combine_kernel<2>), dim3(1), dim3(64), 0, stream, reg,
                               kCacheBlocks, partial.as<double>(), cache.as<double>());
        }
        hipLaunchKernelGGL(snapshot_row_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, c, Po, d,
                           d_desc.as<ColDesc>(), prow_old.as<double>());
        PcdwArgs a = wide_args();
        PcdwParams pp;
        pp.ctl = c;
        pp.a_stride = (size_t)n;
        pp.P = Po;
        pp.d = d;
        pp.reg = reg;
        pp.loss = loss;
        pp.cache_in = cache.as<double>();
        pp.mu = mu;
        pp.beta = beta;
        pp.gamma = gamma;
        pp.eta = eta;
        pp.alpha = 0.0;
        pp.sched0 = prow_old.as<double>();
        pp.sched1 = nullptr;
        pp.wout = nullptr;
        pp.viol_pos = prb_viol.as<double>();
        prof_begin(0, nnz);
        rc = wide_launch<T, 0>(a, pp, A.as<T>());
        if (rc == kNotResident) prof_cancel(0, nnz);
        if (rc) return rc;
        prof_end(0);
        hipLaunchKernelGGL(fold_viol_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), prb_viol.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    template <typename T>
    int lin_wide(double alpha) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        int rc = ensure_wide<T>();
        if (rc) return rc;
        hipLaunchKernelGGL(gather_sched_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), w.as<double>(), prow_old.as<double>());
        PcdwArgs a = wide_args();
        PcdwParams pp;
        pp.ctl = ctl.as<Ctl>();
        pp.a_stride = 0;
        pp.P = nullptr;
        pp.d = d;
        pp.reg = 0;
        pp.loss = loss;
        pp.cache_in = nullptr;
        pp.mu = mu;
        pp.beta = pp.gamma = pp.eta = 0.0;
        pp.alpha = alpha;
        pp.sched0 = prow_old.as<double>();
        pp.sched1 = prb_cn.as<double>();
        pp.wout = w.as<double>();
        pp.viol_pos = prb_viol.as<double>();
        prof_begin(4, nnz);
        rc = wide_launch<T, 1>(a, pp, (T*)nullptr);
        if (rc == kNotResident) prof_cancel(4, nnz);
        if (rc) return rc;
        prof_end(4);
        hipLaunchKernelGGL(fold_viol_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), prb_viol.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    // template parameter for a reference degree: -1 (all-subsets) -> 0
    static int kind_of(int degree) { return degree == -1 ? 0 : degree; }
    bool degree_ok(int degree) const {
        return top_degree == -1 ? degree == -1 : (degree >= 2 && degree <= top_degree);
    }

    // one precompute pass for all components (A_all[s][i][m-1]); needs P^T
    template <typename T, int M>
    int pcd_precompute_all(int order_idx) {
        if (n == 0) return SPFM_OK;
        pt_valid = false;
        int rc = ensure_pt();
        if (rc) return rc;
        const size_t lds = sizeof(T) * (size_t)Kind<M>::AS * kWave * 33;
        HIPC(hipFuncSetAttribute((const void*)pcd_precompute_all_kernel<T, M>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t tiles = (n + 31) / 32;
        const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * 8);
        hipLaunchKernelGGL((pcd_precompute_all_kernel<T, M>), dim3(grid), dim3(kBlock), lds,
                           stream, n, k, rptr.as<int64_t>(), ridx.as<int32_t>(), rval.as<T>(),
                           Pt.as<double>() + (size_t)order_idx * k * d, A.as<T>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }
    template <typename T>
    int pcd_precompute_all_dispatch(int M, int order_idx) {
        switch (M) {
            case 0: return pcd_precompute_all<T, 0>(order_idx);
            case 2: return pcd_precompute_all<T, 2>(order_idx);
            case 3: return pcd_precompute_all<T, 3>(order_idx);
            case 4: return pcd_precompute_all<T, 4>(order_idx);
            case 5: return pcd_precompute_all<T, 5>(order_idx);
            case 6: return pcd_precompute_all<T, 6>(order_idx);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
    }

    template <typename T>
    int pcd_pass_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
        switch (M) {
            case 0: return pcd_pass_body<T, 0>(order_idx, beta, gamma, eta);
            case 2: return pcd_pass_body<T, 2>(order_idx, beta, gamma, eta);
            case 3: return pcd_pass_body<T, 3>(order_idx, beta, gamma, eta);
            case 4: return pcd_pass_body<T, 4>(order_idx, beta, gamma, eta);
            case 5: return pcd_pass_body<T, 5>(order_idx, beta, gamma, eta);
            case 6: return pcd_pass_body<T, 6>(order_idx, beta, gamma, eta);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
    }

    int pcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
                  const int32_t* ic, int n_comp, double* viol) {
        int rc = epoch_prologue();
        if (rc) return rc;
        if (solver != SPFM_SOLVER_PCD) FAIL(SPFM_ERR_INVALID, "engine is not configured for pcd");
        if (order_idx < 0 || order_idx >= n_orders) FAIL(SPFM_ERR_INVALID, "bad order index");
        if (!degree_ok(degree)) FAIL(SPFM_ERR_INVALID, "bad degree");
        if (!ic || n_comp < 0 || n_comp > k) FAIL(SPFM_ERR_INVALID, "bad indices_component");
        for (int q = 0; q < n_comp; ++q)
            if (ic[q] < 0 || ic[q] >= k) FAIL(SPFM_ERR_INVALID, "indices_component out of range");
        rc = ensure_p();
        if (rc) return rc;
        pt_valid = false;
        if (n_comp > 0)
            HIPC(hipMemcpyAsync(comp_order.p, ic, sizeof(int32_t) * (size_t)n_comp,
                                hipMemcpyHostToDevice, stream));
        HIPC(hipMemsetAsync(ctl.p, 0, sizeof(Ctl), stream));
        const std::string key = fkey("pcd", {beta, gamma, eta},
                                     {order_idx, degree, loss, reg, sched_version});
        const int M = kind_of(degree);
        rc = dtype == SPFM_F32 ? pcd_precompute_all_dispatch<float>(M, order_idx)
                               : pcd_precompute_all_dispatch<double>(M, order_idx);
        if (rc) return rc;
        bool use_prb = prb_usable();
        bool use_wide = wide_usable() && M == 2;
        const bool pers_epoch = use_prb || use_wide;
        double* Po_epoch = P.as<double>() + (size_t)order_idx * k * d;
        if (pers_epoch) {
            rc = snapshot_state(Po_epoch, (size_t)k * d, snapP);
            if (rc) return rc;
        }
        for (int pass = 0; pass < n_comp; ++pass) {
            rc = SPFM_OK;
            if (use_wide) {
                rc = dtype == SPFM_F32 ? pcd_pass_wide<float>(order_idx, beta, gamma, eta)
                                       : pcd_pass_wide<double>(order_idx, beta, gamma, eta);
            } else if (use_prb) {
                rc = dtype == SPFM_F32 ? pcd_prb_dispatch<float>(M, order_idx, beta, gamma, eta)
                                       : pcd_prb_dispatch<double>(M, order_idx, beta, gamma, eta);
            }
            if (rc == kNotResident) {
                // the pass was not launched (its helper kernels only picked the component and
                // took snapshots): this and the following passes run on the multi-kernel engine.
                // The component counter was advanced by begin_pass_kernel: step it back.
                mark_not_resident(use_wide ? "wide persistent pcd pass" : "persistent pcd pass");
                use_prb = use_wide = false;
                hipLaunchKernelGGL(unbegin_pass_kernel, dim3(1), dim3(1), 0, stream, ctl.as<Ctl>());
                HIPC(hipGetLastError());
            }
            if (!use_wide && !use_prb) {
                rc = run_cached(key, [&]() {
                    return dtype == SPFM_F32
                               ? pcd_pass_dispatch<float>(M, order_idx, beta, gamma, eta)
                               : pcd_pass_dispatch<double>(M, order_idx, beta, gamma, eta);
                });
            }
            if (rc) return rc;
        }
        pt_valid = false;  // the passes rewrote P; the (d,k) image is stale again
        rc = epoch_epilogue(viol);
        if (rc) return rc;
        if (pers_epoch) {
            bool aborted = false;
            rc = persistent_aborted(&aborted);
            if (rc) return rc;
            if (aborted) {  // all-or-nothing (pcd.py:71-137): back to the epoch's start, redo
                rc = recover_from_abort(Po_epoch, (size_t)k * d, snapP,
                                        use_wide ? "wide persistent pcd pass" : "persistent pcd pass");
                if (rc) return rc;
                return pcd_epoch(order_idx, degree, beta, gamma, eta, ic, n_comp, viol);
            }
        }
        return SPFM_OK;
    }

    // ------------------------------------------------------------------- pbcd
    template <typename T, int M, int L, int C>
    int pbcd_body_lc(int order_idx, double beta, double gamma, double eta) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        double* Po = Pt.as<double>() + (size_t)order_idx * k * d;  // (d,k)
        RegState rs = regstate();
        constexpr int CW = (L == 64) ? C : 1;  // wave-per-column kernels: lanes = 64
        if (n > 0)
            hipLaunchKernelGGL((pbcd_precompute_kernel<T, M>), dim3(cdiv(n * k, kBlock)),
                               dim3(kBlock), 0, stream, n, k, rptr.as<int64_t>(),
                               ridx.as<int32_t>(), rval.as<T>(), Po, A.as<T>());
        const bool chained = (reg == SPFM_REG_SQUAREDL21 || reg == SPFM_REG_OMEGACS);
        if (chained) {
            hipLaunchKernelGGL(pbcd_norms_kernel, dim3(cdiv((int64_t)d * 64, kBlock)),
                               dim3(kBlock), 0, stream, d, k, Po, rs.norms);
            hipLaunchKernelGGL((pbcd_compute_cache_kernel<M>), dim3(1), dim3(kBlock), 0, stream, d,
                               reg, rs);
        }
        const size_t shm = sizeof(double) * ((size_t)(kBlock / L) * k + 16);
        HIPC(hipMemsetAsync(pb_ticket.p, 0, sizeof(int) * 4, stream));  // (resets itself per step)
        const int nb = n_batches();
        for (int b = 0; b < nb; ++b) {
            const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
            if (nc == 0) continue;
            const ColDesc* desc = d_desc.as<ColDesc>() + c0;
            const int64_t bn = prof_on ? batch_nnz(b) : 0;
            prof_begin(2, bn);
            hipLaunchKernelGGL((pbcd_grad_kernel<T, M, L, C>), dim3(nc * kPbW), dim3(kBlock), shm,
                               stream, desc, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                               yy.as<typename Vec2<T>::type>(), Po, k, loss, part.as<double>());
            prof_end(2);
            int rc = allreduce(part.as<double>(), (size_t)nc * kPbW * (k + 1));
            if (rc) return rc;
            const int ncache = top_degree > 0 ? top_degree + 1 : 1;
            if (chained && pbcd_fuse) {
                hipLaunchKernelGGL((pbcd_prep_chain_kernel<M, CW>), dim3(nc), dim3(kWave), 0,
                                   stream, desc, nc, Po, k, part.as<double>(), lams.as<double>(),
                                   reg, mu, beta, gamma, eta, delta.as<double>(),
                                   pold.as<double>(), pb_scal.as<double>(), d, rs, ncache,
                                   pb_ticket.as<int>());
            } else {
                hipLaunchKernelGGL((pbcd_prep_kernel<CW>), dim3(nc), dim3(kWave), 0, stream, desc,
                                   Po, k, part.as<double>(), lams.as<double>(), reg, mu, beta,
                                   gamma, eta, delta.as<double>(), pold.as<double>(),
                                   pb_scal.as<double>());
                if (chained)
                    hipLaunchKernelGGL((pbcd_chain_kernel<M>), dim3(1), dim3(kWave), 0, stream,
                                       desc, nc, d, reg, rs, ncache, pb_scal.as<double>());
            }
            prof_begin(3, bn);
            hipLaunchKernelGGL((pbcd_sync_kernel<T, M, L, C>), dim3(nc * kPbW), dim3(kBlock), 0,
                               stream, desc, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                               yy.as<T>(), lams.as<double>(), k, Po, delta.as<double>(),
                               pold.as<double>(), pb_scal.as<double>(), viol_col.as<double>());
            prof_end(3);
        }
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    // ------------------------------------------------ persistent pbcd pass (one launch)
    static bool pbprb_degree_ok(int M) { return M == 0 || M == 2 || M == 3 || M == 4; }
    bool pbprb_usable(int M) const {
        return persistent && !pers_failed && pb_persistent && (!dist() || peer_ready) &&
               max_batch_cols <= 64 && nnz < ((int64_t)1 << 31) && n > 0 && k <= 62 &&
               pbprb_degree_ok(M);
    }

    // host-side audit of build_pb_stream's output against what the kernel assumes: group
    // boundaries monotone and ending at nnz; every entry a valid CSC position whose row lies in
    // the workgroup's block; slot index inside the group < 64 / NG; a group's entries sorted by
    // slot; at most 64 columns per step.  Returns nullptr or what is wrong.
    const char* validate_pb_stream(int G, int NG, const std::vector<int32_t>& gsp,
                                   const std::vector<int32_t>& src,
                                   const std::vector<uint8_t>& meta) const {
        const int nb = n_batches();
        const size_t stride = (size_t)NG + 1;
        const int64_t rows_per = std::max<int64_t>((n + G - 1) / G, 1);
        const int qm = 64 / NG;
        if (gsp.size() != (size_t)G * nb * stride + 1) return "boundary table has the wrong size";
        if ((int64_t)src.size() != nnz || (int64_t)meta.size() != nnz) return "entry count != nnz";
        for (int b = 0; b < nb; ++b)
            if (batch_ptr[b + 1] - batch_ptr[b] > 64) return "a step has more than 64 columns";
        for (size_t t = 0; t + 1 < gsp.size(); ++t)
            if (gsp[t] > gsp[t + 1] || gsp[t] < 0) return "group boundaries not monotone";
        if (gsp.back() != (int32_t)nnz) return "group boundaries do not end at nnz";
        for (int g = 0; g < G; ++g)
            for (int b = 0; b < nb; ++b)
                for (int grp = 0; grp < NG; ++grp) {
                    const size_t at = ((size_t)g * nb + b) * stride + (size_t)grp;
                    int prev_slot = 0;
                    for (int32_t e = gsp[at]; e < gsp[at + 1]; ++e) {
                        const int32_t pos = src[(size_t)e];
                        if (pos < 0 || pos >= nnz) return "entry points outside the CSC arrays";
                        const int64_t row = h_cidx[(size_t)pos];
                        if (row < 0 || row >= n) return "row index out of range";
                        if (row / rows_per != g) return "entry outside its workgroup's row block";
                        const int slot = meta[(size_t)e] & 0x7f;
                        if (slot >= qm) return "slot index >= slots per group";
                        if (slot < prev_slot) return "a group's entries are not sorted by slot";
                        prev_slot = slot;
                        if (grp + slot * NG >= batch_ptr[b + 1] - batch_ptr[b])
                            return "slot beyond the step's columns";
                    }
                }
        return nullptr;
    }

    // entry stream (workgroup, step, slot, row) for G row blocks; shares the pcd pass's when
    // the workgroup counts agree
    template <typename T>
    int ensure_pb_stream(int NG) {
        int ncu = 0;
        HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        // pbprb_groups workgroups in all: GO dedicated owners + G row workgroups
        const int Gtot = std::max(1, std::min(pbprb_G, ncu));
        int GO = pbprb_owners;
        GO = std::max(0, std::min({GO, 64, Gtot - 1}));
        const int G = Gtot - GO;
        if (pb_stream_ready && pb_stream_G == G && pb_stream_NG == NG && pb_GO == GO)
            return SPFM_OK;
        pb_GO = GO;
        std::vector<int32_t> gsp, src;
        std::vector<uint8_t> meta;
        build_pb_stream(n, h_cptr.data(), h_cidx.data(), order, batch_ptr, G, NG, gsp, src, meta);
        // every bound pbcd_prb_kernel indexes with, checked on the host for problems where that
        // is free (and on request, SPFM_VALIDATE=1): an out-of-range row or slot index would be
        // a device memory fault, i.e. a dead process
        if (nnz < ((int64_t)1 << 22) || getenv("SPFM_VALIDATE")) {
            const char* bad = validate_pb_stream(G, NG, gsp, src, meta);
            if (bad) FAIL(SPFM_ERR_RUNTIME, std::string("internal: pbcd entry stream: ") + bad);
        }
        DevBuf d_src;
        HIPC(d_src.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
        HIPC(pb_sp.alloc(sizeof(int32_t) * gsp.size()));
        HIPC(pb_erow.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1) + 256));
        HIPC(pb_eval.alloc(sizeof(T) * (size_t)(nnz > 0 ? nnz : 1) + 256));
        HIPC(pb_meta.alloc((size_t)(nnz > 0 ? nnz : 1) + 256));
        HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
        HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
        HIPC(pb_stamps.alloc(sizeof(long long) * 16 * (size_t)(G + GO)));
        HIPC(hipMemsetAsync(pb_stamps.p, 0, pb_stamps.bytes, stream));
        HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
        HIPC(hipMemcpyAsync(pb_sp.p, gsp.data(), sizeof(int32_t) * gsp.size(),
                            hipMemcpyHostToDevice, stream));
        if (nnz > 0) {
            HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemcpyAsync(pb_meta.p, meta.data(), (size_t)nnz, hipMemcpyHostToDevice,
                                stream));
            hipLaunchKernelGGL((prb_gather_kernel<T>), dim3(cdiv(nnz, 256)), dim3(256), 0, stream,
                               nnz, d_src.as<int32_t>(), cidx.as<int32_t>(), cval.as<T>(),
                               pb_erow.as<int32_t>(), pb_eval.as<T>());
            HIPC(hipGetLastError());
        }
        HIPC(hipStreamSynchronize(stream));
        pb_stream_G = G;
        pb_stream_NG = NG;
        pb_stream_ready = true;
        return SPFM_OK;
    }

    template <typename T, int M, int L>
    int pbcd_prb_l(int order_idx, double beta, double gamma, double eta) {
        const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
        double* Po = Pt.as<double>() + (size_t)order_idx * k * d;  // (d,k)
        RegState rs = regstate();
        int rc = ensure_pb_stream<T>(kPbPrbThreads / L);
        if (rc) return rc;
        const int G = pb_stream_G;
        // the rows' state as packed records (cache values, yhat, y: one line per row at k <= 30,
        // degree 2, float): the precompute pass of pbcd.py:18-33 writes them
        constexpr int AS = Kind<M>::AS;
        HIPC(pb_rec.alloc(sizeof(T) * (size_t)n * AS * L + 256));
        hipLaunchKernelGGL((pbprb_pack_kernel<T, M, L>), dim3(cdiv(n * L, kBlock)), dim3(kBlock), 0,
                           stream, n, k, rptr.as<int64_t>(), ridx.as<int32_t>(), rval.as<T>(), Po,
                           yy.as<T>(), pb_rec.as<T>());
        const bool chained = (reg == SPFM_REG_SQUAREDL21 || reg == SPFM_REG_OMEGACS);
        if (chained) {
            hipLaunchKernelGGL(pbcd_norms_kernel, dim3(cdiv((int64_t)d * 64, kBlock)),
                               dim3(kBlock), 0, stream, d, k, Po, rs.norms);
            hipLaunchKernelGGL((pbcd_compute_cache_kernel<M>), dim3(1), dim3(kBlock), 0, stream, d,
                               reg, rs);
        }
        HIPC(pb_slabA.alloc(sizeof(double) * 2 * 64 * (size_t)G * L));
        HIPC(pb_slabB.alloc(sizeof(double) * 2 * 64 * L));
        HIPC(hipMemsetAsync(pb_slabA.p, 0, sizeof(double) * 2 * 64 * (size_t)G * L, stream));
        HIPC(hipMemsetAsync(pb_slabB.p, 0, sizeof(double) * 2 * 64 * L, stream));
        {
            int prc = peer_clear(kPeerPbOff, kPeerProbeOff - kPeerPbOff);
            if (prc) return prc;
        }
        PbPrbArgs a;
        a.G = G;
        a.GO = pb_GO;
        a.nb = n_batches();
        a.bptr = d_bptr.as<int32_t>();
        a.jsched = d_order.as<int32_t>();
        a.gsp = pb_sp.as<int32_t>();
        a.erow = pb_erow.as<int32_t>();
        a.emeta = pb_meta.as<uint8_t>();
        a.slabA = pb_slabA.as<double>();
        a.slabB = pb_slabB.as<double>();
        a.rows_per = (int)std::max<int64_t>((n + G - 1) / G, 1);
        a.n_rows = (int)n;
        a.abort_flag = prb_abort.as<unsigned>();
        a.spin_max = spin_max;
        a.n_ranks = peer_ready ? n_ranks : 1;
        a.rank = rank;
        a.slabC = peer_ready ? peer_tab_pb.as<double*>() : nullptr;
        if (peer_ready && L * n_ranks > 64 * 8)
            FAIL(SPFM_ERR_UNSUPPORTED, "persistent pbcd pass: more than 8 ranks");
        a.stamps = pb_stamp_on ? pb_stamps.as<long long>() : nullptr;
        a.dbg = pb_dbg;
        HIPC(pb_dbgbuf.alloc(sizeof(unsigned) * (16 + 4096)));
        if (pb_dbg & 8) HIPC(hipMemsetAsync(pb_dbgbuf.p, 0, sizeof(unsigned) * (16 + 4096), stream));
        if (pb_dbg & 8) {
            unsigned init[16] = {0, 0, 0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            HIPC(hipMemcpyAsync(pb_dbgbuf.p, init, sizeof init, hipMemcpyHostToDevice, stream));
        }
        a.dbg_out = pb_dbgbuf.as<unsigned>();
        const int ncache = top_degree > 0 ? top_degree + 1 : 1;
        prof_begin(2, nnz);
        auto go = [&](auto stamp_tag, auto down_tag) -> int {
            constexpr bool STc = decltype(stamp_tag)::value;
            constexpr bool DWc = decltype(down_tag)::value;
            const size_t lds = std::max(kPrbLds, pbcd_prb_lds_bytes<T, M, L>());
            const int grid = G + (DWc ? pb_GO : 0);
            HIPC(hipFuncSetAttribute((const void*)pbcd_prb_kernel<T, M, L, STc, DWc>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (!resident_ok((const void*)pbcd_prb_kernel<T, M, L, STc, DWc>, kPbPrbThreads, lds,
                             grid))
                return kNotResident;
            hipLaunchKernelGGL((pbcd_prb_kernel<T, M, L, STc, DWc>), dim3(launch_groups(grid)),
                               dim3(kPbPrbThreads),
                               lds, stream, a, pb_eval.as<T>(), pb_rec.as<T>(), Po, k, d,
                               lams.as<double>(), loss, reg, rs, ncache, mu, beta, gamma, eta,
                               prb_viol.as<double>());
            return SPFM_OK;
        };
        constexpr bool can_stamp = std::is_same<T, float>::value && M == 2 && L == 32;
        if (pb_stamp_on && !can_stamp)
            FAIL(SPFM_ERR_UNSUPPORTED, "pbprb_stamps: built for float storage, degree 2, k <= 30");
        const bool down = pb_GO > 0;
        if constexpr (can_stamp) {
            if (pb_stamp_on)
                rc = down ? go(std::true_type{}, std::true_type{})
                          : go(std::true_type{}, std::false_type{});
            else
                rc = down ? go(std::false_type{}, std::true_type{})
                          : go(std::false_type{}, std::false_type{});
        } else {
            rc = down ? go(std::false_type{}, std::true_type{})
                      : go(std::false_type{}, std::false_type{});
        }
        if (rc == kNotResident) prof_cancel(2, nnz);
        if (rc) return rc;
        prof_end(2);
        hipLaunchKernelGGL((pbprb_unpack_kernel<T, AS, L>), dim3(cdiv(n, 256)), dim3(256), 0, stream,
                           n, pb_rec.as<T>(), yy.as<T>());
        hipLaunchKernelGGL(fold_viol_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                           d_desc.as<ColDesc>(), prb_viol.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    template <typename T, int M>
    int pbcd_prb_m(int order_idx, double beta, double gamma, double eta) {
        if (k <= 30) return pbcd_prb_l<T, M, 32>(order_idx, beta, gamma, eta);
        return pbcd_prb_l<T, M, 64>(order_idx, beta, gamma, eta);
    }

    template <typename T>
    int pbcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
        switch (M) {
            case 0: return pbcd_prb_m<T, 0>(order_idx, beta, gamma, eta);
            case 2: return pbcd_prb_m<T, 2>(order_idx, beta, gamma, eta);
            case 3: return pbcd_prb_m<T, 3>(order_idx, beta, gamma, eta);
            case 4: return pbcd_prb_m<T, 4>(order_idx, beta, gamma, eta);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "persistent pbcd pass: degree outside {2,3,4,all-subsets}");
    }

    template <typename T, int M>
    int pbcd_body(int order_idx, double beta, double gamma, double eta) {
        if (k <= 8) return pbcd_body_lc<T, M, 8, 1>(order_idx, beta, gamma, eta);
        if (k <= 16) return pbcd_body_lc<T, M, 16, 1>(order_idx, beta, gamma, eta);
        if (k <= 32) return pbcd_body_lc<T, M, 32, 1>(order_idx, beta, gamma, eta);
        if (k <= 64) return pbcd_body_lc<T, M, 64, 1>(order_idx, beta, gamma, eta);
        if (k <= 128) return pbcd_body_lc<T, M, 64, 2>(order_idx, beta, gamma, eta);
        return pbcd_body_lc<T, M, 64, 4>(order_idx, beta, gamma, eta);
    }

    template <typename T>
    int pbcd_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
        switch (M) {
            case 0: return pbcd_body<T, 0>(order_idx, beta, gamma, eta);
            case 2: return pbcd_body<T, 2>(order_idx, beta, gamma, eta);
            case 3: return pbcd_body<T, 3>(order_idx, beta, gamma, eta);
            case 4: return pbcd_body<T, 4>(order_idx, beta, gamma, eta);
            case 5: return pbcd_body<T, 5>(order_idx, beta, gamma, eta);
            case 6: return pbcd_body<T, 6>(order_idx, beta, gamma, eta);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
    }

    int pbcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
                   double* viol) {
        int rc = epoch_prologue();
        if (rc) return rc;
        if (solver != SPFM_SOLVER_PBCD) FAIL(SPFM_ERR_INVALID, "engine is not configured for pbcd");
        if (order_idx < 0 || order_idx >= n_orders) FAIL(SPFM_ERR_INVALID, "bad order index");
        if (!degree_ok(degree)) FAIL(SPFM_ERR_INVALID, "bad degree");
        rc = ensure_pt();
        if (rc) return rc;
        p_valid = false;
        pbprb_active = 0;
        if (pbprb_usable(kind_of(degree))) {
            pbprb_active = 1;
            double* Pt_epoch = Pt.as<double>() + (size_t)order_idx * k * d;
            rc = snapshot_state(Pt_epoch, (size_t)k * d, snapP);
            if (rc) return rc;
            rc = dtype == SPFM_F32
                     ? pbcd_prb_dispatch<float>(kind_of(degree), order_idx, beta, gamma, eta)
                     : pbcd_prb_dispatch<double>(kind_of(degree), order_idx, beta, gamma, eta);
            if (rc == kNotResident) {  // nothing launched but the epoch's set-up kernels
                mark_not_resident("persistent pbcd pass");
                return pbcd_epoch(order_idx, degree, beta, gamma, eta, viol);
            }
            if (rc) return rc;
            rc = epoch_epilogue(viol);
            if (rc) return rc;
            bool aborted = false;
            rc = persistent_aborted(&aborted);
            if (rc) return rc;
            if (aborted) {  // all-or-nothing (pbcd.py:82-148): back to the epoch's start, redo
                rc = recover_from_abort(Pt_epoch, (size_t)k * d, snapP, "persistent pbcd pass");
                if (rc) return rc;
                return pbcd_epoch(order_idx, degree, beta, gamma, eta, viol);
            }
            return SPFM_OK;
        }
        const std::string key = fkey("pbcd", {beta, gamma, eta},
                                     {order_idx, degree, loss, reg, sched_version});
        rc = run_cached(key, [&]() {
            return dtype == SPFM_F32 ? pbcd_dispatch<float>(kind_of(degree), order_idx, beta, gamma, eta)
                                     : pbcd_dispatch<double>(kind_of(degree), order_idx, beta, gamma, eta);
        });
        if (rc) return rc;
        return epoch_epilogue(viol);
    }

    // ================================================= host-stepped epochs
    // User-defined regularizer objects (regularizer/__init__.py:8-15, base.py:27-34: the
    // reference's duck-typed plug-in protocol) cannot run inside the device chains.  For them the
    // epoch is stepped from the host: per dependent step the device forms the column sums
    // (pcd.py:54-59 / pbcd.py:60-67), the caller applies the update rule with its own
    // prox_cd / prox_bcd and cache hooks in visiting order, the device scatter-updates
    // (pcd.py:124-133 / pbcd.py:135-144).  Two host round trips per step: a path that honours
    // the plug-in surface, not a fast one.  Multi-kernel kernels; with several ranks the sums are
    // all-reduced like any other step.
    int host_order = -1, host_degree = 0;
    std::vector<double> host_stage;

    int host_epoch_begin(int order_idx, int degree) {
        int rc = epoch_prologue();
        if (rc) return rc;
        if (solver != SPFM_SOLVER_PCD && solver != SPFM_SOLVER_PBCD)
            FAIL(SPFM_ERR_INVALID, "host-stepped epochs: configure for pcd or pbcd");
        if (order_idx < 0 || order_idx >= n_orders) FAIL(SPFM_ERR_INVALID, "bad order index");
        if (!degree_ok(degree)) FAIL(SPFM_ERR_INVALID, "bad degree");
        const int M = kind_of(degree);
        if (solver == SPFM_SOLVER_PCD) {
            rc = ensure_p();
            if (rc) return rc;
            pt_valid = false;
            HIPC(hipMemsetAsync(ctl.p, 0, sizeof(Ctl), stream));
            rc = dtype == SPFM_F32 ? pcd_precompute_all_dispatch<float>(M, order_idx)
                                   : pcd_precompute_all_dispatch<double>(M, order_idx);
            if (rc) return rc;
            pt_valid = false;
        } else {
            rc = ensure_pt();
            if (rc) return rc;
            p_valid = false;
            rc = dtype == SPFM_F32 ? host_pbcd_precompute<float>(M, order_idx)
                                   : host_pbcd_precompute<double>(M, order_idx);
            if (rc) return rc;
        }
        host_order = order_idx;
        host_degree = degree;
        return sync();
    }
    template <typename T>
    int host_pbcd_precompute(int M, int order_idx) {
        if (n == 0) return SPFM_OK;
        double* Po = Pt.as<double>() + (size_t)order_idx * k * d;
#define SPFM_HPRE(MM)                                                                        \
    hipLaunchKernelGGL((pbcd_precompute_kernel<T, MM>), dim3(cdiv(n * k, kBlock)), dim3(kBlock), \
                       0, stream, n, k, rptr.as<int64_t>(), ridx.as<int32_t>(), rval.as<T>(), Po, \
                       A.as<T>())
        switch (M) {
            case 0: SPFM_HPRE(0); break;
            case 2: SPFM_HPRE(2); break;
            case 3: SPFM_HPRE(3); break;
            case 4: SPFM_HPRE(4); break;
            case 5: SPFM_HPRE(5); break;
            case 6: SPFM_HPRE(6); break;
            default: FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
        }
#undef SPFM_HPRE
        HIPC(hipGetLastError());
        return SPFM_OK;
    }
    int host_pass_begin(int s) {  // pcd: the component of the following steps (pcd.py:92)
        if (host_order < 0 || solver != SPFM_SOLVER_PCD)
            FAIL(SPFM_ERR_INVALID, "host_pass_begin: call spfm_host_epoch_begin (pcd) first");
        if (s < 0 || s >= k) FAIL(SPFM_ERR_INVALID, "component out of range");
        Ctl hc;
        std::memset(&hc, 0, sizeof hc);
        hc.s = s;
        hc.lam = h_lams[(size_t)s];
        HIPC(hipMemcpyAsync(ctl.p, &hc, sizeof hc, hipMemcpyHostToDevice, stream));
        return sync();
    }
    int host_step_check(int b) {
        if (host_order < 0) FAIL(SPFM_ERR_INVALID, "host step: call spfm_host_epoch_begin first");
        if (b < 0 || b >= n_batches()) FAIL(SPFM_ERR_INVALID, "host step: step index out of range");
        return SPFM_OK;
    }

    template <typename T, int M>
    int host_sums_pcd(int b, double* out) {
        const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
        if (nc == 0) return SPFM_OK;
        double* Po = P.as<double>() + (size_t)host_order * k * d;
        hipLaunchKernelGGL((pcd_grad_kernel<T, M>), dim3(nc), dim3(kBlock), 0, stream, ctl.as<Ctl>(),
                           d_desc.as<ColDesc>() + c0, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                           (size_t)n * Kind<M>::AS, yy.as<typename Vec2<T>::type>(), Po, d, loss,
                           part.as<double>(), pold.as<double>());
        HIPC(hipGetLastError());
        int rc = allreduce(part.as<double>(), (size_t)2 * nc);
        if (rc) return rc;
        HIPC(hipMemcpyAsync(out, part.p, sizeof(double) * 2 * (size_t)nc, hipMemcpyDeviceToHost,
                            stream));
        return sync();
    }
    template <typename T, int M>
    int host_apply_pcd(int b, const double* p_new) {
        const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
        if (nc == 0) return SPFM_OK;
        double* Po = P.as<double>() + (size_t)host_order * k * d;
        HIPC(hipMemcpyAsync(delta.p, p_new, sizeof(double) * (size_t)nc, hipMemcpyHostToDevice,
                            stream));
        hipLaunchKernelGGL(host_apply_pcd_kernel, dim3(cdiv(nc, 64)), dim3(64), 0, stream,
                           ctl.as<Ctl>(), d_desc.as<ColDesc>() + c0, nc, Po, d, pold.as<double>(),
                           delta.as<double>(), viol_col.as<double>());
        hipLaunchKernelGGL((pcd_sync_kernel<T, M>), dim3(nc), dim3(kBlock), 0, stream, ctl.as<Ctl>(),
                           d_desc.as<ColDesc>() + c0, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                           (size_t)n * Kind<M>::AS, yy.as<T>(), delta.as<double>(),
                           pold.as<double>());
        HIPC(hipGetLastError());
        return sync();  // the caller's p_new buffer is free again
    }
    template <typename T, int M, int L, int C>
    int host_sums_pbcd(int b, double* out) {
        const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
        if (nc == 0) return SPFM_OK;
        double* Po = Pt.as<double>() + (size_t)host_order * k * d;
        const size_t shm = sizeof(double) * ((size_t)(kBlock / L) * k + 16);
        hipLaunchKernelGGL((pbcd_grad_kernel<T, M, L, C>), dim3(nc * kPbW), dim3(kBlock), shm, stream,
                           d_desc.as<ColDesc>() + c0, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                           yy.as<typename Vec2<T>::type>(), Po, k, loss, part.as<double>());
        HIPC(hipGetLastError());
        const size_t np = (size_t)nc * kPbW * (k + 1);
        int rc = allreduce(part.as<double>(), np);
        if (rc) return rc;
        host_stage.resize(np);
        HIPC(hipMemcpyAsync(host_stage.data(), part.p, sizeof(double) * np, hipMemcpyDeviceToHost,
                            stream));
        rc = sync();
        if (rc) return rc;
        for (int q = 0; q < nc; ++q)  // the column's kPbW partial vectors in fixed order
            for (int s = 0; s <= k; ++s) {
                double acc = 0.0;
                for (int w = 0; w < kPbW; ++w)
                    acc += host_stage[((size_t)q * kPbW + w) * (k + 1) + s];
                out[(size_t)q * (k + 1) + s] = acc;
            }
        return SPFM_OK;
    }
    template <typename T, int M, int L, int C>
    int host_apply_pbcd(int b, const double* p_new, const double* p_old) {
        const int c0 = batch_ptr[b], nc = batch_ptr[b + 1] - c0;
        if (nc == 0) return SPFM_OK;
        if (!p_old) FAIL(SPFM_ERR_INVALID, "host step (pbcd): p_old is required");
        double* Po = Pt.as<double>() + (size_t)host_order * k * d;
        host_stage.assign((size_t)4 * nc, 0.0);
        for (int q = 0; q < nc; ++q) host_stage[(size_t)4 * q + 2] = 1.0;  // shrink factor f = 1
        HIPC(hipMemcpyAsync(delta.p, p_new, sizeof(double) * (size_t)nc * k, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(pold.p, p_old, sizeof(double) * (size_t)nc * k, hipMemcpyHostToDevice,
                            stream));
        HIPC(hipMemcpyAsync(pb_scal.p, host_stage.data(), sizeof(double) * 4 * (size_t)nc,
                            hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((pbcd_sync_kernel<T, M, L, C>), dim3(nc * kPbW), dim3(kBlock), 0, stream,
                           d_desc.as<ColDesc>() + c0, cidx.as<int32_t>(), cval.as<T>(), A.as<T>(),
                           yy.as<T>(), lams.as<double>(), k, Po, delta.as<double>(),
                           pold.as<double>(), pb_scal.as<double>(), viol_col.as<double>());
        HIPC(hipGetLastError());
        return sync();
    }
    // dispatch on (storage type, degree, component lanes) as the multi-kernel engine does
    template <typename T, int M>
    int host_step_tm(bool sums, int b, double* out, const double* p_new, const double* p_old) {
        if (solver == SPFM_SOLVER_PCD)
            return sums ? host_sums_pcd<T, M>(b, out) : host_apply_pcd<T, M>(b, p_new);
#define SPFM_HPB(LL, CC) \
    return sums ? host_sums_pbcd<T, M, LL, CC>(b, out) : host_apply_pbcd<T, M, LL, CC>(b, p_new, p_old)
        if (k <= 8) SPFM_HPB(8, 1);
        if (k <= 16) SPFM_HPB(16, 1);
        if (k <= 32) SPFM_HPB(32, 1);
        if (k <= 64) SPFM_HPB(64, 1);
        if (k <= 128) SPFM_HPB(64, 2);
        SPFM_HPB(64, 4);
#undef SPFM_HPB
    }
    template <typename T>
    int host_step_t(bool sums, int b, double* out, const double* p_new, const double* p_old) {
        switch (kind_of(host_degree)) {
            case 0: return host_step_tm<T, 0>(sums, b, out, p_new, p_old);
            case 2: return host_step_tm<T, 2>(sums, b, out, p_new, p_old);
            case 3: return host_step_tm<T, 3>(sums, b, out, p_new, p_old);
            case 4: return host_step_tm<T, 4>(sums, b, out, p_new, p_old);
            case 5: return host_step_tm<T, 5>(sums, b, out, p_new, p_old);
            case 6: return host_step_tm<T, 6>(sums, b, out, p_new, p_old);
        }
        FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
    }
    int host_step(bool sums, int b, double* out, const double* p_new, const double* p_old) {
        int rc = host_step_check(b);
        if (rc) return rc;
        if ((sums && !out) || (!sums && !p_new)) FAIL(SPFM_ERR_INVALID, "host step: NULL buffer");
        return dtype == SPFM_F32 ? host_step_t<float>(sums, b, out, p_new, p_old)
                                 : host_step_t<double>(sums, b, out, p_new, p_old);
    }
    int host_epoch_end(double* viol) {
        if (host_order < 0) FAIL(SPFM_ERR_INVALID, "host_epoch_end: no host-stepped epoch open");
        host_order = -1;
        if (solver == SPFM_SOLVER_PCD) pt_valid = false;
        return epoch_epilogue(viol);
    }

    // ================================================================== psgd
    // regularizer.init_cache_psgd exists for l1 / l21 / squaredl12 / squaredl21 only
    // (reference regularizer/*.py); psgd has no all-subsets variant.
    int configure_psgd(int loss_, int reg_, int top_degree_) {
        if (reg_ != SPFM_REG_L1 && reg_ != SPFM_REG_L21 && reg_ != SPFM_REG_SQUAREDL12 &&
            reg_ != SPFM_REG_SQUAREDL21)
            FAIL(SPFM_ERR_INVALID, "this regularizer cannot be used with solver='psgd'");
        if (top_degree_ < 2 || top_degree_ > SPFM_MAX_DEGREE)
            FAIL(SPFM_ERR_UNSUPPORTED, "psgd: degree must be in 2..6");
        if (top_degree_ - (n_orders - 1) < 1)
            FAIL(SPFM_ERR_INVALID, "psgd: more parameter orders than degrees");
        if (k > 64 * kPsgdMaxC) FAIL(SPFM_ERR_UNSUPPORTED, "psgd: n_components > 256 not supported");
        solver = SPFM_SOLVER_PSGD;
        loss = loss_;
        reg = reg_;
        top_degree = top_degree_;
        clear_graphs();
        const size_t np = (size_t)n_orders * k * d;
        const size_t V = (size_t)n_orders * k;
        HIPC(sg_gradP.alloc(sizeof(double) * np));
        HIPC(sg_gradw.alloc(sizeof(double) * (size_t)d));
        HIPC(sg_samples.alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)));
        HIPC(sg_part.alloc(sizeof(double) * 2 * V * kPsgdNB));
        HIPC(sg_cond.alloc(sizeof(double) * V));
        HIPC(sg_thr.alloc(sizeof(double) * V));
        HIPC(sg_theta.alloc(sizeof(double) * V));
        HIPC(sg_done.alloc(sizeof(int) * 4));
        HIPC(sg_conv.alloc(sizeof(int) * V));
        HIPC(sg_norms.alloc(sizeof(double) * (size_t)n_orders * d));
        HIPC(hipMemsetAsync(sg_cond.p, 0, sizeof(double) * V, stream));  // first prox: G = all
        psgd_warm = false;
        HIPC(hipMemsetAsync(sg_gradP.p, 0, sizeof(double) * np, stream));
        HIPC(hipMemsetAsync(sg_gradw.p, 0, sizeof(double) * (size_t)d, stream));
        HIPC(hipMemsetAsync(sg_done.p, 0, sizeof(int) * 4, stream));
        HIPC(scalar.alloc(sizeof(double) * 8));
        if (!h_scalar) HIPC(hipHostMalloc((void**)&h_scalar, sizeof(double) * 8));
        HIPC(hipStreamSynchronize(stream));
        configured = true;
        return SPFM_OK;
    }

    // psgd.py:9-22
    static void psgd_eta(int lr, double eta0, double alpha, double beta, double power_t,
                         int64_t it, double* eta_P, double* eta_w) {
        if (lr == 0) {
            *eta_P = eta0;
            *eta_w = eta0;
        } else if (lr == 1) {
            const double eta_it = eta0 * (double)it;
            *eta_P = eta0 / std::pow(1.0 + eta_it * beta, power_t);
            *eta_w = eta0 / std::pow(1.0 + eta_it * alpha, power_t);
        } else if (lr == 2) {
            *eta_P = 1.0 / (beta * (double)it);
            *eta_w = 1.0 / (alpha * (double)it);
        } else {
            const double eta = eta0 / std::pow((double)it, power_t);
            *eta_P = eta;
            *eta_w = eta;
        }
    }

    template <typename T, int L>
    int psgd_epoch_tl(int degree, double alpha, double beta, double gamma, double eta0, int lr,
                      double power_t, int64_t batch_size, int fit_linear, int64_t* it) {
        const bool mich = (reg == SPFM_REG_SQUAREDL12 || reg == SPFM_REG_SQUAREDL21);
        constexpr int gpb = kBlock / L;
        const int nb_dense = (int)std::min<int64_t>(kPsgdNB, cdiv(d, gpb));
        MichState ms;
        ms.part = sg_part.as<double>();
        ms.cond = sg_cond.as<double>();
        ms.thr = sg_thr.as<double>();
        ms.theta = sg_theta.as<double>();
        ms.conv = sg_conv.as<int>();
        ms.done = sg_done.as<int>();
        ms.V = (reg == SPFM_REG_SQUAREDL12) ? n_orders * k : n_orders;
        ms.NB = nb_dense;
        const int nb_fin = cdiv(ms.V, kBlock / kWave);
        int* h_done = reinterpret_cast<int*>(h_scalar + 4);
        // l1 / l21 (no host round trip inside a minibatch): tabulate the epoch and replay runs
        // of kPsgdRun minibatches from one hipGraph -- the kernels take the batch from a device
        // table, so every (gradient, update) pair has identical arguments
        // (squared-norm prox: the first epoch after configure() starts the support search cold --
        // 6-13 sweeps -- and runs eagerly; afterwards minibatches warm-start each other)
        if (use_graph && !prof_on && !psgd_force_eager && (!mich || psgd_warm)) {
            constexpr int kPsgdRun = 32;
            const int kMichSweeps = psgd_graph_sweeps;  // recorded sweeps per minibatch (2 suffice
                                                        // with the warm start; a failed check
                                                        // redoes the epoch eagerly)
            const int64_t nbat = cdiv(n, batch_size);
            h_sched.resize((size_t)nbat);
            for (int64_t bi = 0; bi < nbat; ++bi) {
                const int64_t pos = bi * batch_size;
                const int B = (int)std::min<int64_t>(batch_size, n - pos);
                double eta_P, eta_w;
                psgd_eta(lr, eta0, alpha, beta, power_t, *it + bi, &eta_P, &eta_w);
                PsgdBatch& e = h_sched[(size_t)bi];
                e.pos = pos;
                e.B = B;
                e.pad = 0;
                e.cp = eta_P / (double)B;
                e.denp = 1.0 + eta_P * beta;
                e.strength = gamma * eta_P / (1 + eta_P * beta);
                e.cw = eta_w / (double)B;
                e.denw = 1 + eta_w * alpha;
            }
            const void* old = sg_sched.p;
            HIPC(sg_sched.alloc(sizeof(PsgdBatch) * (size_t)nbat));
            HIPC(sg_idx.alloc(sizeof(int) * 4));
            if (sg_sched.p != old) clear_graphs();
            HIPC(hipMemcpyAsync(sg_sched.p, h_sched.data(), sizeof(PsgdBatch) * (size_t)nbat,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemsetAsync(sg_idx.p, 0, sizeof(int) * 4, stream));  // idx[0..1], idx[2] = failed
            const size_t np = (size_t)n_orders * k * d;
            if (mich) {  // snapshot for the (rare) eager redo
                HIPC(sg_snapP.alloc(sizeof(double) * np));
                HIPC(sg_snapw.alloc(sizeof(double) * (size_t)d));
                HIPC(sg_snapc.alloc(sizeof(double) * (size_t)ms.V));
                HIPC(hipMemcpyAsync(sg_snapP.p, Pt.p, sizeof(double) * np, hipMemcpyDeviceToDevice,
                                    stream));
                HIPC(hipMemcpyAsync(sg_snapw.p, w.p, sizeof(double) * (size_t)d,
                                    hipMemcpyDeviceToDevice, stream));
                HIPC(hipMemcpyAsync(sg_snapc.p, sg_cond.p, sizeof(double) * (size_t)ms.V,
                                    hipMemcpyDeviceToDevice, stream));
            }
            const int gridg = cdiv(std::min<int64_t>(batch_size, n), gpb);
            auto pair = [&]() {
                hipLaunchKernelGGL((psgd_grad_kernel<T, L>), dim3(gridg), dim3(kBlock), 0, stream,
                                   sg_samples.as<int32_t>(), 0, rptr.as<int64_t>(),
                                   ridx.as<int32_t>(), rval.as<T>(), yy.as<T>(), Pt.as<double>(),
                                   w.as<double>(), lams.as<double>(), n_orders, k, d, degree, loss,
                                   fit_linear, sg_gradP.as<double>(), sg_gradw.as<double>(),
                                   pred_tmp.as<double>(), sg_sched.as<PsgdBatch>(),
                                   sg_idx.as<int>());
                hipLaunchKernelGGL((psgd_update_kernel<L>), dim3(nb_dense), dim3(kBlock), 0, stream,
                                   Pt.as<double>(), sg_gradP.as<double>(), w.as<double>(),
                                   sg_gradw.as<double>(), n_orders, k, d, reg, 0.0, 1.0, 0.0,
                                   fit_linear, 0.0, 1.0, sg_norms.as<double>(), ms,
                                   sg_sched.as<PsgdBatch>(), sg_idx.as<int>());
                if (!mich) return;
                hipLaunchKernelGGL(psgd_mich_finish_kernel, dim3(nb_fin), dim3(kBlock), 0, stream,
                                   ms, 0.0, sg_sched.as<PsgdBatch>(), sg_idx.as<int>());
                for (int sweep = 0; sweep < kMichSweeps; ++sweep) {
                    hipLaunchKernelGGL((psgd_mich_reduce_kernel<L>), dim3(nb_dense), dim3(kBlock),
                                       0, stream, Pt.as<double>(), sg_norms.as<double>(), n_orders,
                                       k, d, reg, ms);
                    hipLaunchKernelGGL(psgd_mich_finish_kernel, dim3(nb_fin), dim3(kBlock), 0,
                                       stream, ms, 0.0, sg_sched.as<PsgdBatch>(), sg_idx.as<int>());
                }
                hipLaunchKernelGGL(psgd_mich_verify_kernel, dim3(1), dim3(kBlock), 0, stream, ms,
                                   sg_idx.as<int>() + 2);
                hipLaunchKernelGGL((psgd_mich_apply_kernel<L>), dim3(nb_dense), dim3(kBlock), 0,
                                   stream, Pt.as<double>(), sg_norms.as<double>(), n_orders, k, d,
                                   reg, sg_thr.as<double>());
            };
            const std::string key = fkey("psgd", {}, {degree, loss, reg, fit_linear, gridg, L,
                                                      (int64_t)sizeof(T), (int64_t)mich,
                                                      (int64_t)psgd_graph_sweeps});
            int64_t done_b = 0;
            for (; done_b + kPsgdRun <= nbat; done_b += kPsgdRun) {
                int rc = run_cached(key, [&]() {
                    for (int q = 0; q < kPsgdRun; ++q) pair();
                    return (int)SPFM_OK;
                });
                if (rc) return rc;
            }
            for (; done_b < nbat; ++done_b) pair();
            HIPC(hipGetLastError());
            int failed = 0;
            if (mich)
                HIPC(hipMemcpyAsync(&failed, sg_idx.as<int>() + 2, sizeof(int),
                                    hipMemcpyDeviceToHost, stream));
            HIPC(hipStreamSynchronize(stream));  // h_sched may be rewritten by the next epoch
            if (!failed) {
                *it += nbat;
                return SPFM_OK;
            }
            // some minibatch needed more sweeps than were recorded: restore and redo eagerly
            HIPC(hipMemcpyAsync(Pt.p, sg_snapP.p, sizeof(double) * np, hipMemcpyDeviceToDevice,
                                stream));
            HIPC(hipMemcpyAsync(w.p, sg_snapw.p, sizeof(double) * (size_t)d,
                                hipMemcpyDeviceToDevice, stream));
            HIPC(hipMemcpyAsync(sg_cond.p, sg_snapc.p, sizeof(double) * (size_t)ms.V,
                                hipMemcpyDeviceToDevice, stream));
            HIPC(hipMemsetAsync(sg_gradP.p, 0, sizeof(double) * np, stream));
            HIPC(hipMemsetAsync(sg_gradw.p, 0, sizeof(double) * (size_t)d, stream));
            psgd_redone += 1;
        }
        for (int64_t pos = 0; pos < n; pos += batch_size) {
            const int B = (int)std::min<int64_t>(batch_size, n - pos);
            prof_begin(0, 0);
            hipLaunchKernelGGL((psgd_grad_kernel<T, L>), dim3(cdiv(B, gpb)), dim3(kBlock), 0, stream,
                               sg_samples.as<int32_t>() + pos, B, rptr.as<int64_t>(),
                               ridx.as<int32_t>(), rval.as<T>(), yy.as<T>(), Pt.as<double>(),
                               w.as<double>(), lams.as<double>(), n_orders, k, d, degree, loss,
                               fit_linear, sg_gradP.as<double>(), sg_gradw.as<double>(),
                               pred_tmp.as<double>() + pos, (const PsgdBatch*)nullptr,
                               (int*)nullptr);
            prof_end(0);
            double eta_P, eta_w;
            psgd_eta(lr, eta0, alpha, beta, power_t, *it, &eta_P, &eta_w);
            const double strength = gamma * eta_P / (1 + eta_P * beta);
            prof_begin(1, 0);
            hipLaunchKernelGGL((psgd_update_kernel<L>), dim3(nb_dense), dim3(kBlock), 0, stream,
                               Pt.as<double>(), sg_gradP.as<double>(), w.as<double>(),
                               sg_gradw.as<double>(), n_orders, k, d, reg, eta_P / (double)B,
                               1.0 + eta_P * beta, strength, fit_linear, eta_w / (double)B,
                               1 + eta_w * alpha, sg_norms.as<double>(), ms,
                               (const PsgdBatch*)nullptr, (int*)nullptr);
            prof_end(1);
            if (mich) {
                prof_begin(2, 0);
                hipLaunchKernelGGL(psgd_mich_finish_kernel, dim3(nb_fin), dim3(kBlock), 0, stream,
                                   ms, strength, (const PsgdBatch*)nullptr, (const int*)nullptr);
                // the iteration is monotone after the first sweep, so it terminates (<= d
                // sweeps; 2-4 with the warm start); the host looks at the flag per chunk
                for (int guard = 0;; ++guard) {
                    for (int sweep = 0; sweep < 2; ++sweep) {
                        hipLaunchKernelGGL((psgd_mich_reduce_kernel<L>), dim3(nb_dense),
                                           dim3(kBlock), 0, stream, Pt.as<double>(),
                                           sg_norms.as<double>(), n_orders, k, d, reg, ms);
                        hipLaunchKernelGGL(psgd_mich_finish_kernel, dim3(nb_fin), dim3(kBlock),
                                           0, stream, ms, strength, (const PsgdBatch*)nullptr,
                                           (const int*)nullptr);
                    }
                    hipLaunchKernelGGL(psgd_mich_check_kernel, dim3(1), dim3(kBlock), 0, stream,
                                       ms);
                    HIPC(hipMemcpyAsync(h_done, sg_done.p, sizeof(int), hipMemcpyDeviceToHost,
                                        stream));
                    HIPC(hipStreamSynchronize(stream));
                    if (*h_done) break;
                    if (guard > d) FAIL(SPFM_ERR_RUNTIME, "psgd: prox support search did not settle");
                }
                hipLaunchKernelGGL((psgd_mich_apply_kernel<L>), dim3(nb_dense), dim3(kBlock), 0,
                                   stream, Pt.as<double>(), sg_norms.as<double>(), n_orders, k, d,
                                   reg, sg_thr.as<double>());
                prof_end(2);
            }
            *it += 1;
        }
        HIPC(hipGetLastError());
        return SPFM_OK;
    }

    // optimizer/psgd.py:125-199: one pass over indices_samples
    int psgd_epoch(int degree, double alpha, double beta, double gamma, double eta0, int lr,
                   double power_t, int64_t batch_size, const int32_t* indices_samples,
                   int64_t n_samples, int fit_linear, int64_t* it, double* sum_loss) {
        if (!have_data || !have_params || !configured)
            FAIL(SPFM_ERR_INVALID, "epoch: data, parameters and configuration are required");
        if (solver != SPFM_SOLVER_PSGD) FAIL(SPFM_ERR_INVALID, "engine is not configured for psgd");
        if (dist()) FAIL(SPFM_ERR_UNSUPPORTED, "psgd: multi-GPU is not supported");
        if (degree != top_degree) FAIL(SPFM_ERR_INVALID, "psgd: degree differs from configure()");
        if (!indices_samples || !it || n_samples != n)
            FAIL(SPFM_ERR_INVALID, "psgd: indices_samples must list every sample once");
        if (batch_size < 1) FAIL(SPFM_ERR_INVALID, "psgd: batch_size must be >= 1");
        if (lr < 0 || lr > 3) FAIL(SPFM_ERR_INVALID, "psgd: learning_rate is not supported.");
        if (*it < 1) FAIL(SPFM_ERR_INVALID, "psgd: it must be >= 1");
        {
            std::vector<char> seen((size_t)n, 0);
            for (int64_t q = 0; q < n; ++q) {
                const int i = indices_samples[q];
                if (i < 0 || i >= n || seen[(size_t)i])
                    FAIL(SPFM_ERR_INVALID, "psgd: indices_samples is not a permutation");
                seen[(size_t)i] = 1;
            }
        }
        if (n == 0) {
            if (sum_loss) *sum_loss = 0.0;
            return SPFM_OK;
        }
        int rc = ensure_pt();
        if (rc) return rc;
        p_valid = false;
        HIPC(hipMemcpyAsync(sg_samples.p, indices_samples, sizeof(int32_t) * (size_t)n,
                            hipMemcpyHostToDevice, stream));
        HIPC(hipStreamSynchronize(stream));  // caller may reuse indices_samples
#define SPFM_PSGD_GO(T, L)                                                                    \
    rc = psgd_epoch_tl<T, L>(degree, alpha, beta, gamma, eta0, lr, power_t, batch_size,        \
                             fit_linear, it)
        if (dtype == SPFM_F32) {
            if (k <= 16) SPFM_PSGD_GO(float, 16);
            else if (k <= 32) SPFM_PSGD_GO(float, 32);
            else SPFM_PSGD_GO(float, 64);
        } else {
            if (k <= 16) SPFM_PSGD_GO(double, 16);
            else if (k <= 32) SPFM_PSGD_GO(double, 32);
            else SPFM_PSGD_GO(double, 64);
        }
#undef SPFM_PSGD_GO
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_partial_kernel, dim3(256), dim3(kBlock), 0, stream,
                           pred_tmp.as<double>(), n, partial.as<double>());
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(kBlock), 0, stream,
                           partial.as<double>(), 256, scalar.as<double>());
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(h_scalar, scalar.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        prof_collect();
        if (sum_loss) *sum_loss = h_scalar[0];
        psgd_warm = true;
        return SPFM_OK;
    }
};

// ======================================================================= C ABI
#define GUARD(h)                  \
    if (!(h)) return SPFM_ERR_INVALID; \
    if (hipSetDevice((h)->device) != hipSuccess) { \
        (h)->err = "hipSetDevice failed";          \
        return SPFM_ERR_RUNTIME;                   \
    }

extern "C" {

int spfm_create(spfm_handle* out, int device_id, int dtype) {
    if (!out) return SPFM_ERR_INVALID;
    *out = nullptr;
    if (dtype != SPFM_F32 && dtype != SPFM_F64) {
        g_create_error = "dtype must be SPFM_F32 or SPFM_F64";
        return SPFM_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return SPFM_ERR_RUNTIME;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_error = "device id out of range";
        return SPFM_ERR_INVALID;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return SPFM_ERR_RUNTIME;
    }
    spfm_engine* h = new spfm_engine();
    h->device = device_id;
    h->dtype = dtype;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->devname = prop.gcnArchName;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete h;
        return SPFM_ERR_RUNTIME;
    }
    *out = h;
    return SPFM_OK;
}

void spfm_destroy(spfm_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
}

const char* spfm_last_error(spfm_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

#ifndef SPFM_BUILD_TAG
#define SPFM_BUILD_TAG "untagged"
#endif
const char* spfm_build_tag(void) { return SPFM_BUILD_TAG; }

int spfm_device_name(spfm_handle h, char* out, int cap) {
    if (!h || !out || cap <= 0) return SPFM_ERR_INVALID;
    snprintf(out, (size_t)cap, "%s", h->devname.c_str());
    return SPFM_OK;
}

int spfm_set_data_csc(spfm_handle h, int64_t n, int32_t d, const int64_t* indptr,
                      const int32_t* indices, const double* data, const double* y) {
    GUARD(h);
    return h->set_data(n, d, indptr, indices, data, y);
}

int spfm_set_data_csr(spfm_handle h, int64_t n, int32_t d, const int64_t* indptr,
                      const int32_t* indices, const double* data, const double* y) {
    GUARD(h);
    return h->set_data_csr(n, d, indptr, indices, data, y);
}

int spfm_set_params(spfm_handle h, int n_orders, int k, int32_t d, const double* P,
                    const double* w, const double* lams) {
    GUARD(h);
    return h->set_params(n_orders, k, d, P, w, lams);
}

int spfm_get_params(spfm_handle h, double* P, double* w) {
    GUARD(h);
    return h->get_params(P, w);
}

int spfm_configure(spfm_handle h, int solver, int loss, int regularizer, int top_degree) {
    GUARD(h);
    return h->configure(solver, loss, regularizer, top_degree);
}

int spfm_init_pred(spfm_handle h, int degree, int fit_linear, int add_lower_deg2) {
    GUARD(h);
    return h->init_pred(degree, fit_linear, add_lower_deg2);
}

int spfm_get_y_pred(spfm_handle h, double* out) {
    GUARD(h);
    if (!h->have_data || !out) return SPFM_ERR_INVALID;
    return h->dtype == SPFM_F32 ? h->get_y_pred_t<float>(out) : h->get_y_pred_t<double>(out);
}

int spfm_loss_sum(spfm_handle h, double* out) {
    GUARD(h);
    if (!h->have_data || !h->configured || !out) {
        h->err = "loss_sum: data and configuration required";
        return SPFM_ERR_INVALID;
    }
    return h->dtype == SPFM_F32 ? h->loss_sum_t<float>(out) : h->loss_sum_t<double>(out);
}

int spfm_predict_csr(spfm_handle h, int64_t n, const int64_t* indptr, const int32_t* indices,
                     const double* data, int degree, int fit_linear, int add_lower_deg2,
                     double* out) {
    GUARD(h);
    if (!h->have_params) {
        h->err = "predict: no parameters set";
        return SPFM_ERR_INVALID;
    }
    if (n < 0 || !indptr || !out) return SPFM_ERR_INVALID;
    return h->dtype == SPFM_F32
               ? h->predict_csr_t<float>(n, indptr, indices, data, degree, fit_linear,
                                         add_lower_deg2, out)
               : h->predict_csr_t<double>(n, indptr, indices, data, degree, fit_linear,
                                          add_lower_deg2, out);
}

int spfm_set_schedule(spfm_handle h, int mode, const int32_t* indices_feature,
                      const int64_t* conflict_indptr, const int32_t* conflict_indices,
                      int64_t conflict_n_rows, int32_t* order_out, int32_t* n_batches_out) {
    GUARD(h);
    return h->set_schedule(mode, indices_feature, conflict_indptr, conflict_indices,
                           conflict_n_rows, order_out, n_batches_out);
}

int spfm_set_schedule_raw(spfm_handle h, const int32_t* order, const int32_t* batch_ptr,
                          int32_t n_batches, const int64_t* conflict_indptr,
                          const int32_t* conflict_indices, int64_t conflict_n_rows) {
    GUARD(h);
    return h->set_schedule_raw(order, batch_ptr, n_batches, conflict_indptr, conflict_indices,
                               conflict_n_rows);
}

int spfm_get_schedule(spfm_handle h, int32_t* order_out, int32_t* batch_ptr_out,
                      int32_t* n_batches_out) {
    if (!h) return SPFM_ERR_INVALID;
    if (!h->have_schedule) {
        h->err = "get_schedule: no schedule installed";
        return SPFM_ERR_INVALID;
    }
    if (order_out)
        std::memcpy(order_out, h->order.data(), sizeof(int32_t) * h->order.size());
    if (batch_ptr_out)
        std::memcpy(batch_ptr_out, h->batch_ptr.data(), sizeof(int32_t) * h->batch_ptr.size());
    if (n_batches_out) *n_batches_out = (int32_t)h->batch_ptr.size() - 1;
    return SPFM_OK;
}

int spfm_schedule_build(int mode, int64_t n_rows, int32_t d, const int64_t* indptr,
                        const int32_t* indices, const int32_t* indices_feature, int max_batch,
                        int32_t* order_out, int32_t* batch_ptr_out, int32_t* n_batches_out) {
    if (n_rows < 0 || d <= 0 || !indptr || !indices_feature || !order_out || !batch_ptr_out ||
        !n_batches_out)
        return SPFM_ERR_INVALID;
    if (max_batch <= 0) max_batch = 4096;
    std::vector<char> seen((size_t)d, 0);
    for (int q = 0; q < d; ++q) {
        const int j = indices_feature[q];
        if (j < 0 || j >= d || seen[(size_t)j]) return SPFM_ERR_INVALID;
        seen[(size_t)j] = 1;
    }
    std::vector<int32_t> order, bp;
    if (mode == SPFM_SCHED_EXACT) {
        order.assign(indices_feature, indices_feature + d);
        schedule_exact(n_rows, d, indptr, indices, indices_feature, max_batch, bp);
    } else if (mode == SPFM_SCHED_COLORED) {
        schedule_colored(n_rows, d, indptr, indices, indices_feature, max_batch, order, bp);
    } else {
        return SPFM_ERR_INVALID;
    }
    std::memcpy(order_out, order.data(), sizeof(int32_t) * (size_t)d);
    std::memcpy(batch_ptr_out, bp.data(), sizeof(int32_t) * bp.size());
    *n_batches_out = (int32_t)bp.size() - 1;
    return SPFM_OK;
}

int spfm_cd_linear_epoch(spfm_handle h, double alpha, double* viol) {
    GUARD(h);
    return h->cd_linear_epoch(alpha, viol);
}

int spfm_pcd_epoch(spfm_handle h, int order_idx, int degree, double beta, double gamma,
                   double eta, const int32_t* indices_component, int n_comp, double* viol) {
    GUARD(h);
    return h->pcd_epoch(order_idx, degree, beta, gamma, eta, indices_component, n_comp, viol);
}

int spfm_pbcd_epoch(spfm_handle h, int order_idx, int degree, double beta, double gamma,
                    double eta, double* viol) {
    GUARD(h);
    return h->pbcd_epoch(order_idx, degree, beta, gamma, eta, viol);
}

int spfm_host_epoch_begin(spfm_handle h, int order_idx, int degree) {
    GUARD(h);
    return h->host_epoch_begin(order_idx, degree);
}
int spfm_host_pass_begin(spfm_handle h, int component) {
    GUARD(h);
    return h->host_pass_begin(component);
}
int spfm_host_step_sums(spfm_handle h, int step, double* sums_out) {
    GUARD(h);
    return h->host_step(true, step, sums_out, nullptr, nullptr);
}
int spfm_host_step_apply(spfm_handle h, int step, const double* p_new, const double* p_old) {
    GUARD(h);
    return h->host_step(false, step, nullptr, p_new, p_old);
}
int spfm_host_epoch_end(spfm_handle h, double* viol) {
    GUARD(h);
    return h->host_epoch_end(viol);
}

int spfm_psgd_epoch(spfm_handle h, int degree, double alpha, double beta, double gamma,
                    double eta0, int learning_rate, double power_t, int64_t batch_size,
                    const int32_t* indices_samples, int64_t n_samples, int fit_linear,
                    int64_t* it, double* sum_loss) {
    GUARD(h);
    return h->psgd_epoch(degree, alpha, beta, gamma, eta0, learning_rate, power_t, batch_size,
                         indices_samples, n_samples, fit_linear, it, sum_loss);
}

int spfm_comm_unique_id(char* id128) {
    if (!id128) return SPFM_ERR_INVALID;
    std::string e;
    if (!g_rccl.load(e)) {
        g_create_error = e;
        return SPFM_ERR_RUNTIME;
    }
    ncclUniqueId_ id;
    if (g_rccl.GetUniqueId(&id) != 0) {
        g_create_error = "ncclGetUniqueId failed";
        return SPFM_ERR_RUNTIME;
    }
    std::memcpy(id128, id.internal, 128);
    return SPFM_OK;
}

int spfm_comm_init(spfm_handle h, const char* id128, int n_ranks, int rank) {
    GUARD(h);
    if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SPFM_ERR_INVALID;
    if (!g_rccl.load(h->err)) return SPFM_ERR_RUNTIME;
    ncclUniqueId_ id;
    std::memcpy(id.internal, id128, 128);
    int rc = g_rccl.CommInitRank(&h->comm, n_ranks, id, rank);
    if (rc != 0) {
        h->err = std::string("ncclCommInitRank: ") +
                 (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
        h->comm = nullptr;
        return SPFM_ERR_RUNTIME;
    }
    h->n_ranks = n_ranks;
    h->rank = rank;
    h->col_norm_reduced = false;
    h->clear_graphs();
    return SPFM_OK;
}

int spfm_comm_init_shm(spfm_handle h, const char* shm_name, int n_ranks, int rank) {
    GUARD(h);
    if (!shm_name || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return SPFM_ERR_INVALID;
    if (h->comm || h->shm.hdr) {
        h->err = "a communicator is already attached";
        return SPFM_ERR_INVALID;
    }
    using Shm = spfm_engine::ShmComm;
    const size_t bytes = sizeof(Shm::Hdr) + sizeof(double) * Shm::kMaxDoubles * (size_t)n_ranks;
    int fd = shm_open(shm_name, O_RDWR | O_CREAT, 0600);
    if (fd < 0) {
        h->err = std::string("shm_open failed: ") + std::strerror(errno);
        return SPFM_ERR_RUNTIME;
    }
    if (ftruncate(fd, (off_t)bytes) != 0) {  // new segments are zero-filled
        h->err = std::string("ftruncate failed: ") + std::strerror(errno);
        close(fd);
        return SPFM_ERR_RUNTIME;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        h->err = std::string("mmap failed: ") + std::strerror(errno);
        return SPFM_ERR_RUNTIME;
    }
    h->shm.hdr = reinterpret_cast<Shm::Hdr*>(p);
    h->shm.slots = reinterpret_cast<double*>(reinterpret_cast<char*>(p) + sizeof(Shm::Hdr));
    h->shm.bytes = bytes;
    h->shm.local_sense = 0;
    h->n_ranks = n_ranks;
    h->rank = rank;
    h->col_norm_reduced = false;
    h->clear_graphs();
    return SPFM_OK;
}

int spfm_peer_alloc(spfm_handle h, char* handle64) {
    GUARD(h);
    if (!handle64) return SPFM_ERR_INVALID;
    if (!h->peer_own) {
        void* p = nullptr;
        // fine-grained device memory: remote stores become visible to local polling loads
        // without a kernel boundary.  There is NO coarse-grained fall-back: plain hipMalloc
        // memory maps just as well over IPC, but does not promise that visibility, and the first
        // persistent pass would spin into its time-out -- the caller takes the per-step
        // collective instead when this fails (sparsepoly_amd.distributed.connect_peers).
        hipError_t ae = hipExtMallocWithFlags(&p, sizeof(double) * spfm_engine::kPeerDoubles,
                                              hipDeviceMallocFinegrained);
        if (ae != hipSuccess) {
            (void)hipGetLastError();
            h->err = std::string("peer slab: fine-grained device memory is not available (") +
                     hipGetErrorString(ae) + "); use the per-step collective";
            return SPFM_ERR_RUNTIME;
        }
        if (hipMemsetAsync(p, 0, sizeof(double) * spfm_engine::kPeerDoubles, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            (void)hipFree(p);
            h->err = "peer slab memset failed";
            return SPFM_ERR_RUNTIME;
        }
        h->peer_own = p;
    }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t ih;
    hipError_t e = hipIpcGetMemHandle(&ih, h->peer_own);
    if (e != hipSuccess) {
        h->err = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e);
        return SPFM_ERR_RUNTIME;
    }
    std::memcpy(handle64, &ih, 64);
    return SPFM_OK;
}

int spfm_peer_connect(spfm_handle h, int n_ranks, int rank, const char* handles) {
    GUARD(h);
    if (!handles || n_ranks < 2 || n_ranks > 8 || rank < 0 || rank >= n_ranks)
        return SPFM_ERR_INVALID;
    if (!h->peer_own) {
        h->err = "spfm_peer_connect: call spfm_peer_alloc first";
        return SPFM_ERR_INVALID;
    }
    if (!h->dist() || h->n_ranks != n_ranks || h->rank != rank) {
        h->err = "spfm_peer_connect: attach the communicator (spfm_comm_init[_shm]) with the "
                 "same ranks first";
        return SPFM_ERR_INVALID;
    }
    h->peer_ptr.assign((size_t)n_ranks, nullptr);
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) {
            h->peer_ptr[(size_t)r] = h->peer_own;
            continue;
        }
        hipIpcMemHandle_t ih;
        std::memcpy(&ih, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            h->err = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e);
            return SPFM_ERR_RUNTIME;
        }
        h->peer_ptr[(size_t)r] = p;
    }
    std::vector<double*> t1((size_t)n_ranks), t2((size_t)n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        t1[(size_t)r] = reinterpret_cast<double*>(h->peer_ptr[(size_t)r]) + spfm_engine::kPeerPcdOff;
        t2[(size_t)r] = reinterpret_cast<double*>(h->peer_ptr[(size_t)r]) + spfm_engine::kPeerPbOff;
    }
    if (h->peer_tab_pcd.alloc(sizeof(double*) * 8) != hipSuccess ||
        h->peer_tab_pb.alloc(sizeof(double*) * 8) != hipSuccess ||
        hipMemcpyAsync(h->peer_tab_pcd.p, t1.data(), sizeof(double*) * (size_t)n_ranks,
                       hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(h->peer_tab_pb.p, t2.data(), sizeof(double*) * (size_t)n_ranks,
                       hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
        h->err = "peer table upload failed";
        return SPFM_ERR_RUNTIME;
    }
    // handshake: every rank's store into every slab must reach a kernel that is already polling
    // (what the persistent passes assume); all ranks are in this call together
    {
        DevBuf okb;
        int ok = 0;
        h->peer_generation += 1;
        const unsigned long long word = 0x5350464d00000000ull + (unsigned)h->peer_generation;
        const unsigned long long ticks = 100ull * 1000 * 1000 * 10;  // 10 s of the 100 MHz counter
        if (okb.alloc(sizeof(int) * 4) != hipSuccess ||
            hipMemsetAsync(okb.p, 0, sizeof(int) * 4, h->stream) != hipSuccess) {
            h->err = "peer handshake: allocation failed";
            return SPFM_ERR_RUNTIME;
        }
        hipLaunchKernelGGL(peer_probe_kernel, dim3(1), dim3(kWave), 0, h->stream,
                           h->peer_tab_pcd.as<double*>(),
                           spfm_engine::kPeerProbeOff - spfm_engine::kPeerPcdOff, n_ranks, rank,
                           word, ticks, okb.as<int>());
        if (hipMemcpyAsync(&ok, okb.p, sizeof(int), hipMemcpyDeviceToHost, h->stream) !=
                hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            h->err = "peer handshake kernel failed";
            return SPFM_ERR_RUNTIME;
        }
        if (!ok) {
            h->err = "peer handshake timed out: a peer's store into this GPU's exchange slab did "
                     "not become visible to a running kernel (use the per-step collective)";
            return SPFM_ERR_RUNTIME;
        }
    }
    h->peer_ready = true;
    h->have_schedule = false;  // the step cap depends on the engine: set the schedule again
    h->prb_ready = false;
    h->pb_stream_ready = false;
    h->wide_ready = false;
    h->clear_graphs();
    return SPFM_OK;
}

int spfm_profile_enable(spfm_handle h, int on) {
    if (!h) return SPFM_ERR_INVALID;
    h->prof_on = on != 0;
    return SPFM_OK;
}

int spfm_profile_get(spfm_handle h, int which, double* ms, int64_t* launches, int64_t* nnz) {
    if (!h || which < 0 || which > 4) return SPFM_ERR_INVALID;
    if (ms) *ms = h->prof[which].ms;
    if (launches) *launches = h->prof[which].launches;
    if (nnz) *nnz = h->prof[which].nnz;
    return SPFM_OK;
}

int spfm_profile_reset(spfm_handle h) {
    if (!h) return SPFM_ERR_INVALID;
    for (auto& ps : h->prof) {
        ps.ms = 0;
        ps.launches = 0;
        ps.nnz = 0;
        ps.used = 0;
    }
    return SPFM_OK;
}

int spfm_set_option(spfm_handle h, const char* key, int value) {
    if (!h || !key) return SPFM_ERR_INVALID;
    const std::string k(key);
    if (k == "use_graph") {
        h->use_graph = value != 0;
    } else if (k == "fuse_chain") {
        h->fuse_chain = value != 0;
    } else if (k == "persistent") {
        h->persistent = value != 0;
        h->prb_ready = false;
    } else if (k == "prb_long") {
        if (value < 16) {
            h->err = "prb_long must be >= 16";
            return SPFM_ERR_INVALID;
        }
        h->prb_long = value;
        h->prb_ready = false;
        h->relax_state = 0;
    } else if (k == "prb_pack") {  // packed row records for degree-3 passes (rows in global memory)
        h->prb_pack = value != 0;
    } else if (k == "co_tenants") {  // concurrent fits: handles sharing the device's CUs
        if (value < 1 || value > 64) {
            h->err = "co_tenants must be in [1, 64]";
            return SPFM_ERR_INVALID;
        }
        h->co_tenants = value;
        // every tenant keeps to its share of the CUs (one persistent workgroup per CU)
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
        const int share = std::max(1, ncu / value);
        if (h->prb_G > share) {
            h->prb_G = share;
            h->prb_ready = false;
            h->relax_state = 0;
        }
        if (h->pbprb_G > share) {
            h->pbprb_G = share;
            h->pb_stream_ready = false;
        }
        h->wide_ready = false;  // the wide pass caps itself (wide_groups)
    } else if (k == "stream_device") {  // entry stream of the 64-column passes: device or host threads
        h->stream_device = value != 0;
        h->prb_ready = false;
    } else if (k == "colour_device") {  // first-fit colouring on the device (default) or by host threads
        h->colour_device = value != 0;
    } else if (k == "ingest_device") {  // CSR -> CSC on the device (default) or by host threads
        h->ingest_device = value != 0;
    } else if (k == "relax") {  // merged steps for schedules of tiny steps
        h->relax_on = value != 0;
        h->relax_state = 0;
    } else if (k == "prb_stamps") {
        h->prb_stamp_on = value != 0;
    } else if (k == "debug_spin_max") {  // test hook: polls before a persistent pass gives up
        if (value < 64) {
            h->err = "debug_spin_max must be >= 64";
            return SPFM_ERR_INVALID;
        }
        h->spin_max = (unsigned)value;
    } else if (k == "debug_drop_group") {  // test hook: the next `value` persistent launches
        h->debug_drop = value;             // lack their last workgroup (they time out)
    } else if (k == "persistent_failed") {  // 0: try the persistent passes again
        h->pers_failed = value != 0;
    } else if (k == "psgd_graph_sweeps") {
        if (value < 0 || value > 64) {
            h->err = "psgd_graph_sweeps must be in 0..64";
            return SPFM_ERR_INVALID;
        }
        h->psgd_graph_sweeps = value;
    } else if (k == "psgd_eager") {
        h->psgd_force_eager = value != 0;
    } else if (k == "pbcd_fuse") {
        h->pbcd_fuse = value != 0;
    } else if (k == "wide") {
        h->wide_on = value != 0;
    } else if (k == "pcdw_groups") {
        if (value < 1) {
            h->err = "pcdw_groups must be >= 1";
            return SPFM_ERR_INVALID;
        }
        h->pcdw_G = value;
        h->wide_ready = false;
    } else if (k == "pbcd_persistent") {
        h->pb_persistent = value != 0;
    } else if (k == "peer_exchange") {
        // 0: give the in-kernel cross-GPU exchange up (a rank could not map its peers): the
        // passes fall back to the per-step collective.  (1 is set by spfm_peer_connect only.)
        if (value != 0) {
            h->err = "peer_exchange: only 0 can be set; connect with spfm_peer_connect";
            return SPFM_ERR_INVALID;
        }
        h->peer_ready = false;
        h->have_schedule = false;
        h->prb_ready = false;
        h->pb_stream_ready = false;
        h->wide_ready = false;
    } else if (k == "probe_xcd") {
        h->probe_xcd = (int)value;
    } else if (k == "probe_lds") {
        h->probe_lds = (int)value;
    } else if (k == "pbprb_dbg") {
        h->pb_dbg = value;
    } else if (k == "wide_min_cols") {
        h->wide_min_cols = value;
    } else if (k == "pcdw_stamps") {
        h->wide_stamp_on = value != 0;
    } else if (k == "pbprb_stamps") {
        h->pb_stamp_on = value != 0;
    } else if (k == "pbprb_owners") {  // dedicated owner workgroups (default 0)
        if (value < 0 || value > 64) {
            h->err = "pbprb_owners must be in 0..64";
            return SPFM_ERR_INVALID;
        }
        h->pbprb_owners = value;
        h->pb_stream_ready = false;
    } else if (k == "pbprb_groups") {
        if (value < 1) {
            h->err = "pbprb_groups must be >= 1";
            return SPFM_ERR_INVALID;
        }
        h->pbprb_G = value;
        h->pb_stream_ready = false;
    } else if (k == "prb_lds") {
        h->prb_lds = value != 0;
    } else if (k == "prb_groups") {
        if (value < 1) {
            h->err = "prb_groups must be >= 1";
            return SPFM_ERR_INVALID;
        }
        h->prb_G = value;
        h->prb_ready = false;
        h->relax_state = 0;
    } else if (k == "max_batch") {
        if (value < 1) {
            h->err = "max_batch must be >= 1";
            return SPFM_ERR_INVALID;
        }
        h->max_batch_opt = value;
    } else {
        h->err = "unknown option: " + k;
        return SPFM_ERR_INVALID;
    }
    h->clear_graphs();
    return SPFM_OK;
}

int spfm_get_option(spfm_handle h, const char* key, int* value) {
    if (!h || !key || !value) return SPFM_ERR_INVALID;
    const std::string k(key);
    if (k == "use_graph") *value = h->use_graph;
    else if (k == "fuse_chain") *value = h->fuse_chain;
    else if (k == "max_batch") *value = h->max_batch_opt;
    else if (k == "persistent") *value = h->persistent;
    else if (k == "prb_groups") *value = h->prb_G;
    else if (k == "prb_lds") *value = h->prb_lds;
    else if (k == "psgd_redone") *value = h->psgd_redone;
    else if (k == "prb_lds_active") *value = h->prb_lds_active;
    else if (k == "pbcd_persistent") *value = h->pb_persistent;
    else if (k == "wide") *value = h->wide_on;
    else if (k == "wide_active") *value = h->have_schedule && h->wide_usable();
    else if (k == "wide_lds_active") *value = h->wide_lr_active;
    else if (k == "pbprb_groups") *value = h->pbprb_G;
    else if (k == "pcdw_groups") *value = h->wide_ready ? h->wide_G : h->pcdw_G;  // 0 = not chosen yet
    else if (k == "pbprb_owners") *value = h->pb_GO;
    else if (k == "pbprb_active") *value = h->pbprb_active;
    else if (k == "persistent_active")
        *value = h->have_schedule && (h->prb_usable() || h->wide_usable());
    else if (k == "relax") *value = h->relax_on;
    else if (k == "ingest_device") *value = h->ingest_device;
    else if (k == "colour_device") *value = h->colour_device;
    else if (k == "colour_device_used") *value = h->colour_device_used;
    else if (k == "stream_device") *value = h->stream_device;
    else if (k == "stream_device_used") *value = h->stream_device_used;
    else if (k == "co_tenants") *value = h->co_tenants;
    else if (k == "ingest_device_used") *value = h->ingest_device_used;
    else if (k == "prb_pack_active") *value = h->prb_pack_active;
    else if (k == "relax_steps")
        *value = h->relax_state == 1 ? (int)h->r_batch_ptr.size() - 1 : 0;
    else if (k == "persistent_fallbacks") *value = h->pers_fallbacks;
    else if (k == "persistent_failed") *value = h->pers_failed;
    else if (k == "n_ranks") *value = h->dist() ? h->n_ranks : 1;
    else if (k == "peer_ready") *value = h->peer_ready;
    else if (k == "wide_min_cols") *value = h->wide_min_cols;
    else {
        h->err = "unknown option: " + k;
        return SPFM_ERR_INVALID;
    }
    return SPFM_OK;
}

int spfm_debug_prb_stamps(spfm_handle h, long long* out, int cap) {
    GUARD(h);
    if ((h->pb_dbg & 8) && out && cap >= 8) {  // diagnostic counters of the persistent pbcd pass
        static unsigned v[16 + 4096];
        if (hipMemcpy(v, h->pb_dbgbuf.p, sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
            return SPFM_ERR_RUNTIME;
        int i = 0;
        for (; i < 16 + 4096 && i < cap; ++i) out[i] = (long long)v[i];
        return i - (i % 16);
    }
    if (h->wide_stamp_on && h->wide_stamps.p && out) {  // wide pcd pass's timers
        const int nv = (int)(h->wide_stamps.bytes / sizeof(long long));
        if (cap < nv) return SPFM_ERR_INVALID;
        if (hipMemcpy(out, h->wide_stamps.p, sizeof(long long) * (size_t)nv,
                      hipMemcpyDeviceToHost) != hipSuccess)
            return SPFM_ERR_RUNTIME;
        return nv;
    }
    if (h->pb_stamp_on && h->pb_stream_ready && out) {  // persistent pbcd pass's timers
        const int nv = 16 * (h->pb_stream_G + h->pb_GO);
        if (cap < nv) return SPFM_ERR_INVALID;
        if (hipMemcpy(out, h->pb_stamps.p, sizeof(long long) * (size_t)nv,
                      hipMemcpyDeviceToHost) != hipSuccess)
            return SPFM_ERR_RUNTIME;
        return nv;
    }
    if (!h->prb_ready || !out) return SPFM_ERR_INVALID;
    const int nval = 16 * h->prb_G;
    if (cap < nval) return SPFM_ERR_INVALID;
    if (hipMemcpy(out, h->prb_stamps.p, sizeof(long long) * (size_t)nval, hipMemcpyDeviceToHost) !=
        hipSuccess)
        return SPFM_ERR_RUNTIME;
    return nval;
}

int spfm_debug_hop_latency(spfm_handle h, int partner, int rounds, double* ns_per_hop,
                           int* xcc_ids /* [2] */) {
    GUARD(h);
    if (partner < 1 || partner > 255 || rounds < 1 || rounds > (1 << 20) || !ns_per_hop)
        return SPFM_ERR_INVALID;
    DevBuf words, info;
    if (words.alloc(sizeof(unsigned long long) * 32) != hipSuccess ||
        info.alloc(sizeof(int) * 4) != hipSuccess)
        return SPFM_ERR_RUNTIME;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return SPFM_ERR_RUNTIME;
    float best = 1e30f;
    int hinfo[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {  // first repetition warms the code path
        (void)hipMemsetAsync(words.p, 0, sizeof(unsigned long long) * 32, h->stream);
        (void)hipMemsetAsync(info.p, 0, sizeof(int) * 4, h->stream);
        (void)hipEventRecord(e0, h->stream);
        hipLaunchKernelGGL(hop_pingpong_kernel, dim3(partner + 1), dim3(kWave), 0, h->stream,
                           words.as<unsigned long long>(), rounds, partner, info.as<int>());
        (void)hipEventRecord(e1, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess) {
            h->err = "hop_pingpong_kernel failed";
            return SPFM_ERR_RUNTIME;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hinfo, info.p, sizeof(int) * 4, hipMemcpyDeviceToHost);
        if (hinfo[2] != 0) {
            h->err = "hop latency probe: partner workgroup did not respond";
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return SPFM_ERR_RUNTIME;
        }
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ns_per_hop = (double)best * 1e6 / (2.0 * rounds);
    if (xcc_ids) {
        xcc_ids[0] = hinfo[0];
        xcc_ids[1] = hinfo[1];
    }
    return SPFM_OK;
}

int spfm_debug_exchange_cost(spfm_handle h, int groups, int ncols, int readers_mod, int rounds,
                             double* ns_per_round) {
    GUARD(h);
    if (groups < 1 || groups > 256 || ncols < 1 || ncols > 64 || readers_mod == 0 ||
        rounds < 1 || rounds > (1 << 20) || !ns_per_round)
        return SPFM_ERR_INVALID;
    DevBuf slab, abortw;
    const size_t bytes = sizeof(double) * 2 * (2 * (size_t)groups + 2) * 64 * 2;
    if (slab.alloc(bytes) != hipSuccess || abortw.alloc(16) != hipSuccess) return SPFM_ERR_RUNTIME;
    PrbArgs a{};
    a.G = groups;
    a.slab = slab.as<double>();
    a.abort_flag = abortw.as<unsigned>();
    a.spin_max = 1u << 21;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return SPFM_ERR_RUNTIME;
    (void)hipFuncSetAttribute((const void*)exchange_probe_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->kPrbLds);
    float best = 1e30f;
    unsigned aborted = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemsetAsync(slab.p, 0, bytes, h->stream);
        (void)hipMemsetAsync(abortw.p, 0, 16, h->stream);
        (void)hipEventRecord(e0, h->stream);
        // probe_xcd = x+1: single-XCD variant on XCD x (8x oversized grid, see the kernel)
        hipLaunchKernelGGL(exchange_probe_kernel, dim3(h->probe_xcd ? 8 * groups + 64 : groups),
                           dim3(readers_mod == -12 ? 768 : 512),
                           h->probe_xcd ? (size_t)h->probe_lds : (size_t)h->kPrbLds, h->stream, a,
                           rounds, ncols, readers_mod, h->probe_xcd);
        (void)hipEventRecord(e1, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess) {
            h->err = "exchange_probe_kernel failed";
            return SPFM_ERR_RUNTIME;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&aborted, abortw.p, sizeof(unsigned), hipMemcpyDeviceToHost);
        if (aborted) break;
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (aborted) {
        h->err = "exchange probe timed out (workgroups not co-resident?)";
        return SPFM_ERR_RUNTIME;
    }
    *ns_per_round = (double)best * 1e6 / rounds;
    return SPFM_OK;
}

int spfm_debug_stream_probe(spfm_handle h, int64_t* bytes_out) {
    GUARD(h);
    if (!h->have_schedule || !h->prb_usable()) {
        h->err = "stream probe: needs a schedule the 64-column persistent pass can run";
        return SPFM_ERR_INVALID;
    }
    int rc = h->dtype == SPFM_F32 ? h->ensure_prb<float>() : h->ensure_prb<double>();
    if (rc) return rc;
    DevBuf sink;
    if (sink.alloc(sizeof(double) * (size_t)h->prb_G * kPrbThreads) != hipSuccess) {
        h->err = "stream probe: allocation failed";
        return SPFM_ERR_RUNTIME;
    }
    const PrbArgs a = h->prb_args();
    if (h->dtype == SPFM_F32)
        hipLaunchKernelGGL((prb_stream_probe_kernel<float>), dim3(h->prb_G), dim3(kPrbThreads), 0,
                           h->stream, a, h->prb_eval.as<float>(), sink.as<double>());
    else
        hipLaunchKernelGGL((prb_stream_probe_kernel<double>), dim3(h->prb_G), dim3(kPrbThreads), 0,
                           h->stream, a, h->prb_eval.as<double>(), sink.as<double>());
    if (hipStreamSynchronize(h->stream) != hipSuccess) {
        h->err = "stream probe kernel failed";
        return SPFM_ERR_RUNTIME;
    }
    // requested bytes: per entry a 4-byte row id and a value; per (workgroup, step) the slot
    // bounds of its columns (+1) as 4-byte words
    if (bytes_out)
        *bytes_out = h->nnz * (int64_t)(4 + h->tsize()) +
                     (int64_t)h->prb_G * ((int64_t)h->d + h->n_batches()) * 4;
    return SPFM_OK;
}

int spfm_debug_branch_counts(spfm_handle h, unsigned* out8, int reset) {
    GUARD(h);
    if (!out8) return SPFM_ERR_INVALID;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return SPFM_ERR_RUNTIME;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_branch_count), sizeof(unsigned) * BR_COUNT) !=
        hipSuccess)
        return SPFM_ERR_RUNTIME;
    if (reset) {
        const unsigned zero[BR_COUNT] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_branch_count), zero, sizeof zero) != hipSuccess)
            return SPFM_ERR_RUNTIME;
    }
    return SPFM_OK;
}

int spfm_set_use_graph(spfm_handle h, int on) {
    if (!h) return SPFM_ERR_INVALID;
    h->use_graph = on != 0;
    if (!on) h->clear_graphs();
    return SPFM_OK;
}

}  // extern "C"
