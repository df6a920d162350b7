// spfm_engine.hip.h -- the host engine behind the C ABI (include/spfm.h): one object per handle.
// The member functions are defined in spfm_engine_*.hip along the engine's seams (core: data,
// parameters, schedule, predict, communicators, recovery, C ABI; pcd: multi-kernel pcd /
// cd_linear and the epoch drivers; prb: persistent 64-column passes, one translation unit per
// storage type; wide: wide persistent passes; pbcd: multi-kernel pbcd; pbprb: persistent pbcd
// pass, one unit per storage type; psgd).  See DESIGN.md for the execution model.
#pragma once
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
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "../../include/spfm.h"
#include "spfm_common.hip.h"
#include "spfm_prb.hip.h"    // PrbArgs
#include "spfm_pcdw.hip.h"   // PcdwArgs, PcdwParams
#include "spfm_psgd.hip.h"   // PsgdBatch

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
                     const std::vector<int32_t>&, int, int, bool, const uint8_t*,
                     std::vector<int32_t>&, std::vector<int32_t>&, std::vector<uint8_t>&,
                     std::vector<uint8_t>&);
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
// spfm_ingest.hip: the pbcd and wide entry streams on the device (same tables as build_pb_stream /
// build_wide_stream)
hipError_t device_pb_stream(int64_t, int32_t, int64_t, int, int, int, int, const int32_t*,
                            const int32_t*, const int64_t*, const int32_t*, const int64_t*,
                            const int32_t*, int32_t*, int32_t*, uint8_t*, uint8_t*, hipStream_t);
hipError_t device_wide_stream(int64_t, int32_t, int64_t, int, int, const int32_t*, const int32_t*,
                              const int64_t*, const int32_t*, const int64_t*, const int32_t*,
                              int32_t*, int32_t*, uint8_t*, hipStream_t);
// spfm_colour.hip: the first-fit colouring on the device (same result as schedule_colored)
hipError_t device_first_fit(int64_t, int32_t, int64_t, const int64_t*, const int32_t*, const int64_t*,
                            const int32_t*, int, int32_t*, int*, int*, hipStream_t);
}  // namespace spfm

using namespace spfm;

typedef struct ncclComm* ncclComm_t;

// --------------------------------------------------------------------- buffers
// A device allocation.  Several DevBufs (of different handles) may refer to ONE allocation
// (`share`: the matrix image and the entry streams of co-tenant fits, spfm_share_data); the
// memory is freed with its last holder.  `alloc` on a shared buffer detaches first, so a handle
// can never write into -- or resize under -- memory that another handle reads.
struct DevBuf {
    struct Block {
        void* p;
        std::atomic<int> refs;
    };
    Block* blk = nullptr;
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (blk && blk->refs.fetch_sub(1) == 1) {
            (void)hipFree(blk->p);
            delete blk;
        }
        blk = nullptr;
        p = nullptr;
        bytes = 0;
    }
    bool shared() const { return blk && blk->refs.load() > 1; }
    hipError_t alloc(size_t b) {
        if (b <= bytes && p && !shared()) return hipSuccess;
        release();
        if (b == 0) b = 16;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, b);
        if (e != hipSuccess) return e;
        blk = new Block{q, {1}};
        p = q;
        bytes = b;
        return e;
    }
    void share(const DevBuf& o) {  // refer to o's allocation (read-only by convention)
        if (o.blk == blk) return;
        release();
        if (!o.blk) return;
        o.blk->refs.fetch_add(1);
        blk = o.blk;
        p = o.p;
        bytes = o.bytes;
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

// Entry streams of the persistent passes, shared by the handles that share one data image
// (spfm_share_data): whoever needs a stream first builds it (under the lock: co-tenant fits of
// one schedule do not build it twice), the others refer to the same device buffers.  The two
// most recent streams of each kind are kept (shuffle=True makes a new one per iteration).
struct StreamCache {
    std::mutex mu;
    struct Prb {
        std::string key;
        DevBuf sp, erow, eval, lmask;
        int has_long = 0;
    };
    struct Pb {
        std::string key;
        DevBuf sp, erow, eval, meta, tab;
    };
    std::vector<std::unique_ptr<Prb>> prb;
    std::vector<std::unique_ptr<Pb>> pb;
    static constexpr size_t kKeep = 2;
};

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
    // several handles on one matrix (concurrent fits, spfm_share_data): the image buffers above
    // are then shared allocations, and so are the entry streams the handles build
    std::shared_ptr<StreamCache> scache;
    uint64_t sched_hash = 0;  // of (order, batch_ptr): names a schedule in `scache`
    int share_data_from(spfm_engine* src, const double* y);

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
    bool keep_last_error = false;  // test hook (spfm_comm_init)
    bool have_pred_args = false;
    int pa_degree = 0, pa_lin = 0, pa_lower = 0;
    DevBuf snapP, snapW, snapC;
    std::map<const void*, int> resident_cache;  // kernel -> workgroups that can be resident
    std::map<std::string, bool> resident_agreed;  // several ranks: the verdict all of them took

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
    int wide_lr_active = 0;  // what the last wide pass used: 0 global rows, 1 all rows in LDS, 2 the first rows of a block in LDS
    int wide_lds_cap = -1;   // option "wide_lds_rows" (see wide_launch)
    bool wide_ep = true;     // option "wide_ep": rows (also) in global memory -> entry-parallel form
    int wide_ep_active = 0;  // what the last wide pass used
    bool wide_rec8 = true;   // option "wide_rec8": squared loss, float: 8-byte (A, residual) row records
    DevBuf w_wbase, w_wsp, w_erow, w_eval, w_slabA, w_slabB;
    // persistent pbcd pass (spfm_pbprb.hip.h): its own workgroup count, hence its own entry
    // stream when that differs from the pcd / cd_linear pass's
    bool pb_persistent = true;
    int pbprb_G = 256;
    int probe_xcd = 0, probe_lds = 60 * 1024;  // diagnostics (spfm_debug_exchange_cost)
    bool pb_stream_ready = false;
    int pb_stream_G = 0, pb_stream_NG = 0;
    DevBuf pb_tab;            // its (workgroup, step) -> slot groups map (gtab)
    // relaxed runs of the persistent pbcd pass (pbcd_prb_kernel CR): merged step boundaries,
    // their entry stream, the conflict tables
    int pbr_state = 0;        // 0 not tried for this schedule, 1 in use, -1 not worth it
    int pbr_G = 0, pb_relax_active = 0;
    std::vector<int32_t> pbr_batch_ptr;
    DevBuf pbr_bptr, pbr_sp, pbr_erow, pbr_eval, pbr_meta, pbr_tab, pbr_cfptr, pbr_cf, pbr_clist,
        pbr_slabR;
    bool pb_balance = true;   // balanced slot groups (0: the fixed map slot q -> group q % NG)
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

    ~spfm_engine();

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

    int shm_barrier();

    int allreduce_shm(double* buf, size_t count);
    int allreduce_shm_piece(double* buf, size_t count);

    int allreduce(double* buf, size_t count);

    int ensure_col_norm();

    int ensure_p();
    int ensure_pt();

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

    template <typename T>
    int upload_images(const int64_t* h_cp, const int32_t* h_ci, const int64_t* h_rp,
                      const int32_t* h_ri, const double* data_csc, const double* data_csr,
                      const int64_t* perm, const double* y);

    template <typename T>
    int set_data_t(const int64_t* indptr, const int32_t* indices, const double* data,
                   const double* y);

    int data_installed(const double* y);

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
                            const double* y);

    int set_data_csr(int64_t n_, int32_t d_, const int64_t* indptr, const int32_t* indices,
                     const double* data, const double* y);

    int set_data(int64_t n_, int32_t d_, const int64_t* indptr, const int32_t* indices,
                 const double* data, const double* y);

    int set_params(int n_orders_, int k_, int32_t d_, const double* P_, const double* w_,
                   const double* lams_);

    int get_params(double* P_, double* w_);

    int configure(int solver_, int loss_, int reg_, int top_degree_);

    int alloc_work();

    // =============================================================== schedule
    // First-fit colouring of the conflict graph in the visiting order `jf` into order / batch_ptr:
    // on the device when the conflict structure is the handle's own matrix (spfm_colour.hip; the
    // same classes as the host form, tests/test_hip_colour.py), else -- global structure of a
    // sharded run, small problems, more than 4096 colours -- by the host threads.
    bool colour_device = true;
    int colour_device_used = 0;
    bool stream_device = true;   // the 64-column pass's entry stream built on the device
    int stream_device_used = 0;
    int pb_stream_device_used = 0, wide_stream_device_used = 0;  // the pbcd / wide entry streams
    int colour_columns(int64_t rows, const int64_t* cp, const int32_t* ci, bool own,
                       const int32_t* jf, int max_batch);

    int set_schedule(int mode, const int32_t* indices_feature, const int64_t* cf_indptr,
                     const int32_t* cf_indices, int64_t cf_rows, int32_t* order_out,
                     int32_t* n_batches_out);

    int install_schedule();

    int set_schedule_raw(const int32_t* order_in, const int32_t* bptr_in, int32_t nb,
                         const int64_t* cf_indptr, const int32_t* cf_indices, int64_t cf_rows);

    int n_batches() const { return (int)batch_ptr.size() - 1; }

    template <typename T, int M>
    void launch_anova(int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv,
                      const double* Pt_o, double* out);
    template <typename T>
    int anova_dispatch(int M, int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv,
                       const double* Pt_o, double* out);

    template <typename T>
    int output_t(int64_t rows, const int64_t* rp, const int32_t* ri, const T* rv, int degree,
                 int fit_linear, int add_lower, double* out);

    template <typename T>
    int init_pred_t(int degree, int fit_linear, int add_lower);

    int init_pred(int degree, int fit_linear, int add_lower);

    template <typename T>
    int get_y_pred_t(double* out);

    template <typename T>
    int loss_sum_t(double* out);

    template <typename T>
    int predict_csr_t(int64_t rows, const int64_t* indptr, const int32_t* indices,
                      const double* data, int degree, int fit_linear, int add_lower,
                      double* out);

    int epoch_prologue();

    int epoch_epilogue(double* viol);

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

    template <typename T>
    int lin_body(double alpha);

    template <typename T, int LOSS>
    int lin_prb(double alpha);
    template <typename T>
    int lin_prb_loss(double alpha);

    void mark_not_resident(const char* what) {
        pers_failed = true;
        pers_fallbacks += 1;
        pers_reason = std::string(what) + ": its workgroups cannot all be resident on this device";
        if (getenv("SPFM_VERBOSE")) fprintf(stderr, "spfm: fall-back: %s\n", pers_reason.c_str());
    }

    int cd_linear_epoch(double alpha, double* viol);

    template <typename T, int M>
    int pcd_pass_body(int order_idx, double beta, double gamma, double eta);

    // ---------------------------------------------------- persistent row-block pass
    // (several ranks: the persistent passes need the peer-mapped exchange slabs)
    bool prb_usable() const {
        return persistent && !pers_failed && (!dist() || peer_ready) && max_batch_cols <= 64 &&
               nnz < ((int64_t)1 << 31) && n > 0;
    }

    bool resident_ok(const void* fn, int threads, size_t lds, int G);
    static constexpr int kNotResident = 1;  // internal: the launch was not made, nothing changed
    int launch_groups(int G) {  // test hook: a launch that lacks its last workgroup times out
        if (debug_drop > 0 && G > 1) {
            --debug_drop;
            return G - 1;
        }
        return G;
    }
    int snapshot_state(const double* params, size_t count, DevBuf& dst);
    int persistent_aborted(bool* out);
    int recover_from_abort(double* params, size_t count, const DevBuf& src, const char* what);

    template <typename T>
    int ensure_prb();
    template <typename T>
    int build_prb_stream(int nb_);

    PrbArgs prb_args();

    // ---- relaxed runs for schedules of tiny steps (degree-2 pcd pass, one GPU)
    // worth trying: the persistent 64-column pass is in use and the strict steps are narrow
    bool relax_candidate() const {
        return relax_on && prb_usable() && !dist() && n_batches() > 0 &&
               (double)d / (double)n_batches() < 12.0;
    }
    template <typename T>
    int ensure_relax();
    PrbArgs relax_args();

    template <typename T, int M, int LOSS>
    int pcd_pass_prb(int order_idx, double beta, double gamma, double eta);

    template <typename T, int M>
    int pcd_prb_loss(int order_idx, double beta, double gamma, double eta);

    template <typename T>
    int pcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta);

    int peer_clear(size_t off_doubles, size_t n_doubles);
    int host_barrier();

    // ------------------------------------------------------ wide persistent passes
    bool wide_usable() const {
        return persistent && !pers_failed && wide_on && (!dist() || peer_ready) &&
               max_batch_cols > 64 && max_batch_cols <= 512 && nnz < ((int64_t)1 << 31) && n > 0;
    }

    int wide_groups(int ncu, size_t lds_max) const;

    template <typename T>
    int ensure_wide();

    PcdwArgs wide_args();

    template <typename T, int KIND>
    int wide_launch(PcdwArgs& a, PcdwParams& pp, T* Aptr);

    template <typename T>
    int pcd_pass_wide(int order_idx, double beta, double gamma, double eta);

    template <typename T>
    int lin_wide(double alpha);

    // template parameter for a reference degree: -1 (all-subsets) -> 0
    static int kind_of(int degree) { return degree == -1 ? 0 : degree; }
    bool degree_ok(int degree) const {
        return top_degree == -1 ? degree == -1 : (degree >= 2 && degree <= top_degree);
    }

    template <typename T, int M>
    int pcd_precompute_all(int order_idx);
    template <typename T>
    int pcd_precompute_all_dispatch(int M, int order_idx);

    template <typename T>
    int pcd_pass_dispatch(int M, int order_idx, double beta, double gamma, double eta);

    int pcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
                  const int32_t* ic, int n_comp, double* viol);

    template <typename T, int M, int L, int C>
    int pbcd_body_lc(int order_idx, double beta, double gamma, double eta);

    // ------------------------------------------------ persistent pbcd pass (one launch)
    static bool pbprb_degree_ok(int M) { return M == 0 || M == 2 || M == 3 || M == 4; }
    bool pbprb_usable(int M) const {
        return persistent && !pers_failed && pb_persistent && (!dist() || peer_ready) &&
               max_batch_cols <= 64 && nnz < ((int64_t)1 << 31) && n > 0 && k <= 62 &&
               pbprb_degree_ok(M);
    }

    const char* validate_pb_stream(int G, int NG, const std::vector<int32_t>& gsp,
                                   const std::vector<int32_t>& src,
                                   const std::vector<uint8_t>& meta,
                                   const std::vector<uint8_t>& tab) const;

    template <typename T>
    int ensure_pb_stream(int NG);
    template <typename T>
    int ensure_pb_relax(int NG);

    template <typename T, int M, int L>
    int pbcd_prb_l(int order_idx, double beta, double gamma, double eta);

    template <typename T, int M>
    int pbcd_prb_m(int order_idx, double beta, double gamma, double eta);

    template <typename T>
    int pbcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta);

    template <typename T, int M>
    int pbcd_body(int order_idx, double beta, double gamma, double eta);

    template <typename T>
    int pbcd_dispatch(int M, int order_idx, double beta, double gamma, double eta);

    int pbcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
                   double* viol);

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

    int host_epoch_begin(int order_idx, int degree);
    template <typename T>
    int host_pbcd_precompute(int M, int order_idx);
    int host_pass_begin(int s);
    int host_step_check(int b) {
        if (host_order < 0) FAIL(SPFM_ERR_INVALID, "host step: call spfm_host_epoch_begin first");
        if (b < 0 || b >= n_batches()) FAIL(SPFM_ERR_INVALID, "host step: step index out of range");
        return SPFM_OK;
    }

    template <typename T, int M>
    int host_sums_pcd(int b, double* out);
    template <typename T, int M>
    int host_apply_pcd(int b, const double* p_new);
    template <typename T, int M, int L, int C>
    int host_sums_pbcd(int b, double* out);
    template <typename T, int M, int L, int C>
    int host_apply_pbcd(int b, const double* p_new, const double* p_old);
    // dispatch on (storage type, degree[, component lanes]) as the multi-kernel engine does
    template <typename T>
    int host_step_pcd_t(bool sums, int b, double* out, const double* p_new);
    template <typename T>
    int host_step_pbcd_t(bool sums, int b, double* out, const double* p_new, const double* p_old);
    int host_step(bool sums, int b, double* out, const double* p_new, const double* p_old);
    int host_epoch_end(double* viol);

    int configure_psgd(int loss_, int reg_, int top_degree_);

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
                      double power_t, int64_t batch_size, int fit_linear, int64_t* it);

    // row_lo / n_global: several ranks (spfm_psgd_epoch_sharded); one rank: 0 / n
    int psgd_epoch(int degree, double alpha, double beta, double gamma, double eta0, int lr,
                   double power_t, int64_t batch_size, const int32_t* indices_samples,
                   int64_t n_samples, int64_t row_lo, int fit_linear, int64_t* it, double* sum_loss);
    // several ranks: per minibatch of the GLOBAL order this rank's samples (first position in
    // its local sample list, count) and the global batch size
    std::vector<int64_t> sg_lpos;
    std::vector<int32_t> sg_lB, sg_gB;

    // diagnostics that need kernels of one translation unit
    int debug_stream_probe(int64_t* bytes_out);  // spfm_engine_pcd.hip
    int debug_write_probe(int bytes, int64_t* bytes_out);
};

// The device branch counters (spfm_common.hip.h: g_branch_count) exist once per translation
// unit; spfm_debug_branch_counts adds the units' copies up.  Each unit that runs chains defines
// its accessor with SPFM_DEFINE_BRANCH_COUNTS(name): out[0..BR_COUNT) += the unit's counters.
#define SPFM_DEFINE_BRANCH_COUNTS(name)                                                          \
    hipError_t name(unsigned* out, int reset) {                                                  \
        unsigned v[spfm::BR_COUNT];                                                              \
        hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(spfm::g_branch_count), sizeof v);       \
        if (e != hipSuccess) return e;                                                           \
        for (int i = 0; i < spfm::BR_COUNT; ++i) out[i] += v[i];                                 \
        if (reset) {                                                                             \
            const unsigned zero[spfm::BR_COUNT] = {0};                                           \
            e = hipMemcpyToSymbol(HIP_SYMBOL(spfm::g_branch_count), zero, sizeof zero);          \
        }                                                                                        \
        return e;                                                                                \
    }
hipError_t spfm_branch_counts_pcd(unsigned* out, int reset);
hipError_t spfm_branch_counts_prb_f32(unsigned* out, int reset);
hipError_t spfm_branch_counts_prb_f64(unsigned* out, int reset);
hipError_t spfm_branch_counts_wide(unsigned* out, int reset);
hipError_t spfm_branch_counts_pbcd(unsigned* out, int reset);
hipError_t spfm_branch_counts_pbprb_f32(unsigned* out, int reset);
hipError_t spfm_branch_counts_pbprb_f64(unsigned* out, int reset);
