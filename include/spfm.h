/*
 * spfm.h -- C ABI of the MI355X (gfx950) sparse factorization-machine
 * proximal coordinate-descent core (libspfm_hip.so).
 *
 * This is the drop-in boundary for the hot path of neonnnnn/sparsepoly:
 * every entry point replaces one interpreter->Numba call (or one NumPy block)
 * of the reference's epoch drivers.  Citations are into the reference tree
 * (sparsepoly/...).  Plain pointers and sizes only; no C++/torch types; no
 * exception crosses the boundary.  All host buffers stay owned by the caller
 * and are copied during the call.
 *
 * Conventions
 *   return 0 = ok; <0 = error (spfm_last_error() gives the text)
 *     SPFM_ERR_INVALID      -> the reference raises ValueError for this input
 *     SPFM_ERR_RUNTIME      -> HIP/RCCL failure
 *     SPFM_ERR_UNSUPPORTED  -> valid for the reference, outside this library
 *   A handle is not thread-safe; distinct handles are independent.  Each handle
 *   owns one HIP stream; calls return after the stream has drained unless noted.
 *   Host arrays are float64 / int32 / int64 as in the reference
 *   (dataset.py:60-66, base.py:41-49); device storage precision is the handle's
 *   dtype (values, A caches, y_pred in f32 or f64; reductions, prox and
 *   parameters always f64).
 */
#ifndef SPFM_H
#define SPFM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spfm_engine* spfm_handle;

#define SPFM_OK 0
#define SPFM_ERR_INVALID (-1)
#define SPFM_ERR_RUNTIME (-2)
#define SPFM_ERR_UNSUPPORTED (-3)

/* storage precision of X values, A caches, y_pred, y on the device */
#define SPFM_F32 0
#define SPFM_F64 1

/* loss.py:74-80 registries */
#define SPFM_LOSS_SQUARED 0
#define SPFM_LOSS_SQUARED_HINGE 1
#define SPFM_LOSS_LOGISTIC 2

/* regularizer/__init__.py:8-15 */
#define SPFM_REG_L1 0
#define SPFM_REG_L21 1
#define SPFM_REG_SQUAREDL12 2
#define SPFM_REG_SQUAREDL21 3
#define SPFM_REG_OMEGATI 4
#define SPFM_REG_OMEGACS 5

#define SPFM_SOLVER_PCD 0
#define SPFM_SOLVER_PBCD 1
#define SPFM_SOLVER_PSGD 2

/* coordinate schedules (spfm_set_schedule) */
#define SPFM_SCHED_EXACT 0   /* keep the given order; batch = maximal run of row-disjoint columns */
#define SPFM_SCHED_COLORED 1 /* first-fit colouring of the column conflict graph; order is permuted */

#define SPFM_MAX_DEGREE 6

/* -- lifetime ------------------------------------------------------------ */
int spfm_create(spfm_handle* out, int device_id, int dtype);
void spfm_destroy(spfm_handle h);
/* h may be NULL: returns the calling thread's last creation error */
const char* spfm_last_error(spfm_handle h);
/* library / device identification, e.g. "gfx950:sramecc+:xnack-" */
int spfm_device_name(spfm_handle h, char* out, int cap);
/* 12 hex digits: hash of the library's sources at build time (csrc/build.sh).  No handle and
 * no device needed.  Measurements kept under profiles/ record it, so a counter file is only
 * ever quoted for the code it was collected with. */
const char* spfm_build_tag(void);

/* -- data ------------------------------------------------------------------
 * Replaces get_dataset(X, order="fortran") (dataset.py:119-123, CSCDataset
 * :94-116) plus col_norm_sq = row_norms(X.T, squared=True)
 * (sparse_factorization_machines.py:406-409).  CSC of the LOCAL row shard:
 * indptr[d+1] (int64), indices[nnz] (int32 row ids in [0,n)), data[nnz], y[n].
 * The library also builds the CSR image it needs for the row-oriented passes. */
int spfm_set_data_csc(spfm_handle h, int64_t n, int32_t d, const int64_t* indptr,
                      const int32_t* indices, const double* data, const double* y);

/* The same from a CSR matrix (what scipy hands the estimators most of the time): indptr[n+1]
 * (int64), indices[nnz] (int32 column ids, sorted and duplicate-free inside each row), data,
 * y[n].  Replaces X.tocsc() of get_dataset as well: the CSC image is built inside, by host
 * threads (SPFM_THREADS, default min(cores, 16)). */
int spfm_set_data_csr(spfm_handle h, int64_t n, int32_t d, const int64_t* indptr,
                      const int32_t* indices, const double* data, const double* y);

/* Several handles on ONE training matrix (round 4): the fits of a regularisation path or a
 * parameter grid -- the use-case the reference serves with warm_start chains
 * (sparse_factorization_machines.py:380-391) -- and one-vs-rest targets (base.py:130-136).  `dst`
 * refers to the device image `src` holds (CSC, CSR, column norms) instead of uploading and
 * transposing its own copy, and the two share the entry streams of the persistent passes either
 * of them builds later.  Same device and storage type, no communicator.  `y` (n doubles): dst's
 * own targets, or NULL = src's.  The image is freed with its last holder; a later
 * spfm_set_data_* on either handle detaches that handle only. */
int spfm_share_data(spfm_handle dst, spfm_handle src, const double* y);

/* -- parameters -------------------------------------------------------------
 * P is (n_orders, k, d) row-major as self.P_ (sparse_factorization_machines.py
 * :383-389), w (d), lams (k, each +-1: :403-404).  d must equal the data's
 * n_features when data is present; a handle without data (predict only) takes d
 * from here.  get copies the live device state back (pcd callbacks see live P_,
 * :227-228). */
int spfm_set_params(spfm_handle h, int n_orders, int k, int32_t d, const double* P,
                    const double* w, const double* lams);
int spfm_get_params(spfm_handle h, double* P, double* w);

/* -- configuration ----------------------------------------------------------
 * loss (base.py:18-25), regularizer (base.py:27-34) and the solver whose cache
 * protocol is initialised: regularizer.init_cache_pcd / init_cache_pbcd(degree,
 * d, k) (sparse_factorization_machines.py:194,282).  Errors as the reference:
 * SquaredL12 degree>2 / SquaredL21 degree!=2 -> SPFM_ERR_INVALID
 * (squaredl12.py:25-26, squaredl21.py:28-29); solver/regularizer pairs the
 * reference cannot run (README.md:28-32) -> SPFM_ERR_INVALID. */
int spfm_configure(spfm_handle h, int solver, int loss, int regularizer, int top_degree);

/* -- initial prediction ----------------------------------------------------
 * y_pred = _get_output(X) (sparse_factorization_machines.py:437-451, kernels.py
 * :71-115,140-153): ANOVA kernel of order `degree` on P[0], + X.w if fit_linear,
 * + the order-2 term on P[1] if add_lower_deg2 (fit_lower='explicit', degree 3). */
int spfm_init_pred(spfm_handle h, int degree, int fit_linear, int add_lower_deg2);
int spfm_get_y_pred(spfm_handle h, double* out);
/* sum_i loss(y_pred_i, y_i) (loss.py:20-21,34-42,61-65) of the local shard */
int spfm_loss_sum(spfm_handle h, double* out);

/* Same computation on a caller-supplied CSR matrix (predict(),
 * sparse_factorization_machines.py:453-458).  indptr[n+1] int64, indices int32
 * column ids in [0,d). */
int spfm_predict_csr(spfm_handle h, int64_t n, const int64_t* indptr, const int32_t* indices,
                     const double* data, int degree, int fit_linear, int add_lower_deg2,
                     double* out);

/* -- coordinate schedule ----------------------------------------------------
 * The reference's epoch functions take the visiting order as an argument
 * (indices_feature: pcd.py:86-87,97; pbcd.py:99,110; cd_linear.py:8,10).
 * This call fixes the order for the following epochs and partitions it into
 * batches of columns that share no row, which the device processes as one
 * dependent step (results equal the sequential sweep in `order_out`).
 *   conflict_indptr/indices: CSC structure used for the disjointness test; pass
 *     NULL to use the local data (single process).  Multi-GPU: pass the GLOBAL
 *     structure so that all ranks derive the identical schedule.
 *   order_out[d]: the order actually used (== indices_feature for EXACT).
 *   n_batches_out: number of dependent steps per sweep. */
int spfm_set_schedule(spfm_handle h, int mode, const int32_t* indices_feature,
                      const int64_t* conflict_indptr, const int32_t* conflict_indices,
                      int64_t conflict_n_rows, int32_t* order_out, int32_t* n_batches_out);

/* Install a schedule computed earlier (spfm_schedule_build / a cached product of a
 * previous fit on the same data): order[d] and batch_ptr[n_batches+1].  The library
 * re-checks that `order` is a permutation and that every batch is row-disjoint on the
 * structure it would have used to build it (conflict_* as in spfm_set_schedule, NULL =
 * local data), because a batch that shares a row would race; SPFM_ERR_INVALID otherwise. */
int spfm_set_schedule_raw(spfm_handle h, const int32_t* order, const int32_t* batch_ptr,
                          int32_t n_batches, const int64_t* conflict_indptr,
                          const int32_t* conflict_indices, int64_t conflict_n_rows);

/* Read back the installed schedule: order_out[d], batch_ptr_out[n_batches+1] (either may
 * be NULL); n_batches_out receives the number of batches. */
int spfm_get_schedule(spfm_handle h, int32_t* order_out, int32_t* batch_ptr_out,
                      int32_t* n_batches_out);

/* Host-only form of the batch construction (no handle, no device): fills
 * order_out[d] and batch_ptr_out[<= d+1] (batch b = order_out[batch_ptr[b] ..
 * batch_ptr[b+1])) and returns the number of batches in n_batches_out.
 * max_batch <= 0 selects the library default (4096 columns per step). */
int spfm_schedule_build(int mode, int64_t n_rows, int32_t d, const int64_t* indptr,
                        const int32_t* indices, const int32_t* indices_feature, int max_batch,
                        int32_t* order_out, int32_t* batch_ptr_out, int32_t* n_batches_out);

/* -- epochs ------------------------------------------------------------------
 * One call = one reference epoch function call.  viol receives sum_viol. */

/* cd_linear._cd_linear_epoch (optimizer/cd_linear.py:8-33) */
int spfm_cd_linear_epoch(spfm_handle h, double alpha, double* viol);

/* pcd.pcd_epoch (optimizer/pcd.py:71-137) on P[order_idx] with `degree`;
 * indices_component[n_comp] as pcd.py:86,92. */
int spfm_pcd_epoch(spfm_handle h, int order_idx, int degree, double beta, double gamma,
                   double eta, const int32_t* indices_component, int n_comp, double* viol);

/* pbcd.pbcd_epoch (optimizer/pbcd.py:82-148) on P[order_idx] */
int spfm_pbcd_epoch(spfm_handle h, int order_idx, int degree, double beta, double gamma,
                    double eta, double* viol);

/* -- host-stepped epochs: user-defined regularizer objects ---------------------------------
 * The reference's regularizers are duck-typed plug-ins (regularizer/__init__.py:8-15,
 * base.py:27-34): any object with init_cache_* / compute_cache_* / prox_cd | prox_bcd /
 * update_cache_* can be registered.  The six built-ins run inside the device chains; for any
 * other object the epoch is stepped from the host.  Between spfm_host_epoch_begin and
 * spfm_host_epoch_end (one reference epoch call; the schedule's dependent steps b = 0 ..
 * n_batches-1 in order, for pcd once per component after spfm_host_pass_begin(s)):
 *   spfm_host_step_sums   the step's column sums: pcd sums_out[ncols][2] = (sum dloss dA,
 *                         sum dA^2) (pcd.py:54-59); pbcd sums_out[ncols][k+1] = grad[0..k),
 *                         sum_s inv_step_sizes[s] (pbcd.py:60-70); all-reduced over the ranks
 *   (caller)              pcd.py:61-68 / pbcd.py:68-79 with the object's prox and cache hooks,
 *                         column by column in the step's visiting order
 *   spfm_host_step_apply  p_new[ncols] (pcd) / p_new, p_old [ncols][k] (pbcd): parameters
 *                         written, rows scatter-updated (pcd.py:119-133 / pbcd.py:135-146)
 * Two host round trips per dependent step: the plug-in surface honoured, not accelerated.
 * sparsepoly_amd.engine.HipEngine.{pcd,pbcd}_epoch_host drive it; the estimators take this path
 * for any registered regularizer that is not one of the six built-in classes. */
int spfm_host_epoch_begin(spfm_handle h, int order_idx, int degree);
int spfm_host_pass_begin(spfm_handle h, int component);
int spfm_host_step_sums(spfm_handle h, int step, double* sums_out);
int spfm_host_step_apply(spfm_handle h, int step, const double* p_new, const double* p_old);
int spfm_host_epoch_end(spfm_handle h, double* viol);

/* psgd.psgd_epoch (optimizer/psgd.py:125-199): one pass over indices_samples (a
 * permutation of 0..n_samples-1; sparse_factorization_machines.py:98,124-125) in
 * minibatches of batch_size rows (the last one may be shorter, psgd.py:177).  Updates
 * every order of P (order o has degree `degree - o`, psgd.py:84-85,120-122) and w.
 *   learning_rate  0 constant | 1 optimal | 2 pegasos | 3 invscaling (psgd.py:9-22,
 *                  sparse_factorization_machines.py:17 LEARNING_RATE)
 *   it             in/out: the reference's self.it_ (starts at 1; +1 per parameter update)
 *   sum_loss       out: sum over samples of loss(y_pred_i, y_i) at visiting time
 * Requires spfm_configure(h, SPFM_SOLVER_PSGD, loss, reg, degree) with reg in {l1, l21,
 * squaredl12, squaredl21}; no schedule is needed.  The training-time y_pred vector
 * (spfm_get_y_pred / spfm_loss_sum) is not maintained by this solver.  One rank; several ranks:
 * spfm_psgd_epoch_sharded. */
int spfm_psgd_epoch(spfm_handle h, int degree, double alpha, double beta, double gamma,
                    double eta0, int learning_rate, double power_t, int64_t batch_size,
                    const int32_t* indices_samples, int64_t n_samples, int fit_linear,
                    int64_t* it, double* sum_loss);
/* The same epoch with the rows sharded over the ranks of the handle's communicator (data-parallel
 * minibatch SGD; the reference has no distributed code -- this is what its loop implies:
 * _update_grads (psgd.py:60-91) sums over the samples of a minibatch, _update_params
 * (psgd.py:94-122) is a function of those sums alone).  The handle holds the rows
 * [row_lo, row_lo + n_local) of a problem of n_global rows; indices_samples is the GLOBAL visiting
 * order (a permutation of 0..n_global-1, identical on every rank).  Every rank forms the gradient
 * of its own samples of a minibatch, the gradients are all-reduced (sum, f64: one collective of
 * n_orders*k*d + d doubles per minibatch), and every rank applies the identical update with the
 * GLOBAL batch size in eta/B -- parameters stay replicated.  sum_loss is the global sum.  Without
 * a communicator (one rank) it is spfm_psgd_epoch. */
int spfm_psgd_epoch_sharded(spfm_handle h, int degree, double alpha, double beta, double gamma,
                            double eta0, int learning_rate, double power_t, int64_t batch_size,
                            const int32_t* indices_samples, int64_t n_global, int64_t row_lo,
                            int fit_linear, int64_t* it, double* sum_loss);

/* -- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------------
 * Rows are sharded; the column partial sums of every step are all-reduced
 * (sum, f64) so that every rank applies the identical prox.  id is an opaque
 * 128-byte RCCL unique id created on rank 0 and shipped by the caller. */
int spfm_comm_unique_id(char* id128);
int spfm_comm_init(spfm_handle h, const char* id128, int n_ranks, int rank);
/* The same sharded protocol with the all-reduce carried through a POSIX shared-memory
 * segment on the host (name as for shm_open, e.g. "/spfm_test_123"; created zero-filled by
 * whichever rank arrives first, the caller unlinks it).  For ranks that share ONE GPU, where
 * RCCL refuses to form a communicator: lets the multi-GPU code path (row shards, per-step
 * all-reduce, replicated chain) be run and checked on a single-GPU machine.  A test and
 * bring-up facility, orders of magnitude slower than RCCL over xGMI. */
int spfm_comm_init_shm(spfm_handle h, const char* shm_name, int n_ranks, int rank);

/* In-kernel cross-GPU exchange for the persistent passes (replaces the per-step collective;
 * SURVEY.md section 5 "one-hop direct-write exchange").  Every rank allocates one exchange
 * slab (spfm_peer_alloc returns its 64-byte hipIpc handle), the caller ships the handles to all
 * ranks (any channel: torch.distributed, MPI, a file) and every rank maps its peers' slabs
 * (spfm_peer_connect; handles = n_ranks x 64 bytes in rank order, n_ranks <= 8).  Requires a
 * communicator with the same ranks (spfm_comm_init / _shm): it carries the one-off set-up
 * reductions and the barrier between launches.  After connecting, set the schedule (again):
 * the persistent passes cap a step at 64 columns.  Works across the GPUs of one node (xGMI)
 * and -- for tests -- between processes that share one GPU. */
int spfm_peer_alloc(spfm_handle h, char* handle64);
/* Maps the peers' slabs and runs a handshake kernel: every rank stores one word into every
 * slab and polls its own until all ranks' words arrived (10 s bound) -- what the persistent
 * passes rely on (a system-scope store into a peer-mapped slab reaches a kernel that is already
 * polling).  All ranks call it together.  The slab is fine-grained device memory; when that
 * cannot be allocated, or the handshake fails, the call returns SPFM_ERR_RUNTIME and the caller
 * keeps the per-step collective (sparsepoly_amd.distributed.connect_peers does, on all ranks). */
int spfm_peer_connect(spfm_handle h, int n_ranks, int rank, const char* handles);

/* -- instrumentation ----------------------------------------------------------
 * Device time (ms, HIP events on the handle's stream) and launch count of the
 * dominant kernel family since the last reset: which = 0 pcd gather (persistent
 * engine: the whole-pass kernel pcd_prb_kernel; multi-kernel engine: pcd_grad_kernel),
 * 1 pcd chain+scatter (multi-kernel engine), 2 pbcd gradient, 3 pbcd sync,
 * 4 cd_linear (lin_prb_kernel or the per-step kernels).  nnz = column entries the
 * timed launches processed.  Timing is only collected when enabled (one event pair
 * per launch; the multi-kernel engine then launches eagerly instead of replaying). */
int spfm_profile_enable(spfm_handle h, int on);
int spfm_profile_get(spfm_handle h, int which, double* ms, int64_t* launches, int64_t* nnz);
int spfm_profile_reset(spfm_handle h);
/* use hipGraph replay for the per-pass launch sequences (default on) */
int spfm_set_use_graph(spfm_handle h, int on);
/* engine tunables, by name: "use_graph" (0/1), "fuse_chain" (0/1: fused chain+sync
 * kernel for steps of <= 64 columns), "max_batch" (columns per dependent step,
 * applies to the next spfm_set_schedule), "persistent" (0/1: one persistent launch
 * per pcd component pass, single GPU; caps steps at 64 columns), "prb_groups"
 * (workgroups of the persistent pass), "prb_long" (entries of one column in one row
 * block above which the whole workgroup, not 4 lanes, processes it), "prb_lds" (0/1, default 1: keep each workgroup's row block -- A and the
 * residual / prediction -- in LDS for the whole pass when it fits: f32 storage, one cache
 * value per row, squared loss or +-1 targets), "prb_stamps" (diagnostic phase timers),
 * "psgd_eager" (1: launch every psgd minibatch eagerly instead of replaying runs of 32 from
 * a hipGraph), "psgd_graph_sweeps" (support-search sweeps recorded per minibatch for the
 * squared-norm prox, default 4; an epoch in which that was not enough is redone eagerly from
 * a snapshot -- spfm_get_option("psgd_redone") counts those).  Round 2: "pbcd_persistent"
 * (0/1: the persistent pbcd pass), "pbprb_groups" (its workgroups, default 256), "wide" (0/1:
 * the wide passes for degree-2 pcd / cd_linear, steps of up to 512 columns), "pcdw_groups"
 * (their workgroups; default 0 = about 160 entries per workgroup and step, at most one per CU),
 * "wide_min_cols" (mean colour-class width below which the schedule is coloured again with 64 columns per class and the 64-column passes run, default
 * 110; 0 = always wide), "peer_exchange" (only 0 can be set: give the in-kernel cross-GPU
 * exchange up after spfm_peer_connect and use the per-step collective), diagnostics
 * "pbprb_stamps", "pcdw_stamps", "probe_xcd" / "probe_lds" (spfm_debug_exchange_cost on one
 * XCD).  They change how a sweep is cut into launches and in which order partial sums are
 * added; "wide" / "wide_min_cols" / "max_batch" also the coloured schedule built next (never an
 * order the caller passed as 'exact'). */
int spfm_set_option(spfm_handle h, const char* key, int value);
/* Round 3: "pbprb_owners" (dedicated owner workgroups of the persistent pbcd pass: removed in
 * round 4, only 0 is accepted), "relax" (0/1, default 1: a schedule of tiny steps --
 * the reference order, fewer than 12 columns per step on average -- is run by the degree-2 pcd
 * pass as merged steps of ~20 consecutive columns whose shared rows the chains replay in order;
 * same result as the sequential sweep), "prb_pack" (0/1, default 1: degree-3 passes with their
 * rows in global memory work on packed 16-byte row records), "persistent_failed" (0: try the
 * persistent passes again after a recorded fall-back), test hooks "debug_spin_max" (polls of one
 * in-kernel wait before a persistent pass gives up, default 2^21) and "debug_drop_group" (the next
 * N persistent launches lack their last workgroup, i.e. time out), "ingest_device" (0/1, default 1:
 * spfm_set_data_csr transposes on the device -- one stable radix sort of the entries by column id;
 * 0, or no room for the sort's scratch: host threads), "colour_device" (0/1, default 1: the
 * first-fit colouring of spfm_set_schedule(SPFM_SCHED_COLORED) runs on the device when the
 * conflict structure is the handle's own matrix -- the same order and batch boundaries as the
 * host form; "colour_device_used" tells), "stream_device" (0/1, default 1: the entry stream of
 * the 64-column passes is built on the device, entry for entry the host builder's), "co_tenants" (1..64, default 1: the number
 * of handles of this process whose persistent passes run at the same time on this device --
 * independent fits, one handle and one host thread each; the handle then sizes its passes to
 * 1/co_tenants of the CUs and its residency check to the shared device.  Set it before the
 * schedule.  Handles are independent objects: calls on DIFFERENT handles may be made from
 * different threads at the same time, calls on one handle must not overlap).
 *
 * The library never issues work on the null stream (every handle owns a non-blocking stream):
 * concurrent handles do not serialise on each other or on the host program's default stream.
 * The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a host program
 * that wants more than three concurrent handles next to its own streams should raise it (the
 * Python package defaults it to 8).
 *
 * Failure semantics of the persistent passes (all-or-nothing epochs, as the reference's epoch
 * functions): if a pass cannot run to its end -- its workgroups are not all resident, a peer GPU
 * does not answer -- the epoch's parameters and regularizer state are restored from a snapshot
 * taken before the launches, y_pred is recomputed from them (arguments of the last
 * spfm_init_pred), the epoch is redone on the multi-kernel engine and the handle keeps using that
 * engine; with several ranks the decision is agreed on through the communicator.  The call
 * still returns SPFM_OK; spfm_get_option("persistent_fallbacks") counts the events. */
/* read back a tunable, or the derived "persistent_active" (1 if the next pcd epoch
 * will use the persistent pass: option on, single GPU, steps of <= 64 columns) and
 * "prb_lds_active" (what the last pcd pass used: 0 global rows, 1 LDS residual form,
 * 2 LDS prediction + label sign), "pbprb_active", "wide_active", "wide_lds_active",
 * "prb_pack_active", "relax_steps" (merged steps per degree-2 pcd sweep, 0 = strict steps),
 * "persistent_fallbacks", "persistent_failed", "n_ranks" (ranks of the attached communicator),
 * "peer_ready" (in-kernel cross-GPU exchange connected and verified), "ingest_device_used"
 * (1 if the last spfm_set_data_csr transposed on the device), "co_tenants" */
int spfm_get_option(spfm_handle h, const char* key, int* value);

/* diagnostic ("prb_stamps" option): accumulated shader cycles per phase of the last
 * persistent pass, 16 values per workgroup (8 control-wave, 8 worker-wave phases);
 * returns the number of values written. */
int spfm_debug_prb_stamps(spfm_handle h, long long* out, int cap);
/* diagnostic: latency of one hand-off "agent-scope store by one workgroup becomes visible to
 * an agent-scope load of another" (the primitive of the persistent pass's exchange), measured
 * by ping-pong between workgroup 0 and workgroup `partner` of one launch (workgroups are
 * dealt round-robin to the 8 XCDs: partner 1 = other XCD, partner 8 = same XCD).
 * xcc_ids[2] receives the XCC_ID register of the two players. */
int spfm_debug_hop_latency(spfm_handle h, int partner, int rounds, double* ns_per_hop,
                           int* xcc_ids);
/* diagnostic: cost of the bare per-step exchange of the persistent pass -- `groups`
 * workgroups publish 64 granule pairs each and sweep all the others', `rounds` times, with
 * no other work (readers_mod > 1: only every readers_mod-th workgroup sweeps and passes its
 * totals on to the rest).  ncols = slots actually read. */
int spfm_debug_exchange_cost(spfm_handle h, int groups, int ncols, int readers_mod, int rounds,
                             double* ns_per_round);

/* diagnostic: how often the device chains took the reference's "numerical error" branches
 * since the last reset -- out[0] omegati.py:97-98 (clip), out[1] omegacs.py:90-96, out[2]
 * omegacs.py:75-76, out[3] squaredl21.py:48-49 (out[4..7] reserved).  reset != 0 clears the
 * counters after reading.  Counters are per process (all handles of the device). */
/* Diagnostic for the counter calibration (profiles/r03_fetch_calibration.txt): one launch that
 * reads the persistent pass's entry stream (slot bounds, rows, values of every step, with the
 * pass's own access pattern) and nothing else; bytes_out receives the bytes it requested.  Run
 * under `rocprofv3 --pmc FETCH_SIZE` to see what the counter reports for that known quantity. */
int spfm_debug_stream_probe(spfm_handle h, int64_t* bytes_out);
/* Diagnostic (tools/write_calibration.py): one launch that stores ONE record of
 * bytes_per_record (4, 8 or 16) bytes per matrix entry at the entry's row -- the scatter pattern of
 * the persistent passes that keep their rows in global memory -- and writes nothing else;
 * *bytes_out = the bytes requested (nnz * bytes_per_record).  Calibrates WRITE_SIZE. */
int spfm_debug_write_probe(spfm_handle h, int bytes_per_record, int64_t* bytes_out);
int spfm_debug_branch_counts(spfm_handle h, unsigned* out8, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SPFM_H */
