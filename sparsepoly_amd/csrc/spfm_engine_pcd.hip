// spfm_engine_pcd.hip -- multi-kernel pcd / cd_linear passes, the pcd and cd_linear epoch drivers, the pcd half of
// the host-stepped epochs
#include "spfm_engine.hip.h"
#include "spfm_pcd.hip.h"
#include "spfm_linear.hip.h"

using namespace spfm;

// ------------------------------------------------------------- cd_linear
template <typename T>
int spfm_engine::lin_body(double alpha) {
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

int spfm_engine::cd_linear_epoch(double alpha, double* viol) {
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
int spfm_engine::pcd_pass_body(int order_idx, double beta, double gamma, double eta) {
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

PrbArgs spfm_engine::prb_args() {
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

PrbArgs spfm_engine::relax_args() {
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

// one precompute pass for all components (A_all[s][i][m-1]); needs P^T
template <typename T, int M>
int spfm_engine::pcd_precompute_all(int order_idx) {
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
int spfm_engine::pcd_precompute_all_dispatch(int M, int order_idx) {
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
int spfm_engine::pcd_pass_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
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

int spfm_engine::pcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
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

int spfm_engine::host_epoch_begin(int order_idx, int degree) {
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

int spfm_engine::host_pass_begin(int s) {  // pcd: the component of the following steps (pcd.py:92)
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

template <typename T, int M>
int spfm_engine::host_sums_pcd(int b, double* out) {
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
int spfm_engine::host_apply_pcd(int b, const double* p_new) {
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

int spfm_engine::host_step(bool sums, int b, double* out, const double* p_new, const double* p_old) {
    int rc = host_step_check(b);
    if (rc) return rc;
    if ((sums && !out) || (!sums && !p_new)) FAIL(SPFM_ERR_INVALID, "host step: NULL buffer");
    if (solver == SPFM_SOLVER_PCD)
        return dtype == SPFM_F32 ? host_step_pcd_t<float>(sums, b, out, p_new)
                                 : host_step_pcd_t<double>(sums, b, out, p_new);
    return dtype == SPFM_F32 ? host_step_pbcd_t<float>(sums, b, out, p_new, p_old)
                             : host_step_pbcd_t<double>(sums, b, out, p_new, p_old);
}

template <typename T>
int spfm_engine::host_step_pcd_t(bool sums, int b, double* out, const double* p_new) {
#define SPFM_HPC(MM) return sums ? host_sums_pcd<T, MM>(b, out) : host_apply_pcd<T, MM>(b, p_new)
    switch (kind_of(host_degree)) {
        case 0: SPFM_HPC(0);
        case 2: SPFM_HPC(2);
        case 3: SPFM_HPC(3);
        case 4: SPFM_HPC(4);
        case 5: SPFM_HPC(5);
        case 6: SPFM_HPC(6);
    }
#undef SPFM_HPC
    FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
}

int spfm_engine::host_epoch_end(double* viol) {
    if (host_order < 0) FAIL(SPFM_ERR_INVALID, "host_epoch_end: no host-stepped epoch open");
    host_order = -1;
    if (solver == SPFM_SOLVER_PCD) pt_valid = false;
    return epoch_epilogue(viol);
}


// tools/fetch_calibration.py: the entry stream of the 64-column pass read exactly as
// pcd_prb_kernel's worker threads read it, and nothing else
int spfm_engine::debug_stream_probe(int64_t* bytes_out) {
    if (!have_schedule || !prb_usable())
        FAIL(SPFM_ERR_INVALID, "stream probe: needs a schedule the 64-column persistent pass can run");
    int rc = dtype == SPFM_F32 ? ensure_prb<float>() : ensure_prb<double>();
    if (rc) return rc;
    DevBuf sink;
    if (sink.alloc(sizeof(double) * (size_t)prb_G * kPrbThreads) != hipSuccess)
        FAIL(SPFM_ERR_RUNTIME, "stream probe: allocation failed");
    const PrbArgs a = prb_args();
    if (dtype == SPFM_F32)
        hipLaunchKernelGGL((prb_stream_probe_kernel<float>), dim3(prb_G), dim3(kPrbThreads), 0,
                           stream, a, prb_eval.as<float>(), sink.as<double>());
    else
        hipLaunchKernelGGL((prb_stream_probe_kernel<double>), dim3(prb_G), dim3(kPrbThreads), 0,
                           stream, a, prb_eval.as<double>(), sink.as<double>());
    if (hipStreamSynchronize(stream) != hipSuccess)
        FAIL(SPFM_ERR_RUNTIME, "stream probe kernel failed");
    // requested bytes: per entry a 4-byte row id and a value; per (workgroup, step) the slot
    // bounds of its columns (+1) as 4-byte words
    if (bytes_out)
        *bytes_out = nnz * (int64_t)(4 + tsize()) + (int64_t)prb_G * ((int64_t)d + n_batches()) * 4;
    return SPFM_OK;
}

// tools/write_calibration.py: one `bytes`-wide store per entry at the entry's row, the scatter
// pattern of the passes that keep their rows in global memory, and nothing else written
int spfm_engine::debug_write_probe(int bytes, int64_t* bytes_out) {
    if (!have_schedule || !prb_usable())
        FAIL(SPFM_ERR_INVALID, "write probe: needs a schedule the 64-column persistent pass can run");
    if (bytes != 4 && bytes != 8 && bytes != 16) FAIL(SPFM_ERR_INVALID, "write probe: 4, 8 or 16 bytes");
    int rc = dtype == SPFM_F32 ? ensure_prb<float>() : ensure_prb<double>();
    if (rc) return rc;
    DevBuf recs;
    if (recs.alloc((size_t)16 * (size_t)(n > 0 ? n : 1)) != hipSuccess)
        FAIL(SPFM_ERR_RUNTIME, "write probe: allocation failed");
    const PrbArgs a = prb_args();
    if (bytes == 4)
        hipLaunchKernelGGL((prb_write_probe_kernel<4>), dim3(prb_G), dim3(kPrbThreads), 0, stream, a,
                           recs.as<float>());
    else if (bytes == 8)
        hipLaunchKernelGGL((prb_write_probe_kernel<8>), dim3(prb_G), dim3(kPrbThreads), 0, stream, a,
                           recs.as<float>());
    else
        hipLaunchKernelGGL((prb_write_probe_kernel<16>), dim3(prb_G), dim3(kPrbThreads), 0, stream,
                           a, recs.as<float>());
    if (hipStreamSynchronize(stream) != hipSuccess) FAIL(SPFM_ERR_RUNTIME, "write probe kernel failed");
    if (bytes_out) *bytes_out = nnz * (int64_t)bytes;
    return SPFM_OK;
}

SPFM_DEFINE_BRANCH_COUNTS(spfm_branch_counts_pcd)

// used by the pbcd unit's host-stepped epochs
template int spfm_engine::pcd_precompute_all_dispatch<float>(int, int);
template int spfm_engine::pcd_precompute_all_dispatch<double>(int, int);
