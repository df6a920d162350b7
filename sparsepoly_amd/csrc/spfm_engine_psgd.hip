// spfm_engine_psgd.hip -- minibatch solver (psgd)
#include "spfm_engine.hip.h"

using namespace spfm;

// ================================================================== psgd
// regularizer.init_cache_psgd exists for l1 / l21 / squaredl12 / squaredl21 only
// (reference regularizer/*.py); psgd has no all-subsets variant.
int spfm_engine::configure_psgd(int loss_, int reg_, int top_degree_) {
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

template <typename T, int L>
int spfm_engine::psgd_epoch_tl(int degree, double alpha, double beta, double gamma, double eta0, int lr,
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
    // (several ranks: every minibatch carries a collective -- eager launches)
    if (use_graph && !prof_on && !psgd_force_eager && !dist() && (!mich || psgd_warm)) {
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
    // one rank: the batches of its n samples; several ranks: the batches of the global order,
    // of which this rank holds samples [lpos, lpos + lB) of its local list (psgd_epoch)
    const bool sharded = dist();
    const int64_t nbat_e = sharded ? (int64_t)sg_gB.size() : cdiv(n, batch_size);
    for (int64_t bi = 0; bi < nbat_e; ++bi) {
        const int64_t pos = sharded ? sg_lpos[(size_t)bi] : bi * batch_size;
        const int lB = sharded ? sg_lB[(size_t)bi] : (int)std::min<int64_t>(batch_size, n - pos);
        const int B = sharded ? sg_gB[(size_t)bi] : lB;  // eta / B: the whole minibatch
        prof_begin(0, 0);
        if (lB > 0)
            hipLaunchKernelGGL((psgd_grad_kernel<T, L>), dim3(cdiv(lB, gpb)), dim3(kBlock), 0,
                               stream, sg_samples.as<int32_t>() + pos, lB, rptr.as<int64_t>(),
                               ridx.as<int32_t>(), rval.as<T>(), yy.as<T>(), Pt.as<double>(),
                               w.as<double>(), lams.as<double>(), n_orders, k, d, degree, loss,
                               fit_linear, sg_gradP.as<double>(), sg_gradw.as<double>(),
                               pred_tmp.as<double>() + pos, (const PsgdBatch*)nullptr,
                               (int*)nullptr);
        prof_end(0);
        if (sharded) {  // sum of the ranks' gradients: identical bits on every rank
            int arc = allreduce(sg_gradP.as<double>(), (size_t)n_orders * k * d);
            if (arc) return arc;
            if (fit_linear) {
                arc = allreduce(sg_gradw.as<double>(), (size_t)d);
                if (arc) return arc;
            }
        }
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
int spfm_engine::psgd_epoch(int degree, double alpha, double beta, double gamma, double eta0, int lr,
               double power_t, int64_t batch_size, const int32_t* indices_samples,
               int64_t n_samples, int64_t row_lo, int fit_linear, int64_t* it, double* sum_loss) {
    if (!have_data || !have_params || !configured)
        FAIL(SPFM_ERR_INVALID, "epoch: data, parameters and configuration are required");
    if (solver != SPFM_SOLVER_PSGD) FAIL(SPFM_ERR_INVALID, "engine is not configured for psgd");
    if (degree != top_degree) FAIL(SPFM_ERR_INVALID, "psgd: degree differs from configure()");
    const bool sharded = dist();
    if (!indices_samples || !it || (!sharded && (n_samples != n || row_lo != 0)))
        FAIL(SPFM_ERR_INVALID, "psgd: indices_samples must list every sample once");
    if (sharded && (row_lo < 0 || row_lo + n > n_samples || n_samples > INT32_MAX))
        FAIL(SPFM_ERR_INVALID, "psgd: the handle's rows [row_lo, row_lo + n) lie outside the "
                               "global sample range");
    if (batch_size < 1) FAIL(SPFM_ERR_INVALID, "psgd: batch_size must be >= 1");
    if (lr < 0 || lr > 3) FAIL(SPFM_ERR_INVALID, "psgd: learning_rate is not supported.");
    if (*it < 1) FAIL(SPFM_ERR_INVALID, "psgd: it must be >= 1");
    {
        std::vector<char> seen((size_t)n_samples, 0);
        for (int64_t q = 0; q < n_samples; ++q) {
            const int i = indices_samples[q];
            if (i < 0 || i >= n_samples || seen[(size_t)i])
                FAIL(SPFM_ERR_INVALID, "psgd: indices_samples is not a permutation");
            seen[(size_t)i] = 1;
        }
    }
    if (n_samples == 0) {
        if (sum_loss) *sum_loss = 0.0;
        return SPFM_OK;
    }
    int rc = ensure_pt();
    if (rc) return rc;
    p_valid = false;
    std::vector<int32_t> local;  // several ranks: this rank's samples in visiting order
    if (sharded) {
        const int64_t nbat = cdiv(n_samples, batch_size);
        sg_lpos.assign((size_t)nbat, 0);
        sg_lB.assign((size_t)nbat, 0);
        sg_gB.assign((size_t)nbat, 0);
        local.reserve((size_t)n);
        for (int64_t bi = 0; bi < nbat; ++bi) {
            const int64_t pos = bi * batch_size;
            const int64_t B = std::min<int64_t>(batch_size, n_samples - pos);
            sg_lpos[(size_t)bi] = (int64_t)local.size();
            for (int64_t q = pos; q < pos + B; ++q) {
                const int64_t i = (int64_t)indices_samples[q] - row_lo;
                if (i >= 0 && i < n) local.push_back((int32_t)i);
            }
            sg_lB[(size_t)bi] = (int32_t)((int64_t)local.size() - sg_lpos[(size_t)bi]);
            sg_gB[(size_t)bi] = (int32_t)B;
        }
        if ((int64_t)local.size() != n)
            FAIL(SPFM_ERR_INVALID, "psgd: the global order does not hold this rank's rows once");
        indices_samples = local.data();
    }
    if (n > 0) {
        HIPC(hipMemcpyAsync(sg_samples.p, indices_samples, sizeof(int32_t) * (size_t)n,
                            hipMemcpyHostToDevice, stream));
        HIPC(hipStreamSynchronize(stream));  // caller may reuse indices_samples
    }
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
    if (sharded) {
        rc = allreduce(scalar.as<double>(), 1);
        if (rc) return rc;
    }
    HIPC(hipMemcpyAsync(h_scalar, scalar.p, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPC(hipStreamSynchronize(stream));
    prof_collect();
    if (sum_loss) *sum_loss = h_scalar[0];
    psgd_warm = true;
    return SPFM_OK;
}
