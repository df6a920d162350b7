// spfm_engine_prb.inc.h -- persistent 64-column passes (pcd_prb_kernel, lin_prb_kernel) of the engine for ONE
// storage type SPFM_TU_T; included by spfm_engine_prb_f32.hip / _f64.hip
#include "spfm_engine.hip.h"
#include "spfm_pcd.hip.h"

using namespace spfm;

template <typename T, int LOSS>
int spfm_engine::lin_prb(double alpha) {
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
int spfm_engine::lin_prb_loss(double alpha) {
    switch (loss) {
        case SPFM_LOSS_SQUARED: return lin_prb<T, LOSS_SQUARED>(alpha);
        case SPFM_LOSS_SQUARED_HINGE: return lin_prb<T, LOSS_SQUARED_HINGE>(alpha);
        default: return lin_prb<T, LOSS_LOGISTIC>(alpha);
    }
}

template <typename T>
int spfm_engine::ensure_prb() {
    if (prb_ready) return SPFM_OK;
    int ncu = 0;
    HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (prb_G > ncu) prb_G = ncu;
    if (prb_G < 1) prb_G = 1;
    const int nb_ = n_batches();
    // handles that share a data image (spfm_share_data) share this stream too: the first one to
    // need it builds it under the cache's lock, the others refer to its buffers
    std::unique_lock<std::mutex> cache_lock;
    std::string ckey;
    bool hit = false;
    if (scache) {
        ckey = fkey("prb", {}, {(int64_t)sched_hash, prb_G, prb_long, (int64_t)sizeof(T), nb_});
        cache_lock = std::unique_lock<std::mutex>(scache->mu);
        for (auto& e : scache->prb)
            if (e->key == ckey) {
                prb_sp.share(e->sp);
                prb_erow.share(e->erow);
                prb_eval.share(e->eval);
                prb_lmask.share(e->lmask);
                prb_has_long = e->has_long;
                stream_device_used = 2;  // = taken from a co-tenant
                hit = true;
                break;
            }
    }
    if (!hit) {
        int rc = build_prb_stream<T>(nb_);
        if (rc) return rc;
        if (scache) {
            auto e = std::make_unique<StreamCache::Prb>();
            e->key = ckey;
            e->sp.share(prb_sp);
            e->erow.share(prb_erow);
            e->eval.share(prb_eval);
            e->lmask.share(prb_lmask);
            e->has_long = prb_has_long;
            scache->prb.insert(scache->prb.begin(), std::move(e));
            if (scache->prb.size() > StreamCache::kKeep) scache->prb.pop_back();
        }
    }
    if (cache_lock.owns_lock()) cache_lock.unlock();
    // the handle's own buffers
    HIPC(prb_slab.alloc(sizeof(double) * 2 * ((size_t)prb_G + 1) * 64 * 2));
    HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
    HIPC(prow_old.alloc(sizeof(double) * (size_t)d));
    HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
    HIPC(prb_cn.alloc(sizeof(double) * (size_t)d));
    HIPC(prb_stamps.alloc(sizeof(long long) * 16 * (size_t)prb_G));
    HIPC(hipMemsetAsync(prb_stamps.p, 0, prb_stamps.bytes, stream));
    HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
    HIPC(hipMemsetAsync(prb_slab.p, 0, prb_slab.bytes, stream));
    hipLaunchKernelGGL(gather_sched_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, d,
                       d_desc.as<ColDesc>(), col_norm.as<double>(), prb_cn.as<double>());
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(stream));
    prb_ready = true;
    return SPFM_OK;
}

// the entry stream of the 64-column passes (sp, erow, eval, lmask): on the device
// (spfm_ingest.hip device_rowblock_stream: two binary searches per (column, row block), a scan,
// a fill -- the host builder's sp / src / lmask exactly, tests/test_hip_stream.py), or by the
// host threads (stream_device=0, no room)
template <typename T>
int spfm_engine::build_prb_stream(int nb_) {
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
    HIPC(hipStreamSynchronize(stream));  // the host staging vectors and d_src die here
    return SPFM_OK;
}

template <typename T>
int spfm_engine::ensure_relax() {
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

template <typename T, int M, int LOSS>
int spfm_engine::pcd_pass_prb(int order_idx, double beta, double gamma, double eta) {
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
int spfm_engine::pcd_prb_loss(int order_idx, double beta, double gamma, double eta) {
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
int spfm_engine::pcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
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


#define SPFM_CAT_(a, b) a##b
#define SPFM_CAT(a, b) SPFM_CAT_(a, b)
SPFM_DEFINE_BRANCH_COUNTS(SPFM_CAT(spfm_branch_counts_prb_, SPFM_TU_TAG))

template int spfm_engine::ensure_prb<SPFM_TU_T>();
template int spfm_engine::lin_prb_loss<SPFM_TU_T>(double);
template int spfm_engine::pcd_prb_dispatch<SPFM_TU_T>(int, int, double, double, double);
