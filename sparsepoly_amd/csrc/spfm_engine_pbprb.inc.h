// spfm_engine_pbprb.inc.h -- persistent pbcd pass (pbcd_prb_kernel) of the engine for ONE storage type
// SPFM_TU_T; included by spfm_engine_pbprb_f32.hip / _f64.hip
#include "spfm_engine.hip.h"
#include "spfm_pbprb.hip.h"

using namespace spfm;

// entry stream (workgroup, step, slot, row) for G row blocks; shares the pcd pass's when
// the workgroup counts agree
template <typename T>
int spfm_engine::ensure_pb_stream(int NG) {
    int ncu = 0;
    HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    const int G = std::max(1, std::min(pbprb_G, ncu));
    if (pb_stream_ready && pb_stream_G == G && pb_stream_NG == NG) return SPFM_OK;
    // handles that share a data image share this stream too (see ensure_prb)
    std::unique_lock<std::mutex> cache_lock;
    std::string ckey;
    if (scache) {
        ckey = fkey("pb", {}, {(int64_t)sched_hash, G, NG, (int64_t)pb_balance, (int64_t)sizeof(T),
                               n_batches()});
        cache_lock = std::unique_lock<std::mutex>(scache->mu);
        for (auto& e : scache->pb)
            if (e->key == ckey) {
                pb_sp.share(e->sp);
                pb_erow.share(e->erow);
                pb_eval.share(e->eval);
                pb_meta.share(e->meta);
                pb_tab.share(e->tab);
                cache_lock.unlock();
                HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
                HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
                HIPC(pb_stamps.alloc(sizeof(long long) * 16 * (size_t)G));
                HIPC(hipMemsetAsync(pb_stamps.p, 0, pb_stamps.bytes, stream));
                HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
                HIPC(hipStreamSynchronize(stream));
                pb_stream_G = G;
                pb_stream_NG = NG;
                pb_stream_ready = true;
                return SPFM_OK;
            }
    }
    const int nb_ = n_batches();
    const size_t ngsp = (size_t)G * nb_ * ((size_t)NG + 1) + 1;
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
    DevBuf d_src;
    HIPC(d_src.alloc(sizeof(int32_t) * nz));
    HIPC(pb_sp.alloc(sizeof(int32_t) * ngsp));
    HIPC(pb_erow.alloc(sizeof(int32_t) * nz + 256));
    HIPC(pb_eval.alloc(sizeof(T) * nz + 256));
    HIPC(pb_meta.alloc(nz + 256));
    HIPC(pb_tab.alloc((size_t)G * (size_t)std::max(nb_, 1) * 64 + 16));
    HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
    HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
    HIPC(pb_stamps.alloc(sizeof(long long) * 16 * (size_t)G));
    HIPC(hipMemsetAsync(pb_stamps.p, 0, pb_stamps.bytes, stream));
    HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
    // the stream: on the device (spfm_ingest.hip device_pb_stream: the host builder's tables
    // exactly, tests/test_hip_stream.py), or by the host threads (small problems -- which are
    // audited --, stream_device=0, no room for the scratch)
    pb_stream_device_used = 0;
    if (stream_device && nnz >= (1 << 20) && !getenv("SPFM_VALIDATE")) {
        const hipError_t e = device_pb_stream(
            n, d, nnz, G, nb_, NG, pb_balance ? 1 : 0, d_order.as<int32_t>(), d_bptr.as<int32_t>(),
            cptr.as<int64_t>(), cidx.as<int32_t>(), rptr.as<int64_t>(), ridx.as<int32_t>(),
            pb_sp.as<int32_t>(), d_src.as<int32_t>(), pb_meta.as<uint8_t>(), pb_tab.as<uint8_t>(),
            stream);
        if (e == hipSuccess) pb_stream_device_used = 1;
        else (void)hipGetLastError();
    }
    if (!pb_stream_device_used) {
        std::vector<int32_t> gsp, src;
        std::vector<uint8_t> meta, tab;
        build_pb_stream(n, h_cptr.data(), h_cidx.data(), order, batch_ptr, G, NG, pb_balance,
                        nullptr, gsp, src, meta, tab);
        // every bound pbcd_prb_kernel indexes with, checked on the host for problems where that
        // is free (and on request, SPFM_VALIDATE=1): an out-of-range row or slot index would be
        // a device memory fault, i.e. a dead process
        if (nnz < ((int64_t)1 << 22) || getenv("SPFM_VALIDATE")) {
            const char* bad = validate_pb_stream(G, NG, gsp, src, meta, tab);
            if (bad) FAIL(SPFM_ERR_RUNTIME, std::string("internal: pbcd entry stream: ") + bad);
        }
        HIPC(hipMemcpyAsync(pb_tab.p, tab.data(), tab.size(), hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(pb_sp.p, gsp.data(), sizeof(int32_t) * gsp.size(),
                            hipMemcpyHostToDevice, stream));
        if (nnz > 0) {
            HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemcpyAsync(pb_meta.p, meta.data(), (size_t)nnz, hipMemcpyHostToDevice,
                                stream));
        }
        HIPC(hipStreamSynchronize(stream));  // the host staging vectors die here
    }
    if (nnz > 0) {
        hipLaunchKernelGGL((prb_gather_kernel<T>), dim3(cdiv(nnz, 256)), dim3(256), 0, stream,
                           nnz, d_src.as<int32_t>(), cidx.as<int32_t>(), cval.as<T>(),
                           pb_erow.as<int32_t>(), pb_eval.as<T>());
        HIPC(hipGetLastError());
    }
    HIPC(hipStreamSynchronize(stream));
    if (scache) {
        auto e = std::make_unique<StreamCache::Pb>();
        e->key = ckey;
        e->sp.share(pb_sp);
        e->erow.share(pb_erow);
        e->eval.share(pb_eval);
        e->meta.share(pb_meta);
        e->tab.share(pb_tab);
        scache->pb.insert(scache->pb.begin(), std::move(e));
        if (scache->pb.size() > StreamCache::kKeep) scache->pb.pop_back();
    }
    pb_stream_G = G;
    pb_stream_NG = NG;
    pb_stream_ready = true;
    return SPFM_OK;
}

// ---- relaxed runs for pbcd: a schedule of tiny strict steps -- the reference's own
// order -- is run as merged steps whose conflict rows every workgroup replays (pbcd_prb_kernel
// CR).  Its own step boundaries, conflict tables and entry stream (fixed slot map, the entries on
// conflict rows left out); degree 2, k <= 30, one GPU.
template <typename T>
int spfm_engine::ensure_pb_relax(int NG) {
    if (pbr_state != 0) return SPFM_OK;
    pbr_state = -1;
    int ncu = 0;
    HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    const int G = std::max(1, std::min(pbprb_G, ncu));
    std::vector<int32_t> cf_ptr, cf_row, cf_qq;
    std::vector<int64_t> cf_ia, cf_ib;
    std::vector<int16_t> clist;
    std::vector<uint8_t> skip;
    schedule_relax(n, d, h_cptr.data(), h_cidx.data(), order.data(), 64, 64, pbr_batch_ptr, cf_ptr,
                   cf_row, cf_qq, cf_ia, cf_ib, clist, skip);
    const int nbr = (int)pbr_batch_ptr.size() - 1;
    if ((double)nbr > 0.6 * (double)n_batches()) return SPFM_OK;  // not worth a second stream
    const size_t ncf = cf_row.size();
    std::vector<PrbConf<T>> hcf(ncf ? ncf : 1);
    {
        std::vector<T> hv((size_t)(nnz > 0 ? nnz : 1));
        HIPC(hipMemcpyAsync(hv.data(), cval.p, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
        HIPC(hipStreamSynchronize(stream));
        for (size_t c = 0; c < ncf; ++c) {
            hcf[c].row = cf_row[c];
            hcf[c].qq = cf_qq[c];
            hcf[c].xa = hv[(size_t)cf_ia[c]];
            hcf[c].xb = hv[(size_t)cf_ib[c]];
        }
    }
    std::vector<int32_t> gsp, src;
    std::vector<uint8_t> meta, tab;
    build_pb_stream(n, h_cptr.data(), h_cidx.data(), order, pbr_batch_ptr, G, NG, false, skip.data(),
                    gsp, src, meta, tab);
    const size_t ne = src.size();
    DevBuf d_src;
    HIPC(d_src.alloc(sizeof(int32_t) * (ne ? ne : 1)));
    HIPC(pbr_bptr.alloc(sizeof(int32_t) * pbr_batch_ptr.size()));
    HIPC(pbr_sp.alloc(sizeof(int32_t) * gsp.size()));
    HIPC(pbr_erow.alloc(sizeof(int32_t) * (ne ? ne : 1) + 256));
    HIPC(pbr_eval.alloc(sizeof(T) * (ne ? ne : 1) + 256));
    HIPC(pbr_meta.alloc((ne ? ne : 1) + 256));
    HIPC(pbr_tab.alloc(tab.size() + 16));
    HIPC(pbr_cfptr.alloc(sizeof(int32_t) * cf_ptr.size()));
    HIPC(pbr_cf.alloc(sizeof(PrbConf<T>) * hcf.size()));
    HIPC(pbr_clist.alloc(sizeof(int16_t) * clist.size() + 64));
    HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
    HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
    HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
    HIPC(hipMemcpyAsync(pbr_bptr.p, pbr_batch_ptr.data(), sizeof(int32_t) * pbr_batch_ptr.size(),
                        hipMemcpyHostToDevice, stream));
    HIPC(hipMemcpyAsync(pbr_sp.p, gsp.data(), sizeof(int32_t) * gsp.size(), hipMemcpyHostToDevice,
                        stream));
    HIPC(hipMemcpyAsync(pbr_tab.p, tab.data(), tab.size(), hipMemcpyHostToDevice, stream));
    HIPC(hipMemcpyAsync(pbr_cfptr.p, cf_ptr.data(), sizeof(int32_t) * cf_ptr.size(),
                        hipMemcpyHostToDevice, stream));
    HIPC(hipMemcpyAsync(pbr_cf.p, hcf.data(), sizeof(PrbConf<T>) * hcf.size(), hipMemcpyHostToDevice,
                        stream));
    HIPC(hipMemcpyAsync(pbr_clist.p, clist.data(), sizeof(int16_t) * clist.size(),
                        hipMemcpyHostToDevice, stream));
    if (ne > 0) {
        HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * ne, hipMemcpyHostToDevice, stream));
        HIPC(hipMemcpyAsync(pbr_meta.p, meta.data(), ne, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((prb_gather_kernel<T>), dim3(cdiv((int64_t)ne, 256)), dim3(256), 0, stream,
                           (int64_t)ne, d_src.as<int32_t>(), cidx.as<int32_t>(), cval.as<T>(),
                           pbr_erow.as<int32_t>(), pbr_eval.as<T>());
        HIPC(hipGetLastError());
    }
    HIPC(hipStreamSynchronize(stream));
    pbr_G = G;
    pbr_state = 1;
    return SPFM_OK;
}

template <typename T, int M, int L>
int spfm_engine::pbcd_prb_l(int order_idx, double beta, double gamma, double eta) {
    const double mu = loss == SPFM_LOSS_SQUARED ? 1.0 : (loss == SPFM_LOSS_LOGISTIC ? 0.25 : 2.0);
    double* Po = Pt.as<double>() + (size_t)order_idx * k * d;  // (d,k)
    RegState rs = regstate();
    // relaxed runs: a schedule of tiny steps (the reference order) on one GPU, degree 2, k <= 30
    bool relaxed = false;
    int rc = SPFM_OK;
    constexpr bool can_cr = M == 2 && L == 32;
    if constexpr (can_cr) {
        if (relax_on && !dist() && n_batches() > 0 &&
            (double)d / (double)n_batches() < 12.0) {
            rc = ensure_pb_relax<T>(kPbPrbThreads / L);
            if (rc) return rc;
            relaxed = pbr_state == 1;
        }
    }
    if (!relaxed) {
        rc = ensure_pb_stream<T>(kPbPrbThreads / L);
        if (rc) return rc;
    }
    const int G = relaxed ? pbr_G : pb_stream_G;
    pb_relax_active = relaxed ? 1 : 0;
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
    a.nb = n_batches();
    a.bptr = d_bptr.as<int32_t>();
    a.jsched = d_order.as<int32_t>();
    a.gsp = pb_sp.as<int32_t>();
    a.erow = pb_erow.as<int32_t>();
    a.emeta = pb_meta.as<uint8_t>();
    a.gtab = pb_tab.as<uint8_t>();
    a.slabA = pb_slabA.as<double>();
    a.slabB = pb_slabB.as<double>();
    a.rows_per = (int)std::max<int64_t>((n + G - 1) / G, 1);
    a.n_rows = (int)n;
    a.abort_flag = prb_abort.as<unsigned>();
    a.spin_max = spin_max;
    a.n_ranks = peer_ready ? n_ranks : 1;
    a.rank = rank;
    a.slabC = peer_ready ? peer_tab_pb.as<double*>() : nullptr;
    a.cf_ptr = nullptr;
    a.cf = nullptr;
    a.clist = nullptr;
    a.slabR = nullptr;
    if (relaxed) {
        a.nb = (int)pbr_batch_ptr.size() - 1;
        a.bptr = pbr_bptr.as<int32_t>();
        a.gsp = pbr_sp.as<int32_t>();
        a.erow = pbr_erow.as<int32_t>();
        a.emeta = pbr_meta.as<uint8_t>();
        a.gtab = pbr_tab.as<uint8_t>();
        a.cf_ptr = pbr_cfptr.as<int32_t>();
        a.cf = pbr_cf.p;
        a.clist = pbr_clist.as<int16_t>();
        HIPC(pbr_slabR.alloc(sizeof(double) * 2 * 64 * L));
        HIPC(hipMemsetAsync(pbr_slabR.p, 0, sizeof(double) * 2 * 64 * L, stream));
        a.slabR = pbr_slabR.as<double>();
    }
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
    auto go = [&](auto stamp_tag) -> int {
        constexpr bool STc = decltype(stamp_tag)::value;
        const size_t lds = std::max(kPrbLds, pbcd_prb_lds_bytes<T, M, L>());
        HIPC(hipFuncSetAttribute((const void*)pbcd_prb_kernel<T, M, L, STc>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (!resident_ok((const void*)pbcd_prb_kernel<T, M, L, STc>, kPbPrbThreads, lds, G))
            return kNotResident;
        hipLaunchKernelGGL((pbcd_prb_kernel<T, M, L, STc>), dim3(launch_groups(G)),
                           dim3(kPbPrbThreads), lds, stream, a, pb_eval.as<T>(), pb_rec.as<T>(), Po,
                           k, d, lams.as<double>(), loss, reg, rs, ncache, mu, beta, gamma, eta,
                           prb_viol.as<double>());
        return SPFM_OK;
    };
    // the diagnostic (timer) instantiation exists for float storage, degree 2, k <= 30
    constexpr bool can_stamp = std::is_same<T, float>::value && M == 2 && L == 32;
    if (pb_stamp_on && !can_stamp)
        FAIL(SPFM_ERR_UNSUPPORTED, "pbprb_stamps: built for float storage, degree 2, k <= 30");
    rc = SPFM_OK;
    bool fired = false;
    if constexpr (can_cr) {
        if (relaxed) {
            auto fire_cr = [&](auto* fn) -> int {
                const size_t lds = std::max(kPrbLds, pbcd_prb_lds_bytes<T, M, L, true>());
                HIPC(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
                if (!resident_ok((const void*)fn, kPbPrbThreads, lds, G)) return kNotResident;
                hipLaunchKernelGGL(fn, dim3(launch_groups(G)), dim3(kPbPrbThreads), lds, stream, a,
                                   pbr_eval.as<T>(), pb_rec.as<T>(), Po, k, d, lams.as<double>(), loss,
                                   reg, rs, ncache, mu, beta, gamma, eta, prb_viol.as<double>());
                return SPFM_OK;
            };
            bool stamped = false;
            if constexpr (can_stamp) {
                if (pb_stamp_on) {
                    HIPC(pb_stamps.alloc(sizeof(long long) * 16 * (size_t)G));
                    HIPC(hipMemsetAsync(pb_stamps.p, 0, pb_stamps.bytes, stream));
                    a.stamps = pb_stamps.as<long long>();
                    rc = fire_cr(&pbcd_prb_kernel<T, M, L, true, true>);
                    stamped = true;
                }
            }
            if (!stamped) rc = fire_cr(&pbcd_prb_kernel<T, M, L, false, true>);
            fired = true;
        }
    }
    if constexpr (can_stamp) {
        if (!fired && pb_stamp_on) {
            rc = go(std::true_type{});
            fired = true;
        }
    }
    if (!fired) rc = go(std::false_type{});
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
int spfm_engine::pbcd_prb_m(int order_idx, double beta, double gamma, double eta) {
    if (k <= 30) return pbcd_prb_l<T, M, 32>(order_idx, beta, gamma, eta);
    return pbcd_prb_l<T, M, 64>(order_idx, beta, gamma, eta);
}

template <typename T>
int spfm_engine::pbcd_prb_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
    switch (M) {
        case 0: return pbcd_prb_m<T, 0>(order_idx, beta, gamma, eta);
        case 2: return pbcd_prb_m<T, 2>(order_idx, beta, gamma, eta);
        case 3: return pbcd_prb_m<T, 3>(order_idx, beta, gamma, eta);
        case 4: return pbcd_prb_m<T, 4>(order_idx, beta, gamma, eta);
    }
    FAIL(SPFM_ERR_UNSUPPORTED, "persistent pbcd pass: degree outside {2,3,4,all-subsets}");
}


#define SPFM_CAT_(a, b) a##b
#define SPFM_CAT(a, b) SPFM_CAT_(a, b)
SPFM_DEFINE_BRANCH_COUNTS(SPFM_CAT(spfm_branch_counts_pbprb_, SPFM_TU_TAG))

template int spfm_engine::pbcd_prb_dispatch<SPFM_TU_T>(int, int, double, double, double);
