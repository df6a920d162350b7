// spfm_engine_wide.hip -- wide persistent passes (pcdw_kernel): steps of up to 512 columns
#include "spfm_engine.hip.h"
#include "spfm_pcd.hip.h"

using namespace spfm;

// Workgroups of the wide pass.  Its exchange (reduce-scatter to slot owners + all-gather)
// grows with the workgroup count, the entry loops shrink with it; measured optimum: about 160
// entries per workgroup and step (the shard of a multi-GPU run -- 1.25 M rows of the
// 10M x 1M problem, 22.7 k entries per step: 23.9 ms per component pass at 128 workgroups,
// 27.3 at 256; 2M x 200k, 36.5 k entries per step: 7.5 us per step at 256, 8.7 at 128).  More
// workgroups than that when the rows fit LDS only then.  An explicit "pcdw_groups" wins;
// concurrent tenants keep to their share of the CUs.
int spfm_engine::wide_groups(int ncu, size_t lds_max) const {
    int g = pcdw_G;
    if (g <= 0) {
        const int64_t steps = std::max<int64_t>(1, (int64_t)batch_ptr.size() - 1);
        const int64_t per_step = nnz / steps;
        g = (int)std::min<int64_t>(ncu, std::max<int64_t>(64, ((per_step / 160 + 15) / 16) * 16));
        auto fits = [&](int gg) {
            const size_t rows_per = ((size_t)n + (size_t)gg - 1) / (size_t)gg;
            return (wide_ep ? kPcdweLdsFixed : kPcdwLdsFixed) + rows_per * 8 + 16 <= lds_max;
        };
        if (!fits(g) && fits(ncu))
            while (g < ncu && !fits(g)) g = std::min(ncu, g + 16);
    }
    g = std::min(g, std::max(1, ncu / co_tenants));
    return std::max(1, std::min(g, ncu));
}

template <typename T>
int spfm_engine::ensure_wide() {
    int ncu = 0, lds_max = 0;
    HIPC(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
    const int G = wide_groups(ncu, (size_t)lds_max);
    if (wide_ready && wide_G == G) return SPFM_OK;
    const int nb_ = n_batches();
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
    // wbase[b] = columns + steps in front of step b (ncols + 1 boundaries per step)
    std::vector<int32_t> wbase((size_t)nb_ + 1, 0);
    for (int b = 0; b < nb_; ++b)
        wbase[(size_t)b + 1] = wbase[(size_t)b] + (batch_ptr[b + 1] - batch_ptr[b]) + 1;
    const size_t tot = (size_t)wbase[(size_t)nb_];
    DevBuf d_src, d_hz;
    HIPC(d_src.alloc(sizeof(int32_t) * nz));
    HIPC(d_hz.alloc(nz));
    HIPC(w_wbase.alloc(sizeof(int32_t) * wbase.size()));
    HIPC(w_wsp.alloc(sizeof(int32_t) * ((size_t)G * tot + 1)));
    HIPC(w_erow.alloc(sizeof(int32_t) * nz + 64));
    HIPC(w_eval.alloc(sizeof(T) * nz + 64));
    HIPC(w_slabA.alloc(sizeof(double) * 2 * 32 * (size_t)G * 32));
    HIPC(w_slabB.alloc(sizeof(double) * 2 * 32 * 32));
    HIPC(prb_abort.alloc(sizeof(unsigned) * 4));
    HIPC(prow_old.alloc(sizeof(double) * (size_t)d));
    HIPC(prb_viol.alloc(sizeof(double) * (size_t)d));
    HIPC(prb_cn.alloc(sizeof(double) * (size_t)d));
    HIPC(hipMemsetAsync(prb_abort.p, 0, sizeof(unsigned) * 4, stream));
    HIPC(hipMemcpyAsync(w_wbase.p, wbase.data(), sizeof(int32_t) * wbase.size(),
                        hipMemcpyHostToDevice, stream));
    // the stream: on the device (spfm_ingest.hip device_wide_stream: the host builder's tables
    // exactly, tests/test_hip_stream.py) or by the host threads
    wide_stream_device_used = 0;
    if (stream_device && nnz >= (1 << 20) && (int64_t)d + nb_ < ((int64_t)1 << 31) / std::max(G, 1)) {
        const hipError_t e = device_wide_stream(
            n, d, nnz, G, nb_, d_order.as<int32_t>(), d_bptr.as<int32_t>(), cptr.as<int64_t>(),
            cidx.as<int32_t>(), rptr.as<int64_t>(), ridx.as<int32_t>(), w_wsp.as<int32_t>(),
            d_src.as<int32_t>(), d_hz.as<uint8_t>(), stream);
        if (e == hipSuccess) wide_stream_device_used = 1;
        else (void)hipGetLastError();
    }
    if (!wide_stream_device_used) {
        std::vector<int32_t> wb2, wsp, src;
        std::vector<uint8_t> hz;
        build_wide_stream(n, h_cptr.data(), h_cidx.data(), order, batch_ptr, G, wb2, wsp, src, hz);
        HIPC(hipMemcpyAsync(w_wsp.p, wsp.data(), sizeof(int32_t) * wsp.size(),
                            hipMemcpyHostToDevice, stream));
        if (nnz > 0) {
            HIPC(hipMemcpyAsync(d_src.p, src.data(), sizeof(int32_t) * (size_t)nnz,
                                hipMemcpyHostToDevice, stream));
            HIPC(hipMemcpyAsync(d_hz.p, hz.data(), (size_t)nnz, hipMemcpyHostToDevice, stream));
        }
        HIPC(hipStreamSynchronize(stream));  // the host staging vectors die here
    }
    if (nnz > 0) {
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
    wide_tot = (int)tot;
    wide_ready = true;
    return SPFM_OK;
}

PcdwArgs spfm_engine::wide_args() {
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
    a.lds_rows = 0;
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
int spfm_engine::wide_launch(PcdwArgs& a, PcdwParams& pp, T* Aptr) {
    int lds_max = 0;
    HIPC(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
    constexpr bool can_lr = std::is_same<T, float>::value;
    // the entry-parallel form (pcdwe_kernel; option "wide_ep" = 0: the thread-per-column form)
    const bool use_ep = wide_ep;
    const size_t fixed = use_ep ? kPcdweLdsFixed : kPcdwLdsFixed;
    const size_t lds_lr = fixed + (size_t)a.rows_per * (KIND == 0 ? 8 : 4) + 16;
    const bool lr_ok = can_lr && prb_lds && loss == SPFM_LOSS_SQUARED;
    const bool use_lr = lr_ok && lds_lr <= (size_t)lds_max && wide_lds_cap < 0;
    // the block does not fit: its first rows in LDS, the others in global memory (LR = 2) -- when
    // at least an eighth of the block gets a place (option "wide_lds_rows": -1 as many as fit,
    // 0 none, n > 0 at most n -- the test hook that makes small problems take this path)
    int hyb_rows = 0;
    if (lr_ok && !use_lr && wide_lds_cap != 0 && !(wide_stamp_on && !use_ep)) {
        const size_t room = (size_t)lds_max > fixed + 64 ? (size_t)lds_max - fixed - 64 : 0;
        hyb_rows = (int)std::min<size_t>(room / (KIND == 0 ? 8 : 4), (size_t)a.rows_per);
        if (wide_lds_cap > 0) hyb_rows = std::min(hyb_rows, wide_lds_cap);
        else if (hyb_rows * 8 < a.rows_per) hyb_rows = 0;
    }
    const bool use_hyb = hyb_rows > 0;
    a.lds_rows = use_hyb ? hyb_rows : 0;
    const size_t lds_hyb = fixed + (size_t)hyb_rows * (KIND == 0 ? 8 : 4) + 16;
    wide_lr_active = use_lr ? 1 : (use_hyb ? 2 : 0);
    wide_ep_active = use_ep ? 1 : 0;
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
    // squared loss, float, entry-parallel form: 8-byte (A, residual) records (option "wide_rec8")
    const bool rec8 = packed && can_lr && use_ep && loss == SPFM_LOSS_SQUARED && wide_rec8 &&
                      !wide_stamp_on;
    if (packed) {
        HIPC(w_rec.alloc(sizeof(PcdwRec<T>) * (size_t)n));
        rec = w_rec.as<PcdwRec<T>>();
        if constexpr (can_lr && KIND == 0) {
            if (rec8)
                hipLaunchKernelGGL(pcdw_pack8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, pp.ctl,
                                   n, pp.a_stride, yy.as<float>(), (const float*)Aptr,
                                   reinterpret_cast<float2*>(rec));
        }
        if (!rec8)
            hipLaunchKernelGGL((pcdw_pack_kernel<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream,
                               pp.ctl, n, pp.a_stride, yy.as<T>(), Aptr, rec);
        HIPC(hipGetLastError());
    }
    auto unpack = [&]() -> int {
        if (packed) {
            if constexpr (can_lr && KIND == 0) {
                if (rec8)
                    hipLaunchKernelGGL(pcdw_unpack8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream,
                                       pp.ctl, n, pp.a_stride, reinterpret_cast<const float2*>(rec),
                                       yy.as<float>(), (float*)Aptr);
            }
            if (!rec8)
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
            int frc = use_lr ? (use_ep ? fire(&pcdwe_kernel<T, KIND, 1, true>, lds)
                                       : fire(&pcdw_kernel<T, KIND, 1, true>, lds))
                      : (use_ep ? (use_hyb ? fire(&pcdwe_kernel<T, KIND, 2, true>, std::max(lds_hyb, kPrbLds))
                                           : fire(&pcdwe_kernel<T, KIND, 0, true>, std::max(fixed, kPrbLds)))
                                : fire(&pcdw_kernel<T, KIND, 0, true>, lds));
            if (frc) return frc;
            return unpack();
        }
    }
    if constexpr (can_lr) {
        if (use_lr)
            return use_ep ? fire(&pcdwe_kernel<T, KIND, 1>, std::max(lds_lr, kPrbLds))
                          : fire(&pcdw_kernel<T, KIND, 1>, std::max(lds_lr, kPrbLds));
        if (use_hyb) {
            if constexpr (KIND == 0) {
                if (rec8) {
                    int hrc8 = fire(&pcdwe_kernel<T, KIND, 2, false, true>, std::max(lds_hyb, kPrbLds));
                    if (hrc8) return hrc8;
                    return unpack();
                }
            }
            int hrc = use_ep ? fire(&pcdwe_kernel<T, KIND, 2>, std::max(lds_hyb, kPrbLds))
                             : fire(&pcdw_kernel<T, KIND, 2>, std::max(lds_hyb, kPrbLds));
            if (hrc) return hrc;
            return unpack();
        }
    }
    if constexpr (can_lr && KIND == 0) {
        if (rec8) {
            int frc8 = fire(&pcdwe_kernel<T, KIND, 0, false, true>, std::max(fixed, kPrbLds));
            if (frc8) return frc8;
            return unpack();
        }
    }
    int frc = use_ep ? fire(&pcdwe_kernel<T, KIND, 0>, std::max(fixed, kPrbLds))
                     : fire(&pcdw_kernel<T, KIND, 0>, kPrbLds);
    if (frc) return frc;
    return unpack();
}

template <typename T>
int spfm_engine::pcd_pass_wide(int order_idx, double beta, double gamma, double eta) {
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
int spfm_engine::lin_wide(double alpha) {
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


SPFM_DEFINE_BRANCH_COUNTS(spfm_branch_counts_wide)

template int spfm_engine::pcd_pass_wide<float>(int, double, double, double);
template int spfm_engine::pcd_pass_wide<double>(int, double, double, double);
template int spfm_engine::lin_wide<float>(double);
template int spfm_engine::lin_wide<double>(double);
