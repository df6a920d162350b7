// spfm_engine_pbcd.hip -- multi-kernel pbcd, the pbcd epoch driver, the pbcd half of the host-stepped epochs
#include "spfm_engine.hip.h"
#include "spfm_pbcd.hip.h"

using namespace spfm;

// ------------------------------------------------------------------- pbcd
template <typename T, int M, int L, int C>
int spfm_engine::pbcd_body_lc(int order_idx, double beta, double gamma, double eta) {
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

// host-side audit of build_pb_stream's output against what the kernel assumes: group
// boundaries monotone and ending at nnz; every entry a valid CSC position whose row lies in
// the workgroup's block; slot index inside the group < 64 / NG; a group's entries sorted by
// slot; at most 64 columns per step.  Returns nullptr or what is wrong.
const char* spfm_engine::validate_pb_stream(int G, int NG, const std::vector<int32_t>& gsp,
                               const std::vector<int32_t>& src,
                               const std::vector<uint8_t>& meta,
                               const std::vector<uint8_t>& tab) const {
    const int nb = n_batches();
    const size_t stride = (size_t)NG + 1;
    const int64_t rows_per = std::max<int64_t>((n + G - 1) / G, 1);
    const int qm = 64 / NG;
    if (gsp.size() != (size_t)G * nb * stride + 1) return "boundary table has the wrong size";
    if (tab.size() != (size_t)G * (size_t)std::max(nb, 1) * 64) return "slot table has the wrong size";
    if ((int64_t)src.size() != nnz || (int64_t)meta.size() != nnz) return "entry count != nnz";
    for (int b = 0; b < nb; ++b)
        if (batch_ptr[b + 1] - batch_ptr[b] > 64) return "a step has more than 64 columns";
    for (size_t t = 0; t + 1 < gsp.size(); ++t)
        if (gsp[t] > gsp[t + 1] || gsp[t] < 0) return "group boundaries not monotone";
    if (gsp.back() != (int32_t)nnz) return "group boundaries do not end at nnz";
    for (int g = 0; g < G; ++g)
        for (int b = 0; b < nb; ++b) {
            // the slot table of (g, b): every slot the kernel writes (this step's columns and
            // those of the exchange buffer's next use) belongs to exactly one (group, t)
            const int nc = batch_ptr[b + 1] - batch_ptr[b];
            const int nc2 = b + 2 < nb ? batch_ptr[b + 3] - batch_ptr[b + 2] : 0;
            const int nw = std::max(nc, nc2);
            const uint8_t* tb = &tab[((size_t)g * nb + b) * 64];
            int seen[64] = {0};
            for (int z = 0; z < 64; ++z)
                if (tb[z] != 0xFF) {
                    if (tb[z] >= nw) return "slot table names a slot nobody reads";
                    if (seen[tb[z]]++) return "slot table names a slot twice";
                }
            for (int q = 0; q < nw; ++q)
                if (!seen[q]) return "slot table misses a slot";
            for (int grp = 0; grp < NG; ++grp) {
                const size_t at = ((size_t)g * nb + b) * stride + (size_t)grp;
                int prev_slot = 0;
                for (int32_t e = gsp[at]; e < gsp[at + 1]; ++e) {
                    const int32_t pos = src[(size_t)e];
                    if (pos < 0 || pos >= nnz) return "entry points outside the CSC arrays";
                    const int64_t row = h_cidx[(size_t)pos];
                    if (row < 0 || row >= n) return "row index out of range";
                    if (row / rows_per != g) return "entry outside its workgroup's row block";
                    const int slot = meta[(size_t)e] & 0x3f;  // (0x80 / 0x40: shared-row flags)
                    if (slot >= qm) return "slot index >= slots per group";
                    if (slot < prev_slot) return "a group's entries are not sorted by slot";
                    prev_slot = slot;
                    const int q = tb[grp * qm + slot];
                    if (q >= nc) return "slot beyond the step's columns";
                    // ... and the entry really belongs to that column
                    const int32_t j = order[(size_t)batch_ptr[b] + (size_t)q];
                    if (pos < h_cptr[(size_t)j] || pos >= h_cptr[(size_t)j + 1])
                        return "entry filed under another column's slot";
                }
            }
        }
    return nullptr;
}

template <typename T, int M>
int spfm_engine::pbcd_body(int order_idx, double beta, double gamma, double eta) {
    if (k <= 8) return pbcd_body_lc<T, M, 8, 1>(order_idx, beta, gamma, eta);
    if (k <= 16) return pbcd_body_lc<T, M, 16, 1>(order_idx, beta, gamma, eta);
    if (k <= 32) return pbcd_body_lc<T, M, 32, 1>(order_idx, beta, gamma, eta);
    if (k <= 64) return pbcd_body_lc<T, M, 64, 1>(order_idx, beta, gamma, eta);
    if (k <= 128) return pbcd_body_lc<T, M, 64, 2>(order_idx, beta, gamma, eta);
    return pbcd_body_lc<T, M, 64, 4>(order_idx, beta, gamma, eta);
}

template <typename T>
int spfm_engine::pbcd_dispatch(int M, int order_idx, double beta, double gamma, double eta) {
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

int spfm_engine::pbcd_epoch(int order_idx, int degree, double beta, double gamma, double eta,
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

template <typename T>
int spfm_engine::host_pbcd_precompute(int M, int order_idx) {
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

template <typename T, int M, int L, int C>
int spfm_engine::host_sums_pbcd(int b, double* out) {
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
int spfm_engine::host_apply_pbcd(int b, const double* p_new, const double* p_old) {
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


template <typename T>
int spfm_engine::host_step_pbcd_t(bool sums, int b, double* out, const double* p_new,
                                  const double* p_old) {
#define SPFM_HPB(MM, LL, CC)                                       \
    return sums ? host_sums_pbcd<T, MM, LL, CC>(b, out)            \
                : host_apply_pbcd<T, MM, LL, CC>(b, p_new, p_old)
#define SPFM_HPM(MM)                    \
    do {                                \
        if (k <= 8) SPFM_HPB(MM, 8, 1);   \
        if (k <= 16) SPFM_HPB(MM, 16, 1); \
        if (k <= 32) SPFM_HPB(MM, 32, 1); \
        if (k <= 64) SPFM_HPB(MM, 64, 1); \
        if (k <= 128) SPFM_HPB(MM, 64, 2); \
        SPFM_HPB(MM, 64, 4);             \
    } while (0)
    switch (kind_of(host_degree)) {
        case 0: SPFM_HPM(0);
        case 2: SPFM_HPM(2);
        case 3: SPFM_HPM(3);
        case 4: SPFM_HPM(4);
        case 5: SPFM_HPM(5);
        case 6: SPFM_HPM(6);
    }
#undef SPFM_HPM
#undef SPFM_HPB
    FAIL(SPFM_ERR_UNSUPPORTED, "degree outside 2..6");
}

SPFM_DEFINE_BRANCH_COUNTS(spfm_branch_counts_pbcd)

// used by the pcd unit (host_epoch_begin, host_step)
template int spfm_engine::host_pbcd_precompute<float>(int, int);
template int spfm_engine::host_pbcd_precompute<double>(int, int);
template int spfm_engine::host_step_pbcd_t<float>(bool, int, double*, const double*, const double*);
template int spfm_engine::host_step_pbcd_t<double>(bool, int, double*, const double*, const double*);
