// spfm_schedule.cpp -- host-side construction of conflict-free coordinate batches.
//
// The reference sweeps coordinates strictly sequentially (optimizer/pcd.py:97,
// pbcd.py:110, cd_linear.py:10).  Two columns that share no row touch disjoint
// parts of A / y_pred, so their gradient reductions and their scatter updates
// commute; only the (scalar) prox + regularizer-cache recurrence must keep the
// order.  A batch = a set of pairwise row-disjoint columns; the device runs one
// batch as one dependent step and the result equals the reference sweep over the
// concatenated batch order (the reference accepts any order: pcd.py:86-87).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace spfm {

// EXACT: keep `order`; cut it into maximal runs of pairwise row-disjoint columns.
void schedule_exact(int64_t n_rows, int32_t d, const int64_t* cptr, const int32_t* cidx,
                    const int32_t* order, int max_batch, std::vector<int32_t>& batch_ptr) {
    std::vector<int32_t> stamp((size_t)n_rows, -1);
    batch_ptr.clear();
    batch_ptr.push_back(0);
    int32_t cur = 0;
    int size = 0;
    for (int32_t pos = 0; pos < d; ++pos) {
        const int32_t j = order[pos];
        bool conflict = (size >= max_batch);
        if (!conflict) {
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii)
                if (stamp[cidx[ii]] == cur) {
                    conflict = true;
                    break;
                }
        }
        if (conflict) {
            ++cur;
            batch_ptr.push_back(pos);
            size = 0;
        }
        for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) stamp[cidx[ii]] = cur;
        ++size;
    }
    batch_ptr.push_back(d);
}

// COLORED: first-fit greedy colouring of the column conflict graph, visiting the
// columns in `order`.  Every row keeps the list of colours already present in it (at
// most its number of entries); a column marks the colours of all its rows in a small
// local bitset and takes the lowest colour that is absent and not full.  Reading the
// rows' short lists (instead of one bitset of all colours per row) keeps the traffic
// per column at ~(entries x colours-so-far-in-row) bytes.  out_order = colour classes
// concatenated (columns keep their relative visiting order inside a class).
void schedule_colored(int64_t n_rows, int32_t d, const int64_t* cptr, const int32_t* cidx,
                      const int32_t* order, int max_batch, std::vector<int32_t>& out_order,
                      std::vector<int32_t>& batch_ptr) {
    // row capacities = entries per row
    std::vector<int64_t> rstart((size_t)n_rows + 1, 0);
    const int64_t nnz = cptr[d];
    for (int64_t ii = 0; ii < nnz; ++ii) rstart[(size_t)cidx[ii] + 1]++;
    for (int64_t i = 0; i < n_rows; ++i) rstart[(size_t)i + 1] += rstart[(size_t)i];
    std::vector<int32_t> rcol((size_t)nnz);          // colours present in each row
    std::vector<int32_t> rcnt((size_t)n_rows, 0);
    std::vector<uint64_t> used, full;
    std::vector<std::vector<int32_t>> classes;
    for (int32_t pos = 0; pos < d; ++pos) {
        const int32_t j = order[pos];
        const size_t W = classes.size() / 64 + 1;
        if (used.size() < W) {
            used.resize(W, 0);
            full.resize(W, 0);
        }
        for (size_t w = 0; w < W; ++w) used[w] = full[w];
        // the rows of a column are scattered over the whole matrix: every access below is a
        // cache miss unless requested ahead (two-stage software prefetch)
        const int64_t cb = cptr[j], ce = cptr[j + 1];
        constexpr int64_t PD = 16;
        for (int64_t ii = cb; ii < ce; ++ii) {
            if (ii + 2 * PD < ce) {
                const int32_t i2 = cidx[ii + 2 * PD];
                __builtin_prefetch(&rstart[(size_t)i2]);
                __builtin_prefetch(&rcnt[(size_t)i2]);
            }
            if (ii + PD < ce) __builtin_prefetch(&rcol[(size_t)rstart[(size_t)cidx[ii + PD]]]);
            const int32_t i = cidx[ii];
            const int32_t* rc = &rcol[(size_t)rstart[(size_t)i]];
            const int32_t cnt = rcnt[(size_t)i];
            for (int32_t t = 0; t < cnt; ++t) used[(size_t)rc[t] >> 6] |= 1ull << (rc[t] & 63);
        }
        size_t c = classes.size();
        for (size_t w = 0; w < W; ++w) {
            if (~used[w]) {
                const size_t cand = w * 64 + (size_t)__builtin_ctzll(~used[w]);
                if (cand < c) c = cand;
                break;
            }
        }
        if (c == classes.size()) classes.emplace_back();
        classes[c].push_back(j);
        if ((int)classes[c].size() >= max_batch) {
            if (full.size() <= (c >> 6)) {
                full.resize((c >> 6) + 1, 0);
                used.resize((c >> 6) + 1, 0);
            }
            full[c >> 6] |= (1ull << (c & 63));
        }
        for (int64_t ii = cb; ii < ce; ++ii) {  // rows are warm from the pass above
            const int32_t i = cidx[ii];
            rcol[(size_t)rstart[(size_t)i] + rcnt[(size_t)i]++] = (int32_t)c;
        }
    }
    out_order.clear();
    out_order.reserve((size_t)d);
    batch_ptr.clear();
    batch_ptr.push_back(0);
    for (auto& cl : classes) {
        if (cl.empty()) continue;
        out_order.insert(out_order.end(), cl.begin(), cl.end());
        batch_ptr.push_back((int32_t)out_order.size());
    }
    if (batch_ptr.size() == 1) batch_ptr.push_back(0);
}

// CSC -> CSR (counting sort; keeps ascending column order inside each row)
void csc_to_csr(int64_t n, int32_t d, const int64_t* cptr, const int32_t* cidx,
                std::vector<int64_t>& rptr, std::vector<int32_t>& ridx,
                std::vector<int64_t>& perm /* csr position -> csc position */) {
    const int64_t nnz = cptr[d];
    rptr.assign((size_t)n + 1, 0);
    for (int64_t ii = 0; ii < nnz; ++ii) rptr[(size_t)cidx[ii] + 1]++;
    for (int64_t i = 0; i < n; ++i) rptr[(size_t)i + 1] += rptr[(size_t)i];
    ridx.resize((size_t)nnz);
    perm.resize((size_t)nnz);
    std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
    for (int32_t j = 0; j < d; ++j)
        for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
            const int64_t dst = fill[(size_t)cidx[ii]]++;
            ridx[(size_t)dst] = j;
            perm[(size_t)dst] = ii;
        }
}

// Entry stream of the persistent row-block pass: entries sorted by (row block g,
// batch b, slot q, row).  sp[(g*nb + b)*65 + q] = first entry of slot q (65 = 64 slots
// + end), src[e] = position of entry e in the CSC arrays.  Requires batches of at
// most 64 columns and nnz < 2^31.
// lmask[(g*nb + b)*2 + {0,1}]: bit q set = slot q holds more than `long_thresh` entries
// of row block g ("long slot": processed by the whole workgroup instead of 4 lanes).
void build_rowblock_stream(int64_t n, const int64_t* cptr, const int32_t* cidx,
                           const std::vector<int32_t>& order,
                           const std::vector<int32_t>& batch_ptr, int G, int long_thresh,
                           std::vector<int32_t>& sp, std::vector<int32_t>& src,
                           std::vector<uint32_t>& lmask) {
    const int nb = (int)batch_ptr.size() - 1;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    sp.assign((size_t)G * nb * 65 + 1, 0);
    // pass 1: counts
    for (int b = 0; b < nb; ++b)
        for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
            const int32_t j = order[(size_t)batch_ptr[b] + q];
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                const int g = (int)(cidx[ii] / rows_per);
                sp[((size_t)g * nb + b) * 65 + q]++;
            }
        }
    // exclusive prefix sum in (g, b, q) order; slots >= ncols of a batch hold 0 entries, so
    // sp[..+q] for q in [ncols, 64] all equal the end of the (g, b) segment
    int64_t run = 0;
    for (size_t t = 0; t < sp.size(); ++t) {
        const int64_t c = sp[t];
        sp[t] = (int32_t)run;
        run += c;
    }
    lmask.assign((size_t)G * nb * 2, 0u);
    for (int g = 0; g < G; ++g)
        for (int b = 0; b < nb; ++b) {
            const size_t base = ((size_t)g * nb + b) * 65;
            for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q)
                if (sp[base + q + 1] - sp[base + q] > long_thresh)
                    lmask[((size_t)g * nb + b) * 2 + (q >> 5)] |= (1u << (q & 31));
        }
    src.resize((size_t)run);
    std::vector<int32_t> fill(sp.begin(), sp.end());
    for (int b = 0; b < nb; ++b)
        for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
            const int32_t j = order[(size_t)batch_ptr[b] + q];
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                const int g = (int)(cidx[ii] / rows_per);
                src[(size_t)fill[((size_t)g * nb + b) * 65 + q]++] = (int32_t)ii;
            }
        }
}

// Entry stream of the persistent pbcd pass: entries sorted by (row block g, batch b, slot
// group q % NG, slot, row).  gsp[(g*nb + b)*(NG+1) + grp] = first entry of group grp (NG+1
// boundaries per (g, b)), src[e] = position in the CSC arrays, meta[e] = slot index inside the
// group (q / NG, < 8) | 0x80 if the entry's row was touched by the previous step (its row
// state must be read after that step's scatter, not prefetched).  Batches of <= 64 columns,
// nnz < 2^31.
void build_pb_stream(int64_t n, const int64_t* cptr, const int32_t* cidx,
                     const std::vector<int32_t>& order, const std::vector<int32_t>& batch_ptr,
                     int G, int NG, std::vector<int32_t>& gsp, std::vector<int32_t>& src,
                     std::vector<uint8_t>& meta) {
    const int nb = (int)batch_ptr.size() - 1;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    const size_t stride = (size_t)NG + 1;
    gsp.assign((size_t)G * nb * stride + 1, 0);
    for (int b = 0; b < nb; ++b)
        for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
            const int32_t j = order[(size_t)batch_ptr[b] + q];
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                const int g = (int)(cidx[ii] / rows_per);
                gsp[((size_t)g * nb + b) * stride + (size_t)(q % NG)]++;
            }
        }
    int64_t run = 0;
    for (size_t t = 0; t < gsp.size(); ++t) {  // exclusive prefix sum; the pad word of each
        const int64_t c = gsp[t];              // (g, b) holds 0 entries = end of the last group
        gsp[t] = (int32_t)run;
        run += c;
    }
    src.resize((size_t)run);
    meta.resize((size_t)run);
    std::vector<int32_t> fill(gsp.begin(), gsp.end());
    std::vector<int32_t> last((size_t)n, -2);  // last step that touched the row
    // slots of one group in ascending order: q = grp, grp + NG, ... -> walk q ascending and the
    // per-group fill pointers keep (slot, row) order inside each group
    for (int b = 0; b < nb; ++b) {
        for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
            const int32_t j = order[(size_t)batch_ptr[b] + q];
            const uint8_t qi = (uint8_t)(q / NG);
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                const int32_t i = cidx[ii];
                const int g = (int)(i / rows_per);
                const size_t e = (size_t)fill[((size_t)g * nb + b) * stride + (size_t)(q % NG)]++;
                src[e] = (int32_t)ii;
                meta[e] = (uint8_t)(qi | (last[(size_t)i] == b - 1 ? 0x80 : 0));
            }
        }
        for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
            const int32_t j = order[(size_t)batch_ptr[b] + q];
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) last[(size_t)cidx[ii]] = b;
        }
    }
}

}  // namespace spfm
