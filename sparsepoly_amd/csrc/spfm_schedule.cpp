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
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
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

// RELAXED runs for the reference order (round 3).  `schedule_exact` cuts the visiting
// order into maximal runs of pairwise row-disjoint columns -- 2.6 columns on BASELINE config 2,
// so a sweep is 38 000 dependent steps that each pay a full exchange.  A relaxed run keeps
// adding consecutive columns although they share a few rows with earlier columns of the run;
// those CONFLICT ROWS are taken out of the row blocks' parallel partial sums and replayed, in
// column order and with their true intermediate state, by the (redundant, local) chain of
// every workgroup -- the dependency no longer crosses the exchange.  Same result as the
// sequential sweep; ~20 columns per step.  Limits that keep the chain's tables small: <= 64
// columns and <= `max_conf` conflict rows per run, a row touched by at most TWO columns of a run,
// <= 8 conflict rows per column.
//   rbptr            run boundaries over `order` (positions)
//   cf_ptr[run+1]    conflict rows of a run: cf_row, cf_qq (slot of the earlier column | slot
//                    of the later one << 8), cf_ia / cf_ib (CSC positions of the two entries)
//   clist[pos*8+t]   the conflict rows of the column at position pos: index inside its run's
//                    list | role << 8 (0 earlier, 1 later column of the row), -1 = none
//   skip[ii]         1: CSC entry ii lies on a conflict row of its run (not in the entry stream)
void schedule_relax(int64_t n_rows, int32_t d, const int64_t* cptr, const int32_t* cidx,
                    const int32_t* order, int max_cols, int max_conf,
                    std::vector<int32_t>& rbptr, std::vector<int32_t>& cf_ptr,
                    std::vector<int32_t>& cf_row, std::vector<int32_t>& cf_qq,
                    std::vector<int64_t>& cf_ia, std::vector<int64_t>& cf_ib,
                    std::vector<int16_t>& clist, std::vector<uint8_t>& skip) {
    constexpr int kPerCol = 8;
    std::vector<int32_t> row_run((size_t)n_rows, -1), row_slot((size_t)n_rows, 0);
    std::vector<uint8_t> row_cnt((size_t)n_rows, 0);
    std::vector<int64_t> row_pos((size_t)n_rows, 0);
    rbptr.assign(1, 0);
    cf_ptr.assign(1, 0);
    cf_row.clear();
    cf_qq.clear();
    cf_ia.clear();
    cf_ib.clear();
    clist.assign((size_t)d * kPerCol, (int16_t)-1);
    skip.assign((size_t)cptr[d], 0);
    int32_t run = 0, start = 0;
    int percol[64];
    for (int t = 0; t < 64; ++t) percol[t] = 0;
    int nconf = 0;
    int extra[64];
    for (int32_t pos = 0; pos < d; ++pos) {
        const int32_t j = order[pos];
        int q = pos - start;
        bool close = q >= max_cols;
        if (!close && q > 0) {  // would the column fit the run's limits?
            for (int t = 0; t < q; ++t) extra[t] = 0;
            int mine = 0;
            for (int64_t ii = cptr[j]; ii < cptr[j + 1] && !close; ++ii) {
                const int32_t i = cidx[ii];
                if (row_run[(size_t)i] != run) continue;
                if (row_cnt[(size_t)i] >= 2) {
                    close = true;  // a third column on the row
                    break;
                }
                ++mine;
                const int qa = row_slot[(size_t)i];
                if (percol[qa] + (++extra[qa]) > kPerCol) close = true;
            }
            if (mine > kPerCol || nconf + mine > max_conf) close = true;
        }
        if (close) {
            cf_ptr.push_back((int32_t)cf_row.size());
            rbptr.push_back(pos);
            ++run;
            start = pos;
            q = 0;
            nconf = 0;
            for (int t = 0; t < 64; ++t) percol[t] = 0;
        }
        for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
            const int32_t i = cidx[ii];
            if (row_run[(size_t)i] == run) {  // second column of this run on row i
                const int qa = row_slot[(size_t)i];
                const int c = nconf++;
                cf_row.push_back(i);
                cf_qq.push_back(qa | (q << 8));
                cf_ia.push_back(row_pos[(size_t)i]);
                cf_ib.push_back(ii);
                skip[(size_t)row_pos[(size_t)i]] = 1;
                skip[(size_t)ii] = 1;
                clist[(size_t)(start + qa) * kPerCol + (size_t)percol[qa]++] = (int16_t)c;
                clist[(size_t)pos * kPerCol + (size_t)percol[q]++] = (int16_t)(c | 0x100);
                row_cnt[(size_t)i] = 2;
            } else {
                row_run[(size_t)i] = run;
                row_cnt[(size_t)i] = 1;
                row_slot[(size_t)i] = q;
                row_pos[(size_t)i] = ii;
            }
        }
    }
    cf_ptr.push_back((int32_t)cf_row.size());
    rbptr.push_back(d);
}

// COLORED: first-fit greedy colouring of the column conflict graph, visiting the
// columns in `order`.  Every row keeps the list of colours already present in it (at
// most its number of entries); a column marks the colours of all its rows in a small
// local bitset and takes the lowest colour that is absent and not full.  Reading the
// rows' short lists (instead of one bitset of all colours per row) keeps the traffic
// per column at ~(entries x colours-so-far-in-row) bytes.  out_order = colour classes
// concatenated (columns keep their relative visiting order inside a class).
void schedule_colored_parallel(int64_t, int32_t, const int64_t*, const int32_t*, const int32_t*, int,
                               int, std::vector<int32_t>&, std::vector<int32_t>&);
int schedule_threads();

void schedule_colored(int64_t n_rows, int32_t d, const int64_t* cptr, const int32_t* cidx,
                      const int32_t* order, int max_batch, std::vector<int32_t>& out_order,
                      std::vector<int32_t>& batch_ptr) {
    // large problems: the parallel form (same greedy rule, deterministic for any thread count)
    if (d >= 4096 && cptr[d] >= (1 << 20)) {
        schedule_colored_parallel(n_rows, d, cptr, cidx, order, max_batch, schedule_threads(),
                                  out_order, batch_ptr);
        return;
    }
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

// ---- parallel form of the same greedy colouring (deterministic: the sequential result) ----
// Rounds of kRoundMax columns taken in visiting order.  (1) In parallel, every column of the
// round collects the set of colours present in its rows AS OF THE START OF THE ROUND (a
// snapshot: assignments of the same round are not seen, so nothing depends on thread timing).
// (2) One thread walks the round in order and gives each column the lowest colour that is
// neither in its snapshot set, nor full, nor taken in this round by a column that shares a row
// with it (merge of the two sorted row lists) -- exactly what the sequential first fit would
// have chosen.  (3) In parallel, the columns append their colour to their rows' lists (atomic
// slot counters: the order inside a row's list varies, the SET does not).  Every column is
// scanned once; the result is identical to schedule_colored's sequential loop for any thread
// count.  (The first version let a column pick ONE tentative colour in (1) and refused it in
// (2) when an earlier column of the round had taken it: 2.2 scans per column on config 2.)
namespace {

class SpinBarrier {
   public:
    explicit SpinBarrier(int n) : n_(n) {}
    void wait() {
        const int gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) == n_ - 1) {
            count_.store(0, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        } else {
            while (gen_.load(std::memory_order_acquire) == gen) std::this_thread::yield();
        }
    }

   private:
    const int n_;
    std::atomic<int> count_{0}, gen_{0};
};

bool columns_share_row(const int64_t* cptr, const int32_t* cidx, int32_t a, int32_t b) {
    int64_t i = cptr[a], ie = cptr[a + 1], j = cptr[b], je = cptr[b + 1];
    while (i < ie && j < je) {
        const int32_t ra = cidx[i], rb = cidx[j];
        if (ra == rb) return true;
        if (ra < rb) ++i;
        else ++j;
    }
    return false;
}

}  // namespace

void schedule_colored_parallel(int64_t n_rows, int32_t d, const int64_t* cptr, const int32_t* cidx,
                               const int32_t* order, int max_batch, int n_threads,
                               std::vector<int32_t>& out_order, std::vector<int32_t>& batch_ptr) {
    // columns per round: more columns mean fewer barriers but more in-round checks by the one
    // resolving thread (measured on 400k x 40k, 7 scanning threads: 32 -> 0.63 s, 64 -> 0.73 s,
    // 128 -> 1.0 s, 21 -> 0.55 s: three columns per scanning thread).  The result does not depend
    // on it.
    int kRoundMax = std::max(8, std::min(64, 3 * std::max(1, n_threads - 1)));
    if (const char* e = std::getenv("SPFM_SCHED_ROUND")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 1024) kRoundMax = v;
    }
    const int64_t nnz = cptr[d];
    // one 16-byte descriptor per row (start of its colour list, entries so far): one cache miss
    // per visited row instead of two
    struct RowList {
        int64_t start;
        std::atomic<int32_t> cnt;
        int32_t pad;
    };
    std::vector<RowList> rows((size_t)n_rows + 1);
    {
        std::vector<int64_t> rstart((size_t)n_rows + 1, 0);
        for (int64_t ii = 0; ii < nnz; ++ii) rstart[(size_t)cidx[ii] + 1]++;
        for (int64_t i = 0; i < n_rows; ++i) rstart[(size_t)i + 1] += rstart[(size_t)i];
        for (int64_t i = 0; i <= n_rows; ++i) {
            rows[(size_t)i].start = rstart[(size_t)i];
            rows[(size_t)i].cnt.store(0, std::memory_order_relaxed);
            rows[(size_t)i].pad = 0;
        }
    }
    std::vector<int32_t> rcol((size_t)nnz);
    std::vector<std::vector<int32_t>> classes;
    std::vector<uint64_t> full;  // bit c: class c holds max_batch columns
    // round state
    std::vector<int32_t> todo(order, order + d);  // columns in visiting order
    size_t head = 0;
    std::vector<int32_t> round_cols, colour((size_t)kRoundMax);
    round_cols.reserve(kRoundMax);
    size_t n_classes = 0, words = 1;
    bool done = false;
    SpinBarrier bar(n_threads);
    std::vector<std::vector<uint64_t>> used((size_t)kRoundMax);  // snapshot colour sets
    // the columns of a round are shared among the threads 1..T-1 (thread 0 prepares and resolves
    // the rounds; it only takes columns when it is alone)
    const int n_work = n_threads > 1 ? n_threads - 1 : 1;

    // (1) the colours present in the rows of column j as of the start of the round (plus the
    // full classes).  The rows of a column are scattered over the whole matrix: descriptor and
    // colour list are requested ahead (two-stage software prefetch), else every row costs two
    // dependent misses
    auto snapshot_set = [&](int32_t j, std::vector<uint64_t>& u) {
        u.assign(words, 0);
        for (size_t w = 0; w < full.size() && w < words; ++w) u[w] = full[w];
        const int64_t cb = cptr[j], ce = cptr[j + 1];
        constexpr int64_t PD = 12;
        for (int64_t ii = cb; ii < ce; ++ii) {
            if (ii + 2 * PD < ce) __builtin_prefetch(&rows[(size_t)cidx[ii + 2 * PD]]);
            if (ii + PD < ce) __builtin_prefetch(&rcol[(size_t)rows[(size_t)cidx[ii + PD]].start]);
            const RowList& r = rows[(size_t)cidx[ii]];
            const int32_t* rc = &rcol[(size_t)r.start];
            const int32_t cnt = r.cnt.load(std::memory_order_relaxed);
            for (int32_t t = 0; t < cnt; ++t) u[(size_t)rc[t] >> 6] |= 1ull << (rc[t] & 63);
        }
    };
    // (3) colour of an accepted column into its rows' lists
    auto commit = [&](int32_t j, int32_t c) {
        const int64_t cb = cptr[j], ce = cptr[j + 1];
        for (int64_t ii = cb; ii < ce; ++ii) {
            if (ii + 12 < ce) __builtin_prefetch(&rows[(size_t)cidx[ii + 12]], 1);
            RowList& r = rows[(size_t)cidx[ii]];
            const int32_t slot = r.cnt.fetch_add(1, std::memory_order_relaxed);
            rcol[(size_t)r.start + slot] = c;
        }
    };
    auto phase1 = [&](int wid) {
        for (size_t q = (size_t)wid; q < round_cols.size(); q += (size_t)n_work)
            snapshot_set(round_cols[q], used[q]);
    };
    auto phase3 = [&](int wid) {
        for (size_t q = (size_t)wid; q < round_cols.size(); q += (size_t)n_work)
            commit(round_cols[q], colour[q]);
    };

    auto worker = [&](int tid) {
        for (;;) {
            bar.wait();  // round prepared by thread 0
            if (done) return;
            phase1(tid - 1);
            bar.wait();  // thread 0 resolves the round
            bar.wait();
            phase3(tid - 1);
            bar.wait();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    // thread 0: drives the rounds
    for (;;) {
        round_cols.clear();
        while ((int)round_cols.size() < kRoundMax && head < todo.size())
            round_cols.push_back(todo[head++]);
        n_classes = classes.size();
        words = n_classes / 64 + 1;
        if (round_cols.empty()) {
            done = true;
            bar.wait();
            break;
        }
        bar.wait();
        if (n_threads == 1) phase1(0);
        bar.wait();
        // (2) first fit in visiting order: snapshot set, classes filled meanwhile, columns of
        // this round that hold the candidate colour
        for (size_t q = 0; q < round_cols.size(); ++q) {
            const std::vector<uint64_t>& u = used[q];
            int32_t c = 0;
            for (;; ++c) {
                if ((size_t)c >= classes.size()) break;  // nothing fits: a new class
                if ((size_t)c < n_classes && ((u[(size_t)c >> 6] >> (c & 63)) & 1ull)) {
                    // jump to the next colour outside the snapshot set
                    size_t w = (size_t)c >> 6;
                    uint64_t free_bits = ~u[w] & (~0ull << (c & 63));
                    while (free_bits == 0 && ++w < words) free_bits = ~u[w];
                    c = (free_bits == 0) ? (int32_t)(words * 64)
                                         : (int32_t)(w * 64 + (size_t)__builtin_ctzll(free_bits));
                    if ((size_t)c >= classes.size()) break;
                }
                if ((int)classes[(size_t)c].size() >= max_batch) continue;
                bool clash = false;
                for (size_t a = 0; a < q && !clash; ++a)
                    if (colour[a] == c &&
                        columns_share_row(cptr, cidx, round_cols[a], round_cols[q]))
                        clash = true;
                if (!clash) break;
            }
            if ((size_t)c >= classes.size()) {
                c = (int32_t)classes.size();
                classes.emplace_back();
            }
            colour[q] = c;
            classes[(size_t)c].push_back(round_cols[q]);
            if ((int)classes[(size_t)c].size() >= max_batch) {
                if (full.size() <= ((size_t)c >> 6)) full.resize(((size_t)c >> 6) + 1, 0);
                full[(size_t)c >> 6] |= 1ull << (c & 63);
            }
        }
        bar.wait();
        if (n_threads == 1) phase3(0);
        bar.wait();
    }
    for (auto& t : pool) t.join();
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

// threads for the parallel colouring: SPFM_THREADS, else min(hardware, 16); 1 = sequential
int schedule_threads() {
    if (const char* e = std::getenv("SPFM_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) return std::min(v, 64);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw, 16u));
}

// run fn(tid) on `n_threads` host threads (thread 0 = the caller)
template <typename F>
static void run_threads(int n_threads, F&& fn) {
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& th : pool) th.join();
}

// Transposition of a compressed sparse structure by host threads, the pattern of every ingest
// step below: the OUTPUT index space (rows of the CSR image, columns of the CSC image, row
// blocks of an entry stream) is cut into one contiguous range per thread; every thread scans
// ALL entries in input order (sequential reads: cheap) and handles those that fall into its
// range, so there are no write conflicts and the order inside an output segment is the input
// order (ascending) -- the result is identical for any thread count.

// CSC -> CSR (keeps ascending column order inside each row)
void csc_to_csr(int64_t n, int32_t d, const int64_t* cptr, const int32_t* cidx,
                std::vector<int64_t>& rptr, std::vector<int32_t>& ridx,
                std::vector<int64_t>& perm /* csr position -> csc position */) {
    const int64_t nnz = cptr[d];
    rptr.assign((size_t)n + 1, 0);
    ridx.resize((size_t)nnz);
    perm.resize((size_t)nnz);
    const int T = (nnz >= (1 << 20)) ? schedule_threads() : 1;
    const int64_t per = (n + T - 1) / T;
    run_threads(T, [&](int tid) {  // counts of the thread's rows
        const int64_t lo = per * tid, hi = std::min<int64_t>(n, lo + per);
        for (int64_t ii = 0; ii < nnz; ++ii) {
            const int64_t i = cidx[ii];
            if (i >= lo && i < hi) rptr[(size_t)i + 1]++;
        }
    });
    for (int64_t i = 0; i < n; ++i) rptr[(size_t)i + 1] += rptr[(size_t)i];
    std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
    run_threads(T, [&](int tid) {
        const int64_t lo = per * tid, hi = std::min<int64_t>(n, lo + per);
        for (int32_t j = 0; j < d; ++j)
            for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                const int64_t i = cidx[ii];
                if (i < lo || i >= hi) continue;
                const int64_t dst = fill[(size_t)i]++;
                ridx[(size_t)dst] = j;
                perm[(size_t)dst] = ii;
            }
    });
}

// CSR -> CSC (dataset.py:119-123's X.tocsc(), on host threads): rows ascending inside each
// column; perm = csc position -> csr position.  Returns false if a row's column indices are
// not strictly ascending (not canonical) or out of range.
bool csr_to_csc(int64_t n, int32_t d, const int64_t* rptr, const int32_t* ridx,
                std::vector<int64_t>& cptr, std::vector<int32_t>& cidx,
                std::vector<int64_t>& perm) {
    const int64_t nnz = rptr[n];
    cptr.assign((size_t)d + 1, 0);
    cidx.resize((size_t)nnz);
    perm.resize((size_t)nnz);
    const int T = (nnz >= (1 << 20)) ? schedule_threads() : 1;
    const int32_t per = (int32_t)((d + T - 1) / T);
    std::vector<char> bad((size_t)T, 0);
    run_threads(T, [&](int tid) {
        const int32_t lo = per * tid, hi = std::min<int32_t>(d, lo + per);
        if (tid == 0)
            for (int64_t i = 0; i < n && !bad[0]; ++i)
                for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii)
                    if (ridx[ii] < 0 || ridx[ii] >= d || (ii > rptr[i] && ridx[ii] <= ridx[ii - 1])) {
                        bad[0] = 1;
                        break;
                    }
        for (int64_t ii = 0; ii < nnz; ++ii) {
            const int32_t j = ridx[ii];
            if (j >= lo && j < hi) cptr[(size_t)j + 1]++;
        }
    });
    if (bad[0]) return false;
    for (int32_t j = 0; j < d; ++j) cptr[(size_t)j + 1] += cptr[(size_t)j];
    std::vector<int64_t> fill(cptr.begin(), cptr.end() - 1);
    run_threads(T, [&](int tid) {
        const int32_t lo = per * tid, hi = std::min<int32_t>(d, lo + per);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t ii = rptr[i]; ii < rptr[i + 1]; ++ii) {
                const int32_t j = ridx[ii];
                if (j < lo || j >= hi) continue;
                const int64_t dst = fill[(size_t)j]++;
                cidx[(size_t)dst] = (int32_t)i;
                perm[(size_t)dst] = ii;
            }
    });
    return true;
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
                           std::vector<uint32_t>& lmask, const uint8_t* skip) {
    // skip (optional): CSC entries that are not part of the stream (relaxed runs: the entries
    // on conflict rows, which the chains replay)
    const int nb = (int)batch_ptr.size() - 1;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    sp.assign((size_t)G * nb * 65 + 1, 0);
    const int64_t nnz = cptr[order.size()];
    const int T = std::min(G, (nnz >= (1 << 20)) ? schedule_threads() : 1);
    const int gper = (G + T - 1) / T;  // row blocks per thread
    // pass 1: counts (a thread owns the row blocks [g0, g1), hence its slice of sp)
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = (int64_t)g1 * rows_per;
        for (int b = 0; b < nb; ++b)
            for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
                const int32_t j = order[(size_t)batch_ptr[b] + q];
                for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                    const int64_t i = cidx[ii];
                    if (i < rlo || i >= rhi) continue;
                    if (skip && skip[(size_t)ii]) continue;
                    sp[((size_t)(i / rows_per) * nb + b) * 65 + q]++;
                }
            }
    });
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
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = (int64_t)g1 * rows_per;
        for (int b = 0; b < nb; ++b)
            for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
                const int32_t j = order[(size_t)batch_ptr[b] + q];
                for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                    const int64_t i = cidx[ii];
                    if (i < rlo || i >= rhi) continue;
                    if (skip && skip[(size_t)ii]) continue;
                    src[(size_t)fill[((size_t)(i / rows_per) * nb + b) * 65 + q]++] = (int32_t)ii;
                }
            }
    });
}

// Entry stream of the persistent pbcd pass: entries sorted by (row block g, step b, slot group,
// slot index t inside the group, row).  gsp[(g*nb + b)*(NG+1) + grp] = first entry of group grp
// (NG+1 boundaries per (g, b)), src[e] = position in the CSC arrays, meta[e] = t (< 8) | 0x80 if
// the entry's row was touched by the previous step (its row state must be read after that
// step's scatter, not prefetched).  Steps of <= 64 columns, nnz < 2^31.
//
// Balanced slot groups (round 4).  A group of L lanes walks its entries one after the other and
// every step waits for the slowest group of the slowest workgroup.  With the fixed map
// "slot q -> group q % NG" the busiest of the 4096 groups of BASELINE config 4 holds 14.9
// entries per step against a mean of 4.4 (groups 0..4 carry three columns, the others two, plus
// the Poisson spread of ~2 entries per column and row block).  Here every (row block, step)
// gets its own map, chosen for its entry counts: columns in descending order of their entries in
// the block go to the group with the fewest entries so far (at most QM = 64 / NG columns per
// group) -- 6.5 entries in the busiest group.  tab[(g*nb + b)*64 + grp*QM + t] = the step's slot
// (column index inside the step) that group grp handles as its t-th, 0xFF = none; every slot
// below `max(ncols(b), ncols(b+2))` is assigned (the kernel rewrites the unused ones with zeros).
// `skip` (relaxed runs, schedule_relax): entries that are not part of the stream -- their rows
// still count as touched by their step.
void build_pb_stream(int64_t n, const int64_t* cptr, const int32_t* cidx,
                     const std::vector<int32_t>& order, const std::vector<int32_t>& batch_ptr,
                     int G, int NG, bool balance, const uint8_t* skip, std::vector<int32_t>& gsp,
                     std::vector<int32_t>& src, std::vector<uint8_t>& meta,
                     std::vector<uint8_t>& tab) {
    const int nb = (int)batch_ptr.size() - 1;
    const int QM = 64 / NG;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    const size_t stride = (size_t)NG + 1;
    gsp.assign((size_t)G * nb * stride + 1, 0);
    tab.assign((size_t)G * (size_t)std::max(nb, 1) * 64, 0xFF);
    const int64_t nnz = cptr[order.size()];
    const int T = std::min(G, (nnz >= (1 << 20)) ? schedule_threads() : 1);
    const int gper = (G + T - 1) / T;
    auto ncols_of = [&](int b) { return (b >= 0 && b < nb) ? batch_ptr[b + 1] - batch_ptr[b] : 0; };
    // the part of column j that lies in the row range [rlo, rhi): rows ascend inside a column
    auto sub_range = [&](int32_t j, int64_t rlo, int64_t rhi, int64_t& lo, int64_t& hi) {
        const int32_t* first = cidx + cptr[j];
        const int32_t* last = cidx + cptr[j + 1];
        lo = cptr[j] + (std::lower_bound(first, last, (int32_t)std::min<int64_t>(rlo, INT32_MAX)) - first);
        hi = cptr[j] + (std::lower_bound(first, last, (int32_t)std::min<int64_t>(rhi, INT32_MAX)) - first);
    };
    // pass 1: per (row block, step) the entries of every slot; the slot -> (group, t) map; the
    // groups' entry counts
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        if (g0 >= g1) return;
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = std::min<int64_t>(n, (int64_t)g1 * rows_per);
        std::vector<int32_t> cnt((size_t)(g1 - g0) * 64);
        int idx[64], load[64], used[64];
        for (int b = 0; b < nb; ++b) {
            const int nc = ncols_of(b);
            const int nw = std::min(64, std::max(nc, ncols_of(b + 2)));
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int q = 0; q < nc; ++q) {
                int64_t lo, hi;
                sub_range(order[(size_t)batch_ptr[b] + q], rlo, rhi, lo, hi);
                for (int64_t ii = lo; ii < hi; ++ii)
                    if (!skip || !skip[ii]) cnt[(size_t)(cidx[ii] / rows_per - g0) * 64 + (size_t)q]++;
            }
            for (int g = g0; g < g1; ++g) {
                const int32_t* c = &cnt[(size_t)(g - g0) * 64];
                uint8_t* tb = &tab[((size_t)g * nb + b) * 64];
                int32_t* gs = &gsp[((size_t)g * nb + b) * stride];
                if (!balance) {  // the fixed map: slot q -> group q % NG, t = q / NG
                    for (int q = 0; q < nw; ++q) {
                        tb[(q % NG) * QM + q / NG] = (uint8_t)q;
                        gs[q % NG] += c[q];
                    }
                    continue;
                }
                for (int q = 0; q < nw; ++q) idx[q] = q;
                std::stable_sort(idx, idx + nw, [&](int x, int y) { return c[x] > c[y]; });
                for (int r = 0; r < NG; ++r) load[r] = used[r] = 0;
                for (int z = 0; z < nw; ++z) {
                    const int q = idx[z];
                    int best = -1;
                    for (int r = 0; r < NG; ++r)
                        if (used[r] < QM && (best < 0 || load[r] < load[best])) best = r;
                    tb[best * QM + used[best]] = (uint8_t)q;
                    used[best]++;
                    load[best] += c[q];
                }
                for (int r = 0; r < NG; ++r) gs[r] = load[r];
            }
        }
    });
    int64_t run = 0;
    for (size_t t = 0; t < gsp.size(); ++t) {  // exclusive prefix sum; the pad word of each
        const int64_t c = gsp[t];              // (g, b) holds 0 entries = end of the last group
        gsp[t] = (int32_t)run;
        run += c;
    }
    src.resize((size_t)run);
    meta.resize((size_t)run);
    std::vector<int32_t> last((size_t)n, -2);    // last step that touched the row (own rows only)
    // pass 2: fill.  Inside a group the entries are ordered by (t, row): the fill position of a
    // slot = the group's first entry + the entries of the group's earlier slots
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        if (g0 >= g1) return;
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = std::min<int64_t>(n, (int64_t)g1 * rows_per);
        std::vector<int32_t> pos((size_t)(g1 - g0) * 64);   // next fill position per (block, slot)
        std::vector<uint8_t> where((size_t)(g1 - g0) * 64);  // grp << 3 | t per (block, slot)
        std::vector<int64_t> lo_q(64), hi_q(64);
        for (int b = 0; b < nb; ++b) {
            const int nc = ncols_of(b);
            std::fill(pos.begin(), pos.end(), 0);
            for (int q = 0; q < nc; ++q) {
                sub_range(order[(size_t)batch_ptr[b] + q], rlo, rhi, lo_q[(size_t)q], hi_q[(size_t)q]);
                for (int64_t ii = lo_q[(size_t)q]; ii < hi_q[(size_t)q]; ++ii)
                    if (!skip || !skip[ii])
                        pos[(size_t)(cidx[ii] / rows_per - g0) * 64 + (size_t)q]++;  // counts first
            }
            for (int g = g0; g < g1; ++g) {
                int32_t* pc = &pos[(size_t)(g - g0) * 64];
                uint8_t* wh = &where[(size_t)(g - g0) * 64];
                const uint8_t* tb = &tab[((size_t)g * nb + b) * 64];
                const int32_t* gs = &gsp[((size_t)g * nb + b) * stride];
                for (int r = 0; r < NG; ++r) {
                    int32_t at = gs[r];
                    for (int t = 0; t < QM; ++t) {
                        const int q = tb[r * QM + t];
                        if (q == 0xFF) continue;
                        wh[q] = (uint8_t)((r << 3) | t);
                        const int32_t c = q < nc ? pc[q] : 0;
                        pc[q] = at;
                        at += c;
                    }
                }
            }
            for (int q = 0; q < nc; ++q)
                for (int64_t ii = lo_q[(size_t)q]; ii < hi_q[(size_t)q]; ++ii) {
                    const int64_t i = cidx[ii];
                    if (skip && skip[ii]) {  // on a conflict row of a relaxed run: not in the
                        last[(size_t)i] = b;  // stream, but the step does touch the row
                        continue;
                    }
                    const size_t gi = (size_t)(i / rows_per);
                    const size_t lq = (gi - (size_t)g0) * 64 + (size_t)q;
                    const size_t e = (size_t)pos[lq]++;
                    src[e] = (int32_t)ii;
                    meta[e] = (uint8_t)((where[lq] & 7) | (last[(size_t)i] == b - 1 ? 0x80 : 0));
                    last[(size_t)i] = b;  // columns of one step share no row: no read-after-write
                }
        }
    });
}

// Entry stream of the WIDE persistent passes (steps of up to 512 columns, one thread per
// column slot): entries sorted by (row block g, step b, slot q, row).  wbase[b] = sum over
// earlier steps of (ncols + 1); wsp[g * tot + wbase[b] + q] = first entry of slot q of step b in
// row block g (ncols + 1 boundaries per step; tot = d + nb).  src[e] = position in the CSC
// arrays; hz[e] = 1 if the entry's row was touched by the previous step.
void build_wide_stream(int64_t n, const int64_t* cptr, const int32_t* cidx,
                       const std::vector<int32_t>& order, const std::vector<int32_t>& batch_ptr,
                       int G, std::vector<int32_t>& wbase, std::vector<int32_t>& wsp,
                       std::vector<int32_t>& src, std::vector<uint8_t>& hz) {
    const int nb = (int)batch_ptr.size() - 1;
    const int64_t rows_per = (n + G - 1) / G > 0 ? (n + G - 1) / G : 1;
    wbase.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb; ++b) wbase[(size_t)b + 1] = wbase[(size_t)b] + (batch_ptr[b + 1] - batch_ptr[b]) + 1;
    const size_t tot = (size_t)wbase[(size_t)nb];
    wsp.assign((size_t)G * tot + 1, 0);
    const int64_t nnz = cptr[order.size()];
    const int T = std::min(G, (nnz >= (1 << 20)) ? schedule_threads() : 1);
    const int gper = (G + T - 1) / T;
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = (int64_t)g1 * rows_per;
        for (int b = 0; b < nb; ++b)
            for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
                const int32_t j = order[(size_t)batch_ptr[b] + q];
                for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                    const int64_t i = cidx[ii];
                    if (i < rlo || i >= rhi) continue;
                    wsp[(size_t)(i / rows_per) * tot + (size_t)wbase[(size_t)b] + (size_t)q]++;
                }
            }
    });
    int64_t run = 0;
    for (size_t t = 0; t < wsp.size(); ++t) {  // exclusive prefix sum; each step's extra word
        const int64_t c = wsp[t];              // counts 0 entries = end of its last slot
        wsp[t] = (int32_t)run;
        run += c;
    }
    src.resize((size_t)run);
    hz.resize((size_t)run);
    std::vector<int32_t> fill(wsp.begin(), wsp.end());
    std::vector<int32_t> last((size_t)n, -2);
    run_threads(T, [&](int tid) {
        const int g0 = gper * tid, g1 = std::min(G, g0 + gper);
        const int64_t rlo = (int64_t)g0 * rows_per, rhi = (int64_t)g1 * rows_per;
        for (int b = 0; b < nb; ++b)
            for (int q = 0; q < batch_ptr[b + 1] - batch_ptr[b]; ++q) {
                const int32_t j = order[(size_t)batch_ptr[b] + q];
                for (int64_t ii = cptr[j]; ii < cptr[j + 1]; ++ii) {
                    const int64_t i = cidx[ii];
                    if (i < rlo || i >= rhi) continue;
                    const size_t e = (size_t)fill[(size_t)(i / rows_per) * tot +
                                                  (size_t)wbase[(size_t)b] + (size_t)q]++;
                    src[e] = (int32_t)ii;
                    hz[e] = (uint8_t)(last[(size_t)i] == b - 1 ? 1 : 0);
                    last[(size_t)i] = b;
                }
            }
    });
}

}  // namespace spfm
