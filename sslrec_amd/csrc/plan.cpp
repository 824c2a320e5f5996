// Native plan builder: COO / CSR of the propagation matrix -> the device layouts of spmm.hip (streamed, packed
// CSR) and spmm_swept.hip (column-swept, LDS accumulators), built on the host in C++ and uploaded with plain HIP
// calls, so that a C caller needs nothing but this library (include/sslrec_hip.h, "native plan builder").
//
// What it replaces: the reference re-coalesces its uncoalesced COO adjacency inside EVERY torch.spmm call
// (models/general_cf/lightgcn.py:28-29 on the tensor built at data_utils/data_handler_general_cf.py:70-73); here the
// conversion happens once.  Every choice below (chunk caps, longest-processing-time-first dealing with
// (load, id) tie-breaking, stable sorts) is deterministic, so a layout is a pure function of the matrix and d.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sslrec_hip.h"

namespace {

typedef std::pair<int64_t, int> LoadId;      // (load, target): the pair with the smallest load, then id, is served next
typedef std::priority_queue<LoadId, std::vector<LoadId>, std::greater<LoadId>> MinHeap;

constexpr int kSweptBlocks = 256;            // one workgroup per CU
constexpr int kSweptWaves = 16;              // 1024-thread workgroups
constexpr int kMaxStreams = 256 * 20;        // streamed kernel: one stream per resident wavefront
constexpr int kMinStream = 64;
constexpr int kRowOverhead = 6;              // cost of finishing a row segment, in entry-equivalents
constexpr int kSweptSlotCost = 1;            // swept layout: what zeroing + flushing one accumulator slot costs its owner, in entry-equivalents

struct HostArray {
    std::shared_ptr<void> keep;              // the std::vector<T> itself, moved in (no copy: these arrays are hundreds of MB)
    const void *data = nullptr;
    size_t elem = 4;
    int64_t count = 0;
    void *dev = nullptr;
    template <class T> void set(std::vector<T> &v) {
        auto held = std::make_shared<std::vector<T>>(std::move(v));
        elem = sizeof(T);
        count = (int64_t)held->size();
        if (held->empty()) held->resize(1);  // callers may hand out the pointer of an empty array
        data = held->data();
        keep = held;
    }
    size_t bytes() const { return (size_t)std::max<int64_t>(count, 1) * elem; }
};

struct Layout {
    int kind = 0;                            // SSLREC_PLAN_SWEPT / SSLREC_PLAN_STREAMED
    int d = 0;
    bool xcd_split = false;
    int64_t xcd_col_pairs = 0;               // swept, XCD split: sum over the 8 XCDs of the distinct columns their rows reference
    std::map<std::string, HostArray> arrays;
    sslrec_swept_t swept = {};
    sslrec_csr_t csr = {};
    sslrec_bundled_t bundled = {};
    float *partial_ws = nullptr;
    bool uploaded = false;
};

}  // namespace

struct sslrec_plan {
    int32_t n_rows = 0, n_cols = 0;
    int64_t nnz = 0;
    std::vector<int64_t> rowptr;             // [n_rows+1]
    std::vector<int32_t> col;                // CSR order: by row, then column (stable: duplicates keep their input order)
    std::vector<float> val;
    std::vector<int64_t> perm;               // CSR position -> input entry
    std::map<int, std::unique_ptr<Layout>> layouts;   // by d * 4 + kind
    int64_t seg_max = 0;                     // chunk cap of the streamed layout's long rows (0 = automatic)
    int64_t n_streams = 0;                   // work streams of the streamed layout (0 = automatic)
    int64_t swept_blocks = 0;                // workgroups of the swept layout: 256 (one per CU, default) or 512 (two per CU)
    int64_t xcd_balance = 0;                 // XCD split: per mille of the entries on XCDs 0-3 (0 = 500)
    mutable int cluster_pays = -1;           // automatic co-clustering: -1 = not tried yet on this matrix, 0 = the clustered dealing did not lower the
                                             // (XCD, column) pairs by a quarter -- later layouts of this plan (other widths) skip the clustering (ADVICE r04:
                                             // 1.7 s per layout at amazon-book size for a dealing that is thrown away), 1 = it did
    int64_t xcd_stagger = 0;                 // XCD split, experiment (EXPERIMENTS.md C.2): XCD k of a class gets k * xcd_stagger per mille of a mean workgroup's
                                             // entries LESS than XCD 0, so the XCDs end their sweeps -- and start their flushes -- one after the other
    int64_t xcd_cluster = -1;                // XCD split, row -> XCD co-clustering: 0 = never (rows dealt to the 4 XCDs of a class by load only),
                                             // n > 0 = always, with n refinement passes; -1 (default) = automatic: built with 4 passes and KEPT only
                                             // when it lowers the layout's distinct (XCD, column) pairs by more than a quarter (a graph with
                                             // communities: real yelp -43 %, launch -11 %; the structure-free headline graph -8 %, launch +9 %:
                                             // the co-clustered dealing balances the streams a little worse, so it has to pay for that)
    int64_t swept_passes = 1;                // allow a swept layout of d/2, d/4, ... columns run in passes (0: never)
    int64_t swept_width = 0;                 // widest swept layout to build (0: the tables' d); tests force passes with it
    int64_t bundled32 = 0;                   // streamed kind at d = 32: 1 = the row-bundled layout instead of the packed one
};

namespace {

// stable counting sort of `idx` by key[idx[i]] in [0, n_keys)
void counting_pass(const std::vector<int64_t> &idx_in, std::vector<int64_t> &idx_out, const std::function<int64_t(int64_t)> &key,
                   int64_t n_keys) {
    std::vector<int64_t> cnt((size_t)n_keys + 1, 0);
    for (int64_t e : idx_in) ++cnt[(size_t)key(e) + 1];
    for (int64_t k = 0; k < n_keys; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
    idx_out.resize(idx_in.size());
    for (int64_t e : idx_in) idx_out[(size_t)cnt[(size_t)key(e)]++] = e;
}

int bipartite_split_of(const sslrec_plan &p) {
    // s such that rows < s hold only columns >= s and rows >= s only columns < s (the (U+I)^2 adjacency of
    // data_handler_general_cf.py:37-73 has s = U); -1 otherwise
    if (p.n_rows != p.n_cols || p.nnz == 0) return -1;
    std::vector<int> cmin((size_t)p.n_rows, p.n_cols), cmax((size_t)p.n_rows, -1);
    for (int r = 0; r < p.n_rows; ++r)
        for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) {
            cmin[r] = std::min(cmin[r], (int)p.col[(size_t)e]);
            cmax[r] = std::max(cmax[r], (int)p.col[(size_t)e]);
        }
    int s = -1;
    for (int r = 0; r < p.n_rows && s < 0; ++r)
        if (cmin[r] < r) s = r;
    if (s <= 0) return -1;
    for (int r = 0; r < p.n_rows; ++r) {
        if (cmax[r] < 0) continue;
        if (r < s ? cmin[r] < s : cmax[r] >= s) return -1;
    }
    return s;
}

// Row -> XCD co-clustering of ONE row class of a bipartite adjacency (option "xcd_cluster" = refinement passes): the 4 XCDs of a
// class each pull every embedding row their output rows reference through their own L2, so the fabric traffic of a launch is ~ the
// sum over XCDs of the DISTINCT columns referenced there.  With rows dealt by load only that is nearly 4 x the table whatever the
// graph looks like; rows that share columns on the same XCD lower it when the graph has communities.  Three steps, deterministic, a
// pure function of the matrix:
//   1. size-constrained label propagation (Meyerhenke et al.'s SCLaP, on the bipartite graph): every row starts as its own cluster;
//      a column takes the most frequent cluster among its rows, a row the most frequent cluster among its columns that still has
//      room (a cluster holds at most 1 / (4 K) of the class's entries); a few sweeps;
//   2. the clusters are packed into the K groups longest-first by entries (LPT) under the groups' accumulator-slot capacity;
//   3. `passes` sweeps of restreaming refinement (linear deterministic greedy): a row moves to the group that references most of its
//      columns, weighted by the group's remaining entry budget -- 12 % of slack in the sweeps before the last, +2 % in the last.
// Summation order inside a row does not depend on where the row runs: results are bit-identical, only the traffic changes.
// rows_desc: the class's rows, descending degree; sets group[r] in [0, K) for them.
void cocluster_rows(const sslrec_plan &p, const std::vector<int> &rows_desc, const std::vector<int64_t> &deg, const std::vector<int64_t> &nch,
                    int K, int passes, int64_t slot_cap_group, std::vector<int> &group) {
    const int n_cls = (int)rows_desc.size();
    if (n_cls == 0) return;
    int64_t tot_e = 0;
    for (int r : rows_desc) tot_e += deg[r];
    // ---- 1. size-constrained label propagation ---------------------------------------------------------------------------------
    // column -> its rows of this class (transposed lists, counting sort)
    std::vector<int64_t> cptr((size_t)p.n_cols + 1, 0);
    for (int r : rows_desc)
        for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) ++cptr[(size_t)p.col[(size_t)e] + 1];
    for (int c = 0; c < p.n_cols; ++c) cptr[(size_t)c + 1] += cptr[(size_t)c];
    std::vector<int> crow((size_t)cptr[(size_t)p.n_cols]);
    {
        std::vector<int64_t> at(cptr.begin(), cptr.end() - 1);
        std::vector<int> asc(rows_desc);
        std::sort(asc.begin(), asc.end());
        for (int r : asc)
            for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) crow[(size_t)at[(size_t)p.col[(size_t)e]]++] = r;
    }
    std::vector<int> lab_r((size_t)p.n_rows, -1), lab_c((size_t)p.n_cols, -1);
    std::vector<int64_t> size_e((size_t)p.n_rows, 0);                    // entries of cluster l (labels are row ids)
    for (int r : rows_desc) { lab_r[r] = r; size_e[r] = deg[r]; }
    const int64_t cap_fine = std::max<int64_t>(tot_e / (4 * K), 1);
    std::vector<int> tmp;
    auto plurality = [&](std::vector<int> &labels) -> int {              // most frequent label, ties to the smaller one; -1 if none
        if (labels.empty()) return -1;
        std::sort(labels.begin(), labels.end());
        int best = labels[0], best_n = 0, run = 0;
        for (size_t i = 0; i < labels.size(); ++i) {
            run = (i > 0 && labels[i] == labels[i - 1]) ? run + 1 : 1;
            if (run > best_n) { best_n = run; best = labels[i]; }
        }
        return best;
    };
    for (int sweep = 0; sweep < 6; ++sweep) {
        for (int c = 0; c < p.n_cols; ++c) {
            if (cptr[(size_t)c] == cptr[(size_t)c + 1]) continue;
            tmp.clear();
            for (int64_t i = cptr[(size_t)c]; i < cptr[(size_t)c + 1]; ++i) tmp.push_back(lab_r[crow[(size_t)i]]);
            lab_c[c] = plurality(tmp);
        }
        for (auto it = rows_desc.rbegin(); it != rows_desc.rend(); ++it) {      // light rows first: hubs join what has formed
            const int r = *it;
            if (deg[r] == 0) continue;
            tmp.clear();
            for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) tmp.push_back(lab_c[(size_t)p.col[(size_t)e]]);
            std::sort(tmp.begin(), tmp.end());
            const int cur = lab_r[r];
            int best = cur, best_n = 0;
            for (size_t i = 0; i < tmp.size();) {
                size_t j = i;
                while (j < tmp.size() && tmp[j] == tmp[i]) ++j;
                const int l = tmp[i], cnt_l = (int)(j - i);
                if (l >= 0 && (l == cur || size_e[(size_t)l] + deg[r] <= cap_fine) && (cnt_l > best_n || (cnt_l == best_n && l < best))) { best_n = cnt_l; best = l; }
                i = j;
            }
            if (best != cur) { size_e[(size_t)cur] -= deg[r]; size_e[(size_t)best] += deg[r]; lab_r[r] = best; }
        }
    }
    // ---- 2. clusters -> K groups, longest first ----------------------------------------------------------------------------------
    std::vector<int64_t> size_s((size_t)p.n_rows, 0);
    for (int r : rows_desc) size_s[(size_t)lab_r[r]] += nch[r];
    std::vector<int> labels;
    for (int r : rows_desc) if (lab_r[r] == r || size_e[(size_t)r] > 0) { if (size_s[(size_t)r] > 0) labels.push_back(r); }
    std::sort(labels.begin(), labels.end());
    labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
    std::stable_sort(labels.begin(), labels.end(), [&](int a, int b) { return size_e[(size_t)a] > size_e[(size_t)b]; });
    std::vector<int64_t> load_e(K, 0), load_s(K, 0);
    std::vector<int> group_of_label((size_t)p.n_rows, 0);
    for (int l : labels) {
        int best = -1;
        for (int k = 0; k < K; ++k)
            if (load_s[k] + size_s[(size_t)l] <= slot_cap_group && (best < 0 || load_e[k] < load_e[best])) best = k;
        if (best < 0) { best = 0; for (int k = 1; k < K; ++k) if (load_s[k] < load_s[best]) best = k; }
        group_of_label[(size_t)l] = best;
        load_e[best] += size_e[(size_t)l]; load_s[best] += size_s[(size_t)l];
    }
    std::vector<int32_t> cnt((size_t)p.n_cols * K, 0);       // entries of group k at column c
    for (int r : rows_desc) {
        const int g = group_of_label[(size_t)lab_r[r]];
        group[r] = g;
        for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) ++cnt[(size_t)p.col[(size_t)e] * K + g];
    }
    // ---- 3. restreaming refinement -------------------------------------------------------------------------------------------------
    const int n_pass = std::max(1, passes);
    for (int pass = 0; pass < n_pass; ++pass) {
        const double cap_e = (pass == n_pass - 1 ? 1.02 : 1.12) * (double)tot_e / K + 1.0;
        for (int r : rows_desc) {
            const int64_t b0 = p.rowptr[r];
            {
                const int g = group[r];
                for (int64_t j = 0; j < deg[r]; ++j) --cnt[(size_t)p.col[(size_t)(b0 + j)] * K + g];
                load_e[g] -= deg[r]; load_s[g] -= nch[r];
            }
            int64_t share[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t j = 0; j < deg[r]; ++j) {
                const int32_t *c = &cnt[(size_t)p.col[(size_t)(b0 + j)] * K];
                for (int k = 0; k < K; ++k) share[k] += c[k] > 0;
            }
            int best = -1;
            double best_score = -1.0;
            for (int k = 0; k < K; ++k) {
                if (load_s[k] + nch[r] > slot_cap_group || (double)(load_e[k] + deg[r]) > cap_e) continue;
                const double room = 1.0 - (double)load_e[k] / cap_e;
                const double score = ((double)share[k] + 1e-3) * room;      // (no shared column yet: the emptiest group)
                if (score > best_score) { best_score = score; best = k; }
            }
            if (best < 0) {                                   // every group at a cap: the one with most slot room takes it
                best = 0;
                for (int k = 1; k < K; ++k) if (load_s[k] < load_s[best]) best = k;
            }
            group[r] = best;
            for (int64_t j = 0; j < deg[r]; ++j) ++cnt[(size_t)p.col[(size_t)(b0 + j)] * K + best];
            load_e[best] += deg[r]; load_s[best] += nch[r];
        }
    }
}

// ---- column-swept layout (mirror of the kernel contract in spmm_swept.hip / sslrec_swept_t) ---------------------
int build_swept(const sslrec_plan &p, int d, int flags, Layout &L, std::string &why) {
    // one 1024-thread workgroup per CU owning all of its LDS; small matrices (a few dozen steps per stream: launch, LDS
    // zeroing and flush dominate) run two half-size workgroups per CU instead -- measured on the real yelp graph
    // (0.36 M entries): 19.8 -> 16.8 us; on the amazon-book-shaped graph (4.8 M) two per CU lose (82 -> 122 us)
    const int G = 256 / d, nw = kSweptWaves, gpb = nw * G;
    const int nb = p.swept_blocks > 0 ? (int)p.swept_blocks : (p.nnz <= 1000000 ? 2 * kSweptBlocks : kSweptBlocks);
    // (narrow rows, d = 16 / 8: a packed word has 12 slot bits, so a workgroup uses at most 4095 of the slots its LDS could hold)
    const int slot_cap = std::min(4095, SSLREC_SWEPT_LDS_BYTES / (nb / kSweptBlocks) / (d * 4));
    const int n = p.n_rows;
    const int64_t nnz = p.nnz;
    if (!((double)n <= 0.985 * (double)nb * slot_cap && p.n_cols <= (1 << 20)) || nnz == 0) {
        why = "output table does not fit the chip's LDS";
        return 1;
    }
    std::vector<int64_t> deg(n);
    int64_t deg_max = 0;
    for (int r = 0; r < n; ++r) { deg[r] = p.rowptr[r + 1] - p.rowptr[r]; deg_max = std::max(deg_max, deg[r]); }
    // heavy rows are cut into INTERLEAVED chunks (entry j -> chunk j % n_chunks), one accumulator slot each
    std::vector<int64_t> nch(n);
    int64_t chunk_cap = 0;
    bool ok = false;
    for (double factor : {0.4, 0.6, 1.0, 2.0, 4.0, 16.0, 1e9}) {
        chunk_cap = std::max<int64_t>(16, (int64_t)(factor * (double)nnz / (double)(nb * gpb)));
        int64_t total = 0, most = 0;
        for (int r = 0; r < n; ++r) {
            nch[r] = std::max<int64_t>(1, (deg[r] + chunk_cap - 1) / chunk_cap);
            total += nch[r];
            most = std::max(most, nch[r]);
        }
        if ((double)total <= 0.985 * nb * slot_cap && most <= slot_cap / 2) { ok = true; break; }
    }
    if (!ok) { why = "accumulator slots do not fit"; return 1; }
    if (n && deg_max > std::max<int64_t>(8192, 4 * (nnz / nb))) { why = "dominated by a giant row"; return 1; }

    std::vector<int> by_deg(n);
    for (int r = 0; r < n; ++r) by_deg[r] = r;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] > deg[b]; });

    // XCD split of a bipartite adjacency: workgroup b runs on XCD b % 8; XCDs 0-3 accumulate one row class, XCDs 4-7
    // the other, so an XCD's L2 sweeps ONE embedding table; the class that overflows its half hands over its coldest rows
    std::vector<char> in_b(n, 0);
    bool split = false;
    if (!(flags & SSLREC_PLAN_NO_XCD_SPLIT) && nb % 8 == 0) {
        const int s = bipartite_split_of(p);
        if (s > 0) {
            const int half_blocks = nb / 2;
            const int64_t cap = (int64_t)(0.97 * half_blocks * slot_cap);
            for (int r = 0; r < n; ++r) in_b[r] = r >= s;
            for (int c = 0; c < 2; ++c) {
                int64_t need = 0;
                for (int r = 0; r < n; ++r) if ((r >= s) == (c == 1)) need += nch[r];
                if (need <= cap) continue;
                std::vector<int> ids;
                for (int r = 0; r < n; ++r) if ((r >= s) == (c == 1)) ids.push_back(r);
                std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return deg[a] < deg[b]; });     // coldest first
                int64_t run = 0;
                size_t k = 0;
                while (k < ids.size() && run + nch[ids[k]] <= need - cap) run += nch[ids[k++]];
                for (size_t j = 0; j <= k && j < ids.size(); ++j) in_b[ids[j]] = (c == 0);
            }
            int64_t sa = 0, sb = 0, ea = 0, eb = 0;
            for (int r = 0; r < n; ++r) { if (in_b[r]) { sb += nch[r]; eb += deg[r]; } else { sa += nch[r]; ea += deg[r]; } }
            // balance: the half with more entries hands its HOTTEST rows over (they bring many entries per slot), the
            // receiving half makes room by returning its coldest rows; `xcd_balance` = share of the entries (per mille)
            // that XCDs 0-3 should hold
            {
                const double want_a = (p.xcd_balance > 0 ? (double)p.xcd_balance : 500.0) / 1000.0 * (double)(ea + eb);
                const bool from_a = (double)ea > want_a;
                std::vector<int> hot, cold;                       // of the giving half (hottest first) / the receiving half (coldest first)
                for (int r : by_deg) if ((in_b[r] != 0) != from_a) hot.push_back(r);
                for (auto it = by_deg.rbegin(); it != by_deg.rend(); ++it) if ((in_b[*it] != 0) == from_a) cold.push_back(*it);
                int64_t &e_give = from_a ? ea : eb, &e_take = from_a ? eb : ea, &s_give = from_a ? sa : sb, &s_take = from_a ? sb : sa;
                const double want_give = from_a ? want_a : (double)(ea + eb) - want_a;
                size_t ic = 0;
                for (size_t ih = 0; ih < hot.size() && (double)e_give - (double)deg[hot[ih]] >= want_give; ++ih) {
                    const int r = hot[ih];
                    while (s_take + nch[r] > cap && ic < cold.size()) {          // make room in the receiving half
                        const int c = cold[ic++];
                        if (deg[c] >= deg[r]) { ic = cold.size(); break; }
                        in_b[c] = from_a ? 0 : 1;
                        s_take -= nch[c]; s_give += nch[c]; e_take -= deg[c]; e_give += deg[c];
                    }
                    if (s_take + nch[r] > cap) break;
                    in_b[r] = from_a ? 1 : 0;
                    s_give -= nch[r]; s_take += nch[r]; e_give -= deg[r]; e_take += deg[r];
                }
            }
            split = (double)sa <= half_blocks * slot_cap * 0.985 && (double)sb <= half_blocks * slot_cap * 0.985 &&
                    (double)std::max(ea, eb) <= 1.15 * (double)(ea + eb) / 2.0;
        }
    }
    // rows -> blocks: longest-processing-time-first on entries, at most slot_cap slots per block
    std::vector<int64_t> used(nb, 0);
    std::vector<int> blk_of_row(n);
    auto lpt = [&](const std::vector<int> &rows_desc, const std::vector<int> &blocks) -> bool {
        MinHeap heap;
        int64_t head = 0;                      // xcd_stagger: a workgroup of XCD k starts with k steps of phantom load
        if (split && p.xcd_stagger > 0 && !blocks.empty()) {
            int64_t tot = 0;
            for (int r : rows_desc) tot += deg[r];
            head = tot / (int64_t)blocks.size() * p.xcd_stagger / 1000;
        }
        for (int b : blocks) heap.push({head * ((b % 8) % 4), b});
        std::vector<LoadId> parked;
        for (int r : rows_desc) {
            parked.clear();
            LoadId top;
            while (true) {
                if (heap.empty()) return false;
                top = heap.top();
                heap.pop();
                if (used[top.second] + nch[r] <= slot_cap) break;
                parked.push_back(top);
            }
            const int b = top.second;
            blk_of_row[r] = b;
            used[b] += nch[r];
            if (used[b] < slot_cap) heap.push({top.first + deg[r] + kSweptSlotCost * nch[r], b});      // (rows without entries still cost their flush:
                                                                                                         // without the term they all land in ONE workgroup)
            for (const LoadId &it : parked) heap.push(it);
        }
        return true;
    };
    bool placed;
    if (!split) {
        std::vector<int> all(nb);
        for (int b = 0; b < nb; ++b) all[b] = b;
        placed = lpt(by_deg, all);
    } else {
        std::vector<int> ra, rb, ba, bb;
        for (int r : by_deg) (in_b[r] ? rb : ra).push_back(r);
        for (int b = 0; b < nb; ++b) (b % 8 < 4 ? ba : bb).push_back(b);
        placed = lpt(ra, ba) && lpt(rb, bb);
        // (automatic mode: the answer is a property of the matrix, not of the width -- a matrix on which it did not pay is not clustered again)
        if (placed && p.xcd_cluster != 0 && !(p.xcd_cluster < 0 && p.cluster_pays == 0)) {      // rows of a class -> its 4 XCDs by shared columns (cocluster_rows), then by load inside an XCD
            auto pairs_of = [&](const std::function<int(int)> &xcd_of_row) {
                std::vector<unsigned char> seen((size_t)p.n_cols, 0);
                for (int r = 0; r < n; ++r) {
                    const unsigned char bit = (unsigned char)(1u << xcd_of_row(r));
                    for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) seen[(size_t)p.col[(size_t)e]] |= bit;
                }
                int64_t t = 0;
                for (unsigned char m : seen) t += __builtin_popcount(m);
                return t;
            };
            const int64_t pairs_plain = pairs_of([&](int r) { return blk_of_row[r] % 8; });
            std::vector<int> grp(n, 0);
            const int64_t cap_g = (int64_t)(0.985 * (nb / 8) * slot_cap);
            const int passes = p.xcd_cluster > 0 ? (int)p.xcd_cluster : 4;
            bool clustered = true;             // false: a clustering ran out of memory (or threw anything else) -- the dealing by load stays
            {   // the two row classes are independent (disjoint rows of grp, everything else read-only): one on a second thread.  Neither
                // an exception on this thread while the other is joinable (std::terminate in ~thread) nor one inside the worker
                // (std::terminate at once) may leave: both bodies catch everything, the worker is always joined.
                std::thread other;
                bool threaded = true, ok_b = true;
                auto run_b = [&] { try { cocluster_rows(p, rb, deg, nch, 4, passes, cap_g, grp); } catch (...) { ok_b = false; } };
                try { other = std::thread(run_b); }
                catch (...) { threaded = false; }                  // (no thread to be had: one after the other)
                try { cocluster_rows(p, ra, deg, nch, 4, passes, cap_g, grp); } catch (...) { clustered = false; }
                if (threaded) other.join(); else run_b();
                clustered = clustered && ok_b;
            }
            const int64_t pairs_cl = clustered ? pairs_of([&](int r) { return (in_b[r] ? 4 : 0) + grp[r]; }) : pairs_plain;
            if (clustered && p.xcd_cluster < 0) p.cluster_pays = (4 * pairs_cl < 3 * pairs_plain) ? 1 : 0;
            if (clustered && (p.xcd_cluster > 0 || 4 * pairs_cl < 3 * pairs_plain)) {
                const std::vector<int64_t> used_plain(used);
                const std::vector<int> blk_plain(blk_of_row);
                std::fill(used.begin(), used.end(), 0);
                bool ok_cl = true;
                for (int k = 0; k < 8 && ok_cl; ++k) {
                    std::vector<int> rk, bk;
                    for (int r : (k < 4 ? ra : rb)) if (grp[r] == k % 4) rk.push_back(r);
                    for (int b = 0; b < nb; ++b) if (b % 8 == k) bk.push_back(b);
                    ok_cl = lpt(rk, bk);
                }
                if (!ok_cl) { used = used_plain; blk_of_row = blk_plain; }      // a group overflowed its workgroups: the dealing by load stays
            }
        }
    }
    if (!placed) { why = "rows do not fit their workgroups"; return 1; }
    int64_t xcd_pairs = 0;                                            // distinct (XCD, column) pairs of the layout
    if (split) {
        std::vector<unsigned char> seen((size_t)p.n_cols, 0);
        for (int r = 0; r < n; ++r) {
            const unsigned char bit = (unsigned char)(1u << (blk_of_row[r] % 8));
            for (int64_t e = p.rowptr[r]; e < p.rowptr[r + 1]; ++e) seen[(size_t)p.col[(size_t)e]] |= bit;
        }
        for (unsigned char m : seen) xcd_pairs += __builtin_popcount(m);
    }

    // slots: a block's rows in row order, the chunks of a row contiguous (the flush adds them in order)
    std::vector<int> row_order(n);
    for (int r = 0; r < n; ++r) row_order[r] = r;
    std::stable_sort(row_order.begin(), row_order.end(), [&](int a, int b) { return blk_of_row[a] < blk_of_row[b]; });
    std::vector<int64_t> slot_start(n);
    {
        int i = 0;
        for (int b = 0; b < nb; ++b) {
            int64_t s = 0;
            while (i < n && blk_of_row[row_order[i]] == b) {
                const int r = row_order[i];
                slot_start[r] = s;
                s += nch[r];
                ++i;
            }
        }
    }
    // chunks -> lane groups of their block, LPT again (longest chunk first, ties by chunk id)
    std::vector<int64_t> v_first(n + 1, 0);
    for (int r = 0; r < n; ++r) v_first[r + 1] = v_first[r] + nch[r];
    const int64_t n_v = v_first[n];
    std::vector<int> v_row((size_t)n_v);
    std::vector<int64_t> v_len((size_t)n_v);
    for (int r = 0; r < n; ++r)
        for (int64_t k = 0; k < nch[r]; ++k) {
            v_row[(size_t)(v_first[r] + k)] = r;
            v_len[(size_t)(v_first[r] + k)] = deg[r] / nch[r] + (k < deg[r] % nch[r]);
        }
    std::vector<int64_t> order_v((size_t)n_v);
    for (int64_t v = 0; v < n_v; ++v) order_v[(size_t)v] = v;
    std::stable_sort(order_v.begin(), order_v.end(), [&](int64_t a, int64_t b) {
        const int ba = blk_of_row[v_row[(size_t)a]], bb = blk_of_row[v_row[(size_t)b]];
        if (ba != bb) return ba < bb;
        return v_len[(size_t)a] > v_len[(size_t)b];
    });
    std::vector<int> v_grp((size_t)n_v);
    {
        size_t i = 0;
        for (int b = 0; b < nb; ++b) {
            MinHeap h;
            for (int g = 0; g < gpb; ++g) h.push({0, g});
            while (i < order_v.size() && blk_of_row[v_row[(size_t)order_v[i]]] == b) {
                const int64_t v = order_v[i++];
                LoadId top = h.top();
                h.pop();
                v_grp[(size_t)v] = top.second;
                h.push({top.first + v_len[(size_t)v] + kSweptSlotCost, top.second});      // a wave flushes the one-slot rows of its lane groups itself
            }
        }
    }
    // flush lists.  A row with ONE slot is owned by one lane group, hence by one wave: that wave writes the row out as soon as
    // ITS sweep is over (no barrier, the stores overlap the slower waves' gathers) -- records [wf_ptr[w], wf_ptr[w+1]) of wave
    // w = block * 16 + wave, in row order.  Rows cut into chunks are spread over lane groups: the block adds their chunks after
    // its barrier -- records [cf_ptr[b], cf_ptr[b+1]), stored behind the wave-owned ones.
    const int n_streams_f = nb * nw;
    std::vector<int32_t> wf_ptr(n_streams_f + 1, 0), cf_ptr(nb + 1, 0), f_row(n), f_start(n), f_n(n);
    {
        std::vector<int> wave_of(n, -1);
        int64_t n_single = 0;
        for (int r = 0; r < n; ++r)
            if (nch[r] == 1) { wave_of[r] = blk_of_row[r] * nw + v_grp[(size_t)v_first[r]] / G; ++wf_ptr[wave_of[r] + 1]; ++n_single; }
        for (int w = 0; w < n_streams_f; ++w) wf_ptr[w + 1] += wf_ptr[w];
        std::vector<int32_t> at(wf_ptr.begin(), wf_ptr.end() - 1);
        for (int r = 0; r < n; ++r) {                                  // ascending row id inside a wave
            if (wave_of[r] < 0) continue;
            const int i = at[wave_of[r]]++;
            f_row[i] = r; f_start[i] = (int32_t)slot_start[r]; f_n[i] = 1;
        }
        int i = (int)n_single, k = 0;
        for (int b = 0; b < nb; ++b) {
            cf_ptr[b] = i;
            while (k < n && blk_of_row[row_order[k]] == b) {
                const int r = row_order[k++];
                if (nch[r] == 1) continue;
                f_row[i] = r; f_start[i] = (int32_t)slot_start[r]; f_n[i] = (int32_t)nch[r];
                ++i;
            }
        }
        cf_ptr[nb] = i;
    }
    // entries (CSR order: by row, then column) -> (lane group, slot); a lane group's stream is sorted by column
    std::vector<int32_t> e_gid((size_t)nnz), e_slot((size_t)nnz);
    for (int r = 0; r < n; ++r) {
        const int64_t b0 = p.rowptr[r];
        for (int64_t j = 0; j < deg[r]; ++j) {
            const int64_t c = j % nch[r];
            e_slot[(size_t)(b0 + j)] = (int32_t)(slot_start[r] + c);
            e_gid[(size_t)(b0 + j)] = blk_of_row[r] * gpb + v_grp[(size_t)(v_first[r] + c)];
        }
    }
    std::vector<int64_t> idx((size_t)nnz), o;
    for (int64_t e = 0; e < nnz; ++e) idx[(size_t)e] = e;
    counting_pass(idx, o, [&](int64_t e) { return (int64_t)p.col[(size_t)e]; }, p.n_cols);
    counting_pass(o, idx, [&](int64_t e) { return (int64_t)e_gid[(size_t)e]; }, (int64_t)nb * gpb);
    o.swap(idx);                                               // o: entries sorted by (lane group, column), stable
    std::vector<int64_t> g_len((size_t)nb * gpb, 0);
    for (int64_t e = 0; e < nnz; ++e) ++g_len[(size_t)e_gid[(size_t)e]];
    // a wave's stream: 64-dword blocks of SB steps; the entry of (step j of the block, lane group g) sits at dword g*LPG + j
    // of the block, once per 16-lane row of the lane group (the kernel broadcasts it from there, spmm_swept.hip)
    const int LPG = 64 / G, SB = std::min(16, LPG), copies = std::max(1, LPG / 16);
    const int n_streams = nb * nw;
    std::vector<int32_t> w_steps(n_streams), w_start(n_streams);
    int64_t n_elem = 0;
    for (int w = 0; w < n_streams; ++w) {
        int64_t longest = 0;
        for (int g = 0; g < G; ++g) longest = std::max(longest, g_len[(size_t)w * G + g]);
        const int64_t steps = (longest + SB - 1) / SB * SB;
        if (n_elem >= 2147483647LL - steps / SB * 64) { why = "swept layout exceeds int32 indexing"; return 1; }
        w_start[w] = (int32_t)n_elem;
        w_steps[w] = (int32_t)steps;
        n_elem += steps / SB * 64;
    }
    std::vector<int32_t> pack((size_t)std::max<int64_t>(n_elem, 1), -1), emap((size_t)std::max<int64_t>(n_elem, 1), -1);
    std::vector<float> val((size_t)std::max<int64_t>(n_elem, 1), 0.f);
    std::vector<int64_t> elem_of((size_t)nnz);
    {
        std::vector<int64_t> seen((size_t)nb * gpb, 0);
        for (int64_t i = 0; i < nnz; ++i) {
            const int64_t e = o[(size_t)i];
            const int gid = e_gid[(size_t)e];
            const int64_t s = seen[(size_t)gid]++;
            const int64_t at = (int64_t)w_start[gid / G] + (s / SB) * 64 + (gid % G) * LPG + (s % SB);
            for (int c = 0; c < copies; ++c) {
                pack[(size_t)at + c * 16] = (int32_t)((uint32_t)p.col[(size_t)e] | ((uint32_t)e_slot[(size_t)e] << 20));
                val[(size_t)at + c * 16] = p.val[(size_t)e];
                emap[(size_t)at + c * 16] = (int32_t)p.perm[(size_t)e];
            }
            elem_of[(size_t)i] = at;
        }
    }
    int64_t n_slots = 1;
    for (int b = 0; b < nb; ++b) n_slots = std::max(n_slots, used[b]);
    L.kind = SSLREC_PLAN_SWEPT;
    L.d = d;
    L.xcd_split = split;
    L.xcd_col_pairs = xcd_pairs;
    L.arrays["pack"].set(pack); L.arrays["val"].set(val);
    L.arrays["w_start"].set(w_start); L.arrays["w_steps"].set(w_steps);
    L.arrays["wf_ptr"].set(wf_ptr); L.arrays["cf_ptr"].set(cf_ptr);
    L.arrays["f_row"].set(f_row); L.arrays["f_start"].set(f_start); L.arrays["f_n"].set(f_n);
    L.arrays["edge_map"].set(emap);
    L.arrays["elem_host"].set(elem_of);         // element of the i-th (lane group, column)-sorted entry
    L.arrays["csr_pos_host"].set(o);            // its CSR position
    sslrec_swept_t &S = L.swept;
    S.n_rows = n; S.n_cols = p.n_cols; S.nnz = (int32_t)nnz; S.d = d;
    S.n_elem = (int32_t)n_elem; S.n_blocks = nb; S.n_slots = (int32_t)n_slots;
    return 0;
}

// ---- streamed, packed CSR (kernel contract in spmm.hip / sslrec_csr_t) -----------------------------------------
struct PhaseTimer {
    bool on = getenv("SSLREC_PLAN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[plan] %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

int build_streamed(const sslrec_plan &p, int d, Layout &L, std::string &why) {
    PhaseTimer tm;
    const int G = 256 / d;
    const int n = p.n_rows;
    const int64_t nnz = p.nnz;
    int64_t n_waves = kMaxStreams;
    if (const char *env = getenv("SSLREC_SPMM_STREAMS")) n_waves = atoll(env);
    if (p.n_streams > 0) n_waves = p.n_streams;
    n_waves = std::min<int64_t>(n_waves, std::max<int64_t>(1, nnz / kMinStream));
    const int64_t chunk_cap = p.seg_max > 0 ? p.seg_max
        : std::max<int64_t>(64, ((nnz + std::max<int64_t>(n_waves, 1) - 1) / std::max<int64_t>(n_waves, 1)) / 2);
    // row segments: short rows whole (row order), then the chunks of long rows (contiguous, balanced sizes)
    std::vector<int64_t> seg_start, seg_len;
    std::vector<int32_t> seg_dst, long_row, long_ptr(1, 0);
    for (int r = 0; r < n; ++r) {
        const int64_t len = p.rowptr[r + 1] - p.rowptr[r];
        if (std::max<int64_t>(1, (len + chunk_cap - 1) / chunk_cap) == 1) {
            seg_dst.push_back(r); seg_start.push_back(p.rowptr[r]); seg_len.push_back(len);
        }
    }
    int64_t n_slots = 0;
    for (int r = 0; r < n; ++r) {
        const int64_t len = p.rowptr[r + 1] - p.rowptr[r];
        const int64_t nc = std::max<int64_t>(1, (len + chunk_cap - 1) / chunk_cap);
        if (nc == 1) continue;
        long_row.push_back(r);
        const int64_t base = len / nc, rem = len % nc;
        for (int64_t k = 0; k < nc; ++k) {
            seg_dst.push_back((int32_t)~(n_slots + k));
            seg_start.push_back(p.rowptr[r] + k * base + std::min(k, rem));
            seg_len.push_back(base + (k < rem));
        }
        n_slots += nc;
        long_ptr.push_back((int32_t)n_slots);
    }
    const int64_t n_seg = (int64_t)seg_len.size();
    tm.mark("segments");
    n_waves = std::max<int64_t>(1, std::min(n_waves, n_seg));
    // deal the segments to the streams, longest first (exact greedy up to 2M segments, boustrophedon beyond)
    std::vector<int64_t> order((size_t)n_seg);
    for (int64_t i = 0; i < n_seg; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return seg_len[(size_t)a] > seg_len[(size_t)b]; });
    std::vector<int64_t> wave_of_rank((size_t)n_seg);
    if (n_seg > 2000000) {
        for (int64_t j = 0; j < n_seg; ++j) {
            const int64_t rnd = j / n_waves, pos = j % n_waves;
            wave_of_rank[(size_t)j] = (rnd % 2 == 0) ? pos : n_waves - 1 - pos;
        }
    } else {
        MinHeap heap;
        for (int k = 0; k < n_waves; ++k) heap.push({0, k});
        for (int64_t j = 0; j < n_seg; ++j) {
            LoadId top = heap.top();
            heap.pop();
            wave_of_rank[(size_t)j] = top.second;
            heap.push({top.first + seg_len[(size_t)order[(size_t)j]] + kRowOverhead, top.second});
        }
    }
    tm.mark("deal");
    // stream layout: segments grouped by stream, inside a stream in dealing order (longest first)
    std::vector<int64_t> rank((size_t)n_seg), by_wave;
    for (int64_t i = 0; i < n_seg; ++i) rank[(size_t)i] = i;
    counting_pass(rank, by_wave, [&](int64_t j) { return wave_of_rank[(size_t)j]; }, n_waves);
    std::vector<int32_t> r_ptr((size_t)n_waves + 1, 0), r_len_loads((size_t)n_seg), r_dst((size_t)n_seg), w_len((size_t)n_waves, 0),
        w_start((size_t)n_waves, 0);
    std::vector<int64_t> r_len_entries((size_t)n_seg), r_start((size_t)n_seg);
    std::vector<int32_t> seg_wave((size_t)n_seg);
    for (int64_t i = 0; i < n_seg; ++i) {
        const int64_t j = by_wave[(size_t)i], sgm = order[(size_t)j], w = wave_of_rank[(size_t)j];
        r_len_entries[(size_t)i] = seg_len[(size_t)sgm];
        r_start[(size_t)i] = seg_start[(size_t)sgm];
        r_dst[(size_t)i] = seg_dst[(size_t)sgm];
        r_len_loads[(size_t)i] = (int32_t)((seg_len[(size_t)sgm] + G - 1) / G);
        seg_wave[(size_t)i] = (int32_t)w;
        ++r_ptr[(size_t)w + 1];
        w_len[(size_t)w] += r_len_loads[(size_t)i];
    }
    for (int64_t w = 0; w < n_waves; ++w) r_ptr[(size_t)w + 1] += r_ptr[(size_t)w];
    int64_t n_elem = 0;
    for (int64_t w = 0; w < n_waves; ++w) {
        const int64_t elems = ((int64_t)w_len[(size_t)w] + 3) / 4 * 4 * G;
        if (n_elem >= 2147483647LL - elems) { why = "packed layout exceeds int32 indexing"; return 1; }
        w_start[(size_t)w] = (int32_t)n_elem;
        n_elem += elems;
    }
    tm.mark("stream table");
    std::vector<int32_t> col((size_t)std::max<int64_t>(n_elem, 1), -1), emap((size_t)std::max<int64_t>(n_elem, 1), -1);
    std::vector<float> val((size_t)std::max<int64_t>(n_elem, 1), 0.f);
    std::vector<int64_t> elem_of, src_index;
    elem_of.reserve((size_t)nnz);
    src_index.reserve((size_t)nnz);
    {
        int64_t slot0 = 0;
        int32_t cur = -1;
        for (int64_t i = 0; i < n_seg; ++i) {
            if (seg_wave[(size_t)i] != cur) { cur = seg_wave[(size_t)i]; slot0 = 0; }
            for (int64_t j = 0; j < r_len_entries[(size_t)i]; ++j) {
                const int64_t k = slot0 + j, load = k / G, sub = k % G;
                const int64_t at = (int64_t)w_start[(size_t)cur] + (load >> 2) * (4 * G) + sub * 4 + (load & 3);
                const int64_t src = r_start[(size_t)i] + j;
                col[(size_t)at] = p.col[(size_t)src];
                val[(size_t)at] = p.val[(size_t)src];
                emap[(size_t)at] = (int32_t)p.perm[(size_t)src];
                elem_of.push_back(at);
                src_index.push_back(src);
            }
            slot0 += (int64_t)r_len_loads[(size_t)i] * G;
        }
    }
    tm.mark("fill");
    const int32_t n_long = (int32_t)long_row.size();      // the vectors are MOVED into the layout below
    L.kind = SSLREC_PLAN_STREAMED;
    L.d = d;
    L.arrays["col"].set(col); L.arrays["val"].set(val);
    L.arrays["w_start"].set(w_start); L.arrays["w_len"].set(w_len); L.arrays["r_ptr"].set(r_ptr);
    L.arrays["r_len"].set(r_len_loads); L.arrays["r_dst"].set(r_dst);
    L.arrays["long_row"].set(long_row); L.arrays["long_ptr"].set(long_ptr);
    L.arrays["edge_map"].set(emap);
    L.arrays["elem_host"].set(elem_of); L.arrays["csr_pos_host"].set(src_index);
    sslrec_csr_t &S = L.csr;
    S.n_rows = n; S.n_cols = p.n_cols; S.nnz = (int32_t)nnz; S.d = d; S.n_elem = (int32_t)n_elem;
    S.n_waves = (int32_t)n_waves; S.n_rseg = (int32_t)n_seg; S.n_long = n_long; S.n_slots = (int32_t)n_slots;
    tm.mark("copy out");
    return 0;
}

// ---- row-bundled streamed layout for narrow tables (kernel contract: spmm_bundle_kernel in spmm.hip / sslrec_bundled_t) ----
int build_bundled(const sslrec_plan &p, int d, Layout &L, std::string &why) {
    PhaseTimer tm;
    const int G = 256 / d, LPG = 64 / G, S = std::min(16, LPG);
    const int n = p.n_rows;
    const int64_t nnz = p.nnz;
    int64_t n_waves = 256 * 32;                       // 8 resident wavefronts per SIMD
    if (const char *env = getenv("SSLREC_SPMM_STREAMS")) n_waves = atoll(env);
    if (p.n_streams > 0) n_waves = p.n_streams;
    n_waves = std::max<int64_t>(1, n_waves);
    // a segment occupies ONE lane group for its length in steps; a wave advances G segments per step
    const int64_t chunk_cap = p.seg_max > 0 ? p.seg_max : std::max<int64_t>(64, nnz / G / n_waves / 2);
    std::vector<int64_t> seg_start, seg_len;
    std::vector<int32_t> seg_dst, long_row, long_ptr(1, 0);
    int64_t n_slots = 0;
    for (int r = 0; r < n; ++r) {
        const int64_t len = p.rowptr[r + 1] - p.rowptr[r];
        const int64_t nc = std::max<int64_t>(1, (len + chunk_cap - 1) / chunk_cap);
        if (nc == 1) {
            seg_dst.push_back(r); seg_start.push_back(p.rowptr[r]); seg_len.push_back(len);
            continue;
        }
        long_row.push_back(r);
        const int64_t base = len / nc, rem = len % nc;
        for (int64_t k = 0; k < nc; ++k) {
            seg_dst.push_back((int32_t)~(n_slots + k));
            seg_start.push_back(p.rowptr[r] + k * base + std::min(k, rem));
            seg_len.push_back(base + (k < rem));
        }
        n_slots += nc;
        long_ptr.push_back((int32_t)n_slots);
    }
    const int64_t n_seg = (int64_t)seg_len.size();
    tm.mark("bundled: segments");
    // segments by length, longest first (stable: ties in segment order); G consecutive ones form a bundle
    int64_t len_max = 0;
    for (int64_t v : seg_len) len_max = std::max(len_max, v);
    std::vector<int64_t> order((size_t)n_seg);
    {
        std::vector<int64_t> cnt((size_t)len_max + 2, 0);
        for (int64_t v : seg_len) ++cnt[(size_t)(len_max - v) + 1];
        for (int64_t k = 0; k <= len_max; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
        for (int64_t i = 0; i < n_seg; ++i) order[(size_t)cnt[(size_t)(len_max - seg_len[(size_t)i])]++] = i;
    }
    const int64_t n_bundles = (n_seg + G - 1) / G;
    if (n_bundles * (int64_t)G >= 2147483647LL) { why = "bundled layout exceeds int32 indexing"; return 1; }
    std::vector<int32_t> b_steps_of((size_t)n_bundles);
    for (int64_t b = 0; b < n_bundles; ++b)
        b_steps_of[(size_t)b] = (int32_t)((seg_len[(size_t)order[(size_t)(b * G)]] + S - 1) / S * S);      // its first member is its longest
    n_waves = std::max<int64_t>(1, std::min(n_waves, n_bundles));
    // bundles -> streams, longest first (they already are in that order): exact greedy
    std::vector<int32_t> wave_of((size_t)n_bundles);
    {
        MinHeap heap;
        for (int k = 0; k < n_waves; ++k) heap.push({0, k});
        for (int64_t b = 0; b < n_bundles; ++b) {
            LoadId top = heap.top();
            heap.pop();
            wave_of[(size_t)b] = top.second;
            heap.push({top.first + b_steps_of[(size_t)b] + 2 * S, top.second});      // + the cost of finishing a bundle
        }
    }
    tm.mark("bundled: deal");
    std::vector<int64_t> ids((size_t)n_bundles), by_wave;
    for (int64_t b = 0; b < n_bundles; ++b) ids[(size_t)b] = b;
    counting_pass(ids, by_wave, [&](int64_t b) { return (int64_t)wave_of[(size_t)b]; }, n_waves);      // a stream's bundles longest first
    std::vector<int32_t> w_ptr((size_t)n_waves + 1, 0), w_start((size_t)n_waves + 1, 0), b_steps((size_t)n_bundles), b_dst((size_t)n_bundles * G);
    for (int64_t b = 0; b < n_bundles; ++b) ++w_ptr[(size_t)wave_of[(size_t)b] + 1];
    for (int64_t w = 0; w < n_waves; ++w) w_ptr[(size_t)w + 1] += w_ptr[(size_t)w];
    std::vector<int64_t> b_start((size_t)n_bundles);
    int64_t n_elem = 0;
    for (int64_t i = 0; i < n_bundles; ++i) {
        const int64_t b = by_wave[(size_t)i];
        if (i == 0 || wave_of[(size_t)b] != wave_of[(size_t)by_wave[(size_t)(i - 1)]]) w_start[(size_t)wave_of[(size_t)b]] = (int32_t)n_elem;
        b_steps[(size_t)i] = b_steps_of[(size_t)b];
        b_start[(size_t)i] = n_elem;
        n_elem += (int64_t)b_steps_of[(size_t)b] / S * 64;
        if (n_elem >= 2147483647LL - 64) { why = "bundled layout exceeds int32 indexing"; return 1; }
    }
    // streams are stored back to back: w_start[w + 1] - w_start[w] = the stream's elements (empty streams included)
    w_start[(size_t)n_waves] = (int32_t)n_elem;
    for (int64_t w = n_waves - 1; w >= 0; --w)
        if (w_ptr[(size_t)w] == w_ptr[(size_t)w + 1]) w_start[(size_t)w] = w_start[(size_t)w + 1];
    tm.mark("bundled: stream table");
    std::vector<int32_t> col((size_t)std::max<int64_t>(n_elem, 1), -1), emap((size_t)std::max<int64_t>(n_elem, 1), -1);
    std::vector<float> val((size_t)std::max<int64_t>(n_elem, 1), 0.f);
    std::vector<int64_t> elem_of, src_index;
    elem_of.reserve((size_t)nnz);
    src_index.reserve((size_t)nnz);
    for (int64_t i = 0; i < n_bundles; ++i) {
        const int64_t b = by_wave[(size_t)i];
        for (int g = 0; g < G; ++g) {
            const int64_t k = b * G + g;
            if (k >= n_seg) { b_dst[(size_t)i * G + g] = SSLREC_BUNDLE_NONE; continue; }
            const int64_t sgm = order[(size_t)k];
            b_dst[(size_t)i * G + g] = seg_dst[(size_t)sgm];
            for (int64_t j = 0; j < seg_len[(size_t)sgm]; ++j) {
                const int64_t at = b_start[(size_t)i] + (j / S) * 64 + (int64_t)g * LPG + (j % S);
                const int64_t src = seg_start[(size_t)sgm] + j;
                col[(size_t)at] = p.col[(size_t)src];
                val[(size_t)at] = p.val[(size_t)src];
                emap[(size_t)at] = (int32_t)p.perm[(size_t)src];
                elem_of.push_back(at);
                src_index.push_back(src);
            }
        }
    }
    tm.mark("bundled: fill");
    const int32_t n_long = (int32_t)long_row.size();
    L.kind = SSLREC_PLAN_BUNDLED;
    L.d = d;
    L.arrays["col"].set(col); L.arrays["val"].set(val);
    L.arrays["w_start"].set(w_start); L.arrays["w_ptr"].set(w_ptr);
    L.arrays["b_steps"].set(b_steps); L.arrays["b_dst"].set(b_dst);
    L.arrays["long_row"].set(long_row); L.arrays["long_ptr"].set(long_ptr);
    L.arrays["edge_map"].set(emap);
    L.arrays["elem_host"].set(elem_of); L.arrays["csr_pos_host"].set(src_index);
    sslrec_bundled_t &B = L.bundled;
    B.n_rows = n; B.n_cols = p.n_cols; B.nnz = (int32_t)nnz; B.d = d; B.n_elem = (int32_t)n_elem;
    B.n_waves = (int32_t)n_waves; B.n_bundles = (int32_t)n_bundles; B.n_long = n_long; B.n_slots = (int32_t)n_slots;
    return 0;
}

int finish_csr(sslrec_plan *p) {
    p->rowptr.assign((size_t)p->n_rows + 1, 0);
    return 0;
}

}  // namespace

extern "C" int sslrec_plan_build_coo(const int64_t *rows, const int64_t *cols, const float *vals, int64_t nnz, int32_t n_rows,
                                     int32_t n_cols, sslrec_plan_t **out) {
    if (!out || nnz < 0 || n_rows < 0 || n_cols < 0 || (nnz > 0 && (!rows || !cols || !vals)) || nnz >= 2147483647LL)
        return SSLREC_E_BADARG;
    for (int64_t e = 0; e < nnz; ++e)
        if (rows[e] < 0 || rows[e] >= n_rows || cols[e] < 0 || cols[e] >= n_cols) return SSLREC_E_BADARG;
    std::unique_ptr<sslrec_plan> p(new sslrec_plan);
    p->n_rows = n_rows; p->n_cols = n_cols; p->nnz = nnz;
    finish_csr(p.get());
    // stable sort by (row, column): duplicates keep their input order, which fixes the summation order
    std::vector<int64_t> idx((size_t)nnz), tmp;
    for (int64_t e = 0; e < nnz; ++e) idx[(size_t)e] = e;
    counting_pass(idx, tmp, [&](int64_t e) { return cols[e]; }, n_cols);
    counting_pass(tmp, idx, [&](int64_t e) { return rows[e]; }, n_rows);
    p->perm.swap(idx);
    p->col.resize((size_t)nnz);
    p->val.resize((size_t)nnz);
    for (int64_t i = 0; i < nnz; ++i) {
        const int64_t e = p->perm[(size_t)i];
        p->col[(size_t)i] = (int32_t)cols[e];
        p->val[(size_t)i] = vals[e];
        ++p->rowptr[(size_t)rows[e] + 1];
    }
    for (int r = 0; r < n_rows; ++r) p->rowptr[(size_t)r + 1] += p->rowptr[(size_t)r];
    *out = p.release();
    return 0;
}

extern "C" int sslrec_plan_build_csr(const int64_t *rowptr, const int32_t *col, const float *val, int32_t n_rows, int32_t n_cols,
                                     sslrec_plan_t **out) {
    if (!out || !rowptr || n_rows < 0 || n_cols < 0 || rowptr[0] != 0) return SSLREC_E_BADARG;
    const int64_t nnz = rowptr[n_rows];
    if (nnz < 0 || nnz >= 2147483647LL || (nnz > 0 && (!col || !val))) return SSLREC_E_BADARG;
    for (int r = 0; r < n_rows; ++r)
        if (rowptr[r + 1] < rowptr[r]) return SSLREC_E_BADARG;
    for (int64_t e = 0; e < nnz; ++e)
        if (col[e] < 0 || col[e] >= n_cols) return SSLREC_E_BADARG;
    std::unique_ptr<sslrec_plan> p(new sslrec_plan);      // entries keep the order they are given in: it IS the summation order
    p->n_rows = n_rows; p->n_cols = n_cols; p->nnz = nnz;
    p->rowptr.assign(rowptr, rowptr + n_rows + 1);
    p->col.assign(col, col + nnz);
    p->val.assign(val, val + nnz);
    p->perm.resize((size_t)nnz);
    for (int64_t e = 0; e < nnz; ++e) p->perm[(size_t)e] = e;
    *out = p.release();
    return 0;
}

extern "C" int sslrec_plan_set_option(sslrec_plan_t *p, const char *name, int64_t value) {
    if (!p || !name || value < 0) return SSLREC_E_BADARG;
    const std::string key(name);
    if (key == "seg_max") p->seg_max = value;
    else if (key == "n_streams") p->n_streams = value;
    else if (key == "swept_blocks" && (value == 0 || value == 256 || value == 512)) p->swept_blocks = value;
    else if (key == "xcd_balance" && value <= 1000) p->xcd_balance = value;
    else if (key == "xcd_stagger" && value <= 200) p->xcd_stagger = value;
    else if (key == "xcd_cluster" && value <= 17) p->xcd_cluster = value == 17 ? -1 : value;      // (17 = automatic, the default)
    else if (key == "swept_passes" && value <= 1) p->swept_passes = value;
    else if (key == "bundled32" && value <= 1) p->bundled32 = value;
    else if (key == "swept_width" && (value == 0 || value == 32 || value == 64 || value == 128 || value == 256)) p->swept_width = value;
    else return SSLREC_E_BADARG;
    return 0;
}

static Layout *layout_of(const sslrec_plan_t *p, int32_t d, int32_t kind) {
    if (!p) return nullptr;
    if (kind == SSLREC_PLAN_AUTO) {
        Layout *L = layout_of(p, d, SSLREC_PLAN_SWEPT);
        return L ? L : layout_of(p, d, SSLREC_PLAN_STREAMED);
    }
    auto it = p->layouts.find(d * 4 + kind);
    return it == p->layouts.end() ? nullptr : it->second.get();
}

extern "C" int sslrec_plan_layout(sslrec_plan_t *p, int32_t d, int32_t kind, int32_t flags) {
    // d = 8 / 16 (feature-sliced tables, sslrec_amd/feature_shard.py): the column-swept layout, and the row-bundled one as
    // their streamed kind
    const bool narrow = d == 8 || d == 16;
    if (!p || (!narrow && d != 32 && d != 64 && d != 128 && d != 256) || kind < 0 || kind > 2) return -SSLREC_E_BADARG;
    if (kind != SSLREC_PLAN_AUTO) {
        if (Layout *L = layout_of(p, d, kind)) return L->kind;
    }
    std::string why;
    if (kind == SSLREC_PLAN_AUTO || kind == SSLREC_PLAN_SWEPT) {
        if (layout_of(p, d, SSLREC_PLAN_SWEPT)) return SSLREC_PLAN_SWEPT;
        // tables wider than the LDS holds run in embedding-column passes over a layout of d/2, d/4, ... columns
        // (sslrec_plan_info reports that width; the SpMM entry points take the tables' own d)
        for (int ds = (p->swept_width > 0 && p->swept_width < d) ? (int)p->swept_width : d; ds >= (narrow ? d : 32); ds /= 2) {
            std::unique_ptr<Layout> L(new Layout);
            if (build_swept(*p, ds, flags, *L, why) == 0) {
                p->layouts[d * 4 + SSLREC_PLAN_SWEPT] = std::move(L);
                return SSLREC_PLAN_SWEPT;
            }
            if (p->swept_passes == 0) break;
        }
        if (kind == SSLREC_PLAN_SWEPT) return -SSLREC_E_BADARG;
    }
    if (Layout *L = layout_of(p, d, SSLREC_PLAN_STREAMED)) return L->kind;
    std::unique_ptr<Layout> L(new Layout);
    const bool bundled = narrow || (d == 32 && p->bundled32);
    if ((bundled ? build_bundled(*p, d, *L, why) : build_streamed(*p, d, *L, why)) != 0) return -SSLREC_E_BADARG;
    const int built = L->kind;
    p->layouts[d * 4 + SSLREC_PLAN_STREAMED] = std::move(L);
    return built;
}

extern "C" int sslrec_plan_host_array(const sslrec_plan_t *p, int32_t d, int32_t kind, const char *name, const void **ptr,
                                      int64_t *count, int32_t *elem_bytes) {
    if (!p || !name || !ptr || !count) return SSLREC_E_BADARG;
    const std::string key(name);
    if (d == 0) {          // the d-independent CSR
        if (key == "rowptr") { *ptr = p->rowptr.data(); *count = (int64_t)p->rowptr.size(); if (elem_bytes) *elem_bytes = 8; return 0; }
        if (key == "col") { *ptr = p->col.data(); *count = (int64_t)p->col.size(); if (elem_bytes) *elem_bytes = 4; return 0; }
        if (key == "val") { *ptr = p->val.data(); *count = (int64_t)p->val.size(); if (elem_bytes) *elem_bytes = 4; return 0; }
        if (key == "perm") { *ptr = p->perm.data(); *count = (int64_t)p->perm.size(); if (elem_bytes) *elem_bytes = 8; return 0; }
        return SSLREC_E_BADARG;
    }
    Layout *L = layout_of(p, d, kind);
    if (!L) return SSLREC_E_BADARG;
    auto it = L->arrays.find(key);
    if (it == L->arrays.end()) return SSLREC_E_BADARG;
    *ptr = it->second.data;
    *count = it->second.count;
    if (elem_bytes) *elem_bytes = (int32_t)it->second.elem;
    return 0;
}

extern "C" int sslrec_plan_info(const sslrec_plan_t *p, int32_t d, int32_t kind, sslrec_plan_info_t *info) {
    if (!p || !info) return SSLREC_E_BADARG;
    memset(info, 0, sizeof(*info));
    info->n_rows = p->n_rows; info->n_cols = p->n_cols; info->nnz = p->nnz;
    Layout *L = d == 0 ? nullptr : layout_of(p, d, kind);
    if (!L) return d == 0 ? 0 : SSLREC_E_BADARG;
    info->kind = L->kind; info->d = L->d; info->xcd_split = L->xcd_split; info->xcd_col_pairs = L->xcd_col_pairs;
    if (L->kind == SSLREC_PLAN_SWEPT) {
        info->n_elem = L->swept.n_elem; info->n_blocks = L->swept.n_blocks; info->n_slots = L->swept.n_slots;
    } else if (L->kind == SSLREC_PLAN_BUNDLED) {
        info->n_elem = L->bundled.n_elem; info->n_streams = L->bundled.n_waves; info->n_rseg = L->bundled.n_bundles;
        info->n_long = L->bundled.n_long; info->n_slots = L->bundled.n_slots;
    } else {
        info->n_elem = L->csr.n_elem; info->n_streams = L->csr.n_waves; info->n_rseg = L->csr.n_rseg;
        info->n_long = L->csr.n_long; info->n_slots = L->csr.n_slots;
    }
    return 0;
}

extern "C" int sslrec_plan_upload(sslrec_plan_t *p, int32_t d, int32_t kind, void *stream) {
    Layout *L = layout_of(p, d, kind);
    if (!L) return SSLREC_E_BADARG;
    if (L->uploaded) return 0;
    hipStream_t st = (hipStream_t)stream;
    for (auto &kv : L->arrays) {
        if (kv.first == "elem_host" || kv.first == "csr_pos_host") continue;
        hipError_t e = hipMalloc(&kv.second.dev, kv.second.bytes());
        if (e != hipSuccess) return (int)e;
        e = hipMemcpyAsync(kv.second.dev, kv.second.data, kv.second.bytes(), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    hipError_t e = hipStreamSynchronize(st);      // the host vectors may be reused after this call returns
    if (e != hipSuccess) return (int)e;
    auto dev = [&](const char *k) { return L->arrays[k].dev; };
    if (L->kind == SSLREC_PLAN_SWEPT) {
        sslrec_swept_t &S = L->swept;
        S.pack = (const int32_t *)dev("pack"); S.val = (const float *)dev("val");
        S.w_start = (const int32_t *)dev("w_start"); S.w_steps = (const int32_t *)dev("w_steps");
        S.wf_ptr = (const int32_t *)dev("wf_ptr"); S.cf_ptr = (const int32_t *)dev("cf_ptr"); S.f_row = (const int32_t *)dev("f_row");
        S.f_start = (const int32_t *)dev("f_start"); S.f_n = (const int32_t *)dev("f_n");
    } else if (L->kind == SSLREC_PLAN_BUNDLED) {
        sslrec_bundled_t &S = L->bundled;
        S.col = (const int32_t *)dev("col"); S.val = (const float *)dev("val");
        S.w_start = (const int32_t *)dev("w_start"); S.w_ptr = (const int32_t *)dev("w_ptr");
        S.b_steps = (const int32_t *)dev("b_steps"); S.b_dst = (const int32_t *)dev("b_dst");
        S.long_row = (const int32_t *)dev("long_row"); S.long_ptr = (const int32_t *)dev("long_ptr");
        if (S.n_slots > 0) {
            e = hipMalloc((void **)&L->partial_ws, (size_t)S.n_slots * L->d * sizeof(float));
            if (e != hipSuccess) return (int)e;
        }
    } else {
        sslrec_csr_t &S = L->csr;
        S.col = (const int32_t *)dev("col"); S.val = (const float *)dev("val");
        S.w_start = (const int32_t *)dev("w_start"); S.w_len = (const int32_t *)dev("w_len"); S.r_ptr = (const int32_t *)dev("r_ptr");
        S.r_len = (const int32_t *)dev("r_len"); S.r_dst = (const int32_t *)dev("r_dst");
        S.long_row = (const int32_t *)dev("long_row"); S.long_ptr = (const int32_t *)dev("long_ptr");
        if (S.n_slots > 0) {
            e = hipMalloc((void **)&L->partial_ws, (size_t)S.n_slots * L->d * sizeof(float));
            if (e != hipSuccess) return (int)e;
        }
    }
    L->uploaded = true;
    return 0;
}

extern "C" const sslrec_swept_t *sslrec_plan_swept(const sslrec_plan_t *p, int32_t d) {
    Layout *L = layout_of(p, d, SSLREC_PLAN_SWEPT);
    return (L && L->uploaded && L->kind == SSLREC_PLAN_SWEPT) ? &L->swept : nullptr;
}

extern "C" const sslrec_csr_t *sslrec_plan_csr(const sslrec_plan_t *p, int32_t d) {
    Layout *L = layout_of(p, d, SSLREC_PLAN_STREAMED);
    return (L && L->uploaded && L->kind == SSLREC_PLAN_STREAMED) ? &L->csr : nullptr;
}

extern "C" const sslrec_bundled_t *sslrec_plan_bundled(const sslrec_plan_t *p, int32_t d) {
    Layout *L = layout_of(p, d, SSLREC_PLAN_STREAMED);
    return (L && L->uploaded && L->kind == SSLREC_PLAN_BUNDLED) ? &L->bundled : nullptr;
}

extern "C" const int32_t *sslrec_plan_edge_map(const sslrec_plan_t *p, int32_t d, int32_t kind) {
    Layout *L = layout_of(p, d, kind);
    return (L && L->uploaded) ? (const int32_t *)L->arrays["edge_map"].dev : nullptr;
}

extern "C" int sslrec_plan_spmm_f32(const sslrec_plan_t *p, int32_t d, const float *X, float *Y, const sslrec_epilogue_t *epi,
                                    void *stream) {
    Layout *L = layout_of(p, d, SSLREC_PLAN_AUTO);      // the swept layout when one was built, else the streamed one
    if (!L || !L->uploaded) return SSLREC_E_BADARG;
    if (L->kind == SSLREC_PLAN_SWEPT) return sslrec_spmm_swept_f32(&L->swept, nullptr, nullptr, nullptr, X, d, Y, epi, stream);
    if (L->kind == SSLREC_PLAN_BUNDLED) return sslrec_spmm_bundled_f32(&L->bundled, nullptr, X, d, Y, epi, L->partial_ws, stream);
    return sslrec_spmm_csr_f32(&L->csr, nullptr, nullptr, nullptr, nullptr, X, d, Y, epi, L->partial_ws, stream);
}

extern "C" void sslrec_plan_free(sslrec_plan_t *p) {
    if (!p) return;
    for (auto &kv : p->layouts) {
        for (auto &a : kv.second->arrays)
            if (a.second.dev) (void)hipFree(a.second.dev);
        if (kv.second->partial_ws) (void)hipFree(kv.second->partial_ws);
    }
    delete p;
}
