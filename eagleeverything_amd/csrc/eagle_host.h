// eagle_host.h -- the HIP-free host pieces of libeaglehip.so: plain C++17, no device types, so that a CPU test binary can build
// them with -fsanitize=address,undefined / -fsanitize=thread (tests/host/, run by tests/test_host_sanitizers.py; GPU sanitizers do
// not exist on the target pool).  Everything here is used by eagle_api.cpp / eagle_ingest.cpp / eagle_i8mfma.hip as is.
#ifndef EAGLE_HOST_H
#define EAGLE_HOST_H
#include <math.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__HIPCC__)
#define EAGLE_HD __host__ __device__
#else
#define EAGLE_HD
#endif

// ------------------------------------------------------------------------------------------------
// Layout of the 8 KiB ctx scratch allocation (flags and small reductions of stream-ordered helpers).  One place, no overlap:
// the loaders (load stream) and the scan (compute stream) of ONE call run concurrently, so two users must never share bytes.
// ------------------------------------------------------------------------------------------------
enum : size_t {
    EAGLE_SCR_SYM = 0,             // int[2]: k_sym_check's verdict on S and V                      (eagle_kernels.hip)
    EAGLE_SCR_CERT_TOTALS = 256,   // long[4]: certification counters summed over marker blocks      (eagle_api.cpp scan_range)
    EAGLE_SCR_INGEST = 512,        // u64[2]: first third-allele / first missing position; double: trace  (eagle_ingest.cpp, eagle_linalg.cpp)
    EAGLE_SCR_LOADER_BAD = 1024,   // int: invalid characters / codes seen by the tile loaders       (eagle_api.cpp, any stream)
    EAGLE_SCR_SCACHE_FLAG = 2048,  // int: cached S differs from the caller's                        (eagle_api.cpp, load stream)
    EAGLE_SCR_DOT_PARTIALS = 4096, // double[256]: partial sums of eagle_dev_dot_matrices            (eagle_kernels.hip)
    EAGLE_SCR_BYTES = 8192
};
static_assert(EAGLE_SCR_SYM + 2 * sizeof(int) <= EAGLE_SCR_CERT_TOTALS, "scratch overlap");
static_assert(EAGLE_SCR_CERT_TOTALS + 4 * sizeof(long) <= EAGLE_SCR_INGEST, "scratch overlap");
static_assert(EAGLE_SCR_INGEST + 2 * sizeof(unsigned long long) <= EAGLE_SCR_LOADER_BAD, "scratch overlap");
static_assert(EAGLE_SCR_LOADER_BAD + sizeof(int) <= EAGLE_SCR_SCACHE_FLAG, "scratch overlap");
static_assert(EAGLE_SCR_SCACHE_FLAG + sizeof(int) <= EAGLE_SCR_DOT_PARTIALS, "scratch overlap");
static_assert(EAGLE_SCR_DOT_PARTIALS + 256 * sizeof(double) <= EAGLE_SCR_BYTES, "scratch overlap");

// ------------------------------------------------------------------------------------------------
// Meeting point of the per-device worker threads of one multi-device call.  arrive(ok, v) blocks until every device has
// arrived and returns the outcome OF THAT ROUND: false if any device had reported a failure by the time the round completed
// (then nobody enters the collective that follows); the largest v of the round is left in `vmax`, the sum of the `add`s in `vsum`.  Every worker calls it the
// same number of times, failed or not.  The outcome is latched per round by the last arriver: a device that leaves round k
// and fails before a slower peer has woken up from round k can only influence round k + 1 -- the peer still sees round k's
// verdict, makes its own round k + 1 arrival, and both leave round k + 1 with `false` (a sticky flag read after the wake-up
// stranded the fast device in round k + 1 for ever).  `failed` stays sticky, so every later round fails too.
// `round_failed` and `vmax` cannot change before the reader's own next arrival: the next round needs it to complete.
// ------------------------------------------------------------------------------------------------
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    long gen = 0;
    bool failed = false, round_failed = false;
    double acc = -HUGE_VAL, vmax = -HUGE_VAL;
    long sacc = 0, vsum = 0;
    bool arrive(bool ok, double v = -HUGE_VAL, long add = 0) {
        std::unique_lock<std::mutex> lk(mu);
        if (!ok) failed = true;
        if (v == v && v > acc) acc = v;
        sacc += add;
        const long g = gen;
        if (++waiting == n) {
            waiting = 0; vmax = acc; acc = -HUGE_VAL; vsum = sacc; sacc = 0; round_failed = failed; gen++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
        return !round_failed;
    }
};

// Contiguous marker ranges, one per device; boundaries at multiples of 256 (kernel tiles, and so that the lead's range is a
// prefix view of a whole-file image the converters may have left resident).
static inline void split_markers(long L, int ndev, std::vector<long>& edge) {
    edge.assign((size_t)ndev + 1, 0);
    const long tiles = (L + 255) / 256;
    for (int k = 1; k < ndev; k++) edge[(size_t)k] = std::min(L, (tiles * k / ndev) * 256);
    edge[(size_t)ndev] = L;
}

// selected_loci rule (calculateMMt_rcpp.cpp:88; calculate_a_and_vara_rcpp.cpp:79; calculate_reduced_a_rcpp.cpp:74): masking fires
// iff element 0 is not NA (NA arrives as NaN).  Returns nullptr, or the message of the argument error.
static inline const char* parse_selected_core(const double* sel, long nsel, long bound, std::vector<long>& out) {
    out.clear();
    if (nsel <= 0 || !sel || isnan(sel[0])) return nullptr;
    for (long i = 0; i < nsel; i++) {
        if (isnan(sel[i])) return "NA in selected_loci after element 0";
        const long v = (long)sel[i];
        if (v < 0 || v >= bound) return "selected_loci index out of range";
        out.push_back(v);
    }
    return nullptr;
}

// Rows (multiple of 256) of a streamed chunk whose padded row length is `row_bytes`: two chunk buffers share `budget` bytes
// ((size_t)-1: no explicit budget -- streaming because HBM is full -- 8 GiB).
static inline long stream_chunk_rows_core(size_t budget, long row_bytes, long total_rows_pad) {
    if (budget == (size_t)-1) budget = (size_t)8 << 30;
    long rows = (long)(budget / 2 / (size_t)row_bytes) / 256 * 256;
    if (rows < 256) rows = 256;
    return rows < total_rows_pad ? rows : total_rows_pad;
}

// Pieces per worker of an XCD's last, partly filled round of `tail` workers on 32 CUs (0 < tail < 32): the workers are cut along
// their column-tile pairs into p equal pieces, the tail * p pieces run in ceil(tail * p / 32) rounds of 1/p worker-time each; p <=
// min(npair, VARA_TAIL_PMAX) minimising that cost (p = 1: one whole worker-time for a round that may be 1/32 full; large p: tail/32).
#define VARA_TAIL_PMAX 16
EAGLE_HD static inline int vara_tail_pieces(int tail, int npair) {
    if (tail <= 0) return 1;
    int best = 1, bn = 1, bd = 1;  // cost bn / bd
    const int pmax = npair < VARA_TAIL_PMAX ? npair : VARA_TAIL_PMAX;
    for (int p = 2; p <= pmax; p++) {
        const int rounds = (tail * p + 31) >> 5;
        if (rounds * bd < bn * p) { best = p; bn = rounds; bd = p; }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// Text side of the ingestion (eagle_ingest.cpp): line index of a memory-mapped file and the whitespace tokeniser.
// ------------------------------------------------------------------------------------------------
static inline void parallel_for(long n, int threads, const std::function<void(long, long, int)>& fn) {
    if (n <= 0) return;
    if (threads <= 1 || n < 2 * threads) { fn(0, n, 0); return; }
    std::vector<std::thread> pool;
    const long per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const long a = t * per, b = std::min(n, a + per);
        if (a >= b) break;
        pool.emplace_back(fn, a, b, t);
    }
    for (auto& th : pool) th.join();
}

// getline() semantics: lines end at '\n'; a non-empty tail without '\n' is a line too.  starts has nlines+1 entries,
// line i is [starts[i], starts[i+1] - 1) except for an unterminated last line, whose end is the file size (`tail`).
struct LineIndex {
    std::vector<size_t> starts;
    bool tail = false;
    size_t size = 0;
    long nlines() const { return (long)starts.size() - 1; }
    size_t begin(long i) const { return starts[(size_t)i]; }
    size_t end(long i) const { return (tail && i == nlines() - 1) ? size : starts[(size_t)i + 1] - 1; }
};

static inline void index_lines_buf(const char* base, size_t size, int threads, LineIndex& ix) {
    ix.size = size;
    ix.tail = false;
    std::vector<std::vector<size_t>> part((size_t)std::max(1, threads));
    parallel_for((long)size, threads, [&](long a, long b, int t) {
        auto& v = part[(size_t)t];
        const char* p = base + a;
        const char* e = base + b;
        while (p < e) {
            const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            v.push_back((size_t)(q - base) + 1);
            p = q + 1;
        }
    });
    ix.starts.clear();
    ix.starts.push_back(0);
    for (auto& v : part) ix.starts.insert(ix.starts.end(), v.begin(), v.end());
    if (ix.starts.back() < size) { ix.starts.push_back(size + 1); ix.tail = true; }
    if (size == 0) ix.starts.assign(1, 0);
}

static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
static inline const char* next_token(const char* p, const char* end, const char** tok, long* len) {
    while (p < end && is_ws(*p)) p++;
    if (p >= end) return nullptr;
    *tok = p;
    while (p < end && !is_ws(*p)) p++;
    *len = p - *tok;
    return p;
}
static inline long count_tokens(const char* p, const char* end) {
    const char* tok;
    long len, n = 0;
    while ((p = next_token(p, end, &tok, &len)) != nullptr) n++;
    return n;
}
#endif
