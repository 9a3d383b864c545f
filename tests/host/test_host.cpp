// CPU test binary for the HIP-free host pieces of libeaglehip.so (csrc/eagle_host.h), built by tests/test_host_sanitizers.py
// with -fsanitize=address,undefined and, for the Rendezvous cases, -fsanitize=thread.  Exit code 0 = every check passed.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <string>

#include "../../eagleeverything_amd/csrc/eagle_host.h"

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
    } while (0)

// ---- Rendezvous --------------------------------------------------------------------------------------------------
// rounds of n workers; worker `bad` reports a failure at its arrival for round `bad_round` (and, like the library's workers,
// keeps arriving).  Returns per worker the verdict it saw per round.
static std::vector<std::vector<int>> run_rounds(int n, int rounds, int bad, int bad_round, bool bad_is_fast) {
    Rendezvous rv;
    rv.n = n;
    std::vector<std::vector<int>> seen((size_t)n, std::vector<int>((size_t)rounds, -1));
    std::vector<std::thread> th;
    for (int k = 0; k < n; k++)
        th.emplace_back([&, k] {
            for (int r = 0; r < rounds; r++) {
                // the slow peers dawdle AFTER waking up from a round, which is when the fast device of the advisor's scenario
                // has already failed and arrived for the next round
                if (bad_is_fast && k != bad) std::this_thread::sleep_for(std::chrono::milliseconds(2));
                const bool ok = !(k == bad && r >= bad_round);
                seen[(size_t)k][(size_t)r] = rv.arrive(ok, (double)(r * 100 + k), (long)(r + k)) ? 1 : 0;
                if (seen[(size_t)k][(size_t)r] == 1 && rv.vmax != (double)(r * 100 + n - 1)) g_fail++;  // the round's maximum, stable until the next arrival
                // ... and the round's SUM (round 4: the markers over the tight threshold, added up over the devices of a scan)
                if (seen[(size_t)k][(size_t)r] == 1 && rv.vsum != (long)n * r + (long)n * (n - 1) / 2) g_fail++;
            }
        });
    for (auto& t : th) t.join();
    return seen;
}

static void test_rendezvous() {
    // no failure: every round true
    for (auto& w : run_rounds(4, 50, -1, 0, false)) for (int v : w) CHECK(v == 1);
    // a device fails between round 2 and round 3 and is FAST (the sticky flag, read after the wake-up, made a slow peer
    // leave round 2 with `false`, skip its round-3 arrival and strand the failed device): every worker must see rounds 0..2 true
    // and rounds 3.. false, and nobody may hang (the join above returning is that check)
    for (int trial = 0; trial < 20; trial++) {
        auto seen = run_rounds(3, 6, 0, 3, true);
        for (auto& w : seen) {
            for (int r = 0; r < 3; r++) CHECK(w[(size_t)r] == 1);
            for (int r = 3; r < 6; r++) CHECK(w[(size_t)r] == 0);
        }
    }
    // the library's control flow: a worker that sees `false` returns early -- legal only if every worker sees the SAME verdict
    // for the same round (else the counts of arrivals diverge and someone waits for ever)
    for (int trial = 0; trial < 20; trial++) {
        Rendezvous rv;
        rv.n = 4;
        std::atomic<int> exits_at[8];
        for (auto& e : exits_at) e = 0;
        std::vector<std::thread> th;
        for (int k = 0; k < 4; k++)
            th.emplace_back([&, k] {
                for (int r = 0; r < 8; r++) {
                    if (k != 1) std::this_thread::sleep_for(std::chrono::microseconds(300 * (k + 1)));
                    const bool ok = !(k == 1 && r == 4);
                    if (!rv.arrive(ok)) { exits_at[r]++; return; }
                }
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < 8; r++) CHECK(exits_at[r] == (r == 4 ? 4 : 0));
    }
    // NaN and -inf values do not disturb the maximum
    {
        Rendezvous rv;
        rv.n = 1;
        CHECK(rv.arrive(true, NAN) && rv.vmax == -HUGE_VAL);
        CHECK(rv.arrive(true, 3.0) && rv.vmax == 3.0);
        CHECK(rv.arrive(true) && rv.vmax == -HUGE_VAL && rv.vsum == 0);
        CHECK(rv.arrive(true, 1.0, 7) && rv.vsum == 7);
        CHECK(rv.arrive(true) && rv.vsum == 0);   // a round's sum does not leak into the next
    }
}

// ---- marker split, selected_loci, chunk rows, tail pieces ---------------------------------------------------------------
static void test_small_rules() {
    std::vector<long> e;
    for (long L : {1L, 255L, 256L, 257L, 1000L, 150001L, 1000000L})
        for (int nd : {1, 2, 3, 5, 8}) {
            split_markers(L, nd, e);
            CHECK((int)e.size() == nd + 1 && e[0] == 0 && e[(size_t)nd] == L);
            for (int k = 0; k < nd; k++) { CHECK(e[(size_t)k] <= e[(size_t)k + 1]); if (k) CHECK(e[(size_t)k] % 256 == 0 || e[(size_t)k] == L); }
        }
    std::vector<long> out;
    const double na = NAN;
    double s1[] = {na, 3, 4};
    CHECK(parse_selected_core(s1, 3, 10, out) == nullptr && out.empty());       // AM()'s c(NA, i1, i2): masking never fires
    double s2[] = {3, 4, 9};
    CHECK(parse_selected_core(s2, 3, 10, out) == nullptr && out.size() == 3 && out[2] == 9);
    double s3[] = {3, na};
    CHECK(parse_selected_core(s3, 2, 10, out) != nullptr);
    double s4[] = {10};
    CHECK(parse_selected_core(s4, 1, 10, out) != nullptr);
    double s5[] = {-1};
    CHECK(parse_selected_core(s5, 1, 10, out) != nullptr);
    CHECK(parse_selected_core(nullptr, 0, 10, out) == nullptr && out.empty());
    CHECK(stream_chunk_rows_core((size_t)-1, 50176, 1 << 30) == (long)(((size_t)8 << 30) / 2 / 50176) / 256 * 256);
    CHECK(stream_chunk_rows_core(1000, 50176, 100000) == 256);                 // never below one tile of rows
    CHECK(stream_chunk_rows_core((size_t)1 << 40, 256, 2048) == 2048);         // never above the file
    CHECK(stream_chunk_rows_core(616562688, 50176, 625152) == 6144);           // the C4 shard in >= 100 chunks
    for (int npair : {1, 2, 5, 20, 100})
        for (int tail = 0; tail < 32; tail++) {
            const int p = vara_tail_pieces(tail, npair);
            CHECK(p >= 1 && p <= std::max(1, std::min(npair, VARA_TAIL_PMAX)));
            for (int q = 1; tail && q <= std::min(npair, VARA_TAIL_PMAX); q++)
                CHECK(((tail * p + 31) >> 5) * q <= ((tail * q + 31) >> 5) * p);
        }
    CHECK(vara_tail_pieces(3, 20) == 10);
}

// ---- line index + tokeniser ----------------------------------------------------------------------------------------
static void test_text() {
    struct Case { std::string text; long nlines; };
    const Case cases[] = {{"", 0}, {"\n", 1}, {"a b\n", 1}, {"a b", 1}, {"a\n\nb\n", 3}, {"0 1 2\n2 1 0\n1 1", 3}, {std::string(100000, 'x') + "\n1 2\n", 2}};
    for (const Case& c : cases)
        for (int threads : {1, 2, 3, 7, 16}) {
            // exact-size heap copy: an over-read by a single byte is an ASan report
            char* buf = (char*)malloc(c.text.size() ? c.text.size() : 1);
            memcpy(buf, c.text.data(), c.text.size());
            LineIndex ix;
            index_lines_buf(buf, c.text.size(), threads, ix);
            CHECK(ix.nlines() == c.nlines);
            std::string rebuilt;
            for (long i = 0; i < ix.nlines(); i++) {
                CHECK(ix.begin(i) <= ix.end(i) && ix.end(i) <= c.text.size());
                rebuilt.append(buf + ix.begin(i), ix.end(i) - ix.begin(i));
                rebuilt.push_back('\n');
            }
            std::string want = c.text;
            if (!want.empty() && want.back() != '\n') want.push_back('\n');
            CHECK(rebuilt == want);
            free(buf);
        }
    // big random text, many threads: the per-thread pieces must concatenate to the sequential index
    std::string big;
    unsigned s = 12345;
    for (int i = 0; i < 400000; i++) { s = s * 1664525u + 1013904223u; big.push_back((s >> 24) % 11 == 0 ? '\n' : (char)('0' + (s >> 24) % 3)); }
    LineIndex a, b;
    index_lines_buf(big.data(), big.size(), 1, a);
    index_lines_buf(big.data(), big.size(), 13, b);
    CHECK(a.starts == b.starts && a.tail == b.tail);
    const char* line = " 0\t1  2\r\n";
    CHECK(count_tokens(line, line + strlen(line)) == 3);
    const char *tok, *p = line;
    long len = 0;
    p = next_token(p, line + strlen(line), &tok, &len);
    CHECK(p && len == 1 && *tok == '0');
    CHECK(count_tokens(line, line) == 0);
    char* tight = (char*)malloc(3);
    memcpy(tight, "A B", 3);                       // no terminator: the tokeniser must stop at `end`
    CHECK(count_tokens(tight, tight + 3) == 2);
    free(tight);
    long total = 0;
    parallel_for(1000, 7, [&](long x, long y, int) { static std::mutex m; std::lock_guard<std::mutex> g(m); total += y - x; });
    CHECK(total == 1000);
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "all" || what == "rendezvous") test_rendezvous();
    if (what == "all" || what == "rules") test_small_rules();
    if (what == "all" || what == "text") test_text();
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("host checks passed (%s)\n", what.c_str());
    return 0;
}
