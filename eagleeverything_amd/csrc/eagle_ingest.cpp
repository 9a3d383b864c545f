// eagle_ingest.cpp -- marker-file ingestion next to the hot path: the three host
// text-conversion entry points of the reference's .Call table, rebuilt around the HBM-resident genotype copy.
//
//   eagle_get_row_column ...... E/src/getRowColumn.cpp:20-72
//   eagle_create_M_ascii ...... E/src/createM_ASCII_rcpp.cpp:18-106 -> CreateASCIInospace.cpp:17-163 (text)
//                                                                   -> CreateASCIInospace_PLINK.cpp:16-249 (PLINK ped)
//   eagle_create_Mt_ascii ..... E/src/createMt_ASCII_rcpp.cpp:14-247
//
// The reference tokenises one line at a time through istringstream, and builds Mt.ascii by re-reading M.ascii once per
// column block.  Here the input is mmap()ed, its lines are indexed and tokenised by `host_threads()` workers straight
// into pinned staging, and everything after tokenisation happens on the device: the PLINK allele table walk (one
// thread per locus, k_plink_code), the genotype decode, the transpose (k_transpose_i8) and the re-encoding of text
// lines (k_encode_ascii).  Both files are written with pwrite() from pinned memory, and -- the point of doing it here --
// the int8 images of M.ascii and Mt.ascii stay resident in HBM under the output paths, so the calculateMMt /
// calculate_a_and_vara calls that follow ReadMarker() never parse a text file at all.
// Messages and return values follow the reference (false -> EAGLE_SOFT_SENTINEL after the messages were sent).
#include <ctype.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <functional>

#include "eagle_ctx.h"

extern "C" int eagle_dev_encode_ascii(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, uint8_t* out, void* stream);
extern "C" int eagle_dev_plink_code(eagle_ctx* ctx, const uint8_t* chars, long rows, long L, long row0, uint8_t* alleles0,
                                    uint8_t* alleles1, int8_t* out, long ld, unsigned long long* first_err,
                                    unsigned long long* first_missing, void* stream);

namespace {

struct MappedFile {
    int fd = -1;
    const char* p = nullptr;
    size_t size = 0;
    ~MappedFile() {
        if (p && size) munmap((void*)p, size);
        if (fd >= 0) close(fd);
    }
};

bool map_file(const char* path, MappedFile& m) {
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return false;
    struct stat st;
    if (fstat(m.fd, &st) != 0) return false;
    m.size = (size_t)st.st_size;
    if (m.size == 0) return true;
    void* p = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (p == MAP_FAILED) { m.size = 0; return false; }
    m.p = (const char*)p;
    (void)madvise(p, m.size, MADV_SEQUENTIAL);
    return true;
}

// parallel_for, LineIndex, index_lines_buf, next_token, count_tokens: eagle_host.h (HIP-free, built under the CPU sanitizers)
void index_lines(const MappedFile& m, int threads, LineIndex& ix) { index_lines_buf(m.p, m.size, threads, ix); }

bool pwrite_all(int fd, const char* src, size_t bytes, off_t off, int threads) {
    std::atomic<bool> ok{true};
    parallel_for((long)bytes, bytes < ((size_t)8 << 20) ? 1 : threads, [&](long a, long b, int) {
        while (a < b) {
            ssize_t w = pwrite(fd, src + a, (size_t)(b - a), off + a);
            if (w <= 0) { ok = false; return; }
            a += w;
        }
    });
    return ok;
}

// the reference echoes the head of the input after converting it (CreateASCIInospace.cpp:139-158, _PLINK.cpp:201-235)
void say_head(eagle_ctx* ctx, const MappedFile& m, const LineIndex& ix, long nrows_file, long ncols_file, int maxcols, const char* what) {
    const long nrowsp = std::min(5L, nrows_file);
    const long ncolsp = std::min((long)maxcols, ncols_file);
    say(ctx, " First %ld lines and %ld columns of the %s. ", nrowsp, ncolsp, what);
    for (long r = 0; r < nrowsp && r < ix.nlines(); r++) {
        std::string row;
        const char *p = m.p + ix.begin(r), *e = m.p + ix.end(r), *tok;
        long len;
        for (long c = 0; c < ncolsp && (p = next_token(p, e, &tok, &len)) != nullptr; c++) { row.append(tok, (size_t)len); row.push_back(' '); }
        say(ctx, "%s", row.c_str());
    }
}

struct RowError {  // first failing row of a chunk (smallest row wins)
    long row = -1;
    int kind = 0;  // 1 unknown token, 2 unequal columns
    long cols = 0;
    std::string token;
};

// Does the whole int8 image (rows_pad x ld) fit beside what is already on the device?
bool fits_resident(size_t bytes) {
    if (bytes > eagle_resident_budget()) return false;
    size_t freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) return false;
    return bytes + ((size_t)2 << 30) < freeb;
}


// Incremental writer of the 2-bit sidecar "<text file>.e2b" (layout: E2bHeader in eagle_ctx.h).  Rows are packed on the
// device from the int8 image the caller already holds, copied to its own pinned buffers and written with pwrite();
// the header goes in last, with the size and mtime of the finished text file, and the file is renamed into place.
struct SidecarWriter {
    eagle_ctx* ctx = nullptr;
    int fd = -1;
    std::string final_path, tmp_path;
    long rows = 0, cols = 0, row_bytes = 0, cap_rows = 0;
    PinBuf pin[2];
    DevBuf dev[2];
    bool active() const { return fd >= 0; }
    ~SidecarWriter() { abandon(); }
    void abandon() {
        if (fd >= 0) { close(fd); fd = -1; (void)unlink(tmp_path.c_str()); }
    }
    bool open_for(eagle_ctx* c, const char* text_path, long nrows, long ncols, long max_rows_per_call) {
        if (!eagle_sidecar_enabled() || nrows <= 0 || ncols <= 0) return false;
        ctx = c; rows = nrows; cols = ncols; cap_rows = std::max(1L, max_rows_per_call);
        row_bytes = ((ncols + 3) / 4 + 15) / 16 * 16;
        final_path = std::string(text_path) + ".e2b";
        tmp_path = final_path + ".tmp";
        (void)unlink(final_path.c_str());  // a sidecar of an older text file of that name must not survive a failed run
        fd = open(tmp_path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fd < 0) return false;
        for (int b = 0; b < 2; b++)
            if (pin[b].alloc((size_t)cap_rows * row_bytes) != hipSuccess || dev[b].alloc((size_t)cap_rows * row_bytes) != hipSuccess) { abandon(); return false; }
        return true;
    }
    // enqueue on the ctx stream: pack `nrows` rows of the int8 tile and copy them to pinned buffer b
    int pack(int b, const int8_t* tile, long nrows, long ld) {
        if (!active() || nrows <= 0) return EAGLE_OK;
        if (nrows > cap_rows) return eagle_fail(ctx, EAGLE_ERR_ARG, "sidecar: chunk larger than announced");
        int rc = eagle_dev_pack2b(ctx, tile, nrows, cols, ld, dev[b].as<uint8_t>(), row_bytes, ctx->stream);
        if (rc) return rc;
        hipError_t e = hipMemcpyAsync(pin[b].p, dev[b].p, (size_t)nrows * row_bytes, hipMemcpyDeviceToHost, ctx->stream);
        return e == hipSuccess ? EAGLE_OK : eagle_fail_hip(ctx, e, "sidecar D2H");
    }
    // after the stream work of pack(b, ...) has completed
    void write(int b, long row0, long nrows, int threads) {
        if (!active() || nrows <= 0) return;
        if (!pwrite_all(fd, (const char*)pin[b].p, (size_t)nrows * row_bytes, (off_t)sizeof(E2bHeader) + (off_t)row0 * row_bytes, threads)) abandon();
    }
    void finish(const char* text_path) {
        if (!active()) return;
        struct stat st;
        if (stat(text_path, &st) != 0) { abandon(); return; }
        E2bHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, "EAGLE2B", 8);
        h.version = 1;
        h.rows = (uint64_t)rows; h.cols = (uint64_t)cols; h.row_bytes = (uint64_t)row_bytes;
        h.src_size = (uint64_t)st.st_size;
        h.src_mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000L + st.st_mtim.tv_nsec;
        const bool ok = pwrite(fd, &h, sizeof h, 0) == (ssize_t)sizeof h && ftruncate(fd, (off_t)sizeof h + (off_t)rows * row_bytes) == 0;
        close(fd);
        fd = -1;
        if (!ok || rename(tmp_path.c_str(), final_path.c_str()) != 0) (void)unlink(tmp_path.c_str());
    }
};

}  // namespace

extern "C" int eagle_get_row_column(eagle_ctx* ctx, const char* fname, long dims_out[2]) {
    if (!ctx || !fname || !dims_out) return EAGLE_ERR_ARG;
    MappedFile m;
    if (!map_file(fname, m)) return failf(ctx, EAGLE_ERR_OPEN, "\n\n ERROR: Could not open  %s\n\n", fname);  // getRowColumn.cpp:35-38
    LineIndex ix;
    index_lines(m, host_threads(), ix);
    dims_out[0] = ix.nlines();
    dims_out[1] = ix.nlines() > 0 ? count_tokens(m.p + ix.begin(0), m.p + ix.end(0)) : 0;
    return EAGLE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// text genotype file -> M.ascii
// ---------------------------------------------------------------------------------------------------------------
static int create_M_text(eagle_ctx* ctx, const char* fname, const char* asciifname, const char* AA, const char* AB, const char* BB,
                         const char* missing, const long dims[2], int quiet) {
    MappedFile m;
    if (!map_file(fname, m)) {
        say(ctx, "ERROR: Text file could not be opened with filename  %s\n", fname);  // CreateASCIInospace.cpp:42-45
        return EAGLE_SOFT_SENTINEL;
    }
    const int fdout = open(asciifname, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fdout < 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", asciifname);
    struct Closer { int fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fdout};
    if (!quiet) { say(ctx, ""); say(ctx, " Reading text File  "); say(ctx, ""); say(ctx, " Loading file "); }
    const int threads = host_threads();
    LineIndex ix;
    index_lines(m, threads, ix);
    const long nlines = ix.nlines(), L = dims[1], stride = L + 1;
    const size_t lAA = strlen(AA), lAB = strlen(AB), lBB = strlen(BB), lMS = strlen(missing);
    const bool single = lAA == 1 && lAB == 1 && lBB == 1;  // one-character codes: table lookup instead of compares

    const long chunk_rows = std::max(1L, std::min(std::max(nlines, 1L), (long)(67108864 / std::max(1L, stride))));
    int rc = eagle_stage_ensure(ctx, (size_t)chunk_rows * stride);
    if (rc) return rc;
    // resident image of M.ascii, filled as the chunks go by (only when the file is what dims says)
    const long n_pad = eagle_pad(nlines), ld = eagle_pad(L);
    int8_t* dev = nullptr;
    DevBuf bad;
    HIPCHK(ctx, bad.alloc(sizeof(int)));
    HIPCHK(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
    if (nlines == dims[0] && nlines > 0 && L > 0 && fits_resident((size_t)n_pad * ld)) {
        if (hipMalloc((void**)&dev, (size_t)n_pad * ld) != hipSuccess) dev = nullptr;
        else HIPCHK(ctx, hipMemsetAsync(dev, 0, (size_t)n_pad * ld, ctx->stream));
    }
    struct DevGuard { int8_t*& p; ~DevGuard() { if (p) (void)hipFree(p); } } guard{dev};
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; b++) if (e[b]) (void)hipEventDestroy(e[b]); } } evg{done};

    RowError err;
    long k = 0;
    for (long r0 = 0; r0 < nlines && err.row < 0; r0 += chunk_rows, k++) {
        const int b = (int)(k & 1);
        const long nr = std::min(chunk_rows, nlines - r0);
        char* buf = (char*)ctx->stage_pin[b];
        if (k >= 2) HIPCHK(ctx, hipEventSynchronize(done[b]));
        std::vector<RowError> terr((size_t)threads);
        parallel_for(nr, threads, [&](long a, long e, int t) {
            RowError& my = terr[(size_t)t];
            for (long r = a; r < e; r++) {
                const char *p = m.p + ix.begin(r0 + r), *end = m.p + ix.end(r0 + r), *tok;
                char* out = buf + r * stride;
                long len, i = 0;
                while ((p = next_token(p, end, &tok, &len)) != nullptr) {
                    char c;
                    if (single && len == 1) {
                        const char ch = *tok;
                        if (ch == BB[0]) c = '2';
                        else if (ch == AB[0]) c = '1';
                        else if (ch == AA[0]) c = '0';
                        else if (lMS == 1 && ch == missing[0]) c = '1';
                        else c = 0;
                    } else if ((size_t)len == lBB && memcmp(tok, BB, lBB) == 0) c = '2';        // CreateASCIInospace.cpp:95
                    else if ((size_t)len == lAB && memcmp(tok, AB, lAB) == 0) c = '1';          // :97
                    else if ((size_t)len == lAA && memcmp(tok, AA, lAA) == 0) c = '0';          // :99
                    else if ((size_t)len == lMS && memcmp(tok, missing, lMS) == 0) c = '1';     // :101-103
                    else c = 0;
                    if (!c) { my.row = r0 + r; my.kind = 1; my.token.assign(tok, (size_t)len); return; }
                    if (i < L) out[i] = c;
                    i++;
                }
                if (i != L) { my.row = r0 + r; my.kind = 2; my.cols = i; return; }              // :122-131
                out[L] = '\n';
            }
        });
        for (auto& e : terr)
            if (e.row >= 0 && (err.row < 0 || e.row < err.row)) err = e;
        const long good = err.row < 0 ? nr : err.row - r0;  // rows of this chunk written before the failure
        if (good > 0 && !pwrite_all(fdout, buf, (size_t)good * stride, (off_t)r0 * stride, threads))
            return failf(ctx, EAGLE_ERR_OPEN, "ERROR: could not write %s", asciifname);
        if (dev && err.row < 0) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->stage_raw[b], buf, (size_t)nr * stride, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipEventRecord(done[b], ctx->stream));
            rc = eagle_dev_decode_ascii(ctx, (const uint8_t*)ctx->stage_raw[b], nr, L, stride, dev + r0 * ld, ld, bad.as<int>(), ctx->stream);
            if (rc) return rc;
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (err.row >= 0) {
        if (err.kind == 1) {                                                                   // :104-116
            if (strcmp(AB, "NA") == 0) say(ctx, "\n Marker file contains marker genotypes that are different to AA=%s BB=%s", AA, BB);
            else say(ctx, "\n Marker file contains marker genotypes that are different to AA=%s AB=%s BB=%s", AA, AB, BB);
            say(ctx, " For example , %s in row %ld", err.token.c_str(), err.row + 1);
            say(ctx, "\n ReadMarker has terminated with errors\n");
        } else {
            say(ctx, "\n");
            say(ctx, "Error:  Marker text file contains an unequal number of columns per row.  ");
            say(ctx, "        The error has occurred at row %ld which contains %ld but ", err.row + 1, err.cols);
            say(ctx, "        it should contain %ld columns of data. ", L);
            say(ctx, "\n");
            say(ctx, " ReadMarkerData has terminated with errors");
        }
        snprintf(ctx->err, sizeof ctx->err, "createM_ASCII: %s at row %ld", err.kind == 1 ? "unknown genotype token" : "unequal number of columns", err.row + 1);
        return EAGLE_SOFT_SENTINEL;
    }
    say_head(ctx, m, ix, dims[0], dims[1], 12, "marker text  file");
    if (dev) {
        close(closer.fd);
        closer.fd = -1;  // the text file is final: its size and mtime key the sidecar and the cache entry
        SidecarWriter sc;
        if (sc.open_for(ctx, asciifname, nlines, L, chunk_rows)) {
            for (long r0 = 0; r0 < nlines && sc.active(); r0 += chunk_rows) {
                const long nr = std::min(chunk_rows, nlines - r0);
                rc = sc.pack(0, dev + r0 * ld, nr, ld);
                if (rc) return rc;
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                sc.write(0, r0, nr, threads);
            }
            sc.finish(asciifname);
        }
        int8_t* give = dev;
        dev = nullptr;
        return eagle_cache_adopt(ctx, asciifname, nlines, L, n_pad, ld, give);
    }
    return EAGLE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// PLINK ped -> M.ascii
// ---------------------------------------------------------------------------------------------------------------
static int create_M_plink(eagle_ctx* ctx, const char* fname, const char* asciifname, const long dims[2], int quiet) {
    MappedFile m;
    if (!map_file(fname, m)) {                                                                  // _PLINK.cpp:38-42
        say(ctx, "ERROR: PLINK ped file could not be opened with filename  %s", fname);
        say(ctx, "ERROR: ReadMarkerData has terminated with errors.  ");
        return EAGLE_SOFT_SENTINEL;
    }
    const int fdout = open(asciifname, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fdout < 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", asciifname);
    struct Closer { int fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fdout};
    const int threads = host_threads();
    LineIndex ix;
    index_lines(m, threads, ix);
    const long nlines = ix.nlines();
    const long ncols_total = dims[1];
    const long L = (long)((ncols_total - 6) / 2.0);                                             // :20
    if (L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "createM_ASCII (PLINK): dims[1] must be 6 + 2 * loci");
    const long in_stride = 2 * L, out_stride = L + 1;
    const long chunk_rows = std::max(1L, std::min(std::max(nlines, 1L), (long)(67108864 / in_stride)));
    int rc = eagle_stage_ensure(ctx, (size_t)chunk_rows * in_stride);
    if (rc) return rc;
    const long n_pad = eagle_pad(nlines), ld = eagle_pad(L);
    const bool keep = nlines > 0 && fits_resident((size_t)n_pad * ld);
    DevBuf image, alleles;
    const long img_rows = keep ? n_pad : eagle_pad(chunk_rows);
    HIPCHK(ctx, image.alloc((size_t)img_rows * ld));
    HIPCHK(ctx, hipMemsetAsync(image.p, 0, (size_t)img_rows * ld, ctx->stream));
    HIPCHK(ctx, alleles.alloc((size_t)2 * L));
    HIPCHK(ctx, hipMemsetAsync(alleles.p, 0, (size_t)2 * L, ctx->stream));
    unsigned long long* flags = (unsigned long long*)((char*)eagle_ctx_scratch(ctx) + EAGLE_SCR_INGEST);   // [0] first third-allele, [1] first missing
    HIPCHK(ctx, hipMemsetAsync(flags, 0xff, 2 * sizeof(unsigned long long), ctx->stream));

    SidecarWriter sc;
    (void)sc.open_for(ctx, asciifname, nlines, L, chunk_rows);
    RowError err;          // unequal number of columns (found on the host)
    unsigned long long h_flags[2] = {~0ull, ~0ull};
    for (long r0 = 0; r0 < nlines; r0 += chunk_rows) {
        const long nr = std::min(chunk_rows, nlines - r0);
        char* cin = (char*)ctx->stage_pin[0];
        std::vector<RowError> terr((size_t)threads);
        parallel_for(nr, threads, [&](long a, long e, int t) {
            RowError& my = terr[(size_t)t];
            for (long r = a; r < e; r++) {
                const char *p = m.p + ix.begin(r0 + r), *end = m.p + ix.end(r0 + r), *tok;
                const long numcols = count_tokens(p, end);                                       // :60-63
                if (numcols != ncols_total) { my.row = r0 + r; my.kind = 2; my.cols = numcols; return; }
                long len;
                for (int i = 0; i <= 5; i++) p = next_token(p, end, &tok, &len);                 // :85-87
                char* out = cin + r * in_stride;
                for (long i = 0; i < in_stride; i++) {                                           // :88-90 one character per read
                    while (p < end && is_ws(*p)) p++;
                    out[i] = p < end ? *p++ : 0;
                }
            }
        });
        for (auto& e : terr)
            if (e.row >= 0 && (err.row < 0 || e.row < err.row)) err = e;
        const long good = err.row < 0 ? nr : err.row - r0;  // rows in front of a malformed line are still coded
        if (good > 0) {
            int8_t* img = image.as<int8_t>() + (keep ? r0 * ld : 0);
            HIPCHK(ctx, hipMemcpyAsync(ctx->stage_raw[0], cin, (size_t)good * in_stride, hipMemcpyHostToDevice, ctx->stream));
            rc = eagle_dev_plink_code(ctx, (const uint8_t*)ctx->stage_raw[0], good, L, r0, alleles.as<uint8_t>(), alleles.as<uint8_t>() + L, img, ld,
                                      flags, flags + 1, ctx->stream);
            if (rc) return rc;
            rc = eagle_dev_encode_ascii(ctx, img, good, L, ld, (uint8_t*)ctx->stage_raw[1], ctx->stream);
            if (rc) return rc;
            HIPCHK(ctx, hipMemcpyAsync(ctx->stage_pin[1], ctx->stage_raw[1], (size_t)good * out_stride, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(h_flags, flags, sizeof h_flags, hipMemcpyDeviceToHost, ctx->stream));
            rc = sc.pack(0, img, good, ld);
            if (rc) return rc;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            sc.write(0, r0, good, threads);
            long ok_rows = good;
            if (h_flags[0] != ~0ull) ok_rows = (long)(h_flags[0] / (unsigned long long)L) - r0;  // the reference stops inside that row
            if (ok_rows > 0 && !pwrite_all(fdout, (const char*)ctx->stage_pin[1], (size_t)ok_rows * out_stride, (off_t)r0 * out_stride, threads))
                return failf(ctx, EAGLE_ERR_OPEN, "ERROR: could not write %s", asciifname);
        }
        if (err.row >= 0 || h_flags[0] != ~0ull) break;
    }
    const bool allele_err = h_flags[0] != ~0ull;
    const unsigned long long stop = allele_err ? h_flags[0] : (err.row >= 0 ? (unsigned long long)err.row * (unsigned long long)L : ~0ull);
    if (h_flags[1] != ~0ull && h_flags[1] < stop) {                                              // :112-121, printed once
        say(ctx, "\n");
        say(ctx, " Warning:  PLINK file contains missing alleles (i.e. 0 or - ) ");
        say(ctx, "           These missing genotypes should be imputed before running Eagle.");
        say(ctx, "           As an approximation, AMpus has set these missing genotypes to heterozygotes. ");
        say(ctx, "           Since Eagle assumes an additive model, heterozygote genotypes do not contribute to the estimation of ");
        say(ctx, "           the additive effects.  ");
        say(ctx, "\n");
    }
    if (allele_err) {                                                                            // :155-161
        const long row = (long)(h_flags[0] / (unsigned long long)L), locus = (long)(h_flags[0] % (unsigned long long)L);
        say(ctx, "\n");
        say(ctx, "Error:  PLINK file cannot contain more than two alleles at a locus.");
        say(ctx, "        The error has occurred at snp locus %ld for individual %ld", locus + 1, row + 1);
        say(ctx, "\n");
        say(ctx, " ReadMarkerData has terminated with errors");
        snprintf(ctx->err, sizeof ctx->err, "createM_ASCII: more than two alleles at locus %ld, individual %ld", locus + 1, row + 1);
        return EAGLE_SOFT_SENTINEL;
    }
    if (err.row >= 0) {                                                                          // :65-74
        say(ctx, "\n");
        say(ctx, "Error:  PLINK file contains an unequal number of columns per row.  ");
        say(ctx, "        The error has occurred at row %ld which contains %ld but ", err.row + 1, err.cols);
        say(ctx, "        it should contain %ld columns of data. ", ncols_total);
        say(ctx, "\n");
        say(ctx, " ReadMarkerData has terminated with errors");
        snprintf(ctx->err, sizeof ctx->err, "createM_ASCII: unequal number of columns at row %ld", err.row + 1);
        return EAGLE_SOFT_SENTINEL;
    }
    say_head(ctx, m, ix, dims[0], dims[1], dims[1] < 25 ? (int)dims[1] : 24, "PLINK ped file");  // :205-214
    close(closer.fd);
    closer.fd = -1;
    sc.finish(asciifname);
    if (keep && nlines == dims[0]) {
        int8_t* give = image.as<int8_t>();
        image.p = nullptr;
        return eagle_cache_adopt(ctx, asciifname, nlines, L, n_pad, ld, give);
    }
    return EAGLE_OK;
}

extern "C" int eagle_create_M_ascii(eagle_ctx* ctx, const char* f_name, const char* f_name_ascii, const char* type, const char* AA,
                                    const char* AB, const char* BB, double max_memory_in_Gbytes, const long dims[2], int quiet,
                                    const char* missing) {
    if (!ctx || !f_name || !f_name_ascii || !type || !dims) return EAGLE_ERR_ARG;
    if (dims[0] < 0 || dims[1] < 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "createM_ASCII: negative dims");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    (void)max_memory_in_Gbytes;  // both memory branches of the reference call the same converter (createM_ASCII_rcpp.cpp:88-95)
    if (strcmp(type, "PLINK") == 0) return create_M_plink(ctx, f_name, f_name_ascii, dims, quiet);
    if (!quiet) say(ctx, " A text file is being assumed as the input data file type. ");         // createM_ASCII_rcpp.cpp:85-86
    if (!AA || !AB || !BB || !missing) return eagle_fail(ctx, EAGLE_ERR_ARG, "createM_ASCII: AA, AB, BB and missing must be strings");
    return create_M_text(ctx, f_name, f_name_ascii, AA, AB, BB, missing, dims, quiet);
}

// ---------------------------------------------------------------------------------------------------------------
// M.ascii -> Mt.ascii
// ---------------------------------------------------------------------------------------------------------------
extern "C" int eagle_create_Mt_ascii(eagle_ctx* ctx, const char* f_name, const char* f_name_ascii, const char* type,
                                     double max_memory_in_Gbytes, const long dims[2], int quiet) {
    if (!ctx || !f_name || !f_name_ascii || !dims) return EAGLE_ERR_ARG;
    const long n = dims[0], L = dims[1];
    if (n <= 0 || L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "createMt_ASCII: dims must be positive");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int threads = host_threads();
    const int fdout = open(f_name_ascii, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fdout < 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", f_name_ascii);
    struct Closer { int fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fdout};
    const long out_stride = n + 1;
    if (ftruncate(fdout, (off_t)L * out_stride) != 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: could not size %s", f_name_ascii);

    // source: the resident image of M.ascii (left by eagle_create_M_ascii or an earlier call), else column windows of the file
    const GenoEntry* src = nullptr;
    int rc = eagle_get_resident(ctx, f_name, n, L, max_memory_in_Gbytes, threads, &src);
    if (rc != EAGLE_OK && rc != 2) return rc;                                                    // createMt_ASCII_rcpp.cpp:79-83 Rcpp::stop
    const long n_pad = eagle_pad(n), ldn = eagle_pad(n), L_pad = eagle_pad(L);
    // marker window: its text (w x (n+1)) fits one staging buffer; multiple of 256 markers
    long w = (long)(67108864 / out_stride) / 256 * 256;
    if (!src && eagle_resident_budget() != (size_t)-1)  // streaming under a resident budget: the window image obeys it too
        w = std::min(w, (long)(eagle_resident_budget() / (size_t)n_pad) / 256 * 256);
    if (w < 256) w = 256;
    if (w > L_pad) w = L_pad;
    rc = eagle_stage_ensure(ctx, (size_t)w * out_stride);
    if (rc) return rc;
    if (!src && !quiet) {                                                                        // :125-129
        say(ctx, " A block transpose is being performed due to lack of memory.  ");
        say(ctx, " Memory parameter availmemGb is set to %g gigabytes", max_memory_in_Gbytes);
        say(ctx, " If possible, increase availmemGb parameter. ");
    }
    const bool keep = fits_resident((size_t)L_pad * ldn);
    DevBuf mt, win;
    HIPCHK(ctx, mt.alloc(keep ? (size_t)L_pad * ldn : (size_t)w * ldn));
    if (keep) HIPCHK(ctx, hipMemsetAsync(mt.p, 0, (size_t)L_pad * ldn, ctx->stream));
    if (!src) HIPCHK(ctx, win.alloc((size_t)n_pad * w));
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; b++) if (e[b]) (void)hipEventDestroy(e[b]); } } evg{done};

    SidecarWriter sc;
    (void)sc.open_for(ctx, f_name_ascii, L, n, w);
    // Window k: transpose + encode on the stream, D2H into pinned buffer k&1; the pwrite of window k-1 overlaps it.
    long pend_c0 = -1, pend_rows = 0;
    int pend_b = 0;
    auto flush = [&]() -> int {
        if (pend_c0 < 0) return EAGLE_OK;
        hipError_t e = hipEventSynchronize(done[pend_b]);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "hipEventSynchronize");
        if (!pwrite_all(fdout, (const char*)ctx->stage_pin[pend_b], (size_t)pend_rows * out_stride, (off_t)pend_c0 * out_stride, threads))
            return failf(ctx, EAGLE_ERR_OPEN, "ERROR: could not write %s", f_name_ascii);
        sc.write(pend_b, pend_c0, pend_rows, threads);
        pend_c0 = -1;
        return EAGLE_OK;
    };
    long k = 0;
    for (long c0 = 0; c0 < L; c0 += w, k++) {
        const int b = (int)(k & 1);
        const long wc = std::min(w, L_pad - c0), real = std::min(w, L - c0);
        const int8_t* sp;
        long sld;
        if (src) { sp = src->dev + c0; sld = src->ld; }
        else {
            rc = flush();  // the tile loader below uses the same staging buffers as the pending write
            if (rc) return rc;
            HIPCHK(ctx, hipMemsetAsync(win.p, 0, (size_t)n_pad * w, ctx->stream));
            rc = eagle_dev_load_ascii(ctx, f_name, 0, n, c0, real, win.as<int8_t>(), w, max_memory_in_Gbytes, threads);
            if (rc) return rc;
            sp = win.as<int8_t>();
            sld = w;
        }
        int8_t* dst = mt.as<int8_t>() + (keep ? c0 * ldn : 0);
        rc = eagle_dev_transpose_i8(ctx, sp, n_pad, wc, sld, dst, ldn, ctx->stream);
        if (rc) return rc;
        rc = eagle_dev_encode_ascii(ctx, dst, real, n, ldn, (uint8_t*)ctx->stage_raw[b], ctx->stream);
        if (rc) return rc;
        if (pend_c0 >= 0 && pend_b == b) { rc = flush(); if (rc) return rc; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage_pin[b], ctx->stage_raw[b], (size_t)real * out_stride, hipMemcpyDeviceToHost, ctx->stream));
        rc = sc.pack(b, dst, real, ldn);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(done[b], ctx->stream));
        const long this_c0 = c0, this_rows = real;
        rc = flush();  // window k-1 goes to disk while window k is on the device
        if (rc) return rc;
        pend_c0 = this_c0; pend_rows = this_rows; pend_b = b;
    }
    rc = flush();
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // createMt_ASCII_rcpp.cpp:227-243
    say(ctx, "\n\n                    Summary of Marker File  ");
    say(ctx, "                   ~~~~~~~~~~~~~~~~~~~~~~~~   ");
    say(ctx, " File type:                   %s", type ? type : "");
    say(ctx, " Reformatted ASCII file name:  %s", f_name);
    say(ctx, " Number of individuals:        %ld", n);
    say(ctx, " Number of loci:               %ld", L);
    say(ctx, " File size (gigabytes):       %g", 3.5 * (double)n * (double)L * 3.0 / 1000000000.0);  // bits_in_int/8 = 31/8 = 3 (integer division)
    say(ctx, " Available memory (gigabytes): %g", max_memory_in_Gbytes);
    say(ctx, "\n\n");
    say(ctx, " The marker file has been Uploaded");
    close(closer.fd);
    closer.fd = -1;  // the text file is final: its size and mtime key the sidecar and the cache entry
    sc.finish(f_name_ascii);
    if (keep) {
        int8_t* give = mt.as<int8_t>();
        mt.p = nullptr;
        return eagle_cache_adopt(ctx, f_name_ascii, L, n, L_pad, ldn, give);
    }
    return EAGLE_OK;
}
