// eagle_api.cpp -- host side of libeaglehip.so: context, genotype tile streamer, and the
// reference-shaped entry points of section 1 of include/eagle_hip.h.
//
// What is replaced (E/ = MyPackage/Eagle/ of the reference):
//   eagle_read_block ............. E/src/ReadBlock.cpp:16-68
//   eagle_calculateMMt ........... E/src/calculateMMt_rcpp.cpp:19-185
//   eagle_calculate_a_and_vara ... E/src/calculate_a_and_vara_rcpp.cpp:22-241
//   eagle_calculate_reduced_a .... E/src/calculate_reduced_a_rcpp.cpp:20-171
//   eagle_last_scan_argmax ....... E/R/find_qtl.R:71-83
//   eagle_last_mmt_normalised .... E/R/calcMMt.R:13
//
// Design: the reference re-parses the text file into an n x L (or L x n) matrix of doubles on every call and,
// when that does not fit `availmemGb`, re-scans the file from line 0 once per block.  Here a text tile is
// pread() into pinned memory by `num_cores` threads, copied to HBM, decoded on the device to int8 {-1,0,1} and
// kept resident (one byte per genotype: 10k x 1M = 10 GB of the 288 GB) for every later call on the same file.
// All arithmetic happens in the HIP kernels of eagle_kernels.hip; there is no host compute path.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "eagle_ctx.h"

thread_local char g_open_err[512];
extern "C" int eagle_fail(eagle_ctx* ctx, int code, const char* msg) {
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s", msg);
    return code;
}
extern "C" int eagle_fail_hip(eagle_ctx* ctx, hipError_t e, const char* where) {
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "HIP error in %s: %s", where, hipGetErrorString(e));
    return EAGLE_ERR_HIP;
}

// Grow-only device arena: the n x n operand images and kernel workspaces of a call are carved from one allocation that
// survives between calls (a find_qtl iteration would otherwise pay ~10 hipMalloc/hipFree pairs of 100s of MB each).
static int arena_reserve(eagle_ctx* ctx, size_t total) {
    ctx->arena_off = 0;
    if (total <= ctx->arena_cap) return EAGLE_OK;
    if (ctx->arena) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_cap = 0; }
    hipError_t e = hipMalloc(&ctx->arena, total);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "arena hipMalloc");
    ctx->arena_cap = total;
    return EAGLE_OK;
}
static size_t arena_round(size_t b) { return (b + 255) / 256 * 256; }
template <class T> static T* arena_take(eagle_ctx* ctx, size_t bytes) {
    T* p = (T*)((char*)ctx->arena + ctx->arena_off);
    ctx->arena_off += arena_round(bytes);
    return p;
}
static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static bool timing_on() { static int v = -1; if (v < 0) v = getenv("EAGLE_HIP_TIMING") ? 1 : 0; return v == 1; }

// ------------------------------------------------------------------------------------------------
extern "C" const char* eagle_open_error(void) { return g_open_err; }

extern "C" eagle_ctx* eagle_open(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_open_err, sizeof g_open_err, "no HIP device available (%s); libeaglehip has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_open_err, sizeof g_open_err, "device %d out of range (0..%d)", device, ndev - 1);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_open_err, sizeof g_open_err, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipSetDevice: %s", hipGetErrorString(e));
        return nullptr;
    }
    eagle_ctx* ctx = new eagle_ctx();
    ctx->device = device;
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    ctx->cu_count = prop.multiProcessorCount;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    if ((e = hipMalloc(&ctx->d_scratch, 4096)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipMalloc: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->load_stream, hipStreamNonBlocking)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    return ctx;
}

extern "C" void eagle_drop_cache(eagle_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& g : ctx->cache) { if (g.dev) (void)hipFree(g.dev); if (g.dev_s) (void)hipFree(g.dev_s); if (g.cshift) (void)hipFree(g.cshift); if (g.l1) (void)hipFree(g.l1); }
    ctx->cache.clear();
    if (ctx->f4_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->f4_buf); ctx->f4_buf = nullptr; ctx->f4_cap = 0; }
}

extern "C" void eagle_close(eagle_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    eagle_drop_cache(ctx);
    if (ctx->d_mmt) (void)hipFree(ctx->d_mmt);
    if (ctx->d_mmt_max) (void)hipFree(ctx->d_mmt_max);
    if (ctx->d_a) (void)hipFree(ctx->d_a);
    if (ctx->d_vara) (void)hipFree(ctx->d_vara);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->gemv_ws) (void)hipFree(ctx->gemv_ws);
    if (ctx->f4_buf) (void)hipFree(ctx->f4_buf);
    if (ctx->gemm_scratch) (void)hipFree(ctx->gemm_scratch);
    for (int b = 0; b < 2; b++) { if (ctx->stage_pin[b]) (void)hipHostFree(ctx->stage_pin[b]); if (ctx->stage_raw[b]) (void)hipFree(ctx->stage_raw[b]); }
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->load_stream) (void)hipStreamDestroy(ctx->load_stream);
    delete ctx;
}

extern "C" void* eagle_ctx_scratch(eagle_ctx* ctx) { return ctx->d_scratch; }
extern "C" void* eagle_ctx_f4_buffer(eagle_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->f4_cap) return ctx->f4_buf;
    if (ctx->f4_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->f4_buf); ctx->f4_buf = nullptr; ctx->f4_cap = 0; }
    hipError_t e = hipMalloc(&ctx->f4_buf, bytes);
    if (e != hipSuccess) { (void)eagle_fail_hip(ctx, e, "fp4 image hipMalloc"); ctx->f4_buf = nullptr; return nullptr; }
    ctx->f4_cap = bytes;
    return ctx->f4_buf;
}
extern "C" const char* eagle_last_error(eagle_ctx* ctx) { return ctx ? ctx->err : g_open_err; }
extern "C" void eagle_set_message_callback(eagle_ctx* ctx, eagle_message_fn fn, void* user) {
    if (ctx) { ctx->msg_fn = fn; ctx->msg_user = user; }
}
extern "C" int eagle_device_info(eagle_ctx* ctx, char* arch_out, int arch_len, int* cu_count, int64_t* hbm_bytes) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (arch_out && arch_len > 0) snprintf(arch_out, arch_len, "%s", ctx->arch);
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return EAGLE_OK;
}
extern "C" int eagle_set_scan_mode(eagle_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 1) return EAGLE_ERR_ARG;
    ctx->scan_mode = mode;
    return EAGLE_OK;
}
extern "C" int eagle_set_scan_slices(eagle_ctx* ctx, int nslices) {
    if (!ctx || nslices < 0 || nslices > 8) return EAGLE_ERR_ARG;
    ctx->scan_slices = nslices;
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// selected_loci rule (calculateMMt_rcpp.cpp:88; calculate_a_and_vara_rcpp.cpp:79; calculate_reduced_a_rcpp.cpp:74):
// masking fires iff element 0 is not NA.  NA arrives as NaN.
// ------------------------------------------------------------------------------------------------
static int parse_selected(eagle_ctx* ctx, const double* sel, long nsel, long bound, std::vector<long>& out) {
    out.clear();
    if (nsel <= 0 || !sel || isnan(sel[0])) return EAGLE_OK;
    for (long i = 0; i < nsel; i++) {
        if (isnan(sel[i])) return eagle_fail(ctx, EAGLE_ERR_ARG, "NA in selected_loci after element 0");
        long v = (long)sel[i];
        if (v < 0 || v >= bound) return eagle_fail(ctx, EAGLE_ERR_ARG, "selected_loci index out of range");
        out.push_back(v);
    }
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// Tile streamer: lines [row0, row0+nrows), characters [col0, col0+ncols) of a no-space ASCII genotype file
// -> int8 {-1,0,1} at dst[r*ld + c] in HBM (padding untouched; callers zero it).
// Fast path: fixed-width file (every line `width` characters + '\n'), pread() straight into pinned memory by
// `threads` workers, double-buffered against the H2D copy + decode kernel.  The decode kernel verifies the
// end-of-line byte of every row, so a file that is not fixed-width is detected, and the general path (scan
// for newlines on the host, copy the first `width` characters of each line) is used instead.
// ------------------------------------------------------------------------------------------------
struct FileInfo {
    int fd = -1;
    off_t size = 0;
    long mtime_ns = 0;
    long width = -1;   // characters per line if fixed-width, else -1
    long nlines = -1;
    ~FileInfo() { if (fd >= 0) close(fd); }
};

static int open_file(eagle_ctx* ctx, const char* path, FileInfo& fi) {
    fi.fd = open(path, O_RDONLY);
    if (fi.fd < 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", path);  // ReadBlock.cpp:42-45
    struct stat st;
    if (fstat(fi.fd, &st) != 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not stat  %s", path);
    fi.size = st.st_size;
    fi.mtime_ns = (long)st.st_mtim.tv_sec * 1000000000L + st.st_mtim.tv_nsec;
    // probe the first line
    char buf[1 << 16];
    long pos = 0, width = -1;
    while (pos < fi.size && width < 0) {
        ssize_t got = pread(fi.fd, buf, sizeof buf, pos);
        if (got <= 0) break;
        void* nl = memchr(buf, '\n', (size_t)got);
        if (nl) width = pos + ((char*)nl - buf);
        pos += got;
    }
    if (width >= 0 && fi.size % (width + 1) == 0) {
        fi.width = width;
        fi.nlines = fi.size / (width + 1);
    } else if (width >= 0 && (fi.size + 1) % (width + 1) == 0) {  // last line without '\n'
        fi.width = width;
        fi.nlines = (fi.size + 1) / (width + 1);
    }
    return EAGLE_OK;
}

static void parallel_pread(int fd, uint8_t* dst, long dst_stride, long nrows, long nbytes, off_t off0, long src_stride,
                           int threads, volatile int* io_err) {
    auto work = [&](long r0, long r1) {
        if (src_stride == dst_stride && nbytes == src_stride) {  // contiguous range
            long total = (r1 - r0) * src_stride, done = 0;
            while (done < total) {
                ssize_t got = pread(fd, dst + r0 * dst_stride + done, (size_t)(total - done), off0 + r0 * src_stride + done);
                if (got <= 0) {  // reading past EOF by the missing final '\n' is fine
                    if (got == 0 && total - done <= 1) { dst[r0 * dst_stride + done] = '\n'; break; }
                    *io_err = 1;
                    return;
                }
                done += got;
            }
            return;
        }
        for (long r = r0; r < r1; r++) {
            long done = 0;
            while (done < nbytes) {
                ssize_t got = pread(fd, dst + r * dst_stride + done, (size_t)(nbytes - done), off0 + r * src_stride + done);
                if (got <= 0) {
                    if (got == 0 && nbytes - done <= 1) { dst[r * dst_stride + done] = '\n'; break; }
                    *io_err = 1;
                    return;
                }
                done += got;
            }
        }
    };
    if (threads <= 1 || nrows < 2 * threads) { work(0, nrows); return; }
    std::vector<std::thread> pool;
    long per = (nrows + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        long r0 = t * per, r1 = std::min(nrows, r0 + per);
        if (r0 >= r1) break;
        pool.emplace_back(work, r0, r1);
    }
    for (auto& th : pool) th.join();
}

// Two pinned host buffers + two device buffers of at least `need` bytes each, owned by the ctx (grow-only).
int eagle_stage_ensure(eagle_ctx* ctx, size_t need) {
    if (need <= ctx->stage_cap) return EAGLE_OK;
    (void)hipStreamSynchronize(ctx->stream);
    for (int b = 0; b < 2; b++) {
        if (ctx->stage_pin[b]) { (void)hipHostFree(ctx->stage_pin[b]); ctx->stage_pin[b] = nullptr; }
        if (ctx->stage_raw[b]) { (void)hipFree(ctx->stage_raw[b]); ctx->stage_raw[b] = nullptr; }
    }
    ctx->stage_cap = 0;
    for (int b = 0; b < 2; b++) {
        HIPCHK(ctx, hipHostMalloc(&ctx->stage_pin[b], need, hipHostMallocDefault));
        HIPCHK(ctx, hipMalloc(&ctx->stage_raw[b], need));
    }
    ctx->stage_cap = need;
    return EAGLE_OK;
}

static int load_tile_fixed(eagle_ctx* ctx, FileInfo& fi, long row0, long nrows, long col0, long ncols, int8_t* dst,
                           long ld, double max_mem_gb, int threads) {
    const long line = fi.width + 1;
    // take the end-of-line byte along when the window reaches the end of the line: the decode kernel then
    // verifies it, which is what detects a file that is not fixed-width after all
    const bool at_end = (col0 + ncols == fi.width);
    const long src_bytes = at_end ? ncols + 1 : ncols;
    const long stride = src_bytes;
    // staging budget: a quarter of availmemGb per buffer, within [one row, 64 MiB]; the two pinned / device staging
    // buffers live in the ctx (page-locking 100s of MB per call costs more than the copy it feeds)
    double budget = max_mem_gb > 0 ? max_mem_gb * 1e9 / 4.0 : 64e6;
    long chunk_rows = (long)std::max(1.0, std::min(budget, 67108864.0) / (double)stride);
    chunk_rows = std::min(chunk_rows, nrows);
    int rcs = eagle_stage_ensure(ctx, (size_t)chunk_rows * stride);
    if (rcs) return rcs;
    char* pin[2] = {(char*)ctx->stage_pin[0], (char*)ctx->stage_pin[1]};
    uint8_t* raw[2] = {(uint8_t*)ctx->stage_raw[0], (uint8_t*)ctx->stage_raw[1]};
    DevBuf bad;
    hipEvent_t done[2] = {nullptr, nullptr};
    HIPCHK(ctx, bad.alloc(sizeof(int)));
    HIPCHK(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    int rc = EAGLE_OK;
    volatile int io_err = 0;
    long k = 0;
    for (long r = 0; r < nrows && rc == EAGLE_OK; r += chunk_rows, k++) {
        const int b = (int)(k & 1);
        const long nr = std::min(chunk_rows, nrows - r);
        if (k >= 2) {
            hipError_t e = hipEventSynchronize(done[b]);  // the copy out of pin[b] two chunks ago has finished
            if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipEventSynchronize"); break; }
        }
        parallel_pread(fi.fd, (uint8_t*)pin[b], stride, nr, src_bytes, (off_t)(row0 + r) * line + col0, line, threads,
                       &io_err);
        if (io_err) { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "short read: file has fewer lines than requested"); break; }
        hipError_t e = hipMemcpyAsync(raw[b], pin[b], (size_t)nr * stride, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipMemcpyAsync H2D"); break; }
        e = hipEventRecord(done[b], ctx->stream);
        if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipEventRecord"); break; }
        rc = eagle_dev_decode_ascii(ctx, raw[b], nr, ncols, stride, dst + r * ld, ld, bad.as<int>(), ctx->stream);
    }
    int nbad = 0;
    if (rc == EAGLE_OK) {
        hipError_t e = hipMemcpyAsync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "decode sync");
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (int b = 0; b < 2; b++) if (done[b]) (void)hipEventDestroy(done[b]);
    if (rc == EAGLE_OK && nbad)
        rc = failf(ctx, EAGLE_ERR_FORMAT, "%d characters outside '0'..'2' (or misplaced line ends) in the requested tile", nbad);
    return rc;
}

// Tile from the 2-bit sidecar "<path>.e2b" when there is a valid one (made from this very text file: same size and
// mtime): a quarter of the bytes to read.  Returns 1 when there is none (the caller parses the text), else a status.
static int load_tile_sidecar(eagle_ctx* ctx, const char* path, const FileInfo& fi, long row0, long nrows, long col0, long ncols,
                             int8_t* dst, long ld, int threads) {
    if (!eagle_sidecar_enabled()) return 1;
    std::string sp = std::string(path) + ".e2b";
    int fd = open(sp.c_str(), O_RDONLY);
    if (fd < 0) return 1;
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    E2bHeader h;
    if (pread(fd, &h, sizeof h, 0) != (ssize_t)sizeof h || memcmp(h.magic, "EAGLE2B", 8) != 0 || h.version != 1) return 1;
    if ((off_t)h.src_size != fi.size || h.src_mtime_ns != fi.mtime_ns) return 1;  // stale: the text file changed
    if ((uint64_t)(row0 + nrows) > h.rows || (uint64_t)(col0 + ncols) > h.cols) return 1;
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < sizeof h + h.rows * h.row_bytes) return 1;
    const long b0 = col0 / 4, b1 = (col0 + ncols + 3) / 4;
    const bool whole_rows = col0 == 0 && (uint64_t)ncols == h.cols;  // then the rows are one contiguous byte range
    const long nb = whole_rows ? (long)h.row_bytes : b1 - b0;
    const long stride = (nb + 15) / 16 * 16;
    long chunk_rows = std::max(1L, std::min(nrows, (long)(67108864 / stride)));
    int rc = eagle_stage_ensure(ctx, (size_t)chunk_rows * stride);
    if (rc) return rc;
    DevBuf bad;
    HIPCHK(ctx, bad.alloc(sizeof(int)));
    HIPCHK(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; b++) if (e[b]) (void)hipEventDestroy(e[b]); } } evg{done};
    volatile int io_err = 0;
    long k = 0;
    for (long r = 0; r < nrows; r += chunk_rows, k++) {
        const int b = (int)(k & 1);
        const long nr = std::min(chunk_rows, nrows - r);
        if (k >= 2) HIPCHK(ctx, hipEventSynchronize(done[b]));
        parallel_pread(fd, (uint8_t*)ctx->stage_pin[b], stride, nr, nb, (off_t)sizeof h + (off_t)(row0 + r) * (off_t)h.row_bytes + b0,
                       (long)h.row_bytes, threads, &io_err);
        if (io_err) { (void)hipStreamSynchronize(ctx->stream); return eagle_fail(ctx, EAGLE_ERR_FORMAT, "short read from the 2-bit sidecar"); }
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage_raw[b], ctx->stage_pin[b], (size_t)nr * stride, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(done[b], ctx->stream));
        rc = eagle_dev_unpack2b(ctx, (const uint8_t*)ctx->stage_raw[b], nr, ncols, stride, (int)(col0 % 4), dst + r * ld, ld, bad.as<int>(), ctx->stream);
        if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    }
    int nbad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nbad) return failf(ctx, EAGLE_ERR_FORMAT, "%d invalid genotype codes in %s", nbad, sp.c_str());
    return EAGLE_OK;
}

// General path: arbitrary line lengths.  Lines are located on the host; the first `col0+ncols` characters of
// each wanted line are required to exist (the reference indexes past the end of a short line: undefined).
static int load_tile_general(eagle_ctx* ctx, FileInfo& fi, long row0, long nrows, long col0, long ncols, int8_t* dst,
                             long ld) {
    FILE* f = fdopen(dup(fi.fd), "r");
    if (!f) return eagle_fail(ctx, EAGLE_ERR_OPEN, "fdopen failed");
    rewind(f);
    const long chunk_rows = std::max(1L, std::min(nrows, (long)(67108864 / std::max(1L, ncols))));
    PinBuf pin;
    DevBuf raw, bad;
    int rc = EAGLE_OK;
    hipError_t e;
    if ((e = pin.alloc((size_t)chunk_rows * ncols)) != hipSuccess || (e = raw.alloc((size_t)chunk_rows * ncols)) != hipSuccess ||
        (e = bad.alloc(sizeof(int))) != hipSuccess) {
        fclose(f);
        return eagle_fail_hip(ctx, e, "alloc");
    }
    (void)hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream);
    char* line = nullptr;
    size_t cap = 0;
    long filled = 0, out_row = 0;
    for (long rr = 0; rr < row0 + nrows && rc == EAGLE_OK; rr++) {
        ssize_t len = getline(&line, &cap, f);
        if (len < 0) { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "file has fewer lines than requested"); break; }
        if (rr < row0) continue;
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
        if (len < col0 + ncols) { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "line shorter than the requested columns"); break; }
        memcpy((char*)pin.p + filled * ncols, line + col0, (size_t)ncols);
        filled++;
        if (filled == chunk_rows || rr == row0 + nrows - 1) {
            e = hipMemcpyAsync(raw.p, pin.p, (size_t)filled * ncols, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "H2D"); break; }
            rc = eagle_dev_decode_ascii(ctx, raw.as<uint8_t>(), filled, ncols, ncols, dst + out_row * ld, ld, bad.as<int>(),
                                        ctx->stream);
            e = hipStreamSynchronize(ctx->stream);  // single staging buffer
            if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "sync"); break; }
            out_row += filled;
            filled = 0;
        }
    }
    free(line);
    fclose(f);
    int nbad = 0;
    if (rc == EAGLE_OK) {
        e = hipMemcpy(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "D2H");
        else if (nbad) rc = failf(ctx, EAGLE_ERR_FORMAT, "%d characters outside '0'..'2' in the requested tile", nbad);
    }
    return rc;
}

// public: load a window of a genotype text file into a caller-owned HBM int8 buffer
extern "C" int eagle_dev_load_ascii(eagle_ctx* ctx, const char* path, long row0, long nrows, long col0, long ncols,
                                    int8_t* dst, long ld, double max_mem_gb, int threads) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (row0 < 0 || nrows < 0 || col0 < 0 || ncols < 0 || ld % 4 || ncols > ld) return eagle_fail(ctx, EAGLE_ERR_ARG, "load_ascii: bad window");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FileInfo fi;
    int rc = open_file(ctx, path, fi);
    if (rc) return rc;
    if (nrows == 0 || ncols == 0) return EAGLE_OK;
    rc = load_tile_sidecar(ctx, path, fi, row0, nrows, col0, ncols, dst, ld, threads);
    if (rc != 1) return rc;
    if (fi.width >= 0) {
        if (row0 + nrows > fi.nlines) return eagle_fail(ctx, EAGLE_ERR_FORMAT, "file has fewer lines than requested");
        if (col0 + ncols > fi.width) return eagle_fail(ctx, EAGLE_ERR_FORMAT, "line shorter than the requested columns");
        rc = load_tile_fixed(ctx, fi, row0, nrows, col0, ncols, dst, ld, max_mem_gb, threads);
        if (rc != EAGLE_ERR_FORMAT) return rc;
        // a misplaced line end means the file is not fixed-width after all: fall through to the line scanner
    }
    return load_tile_general(ctx, fi, row0, nrows, col0, ncols, dst, ld);
}

// How many bytes of genotypes may stay resident per file: EAGLE_HIP_MAX_RESIDENT_GB (tests force the streamed path
// with it), otherwise whatever HBM has free.  Files above it are streamed through HBM in marker chunks.
#define EAGLE_STREAM 2
static size_t resident_budget() {
    const char* e = getenv("EAGLE_HIP_MAX_RESIDENT_GB");
    if (e && *e) return (size_t)(atof(e) * 1e9);
    return (size_t)-1;
}

// Resident genotype tile of a whole file: `rows` lines x first `cols` characters, zero padded to
// [pad128(rows)][pad128(cols)].
// Returns EAGLE_OK (*out set), EAGLE_STREAM (too large: the caller streams marker chunks) or an error.
// reserve_bytes: HBM the caller still needs for operands and workspaces.
static bool file_key(const char* path, off_t* size, long* mtime_ns) {
    struct stat st;
    if (stat(path, &st) != 0) return false;
    *size = st.st_size;
    *mtime_ns = (long)st.st_mtim.tv_sec * 1000000000L + st.st_mtim.tv_nsec;
    return true;
}
const GenoEntry* eagle_cache_find(eagle_ctx* ctx, const char* path, long rows, long cols) {
    off_t size; long mt;
    if (!file_key(path, &size, &mt)) return nullptr;
    for (auto& g : ctx->cache)
        if (g.path == path && g.size == size && g.mtime_ns == mt && g.rows == rows && g.cols == cols) return &g;
    return nullptr;
}
static void cache_drop_path(eagle_ctx* ctx, const char* path) {
    for (size_t i = 0; i < ctx->cache.size();)
        if (ctx->cache[i].path == path) {
            (void)hipFree(ctx->cache[i].dev);
            if (ctx->cache[i].dev_s) (void)hipFree(ctx->cache[i].dev_s);
            if (ctx->cache[i].cshift) (void)hipFree(ctx->cache[i].cshift);
            if (ctx->cache[i].l1) (void)hipFree(ctx->cache[i].l1);
            ctx->cache.erase(ctx->cache.begin() + i);
        } else i++;
}
int eagle_cache_adopt(eagle_ctx* ctx, const char* path, long rows, long cols, long rows_pad, long ld, int8_t* dev) {
    cache_drop_path(ctx, path);
    GenoEntry g;
    if (!file_key(path, &g.size, &g.mtime_ns)) { (void)hipFree(dev); return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", path); }
    g.path = path; g.rows = rows; g.cols = cols; g.rows_pad = rows_pad; g.ld = ld; g.dev = dev;
    ctx->cache.push_back(g);
    return EAGLE_OK;
}

static int get_resident(eagle_ctx* ctx, const char* path, long rows, long cols, double max_mem_gb, int threads,
                        GenoEntry** out, size_t reserve_bytes = (size_t)1 << 30) {
    off_t fsize; long mt;
    if (!file_key(path, &fsize, &mt)) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", path);
    if (const GenoEntry* hit = eagle_cache_find(ctx, path, rows, cols)) { *out = const_cast<GenoEntry*>(hit); return EAGLE_OK; }
    cache_drop_path(ctx, path);  // stale entries of the same path
    GenoEntry g;
    g.path = path; g.size = fsize; g.mtime_ns = mt; g.rows = rows; g.cols = cols;
    g.rows_pad = eagle_pad(rows); g.ld = eagle_pad(cols);
    size_t bytes = (size_t)g.rows_pad * (size_t)g.ld;
    if (bytes > resident_budget()) return EAGLE_STREAM;
    size_t freeb = 0, totalb = 0;
    HIPCHK(ctx, hipMemGetInfo(&freeb, &totalb));
    if (bytes + reserve_bytes > freeb) {
        eagle_drop_cache(ctx);
        HIPCHK(ctx, hipMemGetInfo(&freeb, &totalb));
        if (bytes + reserve_bytes > freeb) return EAGLE_STREAM;  // does not fit beside the operands: stream it
    }
    HIPCHK(ctx, hipMalloc((void**)&g.dev, bytes));
    hipError_t e = hipMemsetAsync(g.dev, 0, bytes, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(g.dev); return eagle_fail_hip(ctx, e, "memset"); }
    int rc = eagle_dev_load_ascii(ctx, path, 0, rows, 0, cols, g.dev, g.ld, max_mem_gb, threads);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(g.dev); return rc; }
    ctx->cache.push_back(g);
    *out = &ctx->cache.back();
    return EAGLE_OK;
}

int eagle_get_resident(eagle_ctx* ctx, const char* path, long rows, long cols, double max_mem_gb, int threads, const GenoEntry** out) {
    GenoEntry* g = nullptr;
    int rc = get_resident(ctx, path, rows, cols, max_mem_gb, threads, &g);
    *out = g;
    return rc;
}
size_t eagle_resident_budget() { return resident_budget(); }

// Rows (multiple of 256) of a streamed chunk whose padded row length is `row_bytes`.
static long stream_chunk_rows(long row_bytes, long total_rows_pad) {
    size_t budget = resident_budget();
    if (budget == (size_t)-1) budget = (size_t)8 << 30;  // 8 GiB chunks when streaming because HBM is full
    long rows = (long)(budget / 2 / (size_t)row_bytes) / 256 * 256;  // two chunk buffers share the budget
    if (rows < 256) rows = 256;
    return rows < total_rows_pad ? rows : total_rows_pad;
}

// Out-of-core streaming: chunk k+1 is read (pread -> pinned -> H2D -> decode, all on ctx->load_stream) while the kernels
// of chunk k run on ctx->stream.  Two chunk buffers; an event per buffer says when its kernels are done.
struct ChunkRing {
    int8_t* buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    long k = 0;
    ~ChunkRing() { for (int b = 0; b < 2; b++) if (done[b]) (void)hipEventDestroy(done[b]); }
    int init(eagle_ctx* ctx) {
        for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        return EAGLE_OK;
    }
    // returns the buffer holding the freshly loaded tile (zero padded to clear_bytes); the caller launches its kernels on
    // ctx->stream and then calls computed()
    int load(eagle_ctx* ctx, const char* path, long row0, long nrows, long col0, long ncols, long ld, size_t clear_bytes, double mem_gb,
             int threads, int8_t** out) {
        const int b = (int)(k & 1);
        if (k >= 2) HIPCHK(ctx, hipEventSynchronize(done[b]));  // the kernels that read this buffer two chunks ago
        hipStream_t main = ctx->stream;
        ctx->stream = ctx->load_stream;  // every loader below works on ctx->stream
        int rc = EAGLE_OK;
        hipError_t e = hipMemsetAsync(buf[b], 0, clear_bytes, ctx->stream);
        if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "chunk memset");
        if (!rc) rc = eagle_dev_load_ascii(ctx, path, row0, nrows, col0, ncols, buf[b], ld, mem_gb, threads);
        if (!rc && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "chunk load sync");
        ctx->stream = main;
        *out = buf[b];
        return rc;
    }
    int computed(eagle_ctx* ctx) {
        HIPCHK(ctx, hipEventRecord(done[(int)(k & 1)], ctx->stream));
        k++;
        return EAGLE_OK;
    }
};

// column-major n x n host matrix -> zero padded np x np device image (row-major image of the transpose)
static int upload_square(eagle_ctx* ctx, const double* host, long n, long np, double* dev) {
    HIPCHK(ctx, hipMemsetAsync(dev, 0, sizeof(double) * np * np, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dev, sizeof(double) * np, host, sizeof(double) * n, sizeof(double) * n, n,
                                 hipMemcpyHostToDevice, ctx->stream));
    return EAGLE_OK;
}
static int upload_vec(eagle_ctx* ctx, const double* host, long n, long np, double* dev) {
    HIPCHK(ctx, hipMemsetAsync(dev, 0, sizeof(double) * np, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dev, host, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int eagle_read_block(eagle_ctx* ctx, const char* asciifname, long start_row, long numcols, long numrows,
                                double* out) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (start_row < 0 || numcols < 0 || numrows < 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "negative dimension");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (numcols == 0 || numrows == 0) {
        FileInfo fi;
        return open_file(ctx, asciifname, fi);
    }
    const long ld = eagle_pad(numcols);
    DevBuf tile, dbl;
    HIPCHK(ctx, tile.alloc((size_t)numrows * ld));
    HIPCHK(ctx, dbl.alloc(sizeof(double) * (size_t)numrows * numcols));
    int rc = eagle_dev_load_ascii(ctx, asciifname, start_row, numrows, 0, numcols, tile.as<int8_t>(), ld, 1.0, 4);
    if (rc) return rc;
    rc = eagle_dev_i8_to_f64_colmajor(ctx, tile.as<int8_t>(), numrows, numcols, ld, dbl.as<double>(), ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, dbl.p, sizeof(double) * (size_t)numrows * numcols, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

extern "C" int eagle_calculateMMt(eagle_ctx* ctx, const char* f_name_ascii, double max_memory_in_Gbytes, int num_cores,
                                  const double* selected_loci, long n_selected, const long dims[2], int quiet,
                                  double* MMt_out) {
    if (!ctx) return EAGLE_ERR_ARG;
    const long n = dims[0], L = dims[1];
    if (n <= 0 || L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<long> sel;
    int rc = parse_selected(ctx, selected_loci, n_selected, L, sel);
    if (rc) return rc;
    say(ctx, " Number of cores being used for calculation is .. %d", num_cores);  // calculateMMt_rcpp.cpp:36
    // calculateMMt_rcpp.cpp:75-76,84,103-106: the reference chooses between one DGEMM and row blocks; a block
    // size of zero rows makes it divide by zero.  The device path needs no such split, but the failure is kept.
    double need = (double)((unsigned long)n * n * sizeof(double) + 2 * ((unsigned long)n * L * sizeof(double))) / 1e9;
    if (!(max_memory_in_Gbytes > need)) {
        double p2 = sqrt(4.0 * (double)L * (double)L + 4.0 * max_memory_in_Gbytes * 1e9 / sizeof(double));
        long rows_in_block = (long)((-2.0 * (double)L + p2) / 2.2);
        if (rows_in_block <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "availmemGb too small: zero rows per block");
        if (!quiet) say(ctx, "number of rows in block is %ld", rows_in_block);  // :107
    }
    const int threads = num_cores > 0 ? num_cores : 1;
    const long np = eagle_pad(n);
    GenoEntry* g = nullptr;
    rc = get_resident(ctx, f_name_ascii, n, L, max_memory_in_Gbytes, threads, &g, sizeof(int32_t) * (size_t)np * np * 2);
    if (rc < 0) return rc;
    DevBuf c32, dsel, win;
    HIPCHK(ctx, c32.alloc(sizeof(int32_t) * (size_t)np * np));
    HIPCHK(ctx, hipMemsetAsync(c32.p, 0, sizeof(int32_t) * (size_t)np * np, ctx->stream));
    if (rc == EAGLE_OK) {
        rc = eagle_dev_mmt_accumulate(ctx, g->dev, np, g->ld, g->ld, c32.as<int32_t>(), ctx->stream);
        if (rc) return rc;
        if (!sel.empty()) {  // :88-92 as an exact rank-k downdate
            HIPCHK(ctx, dsel.alloc(sizeof(long) * sel.size()));
            HIPCHK(ctx, hipMemcpyAsync(dsel.p, sel.data(), sizeof(long) * sel.size(), hipMemcpyHostToDevice, ctx->stream));
            rc = eagle_dev_mmt_downdate(ctx, g->dev, np, g->ld, dsel.as<long>(), (long)sel.size(), c32.as<int32_t>(), ctx->stream);
            if (rc) return rc;
        }
    } else {
        // M.ascii does not fit (or may not stay) in HBM: stream column windows of every line (= marker chunks) and
        // accumulate the exact integer partial products, MMt = sum_w M_w M_w^T.
        const long Lw = stream_chunk_rows(np, eagle_pad(L));  // window width in markers; rows of the window = np
        HIPCHK(ctx, win.alloc((size_t)2 * np * Lw));
        ChunkRing ring;
        if ((rc = ring.init(ctx))) return rc;
        ring.buf[0] = win.as<int8_t>();
        ring.buf[1] = win.as<int8_t>() + (size_t)np * Lw;
        if (!quiet) say(ctx, " M.ascii streamed through HBM in windows of %ld markers", Lw);
        for (long c0 = 0; c0 < L; c0 += Lw) {
            const long nc = std::min(Lw, L - c0);
            int8_t* wtile = nullptr;
            rc = ring.load(ctx, f_name_ascii, 0, n, c0, nc, Lw, (size_t)np * Lw, max_memory_in_Gbytes, threads, &wtile);
            if (rc) return rc;
            rc = eagle_dev_mmt_accumulate(ctx, wtile, np, Lw, Lw, c32.as<int32_t>(), ctx->stream);
            if (rc) return rc;
            std::vector<long> in_win;
            for (long c : sel) if (c >= c0 && c < c0 + nc) in_win.push_back(c - c0);
            if (!in_win.empty()) {
                DevBuf dw;
                HIPCHK(ctx, dw.alloc(sizeof(long) * in_win.size()));
                HIPCHK(ctx, hipMemcpyAsync(dw.p, in_win.data(), sizeof(long) * in_win.size(), hipMemcpyHostToDevice, ctx->stream));
                // duplicates of one column must be dropped across the whole list, which k_mmt_downdate does per call:
                // selected_loci entries are distinct columns in any sane call; duplicates inside one window are handled
                rc = eagle_dev_mmt_downdate(ctx, wtile, np, Lw, dw.as<long>(), (long)in_win.size(), c32.as<int32_t>(), ctx->stream);
                if (rc) return rc;
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
            if ((rc = ring.computed(ctx))) return rc;
        }
    }
    if (ctx->mmt_n != n) {
        if (ctx->d_mmt) { (void)hipFree(ctx->d_mmt); ctx->d_mmt = nullptr; }
        ctx->mmt_n = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_mmt, sizeof(double) * (size_t)n * n));
        ctx->mmt_n = n;
    }
    if (!ctx->d_mmt_max) HIPCHK(ctx, hipMalloc((void**)&ctx->d_mmt_max, sizeof(double)));
    rc = eagle_dev_mmt_finish(ctx, c32.as<int32_t>(), n, np, ctx->d_mmt, n, ctx->d_mmt_max, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(MMt_out, ctx->d_mmt, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

extern "C" int eagle_last_mmt_normalised(eagle_ctx* ctx, double* MMt_norm_out, double* max_out) {
    if (!ctx || !ctx->d_mmt || ctx->mmt_n <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "no MMt result held");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long n = ctx->mmt_n;
    DevBuf tmp;
    HIPCHK(ctx, tmp.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(ctx, hipMemcpyAsync(tmp.p, ctx->d_mmt, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = eagle_dev_mmt_normalise(ctx, tmp.as<double>(), n, n, ctx->d_mmt_max, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(MMt_norm_out, tmp.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    if (max_out) HIPCHK(ctx, hipMemcpyAsync(max_out, ctx->d_mmt_max, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

static int ensure_scan_out(eagle_ctx* ctx, long L_pad) {
    if (ctx->scan_cap >= L_pad) return EAGLE_OK;
    if (ctx->d_a) { (void)hipFree(ctx->d_a); ctx->d_a = nullptr; }
    if (ctx->d_vara) { (void)hipFree(ctx->d_vara); ctx->d_vara = nullptr; }
    ctx->scan_cap = 0;
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_a, sizeof(double) * (size_t)L_pad));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_vara, sizeof(double) * (size_t)L_pad));
    ctx->scan_cap = L_pad;
    return EAGLE_OK;
}

extern "C" int eagle_calculate_a_and_vara(eagle_ctx* ctx, const char* f_name_ascii, const double* selected_loci,
                                          long n_selected, const double* inv_MMt_sqrt, const double* dim_reduced_vara,
                                          double max_memory_in_Gbytes, const long dims[2], const double* a, int quiet,
                                          double* a_out, double* vara_out) {
    if (!ctx) return EAGLE_ERR_ARG;
    const long L = dims[0], n = dims[1];
    if (n <= 0 || L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<long> sel;
    int rc = parse_selected(ctx, selected_loci, n_selected, L, sel);
    if (rc) return rc;
    // calculate_a_and_vara_rcpp.cpp:65 (integer division), :74, :129-144
    double mem_needed = (double)((4UL * (unsigned long)n * (unsigned long)L * sizeof(double)) / 1000000000UL);
    if (!quiet) say(ctx, "Inside internal function calculate_a_and_vara_rcpp: Need memory (gigabytes)  %g", mem_needed);
    if (!(mem_needed < max_memory_in_Gbytes)) {
        say(ctx, " Increasing maxmemGb would improve performance... \n");
        long rows_in_block = (long)(max_memory_in_Gbytes * 1e9 / (double)(4UL * (unsigned long)n * sizeof(double)));
        if (rows_in_block < 0) {
            say(ctx, "Error:  availmemGb is set to %g", max_memory_in_Gbytes);
            say(ctx, "        Cannot even read in a single row of data into memory.");
            a_out[0] = 0.0;
            vara_out[0] = 0.0;
            eagle_fail(ctx, EAGLE_SOFT_SENTINEL, "availmemGb: cannot even read in a single row of data into memory");
            return EAGLE_SOFT_SENTINEL;
        }
        if (rows_in_block == 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "availmemGb too small: zero rows per block");
    }
    const long np = eagle_pad(n), Lp = eagle_pad(L);
    const size_t sq = sizeof(double) * (size_t)np * np;
    const bool use_i8 = ctx->scan_mode == 1 && 64.0 * 512.0 * (double)np < 2147483648.0;
    const int nslices = ctx->scan_slices;
    GenoEntry* g = nullptr;
    rc = get_resident(ctx, f_name_ascii, L, n, max_memory_in_Gbytes, host_threads(), &g,
                      4 * sq + (use_i8 ? (size_t)eagle_vara_i8_workspace_bytes(np, Lp, nslices) + (size_t)Lp * np + 5 * (size_t)Lp +
                                             (size_t)eagle_scan_certify_workspace_bytes(np) : 0) +
                          ((size_t)1 << 30));  // operands, digit + certification workspaces, the re-centred image of the file
    if (rc < 0) return rc;
    const bool streamed = (rc == EAGLE_STREAM);
    const long Lc = streamed ? stream_chunk_rows(np, Lp) : Lp;  // marker rows per pass
    DevBuf dsel;
    const size_t wsb = use_i8 ? (size_t)eagle_vara_i8_workspace_bytes(np, Lc, nslices) : 0;
    const size_t certb = use_i8 ? (size_t)eagle_scan_certify_workspace_bytes(np) : 0;
    const double t0 = now_s();
    if ((rc = arena_reserve(ctx, 4 * arena_round(sq) + 2 * arena_round(sizeof(double) * np) + arena_round(wsb) + arena_round(certb) +
                                     (streamed ? (use_i8 ? 4 : 2) * arena_round((size_t)Lc * np) + 2 * arena_round((size_t)Lc) +
                                                     2 * arena_round(sizeof(int32_t) * (size_t)Lc) : 0))))
        return rc;
    double* Sa = arena_take<double>(ctx, sq);
    double* Va = arena_take<double>(ctx, sq);
    double* tmp = arena_take<double>(ctx, sq);
    double* Wu = arena_take<double>(ctx, sq);
    double* ah = arena_take<double>(ctx, sizeof(double) * np);
    double* v = arena_take<double>(ctx, sizeof(double) * np);
    void* ws = arena_take<char>(ctx, wsb);
    void* cert = arena_take<char>(ctx, certb);
    long* cert_totals = (long*)((char*)ctx->d_scratch + 256);  // {re-evaluated, flagged, fell back}, summed over marker blocks
    ChunkRing ring;
    int8_t* shifted[2] = {nullptr, nullptr};
    int8_t* cs[2] = {nullptr, nullptr};
    int32_t* l1s[2] = {nullptr, nullptr};
    if (streamed) {
        if ((rc = ring.init(ctx))) return rc;
        ring.buf[0] = arena_take<int8_t>(ctx, (size_t)Lc * np);
        ring.buf[1] = arena_take<int8_t>(ctx, (size_t)Lc * np);
        for (int b = 0; b < 2 && use_i8; b++) {
            shifted[b] = arena_take<int8_t>(ctx, (size_t)Lc * np);
            cs[b] = arena_take<int8_t>(ctx, (size_t)Lc);
            l1s[b] = arena_take<int32_t>(ctx, sizeof(int32_t) * (size_t)Lc);
        }
    }
    if ((rc = upload_square(ctx, inv_MMt_sqrt, n, np, Sa))) return rc;
    if ((rc = upload_square(ctx, dim_reduced_vara, n, np, Va))) return rc;
    if ((rc = upload_vec(ctx, a, n, np, ah))) return rc;
    if ((rc = ensure_scan_out(ctx, Lp))) return rc;
    HIPCHK(ctx, hipMemsetAsync(cert_totals, 0, 3 * sizeof(long), ctx->stream));
    double t1 = 0;
    if (timing_on()) { (void)hipStreamSynchronize(ctx->stream); t1 = now_s(); }
    rc = eagle_dev_scan_operands(ctx, Sa, Va, ah, n, np, v, Wu, tmp, ctx->stream);
    if (rc) return rc;
    if (streamed && !quiet) say(ctx, " Mt.ascii streamed through HBM in blocks of %ld markers", Lc);
    // one pass per marker block: the whole file when it is resident, else chunks read back from the file
    for (long r0 = 0; r0 < L; r0 += Lc) {
        const long nr = std::min(Lc, L - r0), nrp = eagle_pad(nr);
        const int8_t* Mt8 = streamed ? nullptr : g->dev;
        const long ldm = streamed ? np : g->ld;
        if (streamed) {
            int8_t* tile = nullptr;
            rc = ring.load(ctx, f_name_ascii, r0, nr, 0, n, np, (size_t)nrp * np, max_memory_in_Gbytes, host_threads(), &tile);
            if (rc) return rc;
            Mt8 = tile;
        }
        if (use_i8) {
            // re-centred image of the markers (kept with a resident file, rebuilt per chunk when streaming)
            const int8_t* Ms = nullptr;
            const int8_t* cv = nullptr;
            const int32_t* l1 = nullptr;
            if (streamed) {
                const int b = (int)(ring.k & 1);
                rc = eagle_dev_marker_shift(ctx, Mt8, nrp, n, np, ldm, shifted[b], cs[b], l1s[b], ctx->stream);
                if (rc) return rc;
                Ms = shifted[b]; cv = cs[b]; l1 = l1s[b];
            } else {
                if (!g->dev_s) {
                    hipError_t e = hipMalloc((void**)&g->dev_s, (size_t)g->rows_pad * g->ld);
                    if (e == hipSuccess) e = hipMalloc((void**)&g->cshift, (size_t)g->rows_pad);
                    if (e == hipSuccess) e = hipMalloc((void**)&g->l1, sizeof(int32_t) * (size_t)g->rows_pad);
                    rc = e == hipSuccess ? eagle_dev_marker_shift(ctx, g->dev, g->rows_pad, n, np, g->ld, g->dev_s, g->cshift, g->l1, ctx->stream)
                                         : eagle_fail_hip(ctx, e, "re-centred image hipMalloc");
                    if (rc) {  // never leave a half-made image behind: the next call would scan garbage
                        if (g->dev_s) (void)hipFree(g->dev_s);
                        if (g->cshift) (void)hipFree(g->cshift);
                        if (g->l1) (void)hipFree(g->l1);
                        g->dev_s = nullptr; g->cshift = nullptr; g->l1 = nullptr;
                        return rc;
                    }
                }
                Ms = g->dev_s; cv = g->cshift; l1 = g->l1;
            }
            // one pass over the genotypes gives a = Mt v and the diagonal term of vara; then the int8 MFMA kernel
            rc = eagle_dev_vara_i8_prepare(ctx, Mt8, nrp, np, ldm, Wu, nslices, ws, v, ctx->d_a + r0, ctx->stream);
            if (rc) return rc;
            rc = eagle_dev_vara_i8_mfma_shifted(ctx, Ms, cv, nrp, np, ldm, nslices, ws, ctx->d_vara + r0, nullptr, ctx->stream);
            if (rc) return rc;
            // a-posteriori certificate: markers the digit bounds cannot settle are re-evaluated by the fp64 kernel, so that
            // which(tsq == max(tsq))[1] on the returned arrays is the marker the fp64 scan selects (find_qtl.R:71-83).  A
            // streamed file is certified block by block against the block's own maximum (a superset of the global candidates).
            rc = eagle_dev_scan_certify(ctx, Mt8, nr, nrp, np, ldm, cv, l1, nslices, ws, Wu, ctx->d_a + r0, ctx->d_vara + r0, cert, ctx->stream);
            if (rc) return rc;
            rc = eagle_dev_cert_accumulate(ctx, cert, cert_totals, ctx->stream);
        } else {
            rc = eagle_dev_gemv_i8(ctx, Mt8, nrp, np, ldm, v, 1.0, ctx->d_a + r0, ctx->stream);
            if (rc) return rc;
            rc = eagle_dev_vara_f64(ctx, Mt8, nrp, np, ldm, Wu, ctx->d_vara + r0, ctx->stream);
        }
        if (rc) return rc;
        if (streamed && (rc = ring.computed(ctx))) return rc;
    }
    if (timing_on()) {
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[eaglehip] scan n=%ld L=%ld%s: alloc+upload %.1f ms, device compute %.1f ms\n", n, L,
                streamed ? " (streamed)" : "", (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    }
    if (rc) return rc;
    if (!sel.empty()) {  // :79-84: a zeroed marker row gives a = 0 and vara = 0 exactly
        HIPCHK(ctx, dsel.alloc(sizeof(long) * sel.size()));
        HIPCHK(ctx, hipMemcpyAsync(dsel.p, sel.data(), sizeof(long) * sel.size(), hipMemcpyHostToDevice, ctx->stream));
        rc = eagle_dev_zero_rows(ctx, ctx->d_a, ctx->d_vara, L, dsel.as<long>(), (long)sel.size(), 0, ctx->stream);
        if (rc) return rc;
    }
    ctx->scan_L = L;
    long totals[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(a_out, ctx->d_a, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(vara_out, ctx->d_vara, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(totals, cert_totals, sizeof totals, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cert_reevaluated = totals[0]; ctx->cert_flagged = totals[1]; ctx->cert_fell_back = totals[2] != 0;
    return EAGLE_OK;
}

extern "C" int eagle_last_scan_certificate(eagle_ctx* ctx, long* n_reevaluated, long* n_flagged, int* fell_back) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (n_reevaluated) *n_reevaluated = ctx->cert_reevaluated;
    if (n_flagged) *n_flagged = ctx->cert_flagged;
    if (fell_back) *fell_back = ctx->cert_fell_back;
    return EAGLE_OK;
}

extern "C" int eagle_extract_geno(eagle_ctx* ctx, const char* f_name_ascii, double max_memory_in_Gbytes, long selected_locus,
                                  const long dims[2], int* column_out) {
    if (!ctx) return EAGLE_ERR_ARG;
    const long n = dims[0], L = dims[1];
    if (n <= 0 || L <= 0 || selected_locus < 0 || selected_locus >= L) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims / locus");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    struct stat st;
    if (stat(f_name_ascii, &st) != 0) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", f_name_ascii);
    const long mt = (long)st.st_mtim.tv_sec * 1000000000L + st.st_mtim.tv_nsec;
    for (auto& g : ctx->cache)
        if (g.path == f_name_ascii && g.size == st.st_size && g.mtime_ns == mt && g.rows == n && g.cols == L) {
            DevBuf col;
            HIPCHK(ctx, col.alloc(sizeof(int) * (size_t)n));
            int rc = eagle_dev_extract_col(ctx, g.dev, n, g.ld, selected_locus, col.as<int>(), ctx->stream);
            if (rc) return rc;
            HIPCHK(ctx, hipMemcpyAsync(column_out, col.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            return EAGLE_OK;
        }
    // not resident: one character per line straight from the file (no n x L parse)
    FileInfo fi;
    int rc = open_file(ctx, f_name_ascii, fi);
    if (rc) return rc;
    if (fi.width >= 0) {
        if (n > fi.nlines) return eagle_fail(ctx, EAGLE_ERR_FORMAT, "file has fewer lines than requested");
        if (selected_locus >= fi.width) return eagle_fail(ctx, EAGLE_ERR_FORMAT, "line shorter than the requested columns");
        for (long r = 0; r < n; r++) {
            char c;
            if (pread(fi.fd, &c, 1, (off_t)r * (fi.width + 1) + selected_locus) != 1) return eagle_fail(ctx, EAGLE_ERR_FORMAT, "short read");
            if (c < '0' || c > '2') return eagle_fail(ctx, EAGLE_ERR_FORMAT, "character outside '0'..'2'");
            column_out[r] = (c - '0') - 1;
        }
        return EAGLE_OK;
    }
    FILE* f = fdopen(dup(fi.fd), "r");
    if (!f) return eagle_fail(ctx, EAGLE_ERR_OPEN, "fdopen failed");
    rewind(f);
    char* line = nullptr;
    size_t cap = 0;
    rc = EAGLE_OK;
    for (long r = 0; r < n; r++) {
        ssize_t len = getline(&line, &cap, f);
        if (len <= selected_locus) { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "file shorter than requested"); break; }
        char c = line[selected_locus];
        if (c < '0' || c > '2') { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "character outside '0'..'2'"); break; }
        column_out[r] = (c - '0') - 1;
    }
    free(line);
    fclose(f);
    return rc;
}

extern "C" int eagle_last_scan_argmax(eagle_ctx* ctx, long* index_out, double* tsqmax_out, long* n_near_ties) {
    if (!ctx || !ctx->d_a || ctx->scan_L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "no scan result held");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf scratch, best;
    HIPCHK(ctx, scratch.alloc(sizeof(double) * 3 * 1024));
    HIPCHK(ctx, best.alloc(sizeof(eagle_best)));
    int rc = eagle_dev_tsq_argmax(ctx, ctx->d_a, ctx->d_vara, ctx->scan_L, nullptr, best.as<eagle_best>(), scratch.as<double>(),
                                  ctx->stream);
    if (rc) return rc;
    eagle_best h;
    HIPCHK(ctx, hipMemcpyAsync(&h, best.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (index_out) *index_out = h.index0 + 1;  // R is 1-based; 0 = every tsq was NaN
    if (tsqmax_out) *tsqmax_out = h.tsqmax;
    if (n_near_ties) *n_near_ties = h.near_ties;
    return EAGLE_OK;
}

extern "C" int eagle_calculate_reduced_a(eagle_ctx* ctx, const char* f_name_ascii, double varG, const double* P,
                                         const double* y, double max_memory_in_Gbytes, const long dims[2],
                                         const double* selected_loci, long n_selected, int quiet, double* ar_out) {
    if (!ctx) return EAGLE_ERR_ARG;
    const long n = dims[0], L = dims[1];
    if (n <= 0 || L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<long> sel;
    int rc = parse_selected(ctx, selected_loci, n_selected, L, sel);
    if (rc) return rc;
    // calculate_reduced_a_rcpp.cpp:56: sizeof(double)/1000000000 is integer 0, so "memory needed" is 0 and the
    // in-memory branch runs iff 0 < availmemGb; otherwise the block size comes out negative (:92-103).
    if (!quiet) {
        say(ctx, "Inside internal function calculate_reduced_a_rcpp. Memory needed (gigabytes): 0");
        say(ctx, "Inside internal function calculate_reduced_a_rcpp. Memory available (gigabytes): %g", max_memory_in_Gbytes);
    }
    if (!(0.0 < max_memory_in_Gbytes)) {
        say(ctx, " Note:  Increasing availmemGb would improve performance... ");
        say(ctx, "Error:  availmemGb is set to %g", max_memory_in_Gbytes);
        ar_out[0] = 0.0;
        eagle_fail(ctx, EAGLE_SOFT_SENTINEL, "availmemGb: cannot even read in a single row of data into memory");
        return EAGLE_SOFT_SENTINEL;
    }
    const long np = eagle_pad(n), Lp = eagle_pad(L);
    GenoEntry* g = nullptr;
    rc = get_resident(ctx, f_name_ascii, L, n, max_memory_in_Gbytes, host_threads(), &g,
                      sizeof(double) * (size_t)np * np + ((size_t)1 << 30));
    if (rc < 0) return rc;
    const bool streamed = (rc == EAGLE_STREAM);
    const long Lc = streamed ? stream_chunk_rows(np, Lp) : Lp;
    DevBuf Pa, yv, py, out, dsel, chunk;
    HIPCHK(ctx, Pa.alloc(sizeof(double) * (size_t)np * np));
    HIPCHK(ctx, yv.alloc(sizeof(double) * np)); HIPCHK(ctx, py.alloc(sizeof(double) * np));
    HIPCHK(ctx, out.alloc(sizeof(double) * Lp));
    ChunkRing ring;
    if (streamed) {
        HIPCHK(ctx, chunk.alloc((size_t)2 * Lc * np));
        if ((rc = ring.init(ctx))) return rc;
        ring.buf[0] = chunk.as<int8_t>();
        ring.buf[1] = chunk.as<int8_t>() + (size_t)Lc * np;
    }
    if ((rc = upload_square(ctx, P, n, np, Pa.as<double>()))) return rc;
    if ((rc = upload_vec(ctx, y, n, np, yv.as<double>()))) return rc;
    rc = eagle_dev_colgemv(ctx, Pa.as<double>(), n, np, yv.as<double>(), py.as<double>(), ctx->stream);  // :82
    if (rc) return rc;
    for (long r0 = 0; r0 < L; r0 += Lc) {  // :83-84, one pass per marker block (the whole file when resident)
        const long nr = std::min(Lc, L - r0), nrp = eagle_pad(nr);
        int8_t* tile = nullptr;
        if (streamed) {
            rc = ring.load(ctx, f_name_ascii, r0, nr, 0, n, np, (size_t)nrp * np, max_memory_in_Gbytes, host_threads(), &tile);
            if (rc) return rc;
        }
        rc = eagle_dev_gemv_i8(ctx, streamed ? tile : g->dev, nrp, np, streamed ? np : g->ld, py.as<double>(), varG,
                               out.as<double>() + r0, ctx->stream);
        if (rc) return rc;
        if (streamed && (rc = ring.computed(ctx))) return rc;
    }
    if (!sel.empty()) {  // :74-78
        HIPCHK(ctx, dsel.alloc(sizeof(long) * sel.size()));
        HIPCHK(ctx, hipMemcpyAsync(dsel.p, sel.data(), sizeof(long) * sel.size(), hipMemcpyHostToDevice, ctx->stream));
        rc = eagle_dev_zero_rows(ctx, out.as<double>(), nullptr, L, dsel.as<long>(), (long)sel.size(), 0, ctx->stream);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(ar_out, out.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}
