// eagle_ctx.h -- the context object and small host helpers shared by eagle_api.cpp and eagle_ingest.cpp
// (private to libeaglehip.so; the public ABI only sees the opaque eagle_ctx*).
#ifndef EAGLE_CTX_H
#define EAGLE_CTX_H
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/types.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eagle_hip.h"
#include "eagle_internal.h"
#include "eagle_host.h"
#include "eagle_w8.h"

struct GenoEntry {
    std::string path;
    off_t size = 0;
    long mtime_ns = 0;
    long rows = 0, cols = 0;          // logical tile held: `rows` lines from line row0, `cols` characters from character col0
    long row0 = 0, col0 = 0;          // (0, 0) and the whole file for a single-device ctx; a marker shard in a multi-device one
    long rows_pad = 0, ld = 0;
    int8_t* dev = nullptr;
    int8_t* dev_s = nullptr;    // re-centred image m - c_i (eagle_dev_marker_shift), made on the first digit-slice scan of the file
    int8_t* cshift = nullptr;   // c_i per row
    int32_t* l1 = nullptr;      // {sum_j |m_ij - c_i|, sum_j (m_ij - c_i)^2} per row (error bounds of the digit-slice scan)
    void* dev_f4 = nullptr;     // fp4 image of `dev` (two genotypes per byte): operand of the MM^T kernel, made by the first calculateMMt on the file
};

struct eagle_ctx {
    int device = -1;
    // multi-device: the ctx eagle_open_devices returns is the LEAD (first device); it owns one sub-context per further device.
    std::vector<eagle_ctx*> peers;
    eagle_ctx* lead = nullptr;           // set in sub-contexts
    double* d_Z = nullptr; long z_L = 0, z_n = 0, z_first = 0; long spectral_L = 0;  // Z = Mt U of the opt-in spectral scan (eagle_spectral.hip), L_pad x n_pad fp64
    void* blas_handle = nullptr;         // rocblas_handle of the opt-in device model algebra (eagle_linalg.cpp)
    void* rccl = nullptr;                // RcclState* of the lead (communicators, one per device), or NULL: host-staged sums
    long scan_first = 0;                 // global index of the first marker of this device's last scan
    int32_t* d_c32 = nullptr; size_t c32_cap = 0;      // partial MM^T accumulator (grow-only, kept between calls)
    int32_t* d_pack = nullptr; size_t pack_cap = 0;    // packed upper tiles of it (multi-device sum)
    int32_t* d_pack2 = nullptr; size_t pack2_cap = 0;  // lead: landing buffer of a peer's packed tiles (host-staged sum)
    hipStream_t stream = nullptr;
    hipStream_t load_stream = nullptr;  // tile loads of the streamed (out-of-core) paths run here, under the kernels of `stream`
    char err[1024] = {0};
    eagle_message_fn msg_fn = nullptr;
    void* msg_user = nullptr;
    int scan_mode = 1;   // 1 = int8 digit slices on the int8 MFMA (default), 0 = fp64 MFMA
    int scan_slices = 0; // 0 = chosen per call from the error bound (3..7), 1..8 = fixed
    int scan_stochastic = 0;  // 1 = digits of W rounded at random (unbiased): probabilistic certificate, one digit fewer (opt-in)
    double scan_budget = 5e-7;  // relative digit budget of the int8 scan (eagle_set_scan_budget): half of the path's 1e-6 tolerance
    double scan_budget_tight = 1e-7;  // tried first (round 4): the budget in force is this one whenever the digits that run certify it too; eagle_set_scan_budget sets both
    double scan_budget_enforced = 0.0; long cert_over_tight = 0;   // (eagle_last_scan_enforced)
    double scan_budget_used = 0.0; int scan_bound_level = 0; double scan_w_err = 0.0;   // of the last digit-slice scan (eagle_last_scan_budget)
    bool spectral_off = false;  // a scan that took a digit off under the spectral bound fell back to fp64: this context stops trying
    std::vector<GenoEntry> cache;
    // results of the last calls, kept in HBM
    double* d_mmt = nullptr; long mmt_n = 0; double* d_mmt_max = nullptr;
    double* d_a = nullptr; double* d_vara = nullptr; long scan_L = 0; long scan_cap = 0;
    double* d_bound = nullptr;  // a-posteriori error bound of every vara_i of the last digit-slice scan that ran in marker blocks / device shards
    long cert_reevaluated = 0, cert_flagged = 0; int cert_fell_back = 0;  // certification counters of the last digit-slice scan
    int scan_digits_used = 0, scan_digits_cut = 0; double scan_specH = 0.0;  // digit slices of the last digit-slice scan (eagle_last_scan_digits)
    double scan_phase_ms[8] = {0}; long scan_blocks = 0;                  // phase clock of the last scan on this device (eagle_last_scan_timing)
    double scan_host_setup_s = 0, scan_range_wall_s = 0, scan_call_wall_s = 0;
    // S = inv_MMt_sqrt of the last scan, kept on the device: MMt^-1/2 is the same matrix in every find_qtl call of an AM() run
    // (scan_range: the next call computes on this copy while the caller's matrix is uploaded and compared under the product)
    double* d_Scache = nullptr; double* d_Sscr = nullptr; long scache_n = 0, scache_np = 0; long scache_hits = 0, scache_misses = 0;
    // host copy of the cached S (n x n as the caller passed it): one resident block on one device verifies the caller's S against it with the
    // host's idle cores (memcmp) instead of sending 8 n^2 bytes over PCIe and through HBM under the vara kernel; h_Scache_n = 0: none
    double* h_Scache = nullptr; long h_Scache_n = 0; size_t h_Scache_cap = 0;
    // above 16,384 padded individuals no second device copy of S is kept (20 GB at n = 50,000): the last scan's S is still in its ARENA slot
    // when nothing has re-laid the arena since, and the next scan computes on it if the host comparison agrees (arena_S_ptr: that slot, else null)
    const void* arena_S_ptr = nullptr; const void* arena_S_base = nullptr; long arena_S_n = 0, arena_S_np = 0;
    // out-of-core bookkeeping of the last streamed call on this device (eagle_last_stream_stats)
    long st_chunks = 0, st_file_bytes = 0;
    double st_pread_s = 0, st_load_wall_s = 0, st_wait_s = 0, st_compute_s = 0, st_total_s = 0, st_starved_s = 0, st_load_first_s = 0;
    void* d_scratch = nullptr;
    void* argmax_ws = nullptr;   // block partials + result of eagle_last_scan_argmax (ctx-owned: no allocation per call)
    void* arena = nullptr; size_t arena_cap = 0, arena_off = 0;  // grow-only device workspace reused across calls
    void* arena_prefetch = nullptr;   // ArenaPrefetch* (eagle_api.cpp): a background hipMalloc of the arena in flight
    void* f4_buf = nullptr; size_t f4_cap = 0;  // fp4 image of the tile eagle_dev_mmt_accumulate is working on
    void* gemm_scratch = nullptr; size_t gemm_scratch_cap = 0;  // split-K partial tiles of the fp64 GEMM's last wave
    void* gemv_ws = nullptr;  // 16 digit-slice rows of the GEMV vectors + their exponents (k_gemv_mfma)
    int* h_flag = nullptr;   // 64 pinned bytes: where an asynchronous device-to-host copy of a flag lands (the deferred verification of the cached S)
    void* stage_pin[2] = {nullptr, nullptr}; void* stage_raw[2] = {nullptr, nullptr}; size_t stage_cap = 0;  // tile streamer
    // per-device launch state (a process may hold one ctx per GPU): dynamic-LDS attributes set on this device, schedule
    // experiment switch of tools/bench_i8_engine.py (0 = shipped)
    bool attr_vara_i8 = false, attr_vara_i8w = false, attr_vara_i8p = false, attr_vara_i8pp = false, attr_vara_i8px = false, attr_syrk_f4w = false, attr_zbuild_i8 = false, attr_vara_f6 = false, attr_gemv = false, attr_w8_gemm = false;
    int tune = 0;
    // W = S (V S) on the int8 engine (eagle_w8.hip): workspace, and what the last call left for the scan that follows it
    int w_mode = 1;            // 0 = always the fp64 GEMM, 1 = int8 digit slices from 4,096 padded individuals up, 2 = int8 at any size (tests)
    void* w8_ws = nullptr; size_t w8_ws_cap = 0; void* w8_host = nullptr;
    bool w8_active = false;    // Wu of the last scan_operands call came from the int8 engine:
    double w8_eta = 0.0;       //   || folded image - truth ||_F <= w8_eta (the per-marker certificate adds w8_eta sum_j m'_j^2)
    const double* w8_r = nullptr;   //   r = S (V (S 1)) in fp64 (the correction vector of the re-centred markers comes from it)
    const double* w8_Wu = nullptr; const double* w8_Sa = nullptr; const double* w8_Va = nullptr; long w8_n = 0;   //   the image it describes, and the operands
    double* w8_tmp = nullptr; void* w8_true_ws = nullptr; size_t w8_true_cap = 0;
    W8Info w8_info;
    void* w8_pipe = nullptr;            // W8Pipe* (eagle_w8.hip): the state between eagle_w8_begin / _vrows / _finish
    int w8_guess_c1 = -1; long w8_guess_np = 0;   // the first product's configuration of the last call: what a pipelined call starts on
    char arch[64] = {0};
    int cu_count = 0;
    int64_t hbm_bytes = 0;
};

extern thread_local char g_open_err[512];
static inline int host_threads() { unsigned h = std::thread::hardware_concurrency(); return h == 0 ? 1 : (h > 16 ? 16 : (int)h); }

static inline int failf(eagle_ctx* ctx, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (ctx) vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
    return code;
}
static inline void say(eagle_ctx* ctx, const char* fmt, ...) {
    if (!ctx->msg_fn) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->msg_fn(buf, ctx->msg_user);
}

#define HIPCHK(ctx, call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return eagle_fail_hip(ctx, e__, #call); \
    } while (0)

// RAII device / pinned buffers so every error path frees what it took
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() { return (T*)p; }
};
struct PinBuf {
    void* p = nullptr;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault); }
};


// Registers a device image (rows_pad x ld int8, zero padded) as the resident copy of the text file at `path`
// (rows lines x cols characters); takes ownership of `dev`.  Stale entries of the same path are dropped.
int eagle_cache_adopt(eagle_ctx* ctx, const char* path, long rows, long cols, long rows_pad, long ld, int8_t* dev);
// Resident copy of a file if the cache holds one that matches the file's current size and mtime, else nullptr.
const GenoEntry* eagle_cache_find(eagle_ctx* ctx, const char* path, long rows, long cols);
int eagle_stage_ensure(eagle_ctx* ctx, size_t need);

// 2-bit sidecar "<text file>.e2b": 64-byte header + rows x row_bytes packed genotype codes.  It is only trusted while
// the text file it was made from still has the recorded size and mtime.
struct E2bHeader {
    char magic[8];        // "EAGLE2B\0"
    uint32_t version;     // 1
    uint32_t reserved;
    uint64_t rows, cols, row_bytes;
    uint64_t src_size;
    int64_t src_mtime_ns;
    uint64_t pad;
};
static_assert(sizeof(E2bHeader) == 64, "E2bHeader is the 64-byte file header");
extern "C" int eagle_dev_pack2b(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, uint8_t* out, long row_bytes, void* stream);
extern "C" int eagle_dev_unpack2b(eagle_ctx* ctx, const uint8_t* raw, long rows, long cols, long stride, int shift, int8_t* out,
                                  long ld_out, int* bad_dev, void* stream);
inline bool eagle_sidecar_enabled() { const char* e = getenv("EAGLE_HIP_SIDECAR"); return !(e && e[0] == '0'); }
// Whole-file resident copy (loads it if needed); EAGLE_OK, 2 (too large for HBM: stream it) or an error.
int eagle_get_resident(eagle_ctx* ctx, const char* path, long rows, long cols, double max_mem_gb, int threads, const GenoEntry** out);
// The same for the window [row0, row0+rows) x [col0, col0+cols) of the file (a marker shard of a multi-device context).
int eagle_get_resident_window(eagle_ctx* ctx, const char* path, long row0, long rows, long col0, long cols, double max_mem_gb, int threads,
                              const GenoEntry** out);
size_t eagle_resident_budget();
size_t eagle_drop_f4_images(eagle_ctx* ctx);   // frees the fp4 MM^T operand images kept with resident files; bytes given back
#endif
