// eagle_ctx.h -- the context object and small host helpers shared by eagle_api.cpp and eagle_ingest.cpp
// (private to libeaglehip.so; the public ABI only sees the opaque eagle_ctx*).
#ifndef EAGLE_CTX_H
#define EAGLE_CTX_H
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <sys/types.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/eagle_hip.h"
#include "eagle_internal.h"

struct GenoEntry {
    std::string path;
    off_t size = 0;
    long mtime_ns = 0;
    long rows = 0, cols = 0;          // logical tile held: all `rows` lines, first `cols` characters
    long rows_pad = 0, ld = 0;
    int8_t* dev = nullptr;
};

struct eagle_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    char err[1024] = {0};
    eagle_message_fn msg_fn = nullptr;
    void* msg_user = nullptr;
    int scan_mode = 1;   // 1 = int8 digit slices on the int8 MFMA (default), 0 = fp64 MFMA
    int scan_slices = 0; // 0 = chosen per call from the error bound (3..7), 1..8 = fixed
    std::vector<GenoEntry> cache;
    // results of the last calls, kept in HBM
    double* d_mmt = nullptr; long mmt_n = 0; double* d_mmt_max = nullptr;
    double* d_a = nullptr; double* d_vara = nullptr; long scan_L = 0; long scan_cap = 0;
    void* d_scratch = nullptr;
    void* arena = nullptr; size_t arena_cap = 0, arena_off = 0;  // grow-only device workspace reused across calls
    void* gemv_ws = nullptr;  // 16 digit-slice rows of the GEMV vectors + their exponents (k_gemv_mfma)
    void* stage_pin[2] = {nullptr, nullptr}; void* stage_raw[2] = {nullptr, nullptr}; size_t stage_cap = 0;  // tile streamer
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
// Whole-file resident copy (loads it if needed); EAGLE_OK, 2 (too large for HBM: stream it) or an error.
int eagle_get_resident(eagle_ctx* ctx, const char* path, long rows, long cols, double max_mem_gb, int threads, const GenoEntry** out);
size_t eagle_resident_budget();
#endif
