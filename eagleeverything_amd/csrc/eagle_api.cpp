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
#include <atomic>
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
// The first allocation of a large arena is the slowest thing a first find_qtl call does (3-6 s for 100 GB at 50,000 individuals:
// the driver maps the pages).  Whoever knows the problem size earlier -- eagle_calculateMMt (AM() calls it once, then spends seconds in
// the host's eigen-decomposition), eagle_prepare_scan, EAGLE_HIP_ARENA_GB at eagle_open -- starts it on a background thread;
// arena_reserve collects it.
struct ArenaPrefetch { std::thread th; void* p = nullptr; size_t bytes = 0; hipError_t e = hipSuccess; };
static void arena_collect(eagle_ctx* ctx) {
    ArenaPrefetch* pf = (ArenaPrefetch*)ctx->arena_prefetch;
    if (!pf) return;
    ctx->arena_prefetch = nullptr;
    if (pf->th.joinable()) pf->th.join();
    if (pf->e == hipSuccess && pf->p) {
        if (pf->bytes > ctx->arena_cap) {
            if (ctx->arena) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->arena); }
            ctx->arena = pf->p;
            ctx->arena_S_ptr = nullptr;
            ctx->arena_cap = pf->bytes;
        } else (void)hipFree(pf->p);
    } else (void)hipGetLastError();
    delete pf;
}
static void arena_prefetch(eagle_ctx* ctx, size_t total, bool only_if_roomy = false) {
    if (ctx->arena_prefetch || total <= ctx->arena_cap || total < ((size_t)4 << 30)) return;
    if (only_if_roomy) {   // a guess about what the caller does next must not crowd the card: at most half of what is free now
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) != hipSuccess || total > freeb / 2) { (void)hipGetLastError(); return; }
    }
    ArenaPrefetch* pf = new ArenaPrefetch;
    pf->bytes = total;
    const int device = ctx->device;
    pf->th = std::thread([pf, device] {
        pf->e = hipSetDevice(device);
        if (pf->e == hipSuccess) pf->e = hipMalloc(&pf->p, pf->bytes);
    });
    ctx->arena_prefetch = pf;
}
// Frees the fp4 operand images of MM^T kept with the resident files (GenoEntry.dev_f4: n L / 2 bytes each, 5 GB at 10,000 x 1,000,000;
// the next calculateMMt on the file re-makes its image in 3 ms).  Returns the bytes given back.
size_t eagle_drop_f4_images(eagle_ctx* ctx) {
    size_t freed = 0;
    for (GenoEntry& g : ctx->cache)
        if (g.dev_f4) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(g.dev_f4);
            g.dev_f4 = nullptr;
            freed += (size_t)g.rows_pad * (size_t)g.ld / 2;
        }
    return freed;
}
static int prepare_scan(eagle_ctx* ctx, long n, long L, bool only_if_roomy);
static int arena_reserve(eagle_ctx* ctx, size_t total) {
    ctx->arena_off = 0;
    arena_collect(ctx);
    if (total <= ctx->arena_cap) return EAGLE_OK;
    if (ctx->arena) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_cap = 0; }
    ctx->arena_S_ptr = nullptr;
    hipError_t e = hipMalloc(&ctx->arena, total);
    if (e != hipSuccess && eagle_drop_f4_images(ctx) > 0) {   // the fp4 MM^T operand images kept with resident files are not in any budget
        (void)hipGetLastError();
        e = hipMalloc(&ctx->arena, total);
    }
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
    if ((e = hipMalloc(&ctx->d_scratch, EAGLE_SCR_BYTES)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipMalloc: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    // EAGLE_HIP_TUNE=9: the compiler-scheduled forms of the two hand-scheduled kernels (k_vara_i8w, k_syrk_f4) for a whole session:
    // same results bit for bit, 3-8 % slower; a switch for ruling the inline-asm kernels out when chasing a problem
    if (const char* tv = getenv("EAGLE_HIP_TUNE")) ctx->tune = atoi(tv);
    if (const char* ag = getenv("EAGLE_HIP_ARENA_GB")) {   // reserve the scan arena at open (background thread), e.g. 100 for 50,000 individuals
        const double gb = atof(ag);
        if (gb > 0.0 && gb < 4096.0) arena_prefetch(ctx, (size_t)(gb * 1e9));
    }
    if (const char* wm = getenv("EAGLE_HIP_W_MODE")) {   // 0 / 1 / 2: eagle_set_w_mode for an R session that has no call for it
        const int m = atoi(wm);
        if (m >= 0 && m <= 2) ctx->w_mode = m;
    }
    // EAGLE_HIP_SCAN_BUDGET=1e-7: the digit budget of the int8 scan for an R session that has no call for it (eagle_set_scan_budget)
    if (const char* bv = getenv("EAGLE_HIP_SCAN_BUDGET")) {
        const double b = atof(bv);
        if (b >= 1e-12 && b <= 5e-7) ctx->scan_budget = ctx->scan_budget_tight = b;
    }
    // the loader stream outranks the compute stream: its decode / unpack / fill kernels are microseconds of work that must get
    // onto CUs the scan kernel's long-lived workgroups fill completely (2 waves x 256 VGPRs per SIMD), as soon as one retires
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->load_stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess) {
        snprintf(g_open_err, sizeof g_open_err, "hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    return ctx;
}

// ------------------------------------------------------------------------------------------------
// Several GPUs behind ONE context (the hook the reference left unused: AM(..., ngpu), E/R/AM.R:185-196, hard-wired to 0 at :214
// and handed to .find_qtl at :450-455).  The lead context owns one sub-context per further device; eagle_calculateMMt,
// eagle_calculate_a_and_vara and eagle_calculate_reduced_a split the file's markers into contiguous ranges (boundaries at
// multiples of 256), one per device, worked by one host thread + stream per device that are joined before the call returns.
// Exchange steps: ONE sum of the partial int32 MM^T (upper tiles) to the lead, the rows of W = S V S shared and all-gathered,
// the lower bounds of the shards' tsq maxima (8 bytes per device, on the host).  Between distinct devices the two device
// collectives are RCCL (ncclReduce / ncclAllGather over xGMI, communicators from ncclCommInitAll, the library dlopen()ed on
// first use); when the list names one card twice (how the path is tested on a one-GPU box) or EAGLE_HIP_COLLECTIVES=host,
// the sum is staged by device-to-device copies and W is computed on every device.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>
struct RcclState {
    void* lib = nullptr;
    std::vector<ncclComm_t> comms;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: a library without it keeps the old behaviour
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::mutex abort_mutex;
    std::atomic<bool> dead{false};   // the communicators were aborted: every later call takes the host-staged exchange
};
// A device whose collective could not be enqueued AFTER the rendezvous that precedes it would leave its peers blocked inside
// theirs: abort every communicator (the peers' collectives end with an error, their calls return it) and retire the RCCL leg of
// this context.  Once, whoever comes first.
static void rccl_abort_all(RcclState* r) {
    std::lock_guard<std::mutex> lock(r->abort_mutex);
    if (r->dead.exchange(true)) return;
    for (ncclComm_t& c : r->comms)
        if (c) { if (r->CommAbort) (void)r->CommAbort(c); c = nullptr; }
}
// The communicator of device k for the collective a worker is about to enqueue, or NULL when the leg was aborted meanwhile (ADVICE r3:
// a peer used to read comms[k] unlocked while rccl_abort_all was writing it).  A handle taken here can still be aborted before the
// enqueue -- RCCL then returns an error from the call, which is the outcome the abort exists for.
static ncclComm_t rccl_comm(RcclState* r, int k) {
    std::lock_guard<std::mutex> lock(r->abort_mutex);
    return r->dead.load() ? nullptr : r->comms[(size_t)k];
}
// EAGLE_HIP_FAULT=reduce | allgather [:<device index>]: the named collective reports a failure instead of running -- on every device,
// or on the one named, while its peers go on into theirs (tests of the abort path; never set in production).
static bool rccl_fault(const char* which, int k) {
    const char* e = getenv("EAGLE_HIP_FAULT");   // (tests set and unset it between calls of one process: a cached copy would go stale)
    if (!e) return false;
    const size_t n = strlen(which);
    if (strncmp(e, which, n) != 0) return false;
    if (e[n] == '\0') return true;
    return e[n] == ':' && atoi(e + n + 1) == k;
}
static RcclState* rccl_open(const int* devices, int ndev, char* err, size_t errlen) {
    RcclState* r = new RcclState();
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((r->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r->lib) { snprintf(err, errlen, "cannot load librccl: %s", dlerror()); delete r; return nullptr; }
    r->CommInitAll = (decltype(r->CommInitAll))dlsym(r->lib, "ncclCommInitAll");
    r->CommDestroy = (decltype(r->CommDestroy))dlsym(r->lib, "ncclCommDestroy");
    r->Reduce = (decltype(r->Reduce))dlsym(r->lib, "ncclReduce");
    r->AllGather = (decltype(r->AllGather))dlsym(r->lib, "ncclAllGather");
    r->GetErrorString = (decltype(r->GetErrorString))dlsym(r->lib, "ncclGetErrorString");
    r->CommAbort = (decltype(r->CommAbort))dlsym(r->lib, "ncclCommAbort");
    if (!r->CommInitAll || !r->CommDestroy || !r->Reduce || !r->AllGather || !r->GetErrorString) {
        snprintf(err, errlen, "librccl lacks an entry point");
        dlclose(r->lib); delete r; return nullptr;
    }
    r->comms.resize(ndev);
    ncclResult_t rc = r->CommInitAll(r->comms.data(), ndev, devices);
    if (rc != ncclSuccess) {
        snprintf(err, errlen, "ncclCommInitAll: %s", r->GetErrorString(rc));
        dlclose(r->lib); delete r; return nullptr;
    }
    return r;
}

extern "C" eagle_ctx* eagle_open_devices(const int* devices, int ndev) {
    if (!devices || ndev <= 0 || ndev > 64) { snprintf(g_open_err, sizeof g_open_err, "eagle_open_devices: bad device list"); return nullptr; }
    eagle_ctx* lead = eagle_open(devices[0]);
    if (!lead) return nullptr;
    for (int k = 1; k < ndev; k++) {
        eagle_ctx* sub = eagle_open(devices[k]);
        if (!sub) { eagle_close(lead); return nullptr; }
        sub->lead = lead;
        lead->peers.push_back(sub);
    }
    bool distinct = true;
    for (int a = 0; a < ndev; a++) for (int b = a + 1; b < ndev; b++) if (devices[a] == devices[b]) distinct = false;
    const char* env = getenv("EAGLE_HIP_COLLECTIVES");
    // EAGLE_HIP_COLLECTIVES=host: host-staged stand-in instead of RCCL; =rccl with ONE device: the exchange steps run anyway, on
    // a communicator of one rank (what a one-GPU box can exercise of the RCCL leg: tests/test_multi_device.py)
    const bool force1 = ndev == 1 && env && strcmp(env, "rccl") == 0;
    if ((ndev > 1 || force1) && distinct && !(env && strcmp(env, "host") == 0)) {
        lead->rccl = rccl_open(devices, ndev, g_open_err, sizeof g_open_err);
        if (!lead->rccl) { eagle_close(lead); return nullptr; }
    }
    (void)hipSetDevice(devices[0]);
    return lead;
}
// EAGLE_HIP_DEVICES="0,1,2,3" (several GPUs), else EAGLE_HIP_DEVICE=<d>, else device 0: what the R-side shim opens.
extern "C" eagle_ctx* eagle_open_env(void) {
    std::vector<int> devs;
    if (const char* e = getenv("EAGLE_HIP_DEVICES")) {
        for (const char* p = e; *p;) {
            char* end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p) { snprintf(g_open_err, sizeof g_open_err, "EAGLE_HIP_DEVICES: cannot parse '%s'", e); return nullptr; }
            devs.push_back((int)v);
            p = end;
            while (*p == ',' || *p == ' ') p++;
        }
    }
    if (devs.empty()) { const char* d = getenv("EAGLE_HIP_DEVICE"); devs.push_back(d ? atoi(d) : 0); }
    return devs.size() == 1 ? eagle_open(devs[0]) : eagle_open_devices(devs.data(), (int)devs.size());
}
extern "C" int eagle_device_count(eagle_ctx* ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

static inline int ndev_of(eagle_ctx* ctx) { return 1 + (int)ctx->peers.size(); }
static inline eagle_ctx* dev_ctx(eagle_ctx* ctx, int k) { return k == 0 ? ctx : ctx->peers[k - 1]; }
// fn(k, ctx_k) on every device: the lead's share on the calling thread (the only one that may send messages to R), one
// worker thread per further device; all joined before return.  First hard error wins, then the soft sentinel.
template <class F> static int run_on_devices(eagle_ctx* ctx, F fn) {
    const int nd = ndev_of(ctx);
    std::vector<int> rc(nd, 0);
    std::vector<std::thread> th;
    for (int k = 1; k < nd; k++) th.emplace_back([&rc, &fn, ctx, k] { rc[k] = fn(k, dev_ctx(ctx, k)); });
    rc[0] = fn(0, ctx);
    for (auto& t : th) t.join();
    (void)hipSetDevice(ctx->device);
    for (int k = 0; k < nd; k++)
        if (rc[k] < 0) {
            if (k) { char buf[1024]; snprintf(buf, sizeof buf, "device %d: %s", dev_ctx(ctx, k)->device, dev_ctx(ctx, k)->err); snprintf(ctx->err, sizeof ctx->err, "%s", buf); }
            return rc[k];
        }
    for (int k = 0; k < nd; k++) if (rc[k] > 0) return rc[k];
    return EAGLE_OK;
}

// this device's resident genotype copies only (safe from a per-device worker thread)
static void drop_cache_local(eagle_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    for (auto& g : ctx->cache) { if (g.dev) (void)hipFree(g.dev); if (g.dev_s) (void)hipFree(g.dev_s); if (g.cshift) (void)hipFree(g.cshift); if (g.l1) (void)hipFree(g.l1); if (g.dev_f4) (void)hipFree(g.dev_f4); }
    ctx->cache.clear();
    if (ctx->f4_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->f4_buf); ctx->f4_buf = nullptr; ctx->f4_cap = 0; }
}
// The resident genotype files AND the grow-only workspaces this context holds between calls (the scan arena -- a background reservation
// in flight included --, the workspace of the int8 W products): everything a caller can get back without closing the context.
static void drop_workspaces(eagle_ctx* c) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    arena_collect(c);
    if (c->arena) { (void)hipFree(c->arena); c->arena = nullptr; c->arena_cap = 0; c->arena_off = 0; }
    c->arena_S_ptr = nullptr;
    if (c->w8_ws) { (void)hipFree(c->w8_ws); c->w8_ws = nullptr; c->w8_ws_cap = 0; }
    if (c->w8_true_ws) { (void)hipFree(c->w8_true_ws); c->w8_true_ws = nullptr; c->w8_true_cap = 0; }
    c->w8_active = false;
}
extern "C" void eagle_drop_cache(eagle_ctx* ctx) {
    if (!ctx) return;
    for (eagle_ctx* p : ctx->peers) { drop_cache_local(p); drop_workspaces(p); }
    drop_cache_local(ctx);
    drop_workspaces(ctx);
    (void)hipSetDevice(ctx->device);
}

extern "C" void eagle_close(eagle_ctx* ctx) {
    if (!ctx) return;
    if (ctx->rccl) {
        RcclState* r = (RcclState*)ctx->rccl;
        for (ncclComm_t c : r->comms) if (c) (void)r->CommDestroy(c);
        delete r;  // the library stays loaded
        ctx->rccl = nullptr;
    }
    for (eagle_ctx* p : ctx->peers) eagle_close(p);
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    eagle_linalg_release(ctx);
    eagle_spectral_release(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    eagle_drop_cache(ctx);
    if (ctx->d_mmt) (void)hipFree(ctx->d_mmt);
    if (ctx->d_mmt_max) (void)hipFree(ctx->d_mmt_max);
    if (ctx->d_a) (void)hipFree(ctx->d_a);
    if (ctx->d_vara) (void)hipFree(ctx->d_vara);
    if (ctx->d_bound) (void)hipFree(ctx->d_bound);
    if (ctx->argmax_ws) (void)hipFree(ctx->argmax_ws);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_Scache) (void)hipFree(ctx->d_Scache);
    if (ctx->d_Sscr) (void)hipFree(ctx->d_Sscr);
    arena_collect(ctx);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->gemv_ws) (void)hipFree(ctx->gemv_ws);
    if (ctx->f4_buf) (void)hipFree(ctx->f4_buf);
    if (ctx->gemm_scratch) (void)hipFree(ctx->gemm_scratch);
    eagle_w8_release(ctx);
    if (ctx->w8_ws) (void)hipFree(ctx->w8_ws);
    if (ctx->w8_true_ws) (void)hipFree(ctx->w8_true_ws);
    if (ctx->w8_host) (void)hipHostFree(ctx->w8_host);
    if (ctx->d_c32) (void)hipFree(ctx->d_c32);
    if (ctx->d_pack) (void)hipFree(ctx->d_pack);
    if (ctx->d_pack2) (void)hipFree(ctx->d_pack2);
    for (int b = 0; b < 2; b++) { if (ctx->stage_pin[b]) (void)hipHostFree(ctx->stage_pin[b]); if (ctx->stage_raw[b]) (void)hipFree(ctx->stage_raw[b]); }
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    free(ctx->h_Scache);
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
    for (eagle_ctx* p : ctx->peers) p->scan_mode = mode;
    return EAGLE_OK;
}
extern "C" int eagle_set_scan_rounding(eagle_ctx* ctx, int stochastic) {
    if (!ctx || stochastic < 0 || stochastic > 1) return EAGLE_ERR_ARG;
    ctx->scan_stochastic = stochastic;
    for (eagle_ctx* p : ctx->peers) p->scan_stochastic = stochastic;
    return EAGLE_OK;
}
extern "C" int eagle_set_scan_budget(eagle_ctx* ctx, double relative_budget) {
    if (ctx && relative_budget == 0.0) {   // the default policy: 1e-7 where the digits that run certify it, else 5e-7
        ctx->scan_budget = 5e-7; ctx->scan_budget_tight = 1e-7; ctx->spectral_off = false;
        for (eagle_ctx* p : ctx->peers) { p->scan_budget = 5e-7; p->scan_budget_tight = 1e-7; p->spectral_off = false; }
        return EAGLE_OK;
    }
    if (!ctx || !(relative_budget >= 1e-12 && relative_budget <= 5e-7)) return EAGLE_ERR_ARG;   // 1.8 x budget is enforced per marker: never above 0.9e-6
    // a budget asked for is THE budget: nothing tighter is tried first (the default context tries 1e-7, then 5e-7; relative_budget = 0 restores that)
    ctx->scan_budget = ctx->scan_budget_tight = relative_budget;
    ctx->spectral_off = false;
    for (eagle_ctx* p : ctx->peers) { p->scan_budget = p->scan_budget_tight = relative_budget; p->spectral_off = false; }
    return EAGLE_OK;
}
// on = 0: this context stops taking a digit off under the spectral bound (what eagle_calculate_a_and_vara does by itself after a
// certificate that overflowed under it); on = 1 re-arms.  For callers of the device-resident entry points, which see the certificate
// themselves (eagleeverything_amd/sharded.py).  Not part of the public ABI.
extern "C" void eagle_dev_set_spectral(eagle_ctx* ctx, int on) { if (ctx) ctx->spectral_off = !on; }
extern "C" int eagle_set_scan_slices(eagle_ctx* ctx, int nslices) {
    if (!ctx || nslices < 0 || nslices > 8) return EAGLE_ERR_ARG;
    ctx->scan_slices = nslices;
    for (eagle_ctx* p : ctx->peers) p->scan_slices = nslices;
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// selected_loci rule (calculateMMt_rcpp.cpp:88; calculate_a_and_vara_rcpp.cpp:79; calculate_reduced_a_rcpp.cpp:74):
// masking fires iff element 0 is not NA.  NA arrives as NaN.
// ------------------------------------------------------------------------------------------------
static int parse_selected(eagle_ctx* ctx, const double* sel, long nsel, long bound, std::vector<long>& out) {
    const char* msg = parse_selected_core(sel, nsel, bound, out);  // eagle_host.h
    return msg ? eagle_fail(ctx, EAGLE_ERR_ARG, msg) : EAGLE_OK;
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
    // a thread per >= 2 MiB of the read, 32 at most: num_cores comes from the caller (R hands detectCores()), and spawning
    // hundreds of threads per 64 MiB staging buffer costs more than the reads
    const long by_size = (nrows * nbytes) >> 21;
    if (threads > 32) threads = 32;
    if (threads > by_size) threads = (int)by_size;
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
    // the bad-character counter lives in the ctx scratch page: a hipMalloc / hipFree per tile would synchronise the device,
    // i.e. wait for the kernels of the previous chunk when the tile is a chunk of a streamed file
    int* const bad = (int*)((char*)ctx->d_scratch + EAGLE_SCR_LOADER_BAD);
    hipEvent_t done[2] = {nullptr, nullptr};
    HIPCHK(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
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
        const double tp = now_s();
        parallel_pread(fi.fd, (uint8_t*)pin[b], stride, nr, src_bytes, (off_t)(row0 + r) * line + col0, line, threads,
                       &io_err);
        ctx->st_pread_s += now_s() - tp;
        ctx->st_file_bytes += nr * src_bytes;
        if (io_err) { rc = eagle_fail(ctx, EAGLE_ERR_FORMAT, "short read: file has fewer lines than requested"); break; }
        hipError_t e = hipMemcpyAsync(raw[b], pin[b], (size_t)nr * stride, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipMemcpyAsync H2D"); break; }
        e = hipEventRecord(done[b], ctx->stream);
        if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipEventRecord"); break; }
        rc = eagle_dev_decode_ascii(ctx, raw[b], nr, ncols, stride, dst + r * ld, ld, bad, ctx->stream);
    }
    int nbad = 0;
    if (rc == EAGLE_OK) {
        hipError_t e = hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
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
    int* const bad = (int*)((char*)ctx->d_scratch + EAGLE_SCR_LOADER_BAD);  // (see load_tile_fixed)
    HIPCHK(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; b++) if (e[b]) (void)hipEventDestroy(e[b]); } } evg{done};
    volatile int io_err = 0;
    long k = 0;
    for (long r = 0; r < nrows; r += chunk_rows, k++) {
        const int b = (int)(k & 1);
        const long nr = std::min(chunk_rows, nrows - r);
        if (k >= 2) HIPCHK(ctx, hipEventSynchronize(done[b]));
        const double tp = now_s();
        parallel_pread(fd, (uint8_t*)ctx->stage_pin[b], stride, nr, nb, (off_t)sizeof h + (off_t)(row0 + r) * (off_t)h.row_bytes + b0,
                       (long)h.row_bytes, threads, &io_err);
        ctx->st_pread_s += now_s() - tp;
        ctx->st_file_bytes += nr * nb;
        if (io_err) { (void)hipStreamSynchronize(ctx->stream); return eagle_fail(ctx, EAGLE_ERR_FORMAT, "short read from the 2-bit sidecar"); }
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage_raw[b], ctx->stage_pin[b], (size_t)nr * stride, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(done[b], ctx->stream));
        rc = eagle_dev_unpack2b(ctx, (const uint8_t*)ctx->stage_raw[b], nr, ncols, stride, (int)(col0 % 4), dst + r * ld, ld, bad, ctx->stream);
        if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    }
    int nbad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
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
static void free_entry(GenoEntry& g) {
    if (g.dev) (void)hipFree(g.dev);
    if (g.dev_s) (void)hipFree(g.dev_s);
    if (g.cshift) (void)hipFree(g.cshift);
    if (g.l1) (void)hipFree(g.l1);
    if (g.dev_f4) (void)hipFree(g.dev_f4);
    g.dev = g.dev_s = g.cshift = nullptr; g.l1 = nullptr; g.dev_f4 = nullptr;
}
// Whole-file resident copy (rows lines x cols characters from the origin), current size and mtime.
const GenoEntry* eagle_cache_find(eagle_ctx* ctx, const char* path, long rows, long cols) {
    off_t size; long mt;
    if (!file_key(path, &size, &mt)) return nullptr;
    for (auto& g : ctx->cache)
        if (g.path == path && g.size == size && g.mtime_ns == mt && g.row0 == 0 && g.col0 == 0 && g.rows == rows && g.cols == cols) return &g;
    return nullptr;
}
// An entry that can serve the window [row0, row0+rows) x [col0, col0+cols): the same window, or an image from the origin that
// holds it as a prefix (all columns and the first `rows` lines, or all lines and the first `cols` characters): what the lead of
// a multi-device context finds when the converters left the whole file resident.
static GenoEntry* cache_find_window(eagle_ctx* ctx, const char* path, off_t size, long mt, long row0, long rows, long col0, long cols) {
    for (auto& g : ctx->cache) {
        if (!(g.path == path && g.size == size && g.mtime_ns == mt)) continue;
        if (g.row0 == row0 && g.col0 == col0 && g.rows == rows && g.cols == cols) return &g;
        if (row0 == 0 && col0 == 0 && g.row0 == 0 && g.col0 == 0 &&
            ((g.cols == cols && g.rows >= rows && rows % 256 == 0) || (g.rows == rows && g.cols >= cols && cols % 256 == 0)))
            return &g;
    }
    return nullptr;
}
static void cache_drop_path(eagle_ctx* ctx, const char* path) {
    for (size_t i = 0; i < ctx->cache.size();)
        if (ctx->cache[i].path == path) {
            free_entry(ctx->cache[i]);
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

// Resident int8 image of the window [row0, row0+rows) x [col0, col0+cols) of a genotype text file, zero padded to
// [pad256(rows)][pad256(cols)] (the whole file: row0 = col0 = 0).
// Returns EAGLE_OK (*out set), EAGLE_STREAM (too large: the caller streams marker chunks) or an error.
// reserve_bytes: HBM the caller still needs for operands and workspaces.
static int get_resident(eagle_ctx* ctx, const char* path, long row0, long rows, long col0, long cols, double max_mem_gb, int threads,
                        GenoEntry** out, size_t reserve_bytes = (size_t)1 << 30) {
    off_t fsize; long mt;
    if (!file_key(path, &fsize, &mt)) return failf(ctx, EAGLE_ERR_OPEN, "ERROR: Could not open  %s", path);
    if (GenoEntry* hit = cache_find_window(ctx, path, fsize, mt, row0, rows, col0, cols)) { *out = hit; return EAGLE_OK; }
    for (size_t i = 0; i < ctx->cache.size();)  // stale entries of the same path (the file changed) and other windows of it
        if (ctx->cache[i].path == path) { free_entry(ctx->cache[i]); ctx->cache.erase(ctx->cache.begin() + i); } else i++;
    GenoEntry g;
    g.path = path; g.size = fsize; g.mtime_ns = mt; g.rows = rows; g.cols = cols; g.row0 = row0; g.col0 = col0;
    g.rows_pad = eagle_pad(rows); g.ld = eagle_pad(cols);
    size_t bytes = (size_t)g.rows_pad * (size_t)g.ld;
    if (bytes > resident_budget()) return EAGLE_STREAM;
    size_t freeb = 0, totalb = 0;
    HIPCHK(ctx, hipMemGetInfo(&freeb, &totalb));
    if (bytes + reserve_bytes > freeb) {
        drop_cache_local(ctx);
        HIPCHK(ctx, hipMemGetInfo(&freeb, &totalb));
        if (bytes + reserve_bytes > freeb) return EAGLE_STREAM;  // does not fit beside the operands: stream it
    }
    HIPCHK(ctx, hipMalloc((void**)&g.dev, bytes));
    hipError_t e = hipMemsetAsync(g.dev, 0, bytes, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(g.dev); return eagle_fail_hip(ctx, e, "memset"); }
    int rc = eagle_dev_load_ascii(ctx, path, row0, rows, col0, cols, g.dev, g.ld, max_mem_gb, threads);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(g.dev); return rc; }
    ctx->cache.push_back(g);
    *out = &ctx->cache.back();
    return EAGLE_OK;
}

int eagle_get_resident(eagle_ctx* ctx, const char* path, long rows, long cols, double max_mem_gb, int threads, const GenoEntry** out) {
    GenoEntry* g = nullptr;
    int rc = get_resident(ctx, path, 0, rows, 0, cols, max_mem_gb, threads, &g);
    *out = g;
    return rc;
}
int eagle_get_resident_window(eagle_ctx* ctx, const char* path, long row0, long rows, long col0, long cols, double max_mem_gb, int threads,
                              const GenoEntry** out) {
    GenoEntry* g = nullptr;
    int rc = get_resident(ctx, path, row0, rows, col0, cols, max_mem_gb, threads, &g);
    *out = g;
    return rc;
}
size_t eagle_resident_budget() { return resident_budget(); }

// Rows (multiple of 256) of a streamed chunk whose padded row length is `row_bytes`.
static long stream_chunk_rows(long row_bytes, long total_rows_pad) {
    return stream_chunk_rows_core(resident_budget(), row_bytes, total_rows_pad);  // eagle_host.h
}

// Out-of-core streaming: chunk k+1 is read (pread -> pinned -> H2D -> decode, all on ctx->load_stream) while the kernels
// of chunk k run on ctx->stream.  Two chunk buffers; an event per buffer says when its kernels are done.
struct ChunkRing {
    int8_t* buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;  // start / end of every chunk's kernels (timed: eagle_last_stream_stats)
    long k = 0;
    double t_begin = 0;
    ~ChunkRing() {
        for (int b = 0; b < 2; b++) if (done[b]) (void)hipEventDestroy(done[b]);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    int init(eagle_ctx* ctx) {
        for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        ctx->st_chunks = ctx->st_file_bytes = 0;
        ctx->st_pread_s = ctx->st_load_wall_s = ctx->st_wait_s = ctx->st_compute_s = ctx->st_total_s = ctx->st_starved_s = ctx->st_load_first_s = 0;
        return EAGLE_OK;
    }
    // returns the buffer holding the freshly loaded tile (zero padded to clear_bytes); the caller launches its kernels on
    // ctx->stream and then calls computed()
    int load(eagle_ctx* ctx, const char* path, long row0, long nrows, long col0, long ncols, long ld, size_t clear_bytes, double mem_gb,
             int threads, int8_t** out) {
        const int b = (int)(k & 1);
        const double t0 = now_s();
        if (k == 0) t_begin = t0;
        if (k >= 2) HIPCHK(ctx, hipEventSynchronize(done[b]));  // the kernels that read this buffer two chunks ago
        const double t1 = now_s();
        hipStream_t main = ctx->stream;
        ctx->stream = ctx->load_stream;  // every loader below works on ctx->stream
        int rc = EAGLE_OK;
        hipError_t e = hipMemsetAsync(buf[b], 0, clear_bytes, ctx->stream);
        if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "chunk memset");
        if (!rc) rc = eagle_dev_load_ascii(ctx, path, row0, nrows, col0, ncols, buf[b], ld, mem_gb, threads);
        if (!rc && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "chunk load sync");
        ctx->stream = main;
        *out = buf[b];
        ctx->st_wait_s += t1 - t0;
        ctx->st_load_wall_s += now_s() - t1;
        if (k == 0) ctx->st_load_first_s = now_s() - t1;
        if (timing_on() && getenv("EAGLE_HIP_TIMING")[0] == '2')
            fprintf(stderr, "[eaglehip]   chunk %ld: host waits %.1f..%.1f ms, loads ..%.1f ms\n", k, (t0 - t_begin) * 1e3, (t1 - t_begin) * 1e3, (now_s() - t_begin) * 1e3);
        hipEvent_t e0 = nullptr;  // the chunk's kernels start here on ctx->stream
        if (!rc && hipEventCreate(&e0) == hipSuccess) { ev.push_back(e0); (void)hipEventRecord(e0, ctx->stream); }
        return rc;
    }
    int computed(eagle_ctx* ctx) {
        hipEvent_t e1 = nullptr;
        if (ev.size() == (size_t)(2 * k + 1) && hipEventCreate(&e1) == hipSuccess) { ev.push_back(e1); (void)hipEventRecord(e1, ctx->stream); }
        HIPCHK(ctx, hipEventRecord(done[(int)(k & 1)], ctx->stream));
        if (timing_on() && getenv("EAGLE_HIP_TIMING")[0] == '2') fprintf(stderr, "[eaglehip]   chunk %ld: kernels launched at %.1f ms\n", k, (now_s() - t_begin) * 1e3);
        k++;
        return EAGLE_OK;
    }
    // after the last chunk: waits for its kernels and closes the books
    void finish(eagle_ctx* ctx) {
        (void)hipStreamSynchronize(ctx->stream);
        double ms = 0, starved = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) ms += t;
            // the compute stream sat idle between the end of chunk k's kernels and the moment chunk k+1 was loaded
            if (i + 2 < ev.size() && hipEventElapsedTime(&t, ev[i + 1], ev[i + 2]) == hipSuccess) starved += t;
        }
        ctx->st_chunks = k;
        ctx->st_compute_s = ms / 1e3;
        ctx->st_starved_s = starved / 1e3;
        ctx->st_total_s = now_s() - t_begin;
        if (timing_on()) {
            const double later = ctx->st_load_wall_s - ctx->st_load_first_s;  // loads that had kernels to hide under
            fprintf(stderr, "[eaglehip] streamed: %ld chunks, %.3f GB read in %.3f s (%.2f GB/s while reading), loads %.3f s (first %.3f s), "
                            "kernels %.3f s, compute stream starved for %.3f s (%.0f %% of the later loads hidden), wall %.3f s\n",
                    k, ctx->st_file_bytes / 1e9, ctx->st_pread_s, ctx->st_pread_s > 0 ? ctx->st_file_bytes / 1e9 / ctx->st_pread_s : 0.0,
                    ctx->st_load_wall_s, ctx->st_load_first_s, ctx->st_compute_s, ctx->st_starved_s,
                    later > 0 ? 100.0 * std::max(0.0, 1.0 - ctx->st_starved_s / later) : 100.0, ctx->st_total_s);
        }
    }
};

// Phase clock of one scan call: HIP events on the compute stream, read after the call's final synchronisation
// (eagle_last_scan_timing).  The interval that ENDS at an event is booked under the event's tag.
enum { PH_START = 0, PH_UPLOAD, PH_W, PH_LOADWAIT, PH_PREPARE, PH_VARA, PH_CERT, PH_D2H, PH_COUNT };
struct PhaseEvents {
    std::vector<hipEvent_t> ev;
    std::vector<int> tag;
    ~PhaseEvents() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    void mark(hipStream_t st, int t) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        if (hipEventRecord(e, st) != hipSuccess) { (void)hipEventDestroy(e); return; }
        ev.push_back(e); tag.push_back(t);
    }
    void sum(double* ms) const {  // ms[PH_COUNT]; the stream has been synchronised
        for (int i = 0; i < PH_COUNT; i++) ms[i] = 0.0;
        for (size_t i = 1; i < ev.size(); i++) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i - 1], ev[i]) == hipSuccess) ms[tag[i]] += t;
        }
    }
};

// any bit of two device arrays different -> *flag != 0 (the cached S against the caller's)
__global__ __launch_bounds__(256) void k_bits_differ(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, long count,
                                                     int* __restrict__ flag) {
    int d = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) d |= a[i] != b[i];
    if (__any(d) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
static int upload_square_on(eagle_ctx* ctx, const double* host, long n, long np, double* dev, hipStream_t st) {
    HIPCHK(ctx, hipMemsetAsync(dev, 0, sizeof(double) * np * np, st));
    HIPCHK(ctx, hipMemcpy2DAsync(dev, sizeof(double) * np, host, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, st));
    return EAGLE_OK;
}
// column-major n x n host matrix -> zero padded np x np device image (row-major image of the transpose)
static int upload_square(eagle_ctx* ctx, const double* host, long n, long np, double* dev) {
    HIPCHK(ctx, hipMemsetAsync(dev, 0, sizeof(double) * np * np, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dev, sizeof(double) * np, host, sizeof(double) * n, sizeof(double) * n, n,
                                 hipMemcpyHostToDevice, ctx->stream));
    return EAGLE_OK;
}
// The same upload through the context's two pinned staging buffers: host threads copy piece k + 1 of the caller's pageable matrix into
// one while the DMA of piece k runs out of the other (the runtime's own pageable path stages single-threaded: 25-27 GB/s on the GPU
// boxes against 50+ this way).  Synchronous for the caller's memory: nothing of `host` is in flight when it returns.
// block_rows / on_block: after every block of `block_rows` rows of the image (the last one runs to n_pad: padding rows are zero) an event is
// recorded on `st` and on_block(r0, r1, event) is called -- the caller makes its compute stream wait for the event and works on the rows.
static int upload_square_staged(eagle_ctx* ctx, const double* host, long n, long np, double* dev, hipStream_t st, long block_rows = 0,
                                const std::function<int(long, long, hipEvent_t)>& on_block = nullptr) {
    const size_t piece = (size_t)64 << 20, rowb = sizeof(double) * (size_t)n;
    if (!on_block && (rowb * (size_t)n < 4 * piece || rowb > piece)) return upload_square_on(ctx, host, n, np, dev, st);
    if (rowb > piece) return eagle_fail(ctx, EAGLE_ERR_ARG, "staged upload: a row does not fit a staging buffer");
    int rc = eagle_stage_ensure(ctx, piece);
    if (rc) return rc;
    HIPCHK(ctx, hipMemsetAsync(dev, 0, sizeof(double) * np * np, st));
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    const long rows_per = (long)(piece / rowb);
    const int threads = std::max(1, std::min(host_threads(), 16));
    hipError_t e = hipSuccess;
    std::vector<hipEvent_t> block_events;
    int rcb = EAGLE_OK;
    long k = 0, block0 = 0;
    double t_wait = 0, t_copy = 0, t_issue = 0, t_cb = 0;   // (EAGLE_HIP_TIMING: where the host side of the upload spends its time)
    const bool timed = timing_on();
    const double t_begin = timed ? now_s() : 0.0;
    for (long r0 = 0; r0 < n && e == hipSuccess && !rcb; k++) {
        long nr = std::min(rows_per, n - r0);
        if (on_block && block_rows > 0) nr = std::min(nr, block0 + block_rows - r0);   // pieces end at block boundaries
        double tq = timed ? now_s() : 0.0;
        if (k >= 2) e = hipEventSynchronize(ev[k & 1]);   // the DMA that last read this buffer
        if (e != hipSuccess) break;
        if (timed) { const double t = now_s(); t_wait += t - tq; tq = t; }
        char* dst = (char*)ctx->stage_pin[k & 1];
        const char* src = (const char*)(host + r0 * n);
        parallel_for((long)(rowb * (size_t)nr), threads, [&](long a, long b, int) { memcpy(dst + a, src + a, (size_t)(b - a)); });
        if (timed) { const double t = now_s(); t_copy += t - tq; tq = t; }
        e = hipMemcpy2DAsync(dev + r0 * np, sizeof(double) * np, dst, rowb, rowb, (size_t)nr, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ev[k & 1], st);
        if (timed) { const double t = now_s(); t_issue += t - tq; tq = t; }
        r0 += nr;
        if (on_block && e == hipSuccess && (r0 == block0 + block_rows || r0 == n)) {
            // the block [block0, r0) is on its way; the last one takes the zero padding rows n .. n_pad along (block_rows is a multiple
            // of 256, so a block that ends exactly at n leaves no padding rows behind)
            const long end = (r0 == n) ? np : r0;
            hipEvent_t be = nullptr;
            e = hipEventCreateWithFlags(&be, hipEventDisableTiming);
            if (e == hipSuccess) { block_events.push_back(be); e = hipEventRecord(be, st); }
            if (e == hipSuccess) rcb = on_block(block0, end, be);
            block0 = end;
            if (timed) t_cb += now_s() - tq;
        }
    }
    if (timed) {
        const double t_loop = now_s() - t_begin;
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[eaglehip] staged upload %ld x %ld (%d threads, %ld pieces): host loop %.1f ms = waiting for a buffer %.1f + memcpy %.1f (%.1f GB/s) + "
                "issue %.1f + block callbacks %.1f; all landed after %.1f ms (%.1f GB/s)\n", n, n, threads, k, t_loop * 1e3, t_wait * 1e3, t_copy * 1e3,
                rowb * (double)n / 1e9 / std::max(t_copy, 1e-9), t_issue * 1e3, t_cb * 1e3, (now_s() - t_begin) * 1e3, rowb * (double)n / 1e9 / (now_s() - t_begin));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // (the staging buffers serve other loaders after this call)
    for (int b = 0; b < 2; b++) (void)hipEventDestroy(ev[b]);
    for (hipEvent_t be : block_events) (void)hipEventDestroy(be);
    if (rcb) { (void)hipStreamSynchronize(st); return rcb; }
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return eagle_fail_hip(ctx, e, "staged upload"); }
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

// grow-only ctx-owned device buffer
template <class T> static int ensure_buf(eagle_ctx* ctx, T** buf, size_t* cap, size_t bytes, const char* what) {
    if (bytes <= *cap) return EAGLE_OK;
    if (*buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    hipError_t e = hipMalloc((void**)buf, bytes);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, what);
    *cap = bytes;
    return EAGLE_OK;
}

// Exact int32 partial  sum_{c in [c0, c1)} m_c m_c^T  into ctx->d_c32 (zeroed here; upper 256-tiles live), with the columns of
// `sel` that fall into the range masked (calculateMMt_rcpp.cpp:88-92 as an exact rank-k downdate).  The window of M.ascii is
// kept resident if it fits, else streamed through HBM in marker windows.
static int mmt_range(eagle_ctx* ctx, const char* path, long n, long L, long c0, long c1, const std::vector<long>& sel, double mem_gb,
                     int threads, int quiet) {
    const long np = eagle_pad(n);
    int rc = ensure_buf(ctx, &ctx->d_c32, &ctx->c32_cap, sizeof(int32_t) * (size_t)np * np, "MM^T accumulator hipMalloc");
    if (rc) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_c32, 0, sizeof(int32_t) * (size_t)np * np, ctx->stream));
    const long Lw_all = c1 - c0;
    if (Lw_all <= 0) return EAGLE_OK;
    GenoEntry* g = nullptr;
    rc = get_resident(ctx, path, 0, n, c0, Lw_all, mem_gb, threads, &g, sizeof(int32_t) * (size_t)np * np);
    if (rc < 0) return rc;
    std::vector<long> in_range;
    for (long c : sel) if (c >= c0 && c < c1) in_range.push_back(c - c0);
    DevBuf dsel, win;
    if (rc == EAGLE_OK) {
        // the fp4 operand image stays with the resident file: a later calculateMMt on it (SummaryAM re-calls .calcMMt,
        // E/R/summary_am.R:140) is the SYRK + finish only
        const long Lwp = eagle_pad(Lw_all);
        if (!g->dev_f4) {
            hipError_t e = hipMalloc(&g->dev_f4, (size_t)np * (size_t)(Lwp / 2));
            if (e != hipSuccess) { g->dev_f4 = nullptr; (void)hipGetLastError(); }
            else if ((rc = eagle_dev_pack_fp4(ctx, g->dev, np, Lwp, g->ld, g->dev_f4, ctx->stream))) { (void)hipFree(g->dev_f4); g->dev_f4 = nullptr; return rc; }
        }
        rc = g->dev_f4 ? eagle_dev_mmt_accumulate_f4(ctx, g->dev_f4, np, Lwp, Lwp / 2, ctx->d_c32, ctx->stream)
                       : eagle_dev_mmt_accumulate(ctx, g->dev, np, Lwp, g->ld, ctx->d_c32, ctx->stream);  // no room: pack into the ctx buffer per call
        if (rc) return rc;
        if (!in_range.empty()) {
            HIPCHK(ctx, dsel.alloc(sizeof(long) * in_range.size()));
            HIPCHK(ctx, hipMemcpyAsync(dsel.p, in_range.data(), sizeof(long) * in_range.size(), hipMemcpyHostToDevice, ctx->stream));
            rc = eagle_dev_mmt_downdate(ctx, g->dev, np, g->ld, dsel.as<long>(), (long)in_range.size(), ctx->d_c32, ctx->stream);
            if (rc) return rc;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        return EAGLE_OK;
    }
    // the window does not fit (or may not stay) in HBM: stream column windows of every line (= marker chunks) and accumulate
    // the exact integer partial products, MMt = sum_w M_w M_w^T.
    const long Lw = stream_chunk_rows(np, eagle_pad(Lw_all));  // window width in markers; rows of the window = np
    HIPCHK(ctx, win.alloc((size_t)2 * np * Lw));
    ChunkRing ring;
    if ((rc = ring.init(ctx))) return rc;
    ring.buf[0] = win.as<int8_t>();
    ring.buf[1] = win.as<int8_t>() + (size_t)np * Lw;
    if (!quiet) say(ctx, " M.ascii streamed through HBM in windows of %ld markers", Lw);
    for (long w0 = 0; w0 < Lw_all; w0 += Lw) {
        const long nc = std::min(Lw, Lw_all - w0);
        int8_t* wtile = nullptr;
        rc = ring.load(ctx, path, 0, n, c0 + w0, nc, Lw, (size_t)np * Lw, mem_gb, threads, &wtile);
        if (rc) return rc;
        rc = eagle_dev_mmt_accumulate(ctx, wtile, np, Lw, Lw, ctx->d_c32, ctx->stream);
        if (rc) return rc;
        std::vector<long> in_win;
        for (long c : in_range) if (c >= w0 && c < w0 + nc) in_win.push_back(c - w0);
        if (!in_win.empty()) {
            DevBuf dw;
            HIPCHK(ctx, dw.alloc(sizeof(long) * in_win.size()));
            HIPCHK(ctx, hipMemcpyAsync(dw.p, in_win.data(), sizeof(long) * in_win.size(), hipMemcpyHostToDevice, ctx->stream));
            // duplicates of one column must be dropped across the whole list, which k_mmt_downdate does per call:
            // selected_loci entries are distinct columns in any sane call; duplicates inside one window are handled
            rc = eagle_dev_mmt_downdate(ctx, wtile, np, Lw, dw.as<long>(), (long)in_win.size(), ctx->d_c32, ctx->stream);
            if (rc) return rc;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        if ((rc = ring.computed(ctx))) return rc;
    }
    ring.finish(ctx);
    return EAGLE_OK;
}

static int download_big(eagle_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);   // below, with the other result paths
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
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    const long packed = eagle_upper_tiles_count(np);
    RcclState* rccl = (RcclState*)ctx->rccl;
    if (rccl && rccl->dead) rccl = nullptr;   // aborted in an earlier call: host-staged sums from now on
    Rendezvous rv;
    rv.n = nd;
    rc = run_on_devices(ctx, [&](int k, eagle_ctx* c) -> int {
        hipError_t e = hipSetDevice(c->device);
        int r = e == hipSuccess ? mmt_range(c, f_name_ascii, n, L, edge[k], edge[k + 1], sel, max_memory_in_Gbytes, threads, quiet)
                                : eagle_fail_hip(c, e, "hipSetDevice");
        if (nd == 1 && !rccl) return r;
        // ONE sum of the partials: the upper 256-tiles of every device's accumulator, packed, to the lead
        if (!r) r = ensure_buf(c, &c->d_pack, &c->pack_cap, sizeof(int32_t) * (size_t)packed, "packed MM^T hipMalloc");
        if (!r) r = eagle_dev_tiles_pack(c, c->d_c32, np, c->d_pack, 0, c->stream);
        if (!r && (e = hipStreamSynchronize(c->stream)) != hipSuccess) r = eagle_fail_hip(c, e, "MM^T partial");
        if (!rv.arrive(r == 0)) return r ? r : eagle_fail(c, EAGLE_ERR_HIP, "another device failed before the MM^T sum");
        if (rccl) {
            const ncclComm_t comm = rccl_comm(rccl, k);
            ncclResult_t nr = (rccl_fault("reduce", k) || !comm) ? ncclSystemError
                                                   : rccl->Reduce(c->d_pack, c->d_pack, (size_t)packed, ncclInt32, ncclSum, 0, comm, c->stream);
            if (nr != ncclSuccess) { r = failf(c, EAGLE_ERR_HIP, "ncclReduce: %s", rccl->GetErrorString(nr)); rccl_abort_all(rccl); }
            else if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) r = eagle_fail_hip(c, e, "ncclReduce sync");
        } else if (k == 0) {  // host-staged stand-in: the peers' packed tiles are copied to the lead and added, one by one
            r = ensure_buf(c, &c->d_pack2, &c->pack2_cap, sizeof(int32_t) * (size_t)packed, "packed MM^T landing buffer");
            for (int p = 1; p < nd && !r; p++) {
                eagle_ctx* pc = dev_ctx(ctx, p);
                e = pc->device == c->device ? hipMemcpyAsync(c->d_pack2, pc->d_pack, sizeof(int32_t) * (size_t)packed, hipMemcpyDeviceToDevice, c->stream)
                                            : hipMemcpyPeerAsync(c->d_pack2, c->device, pc->d_pack, pc->device, sizeof(int32_t) * (size_t)packed, c->stream);
                if (e != hipSuccess) { r = eagle_fail_hip(c, e, "peer copy of a partial MM^T"); break; }
                r = eagle_dev_add_i32(c, c->d_pack, c->d_pack2, packed, c->stream);
            }
            if (!r && (e = hipStreamSynchronize(c->stream)) != hipSuccess) r = eagle_fail_hip(c, e, "staged MM^T sum");
        }
        rv.arrive(r == 0);  // the peers keep their buffers until the lead has read them
        return r;
    });
    if (rc) return rc;
    if ((nd > 1 || rccl) && (rc = eagle_dev_tiles_pack(ctx, ctx->d_c32, np, ctx->d_pack, 1, ctx->stream))) return rc;
    if (ctx->mmt_n != n) {
        if (ctx->d_mmt) { (void)hipFree(ctx->d_mmt); ctx->d_mmt = nullptr; }
        ctx->mmt_n = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_mmt, sizeof(double) * (size_t)n * n));
        ctx->mmt_n = n;
    }
    if (!ctx->d_mmt_max) HIPCHK(ctx, hipMalloc((void**)&ctx->d_mmt_max, sizeof(double)));
    rc = eagle_dev_mmt_finish(ctx, ctx->d_c32, n, np, ctx->d_mmt, n, ctx->d_mmt_max, ctx->stream);
    if (rc) return rc;
    // AM() calls this once, then works on the host for seconds (eigen, REML) before its first find_qtl: the arena of that scan is
    // reserved meanwhile, on a background thread (a no-op below 4 GB)
    (void)prepare_scan(ctx, n, L, true);
    return download_big(ctx, MMt_out, ctx->d_mmt, sizeof(double) * (size_t)n * n);
}

// A large result back into the caller's (pageable) memory: the runtime's own pageable copy moves ~10 GB/s (80 ms for the 800 MB
// of MM^T at n = 10,000, most of a warm eagle_calculateMMt call).  Through the ctx's two pinned staging buffers instead: the DMA of
// piece k+1 (57 GB/s) runs under the host threads' copy of piece k into the destination.  Synchronises ctx->stream.
static int download_big(eagle_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    const size_t piece = (size_t)64 << 20;
    if (bytes < 4 * piece) {
        HIPCHK(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return EAGLE_OK;
    }
    int rc = eagle_stage_ensure(ctx, piece);
    if (rc) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) HIPCHK(ctx, hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    const int threads = std::max(1, std::min(host_threads(), 16));
    const size_t np_ = (bytes + piece - 1) / piece;
    hipError_t e = hipSuccess;
    for (size_t k = 0; k <= np_ && e == hipSuccess; k++) {
        if (k < np_) {   // piece k into buffer k & 1 (its previous content, piece k - 2, was copied out in iteration k - 1)
            const size_t off = k * piece, len = std::min(piece, bytes - off);
            e = hipMemcpyAsync(ctx->stage_pin[k & 1], (const char*)dev_src + off, len, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ev[k & 1], ctx->stream);
        }
        if (k > 0 && e == hipSuccess) {   // piece k - 1 out of its buffer, under the DMA of piece k
            const size_t off = (k - 1) * piece, len = std::min(piece, bytes - off);
            e = hipEventSynchronize(ev[(k - 1) & 1]);
            if (e == hipSuccess) {
                const char* src = (const char*)ctx->stage_pin[(k - 1) & 1];
                char* dst = (char*)host_dst + off;
                parallel_for((long)len, threads, [&](long a, long b, int) { memcpy(dst + a, src + a, (size_t)(b - a)); });
            }
        }
    }
    for (int b = 0; b < 2; b++) (void)hipEventDestroy(ev[b]);
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return eagle_fail_hip(ctx, e, "staged download"); }
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
    if (max_out) HIPCHK(ctx, hipMemcpyAsync(max_out, ctx->d_mmt_max, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return download_big(ctx, MMt_norm_out, tmp.p, sizeof(double) * (size_t)n * n);
}

// what scan_range's setup reserves for a resident scan of Lr markers on n individuals (the same terms)
static size_t scan_arena_bytes(eagle_ctx* ctx, long n, long Lr) {
    const long np = eagle_pad(n), Lp = eagle_pad(Lr > 0 ? Lr : 1);
    const size_t sq = sizeof(double) * (size_t)np * np;
    const bool use_i8 = ctx->scan_mode == 1 && 64.0 * 512.0 * (double)np < 2147483648.0;
    const int nslices = ctx->scan_slices | (ctx->scan_stochastic ? EAGLE_SLICES_STOCHASTIC : 0);
    const size_t wsb = use_i8 ? (size_t)eagle_vara_i8_workspace_bytes(np, Lp, nslices) : 0;
    const size_t certb = use_i8 ? (size_t)eagle_scan_certify_workspace_bytes(np) : 0;
    return 4 * arena_round(sq) + 2 * arena_round(sizeof(double) * np) + arena_round(wsb) + arena_round(certb);
}
static int prepare_scan(eagle_ctx* ctx, long n, long L, bool only_if_roomy) {
    if (!ctx || n <= 0 || L <= 0) return EAGLE_ERR_ARG;
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    (void)hipSetDevice(ctx->device);
    arena_prefetch(ctx, scan_arena_bytes(ctx, n, edge[1] - edge[0]), only_if_roomy);
    for (size_t k = 0; k < ctx->peers.size(); k++) {
        (void)hipSetDevice(ctx->peers[k]->device);
        arena_prefetch(ctx->peers[k], scan_arena_bytes(ctx->peers[k], n, edge[k + 2] - edge[k + 1]), only_if_roomy);
    }
    (void)hipSetDevice(ctx->device);
    return EAGLE_OK;
}
extern "C" int eagle_prepare_scan(eagle_ctx* ctx, long n, long L) { return prepare_scan(ctx, n, L, false); }

static int ensure_scan_out(eagle_ctx* ctx, long L_pad) {
    if (ctx->scan_cap >= L_pad) return EAGLE_OK;
    if (ctx->d_a) { (void)hipFree(ctx->d_a); ctx->d_a = nullptr; }
    if (ctx->d_vara) { (void)hipFree(ctx->d_vara); ctx->d_vara = nullptr; }
    if (ctx->d_bound) { (void)hipFree(ctx->d_bound); ctx->d_bound = nullptr; }
    ctx->scan_cap = 0;
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_a, sizeof(double) * (size_t)L_pad));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_vara, sizeof(double) * (size_t)L_pad));
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_bound, sizeof(double) * (size_t)L_pad));
    ctx->scan_cap = L_pad;
    return EAGLE_OK;
}

// The scan of the markers [m0, m1) of Mt.ascii on ctx's device: a, vara into ctx->d_a / d_vara (and the host ranges
// a_out[m0..m1), vara_out[m0..m1)).  k of nd devices; rv / rccl: the meeting point and the communicators of a multi-device call
// (NULL for one device).  Every device passes the same rendezvous in the same order, failed or not.
static int scan_range(eagle_ctx* ctx, const char* f_name_ascii, long L, long n, long m0, long m1, const std::vector<long>& sel,
                      const double* inv_MMt_sqrt, const double* dim_reduced_vara, const double* a, double max_memory_in_Gbytes, int quiet,
                      double* a_out, double* vara_out, int k, int nd, Rendezvous* rv, RcclState* rccl, bool w_direct = false, bool s_trusted = false) {
    // s_trusted: the device copy of S was uploaded from this very caller matrix a moment ago (the re-run after a deferred
    // verification found another S): no verification this time
    // w_direct (eagle_scan_with_W): inv_MMt_sqrt is not S but W itself, `a` is v = S a_hat; dim_reduced_vara is unused
    const long Lr = m1 - m0;
    const long np = eagle_pad(n), Lp = eagle_pad(Lr > 0 ? Lr : 1);
    const size_t sq = sizeof(double) * (size_t)np * np;
    const bool use_i8 = ctx->scan_mode == 1 && 64.0 * 512.0 * (double)np < 2147483648.0;
    const int nslices = ctx->scan_slices | (ctx->scan_stochastic ? EAGLE_SLICES_STOCHASTIC : 0);
    // W's rows shared between devices + ONE all-gather: only for the fp64 products.  The int8 digit-slice products (round 4) cost half
    // as much, are a deterministic function of S and V -- every device forms the bits a single device forms, no exchange, no all-gather
    // of 8 n^2 bytes, each device's copy of S stays cached and verified -- and so run replicated on every device.
    const bool share_w = !w_direct && rccl && (np / 128) % nd == 0 && !(use_i8 && eagle_w8_wanted(ctx, np));  // the same answer on every device
    int rc = EAGLE_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "hipSetDevice");
#define EAGLE_ARRIVE(v) EAGLE_ARRIVE2(v, 0)
#define EAGLE_ARRIVE2(v, add)                                                                                      \
    do {                                                                                                           \
        if (rv && !rv->arrive(rc == EAGLE_OK, (v), (add))) return rc ? rc : eagle_fail(ctx, EAGLE_ERR_HIP, "another device of the scan failed"); \
        if (!rv && rc) return rc;                                                                                  \
    } while (0)
    GenoEntry* g = nullptr;
    bool streamed = false;
    if (!rc && Lr > 0) {
        // operands, digit + certification workspaces (the arena: what this context already holds of it -- from an earlier call or from
        // the background reservation -- is not free memory any more and must not be counted against the file a second time), the
        // re-centred image of the shard
        arena_collect(ctx);
        const size_t arena_need = 4 * sq + (use_i8 ? (size_t)eagle_vara_i8_workspace_bytes(np, Lp, nslices) + (size_t)eagle_scan_certify_workspace_bytes(np) : 0);
        int r = get_resident(ctx, f_name_ascii, m0, Lr, 0, n, max_memory_in_Gbytes, host_threads(), &g,
                             (arena_need > ctx->arena_cap ? arena_need - ctx->arena_cap : 0) + (use_i8 ? (size_t)Lp * np + 9 * (size_t)Lp : 0) +
                                 ((size_t)1 << 30));
        if (r < 0) rc = r;
        streamed = (r == EAGLE_STREAM);
    }
    EAGLE_ARRIVE(streamed ? 1.0 : 0.0);
    // A scan that does not see all its markers at once -- a file streamed in marker blocks, the shards of several devices -- is
    // certified against ONE lower bound of the maximum over every block of every device, so that the candidates, and with them every
    // returned bit, are those of the one-block scan of the whole file (the per-marker bounds of all blocks stay: 8 bytes per marker).
    const bool bounds_flow = use_i8 && (streamed || rv);
    // one resident block on one device: the caller's S is compared with a HOST copy of the cached one (memcmp on the cores that idle while
    // the card works) -- measured: the 800 MB upload + device comparison under the vara kernel cost the call 3.4 ms of 141.5 at the headline size
    const bool host_verify = !rv && !getenv("EAGLE_HIP_NO_HOST_SVERIFY");   // (a streamed file too: the comparison runs beside the loaders)
    long over_tight_all = 0;   // markers of the whole scan over the tight threshold (CERT_TIGHT_MAX)
    const long Lc = streamed ? stream_chunk_rows(np, Lp) : Lp;  // marker rows per pass
    DevBuf dsel;
    const size_t wsb = use_i8 ? (size_t)eagle_vara_i8_workspace_bytes(np, Lc, nslices) : 0;
    const size_t certb = use_i8 ? (size_t)eagle_scan_certify_workspace_bytes(np) : 0;
    const double t0 = now_s();
    double *Sa = nullptr, *Va = nullptr, *tmp = nullptr, *Wu = nullptr, *ah = nullptr, *v = nullptr;
    bool s_from_cache = false;  // the product runs on the device copy of the last call's S; the caller's S is verified under it
    bool s_check_pending = false;  // ... and the outcome of that verification has not been collected yet
    struct Helper {   // the thread that issues a deferred verification (joined on every way out)
        std::thread t;
        int rc = EAGLE_OK;
        void start(const std::function<int()>& f) { t = std::thread([this, f] { rc = f(); }); }
        int join() { if (t.joinable()) t.join(); return rc; }
        ~Helper() { if (t.joinable()) t.join(); }
    } s_verify;
    int host_flag_store = 0;
    int* const host_flag = &host_flag_store;   // outcome of a HOST-side verification (1: the caller's S differs from the cached one)
    int s_check_kind = 0;   // what s_check_pending waits for: 1 = the device comparison, 2 = the host comparison, 3 = only the refresh of the host copy
    void *ws = nullptr, *cert = nullptr;
    long* cert_totals = (long*)((char*)ctx->d_scratch + EAGLE_SCR_CERT_TOTALS);  // {re-evaluated, flagged, fell back, over the tight threshold}, summed over marker blocks
    ChunkRing ring;
    int8_t* shifted[2] = {nullptr, nullptr};
    int8_t* cs[2] = {nullptr, nullptr};
    int32_t* l1s[2] = {nullptr, nullptr};
    PhaseEvents ph;
    bool s_in_arena = false;   // n_pad > 16,384: the arena slot still holds the last scan's S (see eagle_ctx.h)
    // the device copy is (about to be) this caller's S: its host copy is made off the critical path, joined before the call returns
    auto refresh_host_copy = [&]() {
        const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
        // (20 GB at n = 50,000: only when the host has that to spare four times over -- an overcommitted malloc would succeed and the
        // copy then meet the OOM killer; without a host copy the small sizes verify on the device and the large ones upload S every call)
        const long pages = sysconf(_SC_AVPHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
        if (bytes > ctx->h_Scache_cap && pages > 0 && psz > 0 && (double)bytes > 0.25 * (double)pages * (double)psz) return;
        if (bytes > ctx->h_Scache_cap) {
            free(ctx->h_Scache);
            ctx->h_Scache = (double*)malloc(bytes);
            ctx->h_Scache_cap = ctx->h_Scache ? bytes : 0;
        }
        if (!ctx->h_Scache) return;
        s_check_pending = true;   // (nothing to check: only the join)
        s_check_kind = 3;
        s_verify.start([ctx, inv_MMt_sqrt, n, bytes]() -> int {
            parallel_for((long)bytes, std::max(1, std::min(host_threads(), 16)), [&](long a, long b, int) {
                memcpy((char*)ctx->h_Scache + a, (const char*)inv_MMt_sqrt + a, (size_t)(b - a));
            });
            ctx->h_Scache_n = n;
            return EAGLE_OK;
        });
    };
    auto setup = [&]() -> int {
        const void* kept = (ctx->arena_S_ptr && ctx->arena_S_base == ctx->arena && ctx->arena_S_n == n && ctx->arena_S_np == np) ? ctx->arena_S_ptr : nullptr;
        ctx->arena_S_ptr = nullptr;   // (valid again only when this scan has gone through)
        int r = arena_reserve(ctx, 4 * arena_round(sq) + 2 * arena_round(sizeof(double) * np) + arena_round(wsb) + arena_round(certb) +
                                       (streamed ? (use_i8 ? 4 : 2) * arena_round((size_t)Lc * np) + 2 * arena_round((size_t)Lc) +
                                                       2 * arena_round(2 * sizeof(int32_t) * (size_t)Lc) : 0));
        if (r) return r;
        Sa = arena_take<double>(ctx, sq);
        s_in_arena = kept && kept == (const void*)Sa && ctx->arena_S_base == ctx->arena;
        Va = arena_take<double>(ctx, sq);
        tmp = arena_take<double>(ctx, sq);
        Wu = arena_take<double>(ctx, sq);
        ah = arena_take<double>(ctx, sizeof(double) * np);
        v = arena_take<double>(ctx, sizeof(double) * np);
        ws = arena_take<char>(ctx, wsb);
        cert = arena_take<char>(ctx, certb);
        ph.mark(ctx->stream, PH_START);
        if (streamed) {
            if ((r = ring.init(ctx))) return r;
            ring.buf[0] = arena_take<int8_t>(ctx, (size_t)Lc * np);
            ring.buf[1] = arena_take<int8_t>(ctx, (size_t)Lc * np);
            for (int b = 0; b < 2 && use_i8; b++) {
                shifted[b] = arena_take<int8_t>(ctx, (size_t)Lc * np);
                cs[b] = arena_take<int8_t>(ctx, (size_t)Lc);
                l1s[b] = arena_take<int32_t>(ctx, 2 * sizeof(int32_t) * (size_t)Lc);
            }
        }
        if (w_direct) {  // W and v arrive ready: no n^3 work, W goes straight into the buffer the fold works on
            if ((r = upload_square(ctx, inv_MMt_sqrt, n, np, Wu))) return r;
            if ((r = upload_vec(ctx, a, n, np, v))) return r;
        } else {
            // S = inv_MMt_sqrt is MMt^-1/2, the same matrix in every call of an AM() run, but R builds it anew each time (no pointer
            // identity, and hashing 800 MB costs what copying them costs): the copy of the last call stays on the device, this call
            // computes on it at once, and the caller's matrix is uploaded on the loader stream and compared bit for bit while the
            // n^3 products run; a difference starts the products over (below).  V changes with every call; it arrives in row
            // blocks on the loader stream UNDER the first product, which works on the rows that have landed (below) -- unless the
            // rows of W are shared between devices (share_w: the row-block product there needs all of V at once).
            if (share_w && (r = upload_square_staged(ctx, dim_reduced_vara, n, np, Va, ctx->stream))) return r;
            if ((r = upload_vec(ctx, a, n, np, ah))) return r;
            // (EAGLE_HIP_SCACHE_MAX_NP: tests force the large-n form -- S kept in its arena slot, host comparison -- at small sizes)
            const long scache_max_np = getenv("EAGLE_HIP_SCACHE_MAX_NP") ? atol(getenv("EAGLE_HIP_SCACHE_MAX_NP")) : 16384;
            const bool cacheable = np <= scache_max_np && !getenv("EAGLE_HIP_NO_SCACHE");   // (round 4: also when W's rows are shared)
            if (cacheable && ctx->scache_np != np) {
                if (ctx->d_Scache) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->d_Scache); ctx->d_Scache = nullptr; }
                if (ctx->d_Sscr) { (void)hipFree(ctx->d_Sscr); ctx->d_Sscr = nullptr; }
                ctx->scache_n = ctx->scache_np = 0;
                if (hipMalloc((void**)&ctx->d_Scache, sq) == hipSuccess && hipMalloc((void**)&ctx->d_Sscr, sq) == hipSuccess) ctx->scache_np = np;
                else { if (ctx->d_Scache) (void)hipFree(ctx->d_Scache); ctx->d_Scache = nullptr; (void)hipGetLastError(); }
            }
            if (cacheable && ctx->scache_np == np) {
                // (EAGLE_HIP_EXPERIMENT_TRUST_S: measurement only -- what the verification traffic costs the kernels it runs under; never set it
                // for real work: a changed S would go unnoticed)
                const bool trust = s_trusted || (ctx->scache_n == n && getenv("EAGLE_HIP_EXPERIMENT_TRUST_S"));
                s_from_cache = ctx->scache_n == n && !trust;
                if (!s_from_cache && !trust) {
                    if ((r = upload_square(ctx, inv_MMt_sqrt, n, np, ctx->d_Scache))) return r;
                    ctx->h_Scache_n = 0;
                }
                if (!s_from_cache && host_verify && ctx->h_Scache_n != n) refresh_host_copy();
                ctx->scache_n = n;
                Sa = ctx->d_Scache;
            } else if (!getenv("EAGLE_HIP_NO_SCACHE") && s_in_arena && host_verify && !s_trusted && ctx->h_Scache && ctx->h_Scache_n == n) {
                // n_pad > 16,384: S is where the last scan left it; the host comparison decides at the end of the call
                s_from_cache = true;
            } else {
                if ((r = upload_square(ctx, inv_MMt_sqrt, n, np, Sa))) return r;
                ctx->h_Scache_n = 0;
                if (host_verify && !getenv("EAGLE_HIP_NO_SCACHE")) refresh_host_copy();
            }
        }
        // streamed: every block is worked as a full chunk of Lc rows (one workspace layout for all of them; the rows beyond a short
        // last block are zero and land in the slack behind the shard's results)
        if ((r = ensure_scan_out(ctx, streamed ? (Lr + Lc - 1) / Lc * Lc : Lp))) return r;
        HIPCHK(ctx, hipMemsetAsync(cert_totals, 0, 4 * sizeof(long), ctx->stream));
        ph.mark(ctx->stream, PH_UPLOAD);
        return EAGLE_OK;
    };
    if (!rc) rc = setup();
    const double t_setup = now_s();
    double t1 = 0;
    if (timing_on() && !rc) { (void)hipStreamSynchronize(ctx->stream); t1 = now_s(); }
    // W = S (V S) and v = S a_hat.  Several devices: each computes 1/nd of the rows of W's image and ONE all-gather completes
    // it everywhere (the n^3 part would otherwise not scale); without device collectives every device computes all of it.
    if (share_w || w_direct) { ctx->w8_active = false; ctx->w8_info = W8Info(); ctx->w8_info.declined = 7; }
    if (share_w) {
        const long rows = np / nd, r0 = k * rows;
        if (!rc) rc = eagle_dev_scan_operands_rows(ctx, Sa, Va, ah, n, np, r0, r0 + rows, v, Wu, tmp, ctx->stream);
        if (!rc && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "rows of W");
        EAGLE_ARRIVE(0.0);
        const ncclComm_t comm = rccl_comm(rccl, k);
        ncclResult_t nr = (rccl_fault("allgather", k) || !comm) ? ncclSystemError
                                                  : rccl->AllGather(Wu + r0 * np, Wu, (size_t)(rows * np), ncclDouble, comm, ctx->stream);
        if (nr != ncclSuccess) { rc = failf(ctx, EAGLE_ERR_HIP, "ncclAllGather: %s", rccl->GetErrorString(nr)); rccl_abort_all(rccl); }
        if (!rc) rc = eagle_dev_fold_upper(ctx, Wu, np, ctx->stream);
    } else if (!rc) {
        if (w_direct) rc = eagle_dev_fold_upper(ctx, Wu, np, ctx->stream);
        else {
            // v = S a_hat at once; then V in blocks of 1024 rows of its image: copy on the loader stream, event, the rows X = V S of
            // that block on the compute stream.  PCIe (57 GB/s) delivers a block in half the time its product takes, so all of
            // V's upload but the first block hides under the first n^3 product.
            if (eagle_w8_wanted(ctx, np)) {
                // W on the int8 engine (csrc/eagle_w8.hip): its configuration is chosen from statistics of ALL of V, so V is uploaded
                // whole (loader stream; S's statistics and slices do not wait for it) and the products follow; a call that declines
                // runs the fp64 products on the resident operands
                // Since the first version of round 4 the upload is pipelined when this context has a configuration to guess (its last
                // call's): V goes up in blocks of 1,536 rows through the pinned staging buffers (16 host threads), and each block's
                // statistics, digit slices and columns of the first product run on the compute stream as soon as its rows have landed;
                // the finish step keeps that product only if the rule, on the statistics of ALL of V, chooses the guessed configuration.
                int r8 = eagle_w8_begin(ctx, Sa, Va, ah, n, np, v, Wu, tmp, 1, ctx->stream);
                if (r8 < 0) rc = r8;
                if (!rc) {
                    const long vb = eagle_w8_vrows_block();
                    rc = upload_square_staged(ctx, dim_reduced_vara, n, np, Va, ctx->load_stream, vb, [&](long r0, long r1, hipEvent_t landed) -> int {
                        if (r8) return EAGLE_OK;   // declined at the start (no workspace): only the upload
                        hipError_t ew = hipStreamWaitEvent(ctx->stream, landed, 0);
                        if (ew != hipSuccess) return eagle_fail_hip(ctx, ew, "row block event");
                        const int rv8 = eagle_w8_vrows(ctx, r0, r1, ctx->stream);
                        if (rv8 < 0) return rv8;
                        if (rv8) r8 = 1;
                        return EAGLE_OK;
                    });
                }
                if (!rc && !r8) {
                    r8 = eagle_w8_finish(ctx, ctx->stream);
                    if (r8 < 0) rc = r8;
                }
                if (!rc && r8 == 1) {   // declined: the fp64 products on the operands that are resident now (v is made already)
                    hipEvent_t ev = nullptr;
                    if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "hipEventCreate");
                    if (!rc && ((e = hipEventRecord(ev, ctx->load_stream)) != hipSuccess || (e = hipStreamWaitEvent(ctx->stream, ev, 0)) != hipSuccess))
                        rc = eagle_fail_hip(ctx, e, "V upload event");
                    ctx->w8_active = false;
                    if (!rc) rc = eagle_dev_scan_operands_w_f64(ctx, Sa, Va, np, Wu, tmp, ctx->stream);
                    if (ev) (void)hipEventDestroy(ev);
                }
                if (rc) (void)hipStreamSynchronize(ctx->load_stream);
            } else {
            ctx->w8_active = false;
            ctx->w8_info = W8Info();
            ctx->w8_info.declined = 7;
            rc = eagle_dev_scan_operands_begin(ctx, Sa, ah, n, np, v, tmp, ctx->stream);
            if (!rc && (e = hipMemsetAsync(Va, 0, sq, ctx->load_stream)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "V memset");
            std::vector<hipEvent_t> landed;
            for (long b0 = 0; b0 < np && !rc; b0 += EAGLE_VROWS_BLOCK) {
                const long b1 = std::min(np, b0 + EAGLE_VROWS_BLOCK), hr = std::min(n, b1) - b0;  // rows of the image = columns of the R matrix
                if (hr > 0) {
                    e = hipMemcpy2DAsync(Va + b0 * np, sizeof(double) * np, dim_reduced_vara + b0 * n, sizeof(double) * n, sizeof(double) * n, (size_t)hr,
                                         hipMemcpyHostToDevice, ctx->load_stream);
                    if (e != hipSuccess) { rc = eagle_fail_hip(ctx, e, "upload of a row block of V"); break; }
                }
                hipEvent_t ev = nullptr;
                if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) { rc = eagle_fail_hip(ctx, e, "hipEventCreate"); break; }
                landed.push_back(ev);
                if ((e = hipEventRecord(ev, ctx->load_stream)) != hipSuccess || (e = hipStreamWaitEvent(ctx->stream, ev, 0)) != hipSuccess) {
                    rc = eagle_fail_hip(ctx, e, "row block event"); break;
                }
                rc = eagle_dev_scan_operands_vrows(ctx, Sa, Va, np, b0, b1, tmp, ctx->stream);
            }
            if (!rc) rc = eagle_dev_scan_operands_finish(ctx, Sa, Va, np, Wu, tmp, ctx->stream);
            if (rc) (void)hipStreamSynchronize(ctx->load_stream);  // nothing of the caller's V may still be in flight when we return
            for (hipEvent_t ev : landed) (void)hipEventDestroy(ev);
            }
        }
    }
    if (!rc && s_from_cache) {
        // under the scan: the caller's S to the scratch copy (loader stream), compared with the cached one.  One resident block
        // on one device: nothing below needs the host before the results go back, so the answer is collected only then (S's 800 MB
        // at n = 10,000 then hide under the whole scan, not under W alone); otherwise (marker blocks, several devices: the host is
        // in the loop anyway) right here.
        // Round 4: a copy out of PAGEABLE memory blocks the calling thread until the runtime has staged it (14 ms for 800 MB), and since
        // the int8 W ends with a host read of its statistics nothing was queued behind it any more: the card sat idle for those 14 ms
        // (rocprofv3 timeline of the call, tools/upload_probe.py).  The deferred verification is therefore issued from a helper thread
        // while this one goes on queueing the genotype pass and the vara kernel.
        int* flag = (int*)((char*)ctx->d_scratch + EAGLE_SCR_SCACHE_FLAG);
        if (!ctx->h_flag && (e = hipHostMalloc((void**)&ctx->h_flag, 64, hipHostMallocDefault)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "pinned flag");
        auto verify = [ctx, flag, inv_MMt_sqrt, n, np]() -> int {
            hipError_t ev = hipSetDevice(ctx->device);
            if (ev != hipSuccess) return eagle_fail_hip(ctx, ev, "hipSetDevice");
            *ctx->h_flag = 0;
            ev = hipMemsetAsync(flag, 0, sizeof(int), ctx->load_stream);
            if (ev != hipSuccess) return eagle_fail_hip(ctx, ev, "verification of the cached S");
            int r = upload_square_on(ctx, inv_MMt_sqrt, n, np, ctx->d_Sscr, ctx->load_stream);
            if (r) return r;
            hipLaunchKernelGGL(k_bits_differ, dim3(1024), dim3(256), 0, ctx->load_stream, (const unsigned long long*)ctx->d_Sscr,
                               (const unsigned long long*)ctx->d_Scache, np * np, flag);
            ev = hipMemcpyAsync(ctx->h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->load_stream);
            if (ev != hipSuccess) return eagle_fail_hip(ctx, ev, "verification of the cached S");
            return EAGLE_OK;
        };
        if (!rc) {
            s_check_pending = true;
            if (host_verify && ctx->h_Scache && ctx->h_Scache_n == n) {
                s_check_kind = 2;
                *host_flag = 0;
                s_verify.start([ctx, inv_MMt_sqrt, n, host_flag]() -> int {
                    const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
                    std::atomic<int> differs{0};
                    parallel_for((long)bytes, std::max(1, std::min(host_threads(), 16)), [&](long a, long b, int) {
                        if (memcmp((const char*)ctx->h_Scache + a, (const char*)inv_MMt_sqrt + a, (size_t)(b - a)) != 0) differs.store(1);
                    });
                    *host_flag = differs.load();
                    return EAGLE_OK;
                });
            } else {
                s_check_kind = 1;
                if (streamed || bounds_flow || rv) rc = verify();
                else s_verify.start(verify);
            }
        }
        // (rv: a device of a multi-device call must not defer -- a peer that streams or holds no cached S settles a changed S inline,
        // and a deferred restart of this one would pass every rendezvous of the call a second time)
        if (s_check_pending && ((s_check_kind == 1 && (streamed || bounds_flow || rv)) || rc)) {
            s_check_pending = false;
            (void)s_verify.join();
            e = hipStreamSynchronize(ctx->load_stream);
            if (e != hipSuccess && !rc) rc = eagle_fail_hip(ctx, e, "verification of the cached S");
            if (!rc && s_check_kind == 1 && *ctx->h_flag) {  // another S: it is already on the device -- start the product over with it
                if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = eagle_fail_hip(ctx, e, "scan operands");
                std::swap(ctx->d_Scache, ctx->d_Sscr);
                Sa = ctx->d_Scache;
                ctx->h_Scache_n = 0;
                if (!rc) rc = eagle_dev_scan_operands(ctx, Sa, Va, ah, n, np, v, Wu, tmp, ctx->stream);
                ctx->scache_misses++;
            } else if (!rc) ctx->scache_hits++;
        }
    }
    if (!rc) ph.mark(ctx->stream, PH_W);
    if (!rc && streamed && !quiet) say(ctx, " Mt.ascii streamed through HBM in blocks of %ld markers", Lc);
    // one pass per marker block: the whole shard when it is resident, else chunks read back from the file
    for (long r0 = 0; r0 < Lr && !rc; r0 += Lc) {
        const long nr = std::min(Lc, Lr - r0), nrp = streamed ? Lc : eagle_pad(nr);
        const int8_t* Mt8 = streamed ? nullptr : g->dev;
        const long ldm = streamed ? np : g->ld;
        if (streamed) {
            int8_t* tile = nullptr;
            rc = ring.load(ctx, f_name_ascii, m0 + r0, nr, 0, n, np, (size_t)nrp * np, max_memory_in_Gbytes, host_threads(), &tile);
            if (rc) break;
            Mt8 = tile;
            ph.mark(ctx->stream, PH_LOADWAIT);
        }
        if (use_i8) {
            // re-centred image of the markers (kept with a resident file, rebuilt per chunk when streaming)
            const int8_t* Ms = nullptr;
            const int8_t* cv = nullptr;
            const int32_t* l1 = nullptr;
            if (streamed) {
                const int b = (int)(ring.k & 1);
                rc = eagle_dev_marker_shift(ctx, Mt8, nrp, n, np, ldm, shifted[b], cs[b], l1s[b], ctx->stream);
                if (rc) break;
                Ms = shifted[b]; cv = cs[b]; l1 = l1s[b];
            } else {
                if (!g->dev_s) {
                    e = hipMalloc((void**)&g->dev_s, (size_t)g->rows_pad * g->ld);
                    if (e != hipSuccess && eagle_drop_f4_images(ctx) > 0) { (void)hipGetLastError(); e = hipMalloc((void**)&g->dev_s, (size_t)g->rows_pad * g->ld); }
                    if (e == hipSuccess) e = hipMalloc((void**)&g->cshift, (size_t)g->rows_pad);
                    if (e == hipSuccess) e = hipMalloc((void**)&g->l1, 2 * sizeof(int32_t) * (size_t)g->rows_pad);
                    rc = e == hipSuccess ? eagle_dev_marker_shift(ctx, g->dev, g->rows_pad, n, np, g->ld, g->dev_s, g->cshift, g->l1, ctx->stream)
                                         : eagle_fail_hip(ctx, e, "re-centred image hipMalloc");
                    if (rc) {  // never leave a half-made image behind: the next call would scan garbage
                        if (g->dev_s) (void)hipFree(g->dev_s);
                        if (g->cshift) (void)hipFree(g->cshift);
                        if (g->l1) (void)hipFree(g->l1);
                        g->dev_s = nullptr; g->cshift = nullptr; g->l1 = nullptr;
                        break;
                    }
                }
                Ms = g->dev_s; cv = g->cshift; l1 = g->l1;
            }
            // one pass over the genotypes gives a = Mt v and the diagonal term of vara; then the int8 MFMA kernel
            // (the W-dependent part -- digits of W, rho -- once per scan; per block of a streamed file only the genotype pass)
            rc = eagle_dev_vara_i8_prepare_part(ctx, Mt8, nrp, np, ldm, Wu, nslices, ws, v, ctx->d_a + r0, ctx->stream, !streamed ? 0 : (r0 == 0 ? 0 : 2));
            if (rc) break;
            ph.mark(ctx->stream, PH_PREPARE);
            rc = eagle_dev_vara_i8_mfma_shifted(ctx, Ms, cv, nrp, np, ldm, nslices, ws, ctx->d_vara + r0, nullptr, ctx->stream);
            if (rc) break;
            // markers that fail their budget under the spectral bound get the dropped digit back (a per-marker decision: the same
            // whatever the blocking; dropped on the device when nobody qualifies)
            rc = eagle_dev_vara_i8_extend(ctx, Ms, cv, l1, nr, nrp, np, ldm, nslices, ws, ctx->d_vara + r0, ctx->stream);
            if (rc) break;
            ph.mark(ctx->stream, PH_VARA);
            // a-posteriori certificate: markers the digit bounds cannot settle are re-evaluated by the fp64 kernel, so that
            // which(tsq == max(tsq))[1] on the returned arrays is the marker the fp64 scan selects (find_qtl.R:71-83).  One resident
            // block on one device: bound, selection and re-evaluation right here.  Marker blocks / device shards: only the
            // per-marker bounds now; selection against the global lower bound after the last block (below).
            if (bounds_flow) {
                rc = eagle_dev_cert_bounds(ctx, nr, nrp, np, cv, l1, nslices, ws, ctx->d_vara + r0, ctx->d_bound + r0, ctx->stream);
            } else {
                rc = eagle_dev_scan_certify(ctx, Mt8, nr, nrp, np, ldm, cv, l1, nslices, ws, Wu, ctx->d_a + r0, ctx->d_vara + r0, cert, ctx->stream);
                if (!rc) rc = eagle_dev_cert_accumulate(ctx, cert, cert_totals, ctx->stream);
            }
        } else {
            rc = eagle_dev_gemv_i8(ctx, Mt8, nrp, np, ldm, v, 1.0, ctx->d_a + r0, ctx->stream);
            if (rc) break;
            ph.mark(ctx->stream, PH_PREPARE);
            rc = eagle_dev_vara_f64(ctx, Mt8, nrp, np, ldm, Wu, ctx->d_vara + r0, ctx->stream);
            if (!rc) ph.mark(ctx->stream, PH_VARA);
        }
        if (rc) break;
        if (use_i8) ph.mark(ctx->stream, PH_CERT);
        if (streamed && (rc = ring.computed(ctx))) break;
    }
    if (streamed && !rc) ring.finish(ctx);
    if (bounds_flow) {
        // this shard's lower bound of the maximum tsq (0 for an empty shard) and its markers over the tight threshold: the maximum / the
        // sum over the devices decide for all of them
        eagle_cert_info mine = {};
        if (!rc && Lr > 0) {
            rc = eagle_dev_cert_lb_b(ctx, Lr, ctx->d_a, ctx->d_vara, ctx->d_bound, cert, ws, ctx->stream);
            if (!rc) {
                e = hipMemcpyAsync(&mine, cert, sizeof mine, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "lower bound of the shard's maximum");
            }
        }
        const double lb = mine.lower_bound;
        EAGLE_ARRIVE2(lb, (long)mine.over_tight);
        const double glb = rv ? rv->vmax : lb;
        over_tight_all = rv ? rv->vsum : (long)mine.over_tight;
        if (Lr > 0) {
            eagle_cert_info hd;
            rc = eagle_dev_cert_select_b(ctx, Lr, ctx->d_a, ctx->d_vara, ctx->d_bound, cert, glb, over_tight_all, ws, ctx->stream);
            if (!rc) {
                e = hipMemcpyAsync(&hd, cert, sizeof hd, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "candidate count");
            }
            if (!rc && hd.overflow) {
                // more candidates than the re-evaluation buffer holds (degenerate operands): all of the shard in fp64
                // (on the fp64 products, should W have come from the int8 engine)
                rc = eagle_w8_redo_f64(ctx, np, ctx->stream);
                if (rc) {
                } else if (!streamed) {
                    rc = eagle_dev_vara_f64(ctx, g->dev, Lp, np, g->ld, Wu, ctx->d_vara, ctx->stream);
                } else {
                    ChunkRing again;
                    if (!(rc = again.init(ctx))) {
                        again.buf[0] = ring.buf[0]; again.buf[1] = ring.buf[1];
                        for (long r0 = 0; r0 < Lr && !rc; r0 += Lc) {
                            const long nr = std::min(Lc, Lr - r0), nrp = eagle_pad(nr);
                            int8_t* tile = nullptr;
                            rc = again.load(ctx, f_name_ascii, m0 + r0, nr, 0, n, np, (size_t)nrp * np, max_memory_in_Gbytes, host_threads(), &tile);
                            if (!rc) rc = eagle_dev_vara_f64(ctx, tile, nrp, np, np, Wu, ctx->d_vara + r0, ctx->stream);
                            if (!rc) rc = again.computed(ctx);
                        }
                        if (!rc) again.finish(ctx);
                    }
                }
            } else if (!rc && hd.reevaluated > 0) {
                if (!streamed) {
                    rc = eagle_dev_cert_reevaluate(ctx, g->dev, g->ld, np, Wu, ctx->d_vara, cert, ctx->stream);
                } else {
                    // only the candidates' rows are read back from the file (its 2-bit sidecar when there is one): typically one or two
                    const long cnt = hd.reevaluated;
                    std::vector<long> idx((size_t)cnt);
                    int8_t* rows = eagle_cert_rows(cert);
                    e = hipMemcpyAsync(idx.data(), eagle_cert_indices(cert), sizeof(long) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess) e = hipMemsetAsync(rows, 0, (size_t)((cnt + 127) / 128 * 128) * (size_t)np, ctx->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess) rc = eagle_fail_hip(ctx, e, "candidate list");
                    for (long c = 0; c < cnt && !rc; c++)
                        rc = eagle_dev_load_ascii(ctx, f_name_ascii, m0 + idx[(size_t)c], 1, 0, n, rows + c * np, np, max_memory_in_Gbytes, 1);
                    if (!rc) rc = eagle_dev_cert_reevaluate(ctx, nullptr, np, np, Wu, ctx->d_vara, cert, ctx->stream);
                }
            }
            if (!rc) rc = eagle_dev_cert_accumulate(ctx, cert, cert_totals, ctx->stream);
            if (!rc) ph.mark(ctx->stream, PH_CERT);
        }
    }
#undef EAGLE_ARRIVE
#undef EAGLE_ARRIVE2
    if (s_check_pending) {   // the deferred outcome of the verification of the cached S (nothing of the caller's S may be in flight past here)
        const int rs = s_verify.join();
        if (rs && !rc) rc = rs;
        e = hipStreamSynchronize(ctx->load_stream);
        if (e != hipSuccess && !rc) rc = eagle_fail_hip(ctx, e, "verification of the cached S");
        if (!rc && s_check_kind == 1 && *ctx->h_flag) {
            // another S than the cached one: this scan ran on the wrong operand.  The caller's S is on the device already (the scratch
            // copy): it becomes the cached one and the scan starts over, once, without another verification.
            if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return eagle_fail_hip(ctx, e, "scan on a stale S");
            std::swap(ctx->d_Scache, ctx->d_Sscr);
            ctx->h_Scache_n = 0;
            ctx->scache_misses++;
            return scan_range(ctx, f_name_ascii, L, n, m0, m1, sel, inv_MMt_sqrt, dim_reduced_vara, a, max_memory_in_Gbytes, quiet, a_out, vara_out, k, nd,
                              rv, rccl, w_direct, true);
        } else if (!rc && s_check_kind == 2 && *host_flag) {
            // (host comparison) the caller's S is not on the device: the cache is dropped and the scan starts over with an upload
            if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return eagle_fail_hip(ctx, e, "scan on a stale S");
            ctx->scache_n = 0;
            ctx->h_Scache_n = 0;
            ctx->scache_misses++;
            return scan_range(ctx, f_name_ascii, L, n, m0, m1, sel, inv_MMt_sqrt, dim_reduced_vara, a, max_memory_in_Gbytes, quiet, a_out, vara_out, k, nd,
                              rv, rccl, w_direct, false);
        } else if (!rc && s_check_kind != 3) ctx->scache_hits++;
    }
    if (timing_on() && !rc) {
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[eaglehip] scan n=%ld markers [%ld, %ld) of %ld on device %d%s: alloc+upload %.1f ms, device compute %.1f ms\n", n, m0, m1, L,
                ctx->device, streamed ? " (streamed)" : "", (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    }
    if (rc) return rc;
    std::vector<long> in_range;
    for (long r : sel) if (r >= m0 && r < m1) in_range.push_back(r - m0);
    if (!in_range.empty()) {  // :79-84: a zeroed marker row gives a = 0 and vara = 0 exactly
        HIPCHK(ctx, dsel.alloc(sizeof(long) * in_range.size()));
        HIPCHK(ctx, hipMemcpyAsync(dsel.p, in_range.data(), sizeof(long) * in_range.size(), hipMemcpyHostToDevice, ctx->stream));
        rc = eagle_dev_zero_rows(ctx, ctx->d_a, ctx->d_vara, Lr, dsel.as<long>(), (long)in_range.size(), 0, ctx->stream);
        if (rc) return rc;
    }
    ctx->scan_L = Lr;
    ctx->scan_first = m0;
    long totals[4] = {0, 0, 0, 0};
    if (Lr > 0) {
        HIPCHK(ctx, hipMemcpyAsync(a_out + m0, ctx->d_a, sizeof(double) * (size_t)Lr, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(vara_out + m0, ctx->d_vara, sizeof(double) * (size_t)Lr, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(totals, cert_totals, sizeof totals, hipMemcpyDeviceToHost, ctx->stream));
    struct { double maxabs_off; int S, pad; double bound, sumdiag, R, specH; int S_sliced, pad2; double budget; int e, pad3; unsigned long long lo_sumsq;
             int maxdiag, hi_overflow, spec_try2, level; double wErr, specH1, budget_loose; } vh = {};   // head of the digit workspace (VaraHdr)
    if (use_i8 && Lr > 0) HIPCHK(ctx, hipMemcpyAsync(&vh, ws, sizeof vh, hipMemcpyDeviceToHost, ctx->stream));
    ph.mark(ctx->stream, PH_D2H);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cert_reevaluated = totals[0]; ctx->cert_flagged = totals[1]; ctx->cert_fell_back = totals[2] != 0;
    ctx->scan_digits_used = vh.S; ctx->scan_digits_cut = vh.S_sliced; ctx->scan_specH = vh.specH;
    ctx->scan_budget_used = vh.budget; ctx->scan_bound_level = vh.level; ctx->scan_w_err = vh.wErr;
    // (one resident block counted on the device; blocks / shards: the sum the devices exchanged)
    ctx->cert_over_tight = bounds_flow ? over_tight_all : totals[3];
    ctx->scan_budget_enforced = ctx->cert_over_tight > eagle_cert_tight_max() ? vh.budget_loose : vh.budget;
    // a scan that took a digit off under the spectral bound and then had to redo a block in fp64: markers of this data set sit
    // outside what the bound covers -- this context keeps the worst-case digit count from now on (eagle_set_scan_budget re-arms)
    if (ctx->cert_fell_back && vh.specH > 0.0) ctx->spectral_off = true;
    ph.sum(ctx->scan_phase_ms);
    ctx->scan_blocks = streamed ? ring.k : 1;
    ctx->scan_host_setup_s = t_setup - t0;
    ctx->scan_range_wall_s = now_s() - t0;
    if (!w_direct && Lr > 0 && Sa && Sa != ctx->d_Scache) {   // S sits in its arena slot, verified or uploaded by this call: the next scan may find it there
        ctx->arena_S_ptr = Sa; ctx->arena_S_base = ctx->arena; ctx->arena_S_n = n; ctx->arena_S_np = np;
    }
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
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    Rendezvous rv;
    rv.n = nd;
    RcclState* rccl = (RcclState*)ctx->rccl;
    if (rccl && rccl->dead) rccl = nullptr;   // aborted in an earlier call: W on every device, host-staged exchange
    const double t_call = now_s();
    rc = run_on_devices(ctx, [&](int k, eagle_ctx* c) -> int {
        return scan_range(c, f_name_ascii, L, n, edge[k], edge[k + 1], sel, inv_MMt_sqrt, dim_reduced_vara, a, max_memory_in_Gbytes, quiet,
                          a_out, vara_out, k, nd, nd > 1 || rccl ? &rv : nullptr, rccl);
    });
    ctx->scan_call_wall_s = now_s() - t_call;
    if (rc) return rc;
    for (eagle_ctx* p : ctx->peers) {
        ctx->cert_reevaluated += p->cert_reevaluated; ctx->cert_flagged += p->cert_flagged; ctx->cert_fell_back |= p->cert_fell_back;
        ctx->spectral_off |= p->spectral_off;
        if (!ctx->scan_digits_cut) { ctx->scan_digits_used = p->scan_digits_used; ctx->scan_digits_cut = p->scan_digits_cut; ctx->scan_specH = p->scan_specH; }
    }
    for (eagle_ctx* p : ctx->peers) p->spectral_off = ctx->spectral_off;   // every device of the next scan decides alike
    return EAGLE_OK;
}

// Optional shortcut for the R side (not one of the reference's .Call symbols): the same scan with W = S V S and v = S a_hat
// handed over ready-made.  Inside AM() they need no n^3 product at all: dim_reduced_vara is Henderson's varG I - C22 =
// varG^2 Ze P Ze with Ze = MMt^1/2 (E/R/calculate_reduced_vara.R:21-35) and inv_MMt_sqrt = Ze^-1, so W = varG^2 P and
// v = S (varG Ze P y) = varG P y (E/R/calculate_reduced_a.R:31) -- and .find_qtl holds P (find_qtl.R:9).  The reference-shaped
// entry point cannot know that (it receives opaque matrices) and spends 24 % of a scan at n = 10,000 on S (V S).
extern "C" int eagle_scan_with_W(eagle_ctx* ctx, const char* f_name_ascii, const double* selected_loci, long n_selected, const double* W,
                                 const double* v, double max_memory_in_Gbytes, const long dims[2], int quiet, double* a_out, double* vara_out) {
    if (!ctx) return EAGLE_ERR_ARG;
    const long L = dims[0], n = dims[1];
    if (n <= 0 || L <= 0 || !W || !v) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims / null operand");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<long> sel;
    int rc = parse_selected(ctx, selected_loci, n_selected, L, sel);
    if (rc) return rc;
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    Rendezvous rv;
    rv.n = nd;
    rc = run_on_devices(ctx, [&](int k, eagle_ctx* c) -> int {
        return scan_range(c, f_name_ascii, L, n, edge[k], edge[k + 1], sel, W, nullptr, v, max_memory_in_Gbytes, quiet, a_out, vara_out, k, nd,
                          nd > 1 ? &rv : nullptr, nullptr, true);
    });
    if (rc) return rc;
    for (eagle_ctx* p : ctx->peers) {
        ctx->cert_reevaluated += p->cert_reevaluated; ctx->cert_flagged += p->cert_flagged; ctx->cert_fell_back |= p->cert_fell_back;
        ctx->spectral_off |= p->spectral_off;
        if (!ctx->scan_digits_cut) { ctx->scan_digits_used = p->scan_digits_used; ctx->scan_digits_cut = p->scan_digits_cut; ctx->scan_specH = p->scan_specH; }
    }
    for (eagle_ctx* p : ctx->peers) p->spectral_off = ctx->spectral_off;   // every device of the next scan decides alike
    return EAGLE_OK;
}

extern "C" int eagle_scan_operand_cache_stats(eagle_ctx* ctx, long* hits, long* misses) {
    if (!ctx) return EAGLE_ERR_ARG;
    long h = ctx->scache_hits, m = ctx->scache_misses;
    for (eagle_ctx* p : ctx->peers) { h += p->scache_hits; m += p->scache_misses; }
    if (hits) *hits = h;
    if (misses) *misses = m;
    return EAGLE_OK;
}
extern "C" int eagle_last_stream_stats(eagle_ctx* ctx, eagle_stream_stats* out) {
    if (!ctx || !out) return EAGLE_ERR_ARG;
    out->chunks = ctx->st_chunks; out->file_bytes = ctx->st_file_bytes;
    out->pread_s = ctx->st_pread_s; out->load_s = ctx->st_load_wall_s; out->wait_s = ctx->st_wait_s;
    out->kernel_s = ctx->st_compute_s; out->wall_s = ctx->st_total_s;
    out->load_first_s = ctx->st_load_first_s; out->starved_s = ctx->st_starved_s;
    return EAGLE_OK;
}
// Where the last eagle_calculate_a_and_vara / eagle_scan_with_W call spent its time on device `device_index` of the context
// (0 = the lead): host seconds until the operand uploads were enqueued, then the phases on that device's compute stream.
extern "C" int eagle_last_scan_timing(eagle_ctx* ctx, int device_index, eagle_scan_timing* out) {
    if (!ctx || !out || device_index < 0 || device_index >= ndev_of(ctx)) return EAGLE_ERR_ARG;
    const eagle_ctx* c = dev_ctx(ctx, device_index);
    out->call_wall_s = ctx->scan_call_wall_s;
    out->device_wall_s = c->scan_range_wall_s;
    out->host_setup_s = c->scan_host_setup_s;
    out->upload_ms = c->scan_phase_ms[PH_UPLOAD];
    out->w_ms = c->scan_phase_ms[PH_W];
    out->load_wait_ms = c->scan_phase_ms[PH_LOADWAIT];
    out->prepare_ms = c->scan_phase_ms[PH_PREPARE];
    out->vara_ms = c->scan_phase_ms[PH_VARA];
    out->certify_ms = c->scan_phase_ms[PH_CERT];
    out->d2h_ms = c->scan_phase_ms[PH_D2H];
    out->blocks = c->scan_blocks;
    out->markers = c->scan_L;
    return EAGLE_OK;
}
extern "C" int eagle_last_scan_certificate(eagle_ctx* ctx, long* n_reevaluated, long* n_flagged, int* fell_back) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (n_reevaluated) *n_reevaluated = ctx->cert_reevaluated;
    if (n_flagged) *n_flagged = ctx->cert_flagged;
    if (fell_back) *fell_back = ctx->cert_fell_back;
    return EAGLE_OK;
}

extern "C" int eagle_last_scan_budget(eagle_ctx* ctx, double* budget_used, int* bound_level, double* w_error_bound) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (budget_used) *budget_used = ctx->scan_budget_used;
    if (bound_level) *bound_level = ctx->scan_bound_level;
    if (w_error_bound) *w_error_bound = ctx->scan_w_err;
    return EAGLE_OK;
}
extern "C" int eagle_last_scan_enforced(eagle_ctx* ctx, double* budget_enforced, long* n_over_tight) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (budget_enforced) *budget_enforced = ctx->scan_budget_enforced;
    if (n_over_tight) *n_over_tight = ctx->cert_over_tight;
    return EAGLE_OK;
}
extern "C" int eagle_set_w_mode(eagle_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return EAGLE_ERR_ARG;
    ctx->w_mode = mode;
    for (eagle_ctx* p : ctx->peers) p->w_mode = mode;
    return EAGLE_OK;
}
extern "C" int eagle_last_w_info(eagle_ctx* ctx, eagle_w_info* out) {
    if (!ctx || !out) return EAGLE_ERR_ARG;
    const W8Info& i = ctx->w8_info;
    out->int8 = ctx->w8_active ? 1 : 0;
    out->declined = i.declined;
    out->k1 = i.k1; out->T1 = i.T1; out->pairs1 = i.pairs1;
    out->k2 = i.k2; out->T2 = i.T2; out->pairs2 = i.pairs2;
    out->eta = i.eta; out->eta_x = i.eta_x; out->target = i.target; out->mean_diag = i.mean_diag; out->asym_term = i.asym_term;
    out->pipelined = i.pipelined ? 1 : 0; out->pad = 0;
    return EAGLE_OK;
}
extern "C" int eagle_last_scan_digits(eagle_ctx* ctx, int* digits_used, int* digits_cut, double* spectral_bound) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (digits_used) *digits_used = ctx->scan_digits_used;
    if (digits_cut) *digits_cut = ctx->scan_digits_cut;
    if (spectral_bound) *spectral_bound = ctx->scan_specH;
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
    for (int k = 0; k < ndev_of(ctx); k++) {  // whichever device holds the marker's column of M.ascii (a shard is a column window)
        eagle_ctx* c = dev_ctx(ctx, k);
        for (auto& g : c->cache)
            if (g.path == f_name_ascii && g.size == st.st_size && g.mtime_ns == mt && g.row0 == 0 && g.rows == n && selected_locus >= g.col0 &&
                selected_locus < g.col0 + g.cols) {
                HIPCHK(ctx, hipSetDevice(c->device));
                DevBuf col;
                HIPCHK(ctx, col.alloc(sizeof(int) * (size_t)n));
                int rc = eagle_dev_extract_col(c, g.dev, n, g.ld, selected_locus - g.col0, col.as<int>(), c->stream);
                if (rc) return rc;
                HIPCHK(ctx, hipMemcpyAsync(column_out, col.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(ctx, hipStreamSynchronize(c->stream));
                (void)hipSetDevice(ctx->device);
                return EAGLE_OK;
            }
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

static int local_scan_argmax(eagle_ctx* ctx, eagle_best* h) {
    h->tsqmax = 0.0; h->index0 = -1; h->near_ties = 0;
    if (!ctx->d_a || ctx->scan_L <= 0) return EAGLE_OK;  // an empty shard
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // ctx-owned workspace: a hipMalloc / hipFree pair per call costs milliseconds in a process that has 100+ GB mapped (measured:
    // 13 ms per find_qtl-shaped call inside bench.py), and hipFree synchronises the device
    if (!ctx->argmax_ws) HIPCHK(ctx, hipMalloc(&ctx->argmax_ws, sizeof(double) * 3 * 1024 + 256));
    double* scratch = (double*)ctx->argmax_ws;
    eagle_best* best = (eagle_best*)((char*)ctx->argmax_ws + sizeof(double) * 3 * 1024);
    int rc = eagle_dev_tsq_argmax(ctx, ctx->d_a, ctx->d_vara, ctx->scan_L, nullptr, best, scratch, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(h, best, sizeof *h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (h->index0 >= 0) h->index0 += ctx->scan_first;  // global marker index
    return EAGLE_OK;
}
extern "C" int eagle_last_scan_argmax(eagle_ctx* ctx, long* index_out, double* tsqmax_out, long* n_near_ties) {
    if (!ctx || !ctx->d_a || (ctx->scan_L <= 0 && ctx->peers.empty())) return eagle_fail(ctx, EAGLE_ERR_ARG, "no scan result held");
    // per-device (max tsq, first index) of the shards, merged on the host: the largest tsq, ties -> the smallest global
    // index = which(tsq == max)[1] (find_qtl.R:76-80)
    eagle_best h;
    int rc = local_scan_argmax(ctx, &h);
    if (rc) return rc;
    for (eagle_ctx* p : ctx->peers) {
        eagle_best o;
        if ((rc = local_scan_argmax(p, &o))) { snprintf(ctx->err, sizeof ctx->err, "%s", p->err); return rc; }
        if (o.index0 < 0) continue;
        const double thr = h.index0 >= 0 ? std::max(h.tsqmax, o.tsqmax) * (1.0 - 1e-9) : 0.0;
        const long near = (h.index0 >= 0 && h.tsqmax >= thr ? h.near_ties : 0) + (o.tsqmax >= thr ? o.near_ties : 0);
        if (h.index0 < 0 || o.tsqmax > h.tsqmax || (o.tsqmax == h.tsqmax && o.index0 < h.index0)) { h.tsqmax = o.tsqmax; h.index0 = o.index0; }
        h.near_ties = near;
    }
    (void)hipSetDevice(ctx->device);
    if (h.index0 < 0) h.tsqmax = __builtin_nan("");
    if (index_out) *index_out = h.index0 + 1;  // R is 1-based; 0 = every tsq was NaN
    if (tsqmax_out) *tsqmax_out = h.tsqmax;
    if (n_near_ties) *n_near_ties = h.near_ties;
    return EAGLE_OK;
}

// ar[m0..m1) = varG * Mt[m0..m1) (P y) on ctx's device (calculate_reduced_a_rcpp.cpp:82-84)
static int reduced_a_range(eagle_ctx* ctx, const char* f_name_ascii, long n, long L, long m0, long m1, const std::vector<long>& sel, double varG,
                           const double* P, const double* y, double max_memory_in_Gbytes, double* ar_out) {
    const long Lr = m1 - m0;
    if (Lr <= 0) return EAGLE_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long np = eagle_pad(n), Lp = eagle_pad(Lr);
    GenoEntry* g = nullptr;
    int rc = get_resident(ctx, f_name_ascii, m0, Lr, 0, n, max_memory_in_Gbytes, host_threads(), &g,
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
    for (long r0 = 0; r0 < Lr; r0 += Lc) {  // :83-84, one pass per marker block (the whole shard when resident)
        const long nr = std::min(Lc, Lr - r0), nrp = eagle_pad(nr);
        int8_t* tile = nullptr;
        if (streamed) {
            rc = ring.load(ctx, f_name_ascii, m0 + r0, nr, 0, n, np, (size_t)nrp * np, max_memory_in_Gbytes, host_threads(), &tile);
            if (rc) return rc;
        }
        rc = eagle_dev_gemv_i8(ctx, streamed ? tile : g->dev, nrp, np, streamed ? np : g->ld, py.as<double>(), varG,
                               out.as<double>() + r0, ctx->stream);
        if (rc) return rc;
        if (streamed && (rc = ring.computed(ctx))) return rc;
    }
    if (streamed) ring.finish(ctx);
    std::vector<long> in_range;
    for (long r : sel) if (r >= m0 && r < m1) in_range.push_back(r - m0);
    if (!in_range.empty()) {  // :74-78
        HIPCHK(ctx, dsel.alloc(sizeof(long) * in_range.size()));
        HIPCHK(ctx, hipMemcpyAsync(dsel.p, in_range.data(), sizeof(long) * in_range.size(), hipMemcpyHostToDevice, ctx->stream));
        rc = eagle_dev_zero_rows(ctx, out.as<double>(), nullptr, Lr, dsel.as<long>(), (long)in_range.size(), 0, ctx->stream);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(ar_out + m0, out.p, sizeof(double) * (size_t)Lr, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    return run_on_devices(ctx, [&](int k, eagle_ctx* c) -> int {
        return reduced_a_range(c, f_name_ascii, n, L, edge[k], edge[k + 1], sel, varG, P, y, max_memory_in_Gbytes, ar_out);
    });
}

// ------------------------------------------------------------------------------------------------
// The scan in the eigenbasis of MM^T (include/eagle_hip.h section 1d; kernels and per-device work in eagle_spectral.hip): the
// markers -- and with them Z = Mt U -- split over the devices of the context like everywhere else; no exchange step at all.
// ------------------------------------------------------------------------------------------------
extern "C" int eagle_spectral_prepare(eagle_ctx* ctx, const char* f_name_ascii, const long dims[2], const double* U, double max_memory_in_Gbytes) {
    if (!ctx || !f_name_ascii || !U) return EAGLE_ERR_ARG;
    const long L = dims[0], n = dims[1];
    if (L <= 0 || n <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "bad dims");
    const int nd = ndev_of(ctx);
    std::vector<long> edge;
    split_markers(L, nd, edge);
    int rc = run_on_devices(ctx, [&](int k, eagle_ctx* c) -> int {
        return eagle_spectral_prepare_range(c, f_name_ascii, L, n, edge[k], edge[k + 1], U, max_memory_in_Gbytes);
    });
    ctx->spectral_L = rc ? 0 : L;
    return rc;
}
extern "C" int eagle_spectral_scan(eagle_ctx* ctx, const double* lambda, const double* UtX, const double* Uty, long p, double varE, double varG,
                                   const double* selected_loci, long n_selected, double* a_out, double* vara_out) {
    if (!ctx || !lambda || !UtX || !Uty || !a_out || !vara_out) return EAGLE_ERR_ARG;
    if (ctx->spectral_L <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_scan: eagle_spectral_prepare has not run");
    if (p < 1 || p > 31) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_scan: 1 <= p <= 31 fixed-effect columns");
    const long L = ctx->spectral_L, n = ctx->z_n, np = eagle_pad(n);
    std::vector<long> sel;
    int rc = parse_selected(ctx, selected_loci, n_selected, L, sel);
    if (rc) return rc;
    const int NC = p + 1 <= 16 ? 16 : 32;
    std::vector<double> d(np), G((size_t)np * NC), Cm((size_t)p * p), c1(p);
    if ((rc = eagle_spectral_host_operands(ctx, n, lambda, UtX, Uty, p, varE, varG, NC, d.data(), G.data(), Cm.data(), c1.data()))) return rc;
    return run_on_devices(ctx, [&](int, eagle_ctx* c) -> int {
        return eagle_spectral_scan_range(c, d.data(), G.data(), NC, Cm.data(), c1.data(), p, varG, sel.data(), (long)sel.size(), a_out, vara_out);
    });
}
