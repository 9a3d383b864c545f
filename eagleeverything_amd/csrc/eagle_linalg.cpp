// eagle_linalg.cpp -- SURVEY 8 f-4 behind the C ABI: the dense n x n model algebra of find_qtl / emma.* on the device, for a
// caller that wants it there (the author's own note, MyPackage/MyREADME:1, names eigen() as the bottleneck; his MAGMA attempt
// is E/R/emma_eigen_R_wo_Z.R:9-15).  north_star keeps this algebra on host LAPACK by default; these entry points are the
// opt-in.  Host column-major matrices in and out, like section 1 of include/eagle_hip.h.
//
//   eagle_sym_eig ................. eigen(A, symmetric = TRUE)                 rocSOLVER dsyevd
//   eagle_chol2inv ................ chol2inv(chol(A))                          rocSOLVER dpotrf + dpotri
//   eagle_inverse ................. solve(A)                                   rocSOLVER dgetrf + dgetri
//   eagle_matmul .................. A %*% B                                    this library's fp64 MFMA GEMM (k_gemm_f64_list)
//   eagle_mmt_sqrt_and_sqrtinv .... E/R/calculateMMt_sqrt_and_sqrtinv.R:15-47  the four above, data staying in HBM
//
// rocSOLVER / rocBLAS are LIBRARY calls (fine for this row: it is not the hot path and has no kernel of this repository to
// offer); they are dlopen()ed on first use so that libeaglehip.so itself only needs the HIP runtime.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "eagle_ctx.h"

struct SolverApi {
    void* blas = nullptr;
    void* solver = nullptr;
    rocblas_status (*create_handle)(rocblas_handle*) = nullptr;
    rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*dsyevd)(rocblas_handle, rocblas_evect, rocblas_fill, rocblas_int, double*, rocblas_int, double*, double*, rocblas_int*) = nullptr;
    rocblas_status (*dpotrf)(rocblas_handle, rocblas_fill, rocblas_int, double*, rocblas_int, rocblas_int*) = nullptr;
    rocblas_status (*dpotri)(rocblas_handle, rocblas_fill, rocblas_int, double*, rocblas_int, rocblas_int*) = nullptr;
    rocblas_status (*dgetrf)(rocblas_handle, rocblas_int, rocblas_int, double*, rocblas_int, rocblas_int*, rocblas_int*) = nullptr;
    rocblas_status (*dgetri)(rocblas_handle, rocblas_int, double*, rocblas_int, rocblas_int*, rocblas_int*) = nullptr;
};
static SolverApi g_api;
static std::mutex g_api_mutex;
static bool g_api_ok = false;

static int solver_load(eagle_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_api_mutex);
    if (g_api_ok) return EAGLE_OK;
    for (const char* name : {"librocblas.so.5", "librocblas.so.4", "librocblas.so", "/opt/rocm/lib/librocblas.so"})
        if ((g_api.blas = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!g_api.blas) return failf(ctx, EAGLE_ERR_NODEVICE, "cannot load librocblas: %s", dlerror());
    for (const char* name : {"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"})
        if ((g_api.solver = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!g_api.solver) return failf(ctx, EAGLE_ERR_NODEVICE, "cannot load librocsolver: %s", dlerror());
#define EAGLE_SYM(lib, field, name)                                              \
    g_api.field = (decltype(g_api.field))dlsym(g_api.lib, name);                 \
    if (!g_api.field) return failf(ctx, EAGLE_ERR_NODEVICE, "%s not found in the ROCm libraries", name)
    EAGLE_SYM(blas, create_handle, "rocblas_create_handle");
    EAGLE_SYM(blas, destroy_handle, "rocblas_destroy_handle");
    EAGLE_SYM(blas, set_stream, "rocblas_set_stream");
    EAGLE_SYM(solver, dsyevd, "rocsolver_dsyevd");
    EAGLE_SYM(solver, dpotrf, "rocsolver_dpotrf");
    EAGLE_SYM(solver, dpotri, "rocsolver_dpotri");
    EAGLE_SYM(solver, dgetrf, "rocsolver_dgetrf");
    EAGLE_SYM(solver, dgetri, "rocsolver_dgetri");
#undef EAGLE_SYM
    g_api_ok = true;
    return EAGLE_OK;
}

// per-ctx rocBLAS handle bound to the ctx stream
static int solver_handle(eagle_ctx* ctx, rocblas_handle* h) {
    int rc = solver_load(ctx);
    if (rc) return rc;
    if (!ctx->blas_handle) {
        rocblas_handle hh = nullptr;
        if (g_api.create_handle(&hh) != rocblas_status_success) return eagle_fail(ctx, EAGLE_ERR_HIP, "rocblas_create_handle failed");
        if (g_api.set_stream(hh, ctx->stream) != rocblas_status_success) { g_api.destroy_handle(hh); return eagle_fail(ctx, EAGLE_ERR_HIP, "rocblas_set_stream failed"); }
        ctx->blas_handle = hh;
    }
    *h = (rocblas_handle)ctx->blas_handle;
    return EAGLE_OK;
}
extern "C" void eagle_linalg_release(eagle_ctx* ctx) {
    if (ctx && ctx->blas_handle && g_api_ok) { g_api.destroy_handle((rocblas_handle)ctx->blas_handle); ctx->blas_handle = nullptr; }
}

static int check_n(eagle_ctx* ctx, long n) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (n <= 0 || n > 46340) return eagle_fail(ctx, EAGLE_ERR_ARG, "matrix order must be in 1..46340 (32-bit LAPACK interface)");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "hipSetDevice");
    return EAGLE_OK;
}
static int info_of(eagle_ctx* ctx, const rocblas_int* d_info, int* info) {
    HIPCHK(ctx, hipMemcpyAsync(info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

// A (device, n x n, symmetric, destroyed) -> w ascending (device) and, if vectors, the eigenvectors in the columns of A
static int dev_syevd(eagle_ctx* ctx, double* A, long n, double* w, bool vectors) {
    rocblas_handle h;
    int rc = solver_handle(ctx, &h);
    if (rc) return rc;
    DevBuf E, info;
    HIPCHK(ctx, E.alloc(sizeof(double) * n));
    HIPCHK(ctx, info.alloc(sizeof(rocblas_int)));
    // the LOWER triangle of the column-major matrix is read, like R's eigen(symmetric = TRUE) and LAPACK 'L' (numpy.linalg.eigh)
    if (g_api.dsyevd(h, vectors ? rocblas_evect_original : rocblas_evect_none, rocblas_fill_lower, (rocblas_int)n, A, (rocblas_int)n, w, E.as<double>(),
                     info.as<rocblas_int>()) != rocblas_status_success)
        return eagle_fail(ctx, EAGLE_ERR_HIP, "rocsolver_dsyevd failed");
    int hinfo = 0;
    if ((rc = info_of(ctx, info.as<rocblas_int>(), &hinfo))) return rc;
    if (hinfo != 0) return failf(ctx, EAGLE_ERR_ARG, "eigen-decomposition did not converge (%d off-diagonal elements)", hinfo);
    return EAGLE_OK;
}
// A (device, SPD, destroyed) -> A^-1 in BOTH triangles.  Returns EAGLE_SOFT_SENTINEL when A is not positive definite
// (R: "the leading minor of order k is not positive").
static int dev_chol2inv(eagle_ctx* ctx, double* A, long n) {
    rocblas_handle h;
    int rc = solver_handle(ctx, &h);
    if (rc) return rc;
    DevBuf info;
    HIPCHK(ctx, info.alloc(sizeof(rocblas_int)));
    if (g_api.dpotrf(h, rocblas_fill_upper, (rocblas_int)n, A, (rocblas_int)n, info.as<rocblas_int>()) != rocblas_status_success)
        return eagle_fail(ctx, EAGLE_ERR_HIP, "rocsolver_dpotrf failed");
    int hinfo = 0;
    if ((rc = info_of(ctx, info.as<rocblas_int>(), &hinfo))) return rc;
    if (hinfo != 0) { failf(ctx, EAGLE_SOFT_SENTINEL, "the leading minor of order %d is not positive", hinfo); return EAGLE_SOFT_SENTINEL; }
    if (g_api.dpotri(h, rocblas_fill_upper, (rocblas_int)n, A, (rocblas_int)n, info.as<rocblas_int>()) != rocblas_status_success)
        return eagle_fail(ctx, EAGLE_ERR_HIP, "rocsolver_dpotri failed");
    if ((rc = info_of(ctx, info.as<rocblas_int>(), &hinfo))) return rc;
    if (hinfo != 0) return failf(ctx, EAGLE_ERR_ARG, "dpotri: element (%d, %d) of the Cholesky factor is zero, the inverse could not be computed", hinfo, hinfo);
    return eagle_dev_symmetrize(ctx, A, n, n, ctx->stream);  // column-major "upper" = row-major lower: mirror it
}

extern "C" int eagle_sym_eig(eagle_ctx* ctx, const double* A, long n, double* values_out, double* vectors_out) {
    int rc = check_n(ctx, n);
    if (rc) return rc;
    if (!A || !values_out) return eagle_fail(ctx, EAGLE_ERR_ARG, "sym_eig: null argument");
    DevBuf dA, dw;
    HIPCHK(ctx, dA.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(ctx, dw.alloc(sizeof(double) * n));
    HIPCHK(ctx, hipMemcpyAsync(dA.p, A, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = dev_syevd(ctx, dA.as<double>(), n, dw.as<double>(), vectors_out != nullptr))) return rc;
    // R's eigen() returns the values in DEcreasing order (and the vectors to match): reverse rocSOLVER's ascending order
    std::vector<double> w(n);
    HIPCHK(ctx, hipMemcpyAsync(w.data(), dw.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (long i = 0; i < n; i++) values_out[i] = w[n - 1 - i];
    if (vectors_out) {
        for (long j = 0; j < n; j++)  // column j of the result = column n-1-j of the device matrix
            HIPCHK(ctx, hipMemcpyAsync(vectors_out + j * n, dA.as<double>() + (n - 1 - j) * n, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return EAGLE_OK;
}

extern "C" int eagle_chol2inv(eagle_ctx* ctx, const double* A, long n, double* Ainv_out) {
    int rc = check_n(ctx, n);
    if (rc) return rc;
    if (!A || !Ainv_out) return eagle_fail(ctx, EAGLE_ERR_ARG, "chol2inv: null argument");
    DevBuf dA;
    HIPCHK(ctx, dA.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(ctx, hipMemcpyAsync(dA.p, A, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = dev_chol2inv(ctx, dA.as<double>(), n))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(Ainv_out, dA.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

extern "C" int eagle_inverse(eagle_ctx* ctx, const double* A, long n, double* Ainv_out) {
    int rc = check_n(ctx, n);
    if (rc) return rc;
    if (!A || !Ainv_out) return eagle_fail(ctx, EAGLE_ERR_ARG, "inverse: null argument");
    rocblas_handle h;
    if ((rc = solver_handle(ctx, &h))) return rc;
    DevBuf dA, piv, info;
    HIPCHK(ctx, dA.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(ctx, piv.alloc(sizeof(rocblas_int) * n));
    HIPCHK(ctx, info.alloc(sizeof(rocblas_int)));
    HIPCHK(ctx, hipMemcpyAsync(dA.p, A, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
    if (g_api.dgetrf(h, (rocblas_int)n, (rocblas_int)n, dA.as<double>(), (rocblas_int)n, piv.as<rocblas_int>(), info.as<rocblas_int>()) != rocblas_status_success)
        return eagle_fail(ctx, EAGLE_ERR_HIP, "rocsolver_dgetrf failed");
    int hinfo = 0;
    if ((rc = info_of(ctx, info.as<rocblas_int>(), &hinfo))) return rc;
    if (hinfo != 0) { failf(ctx, EAGLE_SOFT_SENTINEL, "Lapack routine dgesv: system is exactly singular: U[%d,%d] = 0", hinfo, hinfo); return EAGLE_SOFT_SENTINEL; }
    if (g_api.dgetri(h, (rocblas_int)n, dA.as<double>(), (rocblas_int)n, piv.as<rocblas_int>(), info.as<rocblas_int>()) != rocblas_status_success)
        return eagle_fail(ctx, EAGLE_ERR_HIP, "rocsolver_dgetri failed");
    HIPCHK(ctx, hipMemcpyAsync(Ainv_out, dA.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

// C (m x n) = A (m x k) %*% B (k x n), all column-major host matrices, on this library's fp64 MFMA GEMM: the operands are
// zero padded to N x N, N = the next multiple of 128 of max(m, k, n).  The kernel computes row-major C' = A' B'; the
// row-major view of a column-major matrix is its transpose, so it is handed B^T-image first: C^T = B^T A^T.
extern "C" int eagle_matmul(eagle_ctx* ctx, const double* A, const double* B, long m, long k, long n, double* C) {
    if (!ctx) return EAGLE_ERR_ARG;
    if (m <= 0 || k <= 0 || n <= 0 || !A || !B || !C) return eagle_fail(ctx, EAGLE_ERR_ARG, "matmul: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    long N = m > k ? m : k;
    if (n > N) N = n;
    N = (N + 127) / 128 * 128;
    DevBuf dA, dB, dC;
    const size_t sq = sizeof(double) * (size_t)N * N;
    HIPCHK(ctx, dA.alloc(sq)); HIPCHK(ctx, dB.alloc(sq)); HIPCHK(ctx, dC.alloc(sq));
    HIPCHK(ctx, hipMemsetAsync(dA.p, 0, sq, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dB.p, 0, sq, ctx->stream));
    // column-major A (m x k): column j is contiguous -> row j of the row-major image (= A^T), k rows of m doubles
    HIPCHK(ctx, hipMemcpy2DAsync(dA.p, sizeof(double) * N, A, sizeof(double) * m, sizeof(double) * m, k, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dB.p, sizeof(double) * N, B, sizeof(double) * k, sizeof(double) * k, n, hipMemcpyHostToDevice, ctx->stream));
    int rc = eagle_dev_gemm_f64(ctx, dB.as<double>(), dA.as<double>(), dC.as<double>(), N, ctx->stream);  // (B^T)(A^T) = (A B)^T
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy2DAsync(C, sizeof(double) * m, dC.p, sizeof(double) * N, sizeof(double) * m, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}

// E/R/calculateMMt_sqrt_and_sqrtinv.R:15-47 in one call, the matrices staying in HBM between the steps:
//   is.positive.definite(MMt)  (matrixcalc: every eigenvalue, with |lambda| < 1e-8 counted as 0, must be > 0)  -> else
//       EAGLE_SOFT_SENTINEL (the R function prints its message and returns NULL);
//   sqrt = U diag(sqrt(lambda)) U^T  (:25-27);   invsqrt = chol2inv(chol(sqrt))  (:30);
//   *trace_out = sum(diag(sqrt %*% invsqrt)) -- the caller mirrors the "trunc(.) != nrow(MMt)" warning of :35-46.
extern "C" int eagle_mmt_sqrt_and_sqrtinv(eagle_ctx* ctx, const double* MMt, long n, double* sqrt_out, double* invsqrt_out, double* trace_out) {
    int rc = check_n(ctx, n);
    if (rc) return rc;
    if (!MMt || !sqrt_out || !invsqrt_out) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_sqrt_and_sqrtinv: null argument");
    const long N = (n + 127) / 128 * 128;
    const size_t sq = sizeof(double) * (size_t)N * N;
    DevBuf dU, dw, dR, dRt, dS;
    HIPCHK(ctx, dU.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(ctx, dw.alloc(sizeof(double) * n));
    HIPCHK(ctx, hipMemcpyAsync(dU.p, MMt, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = dev_syevd(ctx, dU.as<double>(), n, dw.as<double>(), true))) return rc;
    std::vector<double> w(n);
    HIPCHK(ctx, hipMemcpyAsync(w.data(), dw.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (long i = 0; i < n; i++) {
        const double l = fabs(w[i]) < 1e-8 ? 0.0 : w[i];
        if (!(l > 0.0)) { eagle_fail(ctx, EAGLE_SOFT_SENTINEL, "the matrix multiplication M %*% t(M) is not positive definite"); return EAGLE_SOFT_SENTINEL; }
    }
    // R = row-major view of the eigenvector matrix = U^T (row i = eigenvector i); R2 = diag(lambda^1/4) R, padded to N x N;
    // sqrt = U diag(sqrt lambda) U^T = R2^T R2
    HIPCHK(ctx, dR.alloc(sq)); HIPCHK(ctx, dRt.alloc(sq)); HIPCHK(ctx, dS.alloc(sq));
    HIPCHK(ctx, hipMemsetAsync(dR.p, 0, sq, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dR.p, sizeof(double) * N, dU.p, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = eagle_dev_scale_rows_pow(ctx, dR.as<double>(), n, N, dw.as<double>(), 0.25, ctx->stream))) return rc;
    if ((rc = eagle_dev_transpose_f64(ctx, dR.as<double>(), dRt.as<double>(), N, ctx->stream))) return rc;
    if ((rc = eagle_dev_gemm_f64(ctx, dRt.as<double>(), dR.as<double>(), dS.as<double>(), N, ctx->stream))) return rc;
    if ((rc = eagle_dev_symmetrize_mean(ctx, dS.as<double>(), n, N, ctx->stream))) return rc;  // exact symmetry for the Cholesky below
    HIPCHK(ctx, hipMemcpy2DAsync(sqrt_out, sizeof(double) * n, dS.p, sizeof(double) * N, sizeof(double) * n, n, hipMemcpyDeviceToHost, ctx->stream));
    // invsqrt = chol2inv(chol(sqrt)) on an unpadded copy
    HIPCHK(ctx, hipMemcpy2DAsync(dU.p, sizeof(double) * n, dS.p, sizeof(double) * N, sizeof(double) * n, n, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = dev_chol2inv(ctx, dU.as<double>(), n))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(invsqrt_out, dU.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    if (trace_out) {  // tr(sqrt invsqrt) = sum_ij sqrt_ij invsqrt_ij (both symmetric), fixed-order reduction
        double* d_tr = (double*)((char*)ctx->d_scratch + EAGLE_SCR_INGEST);
        if ((rc = eagle_dev_dot_matrices(ctx, dS.as<double>(), N, dU.as<double>(), n, n, d_tr, ctx->stream))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(trace_out, d_tr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}
