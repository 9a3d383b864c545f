// eagle_spectral.hip -- the marker scan in the eigenbasis of MM^T (opt-in entry points; NOT symbols of the reference).
//
// What the reference computes per find_qtl iteration (E/R/find_qtl.R:5-62 + E/src/calculate_a_and_vara_rcpp.cpp:90-112) is
//     a_i = varG m_i^T P y ,   vara_i = varG^2 m_i^T P m_i ,   P = H^-1 - H^-1 X (X^T H^-1 X)^-1 X^T H^-1 ,   H = varE I + varG K
// (K = the normalised MM^T of calcMMt.R:13; W = S V S = varG^2 P), with an n x n quadratic form per marker:
// 2 L n^2 flop per scan.  K is the SAME matrix in every iteration of an AM() run -- only varE, varG and X change -- so with
// its eigen-decomposition K = U diag(lambda) U^T (which emma.REMLE computes anyway, E/R/emma_eigen_R_wo_Z.R:17) and
//     Z = Mt U      (L x n, once per AM() run),     d_k = 1 / (varE + varG lambda_k),     q_i = (D U^T X)^T z_i   (p numbers),
//     vara_i = varG^2 ( sum_k z_ik^2 d_k  -  q_i^T C q_i ) ,   a_i = varG ( z_i^T D U^T y  -  q_i^T C X^T H^-1 y ) ,   C = (X^T H^-1 X)^-1
// every later scan is ONE pass over Z: 8 n bytes and (p + 2) n multiply-adds per marker -- HBM-bound (80 GB at 10,000 x 1,000,000:
// ~15 ms against ~215 ms for the digit-slice scan of the opaque S, V interface).  The price is the one-time Z = Mt U
// (2 L n^2 fp64 flop, the cost of one fp64-mode scan) and 8 bytes per genotype of HBM.
//
//   k_zbuild ............ Z = Mt8 * U on the fp64 MFMA (int8 A converted in registers; the core of k_vara_f64 with a store epilogue)
//   k_spectral_scan ..... lin[i][0..p] = z_i^T G (G = [D U^T y | D U^T X]) on v_mfma_f64_16x16x4_f64, quad_i = sum_k z_ik^2 d_k on the VALU,
//                         one streaming read of Z                                                          (bound: HBM)
//   k_spectral_finish ... the p x p form per marker, a and vara
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/eagle_hip.h"
#include "eagle_ctx.h"
#include "eagle_internal.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// Z[i][k] = sum_j Mt8[i][j] U[j][k].  128 x 128 tile, 4 waves 2 x 2, wave tile 64 x 64, K block 16 (the fp64 core of
// eagle_kernels.hip restated here with a store epilogue).  B = U row-major [np][np].
// ---------------------------------------------------------------------------------------------------------------
#define ZB_T 128
#define ZB_BK 16
#define ZB_LDB (ZB_T + 4)
__global__ __launch_bounds__(256, 2) void k_zbuild(const int8_t* __restrict__ A8, long lda, const double* __restrict__ B, long ldb,
                                                   double* __restrict__ Z, long ldz, long K) {
    __shared__ __attribute__((aligned(16))) double ldsB[2][ZB_BK * ZB_LDB];
    __shared__ __attribute__((aligned(16))) int8_t ldsA[2][ZB_T * 16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * ZB_T, col0 = (long)blockIdx.y * ZB_T;
    const int8_t* Ablk = A8 + row0 * lda;
    const double* Bblk = B + col0;
    f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    f64x2 sb[4];
    i32x4 sa = {0, 0, 0, 0};
    auto load = [&](long k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = t + 256 * i;
            sb[i] = *(const f64x2*)(Bblk + (k0 + (c >> 6)) * ldb + (c & 63) * 2);
        }
        if (t < 128) sa = *(const i32x4*)(Ablk + (long)t * lda + k0);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = t + 256 * i;
            *(f64x2*)(&ldsB[buf][(c >> 6) * ZB_LDB + (c & 63) * 2]) = sb[i];
        }
        if (t < 128) *(i32x4*)(&ldsA[buf][t * 16]) = sa;
    };
    const long nkb = K / ZB_BK;
    load(0);
    store(0);
    __syncthreads();
    int cur = 0;
    for (long kb = 0; kb < nkb; kb++) {
        const bool more = kb + 1 < nkb;
        if (more) load((kb + 1) * ZB_BK);
        int w4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) w4[m] = *(const int*)(&ldsA[cur][(wr * 64 + m * 16 + i16) * 16 + 4 * g]);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; m++) a[m] = (double)((w4[m] << (24 - 8 * s)) >> 24);
#pragma unroll
            for (int n = 0; n < 4; n++) b[n] = ldsB[cur][(4 * g + s) * ZB_LDB + wc * 64 + n * 16 + i16];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (more) store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                Z[(row0 + wr * 64 + m * 16 + g + 4 * q) * ldz + col0 + wc * 64 + n * 16 + i16] = acc[m][n][q];
}

// ---------------------------------------------------------------------------------------------------------------
// One pass over Z.  Workgroup = 16 waves = 256 markers; wave = 16 markers.  G (np x NC, NC = 16 or 32 columns: D U^T y and the
// p columns of D U^T X, zero padded) and d go through LDS in chunks of SP_KC rows shared by the 16 waves; lane (i16, g) streams
// Z[marker i16][k0 + 4g .. 4g+3] (32 contiguous bytes; the four lane groups of a row cover 128 bytes) and uses element s as the
// k = 4g + s operand of MFMA step s -- the same k permutation as the B fragment.
// ---------------------------------------------------------------------------------------------------------------
#define SP_KC 256
template <int NT>  // NT = 1: up to 16 columns, NT = 2: up to 32
__global__ __launch_bounds__(1024) void k_spectral_scan(const double* __restrict__ Z, long ldz, long K, const double* __restrict__ G,
                                                        const double* __restrict__ dvec, double* __restrict__ lin, double* __restrict__ quad) {
    constexpr int NC = 16 * NT;
    __shared__ __attribute__((aligned(16))) double ldsG[2][SP_KC * NC];
    __shared__ __attribute__((aligned(16))) double ldsD[2][SP_KC];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * 256 + wv * 16;
    const double* zrow = Z + (row0 + i16) * ldz + 4 * g;
    f64x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; n++) acc[n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double qacc = 0.0;
    const long nchunks = K / SP_KC;
    auto stage = [&](long c, int buf) {
        const double* src = G + c * SP_KC * NC;
        for (int e = t * 2; e < SP_KC * NC; e += 2048) *(f64x2*)(&ldsG[buf][e]) = *(const f64x2*)(src + e);
        if (t < SP_KC) ldsD[buf][t] = dvec[c * SP_KC + t];
    };
    stage(0, 0);
    __syncthreads();
    for (long c = 0; c < nchunks; c++) {
        const int buf = (int)(c & 1);
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
        const double* zc = zrow + c * SP_KC;
#pragma unroll 4
        for (int ks = 0; ks < SP_KC / 16; ks++) {
            const f64x2 z01 = *(const f64x2*)(zc + ks * 16);
            const f64x2 z23 = *(const f64x2*)(zc + ks * 16 + 2);
            const double zz[4] = {z01[0], z01[1], z23[0], z23[1]};
            const f64x4 dd = *(const f64x4*)(&ldsD[buf][ks * 16 + 4 * g]);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int kk = ks * 16 + 4 * g + s;
#pragma unroll
                for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(zz[s], ldsG[buf][kk * NC + n * 16 + i16], acc[n], 0, 0, 0);
                qacc += zz[s] * zz[s] * dd[s];
            }
        }
        __syncthreads();
    }
    // C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int q = 0; q < 4; q++) lin[(row0 + g + 4 * q) * NC + n * 16 + i16] = acc[n][q];
    qacc += __shfl_xor(qacc, 16);
    qacc += __shfl_xor(qacc, 32);
    if (g == 0) quad[row0 + i16] = qacc;
}

// a_i = varG (lin_i0 - q_i . c1),  vara_i = varG^2 (quad_i - q_i^T C q_i),  q_i = lin_i[1..p]
__global__ __launch_bounds__(256) void k_spectral_finish(const double* __restrict__ lin, int NC, const double* __restrict__ quad, long L, int p,
                                                         const double* __restrict__ Cm, const double* __restrict__ c1, double varG,
                                                         double* __restrict__ a, double* __restrict__ vara) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    const double* li = lin + i * NC;
    double qc1 = 0.0, qCq = 0.0;
    for (int j = 0; j < p; j++) {
        const double qj = li[1 + j];
        qc1 += qj * c1[j];
        double r = 0.0;
        for (int l = 0; l < p; l++) r += Cm[j * p + l] * li[1 + l];
        qCq += qj * r;
    }
    // A marker that is (numerically) in the column space of X -- one that is already in the model -- has a = vara = 0 in exact
    // arithmetic and rounding noise here (the reference's selected_loci masking, which never fires from AM(), was meant for
    // exactly these): give it the masked values a = vara = 0 (tsq NaN, skipped by na.rm) instead of noise / noise.
    const double r = quad[i] - qCq;
    const bool in_model = !(r > 1e-12 * quad[i]);
    a[i] = in_model ? 0.0 : varG * (li[0] - qc1);
    vara[i] = in_model ? 0.0 : varG * varG * r;
}

// ---------------------------------------------------------------------------------------------------------------
// Entry points
// ---------------------------------------------------------------------------------------------------------------
#define SP_LAUNCH_CHECK(ctx, what)                                            \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) return eagle_fail_hip(ctx, e__, what);         \
    } while (0)

// Device-resident forms (HBM pointers; what bench.py times and what the host entry points below are made of).
//   zbuild : Z[L_pad][n_pad] = Mt8[L_pad][ld] * Ur,  Ur = U row-major [n_pad][n_pad] (Ur[j][k] = U[j][k], zero padded)
//   pass   : lin[L_pad][NC] = Z G,  quad[L_pad] = sum_k Z_ik^2 d_k;  G [n_pad][NC] row-major, NC = 16 or 32
//   finish : a_i = varG (lin_i0 - q_i . c1), vara_i = varG^2 (quad_i - q_i^T C q_i), q_i = lin_i[1..p]
extern "C" int eagle_dev_spectral_zbuild(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Ur, double* Z, void* stream) {
    if (L_pad % 256 || n_pad % 256 || ld % 16 || n_pad > ld || L_pad / ZB_T > 2147483647L || n_pad / ZB_T > 65535)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_zbuild: layout contract violated (L_pad % 256, n_pad % 256)");
    if (L_pad == 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_zbuild, dim3((unsigned)(L_pad / ZB_T), (unsigned)(n_pad / ZB_T)), dim3(256), 0, (hipStream_t)stream, Mt8, ld, Ur, n_pad, Z, n_pad, n_pad);
    SP_LAUNCH_CHECK(ctx, "k_zbuild");
    return EAGLE_OK;
}
extern "C" int eagle_dev_spectral_pass(eagle_ctx* ctx, const double* Z, long L_pad, long n_pad, const double* G, int NC, const double* d, double* lin,
                                       double* quad, void* stream) {
    if (L_pad % 256 || n_pad % SP_KC || (NC != 16 && NC != 32)) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_pass: layout contract violated");
    if (L_pad == 0) return EAGLE_OK;
    if (NC == 16)
        hipLaunchKernelGGL((k_spectral_scan<1>), dim3((unsigned)(L_pad / 256)), dim3(1024), 0, (hipStream_t)stream, Z, n_pad, n_pad, G, d, lin, quad);
    else
        hipLaunchKernelGGL((k_spectral_scan<2>), dim3((unsigned)(L_pad / 256)), dim3(1024), 0, (hipStream_t)stream, Z, n_pad, n_pad, G, d, lin, quad);
    SP_LAUNCH_CHECK(ctx, "k_spectral_scan");
    return EAGLE_OK;
}
extern "C" int eagle_dev_spectral_finish(eagle_ctx* ctx, const double* lin, int NC, const double* quad, long L, long p, const double* Cm, const double* c1,
                                         double varG, double* a, double* vara, void* stream) {
    if (p < 1 || p + 1 > NC) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_finish: 1 <= p < NC");
    if (L <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_spectral_finish, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lin, NC, quad, L, (int)p, Cm, c1, varG, a, vara);
    SP_LAUNCH_CHECK(ctx, "k_spectral_finish");
    return EAGLE_OK;
}

extern "C" void eagle_spectral_release(eagle_ctx* ctx) {
    if (!ctx) return;
    if (ctx->d_Z) { (void)hipFree(ctx->d_Z); ctx->d_Z = nullptr; }
    ctx->z_L = ctx->z_n = 0;
}

// Z of the markers [m0, m1) of Mt.ascii on ctx's device (the whole file on one device; a marker range per device of a
// multi-device context: Z shards by markers exactly like the genotypes).
extern "C" int eagle_spectral_prepare_range(eagle_ctx* ctx, const char* f_name_ascii, long L, long n, long m0, long m1, const double* U,
                                            double max_memory_in_Gbytes) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    eagle_spectral_release(ctx);
    const long Lr = m1 - m0;
    ctx->z_first = m0;
    if (Lr <= 0) { ctx->z_L = 0; ctx->z_n = n; return EAGLE_OK; }
    const long np = eagle_pad(n), Lp = eagle_pad(Lr);
    const GenoEntry* g = nullptr;
    int rc = eagle_get_resident_window(ctx, f_name_ascii, m0, Lr, 0, n, max_memory_in_Gbytes, host_threads(), &g);
    if (rc < 0) return rc;
    if (rc != EAGLE_OK || !g) return eagle_fail(ctx, EAGLE_ERR_NOMEM, "spectral_prepare: the genotype shard must fit in HBM next to Z (8 bytes per genotype)");
    size_t freeb = 0, totalb = 0;
    HIPCHK(ctx, hipMemGetInfo(&freeb, &totalb));
    const size_t zbytes = sizeof(double) * (size_t)Lp * np, ubytes = sizeof(double) * (size_t)np * np;
    if (zbytes + 2 * ubytes + ((size_t)1 << 30) > freeb) return eagle_fail(ctx, EAGLE_ERR_NOMEM, "spectral_prepare: not enough HBM for Z = Mt U (8 bytes per genotype)");
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_Z, zbytes));
    DevBuf Ut, Ur;
    HIPCHK(ctx, Ut.alloc(ubytes));
    HIPCHK(ctx, Ur.alloc(ubytes));
    // column-major U -> row-major image of U^T (zero padded) -> transpose: Ur[j][k] = U[j][k]
    HIPCHK(ctx, hipMemsetAsync(Ut.p, 0, ubytes, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(Ut.p, sizeof(double) * np, U, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = eagle_dev_transpose_f64(ctx, Ut.as<double>(), Ur.as<double>(), np, ctx->stream))) return rc;
    if (ctx->scan_mode == 1 && 128.0 * (double)np < 2147483648.0) {
        // default: six exact int8 digit slices of U on the int8 MFMA; |Z - Mt U| <= n 2^(e+1-48) <= n 2^-47 (|U| <= 1)
        DevBuf ws;
        HIPCHK(ctx, ws.alloc((size_t)eagle_spectral_zbuild_i8_workspace_bytes(np, 6)));
        if ((rc = eagle_dev_spectral_zbuild_i8(ctx, g->dev, Lp, np, g->ld, Ur.as<double>(), ctx->d_Z, ws.p, 6, ctx->stream))) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else if ((rc = eagle_dev_spectral_zbuild(ctx, g->dev, Lp, np, g->ld, Ur.as<double>(), ctx->d_Z, ctx->stream))) {  // eagle_set_scan_mode(0): fp64 MFMA
        return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->z_L = Lr; ctx->z_n = n;
    return EAGLE_OK;
}

// p x p symmetric positive definite: in-place inverse by Cholesky (long double accumulations); false if not positive definite
static bool small_spd_inverse(std::vector<long double>& A, int p) {
    std::vector<long double> Lm((size_t)p * p, 0.0L);
    for (int j = 0; j < p; j++) {
        long double s = A[(size_t)j * p + j];
        for (int k = 0; k < j; k++) s -= Lm[(size_t)j * p + k] * Lm[(size_t)j * p + k];
        if (!(s > 0.0L)) return false;
        const long double d = sqrtl(s);
        Lm[(size_t)j * p + j] = d;
        for (int i = j + 1; i < p; i++) {
            long double v = A[(size_t)i * p + j];
            for (int k = 0; k < j; k++) v -= Lm[(size_t)i * p + k] * Lm[(size_t)j * p + k];
            Lm[(size_t)i * p + j] = v / d;
        }
    }
    // inverse of L (lower), then A^-1 = L^-T L^-1
    std::vector<long double> Li((size_t)p * p, 0.0L);
    for (int j = 0; j < p; j++) {
        Li[(size_t)j * p + j] = 1.0L / Lm[(size_t)j * p + j];
        for (int i = j + 1; i < p; i++) {
            long double v = 0.0L;
            for (int k = j; k < i; k++) v -= Lm[(size_t)i * p + k] * Li[(size_t)k * p + j];
            Li[(size_t)i * p + j] = v / Lm[(size_t)i * p + i];
        }
    }
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) {
            long double v = 0.0L;
            for (int k = (i > j ? i : j); k < p; k++) v += Li[(size_t)k * p + i] * Li[(size_t)k * p + j];
            A[(size_t)i * p + j] = v;
        }
    return true;
}

// Host side of a scan, once per call whatever the number of devices: d, G = [d o U^T y | d o U^T X] (n_pad x NC row-major),
// C = (X^T H^-1 X)^-1 = (UtX^T D UtX)^-1, c1 = C (UtX^T D Uty).
extern "C" int eagle_spectral_host_operands(eagle_ctx* ctx, long n, const double* lambda, const double* UtX, const double* Uty, long p, double varE,
                                            double varG, int NC, double* d, double* G, double* Cm, double* c1) {
    const long np = eagle_pad(n);
    memset(d, 0, sizeof(double) * np);
    memset(G, 0, sizeof(double) * (size_t)np * NC);
    for (long k = 0; k < n; k++) {
        const double h = varE + varG * lambda[k];
        if (!(h > 0.0)) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_scan: varE + varG * lambda must be positive");
        d[k] = 1.0 / h;
        G[(size_t)k * NC] = d[k] * Uty[k];
        for (long j = 0; j < p; j++) G[(size_t)k * NC + 1 + j] = d[k] * UtX[j * n + k];
    }
    std::vector<long double> A((size_t)p * p, 0.0L), bvec(p, 0.0L);
    for (long j = 0; j < p; j++) {
        for (long l = j; l < p; l++) {
            long double s = 0.0L;
            for (long k = 0; k < n; k++) s += (long double)UtX[j * n + k] * (long double)d[k] * (long double)UtX[l * n + k];
            A[(size_t)j * p + l] = s;
            A[(size_t)l * p + j] = s;
        }
        long double s = 0.0L;
        for (long k = 0; k < n; k++) s += (long double)UtX[j * n + k] * (long double)d[k] * (long double)Uty[k];
        bvec[j] = s;
    }
    if (!small_spd_inverse(A, (int)p)) return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_scan: X^T H^-1 X is not positive definite (collinear fixed effects)");
    for (long j = 0; j < p; j++) {
        long double s = 0.0L;
        for (long l = 0; l < p; l++) { Cm[(size_t)j * p + l] = (double)A[(size_t)j * p + l]; s += A[(size_t)j * p + l] * bvec[l]; }
        c1[j] = (double)s;
    }
    return EAGLE_OK;
}

// One device's pass over its shard of Z; results into a_out / vara_out at the shard's global marker positions.
extern "C" int eagle_spectral_scan_range(eagle_ctx* ctx, const double* d, const double* G, int NC, const double* Cm, const double* c1, long p,
                                         double varG, const long* sel, long nsel, double* a_out, double* vara_out) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long L = ctx->z_L, n = ctx->z_n, m0 = ctx->z_first;
    if (L <= 0) return EAGLE_OK;
    const long np = eagle_pad(n), Lp = eagle_pad(L);
    DevBuf dG, dd, dC, dc1, dlin, dquad, da, dv, dsel;
    HIPCHK(ctx, dG.alloc(sizeof(double) * (size_t)np * NC));
    HIPCHK(ctx, dd.alloc(sizeof(double) * np));
    HIPCHK(ctx, dC.alloc(sizeof(double) * (size_t)p * p));
    HIPCHK(ctx, dc1.alloc(sizeof(double) * p));
    HIPCHK(ctx, dlin.alloc(sizeof(double) * (size_t)Lp * NC));
    HIPCHK(ctx, dquad.alloc(sizeof(double) * Lp));
    HIPCHK(ctx, da.alloc(sizeof(double) * Lp));
    HIPCHK(ctx, dv.alloc(sizeof(double) * Lp));
    HIPCHK(ctx, hipMemcpyAsync(dG.p, G, sizeof(double) * (size_t)np * NC, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dd.p, d, sizeof(double) * np, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dC.p, Cm, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dc1.p, c1, sizeof(double) * p, hipMemcpyHostToDevice, ctx->stream));
    int rcs = eagle_dev_spectral_pass(ctx, ctx->d_Z, Lp, np, dG.as<double>(), NC, dd.as<double>(), dlin.as<double>(), dquad.as<double>(), ctx->stream);
    if (rcs) return rcs;
    if ((rcs = eagle_dev_spectral_finish(ctx, dlin.as<double>(), NC, dquad.as<double>(), L, p, dC.as<double>(), dc1.as<double>(), varG, da.as<double>(),
                                         dv.as<double>(), ctx->stream)))
        return rcs;
    std::vector<long> in_range;
    for (long i = 0; i < nsel; i++) if (sel[i] >= m0 && sel[i] < m0 + L) in_range.push_back(sel[i] - m0);
    if (!in_range.empty()) {
        HIPCHK(ctx, dsel.alloc(sizeof(long) * in_range.size()));
        HIPCHK(ctx, hipMemcpyAsync(dsel.p, in_range.data(), sizeof(long) * in_range.size(), hipMemcpyHostToDevice, ctx->stream));
        int rc = eagle_dev_zero_rows(ctx, da.as<double>(), dv.as<double>(), L, dsel.as<long>(), (long)in_range.size(), 0, ctx->stream);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(a_out + m0, da.p, sizeof(double) * L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(vara_out + m0, dv.p, sizeof(double) * L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return EAGLE_OK;
}
