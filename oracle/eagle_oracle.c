/*
 * eagle_oracle.c -- CPU ORACLE for the Eagle/WMAM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C (C11 + OpenMP) restatement of the reference's algorithm for the
 * one path this repository accelerates.  It is the checker for the HIP path and the timed
 * "port" CPU baseline of bench.py; nothing under eagleeverything_amd/ may call into it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library.
 *
 * PARITY STATUS: *** parity unpinned ***
 *   The reference (jcbowden/EagleEverything, Eagle 1.0.3) needs R + Rcpp + RcppEigen, none of
 *   which exist in the build container, so it cannot be compiled or run here, and it ships no
 *   test suite and no recorded expected outputs.  The arithmetic that
 *   lives in third-party code (Eigen dense products via RcppEigen, version unpinned in
 *   DESCRIPTION:42-43) is restated here from its published meaning (C = A*B in IEEE fp64).
 *   What pins this oracle: (i) exact int64 known answers for MM^T on the reference's own demo
 *   genotype files (tests/golden, derived from reference semantics, not from reference output),
 *   (ii) an independent numpy/OpenBLAS restatement (oracle/oracle_np.py) that must agree to
 *   1e-12 relative.  fp64 outputs (a, vara, tsq) therefore carry "parity unpinned".
 *
 * Reference lines followed (E/ = /root/reference/MyPackage/Eagle/):
 *   eo_read_block ............ E/src/ReadBlock.cpp:47-58
 *   eo_calculateMMt .......... E/src/calculateMMt_rcpp.cpp:75-76 (memory test), :84-95 (in-memory),
 *                              :99-174 (row-block branch)
 *   eo_normalise_MMt ......... E/R/calcMMt.R:13
 *   eo_calculate_a_and_vara .. E/src/calculate_a_and_vara_rcpp.cpp:65,74 (memory test), :76-112
 *                              (in-memory), :129-230 (marker-block branch)
 *   eo_calculate_reduced_a ... E/src/calculate_reduced_a_rcpp.cpp:56,65-85
 *   eo_tsq_argmax ............ E/R/find_qtl.R:71-83
 *
 * Matrices that cross this interface are column-major (R / Eigen default).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EO_OK 0
#define EO_ERR_OPEN (-1)      /* ReadBlock.cpp:42-45 Rcpp::stop */
#define EO_ERR_SHORT (-2)     /* file has fewer lines / shorter lines than asked: UB in the reference */
#define EO_ERR_ARG (-3)
#define EO_ERR_NOMEM (-4)
#define EO_SOFT_SENTINEL 1    /* reference printed a message and returned its sentinel value */

static __thread char eo_errbuf[512];
const char* eo_last_error(void) { return eo_errbuf; }
static int eo_fail(int code, const char* fmt, const char* arg) {
    snprintf(eo_errbuf, sizeof eo_errbuf, fmt, arg ? arg : "");
    return code;
}

/* R's NA_real_ is a NaN; R_IsNA() tests its payload.  Any NaN is treated as NA here
 * (a non-NA NaN index is undefined behaviour in the reference). */
static int eo_is_na(double x) { return isnan(x); }

/* ------------------------------------------------------------------------------------------
 * Blocked fp64 GEMM   C[M x N] = A[M x K] * B[K x N]   (generic strides, C row-major)
 * Stands in for the Eigen product the reference calls (calculateMMt_rcpp.cpp:95,
 * calculate_a_and_vara_rcpp.cpp:90-91,97-98,103).  Packed panels, 6x8 register tile,
 * OpenMP over row blocks -- fast enough to serve as the timed CPU baseline.
 * ------------------------------------------------------------------------------------------ */
typedef double v4d __attribute__((vector_size(32), aligned(8)));
#define EO_MR 6
#define EO_NR 8
#define EO_KC 256
#define EO_MC 96
#define EO_NC 2048

/* aligned_alloc wants a size that is a multiple of the alignment (C11 7.22.3.1; found by the ASan build) */
static size_t eo_round64(size_t b) { return (b + 63) / 64 * 64; }
static void eo_pack_b(long kc, long nc, const double* B, long rsb, long csb, double* Bp) {
    /* Bp: panels of EO_NR columns, each panel kc x EO_NR row-major, zero padded */
    for (long j0 = 0; j0 < nc; j0 += EO_NR) {
        long nr = nc - j0 < EO_NR ? nc - j0 : EO_NR;
        for (long k = 0; k < kc; k++) {
            for (long j = 0; j < nr; j++) Bp[j] = B[k * rsb + (j0 + j) * csb];
            for (long j = nr; j < EO_NR; j++) Bp[j] = 0.0;
            Bp += EO_NR;
        }
    }
}
static void eo_pack_a(long mc, long kc, const double* A, long rsa, long csa, double* Ap) {
    /* Ap: panels of EO_MR rows, each panel kc x EO_MR (k-major), zero padded */
    for (long i0 = 0; i0 < mc; i0 += EO_MR) {
        long mr = mc - i0 < EO_MR ? mc - i0 : EO_MR;
        for (long k = 0; k < kc; k++) {
            for (long i = 0; i < mr; i++) Ap[i] = A[(i0 + i) * rsa + k * csa];
            for (long i = mr; i < EO_MR; i++) Ap[i] = 0.0;
            Ap += EO_MR;
        }
    }
}
static inline void eo_micro(long kc, const double* Ap, const double* Bp, double* C, long ldc,
                            long mr, long nr, int accumulate) {
    v4d c[EO_MR][2];
    for (int i = 0; i < EO_MR; i++) { c[i][0] = (v4d){0, 0, 0, 0}; c[i][1] = (v4d){0, 0, 0, 0}; }
    for (long k = 0; k < kc; k++) {
        v4d b0 = *(const v4d*)(Bp + k * EO_NR), b1 = *(const v4d*)(Bp + k * EO_NR + 4);
        for (int i = 0; i < EO_MR; i++) {
            double a = Ap[k * EO_MR + i];
            v4d av = {a, a, a, a};
            c[i][0] += av * b0;
            c[i][1] += av * b1;
        }
    }
    for (long i = 0; i < mr; i++)
        for (long j = 0; j < nr; j++) {
            double v = c[i][j >> 2][j & 3];
            if (accumulate) C[i * ldc + j] += v; else C[i * ldc + j] = v;
        }
}
/* A(i,k) = A[i*rsa + k*csa]; B(k,j) = B[k*rsb + j*csb]; C row-major ldc. */
static int eo_dgemm(long M, long N, long K, const double* A, long rsa, long csa, const double* B, long rsb,
                    long csb, double* C, long ldc) {
    if (M <= 0 || N <= 0) return EO_OK;
    if (K <= 0) {
        for (long i = 0; i < M; i++) memset(C + i * ldc, 0, sizeof(double) * (size_t)N);
        return EO_OK;
    }
    int failed = 0;
    for (long jc = 0; jc < N; jc += EO_NC) {
        long nc = N - jc < EO_NC ? N - jc : EO_NC;
        long ncp = (nc + EO_NR - 1) / EO_NR * EO_NR;
        for (long pc = 0; pc < K; pc += EO_KC) {
            long kc = K - pc < EO_KC ? K - pc : EO_KC;
            double* Bp = (double*)aligned_alloc(64, eo_round64(sizeof(double) * (size_t)(ncp * kc + 8)));
            if (!Bp) return eo_fail(EO_ERR_NOMEM, "out of memory in dgemm%s", NULL);
            eo_pack_b(kc, nc, B + pc * rsb + jc * csb, rsb, csb, Bp);
#pragma omp parallel
            {
                double* Ap = (double*)aligned_alloc(64, eo_round64(sizeof(double) * (size_t)((EO_MC + EO_MR) * kc + 8)));
                if (!Ap) {
#pragma omp atomic write
                    failed = 1;
                }
#pragma omp barrier
                if (!failed) {
#pragma omp for schedule(dynamic, 1)
                    for (long ic = 0; ic < M; ic += EO_MC) {
                        long mc = M - ic < EO_MC ? M - ic : EO_MC;
                        eo_pack_a(mc, kc, A + ic * rsa + pc * csa, rsa, csa, Ap);
                        for (long jr = 0; jr < nc; jr += EO_NR) {
                            long nr = nc - jr < EO_NR ? nc - jr : EO_NR;
                            for (long ir = 0; ir < mc; ir += EO_MR) {
                                long mr = mc - ir < EO_MR ? mc - ir : EO_MR;
                                eo_micro(kc, Ap + ir * kc, Bp + jr * kc, C + (ic + ir) * ldc + jc + jr, ldc, mr, nr,
                                         pc != 0);
                            }
                        }
                    }
                }
                free(Ap);
            }
            free(Bp);
            if (failed) return eo_fail(EO_ERR_NOMEM, "out of memory in dgemm%s", NULL);
        }
    }
    return EO_OK;
}

/* ------------------------------------------------------------------------------------------
 * ReadBlock  (E/src/ReadBlock.cpp:47-58)
 *   for rr in [0, start_row+numrows): getline; if rr >= start_row: M(rowi,ii) = (line[ii]-'0') - 1
 * Internal form writes row-major doubles; the exported form is column-major like Eigen::MatrixXd.
 * ------------------------------------------------------------------------------------------ */
static int eo_read_rows_rm(const char* path, long start_row, long numcols, long numrows, double* out_rm) {
    FILE* f = fopen(path, "r");
    if (!f) return eo_fail(EO_ERR_OPEN, "ERROR: Could not open  %s", path);
    char* line = NULL;
    size_t cap = 0;
    long rowi = 0;
    int rc = EO_OK;
    for (long rr = 0; rr < start_row + numrows; rr++) {
        ssize_t len = getline(&line, &cap, f);
        if (len < 0) { rc = eo_fail(EO_ERR_SHORT, "file %s has fewer lines than requested", path); break; }
        if (rr >= start_row) {
            while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
            if (len < numcols) { rc = eo_fail(EO_ERR_SHORT, "line shorter than numcols in %s", path); break; }
            for (long ii = 0; ii < numcols; ii++) {
                int tmp = line[ii] - '0';
                out_rm[rowi * numcols + ii] = (double)tmp - 1;
            }
            rowi++;
        }
    }
    free(line);
    fclose(f);
    return rc;
}

int eo_read_block(const char* path, long start_row, long numcols, long numrows, double* out_colmajor) {
    if (numcols < 0 || numrows < 0 || start_row < 0) return eo_fail(EO_ERR_ARG, "negative dimension%s", NULL);
    if (numcols == 0 || numrows == 0) {
        FILE* f = fopen(path, "r");
        if (!f) return eo_fail(EO_ERR_OPEN, "ERROR: Could not open  %s", path);
        fclose(f);
        return EO_OK;
    }
    double* rm = (double*)malloc(sizeof(double) * (size_t)numcols * (size_t)numrows);
    if (!rm) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
    int rc = eo_read_rows_rm(path, start_row, numcols, numrows, rm);
    if (rc == EO_OK)
        for (long r = 0; r < numrows; r++)
            for (long c = 0; c < numcols; c++) out_colmajor[r + c * numrows] = rm[r * numcols + c];
    free(rm);
    return rc;
}

/* Convert the selected_loci argument (raw R doubles) to indices, applying the rule that
 * masking happens iff element 0 is not NA (calculateMMt_rcpp.cpp:88, calculate_a_and_vara_rcpp.cpp:79). */
static int eo_sel_active(const double* sel, long nsel) { return nsel > 0 && sel && !eo_is_na(sel[0]); }
static int eo_sel_check(const double* sel, long nsel, long bound) {
    if (!eo_sel_active(sel, nsel)) return EO_OK;
    for (long i = 0; i < nsel; i++) {
        if (eo_is_na(sel[i])) return eo_fail(EO_ERR_ARG, "NA in selected_loci after element 0%s", NULL);
        long v = (long)sel[i];
        if (v < 0 || v >= bound) return eo_fail(EO_ERR_ARG, "selected_loci index out of range%s", NULL);
    }
    return EO_OK;
}

/* ------------------------------------------------------------------------------------------
 * calculateMMt_rcpp  (E/src/calculateMMt_rcpp.cpp:19-185);  dims = (n, L), file = M.ascii
 * ------------------------------------------------------------------------------------------ */
static void eo_zero_cols_rm(double* A, long rows, long cols, const double* sel, long nsel) {
    for (long ii = 0; ii < nsel; ii++) {
        long c = (long)sel[ii];
        for (long r = 0; r < rows; r++) A[r * cols + c] = 0.0;
    }
}

int eo_calculateMMt(const char* path, double max_memory_in_Gbytes, int num_cores, const double* sel, long nsel,
                    long n, long L, double* MMt_colmajor, long* branch_out) {
    if (n <= 0 || L <= 0) return eo_fail(EO_ERR_ARG, "bad dims%s", NULL);
    int rc = eo_sel_check(sel, nsel, L);
    if (rc) return rc;
#ifdef _OPENMP
    if (num_cores > 0) omp_set_num_threads(num_cores); /* :25-35 */
#endif
    const int masked = eo_sel_active(sel, nsel);
    /* :75-76 */
    double memory_needed_in_Gb =
        (double)((unsigned long)n * n * sizeof(double) + 2 * ((unsigned long)n * L * sizeof(double))) / 1000000000.0;
    memset(MMt_colmajor, 0, sizeof(double) * (size_t)n * (size_t)n);
    if (max_memory_in_Gbytes > memory_needed_in_Gb) {
        if (branch_out) *branch_out = 0;
        double* G = (double*)malloc(sizeof(double) * (size_t)n * (size_t)L);
        if (!G) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
        rc = eo_read_rows_rm(path, 0, L, n, G); /* :86 */
        if (rc == EO_OK) {
            if (masked) eo_zero_cols_rm(G, n, L, sel, nsel); /* :88-92 */
            rc = eo_dgemm(n, n, L, G, L, 1, G, 1, L, MMt_colmajor, n); /* :95  (symmetric, layout-free) */
        }
        free(G);
        return rc;
    }
    /* :103-106 */
    double part1 = -2.0 * (double)L;
    double part2 = 4.0 * (double)L * (double)L + 4.0 * max_memory_in_Gbytes * 1000000000.0 / sizeof(double);
    part2 = sqrt(part2);
    long num_rows_in_block = (long)((part1 + part2) / 2.2);
    if (num_rows_in_block <= 0)
        return eo_fail(EO_ERR_ARG, "availmemGb too small: zero rows per block (reference divides by zero)%s", NULL);
    if (branch_out) *branch_out = num_rows_in_block;
    long num_blocks = n / num_rows_in_block; /* :113-118 */
    if (n % num_rows_in_block) num_blocks++;
    double* B1 = (double*)malloc(sizeof(double) * (size_t)num_rows_in_block * (size_t)L);
    double* B2 = (double*)malloc(sizeof(double) * (size_t)num_rows_in_block * (size_t)L);
    double* sub = (double*)malloc(sizeof(double) * (size_t)num_rows_in_block * (size_t)num_rows_in_block);
    if (!B1 || !B2 || !sub) { free(B1); free(B2); free(sub); return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL); }
    for (long i = 0; i < num_blocks && rc == EO_OK; i++) { /* :121 */
        long s1 = i * num_rows_in_block, r1 = num_rows_in_block;
        if (s1 + r1 > n) r1 = n - s1;
        rc = eo_read_rows_rm(path, s1, L, r1, B1); /* :129 */
        if (rc) break;
        if (masked) eo_zero_cols_rm(B1, r1, L, sel, nsel); /* :133-137 */
        rc = eo_dgemm(r1, r1, L, B1, L, 1, B1, 1, L, sub, r1); /* :138 */
        if (rc) break;
        for (long a = 0; a < r1; a++) /* :140 */
            for (long b = 0; b < r1; b++) MMt_colmajor[(s1 + a) + (s1 + b) * n] = sub[a * r1 + b];
        for (long j = i + 1; j < num_blocks; j++) { /* :142 */
            long s2 = j * num_rows_in_block, r2 = num_rows_in_block;
            if (s2 + r2 > n) r2 = n - s2;
            rc = eo_read_rows_rm(path, s2, L, r2, B2); /* :148 */
            if (rc) break;
            if (masked) eo_zero_cols_rm(B2, r2, L, sel, nsel); /* :156-160 */
            rc = eo_dgemm(r1, r2, L, B1, L, 1, B2, 1, L, sub, r2); /* :161 */
            if (rc) break;
            for (long a = 0; a < r1; a++)
                for (long b = 0; b < r2; b++) {
                    MMt_colmajor[(s1 + a) + (s2 + b) * n] = sub[a * r2 + b]; /* :163 */
                    MMt_colmajor[(s2 + b) + (s1 + a) * n] = sub[a * r2 + b]; /* :165 */
                }
        }
    }
    free(B1); free(B2); free(sub);
    return rc;
}

/* In-memory MM^T from an int8 {-1,0,1} individual-major matrix (n x L, row stride ld); converts to
 * doubles first exactly as ReadBlock would hand them to Eigen.  Used by the CPU-baseline leg. */
int eo_mmt_from_i8(const int8_t* M8, long n, long L, long ld, double* MMt_colmajor) {
    double* G = (double*)malloc(sizeof(double) * (size_t)n * (size_t)L);
    if (!G) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < n; r++)
        for (long c = 0; c < L; c++) G[r * L + c] = (double)M8[r * ld + c];
    int rc = eo_dgemm(n, n, L, G, L, 1, G, 1, L, MMt_colmajor, n);
    free(G);
    return rc;
}

/* E/R/calcMMt.R:13   MMt <- MMt/max(MMt) + diag(0.95, nrow(MMt)) */
void eo_normalise_MMt(double* MMt, long n) {
    double mx = -INFINITY;
    for (long i = 0; i < n * n; i++) if (MMt[i] > mx) mx = MMt[i];
    for (long i = 0; i < n * n; i++) MMt[i] = MMt[i] / mx;
    for (long i = 0; i < n; i++) MMt[i + i * n] += 0.95;
}

/* ------------------------------------------------------------------------------------------
 * calculate_a_and_vara_rcpp core on a block of markers held as doubles (row-major rows x n).
 *   v = S*ahat                  (:90 / :192)
 *   a = Mt*v                    (:91 / :193)
 *   W = S*(V*S)                 (:97-98 / :197-198)  -- hoisted by the caller, identical values
 *   T = Mt*W                    (:103 / :204)
 *   vara_i = T.row(i).Mt.row(i) (:110-112 / :214-216)
 * ------------------------------------------------------------------------------------------ */
static int eo_scan_block(const double* Mt, long rows, long n, const double* v, const double* W_rm, double* a_out,
                         double* vara_out) {
    /* a = Mt * v : one dot product per marker, k ascending */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < rows; i++) {
        const double* m = Mt + i * n;
        double s = 0.0;
        for (long k = 0; k < n; k++) s += m[k] * v[k];
        a_out[i] = s;
    }
    /* T = Mt * W in row chunks to bound memory, then the row-dot */
    const long CH = 4096;
    double* T = (double*)malloc(sizeof(double) * (size_t)(rows < CH ? rows : CH) * (size_t)n);
    if (!T) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
    int rc = EO_OK;
    for (long r0 = 0; r0 < rows && rc == EO_OK; r0 += CH) {
        long rr = rows - r0 < CH ? rows - r0 : CH;
        rc = eo_dgemm(rr, n, n, Mt + r0 * n, n, 1, W_rm, n, 1, T, n);
        if (rc) break;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < rr; i++) {
            const double* t = T + i * n;
            const double* m = Mt + (r0 + i) * n;
            double s = 0.0;
            for (long k = 0; k < n; k++) s += t[k] * m[k];
            vara_out[r0 + i] = s;
        }
    }
    free(T);
    return rc;
}

/* v = S*ahat ; W = S*(V*S), W returned row-major.  S, V column-major n x n. */
static int eo_scan_operands(const double* S, const double* V, const double* ahat, long n, double* v, double* W_rm) {
    for (long i = 0; i < n; i++) v[i] = 0.0;
    for (long k = 0; k < n; k++) { /* column-major GEMV, column at a time */
        double ak = ahat[k];
        const double* col = S + k * n;
        for (long i = 0; i < n; i++) v[i] += col[i] * ak;
    }
    double* VS = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    if (!VS) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
    /* VS (row-major) = V * S ; X(i,k) col-major = X[i + k*n] */
    int rc = eo_dgemm(n, n, n, V, 1, n, S, 1, n, VS, n);
    if (rc == EO_OK) rc = eo_dgemm(n, n, n, S, 1, n, VS, n, 1, W_rm, n);
    free(VS);
    return rc;
}

int eo_calculate_a_and_vara(const char* path, const double* sel, long nsel, const double* S, const double* V,
                            double max_memory_in_Gbytes, long L, long n, const double* ahat, double* a_out,
                            double* vara_out, long* branch_out) {
    if (n <= 0 || L <= 0) return eo_fail(EO_ERR_ARG, "bad dims%s", NULL);
    int rc = eo_sel_check(sel, nsel, L);
    if (rc) return rc;
    const int masked = eo_sel_active(sel, nsel);
    /* :65  integer arithmetic, then conversion to double */
    double mem_bytes_needed = (double)((4UL * (unsigned long)n * (unsigned long)L * sizeof(double)) / 1000000000UL);
    double* v = (double*)malloc(sizeof(double) * (size_t)n);
    double* W = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    if (!v || !W) { free(v); free(W); return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL); }
    if (mem_bytes_needed < max_memory_in_Gbytes) { /* :74 */
        if (branch_out) *branch_out = 0;
        double* Mt = (double*)malloc(sizeof(double) * (size_t)L * (size_t)n);
        if (!Mt) { free(v); free(W); return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL); }
        rc = eo_read_rows_rm(path, 0, n, L, Mt); /* :76 */
        if (rc == EO_OK && masked) /* :79-84 */
            for (long ii = 0; ii < nsel; ii++) memset(Mt + (long)sel[ii] * n, 0, sizeof(double) * (size_t)n);
        if (rc == EO_OK) rc = eo_scan_operands(S, V, ahat, n, v, W);
        if (rc == EO_OK) rc = eo_scan_block(Mt, L, n, v, W, a_out, vara_out);
        free(Mt); free(v); free(W);
        return rc;
    }
    /* :129-130 */
    long num_rows_in_block = (long)(max_memory_in_Gbytes * 1000000000.0 / (double)(4UL * (unsigned long)n * sizeof(double)));
    if (num_rows_in_block < 0) { /* :133-144  sentinel List(a=0, vara=0) */
        free(v); free(W);
        a_out[0] = 0.0; vara_out[0] = 0.0;
        eo_fail(EO_SOFT_SENTINEL, "availmemGb: cannot even read in a single row of data into memory%s", NULL);
        return EO_SOFT_SENTINEL;
    }
    if (num_rows_in_block == 0) {
        free(v); free(W);
        return eo_fail(EO_ERR_ARG, "availmemGb too small: zero rows per block (reference divides by zero)%s", NULL);
    }
    if (branch_out) *branch_out = num_rows_in_block;
    long num_blocks = L / num_rows_in_block; /* :150-152 */
    if (L % num_rows_in_block) num_blocks++;
    rc = eo_scan_operands(S, V, ahat, n, v, W); /* :192,:197-198 (recomputed per block there) */
    double* Mt = (double*)malloc(sizeof(double) * (size_t)num_rows_in_block * (size_t)n);
    if (!Mt) rc = eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
    for (long i = 0; i < num_blocks && rc == EO_OK; i++) { /* :157 */
        long s1 = i * num_rows_in_block, r1 = num_rows_in_block;
        if (s1 + r1 > L) r1 = L - s1;
        rc = eo_read_rows_rm(path, s1, n, r1, Mt); /* :165 */
        if (rc) break;
        if (masked) /* :176-190 */
            for (long ii = 0; ii < nsel; ii++)
                if (sel[ii] >= (double)s1 && sel[ii] < (double)(s1 + r1))
                    memset(Mt + ((long)sel[ii] - s1) * n, 0, sizeof(double) * (size_t)n);
        rc = eo_scan_block(Mt, r1, n, v, W, a_out + s1, vara_out + s1); /* :192-225 */
    }
    free(Mt); free(v); free(W);
    return rc;
}

/* In-memory branch on an int8 marker-major matrix (L x n, row stride ld) -- CPU-baseline leg.
 * Converts to doubles first (what ReadBlock yields), then the same operation order. */
int eo_scan_from_i8(const int8_t* Mt8, long L, long n, long ld, const double* S, const double* V, const double* ahat,
                    double* a_out, double* vara_out) {
    double* v = (double*)malloc(sizeof(double) * (size_t)n);
    double* W = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double* Mt = (double*)malloc(sizeof(double) * (size_t)L * (size_t)n);
    if (!v || !W || !Mt) { free(v); free(W); free(Mt); return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL); }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < L; r++)
        for (long c = 0; c < n; c++) Mt[r * n + c] = (double)Mt8[r * ld + c];
    int rc = eo_scan_operands(S, V, ahat, n, v, W);
    if (rc == EO_OK) rc = eo_scan_block(Mt, L, n, v, W, a_out, vara_out);
    free(v); free(W); free(Mt);
    return rc;
}

/* Same, with W and v supplied (lets the baseline time the marker-dependent part alone). */
int eo_scan_from_i8_with_W(const int8_t* Mt8, long L, long n, long ld, const double* v, const double* W_rm,
                           double* a_out, double* vara_out) {
    double* Mt = (double*)malloc(sizeof(double) * (size_t)L * (size_t)n);
    if (!Mt) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < L; r++)
        for (long c = 0; c < n; c++) Mt[r * n + c] = (double)Mt8[r * ld + c];
    int rc = eo_scan_block(Mt, L, n, v, W_rm, a_out, vara_out);
    free(Mt);
    return rc;
}
int eo_scan_operands_pub(const double* S, const double* V, const double* ahat, long n, double* v, double* W_rm) {
    return eo_scan_operands(S, V, ahat, n, v, W_rm);
}

/* ------------------------------------------------------------------------------------------
 * calculate_reduced_a_rcpp  (E/src/calculate_reduced_a_rcpp.cpp:20-171); dims = (n, L) of M,
 * file = Mt.ascii (L lines of n chars).  mem_bytes_needed multiplies by the integer
 * sizeof(double)/1000000000 == 0 (:56), so the in-memory branch (:65-85) runs whenever
 * max_memory_in_Gbytes > 0; otherwise the block branch computes a negative row count and
 * returns the 1x1 zero sentinel (:92-103).
 * ------------------------------------------------------------------------------------------ */
int eo_calculate_reduced_a(const char* path, double varG, const double* P, const double* y,
                           double max_memory_in_Gbytes, long n, long L, const double* sel, long nsel,
                           double* ar_out) {
    if (n <= 0 || L <= 0) return eo_fail(EO_ERR_ARG, "bad dims%s", NULL);
    int rc = eo_sel_check(sel, nsel, L);
    if (rc) return rc;
    if (!(0.0 < max_memory_in_Gbytes)) {
        ar_out[0] = 0.0;
        eo_fail(EO_SOFT_SENTINEL, "availmemGb: cannot even read in a single row of data into memory%s", NULL);
        return EO_SOFT_SENTINEL;
    }
    double* Mt = (double*)malloc(sizeof(double) * (size_t)L * (size_t)n);
    double* py = (double*)calloc((size_t)n, sizeof(double));
    if (!Mt || !py) { free(Mt); free(py); return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL); }
    rc = eo_read_rows_rm(path, 0, n, L, Mt); /* :71 */
    if (rc == EO_OK) {
        if (eo_sel_active(sel, nsel)) /* :74-78 */
            for (long ii = 0; ii < nsel; ii++) memset(Mt + (long)sel[ii] * n, 0, sizeof(double) * (size_t)n);
        for (long k = 0; k < n; k++) { /* :82  ar = P * y */
            double yk = y[k];
            const double* col = P + k * n;
            for (long i = 0; i < n; i++) py[i] += col[i] * yk;
        }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < L; i++) { /* :83-84 */
            const double* m = Mt + i * n;
            double s = 0.0;
            for (long k = 0; k < n; k++) s += m[k] * py[k];
            ar_out[i] = varG * s;
        }
    }
    free(Mt); free(py);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * E/R/find_qtl.R:71-83   tsq <- a^2/vara ; indx <- which(tsq == max(tsq, na.rm=TRUE))[1]
 * Returns the 1-based index (0 if every tsq is NaN).  NaN (0/0 at masked rows) is skipped.
 * ------------------------------------------------------------------------------------------ */
long eo_tsq_argmax(const double* a, const double* vara, long L, double* tsq_out, double* max_out) {
    double mx = -INFINITY;
    long idx = 0;
    for (long i = 0; i < L; i++) {
        double t = (a[i] * a[i]) / vara[i];
        if (tsq_out) tsq_out[i] = t;
        if (!isnan(t) && (idx == 0 || t > mx)) { mx = t; idx = i + 1; }
    }
    if (max_out) *max_out = idx ? mx : NAN;
    return idx;
}

/* E/src/extract_geno_rcpp.cpp:16-89: column `selected_locus` (0-based) of M.ascii as ints; both of its branches
 * (whole file / row blocks) yield genoMat(i, selected_locus) for every line i. */
int eo_extract_geno(const char* path, long selected_locus, long n, long L, int* out) {
    if (selected_locus < 0 || selected_locus >= L) return eo_fail(EO_ERR_ARG, "bad locus%s", NULL);
    double* row = (double*)malloc(sizeof(double) * (size_t)L);
    if (!row) return eo_fail(EO_ERR_NOMEM, "out of memory%s", NULL);
    FILE* f = fopen(path, "r");
    if (!f) { free(row); return eo_fail(EO_ERR_OPEN, "ERROR: Could not open  %s", path); }
    char* line = NULL;
    size_t cap = 0;
    int rc = EO_OK;
    for (long r = 0; r < n; r++) {
        ssize_t len = getline(&line, &cap, f);
        if (len < 0 || len <= selected_locus) { rc = eo_fail(EO_ERR_SHORT, "file %s shorter than requested", path); break; }
        out[r] = (int)((double)(line[selected_locus] - '0') - 1);
    }
    free(line); free(row); fclose(f);
    return rc;
}

int eo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void eo_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#endif
}
