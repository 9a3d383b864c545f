/*
 * eagle_hip.h -- C ABI of libeaglehip.so, the MI355X (gfx950) backend for the Eagle/WMAM hot path.
 *
 * The drop-in boundary of the reference is the .Call table of Eagle.so
 * (E/src/RcppExports.cpp:154-170, E/ = MyPackage/Eagle/).  The functions of section 1 below are what the
 * bodies of the reference's exported C++ functions are replaced with; each one cites the interface it
 * replaces.  The R-side binding (plain R C API, no Rcpp) is shown in INTEGRATION.md and in
 * eagleeverything_amd/shim/ (one Rcpp-typed translation unit per exported function + eagle_backend.h).
 *
 * Conventions
 *   - plain pointers and sizes only; matrices crossing section 1 are COLUMN-major doubles (R / Eigen).
 *   - every function returns an int status: 0 ok, < 0 hard error (text via eagle_last_error), 1 = the
 *     reference's "soft" sentinel return (message printed, placeholder value written, see each function).
 *   - selected_loci is passed as the raw R doubles (NA = NaN).  Masking fires iff element 0 is not NA
 *     (calculateMMt_rcpp.cpp:88, calculate_a_and_vara_rcpp.cpp:79, calculate_reduced_a_rcpp.cpp:74).
 *   - caller owns every pointer it passes; nothing is retained past return except the HBM-resident
 *     genotype cache inside the ctx (keyed by path, size, mtime), freed by eagle_close / eagle_drop_cache.
 *   - no function throws; none calls back except through the message callback, on the calling thread.
 *   - there is NO CPU fallback: without a gfx950 device eagle_open fails.
 */
#ifndef EAGLE_HIP_H
#define EAGLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAGLE_OK 0
#define EAGLE_SOFT_SENTINEL 1
#define EAGLE_ERR_OPEN (-1)    /* ReadBlock.cpp:42-45: "ERROR: Could not open <file>" */
#define EAGLE_ERR_FORMAT (-2)  /* short file / short line / character outside '0'..'2' */
#define EAGLE_ERR_ARG (-3)
#define EAGLE_ERR_NOMEM (-4)
#define EAGLE_ERR_HIP (-5)
#define EAGLE_ERR_NODEVICE (-6)

typedef struct eagle_ctx eagle_ctx;

/* Receives what the reference sends through its `message` R closure argument
 * (calculateMMt_rcpp.cpp:36,107; calculate_a_and_vara_rcpp.cpp:69,124). */
typedef void (*eagle_message_fn)(const char* text, void* user);

/* ---------------------------------------------------------------------------------------------
 * 0. Context
 * ------------------------------------------------------------------------------------------- */
/* Opens HIP device `device` (must be gfx950).  Returns NULL on failure; the reason is available from
 * eagle_open_error().  One ctx per device; a process that drives several GPUs opens one ctx each. */
eagle_ctx* eagle_open(int device);
/* Several GPUs of one node behind ONE context -- the reference's unused hook is AM(..., ngpu) (E/R/AM.R:185-196, forced to 0 at
 * :214, handed to .find_qtl at :450-455).  eagle_calculateMMt / eagle_calculate_a_and_vara / eagle_calculate_reduced_a then
 * split the file's markers into ndev contiguous ranges (boundaries at multiples of 256 markers), one host thread + stream
 * per device, all joined before the call returns; results land in the caller's arrays exactly as with one device
 * (bit-identical: integer MM^T, per-marker scan, and a certification that exchanges the shards' bounds).  Exchange steps:
 * one int32 sum of the partial MM^T to the first device (ncclReduce over xGMI), the rows of W = S V S shared and
 * all-gathered (ncclAllGather), 8 bytes per device through the host for the certificate.  RCCL is dlopen()ed on first use;
 * a list that names one device twice (the one-GPU test configuration) or EAGLE_HIP_COLLECTIVES=host stages the sum through
 * device copies and computes W on every device instead.
 * eagle_open_env: the devices of EAGLE_HIP_DEVICES="0,1,..." if set, else EAGLE_HIP_DEVICE, else device 0. */
eagle_ctx* eagle_open_devices(const int* devices, int ndev);
eagle_ctx* eagle_open_env(void);
int eagle_device_count(eagle_ctx* ctx);
const char* eagle_open_error(void);
void eagle_close(eagle_ctx* ctx);
const char* eagle_last_error(eagle_ctx* ctx);
void eagle_set_message_callback(eagle_ctx* ctx, eagle_message_fn fn, void* user);
/* Drops HBM-resident genotype copies kept between calls. */
/* (round 4: also the grow-only workspaces held between calls -- the scan arena, a background reservation of it, the workspace of
 * the int8 W products; the next call allocates them again) */
void eagle_drop_cache(eagle_ctx* ctx);
/* "gfx950", CU count, HBM bytes -- for logs and bench JSON. */
int eagle_device_info(eagle_ctx* ctx, char* arch_out, int arch_len, int* cu_count, int64_t* hbm_bytes);
/* vara kernel: 1 (default) = exact int8 digit slices of W on v_mfma_i32_32x32x32_i8, 0 = fp64 MFMA
 * (v_mfma_f64_16x16x4_f64; also taken automatically when n is too large for the int32 tile sums).
 * eagle_set_scan_slices: S = 1..8 base-256 digits of the off-diagonal part of W, or 0 (default) = chosen per call:
 * the smallest S in 3..7 whose worst-case bound is below the budget (eagle_set_scan_budget, default 5e-7) times 0.5 * sum_k |W_kk| -- half
 * of the 1e-6 relative tolerance of this path, relative to the smallest diagonal term a marker can have (at least half of its genotypes are
 * non-zero in the g-1 coding); measured errors are ~1000x below the bound.  The diagonal term
 * sum_k m_ik^2 W_kk is evaluated in fp64; every vara_i then differs from the exact m_i^T W m_i by at most
 * (sum_j |m_ij|)^2 * 2^(e+1-8S), max_{j!=k} |W_jk + W_kj| < 2^e (or < 1.96 * 2^e: the exponent is lowered by one when the largest
 * entry's mantissa leaves the balanced digits room), plus fp64 rounding of an n-term and an S-term sum. */
int eagle_set_scan_mode(eagle_ctx* ctx, int mode);
int eagle_set_scan_slices(eagle_ctx* ctx, int nslices);
/* Round 3.  The relative digit budget of the int8 scan, 1e-12 .. 5e-7 (default 5e-7 = half of the path's 1e-6 tolerance; rounds 1-2
 * used 1e-7, which this call restores).  It enters twice: the automatic digit count is the smallest S whose bound keeps a typical
 * marker inside the budget, and the certificate sends every marker whose OWN bound exceeds 1.8 x budget (0.9e-6 by default) to the
 * fp64 kernel.  With the automatic count the library also tries ONE DIGIT FEWER than the worst-case bound asks for, under a spectral
 * bound of the truncation error: |error_i| <= (u/2)(||Ds||_2 + (n_pad-1)/2) sum_j m'_ij^2, Ds = the symmetrised last digit of W,
 * ||Ds||_2^2 <= max row sum of |Ds Ds| from an exact int8 MFMA Gram product, or, one level down, max_j (Ds Ds)_jj + ||offdiag(Ds Ds)||_2
 * with the off-diagonal part bounded the same way (csrc/eagle_i8mfma.hip, k_spectral_decide) -- rigorous, deterministic, a function of
 * W alone.  At n = 10,000 this takes the scan of an AM() run from 4 digit slices to 3, with the default budget and with 1e-7 alike. */
int eagle_set_scan_budget(eagle_ctx* ctx, double relative_budget);
/* How the digits of W are rounded (digit-slice mode).  0 (default): to nearest -- every vara_i is GUARANTEED within
 * l1_i^2 / 2 * 2^(e+1-8S) of the exact quadratic form (l1_i = sum_j |m'_ij| of the re-centred marker).  1: stochastic
 * rounding -- each entry rounded down or up at random with the probabilities that make it unbiased, from a counter-based
 * generator keyed by the entry's position (independent of the data, reproducible).  The truncation errors are then
 * independent and zero-mean, Hoeffding's inequality bounds marker i's error by 8.355 q2_i 2^(e+1-8S), q2_i = sum_j m'_ij^2,
 * with failure probability below 1e-30 per marker (1e-22 over every marker of every scan of an AM() run), and the automatic
 * digit count drops by one (C2 / C3: S = 3 instead of 4, a quarter less matrix work).  The certificate is then a
 * probabilistic one; flagging, fp64 re-evaluation and the selected marker work exactly as in mode 0.  In the
 * device-resident entry points the same switch is bit 8 of the `nslices` argument (EAGLE_SLICES_STOCHASTIC). */
#define EAGLE_SLICES_STOCHASTIC 0x100
int eagle_set_scan_rounding(eagle_ctx* ctx, int stochastic);
/* Round 4.  Which engine forms W = S (V S) for a digit-slice scan (calculate_a_and_vara_rcpp.cpp:97-98).  1 (default): from 4,096
 * padded individuals up, the two n^3 products run on the int8 MFMA from exact base-256 digit slices of the off-diagonal parts of S, V
 * and X = V S (diagonal parts exactly, in fp64), under a rigorous Frobenius-norm bound eta of the error of the W delivered; the scan's
 * per-marker certificate adds eta * sum_j m'_ij^2 to its bound, the correction vector of the re-centred markers comes from
 * r = S (V (S 1)) in fp64, and the markers the certificate re-evaluates are computed as m^T (S (V (S m))) in fp64.  A call whose
 * operands the configurations on offer cannot certify to 2 % of the digit budget (wild scaling, cancelling V, non-finite or visibly
 * asymmetric matrices, no workspace) DECLINES and the fp64 GEMM runs as before.  0: always the fp64 GEMM.  2: the int8 engine at any
 * size (tests).  eagle_set_scan_mode(0) never uses it.  csrc/eagle_w8.hip. */
int eagle_set_w_mode(eagle_ctx* ctx, int mode);
typedef struct eagle_w_info {
    int int8;            /* 1: the W of the last scan_operands call came from the int8 engine */
    int declined;        /* else why not: 1 non-finite, 2 asymmetric / no yardstick, 3 / 4 no configuration for V S / S X, 5 no workspace,
                            6 the finished W failed the check against its own diagonal, 7 switched off or too small, 8 replaced by the fp64
                            products after the certificate overflowed */
    int k1, T1, pairs1;  /* V S: digits per operand, highest level p + q, int8 products (n_pad^3 MACs each) */
    int k2, T2, pairs2;  /* S X (upper triangle only) */
    double eta;          /* || folded W - truth ||_F bound; a marker's vara carries eta * sum_j m'_j^2 of it */
    double eta_x;        /* || X computed - V S ||_F bound */
    double target;       /* what the configurations were chosen for (0.02 x budget x estimate of mean |W_kk|) */
    double mean_diag;    /* mean |W_kk| of the W delivered */
    double asym_term;    /* share of eta that pays for || S - S^T ||_F, || V - V^T ||_F (measured) */
    int pipelined;       /* 1: eagle_calculate_a_and_vara formed V S column block by column block while V was still arriving over PCIe (on the
                            configuration of the context's last call, which the rule then confirmed: the bits are the same either way) */
    int pad;
} eagle_w_info;
int eagle_last_w_info(eagle_ctx* ctx, eagle_w_info* out);

/* ---------------------------------------------------------------------------------------------
 * 1. Reference-shaped entry points (host pointers, files on disk)
 * ------------------------------------------------------------------------------------------- */

/* Replaces  Eigen::MatrixXd ReadBlock(std::string asciifname, long start_row, long numcols,
 *           long numrows_in_block)               E/src/ReadBlock.cpp:16-68, RcppExports.cpp:9-21
 * out: numrows_in_block x numcols column-major, value (char-'0')-1.  The text tile is decoded on the
 * device (raw bytes -> HBM -> int8 -> fp64) and copied back. */
int eagle_read_block(eagle_ctx* ctx, const char* asciifname, long start_row, long numcols,
                     long numrows_in_block, double* out);

/* Replaces  Eigen::MatrixXd calculateMMt_rcpp(CharacterVector f_name_ascii, double max_memory_in_Gbytes,
 *           int num_cores, NumericVector selected_loci, std::vector<long> dims, bool quiet,
 *           Function message)                    E/src/calculateMMt_rcpp.cpp:19-185, RcppExports.cpp:37-52
 * dims = (n, L) of M.ascii.  MMt_out: n x n column-major.  max_memory_in_Gbytes bounds HOST staging
 * (the genotype tiles are streamed through pinned memory); num_cores bounds host reader threads.
 * A file that does not fit in HBM (or exceeds the environment variable EAGLE_HIP_MAX_RESIDENT_GB) is streamed in
 * marker windows and the exact integer partial products are accumulated; the same holds for the two scans below. */
int eagle_calculateMMt(eagle_ctx* ctx, const char* f_name_ascii, double max_memory_in_Gbytes, int num_cores,
                       const double* selected_loci, long n_selected, const long dims[2], int quiet,
                       double* MMt_out);

/* Replaces  Rcpp::List calculate_a_and_vara_rcpp(CharacterVector f_name_ascii, NumericVector selected_loci,
 *           Map<MatrixXd> inv_MMt_sqrt, Map<MatrixXd> dim_reduced_vara, double max_memory_in_Gbytes,
 *           std::vector<long> dims, VectorXd a, bool quiet, Function message)
 *                                                E/src/calculate_a_and_vara_rcpp.cpp:22-241, RcppExports.cpp:54-71
 * dims = (L, n) of Mt.ascii.  inv_MMt_sqrt, dim_reduced_vara: n x n column-major; a: n.
 * a_out, vara_out: L doubles each (the "a" and "vara" L x 1 matrices of the returned list).
 * Returns EAGLE_SOFT_SENTINEL with a_out[0] = vara_out[0] = 0 where the reference returns
 * List(a = 0, vara = 0) (:133-144). */
int eagle_calculate_a_and_vara(eagle_ctx* ctx, const char* f_name_ascii, const double* selected_loci,
                               long n_selected, const double* inv_MMt_sqrt, const double* dim_reduced_vara,
                               double max_memory_in_Gbytes, const long dims[2], const double* a, int quiet,
                               double* a_out, double* vara_out);

/* Replaces  Eigen::MatrixXd calculate_reduced_a_rcpp(CharacterVector f_name_ascii, double varG,
 *           Map<MatrixXd> P, Map<MatrixXd> y, double max_memory_in_Gbytes, std::vector<long> dims,
 *           NumericVector selected_loci, bool quiet, Function message)
 *                                                E/src/calculate_reduced_a_rcpp.cpp:20-171, RcppExports.cpp:73-90
 * dims = (n, L) of M; the file is Mt.ascii.  ar_out: L doubles.  Returns EAGLE_SOFT_SENTINEL with
 * ar_out[0] = 0 where the reference returns its 1 x 1 zero matrix (:94-103). */
int eagle_calculate_reduced_a(eagle_ctx* ctx, const char* f_name_ascii, double varG, const double* P,
                              const double* y, double max_memory_in_Gbytes, const long dims[2],
                              const double* selected_loci, long n_selected, int quiet, double* ar_out);

/* Replaces  Eigen::VectorXi extract_geno_rcpp(CharacterVector f_name_ascii, double max_memory_in_Gbytes,
 *           long selected_locus, std::vector<long> dims)   E/src/extract_geno_rcpp.cpp:16-89, RcppExports.cpp:128-141
 * Column `selected_locus` (0-based) of M.ascii as ints -1/0/1 (constructX appends it to the design matrix,
 * E/R/constructX.R:16-19).  dims = (n, L).  Served from the HBM-resident copy of the file when eagle_calculateMMt
 * has loaded it (always the case inside AM(), AM.R:403 vs :414-417); otherwise one character per line is read from
 * the file -- the reference parses the whole n x L file into doubles for this. */
int eagle_extract_geno(eagle_ctx* ctx, const char* f_name_ascii, double max_memory_in_Gbytes, long selected_locus,
                       const long dims[2], int* column_out);

/* ---------------------------------------------------------------------------------------------
 * 1b. Marker-file ingestion (the producers of M.ascii / Mt.ascii that ReadMarker() calls, E/R/create_ascii.R:27-58).
 *     Same results byte for byte; additionally the int8 images of both files stay resident in HBM under the OUTPUT
 *     paths, so the eagle_calculateMMt / eagle_calculate_a_and_vara calls that follow never parse text, and a 2-bit
 *     sidecar "<output>.e2b" (64-byte header + packed rows, a quarter of the text bytes) is left beside each text file:
 *     every loader of this library reads it instead of the text while the text file keeps the size and mtime recorded in
 *     it (EAGLE_HIP_SIDECAR=0 disables writing and reading).
 * ------------------------------------------------------------------------------------------- */

/* Replaces  std::vector<long> getRowColumn(std::string fname)      E/src/getRowColumn.cpp:20-72, RcppExports.cpp:143-151
 * dims_out = (lines, whitespace-separated tokens of the first line). */
int eagle_get_row_column(eagle_ctx* ctx, const char* fname, long dims_out[2]);

/* Replaces  bool createM_ASCII_rcpp(CharacterVector f_name, CharacterVector f_name_ascii, CharacterVector type,
 *           std::string AA, std::string AB, std::string BB, double max_memory_in_Gbytes, std::vector<long> dims,
 *           bool quiet, Function message, std::string missing)
 *                                                E/src/createM_ASCII_rcpp.cpp:18-106, RcppExports.cpp:92-111
 * type "PLINK": f_name is a ped file, dims = (individuals, 6 + 2*loci), AA/AB/BB/missing unused
 *               (E/src/CreateASCIInospace_PLINK.cpp:16-249: first-seen allele order, 0 / - = missing -> het);
 * otherwise   : a whitespace-separated genotype table, dims = (individuals, loci), tokens BB -> '2', AB -> '1',
 *               AA -> '0', missing -> '1'  (E/src/CreateASCIInospace.cpp:84-105).
 * Returns EAGLE_OK where the reference returns true, EAGLE_SOFT_SENTINEL where it prints its messages and returns
 * false (unknown token, unequal number of columns, third allele; eagle_last_error holds a one-line summary and the
 * output file keeps the rows converted before the failure, like the reference's). */
int eagle_create_M_ascii(eagle_ctx* ctx, const char* f_name, const char* f_name_ascii, const char* type, const char* AA,
                         const char* AB, const char* BB, double max_memory_in_Gbytes, const long dims[2], int quiet,
                         const char* missing);

/* Replaces  void createMt_ASCII_rcpp(CharacterVector f_name, CharacterVector f_name_ascii, CharacterVector type,
 *           double max_memory_in_Gbytes, std::vector<long> dims, bool quiet, Function message)
 *                                                E/src/createMt_ASCII_rcpp.cpp:14-247, RcppExports.cpp:113-126
 * f_name = M.ascii (dims = (n, L)), f_name_ascii = Mt.ascii to write (L lines of n characters).  The transpose runs on
 * the device from the resident image of M.ascii (loaded if it is not there; column windows if it does not fit). */
int eagle_create_Mt_ascii(eagle_ctx* ctx, const char* f_name, const char* f_name_ascii, const char* type,
                          double max_memory_in_Gbytes, const long dims[2], int quiet);

/* ---------------------------------------------------------------------------------------------
 * 1c. Dense n x n model algebra on the device (OPT-IN: north_star keeps calculateH / calculateP / emma.* on
 *     host LAPACK, and nothing above calls these).  Once the scan takes tens of milliseconds the ~10-15 O(n^3) base-R calls
 *     of a find_qtl iteration are the whole run time (the author's note MyPackage/MyREADME:1 names eigen(); his MAGMA
 *     attempt is E/R/emma_eigen_R_wo_Z.R:9-15).  Column-major host matrices in and out, as R holds them.  The
 *     factorisations are rocSOLVER library calls (dlopen()ed on first use), the products this library's fp64 MFMA GEMM.
 * ------------------------------------------------------------------------------------------- */
/* eigen(A, symmetric = TRUE): values in DEcreasing order as R returns them, the matching eigenvectors in the columns of
 * vectors_out (NULL: only.values = TRUE).  Used by E/R/emma_eigen_L_wo_Z.R:3, emma_eigen_R_wo_Z.R:17, calculateMMt_sqrt_and_sqrtinv.R:25.
 * Only the LOWER triangle of the column-major matrix is read (as R's eigen(symmetric = TRUE) and LAPACK uplo 'L' do): for an input
 * that is symmetric only to rounding, a row-major buffer handed over as its transpose has its UPPER triangle read instead and the
 * result differs at rounding level.  Must not run under `rocprofv3 --pmc` (rocSOLVER aborts under counter collection,
 * profiles/r02_eigh_under_pmc.log). */
int eagle_sym_eig(eagle_ctx* ctx, const double* A, long n, double* values_out, double* vectors_out);
/* chol2inv(chol(A)) (calculateMMt_sqrt_and_sqrtinv.R:30, calculateP.R:27, calculate_reduced_vara.R:27).  Returns
 * EAGLE_SOFT_SENTINEL when A is not positive definite (R's chol() error text in eagle_last_error).  Only the UPPER triangle of
 * the column-major matrix is read (as R's chol() does); a row-major buffer handed over as its transpose has its lower triangle read. */
int eagle_chol2inv(eagle_ctx* ctx, const double* A, long n, double* Ainv_out);
/* solve(A) (calculate_reduced_vara.R:31-33, calculateP.R:28).  EAGLE_SOFT_SENTINEL for a singular matrix. */
int eagle_inverse(eagle_ctx* ctx, const double* A, long n, double* Ainv_out);
/* C (m x n) = A (m x k) %*% B (k x n) on the fp64 MFMA GEMM of the scan operands (v_mfma_f64_16x16x4_f64). */
int eagle_matmul(eagle_ctx* ctx, const double* A, const double* B, long m, long k, long n, double* C);
/* E/R/calculateMMt_sqrt_and_sqrtinv.R:15-47 in one call: EAGLE_SOFT_SENTINEL if MMt is not positive definite by
 * matrixcalc::is.positive.definite's rule (:15; eigenvalues below 1e-8 in magnitude count as 0); sqrt_out = U sqrt(L) U^T
 * (:25-27), invsqrt_out = chol2inv(chol(sqrt)) (:30), *trace_out = sum(diag(sqrt %*% invsqrt)), whose truncation the R code
 * compares with nrow(MMt) (:35-46; may be NULL). */
int eagle_mmt_sqrt_and_sqrtinv(eagle_ctx* ctx, const double* MMt, long n, double* sqrt_out, double* invsqrt_out, double* trace_out);

/* Optional shortcut (NOT one of the reference's .Call symbols; a maintainer who edits find_qtl.R may use it): the scan of
 * eagle_calculate_a_and_vara with W = S V S (n x n) and v = S a_hat (n) handed over ready-made.  Inside AM() neither needs an
 * n^3 product: dim_reduced_vara = varG I - C22 = varG^2 Ze P Ze with Ze = MMt^1/2 (E/R/calculate_reduced_vara.R:21-35),
 * inv_MMt_sqrt = Ze^-1, hence W = varG^2 P and v = varG P y, and .find_qtl holds P (E/R/find_qtl.R:9, calculateP.R:27-28).
 * Same outputs, sentinel-free (no availmemGb branch rules: those belong to the reference-shaped call). */
int eagle_scan_with_W(eagle_ctx* ctx, const char* f_name_ascii, const double* selected_loci, long n_selected, const double* W,
                      const double* v, double max_memory_in_Gbytes, const long dims[2], int quiet, double* a_out, double* vara_out);

/* ---------------------------------------------------------------------------------------------
 * 1d. The scan in the eigenbasis of MM^T (OPT-IN; not symbols of the reference -- a maintainer who edits find_qtl.R may use
 *     them; the reference-shaped eagle_calculate_a_and_vara above stays the drop-in).
 *     Per iteration the reference evaluates a_i = varG m_i^T P y and vara_i = varG^2 m_i^T P m_i (W = S V S = varG^2 P, see
 *     eagle_scan_with_W) with P = H^-1 - H^-1 X (X^T H^-1 X)^-1 X^T H^-1, H = varE I + varG K, K = the normalised MM^T of
 *     calcMMt.R:13 -- an n x n quadratic form per marker.  K does not change during an AM() run (only varE, varG, X do), so with
 *     K = U diag(lambda) U^T (emma.REMLE computes it anyway, E/R/emma_eigen_R_wo_Z.R:17) and Z = Mt U made ONCE,
 *     every scan is one streaming pass over Z: 8 n bytes and (p + 2) n multiply-adds per marker, HBM-bound.
 * ------------------------------------------------------------------------------------------- */
/* Once per AM() run: Z = Mt U (L x n fp64, 8 bytes per genotype, kept in HBM by the ctx) from the resident int8 image of
 * Mt.ascii (dims = (L, n)) and the eigenvectors U of K (n x n column-major, any order, the same order as lambda below). */
int eagle_spectral_prepare(eagle_ctx* ctx, const char* f_name_ascii, const long dims[2], const double* U, double max_memory_in_Gbytes);
/* Per iteration: lambda (n eigenvalues of K), UtX = U^T X (n x p column-major, p = columns of the fixed-effects design,
 * 1 <= p <= 31), Uty = U^T y (n), the variance components.  a_out, vara_out: L doubles, the values
 * eagle_calculate_a_and_vara returns for the S, V, a_hat that find_qtl.R:5-49 builds from the same K, X, y, varE, varG.
 * selected_loci: the reference's masking rule (element 0 NA: none). */
int eagle_spectral_scan(eagle_ctx* ctx, const double* lambda, const double* UtX, const double* Uty, long p, double varE, double varG,
                        const double* selected_loci, long n_selected, double* a_out, double* vara_out);

/* Replaces the R tail of .find_qtl:  tsq <- a^2/vara ; which(tsq == max(tsq, na.rm=TRUE))[1]
 *                                                E/R/find_qtl.R:71-83
 * Evaluated on the device on the a / vara of the LAST eagle_calculate_a_and_vara call of this ctx (still in
 * HBM).  index_out is 1-based (0 if every tsq is NaN).  In digit-slice mode the arrays were certified inside the scan call
 * (eagle_dev_scan_certify below: every marker that could be the arg-max carries its fp64-kernel value), so the index is
 * the one scan mode 0 returns and the one R's own which(tsq == max(tsq))[1] finds on the returned a / vara.
 * n_near_ties is informational: markers whose tsq is within a relative 1e-9 of the maximum (1 = unambiguous). */
int eagle_last_scan_argmax(eagle_ctx* ctx, long* index_out, double* tsqmax_out, long* n_near_ties);

/* MMt/max(MMt) + 0.95 I  (E/R/calcMMt.R:13) of the LAST eagle_calculateMMt result, on the device. */
int eagle_last_mmt_normalised(eagle_ctx* ctx, double* MMt_norm_out, double* max_out);

/* ---------------------------------------------------------------------------------------------
 * 2. Device-resident entry points (all pointers are HBM addresses on ctx's device; `stream` is a
 *    hipStream_t passed as void*, NULL = default stream).  Used by the marker-sharded multi-GPU driver
 *    and by bench.py, which keep genotype shards resident in HBM between calls.
 *
 *    Layout contract for genotype matrices (int8, values {-1,0,1}):
 *      Mt8: marker-major  [L_pad][ld]  ld >= n, ld % 256 == 0, L_pad % 256 == 0, padding bytes ZERO
 *      M8 : individual-major [n_pad][ld] ld >= L, ld % 256 == 0, n_pad % 256 == 0, padding bytes ZERO
 *    fp64 square operands: row-major [np][np], np = eagle_pad(n) (next multiple of 256), padding ZERO.
 * ------------------------------------------------------------------------------------------- */
long eagle_pad(long x);

/* Lines [row0,row0+nrows) x characters [col0,col0+ncols) of a no-space ASCII genotype file (M.ascii / Mt.ascii,
 * as ReadBlock reads them, E/src/ReadBlock.cpp:47-58) -> int8 at dst[r*ld + c] in HBM.  Fixed-width files are
 * pread() into pinned memory by `threads` workers, double-buffered against the H2D copy and the decode kernel;
 * max_mem_gb bounds the pinned staging.  A marker shard is a row range of Mt.ascii or a column window of M.ascii. */
int eagle_dev_load_ascii(eagle_ctx* ctx, const char* path, long row0, long nrows, long col0, long ncols, int8_t* dst,
                         long ld, double max_mem_gb, int threads);

/* raw text tile (rows of `line_stride` bytes, first `cols` bytes used) -> int8 (value char-'0'-1);
 * *bad_chars_dev (device int, caller-zeroed) counts bytes outside '0'..'2'. */
int eagle_dev_decode_ascii(eagle_ctx* ctx, const uint8_t* raw, long rows, long cols, long line_stride,
                           int8_t* out, long ld_out, int* bad_chars_dev, void* stream);
int eagle_dev_transpose_i8(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, int8_t* out,
                           long ld_out, void* stream);
int eagle_dev_i8_to_f64_colmajor(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in,
                                 double* out_colmajor, void* stream);

/* C32[np][np] (int32, row-major, caller-zeroed) += M8 * M8^T over the marker columns [0, L_pad).
 * Upper-triangular tiles only; exact integers, order independent (integer atomics).  The products run on
 * v_mfma_scale_f32_32x32x64_f8f6f4 with both operands fp4 (-1, 0, +1 are exact e2m1 numbers; fp32 partial sums of a K
 * split are exact integers): eagle_dev_mmt_accumulate packs the int8 image into a ctx-owned fp4 buffer first,
 * eagle_dev_mmt_accumulate_f4 takes an image the caller made with eagle_dev_pack_fp4 (M4[n_pad][ld4 bytes], two
 * genotypes per byte).  n_pad % 256 == 0, L_pad % 256 == 0. */
int eagle_dev_mmt_accumulate(eagle_ctx* ctx, const int8_t* M8, long n_pad, long L_pad, long ld, int32_t* C32,
                             void* stream);
int eagle_dev_mmt_accumulate_f4(eagle_ctx* ctx, const void* M4, long n_pad, long L_pad, long ld4, int32_t* C32, void* stream);
/* C32 -= sum over the listed marker columns of m_c m_c^T  (the selected_loci masking of
 * calculateMMt_rcpp.cpp:88-92 applied as an exact rank-k downdate). cols_dev: device array of long. */
int eagle_dev_mmt_downdate(eagle_ctx* ctx, const int8_t* M8, long n_pad, long ld, const long* cols_dev,
                           long ncols, int32_t* C32, void* stream);
/* Mirror the upper triangle, convert to fp64 (n x n, leading dimension ld_out, symmetric so layout-free),
 * and write max(MMt) to *max_dev (device double). */
int eagle_dev_mmt_finish(eagle_ctx* ctx, const int32_t* C32, long n, long n_pad, double* MMt, long ld_out,
                         double* max_dev, void* stream);
int eagle_dev_mmt_normalise(eagle_ctx* ctx, double* MMt, long n, long ld, const double* max_dev, void* stream);

/* Scan operands from the reference's arguments (n_pad x n_pad row-major images of the COLUMN-major
 * inputs, i.e. Sa = S^T, Va = V^T):  v = S a_hat ; Wu = upper-triangular fold of W = S (V S) with
 * Wu[j][k] = W[j][k] + W[k][j] (j<k), W[k][k] (j=k), 0 (j>k), so that m^T W m = sum_k m_k sum_{j<=k} m_j Wu[j][k].
 * tmp: n_pad*n_pad doubles of scratch. */
int eagle_dev_scan_operands(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n,
                            long n_pad, double* v_out, double* Wu_out, double* tmp, void* stream);
/* Marker-sharded runs share the n^3 part: each rank computes v and the rows [row0, row1) (multiples of 128) of the W^T
 * image (full rows), the row blocks are all-gathered, and eagle_dev_fold_upper folds the complete image in place. */
int eagle_dev_scan_operands_rows(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long n_pad,
                                 long row0, long row1, double* v_out, double* Wt_out, double* tmp, void* stream);
int eagle_dev_fold_upper(eagle_ctx* ctx, double* W, long n_pad, void* stream);
/* a_i = scale * sum_j Mt8[i][j] v[j] for L_pad rows: calculate_a_and_vara_rcpp.cpp:91, calculate_reduced_a_rcpp.cpp:83-84.
 * HBM-bound pass on the int8 MFMA (v as 8 exact base-256 digit rows, int32 sums, one rounding per output).
 * L_pad % 16 == 0, n_pad % 256 == 0, ld % 16 == 0. */
int eagle_dev_gemv_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v,
                      double scale, double* out, void* stream);
/* vara_i = m_i^T W m_i for L_pad rows (calculate_a_and_vara_rcpp.cpp:103-112), fp64 MFMA kernel. */
int eagle_dev_vara_f64(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                       double* vara_out, void* stream);
/* Same result from int8 digit slices of the off-diagonal part of Wu on the int8 MFMA (exact integer partial sums)
 * plus the fp64 diagonal term.  nslices: 0 = automatic (see eagle_set_scan_slices), 1..8 = fixed.
 * ws: workspace, eagle_vara_i8_workspace_bytes(n_pad, L_pad, nslices) bytes; its first 48 bytes are written by the
 * device as { double max|offdiag|; int32 S_used; int32 pad; double bound; double sum|diag|; double R = sum_{j<k} Wu_jk }.
 * err_bound_dev (device double, may be NULL): the absolute error bound n_pad^2 * 2^(e+1-8S) of every vara_i. */
int64_t eagle_vara_i8_workspace_bytes(long n_pad, long L_pad, int nslices);
/* The same in two phases, so that one pass over the genotype bytes serves both a = Mt8 v (if v != NULL; written to
 * a_out) and the diagonal term, and so that the MFMA kernel can be timed on its own:
 *   prepare: max |off-diagonal|, slice count, diagonal vector, fused genotype pass, digit slices;
 *   mfma   : k_vara_i8 over all (marker tile, slice) workers + the S-term finish into vara_out. */
int eagle_dev_vara_i8_prepare(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                              int nslices, void* ws, const double* v, double* a_out, void* stream);
int eagle_dev_vara_i8_mfma(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, int nslices, void* ws,
                           double* vara_out, double* err_bound_dev, void* stream);
/* Re-centred markers for the digit-slice kernel: Mt8s[i][j] = Mt8[i][j] - c_i for the n real individuals (padding stays
 * zero), c_i in {-1,0,+1} = the majority genotype of marker i, cshift[i] = c_i.  eagle_dev_vara_i8_mfma_shifted runs
 * the MFMA kernel on Mt8s (rare-variant markers become sparse rows, so their truncation error bound
 * (sum_j |m'_ij|)^2 / 2 * 2^(e+1-8S) shrinks with their own diagonal term) and adds c_i m_i^T rho - c_i^2 R in fp64;
 * prepare (always on the ORIGINAL image) has left rho, R and m^T rho in the workspace.
 * l1norm (may be NULL; 2 * L_pad int32): l1norm[2i] = sum_j |Mt8s[i][j]|, l1norm[2i+1] = sum_j Mt8s[i][j]^2, which
 * eagle_dev_scan_certify turns into marker i's error bound. */
int eagle_dev_marker_shift(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n, long n_pad, long ld, int8_t* Mt8s,
                           int8_t* cshift, int32_t* l1norm, void* stream);
int eagle_dev_vara_i8_mfma_shifted(eagle_ctx* ctx, const int8_t* Mt8s, const int8_t* cshift, long L_pad, long n_pad, long ld,
                                   int nslices, void* ws, double* vara_out, double* err_bound_dev, void* stream);
/* Between eagle_dev_vara_i8_mfma_shifted and the certification (same image, cshift, l1norm, workspace), automatic digit count only:
 * when the scan ran on one digit fewer than cut (spectral bound, eagle_set_scan_budget), every marker whose own bound
 * min(specH q2_i, l1_i^2/2 * 128.5 u) exceeds 1.8 x budget of |vara_i| is given the dropped digit back: its rows are gathered, the
 * vara kernel runs once more on them with the last digit slice alone, and vara_i becomes, bit for bit, the value a scan on all the
 * cut digits gives that marker (the certificate then uses the rounding bound of that digit count for it).  Up to min(65,536, L_pad)
 * markers per call; more: nobody is extended and the certificate decides (fp64 fallback of the block).  Nothing runs when no marker
 * qualifies.  A call that skips this step is still certified correctly -- its flagged markers go to the fp64 kernel instead. */
int eagle_dev_vara_i8_extend(eagle_ctx* ctx, const int8_t* Mt8s, const int8_t* cshift, const int32_t* l1norm, long L, long L_pad, long n_pad,
                             long ld, int nslices, void* ws, double* vara, void* stream);
int eagle_dev_vara_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                      int nslices, void* ws, double* vara_out, double* err_bound_dev, void* stream);
/* Certification of a digit-slice scan (after eagle_dev_vara_i8_mfma_shifted, before the arg-max), all on the device:
 * every vara_i carries the a-posteriori bound b_i = l1_i^2 / 2 * 2^(e+1-8S) (stochastic rounding: 8.355 q2_i 2^(e+1-8S)) + fp64
 * rounding terms; markers with
 * b_i > 1e-7 |vara_i|, and every marker that the bounds cannot exclude from being the arg-max of tsq = a^2 / vara
 * (a_i^2 / (vara_i - b_i) >= max_j a_j^2 / (vara_j + b_j)), are re-evaluated by the fp64 MFMA kernel on the ORIGINAL image
 * Mt8 -- bitwise the value eagle_dev_vara_f64 gives that marker -- and written back into vara.  The arg-max of tsq over the
 * certified arrays is then the arg-max of the fp64 scan (find_qtl.R:71-83 selects the same marker in either mode).
 * L = real markers of the block (rows beyond it are padding), L_pad / nslices / vara_ws as passed to prepare + mfma.
 * cert_ws: eagle_scan_certify_workspace_bytes(n_pad) bytes; its head is an eagle_cert_info the host may copy back.  If more
 * than 2048 markers qualify (degenerate operands) the whole block is redone in fp64 (overflow = 1).
 * Round 4: over_tight = markers whose bound exceeds 1.8 x the budget in force; with the tight budget (1e-7) in force and more than 512
 * of them (structured populations: quadratic forms that cancel against their diagonal term) the certificate enforces 1.8 x the default
 * budget (5e-7) instead -- `flagged` counts against the threshold that was enforced. */
typedef struct { double lower_bound; int32_t reevaluated; int32_t overflow; int32_t flagged; int32_t over_tight; } eagle_cert_info;
int64_t eagle_scan_certify_workspace_bytes(long n_pad);
int eagle_dev_scan_certify(eagle_ctx* ctx, const int8_t* Mt8, long L, long L_pad, long n_pad, long ld, const int8_t* cshift,
                           const int32_t* l1norm, int nslices, void* vara_ws, const double* Wu, const double* a, double* vara,
                           void* cert_ws, void* stream);
/* The same in two phases, for scans whose markers are spread over several devices / ranks: _lb leaves the lower bound of
 * this block's maximum tsq in cert_ws (eagle_cert_info.lower_bound); the caller takes the maximum over all blocks and hands
 * it to _apply as lb_override (NaN: the block's own), so that every block selects exactly the candidates a single scan of
 * all markers would. */
int eagle_dev_scan_certify_lb(eagle_ctx* ctx, long L, long L_pad, long n_pad, const int8_t* cshift, const int32_t* l1norm, int nslices,
                              void* vara_ws, const double* a, const double* vara, void* cert_ws, void* stream);
int eagle_dev_scan_certify_apply(eagle_ctx* ctx, const int8_t* Mt8, long L, long L_pad, long n_pad, long ld, const int8_t* cshift,
                                 const int32_t* l1norm, int nslices, void* vara_ws, const double* Wu, const double* a, double* vara,
                                 void* cert_ws, double lb_override, void* stream);
/* Certification counters of the LAST eagle_calculate_a_and_vara call of this ctx in digit-slice mode (summed over the
 * marker blocks of a streamed file): markers re-evaluated in fp64, of which flagged by their own error bound, and
 * whether a block fell back to the fp64 kernel entirely. */
int eagle_last_scan_certificate(eagle_ctx* ctx, long* n_reevaluated, long* n_flagged, int* fell_back);
/* Digit slices of the LAST eagle_calculate_a_and_vara call in digit-slice mode: used by the scan, cut from W (the same, or one more
 * when the spectral bound took the last one off), and that bound (|error_i| <= spectral_bound * sum_j m'_ij^2; 0 = not in use). */
int eagle_last_scan_digits(eagle_ctx* ctx, int* digits_used, int* digits_cut, double* spectral_bound);
/* Round 4.  The budget IN FORCE for the last digit-slice scan (what its certificate enforced per marker is 1.8 x this), the level of the
 * spectral bound that took the last digit off (0: none, 1, 2) and the error bound of W itself when the int8 engine formed it (0: fp64
 * products).  A context whose budget was never set tries 1e-7 first and falls back to 5e-7: the tight budget is in force whenever
 * the digits that run anyway certify a marker with q2 = n_pad to it (worst-case bound, or the spectral bound at level 1 / 2);
 * eagle_set_scan_budget(b) makes b the only budget, eagle_set_scan_budget(0) restores the default policy. */
int eagle_last_scan_budget(eagle_ctx* ctx, double* budget_used, int* bound_level, double* w_error_bound);
/* What the certificate of the last digit-slice scan ENFORCED per marker (bound <= 1.8 x budget_enforced x |vara_i|, else fp64): the
 * budget in force, or the default behind a tight one when more than 512 markers of the whole scan (all blocks, all devices) missed the
 * tight threshold (n_over_tight; see eagle_cert_info). */
int eagle_last_scan_enforced(eagle_ctx* ctx, double* budget_enforced, long* n_over_tight);
/* Round 4.  The first eagle_calculate_a_and_vara of a context allocates its device arena (four n x n fp64 images + the digit-slice and
 * certification workspaces: 100 GB at 50,000 individuals, 3-6 s of hipMalloc).  This call starts that allocation on a background
 * thread and returns at once; the scan collects it.  eagle_calculateMMt calls it by itself (AM() calls calcMMt once and then works
 * on the host for seconds); EAGLE_HIP_ARENA_GB=<GB> does the same at eagle_open.  dims as R has them: n individuals, L markers. */
int eagle_prepare_scan(eagle_ctx* ctx, long n, long L);
/* Out-of-core bookkeeping of the LAST call of this ctx that streamed its file through HBM in marker chunks (a file larger than
 * free HBM or than EAGLE_HIP_MAX_RESIDENT_GB; the lead device's share in a multi-device context): what SURVEY 8(d) asks to be
 * reported for the streamed configurations.  The loader of chunk k+1 (pread -> pinned -> H2D -> decode / 2-bit unpack) runs
 * under the kernels of chunk k.  Fraction of the load time hidden = 1 - starved_s / (load_s - load_first_s). */
typedef struct eagle_stream_stats {
    long chunks;         /* marker chunks */
    long file_bytes;     /* bytes read from the genotype file (text, or its 2-bit sidecar) */
    double pread_s;      /* host seconds inside the parallel preads (file_bytes / pread_s = storage rate) */
    double load_s;       /* host seconds in the chunk loaders (read + H2D + decode, synchronised per chunk) */
    double wait_s;       /* host seconds waiting for a free chunk buffer (kernels of chunk k-2 still running) */
    double kernel_s;     /* device seconds of the chunks' kernels (HIP events on the compute stream) */
    double wall_s;       /* first load to last kernel (includes whatever the compute stream had queued before chunk 0, e.g. W = S V S) */
    double load_first_s; /* the first chunk's load: nothing of this call's chunk kernels to hide under */
    double starved_s;    /* device seconds the compute stream sat idle between chunks because the next one was not loaded yet */
} eagle_stream_stats;
int eagle_last_stream_stats(eagle_ctx* ctx, eagle_stream_stats* out);
/* Where the LAST eagle_calculate_a_and_vara / eagle_scan_with_W call spent its time on device `device_index` of the context
 * (0 = the lead, which works on the calling thread): the accounting between the device-resident step bench.py times and what
 * the .Call-shaped entry point costs (E/R/calculate_a_and_vara.R:20-31 is the caller).  The *_ms fields are HIP-event intervals
 * on that device's compute stream, summed over the marker blocks of a streamed file. */
typedef struct eagle_scan_timing {
    double call_wall_s;   /* the whole call on the calling thread (argument checks, all devices, joins) */
    double device_wall_s; /* this device's worker: resident lookup / file load, uploads, kernels, results back */
    double host_setup_s;  /* of it: host seconds until the operand uploads were enqueued (resident lookup or file load, arena,
                             staging of V, a_hat -- and S when it is not the cached one -- out of pageable memory) */
    double upload_ms;     /* stream time from the first upload to the last (V, a_hat, S on a cache miss) */
    double w_ms;          /* v = S a_hat, W = S (V S): symmetry check, two products, fold (+ all-gather of W's rows) */
    double load_wait_ms;  /* streamed files: compute stream waiting for a marker block to be loaded */
    double prepare_ms;    /* digits of W, rho, ONE genotype pass (a = Mt v, diagonal term of vara) */
    double vara_ms;       /* the vara kernel (int8 digit slices + finish, or the fp64 kernel) */
    double certify_ms;    /* error bounds, candidate selection, fp64 re-evaluation */
    double d2h_ms;        /* row masking + a, vara back into the caller's arrays */
    long blocks;          /* marker blocks (1 = resident) */
    long markers;         /* markers of this device's shard */
} eagle_scan_timing;
int eagle_last_scan_timing(eagle_ctx* ctx, int device_index, eagle_scan_timing* out);
/* eagle_calculate_a_and_vara keeps the last call's S = inv_MMt_sqrt on the device (MMt^-1/2 is the same matrix in every find_qtl
 * call of an AM() run; n_pad <= 16,384, 2 x 8 n_pad^2 bytes): the next call starts its n^3 products on that copy and meanwhile
 * uploads the caller's matrix and compares the two bit for bit; a difference starts the products over with the new matrix, so the
 * result never depends on the cache.  hits: calls whose S was the cached one (its PCIe upload hidden under the product);
 * misses: calls that started over.  EAGLE_HIP_NO_SCACHE=1 disables the mechanism. */
int eagle_scan_operand_cache_stats(eagle_ctx* ctx, long* hits, long* misses);
/* The same quadratic form on the block-scaled matrix path (v_mfma_scale_f32_32x32x64_f8f6f4): genotypes as fp4, balanced
 * base-33 digits of Wu as fp6 (every integer in [-16,16] is an e2m3 number / 8), exact fp32 sums, twice the MAC rate of
 * the int8 instruction.  Mt4: [L_pad][n_pad/2] bytes made once per genotype matrix by eagle_dev_pack_fp4 (two genotypes
 * per byte).  nslices: 0 = automatic (same 1e-7 worst-case criterion), 1..12 = fixed; absolute error bound n_pad^2 * 2^(e-5S).
 * The workspace head has the layout of the int8 one ({max|offdiag|; S; f; bound; sum|diag|}). */
int64_t eagle_vara_f6_workspace_bytes(long n_pad, long L_pad, int nslices);
int eagle_dev_pack_fp4(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, void* Mt4, void* stream);
/* The operand image of eagle_dev_mmt_accumulate_f4 straight from the MARKER-major genotypes, in one pass: M4[n_pad][ld4 bytes],
 * byte b of row j = markers 2b (low nibble) and 2b+1 (high nibble) of individual j as e2m1 codes, from Mt8[L_pad][ld].  The same
 * bytes as eagle_dev_transpose_i8 followed by eagle_dev_pack_fp4, without the individual-major int8 image in between.
 * L_pad % 256 == 0, n_pad % 128 == 0, ld % 16 == 0, ld4 % 16 == 0, ld4 >= L_pad / 2. */
int eagle_dev_transpose_pack_fp4(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, void* M4, long ld4, void* stream);
int eagle_dev_vara_f6_prepare(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                              int nslices, void* ws, const double* v, double* a_out, void* stream);
int eagle_dev_vara_f6_mfma(eagle_ctx* ctx, const int8_t* Mt8, const void* Mt4, long L_pad, long n_pad, long ld, int nslices,
                           void* ws, double* vara_out, double* err_bound_dev, void* stream);
/* The spectral scan (section 1d) on device-resident data: Z[L_pad][n_pad] = Mt8 * Ur with Ur = U row-major [n_pad][n_pad]
 * (Ur[j][k] = U[j][k], zero padded); one pass over Z: lin[L_pad][NC] = Z G and quad[i] = sum_k Z_ik^2 d_k with G [n_pad][NC]
 * row-major, NC = 16 or 32 (column 0 = d o U^T y, columns 1..p = d o U^T X, the rest zero); the finish:
 * a_i = varG (lin_i0 - q_i . c1), vara_i = varG^2 (quad_i - q_i^T C q_i), q_i = lin_i[1..p], C p x p row-major, c1 p. */
int eagle_dev_spectral_zbuild(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Ur, double* Z, void* stream);
/* The same Z from S exact int8 digit slices of U on the int8 MFMA (the tile engine of the digit-slice scan, dense K loop, store
 * epilogue): |Z_ik - (Mt U)_ik| <= (sum_j |m_ij|) 2^(e+1-8S), max|U| < 2^e <= 1.  ws: eagle_spectral_zbuild_i8_workspace_bytes bytes.
 * eagle_spectral_prepare uses S = 6 (error <= n 2^-47) unless eagle_set_scan_mode(ctx, 0) asks for the fp64 form. */
int64_t eagle_spectral_zbuild_i8_workspace_bytes(long n_pad, int nslices);
int eagle_dev_spectral_zbuild_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Ur, double* Z, void* ws,
                                 int nslices, void* stream);
int eagle_dev_spectral_pass(eagle_ctx* ctx, const double* Z, long L_pad, long n_pad, const double* G, int NC, const double* d, double* lin,
                            double* quad, void* stream);
int eagle_dev_spectral_finish(eagle_ctx* ctx, const double* lin, int NC, const double* quad, long L, long p, const double* Cm, const double* c1,
                              double varG, double* a, double* vara, void* stream);
/* zero a[i], vara[i] at the listed rows (row masking of calculate_a_and_vara_rcpp.cpp:79-84: a zeroed
 * marker row yields exactly a = 0, vara = 0). rows_dev: device array of long, entries outside [0,L) ignored. */
int eagle_dev_zero_rows(eagle_ctx* ctx, double* a, double* vara, long L, const long* rows_dev, long nrows,
                        long row_offset, void* stream);
/* tsq = a^2/vara, first index of the maximum ignoring NaN (find_qtl.R:71-83).
 * best_dev: device struct {double tsqmax; long index0; long near_ties;} index0 = -1 if all NaN.
 * tsq_out may be NULL. block_scratch: 3*1024 doubles. */
typedef struct { double tsqmax; long index0; long near_ties; } eagle_best;
int eagle_dev_tsq_argmax(eagle_ctx* ctx, const double* a, const double* vara, long L, double* tsq_out,
                         eagle_best* best_dev, double* block_scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAGLE_HIP_H */
