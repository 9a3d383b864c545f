// eagle_internal.h -- shared between eagle_api.cpp and eagle_kernels.hip (not part of the public ABI)
#ifndef EAGLE_INTERNAL_H
#define EAGLE_INTERNAL_H
#include <hip/hip_runtime.h>

#include "../../include/eagle_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
int eagle_fail(eagle_ctx* ctx, int code, const char* msg);
int eagle_fail_hip(eagle_ctx* ctx, hipError_t e, const char* where);
// C = A * B, all row-major np x np fp64, np % 128 == 0
int eagle_dev_gemm_f64(eagle_ctx* ctx, const double* A, const double* B, double* C, long np, void* stream);
// eagle_dev_scan_operands in three steps (eagle_kernels.hip): the caller may upload V in row blocks under the first product
#define EAGLE_VROWS_BLOCK 1024  /* rows of V's image per upload + product step (a multiple of 256) */
int eagle_dev_scan_operands_begin(eagle_ctx* ctx, const double* Sa, const double* ahat, long n, long n_pad, double* v_out, double* tmp, void* stream);
int eagle_dev_scan_operands_vrows(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, long row0, long row1, double* tmp, void* stream);
int eagle_dev_scan_operands_finish(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, double* Wu_out, double* tmp, void* stream);
// the two n^3 products alone (v is made by _begin): W's folded image from resident operands on the fp64 GEMM
int eagle_dev_scan_operands_w_f64(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, double* Wu_out, double* tmp, void* stream);
// W = S (V S) from int8 digit slices (eagle_w8.hip): EAGLE_OK, 1 = declined (run the fp64 products), < 0 error
int eagle_dev_scan_operands_w8(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long n_pad, double* v_out, double* Wu_out,
                               double* tmp, void* stream);
int eagle_dev_colgemv_parts(eagle_ctx* ctx, const double* At, long n, long n_pad, const double* x, double* out, double* part, void* stream);
// the same in three steps (eagle_w8.hip): V may arrive in row blocks of eagle_w8_vrows_block() rows under the first product
int eagle_w8_begin(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long n_pad, double* v_out, double* Wu_out, double* tmp,
                   int allow_guess, void* stream);
int eagle_w8_vrows(eagle_ctx* ctx, long r0, long r1, void* stream);
int eagle_w8_finish(eagle_ctx* ctx, void* stream);
int eagle_w8_vrows_block(void);
int eagle_w8_rho(eagle_ctx* ctx, const double* Wu, long n_pad, double* rho, void* stream);
int eagle_w8_true_vara(eagle_ctx* ctx, const int8_t* rows8, long count, long n_pad, long ld, const long* dst_dev, double* out, void* stream);
int eagle_w8_redo_f64(eagle_ctx* ctx, long n_pad, void* stream);
// out = A x where At is the row-major image of A^T (i.e. the column-major R matrix), n_pad % 64 == 0
int eagle_dev_colgemv(eagle_ctx* ctx, const double* At, long n, long n_pad, const double* x, double* out, void* stream);
// 4 KiB of ctx-owned device scratch (flags, small reductions); stream-ordered use only
void* eagle_ctx_scratch(eagle_ctx* ctx);
// grow-only ctx-owned device buffer for the fp4 image of a genotype tile (NULL + last_error on failure)
void* eagle_ctx_f4_buffer(eagle_ctx* ctx, size_t bytes);
int eagle_dev_gemv2_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v, const double* w,
                       double scale, double* out_a, double* out_d, void* stream);
int eagle_dev_gemv3_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v, const double* w,
                       const double* x, double scale, double* out_a, double* out_d, double* out_x, void* stream);
int eagle_dev_vara_f64_gated(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu, double* vara_out,
                             const int* run_if, void* stream);
long eagle_vara_f64_split_partial_doubles(long rows_cap, long n_pad);
int eagle_dev_vara_f64_split(eagle_ctx* ctx, const int8_t* rows8, long rows_cap, long n_pad, long ld, const double* Wu,
                             const int* count_dev, const long* dst_dev, double* partial, double* out, void* stream);
long eagle_upper_tiles_count(long n_pad);  // int32 elements of the packed upper 256-tiles of an n_pad x n_pad matrix
int eagle_dev_tiles_pack(eagle_ctx* ctx, int32_t* C32, long n_pad, int32_t* packed, int unpack, void* stream);
int eagle_dev_add_i32(eagle_ctx* ctx, int32_t* dst, const int32_t* src, long count, void* stream);
// eagle_dev_vara_i8_prepare in parts: 0 = all of it, 1 = the W-dependent part only (once per scan), 2 = the block part only
int eagle_dev_vara_i8_prepare_part(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu, int nslices, void* ws,
                                   const double* v, double* a_out, void* stream, int part);
int eagle_dev_cert_accumulate(eagle_ctx* ctx, const void* cert_ws, long* totals_dev, void* stream);
// certification of a scan cut into marker blocks / device shards against ONE lower bound (eagle_i8mfma.hip, "The same certification ...")
int eagle_dev_cert_bounds(eagle_ctx* ctx, long L, long L_pad, long n_pad, const int8_t* cshift, const int32_t* l1norm, int nslices,
                          const void* vara_ws, const double* vara, double* bound, void* stream);
int eagle_dev_cert_lb_b(eagle_ctx* ctx, long L, const double* a, const double* vara, const double* bound, void* cert_ws, const void* vara_ws,
                        void* stream);
int eagle_cert_tight_max(void);
int eagle_dev_cert_select_b(eagle_ctx* ctx, long L, const double* a, const double* vara, const double* bound, void* cert_ws, double lb,
                            long over_tight, const void* vara_ws, void* stream);
int8_t* eagle_cert_rows(void* cert_ws);
long* eagle_cert_indices(void* cert_ws);
int eagle_dev_cert_reevaluate(eagle_ctx* ctx, const int8_t* Mt8, long ld, long n_pad, const double* Wu, double* vara, void* cert_ws, void* stream);
int eagle_dev_symmetrize(eagle_ctx* ctx, double* A, long n, long ld, void* stream);
int eagle_dev_symmetrize_mean(eagle_ctx* ctx, double* A, long n, long ld, void* stream);
int eagle_dev_scale_rows_pow(eagle_ctx* ctx, double* R, long n, long ld, const double* w, double p, void* stream);
int eagle_dev_transpose_f64(eagle_ctx* ctx, const double* in, double* out, long N, void* stream);
int eagle_dev_dot_matrices(eagle_ctx* ctx, const double* A, long lda, const double* B, long ldb, long n, double* out, void* stream);
void eagle_linalg_release(eagle_ctx* ctx);
void eagle_spectral_release(eagle_ctx* ctx);
int eagle_spectral_prepare_range(eagle_ctx* ctx, const char* f_name_ascii, long L, long n, long m0, long m1, const double* U, double max_memory_in_Gbytes);
int eagle_spectral_host_operands(eagle_ctx* ctx, long n, const double* lambda, const double* UtX, const double* Uty, long p, double varE, double varG,
                                 int NC, double* d, double* G, double* Cm, double* c1);
int eagle_spectral_scan_range(eagle_ctx* ctx, const double* d, const double* G, int NC, const double* Cm, const double* c1, long p, double varG,
                              const long* sel, long nsel, double* a_out, double* vara_out);
int eagle_dev_extract_col(eagle_ctx* ctx, const int8_t* M8, long n, long ld, long col, int* out, void* stream);
#ifdef __cplusplus
}
struct eagle_ctx;
bool eagle_w8_wanted(const eagle_ctx* ctx, long n_pad);
void eagle_w8_release(eagle_ctx* ctx);
#endif
#endif
