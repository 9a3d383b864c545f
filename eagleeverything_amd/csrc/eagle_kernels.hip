// eagle_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Eagle hot path and their launchers
// (section 2 of include/eagle_hip.h).  Written for 64-wide wavefronts and MFMA; no other target.
//
//   k_decode_ascii ........ text tile '0','1','2' -> int8 {-1,0,1}     (E/src/ReadBlock.cpp:52-55)
//   (k_syrk_i8 / k_vara_i8: the int8 MFMA tile engine lives in eagle_i8mfma.hip)
//   k_gemm_f64_list ....... fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64): W = S (V S)   (calculate_a_and_vara_rcpp.cpp:97-98)
//   k_vara_f64<..> ........ same core, A = int8: T = Mt W fused with the row-dot    (calculate_a_and_vara_rcpp.cpp:103-112)
//   k_gemv_mfma ........... a = Mt v (+ diagonal term of vara)          (calculate_a_and_vara_rcpp.cpp:91,
//                                                                        calculate_reduced_a_rcpp.cpp:83-84)
//   k_tsq_* ............... tsq = a^2/vara, first arg-max ignoring NaN  (E/R/find_qtl.R:71-83)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eagle_hip.h"
#include "eagle_ctx.h"
#include "eagle_internal.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// decode: raw[row*line_stride + c] in {'0','1','2'} -> out[row*ld_out + c] = c - '0' - 1 ; pad = 0.
// Also checks the fixed-width assumption: byte `cols` of every line must be '\n' (or '\r') when
// line_stride > cols.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode_ascii(const uint8_t* __restrict__ raw, long rows, long cols,
                                                      long line_stride, int8_t* __restrict__ out, long ld_out,
                                                      int* __restrict__ bad) {
    const long row = blockIdx.y;
    const long c4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c4 >= ld_out) return;
    const uint8_t* src = raw + row * line_stride;
    int nbad = 0;
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        long c = c4 + q;
        int v = 0;
        if (c < cols) {
            int ch = src[c];
            v = ch - '0' - 1;
            if (ch < '0' || ch > '2') nbad++;
        }
        packed |= ((uint32_t)(uint8_t)(int8_t)v) << (8 * q);
    }
    const bool owns_eol = (cols >= c4 && cols < c4 + 4) || (cols == ld_out && c4 + 4 == ld_out);
    if (owns_eol && line_stride > cols) {
        int ch = src[cols];
        if (ch != '\n' && ch != '\r') nbad++;
    }
    *(uint32_t*)(out + row * ld_out + c4) = packed;
    if (nbad) atomicAdd(bad, nbad);
}

// 64x64-byte tile transpose through LDS. rows, cols multiples of 64.
__global__ __launch_bounds__(256) void k_transpose_i8(const int8_t* __restrict__ in, long ld_in,
                                                      int8_t* __restrict__ out, long ld_out) {
    __shared__ __attribute__((aligned(16))) int8_t tile[64][80];
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
    const int t = threadIdx.x;
    {
        int r = t >> 2, ch = (t & 3) * 16;
        i32x4 v = *(const i32x4*)(in + (r0 + r) * ld_in + c0 + ch);
        *(i32x4*)(&tile[r][ch]) = v;
    }
    __syncthreads();
    {
        int c = t >> 2, ch = (t & 3) * 16;  // output row c0+c, bytes r0+ch .. +15
        union { i32x4 v; int8_t b[16]; } u;
#pragma unroll
        for (int q = 0; q < 16; q++) u.b[q] = tile[ch + q][c];
        *(i32x4*)(out + (c0 + c) * ld_out + r0 + ch) = u.v;
    }
}

__global__ __launch_bounds__(256) void k_i8_to_f64_colmajor(const int8_t* __restrict__ in, long rows, long cols,
                                                            long ld_in, double* __restrict__ out) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cols) return;
    long r = idx % rows, c = idx / rows;
    out[idx] = (double)in[r * ld_in + c];
}

__global__ __launch_bounds__(256) void k_mmt_downdate(const int8_t* __restrict__ M8, long n_pad, long ld,
                                                      const long* __restrict__ cols, long ncols,
                                                      int32_t* __restrict__ C, long ldc) {
    long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n_pad) return;
    int s = 0;
    for (long q = 0; q < ncols; q++) {
        long c = cols[q];
        bool dup = false;  // zeroing a column twice is zeroing it once
        for (long p = 0; p < q; p++) dup |= (cols[p] == c);
        if (!dup) s += (int)M8[i * ld + c] * (int)M8[j * ld + c];
    }
    // only the upper-triangular tiles of C are live
    if ((i >> 8) <= (j >> 8)) C[i * ldc + j] -= s;
}

// One block per 32x32 tile on or above the diagonal: coalesced read of the int32 tile, coalesced write of the fp64
// tile and (through LDS) of its mirror image; block max -> atomicMax on the raw bits (the maximum is non-negative).
__global__ __launch_bounds__(256) void k_mmt_finish(const int32_t* __restrict__ C, long n, long ldc,
                                                    double* __restrict__ out, long ld_out,
                                                    unsigned long long* __restrict__ maxbits) {
    const long bi = (long)blockIdx.y * 32, bj = (long)blockIdx.x * 32;
    if (bj < bi) return;
    __shared__ double t[32][33];
    __shared__ double sm[4];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double m = 0.0;
    for (int r = ty; r < 32; r += 8) {
        const long i = bi + r, j = bj + tx;
        double v = 0.0;
        if (i < n && j < n) {
            v = (double)C[i * ldc + j];
            out[i * ld_out + j] = v;
            m = v > m ? v : m;
        }
        t[r][tx] = v;
    }
    __syncthreads();
    if (bj > bi)
        for (int r = ty; r < 32; r += 8) {
            const long j = bj + r, i = bi + tx;  // out[j][i] = C[i][j]
            if (i < n && j < n) out[j * ld_out + i] = t[tx][r];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double x = __shfl_down(m, o); m = x > m ? x : m; }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        if (m > 0.0) atomicMax(maxbits, (unsigned long long)__double_as_longlong(m));
    }
}

__global__ __launch_bounds__(256) void k_mmt_normalise(double* __restrict__ A, long n, long ld,
                                                       const double* __restrict__ maxv) {
    long j = (long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    double v = A[i * ld + j] / *maxv;  // E/R/calcMMt.R:13
    if (i == j) v = v + 0.95;
    A[i * ld + j] = v;
}

// ------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM core, C = A * B, v_mfma_f64_16x16x4_f64.
// Block = 256 threads = 4 waves (2 x 2); block tile 128 x 128; wave tile 64 x 64 = 4 x 4 MFMA tiles;
// K block 16.  Inside a K block the lane group g = lane>>4 owns k = 4g + s for MFMA step s, so a lane reads
// 4 consecutive k of its A row at once (the k order of an fp64 sum is a free choice; it is fixed, hence
// deterministic).
//   AMODE 0: A is fp64 row-major (k_gemm_f64_list).   AMODE 1: A is int8 row-major (genotypes), converted in registers
//   (k_vara_f64, which fuses the row-dot out[i] = sum_c C[i][c] * A8[i][c] into the tile epilogue).
// LDS rows: A f64 [128][16] doubles (+2 pad), A i8 [128][16] bytes, B [16][128] doubles (+4 pad).
// ------------------------------------------------------------------------------------------------
#define GF_T 128
#define GF_BK 16
#define GF_LDB (GF_T + 4)   /* doubles per B row in LDS: 4*stride*8 mod 256 == 128 -> the 4 k-groups hit distinct banks */
#define GF_LDA (GF_BK + 4)  /* doubles per A row in LDS (pitch 20: 51.2 -> 50.3 ms for W at n = 10,000 against pitch 18; 24 is 10 % slower) */

template <int AMODE>
struct GfStage {
    f64x2 b[4];
    f64x2 a[AMODE == 0 ? 4 : 1];
    i32x4 a8;
};

template <int AMODE>
__device__ __forceinline__ void gf_load(GfStage<AMODE>& s, const void* __restrict__ Ablk, long lda,
                                        const double* __restrict__ Bblk, long ldb, long k0, int t) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c = t + 256 * i;           // 1024 chunks of 2 doubles: 16 rows x 64 chunks
        int kr = c >> 6, cc = c & 63;
        s.b[i] = *(const f64x2*)(Bblk + (k0 + kr) * ldb + cc * 2);
    }
    if (AMODE == 0) {
        const double* A = (const double*)Ablk;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int c = t + 256 * i;       // 1024 chunks: 128 rows x 8 chunks
            int row = c >> 3, cc = c & 7;
            s.a[i] = *(const f64x2*)(A + (long)row * lda + k0 + cc * 2);
        }
    } else {
        const int8_t* A = (const int8_t*)Ablk;
        if (t < 128) s.a8 = *(const i32x4*)(A + (long)t * lda + k0);
    }
}
template <int AMODE, int LDA = GF_LDA>
__device__ __forceinline__ void gf_store(const GfStage<AMODE>& s, double* ldsA, double* ldsB, int t) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c = t + 256 * i;
        int kr = c >> 6, cc = c & 63;
        *(f64x2*)(ldsB + kr * GF_LDB + cc * 2) = s.b[i];
    }
    if (AMODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int c = t + 256 * i;
            int row = c >> 3, cc = c & 7;
            *(f64x2*)(ldsA + row * LDA + cc * 2) = s.a[i];
        }
    } else {
        if (t < 128) *(i32x4*)((int8_t*)ldsA + t * 16) = s.a8;
    }
}
// mlim: only the first mlim 16-row tiles of the wave's four are computed (the SPLIT vara kernel re-evaluates a handful of
// rows; the MFMA sequence of a computed element is the same whatever mlim is)
template <int AMODE, int LDA = GF_LDA>
__device__ __forceinline__ void gf_compute(f64x4 (&acc)[4][4], const double* ldsA, const double* ldsB, int wr, int wc,
                                           int lane, int mlim = 4) {
    const int i16 = lane & 15, g = lane >> 4;
    int w4[4];  // AMODE 1: the lane's 4 consecutive genotype bytes of each row tile
    if (AMODE == 1) {
#pragma unroll
        for (int m = 0; m < 4; m++) w4[m] = *(const int*)((const int8_t*)ldsA + (wr * 64 + m * 16 + i16) * 16 + 4 * g);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        double a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (AMODE == 0) a[m] = ldsA[(wr * 64 + m * 16 + i16) * LDA + 4 * g + s];
            else a[m] = (double)((w4[m] << (24 - 8 * s)) >> 24);  // sign-extended byte s
        }
#pragma unroll
        for (int n = 0; n < 4; n++) b[n] = ldsB[(4 * g + s) * GF_LDB + wc * 64 + n * 16 + i16];
#pragma unroll
        for (int m = 0; m < 4; m++)
            if (m < mlim) {
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
            }
    }
}

// LDS bytes: A: 128*20*8 = 20480 (f64) ; B: 16*132*8 = 16896 ; x2 buffers (two workgroups per CU: 149.5 KiB)
#define GF_LDSA_DOUBLES (GF_T * GF_LDA)
#define GF_LDSB_DOUBLES (GF_BK * GF_LDB)

// The fp64 vara kernel (calculate_a_and_vara_rcpp.cpp:103-112): vara_i = sum_k m_ik sum_{j<=k} m_ij Wu[j][k] for the 128 markers
// of a row block, A = int8 genotypes converted in registers, B = Wu (upper triangular fold of W, so column tile ct only
// needs k < (ct+1)*128).  The summation order is FIXED and is the definition of this library's fp64 result:
//   s_{ct,c}(i, wc) = the row-dot of column tile ct, restricted to the 64 columns of wave column wc, of the partial product over
//                     the K chunk c = [2048 c, 2048 (c+1)): the MFMA chain over the chunk, then per lane a 4-term chain over
//                     the lane's columns, then a butterfly over the 16 lanes of the row;
//   P_wc(i)         = the s_{ct,c} added one by one, ct ascending, c ascending inside a tile;   vara_i = P_0(i) + P_1(i).
// SPLIT = false: one workgroup walks every (column tile, chunk) of its row block and keeps P in registers.
// SPLIT = true : grid.y = column tile, grid.z = chunk; the workgroup writes s_{ct,c} to partial[rb][ct][c][wc][128] and
//                k_vara_f64_sum forms the same chain -- bitwise the same vara for a row, whichever form computed it and
//                whatever rows share its block.  That is what lets the digit-slice scan re-evaluate a handful of markers in
//                fp64 on the whole chip (k_cert_*, eagle_i8mfma.hip: the longest workgroup is one 2048-deep chunk, 0.3 ms,
//                instead of a 10240-deep tile) and promise the fp64-mode result for them.
// gate (may be NULL): SPLIT = false -> the launch is dropped unless *gate != 0; SPLIT = true -> row blocks at or beyond
// *gate rows are dropped.
#define GF_KC 2048  /* K chunk of the canonical summation order */
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_vara_f64(const int8_t* __restrict__ A8, long lda, const double* __restrict__ B, long ldb,
                                                     double* __restrict__ out, int n_coltiles, long K, const int* __restrict__ gate,
                                                     double* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) double lds[2][GF_LDSA_DOUBLES + GF_LDSB_DOUBLES];
    if (gate) {
        if (!SPLIT && *gate == 0) return;
        if (SPLIT && (long)blockIdx.x * GF_T >= (long)*gate) return;
    }
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * GF_T;
    const int8_t* Ablk = A8 + row0 * lda;
    // SPLIT with a row count: 16-row tiles entirely beyond the count are not computed (wave-uniform)
    int mlim = 4;
    if (SPLIT && gate) {
        const long left = (long)*gate - row0 - wr * 64;
        mlim = left <= 0 ? 0 : (left >= 64 ? 4 : (int)((left + 15) >> 4));
        mlim = __builtin_amdgcn_readfirstlane(mlim);
    }
    const int nchunk_max = (int)((K + GF_KC - 1) / GF_KC);
    double P[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) P[m][q] = 0.0;
    const int ct0 = SPLIT ? (int)blockIdx.y : 0;
    const int ct1 = SPLIT ? (int)blockIdx.y + 1 : n_coltiles;
    for (int ct = ct0; ct < ct1; ct++) {
        const double* Bblk = B + (long)ct * GF_T;
        const long kend = (long)(ct + 1) * GF_T < K ? (long)(ct + 1) * GF_T : K;
        const int c0 = SPLIT ? (int)blockIdx.z : 0;
        const int c1 = SPLIT ? (int)blockIdx.z + 1 : nchunk_max;
        for (int c = c0; c < c1; c++) {
            const long kc0 = (long)c * GF_KC;
            if (kc0 >= kend) break;
            const long kc1 = kc0 + GF_KC < kend ? kc0 + GF_KC : kend;
            const long nkb = (kc1 - kc0) / GF_BK;
            f64x4 acc[4][4];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
            GfStage<1> st;
            gf_load<1>(st, Ablk, lda, Bblk, ldb, kc0, t);
            __syncthreads();  // the previous chunk's readers are done with buffer 0
            gf_store<1>(st, lds[0], lds[0] + GF_LDSA_DOUBLES, t);
            __syncthreads();
            int cur = 0;
            for (long kb = 0; kb < nkb; kb++) {
                const bool more = kb + 1 < nkb;
                if (more) gf_load<1>(st, Ablk, lda, Bblk, ldb, kc0 + (kb + 1) * GF_BK, t);
                gf_compute<1>(acc, lds[cur], lds[cur] + GF_LDSA_DOUBLES, wr, wc, lane, mlim);
                if (more) gf_store<1>(st, lds[cur ^ 1], lds[cur ^ 1] + GF_LDSA_DOUBLES, t);
                __syncthreads();
                cur ^= 1;
            }
            // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const long r = row0 + wr * 64 + m * 16 + g + 4 * q;
                    const int8_t* mr = A8 + r * lda + (long)ct * GF_T + wc * 64 + i16;
                    double s = 0.0;
#pragma unroll
                    for (int n = 0; n < 4; n++) s += acc[m][n][q] * (double)mr[n * 16];
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    s += __shfl_xor(s, 4);
                    s += __shfl_xor(s, 8);
                    if (SPLIT) {
                        if (i16 == 0)
                            partial[((((long)blockIdx.x * n_coltiles + ct) * nchunk_max + c) * 2 + wc) * GF_T + wr * 64 + m * 16 + g + 4 * q] = s;
                    } else {
                        P[m][q] += s;
                    }
                }
        }
    }
    if (!SPLIT) {
        __syncthreads();
        double* red = lds[0];  // [2][128]
        if (i16 == 0) {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int q = 0; q < 4; q++) red[wc * 128 + wr * 64 + m * 16 + g + 4 * q] = P[m][q];
        }
        __syncthreads();
        if (t < 128) out[row0 + t] = red[t] + red[128 + t];
    }
}
// vara of the rows of the SPLIT form: out[dst ? dst[r] : r] = P_0 + P_1, P_wc the chain of the partial sums in the order above.
__global__ __launch_bounds__(128) void k_vara_f64_sum(const double* __restrict__ partial, int n_coltiles, long K, const int* __restrict__ count,
                                                      const long* __restrict__ dst, double* __restrict__ out) {
    const long r = (long)blockIdx.x * GF_T + threadIdx.x;
    if (count && r >= (long)*count) return;
    const int nchunk_max = (int)((K + GF_KC - 1) / GF_KC);
    const double* p = partial + (long)blockIdx.x * n_coltiles * nchunk_max * 2 * GF_T + threadIdx.x;
    double P0 = 0.0, P1 = 0.0;
    for (int ct = 0; ct < n_coltiles; ct++) {
        const long kend = (long)(ct + 1) * GF_T < K ? (long)(ct + 1) * GF_T : K;
        for (int c = 0; c < nchunk_max && (long)c * GF_KC < kend; c++) {
            P0 += p[(((long)ct * nchunk_max + c) * 2 + 0) * GF_T];
            P1 += p[(((long)ct * nchunk_max + c) * 2 + 1) * GF_T];
        }
    }
    out[dst ? dst[r] : r] = P0 + P1;
}

// In-place fold  W -> Wu:  Wu[j][k] = W[j][k] + W[k][j] (j<k) ; W[k][k] (j==k) ; 0 (j>k).
// One block per pair of mirrored 32x32 tiles.  If *sym (S and V symmetric, so W = S V S is, and only the 128-tiles on
// or above the diagonal were computed) an off-diagonal 128-tile contributes 2 W[j][k] instead.
__global__ __launch_bounds__(256) void k_fold_upper(double* __restrict__ W, long np, const int* __restrict__ sym) {
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    if (bk < bj) return;
    __shared__ double t1[32][33], t2[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const bool diag = (bk == bj);
    const bool twice = (*sym != 0) && ((bj >> 7) != (bk >> 7));
    for (int r = ty; r < 32; r += 8) {
        t1[r][tx] = W[(bj + r) * np + bk + tx];
        if (!diag && !twice) t2[r][tx] = W[(bk + r) * np + bj + tx];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long j = bj + r, k = bk + tx;
        double v;
        if (diag) v = (j < k) ? t1[r][tx] + t1[tx][r] : (j == k ? t1[r][tx] : 0.0);
        else v = twice ? 2.0 * t1[r][tx] : t1[r][tx] + t2[tx][r];
        W[j * np + k] = v;
        if (!diag) W[(bk + r) * np + bj + tx] = 0.0;
    }
}

// *sym = 1 iff both square matrices equal their transposes up to 1e-12 of the larger of the two diagonal entries
// involved (they are MMt^-1/2 and a variance matrix in every Eagle run; an arbitrary caller may pass anything).
__global__ __launch_bounds__(256) void k_sym_check(const double* __restrict__ A, const double* __restrict__ B, long np,
                                                   int* __restrict__ sym) {
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    if (bk < bj) return;  // (the diagonal 32-tiles too: round 3 found an asymmetry inside one of them going unnoticed)
    __shared__ double t2[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        t2[0][r][tx] = A[(bk + r) * np + bj + tx];
        t2[1][r][tx] = B[(bk + r) * np + bj + tx];
    }
    __syncthreads();
    bool bad = false;
    for (int r = ty; r < 32; r += 8) {
        const long j = bj + r, k = bk + tx;
        const double a = A[j * np + k], at = t2[0][tx][r], b = B[j * np + k], bt = t2[1][tx][r];
        const double sa = fmax(fmax(fabs(A[j * np + j]), fabs(A[k * np + k])), fmax(fabs(a), fabs(at)));
        const double sb = fmax(fmax(fabs(B[j * np + j]), fabs(B[k * np + k])), fmax(fabs(b), fabs(bt)));
        if (!(fabs(a - at) <= 1e-12 * sa) || !(fabs(b - bt) <= 1e-12 * sb)) bad = true;
    }
    if (bad) *sym = 0;
}
__global__ void k_set_int(int* p, int v) { *p = v; }

// out[i] = sum_j At[j][i] * x[j]   (At row-major = column-major image of the R matrix: out = A x)
// Block = 4 waves x 64 columns; wave w sums j = w, w+4, ... ; fixed-order LDS combine.
__global__ __launch_bounds__(256) void k_colgemv(const double* __restrict__ At, long n, long np,
                                                 const double* __restrict__ x, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    long j = w;
    for (; j + 12 < n; j += 16) {
        s0 += At[j * np + i] * x[j];
        s1 += At[(j + 4) * np + i] * x[j + 4];
        s2 += At[(j + 8) * np + i] * x[j + 8];
        s3 += At[(j + 12) * np + i] * x[j + 12];
    }
    for (; j < n; j += 4) s0 += At[j * np + i] * x[j];
    __shared__ double red[4][64];
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0) out[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// The same product with the j range cut into gridDim.y parts (more workgroups than 64-column strips: an n x n matrix is
// only n/64 strips wide): part[y][i] holds the partial sum of part y; k_colgemv_sum adds the parts in a fixed order.
__global__ __launch_bounds__(256) void k_colgemv_part(const double* __restrict__ At, long n, long np, const double* __restrict__ x,
                                                      double* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const long per = ((n + gridDim.y - 1) / gridDim.y + 3) / 4 * 4;
    const long j0 = (long)blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
    double s0 = 0, s1 = 0;
    long j = j0 + w;
    for (; j + 4 < j1; j += 8) {
        s0 += At[j * np + i] * x[j];
        s1 += At[(j + 4) * np + i] * x[j + 4];
    }
    for (; j < j1; j += 4) s0 += At[j * np + i] * x[j];
    __shared__ double red[4][64];
    red[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0) part[(long)blockIdx.y * np + i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void k_colgemv_sum(const double* __restrict__ part, long np, int nparts, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    double s = part[i];
    for (int y = 1; y < nparts; y++) s += part[(long)y * np + i];
    out[i] = s;
}

__global__ __launch_bounds__(256) void k_extract_col(const int8_t* __restrict__ M8, long n, long ld, long col, int* __restrict__ out) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int)M8[i * ld + col];
}

__global__ __launch_bounds__(256) void k_zero_rows(double* __restrict__ a, double* __restrict__ vara, long L,
                                                   const long* __restrict__ rows, long nrows, long row_offset) {
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nrows) return;
    long r = rows[q] - row_offset;
    if (r < 0 || r >= L) return;
    if (a) a[r] = 0.0;
    if (vara) vara[r] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// tsq = a^2 / vara ; first index of the maximum, NaN ignored (find_qtl.R:71-83).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void best_merge(double& bv, long& bi, double v, long i) {
    // (v, i) beats (bv, bi) if bi < 0, or v > bv, or v == bv and i < bi ; i < 0 means "none"
    if (i >= 0 && (bi < 0 || v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
}
__global__ __launch_bounds__(256) void k_tsq_partial(const double* __restrict__ a, const double* __restrict__ vara, long L,
                                                     double* __restrict__ tsq_out, double* __restrict__ pv,
                                                     long* __restrict__ pi) {
    double bv = 0.0;
    long bi = -1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        double x = a[i];
        double t = (x * x) / vara[i];
        if (tsq_out) tsq_out[i] = t;
        if (t == t) best_merge(bv, bi, t, i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_down(bv, o);
        long oi = __shfl_down(bi, o);
        best_merge(bv, bi, ov, oi);
    }
    __shared__ double sv[4];
    __shared__ long si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) best_merge(bv, bi, sv[w], si[w]);
        pv[blockIdx.x] = bv;
        pi[blockIdx.x] = bi;
    }
}
__global__ __launch_bounds__(256) void k_tsq_final(const double* __restrict__ pv, const long* __restrict__ pi, int nparts,
                                                   eagle_best* __restrict__ best) {
    double bv = 0.0;
    long bi = -1;
    for (int p = threadIdx.x; p < nparts; p += 256) best_merge(bv, bi, pv[p], pi[p]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_down(bv, o);
        long oi = __shfl_down(bi, o);
        best_merge(bv, bi, ov, oi);
    }
    __shared__ double sv[4];
    __shared__ long si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) best_merge(bv, bi, sv[w], si[w]);
        best->tsqmax = bi >= 0 ? bv : __longlong_as_double(0x7ff8000000000000LL);
        best->index0 = bi;
        best->near_ties = 0;
    }
}
__global__ __launch_bounds__(256) void k_tsq_near(const double* __restrict__ a, const double* __restrict__ vara, long L,
                                                  eagle_best* __restrict__ best) {
    const double mx = best->tsqmax;
    if (!(mx == mx)) return;
    const double thr = isinf(mx) ? mx : mx - fabs(mx) * 1e-9;
    int cnt = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        double x = a[i];
        double t = (x * x) / vara[i];
        if (t == t && t >= thr) cnt++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd((unsigned long long*)&best->near_ties, (unsigned long long)cnt);
}

// ================================================================================================
// launchers (section 2 of eagle_hip.h)
// ================================================================================================
#define LAUNCH_CHECK(ctx)                                                   \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) return eagle_fail_hip(ctx, e__, __func__);   \
    } while (0)

extern "C" long eagle_pad(long x) { return (x + 255) / 256 * 256; }

extern "C" int eagle_dev_decode_ascii(eagle_ctx* ctx, const uint8_t* raw, long rows, long cols, long line_stride,
                                      int8_t* out, long ld_out, int* bad_chars_dev, void* stream) {
    if (rows <= 0) return EAGLE_OK;
    if (ld_out % 4 || cols > ld_out || rows > 65535L * 32768L) return eagle_fail(ctx, EAGLE_ERR_ARG, "decode_ascii: bad shape");
    // grid.y is limited to 65535: walk row bands
    for (long r0 = 0; r0 < rows; r0 += 65535) {
        long nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((ld_out / 4 + 255) / 256), (unsigned)nr);
        hipLaunchKernelGGL(k_decode_ascii, grid, dim3(256), 0, (hipStream_t)stream, raw + r0 * line_stride, nr, cols,
                           line_stride, out + r0 * ld_out, ld_out, bad_chars_dev);
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_transpose_i8(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, int8_t* out,
                                      long ld_out, void* stream) {
    if (rows % 64 || cols % 64 || ld_in % 16 || ld_out % 16 || cols > ld_in || rows > ld_out)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "transpose_i8: dims must be multiples of 64");
    for (long r0 = 0; r0 < rows; r0 += 64L * 65535) {
        long nr = rows - r0 < 64L * 65535 ? rows - r0 : 64L * 65535;
        dim3 grid((unsigned)(cols / 64), (unsigned)(nr / 64));
        hipLaunchKernelGGL(k_transpose_i8, grid, dim3(256), 0, (hipStream_t)stream, in + r0 * ld_in, ld_in, out + r0,
                           ld_out);
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_i8_to_f64_colmajor(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in,
                                            double* out_colmajor, void* stream) {
    long tot = rows * cols;
    if (tot <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_i8_to_f64_colmajor, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       rows, cols, ld_in, out_colmajor);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_mmt_downdate(eagle_ctx* ctx, const int8_t* M8, long n_pad, long ld, const long* cols_dev,
                                      long ncols, int32_t* C32, void* stream) {
    if (ncols <= 0) return EAGLE_OK;
    dim3 grid((unsigned)((n_pad + 255) / 256), (unsigned)n_pad);
    if (n_pad > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_downdate: n too large");
    hipLaunchKernelGGL(k_mmt_downdate, grid, dim3(256), 0, (hipStream_t)stream, M8, n_pad, ld, cols_dev, ncols, C32, n_pad);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_mmt_finish(eagle_ctx* ctx, const int32_t* C32, long n, long n_pad, double* MMt, long ld_out,
                                    double* max_dev, void* stream) {
    if (n > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_finish: n too large");
    hipError_t e = hipMemsetAsync(max_dev, 0, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "mmt_finish memset");
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(k_mmt_finish, grid, dim3(256), 0, (hipStream_t)stream, C32, n, n_pad, MMt, ld_out,
                       (unsigned long long*)max_dev);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_mmt_normalise(eagle_ctx* ctx, double* MMt, long n, long ld, const double* max_dev, void* stream) {
    if (n > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_normalise: n too large");
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(k_mmt_normalise, grid, dim3(256), 0, (hipStream_t)stream, MMt, n, ld, max_dev);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// Partial MM^T of several devices -> one sum.  Only the 256 x 256 tiles on or above the diagonal of C32 are live, so only
// they travel: packed[t] (t = ti*nt - ti(ti-1)/2 + tj - ti) is the tile (ti, tj) as 65536 contiguous int32.
//   k_tiles_pack   : C32 upper tiles -> packed          k_tiles_unpack : packed -> C32 upper tiles
//   k_add_i32      : dst += src (exact integer sum; the host-staged stand-in for the RCCL reduce)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tiles_pack(int32_t* __restrict__ C, long np, int32_t* __restrict__ packed, int nt, int unpack) {
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;
    const long t = (long)ti * nt - (long)ti * (ti - 1) / 2 + (tj - ti);
    int32_t* tile = packed + t * 65536;
    for (int e = threadIdx.x * 4; e < 65536; e += 1024) {
        const int r = e >> 8, c = e & 255;
        i32x4* g = (i32x4*)(C + ((long)ti * 256 + r) * np + (long)tj * 256 + c);
        i32x4* p = (i32x4*)(tile + e);
        if (unpack) *g = *p; else *p = *g;
    }
}
__global__ __launch_bounds__(256) void k_add_i32(int32_t* __restrict__ dst, const int32_t* __restrict__ src, long count4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count4; i += (long)gridDim.x * 256) {
        i32x4 a = ((const i32x4*)dst)[i], b = ((const i32x4*)src)[i];
        ((i32x4*)dst)[i] = a + b;
    }
}
extern "C" long eagle_upper_tiles_count(long n_pad) { const long nt = n_pad / 256; return nt * (nt + 1) / 2 * 65536; }
extern "C" int eagle_dev_tiles_pack(eagle_ctx* ctx, int32_t* C32, long n_pad, int32_t* packed, int unpack, void* stream) {
    if (n_pad % 256 || n_pad <= 0 || n_pad / 256 > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "tiles_pack: bad padding");
    const int nt = (int)(n_pad / 256);
    hipLaunchKernelGGL(k_tiles_pack, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, C32, n_pad, packed, nt, unpack);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_add_i32(eagle_ctx* ctx, int32_t* dst, const int32_t* src, long count, void* stream) {
    if (count % 4) return eagle_fail(ctx, EAGLE_ERR_ARG, "add_i32: count must be a multiple of 4");
    if (count == 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_add_i32, dim3(2048), dim3(256), 0, (hipStream_t)stream, dst, src, count / 4);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// C = A * B over a LIST of 128 x 128 output tiles, with the last partial wave of workgroups split along K.
// 512 workgroups are resident (2 per CU); 1600 tiles (np = 5120) are 3.125 waves, so a plain grid leaves 7/8 of the chip
// idle through a fourth wave.  The first floor(T/512)*512 tiles run whole; each remaining tile is cut into `split` K
// ranges whose partial tiles go to a scratch buffer and are added in a fixed order by k_gemm_f64_tail (deterministic).
// skip_if / skip_val: the whole launch is dropped on the device when *skip_if == skip_val (the lower-triangle launch of
// a product the symmetry check found symmetric).
// ------------------------------------------------------------------------------------------------
template <int LDA, int XCDMAP>
__global__ __launch_bounds__(256, 2) void k_gemm_f64_list(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                          double* __restrict__ C, long ldc, long K, const int* __restrict__ tiles, int n_main,
                                                          int split, double* __restrict__ scratch, const int* __restrict__ skip_if, int skip_val) {
    constexpr int LDSA = GF_T * LDA;
    __shared__ __attribute__((aligned(16))) double lds[2][LDSA + GF_LDSB_DOUBLES];
    if (skip_if && *skip_if == skip_val) return;
    int b = blockIdx.x;
    // XCDMAP: workgroup b runs on XCD b % 8 (observed dealing); give each XCD a contiguous range of the whole tiles of the list
    if (XCDMAP && b < n_main) b = (b & 7) * (n_main >> 3) + (b >> 3);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long nkb_all = K / GF_BK;
    int tile;
    long kb0 = 0, kb1 = nkb_all;
    double* out;
    long ldo;
    if (b < n_main) {
        tile = tiles[b];
        out = nullptr; ldo = ldc;
    } else {
        const int u = b - n_main, tt = u / split, ks = u - tt * split;
        tile = tiles[n_main + tt];
        kb0 = nkb_all * ks / split;
        kb1 = nkb_all * (ks + 1) / split;
        out = scratch + (long)u * (GF_T * GF_T);
        ldo = GF_T;
    }
    const long row0 = (long)(tile >> 16) * GF_T, col0 = (long)(tile & 0xffff) * GF_T;
    const void* Ablk = (const void*)(A + row0 * lda);
    const double* Bblk = B + col0;
    f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    GfStage<0> st;
    gf_load<0>(st, Ablk, lda, Bblk, ldb, kb0 * GF_BK, t);
    gf_store<0, LDA>(st, lds[0], lds[0] + LDSA, t);
    __syncthreads();
    int cur = 0;
    for (long kb = kb0; kb < kb1; kb++) {
        const bool more = kb + 1 < kb1;
        if (more) gf_load<0>(st, Ablk, lda, Bblk, ldb, (kb + 1) * GF_BK, t);
        gf_compute<0, LDA>(acc, lds[cur], lds[cur] + LDSA, wr, wc, lane);
        if (more) gf_store<0, LDA>(st, lds[cur ^ 1], lds[cur ^ 1] + LDSA, t);
        __syncthreads();
        cur ^= 1;
    }
    double* dst = out ? out : C + row0 * ldc + col0;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) dst[(long)(wr * 64 + m * 16 + g + 4 * q) * ldo + wc * 64 + n * 16 + i16] = acc[m][n][q];
}
// The same tile with EIGHT waves (4 x 2, wave tile 32 x 64 = 2 x 4 MFMA tiles, 64 accumulator registers): under 128 VGPRs, so two
// workgroups = 16 waves = 4 per SIMD are resident and the matrix pipe has four waves to choose from around every barrier
// instead of two (the 4-wave form keeps the fp64 pipe 83 % busy).  Same K order per element: bitwise the same C.
// Sixteen waves (wave tile 16 x 64, 64 VGPRs with spills) were measured too: 54.8 ms against 48.4 ms at n = 10,000, not kept.
template <int LDA>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gemm_f64_list8(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                           double* __restrict__ C, long ldc, long K, const int* __restrict__ tiles, int n_main,
                                                           int split, double* __restrict__ scratch, const int* __restrict__ skip_if, int skip_val) {
    constexpr int LDSA = GF_T * LDA;
    __shared__ __attribute__((aligned(16))) double lds[2][LDSA + GF_LDSB_DOUBLES];
    if (skip_if && *skip_if == skip_val) return;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long nkb_all = K / GF_BK;
    int tile;
    long kb0 = 0, kb1 = nkb_all;
    double* out;
    long ldo;
    if (b < n_main) {
        tile = tiles[b];
        out = nullptr; ldo = ldc;
    } else {
        const int u = b - n_main, tt = u / split, ks = u - tt * split;
        tile = tiles[n_main + tt];
        kb0 = nkb_all * ks / split;
        kb1 = nkb_all * (ks + 1) / split;
        out = scratch + (long)u * (GF_T * GF_T);
        ldo = GF_T;
    }
    const long row0 = (long)(tile >> 16) * GF_T, col0 = (long)(tile & 0xffff) * GF_T;
    const double* Ablk = A + row0 * lda;
    const double* Bblk = B + col0;
    f64x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    f64x2 sa[2], sb[2];
    auto load = [&](long k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = t + 512 * i;
            sb[i] = *(const f64x2*)(Bblk + (k0 + (c >> 6)) * ldb + (c & 63) * 2);
            sa[i] = *(const f64x2*)(Ablk + (long)(c >> 3) * lda + k0 + (c & 7) * 2);
        }
    };
    auto store = [&](double* ldsA, double* ldsB) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = t + 512 * i;
            *(f64x2*)(ldsB + (c >> 6) * GF_LDB + (c & 63) * 2) = sb[i];
            *(f64x2*)(ldsA + (c >> 3) * LDA + (c & 7) * 2) = sa[i];
        }
    };
    load(kb0 * GF_BK);
    store(lds[0], lds[0] + LDSA);
    __syncthreads();
    int cur = 0;
    for (long kb = kb0; kb < kb1; kb++) {
        const bool more = kb + 1 < kb1;
        if (more) load((kb + 1) * GF_BK);
        const double* ldsA = lds[cur];
        const double* ldsB = lds[cur] + LDSA;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            double a[2], bb[4];
#pragma unroll
            for (int m = 0; m < 2; m++) a[m] = ldsA[(wr * 32 + m * 16 + i16) * LDA + 4 * g + s4];
#pragma unroll
            for (int n = 0; n < 4; n++) bb[n] = ldsB[(4 * g + s4) * GF_LDB + wc * 64 + n * 16 + i16];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
        }
        if (more) store(lds[cur ^ 1], lds[cur ^ 1] + LDSA);
        __syncthreads();
        cur ^= 1;
    }
    double* dst = out ? out : C + row0 * ldc + col0;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) dst[(long)(wr * 32 + m * 16 + g + 4 * q) * ldo + wc * 64 + n * 16 + i16] = acc[m][n][q];
}
__global__ __launch_bounds__(256) void k_gemm_f64_tail(const double* __restrict__ scratch, const int* __restrict__ tail_tiles, int split,
                                                       double* __restrict__ C, long ldc, const int* __restrict__ skip_if, int skip_val) {
    if (skip_if && *skip_if == skip_val) return;
    const int tile = tail_tiles[blockIdx.x >> 4], part = blockIdx.x & 15;  // 16 workgroups per tile, 8 rows each
    double* dst = C + (long)(tile >> 16) * GF_T * ldc + (long)(tile & 0xffff) * GF_T;
    const double* src = scratch + (long)(blockIdx.x >> 4) * split * (GF_T * GF_T);
    for (int e = part * 1024 + threadIdx.x; e < (part + 1) * 1024; e += 256) {
        double s = src[e];
        for (int k = 1; k < split; k++) s += src[(long)k * (GF_T * GF_T) + e];  // ascending K: fixed order
        dst[(long)(e >> 7) * ldc + (e & 127)] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// k_gemm_f64_dma (round 3; the W = S (V S) products at n_pad % 256 == 0): C = A * B on 256 x 128 output tiles,
// operands global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write), XOR-swizzled unpadded tiles,
// fragments software-pipelined by half K blocks, ONE barrier per K block placed inside a run of MFMAs.
//   8 waves (4 x 2), wave tile 64 x 64 = 4 x 4 MFMA tiles (128 accumulator VGPRs), one workgroup per CU; a 256 x 128 tile streams
//   (256 + 128) K doubles per 256 * 128 * K MACs: 3/4 of the panel bytes of the 128 x 128 form.
//   K block = 16.  LDS stage = A [256 rows][128 B] (32 KiB) + B [16 k][1 KiB] (16 KiB); two stages = 96 KiB.
//   A: 16-byte chunk c (k = 2c, 2c+1) of row r lives at chunk c ^ ((r >> 1) & 7); lane (i16, g) reads chunks g and 4 + g of its
//      row with ds_read_b128 -- the four 16-lane groups of that instruction each cover all 16 granules of a 256-byte bank line
//      (checked by tests/test_kernel_index_maps.py) -- so lane group g owns k in {2g, 2g+1, 8+2g, 9+2g} and MFMA step s of a
//      K block multiplies k = 2g + (s & 1) + 8 (s >> 1), g = 0..3.  (The k order inside an fp64 sum is a free choice; this one
//      is FIXED and is the definition of this kernel's result.  It differs from k_gemm_f64_list8's {s, 4+s, 8+s, 12+s}, so the two
//      kernels agree to rounding, not bit for bit.)
//   B: row k is 1 KiB; its 128-byte block nt (16 columns) lives at block nt ^ ((k >> 1) & 1), so the two lane groups a
//      ds_read_b64 serves together (k and k + 2, same columns) sit in different halves of the bank line.
//   DMA: a wave instruction writes 1 KiB of consecutive LDS bytes; the swizzles are applied to the per-lane SOURCE offset
//      (two VGPRs for A: even / odd groups of 8 rows; one for B: the rows 2w, 2w+1 of wave w share (k >> 1) & 1).
//   Pipeline of K block kb (stage buffer kb & 1; X = the fragments of k-half 0, Y = of k-half 1):
//        read Y(kb)                     | 32 MFMAs on X(kb)
//                                       | 16 MFMAs on Y(kb)
//        barrier  -- every wave's DMA pieces of stage kb+1 have landed (issued a whole K block ago), all reads of stage kb are done
//        DMA stage kb+2 -> buffer of stage kb ; read X(kb+1)   | 16 MFMAs on Y(kb)
//   Tiles: (row128 << 16) | col128 -- a tile starts at any multiple of 128 rows and covers 256; rows at or beyond `row_end` are
//   read as zero (buffer descriptor bounds) and not stored (the last tile of a row range whose length is an odd multiple of 128).
//   TRANS: the tile is stored transposed, C[col][row] (the symmetric-operand pipeline of eagle_dev_scan_operands).
// ------------------------------------------------------------------------------------------------
#define G2_TM 256
#define G2_TN 128
#define G2_A_BYTES (G2_TM * GF_BK * 8)
#define G2_B_BYTES (GF_BK * G2_TN * 8)
#define G2_STAGE (G2_A_BYTES + G2_B_BYTES)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <bool TRANS>
__global__ __launch_bounds__(512, 2) void k_gemm_f64_dma(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                         double* __restrict__ C, long ldc, long K, long row_end, const int* __restrict__ tiles,
                                                         int n_main, int split, double* __restrict__ scratch, const int* __restrict__ skip_if,
                                                         int skip_val) {
    __shared__ __attribute__((aligned(1024))) char lds[2][G2_STAGE];
    if (skip_if && *skip_if == skip_val) return;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long nkb_all = K / GF_BK;
    int tile;
    long kb0 = 0, kb1 = nkb_all;
    double* out;
    long ldo;
    if (b < n_main) {
        tile = tiles[b];
        out = nullptr; ldo = ldc;
    } else {
        const int u = b - n_main, tt = u / split, ks = u - tt * split;
        tile = tiles[n_main + tt];
        kb0 = nkb_all * ks / split;
        kb1 = nkb_all * (ks + 1) / split;
        out = scratch + (long)u * (G2_TM * G2_TN);
        ldo = G2_TN;
    }
    const long row0 = (long)(tile >> 16) * 128, col0 = (long)(tile & 0xffff) * 128;
    const long rows_here = row_end - row0 < G2_TM ? row_end - row0 : G2_TM;
    const int lda8 = (int)(lda * 8), ldb8 = (int)(ldb * 8);
    // A: rows of this tile, all K (256 * lda * 8 bytes < 2^32); B: re-based per K block (16 rows), columns of this tile
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + row0 * lda), 0, (int)(rows_here * lda8), 0x00020000);
    const int voffAE = (lane >> 3) * lda8 + (((lane & 7) ^ (lane >> 4)) << 4), voffAO = voffAE ^ 64;
    const int voffB = (((lane >> 3) ^ (w & 1)) << 7) + ((lane & 7) << 4);
    auto dma = [&](long kb, char* st) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int grp = w * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(st + grp * 1024), 16, (i & 1) ? voffAO : voffAE,
                                                     grp * 8 * lda8 + (int)(kb * GF_BK * 8), 0, 0);
        }
        const __amdgpu_buffer_rsrc_t rsB =
            __builtin_amdgcn_make_buffer_rsrc((void*)(B + kb * GF_BK * ldb + col0), 0, GF_BK * ldb8, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int k = w * 2 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(st + G2_A_BYTES + k * 1024), 16, voffB, k * ldb8, 0, 0);
        }
    };
    // fragment addresses: per-lane LDS pointers of both stages, made once -- inside the K loop every read is a base register + an
    // immediate offset and the loop counters are scalar ints: a vector instruction between two fp64 MFMAs of a wave costs about 6
    // of the matrix pipe's cycles even with a second wave on the SIMD (tools/ubench/f64_ceiling.hip: 0.99 of the peak with no VALU
    // in the loop, 0.92 with one v_mov per MFMA), so the loop holds nothing but fragment reads, DMA issues and MFMAs
    const int offA = (wr * 64 + i16) * 128 + ((g ^ (i16 >> 1)) << 4);  // chunk g of row wr*64 + m*16 + i16 (+ m * 2048); chunk 4+g: ^ 64
    const int offB = G2_A_BYTES + (2 * g) * 1024 + i16 * 8;             // row k = 2g (+ 1024 for k+1, + 8192 for the second k-half)
    const char* pAx[2] = {lds[0] + offA, lds[1] + offA};
    const char* pAy[2] = {lds[0] + (offA ^ 64), lds[1] + (offA ^ 64)};
    const char* pB[2][4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int blk = ((wc * 4 + n) ^ (g & 1)) << 7;
        pB[0][n] = lds[0] + offB + blk;
        pB[1][n] = lds[1] + offB + blk;
    }
    f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    f64x2 ax[4], ay[4];
    double bx[4][2], by[4][2];
#define G2_READ_X(ST)                                                              \
    do {                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; m++) ax[m] = *(const f64x2*)(pAx[ST] + m * 2048); \
        _Pragma("unroll") for (int n = 0; n < 4; n++) {                            \
            bx[n][0] = *(const double*)(pB[ST][n]);                                \
            bx[n][1] = *(const double*)(pB[ST][n] + 1024);                         \
        }                                                                          \
    } while (0)
#define G2_READ_Y(ST)                                                              \
    do {                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; m++) ay[m] = *(const f64x2*)(pAy[ST] + m * 2048); \
        _Pragma("unroll") for (int n = 0; n < 4; n++) {                            \
            by[n][0] = *(const double*)(pB[ST][n] + 8192);                         \
            by[n][1] = *(const double*)(pB[ST][n] + 8192 + 1024);                  \
        }                                                                          \
    } while (0)
#define G2_MFMA(A_, B_, M0, M1)                                                    \
    do {                                                                           \
        _Pragma("unroll") for (int e = 0; e < 2; e++)                              \
            _Pragma("unroll") for (int m = M0; m < M1; m++)                        \
                _Pragma("unroll") for (int n = 0; n < 4; n++)                      \
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[m][e], B_[n][e], acc[m][n], 0, 0, 0); \
    } while (0)
    // One REGION = the code between two barriers, on stage buffer ST holding K block kb_ (visible since the barrier before it):
    //   DMA of K block kb_+1 into the other buffer (all its readers passed that barrier) | fragment reads X(kb_) | the last 16 MFMAs
    //   of K block kb_-1 (on its Y fragments) | fragment reads Y(kb_) | 48 MFMAs of K block kb_ | barrier.
    // The sched_group_barriers spell that order out for hipcc (it otherwise parks fragment reads behind the MFMAs and then waits for
    // them in front of the barrier); the sched_barrier keeps the barrier behind the MFMAs, so a wave arrives with matrix work in flight.
#define G2_REGION(ST, kb_, first, more)                                            \
    do {                                                                           \
        if (more) dma((kb_) + 1, lds[(ST) ^ 1]);                                   \
        G2_READ_X(ST);                                                             \
        if (!(first)) G2_MFMA(ay, by, 2, 4);                                       \
        G2_READ_Y(ST);                                                             \
        G2_MFMA(ax, bx, 0, 4);                                                     \
        G2_MFMA(ay, by, 0, 2);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                         \
        if (!(first)) __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);          \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                         \
        __syncthreads(); /* this wave's DMA of K block kb_+1 (vmcnt) and its reads of this stage (lgkmcnt), then the barrier */ \
    } while (0)
    const int nkb = (int)(kb1 - kb0);
    dma(kb0, lds[0]);
    __syncthreads();
    G2_REGION(0, kb0, true, nkb > 1);
    int i = 1;
    for (; i + 2 <= nkb; i += 2) {   // two K blocks per trip: the stage buffer of every read is a compile-time constant
        G2_REGION(1, kb0 + i, false, true);
        G2_REGION(0, kb0 + i + 1, false, i + 2 < nkb);
    }
    if (i < nkb) G2_REGION(1, kb0 + i, false, false);
    G2_MFMA(ay, by, 2, 4);   // the tail of the last K block
#undef G2_REGION
#undef G2_MFMA
#undef G2_READ_X
#undef G2_READ_Y
    // C/D map of v_mfma_f64_16x16x4: lane (i16, g), register q holds row g + 4 q, column i16 of the 16 x 16 tile
    double* dst = out ? out : C;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const long rl = wr * 64 + m * 16 + g + 4 * q, cl = wc * 64 + n * 16 + i16;
                if (rl >= rows_here) continue;
                if (out) dst[rl * ldo + cl] = acc[m][n][q];
                else if (TRANS) dst[(col0 + cl) * ldc + row0 + rl] = acc[m][n][q];
                else dst[(row0 + rl) * ldc + col0 + cl] = acc[m][n][q];
            }
}
// partial tiles of the split-K tail added in ascending K order (fixed: deterministic); 32 workgroups per tile, 8 rows each
template <bool TRANS>
__global__ __launch_bounds__(256) void k_gemm_f64_dma_tail(const double* __restrict__ scratch, const int* __restrict__ tail_tiles, int split,
                                                           double* __restrict__ C, long ldc, long row_end, const int* __restrict__ skip_if, int skip_val) {
    if (skip_if && *skip_if == skip_val) return;
    const int tile = tail_tiles[blockIdx.x >> 5], part = blockIdx.x & 31;
    const long row0 = (long)(tile >> 16) * 128, col0 = (long)(tile & 0xffff) * 128;
    const double* src = scratch + (long)(blockIdx.x >> 5) * split * (G2_TM * G2_TN);
    for (int e = part * 1024 + threadIdx.x; e < (part + 1) * 1024; e += 256) {
        double s = src[e];
        for (int k = 1; k < split; k++) s += src[(long)k * (G2_TM * G2_TN) + e];
        const long rl = e >> 7, cl = e & 127;
        if (row0 + rl >= row_end) continue;
        if (TRANS) C[(col0 + cl) * ldc + row0 + rl] = s;
        else C[(row0 + rl) * ldc + col0 + cl] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// k_vara_f64d (round 3): the fp64 vara kernel -- scan mode 0, the re-evaluation engine of the certification, the overflow fallback --
// rebuilt on the lessons of k_gemm_f64_dma: every vector instruction between two fp64 MFMAs of a wave costs ~6 of the matrix pipe's
// cycles (tools/ubench/f64_ceiling.hip), and k_vara_f64 converted its int8 A fragments to doubles inside the K loop (12 VALU per
// 16 MFMAs, every element converted by both column waves) on top of register-staged B tiles.  Here
//   * the genotype bytes of a K block are converted ONCE by the staging threads (8 per thread) and written to LDS as an fp64 tile
//     [128 rows][128 B] with k_gemm_f64_dma's chunk swizzle; the fragments come back with two ds_read_b128 per 16-row tile;
//   * Wu's K block arrives by LDS-DMA ([16 k][1 KiB], 128-byte block swizzle), no VGPR staging;
//   * same lane-group ownership of k as k_gemm_f64_dma ({2g, 2g+1, 8+2g, 9+2g}), same barrier-to-barrier regions;
//   * 4 waves 2 x 2, wave tile 64 x 64, two workgroups per CU (64 KiB of LDS each): two waves per SIMD.
// Everything that DEFINES the result is as in k_vara_f64 except the k order inside a K block: per (row, column tile, 2048-deep K
// chunk, 64-column half) the MFMA chain over the chunk, a 4-term chain + 16-lane butterfly, the partial row-dots added tiles
// ascending, chunks ascending, the two halves last.  SPLIT = true puts every (tile, chunk) on its own workgroup and k_vara_f64_sum
// forms the same chain: bitwise the same value for a row whichever form computed it.  (tune 28: the round-2 kernel, for A/B runs.)
// ------------------------------------------------------------------------------------------------
#define V2_A_BYTES (GF_T * GF_BK * 8)
#define V2_B_BYTES (GF_BK * GF_T * 8)
#define V2_STAGE (V2_A_BYTES + V2_B_BYTES)
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_vara_f64d(const int8_t* __restrict__ A8, long lda, const double* __restrict__ B, long ldb,
                                                      double* __restrict__ out, int n_coltiles, long K, const int* __restrict__ gate,
                                                      double* __restrict__ partial) {
    __shared__ __attribute__((aligned(1024))) char lds[2][V2_STAGE];
    __shared__ double Psum[2][GF_T];  // the running row-dots P_wc(row) of the non-SPLIT form (in LDS: 32 VGPRs the K loop needs)
    // SPLIT: grid (column tile, K chunk, row block) -- the row block is the SLOWEST index: workgroup ids go round the XCDs, and with the
    // row block fastest the few blocks that hold candidates (normally the first of 16) had all their workgroups on one XCD (1.16 ms for
    // one 128-row block; rocprofv3, round 3)
    const int rb = SPLIT ? (int)blockIdx.z : (int)blockIdx.x;
    if (gate) {
        if (!SPLIT && *gate == 0) return;
        if (SPLIT && (long)rb * GF_T >= (long)*gate) return;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const long row0 = (long)rb * GF_T;
    const int8_t* Ablk = A8 + row0 * lda;
    int mlim = 4;  // SPLIT with a row count: 16-row tiles entirely beyond the count are not computed (wave-uniform)
    if (SPLIT && gate) {
        const long left = (long)*gate - row0 - wr * 64;
        mlim = left <= 0 ? 0 : (left >= 64 ? 4 : (int)((left + 15) >> 4));
        mlim = __builtin_amdgcn_readfirstlane(mlim);
    }
    const int nchunk_max = (int)((K + GF_KC - 1) / GF_KC);
    // staging of A: thread t converts the 8 bytes k0 + 8 ah .. of row ar and writes chunks 4 ah + j (k = 8 ah + 2j, 2j+1), swizzled
    const int ar = t >> 1, ah = t & 1;
    const int8_t* Arow = Ablk + (long)ar * lda + 8 * ah;
    int wrA[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wrA[j] = ar * 128 + (((4 * ah + j) ^ ((ar >> 1) & 7)) << 4);
    // DMA of B: wave w issues rows 4w .. 4w+3 of the K block; rows 4w, 4w+1 have (k >> 1) & 1 = 0, rows 4w+2, 4w+3 have 1
    const int ldb8 = (int)(ldb * 8);
    const int voffB0 = ((lane >> 3) << 7) + ((lane & 7) << 4), voffB1 = voffB0 ^ 128;
    const int offA = (wr * 64 + i16) * 128 + ((g ^ (i16 >> 1)) << 4);
    const int offB = V2_A_BYTES + (2 * g) * 1024 + i16 * 8;
    const char* pAx[2] = {lds[0] + offA, lds[1] + offA};
    const char* pAy[2] = {lds[0] + (offA ^ 64), lds[1] + (offA ^ 64)};
    const char* pB[2][4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const int blk = ((wc * 4 + n) ^ (g & 1)) << 7;
        pB[0][n] = lds[0] + offB + blk;
        pB[1][n] = lds[1] + offB + blk;
    }
    if (!SPLIT) { Psum[t >> 7][t & 127] = 0.0; }   // (made visible by the barriers of the first chunk)
    const int ct0 = SPLIT ? (int)blockIdx.x : 0;
    const int ct1 = SPLIT ? (int)blockIdx.x + 1 : n_coltiles;
    for (int ct = ct0; ct < ct1; ct++) {
        const double* Bblk = B + (long)ct * GF_T;
        const long kend = (long)(ct + 1) * GF_T < K ? (long)(ct + 1) * GF_T : K;
        const int c0 = SPLIT ? (int)blockIdx.y : 0;
        const int c1 = SPLIT ? (int)blockIdx.y + 1 : nchunk_max;
        for (int c = c0; c < c1; c++) {
            const long kc0 = (long)c * GF_KC;
            if (kc0 >= kend) break;
            const long kc1 = kc0 + GF_KC < kend ? kc0 + GF_KC : kend;
            const int nkb = (int)((kc1 - kc0) / GF_BK);
            f64x4 acc[4][4];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
            i32x2 a8;   // the 8 genotype bytes this thread stages next
            auto a_load = [&](int kb) { a8 = *(const i32x2*)(Arow + kc0 + (long)kb * GF_BK); };
            auto a_store = [&](char* st) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int d = a8[j >> 1], sh = (j & 1) * 16;
                    f64x2 v;
                    v[0] = (double)((d << (24 - sh)) >> 24);
                    v[1] = (double)((d << (16 - sh)) >> 24);
                    *(f64x2*)(st + wrA[j]) = v;
                }
            };
            auto b_dma = [&](int kb, char* st) {
                const __amdgpu_buffer_rsrc_t rsB =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(Bblk + (kc0 + (long)kb * GF_BK) * ldb), 0, GF_BK * ldb8, 0x00020000);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = w * 4 + i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(st + V2_A_BYTES + k * 1024), 16, (i & 2) ? voffB1 : voffB0, k * ldb8, 0, 0);
                }
            };
            f64x2 ax[4], ay[4];
            double bx[4][2], by[4][2];
#define V2_READ_X(ST)                                                              \
    do {                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; m++) ax[m] = *(const f64x2*)(pAx[ST] + m * 2048); \
        _Pragma("unroll") for (int n = 0; n < 4; n++) {                            \
            bx[n][0] = *(const double*)(pB[ST][n]);                                \
            bx[n][1] = *(const double*)(pB[ST][n] + 1024);                         \
        }                                                                          \
    } while (0)
#define V2_READ_Y(ST)                                                              \
    do {                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; m++) ay[m] = *(const f64x2*)(pAy[ST] + m * 2048); \
        _Pragma("unroll") for (int n = 0; n < 4; n++) {                            \
            by[n][0] = *(const double*)(pB[ST][n] + 8192);                         \
            by[n][1] = *(const double*)(pB[ST][n] + 8192 + 1024);                  \
        }                                                                          \
    } while (0)
#define V2_MFMA(A_, B_, M0, M1)                                                    \
    do {                                                                           \
        _Pragma("unroll") for (int e = 0; e < 2; e++)                              \
            _Pragma("unroll") for (int m = M0; m < M1; m++)                        \
                if (!SPLIT || m < mlim) {                                          \
                    _Pragma("unroll") for (int n = 0; n < 4; n++)                  \
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[m][e], B_[n][e], acc[m][n], 0, 0, 0); \
                }                                                                  \
    } while (0)
            // region = the code between two barriers, on stage buffer ST holding K block kb_ (see k_gemm_f64_dma): stage kb_+1 is
            // made here -- A bytes (loaded a region ago) converted and written, B by DMA -- and the A bytes of kb_+2 are requested
#define V2_REGION(ST, kb_, first, more, more2)                                     \
    do {                                                                           \
        if (more) a_store(lds[(ST) ^ 1]);                                          \
        V2_READ_X(ST);                                                             \
        if (!(first)) V2_MFMA(ay, by, 2, 4);                                       \
        V2_READ_Y(ST);                                                             \
        if (!SPLIT) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                     \
            if (!(first)) __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);      \
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                     \
            __builtin_amdgcn_sched_barrier(0);                                     \
        }                                                                          \
        /* the DMA of the next stage's Wu rows and the request for the A bytes after next go out BEHIND this region's last LDS  */ \
        /* read: hipcc puts an s_waitcnt vmcnt(0) in front of the first LDS read that follows an LDS-DMA here (measured: 0.81   */ \
        /* of the peak with the DMA at the top of the region, every wave waiting a memory latency per K block)                  */ \
        if (more) b_dma((kb_) + 1, lds[(ST) ^ 1]);                                  \
        if (more2) a_load((kb_) + 2);                                              \
        V2_MFMA(ax, bx, 0, 4);                                                     \
        V2_MFMA(ay, by, 0, 2);                                                     \
        if (!SPLIT) __builtin_amdgcn_sched_barrier(0);                             \
        __syncthreads();                                                           \
    } while (0)
            a_load(0);
            __syncthreads();   // the previous chunk's readers are done with both stage buffers
            a_store(lds[0]);
            b_dma(0, lds[0]);
            if (nkb > 1) a_load(1);
            __syncthreads();
            V2_REGION(0, 0, true, nkb > 1, nkb > 2);
            int i = 1;
            for (; i + 2 <= nkb; i += 2) {
                V2_REGION(1, i, false, true, i + 2 < nkb);
                V2_REGION(0, i + 1, false, i + 2 < nkb, i + 3 < nkb);
            }
            if (i < nkb) V2_REGION(1, i, false, false, false);
            V2_MFMA(ay, by, 2, 4);
#undef V2_REGION
#undef V2_MFMA
#undef V2_READ_X
#undef V2_READ_Y
            // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const long r = row0 + wr * 64 + m * 16 + g + 4 * q;
                    const int8_t* mr = A8 + r * lda + (long)ct * GF_T + wc * 64 + i16;
                    double sdot = 0.0;
#pragma unroll
                    for (int n = 0; n < 4; n++) sdot += acc[m][n][q] * (double)mr[n * 16];
                    sdot += __shfl_xor(sdot, 1);
                    sdot += __shfl_xor(sdot, 2);
                    sdot += __shfl_xor(sdot, 4);
                    sdot += __shfl_xor(sdot, 8);
                    if (SPLIT) {
                        if (i16 == 0)
                            partial[((((long)rb * n_coltiles + ct) * nchunk_max + c) * 2 + wc) * GF_T + wr * 64 + m * 16 + g + 4 * q] = sdot;
                    } else if (i16 == 0) {
                        Psum[wc][wr * 64 + m * 16 + g + 4 * q] += sdot;   // this lane alone owns the element: same chain as a register
                    }
                }
        }
    }
    if (!SPLIT) {
        __syncthreads();
        if (t < 128) out[row0 + t] = Psum[0][t] + Psum[1][t];
    }
}

// device tile lists, cached per (device, tiles per side, kind): 0 = all tiles, 1 = row tile <= column tile, 2 = row tile > column tile
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
static std::map<std::tuple<int, int, int, int, int>, std::pair<int*, long>> g_gemm_lists;
static std::mutex g_gemm_lists_mutex;
// kind: 0 = all tiles, 1 = row tile <= column tile, 2 = row tile > column tile; restricted to row tiles [rt0, rt1)
static int gemm_tile_list(eagle_ctx* ctx, int nt, int kind, int rt0, int rt1, const int** out, long* count) {
    std::lock_guard<std::mutex> lock(g_gemm_lists_mutex);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto key = std::make_tuple(dev, nt, kind, rt0, rt1);
    auto it = g_gemm_lists.find(key);
    if (it != g_gemm_lists.end()) { *out = it->second.first; *count = it->second.second; return EAGLE_OK; }
    std::vector<int> h;
    const int base_kind = kind & 15;
    if (kind & 16) {  // 8 x 8 super-tiles (experiment, tune 23 / 24): 64 consecutive entries share 8 row and 8 column panels
        for (int si = rt0; si < rt1; si += 8)
            for (int sj = 0; sj < nt; sj += 8)
                for (int i = si; i < si + 8 && i < rt1; i++)
                    for (int j = sj; j < sj + 8 && j < nt; j++)
                        if (base_kind == 0 || (base_kind == 1 && i <= j) || (base_kind == 2 && i > j)) h.push_back((i << 16) | j);
    } else
    for (int i = rt0; i < rt1; i++)   // row tile outer: consecutive workgroups share an A row panel (measured 4 % faster than column-major)
        for (int j = 0; j < nt; j++)
            if (base_kind == 0 || (base_kind == 1 && i <= j) || (base_kind == 2 && i > j)) h.push_back((i << 16) | j);
    int* d = nullptr;
    if (!h.empty()) {
        hipError_t e = hipMalloc((void**)&d, h.size() * sizeof(int));
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "gemm tile list alloc");
        e = hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return eagle_fail_hip(ctx, e, "gemm tile list copy"); }
    }
    g_gemm_lists[key] = std::make_pair(d, (long)h.size());
    *out = d;
    *count = (long)h.size();
    return EAGLE_OK;
}

// Tile list of k_gemm_f64_dma: 256-row tiles starting at 128-row tile rt0, rt0 + 2, ... (the last may be half inside [rt0, rt1)),
// 128-column tiles; kind 0 = all, 1 = tiles holding an element on or above the diagonal (column tile >= first row tile),
// 2 = the others, 3 = tiles holding an element on or below the diagonal (column tile <= last 128-row tile of the tile).
// Row tile outer: consecutive workgroups share an A row panel.
static int gemm_dma_tile_list(eagle_ctx* ctx, int nt, int kind, int rt0, int rt1, const int** out, long* count) {
    std::lock_guard<std::mutex> lock(g_gemm_lists_mutex);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto key = std::make_tuple(dev, nt, kind | 256, rt0, rt1);
    auto it = g_gemm_lists.find(key);
    if (it != g_gemm_lists.end()) { *out = it->second.first; *count = it->second.second; return EAGLE_OK; }
    std::vector<int> h;
    for (int i = rt0; i < rt1; i += 2)
        for (int j = 0; j < nt; j++)
            if (kind == 0 || (kind == 1 && j >= i) || (kind == 2 && j < i) || (kind == 3 && j <= i + 1)) h.push_back((i << 16) | j);
    int* d = nullptr;
    if (!h.empty()) {
        hipError_t e = hipMalloc((void**)&d, h.size() * sizeof(int));
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "gemm tile list alloc");
        e = hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return eagle_fail_hip(ctx, e, "gemm tile list copy"); }
    }
    g_gemm_lists[key] = std::make_pair(d, (long)h.size());
    *out = d;
    *count = (long)h.size();
    return EAGLE_OK;
}
// C[rows rt0*128 .. rt1*128) = A * B on k_gemm_f64_dma (n_pad % 256 == 0); trans: C^T is written instead (C[col][row]).
static int gemm_f64_dma_tiles(eagle_ctx* ctx, const double* A, const double* B, double* C, long np, int kind, const int* skip_if, int skip_val,
                              void* stream, int rt0, int rt1, bool trans) {
    const int* tiles = nullptr;
    long count = 0;
    int rc = gemm_dma_tile_list(ctx, (int)(np / 128), kind, rt0, rt1, &tiles, &count);
    if (rc || count == 0) return rc;
    const long slots = ctx->cu_count > 0 ? ctx->cu_count : 256;  // one workgroup per CU
    const long n_main = count / slots * slots, tail = count - n_main;
    const long nkb = np / GF_BK;
    int split = 1;
    if (tail > 0) {
        double best = 1.0;
        for (int sp = 2; sp <= 8; sp++) {
            if (nkb / sp < 8 || (double)tail * sp * G2_TM * G2_TN * 8.0 > 256e6) break;
            const double cost = (double)((tail * sp + slots - 1) / slots) / sp;
            if (cost < best - 1e-9) { best = cost; split = sp; }
        }
    }
    double* scratch = nullptr;
    if (split > 1) {
        const size_t need = (size_t)tail * split * G2_TM * G2_TN * sizeof(double);
        if (need > ctx->gemm_scratch_cap) {
            if (ctx->gemm_scratch) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(ctx->gemm_scratch); ctx->gemm_scratch = nullptr; ctx->gemm_scratch_cap = 0; }
            hipError_t e = hipMalloc(&ctx->gemm_scratch, need);
            if (e != hipSuccess) return eagle_fail_hip(ctx, e, "gemm scratch");
            ctx->gemm_scratch_cap = need;
        }
        scratch = (double*)ctx->gemm_scratch;
    }
    const long blocks = split > 1 ? n_main + tail * split : count;
    const long row_end = (long)rt1 * 128;
    const int nm = (int)(split > 1 ? n_main : count);
    if (trans) {
        hipLaunchKernelGGL((k_gemm_f64_dma<true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, A, np, B, np, C, np, np, row_end, tiles, nm, split,
                           scratch, skip_if, skip_val);
        if (split > 1)
            hipLaunchKernelGGL((k_gemm_f64_dma_tail<true>), dim3((unsigned)(tail * 32)), dim3(256), 0, (hipStream_t)stream, scratch, tiles + n_main, split, C, np,
                               row_end, skip_if, skip_val);
    } else {
        hipLaunchKernelGGL((k_gemm_f64_dma<false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, A, np, B, np, C, np, np, row_end, tiles, nm, split,
                           scratch, skip_if, skip_val);
        if (split > 1)
            hipLaunchKernelGGL((k_gemm_f64_dma_tail<false>), dim3((unsigned)(tail * 32)), dim3(256), 0, (hipStream_t)stream, scratch, tiles + n_main, split, C, np,
                               row_end, skip_if, skip_val);
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

static int gemm_f64_tiles(eagle_ctx* ctx, const double* A, const double* B, double* C, long np, int kind, const int* skip_if, int skip_val,
                          void* stream, int rt0 = 0, int rt1 = -1) {
    if (np % GF_T || np <= 0 || np / GF_T > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "gemm_f64: size must be a multiple of 128");
    // n_pad % 256 == 0 (every size eagle_pad produces): the 256 x 128 LDS-DMA kernel; tune 21..27 keep the 128 x 128 forms for A/B runs
    if (np % 256 == 0 && 256 * np * 8 < 2147483648L && !(ctx->tune >= 21 && ctx->tune <= 27))
        return gemm_f64_dma_tiles(ctx, A, B, C, np, kind, skip_if, skip_val, stream, rt0, rt1 < 0 ? (int)(np / GF_T) : rt1, false);
    const int* tiles = nullptr;
    long count = 0;
    if (rt1 < 0) rt1 = (int)(np / GF_T);
    const int tune = ctx->tune;
    const bool super_tiles = tune == 23 || tune == 24;
    int rc = gemm_tile_list(ctx, (int)(np / GF_T), kind | (super_tiles ? 16 : 0), rt0, rt1, &tiles, &count);
    if (rc || count == 0) return rc;
    const long slots = 2L * (ctx->cu_count > 0 ? ctx->cu_count : 256);  // resident workgroups (2 per CU)
    const long n_main = count / slots * slots, tail = count - n_main;
    const long nkb = np / GF_BK;
    int split = 1;
    if (tail > 0) {
        double best = 1.0;
        for (int sp = 2; sp <= 8; sp++) {
            if (nkb / sp < 8 || (double)tail * sp * GF_T * GF_T * 8.0 > 160e6) break;
            const double cost = (double)((tail * sp + slots - 1) / slots) / sp;
            if (cost < best - 1e-9) { best = cost; split = sp; }
        }
    }
    double* scratch = nullptr;
    if (split > 1) {
        const size_t need = (size_t)tail * split * GF_T * GF_T * sizeof(double);
        if (need > ctx->gemm_scratch_cap) {
            if (ctx->gemm_scratch) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(ctx->gemm_scratch); ctx->gemm_scratch = nullptr; ctx->gemm_scratch_cap = 0; }
            hipError_t e = hipMalloc(&ctx->gemm_scratch, need);
            if (e != hipSuccess) return eagle_fail_hip(ctx, e, "gemm scratch");
            ctx->gemm_scratch_cap = need;
        }
        scratch = (double*)ctx->gemm_scratch;
    }
    const long blocks = split > 1 ? n_main + tail * split : count;
#define GEMM_LAUNCH(LDA_, X_) hipLaunchKernelGGL((k_gemm_f64_list<LDA_, X_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, np, B, np, C, np, np, tiles, \
                       (int)(split > 1 ? n_main : count), split, scratch, skip_if, skip_val)
    switch (tune) {  // 21..26: schedule experiments of tools/bench_gemm.py (profiles/r02_gemm_ab.txt); anything else: the shipped form
        case 21: GEMM_LAUNCH(18, 0); break;      // round 1's A pitch, four waves
        case 22: GEMM_LAUNCH(24, 0); break;
        case 23: GEMM_LAUNCH(GF_LDA, 1); break;  // XCD-contiguous 8 x 8 super-tile order: neutral
        case 24: GEMM_LAUNCH(18, 1); break;
        case 26: GEMM_LAUNCH(GF_LDA, 0); break;  // four waves per tile, pitch 20 (shipped until the eight-wave form)
        default:  // (tune 25, and sizes that are not a multiple of 256) eight waves per tile: 50.4 -> 48.4 ms for W at n = 10,000 (66.6 TF incl. v, symmetry check, fold), same bits
            hipLaunchKernelGGL((k_gemm_f64_list8<GF_LDA>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, A, np, B, np, C, np, np, tiles,
                               (int)(split > 1 ? n_main : count), split, scratch, skip_if, skip_val);
    }
#undef GEMM_LAUNCH
    if (split > 1)
        hipLaunchKernelGGL(k_gemm_f64_tail, dim3((unsigned)(tail * 16)), dim3(256), 0, (hipStream_t)stream, scratch, tiles + n_main, split, C, np, skip_if,
                           skip_val);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_gemm_f64(eagle_ctx* ctx, const double* A, const double* B, double* C, long np, void* stream) {
    return gemm_f64_tiles(ctx, A, B, C, np, 0, nullptr, 0, stream);
}
// Upper tiles always; the lower ones only when *sym == 0 (k_fold_upper then takes both halves from memory).
static int gemm_f64_upper(eagle_ctx* ctx, const double* A, const double* B, double* C, long np, const int* sym, void* stream) {
    int rc = gemm_f64_tiles(ctx, A, B, C, np, 1, nullptr, 0, stream);
    if (rc) return rc;
    return gemm_f64_tiles(ctx, A, B, C, np, 2, sym, 1, stream);
}

// W = S (V S) in three steps, so that the caller can overlap the PCIe upload of V with the first product
// (eagle_api.cpp scan_range: V arrives in row blocks on the loader stream; the device copy of S is already there):
//   _begin   v = S a_hat (needs Sa, a_hat only).
//   _vrows   X[r0, r1) = Va[r0, r1) * Sa for one block of rows of Va's image (r0, r1 multiples of 128) into tmp -- needs only
//            those rows of Va.  For SYMMETRIC operands (every Eagle run: MMt^-1/2 and a variance matrix) X = V S.
//   _finish  symmetry check of Sa, Va; if symmetric: the 256 x 128 tiles of W' = Sa * X that hold an element on or below the
//            diagonal, stored TRANSPOSED into Wu (W is symmetric: only one triangle is needed, the fold doubles it);
//            if not: the general products Xt = Sa * Va, Wt = Xt * Sa (both triangles) replace everything -- launched always,
//            dropped on the device by the flag the check left, so there is no host round trip either way; then the fold.
// For exactly symmetric images the symmetric pipeline forms, tile for tile, the very sums of the general products (the same
// products in the same k order, a * b = b * a); for images symmetric only to rounding the two differ at rounding level, as the
// "upper tiles doubled" shortcut always did.  eagle_dev_scan_operands = the three steps on resident operands, so a device-resident
// step (bench.py) and the reference-shaped call (upload pipelined) return the same bits.
extern "C" int eagle_dev_scan_operands_begin(eagle_ctx* ctx, const double* Sa, const double* ahat, long n, long n_pad, double* v_out, double* tmp,
                                             void* stream) {
    if (n_pad % GF_T || n > n_pad) return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_operands: bad padding");
    hipStream_t s = (hipStream_t)stream;
    // v = S a_hat ; Sa is the row-major image of S^T, so v_i = sum_j Sa[j][i] a_hat[j]
    hipError_t e = hipMemsetAsync(v_out, 0, sizeof(double) * n_pad, s);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "scan_operands memset");
    // (tmp is free until the first product writes it: 8 partial vectors live there)
    hipLaunchKernelGGL(k_colgemv_part, dim3((unsigned)(n_pad / 64), 8), dim3(256), 0, s, Sa, n, n_pad, ahat, tmp);
    hipLaunchKernelGGL(k_colgemv_sum, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, tmp, n_pad, 8, v_out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
static bool scan_operands_pipelined(const eagle_ctx* ctx, long n_pad) {
    return n_pad % 256 == 0 && 256 * n_pad * 8 < 2147483648L && !(ctx->tune >= 21 && ctx->tune <= 27);
}
extern "C" int eagle_dev_scan_operands_vrows(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, long row0, long row1, double* tmp,
                                             void* stream) {
    if (n_pad % GF_T || row0 % GF_T || row1 % GF_T || row0 < 0 || row1 > n_pad || row0 >= row1)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_operands_vrows: bad padding or row block");
    if (!scan_operands_pipelined(ctx, n_pad)) return EAGLE_OK;  // (the 128 x 128 kernels of the A/B runs: all of it in _finish)
    return gemm_f64_dma_tiles(ctx, Va, Sa, tmp, n_pad, 0, nullptr, 0, stream, (int)(row0 / GF_T), (int)(row1 / GF_T), false);
}
extern "C" int eagle_dev_scan_operands_finish(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, double* Wu_out, double* tmp, void* stream) {
    if (n_pad % GF_T) return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_operands: bad padding");
    hipStream_t s = (hipStream_t)stream;
    int* sym = (int*)((char*)eagle_ctx_scratch(ctx) + EAGLE_SCR_SYM);
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, sym, 1);
    dim3 g32((unsigned)(n_pad / 32), (unsigned)(n_pad / 32));
    hipLaunchKernelGGL(k_sym_check, g32, dim3(256), 0, s, Sa, Va, n_pad, sym);
    LAUNCH_CHECK(ctx);
    int rc;
    const int nt = (int)(n_pad / GF_T);
    if (scan_operands_pipelined(ctx, n_pad)) {
        // symmetric: W' = Sa * X on the tiles with column tile <= last row tile ... i.e. NOT strictly above the diagonal, transposed
        // into the upper tiles of Wu (skipped when the check failed)
        rc = gemm_f64_dma_tiles(ctx, Sa, tmp, Wu_out, n_pad, 3, sym, 0, stream, 0, nt, true);
        if (rc) return rc;
        // not symmetric: the general products (skipped when the check passed)
        rc = gemm_f64_dma_tiles(ctx, Sa, Va, tmp, n_pad, 0, sym, 1, stream, 0, nt, false);
        if (!rc) rc = gemm_f64_dma_tiles(ctx, tmp, Sa, Wu_out, n_pad, 0, sym, 1, stream, 0, nt, false);
        if (rc) return rc;
    } else {
        // Xt = (V S)^T = S^T V^T = Sa * Va ; Wt = (S X)^T = X^T S^T = Xt * Sa        (row-major images)
        rc = eagle_dev_gemm_f64(ctx, Sa, Va, tmp, n_pad, stream);
        if (rc) return rc;
        rc = gemm_f64_upper(ctx, tmp, Sa, Wu_out, n_pad, sym, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_fold_upper, g32, dim3(256), 0, s, Wu_out, n_pad, sym);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_scan_operands_w_f64(eagle_ctx* ctx, const double* Sa, const double* Va, long n_pad, double* Wu_out, double* tmp, void* stream) {
    int rc = EAGLE_OK;
    // the same 1024-row blocks the reference-shaped call uses while V arrives: each launch cuts its own split-K tail, so only
    // the same blocking gives the same sums
    for (long b0 = 0; b0 < n_pad && !rc; b0 += EAGLE_VROWS_BLOCK)
        rc = eagle_dev_scan_operands_vrows(ctx, Sa, Va, n_pad, b0, b0 + EAGLE_VROWS_BLOCK < n_pad ? b0 + EAGLE_VROWS_BLOCK : n_pad, tmp, stream);
    if (!rc) rc = eagle_dev_scan_operands_finish(ctx, Sa, Va, n_pad, Wu_out, tmp, stream);
    return rc;
}
// Which engine forms W: the int8 digit-slice products (eagle_w8.hip) for a digit-slice scan from 4,096 padded individuals up
// (eagle_set_w_mode: 0 = never, 2 = at any size), unless they decline; the fp64 GEMM otherwise.
bool eagle_w8_wanted(const eagle_ctx* ctx, long n_pad) {
    return ctx->scan_mode == 1 && (ctx->w_mode == 2 || (ctx->w_mode == 1 && n_pad >= 4096));
}
extern "C" int eagle_dev_scan_operands(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n,
                                       long n_pad, double* v_out, double* Wu_out, double* tmp, void* stream) {
    if (eagle_w8_wanted(ctx, n_pad)) {
        const int r8 = eagle_dev_scan_operands_w8(ctx, Sa, Va, ahat, n, n_pad, v_out, Wu_out, tmp, stream);
        if (r8 != 1) return r8;   // done, or failed; 1: declined -- the fp64 products below
    } else {
        ctx->w8_info = W8Info();
        ctx->w8_info.declined = 7;
    }
    ctx->w8_active = false;
    int rc = eagle_dev_scan_operands_begin(ctx, Sa, ahat, n, n_pad, v_out, tmp, stream);
    if (!rc) rc = eagle_dev_scan_operands_w_f64(ctx, Sa, Va, n_pad, Wu_out, tmp, stream);
    return rc;
}

// The same operands with the n^3 work shared between ranks (marker-sharded multi-GPU run): this call computes v = S a_hat
// and only the rows [row0, row1) (multiples of 128) of W^T's row-major image -- Xt rows = Sa rows * Va, Wt rows = Xt rows * Sa --
// into Wt_out[row0.., :], full rows, no symmetry shortcut.  The caller all-gathers the row blocks and then folds the
// complete image with eagle_dev_fold_upper.
extern "C" int eagle_dev_scan_operands_rows(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long n_pad,
                                            long row0, long row1, double* v_out, double* Wt_out, double* tmp, void* stream) {
    ctx->w8_active = false;
    if (n_pad % GF_T || n > n_pad || row0 % GF_T || row1 % GF_T || row0 < 0 || row1 > n_pad || row0 >= row1)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_operands_rows: bad padding or row block");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(v_out, 0, sizeof(double) * n_pad, s);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "scan_operands memset");
    // (tmp is free until the first product writes it: 8 partial vectors live there)
    hipLaunchKernelGGL(k_colgemv_part, dim3((unsigned)(n_pad / 64), 8), dim3(256), 0, s, Sa, n, n_pad, ahat, tmp);
    hipLaunchKernelGGL(k_colgemv_sum, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, tmp, n_pad, 8, v_out);
    LAUNCH_CHECK(ctx);
    const int rt0 = (int)(row0 / GF_T), rt1 = (int)(row1 / GF_T);
    int rc = gemm_f64_tiles(ctx, Sa, Va, tmp, n_pad, 0, nullptr, 0, stream, rt0, rt1);
    if (rc) return rc;
    return gemm_f64_tiles(ctx, tmp, Sa, Wt_out, n_pad, 0, nullptr, 0, stream, rt0, rt1);
}
// In-place fold of a COMPLETE W image: Wu[j][k] = W[j][k] + W[k][j] (j<k), W[k][k], 0 below.
extern "C" int eagle_dev_fold_upper(eagle_ctx* ctx, double* W, long n_pad, void* stream) {
    if (ctx->w8_Wu == W) ctx->w8_active = false;
    if (n_pad % 32) return eagle_fail(ctx, EAGLE_ERR_ARG, "fold_upper: bad padding");
    int* sym = (int*)((char*)eagle_ctx_scratch(ctx) + EAGLE_SCR_SYM);
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, (hipStream_t)stream, sym, 0);
    dim3 g32((unsigned)(n_pad / 32), (unsigned)(n_pad / 32));
    hipLaunchKernelGGL(k_fold_upper, g32, dim3(256), 0, (hipStream_t)stream, W, n_pad, sym);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

// the same product with the j range cut into 8 parts (more workgroups: 2x the rate of the one-kernel form at n = 10,000);
// part: 8 n_pad doubles of scratch
extern "C" int eagle_dev_colgemv_parts(eagle_ctx* ctx, const double* At, long n, long n_pad, const double* x, double* out, double* part, void* stream) {
    if (n_pad % 64) return eagle_fail(ctx, EAGLE_ERR_ARG, "colgemv: bad padding");
    hipLaunchKernelGGL(k_colgemv_part, dim3((unsigned)(n_pad / 64), 8), dim3(256), 0, (hipStream_t)stream, At, n, n_pad, x, part);
    hipLaunchKernelGGL(k_colgemv_sum, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, n_pad, 8, out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_colgemv(eagle_ctx* ctx, const double* At, long n, long n_pad, const double* x, double* out,
                                 void* stream) {
    if (n_pad % 64) return eagle_fail(ctx, EAGLE_ERR_ARG, "colgemv: bad padding");
    hipLaunchKernelGGL(k_colgemv, dim3((unsigned)(n_pad / 64)), dim3(256), 0, (hipStream_t)stream, At, n, n_pad, x, out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}


// ------------------------------------------------------------------------------------------------
// Genotype pass on the int8 MFMA:  a_i = scale * sum_k m_ik v_k   and   d_i = sum_k m_ik^2 w_k   in one sweep over the
// genotype bytes (E/src/calculate_a_and_vara_rcpp.cpp:90-91, calculate_reduced_a_rcpp.cpp:83-84; d is the diagonal term
// of the int8 vara path).  HBM-bound: L*n genotype bytes, read once.
//
// The fp64 vectors are turned into exact fixed point first (k_slice_vec): Q_k = rint(v_k * 2^(62-e)), max|v| < 2^e, as 8
// balanced base-256 digits (peeled least significant first), so v_k = 2^(e-62) sum_s 256^s D_s[k] up to 2^(e-63) for
// elements more than 10 binades below the largest and exactly otherwise.  B = [D_0..D_7 of v ; D_0..D_7 of w] is a
// 16 x n_pad int8 matrix; v_mfma_i32_16x16x64_i8 gives P[i][s] = sum_k m_ik D_s[k] exactly (|P| <= 128 n), and
// a_i = 2^(e-62) sum_s 256^s P[i][s] is assembled in int64 halves and rounded once.  m^2 = m & 1 for m in {-1,0,1}.
// Two MFMAs (32 matrix cycles) per KiB of genotypes leave the kernel on the HBM stream: 0.478 ms for a and d at C2 (5.35 TB/s;
// a alone 0.464 ms, 5.5 TB/s), where the fp64 VALU form it replaced (int8 -> fp64 convert + FMA per genotype, vectors in
// LDS) took 0.62 ms for the pair (tools/bench_gemv.py history in DESIGN.md 3.4).  Nontemporal loads cost 12 %.
// Wave = 16 markers x all k; lane (r = l&15, q = l>>4) loads the 16 bytes k = 64 step + 16 q .. of marker r and the same
// k range of digit row r from LDS (16-byte chunks XOR-swizzled with r: conflict-free without padding, so that
// n_pad = 10240 fits the 160 KiB exactly).
// ------------------------------------------------------------------------------------------------
#define GV_MAXN 10240
// B rows: 0..7 digits of v, 8..15 digits of w, and (x != NULL) 16..23 digits of a third vector x, 24..31 zero.
__global__ __launch_bounds__(1024) void k_slice_vec(const double* __restrict__ v, const double* __restrict__ w, const double* __restrict__ x,
                                                    int n_pad, int8_t* __restrict__ B, int* __restrict__ exps) {
    __shared__ double red[3][16];
    __shared__ int ex[3];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double mv = 0.0, mw = 0.0, mx = 0.0;
    for (int k = t; k < n_pad; k += 1024) {
        mv = fmax(mv, fabs(v[k]));
        if (w) mw = fmax(mw, fabs(w[k]));
        if (x) mx = fmax(mx, fabs(x[k]));
    }
    for (int o = 32; o > 0; o >>= 1) { mv = fmax(mv, __shfl_xor(mv, o)); mw = fmax(mw, __shfl_xor(mw, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { red[0][wv] = mv; red[1][wv] = mw; red[2][wv] = mx; }
    __syncthreads();
    if (t < 3) {
        double m = 0.0;
        for (int i = 0; i < 16; i++) m = fmax(m, red[t][i]);
        int e = 0;
        if (m > 0.0 && m < INFINITY) (void)frexp(m, &e);  // m = f 2^e, f in [0.5,1)
        ex[t] = e;
        exps[t] = e;
    }
    __syncthreads();
    for (int idx = t; idx < (x ? 4 : 2) * n_pad; idx += 1024) {
        const int which = idx / n_pad, k = idx - which * n_pad;
        long long Q = 0;
        if (which == 0) Q = llrint(ldexp(v[k], 62 - ex[0]));
        else if (which == 1) { if (w) Q = llrint(ldexp(w[k], 62 - ex[1])); }
        else if (which == 2) Q = llrint(ldexp(x[k], 62 - ex[2]));
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const long long d = ((Q + 128) & 255) - 128;
            Q = (Q - d) >> 8;
            B[(long)(8 * which + s) * n_pad + k] = (int8_t)d;
        }
    }
}

typedef int gv_i32x4 __attribute__((ext_vector_type(4)));
#define GV_U 8  /* 16-byte loads per lane in flight: 8 KiB per wave, 128 KiB per CU */
// X3: a third vector x (digit rows 16..23 of B, a second 16-row LDS image): out_x = Mt8 x in the same sweep (n_pad <= 5120).
template <bool SQ, bool X3>
__global__ __launch_bounds__(1024) void k_gemv_mfma(const int8_t* __restrict__ Mt8, long L_pad, int n_pad, long ld,
                                                    const int8_t* __restrict__ B, const int* __restrict__ exps, double scale,
                                                    double* __restrict__ out_a, double* __restrict__ out_d, double* __restrict__ out_x,
                                                    int accumulate) {
    extern __shared__ __attribute__((aligned(16))) int8_t lB[];  // [16 or 32][n_pad], chunk c of row r stored at chunk c ^ (r & 15)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nchunk = n_pad >> 4;
    for (int idx = t; idx < (X3 ? 32 : 16) * nchunk; idx += 1024) {
        const int r = idx / nchunk, c = idx - r * nchunk;
        *(gv_i32x4*)(lB + (long)r * n_pad + ((c ^ (r & 15)) << 4)) = *(const gv_i32x4*)(B + (long)r * n_pad + (c << 4));
    }
    __syncthreads();
    const int r = lane & 15, q = lane >> 4;
    const int8_t* brow = lB + (long)r * n_pad;
    const int8_t* brow2 = lB + (long)(16 + r) * n_pad;
    const double sa = scale * ldexp(1.0, exps[0] - 62), sd = ldexp(1.0, exps[1] - 62), sx = X3 ? ldexp(1.0, exps[2] - 62) : 0.0;
    const long ngroups = L_pad >> 4;
    for (long g = (long)blockIdx.x * 16 + wv; g < ngroups; g += (long)gridDim.x * 16) {
        const int8_t* ap = Mt8 + (g * 16 + r) * ld + (q << 4);
        gv_i32x4 accA = {0, 0, 0, 0}, accD = {0, 0, 0, 0}, accX = {0, 0, 0, 0};
        const int nsteps = n_pad >> 6;  // n_pad % 256 == 0 in every caller: a multiple of 4
        for (int st = 0; st < nsteps; st += GV_U) {
            gv_i32x4 a[GV_U], b[GV_U];
#pragma unroll
            for (int u = 0; u < GV_U; u++)
                a[u] = st + u < nsteps ? *(const gv_i32x4*)(ap + ((st + u) << 6)) : (gv_i32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < GV_U; u++)
                b[u] = st + u < nsteps ? *(const gv_i32x4*)(brow + (((4 * (st + u) + q) ^ r) << 4)) : (gv_i32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < GV_U; u++) {
                accA = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], b[u], accA, 0, 0, 0);
                if (SQ) {
                    const gv_i32x4 a2 = a[u] & (gv_i32x4){0x01010101, 0x01010101, 0x01010101, 0x01010101};
                    accD = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b[u], accD, 0, 0, 0);
                }
                if (X3) {
                    const gv_i32x4 b2 = st + u < nsteps ? *(const gv_i32x4*)(brow2 + (((4 * (st + u) + q) ^ r) << 4)) : (gv_i32x4){0, 0, 0, 0};
                    accX = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], b2, accX, 0, 0, 0);
                }
            }
        }
        // lane holds P[row = 4q + e][col = r]: cols 0..7 of accA are the digits of v, cols 8..15 of accD those of w
        const int s = r & 7;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const long long val = (r < 8) ? (long long)accA[e] : (long long)(SQ ? accD[e] : 0);
            long long hi = s >= 4 ? val << (8 * (s - 4)) : 0, lo = s < 4 ? val << (8 * s) : 0;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { hi += __shfl_xor(hi, o); lo += __shfl_xor(lo, o); }
            if (s == 0) {
                const long row = g * 16 + 4 * q + e;
                const double x = (double)hi * 4294967296.0 + (double)lo;
                if (r == 0) { if (out_a) out_a[row] = accumulate ? out_a[row] + sa * x : sa * x; }
                else if (SQ) out_d[row] = accumulate ? out_d[row] + sd * x : sd * x;
            }
            if (X3) {  // cols 0..7 of accX: the digits of the third vector
                const long long vx = (r < 8) ? (long long)accX[e] : 0;
                long long hx = s >= 4 ? vx << (8 * (s - 4)) : 0, lx = s < 4 ? vx << (8 * s) : 0;
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { hx += __shfl_xor(hx, o); lx += __shfl_xor(lx, o); }
                if (r == 0) {
                    const double xv = sx * ((double)hx * 4294967296.0 + (double)lx);
                    out_x[g * 16 + 4 * q + e] = accumulate ? out_x[g * 16 + 4 * q + e] + xv : xv;
                }
            }
        }
    }
}

// a = scale * Mt8 v, d_i = sum_j Mt8[i][j]^2 w[j] and (x != NULL) out_x = Mt8 x in one pass over the genotypes (w may be
// NULL: no d; out_a may be NULL: no a).  Individuals beyond GV_MAXN columns are taken in further sweeps of GV_MAXN columns
// that add into the outputs (5120 columns per sweep when the third vector's second LDS image rides along).
extern "C" int eagle_dev_gemv3_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v,
                                  const double* w, const double* x, double scale, double* out_a, double* out_d, double* out_x, void* stream) {
    if (L_pad % 16 || n_pad % 256 || ld % 16 || n_pad > ld) return eagle_fail(ctx, EAGLE_ERR_ARG, "gemv_i8: layout contract violated (L_pad % 16, n_pad % 256, ld % 16)");
    if (L_pad == 0 || n_pad == 0) return EAGLE_OK;
    if (!ctx->gemv_ws) {
        hipError_t e = hipMalloc(&ctx->gemv_ws, 32 * GV_MAXN + 256);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "gemv workspace");
    }
    int8_t* B = (int8_t*)ctx->gemv_ws;
    int* exps = (int*)(B + 32 * GV_MAXN);
    if (!ctx->attr_gemv) {  // per device
        hipError_t e = hipFuncSetAttribute((const void*)k_gemv_mfma<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemv_mfma<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemv_mfma<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "hipFuncSetAttribute");
        ctx->attr_gemv = true;
    }
    long blocks = 256;
    const long groups = L_pad / 16;
    if (blocks > (groups + 15) / 16) blocks = (groups + 15) / 16;
    if (x && w) {  // three vectors: two 16-row LDS images, 5120 individuals per sweep; every genotype byte is still read once
        for (long k0 = 0; k0 < n_pad; k0 += GV_MAXN / 2) {
            const int nk = (int)(n_pad - k0 < GV_MAXN / 2 ? n_pad - k0 : GV_MAXN / 2);
            hipLaunchKernelGGL(k_slice_vec, dim3(1), dim3(1024), 0, (hipStream_t)stream, v + k0, w + k0, x + k0, nk, B, exps);
            hipLaunchKernelGGL((k_gemv_mfma<true, true>), dim3((unsigned)blocks), dim3(1024), (size_t)32 * nk, (hipStream_t)stream, Mt8 + k0, L_pad, nk,
                               ld, B, exps, scale, out_a, out_d, out_x, k0 > 0);
        }
        LAUNCH_CHECK(ctx);
        return EAGLE_OK;
    }
    for (long k0 = 0; k0 < n_pad; k0 += GV_MAXN) {
        const int nk = (int)(n_pad - k0 < GV_MAXN ? n_pad - k0 : GV_MAXN);
        hipLaunchKernelGGL(k_slice_vec, dim3(1), dim3(1024), 0, (hipStream_t)stream, v + k0, w ? w + k0 : nullptr, (const double*)nullptr, nk, B, exps);
        if (w) hipLaunchKernelGGL((k_gemv_mfma<true, false>), dim3((unsigned)blocks), dim3(1024), (size_t)16 * nk, (hipStream_t)stream, Mt8 + k0, L_pad, nk, ld,
                                  B, exps, scale, out_a, out_d, (double*)nullptr, k0 > 0);
        else hipLaunchKernelGGL((k_gemv_mfma<false, false>), dim3((unsigned)blocks), dim3(1024), (size_t)16 * nk, (hipStream_t)stream, Mt8 + k0, L_pad, nk, ld,
                                B, exps, scale, out_a, out_d, (double*)nullptr, k0 > 0);
    }
    LAUNCH_CHECK(ctx);
    if (x) return eagle_dev_gemv3_i8(ctx, Mt8, L_pad, n_pad, ld, x, nullptr, nullptr, 1.0, out_x, nullptr, nullptr, stream);
    return EAGLE_OK;
}
extern "C" int eagle_dev_gemv2_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v,
                                  const double* w, double scale, double* out_a, double* out_d, void* stream) {
    return eagle_dev_gemv3_i8(ctx, Mt8, L_pad, n_pad, ld, v, w, nullptr, scale, out_a, out_d, nullptr, stream);
}

extern "C" int eagle_dev_gemv_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* v,
                                 double scale, double* out, void* stream) {
    return eagle_dev_gemv2_i8(ctx, Mt8, L_pad, n_pad, ld, v, nullptr, scale, out, nullptr, stream);
}

extern "C" int eagle_dev_vara_f64(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                  double* vara_out, void* stream) {
    return eagle_dev_vara_f64_gated(ctx, Mt8, L_pad, n_pad, ld, Wu, vara_out, nullptr, stream);
}
// run_if (device int, may be NULL): the launch is dropped on the device unless *run_if != 0.
extern "C" int eagle_dev_vara_f64_gated(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                        double* vara_out, const int* run_if, void* stream) {
    if (L_pad % GF_T || n_pad % GF_T || ld % 16 || n_pad > ld)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_f64: layout contract violated");
    if (L_pad == 0) return EAGLE_OK;
    dim3 grid((unsigned)(L_pad / GF_T));
    if (ctx->tune == 28)  // the round-2 kernel (A/B runs)
        hipLaunchKernelGGL((k_vara_f64<false>), grid, dim3(256), 0, (hipStream_t)stream, Mt8, ld, Wu, n_pad, vara_out, (int)(n_pad / GF_T), n_pad,
                           run_if, (double*)nullptr);
    else
        hipLaunchKernelGGL((k_vara_f64d<false>), grid, dim3(256), 0, (hipStream_t)stream, Mt8, ld, Wu, n_pad, vara_out, (int)(n_pad / GF_T), n_pad,
                           run_if, (double*)nullptr);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
// The same values for the first *count_dev rows (count <= rows_cap, a multiple of 128) of a compact row buffer, with the
// (column tile, K chunk) pairs spread over the chip (a few rows would otherwise sit on one CU): partial needs
// eagle_vara_f64_split_partial_doubles(rows_cap, n_pad) doubles.  Results go to out[dst_dev[r]] (dst_dev == NULL: out[r]).
extern "C" long eagle_vara_f64_split_partial_doubles(long rows_cap, long n_pad) {
    return (rows_cap / GF_T) * (n_pad / GF_T) * ((n_pad + GF_KC - 1) / GF_KC) * 2 * GF_T;
}
extern "C" int eagle_dev_vara_f64_split(eagle_ctx* ctx, const int8_t* rows8, long rows_cap, long n_pad, long ld, const double* Wu,
                                        const int* count_dev, const long* dst_dev, double* partial, double* out, void* stream) {
    if (rows_cap % GF_T || n_pad % GF_T || ld % 16 || n_pad > ld || n_pad / GF_T > 65535)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_f64_split: layout contract violated");
    if (rows_cap == 0) return EAGLE_OK;
    const int nct = (int)(n_pad / GF_T);
    dim3 grid((unsigned)(rows_cap / GF_T), (unsigned)nct, (unsigned)((n_pad + GF_KC - 1) / GF_KC));
    if (ctx->tune == 28)
        hipLaunchKernelGGL((k_vara_f64<true>), grid, dim3(256), 0, (hipStream_t)stream, rows8, ld, Wu, n_pad, (double*)nullptr, nct, n_pad, count_dev,
                           partial);
    else
        hipLaunchKernelGGL((k_vara_f64d<true>), dim3(grid.y, grid.z, grid.x), dim3(256), 0, (hipStream_t)stream, rows8, ld, Wu, n_pad, (double*)nullptr, nct,
                           n_pad, count_dev, partial);
    hipLaunchKernelGGL(k_vara_f64_sum, dim3((unsigned)(rows_cap / GF_T)), dim3(GF_T), 0, (hipStream_t)stream, partial, nct, n_pad, count_dev, dst_dev, out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_extract_col(eagle_ctx* ctx, const int8_t* M8, long n, long ld, long col, int* out, void* stream) {
    if (n <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_extract_col, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M8, n, ld, col, out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_zero_rows(eagle_ctx* ctx, double* a, double* vara, long L, const long* rows_dev, long nrows,
                                   long row_offset, void* stream) {
    if (nrows <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, vara, L,
                       rows_dev, nrows, row_offset);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

extern "C" int eagle_dev_tsq_argmax(eagle_ctx* ctx, const double* a, const double* vara, long L, double* tsq_out,
                                    eagle_best* best_dev, double* block_scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int nparts = (int)((L + 255) / 256);
    if (nparts > 1024) nparts = 1024;
    if (nparts < 1) nparts = 1;
    double* pv = block_scratch;
    long* pi = (long*)(block_scratch + 1024);
    hipLaunchKernelGGL(k_tsq_partial, dim3(nparts), dim3(256), 0, s, a, vara, L, tsq_out, pv, pi);
    hipLaunchKernelGGL(k_tsq_final, dim3(1), dim3(256), 0, s, pv, pi, nparts, best_dev);
    hipLaunchKernelGGL(k_tsq_near, dim3(nparts), dim3(256), 0, s, a, vara, L, best_dev);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}


// ------------------------------------------------------------------------------------------------
// Marker-file ingestion.
//   k_encode_ascii : int8 {-1,0,1} rows -> text lines '0','1','2' + '\n'   (what CreateASCIInospace.cpp:132-135 and
//                    createMt_ASCII_rcpp.cpp:104-118 write)
//   k_plink_code   : PLINK allele characters -> genotype codes, one thread per locus walking the individuals in file
//                    order, because the reference's allele table evolves row by row
//                    (E/src/CreateASCIInospace_PLINK.cpp:95-187)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_encode_ascii(const int8_t* __restrict__ in, long rows, long cols, long ld_in,
                                                      uint8_t* __restrict__ out) {
    const long row = blockIdx.y;
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c > cols) return;
    out[row * (cols + 1) + c] = c == cols ? (uint8_t)'\n' : (uint8_t)('0' + 1 + in[row * ld_in + c]);
}

extern "C" int eagle_dev_encode_ascii(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, uint8_t* out, void* stream) {
    if (rows <= 0) return EAGLE_OK;
    for (long r0 = 0; r0 < rows; r0 += 65535) {
        long nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((cols + 1 + 255) / 256), (unsigned)nr);
        hipLaunchKernelGGL(k_encode_ascii, grid, dim3(256), 0, (hipStream_t)stream, in + r0 * ld_in, nr, cols, ld_in, out + r0 * (cols + 1));
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

// chars: rows x 2L allele characters (one per allele, in file order) of individuals row0 .. row0+rows-1.
// alleles0/alleles1: the per-locus allele table, carried across calls.  out: int8 genotype - 1 at out[r*ld + i].
// first_err / first_missing: smallest row-major position (row*L + locus) of a third allele / of a missing allele.
__global__ __launch_bounds__(256) void k_plink_code(const uint8_t* __restrict__ chars, long rows, long L, long row0,
                                                    uint8_t* __restrict__ alleles0, uint8_t* __restrict__ alleles1,
                                                    int8_t* __restrict__ out, long ld, unsigned long long* __restrict__ first_err,
                                                    unsigned long long* __restrict__ first_missing) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    uint8_t a0 = alleles0[i], a1 = alleles1[i];
    bool dead = false;  // a third allele was met: the reference has returned, later rows are never coded
    for (long r = 0; r < rows; r++) {
        uint8_t c0 = chars[r * 2 * L + 2 * i], c1 = chars[r * 2 * L + 2 * i + 1];
        const bool miss = c0 == '0' || c1 == '0' || c0 == '-' || c1 == '-';
        if (row0 + r == 0) {                                   // :95-106
            if (miss) { a0 = 'I'; a1 = 'I'; } else { a0 = c0; a1 = c1; }
        }
        int8_t code = 0;                                       // het / missing -> '1' -> 0
        if (!dead) {
            if (miss) {                                        // :110-123
                atomicMin(first_missing, (unsigned long long)((row0 + r) * L + i));
                c0 = 'I';
                c1 = 'I';
            }
#pragma unroll
            for (int j = 1; j >= 0; --j) {                     // :127-165
                const uint8_t c = j ? c1 : c0;
                if (c != a0 && c != a1 && c != 'I') {
                    if (a0 == 'I') a0 = c;
                    else if (a1 == 'I') a1 = c;
                    else if (a0 == a1) a1 = c;
                    else if (!dead) { dead = true; atomicMin(first_err, (unsigned long long)((row0 + r) * L + i)); }
                }
            }
            if (c0 == 'I' || c1 == 'I' || c0 != c1) code = 0;  // :170-184
            else code = c0 == a0 ? -1 : 1;
        }
        out[r * ld + i] = code;
    }
    alleles0[i] = a0;
    alleles1[i] = a1;
}

extern "C" int eagle_dev_plink_code(eagle_ctx* ctx, const uint8_t* chars, long rows, long L, long row0, uint8_t* alleles0,
                                    uint8_t* alleles1, int8_t* out, long ld, unsigned long long* first_err,
                                    unsigned long long* first_missing, void* stream) {
    if (rows <= 0 || L <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_plink_code, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chars, rows, L, row0, alleles0,
                       alleles1, out, ld, first_err, first_missing);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// 2-bit sidecar of a genotype text file (the reference itself once bit-packed its genotypes,
// MP/RcppFunctions.cpp.gpu:224-358): genotype code g = m + 1 in {0,1,2}, code of column c in bits 2(c%4) of byte c/4.
//   k_pack2b   : int8 {-1,0,1} rows -> packed rows
//   k_unpack2b : packed row window -> int8 rows (first code at 2-bit index `shift` of the first byte); code 3 is invalid
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack2b(const int8_t* __restrict__ in, long rows, long cols, long ld_in, uint8_t* __restrict__ out,
                                                long row_bytes) {
    const long row = blockIdx.y;
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= row_bytes) return;
    unsigned v = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const long c = 4 * b + q;
        if (c < cols) v |= ((unsigned)(in[row * ld_in + c] + 1) & 3u) << (2 * q);
    }
    out[row * row_bytes + b] = (uint8_t)v;
}
__global__ __launch_bounds__(256) void k_unpack2b(const uint8_t* __restrict__ raw, long rows, long cols, long stride, int shift,
                                                  int8_t* __restrict__ out, long ld_out, int* __restrict__ bad) {
    const long row = blockIdx.y;
    const long c4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c4 >= ld_out) return;
    const uint8_t* src = raw + row * stride;
    uint32_t packed = 0;
    int nbad = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const long c = c4 + q;
        int v = 0;
        if (c < cols) {
            const long idx = c + shift;
            const int g = (src[idx >> 2] >> (2 * (idx & 3))) & 3;
            if (g == 3) nbad++;
            v = g - 1;
        }
        packed |= ((uint32_t)(uint8_t)(int8_t)v) << (8 * q);
    }
    *(uint32_t*)(out + row * ld_out + c4) = packed;
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int eagle_dev_pack2b(eagle_ctx* ctx, const int8_t* in, long rows, long cols, long ld_in, uint8_t* out, long row_bytes, void* stream) {
    if (rows <= 0 || row_bytes <= 0) return EAGLE_OK;
    for (long r0 = 0; r0 < rows; r0 += 65535) {
        long nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((row_bytes + 255) / 256), (unsigned)nr);
        hipLaunchKernelGGL(k_pack2b, grid, dim3(256), 0, (hipStream_t)stream, in + r0 * ld_in, nr, cols, ld_in, out + r0 * row_bytes, row_bytes);
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_unpack2b(eagle_ctx* ctx, const uint8_t* raw, long rows, long cols, long stride, int shift, int8_t* out,
                                  long ld_out, int* bad_dev, void* stream) {
    if (rows <= 0) return EAGLE_OK;
    if (ld_out % 4) return eagle_fail(ctx, EAGLE_ERR_ARG, "unpack2b: ld_out must be a multiple of 4");
    for (long r0 = 0; r0 < rows; r0 += 65535) {
        long nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid((unsigned)((ld_out / 4 + 255) / 256), (unsigned)nr);
        hipLaunchKernelGGL(k_unpack2b, grid, dim3(256), 0, (hipStream_t)stream, raw + r0 * stride, nr, cols, stride, shift, out + r0 * ld_out,
                           ld_out, bad_dev);
    }
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}


// ------------------------------------------------------------------------------------------------
// Small fp64 helpers of the device model algebra (eagle_linalg.cpp).  Matrices are n x n inside a buffer of
// leading dimension ld; "row-major" below is just the memory order (a symmetric matrix is layout-free).
// ------------------------------------------------------------------------------------------------
// mode 0: A[c][r] = A[r][c] for c < r (mirror the triangle potri wrote); mode 1: both = their mean
__global__ __launch_bounds__(256) void k_symmetrize(double* __restrict__ A, long n, long ld, int mode) {
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    if (bk > bj) return;  // tiles with row block >= column block
    __shared__ double t1[32][33], t2[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long i = bj + r, j = bk + tx;
        t1[r][tx] = (i < n && j < n) ? A[i * ld + j] : 0.0;          // lower-side tile (row block bj, column block bk)
        const long i2 = bk + r, j2 = bj + tx;
        t2[r][tx] = (i2 < n && j2 < n) ? A[i2 * ld + j2] : 0.0;      // its mirror tile
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long i = bj + r, j = bk + tx;                            // element (i, j) of the lower-side tile, mirror (j, i)
        if (i >= n || j >= n || j >= i) continue;
        const double lo = t1[r][tx], up = t2[tx][r];
        const double v = mode ? 0.5 * (lo + up) : lo;
        A[i * ld + j] = v;
        A[j * ld + i] = v;
    }
}
__global__ __launch_bounds__(256) void k_scale_rows_pow(double* __restrict__ R, long n, long ld, const double* __restrict__ w, double p) {
    const long i = blockIdx.y, j = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || j >= n) return;
    R[i * ld + j] *= pow(w[i], p);
}
__global__ __launch_bounds__(256) void k_transpose_f64(const double* __restrict__ in, double* __restrict__ out, long N) {
    __shared__ double t[32][33];
    const long bi = (long)blockIdx.y * 32, bj = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) t[r][tx] = in[(bi + r) * N + bj + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) out[(bj + r) * N + bi + tx] = t[tx][r];
}
// part[b] = sum over the rows b, b + gridDim.x, ... of sum_j A[i][j] B[i][j]; k_dot_final adds the parts in order
__global__ __launch_bounds__(256) void k_dot_rows(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, long n,
                                                  double* __restrict__ part) {
    double s = 0.0;
    for (long i = blockIdx.x; i < n; i += gridDim.x)
        for (long j = threadIdx.x; j < n; j += 256) s += A[i * lda + j] * B[i * ldb + j];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_dot_final(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    double s = 0.0;
    for (int b = 0; b < nparts; b++) s += part[b];
    *out = s;
}
extern "C" int eagle_dev_symmetrize(eagle_ctx* ctx, double* A, long n, long ld, void* stream) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(k_symmetrize, grid, dim3(256), 0, (hipStream_t)stream, A, n, ld, 0);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_symmetrize_mean(eagle_ctx* ctx, double* A, long n, long ld, void* stream) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(k_symmetrize, grid, dim3(256), 0, (hipStream_t)stream, A, n, ld, 1);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_scale_rows_pow(eagle_ctx* ctx, double* R, long n, long ld, const double* w, double p, void* stream) {
    if (n > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "scale_rows: n too large");
    hipLaunchKernelGGL(k_scale_rows_pow, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, R, n, ld, w, p);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_transpose_f64(eagle_ctx* ctx, const double* in, double* out, long N, void* stream) {
    if (N % 32 || N / 32 > 65535) return eagle_fail(ctx, EAGLE_ERR_ARG, "transpose_f64: bad size");
    hipLaunchKernelGGL(k_transpose_f64, dim3((unsigned)(N / 32), (unsigned)(N / 32)), dim3(256), 0, (hipStream_t)stream, in, out, N);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
extern "C" int eagle_dev_dot_matrices(eagle_ctx* ctx, const double* A, long lda, const double* B, long ldb, long n, double* out, void* stream) {
    double* part = (double*)((char*)eagle_ctx_scratch(ctx) + EAGLE_SCR_DOT_PARTIALS);  // 256 partial sums, their own region of the ctx scratch
    hipLaunchKernelGGL(k_dot_rows, dim3(256), dim3(256), 0, (hipStream_t)stream, A, lda, B, ldb, n, part);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(1), 0, (hipStream_t)stream, part, 256, out);
    LAUNCH_CHECK(ctx);
    return EAGLE_OK;
}
