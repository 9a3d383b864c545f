// eagle_w8.hip -- W = S (V S) of the marker scan on the int8 MFMA tile engine       (E/src/calculate_a_and_vara_rcpp.cpp:97-98, :197-198)
//
// The digit-slice scan keeps 24 bits of W's off-diagonal and certifies every vara_i a posteriori -- and until round 3 it formed that W
// on the 78.6 TFLOP/s fp64 pipe (30 % of the step at 10,000 individuals, 74 % at 50,000) next to a 5 POP/s int8 engine.  Here the two
// n^3 products run on the int8 engine from exact base-256 digit slices (an Ozaki-type splitting), with a RIGOROUS Frobenius-norm
// bound of the error of the W they deliver, which the scan's per-marker certificate carries next to its truncation term.
//
// Splitting.  S = D + F, V = Dv + Fv (diagonal + off-diagonal part of the row-major images Sa, Va).  MMt^-1/2 and a variance
// matrix are diagonally dominant: the diagonal parts are handled exactly, element by element, in fp64 --
//     X[j][k] = (S V^T)[j][k] = D_j Dv_j [j = k]  +  D_j Fv[k][j]  +  F[j][k] Dv_k  +  G1[j][k],    G1 = F Fv^T   (NT product of the images),
//     W[i][j] = (S X^T)[i][j] = D_i Dx_i [i = j]  +  D_i Fx[j][i]  +  F[i][j] Dx_j  +  G2[i][j],    G2 = F Fx^T,  X = Dx + Fx,
// and only G1, G2 are matrix products.  Their operands are flat (no dominant entry), so a per-ROW power-of-two scale wastes no bits:
//     F[i][l] = 2^(e_i+2) ( sum_{p=1..6} 256^-p a_p[i][l]  +  r ),   a_p in [-128, 127] balanced digits,  |r| <= 256^-6 / 2,
//     G[i][j] = 2^(e_i+f_j+4) sum_{p,q} 256^-(p+q) (a_p b_q^T)[i][j],   every a_p b_q^T an EXACT int32 product on v_mfma_i32_32x32x32_i8.
// Products of equal weight p + q = t share an accumulator (level t); a configuration (k, T) computes the pairs p, q <= k, p + q <= T.
// Level sums are exact integers, so the result does not depend on tiling, panel cuts or which device formed which rows: a row-sharded
// multi-GPU evaluation returns the single-device bits.  The levels are combined in fp64 in a fixed order.
//
// Error bound (everything in the Frobenius norm, which survives taking the upper triangle: the scan works on the folded image).
// For a product G = A B^T with A = sum_p 2^(e+2) 256^-p A_p + dA,  B likewise:
//     || G - G_computed ||_F  <=  sum_{(p,q) not computed} 256^-(p+q) Phi_p(A) Phi_q(B)  +  T_A (||B||_F + T_B)  +  ||A||_F T_B,
//     Phi_p(A)^2 = sum_i 4^(e_i+2) sum_l a_p[i][l]^2   (exact integer row sums, scaled),     T_A = sqrt(n_pad) 2^(-8*6-1) sqrt(sum_i 4^(e_i+2)),
// then  ||X_c - X||_F <= eta_1 + rounding,  ||W_c - W||_F <= (max|D| + ||F||_F) ||X_c - X||_F + eta_2 + rounding + asymmetry terms
// (the images are used where their transposes are meant; the measured max |M - M^T| pays for it), and the folded image built from the
// upper triangle of W_c is within sqrt(2) of that.  For a re-centred marker row m':  | m'^T (W_c - W) m' | <= eta_W sum_j m'_j^2.
// The host picks, for each product, the cheapest configuration that keeps eta_W below W8_TARGET x budget x mean|W_kk|; if none of the
// configurations does (wild scaling, cancelling V, non-finite or visibly asymmetric operands) the call DECLINES and the caller runs the
// fp64 GEMM as before.  Terms that must not carry eta_W at all are taken from elsewhere: the correction vector of the re-centred
// markers from r = S (V (S 1)) (three fp64 matrix-vector products), and the markers the certificate re-evaluates from
// m^T (S (V (S m))) in fp64 (k_w8_mgemv below) -- so the selected marker is still decided on fp64 values.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <algorithm>
#include <functional>
#include <string.h>

#include "../../include/eagle_hip.h"
#include "eagle_ctx.h"
#include "eagle_internal.h"
#include "eagle_t8.h"
#include "eagle_w8.h"

// ------------------------------------------------------------------------------------------------
// Row statistics of an n_pad x n_pad fp64 image: diagonal, largest off-diagonal magnitude, off-diagonal sum of squares.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w8_rowstats(const double* __restrict__ M, long np, double* __restrict__ d, double* __restrict__ mx,
                                                     double* __restrict__ ssq, int* __restrict__ bad, long row0) {
    const long i = row0 + blockIdx.x;
    const double* row = M + i * np;
    double m = 0.0, s = 0.0;
    int nf = 0;
    for (long l = threadIdx.x; l < np; l += 256) {
        const double v = row[l];
        if (!isfinite(v)) nf = 1;
        if (l != i) {
            const double a = fabs(v);
            m = a > m ? a : m;
            s += v * v;
        }
    }
    __shared__ double rm[256], rs[256];
    __shared__ int rb[256];
    rm[threadIdx.x] = m; rs[threadIdx.x] = s; rb[threadIdx.x] = nf;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            rm[threadIdx.x] = rm[threadIdx.x + o] > rm[threadIdx.x] ? rm[threadIdx.x + o] : rm[threadIdx.x];
            rs[threadIdx.x] += rs[threadIdx.x + o];
            rb[threadIdx.x] |= rb[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d[i] = row[i];
        mx[i] = rm[0];
        ssq[i] = rs[0];
        if (rb[0]) *bad = 1;
    }
}

// part[block] = this block's share of || M - M^T ||_F^2 (one block per pair of mirrored 32 x 32 tiles, every block writes: a fixed-order
// sum of `part` in k_w8_reduce is deterministic).  The products below use the images where their transposes are meant; what that
// changes in the symmetric part of S V S is bounded with these two numbers (eagle_dev_scan_operands_w8).
__global__ __launch_bounds__(256) void k_w8_asymsq(const double* __restrict__ M, long np, double* __restrict__ part) {
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    const long pid = (long)blockIdx.y * gridDim.x + blockIdx.x;
    if (bk < bj) { if (threadIdx.x == 0) part[pid] = 0.0; return; }
    __shared__ double t2[32][33];
    __shared__ double red[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) t2[r][tx] = M[(bk + r) * np + bj + tx];
    __syncthreads();
    double s = 0.0;
    for (int r = ty; r < 32; r += 8) {
        const double df = M[(bj + r) * np + bk + tx] - t2[tx][r];
        s += df * df;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[pid] = bk == bj ? red[0] : 2.0 * red[0];
}

// Digit slices of the off-diagonal part of one row: D[p][i][l] = digit p (0 = most significant) of round(M[i][l] 2^(8*6 - e_i - 2)),
// balanced, peeled least significant first with a carry (as k_slice_w); dssq[p][i] = sum_l D[p][i][l]^2 (exact).
__global__ __launch_bounds__(256) void k_w8_slice(const double* __restrict__ M, long np, const double* __restrict__ mx, int8_t* __restrict__ D,
                                                  long sstride, unsigned long long* __restrict__ dssq, int* __restrict__ eout, long row0) {
    const long i = row0 + blockIdx.x;
    const double* row = M + i * np;
    const double mxi = mx[i];
    const int e = w_scale_exp(mxi);
    const int sh = 8 * W8_KMAX - e - 2;
    unsigned long long sq[W8_KMAX];
#pragma unroll
    for (int p = 0; p < W8_KMAX; p++) sq[p] = 0;
    for (long l0 = (long)threadIdx.x * 16; l0 < np; l0 += 256 * 16) {
        union { i32x4 v; int8_t b[16]; } dg[W8_KMAX];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const long l = l0 + u;
            long long Q = 0;
            if (l != i && mxi > 0.0) Q = llrint(ldexp(row[l], sh));
#pragma unroll
            for (int p = W8_KMAX - 1; p >= 0; p--) {
                const long long dd = ((Q + 128) & 255) - 128;
                Q = (Q - dd) >> 8;
                dg[p].b[u] = (int8_t)dd;
                sq[p] += (unsigned long long)(dd * dd);
            }
        }
#pragma unroll
        for (int p = 0; p < W8_KMAX; p++) *(i32x4*)(D + (long)p * sstride + i * np + l0) = dg[p].v;
    }
    __shared__ unsigned long long red[W8_KMAX][256];
#pragma unroll
    for (int p = 0; p < W8_KMAX; p++) red[p][threadIdx.x] = sq[p];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int p = 0; p < W8_KMAX; p++) red[p][threadIdx.x] += red[p][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < W8_KMAX) dssq[(long)threadIdx.x * np + i] = red[threadIdx.x][0];
    if (threadIdx.x == 0) eout[i] = e;
}

// One block: the numbers the host's configuration choice needs (fixed summation order).  d2: optional second diagonal for the
// estimate sum_i d2_i^2 |d_i| of sum_k |W_kk|.
__global__ __launch_bounds__(1024) void k_w8_reduce(long np, const double* __restrict__ d, const double* __restrict__ mx, const double* __restrict__ ssq,
                                                    const int* __restrict__ e, const unsigned long long* __restrict__ dssq, const double* __restrict__ d2,
                                                    const int* __restrict__ bad, const double* __restrict__ asympart, long nasym, W8Stats* __restrict__ out) {
    double maxd = 0.0, fro2 = 0.0, es2 = 0.0, wd = 0.0, phi2[W8_KMAX];
#pragma unroll
    for (int p = 0; p < W8_KMAX; p++) phi2[p] = 0.0;
    for (long i = threadIdx.x; i < np; i += 1024) {
        const double a = fabs(d[i]);
        maxd = a > maxd ? a : maxd;
        fro2 += ssq[i];
        if (mx[i] > 0.0) {
            const double sc = ldexp(1.0, 2 * (e[i] + 2));
            es2 += sc;
#pragma unroll
            for (int p = 0; p < W8_KMAX; p++) phi2[p] += sc * (double)dssq[(long)p * np + i];
        }
        if (d2) wd += d2[i] * d2[i] * a;
    }
    __shared__ double red[1024];
    auto sum = [&](double x) -> double {
        __syncthreads();
        red[threadIdx.x] = x;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        return red[0];
    };
    __syncthreads();
    red[threadIdx.x] = maxd;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x + o] > red[threadIdx.x] ? red[threadIdx.x + o] : red[threadIdx.x];
        __syncthreads();
    }
    const double gmaxd = red[0];
    const double gfro2 = sum(fro2), ges2 = sum(es2), gwd = sum(wd);
    double as = 0.0;
    if (asympart)
        for (long i = threadIdx.x; i < nasym; i += 1024) as += asympart[i];
    const double gas = sum(as);
    double gphi2[W8_KMAX];
#pragma unroll
    for (int p = 0; p < W8_KMAX; p++) gphi2[p] = sum(phi2[p]);
    if (threadIdx.x == 0) {
        out->maxd = gmaxd; out->fro2 = gfro2; out->es2 = ges2; out->wdsum = gwd;
#pragma unroll
        for (int p = 0; p < W8_KMAX; p++) out->phi2[p] = gphi2[p];
        out->bad = bad ? *bad : 0;
        out->asym = asympart ? sqrt(gas) : 0.0;   // || M - M^T ||_F
    }
}

// ------------------------------------------------------------------------------------------------
// The products.  One workgroup = one 256 x 256 output tile of ONE accumulation group: the (at most 8) digit pairs of one level whose
// worst-case sum fits an int32 (n_pad x pairs x 2^14 < 2^31), streamed through the double-buffered LDS-DMA pipeline of the tile
// engine as one long K loop.  Work list per XCD (workgroup b runs on XCD b % 8, positions in order): units of up to 4 x 8 tiles of one
// group, dealt to the XCDs by cost, so that the ~32 workgroups an XCD runs at a time stream the same few digit panels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_w8_gemm(const int8_t* __restrict__ As, const int8_t* __restrict__ Bs, long sstride, long ld,
                                                    const unsigned* __restrict__ work, int maxlen, const W8Group* __restrict__ groups,
                                                    int32_t* __restrict__ L, long img_elems, long ldc, int row_tile0, int nstages, long col0) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    if (pos >= maxlen) return;
    const unsigned wk = work[(long)xcd * maxlen + pos];
    if (wk == 0xFFFFFFFFu) return;
    const int ti = (int)(wk >> 20), tj = (int)((wk >> 8) & 0xfffu), g = (int)(wk & 0xffu);
    const W8Group* gp = groups + g;
    const int gnp = gp->npairs;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld;
    const T8Lane ln = t8_lane(lane, ldi);
    const int8_t* Arow = As + (long)ti * T8 * ld;
    const int8_t* Brow = Bs + (long)tj * T8 * ld;
    __amdgpu_buffer_rsrc_t rsA = t8_rsrc(Arow + (long)gp->p[0] * sstride, ldi);
    __amdgpu_buffer_rsrc_t rsB = t8_rsrc(Brow + (long)gp->q[0] * sstride, ldi);
    i32x16 acc[4][2];
    t8_zero(acc);
    t8_stage(rsA, ln, ldi, 0, lds[0][0], w);
    t8_stage(rsB, ln, ldi, 0, lds[0][1], w);
    __syncthreads();
    int cur = 0;
    const T8Read rd = t8_read_init(wr, wc, lane);
    const int total = gnp * nstages;
    int s = 0, pr = 0;
    for (int it = 0; it < total; it++) {
        int s1 = s + 1, pr1 = pr;
        if (s1 == nstages) { s1 = 0; pr1 = pr + 1; }
        const bool more = it + 1 < total;
        if (more && pr1 != pr) {
            rsA = t8_rsrc(Arow + (long)gp->p[pr1] * sstride, ldi);
            rsB = t8_rsrc(Brow + (long)gp->q[pr1] * sstride, ldi);
        }
        t8_stage_compute<0>(acc, lds[cur][0], lds[cur][1], rd, more, rsA, ln, ldi, s1 * BK8, lds[cur ^ 1][0], rsB, ln, ldi, s1 * BK8,
                            lds[cur ^ 1][1], w);
        __syncthreads();
        cur ^= 1;
        s = s1; pr = pr1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    int32_t* Lg = L + (long)g * img_elems + ((long)(ti - row_tile0) * T8 + wr * 128 + 4 * (lane >> 5)) * ldc + ((long)tj * T8 - col0) + wc * 64 + (lane & 31);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) Lg[(long)(m * 32 + (q & 3) + 8 * (q >> 2)) * ldc + n * 32] = acc[m][n][q];
}

// The same product on the 384 x 256 tile with the asm-pipelined k-step of k_vara_i8p (eagle_t8.h): LDS reads, LDS-DMA and MFMAs of one wave
// overlap.  Engine operand "a" (384-row tiles; the LANES of the transposed 32 x 32 result tiles) = the operand whose rows are the
// output COLUMNS j (digit slices of V / X), operand "b" (256-row tiles; the registers) = the one whose rows are the output ROWS i
// (slices of S): a result register is 32 consecutive j of one row i, stored as one 128-byte run.  Work entry: ti = 256-row tile of S,
// tj = 384-row tile of the column operand (the last one may be short: rows beyond n_pad read as zero and are not stored).
__global__ __launch_bounds__(512, 2) void k_w8_gemm_p(const int8_t* __restrict__ Ss, const int8_t* __restrict__ Cs, long sstride, long ld,
                                                      const unsigned* __restrict__ work, int maxlen, const W8Group* __restrict__ groups,
                                                      int32_t* __restrict__ L, long img_elems, long ldc, int row_tile0, int nstages, long np, long col0) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];  // [2][A 48 KiB | B 32 KiB]
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    if (pos >= maxlen) return;
    const unsigned wk = __builtin_amdgcn_readfirstlane(work[(long)xcd * maxlen + pos]);
    if (wk == 0xFFFFFFFFu) return;
    const int ti = (int)(wk >> 20), tj = (int)((wk >> 8) & 0xfffu), g = (int)(wk & 0xffu);
    const W8Group* gp = groups + g;
    const int gnp = __builtin_amdgcn_readfirstlane(gp->npairs);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;   // 4 x 2 waves, wave tile 96 (a) x 128 (b)
    const int ldi = (int)ld;
    const T8Lane ln = t8_lane(lane, ldi);
    const int8_t* Acol = Cs + (long)tj * TW_M * ld;
    const int8_t* Brow = Ss + (long)ti * T8 * ld;
    const long rows_left = np - (long)tj * TW_M;
    const int rows_here = __builtin_amdgcn_readfirstlane((int)(rows_left < TW_M ? rows_left : TW_M));
    const int8_t* baseA = Acol + (long)gp->q[0] * sstride;
    const int8_t* baseB = Brow + (long)gp->p[0] * sstride;
    tw_stage<6>(__builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, rows_here * ldi, 0x00020000), ln, ldi, 0, ldsv, w);
    tw_stage<4>(t8_rsrc(baseB, ldi), ln, ldi, 0, ldsv + TW_ABYTES, w);
    __syncthreads();
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    constexpr int STG = TW_ABYTES + TILE_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) int8_t*)ldsv;
    const unsigned offA = lds0 + wr * (96 * BK8) + r * BK8, offB = lds0 + TW_ABYTES + wc * (128 * BK8) + r * BK8;
    unsigned ch[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) ch[ks] = ((2 * ks + h) ^ swz) << 4;
    XDma dA, dB;
    dA.st = dB.st = __builtin_amdgcn_readfirstlane(8 * ldi);
    dA.vE = dB.vE = ln.voffE; dA.vO = dB.vO = ln.voffO;
    const int total = gnp * nstages;   // the flattened (pair, K stage) sequence of this accumulation group
    int snext = 1, nkt = 1, npr = 0;   // the next stage to fetch: global index, K stage, pair   (nstages >= 2)
    unsigned nrecA = 0, nrecB = 0;
    // (the descriptors are rebuilt from scalars at each use: a loop-carried SGPR vector ends up in VGPRs)
    auto rs_fresh = [&] { dA.rs = x_rsrc(baseA, nrecA); dB.rs = x_rsrc(baseB, nrecB); };
    auto dma_arm = [&](int into) {
        const unsigned base = lds0 + into * STG;
        if (nkt == nstages) { nkt = 0; npr++; }
        const int left = __builtin_amdgcn_readfirstlane(total - snext);
        const unsigned on = (unsigned)max(min(left, 1), 0);
        if (on && nkt == 0) {   // the first stage of the next digit pair: other slices of the same row ranges
            baseA = Acol + (long)gp->q[npr] * sstride;
            baseB = Brow + (long)gp->p[npr] * sstride;
        }
        dA.m0 = __builtin_amdgcn_readfirstlane(base + (w * 6) * 1024 - 1024);
        dB.m0 = __builtin_amdgcn_readfirstlane(base + TW_ABYTES + (w * 4) * 1024 - 1024);
        dA.so = __builtin_amdgcn_readfirstlane((unsigned)((w * 6) * 8 * ldi + nkt * BK8 - 8 * ldi));
        dB.so = __builtin_amdgcn_readfirstlane((unsigned)((w * 4) * 8 * ldi + nkt * BK8 - 8 * ldi));
        // num_records = 0 when nothing is left to fetch: the loads then write zeros nobody reads
        nrecA = __builtin_amdgcn_readfirstlane((unsigned)(rows_here * ldi) * on);
        nrecB = __builtin_amdgcn_readfirstlane((unsigned)(T8 * ldi) * on);
        snext++; nkt++;
    };
    dma_arm(1);
    rs_fresh();
    tx_dma3(dA);
    sx_uniform(dA);
    int buf = 0;
    i32x4 fa[2][3], fb[4];
    TxAcc c;
    tx_prologue(fa[0], fb, offA + ch[0], offB + ch[0]);
    auto stage_rest = [&] {
        const unsigned sa = offA + buf * STG, sb = offB + buf * STG;
        rs_fresh();
        tx_kstep<false, 2>(c, fa[1], fa[0], fb, sa + ch[2], sb + ch[2], dA, dB);
        sx_uniform(dB);
        tx_kstep<false, 0>(c, fa[0], fa[1], fb, sa + ch[3], sb + ch[3], dA, dB);
        dma_arm(buf);
        rs_fresh();
        buf ^= 1;
        tx_klast(c, fa[1], fa[0], fb, offA + buf * STG + ch[0], offB + buf * STG + ch[0], dA);
        sx_uniform(dA);
    };
    rs_fresh();
    tx_kstep<true, 1>(c, fa[0], fa[1], fb, offA + ch[1], offB + ch[1], dA, dB);
    sx_uniform(dA); sx_uniform(dB);
    stage_rest();
    for (int s = 1; s < total; s++) {
        rs_fresh();
        tx_kstep<false, 1>(c, fa[0], fa[1], fb, offA + buf * STG + ch[1], offB + buf * STG + ch[1], dA, dB);
        sx_uniform(dA); sx_uniform(dB);
        stage_rest();
    }
    // the MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire; the stale re-loads have to land too
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // c[m][n][x]: lane r = row wr*96 + m*32 + r of the a tile (output column j), register = row wc*128 + n*32 + (x&3) + 8(x>>2) + 4h of the b tile (row i)
    int32_t* Lg = L + (long)g * img_elems + ((long)(ti - row_tile0) * T8 + wc * 128 + 4 * h) * ldc;
    const long j0 = (long)tj * TW_M + wr * 96 + r;
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const long j = j0 + m * 32;
        if (j < np) {
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int x = 0; x < 16; x++) Lg[(long)(n * 32 + (x & 3) + 8 * (x >> 2)) * ldc + (j - col0)] = c[m][n][x];
        }
    }
}

// sum_g 256^-level_g L_g[.] in fp64: groups are listed by ascending level; the smallest terms are added first (fixed order)
__device__ __forceinline__ double w8_levels(const int32_t* __restrict__ L, long img_elems, long off, const W8Group* __restrict__ groups, int ngroups) {
    double s = 0.0;
    for (int g = ngroups - 1; g >= 0; g--) s += ldexp((double)L[(long)g * img_elems + off], -8 * groups[g].level);
    return s;
}

// X[j][k] = (Sa Va^T)[j][k] for one 32 x 32 tile: rows j of the panel [row0, ...), columns k of the window [col0, col0 + ncols).  The
// element-wise term D_j Va[k][j] reads V's image TRANSPOSED (through LDS): it is row k of V that the product's digit slices come from,
// so X is exactly Sa Va^T -- and a column window needs only the rows of V that have landed (eagle_w8_vrows).
__global__ __launch_bounds__(256) void k_w8_combine1(const double* __restrict__ Sa, const double* __restrict__ Va, long np, const double* __restrict__ dS,
                                                     const double* __restrict__ dV, const int* __restrict__ eS, const int* __restrict__ eV,
                                                     const int32_t* __restrict__ L, long img_elems, const W8Group* __restrict__ groups, int ngroups,
                                                     long row0, double* __restrict__ X, long col0, long ncols) {
    const long bj = row0 + (long)blockIdx.y * 32, bk = col0 + (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    __shared__ double vt[32][33];   // vt[a][b] = Va[bk + a][bj + b]
    for (int r = ty; r < 32; r += 8) vt[r][tx] = Va[(bk + r) * np + bj + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long j = bj + r, k = bk + tx;
        const double g = ldexp(w8_levels(L, img_elems, (j - row0) * ncols + (k - col0), groups, ngroups), eS[j] + eV[k] + 4);
        double x;
        if (j == k) x = dS[j] * dV[j] + g;
        else x = (dS[j] * vt[tx][r] + Sa[j * np + k] * dV[k]) + g;
        X[j * np + k] = x;
    }
}

// The folded image of W from its upper triangle: Wu[i][j] = 2 W[i][j] (i < j), W[i][i], 0 below; one block per 32 x 32 tile.
// (Sa and Wu may be the same buffer: an element above the diagonal is read, then written, by one thread; nothing below it is read)
__global__ __launch_bounds__(256) void k_w8_combine2(const double* Sa, const double* __restrict__ X, long np, const double* __restrict__ dS,
                                                     const double* __restrict__ dX, const int* __restrict__ eS, const int* __restrict__ eX,
                                                     const int32_t* __restrict__ L, long img_elems, const W8Group* __restrict__ groups, int ngroups,
                                                     long row0, double* Wu) {
    const long bi = (long)blockIdx.y + row0 / 32, bj = blockIdx.x;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (bj < bi) {
        for (int r = ty; r < 32; r += 8) Wu[(bi * 32 + r) * np + bj * 32 + tx] = 0.0;
        return;
    }
    __shared__ double xt[32][33];   // xt[a][b] = X[32 bj + a][32 bi + b]
    for (int r = ty; r < 32; r += 8) xt[r][tx] = X[(bj * 32 + r) * np + bi * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long i = bi * 32 + r, j = bj * 32 + tx;
        double v = 0.0;
        if (i <= j) {
            const double g = ldexp(w8_levels(L, img_elems, (i - row0) * np + j, groups, ngroups), eS[i] + eX[j] + 4);
            if (i == j) v = dS[i] * dX[i] + g;
            else v = 2.0 * ((dS[i] * xt[tx][r] + Sa[i * np + j] * dX[j]) + g);
        }
        Wu[i * np + j] = v;
    }
}

// out_i = sum_j At[i][j] x_j (the transposed product of k_colgemv: one wave per row, fixed order)
__global__ __launch_bounds__(256) void k_w8_rowgemv(const double* __restrict__ At, long n, long np, const double* __restrict__ x, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double s0 = 0.0, s1 = 0.0;
    long j = lane;
    for (; j + 64 < n; j += 128) { s0 += At[i * np + j] * x[j]; s1 += At[i * np + j + 64] * x[j + 64]; }
    if (j < n) s0 += At[i * np + j] * x[j];
    double v = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) out[i] = v;
}
__global__ __launch_bounds__(256) void k_w8_mean2(const double* __restrict__ a, const double* __restrict__ b, long np, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < np) out[i] = 0.5 * (a[i] + b[i]);
}
__global__ __launch_bounds__(256) void k_w8_fill_ones(double* __restrict__ x, long n, long np) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < np) x[i] = i < n ? 1.0 : 0.0;
}
// sum_k |Wu[k][k]| in a fixed order (one block)
__global__ __launch_bounds__(1024) void k_w8_sumdiag(const double* __restrict__ Wu, long np, double* __restrict__ out) {
    double s = 0.0;
    for (long k = threadIdx.x; k < np; k += 1024) s += fabs(Wu[k * np + k]);
    __shared__ double red[1024];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}
// rho_j = 2 (r_j - W_jj): the off-diagonal row sums of the folded image from r = W 1 (see k_marker_shift in eagle_i8mfma.hip)
__global__ __launch_bounds__(256) void k_w8_rho_from_r(const double* __restrict__ r, const double* __restrict__ Wu, long np, double* __restrict__ rho) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j < np) rho[j] = 2.0 * (r[j] - Wu[j * np + j]);
}
extern "C" int eagle_w8_rho(eagle_ctx* ctx, const double* Wu, long n_pad, double* rho, void* stream) {
    hipLaunchKernelGGL(k_w8_rho_from_r, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const double*)ctx->w8_r, Wu, n_pad, rho);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_w8_rho_from_r");
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// Re-evaluation of single markers against the TRUE W: vara = m^T (S (V (S m))) in fp64, 16 or 64 markers per pass over a matrix.
//   out[c][i] = sum_j At[j][i] x[c][j]  (At = row-major image of A^T: out_c = A x_c) on the fp64 MFMA (v_mfma_f64_16x16x4_f64: 16 markers
//   x 16 columns x 4 j per instruction).  Every (c, i) is summed in a fixed order whatever else is in the batch and whichever row of a
//   tile the marker sits in: the j range is cut into W8_JS parts, inside a part wave w takes the groups of four j number w, w + 4, ...
//   in ascending order, then (w0 + w1) + (w2 + w3), then the parts in order (k_w8_mgemv_sum).  A structured panel sends hundreds of
//   markers here (tests/test_gpu_structure.py): 64 of them share one read of the matrix; a headline scan sends one or two.
// ------------------------------------------------------------------------------------------------
#define W8_MG 16    // markers per pass, a handful of candidates (one MFMA row tile)
#define W8_MGL 64   // ... many candidates (four row tiles)
#define W8_JS 16
#define W8_TRUE_CHUNK 256
typedef double w8_f64x4 __attribute__((ext_vector_type(4)));
// part[js][c][i] for the 64 columns i of blockIdx.z, the MT * 16 markers of blockIdx.x, the js-th part of j (blockIdx.y)
template <int MT>
__global__ __launch_bounds__(256) void k_w8_mgemv_part(const double* __restrict__ At, long n, long np, const double* __restrict__ X, double* __restrict__ part,
                                                       int capr) {
    const int c0 = blockIdx.x * (MT * 16), js = blockIdx.y;
    const long i0 = (long)blockIdx.z * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
    const long n4 = (n + 3) / 4 * 4;   // (rows n .. np of the images and columns n .. np of x are zero)
    const long per = ((n + W8_JS - 1) / W8_JS + 3) / 4 * 4;
    const long ja = (long)js * per, jb = ja + per < n4 ? ja + per : n4;
    w8_f64x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[m][q] = (w8_f64x4){0.0, 0.0, 0.0, 0.0};
    // operand maps of the instruction: A lane (i16, g) = x[marker i16][j + g], B lane (i16, g) = At[j + g][column i16]
    const double* xa = X + (long)(c0 + i16) * np + g;
    const double* bt = At + (long)g * np + i0 + i16;
    // (the operands of the next group are in flight while this one multiplies)
    double a[MT], b[4];
    long j = ja + 4 * w;
    if (j < jb) {
#pragma unroll
        for (int q = 0; q < 4; q++) b[q] = bt[j * np + 16 * q];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = xa[(long)m * 16 * np + j];
    }
    for (; j < jb; j += 16) {
        double an[MT], bn[4];
        const long jn = j + 16 < jb ? j + 16 : j;   // (the last group loads itself again: no branch around the loads)
#pragma unroll
        for (int q = 0; q < 4; q++) bn[q] = bt[jn * np + 16 * q];
#pragma unroll
        for (int m = 0; m < MT; m++) an[m] = xa[(long)m * 16 * np + jn];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[q], acc[m][q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) b[q] = bn[q];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = an[m];
    }
    // the four waves' tiles, one row tile at a time: C/D map row = g + 4 r, column = i16
    __shared__ double sm[4][4][256];   // [wave][column tile][lane * 4 + r]: 32 KiB
#pragma unroll
    for (int m = 0; m < MT; m++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) *(w8_f64x4*)&sm[w][q][lane * 4] = acc[m][q];
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 256) {   // e = row * 64 + column of the 16 x 64 block
            const int row = e >> 6, col = e & 63, q = col >> 4;
            const int k = (((row & 3) << 4) | (col & 15)) * 4 + (row >> 2);   // lane (g = row % 4, i16 = col % 16), register row / 4
            part[((long)js * capr + c0 + m * 16 + row) * np + i0 + col] = (sm[0][q][k] + sm[1][q][k]) + (sm[2][q][k] + sm[3][q][k]);
        }
    }
}
__global__ __launch_bounds__(256) void k_w8_mgemv_sum(const double* __restrict__ part, long np, int capr, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (i >= np) return;
    double s = 0.0;
    for (int js = 0; js < W8_JS; js++) s += part[((long)js * capr + c) * np + i];
    out[(long)c * np + i] = s;
}
// x[c][j] = (double) rows8[c][j] for c < count, zero rows up to the next multiple of 16
__global__ __launch_bounds__(256) void k_w8_rows_f64(const int8_t* __restrict__ rows8, long ld, long np, int count, double* __restrict__ X) {
    const int c = blockIdx.y;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j < np) X[(long)c * np + j] = c < count ? (double)rows8[(long)c * ld + j] : 0.0;
}
// out[dst[c]] = sum_i T[c][i] U[c][i] (one block per marker, fixed order)
__global__ __launch_bounds__(256) void k_w8_rowdot(const double* __restrict__ T, const double* __restrict__ U, long np, const long* __restrict__ dst,
                                                   double* __restrict__ out) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < np; i += 256) s += T[(long)c * np + i] * U[(long)c * np + i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[dst ? dst[c] : c] = red[0];
}

// out[dst[c]] = m_c^T S V S m_c = m_c^T (S (V (S m_c))) for the `count` rows of rows8 (host count), 256 rows at a time through a ctx-owned
// buffer: three products of the same kind (the image Sa is the row-major S^T: out_c = S x_c sums Sa[j][i] x[c][j] over j), then m . r.
// S, V: the operands of the last eagle_dev_scan_operands_w8 (still alive: it is the caller's scan).
extern "C" int eagle_w8_true_vara(eagle_ctx* ctx, const int8_t* rows8, long count, long n_pad, long ld, const long* dst_dev, double* out, void* stream) {
    if (!ctx->w8_Sa || !ctx->w8_Va || ctx->w8_n <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "w8_true_vara: no operands on record");
    if (count <= 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t need = (size_t)(3 + W8_JS) * W8_TRUE_CHUNK * n_pad * sizeof(double);
    if (need > ctx->w8_true_cap) {
        if (ctx->w8_true_ws) { (void)hipStreamSynchronize(s); (void)hipFree(ctx->w8_true_ws); ctx->w8_true_ws = nullptr; ctx->w8_true_cap = 0; }
        hipError_t e = hipMalloc(&ctx->w8_true_ws, need);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "w8 re-evaluation buffer");
        ctx->w8_true_cap = need;
    }
    double* X = (double*)ctx->w8_true_ws;
    double* T = X + (size_t)W8_TRUE_CHUNK * n_pad;
    double* U = T + (size_t)W8_TRUE_CHUNK * n_pad;
    double* part = U + (size_t)W8_TRUE_CHUNK * n_pad;
    const long n = ctx->w8_n;
    for (long c0 = 0; c0 < count; c0 += W8_TRUE_CHUNK) {
        const int cnt = (int)std::min<long>(W8_TRUE_CHUNK, count - c0);
        // a handful of candidates: one row tile of 16 per pass over the matrices; many (a structured panel flags hundreds): 64 per pass --
        // the same bits per marker
        const bool many = cnt > W8_MG;
        const int mg = many ? W8_MGL : W8_MG;
        const int capr = (cnt + mg - 1) / mg * mg;
        hipLaunchKernelGGL(k_w8_rows_f64, dim3((unsigned)((n_pad + 255) / 256), (unsigned)capr), dim3(256), 0, s, rows8 + c0 * ld, ld, n_pad, cnt, X);
        const dim3 gp((unsigned)(capr / mg), W8_JS, (unsigned)(n_pad / 64)), gs((unsigned)((n_pad + 255) / 256), (unsigned)capr);
        const double* mats[3] = {ctx->w8_Sa, ctx->w8_Va, ctx->w8_Sa};      // t = S m,  u = V t,  r = S u
        const double* src[3] = {X, T, U};
        double* dstv[3] = {T, U, T};
        for (int k = 0; k < 3; k++) {
            if (many) hipLaunchKernelGGL((k_w8_mgemv_part<W8_MGL / 16>), gp, dim3(256), 0, s, mats[k], n, n_pad, src[k], part, capr);
            else hipLaunchKernelGGL((k_w8_mgemv_part<W8_MG / 16>), gp, dim3(256), 0, s, mats[k], n, n_pad, src[k], part, capr);
            hipLaunchKernelGGL(k_w8_mgemv_sum, gs, dim3(256), 0, s, (const double*)part, n_pad, capr, dstv[k]);
        }
        hipLaunchKernelGGL(k_w8_rowdot, dim3((unsigned)cnt), dim3(256), 0, s, (const double*)X, (const double*)T, n_pad, dst_dev ? dst_dev + c0 : nullptr,
                           dst_dev ? out : out + c0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "w8_true_vara");
    return EAGLE_OK;
}
// The scan found its certificate overflowing on a W from the int8 engine (degenerate operands): the fp64 products replace the image
// in place, from the operands on record, and the context forgets the int8 result.
extern "C" int eagle_w8_redo_f64(eagle_ctx* ctx, long n_pad, void* stream) {
    if (!ctx->w8_active) return EAGLE_OK;
    ctx->w8_active = false;
    ctx->w8_info.declined = 8;
    return eagle_dev_scan_operands_w_f64(ctx, ctx->w8_Sa, ctx->w8_Va, n_pad, (double*)ctx->w8_Wu, ctx->w8_tmp, stream);
}

// ------------------------------------------------------------------------------------------------
// Host side: configurations, work lists, the bound, the pipeline.
// ------------------------------------------------------------------------------------------------
int w8_config_pairs(const W8Config& c) {
    int np = 0;
    for (int p = 1; p <= c.k; p++)
        for (int q = 1; q <= c.k; q++) np += (p + q <= c.T);
    return np;
}
// accumulation groups of a configuration: pairs of one level, at most maxp per group, levels ascending
std::vector<W8Group> w8_groups(const W8Config& c, int maxp) {
    std::vector<W8Group> gs;
    if (maxp > 8) maxp = 8;
    for (int t = 2; t <= c.T; t++) {
        W8Group g = {};
        g.level = t;
        for (int p = 1; p <= c.k; p++) {
            const int q = t - p;
            if (q < 1 || q > c.k) continue;
            if (g.npairs == maxp) { gs.push_back(g); g = {}; g.level = t; }
            g.p[g.npairs] = (unsigned char)(p - 1);
            g.q[g.npairs] = (unsigned char)(q - 1);
            g.npairs++;
        }
        if (g.npairs) gs.push_back(g);
    }
    return gs;
}
// The work list of one product over the row tiles [rt0, rt1): per XCD `maxlen` entries (ti << 20 | tj << 8 | group, 0xFFFFFFFF = none).
// Row tiles of ti_rows rows (i), column tiles of tj_rows rows (j), ntj of them; upper: only tiles that hold an element with j >= i.
// Units = (ui x uj super-tile, group) -- what an XCD's 32 workgroups run at a time -- dealt longest first to the XCD with the least
// work so far.
void w8_work_list(int rt0, int rt1, int ntj, int ti_rows, int tj_rows, int ui, int uj, bool upper, const std::vector<W8Group>& gs,
                  std::vector<unsigned>& out, int* maxlen_out, int tj0) {
    struct Unit { int cost; std::vector<unsigned> items; };
    std::vector<Unit> units;
    for (int si = rt0; si < rt1; si += ui)
        for (int sj = tj0; sj < ntj; sj += uj)
            for (size_t g = 0; g < gs.size(); g++) {
                Unit u;
                for (int i = si; i < si + ui && i < rt1; i++)
                    for (int j = sj; j < sj + uj && j < ntj; j++)
                        if (!upper || (long)j * tj_rows + tj_rows - 1 >= (long)i * ti_rows) u.items.push_back(((unsigned)i << 20) | ((unsigned)j << 8) | (unsigned)g);
                if (u.items.empty()) continue;
                u.cost = (int)u.items.size() * gs[g].npairs;
                units.push_back(std::move(u));
            }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.cost > b.cost; });
    std::vector<unsigned> lists[8];
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Unit& u : units) {
        int best = 0;
        for (int x = 1; x < 8; x++) if (load[x] < load[best]) best = x;
        load[best] += u.cost;
        lists[best].insert(lists[best].end(), u.items.begin(), u.items.end());
    }
    size_t maxlen = 0;
    for (int x = 0; x < 8; x++) maxlen = std::max(maxlen, lists[x].size());
    out.assign(8 * maxlen, 0xFFFFFFFFu);
    for (int x = 0; x < 8; x++) std::copy(lists[x].begin(), lists[x].end(), out.begin() + x * maxlen);
    *maxlen_out = (int)maxlen;
}

// || G - G_computed ||_F of a product A B^T under configuration c (see the head of the file); every step rounded up
double w8_product_bound(const W8Stats& A, const W8Stats& B, const W8Config& c, long np) {
    const double up = 1.0 + 1e-9;
    double phiA[W8_KMAX], phiB[W8_KMAX];
    for (int p = 0; p < W8_KMAX; p++) { phiA[p] = sqrt(A.phi2[p] * up) * up; phiB[p] = sqrt(B.phi2[p] * up) * up; }
    double drop = 0.0;
    for (int p = 1; p <= W8_KMAX; p++)
        for (int q = 1; q <= W8_KMAX; q++)
            if (!(p <= c.k && q <= c.k && p + q <= c.T)) drop += ldexp(phiA[p - 1] * phiB[q - 1], -8 * (p + q)) * up;
    const double TA = sqrt((double)np * A.es2 * up) * ldexp(1.0, -8 * W8_KMAX - 1) * up;
    const double TB = sqrt((double)np * B.es2 * up) * ldexp(1.0, -8 * W8_KMAX - 1) * up;
    const double fA = sqrt(A.fro2 * up) * up, fB = sqrt(B.fro2 * up) * up;
    return (drop + TA * (fB + TB) + fA * TB) * up;
}

// Host-only hooks for the CPU tests (tests/test_w8_host.py): the work list and the bound exactly as the pipeline uses them.
extern "C" int eagle_w8_host_work_list(int nt, int rt0, int rt1, int upper, int piped, int k, int T, int maxp, unsigned* out, long cap, int* maxlen,
                                       W8Group* groups_out, int* ngroups) {
    const W8Config c = {k, T};
    const std::vector<W8Group> gs = w8_groups(c, maxp);
    std::vector<unsigned> wl;
    int ml = 0;
    if (piped) w8_work_list(rt0, rt1, (int)(((long)nt * T8 + TW_M - 1) / TW_M), T8, TW_M, 8, 4, upper != 0, gs, wl, &ml, 0);
    else w8_work_list(rt0, rt1, nt, T8, T8, 4, 8, upper != 0, gs, wl, &ml, 0);
    if ((long)wl.size() > cap || gs.size() > 64) return -1;
    std::copy(wl.begin(), wl.end(), out);
    std::copy(gs.begin(), gs.end(), groups_out);
    *maxlen = ml;
    *ngroups = (int)gs.size();
    return 0;
}
extern "C" double eagle_w8_host_bound(const W8Stats* A, const W8Stats* B, int k, int T, long np) {
    const W8Config c = {k, T};
    return w8_product_bound(*A, *B, c, np);
}

static const W8Config W8_CONFIGS[] = {{3, 4}, {3, 5}, {4, 5}, {4, 6}, {5, 6}, {5, 7}, {6, 7}, {6, 8}, {6, 9}, {6, 12}};
static const int W8_NCONFIGS = (int)(sizeof(W8_CONFIGS) / sizeof(W8_CONFIGS[0]));
// cheapest configuration whose bound is <= limit (-1: none)
static int w8_choose(const W8Stats& A, const W8Stats& B, long np, double limit, double* bound_out) {
    int best = -1, best_pairs = 1 << 30;
    double best_b = 0.0;
    for (int c = 0; c < W8_NCONFIGS; c++) {
        const double b = w8_product_bound(A, B, W8_CONFIGS[c], np);
        const int pairs = w8_config_pairs(W8_CONFIGS[c]);
        if (getenv("EAGLE_HIP_W8_DEBUG")) fprintf(stderr, "[eaglehip w8] config (%d,%d) %d pairs: bound %.3e (limit %.3e)\n", W8_CONFIGS[c].k, W8_CONFIGS[c].T, pairs, b, limit);
        if (b <= limit && pairs < best_pairs) { best = c; best_pairs = pairs; best_b = b; }
    }
    if (bound_out) *bound_out = best_b;
    return best;
}

struct W8List { unsigned* work = nullptr; W8Group* groups = nullptr; int maxlen = 0, ngroups = 0; };
static std::map<std::tuple<int, int, int, int, int, int, int, int, int>, W8List> g_w8_lists;   // (device, nt, rt0, rt1, upper | engine, config, maxp, column window)
static std::mutex g_w8_mutex;
// c0, c1: the column window [c0, c1) of the product (multiples of 1536 = 4 x 384 = 6 x 256, or c1 = n_pad); -1: all columns
static int w8_get_list(eagle_ctx* ctx, int nt, int rt0, int rt1, bool upper, bool piped, int cfg, int maxp, W8List* out, long c0 = -1, long c1 = -1) {
    std::lock_guard<std::mutex> lock(g_w8_mutex);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const long np = (long)nt * T8;
    const int tjr = piped ? TW_M : T8;
    const int tj0 = c0 < 0 ? 0 : (int)(c0 / tjr), tj1 = c0 < 0 ? (int)((np + tjr - 1) / tjr) : (int)((c1 + tjr - 1) / tjr);
    auto key = std::make_tuple(dev, nt, rt0, rt1, (upper ? 1 : 0) | (piped ? 2 : 0), cfg, maxp, tj0, tj1);
    auto it = g_w8_lists.find(key);
    if (it != g_w8_lists.end()) { *out = it->second; return EAGLE_OK; }
    const std::vector<W8Group> gs = w8_groups(W8_CONFIGS[cfg], maxp);
    std::vector<unsigned> wl;
    W8List l;
    if (piped) w8_work_list(rt0, rt1, tj1, T8, TW_M, 8, 4, upper, gs, wl, &l.maxlen, tj0);
    else w8_work_list(rt0, rt1, tj1, T8, T8, 4, 8, upper, gs, wl, &l.maxlen, tj0);
    l.ngroups = (int)gs.size();
    hipError_t e = hipMalloc((void**)&l.work, wl.size() * sizeof(unsigned) + 16);
    if (e == hipSuccess) e = hipMalloc((void**)&l.groups, gs.size() * sizeof(W8Group));
    if (e == hipSuccess) e = hipMemcpy(l.work, wl.data(), wl.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(l.groups, gs.data(), gs.size() * sizeof(W8Group), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (l.work) (void)hipFree(l.work);
        if (l.groups) (void)hipFree(l.groups);
        return eagle_fail_hip(ctx, e, "w8 work list");
    }
    g_w8_lists[key] = l;
    *out = l;
    return EAGLE_OK;
}

// workspace: [ slices A: 6 n^2 | slices B: 6 n^2 | levels | per-matrix vectors ... ]
struct W8Ws {
    int8_t *sA, *sB;
    int32_t* levels; size_t level_bytes;
    double *dS, *mxS, *ssqS, *dV, *mxV, *ssqV, *dX, *mxX, *ssqX, *r, *r1, *r2, *sumdiag, *gvpart;
    int *eS, *eV, *eX, *bad;
    unsigned long long *dssqS, *dssqV, *dssqX;
    double* asympart;
    W8Stats* stats;  // [3] device
};
static size_t r256(size_t x) { return (x + 255) / 256 * 256; }
#define W8_LEVEL_CAP ((size_t)3500 << 20)
static int w8_workspace(eagle_ctx* ctx, long np, W8Ws* w) {
    const size_t nn = (size_t)np * np;
    const size_t full_levels = 12 * nn * sizeof(int32_t);   // the largest configuration: 11 levels (+ splits at very large n)
    size_t cap = np >= 32768 ? 4 * W8_LEVEL_CAP : W8_LEVEL_CAP;   // (14 GB at 50,000 individuals: panels of 24 row tiles instead of 4)
    if (const char* e = getenv("EAGLE_HIP_W8_LEVEL_MB")) {   // tests: small level images force the products through several row panels
        const long mb = atol(e);
        if (mb >= 1) cap = (size_t)mb << 20;
    }
    const size_t level_bytes = full_levels < cap ? full_levels : cap;
    const size_t vec = r256(sizeof(double) * (size_t)np), dss = r256(sizeof(unsigned long long) * W8_KMAX * (size_t)np);
    const size_t asymb = r256(sizeof(double) * (size_t)(np / 32) * (size_t)(np / 32));
    const size_t need = 2 * r256(W8_KMAX * nn) + r256(level_bytes) + 13 * vec + 8 * vec + 4 * vec + 3 * dss + asymb + 4096;
    if (need > ctx->w8_ws_cap || (getenv("EAGLE_HIP_W8_LEVEL_MB") && need != ctx->w8_ws_cap)) {
        if (ctx->w8_ws) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->w8_ws); ctx->w8_ws = nullptr; ctx->w8_ws_cap = 0; }
        if (hipMalloc(&ctx->w8_ws, need) != hipSuccess) {
            (void)hipGetLastError();
            ctx->w8_ws = nullptr;
            if (!(eagle_drop_f4_images(ctx) > 0 && hipMalloc(&ctx->w8_ws, need) == hipSuccess)) { (void)hipGetLastError(); ctx->w8_ws = nullptr; return 1; }   // no room: decline
        }
        ctx->w8_ws_cap = need;
    }
    char* p = (char*)ctx->w8_ws;
    auto take = [&](size_t b) { char* q = p; p += r256(b); return q; };
    w->sA = (int8_t*)take(W8_KMAX * nn);
    w->sB = (int8_t*)take(W8_KMAX * nn);
    w->levels = (int32_t*)take(level_bytes);
    w->level_bytes = level_bytes;
    double** dv[] = {&w->dS, &w->mxS, &w->ssqS, &w->dV, &w->mxV, &w->ssqV, &w->dX, &w->mxX, &w->ssqX, &w->r, &w->r1, &w->r2, &w->sumdiag};
    for (double** x : dv) *x = (double*)take(vec);
    w->gvpart = (double*)take(8 * vec);
    int** iv[] = {&w->eS, &w->eV, &w->eX, &w->bad};
    for (int** x : iv) *x = (int*)take(vec);
    unsigned long long** uv[] = {&w->dssqS, &w->dssqV, &w->dssqX};
    for (unsigned long long** x : uv) *x = (unsigned long long*)take(dss);
    w->asympart = (double*)take(asymb);
    w->stats = (W8Stats*)take(3 * sizeof(W8Stats));
    return 0;
}

// row statistics + digit slices of the rows [r0, r1) of one operand
static int w8_rows_of(eagle_ctx* ctx, const double* M, long np, long r0, long r1, double* d, double* mx, double* ssq, int* e, unsigned long long* dssq,
                      int8_t* slices, W8Ws& w, hipStream_t s) {
    hipLaunchKernelGGL(k_w8_rowstats, dim3((unsigned)(r1 - r0)), dim3(256), 0, s, M, np, d, mx, ssq, w.bad, r0);
    hipLaunchKernelGGL(k_w8_slice, dim3((unsigned)(r1 - r0)), dim3(256), 0, s, M, np, (const double*)mx, slices, np * np, dssq, e, r0);
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) return eagle_fail_hip(ctx, er, "w8 row statistics");
    return EAGLE_OK;
}
// the numbers the configuration choice needs, of an operand whose rows have all been through w8_rows_of; asym: also || M - M^T ||_F
static int w8_reduce_of(eagle_ctx* ctx, const double* M, long np, double* d, double* mx, double* ssq, int* e, unsigned long long* dssq, const double* d2,
                        bool asym, W8Ws& w, W8Stats* out_dev, hipStream_t s) {
    const long nb = np / 32;
    if (asym) hipLaunchKernelGGL(k_w8_asymsq, dim3((unsigned)nb, (unsigned)nb), dim3(256), 0, s, M, np, w.asympart);
    hipLaunchKernelGGL(k_w8_reduce, dim3(1), dim3(1024), 0, s, np, (const double*)d, (const double*)mx, (const double*)ssq, (const int*)e,
                       (const unsigned long long*)dssq, d2, (const int*)w.bad, asym ? (const double*)w.asympart : nullptr, nb * nb, out_dev);
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) return eagle_fail_hip(ctx, er, "w8 statistics");
    return EAGLE_OK;
}
// statistics + digit slices of one operand
static int w8_stats_of(eagle_ctx* ctx, const double* M, long np, double* d, double* mx, double* ssq, int* e, unsigned long long* dssq, int8_t* slices,
                       const double* d2, bool asym, W8Ws& w, W8Stats* out_dev, hipStream_t s) {
    hipError_t er = hipMemsetAsync(w.bad, 0, sizeof(int), s);
    if (er != hipSuccess) return eagle_fail_hip(ctx, er, "w8 memset");
    int rc = w8_rows_of(ctx, M, np, 0, np, d, mx, ssq, e, dssq, slices, w, s);
    if (rc) return rc;
    return w8_reduce_of(ctx, M, np, d, mx, ssq, e, dssq, d2, asym, w, out_dev, s);
}

// c0, c1: the column window of the product (-1: all columns); the level images then hold that window only (row pitch c1 - c0)
static int w8_product(eagle_ctx* ctx, int cfg, bool upper, long np, const int8_t* sA, const int8_t* sB, W8Ws& w, hipStream_t s,
                      const std::function<void(const W8List&, long, long, long)>& combine, long c0 = -1, long c1 = -1) {
    const long col0 = c0 < 0 ? 0 : c0, ncols = c0 < 0 ? np : c1 - c0;
    const int nt = (int)(np / T8);
    int maxp = (int)(131071 / np);
    if (maxp < 1) return 1;
    if (maxp > 8) maxp = 8;
    const int ngroups = (int)w8_groups(W8_CONFIGS[cfg], maxp).size();
    // rows per panel so that the level images of all groups fit
    long rt_per_panel = (long)(w.level_bytes / ((size_t)ngroups * T8 * ncols * sizeof(int32_t)));
    if (rt_per_panel < 1) return 1;
    if (rt_per_panel >= nt) rt_per_panel = nt;
    else if (rt_per_panel > 8) rt_per_panel = rt_per_panel / 8 * 8;   // whole super-tile rows (8 row tiles in the 384 x 256 tiling, 4 in the other)
    else if (rt_per_panel > 4) rt_per_panel = 4;
    for (int rt0 = 0; rt0 < nt; rt0 += (int)rt_per_panel) {
        const int rt1 = (int)std::min<long>(nt, rt0 + rt_per_panel);
        W8List l;
        const bool piped = ctx->tune != 31 && (double)np * TW_M < 2147483648.0;   // tune 31: the compiler-scheduled 256 x 256 form (A/B runs)
        int rc = w8_get_list(ctx, nt, rt0, rt1, upper, piped, cfg, maxp, &l, c0, c1);
        if (rc) return rc;
        const long img_elems = (long)(rt1 - rt0) * T8 * ncols;
        if (piped) {
            if (!ctx->attr_w8_gemm) {
                hipError_t ea = hipFuncSetAttribute((const void*)k_w8_gemm_p, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TW_ABYTES + TILE_BYTES));
                if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_w8_gemm_p)");
                ctx->attr_w8_gemm = true;
            }
            hipLaunchKernelGGL(k_w8_gemm_p, dim3((unsigned)(8 * l.maxlen)), dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, sA, sB, np * np, np, (const unsigned*)l.work,
                               l.maxlen, (const W8Group*)l.groups, w.levels, img_elems, ncols, rt0, (int)(np / BK8), np, col0);
        } else
        hipLaunchKernelGGL(k_w8_gemm, dim3((unsigned)(8 * l.maxlen)), dim3(512), 0, s, sA, sB, np * np, np, (const unsigned*)l.work, l.maxlen,
                           (const W8Group*)l.groups, w.levels, img_elems, ncols, rt0, (int)(np / BK8), col0);
        combine(l, img_elems, (long)rt0 * T8, (long)(rt1 - rt0) * T8);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "w8 product");
    return EAGLE_OK;
}

static int w8_fetch(eagle_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!ctx->w8_host) {
        hipError_t e = hipHostMalloc((void**)&ctx->w8_host, 4096, hipHostMallocDefault);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "w8 pinned block");
    }
    hipError_t e = hipMemcpyAsync(ctx->w8_host, src, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "w8 statistics to the host");
    memcpy(dst, ctx->w8_host, bytes);
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// The pipeline in three steps, so that the reference-shaped call can work on V while it arrives over PCIe:
//   eagle_w8_begin   workspace, v = S a_hat, statistics + digit slices of S;
//   eagle_w8_vrows   the rows [r0, r1) of V's image have landed: their statistics and digit slices -- and, when the caller allows a
//                    GUESS of the first product's configuration (the one the last call of this context chose: V changes little
//                    between the find_qtl calls of an AM() run), the columns [r0, r1) of X = S V^T at once (exact integer level sums:
//                    the same bits whatever the column blocking);
//   eagle_w8_finish  statistics of all of V, the configuration the RULE chooses; the first product is kept only if the guess WAS that
//                    configuration (else it is formed anew: the result never depends on the guess), then S X^T, the bound, r.
// eagle_dev_scan_operands_w8 = the three steps on resident operands, no guess.
// All three return EAGLE_OK, 1 = declined (run the fp64 products), or an error.
// ------------------------------------------------------------------------------------------------
#define W8_VBLOCK 1536   /* rows of V per pipelined step: 4 column tiles of 384 = 6 of 256 */
struct W8Pipe {
    bool open = false, guessing = false, product1_done = false;
    int guess = -1;
    long n = 0, np = 0, rows_done = 0;
    const double *Sa = nullptr, *Va = nullptr;
    double *v = nullptr, *Wu = nullptr, *tmp = nullptr;
    W8Ws w;
};
static W8Pipe* w8_pipe(eagle_ctx* ctx) {
    if (!ctx->w8_pipe) ctx->w8_pipe = new W8Pipe;
    return (W8Pipe*)ctx->w8_pipe;
}
void eagle_w8_release(eagle_ctx* ctx) {
    delete (W8Pipe*)ctx->w8_pipe;
    ctx->w8_pipe = nullptr;
}
static int w8_decline(eagle_ctx* ctx, W8Info& info, int why) {
    info.declined = why;
    ctx->w8_info = info;
    w8_pipe(ctx)->open = false;
    return 1;
}
extern "C" int eagle_w8_vrows_block(void) { return W8_VBLOCK; }

extern "C" int eagle_w8_begin(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long np, double* v_out, double* Wu_out,
                              double* tmp, int allow_guess, void* stream) {
    ctx->w8_active = false;
    ctx->w8_eta = 0.0;
    ctx->w8_info = W8Info();
    W8Pipe* P = w8_pipe(ctx);
    P->open = false;
    if (np % T8 || n > np || n <= 0 || (double)np * T8 >= 2147483648.0 || np / T8 > 4095) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (w8_workspace(ctx, np, &P->w)) { ctx->w8_info.declined = 5; return 1; }
    W8Ws& w = P->w;
    int rc = eagle_dev_scan_operands_begin(ctx, Sa, ahat, n, np, v_out, tmp, stream);
    if (rc) return rc;
    // statistics + digit slices of the off-diagonal part F of S (V's follow row block by row block: eagle_w8_vrows)
    rc = w8_stats_of(ctx, Sa, np, w.dS, w.mxS, w.ssqS, w.eS, w.dssqS, w.sA, nullptr, true, w, w.stats + 0, s);
    if (rc) return rc;
    hipError_t er = hipMemsetAsync(w.bad, 0, sizeof(int), s);   // (S's flag is in its statistics already; V's rows raise it again)
    if (er != hipSuccess) return eagle_fail_hip(ctx, er, "w8 memset");
    P->open = true;
    P->n = n; P->np = np; P->Sa = Sa; P->Va = Va; P->v = v_out; P->Wu = Wu_out; P->tmp = tmp;
    P->rows_done = 0;
    P->product1_done = false;
    // a guess is only worth having when the level images of one column block fit the workspace in one row panel
    const int maxp = (int)std::min<long>(8, 131071 / np);
    P->guessing = allow_guess && ctx->w8_guess_np == np && ctx->w8_guess_c1 >= 0 && maxp >= 1 &&
                  w8_groups(W8_CONFIGS[ctx->w8_guess_c1], maxp).size() * (size_t)np * (size_t)std::min<long>(np, W8_VBLOCK + 255) * sizeof(int32_t) <= w.level_bytes;
    P->guess = P->guessing ? ctx->w8_guess_c1 : -1;
    return EAGLE_OK;
}

static void w8_launch_combine1(W8Pipe* P, const W8List& l, long img, long row0, long rows, long c0, long ncols, hipStream_t s) {
    W8Ws& w = P->w;
    hipLaunchKernelGGL(k_w8_combine1, dim3((unsigned)(ncols / 32), (unsigned)(rows / 32)), dim3(256), 0, s, P->Sa, P->Va, P->np, (const double*)w.dS,
                       (const double*)w.dV, (const int*)w.eS, (const int*)w.eV, (const int32_t*)w.levels, img, (const W8Group*)l.groups, l.ngroups, row0, P->tmp,
                       c0, ncols);
}

extern "C" int eagle_w8_vrows(eagle_ctx* ctx, long r0, long r1, void* stream) {
    W8Pipe* P = w8_pipe(ctx);
    if (!P->open) return 1;
    if (r0 != P->rows_done || r1 <= r0 || r1 > P->np) return eagle_fail(ctx, EAGLE_ERR_ARG, "w8_vrows: row blocks must arrive in order");
    hipStream_t s = (hipStream_t)stream;
    W8Ws& w = P->w;
    int rc = w8_rows_of(ctx, P->Va, P->np, r0, r1, w.dV, w.mxV, w.ssqV, w.eV, w.dssqV, w.sB, w, s);
    if (rc) return rc;
    P->rows_done = r1;
    if (P->guessing && r0 % W8_VBLOCK == 0 && (r1 % W8_VBLOCK == 0 || r1 == P->np)) {
        // the columns [r0, r1) of X on the guessed configuration (kept by eagle_w8_finish only if the rule chooses the same one)
        rc = w8_product(ctx, P->guess, false, P->np, w.sA, w.sB, w, s,
                        [&](const W8List& l, long img, long row0, long rows) { w8_launch_combine1(P, l, img, row0, rows, r0, r1 - r0, s); }, r0, r1);
        if (rc < 0) return rc;
        if (rc) P->guessing = false;
    } else P->guessing = false;
    return EAGLE_OK;
}

extern "C" int eagle_w8_finish(eagle_ctx* ctx, void* stream) {
    W8Pipe* P = w8_pipe(ctx);
    if (!P->open) return 1;
    if (P->rows_done != P->np) return eagle_fail(ctx, EAGLE_ERR_ARG, "w8_finish: rows of V missing");
    P->open = false;
    hipStream_t s = (hipStream_t)stream;
    W8Ws& w = P->w;
    const double *Sa = P->Sa, *Va = P->Va;
    double *tmp = P->tmp, *Wu_out = P->Wu;
    const long n = P->n, np = P->np;
    int rc = w8_reduce_of(ctx, Va, np, w.dV, w.mxV, w.ssqV, w.eV, w.dssqV, w.dS, true, w, w.stats + 1, s);
    if (rc) return rc;
    // r = the row sums of sym(S V S) in fp64: the mean of (Sa Va Sa) 1 (row-type products) and (Sa Va Sa)^T 1 (column-type)
    hipLaunchKernelGGL(k_w8_fill_ones, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, w.r2, n, np);
    rc = eagle_dev_colgemv_parts(ctx, Sa, n, np, w.r2, w.r1, w.gvpart, stream);
    if (!rc) rc = eagle_dev_colgemv_parts(ctx, Va, n, np, w.r1, w.r, w.gvpart, stream);
    if (!rc) rc = eagle_dev_colgemv_parts(ctx, Sa, n, np, w.r, w.r1, w.gvpart, stream);     // r1 = (Sa Va Sa)^T 1
    if (rc) return rc;
    hipLaunchKernelGGL(k_w8_rowgemv, dim3((unsigned)(np / 4)), dim3(256), 0, s, Sa, n, np, (const double*)w.r2, w.r);
    hipLaunchKernelGGL(k_w8_rowgemv, dim3((unsigned)(np / 4)), dim3(256), 0, s, Va, n, np, (const double*)w.r, w.gvpart);
    hipLaunchKernelGGL(k_w8_rowgemv, dim3((unsigned)(np / 4)), dim3(256), 0, s, Sa, n, np, (const double*)w.gvpart, w.r);   // r = (Sa Va Sa) 1
    hipLaunchKernelGGL(k_w8_mean2, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (const double*)w.r, (const double*)w.r1, np, w.r);
    W8Stats st[3];
    if ((rc = w8_fetch(ctx, st, w.stats, 2 * sizeof(W8Stats), s))) return rc;
    W8Info info;
    if (st[0].bad || st[1].bad) return w8_decline(ctx, info, 1);
    const double up = 1.0 + 1e-9;
    const double normS = (st[0].maxd + sqrt(st[0].fro2 * up)) * up, normV = (st[1].maxd + sqrt(st[1].fro2 * up)) * up;
    const double wd_est = st[1].wdsum / (double)n;               // mean_k |W_kk| ~ mean_k D_k^2 |Dv_k|
    // chosen for the TIGHT budget the scan tries first (k_spectral_decide): a W error that ate the little room the spectral bound leaves there
    // would cost the scan its tighter certificate (or a digit); accepted, at the end, against the budget the scan falls back to
    const double target = W8_TARGET * ctx->scan_budget_tight * wd_est;
    // The images stand in for their transposes (NT products, upper triangle only).  With A = (Sa - Sa^T)/2, B = (Va - Va^T)/2:
    // computed W' = Sa Va^T Sa^T = Sa Vs Sa^T - Sa B Sa^T against T' = Sa Va Sa, whose symmetric part is what the scan's quadratic
    // forms see:  || W' - T' ||_F <= 2 ||S|| (||V|| ||A||_F + (||A|| + max|D|) ||B||_F) (generous),  and folding the upper triangle of T' instead of
    // symmetrising it costs || antisym(T') ||_F <= 2 ||A||_F ||V|| ||S|| + ||S||^2 ||B||_F (+ second order).  Visibly asymmetric
    // operands make this term large and the call declines (the fp64 path then takes its general products).
    const double nA = 0.5 * st[0].asym * up, nB = 0.5 * st[1].asym * up;
    const double asym = (M_SQRT2 * 2.0 * normS * (normV * nA + (nA + st[0].maxd) * nB) + 2.0 * nA * normV * normS + normS * normS * nB +
                         2.0 * (nA * nA * normV + 2.0 * nA * nB * normS)) * up;
    info.asym_term = asym;
    if (!(target > 0.0) || !(asym <= 0.25 * target)) return w8_decline(ctx, info, 2);
    double b1 = 0.0;
    // (the bounds fall ~13x per configuration step: the second product is left with at least 30 % of the target, rarely one step's worth)
    const int c1 = w8_choose(st[0], st[1], np, 0.7 * (target - asym) / (normS * M_SQRT2), &b1);
    if (c1 < 0) return w8_decline(ctx, info, 3);
    ctx->w8_guess_c1 = c1;
    ctx->w8_guess_np = np;
    info.pipelined = P->guessing && P->guess == c1;
    if (!info.pipelined) {   // no guess, or not the configuration the rule chooses: X = S V^T now, all of it
        rc = w8_product(ctx, c1, false, np, w.sA, w.sB, w, s,
                        [&](const W8List& l, long img, long row0, long rows) { w8_launch_combine1(P, l, img, row0, rows, 0, np, s); });
        if (rc) { if (rc == 1) return w8_decline(ctx, info, 5); return rc; }
    }
    // X = tmp: statistics + slices (over Fv's)
    rc = w8_stats_of(ctx, tmp, np, w.dX, w.mxX, w.ssqX, w.eX, w.dssqX, w.sB, nullptr, false, w, w.stats + 2, s);
    if (rc) return rc;
    if ((rc = w8_fetch(ctx, st + 2, w.stats + 2, sizeof(W8Stats), s))) return rc;
    if (st[2].bad) return w8_decline(ctx, info, 1);
    // rounding of the element-wise terms and the level sums of X: a handful of roundings on terms of these sizes
    const double fS = sqrt(st[0].fro2 * up) * up, fV = sqrt(st[1].fro2 * up) * up, fX = sqrt(st[2].fro2 * up) * up;
    const double round1 = ldexp(st[0].maxd * fV + fS * st[1].maxd + fS * fV + st[0].maxd * st[1].maxd * sqrt((double)np), -49);
    const double etaX = (b1 + round1) * up;
    const double used = (normS * etaX * M_SQRT2 + asym) * up;
    double b2 = 0.0;
    const int c2 = w8_choose(st[0], st[2], np, (target - used) / M_SQRT2 * 0.98, &b2);
    if (c2 < 0) return w8_decline(ctx, info, 4);
    rc = w8_product(ctx, c2, true, np, w.sA, w.sB, w, s, [&](const W8List& l, long img, long row0, long rows) {
        hipLaunchKernelGGL(k_w8_combine2, dim3((unsigned)(np / 32), (unsigned)(rows / 32)), dim3(256), 0, s, Sa, (const double*)tmp, np, (const double*)w.dS,
                           (const double*)w.dX, (const int*)w.eS, (const int*)w.eX, (const int32_t*)w.levels, img, (const W8Group*)l.groups, l.ngroups, row0,
                           Wu_out);
    });
    if (rc) { if (rc == 1) return w8_decline(ctx, info, 5); return rc; }
    const double round2 = ldexp(st[0].maxd * fX + fS * st[2].maxd + fS * fX + st[0].maxd * st[2].maxd * sqrt((double)np), -49);
    const double eta = ((normS * etaX + b2 + round2) * M_SQRT2 + asym) * up;
    // the a-posteriori check against the diagonal of the W just made
    hipLaunchKernelGGL(k_w8_sumdiag, dim3(1), dim3(1024), 0, s, (const double*)Wu_out, np, w.sumdiag);
    double sumdiag = 0.0;
    if ((rc = w8_fetch(ctx, &sumdiag, w.sumdiag, sizeof(double), s))) return rc;
    info.config1 = c1; info.config2 = c2;
    info.k1 = W8_CONFIGS[c1].k; info.T1 = W8_CONFIGS[c1].T; info.k2 = W8_CONFIGS[c2].k; info.T2 = W8_CONFIGS[c2].T;
    info.pairs1 = w8_config_pairs(W8_CONFIGS[c1]); info.pairs2 = w8_config_pairs(W8_CONFIGS[c2]);
    info.eta = eta; info.eta_x = etaX; info.bound1 = b1; info.bound2 = b2; info.norm_s = normS; info.target = target;
    info.mean_diag = sumdiag / (double)n;
    if (!(eta <= W8_ACCEPT * ctx->scan_budget * info.mean_diag)) return w8_decline(ctx, info, 6);
    ctx->w8_info = info;
    ctx->w8_active = true;
    ctx->w8_eta = eta;
    ctx->w8_r = w.r;
    ctx->w8_Wu = Wu_out;
    ctx->w8_Sa = Sa;
    ctx->w8_Va = Va;
    ctx->w8_n = n;
    ctx->w8_tmp = tmp;
    return EAGLE_OK;
}

extern "C" int eagle_dev_scan_operands_w8(eagle_ctx* ctx, const double* Sa, const double* Va, const double* ahat, long n, long np, double* v_out,
                                          double* Wu_out, double* tmp, void* stream) {
    int rc = eagle_w8_begin(ctx, Sa, Va, ahat, n, np, v_out, Wu_out, tmp, 0, stream);
    if (!rc) rc = eagle_w8_vrows(ctx, 0, np, stream);
    if (!rc) rc = eagle_w8_finish(ctx, stream);
    return rc;
}
