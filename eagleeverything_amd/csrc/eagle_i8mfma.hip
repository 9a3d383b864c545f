// eagle_i8mfma.hip -- the int8 MFMA tile engine (v_mfma_i32_32x32x32_i8) and the two kernels built on it.
//
//   k_syrk_i8  : MM^T partial sums, C32 += M8 * M8^T                        (E/src/calculateMMt_rcpp.cpp:95)
//   k_vara_i8  : vara_i = m_i^T W m_i from exact int8 digit slices of W     (E/src/calculate_a_and_vara_rcpp.cpp:97-112)
//
// Tile engine.  Workgroup = 512 threads = 8 waves arranged 2 (M) x 4 (N); block tile 256 x 256; wave tile
// 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator registers); K step 128 bytes.  Both operands are
// "row-major, K contiguous" int8 (NT product C[i][j] = sum_k A[i][k] B[j][k]).  Operand tiles (256 rows x 128 B
// = 32 KiB each) go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 rows per wave-instruction, no
// VGPR staging, no ds_write), double buffered (2 x 64 KiB), the loads of stage t+1 in flight under the 32 MFMAs
// per wave of stage t.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7, applied to
// the per-lane SOURCE address of the DMA (its LDS destination is lane-linear) and to the ds_read_b128 address,
// which makes the 32-row fragment reads bank-conflict free.
// A and B fragments use the same (lane>>5, byte) -> k map, so the k order inside the instruction is immaterial
// for an exact integer sum.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eagle_hip.h"
#include "eagle_ctx.h"
#include "eagle_internal.h"

#include "eagle_t8.h"

// ------------------------------------------------------------------------------------------------
// MM^T: grid.x = upper-triangular 256-tile pairs, grid.y = K splits.  Integer atomics: exact, any order.
// ------------------------------------------------------------------------------------------------
template <int TUNE>
__global__ __launch_bounds__(512, 2) void k_syrk_i8(const int8_t* __restrict__ M8, long ld, const int* __restrict__ pairs,
                                                    int npairs, int nblocks, long nstages, long stages_per_split,
                                                    int32_t* __restrict__ C, long ldc) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    // XCD-aware order (speed only): workgroup b runs on XCD b%8 (observed dealing); give each XCD a contiguous range
    // of the logical work list, which is ordered K-split major and, inside a split, by G x G super-tiles of the
    // upper triangle, so the ~32 workgroups resident on one XCD stream the same few row/column panels of M8.
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (lid >= nblocks) return;
    const int split = lid / npairs;
    const int pr = pairs[lid - split * npairs];
    const int ti = pr >> 16, tj = pr & 0xffff;
    const long s0 = (long)split * stages_per_split;
    long s1 = s0 + stages_per_split;
    if (s1 > nstages) s1 = nstages;
    if (s0 >= s1) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld;
    const T8Lane ln = t8_lane(lane, ldi);
    const __amdgpu_buffer_rsrc_t rsA = t8_rsrc(M8 + (long)ti * T8 * ld, ldi);
    const __amdgpu_buffer_rsrc_t rsB = t8_rsrc(M8 + (long)tj * T8 * ld, ldi);
    i32x16 acc[4][2];
    t8_zero(acc);
    t8_stage(rsA, ln, ldi, (int)(s0 * BK8), lds[0][0], w);
    t8_stage(rsB, ln, ldi, (int)(s0 * BK8), lds[0][1], w);
    __syncthreads();
    int cur = 0;
    const T8Read rd = t8_read_init(wr, wc, lane);
    for (long s = s0; s < s1; s++) {
        const int kn = (int)((s + 1) * BK8);
        t8_stage_compute<TUNE>(acc, lds[cur][0], lds[cur][1], rd, s + 1 < s1, rsA, ln, ldi, kn, lds[cur ^ 1][0], rsB, ln, ldi, kn,
                               lds[cur ^ 1][1], w);
        __syncthreads();
        cur ^= 1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = lane & 31, rq = 4 * (lane >> 5);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                long i = (long)ti * T8 + wr * 128 + m * 32 + (q & 3) + 8 * (q >> 2) + rq;
                long j = (long)tj * T8 + wc * 64 + n * 32 + col;
                int v = acc[m][n][q];
                if (v) atomicAdd(&C[i * ldc + j], v);
            }
}

// Upper-triangular tile pairs (ti<<16 | tj) in super-tile order, cached on the device per tile count.
#include <map>
#include <mutex>
#include <vector>
static std::map<std::pair<int, int>, int*> g_pair_tables;  // (device, nt) -> device table
static std::mutex g_pair_mutex;                             // one worker thread per device may come through here
static int syrk_pair_table(eagle_ctx* ctx, int nt, const int** out, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto key = std::make_pair(dev, nt);
    auto it = g_pair_tables.find(key);
    if (it != g_pair_tables.end()) { *out = it->second; return EAGLE_OK; }
    const int G = 6;  // 6 x 6 tiles = 36 workgroups ~ one XCD's worth share 6 row panels + 6 column panels
    std::vector<int> h;
    for (int si = 0; si < nt; si += G)
        for (int sj = si; sj < nt; sj += G)
            for (int i = si; i < si + G && i < nt; i++)
                for (int j = (sj > i ? sj : i); j < sj + G && j < nt; j++) h.push_back((i << 16) | j);
    int* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, h.size() * sizeof(int));
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "pair table alloc");
    e = hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return eagle_fail_hip(ctx, e, "pair table copy"); }
    g_pair_tables[key] = d;
    *out = d;
    return EAGLE_OK;
}

// Experiment switch for the tile-engine schedule (tools/bench_*.py); 0 is the shipped default.  Per ctx.  Not part of the public
// ABI.  The switch also decides the layout eagle_dev_vara_i8_prepare gives the digit slices (vara_piped below): set it BEFORE the
// prepare call of the scan it is meant for.
extern "C" void eagle_dev_set_tune(eagle_ctx* ctx, int v) { if (ctx) ctx->tune = v; }

// The int8 form of the MM^T kernel (the shipped one is k_syrk_f4 below: same engine, fp4 operands, twice the markers per
// stage).  Kept for the schedule ablations of tools/bench_i8_engine.py; not part of the public ABI.
extern "C" int eagle_dev_mmt_accumulate_i8(eagle_ctx* ctx, const int8_t* M8, long n_pad, long L_pad, long ld, int32_t* C32,
                                           void* stream) {
    if (n_pad % T8 || L_pad % BK8 || ld % 128 || L_pad > ld || n_pad <= 0 || (double)ld * T8 >= 2147483648.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_accumulate: layout contract violated (n_pad % 256, L_pad % 128, ld % 128, ld < 2^23)");
    if (L_pad == 0) return EAGLE_OK;
    const int nt = (int)(n_pad / T8);
    const long npairs = (long)nt * (nt + 1) / 2;
    const long nstages = L_pad / BK8;
    // one workgroup per CU (128 KiB LDS): aim at ~10 waves of 256 workgroups, K runs of at least 16 stages
    long want = (10L * 256 + npairs - 1) / npairs;
    long maxsplit = nstages / 16 > 0 ? nstages / 16 : 1;
    long nsplit = want < maxsplit ? want : maxsplit;
    if (nsplit < 1) nsplit = 1;
    long per = (nstages + nsplit - 1) / nsplit;
    nsplit = (nstages + per - 1) / per;
    const long nblocks = npairs * nsplit;
    if (nblocks >= (1L << 30)) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_accumulate: too many workgroups");
    const int* pairs = nullptr;
    int rc = syrk_pair_table(ctx, nt, &pairs, (hipStream_t)stream);
    if (rc) return rc;
    dim3 grid((unsigned)((nblocks + 7) / 8 * 8));
#define SYRK_LAUNCH(T) hipLaunchKernelGGL(k_syrk_i8<T>, grid, dim3(512), 0, (hipStream_t)stream, M8, ld, pairs, (int)npairs, (int)nblocks, nstages, per, C32, n_pad)
    switch (ctx->tune) {
        case 3: SYRK_LAUNCH(3); break;
        case 4: SYRK_LAUNCH(4); break;
        default: SYRK_LAUNCH(0);
    }
#undef SYRK_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_syrk_i8");
    return EAGLE_OK;
}

// ================================================================================================
// vara from int8 digit slices of Wu.
//
//   Wu (upper triangular fold of W, fp64) = 2^(e+2) * sum_{s<S} D_s * 256^-(s+1)  +  R,   |R_jk| <= 2^(e+1-8S)
//   with max|Wu| < 2^e (< 1.96 * 2^e when the mantissa leaves room: w_scale_exp), D_s in [-128,127]: the balanced base-256 digits of the exact integer
//   round(Wu * 2^(8S-e-2)) (below 2^(8S-2) <= 2^62, so int64 arithmetic; a balanced digit set needs 257 values per
//   level when produced most-significant first, so the digits are peeled least-significant first with a carry).
//   q_s[i] = sum_k m_ik sum_{j<=k} m_ij D_s[j][k]   is an exact integer (int32 MFMA sums, int64 across tiles), so
//   vara_i = 2^(e+2) * sum_s 256^-(s+1) q_s[i]  carries only the truncation R: |error| <= (sum_j |m_ij|)^2 * 2^(e+1-8S),
//   and the final S-term fp64 sum (smallest term first).  S = 7 covers the whole 53-bit mantissa of the largest
//   element of W.  Nothing depends on the order in which tiles finish: bitwise reproducible.
//
// Slices are stored transposed, Bs[s][k][j] = D_s[j][k] (row = output column k, K = j contiguous), so the scan is the
// NT product  T_s[i][k] = sum_j Mt8[i][j] * Bs[s][k][j]  with k-tile ct needing only j < (ct+1)*256 (lower
// triangular image).  The row-dot with m_ik is fused into the tile epilogue.
//
// Work decomposition.  A marker tile (256 markers) is served by S workgroups, one per digit slice ("workers").  Every
// worker walks the same sequence of column-tile pairs {p, nct-1-p} (each pair costs (nct+1)*2 stages) and the same k
// order inside them, so at any moment the S workers of a marker tile want the SAME genotype stage and differ only in
// the W-slice they stream: when they sit on one XCD (workgroup b runs on XCD b%8, observed dealing; the workers of a
// marker tile are given consecutive slots of one XCD) the genotype panel is fetched once into that XCD's L2 and read
// S times from there, and equal-numbered workers of the XCD's other resident marker tiles stream the same W-slice
// tiles in lock-step.  This placement is a guess that only affects speed.  The integer partial row-dots of the
// workers meet in q[s][i] by int64 atomics (one flush per worker).
// ================================================================================================

struct VaraHdr {        // head of the workspace, written on the device, never read by the host
    double maxabs_off;  // max |Wu[j][k]|, j != k
    int S;              // digit slices in use
    int pad;
    double bound;       // n_pad^2 * 2^(e+1-8S): absolute error bound of every vara_i
    double sumdiag;     // sum_k |Wu[k][k]|
    double R;           // sum_{j<k} Wu[j][k] (the off-diagonal quadratic form of the all-ones vector)
    double specH;       // > 0: the scan runs on S = S_sliced - 1 digits and |digit error of marker i| <= specH * sum_j m'_ij^2 (k_spectral_decide)
    int S_sliced;       // digits k_slice_w cut (the scale of the integers Q); S_sliced - S is 0 or 1
    int pad2;
    double budget;      // the digit budget of this scan (eagle_set_scan_budget)
    int e;              // the scale exponent w_scale_exp(maxabs_off): the digits are those of round(Wu * 2^(8 S_sliced - e - 2))
    int pad3;
    // second level of the spectral bound (k_gram_hi_i8): sum of squares of the low part of offdiag(Ds Ds), its largest diagonal entry,
    // whether a high part left int8, whether level 1 declined and level 2 is to run, and the level that took the digit off (0: none)
    unsigned long long lo_sumsq;
    int maxdiag, hi_overflow, spec_try2, level;
    // W itself came from int8 digit slices (eagle_w8.hip): || Wu - truth ||_F <= wErr, i.e. |error of marker i| <= wErr sum_j m'_ij^2 on top
    // of the digit terms (0: the fp64 products)
    double wErr;
    // round 4: the budget is tried TIGHT first (1e-7 unless eagle_set_scan_budget fixed one): `budget` above is the one in force for this
    // scan -- the tight one if the digits that run certify a marker with q2 = n_pad to it, else the default; specH1 = level 1's bound while
    // level 2 is being tried
    double specH1;
    // the default budget behind a tight one in force (= budget otherwise): what the certificate ENFORCES per marker falls back to it when
    // more than CERT_TIGHT_MAX markers of the scan miss the tight threshold (see k_cert_select)
    double budget_loose;
};
__global__ __launch_bounds__(256) void k_absmax_offdiag(const double* __restrict__ x, long np, unsigned long long* __restrict__ bits) {
    double m = 0.0;
    const long n = np * np;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        double v = fabs(x[i]);
        if (i % (np + 1) == 0) v = 0.0;  // the diagonal is handled in fp64 (k_vara_prep / gemv_sq)
        m = v > m ? v : m;                // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(m, o); m = y > m ? y : m; }
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(bits, (unsigned long long)__double_as_longlong(m));
}

// Digit-count rule (forced = 0).  The path's stated tolerance is 1e-6 relative on the score statistics; the digits get a
// tenth of it: the smallest S whose WORST-CASE bound n^2 * 2^(e+1-8S) -- every truncation error aligned, a marker with n
// non-zero entries -- is below 1e-7 of 0.5 * sum_k |W_kk|, the diagonal term of a marker with n/2 non-zero entries.
// Per marker the bound is (sum_j |m'_ij|)^2 / 2 * 2^(e+1-8S) on the RE-CENTRED genotypes m' (k_marker_shift below), so it
// scales with the square of the marker's non-majority count while its quadratic form scales with the count itself:
// rare-variant markers, whose vara is far below the diagonal term of the raw g-1 coding, keep the same relative bound,
// and monomorphic markers (vara = 0 up to fp64 noise in the reference too) take no digit error at all.
// Measured errors sit three orders below the bound (C2, S = 4: bound 1.7e-8, largest error against an fp64 evaluation
// 1.5e-11).  eagle_set_scan_slices forces more digits.
// Round 3: the budget is a property of the context (eagle_set_scan_budget, default VARA_DIGIT_BUDGET = 5e-7 = half the path's
// tolerance; 1e-7 until round 2), and one digit fewer than the worst case asks for is taken when the SPECTRAL bound below allows it.
#define VARA_DIGIT_BUDGET 5e-7
// What the certificate enforces per marker: a bound above VARA_FLAG_FACTOR x budget sends the marker to the fp64 kernel (0.9e-6 of
// the 1e-6 tolerance with the default budget; the rest covers the fp64 roundings, orders of magnitude smaller).
#define VARA_FLAG_FACTOR 1.8
// Stochastic rounding (EAGLE_SLICES_STOCHASTIC, opt-in).  With round-to-nearest the truncation errors R_jk of the digits are
// only known to lie in [-delta, delta], delta = 2^(e+1-8S), and the guaranteed bound must add them up in absolute value:
// l1^2/2 * delta.  If instead each entry is rounded down or up AT RANDOM with the probabilities that make the rounding
// unbiased (a counter-based generator keyed by the entry's position, independent of the data), the R_jk are independent,
// zero-mean and confined to an interval of width 2 delta, and Hoeffding's inequality bounds their weighted sum:
//     P( |sum_{j<k} m'_j m'_k R_jk| >= t )  <=  2 exp( -2 t^2 / sum_{j<k} (2 delta m'_j m'_k)^2 )  <=  2 exp( -(t / (delta q2))^2 ),
// q2 = sum_j m'_j^2.  With t = 8.355 delta q2 the right side is 1e-30 per marker -- below 1e-22 over every marker of every
// scan of an AM() run -- and the bound grows with the marker's non-zero count instead of its square: one digit fewer carries
// the same 1e-7 budget (n = 5000 .. 10000).  The certificate becomes a probabilistic one (over the library's own random
// bits, for any input); everything downstream -- per-marker flagging, fp64 re-evaluation of the candidates -- is unchanged.
#define VARA_HOEFFDING_K 8.355  /* sqrt(ln(2 / 1e-30)) */
// One block: dW[k] = Wu[k][k] (contiguous copy), sumdiag in a fixed order, then the slice count (forced = 1..8: that S).
__global__ __launch_bounds__(256) void k_vara_prep(const double* __restrict__ Wu, long n_pad, int forced_arg, VaraHdr* __restrict__ hdr,
                                                   double* __restrict__ dW, double budget, double wErr, double tight) {
    const int forced = forced_arg & 0xff, stochastic = (forced_arg & EAGLE_SLICES_STOCHASTIC) ? 1 : 0;
    double s = 0.0;
    for (long k = threadIdx.x; k < n_pad; k += 256) {
        double d = Wu[k * n_pad + k];
        dW[k] = d;
        s += fabs(d);
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mx = hdr->maxabs_off;
        const int e = w_scale_exp(mx);
        const double nn = (double)n_pad * (double)n_pad;
        int S = forced;
        if (S <= 0) {
            const double target = budget * 0.5 * red[0];
            S = 7;
            for (int c = stochastic ? 2 : 3; c <= 7; c++) {
                // worst case over markers: every entry non-zero (nearest: errors aligned; stochastic: q2 = n, the Hoeffding radius)
                const double b = stochastic ? VARA_HOEFFDING_K * (double)n_pad * ldexp(1.0, e + 1 - 8 * c) : ldexp(nn, e + 1 - 8 * c);
                if (b + wErr * (double)n_pad <= target) { S = c; break; }   // (wErr: the marker with q2 = n_pad again)
            }
        }
        if (mx == 0.0) S = 1;
        hdr->S = S;
        hdr->S_sliced = S;
        hdr->specH = 0.0;
        // in force: the tight budget if the worst-case bound of the digits chosen meets it too (a forced count: the context's budget)
        hdr->budget = (forced <= 0 && mx > 0.0 && tight < budget &&
                       (stochastic ? VARA_HOEFFDING_K * (double)n_pad * ldexp(1.0, e + 1 - 8 * S) : ldexp(nn, e + 1 - 8 * S)) + wErr * (double)n_pad <= tight * 0.5 * red[0])
                          ? tight : budget;
        hdr->specH1 = 0.0;
        hdr->budget_loose = budget;
        hdr->wErr = wErr;
        hdr->e = e;
        hdr->pad = stochastic;
        hdr->sumdiag = red[0];
        hdr->bound = mx > 0.0 ? ldexp(nn, e + 1 - 8 * S) * (stochastic ? 2.0 : 1.0) : 0.0;  // what can never be exceeded
    }
}

// ------------------------------------------------------------------------------------------------
// One digit fewer under a SPECTRAL bound of the truncation error (round 3).
//
// With S digits cut (k_slice_w), Wu_jk / u = Q_jk + rho_jk, u = 2^(e+2-8S), |rho_jk| <= 1/2, Q = sum_s d_s 256^(S-1-s).  A scan on the
// leading S-1 digits alone leaves, for marker i (re-centred row m'),
//     err_i = sum_{j<k} m'_j m'_k (Wu_jk - u (Q_jk - d_jk))  =  (u/2) m'^T (Ds + P) m',     d = the LAST digit, Ds = d + d^T, P = rho + rho^T,
// so |err_i| <= (u/2) (||Ds||_2 + ||P||_2) q2_i with q2_i = sum_j m'_j^2 -- against the worst case l1_i^2/2 * 128.5 u of errors that all
// line up.  The last digit of W is as good as random: ||Ds||_2 ~ 2 sqrt(n) sigma_d ~ 150 sqrt(n), and the spectral bound wins by
// ~ l1^2 / (q2 sqrt(n)): two orders of magnitude for a common marker at n = 10,000, which with the budget above pays for a whole digit
// (a quarter of the matrix work of the scan) on the operands of an AM() run.
// The bound has to be RIGOROUS, and cheap next to the 33 ms it saves:  ||P||_2 <= ||P||_inf <= (n_pad - 1)/2, and
//     ||Ds||_2^2 = lambda_max(Ds Ds) <= max_j sum_k |(Ds Ds)_jk|          (Gershgorin on the Gram matrix)
// where Ds Ds is an EXACT int32 matrix product of an int8 matrix with itself: the tile engine above, 2 n^3 int8 MAC-flop (~1 ms at
// n = 10,240), never stored -- the tile epilogue adds |.| into per-row int64 sums (k_gram_rowabs_i8).  Row sums of a random Gram matrix
// overestimate lambda_max by ~0.45 n^(1/4) in the norm (5.3x at n = 10,240: tools/diag_spectral.py prints it against a power iteration);
// enough.  k_spectral_decide then takes S-1 digits if a marker with q2 = n_pad stays inside the budget against 0.5 sum_k |W_kk| -- the
// worst-case rule's yardstick -- and leaves specH = (u/2)(||Ds|| + (n_pad-1)/2) in the header: cert_bound uses
// min(specH q2_i, l1_i^2/2 * 128.5 u) per marker.  Markers whose own bound still exceeds the budget are re-evaluated in fp64 like any
// flagged marker; should more than CERT_CAP of them do (the fp64 fallback of the whole block), the context stops taking the digit
// off (eagle_api.cpp).  Deterministic, a function of W alone: every device and every marker block of a scan decides the same.
// Not used with a forced digit count, with stochastic rounding, or under tune 29 (A/B switch).
// ------------------------------------------------------------------------------------------------
// Ds[i][j] = last digit of Wu[min(i,j)][max(i,j)], 0 on the diagonal: symmetric int8 image, n_pad x n_pad (same llrint as k_slice_w)
__global__ __launch_bounds__(256) void k_last_digit_sym(const double* __restrict__ Wu, long np, const VaraHdr* __restrict__ hdr, int8_t* __restrict__ Ds,
                                                        int smax) {
    __shared__ double tile[32][33];
    if (hdr->S_sliced >= smax) return;   // every slot of the slice area holds a digit in use: no spare one for this image (S = 7: never taken down)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long bx = blockIdx.x, by = blockIdx.y;
    const long lo = bx < by ? bx : by, hi = bx < by ? by : bx;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = Wu[(lo * 32 + r) * np + hi * 32 + tx];  // tile[a][b] = Wu[32 lo + a][32 hi + b]
    __syncthreads();
    const int e = hdr->e;
    const int S = hdr->S_sliced;
    for (int r = ty; r < 32; r += 8) {
        double v;   // element (i = 32 by + r, j = 32 bx + tx)
        if (by < bx) v = tile[r][tx];
        else if (by > bx) v = tile[tx][r];
        else v = r < tx ? tile[r][tx] : (r > tx ? tile[tx][r] : 0.0);
        const long long Q = llrint(ldexp(v, 8 * S - (e + 2)));
        Ds[(by * 32 + r) * np + bx * 32 + tx] = (int8_t)(((Q + 128) & 255) - 128);
    }
}

// rs[i] += sum_j |(D D^T)_ij| over the upper-triangular 256-tile pairs of the symmetric product (both the row sums of a tile and, off
// the diagonal, its column sums = the row sums of the mirrored tile).  Whole K per workgroup: |.| does not commute with a K split.
__global__ __launch_bounds__(512, 2) void k_gram_rowabs_i8(const int8_t* __restrict__ D, long ld, const int* __restrict__ pairs, int npairs,
                                                           long nstages, unsigned long long* __restrict__ rs, const int* __restrict__ gate) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    if (gate && *gate == 0) return;   // level 2 only when level 1 declined
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);   // an XCD gets a contiguous range of the super-tile ordered list
    if (lid >= npairs) return;
    const int pr = pairs[lid];
    const int ti = pr >> 16, tj = pr & 0xffff;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld;
    const T8Lane ln = t8_lane(lane, ldi);
    const __amdgpu_buffer_rsrc_t rsA = t8_rsrc(D + (long)ti * T8 * ld, ldi);
    const __amdgpu_buffer_rsrc_t rsB = t8_rsrc(D + (long)tj * T8 * ld, ldi);
    i32x16 acc[4][2];
    t8_zero(acc);
    t8_stage(rsA, ln, ldi, 0, lds[0][0], w);
    t8_stage(rsB, ln, ldi, 0, lds[0][1], w);
    __syncthreads();
    int cur = 0;
    const T8Read rd = t8_read_init(wr, wc, lane);
    for (long s = 0; s < nstages; s++) {
        const int kn = (int)((s + 1) * BK8);
        t8_stage_compute<0>(acc, lds[cur][0], lds[cur][1], rd, s + 1 < nstages, rsA, ln, ldi, kn, lds[cur ^ 1][0], rsB, ln, ldi, kn,
                            lds[cur ^ 1][1], w);
        __syncthreads();
        cur ^= 1;
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  |acc| <= 128^2 K: int64 sums throughout.
    if (ti != tj) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
            unsigned long long cs = 0;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int q = 0; q < 16; q++) { const int v = acc[m][n][q]; cs += (unsigned long long)(v < 0 ? -v : v); }
            cs += __shfl_xor(cs, 32);
            if (lane < 32) atomicAdd(&rs[(long)tj * T8 + wc * 64 + n * 32 + lane], cs);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int v0 = acc[m][0][q], v1 = acc[m][1][q];
            unsigned long long r = (unsigned long long)(v0 < 0 ? -v0 : v0) + (unsigned long long)(v1 < 0 ? -v1 : v1);
            r += __shfl_xor(r, 1);
            r += __shfl_xor(r, 2);
            r += __shfl_xor(r, 4);
            r += __shfl_xor(r, 8);
            r += __shfl_xor(r, 16);
            if ((lane & 31) == 0) atomicAdd(&rs[(long)ti * T8 + wr * 128 + m * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)], r);
        }
}

// Second level (only when the first declined).  ||Ds||_2^2 = lambda_max(G), G = Ds Ds, and Gershgorin's row sums lose a factor ~0.45 n^(1/4)
// in the norm because they ignore the signs of G's off-diagonal part E -- as random as Ds itself, but ~sqrt(n) sigma^2 in size against the
// diagonal's n sigma^2.  So split:  lambda_max(G) <= max_j G_jj + ||E||_2  (Weyl),  E = 2^s E_hi + E_lo  with E_hi = round(E / 2^s) an int8
// matrix (s chosen from n so that 8 standard deviations of a random Gram entry fit; an entry that does not fit raises hi_overflow and the
// level declines),  ||E||_2 <= 2^s ||E_hi||_2 + ||E_lo||_F,  and  ||E_hi||_2^2 <= max row sum |E_hi E_hi|  is the first level again, one
// Gram product further down.  This kernel forms G tile by tile (ALL 256-tile pairs: the stores stay row-wise), writes E_hi, and leaves
// max_j G_jj and the exact sum of squares of E_lo in the header.  Measured: the norm bound 2.2x tighter than level 1 at n = 10,240.
__global__ __launch_bounds__(512, 2) void k_gram_hi_i8(const int8_t* __restrict__ D, long ld, int nt, long nstages, int shift, VaraHdr* __restrict__ hdr,
                                                       int8_t* __restrict__ Ehi) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    if (!hdr->spec_try2) return;
    // 6 x 6 super-tiles (round 4): the ~32 workgroups an XCD runs at a time share 6 row panels and 6 column panels instead of one row panel
    // and 32 column panels (L2 hit rate 48 % at n_pad = 50,176 with the row-major order); slots of a super-tile beyond the edge leave at once
    const int nsc = (nt + 5) / 6;
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (lid >= nsc * nsc * 36) return;
    const int sid = lid / 36, wi = lid - sid * 36;
    const int ti = (sid / nsc) * 6 + wi / 6, tj = (sid % nsc) * 6 + wi % 6;
    if (ti >= nt || tj >= nt) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld;
    const T8Lane ln = t8_lane(lane, ldi);
    const __amdgpu_buffer_rsrc_t rsA = t8_rsrc(D + (long)ti * T8 * ld, ldi);
    const __amdgpu_buffer_rsrc_t rsB = t8_rsrc(D + (long)tj * T8 * ld, ldi);
    i32x16 acc[4][2];
    t8_zero(acc);
    t8_stage(rsA, ln, ldi, 0, lds[0][0], w);
    t8_stage(rsB, ln, ldi, 0, lds[0][1], w);
    __syncthreads();
    int cur = 0;
    const T8Read rd = t8_read_init(wr, wc, lane);
    for (long s = 0; s < nstages; s++) {
        const int kn = (int)((s + 1) * BK8);
        t8_stage_compute<0>(acc, lds[cur][0], lds[cur][1], rd, s + 1 < nstages, rsA, ln, ldi, kn, lds[cur ^ 1][0], rsB, ln, ldi, kn,
                            lds[cur ^ 1][1], w);
        __syncthreads();
        cur ^= 1;
    }
    const int half = 1 << (shift - 1);
    unsigned long long sq = 0;
    int dmax = 0, over = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const long i = (long)ti * T8 + wr * 128 + m * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const long j = (long)tj * T8 + wc * 64 + n * 32 + (lane & 31);
                const int v = acc[m][n][q];
                int hi = 0;
                if (i == j) dmax = v > dmax ? v : dmax;
                else {
                    hi = (v + half) >> shift;                 // floor((v + 2^(s-1)) / 2^s): lo in [-2^(s-1), 2^(s-1))
                    const long long lo = (long long)v - ((long long)hi << shift);
                    sq += (unsigned long long)(lo * lo);
                    over |= (hi > 127) | (hi < -128);
                }
                Ehi[i * ld + j] = (int8_t)hi;
            }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_xor(sq, o);
        const int d2 = __shfl_xor(dmax, o);
        dmax = d2 > dmax ? d2 : dmax;
        over |= __shfl_xor(over, o);
    }
    if (lane == 0) {
        if (sq) atomicAdd(&hdr->lo_sumsq, sq);
        if (dmax) atomicMax(&hdr->maxdiag, dmax);
        if (over) atomicOr(&hdr->hi_overflow, 1);
    }
}

// One block: g = max_j rs[j]; the decision (see above).  rs is exact, the fp64 steps round up.  level 1: rs = row sums of |Ds Ds|;
// level 2 (spec_try2 left by level 1): rs = row sums of |E_hi E_hi|, with max_j G_jj and ||E_lo||_F^2 in the header.
__global__ __launch_bounds__(256) void k_spectral_decide(const unsigned long long* __restrict__ rs, long n_pad, VaraHdr* __restrict__ hdr, double budget,
                                                         double tight, int smax, int level, int shift) {
    if (level == 2 && !hdr->spec_try2) return;
    unsigned long long g = 0;
    for (long k = threadIdx.x; k < n_pad; k += 256) g = rs[k] > g ? rs[k] : g;
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = g;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x + o] > red[threadIdx.x] ? red[threadIdx.x + o] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mx = hdr->maxabs_off;
        const int S = hdr->S_sliced;
        if (mx > 0.0 && S >= 2 && S < smax && hdr->S == S && !hdr->pad) {
            const int e = hdr->e;
            const double u = ldexp(1.0, e + 2 - 8 * S);
            const double up = 1.0 + 0x1p-50;
            double H = 0.0;   // this level's bound (level 2 with an entry of E_hi outside int8: none)
            if (!(level == 2 && hdr->hi_overflow)) {
                double normsq = (double)red[0] * up;       // red[0] < 2^53 for n_pad < 2^19: the conversion is exact or rounds within `up`
                if (level == 2)   // max_j G_jj + 2^s sqrt(max row sum |E_hi E_hi|) + ||E_lo||_F, every step rounded up
                    normsq = ((double)hdr->maxdiag + ldexp(sqrt(normsq) * up, shift) + sqrt((double)hdr->lo_sumsq * up) * up) * up;
                const double normDs = sqrt(normsq) * up;
                H = 0.5 * u * (normDs + 0.5 * (double)(n_pad - 1)) * up;
            }
            // a marker with q2 = n_pad inside budget b of 0.5 sum_k |W_kk|, the error of W itself (wErr) included
            auto pass = [&](double Hx, double b) { return Hx > 0.0 && (Hx + hdr->wErr) * (double)n_pad <= b * 0.5 * hdr->sumdiag; };
            auto take = [&](double Hx, int lev, double b) {
                hdr->S = S - 1;
                hdr->specH = Hx;
                hdr->level = lev;
                hdr->budget = b;
                hdr->bound = ldexp((double)n_pad * (double)n_pad, e + 1 - 8 * (S - 1)) * (1.0 + 0x1p-8);
            };
            // Order: the tight budget first (level 1, then level 2), then the default (level 1's bound, then level 2's).
            if (level == 1) {
                if (pass(H, tight)) take(H, 1, tight);
                else if (S < smax - 1) { hdr->specH1 = H; hdr->spec_try2 = 1; }   // a spare slot of the slice area for E_hi: level 2 decides
                else if (pass(H, budget)) take(H, 1, budget);
            } else {
                if (pass(H, tight)) take(H, 2, tight);
                else if (pass(hdr->specH1, budget)) take(hdr->specH1, 1, budget);
                else if (pass(H, budget)) take(H, 2, budget);
            }
        }
        if (level == 2) hdr->spec_try2 = 0;
    }
}

// Bs[s][k][j] = digit s of Wu[j][k] (j != k; the diagonal goes through dW).  32x32 tiles through LDS so both sides
// are coalesced.
// perm128: the rows of Bs (= columns c of Wu) are stored permuted inside every block of 128 for k_vara_i8p, whose 32 x 32 result
// tiles hold the W columns in the register dimension: lane half h, column tile n, register x is row n * 32 + 8 (x >> 2) + (x & 3) + 4 h
// of the wave's 128; storing column c = 64 h + 16 n + x there makes the 64 columns of a lane CONSECUTIVE, and the tile epilogue
// fetches their genotype bytes with four 16-byte loads from the lane's own marker row.
__global__ __launch_bounds__(256) void k_slice_w(const double* __restrict__ Wu, long np, const VaraHdr* __restrict__ hdr,
                                                 int8_t* __restrict__ Bs, int perm128) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = Wu[(bj + r) * np + bk + tx];  // tile[jj][kk]
    __syncthreads();
    const int e = hdr->e;   // w_scale_exp: mx < 2^e, or mx < 1.96 * 2^e when the mantissa leaves room
    const int nslices = hdr->S;
    for (int r = ty; r < 32; r += 8) {
        // output element (k = bk + r, j = bj + tx):  Q = round(Wu * 2^(8S - e - 2)) is an exact integer below 0.49 * 256^S + 1
        // (llrint of a double of that size is exact); its balanced base-256 digits, least significant first,
        // d = ((Q + 128) mod 256) - 128 in [-128,127], Q <- (Q - d)/256; the leading digit ends in [-126,126].
        long long Q = 0;
        if (bk + r != bj + tx) {
            const double yv = ldexp(tile[tx][r], 8 * nslices - (e + 2));
            if (hdr->pad) {  // unbiased random rounding, keyed by the entry's position (see VARA_HOEFFDING_K above)
                const double fl = floor(yv);
                unsigned long long z = (unsigned long long)((bj + tx) * np + bk + r) + 0x9E3779B97F4A7C15ULL;  // splitmix64
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                z ^= z >> 31;
                z += 0x9E3779B97F4A7C15ULL;                                                                      // second round
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                z ^= z >> 31;
                const double u = (double)(z >> 11) * 0x1p-53;  // uniform on multiples of 2^-53 in [0, 1)
                Q = (long long)fl + (u < yv - fl ? 1 : 0);
            } else {
                Q = llrint(yv);
            }
        }
        const long c = bk + r;
        const long c7 = c & 127, cx = c7 & 15;
        const long crow = perm128 ? (c & ~127L) | (((c7 >> 4) & 3) << 5) | ((cx >> 2) << 3) | (cx & 3) | ((c7 >> 6) << 2) : c;
        for (int s = nslices - 1; s >= 0; s--) {
            long long d = ((Q + 128) & 255) - 128;
            Q = (Q - d) >> 8;
            Bs[(long)s * np * np + crow * np + bj + tx] = (int8_t)d;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Marker re-centring.  The truncation of W to S digits costs every vara_i at most (sum_j |m_ij|)^2 times the last-digit
// weight -- harmless next to the diagonal term of a marker with many non-zero genotypes, but a rare-variant marker is
// almost the constant vector c (c = -1 in the g-1 coding), W annihilates constants when the model has an intercept, and
// its true vara is far below its diagonal term.  So the digit kernel is fed m' = m - c_i 1 (c_i = the marker's majority
// genotype, over the n real individuals only): rare variants become sparse rows, monomorphic ones vanish, and
//   sum_{j<k} m_j m_k Wu_jk  =  sum_{j<k} m'_j m'_k Wu_jk  +  c_i (m^T rho)  -  c_i^2 R,
//   rho_j = sum_{k>j} Wu_jk + sum_{k<j} Wu_kj,   R = sum_{j<k} Wu_jk = (1/2) sum_j rho_j,
// with the two correction terms in fp64 (m^T rho is one more column of the genotype pass).  The digit error of marker i
// is then bounded by (sum_j |m'_ij|)^2 / 2 * 2^(e+1-8S): it shrinks with the marker's own diagonal term.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_marker_shift(const int8_t* __restrict__ Mt8, long L_pad, int n, long ld, int8_t* __restrict__ Mt8s,
                                                      int8_t* __restrict__ cshift, int32_t* __restrict__ l1norm) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per marker
    if (row >= L_pad) return;
    const int8_t* src = Mt8 + row * ld;
    int neg = 0, pos = 0;
    for (int j = lane * 16; j < n; j += 64 * 16) {
        const i32x4 x = *(const i32x4*)(src + j);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned u = (unsigned)x[q];
            // bytes are 0x00, 0x01, 0xFF; bytes at j >= n inside the last 16-byte group are padding zeros
            neg += __builtin_popcount(u & 0x80808080u);
            pos += __builtin_popcount(u & ~(u >> 7) & 0x01010101u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { neg += __shfl_xor(neg, o); pos += __shfl_xor(pos, o); }
    const int zer = n - neg - pos;
    int c = 0;  // the commonest genotype; ties: any fixed rule will do
    if (neg > zer && neg >= pos) c = -1;
    else if (pos > zer && pos > neg) c = 1;
    // heterozygote majority (a common variant: its digit bound has room to spare): centre on the commoner homozygote instead, so
    // that the stored row has no negative entries at all -- the matrix unit draws less power on such an operand (tools/ubench/
    // mfma_ceiling.hip: +2 %; measured on the kernel: 141.0 -> 138.0 ms).  A marker with heterozygotes only keeps c = 0: its row is 0.
    else if (neg | pos) c = neg >= pos ? -1 : 1;
    if (lane == 0) {
        cshift[row] = (int8_t)c;
        // {sum_j |m'_ij|, sum_j m'_ij^2} of the re-centred marker: the per-marker digit error bound is l1^2 / 2 * 2^(e+1-8S)
        // (round to nearest) or 8.355 q2 2^(e+1-8S) (stochastic rounding); k_cert_*
        if (l1norm) {
            l1norm[2 * row] = c == 0 ? neg + pos : (c < 0 ? zer + 2 * pos : zer + 2 * neg);
            l1norm[2 * row + 1] = c == 0 ? neg + pos : (c < 0 ? zer + 4 * pos : zer + 4 * neg);
        }
    }
    // The digit kernel's term is a quadratic form in the re-centred row, so the row may be stored with either sign: the one with
    // fewer negative entries (none at all when a homozygote is the majority) -- the matrix unit draws measurably less power on an
    // operand of small non-negative bytes than on one with 0xFF / 0xFE bytes in it (tools/ubench/mfma_ceiling.hip: +3 %).
    const bool flip = c > 0 || (c == 0 && neg > pos);
    int8_t* dst = Mt8s + row * ld;
    for (int j = lane * 16; j < (int)ld; j += 64 * 16) {
        i32x4 x = {0, 0, 0, 0};
        if (j < n) {
            x = *(const i32x4*)(src + j);
            if (c != 0 || flip) {
                union { i32x4 v; int8_t b[16]; } u;
                u.v = x;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int v = u.b[q] - c;
                    u.b[q] = (j + q < n) ? (int8_t)(flip ? -v : v) : (int8_t)0;
                }
                x = u.v;
            }
        }
        *(i32x4*)(dst + j) = x;
    }
}
extern "C" int eagle_dev_marker_shift(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n, long n_pad, long ld, int8_t* Mt8s,
                                      int8_t* cshift, int32_t* l1norm, void* stream) {
    if (ld % 16 || n > n_pad || n_pad > ld || n <= 0) return eagle_fail(ctx, EAGLE_ERR_ARG, "marker_shift: layout contract violated");
    if (L_pad <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_marker_shift, dim3((unsigned)((L_pad + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Mt8, L_pad, (int)n, ld, Mt8s, cshift, l1norm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_marker_shift");
    return EAGLE_OK;
}

// rowpart[j] = sum_{k>j} Wu[j][k] (one block per row, fixed tree order)
__global__ __launch_bounds__(256) void k_rho_rows(const double* __restrict__ Wu, long np, double* __restrict__ rho) {
    const long j = blockIdx.x;
    double s = 0.0;
    for (long k = j + 1 + threadIdx.x; k < np; k += 256) s += Wu[j * np + k];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) rho[j] = red[0];
}
// colpart[by][j] = sum over the 256 rows k of chunk `by` with k < j of Wu[k][j] (coalesced across the columns of a block)
__global__ __launch_bounds__(256) void k_rho_cols(const double* __restrict__ Wu, long np, double* __restrict__ colpart) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x, k0 = (long)blockIdx.y * 256;
    double s = 0.0;
    if (k0 < j)
        for (long k = k0; k < k0 + 256 && k < j; k++) s += Wu[k * np + j];
    colpart[(long)blockIdx.y * np + j] = s;
}
// rho[j] += sum_by colpart[by][j] (every column on its own, chunks in order), then R = (1/2) sum_j rho_j in one block and a fixed order:
// thread t of 1,024 adds the columns j = t, t + 1024, ... in order, then a tree over the threads.  (Two kernels since round 3: the
// column sums used to run inside the one block as well, 0.16 ms for 3.4 MB; the values are the same bits.)
__global__ __launch_bounds__(256) void k_rho_colsum(const double* __restrict__ colpart, long np, double* __restrict__ rho) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
    const int nchunk = (int)(np / 256);
    double s = rho[j];
    for (int b = 0; b < nchunk; b++) s += colpart[(long)b * np + j];
    rho[j] = s;
}
__global__ __launch_bounds__(1024) void k_rho_final(long np, const double* __restrict__ rho, VaraHdr* __restrict__ hdr) {
    double tot = 0.0;
    for (long j = threadIdx.x; j < np; j += 1024) tot += rho[j];
    __shared__ double red[1024];
    red[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) hdr->R = 0.5 * red[0];
}

struct VaraIt {  // position in a worker's flattened stage sequence: pair p, half (tile p, then tile nct-1-p), stage kt
    int p, half, ct, kt, nk;
    bool valid;
};
__device__ __forceinline__ void vit_set_tile(VaraIt& it, int nct, int npair) {
    while (it.p < npair) {
        int ct = it.half == 0 ? it.p : nct - 1 - it.p;
        if (it.half == 1 && ct == it.p) { it.p++; it.half = 0; continue; }  // middle tile of an odd nct: once only
        it.ct = ct;
        it.kt = 0;
        it.nk = (T8 / BK8) * (ct + 1);
        it.valid = true;
        return;
    }
    it.valid = false;
}
__device__ __forceinline__ void vit_advance(VaraIt& it, int nct, int npair) {
    if (++it.kt < it.nk) return;
    if (it.half == 0) it.half = 1; else { it.half = 0; it.p++; }
    vit_set_tile(it, nct, npair);
}

__global__ __launch_bounds__(512, 2) void k_vara_i8(const int8_t* __restrict__ Mt8, long ld, int ntm, const int8_t* __restrict__ Bs,
                                                    long np, const VaraHdr* __restrict__ hdr, long long* __restrict__ q, long Lp) {
    // LDS: genotype (A) stages in a ring of THREE, W-digit (B) stages in a ring of two = 160 KiB.  The third A buffer is
    // what lets the tile epilogue read its genotype bytes from LDS: a tile's 256 output columns are exactly the genotype
    // columns of its last two K stages, and with three buffers both are still there when the last MFMA has retired.
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];
    int8_t* const ldsA0 = ldsv;
    int8_t* const ldsB0 = ldsv + 3 * TILE_BYTES;
    // XCD-aware placement (speed only): b -> (xcd, slot); slot -> (marker tile within the XCD's sequence, worker)
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int nslices = hdr->S;  // the grid is sized for the largest slice count; surplus workgroups leave here
    const int mt = (slot / nslices) * 8 + xcd, sl = slot % nslices;  // worker = digit slice
    if (mt >= ntm) return;
    const int nct = (int)(np / T8), npair = (nct + 1) / 2;
    const int8_t* Bsl = Bs + (long)sl * np * np;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int8_t* Ablk = Mt8 + (long)mt * T8 * ld;
    const int ldi = (int)ld, npi = (int)np;
    const T8Lane lnA = t8_lane(lane, ldi), lnB = t8_lane(lane, npi);
    const __amdgpu_buffer_rsrc_t rsA = t8_rsrc(Ablk, ldi);

    VaraIt cur, nxt;
    cur.p = 0; cur.half = 0; vit_set_tile(cur, nct, npair);
    if (!cur.valid) return;
    nxt = cur;
    i32x16 acc[4][2];
    t8_zero(acc);
    // Tile epilogue: per 32-row MFMA tile m a reduce-scatter butterfly over lane bits 0..3 leaves lane l with the
    // sum (over 16 of the 32 column lanes) of register x = xsel, one more exchange over bit 4 completes it.  keep[m]
    // accumulates those row sums across the tiles of a slice (int64) and is flushed once per slice.
    long long keep[4] = {0, 0, 0, 0};
    const int col = lane & 31, hrow = 4 * (lane >> 5);
    const int xsel = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);

    // epilogue addressing: lane (col = l&31, h = l>>5) of wave (wr, wc) wants byte (row, colT) of a 128-column A stage,
    // row = wr*128 + m*32 + rx + 4h (rx = (x&3) + 8(x>>2)), colT = (wc&1)*64 + col (+32 for the second MFMA tile); the
    // stage image stores 16-byte chunk c of a row at chunk c ^ ((row>>1)&7), and ((rx + 4h)>>1)&7 = base_x ^ 2h with
    // base_x in {0,1,4,5} (disjoint bits), so the byte sits at rowpart + (laneP ^ (base_x << 4)): four per-lane address
    // registers per column and compile-time offsets cover all 128 reads.
    const int colT = (wc & 1) * 64 + col;
    const int laneP = (wr * 128 + hrow) * BK8 + (((colT >> 4) ^ (2 * (lane >> 5))) << 4) + (colT & 15);
    int pe0[4], pe1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int bx = ((k & 1) | ((k & 2) << 1)) << 4;  // base_x << 4 for base_x = 0, 1, 4, 5
        pe0[k] = laneP ^ bx;
        pe1[k] = pe0[k] ^ 32;                              // second column tile: chunk + 2
    }

    t8_stage(rsA, lnA, ldi, nxt.kt * BK8, ldsA0, w);
    t8_stage(t8_rsrc(Bsl + (long)nxt.ct * T8 * np, npi), lnB, npi, nxt.kt * BK8, ldsB0, w);
    vit_advance(nxt, nct, npair);
    __syncthreads();
    int ia = 0, ib = 0;  // ring positions of the stage being computed
    while (cur.valid) {
        const int ian = ia == 2 ? 0 : ia + 1;
        if (nxt.valid) {
            t8_stage(rsA, lnA, ldi, nxt.kt * BK8, ldsA0 + ian * TILE_BYTES, w);
            t8_stage(t8_rsrc(Bsl + (long)nxt.ct * T8 * np, npi), lnB, npi, nxt.kt * BK8, ldsB0 + (ib ^ 1) * TILE_BYTES, w);
            vit_advance(nxt, nct, npair);
        }
        t8_compute(acc, ldsA0 + ia * TILE_BYTES, ldsB0 + ib * TILE_BYTES, wr, wc, lane);
        if (cur.kt == cur.nk - 1) {
            // tile done: v[x] = sum over this lane's 2 columns of T[row][col] * m[row][col]; the genotype bytes of columns
            // 0..127 of the tile are the A stage before this one (waves wc = 0, 1), those of 128..255 this one (wc = 2, 3)
            const int8_t* mst = ldsA0 + ((wc & 2) ? ia : (ia == 0 ? 2 : ia - 1)) * TILE_BYTES;
            const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int v16[16], v8[8], v4[4], v2[2];
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const int rx = (x & 3) + 8 * (x >> 2);
                    const int k = ((rx >> 1) & 1) | (((rx >> 1) & 4) >> 1);  // index of base_x = (rx>>1)&7 in {0,1,4,5}
                    const int off = (m * 32 + rx) * BK8;
                    const int m0 = (int)mst[pe0[k] + off];
                    const int m1 = (int)mst[pe1[k] + off];
                    v16[x] = acc[m][0][x] * m0 + acc[m][1][x] * m1;
                    acc[m][0][x] = 0;
                    acc[m][1][x] = 0;
                }
#pragma unroll
                for (int i = 0; i < 8; i++) { int snd = b0 ? v16[i] : v16[i + 8]; int kp = b0 ? v16[i + 8] : v16[i]; v8[i] = kp + __shfl_xor(snd, 1); }
#pragma unroll
                for (int i = 0; i < 4; i++) { int snd = b1 ? v8[i] : v8[i + 4]; int kp = b1 ? v8[i + 4] : v8[i]; v4[i] = kp + __shfl_xor(snd, 2); }
#pragma unroll
                for (int i = 0; i < 2; i++) { int snd = b2 ? v4[i] : v4[i + 2]; int kp = b2 ? v4[i + 2] : v4[i]; v2[i] = kp + __shfl_xor(snd, 4); }
                int v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
                v1 += __shfl_xor(v1, 16);
                keep[m] += v1;
                __builtin_amdgcn_sched_barrier(0);  // keep the 4 m-tiles' epilogues apart: bounds live registers
            }
            vit_advance(cur, nct, npair);  // kt == nk-1: moves to the next tile (or ends)
        } else {
            cur.kt++;
        }
        __syncthreads();
        ia = ian;
        ib ^= 1;
    }
    // this worker's slice is complete: add into q[slice][row] (int64 atomics: exact, order independent)
    long long* qs = q + (long)sl * Lp + (long)mt * T8 + wr * 128 + hrow + (xsel & 3) + 8 * (xsel >> 2);
#pragma unroll
    for (int m = 0; m < 4; m++)
        if ((lane & 16) == 0 && keep[m]) atomicAdd((unsigned long long*)&qs[m * 32], (unsigned long long)keep[m]);
}

// ------------------------------------------------------------------------------------------------
// k_vara_i8w: the same computation on a 384 (markers) x 256 (columns) workgroup tile.
// The 256 x 256 form reads 6 fragments from LDS per 8 MFMAs and fills 64 KiB of LDS per 32 MFMAs per wave; at 55 % MFMA busy
// the LDS port (ds_read_b128 + LDS-DMA writes: 256 KiB per stage against 128 B/clk) is as loaded as the matrix pipe.  Here the
// 8 waves sit 4 (M) x 2 (N) with a 96 x 128 wave tile = 3 x 4 MFMA tiles (192 accumulator registers): 7 fragment reads per
// 12 MFMAs (-22 % LDS reads per MAC) and 80 KiB of fill per 48 MFMAs per wave (-17 % L2 -> LDS bytes per MAC).  Stages are
// double buffered (2 x (48 + 32) KiB = all 160 KiB), so the tile epilogue takes its genotype bytes from global memory (L2).
// Same integer sums as k_vara_i8: bit-identical q.  int32 butterfly over 128 columns per wave: n_pad < 32768 (k_vara_i8p<2> lifts that).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_vara_i8w(const int8_t* __restrict__ Mt8, long ld, int ntm, const int8_t* __restrict__ Bs,
                                                     long np, const VaraHdr* __restrict__ hdr, long long* __restrict__ q, long Lp, int cut_last_round) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];  // [2][A 48 KiB | B 32 KiB]
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int nslices = hdr->S;
    const int nct = (int)(np / T8), npair = (nct + 1) / 2;
    // Work per XCD: its marker tiles (mt = 8 g + xcd) x the S slices = `wx` workers on 32 CUs.  The workers of the last, partly
    // filled round are cut into `psplit` pieces along their column-tile pairs (vara_tail_pieces), so that the round costs about
    // tail/32 of a worker-time instead of a whole one (a 125,000-marker shard -- the headline problem on 8 GPUs -- is 5.1 rounds:
    // 6 worker-times without the cut, 5.2 with it).  Integer atomics into q: the result does not change.
    const int groups = (ntm + 7) >> 3, wx = groups * nslices, full = (wx >> 5) << 5, tail = wx - full;
    const int psplit_tail = cut_last_round ? vara_tail_pieces(tail, npair) : 1;
    int psplit = psplit_tail;
    int worker = slot, piece = 0;
    if (slot >= full) { const int u = slot - full; worker = full + u / psplit; piece = u - (u / psplit) * psplit; }
    else psplit = 1;  // only the workers of the last round are cut
    if (worker >= wx) return;
    const int mt = (worker / nslices) * 8 + xcd, sl = worker - (worker / nslices) * nslices;
    if (mt >= ntm) return;
    const int pair0 = npair * piece / psplit, pair1 = npair * (piece + 1) / psplit;
    const int8_t* Bsl = Bs + (long)sl * np * np;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;   // 4 x 2 waves
    const int ldi = (int)ld, npi = (int)np;
    const T8Lane lnA = t8_lane(lane, ldi), lnB = t8_lane(lane, npi);
    const long row0 = (long)mt * TW_M;
    const long rows_here = Lp - row0 < TW_M ? Lp - row0 : TW_M;  // the last marker tile may be short: rows beyond read as zero
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(Mt8 + row0 * ld), 0, (int)(rows_here * ld), 0x00020000);

    VaraIt cur, nxt;
    cur.p = pair0; cur.half = 0; vit_set_tile(cur, nct, pair1);
    if (!cur.valid) return;
    nxt = cur;
    i32x16 acc[3][4];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) acc[m][n][x] = 0;
    long long keep[3] = {0, 0, 0};
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    const int offA = wr * (96 * BK8) + r * BK8, offB = TW_ABYTES + wc * (128 * BK8) + r * BK8;
    int ch[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) ch[ks] = ((2 * ks + h) ^ swz) << 4;
    const int xsel = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
    // epilogue: genotype byte (row = wr*96 + m*32 + rx + 4h, column = ct*256 + wc*128 + n*32 + r) through the tile's buffer
    const int evoff = (wr * 96 + 4 * h) * ldi + wc * 128 + r;

    tw_stage<6>(rsA, lnA, ldi, nxt.kt * BK8, ldsv, w);
    tw_stage<4>(t8_rsrc(Bsl + (long)nxt.ct * T8 * np, npi), lnB, npi, nxt.kt * BK8, ldsv + TW_ABYTES, w);
    vit_advance(nxt, nct, pair1);
    __syncthreads();
    int buf = 0;
    while (cur.valid) {
        const int8_t* st = ldsv + buf * (TW_ABYTES + TILE_BYTES);
        if (nxt.valid) {
            int8_t* nx = ldsv + (buf ^ 1) * (TW_ABYTES + TILE_BYTES);
            tw_stage<6>(rsA, lnA, ldi, nxt.kt * BK8, nx, w);
            tw_stage<4>(t8_rsrc(Bsl + (long)nxt.ct * T8 * np, npi), lnB, npi, nxt.kt * BK8, nx + TW_ABYTES, w);
            vit_advance(nxt, nct, pair1);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) tw_kstep(acc, st + offA, st + offB, ch[ks]);
        if (cur.kt == cur.nk - 1) {
            const int ecol = cur.ct * T8;
            const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                int v16[16], v8[8], v4[4], v2[2];
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const int so = (m * 32 + (x & 3) + 8 * (x >> 2)) * ldi + ecol;
                    int sacc = 0;
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        const int g = (int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rsA, evoff, so + n * 32, 0);
                        sacc += acc[m][n][x] * g;
                        acc[m][n][x] = 0;
                    }
                    v16[x] = sacc;
                }
#pragma unroll
                for (int i = 0; i < 8; i++) { int snd = b0 ? v16[i] : v16[i + 8]; int kp = b0 ? v16[i + 8] : v16[i]; v8[i] = kp + __shfl_xor(snd, 1); }
#pragma unroll
                for (int i = 0; i < 4; i++) { int snd = b1 ? v8[i] : v8[i + 4]; int kp = b1 ? v8[i + 4] : v8[i]; v4[i] = kp + __shfl_xor(snd, 2); }
#pragma unroll
                for (int i = 0; i < 2; i++) { int snd = b2 ? v4[i] : v4[i + 2]; int kp = b2 ? v4[i + 2] : v4[i]; v2[i] = kp + __shfl_xor(snd, 4); }
                int v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
                v1 += __shfl_xor(v1, 16);
                keep[m] += v1;
                __builtin_amdgcn_sched_barrier(0);
            }
            vit_advance(cur, nct, pair1);
        } else {
            cur.kt++;
        }
        __syncthreads();
        buf ^= 1;
    }
    const long qrow = row0 + wr * 96 + 4 * h + (xsel & 3) + 8 * (xsel >> 2);
    long long* qs = q + (long)sl * Lp + qrow;
#pragma unroll
    for (int m = 0; m < 3; m++)
        if ((lane & 16) == 0 && keep[m] && qrow + m * 32 < Lp) atomicAdd((unsigned long long*)&qs[m * 32], (unsigned long long)keep[m]);
}

// ------------------------------------------------------------------------------------------------
// k_vara_i8p: k_vara_i8w (384 x 256 tile, 8 waves 4 x 2, wave tile 96 x 128 = 3 x 4 MFMA tiles) with the k-step written as
// inline asm so that LDS reads, LDS-DMA loads and MFMAs overlap inside ONE wave.  hipcc's schedule of the k-step is "7 fragment
// reads, wait, 12 MFMAs"; the two waves of a SIMD advance in step (they alternate on the matrix pipe), so both sit in the
// read phase together: 56 KiB of fragment reads per k-step = 450 cycles of the LDS pipe against 770 cycles of MFMA, unhidden.
// Here a k-step issues, between its own MFMAs, the fragment reads of the NEXT k-step:
//   * MFMA order column-major (column n = W-digit fragment b_n): b_n is re-loaded in place when its column is done, 9 MFMAs
//     before its next use; the genotype fragments are double-buffered (a / an, +12 VGPRs), loaded 11 MFMAs ahead;
//   * LDS returns in order, so the waits are counted (issue order per k-step: an0 an1 b0 an2 b1 b2 b3).  An MFMA reads its A/B
//     sources as it issues and LDS data comes back tens of cycles later, so re-loading a source register right behind the last
//     MFMA that reads it is safe;
//   * the next stage's ten LDS-DMA loads per wave go out one at a time behind every second MFMA (3 behind the barrier, 6 in
//     the next k-step, the last in the one after) instead of as a burst at the top of the stage (the burst alone costs 3-4 %, tune 11-15 experiments);
//   * the first k-step of a tile uses the inline constant 0 as the C operand: no zeroing pass;
//   * the stage barrier sits in the middle of the last k-step (after 6 of its 12 MFMAs: all fragment reads of this stage have
//     returned; the re-loads behind the barrier read the other buffer), so each wave reaches it with matrix work in flight;
//   * the W digits are SrcA and the genotypes SrcB (below): transposed result tiles, lane = marker, register = W column; with
//     the digit slices' columns permuted inside blocks of 128 (k_slice_w) a lane's 64 columns are 64 consecutive genotype bytes
//     of its own marker row, and the tile epilogue is four 16-byte loads and 64 multiply-adds per lane, no cross-lane reduction.
// Every asm block that advances DMA addresses with s_add_u32 declares the "scc" clobber (hipcc keeps compare results in SCC
// across asm statements that do not).
// ------------------------------------------------------------------------------------------------
// (the tile row-dot of a lane sums 64 products in int32: |sum| <= 32768 K for a K-deep tile, i.e. n_pad < 65536)
// PACE (experiment of round 3, tools/bench_vara.py tune 14): the 32 workers an XCD runs at a time meet at a soft
// barrier before every column-tile pair -- one agent-scope counter per (XCD, round, pair), bounded spin (a worker that is not
// co-resident with its round only costs the others the spin limit, never a hang) -- so that they stream the shared genotype and
// W-digit stages in step and the L2 serves them once.  Same integer sums: bit-identical q.
// EXT: the same kernel under a second name for the run on the compact image of eagle_dev_vara_i8_extend -- a launch that is dropped
// on the device whenever nobody qualifies, and that a profile should not average into the scan's launches.
template <bool PACE, bool EXT = false>
__global__ __launch_bounds__(512, 2) void k_vara_i8p(const int8_t* __restrict__ Mt8, long ld, int ntm, const int8_t* __restrict__ Bs,
                                                     long np, const VaraHdr* __restrict__ hdr, long long* __restrict__ q, long Lp, int cut_last_round,
                                                     int* __restrict__ pace) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];  // [2][A 48 KiB | B 32 KiB]
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int nslices = hdr->S;
    const int nct = (int)(np / T8), npair = (nct + 1) / 2;
    const int groups = (ntm + 7) >> 3, wx = groups * nslices, full = (wx >> 5) << 5, tail = wx - full;  // see k_vara_i8w
    const int psplit_tail = cut_last_round ? vara_tail_pieces(tail, npair) : 1;
    int psplit = psplit_tail;
    int worker = slot, piece = 0;
    if (slot >= full) { const int u = slot - full; worker = full + u / psplit; piece = u - (u / psplit) * psplit; }
    else psplit = 1;
    if (worker >= wx) return;
    const int mt = (worker / nslices) * 8 + xcd, sl = worker - (worker / nslices) * nslices;
    if (mt >= ntm) return;
    const int pair0 = npair * piece / psplit, pair1 = npair * (piece + 1) / psplit;
    const int8_t* Bsl = Bs + (long)sl * np * np;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;   // 4 x 2 waves
    const int ldi = (int)ld, npi = (int)np;
    const T8Lane lnA = t8_lane(lane, ldi), lnB = t8_lane(lane, npi);
    const long row0 = (long)mt * TW_M;
    const long rows_here = Lp - row0 < TW_M ? Lp - row0 : TW_M;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(Mt8 + row0 * ld), 0, (int)(rows_here * ld), 0x00020000);

    VaraIt cur, nxt;
    cur.p = pair0; cur.half = 0; vit_set_tile(cur, nct, pair1);
    if (!cur.valid) return;
    nxt = cur;
    int pace_expected = 0;
    int* pace_row = nullptr;
    if (PACE && slot < full) {  // whole workers only; the cut last round runs unpaced
        const int round = slot >> 5;
        for (int u = 0; u < 32; u++) pace_expected += (((round * 32 + u) / nslices) * 8 + xcd) < ntm;
        pace_row = pace + ((long)xcd * ((wx + 31) >> 5) + round) * npair;
    }
    long long keep[3] = {0, 0, 0};
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    constexpr int STG = TW_ABYTES + TILE_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) int8_t*)ldsv;
    const unsigned offA = lds0 + wr * (96 * BK8) + r * BK8, offB = lds0 + TW_ABYTES + wc * (128 * BK8) + r * BK8;
    unsigned ch[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) ch[ks] = ((2 * ks + h) ^ swz) << 4;
    const int evoff = (wr * 96 + r) * ldi + wc * 128 + h * 64;  // epilogue: this lane's marker row and its 64 genotype columns (k_slice_w, perm128)
    // stage 0 as a burst, then the pipeline: eight waves issue the DMA of a stage, A 48 row groups (6 per wave), B 32 (4 per wave)
    tw_stage<6>(rsA, lnA, ldi, nxt.kt * BK8, ldsv, w);
    tw_stage<4>(t8_rsrc(Bsl + (long)nxt.ct * T8 * np, npi), lnB, npi, nxt.kt * BK8, ldsv + TW_ABYTES, w);
    vit_advance(nxt, nct, pair1);
    __syncthreads();
    XDma dA, dB;
    dA.st = 8 * ldi; dA.vE = lnA.voffE; dA.vO = lnA.voffO;
    dB.st = 8 * npi; dB.vE = lnB.voffE; dB.vO = lnB.voffO;
    const i32x4 rsA_raw = x_rsrc(Mt8 + row0 * ld, (unsigned)(rows_here * ld)), rs_none = x_rsrc(Mt8, 0);
    auto dma_arm = [&](int into) {  // the sequence of the next stage to fetch, into buffer `into`
        const bool on = nxt.valid;
        const unsigned base = lds0 + into * STG;
        dA.m0 = base + (w * 6) * 1024 - 1024;
        dB.m0 = base + TW_ABYTES + (w * 4) * 1024 - 1024;
        dA.so = (unsigned)((w * 6) * 8 * ldi + nxt.kt * BK8 - 8 * ldi);
        dB.so = (unsigned)((w * 4) * 8 * npi + nxt.kt * BK8 - 8 * npi);
        dA.rs = on ? rsA_raw : rs_none;
        dB.rs = on ? x_rsrc(Bsl + (long)nxt.ct * T8 * np, (unsigned)(T8 * npi)) : rs_none;
        if (on) vit_advance(nxt, nct, pair1);
    };
    dma_arm(1);
    tx_dma3(dA);
    int buf = 0;
    i32x4 fa[2][3], fb[4];
    TxAcc c;
    tx_prologue(fa[0], fb, offA + ch[0], offB + ch[0]);
    // the stage's k-steps 1-3; the DMA sequence of the stage after next is armed before the last one, which carries the barrier
    auto stage_rest = [&] {
        const unsigned sa = offA + buf * STG, sb = offB + buf * STG;
        tx_kstep<false, 2>(c, fa[1], fa[0], fb, sa + ch[2], sb + ch[2], dA, dB);
        tx_kstep<false, 0>(c, fa[0], fa[1], fb, sa + ch[3], sb + ch[3], dA, dB);
        dma_arm(buf);
        buf ^= 1;
        tx_klast(c, fa[1], fa[0], fb, offA + buf * STG + ch[0], offB + buf * STG + ch[0], dA);
    };
    while (cur.valid) {
        const int done_ct = cur.ct, nk = cur.nk;  // nk >= 2
        if (PACE && pace_row && cur.half == 0) {
            if (t == 0) {
                int* cnt = pace_row + cur.p;
                (void)__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int it = 0; it < 2000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pace_expected; it++)
                    __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
        tx_kstep<true, 1>(c, fa[0], fa[1], fb, offA + buf * STG + ch[1], offB + buf * STG + ch[1], dA, dB);
        stage_rest();
        for (int kt = 1; kt < nk; kt++) {
            tx_kstep<false, 1>(c, fa[0], fa[1], fb, offA + buf * STG + ch[1], offB + buf * STG + ch[1], dA, dB);
            stage_rest();
        }
        cur.kt = nk - 1;
        vit_advance(cur, nct, pair1);
        {
            // the MFMAs above are opaque to the compiler's hazard recogniser: let the last ones retire before the accumulators are
            // read; and have the re-loads landed, in case the fragments are dead from here (last tile)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            const int ecol = done_ct * T8;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                // lane (r, h) holds, for its marker row wr*96 + m*32 + r, the sums over K against the 64 W columns 64 h + 16 n + x of
                // the wave's 128 (k_slice_w, perm128): their genotype bytes are 64 consecutive bytes of that row
                i32x4 g[4];
#pragma unroll
                for (int n = 0; n < 4; n++) g[n] = __builtin_amdgcn_raw_buffer_load_b128(rsA, evoff + m * 32 * ldi, ecol + n * 16, 0);
                int sacc = 0;
#pragma unroll
                for (int n = 0; n < 4; n++)
#pragma unroll
                    for (int x = 0; x < 16; x++) sacc += c[m][n][x] * ((g[n][x >> 2] << (24 - 8 * (x & 3))) >> 24);
                keep[m] += sacc;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long qrow = row0 + wr * 96 + r;  // both halves h of a marker's lane pair add their part
    long long* qs = q + (long)sl * Lp + qrow;
#pragma unroll
    for (int m = 0; m < 3; m++)
        if (keep[m] && qrow + m * 32 < Lp) atomicAdd((unsigned long long*)&qs[m * 32], (unsigned long long)keep[m]);
}

// ------------------------------------------------------------------------------------------------
// k_zbuild_i8: Z = Mt8 * U for the spectral scan (eagle_spectral.hip) from exact int8 digit slices of U, on the tile engine of
// k_vara_i8w (384 x 256 tile, 8 waves 4 x 2, 3 x 4 MFMA tiles per wave) with a dense K loop and a store epilogue:
//   U = 2^(e+2) sum_{s<S} D_s 256^-(s+1) + R,  max|U| < 2^e,  |R_jk| <= 2^(e+1-8S);   Us[s][k][j] = D_s[j][k]  (k = output column)
//   T_s[i][k] = sum_j Mt8[i][j] D_s[j][k]  exactly (int32: |T| <= 128 n_pad);   Z[i][k] = 2^(e+2) sum_s 256^-(s+1) T_s[i][k],
// the slices walked least significant first inside one workgroup, each added into its tile of Z (owned by that workgroup alone).
// |Z_ik - (Mt U)_ik| <= (sum_j |m_ij|) 2^(e+1-8S) + S-term fp64 rounding.  Against the fp64 MFMA form (k_zbuild) this is S int8
// products instead of one fp64 product: 64x the matrix rate per product.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_absmax_all(const double* __restrict__ x, long count, unsigned long long* __restrict__ bits) {
    double m = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        const double v = fabs(x[i]);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double y = __shfl_down(m, o); m = y > m ? y : m; }
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(bits, (unsigned long long)__double_as_longlong(m));
}
// Us[s][k][j] = digit s of Ur[j][k]; 32 x 32 tiles through LDS so that both sides are coalesced
__global__ __launch_bounds__(256) void k_slice_u(const double* __restrict__ Ur, long np, const unsigned long long* __restrict__ maxbits, int nslices,
                                                 int8_t* __restrict__ Us) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long bj = (long)blockIdx.y * 32, bk = (long)blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = Ur[(bj + r) * np + bk + tx];  // tile[jj][kk]
    __syncthreads();
    int e = 0;
    const double mx = __longlong_as_double((long long)*maxbits);
    if (mx > 0.0) (void)frexp(mx, &e);
    for (int r = ty; r < 32; r += 8) {
        long long Q = llrint(ldexp(tile[tx][r], 8 * nslices - (e + 2)));   // element (k = bk + r, j = bj + tx)
        for (int s = nslices - 1; s >= 0; s--) {
            const long long d = ((Q + 128) & 255) - 128;
            Q = (Q - d) >> 8;
            Us[(long)s * np * np + (bk + r) * np + bj + tx] = (int8_t)d;
        }
    }
}
__global__ __launch_bounds__(512, 2) void k_zbuild_i8(const int8_t* __restrict__ Mt8, long ld, int ntm, const int8_t* __restrict__ Us, long np,
                                                      int nslices, const unsigned long long* __restrict__ maxbits, double* __restrict__ Z, long Lp) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];  // [2][A 48 KiB | B 32 KiB]
    // XCD-aware placement (speed only): an XCD's 32 CUs work on 4 marker tiles x 8 column tiles in lock-step along K, so a
    // stage of either operand is fetched into that L2 once and read 8 / 4 times from there
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nct = (int)(np / T8), ncg = (nct + 7) >> 3;
    const int within = slot & 31, grp = slot >> 5;
    const int mt = (((grp / ncg) * 4 + (within >> 3)) << 3) + xcd, ct = (grp % ncg) * 8 + (within & 7);
    if (mt >= ntm || ct >= nct) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int ldi = (int)ld, npi = (int)np;
    const T8Lane lnA = t8_lane(lane, ldi), lnB = t8_lane(lane, npi);
    const long row0 = (long)mt * TW_M;
    const long rows_here = Lp - row0 < TW_M ? Lp - row0 : TW_M;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(Mt8 + row0 * ld), 0, (int)(rows_here * ld), 0x00020000);
    int e = 0;
    const double mx = __longlong_as_double((long long)*maxbits);
    if (mx > 0.0) (void)frexp(mx, &e);
    i32x16 acc[3][4];
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) acc[m][n][x] = 0;
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    const int offA = wr * (96 * BK8) + r * BK8, offB = TW_ABYTES + wc * (128 * BK8) + r * BK8;
    int ch[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) ch[ks] = ((2 * ks + h) ^ swz) << 4;
    // the workgroup's tile of Z: rows row0 .. row0 + rows_here, columns ct*256 .. +256 (byte offsets fit 32 bits: 384 rows x 8 np)
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)(Z + row0 * np + (long)ct * T8), 0,
                                                                         (int)(((rows_here - 1) * np + T8) * 8), 0x00020000);
    const int zvoff = ((wr * 96 + 4 * h) * npi + wc * 128 + r) * 8;
    const int nk = (int)(np / BK8), total = nslices * nk;  // flattened (slice, K stage) sequence, least significant slice first
    auto stage = [&](int qi, int8_t* dst) {
        const int s = nslices - 1 - qi / nk, kt = qi - (qi / nk) * nk;
        tw_stage<6>(rsA, lnA, ldi, kt * BK8, dst, w);
        tw_stage<4>(t8_rsrc(Us + (long)s * np * np + (long)ct * T8 * np, npi), lnB, npi, kt * BK8, dst + TW_ABYTES, w);
    };
    stage(0, ldsv);
    __syncthreads();
    int buf = 0;
    for (int qi = 0; qi < total; qi++) {
        const int8_t* st = ldsv + buf * (TW_ABYTES + TILE_BYTES);
        if (qi + 1 < total) stage(qi + 1, ldsv + (buf ^ 1) * (TW_ABYTES + TILE_BYTES));
#pragma unroll
        for (int ks = 0; ks < 4; ks++) tw_kstep(acc, st + offA, st + offB, ch[ks]);
        const int kt = qi - (qi / nk) * nk;
        if (kt == nk - 1) {  // slice done: Z tile (+)= 2^(e+2-8(s+1)) T_s, through the tile's buffer descriptor (rows beyond L_pad dropped)
            const int s = nslices - 1 - qi / nk;
            const double scale = ldexp(1.0, e + 2 - 8 * (s + 1));
            const bool first = s == nslices - 1;
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const int so = ((m * 32 + (x & 3) + 8 * (x >> 2)) * npi) * 8;
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        double v = scale * (double)acc[m][n][x];
                        if (!first) v += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsZ, zvoff, so + n * 256, 0));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsZ, zvoff, so + n * 256, 0);
                        acc[m][n][x] = 0;
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one accumulator row at a time: bounds the live registers
                }
        }
        __syncthreads();
        buf ^= 1;
    }
}
// Us: nslices * n_pad * n_pad bytes + 16.  nslices = 0: chosen so that n_pad * 2^(e+1-8S) <= 1e-10 (Z entries are O(1)).
extern "C" int64_t eagle_spectral_zbuild_i8_workspace_bytes(long n_pad, int nslices) {
    return (int64_t)((size_t)(nslices > 0 ? nslices : 7) * n_pad * n_pad + 256);
}
extern "C" int eagle_dev_spectral_zbuild_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Ur, double* Z, void* ws,
                                            int nslices, void* stream) {
    if (L_pad % T8 || n_pad % T8 || ld % 128 || n_pad > ld || nslices < 1 || nslices > 7 || (double)ld * TW_M >= 2147483648.0 ||
        128.0 * (double)n_pad >= 2147483648.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "spectral_zbuild_i8: layout contract violated (L_pad % 256, n_pad % 256, 1 <= nslices <= 7)");
    if (L_pad == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* maxbits = (unsigned long long*)ws;
    int8_t* Us = (int8_t*)ws + 256;
    hipError_t er = hipMemsetAsync(ws, 0, 256, s);
    if (er != hipSuccess) return eagle_fail_hip(ctx, er, "zbuild_i8 memset");
    hipLaunchKernelGGL(k_absmax_all, dim3(1024), dim3(256), 0, s, Ur, n_pad * n_pad, maxbits);
    hipLaunchKernelGGL(k_slice_u, dim3((unsigned)(n_pad / 32), (unsigned)(n_pad / 32)), dim3(256), 0, s, Ur, n_pad, maxbits, nslices, Us);
    if (!ctx->attr_zbuild_i8) {
        hipError_t ea = hipFuncSetAttribute((const void*)k_zbuild_i8, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TW_ABYTES + TILE_BYTES));
        if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_zbuild_i8)");
        ctx->attr_zbuild_i8 = true;
    }
    const int ntm = (int)((L_pad + TW_M - 1) / TW_M), nct = (int)(n_pad / T8), ncg = (nct + 7) / 8;
    const int mg = ((ntm + 7) / 8 + 3) / 4;  // groups of 4 marker tiles per XCD
    hipLaunchKernelGGL(k_zbuild_i8, dim3((unsigned)(8 * mg * ncg * 32)), dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, Mt8, ld, ntm, Us, n_pad, nslices, maxbits, Z,
                       L_pad);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return eagle_fail_hip(ctx, e2, "k_zbuild_i8");
    return EAGLE_OK;
}

// vara_i from the exact integer row-dots q_s[i] of `nslices` digit slices (smallest term first), the fp64 diagonal term and the
// re-centring corrections
__device__ __forceinline__ double vara_from_q(const long long* __restrict__ q, long Lp, long i, int nslices, const VaraHdr* __restrict__ hdr,
                                              const double* __restrict__ vdiag, const int8_t* __restrict__ cshift, const double* __restrict__ mrho) {
    const int e = hdr->e;
    double s = 0.0;
    for (int k = nslices - 1; k >= 0; k--) s += ldexp((double)q[(long)k * Lp + i], e + 2 - 8 * (k + 1));  // smallest first
    const double c = cshift ? (double)cshift[i] : 0.0;  // re-centred marker: off(m) = off(m') + c m^T rho - c^2 R
    if (c != 0.0) s += c * mrho[i] - c * c * hdr->R;
    return vdiag[i] + s;  // diagonal term sum_k m_ik^2 W_kk (fp64) + exact-integer off-diagonal term
}
__global__ __launch_bounds__(256) void k_vara_i8_finish(const long long* __restrict__ q, long Lp, const VaraHdr* __restrict__ hdr,
                                                        const double* __restrict__ vdiag, const int8_t* __restrict__ cshift,
                                                        const double* __restrict__ mrho, double* __restrict__ vara) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Lp) return;
    vara[i] = vara_from_q(q, Lp, i, hdr->S, hdr, vdiag, cshift, mrho);
}

// ------------------------------------------------------------------------------------------------
// Certification of a digit-slice scan (find_qtl.R:71-83 must select the marker the fp64 arithmetic selects).
//
// Every vara_i of k_vara_i8_finish differs from the exact m_i^T W m_i by at most
//     b_i = min(l1_i^2 / 2 * 2^(e+1-8S), specH q2_i [k_spectral_decide])  +  2^-48 (|diag_i| + |vara_i - diag_i| + 2|c_i m_i^T rho| + 2 c_i^2 |R|)  +  2^-50 sum_k |W_kk|,
// l1_i = sum_j |m'_ij| of the re-centred marker (k_marker_shift): the first term is the truncation of W to S digits
// (|sum_{j<k} m'_j m'_k R_jk| <= 2^(e+1-8S) ((sum|m'_j|)^2 - sum m'_j^2)/2), the others cover the fp64 roundings of the
// handful of terms the finish kernel adds (each an exactly evaluated integer sum rounded once).
// Markers are RE-EVALUATED with the fp64 kernel (k_vara_f64<SPLIT>, bitwise the value scan mode 0 gives that marker) when
//   (1) b_i > 1.8 budget |vara_i|                -- the digit result is not certified to 0.9 of the path's 1e-6 tolerance (default budget 5e-7)
//                                                   (a quadratic form that cancels against its diagonal term, or W with huge
//                                                   off-diagonal entries), or
//   (2) a_i^2 / (vara_i - b_i) >= LB (1 - 1e-9),  LB = max_j a_j^2 / (vara_j + b_j)  -- marker i cannot be excluded from being
//                                                   the arg-max: LB is a lower bound of the true maximum, the left side an upper
//                                                   bound of marker i's true tsq (vara_i - b_i <= 0: no bound, re-evaluated).
// After that the arg-max over [fp64 values of the candidates, digit values of everybody else] is the arg-max of the fp64
// scan: every marker that can win carries its fp64 value, every other marker is strictly below the winner in both.
// At most CERT_CAP markers are re-evaluated per call; if more qualify (degenerate operands) the whole block is redone by the
// fp64 kernel (device-side gate, no host round trip).
// ------------------------------------------------------------------------------------------------
#define CERT_CAP 2048
// Round 4, two-tier enforcement.  With the tight budget in force (1e-7) the per-marker threshold is 1.8e-7 |vara_i|.  Operands of a
// structured population leave thousands of markers whose quadratic form cancels against its diagonal term between the two thresholds:
// re-evaluating them all would overflow CERT_CAP and throw the whole scan back to fp64 for nothing (they are inside the path's tolerance
// under the default budget).  So the markers over the TIGHT threshold are counted over the whole scan first (every block, every device:
// the decision does not depend on how the markers were cut up); more than CERT_TIGHT_MAX of them and the certificate enforces the
// default budget (1.8 x 5e-7 = 0.9 of the tolerance, round 3's rule), else the tight one.  eagle_cert_info.over_tight reports the count.
#define CERT_TIGHT_MAX 512
struct CertHdr { unsigned long long lb_bits; int count; int overflow; int flagged; int tight; };  // = eagle_cert_info of the public header

struct CertCtx { double delta, absR, sumdiag, specH, flag_rel, flag_loose, wErr; int stochastic; };
__device__ __forceinline__ CertCtx cert_ctx(const VaraHdr* hdr) {
    CertCtx c;
    const double mx = hdr->maxabs_off;
    const int e = hdr->e;
    c.delta = mx > 0.0 ? ldexp(1.0, e + 1 - 8 * hdr->S) : 0.0;
    c.specH = hdr->specH;
    c.wErr = hdr->wErr;
    if (c.specH > 0.0) c.delta *= 1.0 + 0x1p-8;   // the leading S digits of an (S+1)-digit rounding: |R_jk| <= (128 + 1/2) u
    c.absR = fabs(hdr->R);
    c.flag_rel = VARA_FLAG_FACTOR * hdr->budget;
    c.flag_loose = VARA_FLAG_FACTOR * hdr->budget_loose;
    c.sumdiag = hdr->sumdiag;
    c.stochastic = hdr->pad;
    return c;
}
__device__ __forceinline__ double cert_bound(const CertCtx& cc, const int32_t* l1q2, long i, int c, double vdiag, double mrho, double vara,
                                             int ext = 0) {
    const double l = (double)l1q2[2 * i];
    if (ext && cc.specH > 0.0) {   // a marker that got the dropped digit back (eagle_dev_vara_i8_extend): all digits cut, rounding only
        double mag = fabs(vdiag) + fabs(vara - vdiag);
        if (c != 0) mag += 2.0 * (fabs(mrho) + cc.absR);
        return 0.5 * l * l * (cc.delta * (0x1p-8 / (1.0 + 0x1p-8))) + cc.wErr * (double)l1q2[2 * i + 1] + 0x1p-48 * mag + 0x1p-50 * cc.sumdiag;
    }
    // round to nearest: guaranteed; stochastic rounding: exceeded with probability below 1e-30 per marker (and never above
    // the guaranteed l1^2 * delta of an interval of twice the width)
    double b = cc.stochastic ? fmin(VARA_HOEFFDING_K * (double)l1q2[2 * i + 1] * cc.delta, l * l * cc.delta) : 0.5 * l * l * cc.delta;
    if (cc.specH > 0.0) b = fmin(b, cc.specH * (double)l1q2[2 * i + 1]);   // the spectral bound of k_spectral_decide
    double mag = fabs(vdiag) + fabs(vara - vdiag);
    if (c != 0) mag += 2.0 * (fabs(mrho) + cc.absR);
    return b + cc.wErr * (double)l1q2[2 * i + 1] + 0x1p-48 * mag + 0x1p-50 * cc.sumdiag;
}
__global__ __launch_bounds__(256) void k_cert_lb(const double* __restrict__ a, const double* __restrict__ vara, long L,
                                                 const int32_t* __restrict__ l1, const int8_t* __restrict__ cshift,
                                                 const double* __restrict__ vdiag, const double* __restrict__ mrho,
                                                 const VaraHdr* __restrict__ hdr, CertHdr* __restrict__ ch, const unsigned char* __restrict__ xflag) {
    const CertCtx cc = cert_ctx(hdr);
    double best = 0.0;
    int nt = 0;   // markers over the tight threshold (CERT_TIGHT_MAX)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        const double x = a[i], v = vara[i];
        if (!(isfinite(x) && isfinite(v))) continue;
        const double b = cert_bound(cc, l1, i, cshift[i], vdiag[i], mrho[i], v, xflag ? xflag[i] : 0);
        nt += b > cc.flag_rel * fabs(v);
        const double up = v + b;
        if (!(up > 0.0)) continue;
        const double lb = (x * x) / up;
        best = lb > best ? lb : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(best, o); best = y > best ? y : best; nt += __shfl_down(nt, o); }
    if ((threadIdx.x & 63) == 0 && best > 0.0 && isfinite(best)) atomicMax(&ch->lb_bits, (unsigned long long)__double_as_longlong(best));
    if ((threadIdx.x & 63) == 0 && nt) atomicAdd(&ch->tight, nt);
}
__global__ __launch_bounds__(256) void k_cert_select(const double* __restrict__ a, const double* __restrict__ vara, long L,
                                                     const int32_t* __restrict__ l1, const int8_t* __restrict__ cshift,
                                                     const double* __restrict__ vdiag, const double* __restrict__ mrho,
                                                     const VaraHdr* __restrict__ hdr, CertHdr* __restrict__ ch, long* __restrict__ idx,
                                                     double lb_override, const unsigned char* __restrict__ xflag) {
    const CertCtx cc = cert_ctx(hdr);
    // lb_override (not NaN): the lower bound of the maximum over ALL shards of a multi-device scan (exchanged on the host),
    // so that every device selects exactly the candidates a single-device scan of the whole file selects
    const double lb = lb_override == lb_override ? lb_override : __longlong_as_double((long long)ch->lb_bits);
    const double thr = lb * (1.0 - 1e-9);
    const double flag_rel = ch->tight > CERT_TIGHT_MAX ? cc.flag_loose : cc.flag_rel;   // (k_cert_lb counted; nobody writes it here)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        const double x = a[i], v = vara[i];
        if (!(isfinite(x) && isfinite(v))) continue;  // NaN / Inf operands: the fp64 kernel gives the same
        const double b = cert_bound(cc, l1, i, cshift[i], vdiag[i], mrho[i], v, xflag ? xflag[i] : 0);
        const bool flagged = b > flag_rel * fabs(v);
        const double den = v - b;
        const bool cand = !(den > 0.0) || (x * x) / den >= thr;
        if (flagged) atomicAdd(&ch->flagged, 1);
        if (flagged || cand) {
            const int k = atomicAdd(&ch->count, 1);
            if (k < CERT_CAP) idx[k] = i; else ch->overflow = 1;
        }
    }
}
// rows[k][0..n_pad) = Mt8[idx[k]][0..n_pad) for k < count, zero rows up to the next multiple of 128 (the fp64 kernel works on
// whole 128-row blocks)
__global__ __launch_bounds__(256) void k_cert_gather(const int8_t* __restrict__ Mt8, long ld, long n_pad, const CertHdr* __restrict__ ch,
                                                     const long* __restrict__ idx, int8_t* __restrict__ rows) {
    const int cnt = ch->count < CERT_CAP ? ch->count : CERT_CAP;
    const int k = blockIdx.x;
    if (k >= (cnt + 127) / 128 * 128) return;
    const int8_t* src = k < cnt ? Mt8 + idx[k] * ld : nullptr;
    for (long j = (long)threadIdx.x * 16; j < n_pad; j += 256 * 16) {
        i32x4 x = {0, 0, 0, 0};
        if (src) x = *(const i32x4*)(src + j);
        *(i32x4*)(rows + (long)k * n_pad + j) = x;
    }
}

// totals[0..3] += {re-evaluated rows, flagged rows, overflow, rows over the tight threshold} of one certification
__global__ void k_cert_accumulate(const CertHdr* __restrict__ ch, long* __restrict__ totals) {
    totals[0] += ch->count < CERT_CAP ? ch->count : CERT_CAP;
    totals[1] += ch->flagged;
    totals[2] += ch->overflow;
    totals[3] += ch->tight;
}
extern "C" int eagle_dev_cert_accumulate(eagle_ctx* ctx, const void* cert_ws, long* totals_dev, void* stream) {
    hipLaunchKernelGGL(k_cert_accumulate, dim3(1), dim3(1), 0, (hipStream_t)stream, (const CertHdr*)cert_ws, totals_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_accumulate");
    return EAGLE_OK;
}

static int ws_smax(int nslices);  // layout of the vara workspace (defined with it, below)
static size_t ws_vd_off(long n_pad, long L_pad, int smax);
static size_t ws_xflag_off(long n_pad, long L_pad, int smax);
static size_t ws_mr_off(long n_pad, long L_pad, int smax);
static size_t cert_idx_off() { return 256; }
static size_t cert_rows_off() { return cert_idx_off() + (size_t)CERT_CAP * sizeof(long); }
static size_t cert_part_off(long n_pad) { return (cert_rows_off() + (size_t)CERT_CAP * n_pad + 255) / 256 * 256; }
extern "C" int64_t eagle_scan_certify_workspace_bytes(long n_pad) {
    return (int64_t)(cert_part_off(n_pad) + (size_t)eagle_vara_f64_split_partial_doubles(CERT_CAP, n_pad) * sizeof(double));
}
// Phase 1: lower bound of the block's maximum tsq into the head of cert_ws (eagle_cert_info.lower_bound).
extern "C" int eagle_dev_scan_certify_lb(eagle_ctx* ctx, long L, long L_pad, long n_pad, const int8_t* cshift, const int32_t* l1norm,
                                         int nslices, void* vara_ws, const double* a, const double* vara, void* cert_ws, void* stream) {
    if (n_pad % T8 || L_pad % T8 || L < 0 || L > L_pad || !cshift || !l1norm || !cert_ws)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_certify: layout contract violated");
    hipStream_t s = (hipStream_t)stream;
    CertHdr* ch = (CertHdr*)cert_ws;
    hipError_t e = hipMemsetAsync(ch, 0, 256, s);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "scan_certify memset");
    if (L == 0) return EAGLE_OK;
    const int smax = ws_smax(nslices);
    const VaraHdr* hdr = (const VaraHdr*)vara_ws;
    const double* vdiag = (const double*)((const char*)vara_ws + ws_vd_off(n_pad, L_pad, smax));
    const double* mrho = (const double*)((const char*)vara_ws + ws_mr_off(n_pad, L_pad, smax));
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cert_lb, dim3(blocks), dim3(256), 0, s, a, vara, L, l1norm, cshift, vdiag, mrho, hdr, ch,
                       (const unsigned char*)vara_ws + ws_xflag_off(n_pad, L_pad, smax));
    e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_lb");
    return EAGLE_OK;
}
// Phase 2: select (against the block's own lower bound, or lb_override if it is not NaN), gather, re-evaluate, write back.
extern "C" int eagle_dev_scan_certify_apply(eagle_ctx* ctx, const int8_t* Mt8, long L, long L_pad, long n_pad, long ld, const int8_t* cshift,
                                            const int32_t* l1norm, int nslices, void* vara_ws, const double* Wu, const double* a, double* vara,
                                            void* cert_ws, double lb_override, void* stream) {
    if (n_pad % T8 || L_pad % T8 || ld % 16 || n_pad > ld || L < 0 || L > L_pad || !cshift || !l1norm || !cert_ws)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "scan_certify: layout contract violated");
    if (L == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    CertHdr* ch = (CertHdr*)cert_ws;
    const int smax = ws_smax(nslices);
    const VaraHdr* hdr = (const VaraHdr*)vara_ws;
    const double* vdiag = (const double*)((const char*)vara_ws + ws_vd_off(n_pad, L_pad, smax));
    const double* mrho = (const double*)((const char*)vara_ws + ws_mr_off(n_pad, L_pad, smax));
    long* idx = (long*)((char*)cert_ws + cert_idx_off());
    int8_t* rows = (int8_t*)cert_ws + cert_rows_off();
    double* partial = (double*)((char*)cert_ws + cert_part_off(n_pad));
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cert_select, dim3(blocks), dim3(256), 0, s, a, vara, L, l1norm, cshift, vdiag, mrho, hdr, ch, idx, lb_override,
                       (const unsigned char*)vara_ws + ws_xflag_off(n_pad, L_pad, smax));
    hipLaunchKernelGGL(k_cert_gather, dim3(CERT_CAP), dim3(256), 0, s, Mt8, ld, n_pad, ch, idx, rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "scan_certify");
    if (ctx->w8_active && ctx->w8_Wu == Wu) {
        // W from the int8 engine carries its own error: the candidates are re-evaluated against the TRUE W, m^T (S (V (S m))) in fp64
        // (the host reads the count: one small round trip); an overflowing certificate first replaces W by the fp64 products
        CertHdr h;
        e = hipMemcpyAsync(&h, ch, sizeof h, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "candidate count");
        if (!h.overflow) return eagle_w8_true_vara(ctx, rows, h.count, n_pad, n_pad, idx, vara, stream);
        int r8 = eagle_w8_redo_f64(ctx, n_pad, stream);
        if (r8) return r8;
        return eagle_dev_vara_f64_gated(ctx, Mt8, L_pad, n_pad, ld, Wu, vara, &ch->overflow, stream);
    }
    int rc = eagle_dev_vara_f64_split(ctx, rows, CERT_CAP, n_pad, n_pad, Wu, &ch->count, idx, partial, vara, stream);
    if (rc) return rc;
    // more than CERT_CAP markers qualified: the whole block in fp64 (dropped on the device otherwise)
    return eagle_dev_vara_f64_gated(ctx, Mt8, L_pad, n_pad, ld, Wu, vara, &ch->overflow, stream);
}
extern "C" int eagle_dev_scan_certify(eagle_ctx* ctx, const int8_t* Mt8, long L, long L_pad, long n_pad, long ld, const int8_t* cshift,
                                      const int32_t* l1norm, int nslices, void* vara_ws, const double* Wu, const double* a, double* vara,
                                      void* cert_ws, void* stream) {
    int rc = eagle_dev_scan_certify_lb(ctx, L, L_pad, n_pad, cshift, l1norm, nslices, vara_ws, a, vara, cert_ws, stream);
    if (rc) return rc;
    return eagle_dev_scan_certify_apply(ctx, Mt8, L, L_pad, n_pad, ld, cshift, l1norm, nslices, vara_ws, Wu, a, vara, cert_ws,
                                        __builtin_nan(""), stream);
}

// ------------------------------------------------------------------------------------------------
// The same certification for a scan that does not see all its markers at once (a file streamed through HBM in marker blocks, the
// shards of a multi-device context): ONE lower bound over every block of every device, so that the returned bits do not depend
// on how the markers were cut up (a block-by-block certificate re-evaluated a superset of candidates, and up
// to 64 vara values differed at 1e-9 between a streamed and a resident scan of the same file).
//   per block, right after the vara kernel:  bound[i] = b_i                      (eagle_dev_cert_bounds; 8 bytes per marker stay)
//   after the last block:                    LB = max_i a_i^2 / (vara_i + b_i)   (eagle_dev_cert_lb_b; devices exchange theirs)
//                                            idx[] = markers with b_i > 1.8 budget |vara_i| or a_i^2 / (vara_i - b_i) >= LB (1 - 1e-9)
//                                                                                 (eagle_dev_cert_select_b)
//   the caller gathers those rows (from the resident image, or by re-reading just them from the file) and runs
//   eagle_dev_vara_f64_split on them: bitwise the values, and so the selected marker, of the one-block scan.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cert_bounds(const double* __restrict__ vara, long L, const int32_t* __restrict__ l1,
                                                     const int8_t* __restrict__ cshift, const double* __restrict__ vdiag,
                                                     const double* __restrict__ mrho, const VaraHdr* __restrict__ hdr, double* __restrict__ bound,
                                                     const unsigned char* __restrict__ xflag) {
    const CertCtx cc = cert_ctx(hdr);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256)
        bound[i] = cert_bound(cc, l1, i, cshift[i], vdiag[i], mrho[i], vara[i], xflag ? xflag[i] : 0);
}
__global__ __launch_bounds__(256) void k_cert_lb_b(const double* __restrict__ a, const double* __restrict__ vara, const double* __restrict__ bound,
                                                   long L, CertHdr* __restrict__ ch, const VaraHdr* __restrict__ hdr) {
    double best = 0.0;
    const double flag_tight = VARA_FLAG_FACTOR * hdr->budget;
    int nt = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        const double x = a[i], v = vara[i];
        if (!(isfinite(x) && isfinite(v))) continue;
        nt += bound[i] > flag_tight * fabs(v);
        const double up = v + bound[i];
        if (!(up > 0.0)) continue;
        const double lb = (x * x) / up;
        best = lb > best ? lb : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(best, o); best = y > best ? y : best; nt += __shfl_down(nt, o); }
    if ((threadIdx.x & 63) == 0 && best > 0.0 && isfinite(best)) atomicMax(&ch->lb_bits, (unsigned long long)__double_as_longlong(best));
    if ((threadIdx.x & 63) == 0 && nt) atomicAdd(&ch->tight, nt);
}
__global__ __launch_bounds__(256) void k_cert_select_b(const double* __restrict__ a, const double* __restrict__ vara, const double* __restrict__ bound,
                                                       long L, CertHdr* __restrict__ ch, long* __restrict__ idx, double lb, int loose,
                                                       const VaraHdr* __restrict__ hdr) {
    // the budget in force for this scan, or the default behind it (`loose`: more than CERT_TIGHT_MAX markers of the WHOLE scan over the tight one)
    const double flag_rel = VARA_FLAG_FACTOR * (loose ? hdr->budget_loose : hdr->budget);
    const double thr = lb * (1.0 - 1e-9);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        const double x = a[i], v = vara[i];
        if (!(isfinite(x) && isfinite(v))) continue;
        const double b = bound[i];
        const bool flagged = b > flag_rel * fabs(v);
        const double den = v - b;
        const bool cand = !(den > 0.0) || (x * x) / den >= thr;
        if (flagged) atomicAdd(&ch->flagged, 1);
        if (flagged || cand) {
            const int k = atomicAdd(&ch->count, 1);
            if (k < CERT_CAP) idx[k] = i; else ch->overflow = 1;
        }
    }
}
// bound[0..L) of one marker block (vara_ws: the block's workspace as passed to prepare + mfma; vara: the block's values)
extern "C" int eagle_dev_cert_bounds(eagle_ctx* ctx, long L, long L_pad, long n_pad, const int8_t* cshift, const int32_t* l1norm, int nslices,
                                     const void* vara_ws, const double* vara, double* bound, void* stream) {
    if (n_pad % T8 || L_pad % T8 || L < 0 || L > L_pad || !cshift || !l1norm || !bound)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "cert_bounds: layout contract violated");
    if (L == 0) return EAGLE_OK;
    const int smax = ws_smax(nslices);
    const VaraHdr* hdr = (const VaraHdr*)vara_ws;
    const double* vdiag = (const double*)((const char*)vara_ws + ws_vd_off(n_pad, L_pad, smax));
    const double* mrho = (const double*)((const char*)vara_ws + ws_mr_off(n_pad, L_pad, smax));
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cert_bounds, dim3(blocks), dim3(256), 0, (hipStream_t)stream, vara, L, l1norm, cshift, vdiag, mrho, hdr, bound,
                       (const unsigned char*)vara_ws + ws_xflag_off(n_pad, L_pad, smax));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_bounds");
    return EAGLE_OK;
}
// head of cert_ws zeroed, then eagle_cert_info.lower_bound = max_i a_i^2 / (vara_i + bound_i) over the L markers (0: none positive)
extern "C" int eagle_dev_cert_lb_b(eagle_ctx* ctx, long L, const double* a, const double* vara, const double* bound, void* cert_ws,
                                   const void* vara_ws, void* stream) {
    if (L < 0 || !cert_ws || !vara_ws) return eagle_fail(ctx, EAGLE_ERR_ARG, "cert_lb_b: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(cert_ws, 0, 256, s);
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "cert_lb_b memset");
    if (L == 0) return EAGLE_OK;
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cert_lb_b, dim3(blocks), dim3(256), 0, s, a, vara, bound, L, (CertHdr*)cert_ws, (const VaraHdr*)vara_ws);
    e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_lb_b");
    return EAGLE_OK;
}
// candidates against `lb` into the index list of cert_ws (count / flagged / overflow in its head, which eagle_dev_cert_lb_b zeroed);
// over_tight: markers of the whole scan over the tight threshold (the sum of eagle_cert_info.over_tight over blocks and devices)
extern "C" int eagle_cert_tight_max(void) { return CERT_TIGHT_MAX; }
extern "C" int eagle_dev_cert_select_b(eagle_ctx* ctx, long L, const double* a, const double* vara, const double* bound, void* cert_ws, double lb,
                                       long over_tight, const void* vara_ws, void* stream) {
    if (L < 0 || !cert_ws || !vara_ws) return eagle_fail(ctx, EAGLE_ERR_ARG, "cert_select_b: bad arguments");
    if (L == 0) return EAGLE_OK;
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cert_select_b, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, vara, bound, L, (CertHdr*)cert_ws,
                       (long*)((char*)cert_ws + cert_idx_off()), lb, over_tight > CERT_TIGHT_MAX ? 1 : 0, (const VaraHdr*)vara_ws);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_select_b");
    return EAGLE_OK;
}
// Re-evaluation of the selected markers in fp64 once their genotype rows sit in cert_ws (eagle_cert_rows): rows of the resident
// image are gathered here (Mt8 != NULL); rows of a streamed file were put there by the caller (Mt8 == NULL).  vara[idx[k]] is
// overwritten with bitwise the value eagle_dev_vara_f64 gives that marker.
extern "C" int8_t* eagle_cert_rows(void* cert_ws) { return (int8_t*)cert_ws + cert_rows_off(); }
extern "C" long* eagle_cert_indices(void* cert_ws) { return (long*)((char*)cert_ws + cert_idx_off()); }
extern "C" int eagle_dev_cert_reevaluate(eagle_ctx* ctx, const int8_t* Mt8, long ld, long n_pad, const double* Wu, double* vara, void* cert_ws, void* stream) {
    CertHdr* ch = (CertHdr*)cert_ws;
    long* idx = (long*)((char*)cert_ws + cert_idx_off());
    int8_t* rows = (int8_t*)cert_ws + cert_rows_off();
    double* partial = (double*)((char*)cert_ws + cert_part_off(n_pad));
    if (Mt8) {
        hipLaunchKernelGGL(k_cert_gather, dim3(CERT_CAP), dim3(256), 0, (hipStream_t)stream, Mt8, ld, n_pad, ch, idx, rows);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_cert_gather");
    }
    if (ctx->w8_active && ctx->w8_Wu == Wu) {   // (see eagle_dev_scan_certify_apply)
        CertHdr h;
        hipError_t e = hipMemcpyAsync(&h, ch, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "candidate count");
        return eagle_w8_true_vara(ctx, rows, h.count < CERT_CAP ? h.count : CERT_CAP, n_pad, n_pad, idx, vara, stream);
    }
    return eagle_dev_vara_f64_split(ctx, rows, CERT_CAP, n_pad, n_pad, Wu, &ch->count, idx, partial, vara, stream);
}

__global__ void k_vara_i8_bound(const VaraHdr* __restrict__ hdr, double* __restrict__ out, int* __restrict__ slices_out) {
    *out = hdr->bound;
    if (slices_out) *slices_out = hdr->S;
}

// workspace: [ VaraHdr (padded to 256) | q: Smax*L_pad int64 | dW: n_pad f64 | vdiag: L_pad f64 | mrho: L_pad f64 | rho: n_pad f64 |
//              column partials of rho: (n_pad/256 + 1) * n_pad f64 | Bs: Smax*n_pad*n_pad int8 ]
#define VARA_SMAX_AUTO 7
static int ws_smax(int nslices) { return (nslices & 0xff) > 0 ? (nslices & 0xff) : VARA_SMAX_AUTO; }  // bit 8: EAGLE_SLICES_STOCHASTIC
static size_t ws_q_off() { return 256; }
static size_t ws_dw_off(long L_pad, int smax) { return 256 + (size_t)smax * L_pad * 8; }
static size_t ws_vd_off(long n_pad, long L_pad, int smax) { return ws_dw_off(L_pad, smax) + (size_t)n_pad * 8; }
static size_t ws_mr_off(long n_pad, long L_pad, int smax) { return ws_vd_off(n_pad, L_pad, smax) + (size_t)L_pad * 8; }           // m^T rho, L_pad f64
static size_t ws_rho_off(long n_pad, long L_pad, int smax) { return ws_mr_off(n_pad, L_pad, smax) + (size_t)L_pad * 8; }          // rho, n_pad f64
static size_t ws_cp_off(long n_pad, long L_pad, int smax) { return ws_rho_off(n_pad, L_pad, smax) + (size_t)n_pad * 8; }          // column partials
static size_t ws_bs_off(long n_pad, long L_pad, int smax) {
    return (ws_cp_off(n_pad, L_pad, smax) + (size_t)n_pad * (size_t)(n_pad / T8 + 1) * 8 + 255) / 256 * 256;
}

// Behind the slice slots: the area of eagle_dev_vara_i8_extend (markers that get the dropped digit back, see there)
//   [ ExtHdr 256 B | idx: cap int32 | flag: L_pad bytes | qc: cap int64 | Xc: cap x n_pad int8 ],  cap = min(65,536, L_pad) rounded to 768
#define EXT_CAP_MAX 65536
struct ExtHdr { VaraHdr h2; int count; int overflow; };   // h2: a header for the vara kernels, which read only its S (1: run, 0: every worker leaves)
static long ext_cap(long L_pad) {
    long mx = EXT_CAP_MAX;
    if (const char* e = getenv("EAGLE_HIP_EXT_CAP")) {   // tests: a small compact image forces the extension through several passes
        const long v = atol(e);
        if (v >= 768 && v < EXT_CAP_MAX) mx = v;
    }
    const long c = L_pad < mx ? L_pad : mx;
    return (c + 767) / 768 * 768;
}
static size_t r256(size_t x) { return (x + 255) / 256 * 256; }
static size_t ws_ext_off(long n_pad, long L_pad, int smax) { return r256(ws_bs_off(n_pad, L_pad, smax) + (size_t)smax * n_pad * n_pad); }
static size_t ws_xidx_off(long n_pad, long L_pad, int smax) { return ws_ext_off(n_pad, L_pad, smax) + 256; }
static size_t ws_xflag_off(long n_pad, long L_pad, int smax) { return ws_xidx_off(n_pad, L_pad, smax) + r256(4 * (size_t)ext_cap(L_pad)); }
static size_t ws_xq_off(long n_pad, long L_pad, int smax) { return ws_xflag_off(n_pad, L_pad, smax) + r256((size_t)L_pad); }
static size_t ws_xc_off(long n_pad, long L_pad, int smax) { return ws_xq_off(n_pad, L_pad, smax) + r256(8 * (size_t)ext_cap(L_pad)); }

extern "C" int64_t eagle_vara_i8_workspace_bytes(long n_pad, long L_pad, int nslices) {
    const int smax = ws_smax(nslices);
    return (int64_t)(ws_xc_off(n_pad, L_pad, smax) + (size_t)ext_cap(L_pad) * (size_t)n_pad);
}

static int vara_i8_check(eagle_ctx* ctx, long L_pad, long n_pad, long ld, int nslices) {
    if (L_pad % T8 || n_pad % T8 || ld % 128 || n_pad > ld || nslices < 0 || (nslices & ~EAGLE_SLICES_STOCHASTIC) > 8 || (double)ld * T8 >= 2147483648.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_i8: layout contract violated (L_pad % 256, n_pad % 256, 0 <= nslices <= 8)");
    // int32 tile row-sum: 64 columns per wave x |T*m'| <= 2*2*128*n_pad each, |m'| <= 2 (accumulation across tiles is int64)
    if (64.0 * 512.0 * (double)n_pad >= 2147483648.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_i8: n too large for the int32 per-slice partial sums; use the fp64 kernel");
    return EAGLE_OK;
}

// Phase 1: max |off-diagonal|, slice count, diagonal vector, ONE fused pass over the genotypes for
// vdiag_i = sum_k m_ik^2 W_kk (and a = Mt8 v if v != NULL), digit slices of the off-diagonal part.
// Which vara kernel a context runs at this size (the prepare step lays the digit slices out for it): the pipelined 384 x 256 kernel
// below 65,536 padded individuals unless a tune switch asks for one of the others (tools/bench_vara.py: 8 = the 256 x 256 form,
// 9 = the compiler-scheduled 384 x 256 form).
static bool vara_piped(const eagle_ctx* ctx, long n_pad) { return ctx->tune != 8 && ctx->tune != 9 && n_pad < 65536; }

// part 0: everything (one block = the whole scan).  A scan that walks several marker blocks of the SAME L_pad through one workspace
// (a streamed file) does the W-dependent work once -- part 1: header, digits of W, rho; 4 n_pad^2 S bytes written -- and per block
// only part 2: q zeroed, ONE pass over the block's genotypes (a = Mt8 v, the diagonal term, m^T rho).
extern "C" int eagle_dev_vara_i8_prepare_part(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                              int nslices, void* ws, const double* v, double* a_out, void* stream, int part) {
    int rc = vara_i8_check(ctx, L_pad, n_pad, ld, nslices);
    if (rc) return rc;
    if (L_pad == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int smax = ws_smax(nslices);
    VaraHdr* hdr = (VaraHdr*)ws;
    double* dW = (double*)((char*)ws + ws_dw_off(L_pad, smax));
    double* vdiag = (double*)((char*)ws + ws_vd_off(n_pad, L_pad, smax));
    int8_t* Bs = (int8_t*)((char*)ws + ws_bs_off(n_pad, L_pad, smax));
    double* rho = (double*)((char*)ws + ws_rho_off(n_pad, L_pad, smax));
    hipError_t e = part == 2 ? hipMemsetAsync((char*)ws + ws_q_off(), 0, ws_dw_off(L_pad, smax) - ws_q_off(), s)   // q only
                             : hipMemsetAsync(ws, 0, ws_dw_off(L_pad, smax), s);                                     // header and q
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_i8 memset");
    if (part != 1) {   // per marker block: nobody has been given the dropped digit back yet (eagle_dev_vara_i8_extend)
        e = hipMemsetAsync((char*)ws + ws_ext_off(n_pad, L_pad, smax), 0, 256, s);
        if (e == hipSuccess) e = hipMemsetAsync((char*)ws + ws_xflag_off(n_pad, L_pad, smax), 0, (size_t)L_pad, s);
        if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_i8 extension memset");
    }
    if (part != 2) {
        hipLaunchKernelGGL(k_absmax_offdiag, dim3(1024), dim3(256), 0, s, Wu, n_pad, (unsigned long long*)&hdr->maxabs_off);
        // W from the int8 engine (eagle_w8.hip, the image it left is the one being prepared): its error bound rides in the header, and
        // the correction vector of the re-centred markers comes from r = S (V (S 1)) instead of the row sums of this image
        const bool w8 = ctx->w8_active && ctx->w8_Wu == Wu;
        hipLaunchKernelGGL(k_vara_prep, dim3(1), dim3(256), 0, s, Wu, n_pad, nslices, hdr, dW, ctx->scan_budget, w8 ? ctx->w8_eta : 0.0, ctx->scan_budget_tight);
        // correction terms of the re-centred markers: rho and R from Wu, m^T rho from the genotype pass
        double* colpart = (double*)((char*)ws + ws_cp_off(n_pad, L_pad, smax));
        if (w8) {
            rc = eagle_w8_rho(ctx, Wu, n_pad, rho, stream);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_rho_rows, dim3((unsigned)n_pad), dim3(256), 0, s, Wu, n_pad, rho);
            hipLaunchKernelGGL(k_rho_cols, dim3((unsigned)(n_pad / 256), (unsigned)(n_pad / 256)), dim3(256), 0, s, Wu, n_pad, colpart);
            hipLaunchKernelGGL(k_rho_colsum, dim3((unsigned)(n_pad / 256)), dim3(256), 0, s, colpart, n_pad, rho);
        }
        hipLaunchKernelGGL(k_rho_final, dim3(1), dim3(1024), 0, s, n_pad, rho, hdr);
    }
    if (part != 1) {
        // ONE pass over the genotypes: a = Mt8 v (if asked for; a NULL a_out drops it), the diagonal term, and m^T rho
        double* mrho = (double*)((char*)ws + ws_mr_off(n_pad, L_pad, smax));
        rc = eagle_dev_gemv3_i8(ctx, Mt8, L_pad, n_pad, ld, v ? v : dW, dW, rho, 1.0, v ? a_out : nullptr, vdiag, mrho, stream);
        if (rc) return rc;
    }
    if (part != 2) {
        dim3 g2((unsigned)(n_pad / 32), (unsigned)(n_pad / 32));
        hipLaunchKernelGGL(k_slice_w, g2, dim3(256), 0, s, Wu, n_pad, hdr, Bs, vara_piped(ctx, n_pad) ? 1 : 0);
        // one digit fewer if the spectral bound of the last digit allows it (automatic digit count, round to nearest): the last digit's
        // symmetric image goes into the spare slot of the slice area (at most 6 of the 7 are in use), the row sums into the column
        // partials of rho (free again after k_rho_final)
        if (nslices == 0 && !ctx->spectral_off && ctx->tune != 29) {
            int8_t* Ds = Bs + (size_t)(smax - 1) * n_pad * n_pad;
            unsigned long long* rsum = (unsigned long long*)((char*)ws + ws_cp_off(n_pad, L_pad, smax));
            const int* pairs = nullptr;
            const int nt = (int)(n_pad / T8);
            rc = syrk_pair_table(ctx, nt, &pairs, s);
            if (rc) return rc;
            const int npairs = nt * (nt + 1) / 2;
            hipLaunchKernelGGL(k_last_digit_sym, g2, dim3(256), 0, s, Wu, n_pad, hdr, Ds, smax);
            e = hipMemsetAsync(rsum, 0, sizeof(unsigned long long) * (size_t)n_pad, s);
            if (e != hipSuccess) return eagle_fail_hip(ctx, e, "spectral row sums memset");
            hipLaunchKernelGGL(k_gram_rowabs_i8, dim3((unsigned)((npairs + 7) / 8 * 8)), dim3(512), 0, s, Ds, n_pad, pairs, npairs, n_pad / BK8, rsum,
                               (const int*)nullptr);
            hipLaunchKernelGGL(k_spectral_decide, dim3(1), dim3(256), 0, s, rsum, n_pad, hdr, ctx->scan_budget, ctx->scan_budget_tight, smax, 1, 0);
            // second level, dropped on the device unless the first declined: E_hi into the next spare slot, then its Gram row sums
            int8_t* Ehi = Bs + (size_t)(smax - 2) * n_pad * n_pad;
            int shift = 8;   // 127 * 2^shift >= 8 standard deviations sqrt(n) 74^2 of a random Gram entry
            while (shift < 23 && 127.0 * (double)(1 << shift) < 8.0 * 5476.0 * sqrt((double)n_pad)) shift++;
            const int nsc = (nt + 5) / 6;
            hipLaunchKernelGGL(k_gram_hi_i8, dim3((unsigned)((nsc * nsc * 36 + 7) / 8 * 8)), dim3(512), 0, s, Ds, n_pad, nt, n_pad / BK8, shift, hdr, Ehi);
            e = hipMemsetAsync(rsum, 0, sizeof(unsigned long long) * (size_t)n_pad, s);
            if (e != hipSuccess) return eagle_fail_hip(ctx, e, "spectral row sums memset");
            hipLaunchKernelGGL(k_gram_rowabs_i8, dim3((unsigned)((npairs + 7) / 8 * 8)), dim3(512), 0, s, Ehi, n_pad, pairs, npairs, n_pad / BK8, rsum,
                               (const int*)&hdr->spec_try2);
            hipLaunchKernelGGL(k_spectral_decide, dim3(1), dim3(256), 0, s, rsum, n_pad, hdr, ctx->scan_budget, ctx->scan_budget_tight, smax, 2, shift);
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_i8_prepare");
    return EAGLE_OK;
}
extern "C" int eagle_dev_vara_i8_prepare(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                         int nslices, void* ws, const double* v, double* a_out, void* stream) {
    return eagle_dev_vara_i8_prepare_part(ctx, Mt8, L_pad, n_pad, ld, Wu, nslices, ws, v, a_out, stream, 0);
}

// The int8 MFMA kernel over all (marker tile, slice) workers of an image: q[s][i] += ... for the hdr->S slices at Bs (grid sized for smax).
static int vara_i8_launch(eagle_ctx* ctx, const int8_t* Mt8s, long L_pad, long n_pad, long ld, int smax, const VaraHdr* hdr, const int8_t* Bs,
                          long long* q, hipStream_t s, bool ext = false) {
    const int ntm = (int)(L_pad / T8);
    const int groups = (ntm + 7) / 8;
    if (!ctx->attr_vara_i8) {  // per device: a second ctx on another GPU must set it again
        hipError_t ea = hipFuncSetAttribute((const void*)k_vara_i8, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE_BYTES);
        if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_vara_i8)");
        ctx->attr_vara_i8 = true;
    }
    // The 384 x 256 tile is the default (C2: 23.05 -> 21.91 ms, C3 shape: 45.2 -> 43.1 ms per 262144 markers; profiles/
    // r02_ab_vara_tile.txt), since round 2 in its asm-pipelined form k_vara_i8p (another -4 to -5 %, profiles/r02_ab_vara_pipe.txt).
    // All forms give bit-identical q.  The 256 x 256 form serves n_pad >= 65536 (int32 range of the tile row-dots).  A/B switches of tools/bench_vara.py: tune 8 = the 256 x 256 form, 9 = the compiler-scheduled 384 x 256
    // form (k_vara_i8w), 7 = whole workers in the last round.
    if (vara_piped(ctx, n_pad) || (ctx->tune == 9 && n_pad < 32768)) {
        const bool piped = vara_piped(ctx, n_pad);
        // paced workers (a soft barrier per column-tile pair, round 3's tune 14): 1 % slower at n_pad = 10,240, where the L2 serves 68-72 % of
        // the stage fills anyway, 3.3 % FASTER at n_pad = 50,176, where the unpaced workers drift apart over the 5x longer K loops and the hit
        // rate falls to 53 % (profiles/r04_rocprof_C4, r04_vara_pace_50k.txt): on from 32,768 padded individuals up; same integer sums either way
        const bool pace = piped && !ext && (ctx->tune == 14 || (ctx->tune == 0 && n_pad >= 32768));
        const bool px = piped && ext;
        const void* kfn = !piped ? (const void*)k_vara_i8w
                                 : (pace ? (const void*)k_vara_i8p<true> : (px ? (const void*)k_vara_i8p<false, true> : (const void*)k_vara_i8p<false>));
        bool& attr = !piped ? ctx->attr_vara_i8w : (pace ? ctx->attr_vara_i8pp : (px ? ctx->attr_vara_i8px : ctx->attr_vara_i8p));
        if (!attr) {
            hipError_t ea = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TW_ABYTES + TILE_BYTES));
            if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_vara_i8p/w)");
            attr = true;
        }
        const int ntw = (int)((L_pad + TW_M - 1) / TW_M), gw = (ntw + 7) / 8;
        // per XCD at most gw * smax workers, plus at most 31 * VARA_TAIL_PMAX more blocks when the last round is cut into pieces
        const dim3 gridw((unsigned)(8 * (gw * smax + 31 * VARA_TAIL_PMAX)));
        const int cut = ctx->tune == 7 ? 0 : 1;
        if (!piped) hipLaunchKernelGGL(k_vara_i8w, gridw, dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, Mt8s, ld, ntw, Bs, n_pad, hdr, q, L_pad, cut);
        else if (px) hipLaunchKernelGGL((k_vara_i8p<false, true>), gridw, dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, Mt8s, ld, ntw, Bs, n_pad, hdr, q, L_pad, cut, (int*)nullptr);
        else if (!pace) hipLaunchKernelGGL(k_vara_i8p<false>, gridw, dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, Mt8s, ld, ntw, Bs, n_pad, hdr, q, L_pad, cut, (int*)nullptr);
        else {
            // counters [8 XCDs][rounds][pairs], zeroed per launch, in the ctx's GEMM scratch (free during the scan)
            const size_t need = sizeof(int) * 8 * (size_t)((gw * smax + 31) / 32 + 1) * (size_t)((n_pad / T8 + 1) / 2);
            if (need > ctx->gemm_scratch_cap) {
                if (ctx->gemm_scratch) { (void)hipStreamSynchronize(s); (void)hipFree(ctx->gemm_scratch); ctx->gemm_scratch = nullptr; ctx->gemm_scratch_cap = 0; }
                hipError_t ea = hipMalloc(&ctx->gemm_scratch, need);
                if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "pace counters");
                ctx->gemm_scratch_cap = need;
            }
            hipError_t ea = hipMemsetAsync(ctx->gemm_scratch, 0, need, s);
            if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "pace counters memset");
            hipLaunchKernelGGL(k_vara_i8p<true>, gridw, dim3(512), 2 * (TW_ABYTES + TILE_BYTES), s, Mt8s, ld, ntw, Bs, n_pad, hdr, q, L_pad, cut, (int*)ctx->gemm_scratch);
        }
    } else
    hipLaunchKernelGGL(k_vara_i8, dim3((unsigned)(groups * 8 * smax)), dim3(512), 5 * TILE_BYTES, s, Mt8s, ld, ntm, Bs, n_pad, hdr, q, L_pad);
    return EAGLE_OK;
}

// Phase 2: the int8 MFMA kernel over all (marker tile, slice) workers + the S-term finish.
// Mt8s / cshift: the re-centred genotype image and the per-marker shifts of eagle_dev_marker_shift (prepare ran on the
// original image); cshift == NULL: Mt8s is the original image and no correction applies.
extern "C" int eagle_dev_vara_i8_mfma_shifted(eagle_ctx* ctx, const int8_t* Mt8s, const int8_t* cshift, long L_pad, long n_pad, long ld,
                                              int nslices, void* ws, double* vara_out, double* err_bound_dev, void* stream) {
    int rc = vara_i8_check(ctx, L_pad, n_pad, ld, nslices);
    if (rc) return rc;
    if (L_pad == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int smax = ws_smax(nslices);
    VaraHdr* hdr = (VaraHdr*)ws;
    long long* q = (long long*)((char*)ws + ws_q_off());
    double* vdiag = (double*)((char*)ws + ws_vd_off(n_pad, L_pad, smax));
    double* mrho = (double*)((char*)ws + ws_mr_off(n_pad, L_pad, smax));
    int8_t* Bs = (int8_t*)((char*)ws + ws_bs_off(n_pad, L_pad, smax));
    rc = vara_i8_launch(ctx, Mt8s, L_pad, n_pad, ld, smax, hdr, Bs, q, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_vara_i8_finish, dim3((unsigned)((L_pad + 255) / 256)), dim3(256), 0, s, q, L_pad, hdr, vdiag, cshift, mrho, vara_out);
    if (err_bound_dev) hipLaunchKernelGGL(k_vara_i8_bound, dim3(1), dim3(1), 0, s, hdr, err_bound_dev, (int*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_i8_mfma");
    return EAGLE_OK;
}
// ------------------------------------------------------------------------------------------------
// Giving single markers the dropped digit back (round 3).  Under the spectral bound the scan runs on S = S_sliced - 1 digits and
// |digit error_i| <= specH q2_i.  A marker whose quadratic form is far below q2_i mean(W_kk) -- one that IS a top eigenvector of the
// kinship, in a panel of strongly diverged sub-populations -- fails its budget under that bound although the digit that was cut would
// settle it.  Instead of sending thousands of such markers to the fp64 kernel (or the block to the fp64 fallback), they get the last
// digit's term exactly: rows gathered from the re-centred image into a compact image, ONE more run of the vara kernel on it with the
// last digit slice only, q[S_sliced-1][i] filled in, vara_i re-formed from all S_sliced digits (bitwise the value a scan on S_sliced
// digits gives that marker) and a flag that makes cert_bound use the rounding bound of S_sliced digits for it.  A per-marker decision
// from the marker's own numbers: the same whatever the blocking.  Everything is dropped on the device when nobody qualifies (the
// vara kernel sees S = 0: every worker leaves).  More than `cap` markers in a block: nobody is extended, the certificate will
// overflow and the block take the fp64 fallback as before.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ext_select(const double* __restrict__ vara, long L, const int32_t* __restrict__ l1q2, const VaraHdr* __restrict__ hdr,
                                                    ExtHdr* __restrict__ xh, int* __restrict__ idx, unsigned char* __restrict__ flag, int cap) {
    const double H = hdr->specH;
    if (!(H > 0.0)) return;
    const double delta = ldexp(1.0, hdr->e + 1 - 8 * hdr->S) * (1.0 + 0x1p-8), thr = VARA_FLAG_FACTOR * hdr->budget;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
        const double v = vara[i];
        if (!isfinite(v) || flag[i]) continue;   // (flag: extended by an earlier pass of this block)
        const double l = (double)l1q2[2 * i];
        const double b = fmin(H * (double)l1q2[2 * i + 1], 0.5 * l * l * delta) + hdr->wErr * (double)l1q2[2 * i + 1];
        if (b > thr * fabs(v)) {
            const int k = atomicAdd(&xh->count, 1);
            if (k < cap) { idx[k] = (int)i; flag[i] = 1; } else xh->overflow = 1;
        }
    }
}
// S of the second header: 1 = run the last digit slice on the compact image.  The slice is copied to the fixed slot `spare` (the
// kernels take their slice base from the host, which does not know S_sliced); no spare slot: nobody is extended.
__global__ void k_ext_head(const VaraHdr* __restrict__ hdr, ExtHdr* __restrict__ xh, unsigned char* __restrict__ flag, const int* __restrict__ idx, int cap,
                           int spare) {
    // more than `cap` qualified: the first cap are extended in this pass, the rest by the next one (eagle_dev_vara_i8_extend)
    const bool on = xh->count > 0 && hdr->S_sliced - 1 < spare;
    if (!on && threadIdx.x == 0 && xh->count > 0) {   // take the flags back: these markers keep the spectral bound (and will be re-evaluated)
        const int c = xh->count < cap ? xh->count : cap;
        for (int k = 0; k < c; k++) flag[idx[k]] = 0;
        xh->count = 0;
    }
    if (threadIdx.x == 0) {
        if (xh->count > cap) xh->count = cap;
        xh->h2.S = on ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_ext_copy_slice(const ExtHdr* __restrict__ xh, const VaraHdr* __restrict__ hdr, int8_t* __restrict__ Bs, long nn, int spare) {
    if (!xh->h2.S) return;
    const i32x4* src = (const i32x4*)(Bs + (long)(hdr->S_sliced - 1) * nn);
    i32x4* dst = (i32x4*)(Bs + (long)spare * nn);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nn / 16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// Xc[k][0..n_pad) = Mt8s[idx[k]][0..n_pad) for k < count, zero rows behind them up to cap; qc zeroed
__global__ __launch_bounds__(256) void k_ext_gather(const int8_t* __restrict__ Mt8s, long ld, long n_pad, const ExtHdr* __restrict__ xh, const int* __restrict__ idx,
                                                    int8_t* __restrict__ Xc, long long* __restrict__ qc) {
    if (!xh->h2.S) return;
    const int k = blockIdx.x;
    const int8_t* src = k < xh->count ? Mt8s + (long)idx[k] * ld : nullptr;
    for (long j = (long)threadIdx.x * 16; j < n_pad; j += 256 * 16) {
        i32x4 x = {0, 0, 0, 0};
        if (src) x = *(const i32x4*)(src + j);
        *(i32x4*)(Xc + (long)k * n_pad + j) = x;
    }
    if (threadIdx.x == 0) qc[k] = 0;
}
__global__ __launch_bounds__(256) void k_ext_apply(const ExtHdr* __restrict__ xh, const int* __restrict__ idx, const long long* __restrict__ qc,
                                                   long long* __restrict__ q, long Lp, const VaraHdr* __restrict__ hdr, const double* __restrict__ vdiag,
                                                   const int8_t* __restrict__ cshift, const double* __restrict__ mrho, double* __restrict__ vara) {
    if (!xh->h2.S) return;
    const int cnt = xh->count;
    const int Sc = hdr->S_sliced;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const long i = idx[k];
        q[(long)(Sc - 1) * Lp + i] = qc[k];
        vara[i] = vara_from_q(q, Lp, i, Sc, hdr, vdiag, cshift, mrho);
    }
}
// After eagle_dev_vara_i8_mfma_shifted and before the certification, same workspace / image / cshift / l1norm {sum |m'|, sum m'^2}.
extern "C" int eagle_dev_vara_i8_extend(eagle_ctx* ctx, const int8_t* Mt8s, const int8_t* cshift, const int32_t* l1norm, long L, long L_pad, long n_pad,
                                        long ld, int nslices, void* ws, double* vara, void* stream) {
    int rc = vara_i8_check(ctx, L_pad, n_pad, ld, nslices);
    if (rc) return rc;
    if (L <= 0 || nslices != 0 || !l1norm || !cshift) return EAGLE_OK;   // the spectral bound is only ever in use with the automatic digit count
    hipStream_t s = (hipStream_t)stream;
    const int smax = ws_smax(nslices);
    const VaraHdr* hdr = (const VaraHdr*)ws;
    long long* q = (long long*)((char*)ws + ws_q_off());
    const double* vdiag = (const double*)((char*)ws + ws_vd_off(n_pad, L_pad, smax));
    const double* mrho = (const double*)((char*)ws + ws_mr_off(n_pad, L_pad, smax));
    int8_t* Bs = (int8_t*)((char*)ws + ws_bs_off(n_pad, L_pad, smax));
    ExtHdr* xh = (ExtHdr*)((char*)ws + ws_ext_off(n_pad, L_pad, smax));
    int* idx = (int*)((char*)ws + ws_xidx_off(n_pad, L_pad, smax));
    unsigned char* flag = (unsigned char*)ws + ws_xflag_off(n_pad, L_pad, smax);
    long long* qc = (long long*)((char*)ws + ws_xq_off(n_pad, L_pad, smax));
    int8_t* Xc = (int8_t*)ws + ws_xc_off(n_pad, L_pad, smax);
    const long cap = ext_cap(L_pad);
    const int spare = smax - 3;   // slot 4 of 7: slots 5 and 6 hold E_hi and Ds of the spectral bound
    unsigned blocks = (unsigned)((L + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    // A block with more qualifying markers than the compact image holds (cap = min(65,536, L_pad)) is worked in up to four passes of
    // cap markers (a per-block overflow rule made the returned bits depend on the blocking); a pass with nobody left drops
    // its kernels on the device.  More than 4 cap in ONE block: the rest keep the spectral bound and go to the certificate.
    const int passes = cap < L_pad ? 4 : 1;
    for (int pass = 0; pass < passes; pass++) {
        if (pass) {
            hipError_t em = hipMemsetAsync(&xh->count, 0, 2 * sizeof(int), s);   // count, overflow (the flags stay)
            if (em != hipSuccess) return eagle_fail_hip(ctx, em, "vara_i8_extend memset");
        }
        hipLaunchKernelGGL(k_ext_select, dim3(blocks), dim3(256), 0, s, vara, L, l1norm, hdr, xh, idx, flag, (int)cap);
        hipLaunchKernelGGL(k_ext_head, dim3(1), dim3(64), 0, s, hdr, xh, flag, idx, (int)cap, spare);
        if (pass == 0) hipLaunchKernelGGL(k_ext_copy_slice, dim3(1024), dim3(256), 0, s, xh, hdr, Bs, n_pad * n_pad, spare);
        hipLaunchKernelGGL(k_ext_gather, dim3((unsigned)cap), dim3(256), 0, s, Mt8s, ld, n_pad, xh, idx, Xc, qc);
        rc = vara_i8_launch(ctx, Xc, cap, n_pad, n_pad, 1, &xh->h2, Bs + (size_t)spare * n_pad * n_pad, qc, s, true);
        if (rc) return rc;
        hipLaunchKernelGGL(k_ext_apply, dim3(64), dim3(256), 0, s, xh, idx, qc, q, L_pad, hdr, vdiag, cshift, mrho, vara);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_i8_extend");
    return EAGLE_OK;
}

extern "C" int eagle_dev_vara_i8_mfma(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, int nslices, void* ws,
                                      double* vara_out, double* err_bound_dev, void* stream) {
    return eagle_dev_vara_i8_mfma_shifted(ctx, Mt8, nullptr, L_pad, n_pad, ld, nslices, ws, vara_out, err_bound_dev, stream);
}

extern "C" int eagle_dev_vara_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                 int nslices, void* ws, double* vara_out, double* err_bound_dev, void* stream) {
    int rc = eagle_dev_vara_i8_prepare(ctx, Mt8, L_pad, n_pad, ld, Wu, nslices, ws, nullptr, nullptr, stream);
    if (rc) return rc;
    return eagle_dev_vara_i8_mfma(ctx, Mt8, L_pad, n_pad, ld, nslices, ws, vara_out, err_bound_dev, stream);
}

// ================================================================================================
// vara on the block-scaled matrix path: genotypes as fp4 (e2m1: -1, 0, +1 are exact), digits of Wu as fp6 (e2m3).
//
// v_mfma_scale_f32_32x32x64_f8f6f4 multiplies K = 64 per instruction in the cycles the int8 form needs for K = 32, and an
// e2m3 number holds every integer d in [-16, 16] as d/8.  So Wu is cut into balanced base-33 digits instead of base-256
// ones (5.04 bits per digit at twice the MAC rate = 10.1 bits per matrix cycle against 8), the products m * d/8 and their
// fp32 sums are exact (multiples of 1/8 below 2^24), and everything downstream of the accumulators is the integer
// arithmetic of the int8 path.  tools/ubench/fp6_probe.hip: bit-exact against the integer product, 6.0 POP/s against
// 3.04 POP/s for the int8 instruction in the same register-resident loop on random operands.
//
//   Wu = 2^-f * sum_{s<S} D_s 33^s + R,  |R_jk| <= 2^(-f-1),  f = 5S - 1 - e,  max|Wu| < 2^e,  D_s in [-16,16]
//   (Q = rint(Wu 2^f) is below 2^(5S-1) <= (33^S - 1)/2, the largest balanced S-digit number, for S <= 12.)
//
// Operand images.  A: Mt4[L_pad][n_pad/2], two genotypes per byte (low nibble first) -- a 256-marker x 256-individual
// stage is 256 rows x 128 bytes, the same LDS image, DMA and fragment addressing as the int8 engine's 128-individual
// stage.  B: per (slice, column tile ct, stage kt <= ct) one 48 KiB blob already in LDS order: a lane's fragment of 32
// digits is a 24-byte little-endian 6-bit stream, stored as its first 16 bytes in a 256 x 128-byte plane (chunk index
// XOR (row>>1)&7, read with ds_read_b128 like the A image) and its last 8 bytes in a 256 x 64-byte plane (8-byte slot
// XOR (row>>2)&7, ds_read_b64, conflict-free); the blob is copied global -> LDS by 48 linear 1 KiB LDS-DMA pieces.
// A stage is 80 KiB, double buffered = the whole 160 KiB of LDS.
// ================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define F6_A_BYTES 32768
#define F6_B16_BYTES 32768
#define F6_B8_BYTES 16384
#define F6_BLOB (F6_B16_BYTES + F6_B8_BYTES)
#define F6_STAGE (F6_A_BYTES + F6_BLOB)
#define VARA6_SMAX_AUTO 11

// 16 genotypes (one 16-byte load) -> 8 bytes per thread.  int8 m in {0x00, 0x01, 0xFF} -> e2m1 code ((m & 1) << 1) | ((m & 0x80) >> 4)
// = {0x0, 0x2, 0xA}; the codes of bytes 0,1 / 2,3 of a dword fold into its bytes 0 / 2 with one shift-or.
__global__ __launch_bounds__(256) void k_pack_fp4(const int8_t* __restrict__ in, long rows, long ld_in, uint8_t* __restrict__ out, long ld4) {
    const long row = blockIdx.y;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;  // group of 16 genotypes
    if (g * 8 >= ld4) return;
    const i32x4 x = *(const i32x4*)(in + row * ld_in + g * 16);
    unsigned h[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned u = (unsigned)x[i];
        const unsigned c = ((u & 0x01010101u) << 1) | ((u & 0x80808080u) >> 4);
        const unsigned t = c | (c >> 4);
        h[i] = (t & 0xffu) | ((t >> 8) & 0xff00u);
    }
    i32x2 o;
    o[0] = (int)(h[0] | (h[1] << 16));
    o[1] = (int)(h[2] | (h[3] << 16));
    *(i32x2*)(out + row * ld4 + g * 8) = o;
}
extern "C" int eagle_dev_pack_fp4(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, void* Mt4, void* stream) {
    if (n_pad % T8 || ld % 16 || n_pad > ld) return eagle_fail(ctx, EAGLE_ERR_ARG, "pack_fp4: layout contract violated");
    if (L_pad <= 0) return EAGLE_OK;
    for (long r0 = 0; r0 < L_pad; r0 += 65535) {
        const long nr = L_pad - r0 < 65535 ? L_pad - r0 : 65535;
        dim3 grid((unsigned)((n_pad / 16 + 255) / 256), (unsigned)nr);
        hipLaunchKernelGGL(k_pack_fp4, grid, dim3(256), 0, (hipStream_t)stream, Mt8 + r0 * ld, nr, ld, (uint8_t*)Mt4 + r0 * (n_pad / 2), n_pad / 2);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_pack_fp4");
    return EAGLE_OK;
}

// The operand image of the MM^T kernel straight from the marker-major genotypes: M4[individual][marker / 2] (fp4, two per byte,
// the even marker in the low nibble) from Mt8[marker][individual] in ONE pass -- a 256-marker x 128-individual tile goes through LDS,
// each thread gathers 128 markers of its individual (all lanes of a wave read one LDS row: 64 consecutive bytes, conflict
// free) and writes their 64 bytes.  Replaces k_transpose_i8 + k_pack_fp4 (read 2 + write 1.5 bytes per genotype, and a second
// int8 image of the shard) by read 1 + write 0.5.  Same bytes as the two-pass form (tests/test_gpu_parity.py).
__global__ __launch_bounds__(256) void k_transpose_pack_fp4(const int8_t* __restrict__ in, long ld_in, uint8_t* __restrict__ out, long ld4) {
    // 256 markers x 128 individuals per block: whole 128-byte lines on both sides (with 64-individual tiles every input line was
    // fetched by two blocks far apart in the grid: 25 GB of traffic for 15 GB of data at 10,000 x 1,000,000)
    __shared__ __attribute__((aligned(16))) int8_t tile[256][128];
    const long r0 = (long)blockIdx.x * 256, c0 = (long)blockIdx.y * 128;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = (t >> 3) + 32 * i, ch = (t & 7) * 16;
        *(i32x4*)(&tile[row][ch]) = *(const i32x4*)(in + (r0 + row) * ld_in + c0 + ch);
    }
    __syncthreads();
    const int c = t & 127, part = t >> 7;   // a wave reads 64 consecutive bytes of one LDS row per instruction: conflict free
    const uint8_t* col = (const uint8_t*)&tile[part * 128][c];
    i32x4 ov[4];
#pragma unroll
    for (int v4 = 0; v4 < 4; v4++) {
        unsigned o[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned hw[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                // four genotypes of this individual into one dword, then k_pack_fp4's dword-wide code fold
                const uint8_t* b = col + ((v4 * 4 + w) * 8 + hh * 4) * 128;
                const unsigned x = (unsigned)b[0] | ((unsigned)b[128] << 8) | ((unsigned)b[256] << 16) | ((unsigned)b[384] << 24);
                const unsigned cc = ((x & 0x01010101u) << 1) | ((x & 0x80808080u) >> 4);
                const unsigned tt = cc | (cc >> 4);
                hw[hh] = (tt & 0xffu) | ((tt >> 8) & 0xff00u);
            }
            o[w] = hw[0] | (hw[1] << 16);
        }
        ov[v4] = i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    }
    // the 128 output rows (128 bytes each) go out as whole lines: staged in LDS (the input tile is dead; pitch 144 bytes keeps the
    // 16-byte writes of 8 neighbouring rows on different banks), then 8 rows x 128 contiguous bytes per wave instruction
    __syncthreads();
    uint8_t* stage = (uint8_t*)&tile[0][0];
#pragma unroll
    for (int v4 = 0; v4 < 4; v4++) *(i32x4*)(stage + c * 144 + part * 64 + 16 * v4) = ov[v4];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int chunk = t + 256 * i, row = chunk >> 3, piece = chunk & 7;
        *(i32x4*)(out + (c0 + row) * ld4 + r0 / 2 + piece * 16) = *(const i32x4*)(stage + row * 144 + piece * 16);
    }
}
extern "C" int eagle_dev_transpose_pack_fp4(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, void* M4, long ld4, void* stream) {
    if (L_pad % 256 || n_pad % 128 || ld % 16 || n_pad > ld || ld4 % 16 || ld4 < L_pad / 2 || n_pad / 128 > 65535)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "transpose_pack_fp4: layout contract violated (L_pad % 256, n_pad % 128, ld % 16, ld4 % 16)");
    if (L_pad <= 0 || n_pad <= 0) return EAGLE_OK;
    hipLaunchKernelGGL(k_transpose_pack_fp4, dim3((unsigned)(L_pad / 256), (unsigned)(n_pad / 128)), dim3(256), 0, (hipStream_t)stream, Mt8, ld,
                       (uint8_t*)M4, ld4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_transpose_pack_fp4");
    return EAGLE_OK;
}

// One block: dW, sumdiag, digit count (hdr->S), scale exponent f (hdr->pad) and the error bound n_pad^2 2^(-f-1).
__global__ __launch_bounds__(256) void k_vara_prep6(const double* __restrict__ Wu, long n_pad, int forced, VaraHdr* __restrict__ hdr, double budget,
                                                    double* __restrict__ dW) {
    double s = 0.0;
    for (long k = threadIdx.x; k < n_pad; k += 256) {
        double d = Wu[k * n_pad + k];
        dW[k] = d;
        s += fabs(d);
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        const double mx = hdr->maxabs_off;
        if (mx > 0.0) (void)frexp(mx, &e);
        const double nn = (double)n_pad * (double)n_pad;
        int S = forced;
        if (S <= 0) {
            const double target = budget * 0.5 * red[0];
            S = VARA6_SMAX_AUTO;
            for (int c = 4; c <= VARA6_SMAX_AUTO; c++)
                if (ldexp(nn, e - 5 * c) <= target) { S = c; break; }
        }
        if (mx == 0.0) S = 1;
        hdr->S = S;
        hdr->pad = 5 * S - 1 - e;  // f
        hdr->sumdiag = red[0];
        hdr->bound = mx > 0.0 ? ldexp(nn, e - 5 * S) : 0.0;
    }
}

__device__ __forceinline__ long f6_blob_index(int ct, int kt) { return (long)ct * (ct + 1) / 2 + kt; }

// Digits of the 32 x 32 tile Wu[j0.., k0..] -> its fragment (rows k0%256.., fragment (j0%256)/32) of blob (ct, kt), all slices.
__global__ __launch_bounds__(256) void k_slice_w6(const double* __restrict__ Wu, long np, const VaraHdr* __restrict__ hdr,
                                                  uint8_t* __restrict__ Bs6, long nblobs) {
    const long j0 = (long)blockIdx.y * 32, k0 = (long)blockIdx.x * 32;
    const int ct = (int)(k0 >> 8), kt = (int)(j0 >> 8);
    if (kt > ct) return;
    __shared__ double tile[32][33];
    __shared__ uint8_t codes[12][32][32];  // [slice][kk][jj]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = Wu[(j0 + r) * np + k0 + tx];  // tile[jj][kk]
    __syncthreads();
    const int S = hdr->S, f = hdr->pad;
    for (int jj = ty * 4; jj < ty * 4 + 4; jj++) {
        long long Q = (j0 + jj == k0 + tx) ? 0 : llrint(ldexp(tile[jj][tx], f));
        for (int s = 0; s < S; s++) {
            long long d = Q % 33;           // (-33, 33)
            if (d > 16) d -= 33;
            if (d < -16) d += 33;
            Q = (Q - d) / 33;               // exact
            codes[s][tx][jj] = (uint8_t)((d < 0 ? 0x20 : 0) | (int)(d < 0 ? -d : d));  // e2m3 of d/8: magnitude code = |d|
        }
    }
    __syncthreads();
    const int frag = (int)((j0 & 255) >> 5);
    for (int item = threadIdx.x; item < S * 32 * 8; item += 256) {
        const int g = item & 7, kk = (item >> 3) & 31, s = item >> 8;
        const unsigned c0 = codes[s][kk][4 * g], c1 = codes[s][kk][4 * g + 1], c2 = codes[s][kk][4 * g + 2], c3 = codes[s][kk][4 * g + 3];
        const unsigned w24 = c0 | (c1 << 6) | (c2 << 12) | (c3 << 18);
        const int c = (int)(k0 & 255) + kk;  // row of the blob
        uint8_t* blob = Bs6 + ((size_t)s * nblobs + f6_blob_index(ct, kt)) * F6_BLOB;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int p = 3 * g + b;
            uint8_t* dst = p < 16 ? blob + c * 128 + ((frag ^ ((c >> 1) & 7)) << 4) + p
                                  : blob + F6_B16_BYTES + c * 64 + ((frag ^ ((c >> 2) & 7)) << 3) + (p - 16);
            *dst = (uint8_t)(w24 >> (8 * b));
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_vara_f6(const int8_t* __restrict__ Mt8, long ld, const uint8_t* __restrict__ Mt4, long ld4, int ntm,
                                                    const uint8_t* __restrict__ Bs6, long nblobs, long np, const VaraHdr* __restrict__ hdr,
                                                    long long* __restrict__ q, long Lp) {
    extern __shared__ __attribute__((aligned(1024))) int8_t lds6[];  // 2 x (A 32 KiB | B16 32 KiB | B8 16 KiB)
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int nslices = hdr->S;
    const int mt = (slot / nslices) * 8 + xcd, sl = slot % nslices;
    if (mt >= ntm) return;
    const int nct = (int)(np / T8), npair = (nct + 1) / 2;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld, ld4i = (int)ld4;
    const T8Lane lnA = t8_lane(lane, ld4i);
    const __amdgpu_buffer_rsrc_t rsA4 = t8_rsrc((const int8_t*)Mt4 + (long)mt * T8 * ld4, ld4i);
    const __amdgpu_buffer_rsrc_t rsA8 = t8_rsrc(Mt8 + (long)mt * T8 * ld, ldi);  // epilogue: genotype bytes
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Bs6 + (size_t)sl * nblobs * F6_BLOB), 0, (int)(nblobs * F6_BLOB), 0x00020000);
    const int lane16 = lane << 4;

    VaraIt cur, nxt;  // nk counts 256-individual stages here
    auto set_tile = [&](VaraIt& it) {
        while (it.p < npair) {
            int ct = it.half == 0 ? it.p : nct - 1 - it.p;
            if (it.half == 1 && ct == it.p) { it.p++; it.half = 0; continue; }
            it.ct = ct; it.kt = 0; it.nk = ct + 1; it.valid = true;
            return;
        }
        it.valid = false;
    };
    auto advance = [&](VaraIt& it) {
        if (++it.kt < it.nk) return;
        if (it.half == 0) it.half = 1; else { it.half = 0; it.p++; }
        set_tile(it);
    };
    auto stage = [&](const VaraIt& it, int8_t* dst) {
        t8_stage(rsA4, lnA, ld4i, it.kt * 128, dst, w);  // 256 individuals = 128 bytes of every marker row
        const int blob = (int)f6_blob_index(it.ct, it.kt) * F6_BLOB;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int piece = w * 6 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + F6_A_BYTES + piece * 1024), 16, lane16,
                                                     blob + piece * 1024, 0, 0);
        }
    };
    cur.p = 0; cur.half = 0; set_tile(cur);
    if (!cur.valid) return;
    nxt = cur;
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) acc[m][n][x] = 0.f;
    long long keep[4] = {0, 0, 0, 0};
    const int r = lane & 31, h = lane >> 5;
    const int col = r, hrow = 4 * h;
    const int xsel = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
    // fragment read offsets (bytes) inside a stage
    const int offA = (wr * 128 + r) * 128, offB16 = F6_A_BYTES + (wc * 64 + r) * 128, offB8 = F6_A_BYTES + F6_B16_BYTES + (wc * 64 + r) * 64;
    int ch16[4], ch8[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        ch16[ks] = ((2 * ks + h) ^ ((r >> 1) & 7)) << 4;
        ch8[ks] = ((2 * ks + h) ^ ((r >> 2) & 7)) << 3;
    }

    stage(nxt, lds6);
    advance(nxt);
    __syncthreads();
    int buf = 0;
    while (cur.valid) {
        const int8_t* st = lds6 + buf * F6_STAGE;
        if (nxt.valid) {
            stage(nxt, lds6 + (buf ^ 1) * F6_STAGE);
            advance(nxt);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            i32x8 a[4], bb[2];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const i32x4 v = *(const i32x4*)(st + offA + m * (32 * 128) + ch16[ks]);
                a[m][0] = v[0]; a[m][1] = v[1]; a[m][2] = v[2]; a[m][3] = v[3];
                a[m][4] = 0; a[m][5] = 0; a[m][6] = 0; a[m][7] = 0;
            }
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const i32x4 v = *(const i32x4*)(st + offB16 + n * (32 * 128) + ch16[ks]);
                const i32x2 u = *(const i32x2*)(st + offB8 + n * (32 * 64) + ch8[ks]);
                bb[n][0] = v[0]; bb[n][1] = v[1]; bb[n][2] = v[2]; bb[n][3] = v[3]; bb[n][4] = u[0]; bb[n][5] = u[1];
                bb[n][6] = 0; bb[n][7] = 0;
            }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], bb[n], acc[m][n], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        if (cur.kt == cur.nk - 1) {
            const int mvoff = (wr * 128 + hrow) * ldi + wc * 64 + col;
            const int msoff = cur.ct * T8;
            const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int v16[16], v8[8], v4[4], v2[2];
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const int so = (m * 32 + (x & 3) + 8 * (x >> 2)) * ldi + msoff;
                    const int m0 = (int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rsA8, mvoff, so, 0);
                    const int m1 = (int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rsA8, mvoff, so + 32, 0);
                    v16[x] = (int)(acc[m][0][x] * 8.0f) * m0 + (int)(acc[m][1][x] * 8.0f) * m1;  // accumulators are exact multiples of 1/8
                    acc[m][0][x] = 0.f;
                    acc[m][1][x] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; i++) { int snd = b0 ? v16[i] : v16[i + 8]; int kp = b0 ? v16[i + 8] : v16[i]; v8[i] = kp + __shfl_xor(snd, 1); }
#pragma unroll
                for (int i = 0; i < 4; i++) { int snd = b1 ? v8[i] : v8[i + 4]; int kp = b1 ? v8[i + 4] : v8[i]; v4[i] = kp + __shfl_xor(snd, 2); }
#pragma unroll
                for (int i = 0; i < 2; i++) { int snd = b2 ? v4[i] : v4[i + 2]; int kp = b2 ? v4[i + 2] : v4[i]; v2[i] = kp + __shfl_xor(snd, 4); }
                int v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
                v1 += __shfl_xor(v1, 16);
                keep[m] += v1;
                __builtin_amdgcn_sched_barrier(0);
            }
            advance(cur);
        } else {
            cur.kt++;
        }
        __syncthreads();
        buf ^= 1;
    }
    long long* qs = q + (long)sl * Lp + (long)mt * T8 + wr * 128 + hrow + (xsel & 3) + 8 * (xsel >> 2);
#pragma unroll
    for (int m = 0; m < 4; m++)
        if ((lane & 16) == 0 && keep[m]) atomicAdd((unsigned long long*)&qs[m * 32], (unsigned long long)keep[m]);
}

__global__ __launch_bounds__(256) void k_vara_f6_finish(const long long* __restrict__ q, long Lp, const VaraHdr* __restrict__ hdr,
                                                        const double* __restrict__ vdiag, double* __restrict__ vara) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Lp) return;
    const int S = hdr->S, f = hdr->pad;
    double x = 0.0;
    for (int s = S - 1; s >= 0; s--) x = x * 33.0 + (double)q[(long)s * Lp + i];  // Horner, most significant digit first
    vara[i] = vdiag[i] + ldexp(x, -f);
}

// workspace: [ VaraHdr (256) | q: Smax*L_pad int64 | dW: n_pad f64 | vdiag: L_pad f64 | Bs6: Smax * nblobs * 48 KiB ]
static int ws6_smax(int nslices) { return nslices > 0 ? nslices : VARA6_SMAX_AUTO; }
static long f6_nblobs(long n_pad) { const long nct = n_pad / T8; return nct * (nct + 1) / 2; }
extern "C" int64_t eagle_vara_f6_workspace_bytes(long n_pad, long L_pad, int nslices) {
    const int smax = ws6_smax(nslices);
    return (int64_t)(ws_bs_off(n_pad, L_pad, smax) + (size_t)smax * f6_nblobs(n_pad) * F6_BLOB);
}
static int vara_f6_check(eagle_ctx* ctx, long L_pad, long n_pad, long ld, int nslices) {
    if (L_pad % T8 || n_pad % T8 || ld % 128 || n_pad > ld || nslices < 0 || nslices > 12 || (double)ld * T8 >= 2147483648.0 ||
        (double)f6_nblobs(n_pad) * F6_BLOB >= 4294967296.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_f6: layout contract violated (L_pad % 256, n_pad % 256, 0 <= nslices <= 12)");
    return EAGLE_OK;
}

extern "C" int eagle_dev_vara_f6_prepare(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu, int nslices,
                                         void* ws, const double* v, double* a_out, void* stream) {
    int rc = vara_f6_check(ctx, L_pad, n_pad, ld, nslices);
    if (rc) return rc;
    if (L_pad == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int smax = ws6_smax(nslices);
    VaraHdr* hdr = (VaraHdr*)ws;
    double* dW = (double*)((char*)ws + ws_dw_off(L_pad, smax));
    double* vdiag = (double*)((char*)ws + ws_vd_off(n_pad, L_pad, smax));
    uint8_t* Bs6 = (uint8_t*)((char*)ws + ws_bs_off(n_pad, L_pad, smax));
    hipError_t e = hipMemsetAsync(ws, 0, ws_dw_off(L_pad, smax), s);  // header and q
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_f6 memset");
    hipLaunchKernelGGL(k_absmax_offdiag, dim3(1024), dim3(256), 0, s, Wu, n_pad, (unsigned long long*)&hdr->maxabs_off);
    hipLaunchKernelGGL(k_vara_prep6, dim3(1), dim3(256), 0, s, Wu, n_pad, nslices, hdr, ctx->scan_budget, dW);
    rc = eagle_dev_gemv2_i8(ctx, Mt8, L_pad, n_pad, ld, v ? v : dW, dW, 1.0, v ? a_out : nullptr, vdiag, stream);
    if (rc) return rc;
    dim3 g2((unsigned)(n_pad / 32), (unsigned)(n_pad / 32));
    hipLaunchKernelGGL(k_slice_w6, g2, dim3(256), 0, s, Wu, n_pad, hdr, Bs6, f6_nblobs(n_pad));
    e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_f6_prepare");
    return EAGLE_OK;
}

extern "C" int eagle_dev_vara_f6_mfma(eagle_ctx* ctx, const int8_t* Mt8, const void* Mt4, long L_pad, long n_pad, long ld, int nslices,
                                      void* ws, double* vara_out, double* err_bound_dev, void* stream) {
    int rc = vara_f6_check(ctx, L_pad, n_pad, ld, nslices);
    if (rc) return rc;
    if (L_pad == 0) return EAGLE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int smax = ws6_smax(nslices);
    VaraHdr* hdr = (VaraHdr*)ws;
    long long* q = (long long*)((char*)ws + ws_q_off());
    double* vdiag = (double*)((char*)ws + ws_vd_off(n_pad, L_pad, smax));
    uint8_t* Bs6 = (uint8_t*)((char*)ws + ws_bs_off(n_pad, L_pad, smax));
    if (!ctx->attr_vara_f6) {
        hipError_t ea = hipFuncSetAttribute((const void*)k_vara_f6, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * F6_STAGE);
        if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_vara_f6)");
        ctx->attr_vara_f6 = true;
    }
    const int ntm = (int)(L_pad / T8);
    const int groups = (ntm + 7) / 8;
    hipLaunchKernelGGL(k_vara_f6, dim3((unsigned)(groups * 8 * smax)), dim3(512), 2 * F6_STAGE, s, Mt8, ld, (const uint8_t*)Mt4, n_pad / 2, ntm, Bs6,
                       f6_nblobs(n_pad), n_pad, hdr, q, L_pad);
    hipLaunchKernelGGL(k_vara_f6_finish, dim3((unsigned)((L_pad + 255) / 256)), dim3(256), 0, s, q, L_pad, hdr, vdiag, vara_out);
    if (err_bound_dev) hipLaunchKernelGGL(k_vara_i8_bound, dim3(1), dim3(1), 0, s, hdr, err_bound_dev, (int*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "vara_f6_mfma");
    return EAGLE_OK;
}

// ------------------------------------------------------------------------------------------------
// MM^T on the same instruction with BOTH operands fp4: C32 += M M^T from the individual-major fp4 image M4[n_pad][L_pad/2].
// A stage is 256 markers = 128 bytes per individual row -- the int8 engine's LDS image, DMA and fragment addressing
// unchanged -- so the same 64 KiB fill and the same 32 matrix instructions per wave now cover twice the markers.
// Products are +-1, fp32 partial sums are exact integers (a K split is far below 2^24 markers), int32 atomics as before.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_syrk_f4(const uint8_t* __restrict__ M4, long ld4, const int* __restrict__ pairs, int npairs,
                                                    int nblocks, long nstages, long stages_per_split, int32_t* __restrict__ C, long ldc) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (lid >= nblocks) return;
    const int split = lid / npairs;
    const int pr = pairs[lid - split * npairs];
    const int ti = pr >> 16, tj = pr & 0xffff;
    const long s0 = (long)split * stages_per_split;
    long s1 = s0 + stages_per_split;
    if (s1 > nstages) s1 = nstages;
    if (s0 >= s1) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld4;
    const T8Lane ln = t8_lane(lane, ldi);
    const __amdgpu_buffer_rsrc_t rsA = t8_rsrc((const int8_t*)M4 + (long)ti * T8 * ld4, ldi);
    const __amdgpu_buffer_rsrc_t rsB = t8_rsrc((const int8_t*)M4 + (long)tj * T8 * ld4, ldi);
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) acc[m][n][x] = 0.f;
    t8_stage(rsA, ln, ldi, (int)(s0 * BK8), lds[0][0], w);
    t8_stage(rsB, ln, ldi, (int)(s0 * BK8), lds[0][1], w);
    __syncthreads();
    int cur = 0;
    const T8Read rd = t8_read_init(wr, wc, lane);
    for (long s = s0; s < s1; s++) {
        if (s + 1 < s1) {
            const int kn = (int)((s + 1) * BK8);
            t8_stage(rsA, ln, ldi, kn, lds[cur ^ 1][0], w);
            t8_stage(rsB, ln, ldi, kn, lds[cur ^ 1][1], w);
        }
        const int8_t* pa = lds[cur][0] + rd.offA;
        const int8_t* pb = lds[cur][1] + rd.offB;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            i32x8 a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const i32x4 v = *(const i32x4*)(pa + m * (32 * BK8) + rd.ch[ks]);
                a[m][0] = v[0]; a[m][1] = v[1]; a[m][2] = v[2]; a[m][3] = v[3]; a[m][4] = 0; a[m][5] = 0; a[m][6] = 0; a[m][7] = 0;
            }
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const i32x4 v = *(const i32x4*)(pb + n * (32 * BK8) + rd.ch[ks]);
                b[n][0] = v[0]; b[n][1] = v[1]; b[n][2] = v[2]; b[n][3] = v[3]; b[n][4] = 0; b[n][5] = 0; b[n][6] = 0; b[n][7] = 0;
            }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[n], acc[m][n], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        __syncthreads();
        cur ^= 1;
    }
    const int col = lane & 31, rq = 4 * (lane >> 5);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) {
                long i = (long)ti * T8 + wr * 128 + m * 32 + (x & 3) + 8 * (x >> 2) + rq;
                long j = (long)tj * T8 + wc * 64 + n * 32 + col;
                int v = (int)acc[m][n][x];
                if (v) atomicAdd(&C[i * ldc + j], v);
            }
}

// ------------------------------------------------------------------------------------------------
// k_syrk_f4p: k_syrk_f4 with the k-step in inline asm, software pipelined like k_vara_i8p: both fragment sets double-buffered
// (24 + 24 VGPRs), the six reads of the next k-step issued behind the first six MFMAs of this one (issue order a0 b0 a1 b1 a2 a3,
// counted waits), the eight LDS-DMA loads of the next stage one behind each MFMA that follows the stage barrier (6 in the
// stage's last k-step, whose barrier sits behind its second MFMA, 2 in the next), C = 0 as an inline constant in the very first k-step.
// Same sums of the same exact integers: bit-identical C.
// ------------------------------------------------------------------------------------------------
#define S_MF(c, a, b) "v_mfma_scale_f32_32x32x64_f8f6f4 %[" #c "], %[" #a "], %[" #b "], %[" #c "], %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"
#define S_MZ(c, a, b) "v_mfma_scale_f32_32x32x64_f8f6f4 %[" #c "], %[" #a "], %[" #b "], 0, %[sc], %[sc] op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"
#define S_KSTEP(M, D1, D2, D3, D4)                                                             \
    X_WT(4) M(c00, a0, b0) X_LD(x0, pa, 0) D1                                                  \
    X_WT(3) M(c01, a0, b1) X_LD(y0, pb, 0)                                                     \
            M(c10, a1, b0) X_LD(x1, pa, 4096) D2                                               \
            M(c11, a1, b1) X_LD(y1, pb, 4096)                                                  \
    X_WT(5) M(c20, a2, b0) X_LD(x2, pa, 8192) D3                                               \
            M(c21, a2, b1) X_LD(x3, pa, 12288)                                                 \
    X_WT(6) M(c30, a3, b0) D4                                                                  \
            M(c31, a3, b1)
#define S_KLAST                                                                                \
    X_WT(4) S_MF(c00, a0, b0) X_WT(2) S_MF(c01, a0, b1)                                        \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t"                                           \
    X_LD(x0, pa, 0) X_LD(y0, pb, 0) X_LD(x1, pa, 4096) X_LD(y1, pb, 4096) X_LD(x2, pa, 8192) X_LD(x3, pa, 12288) \
    S_MF(c10, a1, b0) X_DM(vE, a) S_MF(c11, a1, b1) X_DM(vO, a) S_MF(c20, a2, b0) X_DM(vE, a) S_MF(c21, a2, b1) X_DM(vO, a) \
    S_MF(c30, a3, b0) X_DM(vE, b) S_MF(c31, a3, b1) X_DM(vO, b)
#define S_ACC_RW(m) [c##m##0] "+v"(c[m][0]), [c##m##1] "+v"(c[m][1])
#define S_ACC_W(m) [c##m##0] "=&v"(c[m][0]), [c##m##1] "=&v"(c[m][1])
#define S_NEXT [x0] "=&v"(g.a[0]), [x1] "=&v"(g.a[1]), [x2] "=&v"(g.a[2]), [x3] "=&v"(g.a[3]), [y0] "=&v"(g.b[0]), [y1] "=&v"(g.b[1])
#define S_CUR [a0] "v"(f.a[0]), [a1] "v"(f.a[1]), [a2] "v"(f.a[2]), [a3] "v"(f.a[3]), [b0] "v"(f.b[0]), [b1] "v"(f.b[1]), [sc] "v"(sc), [pa] "v"(pa), [pb] "v"(pb)
struct SxFrag { i32x4 a[4], b[2]; };
typedef f32x16 SxAcc[4][2];
// k-step on fragments f, loading g for the next k-step from LDS byte addresses pa / pb.
// DMA: 0 = none; 1 = the stage's first k-step: the last two loads (2, 3) of the column-tile sequence
template <bool FIRST, int DMA>
__device__ __forceinline__ void sx_kstep(SxAcc& c, const SxFrag& f, SxFrag& g, unsigned pa, unsigned pb, int sc, XDma& db) {
    if (DMA == 1 && FIRST)
        asm volatile(S_KSTEP(S_MZ, X_DM(vE, b), X_DM(vO, b), , )
                     : S_ACC_W(0), S_ACC_W(1), S_ACC_W(2), S_ACC_W(3), S_NEXT, X_DMA_OUT(b, db) : S_CUR, X_DMA_IN(b, db) : "memory", "scc");
    else if (DMA == 1)
        asm volatile(S_KSTEP(S_MF, X_DM(vE, b), X_DM(vO, b), , )
                     : S_ACC_RW(0), S_ACC_RW(1), S_ACC_RW(2), S_ACC_RW(3), S_NEXT, X_DMA_OUT(b, db) : S_CUR, X_DMA_IN(b, db) : "memory", "scc");
    else
        asm volatile(S_KSTEP(S_MF, , , , ) : S_ACC_RW(0), S_ACC_RW(1), S_ACC_RW(2), S_ACC_RW(3), S_NEXT : S_CUR : "memory");
    if (DMA) sx_uniform(db);
}
__device__ __forceinline__ void sx_klast(SxAcc& c, const SxFrag& f, SxFrag& g, unsigned pa, unsigned pb, int sc, XDma& da, XDma& db) {
    asm volatile(S_KLAST : S_ACC_RW(0), S_ACC_RW(1), S_ACC_RW(2), S_ACC_RW(3), S_NEXT, X_DMA_OUT(a, da), X_DMA_OUT(b, db) : S_CUR, X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    sx_uniform(da);
    sx_uniform(db);
}
__device__ __forceinline__ void sx_prologue(SxFrag& g, unsigned pa, unsigned pb) {
    asm volatile(X_LD(x0, pa, 0) X_LD(y0, pb, 0) X_LD(x1, pa, 4096) X_LD(y1, pb, 4096) X_LD(x2, pa, 8192) X_LD(x3, pa, 12288) X_WT(0)
                 : S_NEXT : [pa] "v"(pa), [pb] "v"(pb) : "memory");
}
__device__ __forceinline__ void sx_dma6(XDma& da, XDma& db) {  // pipeline fill: what the last k-step issues behind its barrier
    asm volatile(X_DM(vE, a) X_DM(vO, a) X_DM(vE, a) X_DM(vO, a) X_DM(vE, b) X_DM(vO, b) : X_DMA_OUT(a, da), X_DMA_OUT(b, db) : X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    sx_uniform(da);
    sx_uniform(db);
}
__global__ __launch_bounds__(512, 2) void k_syrk_f4p(const uint8_t* __restrict__ M4, long ld4, const int* __restrict__ pairs, int npairs,
                                                     int nblocks, long nstages, long stages_per_split, int32_t* __restrict__ C, long ldc) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][TILE_BYTES];
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (lid >= nblocks) return;
    const int split = __builtin_amdgcn_readfirstlane(lid / npairs);  // (the division runs on the vector unit)
    const int pr = __builtin_amdgcn_readfirstlane(pairs[lid - split * npairs]);
    const int ti = pr >> 16, tj = pr & 0xffff;
    const int s0 = split * (int)stages_per_split;  // stage counters fit an int (ld4 * 256 < 2^31); 64-bit compares would run on the vector unit
    int s1 = s0 + (int)stages_per_split;
    if (s1 > (int)nstages) s1 = (int)nstages;
    if (s0 >= s1) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int ldi = (int)ld4;
    const T8Lane ln = t8_lane(lane, ldi);
    const int8_t* baseA = (const int8_t*)M4 + (long)ti * T8 * ld4;
    const int8_t* baseB = (const int8_t*)M4 + (long)tj * T8 * ld4;
    t8_stage(t8_rsrc(baseA, ldi), ln, ldi, s0 * BK8, lds[0][0], w);
    t8_stage(t8_rsrc(baseB, ldi), ln, ldi, s0 * BK8, lds[0][1], w);
    __syncthreads();
    const T8Read rd = t8_read_init(wr, wc, lane);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) int8_t*)&lds[0][0][0];
    const unsigned offA = lds0 + rd.offA, offB = lds0 + TILE_BYTES + rd.offB;
    constexpr unsigned STG = 2 * TILE_BYTES;
    XDma dA, dB;
    dA.st = dB.st = __builtin_amdgcn_readfirstlane(8 * ldi);
    dA.vE = dB.vE = ln.voffE; dA.vO = dB.vO = ln.voffO;
    int snext = s0 + 1;  // the next stage to fetch
    unsigned nrec = 0;
    // (the descriptors are rebuilt from scalars at each use: a loop-carried SGPR vector ends up in VGPRs)
    auto rs_fresh = [&] { dA.rs = x_rsrc(baseA, nrec); dB.rs = x_rsrc(baseB, nrec); };
    auto dma_arm = [&](int into) {
        const unsigned base = lds0 + into * STG + (w * 4) * 1024 - 1024;
        dA.m0 = __builtin_amdgcn_readfirstlane(base); dB.m0 = dA.m0 + TILE_BYTES;
        dA.so = dB.so = __builtin_amdgcn_readfirstlane((unsigned)((w * 4) * 8 * ldi + snext * BK8 - 8 * ldi));
        // num_records = 0 when nothing is left to fetch: the loads then write zeros nobody reads
        const int left = __builtin_amdgcn_readfirstlane(s1 - snext);
        nrec = __builtin_amdgcn_readfirstlane((unsigned)(T8 * ldi) * (unsigned)max(min(left, 1), 0));
        snext++;
    };
    dma_arm(1);
    rs_fresh();
    sx_dma6(dA, dB);
    const int sc = 0x7f7f7f7f;
    SxFrag f0, f1;
    SxAcc c;
    sx_prologue(f0, offA + rd.ch[0], offB + rd.ch[0]);
    int buf = 0;
    auto stage_rest = [&] {
        const unsigned sa = offA + buf * STG, sb = offB + buf * STG;
        sx_kstep<false, 0>(c, f1, f0, sa + rd.ch[2], sb + rd.ch[2], sc, dB);
        sx_kstep<false, 0>(c, f0, f1, sa + rd.ch[3], sb + rd.ch[3], sc, dB);
        dma_arm(buf);
        rs_fresh();
        buf ^= 1;
        sx_klast(c, f1, f0, offA + buf * STG + rd.ch[0], offB + buf * STG + rd.ch[0], sc, dA, dB);
    };
    rs_fresh();
    sx_kstep<true, 1>(c, f0, f1, offA + rd.ch[1], offB + rd.ch[1], sc, dB);
    stage_rest();
    for (int s = s0 + 1; s < s1; s++) {
        rs_fresh();
        sx_kstep<false, 1>(c, f0, f1, offA + buf * STG + rd.ch[1], offB + buf * STG + rd.ch[1], sc, dB);
        stage_rest();
    }
    // the MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire; the stale re-loads have to land before
    // their registers are reused
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const int col = lane & 31, rq = 4 * (lane >> 5);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) {
                long i = (long)ti * T8 + wr * 128 + m * 32 + (x & 3) + 8 * (x >> 2) + rq;
                long j = (long)tj * T8 + wc * 64 + n * 32 + col;
                int v = (int)c[m][n][x];
                if (v) atomicAdd(&C[i * ldc + j], v);
            }
}

// ------------------------------------------------------------------------------------------------
// k_syrk_f4w: MM^T on 384 x 256 tiles (row tile I of 384 individuals x column tile J of 256), the tile shape and the pipelined
// k-step of k_vara_i8p with the fp4 x fp4 instruction.  The 256 x 256 kernel is bound by the L2 -> LDS fill, not by the matrix unit
// (a stage's 64 KiB take longer to arrive than its 32 MFMAs per wave take to run; tools/ubench: the bare loop holds 7 POP/s on
// genotype operands, the kernel 4.8): this tile moves 80 KiB per 48 MFMAs per wave, 17 % fewer bytes per MAC.  Rectangular tiles on
// a symmetric output: tile (I, J) is needed when it holds an element of a 256-block on or above the diagonal, i.e. 3 I / 2 <= J
// (2.4 % more MACs than the square tiling at n = 10,000); elements of blocks below the diagonal are not stored; the last row
// tile may be short (rows beyond n_pad read as zero and are not stored).  Same exact integers as the other forms: bit-identical C.
// ------------------------------------------------------------------------------------------------
typedef f32x16 WxAcc[3][4];
template <bool FIRST, int DMA>
__device__ __forceinline__ void wx_kstep(WxAcc& c, i32x4 (&a)[3], i32x4 (&an)[3], i32x4 (&b)[4], unsigned pa, unsigned pb, int sc, XDma& da, XDma& db) {
    if (DMA == 1 && FIRST)
        asm volatile(X_KSTEP(S_MZ, X_DM(vO, a), X_DM(vE, a), X_DM(vO, a), X_DM(vE, b), X_DM(vO, b), X_DM(vE, b))
                     : X_ACC_W(0), X_ACC_W(1), X_ACC_W(2), X_FRAGS, X_DMA_OUT(a, da), X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(sc), X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    else if (DMA == 1)
        asm volatile(X_KSTEP(S_MF, X_DM(vO, a), X_DM(vE, a), X_DM(vO, a), X_DM(vE, b), X_DM(vO, b), X_DM(vE, b))
                     : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(a, da), X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(sc), X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    else if (DMA == 2)
        asm volatile(X_KSTEP(S_MF, X_DM(vO, b), , , , , )
                     : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(sc), X_DMA_IN(b, db) : "memory", "scc");
    else
        asm volatile(X_KSTEP(S_MF, , , , , , ) : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS : [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(sc) : "memory");
    if (DMA == 1) sx_uniform(da);
    if (DMA) sx_uniform(db);
}
__device__ __forceinline__ void wx_klast(WxAcc& c, i32x4 (&a)[3], i32x4 (&an)[3], i32x4 (&b)[4], unsigned pa, unsigned pb, int sc, XDma& da) {
    asm volatile(X_KLAST_G(S_MF, X_DM(vE, a), X_DM(vO, a), X_DM(vE, a))
                 : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(a, da) : [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(sc), X_DMA_IN(a, da) : "memory", "scc");
    sx_uniform(da);
}
__global__ __launch_bounds__(512, 2) void k_syrk_f4w(const uint8_t* __restrict__ M4, long ld4, const int* __restrict__ pairs, int npairs,
                                                     int nblocks, long nstages, long stages_per_split, int32_t* __restrict__ C, long ldc, long n_pad) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ldsv[];  // [2][A 48 KiB | B 32 KiB]
    const int cpx = (gridDim.x + 7) / 8;
    const int lid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (lid >= nblocks) return;
    const int split = __builtin_amdgcn_readfirstlane(lid / npairs);
    const int pr = __builtin_amdgcn_readfirstlane(pairs[lid - split * npairs]);
    const int ti = pr >> 16, tj = pr & 0xffff;   // row tile of 384, column tile of 256
    const int s0 = split * (int)stages_per_split;
    int s1 = s0 + (int)stages_per_split;
    if (s1 > (int)nstages) s1 = (int)nstages;
    if (s0 >= s1) return;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;   // 4 x 2 waves, wave tile 96 x 128
    const int ldi = (int)ld4;
    const T8Lane ln = t8_lane(lane, ldi);
    const int8_t* baseA = (const int8_t*)M4 + (long)ti * TW_M * ld4;
    const int8_t* baseB = (const int8_t*)M4 + (long)tj * T8 * ld4;
    const long rows_left = n_pad - (long)ti * TW_M;
    const int rows_here = __builtin_amdgcn_readfirstlane((int)(rows_left < TW_M ? rows_left : TW_M));
    tw_stage<6>(__builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, rows_here * ldi, 0x00020000), ln, ldi, s0 * BK8, ldsv, w);
    tw_stage<4>(t8_rsrc(baseB, ldi), ln, ldi, s0 * BK8, ldsv + TW_ABYTES, w);
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
    int snext = s0 + 1;  // the next stage to fetch
    unsigned nrecA = 0, nrecB = 0;
    // (the descriptors are rebuilt from scalars at each use: a loop-carried SGPR vector ends up in VGPRs)
    auto rs_fresh = [&] { dA.rs = x_rsrc(baseA, nrecA); dB.rs = x_rsrc(baseB, nrecB); };
    auto dma_arm = [&](int into) {
        const unsigned base = lds0 + into * STG;
        dA.m0 = __builtin_amdgcn_readfirstlane(base + (w * 6) * 1024 - 1024);
        dB.m0 = __builtin_amdgcn_readfirstlane(base + TW_ABYTES + (w * 4) * 1024 - 1024);
        dA.so = __builtin_amdgcn_readfirstlane((unsigned)((w * 6) * 8 * ldi + snext * BK8 - 8 * ldi));
        dB.so = __builtin_amdgcn_readfirstlane((unsigned)((w * 4) * 8 * ldi + snext * BK8 - 8 * ldi));
        // num_records = 0 when nothing is left to fetch: the loads then write zeros nobody reads
        const int left = __builtin_amdgcn_readfirstlane(s1 - snext);
        const unsigned on = (unsigned)max(min(left, 1), 0);
        nrecA = __builtin_amdgcn_readfirstlane((unsigned)(rows_here * ldi) * on);
        nrecB = __builtin_amdgcn_readfirstlane((unsigned)(T8 * ldi) * on);
        snext++;
    };
    dma_arm(1);
    rs_fresh();
    tx_dma3(dA);
    sx_uniform(dA);
    const int sc = 0x7f7f7f7f;
    int buf = 0;
    i32x4 fa[2][3], fb[4];
    WxAcc c;
    tx_prologue(fa[0], fb, offA + ch[0], offB + ch[0]);
    auto stage_rest = [&] {
        const unsigned sa = offA + buf * STG, sb = offB + buf * STG;
        rs_fresh();
        wx_kstep<false, 2>(c, fa[1], fa[0], fb, sa + ch[2], sb + ch[2], sc, dA, dB);
        wx_kstep<false, 0>(c, fa[0], fa[1], fb, sa + ch[3], sb + ch[3], sc, dA, dB);
        dma_arm(buf);
        rs_fresh();
        buf ^= 1;
        wx_klast(c, fa[1], fa[0], fb, offA + buf * STG + ch[0], offB + buf * STG + ch[0], sc, dA);
    };
    rs_fresh();
    wx_kstep<true, 1>(c, fa[0], fa[1], fb, offA + ch[1], offB + ch[1], sc, dA, dB);
    stage_rest();
    for (int s = s0 + 1; s < s1; s++) {
        rs_fresh();
        wx_kstep<false, 1>(c, fa[0], fa[1], fb, offA + buf * STG + ch[1], offB + buf * STG + ch[1], sc, dA, dB);
        stage_rest();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // element (i, j): lane = column j, register = row i; only 256-blocks on or above the diagonal are live in C
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int x = 0; x < 16; x++) {
                const long i = (long)ti * TW_M + wr * 96 + m * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                const long j = (long)tj * T8 + wc * 128 + n * 32 + r;
                const int v = (int)c[m][n][x];
                if (v && i < n_pad && (i >> 8) <= (j >> 8)) atomicAdd(&C[i * ldc + j], v);
            }
}
// pair table of the 384 x 256 tiling: (I << 16) | J for 3 I / 2 <= J, in super-tiles of 4 row tiles x 8 column tiles (32 workgroups
// = what an XCD runs at a time share 4 row panels + 8 column panels)
static int syrk_pair_table_w(eagle_ctx* ctx, int nti, int ntj, const int** out, long* npairs) {
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::map<std::pair<int, int>, std::pair<int*, long>> tables;
    auto key = std::make_pair(dev, ntj);
    auto it = tables.find(key);
    if (it != tables.end()) { *out = it->second.first; *npairs = it->second.second; return EAGLE_OK; }
    std::vector<int> h;
    for (int si = 0; si < nti; si += 4)
        for (int sj = (3 * si / 2) / 8 * 8; sj < ntj; sj += 8)
            for (int i = si; i < si + 4 && i < nti; i++)
                for (int j = sj; j < sj + 8 && j < ntj; j++)
                    if (3 * i / 2 <= j) h.push_back((i << 16) | j);
    int* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, h.size() * sizeof(int));
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "pair table alloc");
    e = hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return eagle_fail_hip(ctx, e, "pair table copy"); }
    tables[key] = std::make_pair(d, (long)h.size());
    *out = d;
    *npairs = (long)h.size();
    return EAGLE_OK;
}

static long ctx_cu_count(eagle_ctx* ctx) {
    int cu = 0;
    (void)eagle_device_info(ctx, nullptr, 0, &cu, nullptr);
    return cu > 0 ? cu : 256;
}
// C32[np][np] += M M^T over the marker columns [0, L_pad) of the fp4 image M4[n_pad][ld4 bytes] (eagle_dev_pack_fp4 of M8).
extern "C" int eagle_dev_mmt_accumulate_f4(eagle_ctx* ctx, const void* M4, long n_pad, long L_pad, long ld4, int32_t* C32, void* stream) {
    if (n_pad % T8 || L_pad % 256 || ld4 % 128 || L_pad > ld4 * 2 || n_pad <= 0 || (double)ld4 * T8 >= 2147483648.0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_accumulate_f4: layout contract violated (n_pad % 256, L_pad % 256, ld4 % 128)");
    if (L_pad == 0) return EAGLE_OK;
    const int nt = (int)(n_pad / T8);
    const long nstages = L_pad / 256;
    // tune 9: k_syrk_f4 (256 x 256 tiles, compiler-scheduled), 10: k_syrk_f4p (256 x 256, pipelined k-step), else k_syrk_f4w (384 x 256)
    // (below ~3,000 individuals the 384-row tiles pad too much and leave too few workgroups: the 256 x 256 form is as fast or faster)
    const bool wide = ctx->tune != 9 && ctx->tune != 10 && (n_pad >= 3072 || ctx->tune == 11) && (double)ld4 * TW_M < 2147483648.0;
    const int* pairs = nullptr;
    long npairs = (long)nt * (nt + 1) / 2;
    int rc;
    if (wide) rc = syrk_pair_table_w(ctx, (int)((n_pad + TW_M - 1) / TW_M), nt, &pairs, &npairs);
    else rc = syrk_pair_table(ctx, nt, &pairs, (hipStream_t)stream);
    if (rc) return rc;
    long want = (10L * 256 + npairs - 1) / npairs;
    long maxsplit = nstages / 16 > 0 ? nstages / 16 : 1;
    long nsplit = want < maxsplit ? want : maxsplit;
    if (nsplit < 1) nsplit = 1;
    {   // one workgroup per CU: among the split counts near `want`, take the one whose last wave of workgroups is fullest
        const long slots = ctx_cu_count(ctx);
        double best = 1e30;
        long pick = nsplit;
        for (long sp = nsplit > 4 ? nsplit - 4 : 1; sp <= nsplit + 4 && sp <= maxsplit; sp++) {
            const long B = npairs * sp;
            const double waste = (double)((B + slots - 1) / slots * slots) / (double)B;
            if (waste < best - 1e-9) { best = waste; pick = sp; }
        }
        nsplit = pick;
    }
    long per = (nstages + nsplit - 1) / nsplit;
    if (per * 256 >= (1L << 24)) per = (1L << 24) / 256 - 1;  // fp32 partial sums stay exact integers
    nsplit = (nstages + per - 1) / per;
    const long nblocks = npairs * nsplit;
    if (nblocks >= (1L << 30)) return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_accumulate_f4: too many workgroups");
    dim3 grid((unsigned)((nblocks + 7) / 8 * 8));
    if (wide) {
        if (!ctx->attr_syrk_f4w) {
            hipError_t ea = hipFuncSetAttribute((const void*)k_syrk_f4w, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TW_ABYTES + TILE_BYTES));
            if (ea != hipSuccess) return eagle_fail_hip(ctx, ea, "hipFuncSetAttribute(k_syrk_f4w)");
            ctx->attr_syrk_f4w = true;
        }
        hipLaunchKernelGGL(k_syrk_f4w, grid, dim3(512), 2 * (TW_ABYTES + TILE_BYTES), (hipStream_t)stream, (const uint8_t*)M4, ld4, pairs, (int)npairs, (int)nblocks, nstages, per, C32, n_pad, n_pad);
    } else if (ctx->tune == 9) hipLaunchKernelGGL(k_syrk_f4, grid, dim3(512), 0, (hipStream_t)stream, (const uint8_t*)M4, ld4, pairs, (int)npairs, (int)nblocks, nstages, per, C32, n_pad);
    else hipLaunchKernelGGL(k_syrk_f4p, grid, dim3(512), 0, (hipStream_t)stream, (const uint8_t*)M4, ld4, pairs, (int)npairs, (int)nblocks, nstages, per, C32, n_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return eagle_fail_hip(ctx, e, "k_syrk_f4");
    return EAGLE_OK;
}

// Public int8 entry: packs the int8 image to fp4 (one extra read of it, 0.3 bytes written per genotype byte) into a
// ctx-owned buffer and runs k_syrk_f4.  Callers that keep the fp4 image themselves use eagle_dev_mmt_accumulate_f4.
extern "C" int eagle_dev_mmt_accumulate(eagle_ctx* ctx, const int8_t* M8, long n_pad, long L_pad, long ld, int32_t* C32, void* stream) {
    if (n_pad % T8 || L_pad % 256 || ld % 16 || L_pad > ld || n_pad <= 0)
        return eagle_fail(ctx, EAGLE_ERR_ARG, "mmt_accumulate: layout contract violated (n_pad % 256, L_pad % 256, ld % 16)");
    if (L_pad == 0) return EAGLE_OK;
    void* buf = eagle_ctx_f4_buffer(ctx, (size_t)n_pad * (size_t)(L_pad / 2));
    if (!buf) return EAGLE_ERR_HIP;
    int rc = eagle_dev_pack_fp4(ctx, M8, n_pad, L_pad, ld, buf, stream);
    if (rc) return rc;
    return eagle_dev_mmt_accumulate_f4(ctx, buf, n_pad, L_pad, L_pad / 2, C32, stream);
}
