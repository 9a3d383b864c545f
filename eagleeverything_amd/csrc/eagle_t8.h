// eagle_t8.h -- the int8 MFMA tile engine shared by eagle_i8mfma.hip and eagle_w8.hip (see the head of eagle_i8mfma.hip).
#ifndef EAGLE_T8_H
#define EAGLE_T8_H
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define T8 256          /* block tile (rows of A, rows of B) */
#define BK8 128         /* K bytes per stage */
#define TILE_BYTES (T8 * BK8)

// Per-lane constants of the tile engine.
//  DMA source: group g = 4w+i covers rows 8g .. 8g+7; lane l writes LDS bytes [1024 g + 16 l, +16) = row 8g + (l>>3),
//  physical chunk l&7, which must hold logical chunk (l&7) ^ ((row>>1)&7) = (l&7) ^ ((l>>4) + 4(i&1)).  So the per-lane
//  byte offset is voffE for even i and voffE ^ 64 for odd i (ld % 128 == 0), everything else is wave-uniform and goes
//  into the scalar offset of a buffer_load ... lds.
//  Fragment read: lane (r = l&31, h = l>>5) reads logical chunk 2ks+h of row R+r at physical chunk (2ks+h) ^ ((r>>1)&7).
struct T8Lane {
    int voffE, voffO;  // DMA source offsets (bytes) for even / odd row groups
};
__device__ __forceinline__ T8Lane t8_lane(int lane, int ld) {
    T8Lane x;
    x.voffE = (lane >> 3) * ld + (((lane & 7) ^ (lane >> 4)) << 4);
    x.voffO = x.voffE ^ 64;
    return x;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t t8_rsrc(const int8_t* base, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, T8 * ld, 0x00020000);
}
// One operand tile (256 rows x 128 B at byte column k0): 32 groups of 8 rows; wave w issues groups 4w .. 4w+3.
__device__ __forceinline__ void t8_stage(__amdgpu_buffer_rsrc_t rs, const T8Lane& ln, int ld, int k0, int8_t* ldsTile, int w) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int grp = w * 4 + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsTile + grp * 1024), 16,
                                                 (i & 1) ? ln.voffO : ln.voffE, grp * 8 * ld + k0, 0, 0);
    }
}

// One k-step (32 bytes of K) of the wave tile: 6 fragment reads + 8 MFMAs.
__device__ __forceinline__ void t8_kstep(i32x16 (&acc)[4][2], const int8_t* pa, const int8_t* pb, int ch) {
    i32x4 a[4], b[2];
#pragma unroll
    for (int m = 0; m < 4; m++) a[m] = *(const i32x4*)(pa + m * (32 * BK8) + ch);
#pragma unroll
    for (int n = 0; n < 2; n++) b[n] = *(const i32x4*)(pb + n * (32 * BK8) + ch);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
}
struct T8Read {  // per-lane LDS read bases and the 4 swizzled chunk offsets of a stage
    int offA, offB, ch[4];
};
__device__ __forceinline__ T8Read t8_read_init(int wr, int wc, int lane) {
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    T8Read x;
    x.offA = wr * (128 * BK8) + r * BK8;
    x.offB = wc * (64 * BK8) + r * BK8;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) x.ch[ks] = ((2 * ks + h) ^ swz) << 4;
    return x;
}
__device__ __forceinline__ void t8_compute(i32x16 (&acc)[4][2], const int8_t* ldsA, const int8_t* ldsB, int wr, int wc,
                                           int lane) {
    const T8Read rd = t8_read_init(wr, wc, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) t8_kstep(acc, ldsA + rd.offA, ldsB + rd.offB, rd.ch[ks]);
}

// One pipeline stage: DMA the next stage into (nA, nB) if `more`, then the 4 k-steps of the current stage.
// TUNE 3 / 4 are ablations for tools/bench_i8_engine.py (3: no DMA, 4: DMA + one k-step); results are wrong there.
// Measured on the MM^T SYRK (n = 5000, L = 262144, same process, interleaved): full 2.93 ms, no-DMA 2.32 ms, DMA-only
// 2.78 ms: the kernel is bound by the L2 -> LDS fill rate (~40 GB/s per CU at a 70-80 % L2 hit rate), not by MFMA issue.
// Tried and NOT faster (kept out of the code): staggering the DMA issue of waves 4-7 by half a stage (-13 %), spreading
// the DMA pieces over the k-steps (+-2 %), software-pipelined fragment reads pinned with sched_group_barrier (-3 %),
// a 10-slot 160 KiB LDS ring with 96 KiB in flight and counted vmcnt (-3 %; DMA-only 2.55 ms), a 4-deep ring of 64-byte
// K stages (-9 %), tile-major pre-swizzled operand copies so that every DMA instruction reads 1 KiB of consecutive
// bytes (+-1 %), odd leading dimensions against channel aliasing (+-1 %), a descending K order for every second tile so
// that each sweep of the genotype panel starts on the lines the previous sweep left in L2 (0 %), and L2 prefetch of the
// stage wanted 3 stages ahead by one designated leader per sharing group (-10 %: the leader's in-order vmcnt wait now
// includes its own HBM-latency loads and it becomes the straggler of its group), and the v_mfma_i32_16x16x64_i8 shape
// (8 x 4 tiles per wave; +7 % in the stand-alone tools/ubench/tile_geom.hip where the chip gives the clock back, -2.5 %
// here: 5.66 vs 5.52 ms on the C2 SYRK).  Last, 2-bit packed genotype operands expanded in registers (perm LUT, 11 VALU
// per 16 genotypes) and written to the same LDS image by ds_write_b128, which cuts the L2 -> CU bytes of the SYRK 4x:
// 5.39 vs 5.41 ms, bit-identical result.  With the fill bytes quartered and the time unchanged the fill rate is not the
// whole story either: every variant lands on the same ~2.5 POP/s, where MFMA-busy x clock is what the chip sustains on
// random int8 operands (1.9 GHz at 55 % busy here; the guide's LDS-read + MFMA loops hold 1.5-1.7 GHz when denser).
// Splitting a worker's column-tile pairs over 2 / 5 workgroups (fewer marker tiles resident per XCD, so that their genotype
// panels stay in L2 across the column tiles): -1 % / -7 % (22.6 -> 22.9 / 24.2 ms), the W-digit tiles then have fewer sharers.
// tools/ubench/fill_rate.hip measures what bounds it: filling 64 KiB of LDS takes 1.15-1.2 us per CU when every line
// is an L2 hit and 3.1 us when every line comes from HBM, by LDS-DMA and by register staging alike; at the 72-82 %
// hit rate of these kernels (rocprofv3 TCC_HIT/TCC_REQ) that is 1.5-1.7 us per stage against 0.9-1.1 us of MFMA work.
template <int TUNE>
__device__ __forceinline__ void t8_stage_compute(i32x16 (&acc)[4][2], const int8_t* ldsA, const int8_t* ldsB, const T8Read& rd,
                                                 bool more, __amdgpu_buffer_rsrc_t rsA, const T8Lane& lnA, int ldA, int kA,
                                                 int8_t* nA, __amdgpu_buffer_rsrc_t rsB, const T8Lane& lnB, int ldB, int kB,
                                                 int8_t* nB, int w) {
    const int8_t* pa = ldsA + rd.offA;
    const int8_t* pb = ldsB + rd.offB;
    if (TUNE != 3 && more) { t8_stage(rsA, lnA, ldA, kA, nA, w); t8_stage(rsB, lnB, ldB, kB, nB, w); }
#pragma unroll
    for (int ks = 0; ks < (TUNE == 4 ? 1 : 4); ks++) t8_kstep(acc, pa, pb, rd.ch[ks]);
}

__device__ __forceinline__ void t8_zero(i32x16 (&acc)[4][2]) {
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[m][n][q] = 0;
}
// Scale exponent of the digits: max|Wu_jk| (j != k) < 2^e, lowered by one when the mantissa leaves room (round 3) -- a balanced S-digit
// number reaches 127 (256^S - 1)/255 = 0.498 * 256^S, and |Q| <= max|Wu| 2^(8S-e-2) + 1 stays below 0.49 * 256^S + 1 for a mantissa up
// to 0.98: one more bit of resolution for the same digits on 96 % of all scales (rounds 1-2 always kept the leading digit inside
// [-65, 65]).  A power of two, so that Wu * 2^(8S-e-2) is exact and llrint rounds the true value.
__device__ __forceinline__ int w_scale_exp(double mx) {
    int e = 0;
    if (mx > 0.0) {
        const double f = frexp(mx, &e);
        if (f <= 0.98) e -= 1;
    }
    return e;
}


// ---- the 384 x 256 tile and its asm-pipelined k-step (k_vara_i8p, k_syrk_f4w, k_w8_gemm) ----
#define TW_M 384
#define TW_ABYTES (TW_M * BK8)
__device__ __forceinline__ void tw_kstep(i32x16 (&acc)[3][4], const int8_t* pa, const int8_t* pb, int ch) {
    i32x4 a[3], b[4];
#pragma unroll
    for (int m = 0; m < 3; m++) a[m] = *(const i32x4*)(pa + m * (32 * BK8) + ch);
#pragma unroll
    for (int n = 0; n < 4; n++) b[n] = *(const i32x4*)(pb + n * (32 * BK8) + ch);
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
}
// vara_tail_pieces (pieces per worker of an XCD's last, partly filled round): eagle_host.h
// `groups` row groups (8 rows each) of an operand tile per wave: wave w issues groups w*groups .. (groups is even)
template <int GROUPS>
__device__ __forceinline__ void tw_stage(__amdgpu_buffer_rsrc_t rs, const T8Lane& ln, int ld, int k0, int8_t* ldsTile, int w) {
#pragma unroll
    for (int i = 0; i < GROUPS; i++) {
        const int grp = w * GROUPS + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsTile + grp * 1024), 16,
                                                 (i & 1) ? ln.voffO : ln.voffE, grp * 8 * ld + k0, 0, 0);
    }
}
// SrcA = the W-digit fragment b, SrcB = the genotype fragment a: the matrix unit draws far less power when its SrcB operand is the
// low-entropy one (tools/ubench/mfma_ceiling.hip: a bare loop on these operand statistics holds 3.75 POP/s this way round,
// 3.29 POP/s the other), and this kernel runs against the power limit.  The 32 x 32 result tiles come out transposed: lane =
// marker, register = W column.
#define X_MF(c, a, b) "v_mfma_i32_32x32x32_i8 %[" #c "], %[" #b "], %[" #a "], %[" #c "]\n\t"
#define X_MZ(c, a, b) "v_mfma_i32_32x32x32_i8 %[" #c "], %[" #b "], %[" #a "], 0\n\t"
#define X_LD(d, p, off) "ds_read_b128 %[" #d "], %[" #p "] offset:" #off "\n\t"
#define X_WT(n) "s_waitcnt lgkmcnt(" #n ")\n\t"
// one LDS-DMA load of the stage being fetched, operand set s (a = genotype rows, b = W-digit rows): next row group
// (LDS +1 KiB, source + 8 rows), even / odd group lane offsets
// (the asm statements that use it declare the "scc" clobber: s_add_u32 writes SCC, and hipcc keeps compare results live across asm)
#define X_DM(vo, s) "s_add_u32 %[m0" #s "], %[m0" #s "], 0x400\n\ts_mov_b32 m0, %[m0" #s "]\n\ts_add_u32 %[so" #s "], %[so" #s "], %[st" #s "]\n\t" \
                    "buffer_load_dwordx4 %[" #vo #s "], %[rs" #s "], %[so" #s "] offen lds\n\t"
#define X_NO(vo, s)
// k-steps 0-2 (M = X_MF or X_MZ; D1-D6: DMA slots behind every second MFMA): queue on entry an0 an1 b0 an2 b1 b2 b3
// (as a0-a2 here), loads x0-x2, b0-b3
#define X_KSTEP(M, D1, D2, D3, D4, D5, D6)                                                                  \
    X_WT(4) M(c00, a0, b0) X_LD(x0, pa, 0) M(c10, a1, b0) X_LD(x1, pa, 4096) D1                             \
    X_WT(5) M(c20, a2, b0) X_LD(b0, pb, 0)                                                                  \
    X_WT(5) M(c01, a0, b1) X_LD(x2, pa, 8192) D2 M(c11, a1, b1) M(c21, a2, b1) X_LD(b1, pb, 4096) D3        \
    X_WT(6) M(c02, a0, b2) M(c12, a1, b2) D4 M(c22, a2, b2) X_LD(b2, pb, 8192)                              \
    X_WT(6) M(c03, a0, b3) D5 M(c13, a1, b3) M(c23, a2, b3) X_LD(b3, pb, 12288) D6
// the stage's last k-step: no loads before the barrier; behind it the loads of the next stage's first k-step in queue order
// and the first three DMA loads of the stage after next
#define X_KLAST_G(M, DA1, DA2, DA3)                                                                          \
    X_WT(4) M(c00, a0, b0) M(c10, a1, b0) X_WT(3) M(c20, a2, b0)                                            \
    X_WT(2) M(c01, a0, b1) M(c11, a1, b1) M(c21, a2, b1)                                                    \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t"                                                        \
    X_LD(x0, pa, 0) X_LD(x1, pa, 4096) X_LD(b0, pb, 0) X_LD(x2, pa, 8192) X_LD(b1, pb, 4096)                \
    M(c02, a0, b2) M(c12, a1, b2) DA1 M(c22, a2, b2) X_LD(b2, pb, 8192)                                     \
    M(c03, a0, b3) DA2 M(c13, a1, b3) M(c23, a2, b3) X_LD(b3, pb, 12288) DA3
#define X_KLAST X_KLAST_G(X_MF, X_DM(vE, a), X_DM(vO, a), X_DM(vE, a))
#define X_ACC_RW(m) [c##m##0] "+v"(c[m][0]), [c##m##1] "+v"(c[m][1]), [c##m##2] "+v"(c[m][2]), [c##m##3] "+v"(c[m][3])
#define X_ACC_W(m) [c##m##0] "=&v"(c[m][0]), [c##m##1] "=&v"(c[m][1]), [c##m##2] "=&v"(c[m][2]), [c##m##3] "=&v"(c[m][3])
#define X_FRAGS [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [x0] "=&v"(an[0]), [x1] "=&v"(an[1]), [x2] "=&v"(an[2]), \
                [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3])
#define X_DMA_OUT(s, d) [m0##s] "+s"(d.m0), [so##s] "+s"(d.so)
#define X_DMA_IN(s, d) [st##s] "s"(d.st), [rs##s] "s"(d.rs), [vE##s] "v"(d.vE), [vO##s] "v"(d.vO)
typedef i32x16 TxAcc[3][4];
// state of one operand's DMA sequence: LDS address of the last issued row group (m0), its source offset (so), the stride of a
// row group in the source (st = 8 rows), the buffer descriptor (num_records = 0 when there is nothing left to fetch: the loads
// then write zeros into a buffer nobody reads again) and the even / odd lane offsets
struct XDma { unsigned m0, so, st; i32x4 rs; int vE, vO; };
__device__ __forceinline__ i32x4 x_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void tx_prologue(i32x4 (&an)[3], i32x4 (&b)[4], unsigned pa, unsigned pb) {
    asm volatile(X_LD(x0, pa, 0) X_LD(x1, pa, 4096) X_LD(b0, pb, 0) X_LD(x2, pa, 8192) X_LD(b1, pb, 4096) X_LD(b2, pb, 8192) X_LD(b3, pb, 12288) X_WT(0)
                 : [x0] "=&v"(an[0]), [x1] "=&v"(an[1]), [x2] "=&v"(an[2]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3])
                 : [pa] "v"(pa), [pb] "v"(pb) : "memory");
}
__device__ __forceinline__ void tx_dma3(XDma& da) {  // three loads on their own (pipeline fill)
    asm volatile(X_DM(vE, a) X_DM(vO, a) X_DM(vE, a) : X_DMA_OUT(a, da) : X_DMA_IN(a, da) : "memory", "scc");
}
// k-step on fragments a (genotype rows) / b, loading an / b for the next k-step from LDS byte addresses pa / pb.
// DMA: 0 = none; 1 = the stage's first k-step: loads 3-5 of the genotype sequence, 0-2 of the W-digit one; 2 = the second: the last
template <bool FIRST, int DMA>
__device__ __forceinline__ void tx_kstep(TxAcc& c, i32x4 (&a)[3], i32x4 (&an)[3], i32x4 (&b)[4], unsigned pa, unsigned pb, XDma& da, XDma& db) {
    if (DMA == 1 && FIRST)
        asm volatile(X_KSTEP(X_MZ, X_DM(vO, a), X_DM(vE, a), X_DM(vO, a), X_DM(vE, b), X_DM(vO, b), X_DM(vE, b))
                     : X_ACC_W(0), X_ACC_W(1), X_ACC_W(2), X_FRAGS, X_DMA_OUT(a, da), X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    else if (DMA == 1)
        asm volatile(X_KSTEP(X_MF, X_DM(vO, a), X_DM(vE, a), X_DM(vO, a), X_DM(vE, b), X_DM(vO, b), X_DM(vE, b))
                     : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(a, da), X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), X_DMA_IN(a, da), X_DMA_IN(b, db) : "memory", "scc");
    else if (DMA == 2)
        asm volatile(X_KSTEP(X_MF, X_DM(vO, b), , , , , )
                     : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(b, db) : [pa] "v"(pa), [pb] "v"(pb), X_DMA_IN(b, db) : "memory", "scc");
    else
        asm volatile(X_KSTEP(X_MF, , , , , , ) : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS : [pa] "v"(pa), [pb] "v"(pb) : "memory");
}
// last k-step of a stage; da: the genotype DMA sequence of the stage after next (armed before the call), three of its loads go out here
__device__ __forceinline__ void tx_klast(TxAcc& c, i32x4 (&a)[3], i32x4 (&an)[3], i32x4 (&b)[4], unsigned pa, unsigned pb, XDma& da) {
    asm volatile(X_KLAST : X_ACC_RW(0), X_ACC_RW(1), X_ACC_RW(2), X_FRAGS, X_DMA_OUT(a, da) : [pa] "v"(pa), [pb] "v"(pb), X_DMA_IN(a, da) : "memory", "scc");
}
// hipcc's uniformity analysis calls the scalar outputs of these asm blocks divergent in this kernel (not in k_vara_i8p), puts their
// loop-carried copies into VGPRs and then cannot feed them to the next block's "s" operands: say it explicitly
__device__ __forceinline__ void sx_uniform(XDma& d) {
    d.m0 = __builtin_amdgcn_readfirstlane(d.m0);
    d.so = __builtin_amdgcn_readfirstlane(d.so);
}
#endif
