// Micro-benchmark: int8 NT tile product C[i][j] = sum_k A[i][k] B[j][k] on v_mfma_i32_32x32x32_i8 for several block
// geometries of the tile engine (LDS-DMA staging, XOR-swizzled 128-byte LDS rows, 2 stage buffers).
//   G0: 8 waves 2x4, wave tile 128x64  (block 256x256, 128 acc regs, 2 waves/SIMD)   -- the shipped engine
//   G1: 4 waves 2x2, wave tile 128x128 (block 256x256, 256 acc regs, 1 wave/SIMD)
//   G2: 4 waves 2x2, wave tile 128x192 (block 256x384, 384 acc regs, 1 wave/SIMD)
// Every block owns one output tile and runs the whole K; A and B panels are shared between blocks like in k_vara_i8
// (blocks with equal blockIdx.y share A, equal blockIdx.x share B).  Prints TOP/s and checks one tile on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define BK 128

template <int WAVES_M, int WAVES_N, int WM, int WN>
struct Geo {
    static constexpr int NW = WAVES_M * WAVES_N, TM = WAVES_M * WM * 32, TN = WAVES_N * WN * 32;
    static constexpr int A_BYTES = TM * BK, B_BYTES = TN * BK, STAGE = A_BYTES + B_BYTES;
};

template <class G>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, int voffE, int ld, int k0, int8_t* lds, int rows, int w) {
    const int groups = rows / 8, per = groups / G::NW;  // groups of 8 rows, split evenly over the waves
#pragma unroll
    for (int i = 0; i < per; i++) {
        const int grp = w * per + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + grp * 1024), 16,
                                                 (grp & 1) ? (voffE ^ 64) : voffE, grp * 8 * ld + k0, 0, 0);
    }
}

template <class G, int WM, int WN>
__global__ __launch_bounds__(G::NW * 64) void k_tile(const int8_t* __restrict__ A, const int8_t* __restrict__ B, long ld, int nstages,
                                                       int* __restrict__ C, long ldc, int wavesN) {
    extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w / wavesN, wc = w % wavesN;
    const int ldi = (int)ld;
    const int voffE = (lane >> 3) * ldi + (((lane & 7) ^ (lane >> 4)) << 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)blockIdx.y * G::TM * ld), 0, G::TM * ldi, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (long)blockIdx.x * G::TN * ld), 0, G::TN * ldi, 0x00020000);
    i32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int n = 0; n < WN; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[m][n][q] = 0;
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    stage_tile<G>(rsA, voffE, ldi, 0, lds, G::TM, w);
    stage_tile<G>(rsB, voffE, ldi, 0, lds + G::A_BYTES, G::TN, w);
    __syncthreads();
    int cur = 0;
    for (int st = 0; st < nstages; st++) {
        int8_t* nb = lds + (cur ^ 1) * G::STAGE;
        if (st + 1 < nstages) {
            stage_tile<G>(rsA, voffE, ldi, (st + 1) * BK, nb, G::TM, w);
            stage_tile<G>(rsB, voffE, ldi, (st + 1) * BK, nb + G::A_BYTES, G::TN, w);
        }
        const int8_t* pa = lds + cur * G::STAGE + (wr * WM * 32 + r) * BK;
        const int8_t* pb = lds + cur * G::STAGE + G::A_BYTES + (wc * WN * 32 + r) * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int ch = ((2 * ks + h) ^ swz) << 4;
            i32x4 a[WM], b[WN];
#pragma unroll
            for (int m = 0; m < WM; m++) a[m] = *(const i32x4*)(pa + m * 32 * BK + ch);
#pragma unroll
            for (int n = 0; n < WN; n++) b[n] = *(const i32x4*)(pb + n * 32 * BK + ch);
#pragma unroll
            for (int m = 0; m < WM; m++)
#pragma unroll
                for (int n = 0; n < WN; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }
    const int col = lane & 31, rq = 4 * (lane >> 5);
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int n = 0; n < WN; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                long i = (long)blockIdx.y * G::TM + (wr * WM + m) * 32 + (q & 3) + 8 * (q >> 2) + rq;
                long j = (long)blockIdx.x * G::TN + (wc * WN + n) * 32 + col;
                C[i * ldc + j] = acc[m][n][q];
            }
}



// G1p: 4 waves 2x2, wave tile 128x128 (256 accumulators, one wave per SIMD) with the fragments of k-step ks+1 loaded while
// the 16 MFMAs of k-step ks issue (register double buffering: the single wave of a SIMD has no partner to hide LDS latency).
template <class G>
__global__ __launch_bounds__(256) void k_tile_g1p(const int8_t* __restrict__ A, const int8_t* __restrict__ B, long ld, int nstages,
                                                   int* __restrict__ C, long ldc) {
    extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int ldi = (int)ld;
    const int voffE = (lane >> 3) * ldi + (((lane & 7) ^ (lane >> 4)) << 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)blockIdx.y * G::TM * ld), 0, G::TM * ldi, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (long)blockIdx.x * G::TN * ld), 0, G::TN * ldi, 0x00020000);
    i32x16 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[m][n][q] = 0;
    const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
    stage_tile<G>(rsA, voffE, ldi, 0, lds, G::TM, w);
    stage_tile<G>(rsB, voffE, ldi, 0, lds + G::A_BYTES, G::TN, w);
    __syncthreads();
    int cur = 0;
    i32x4 a[2][4], b[2][4];
    auto load_frags = [&](int slot, const int8_t* base, int ks) {
        const int ch = ((2 * ks + h) ^ swz) << 4;
        const int8_t* pa = base + (wr * 128 + r) * BK;
        const int8_t* pb = base + G::A_BYTES + (wc * 128 + r) * BK;
#pragma unroll
        for (int m = 0; m < 4; m++) a[slot][m] = *(const i32x4*)(pa + m * 32 * BK + ch);
#pragma unroll
        for (int n = 0; n < 4; n++) b[slot][n] = *(const i32x4*)(pb + n * 32 * BK + ch);
    };
    load_frags(0, lds, 0);
    for (int st = 0; st < nstages; st++) {
        int8_t* nb = lds + (cur ^ 1) * G::STAGE;
        if (st + 1 < nstages) {
            stage_tile<G>(rsA, voffE, ldi, (st + 1) * BK, nb, G::TM, w);
            stage_tile<G>(rsB, voffE, ldi, (st + 1) * BK, nb + G::A_BYTES, G::TN, w);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks < 3) load_frags((ks + 1) & 1, lds + cur * G::STAGE, ks + 1);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks & 1][m], b[ks & 1][n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
        if (st + 1 < nstages) load_frags(0, lds + cur * G::STAGE, 0);
    }
    const int col = lane & 31, rq = 4 * (lane >> 5);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                long i = (long)blockIdx.y * G::TM + (wr * 4 + m) * 32 + (q & 3) + 8 * (q >> 2) + rq;
                long j = (long)blockIdx.x * G::TN + (wc * 4 + n) * 32 + col;
                C[i * ldc + j] = acc[m][n][q];
            }
}

// Same engine on v_mfma_i32_16x16x64_i8: wave tile (WM*16) x (WN*16), two K=64 steps per 128-byte stage.
template <class G, int WM, int WN>
__global__ __launch_bounds__(G::NW * 64) void k_tile16(const int8_t* __restrict__ A, const int8_t* __restrict__ B, long ld, int nstages,
                                                         int* __restrict__ C, long ldc, int wavesN) {
    extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w / wavesN, wc = w % wavesN;
    const int ldi = (int)ld;
    const int voffE = (lane >> 3) * ldi + (((lane & 7) ^ (lane >> 4)) << 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)blockIdx.y * G::TM * ld), 0, G::TM * ldi, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (long)blockIdx.x * G::TN * ld), 0, G::TN * ldi, 0x00020000);
    i32x4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int n = 0; n < WN; n++) acc[m][n] = (i32x4){0, 0, 0, 0};
    const int r = lane & 15, q = lane >> 4, swz = (r >> 1) & 7;
    stage_tile<G>(rsA, voffE, ldi, 0, lds, G::TM, w);
    stage_tile<G>(rsB, voffE, ldi, 0, lds + G::A_BYTES, G::TN, w);
    __syncthreads();
    int cur = 0;
    for (int st = 0; st < nstages; st++) {
        int8_t* nb = lds + (cur ^ 1) * G::STAGE;
        if (st + 1 < nstages) {
            stage_tile<G>(rsA, voffE, ldi, (st + 1) * BK, nb, G::TM, w);
            stage_tile<G>(rsB, voffE, ldi, (st + 1) * BK, nb + G::A_BYTES, G::TN, w);
        }
        const int8_t* pa = lds + cur * G::STAGE + (wr * WM * 16 + r) * BK;
        const int8_t* pb = lds + cur * G::STAGE + G::A_BYTES + (wc * WN * 16 + r) * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int ch = ((4 * ks + q) ^ swz) << 4;
            i32x4 a[WM], b[WN];
#pragma unroll
            for (int m = 0; m < WM; m++) a[m] = *(const i32x4*)(pa + m * 16 * BK + ch);
#pragma unroll
            for (int n = 0; n < WN; n++) b[n] = *(const i32x4*)(pb + n * 16 * BK + ch);
#pragma unroll
            for (int m = 0; m < WM; m++)
#pragma unroll
                for (int n = 0; n < WN; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int n = 0; n < WN; n++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                long i = (long)blockIdx.y * G::TM + (wr * WM + m) * 16 + 4 * q + e;
                long j = (long)blockIdx.x * G::TN + (wc * WN + n) * 16 + r;
                C[i * ldc + j] = acc[m][n][e];
            }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SH16 = false>
static void run(const char* name, const int8_t* dA, const int8_t* dB, long ld, int K, int Mrows, int Nrows, int* dC, const std::vector<int8_t>& hA,
                const std::vector<int8_t>& hB) {
    using G = Geo<WAVES_M, WAVES_N, WM, WN>;
    const int gx = Nrows / G::TN, gy = Mrows / G::TM;
    const size_t ldsb = 2 * (size_t)G::STAGE;
    auto kern = SH16 ? k_tile16<G, WM * 2, WN * 2> : k_tile<G, WM, WN>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    const long ldc = (long)gx * G::TN;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(G::NW * 64), ldsb, 0, dA, dB, ld, K / BK, dC, ldc, WAVES_N);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // check a few entries on the host
    std::vector<int> hC((size_t)8 * ldc);
    CHECK(hipMemcpy(hC.data(), dC + (long)(gy * G::TM - 8) * ldc, hC.size() * sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int ii = 0; ii < 8; ii++)
        for (long j = 0; j < ldc; j += 97) {
            long i = (long)gy * G::TM - 8 + ii;
            int s = 0;
            for (int k = 0; k < K; k++) s += (int)hA[i * ld + k] * (int)hB[j * ld + k];
            if (s != hC[(size_t)ii * ldc + j]) bad++;
        }
    double ops = 2.0 * gy * G::TM * (double)gx * G::TN * K;
    printf("%s: tile %dx%d, %d waves, LDS %zu KiB, grid %dx%d: %.3f ms  %.0f TOP/s  (%s)\n", name, G::TM, G::TN, G::NW, ldsb / 1024, gx, gy, best,
           ops / best / 1e9, bad ? "WRONG" : "ok");
}

static void run_g1p(const int8_t* dA, const int8_t* dB, long ld, int K, int Mrows, int Nrows, int* dC, const std::vector<int8_t>& hA,
                    const std::vector<int8_t>& hB) {
    using G = Geo<2, 2, 4, 4>;
    const int gx = Nrows / G::TN, gy = Mrows / G::TM;
    const size_t ldsb = 2 * (size_t)G::STAGE;
    auto kern = k_tile_g1p<G>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    const long ldc = (long)gx * G::TN;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), ldsb, 0, dA, dB, ld, K / BK, dC, ldc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    std::vector<int> hC((size_t)8 * ldc);
    CHECK(hipMemcpy(hC.data(), dC + (long)(gy * G::TM - 8) * ldc, hC.size() * sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int ii = 0; ii < 8; ii++)
        for (long j = 0; j < ldc; j += 97) {
            long i = (long)gy * G::TM - 8 + ii;
            int s = 0;
            for (int k = 0; k < K; k++) s += (int)hA[i * ld + k] * (int)hB[j * ld + k];
            if (s != hC[(size_t)ii * ldc + j]) bad++;
        }
    double ops = 2.0 * gy * G::TM * (double)gx * G::TN * K;
    printf("G1p (register double-buffered fragments): %.3f ms  %.0f TOP/s  (%s)\n", best, ops / best / 1e9, bad ? "WRONG" : "ok");
}

int main() {
    const int K = 5120;
    const long ld = K;
    const int Mrows = 256 * 96, Nrows = 768 * 20;  // 24576 x 15360 outputs
    std::vector<int8_t> hA((size_t)Mrows * ld), hB((size_t)Nrows * ld);
    srand(1);
    for (auto& x : hA) x = (int8_t)(rand() % 3 - 1);
    for (auto& x : hB) x = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB; int* dC;
    CHECK(hipMalloc((void**)&dA, hA.size())); CHECK(hipMalloc((void**)&dB, hB.size()));
    CHECK(hipMalloc((void**)&dC, (size_t)Mrows * Nrows * sizeof(int)));
    CHECK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice));
    for (int round = 0; round < 2; round++) {
        run<2, 4, 4, 2>("G0", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
        run<2, 4, 4, 2, true>("G0/16x16x64", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
        run<2, 2, 4, 4>("G1", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
        run_g1p(dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
        run<2, 2, 4, 5>("G2b", dA, dB, ld, K, Mrows, 320 * 48, dC, hA, hB);
        run<2, 2, 4, 6>("G2", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
    }
    // genotype coding: HWE-like genotype frequencies (g=0: 55 %, g=1: 35 %, g=2: 10 %) as m = g-1 (0xFF common) vs as g (0x00 common)
    for (int coding = 0; coding < 2; coding++) {
        for (auto& x : hA) { int u = rand() % 100; int g = u < 55 ? 0 : (u < 90 ? 1 : 2); x = (int8_t)(coding ? g : g - 1); }
        CHECK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            run<2, 4, 4, 2>(coding ? "G0 A=g in {0,1,2}" : "G0 A=g-1 in {-1,0,1}", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
            run<2, 4, 4, 2, true>(coding ? "G0/16 A=g" : "G0/16 A=g-1", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
        }
    }
    // the same two on all-zero operands: the gap to the random-data time is clock the chip gives back under load
    CHECK(hipMemset(dA, 0, hA.size())); CHECK(hipMemset(dB, 0, hB.size()));
    std::fill(hA.begin(), hA.end(), 0); std::fill(hB.begin(), hB.end(), 0);
    run<2, 4, 4, 2>("G0 zeros", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
    run<2, 4, 4, 2, true>("G0/16x16x64 zeros", dA, dB, ld, K, Mrows, Nrows, dC, hA, hB);
    return 0;
}
