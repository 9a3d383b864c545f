// What the chip sustains on v_mfma_f64_16x16x4_f64 with nothing else going on: register-resident loops, every SIMD issuing back to
// back, random operands, ~0.2 s per case -- the ceiling the fp64 kernels of this library (k_gemm_f64_list8, k_vara_f64, k_zbuild)
// are to be measured against (the data sheet says 78.6 TFLOP/s = 128 flop/clk/CU at 2.4 GHz).
// Cases: waves per SIMD 1 / 2 / 4 (blocks of 256 / 512 threads, 1 or 2 per CU), accumulator tiles per wave 8 (2 x 4) or 16 (4 x 4).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/f64_ceiling.hip -o tools/ubench/f64_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// the same loop without the operand rotation (no VALU between the MFMAs at all)
template <int TM, int TN, int THREADS, int MINB>
__global__ __launch_bounds__(THREADS, MINB) void k_loop_fixed(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    double a[TM], b[TN];
    for (int m = 0; m < TM; m++) a[m] = A[(t + THREADS * m) & 4095];
    for (int n = 0; n < TN; n++) b[n] = B[(t + THREADS * n) & 4095];
    f64x4 c[TM][TN];
    for (int m = 0; m < TM; m++) for (int n = 0; n < TN; n++) c[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < TM; m++)
#pragma unroll
            for (int n = 0; n < TN; n++) c[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], c[m][n], 0, 0, 0);
    }
    double s = 0;
    for (int m = 0; m < TM; m++) for (int n = 0; n < TN; n++) for (int i = 0; i < 4; i++) s += c[m][n][i];
    out[(size_t)blockIdx.x * THREADS + t] = s;
}
template <int TM, int TN, int THREADS, int MINB>
static void run_fixed(const char* name, const double* dA, const double* dB, double* dout, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_loop_fixed<TM, TN, THREADS, MINB>), dim3(blocks), dim3(THREADS), 0, 0, dA, dB, dout, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) iters = (int)(iters * 200.0 / ms);
    }
    const double flop = 2.0 * 16 * 16 * 4 * TM * TN * (double)iters * (THREADS / 64) * blocks;
    printf("%-58s fixed  operands: %8.2f ms  %6.2f TFLOP/s = %.3f of 78.6\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
}

template <int TM, int TN, int THREADS, int MINB>
__global__ __launch_bounds__(THREADS, MINB) void k_loop(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    double a[TM], b[TN];
    for (int m = 0; m < TM; m++) a[m] = A[(t + THREADS * m) & 4095];
    for (int n = 0; n < TN; n++) b[n] = B[(t + THREADS * n) & 4095];
    f64x4 c[TM][TN];
    for (int m = 0; m < TM; m++) for (int n = 0; n < TN; n++) c[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < TM; m++)
#pragma unroll
            for (int n = 0; n < TN; n++) c[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], c[m][n], 0, 0, 0);
        const double ta = a[0];
#pragma unroll
        for (int m = 0; m + 1 < TM; m++) a[m] = a[m + 1];
        a[TM - 1] = ta;
        const double tb = b[0];
#pragma unroll
        for (int n = 0; n + 1 < TN; n++) b[n] = b[n + 1];
        b[TN - 1] = tb;
    }
    double s = 0;
    for (int m = 0; m < TM; m++) for (int n = 0; n < TN; n++) for (int i = 0; i < 4; i++) s += c[m][n][i];
    out[(size_t)blockIdx.x * THREADS + t] = s;
}

template <int TM, int TN, int THREADS, int MINB>
static void run(const char* name, const double* dA, const double* dB, double* dout, int blocks, int zeros) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_loop<TM, TN, THREADS, MINB>), dim3(blocks), dim3(THREADS), 0, 0, dA, dB, dout, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) iters = (int)(iters * 200.0 / ms);  // ~0.2 s per timed launch
    }
    const double flop = 2.0 * 16 * 16 * 4 * TM * TN * (double)iters * (THREADS / 64) * blocks;
    printf("%-58s %s operands: %8.2f ms  %6.2f TFLOP/s = %.3f of 78.6\n", name, zeros ? "zero  " : "random", ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
}

int main() {
    std::vector<double> h(4096), z(4096, 0.0);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    double *dA, *dB, *dZ, *dout;
    CHECK(hipMalloc(&dA, 4096 * 8)); CHECK(hipMalloc(&dB, 4096 * 8)); CHECK(hipMalloc(&dZ, 4096 * 8)); CHECK(hipMalloc(&dout, (size_t)1024 * 1024 * 8));
    CHECK(hipMemcpy(dA, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    CHECK(hipMemcpy(dB, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dZ, z.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (int zeros = 0; zeros < 2; zeros++) {
        const double* a = zeros ? dZ : dA;
        const double* b = zeros ? dZ : dB;
        run<4, 4, 256, 1>("1 wave / SIMD, 4 x 4 tiles (256 threads, 1 block per CU)", a, b, dout, 256, zeros);
        run<2, 4, 512, 1>("2 waves / SIMD, 2 x 4 tiles (512 threads, 1 block per CU)", a, b, dout, 256, zeros);
        run<2, 4, 512, 2>("4 waves / SIMD, 2 x 4 tiles (512 threads, 2 blocks per CU)", a, b, dout, 512, zeros);
        run<4, 4, 256, 2>("2 waves / SIMD, 4 x 4 tiles (256 threads, 2 blocks per CU)", a, b, dout, 512, zeros);
    }
    run_fixed<4, 4, 256, 1>("1 wave / SIMD, 4 x 4 tiles, no VALU in the loop", dA, dB, dout, 256);
    run_fixed<4, 4, 256, 2>("2 waves / SIMD, 4 x 4 tiles, no VALU in the loop", dA, dB, dout, 512);
    run_fixed<4, 4, 768, 1>("3 waves / SIMD, 4 x 4 tiles, no VALU in the loop (768 threads)", dA, dB, dout, 256);
    run_fixed<2, 4, 512, 2>("4 waves / SIMD, 2 x 4 tiles, no VALU in the loop", dA, dB, dout, 512);
    run_fixed<2, 4, 512, 1>("2 waves / SIMD, 2 x 4 tiles, no VALU in the loop", dA, dB, dout, 256);
    run_fixed<2, 2, 1024, 2>("8 waves / SIMD, 2 x 2 tiles, no VALU in the loop", dA, dB, dout, 512);
    run_fixed<2, 4, 1024, 1>("4 waves / SIMD, 2 x 4 tiles, no VALU in the loop (1024 threads, 1 block)", dA, dB, dout, 256);
    return 0;
}
