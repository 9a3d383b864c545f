// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with A = fp4 (e2m1: genotypes -1/0/1) and B = fp6 (e2m3: digits -16..16 as d/8):
// (1) operand layout check against an exact integer product, (2) throughput of a register-resident loop on random data
// next to the int8 32x32x32 loop.  Next-round lever for k_vara_i8: base-33 digits at twice the int8 MAC rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_once(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, float* __restrict__ C) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (int)A[l * 8 + i]; b[i] = (int)B[l * 8 + i]; }
    f32x16 c;
    for (int i = 0; i < 16; i++) c[i] = 0.f;
    // cbsz = 4 (A fp4), blgp = 2 (B fp6 e2m3); scales 1.0 (E8M0 127)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; i++) C[l * 16 + i] = c[i];
}

template <int MODE>  // 0: fp4 x fp6 scaled 32x32x64, 1: int8 32x32x32
__global__ __launch_bounds__(256) void k_loop(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, float* __restrict__ out, int iters) {
    const int l = threadIdx.x & 63;
    i32x8 a[2], b[2];
    for (int u = 0; u < 2; u++)
        for (int i = 0; i < 8; i++) { a[u][i] = (int)A[(l + 64 * u) * 8 + i]; b[u][i] = (int)B[(l + 64 * u) * 8 + i]; }
    if (MODE == 0) {
        f32x16 c[2][2];
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) c[m][n][i] = 0.f;
        for (int it = 0; it < iters; it++)
            for (int m = 0; m < 2; m++)
                for (int n = 0; n < 2; n++) c[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[n], c[m][n], 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        float s = 0;
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) s += c[m][n][i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        i32x16 c[2][2];
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) c[m][n][i] = 0;
        i32x4 a4[2], b4[2];
        for (int u = 0; u < 2; u++) for (int i = 0; i < 4; i++) { a4[u][i] = a[u][i]; b4[u][i] = b[u][i]; }
        for (int it = 0; it < iters; it++)
            for (int m = 0; m < 2; m++)
                for (int n = 0; n < 2; n++) c[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4[m], b4[n], c[m][n], 0, 0, 0);
        int s = 0;
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) s += c[m][n][i];
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;
    }
}

static int fp4_code(int v) { return v == 0 ? 0 : (v > 0 ? 0x2 : 0xA); }            // +-1.0
static int fp6_code(int d) {                                                        // d/8, |d| <= 16
    int s = d < 0, m = abs(d), code;
    if (m < 8) code = m; else if (m < 16) code = (1 << 3) | (m - 8); else code = (2 << 3) | 0;  // 16 = 2.0 * 8/8
    return (s << 5) | code;
}

int main() {
    srand(3);
    std::vector<int> Ai(32 * 64), Bi(64 * 32);  // A[row][k], B[k][col]
    for (auto& x : Ai) x = rand() % 3 - 1;
    for (auto& x : Bi) x = rand() % 33 - 16;
    // hypothesis: lane l holds row/col l&31, k = 32*(l>>5) + i (i < 32); fp4: nibble i of the lane's 16 bytes, low nibble first;
    // fp6: 6-bit field i of the lane's 24-byte little-endian bit stream.
    std::vector<uint32_t> A(64 * 8, 0), B(64 * 8, 0);
    for (int l = 0; l < 64; l++) {
        uint8_t ab[32] = {0}, bb[32] = {0};
        for (int i = 0; i < 32; i++) {
            const int k = 32 * (l >> 5) + i;
            ab[i >> 1] |= fp4_code(Ai[(l & 31) * 64 + k]) << (4 * (i & 1));
            const int code = fp6_code(Bi[k * 32 + (l & 31)]);
            const int bit = 6 * i;
            bb[bit >> 3] |= (code << (bit & 7)) & 0xff;
            if ((bit & 7) > 2) bb[(bit >> 3) + 1] |= code >> (8 - (bit & 7));
        }
        memcpy(&A[l * 8], ab, 32);
        memcpy(&B[l * 8], bb, 32);
    }
    uint32_t *dA, *dB; float* dC;
    CHECK(hipMalloc((void**)&dA, 128 * 8 * 4)); CHECK(hipMalloc((void**)&dB, 128 * 8 * 4)); CHECK(hipMalloc((void**)&dC, 256 * 1024 * 4));
    CHECK(hipMemset(dA, 0, 128 * 8 * 4)); CHECK(hipMemset(dB, 0, 128 * 8 * 4));
    CHECK(hipMemcpy(dA, A.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(64 * 16);
    CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int q = 0; q < 16; q++) {
            const int col = l & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5);
            long s = 0;
            for (int k = 0; k < 64; k++) s += (long)Ai[row * 64 + k] * Bi[k * 32 + col];
            if (C[l * 16 + q] * 8.0f != (float)s) { if (bad < 6) printf("  C[%d][%d] = %g*8 = %g, expected %ld\n", row, col, C[l * 16 + q], C[l * 16 + q] * 8, s); bad++; }
        }
    printf("layout check (fp4 x fp6, 32x32x64): %d of 1024 outputs differ from the exact integer product\n", bad);

    // throughput: 1024 blocks x 4 waves, 4 MFMAs per iteration per wave
    std::vector<uint32_t> R(128 * 8);
    for (auto& x : R) x = (uint32_t)rand() * 2654435761u;
    // keep fp6/fp4 random codes finite: every 6-bit / 4-bit code is a finite number in e2m3 / e2m1, so raw random bits are fine
    CHECK(hipMemcpy(dA, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    for (auto& x : R) x = (uint32_t)rand() * 2246822519u;
    CHECK(hipMemcpy(dB, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 1024;
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_loop<0>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, iters);
            else hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double macs = (double)blocks * 4 * iters * 4 * 32 * 32 * (mode == 0 ? 64 : 32);
            printf("%s: %.2f ms  %.0f TOP/s   (%.1f useful digit-bits x TMAC/s: %.0f)\n", mode == 0 ? "fp4 x fp6 32x32x64" : "int8 32x32x32     ", ms,
                   2 * macs / ms / 1e9, mode == 0 ? 5.04 : 8.0, (mode == 0 ? 5.04 : 8.0) * macs / ms / 1e9);
        }
    return 0;
}
