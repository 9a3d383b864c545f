// What the chip sustains on v_mfma_i32_32x32x32_i8 with nothing else going on: a register-resident loop (2 waves per SIMD, 12
// accumulator tiles per wave like k_vara_i8p, no LDS, no memory traffic) running for ~150 ms, i.e. in the power-limited steady
// state, on operands with the statistics of the scan: A = re-centred genotypes (values in {-2..2}, about half of them zero) or
// uniformly random bytes or zeros; B = balanced base-256 digits of W (uniformly random bytes) or zeros.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_ceiling.hip -o tools/ubench/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512, 2) void k_loop(const i32x4* __restrict__ A, const i32x4* __restrict__ B, int* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    i32x4 a[3], b[4];
    for (int m = 0; m < 3; m++) a[m] = A[(t + 512 * m) & 4095];
    for (int n = 0; n < 4; n++) b[n] = B[(t + 512 * n) & 4095];
    i32x16 c[3][4];
    for (int m = 0; m < 3; m++) for (int n = 0; n < 4; n++) for (int i = 0; i < 16; i++) c[m][n][i] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int n = 0; n < 4; n++) c[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], c[m][n], 0, 0, 0);
        // rotate the fragments so that consecutive MFMAs see different operand bits (as the real kernel does every k-step)
        const i32x4 ta = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = ta;
        const i32x4 tb = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = tb;
    }
    int s = 0;
    for (int m = 0; m < 3; m++) for (int n = 0; n < 4; n++) for (int i = 0; i < 16; i++) s += c[m][n][i];
    out[blockIdx.x * 512 + t] = s;
}

// the other int8 shape, 16 x 16 x 64 (same MACs per wave as 12 tiles of 32 x 32 x 32: 24 tiles of 16 x 16, 8 accumulator registers each)
typedef int i32x4b __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k_loop16(const i32x4* __restrict__ A, const i32x4* __restrict__ B, int* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    i32x4 a[6], b[8];
    for (int m = 0; m < 6; m++) a[m] = A[(t + 512 * m) & 4095];
    for (int n = 0; n < 8; n++) b[n] = B[(t + 512 * n) & 4095];
    i32x4 c[6][8];
    for (int m = 0; m < 6; m++) for (int n = 0; n < 8; n++) for (int i = 0; i < 4; i++) c[m][n][i] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 6; m++)
#pragma unroll
            for (int n = 0; n < 8; n++) c[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[n], c[m][n], 0, 0, 0);
        const i32x4 ta = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = a[4]; a[4] = a[5]; a[5] = ta;
        const i32x4 tb = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = b[4]; b[4] = b[5]; b[5] = b[6]; b[6] = b[7]; b[7] = tb;
    }
    int s = 0;
    for (int m = 0; m < 6; m++) for (int n = 0; n < 8; n++) for (int i = 0; i < 4; i++) s += c[m][n][i];
    out[blockIdx.x * 512 + t] = s;
}

// the MM^T kernel's instruction: fp4 x fp4 on the block-scaled MFMA (K = 64), wave shape of k_syrk_f4p (4 x 2 tiles)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 2) void k_loop_f4(const i32x4* __restrict__ A, const i32x4* __restrict__ B, float* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    i32x8 a[4], b[2];
    for (int m = 0; m < 4; m++) { const i32x4 v = A[(t + 512 * m) & 4095]; a[m] = i32x8{v[0], v[1], v[2], v[3], 0, 0, 0, 0}; }
    for (int n = 0; n < 2; n++) { const i32x4 v = B[(t + 512 * n) & 4095]; b[n] = i32x8{v[0], v[1], v[2], v[3], 0, 0, 0, 0}; }
    f32x16 c[4][2];
    for (int m = 0; m < 4; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) c[m][n][i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) c[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[n], c[m][n], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        const i32x8 ta = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = ta;
        const i32x8 tb = b[0]; b[0] = b[1]; b[1] = tb;
    }
    float s = 0;
    for (int m = 0; m < 4; m++) for (int n = 0; n < 2; n++) for (int i = 0; i < 16; i++) s += c[m][n][i];
    out[blockIdx.x * 512 + t] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;   // 12 MFMAs per iteration and wave; 2 waves per SIMD -> ~0.15-0.25 s
    srand(5);
    std::vector<int8_t> geno(65536), genof(65536), genou(65536), rnd(65536), zero(65536, 0);
    for (auto& x : rnd) x = (int8_t)(rand() & 0xff);
    for (auto& x : geno) {  // HWE genotypes with p ~ U(0.05, 0.5), re-centred on the marker's majority genotype (per 16-byte run)
        x = 0;
    }
    for (size_t r = 0; r < geno.size(); r += 16) {
        const double p = 0.05 + 0.45 * (rand() / (double)RAND_MAX), q = 1 - p;
        const double f0 = q * q, f1 = 2 * p * q;
        const int maj = f0 >= f1 && f0 >= p * p ? 0 : (f1 >= p * p ? 1 : 2);
        for (int i = 0; i < 16; i++) {
            const double u = rand() / (double)RAND_MAX;
            const int g = u < f0 ? 0 : (u < f0 + f1 ? 1 : 2);
            geno[r + i] = (int8_t)(g - maj);
            genof[r + i] = (int8_t)(maj == 1 && f0 > p * p ? maj - g : g - maj);  // sign chosen so that the commoner non-zero value is +1
            genou[r + i] = (int8_t)(maj == 2 ? 2 - g : g);                        // never negative: only the majority-2 rows re-centred (and flipped)
        }
    }
    i32x4 *dA, *dB; int* dOut;
    CHECK(hipMalloc((void**)&dA, 65536)); CHECK(hipMalloc((void**)&dB, 65536)); CHECK(hipMalloc((void**)&dOut, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; const int8_t* a; const int8_t* b; } cases[] = {
        {"A genotypes (re-centred) x B random digits", geno.data(), rnd.data()},
        {"A random digits x B genotypes (roles swapped)", rnd.data(), geno.data()},
        {"A random digits x B genotypes, sign of a row chosen so that +1 is commoner than -1", rnd.data(), genof.data()},
        {"A random digits x B genotypes never negative (heterozygote-majority rows not re-centred)", rnd.data(), genou.data()},
        {"A random bytes          x B random digits", rnd.data(), rnd.data()},
        {"A zeros                 x B random digits", zero.data(), rnd.data()},
        {"A genotypes (re-centred) x B zeros        ", geno.data(), zero.data()},
    };
    for (auto& cs : cases) {
        CHECK(hipMemcpy(dA, cs.a, 65536, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, cs.b, 65536, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_loop, dim3(256), dim3(512), 0, 0, dA, dB, dOut, iters);   // one workgroup of 8 waves per CU
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = 2.0 * 256 * 8 * (double)iters * 12 * 32 * 32 * 32;
            if (rep) printf("%s: %8.2f ms  %.2f POP/s  (= %.3f of 5 POP/s)\n", cs.name, ms, ops / ms / 1e12, ops / ms / 1e12 / 5.0);
        }
    }
    struct { const char* name; const int8_t* a; const int8_t* b; } cases16[] = {
        {"16x16x64: A random digits x B genotypes", rnd.data(), genof.data()},
        {"16x16x64: A genotypes x B random digits", genof.data(), rnd.data()},
    };
    for (auto& cs : cases16) {
        CHECK(hipMemcpy(dA, cs.a, 65536, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, cs.b, 65536, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_loop16, dim3(256), dim3(512), 0, 0, dA, dB, dOut, iters / 2);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = 2.0 * 256 * 8 * (double)(iters / 2) * 48 * 16 * 16 * 64;
            if (rep) printf("%s: %8.2f ms  %.2f POP/s  (= %.3f of 5 POP/s)\n", cs.name, ms, ops / ms / 1e12, ops / ms / 1e12 / 5.0);
        }
    }
    // fp4 codes (e2m1), two per byte: -1 / 0 / +1 as 0xA / 0x0 / 0x2 (the g - 1 coding of MM^T) against 0 / 1 / 2 as 0x0 / 0x2 / 0x4
    std::vector<int8_t> f4m(65536), f4p(65536);
    for (size_t r = 0; r < f4m.size(); r += 16) {
        const double p = 0.05 + 0.45 * (rand() / (double)RAND_MAX), q = 1 - p;
        for (int i = 0; i < 16; i++) {
            int codes_m = 0, codes_p = 0;
            for (int hnib = 0; hnib < 2; hnib++) {
                const double u = rand() / (double)RAND_MAX;
                const int g = u < q * q ? 0 : (u < q * q + 2 * p * q ? 1 : 2);
                codes_m |= (g == 0 ? 0xA : (g == 1 ? 0x0 : 0x2)) << (4 * hnib);
                codes_p |= (g == 0 ? 0x0 : (g == 1 ? 0x2 : 0x4)) << (4 * hnib);
            }
            f4m[r + i] = (int8_t)codes_m; f4p[r + i] = (int8_t)codes_p;
        }
    }
    struct { const char* name; const int8_t* a; } cases4[] = {
        {"fp4 x fp4, genotypes as -1 / 0 / +1 (MM^T as shipped)", f4m.data()},
        {"fp4 x fp4, genotypes as  0 / 1 / 2  (non-negative)    ", f4p.data()},
        {"fp4 x fp4, random nibbles                              ", rnd.data()},
    };
    for (auto& cs : cases4) {
        CHECK(hipMemcpy(dA, cs.a, 65536, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, cs.a, 65536, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_loop_f4, dim3(256), dim3(512), 0, 0, dA, dB, (float*)dOut, iters * 3 / 2);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = 2.0 * 256 * 8 * (double)(iters * 3 / 2) * 8 * 32 * 32 * 64;
            if (rep) printf("%s: %8.2f ms  %.2f POP/s  (= %.3f of 10 POP/s)\n", cs.name, ms, ops / ms / 1e12, ops / ms / 1e12 / 10.0);
        }
    }
    return 0;
}
