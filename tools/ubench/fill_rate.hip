// Micro-benchmark: per-CU fill rate of a 256x128-byte operand tile pair into LDS, by staging method.
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), 2 x 32 KiB per stage, vmcnt(0)+barrier per stage
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging), same bytes
//   mode 2: A by LDS-DMA, B by register staging (two paths at once)
// Source: a matrix [rows][ld] int8; each block streams its own 256-row panel (like the real kernels) along K.
// hit = 1: every block reads the SAME two panels (L2-resident after the first touch); hit = 0: distinct panels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rs, int voffE, int ld, int k0, int8_t* lds, int w) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int grp = w * 4 + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + grp * 1024), 16,
                                                 (i & 1) ? (voffE ^ 64) : voffE, grp * 8 * ld + k0, 0, 0);
    }
}
template <int MODE>
__global__ __launch_bounds__(512) void k_fill(const int8_t* __restrict__ X, long ld, int nstages, int hit, int* sink) {
    __shared__ __attribute__((aligned(1024))) int8_t lds[2][2][32768];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const long panelA = hit ? 0 : (long)blockIdx.x * 2, panelB = panelA + 1;
    const int8_t* A = X + panelA * 256 * ld;
    const int8_t* B = X + panelB * 256 * ld;
    const int ldi = (int)ld;
    const int voffE = (lane >> 3) * ldi + (((lane & 7) ^ (lane >> 4)) << 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 256 * ldi, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 256 * ldi, 0x00020000);
    int acc = 0;
    for (int st = 0; st < nstages; st++) {
        const int buf = st & 1, k0 = st * 128;
        if (MODE == 0 || MODE == 2) dma_tile(rsA, voffE, ldi, k0, lds[buf][0], w);
        if (MODE == 0) dma_tile(rsB, voffE, ldi, k0, lds[buf][1], w);
        if (MODE == 1 || MODE == 2) {
            i32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = t + 512 * i;  // 2048 chunks of 16 B: 256 rows x 8 chunks
                const int row = c >> 3, ch = c & 7;
                r[i] = *(const i32x4*)(B + (long)row * ld + k0 + ch * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = t + 512 * i;
                const int row = c >> 3, ch = c & 7;
                *(i32x4*)(lds[buf][1] + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = r[i];
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = t + 512 * i;
                    const int row = c >> 3, ch = c & 7;
                    r[i] = *(const i32x4*)(A + (long)row * ld + k0 + ch * 16);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = t + 512 * i;
                    const int row = c >> 3, ch = c & 7;
                    *(i32x4*)(lds[buf][0] + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = r[i];
                }
            }
        }
        __syncthreads();
        acc += *(const int*)(lds[buf][0] + t * 4) + *(const int*)(lds[buf][1] + t * 4);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main(int argc, char** argv) {
    const long ld = 16384 + 128, rows = 256L * 2 * 1024;  // 1024 blocks x 2 panels, K = 16384 -> 128 stages
    const int nstages = 128;
    int8_t* X; int* sink;
    CHECK(hipMalloc((void**)&X, rows * ld));
    CHECK(hipMemset(X, 1, rows * ld));
    CHECK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int hit = 1; hit >= 0; hit--)
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_fill<0>, dim3(1024), dim3(512), 0, 0, X, ld, nstages, hit, sink);
                if (mode == 1) hipLaunchKernelGGL(k_fill<1>, dim3(1024), dim3(512), 0, 0, X, ld, nstages, hit, sink);
                if (mode == 2) hipLaunchKernelGGL(k_fill<2>, dim3(1024), dim3(512), 0, 0, X, ld, nstages, hit, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            double bytes = 1024.0 * nstages * 65536.0;
            printf("hit=%d mode=%d: %.3f ms  %.2f TB/s aggregate  %.1f GB/s per CU  (%.2f us per 64 KiB stage)\n", hit, mode, best,
                   bytes / best / 1e9, bytes / best / 1e6 / 256, best * 1e3 / (1024.0 / 256 * nstages));
        }
    return 0;
}
