// Which f16 MFMA shape does the chip sustain the higher FLOP/s on under full load?  (MI355X_MICROARCH.md "DVFS give-back" item 7: on bf16 the 16x16x32 loop delivered
// ~1.15x the 32x32x16 loop at equal cycles per FLOP, because the chip holds a higher clock on it.)  The 256 x 256 GEMM kernel's K loop runs at the matrix pipe's
// issue rate and at the power-limited clock (profiles/r03_d_gemm_v4_kernel_stamps.log: 1.44 GHz), so the shape is the one in-loop lever left.
// Both loops: 2 waves per SIMD, every CU, 128 accumulator registers per lane (the wave tile of the GEMM kernel), random fp16 operands re-read from LDS
// (ds_read_b128, as in the kernel) every step; the same FLOPs per step.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape_f16.hip -o tools/micro/mfma_shape_f16.bin && ./tools/micro/mfma_shape_f16.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>          // 0: 32x32x16, 8 accumulator tiles; 1: 16x16x32, 32 accumulator tiles
__global__ __launch_bounds__(512, 2) void k(const uint16_t* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[16384];            // 32 KB of operand fragments
    for (int i = threadIdx.x; i < 16384; i += 512) sm[i] = src[(blockIdx.x * 16384 + i) & 0xFFFFF];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f16x8* fr = reinterpret_cast<const f16x8*>(sm) + lane;             // fragment f of step s: fr[64 * ((s + f) & 31)]
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[8];
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // one k16 step of a 128 x 64 wave tile: 4 + 2 fragments x 2 planes = 12 reads, 8 tiles x 3 products = 24 MFMAs
            f16x8 a[4][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) for (int p = 0; p < 2; ++p) a[i][p] = fr[64 * ((it + 2 * i + p + wave) & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) for (int p = 0; p < 2; ++p) b[j][p] = fr[64 * ((it + 8 + 2 * j + p + wave) & 31)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][0], a[i][1], acc[2 * i + j], 0, 0, 0);
                    acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][1], a[i][0], acc[2 * i + j], 0, 0, 0);
                    acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][0], a[i][0], acc[2 * i + j], 0, 0, 0);
                }
        }
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    } else {
        f32x4 acc[32];
        for (int n = 0; n < 32; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; it += 2) {
            // one k32 step of the same wave tile (= two k16 steps of the loop above): 8 + 4 fragments x 2 planes = 24 reads, 32 tiles x 3 products = 96 MFMAs
            f16x8 a[8][2], b[4][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) for (int p = 0; p < 2; ++p) a[i][p] = fr[64 * ((it + 2 * i + p + wave) & 31)];
#pragma unroll
            for (int j = 0; j < 4; ++j) for (int p = 0; p < 2; ++p) b[j][p] = fr[64 * ((it + 16 + 2 * j + p + wave) & 31)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][1], acc[4 * i + j], 0, 0, 0);
                    acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][1], a[i][0], acc[4 * i + j], 0, 0, 0);
                    acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][0], a[i][0], acc[4 * i + j], 0, 0, 0);
                }
        }
        for (int n = 0; n < 32; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    const int iters = 4096;          // k16 steps per launch
    uint16_t* hsrc = (uint16_t*)malloc((1 << 20) * 2);
    uint32_t st = 777u;
    for (int i = 0; i < (1 << 20); ++i) { st = st * 1664525u + 1013904223u; const float v = ((st >> 8) / 16777216.0f - 0.5f) * 4.0f; _Float16 h = (_Float16)v; hsrc[i] = *(uint16_t*)&h; }
    uint16_t* src; float* out; unsigned long long* clk;
    hipMalloc(&src, (1 << 20) * 2); hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&clk, 16);
    hipMemcpy(src, hsrc, (1 << 20) * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
        for (int shape = 0; shape < 2; ++shape) {
            const int launches = 300;       // ~0.5 s of back-to-back launches per arm: the clock the chip holds under this load
            hipEventRecord(e0);
            for (int l = 0; l < launches; ++l) {
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, out, iters, clk);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, out, iters, clk);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double flops = 256.0 * 8 * (double)iters * 24 * 32768.0 * launches;       // 24 32x32x16 MFMAs (= 96 16x16x32 per two steps) per wave and k16 step
            printf("round %d %-10s: %.1f us per launch, %.0f TFLOP/s (MFMA rate), in-kernel clock %.2f GHz, %.1f cycles per k16 step and wave pair\n", round, shape ? "16x16x32" : "32x32x16",
                   ms * 1e3 / launches, flops / ms / 1e9, (double)c[0] / ((double)c[1] * 10.0), (double)c[0] / iters);
        }
    return 0;
}
