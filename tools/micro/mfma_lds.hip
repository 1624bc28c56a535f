// Micro-benchmark: the inner K-step of gemm_bf16x3_v2 (18 ds_read_b128 + 24 MFMA + barrier) with no global traffic.
// Variants: 0 = MFMA only (fragments fixed), 1 = + LDS reads each step, 2 = + s_barrier each step,
//           3 = + the LDS-DMA stream of the real kernel (6 global_load_lds_dwordx4 per wave per step, L2-resident source),
//           4 = as 3 but the 6 DMA instructions are spread between the MFMAs instead of issued together after the barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#ifndef RANDOM_DATA
#define RANDOM_DATA 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, const uint16_t* src) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // RANDOM_DATA: realistic high-toggle bf16 operands (sign/exponent/mantissa all varying) instead of a smooth pattern
    for (int i = tid; i < 3 * 6 * 128 * 32; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        sm[i] = RANDOM_DATA ? (uint16_t)(((h & 0x8000u)) | (0x3c00u + ((h >> 3) & 0x07ffu))) : (uint16_t)(0x3c00 + (i * 7 % 97));
    }
    __syncthreads();
    const int wm = wave >> 2, wn = wave & 3, sw = (li >> 2) & 3;
    const int offa0 = (wm * 64 + li) * 32, offb = (wn * 32 + li) * 32, ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    bf16x8 fa[2][3][2], fb[2][3];
    for (int s = 0; s < 2; ++s) for (int p = 0; p < 3; ++p) { fb[s][p] = *(bf16x8*)(sm + (3 + p) * 4096 + offb + (s ? ch1 : ch0)); for (int i = 0; i < 2; ++i) fa[s][p][i] = *(bf16x8*)(sm + p * 4096 + offa0 + i * 1024 + (s ? ch1 : ch0)); }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const uint16_t* gsrc = src + (size_t)blockIdx.x * 0 + (size_t)(wave * 64 + lane) * 8;     // every block streams the same 48 KB x 64 steps (L2 hits)
    auto issue1 = [&](int t, int q) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gsrc + (size_t)(t & 63) * 24576 + q * 4096), (lds_ptr_t)(sm + ((t + 2) % 3) * 24576 + q * 4096 + wave * 512), 16, 0, 0);
    };
    for (int t = 0; t < iters; ++t) {
        if (VAR >= 3) { if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        if (VAR >= 2) __builtin_amdgcn_s_barrier();
        if (VAR == 3) { for (int q = 0; q < 6; ++q) issue1(t, q); }
        if (VAR >= 1) {
            const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sm + (t % 3) * 6 * 128 * 32);
            const uint32_t aa0 = sb + 2 * (offa0 + ch0), aa1 = sb + 2 * (offa0 + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
            RD(fa[0][0][0], aa0, 0); RD(fb[0][2], ab0, 40960); RD(fa[0][1][0], aa0, 8192); RD(fb[0][1], ab0, 32768); RD(fa[0][2][0], aa0, 16384); RD(fb[0][0], ab0, 24576);
            RD(fa[0][0][1], aa0, 2048); RD(fa[0][1][1], aa0, 10240); RD(fa[0][2][1], aa0, 18432);
            RD(fa[1][0][0], aa1, 0); RD(fb[1][2], ab1, 40960); RD(fa[1][1][0], aa1, 8192); RD(fb[1][1], ab1, 32768); RD(fa[1][2][0], aa1, 16384); RD(fb[1][0], ab1, 24576);
            RD(fa[1][0][1], aa1, 2048); RD(fa[1][1][1], aa1, 10240); RD(fa[1][2][1], aa1, 18432);
#undef RD
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (VAR >= 1) { if (s == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][2], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][2][i], fb[s][0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][0], acc[i], 0, 0, 0);
                if (VAR == 4) { issue1(t, 2 * (2 * s + i) % 6); if (s == 0) issue1(t, (2 * (2 * s + i) + 1) % 6); __builtin_amdgcn_sched_barrier(0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 512 + tid] = s;
}
template <int VAR>
void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    uint16_t* src; (void)hipMalloc(&src, 64 * 24576 * 2 + 65536); (void)hipMemset(src, 0x3c, 64 * 24576 * 2 + 65536);
    const int iters = 4000; const size_t lds = 3 * 6 * 128 * 32 * 2;
    (void)hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(512), lds, 0, out, iters, src); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); }
    const double mfma_per_simd = (double)iters * 24 * 2;
    printf("%-44s %.3f ms  %.1f ns per MFMA per SIMD  (%.0f TFLOP/s bf16)\n", name, ms, ms * 1e6 / mfma_per_simd, 256.0 * 8 * iters * 24 * 32768.0 / ms / 1e9);
    (void)hipFree(out);
}
int main() { run<0>("MFMA only (24 per step, 2 waves/SIMD)"); run<1>("+ 18 ds_read_b128 per step"); run<2>("+ s_barrier per step"); run<3>("+ 6 LDS-DMA per wave per step (after barrier)"); run<4>("+ 6 LDS-DMA spread between MFMAs"); return 0; }
