// Issue rate of the vector instructions the attention softmax is made of (one wave per SIMD, 256 CUs busy): cycles per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/micro/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, float seed, int iters) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = 1.0001f, b1 = 0.5f;
    uint32_t u0 = threadIdx.x, u1 = 77u, u2 = 5u, u3 = 9u, v0 = threadIdx.x * 3u, v1 = 0xffu;
    float c0 = a0 * 2.f, c1 = c0 + 1.f, c2 = c0 + 2.f, c3 = c0 + 3.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {b0, b1}, p3 = {b1, b0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // 4 independent chains x 16 = 64 instructions per iteration
        if (OP == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %4\n v_cvt_pk_f16_f32 %2, %4, %4\n v_cvt_pk_f16_f32 %3, %5, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));) }
        if (OP == 2) { REP16(asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
        if (OP == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %5, %4\n v_pk_fma_f32 %3, %3, %5, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2), "v"(p3));) }
        if (OP == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 5) { REP16(asm volatile("v_fma_mixlo_f16 %0, %4, %5, %4 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %5, %4, %5 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %4, %5 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %5, %5, %4 op_sel_hi:[0,0,0]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));) }
        if (OP == 6) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0xffffe000u));) }
        if (OP == 7) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2), "v"(p3));) }
        if (OP == 8) { REP16(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %5, %4\n v_max3_f32 %3, %3, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (OP == 9) { REP16(asm volatile("v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
        if (OP == 10) { REP16(asm volatile("v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %5\n v_pk_mul_f16 %3, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x3c003c00u), "v"(0x3c003c00u));) }
        if (OP == 11) { REP16(asm volatile("v_lshrrev_b32 %0, 13, %0\n v_lshrrev_b32 %1, 13, %1\n v_sub_u32 %2, %2, %4\n v_perm_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x1c000u), "v"(0x07060302u));) }
        if (OP == 12) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %9, %8\n v_fma_f32 %3, %3, %9, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b0), "v"(b1));
                               asm volatile("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b0), "v"(b1));) }
        if (OP == 13) { REP16(asm volatile("v_and_b32 %0, %4, %5\n v_and_b32 %1, %4, %5\n v_and_b32 %2, %5, %4\n v_and_b32 %3, %5, %4" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(v0), "v"(v1));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1] + (float)(u0 ^ u1 ^ u2 ^ u3) + c0 + c1 + c2 + c3 == 12345.678f) out[1] = 1;
}

template <int OP> void run(const char* name, unsigned long long* d, int waves_per_simd) {
    const int iters = 2000;
    const int per_iter = (OP == 12) ? 128 : 64;
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, 0.5f, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, 0.5f, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("  [%7.1f us wall: %6.2f ns per instruction per wave]", ms * 1e3, ms * 1e6 / (iters * (double)per_iter));
    // s_memtime ticks at 100 MHz on this part: convert with the measured duration of a known-rate loop instead - report relative to v_fma_f32
    printf("%-28s %d wave(s)/SIMD: %8.3f memtime ticks per instruction\n", name, waves_per_simd, (double)h[0] / (iters * (double)per_iter));
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int w = 1; w <= 4; w *= 2) {
        run<4>("v_fma_f32 (reference)", d, w);
        run<12>("v_fma_f32, 8 chains", d, w);
        run<13>("v_and_b32, no dependence", d, w);
        run<3>("v_pk_fma_f32", d, w);
        run<7>("v_pk_add_f32", d, w);
        run<8>("v_max3_f32", d, w);
        run<6>("v_and_b32", d, w);
        run<11>("lshr/sub/perm mix", d, w);
        run<0>("v_exp_f32", d, w);
        run<1>("v_cvt_pk_f16_f32", d, w);
        run<2>("v_cvt_f32_f16", d, w);
        run<9>("v_cvt_f32_f16 sdwa", d, w);
        run<5>("v_fma_mixlo_f16", d, w);
        run<10>("v_pk_mul_f16", d, w);
    }
    return 0;
}
