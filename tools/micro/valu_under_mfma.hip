// Does a wave's vector-ALU issue rate suffer while the OTHER wave of its SIMD feeds the matrix pipe back to back?  512-thread workgroups, one per CU:
// waves 0-3 run a vector loop (64 independent-enough instructions per iteration), waves 4-7 either idle (mode 0), run dependent v_mfma_f32_32x32x16_f16
// back to back (mode 1) or the same vector loop (mode 2).   hipcc --offload-arch=gfx950 -O3 -o /tmp/vum tools/micro/valu_under_mfma.hip && /tmp/vum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__device__ __forceinline__ void valu_loop(int iters, float& a0, float& a1, float& a2, float& a3, unsigned& u0, unsigned& u1) {
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(1.0001f), "v"(0.5f));) }
        if (OP == 1) { REP16(asm volatile("v_exp_f32 %0, %0\n v_cvt_pk_f16_f32 %4, %1, %2\n v_cvt_f32_f16 %3, %5\n v_max3_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0) : "v"(u1));) }
    }
}

template <int OP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f; unsigned u0 = 1, u1 = 0x3c003c00u;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f16x8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave < 4) valu_loop<OP>(iters, a0, a1, a2, a3, u0, u1);
    else if (mode == 1) { for (int i = 0; i < iters * 8; ++i) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc, 0, 0, 0); } }
    else if (mode == 2) valu_loop<OP>(iters, a0, a1, a2, a3, u0, u1);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    if (a0 + a1 + a2 + a3 + acc[0] + acc[7] + (float)u0 == 1234.5f) out[15] = 1;
}

template <int OP> void run(const char* name, unsigned long long* d) {
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, d, iters, mode); hipDeviceSynchronize(); }
        unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        const char* mn[3] = {"partner idle", "partner: back-to-back MFMA", "partner: the same vector loop"};
        printf("%-22s %-32s vector wave %7.2f ns per instruction;  partner wave %8.1f us total\n", name, mn[mode], h[0] * 10.0 / (iters * 64.0), h[4] * 0.01);
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 128);
    run<0>("v_fma_f32", d);
    run<1>("exp/cvt_pk/cvt/max3 mix", d);
    return 0;
}
