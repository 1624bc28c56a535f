// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from registers for 1/2/4 accumulator chains, 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j * 0.25f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 24 * (threads / 256);     // waves per SIMD = threads/256
    const double flops = 256.0 * (threads / 64) * iters * 24 * 32768.0;
    printf("%-28s threads/WG %3d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, threads, ms, flops / ms / 1e9, ms * 1e6 / mfma_per_simd);
    hipFree(out);
}
int main() {
    run<1>(256, "1 chain, 1 wave/SIMD"); run<2>(256, "2 chains, 1 wave/SIMD"); run<4>(256, "4 chains, 1 wave/SIMD");
    run<1>(512, "1 chain, 2 waves/SIMD"); run<2>(512, "2 chains, 2 waves/SIMD"); run<4>(512, "4 chains, 2 waves/SIMD");
    return 0;
}
