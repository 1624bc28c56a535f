// Micro-benchmark: do vector-ALU instructions overlap with v_mfma_f32_32x32x16_bf16 on one SIMD?
// Per loop iteration: 24 MFMAs (2 accumulator chains) and NV independent v_fma_f32 (or v_exp_f32) per MFMA, interleaved
// in program order (sched_barrier keeps the order).  Compare "MFMA only", "VALU only" and "both".
// hipcc -w --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NV, int TRANS>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j * 0.25f); }
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x;
    const float c0 = seed * 0.999f, c1 = seed * 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            if (NM) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                if (TRANS == 1) v[w & 7] = __builtin_amdgcn_exp2f(v[w & 7]);
                else if (TRANS == 2) v[w & 7] = __uint_as_float(__float_as_uint(v[w & 7]) & (0xFFFF0000u | it));          // v_and_b32
                else if (TRANS == 3) v[w & 7] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(v[w & 7]), __float_as_uint(v[(w + 1) & 7]), 0x07060302u));
                else if (TRANS == 4) v[w & 7] = v[w & 7] - c1;                                                            // v_sub_f32
                else if (TRANS == 5) v[w & 7] = fmaxf(v[w & 7], v[(w + 3) & 7]);                                          // v_max_f32
                else if (TRANS == 6) v[w & 7] = __uint_as_float(__float_as_uint(v[w & 7]) + 3u);                          // v_add_u32
                else if (TRANS == 7) v[w & 7] = (v[(w + 1) & 7] > c0) ? v[w & 7] : c1;                                    // v_cmp + v_cndmask
                else v[w & 7] = __builtin_fmaf(v[w & 7], c0, c1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV, int TRANS>
void run(int threads, const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NV, TRANS>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %.3f ms  -> %.1f ns per (MFMA + %d VALU) slot per wave\n", name, threads / 256, ms, ms * 1e6 / (iters * 24.0), NV);
    hipFree(out);
}
int main() {
    for (int th = 256; th <= 512; th += 256) {
        run<1, 0, 0>(th, "MFMA only");
        run<0, 8, 0>(th, "8 v_fma only");     run<1, 8, 0>(th, "MFMA + 8 v_fma");
        run<0, 2, 1>(th, "2 v_exp only");     run<1, 2, 1>(th, "MFMA + 2 v_exp");   run<1, 4, 1>(th, "MFMA + 4 v_exp");
        run<0, 8, 2>(th, "8 v_and only");     run<1, 8, 2>(th, "MFMA + 8 v_and");
        run<0, 8, 3>(th, "8 v_perm only");    run<1, 8, 3>(th, "MFMA + 8 v_perm");
        run<0, 8, 4>(th, "8 v_sub only");     run<1, 8, 4>(th, "MFMA + 8 v_sub");
        run<0, 8, 5>(th, "8 v_max only");     run<1, 8, 5>(th, "MFMA + 8 v_max");
        run<0, 8, 6>(th, "8 v_add_u32 only"); run<1, 8, 6>(th, "MFMA + 8 v_add_u32");
        run<0, 8, 7>(th, "8 cmp+cndmask only"); run<1, 8, 7>(th, "MFMA + 8 cmp+cndmask");
    }
    return 0;
}
