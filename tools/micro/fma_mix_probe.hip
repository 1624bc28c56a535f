// Does v_fma_mixlo_f16 / v_fma_mixhi_f16 round h * (-1) + x to fp16 exactly like (_Float16)(x - (float)h), also when the result is an fp16 subnormal?
// hipcc --offload-arch=gfx950 -O2 tools/micro/fma_mix_probe.hip -o tools/micro/fma_mix_probe.bin && ./tools/micro/fma_mix_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void k(const float* x, uint32_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    const uint32_t h = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
    uint32_t l;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    out[2 * i] = l;
    out[2 * i + 1] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
}
int main() {
    const int n = 1 << 20;
    float* hx = (float*)malloc(2 * n * sizeof(float));
    uint32_t s = 12345u;
    for (int i = 0; i < 2 * n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)(s >> 8) / 16777216.0f - 0.5f;           // [-0.5, 0.5)
        const int e = (int)((s >> 3) % 24) - 20;                          // magnitudes 2^-20 .. 2^3
        hx[i] = ldexpf(u, e);
    }
    float* dx; uint32_t* dout; uint32_t* ho = (uint32_t*)malloc(2 * n * sizeof(uint32_t));
    hipMalloc(&dx, 2 * n * sizeof(float)); hipMalloc(&dout, 2 * n * sizeof(uint32_t));
    hipMemcpy(dx, hx, 2 * n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    long bad = 0, sub = 0;
    for (int i = 0; i < n; ++i) {
        if (ho[2 * i] != ho[2 * i + 1]) { if (bad < 8) printf("mismatch at %d: x = %.9g %.9g  mix %08x  ref %08x\n", i, hx[2 * i], hx[2 * i + 1], ho[2 * i], ho[2 * i + 1]); ++bad; }
        if (((ho[2 * i + 1] & 0x7C00u) == 0 && (ho[2 * i + 1] & 0x3FFu)) || ((ho[2 * i + 1] & 0x7C000000u) == 0 && (ho[2 * i + 1] & 0x3FF0000u))) ++sub;
    }
    printf("%ld of %d pairs differ; %ld pairs have a subnormal low plane in the reference\n", bad, n, sub);
    return bad ? 1 : 0;
}
