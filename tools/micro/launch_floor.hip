// Back-to-back launch floor on one stream: empty kernel, one dependent load, a 16 MB stream at several memory-level parallelisms.
// hipcc -O3 --offload-arch=gfx950 tools/micro/launch_floor.hip -o tools/micro/launch_floor.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_empty() {}
__global__ void k_one(const float* a, float* o) { o[blockIdx.x * blockDim.x + threadIdx.x] = a[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
// every thread sums `per` float4 loads, UNR of them in flight
template <int UNR>
__global__ void k_stream(const f32x4* a, float* o, int per) {
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < per; i += UNR) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = a[base + (size_t)(i + u) * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) o[base] = 1.f;
}
template <typename F> float timeit(F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f(i);
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f(i);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    const size_t big = (size_t)1 << 30;       // 1 GiB: cycle through it so a 16 MB stream is never L2 / MALL resident
    float *a, *o; CK(hipMalloc(&a, big)); CK(hipMalloc(&o, 64 << 20)); CK(hipMemset(a, 0, big)); CK(hipMemset(o, 0, 64 << 20));
    printf("empty kernel back-to-back: %.2f us\n", timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 2000));
    printf("empty kernel 256 WGs x 512: %.2f us\n", timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0); }, 2000));
    printf("one load+store, 16 WGs x 256: %.2f us\n", timeit([&](int) { hipLaunchKernelGGL(k_one, dim3(16), dim3(256), 0, 0, a, o); }, 2000));
    const size_t bytes = 16 << 20;
    for (int wgs : {64, 128, 256, 512, 1024}) {
        for (int thr : {256, 512}) {
            const int per = (int)(bytes / 16 / ((size_t)wgs * thr));
            if (per < 1) continue;
            auto run = [&](int unr) {
                return timeit([&](int i) {
                    const f32x4* p = reinterpret_cast<const f32x4*>(a) + (size_t)(i % 60) * (bytes / 16);
                    if (unr == 1) hipLaunchKernelGGL(k_stream<1>, dim3(wgs), dim3(thr), 0, 0, p, o, per);
                    else if (unr == 4) hipLaunchKernelGGL(k_stream<4>, dim3(wgs), dim3(thr), 0, 0, p, o, per);
                    else hipLaunchKernelGGL(k_stream<8>, dim3(wgs), dim3(thr), 0, 0, p, o, per);
                }, 600);
            };
            printf("16 MB stream, %4d WGs x %d thr, %3d loads/thread: in flight 1: %6.2f us", wgs, thr, per, run(1));
            if (per % 4 == 0) printf("  4: %6.2f us", run(4));
            if (per % 8 == 0) printf("  8: %6.2f us", run(8));
            printf("\n");
        }
    }
    return 0;
}
