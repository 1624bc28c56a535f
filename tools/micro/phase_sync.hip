// What does a dependency between two phases cost INSIDE one persistent kernel, against a kernel boundary?
// Ticket-ordered persistent kernel (deadlock-free with any number of resident workgroups: a workgroup only ever waits for tickets lower than its own, and the
// holder of the lowest unfinished ticket is running by construction): tile = atomicAdd(ticket), phase = tile / T; before touching the previous phase's output
// it waits for done[phase - 1] == T (agent-scope acquire), after its stores it does an agent-scope release + atomicAdd(done[phase]).
// Work per tile models a skinny GEMM launch of the sampler's small stages: stream `wbytes / T` of weights (never re-used), read 256 B of another workgroup's
// previous-phase output, write 256 B.  Compared with the same phases as separate launches on one stream.
// hipcc -O3 --offload-arch=gfx950 tools/micro/phase_sync.hip -o tools/micro/phase_sync.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THR = 512;
constexpr unsigned long long TIMEOUT_TICKS = 20ull * 100000ull;      // 20 ms of the 100 MHz constant clock

struct Args {
    const f32x4* w; size_t w_phase_stride;   // f32x4 elements between the weights of consecutive phases
    float* act;                              // [P + 1][T][64]
    unsigned* ticket; unsigned* done;        // done[P]
    int* abort_flag;
    int P, T, per, prefetch;                 // per = f32x4 loads per thread and tile (<= 8)
};

__device__ __forceinline__ float tile_work(const f32x4* w, const float* in, float* out, int per, const f32x4* pre, bool use_pre) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (use_pre) {
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < per) s += pre[u];
    } else {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < per) v[u] = __builtin_nontemporal_load(w + (size_t)u * THR + threadIdx.x);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < per) s += v[u];
    }
    float r = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x < 64) out[threadIdx.x] = in[threadIdx.x] * 0.5f + 1.0f + r;
    return r;
}

__global__ __launch_bounds__(THR) void persistent_kernel(Args a) {
    __shared__ unsigned s_t;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= (unsigned)(a.P * a.T)) return;
        const int p = t / a.T, tile = t - p * a.T;
        const f32x4* w = a.w + (size_t)p * a.w_phase_stride + (size_t)tile * a.per * THR;
        f32x4 pre[8];
        if (a.prefetch) {
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < a.per) pre[u] = __builtin_nontemporal_load(w + (size_t)u * THR + threadIdx.x);
        }
        if (p > 0) {
            if (threadIdx.x == 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(a.done + p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.T) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) { *a.abort_flag = 1; break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int src = (tile * 7 + 3) % a.T;
        tile_work(w, a.act + ((size_t)p * a.T + src) * 64, a.act + ((size_t)(p + 1) * a.T + tile) * 64, a.per, pre, a.prefetch != 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.done + p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(THR) void phase_kernel(Args a, int p) {
    const int tile = blockIdx.x;
    const f32x4* w = a.w + (size_t)p * a.w_phase_stride + (size_t)tile * a.per * THR;
    const int src = (tile * 7 + 3) % a.T;
    f32x4 pre[8];
    tile_work(w, a.act + ((size_t)p * a.T + src) * 64, a.act + ((size_t)(p + 1) * a.T + tile) * 64, a.per, pre, false);
}

int main(int argc, char** argv) {
    const int P = 160, T = 256;
    const size_t big = (size_t)3 << 30;
    float* wbuf; CK(hipMalloc(&wbuf, big)); CK(hipMemset(wbuf, 0, big));
    float *act, *act_ref; CK(hipMalloc(&act, (size_t)(P + 1) * T * 64 * 4)); CK(hipMalloc(&act_ref, (size_t)(P + 1) * T * 64 * 4));
    unsigned* ctr; CK(hipMalloc(&ctr, (P + 16) * 4));
    int* abortf; CK(hipMalloc(&abortf, 4)); CK(hipMemset(abortf, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h0((size_t)T * 64);
    for (size_t i = 0; i < h0.size(); ++i) h0[i] = (float)(i % 97) * 0.01f;
    for (int per : {0, 2, 8}) {
        const size_t wphase = (size_t)T * per * THR;                 // f32x4 per phase
        const double mb = wphase * 16.0 / 1e6;
        Args a{reinterpret_cast<const f32x4*>(wbuf), wphase, act_ref, ctr, ctr + 16, abortf, P, T, per, 0};
        // reference: one launch per phase
        CK(hipMemcpy(act_ref, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
        float ms_l = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(T), dim3(THR), 0, st, a, p);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_l, e0, e1));
        }
        std::vector<float> ref((size_t)T * 64), got((size_t)T * 64);
        CK(hipMemcpy(ref.data(), act_ref + (size_t)P * T * 64, ref.size() * 4, hipMemcpyDeviceToHost));
        for (int grid : {256, 512}) {
            for (int prefetch : {0, 1}) {
                Args b = a; b.act = act; b.prefetch = prefetch;
                CK(hipMemcpy(act, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
                float ms_p = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(ctr, 0, (P + 16) * 4, st));
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(THR), 0, st, b);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_p, e0, e1));
                }
                int ab = 0; CK(hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(got.data(), act + (size_t)P * T * 64, got.size() * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
                printf("weights %5.1f MB/phase: launches %6.2f us/phase | persistent grid %3d prefetch %d: %6.2f us/phase  (abort %d, mismatches %zu)\n", mb,
                       ms_l * 1e3 / P, grid, prefetch, ms_p * 1e3 / P, ab, bad);
                fflush(stdout);
                if (ab) { printf("ABORT flag set: stopping\n"); return 2; }
            }
        }
    }
    return 0;
}
