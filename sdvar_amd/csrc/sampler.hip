// Token sampling and the acceptance scan (HBM-bound row kernels, one 256-thread workgroup per token).
//
//   cfg_sample      CFG combine + top-k + top-p + multinomial draw          /root/reference/models/var.py:199-202,
//                                                                           models/helpers.py:6-19
//   verify_accept   CFG combine + argmax_V + compare with the draft ids + per-stage match counts, then the
//                   "leading stages with batch match rate >= thr" scan       models/var.py:1062-1067, 1199-1222
//   noise_fill      the Philox Exp(1) stream of sdvar_amd/noise.py, for tests of the device-noise mode
//
// Exactness notes (the ids must equal the reference's):
//   * CFG is (1+t)*cond - t*uncond with two roundings then a subtraction, as torch evaluates it: no fma contraction.
//   * top-k keeps ties at the k-th value (logits < kth are dropped); the k-th value comes from an exact radix select.
//   * top-p follows helpers.py:12-15: ascending sort (bitonic, keys (value, index)), softmax of the sorted row,
//     cumulative sum accumulated in double and rounded to float per element (ATen's CPU cumsum accumulates float in
//     double), removed iff cumsum <= float(1 - top_p), last element always kept.
//   * the draw is argmax(p / q), q ~ Exp(1): torch.multinomial's own formulation; q is either an explicit
//     (B*l, V) input (parity mode) or generated in-kernel from Philox4x32-10 (device mode).
#include "common.h"

namespace sdvar {

constexpr int NV = 4096;          // sort width (V <= NV, padded with keys below -inf)
constexpr int VPT = NV / 256;     // values per thread = 16 (4 float4 chunks)

__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// block-wide helpers over 256 threads (4 waves); `red` is LDS scratch of >= 8 doubles
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

struct SampleArgs {
    const float* logits;      // (2B, l, V)
    const float* q;           // (B*l, V) or null -> philox
    long long* ids;           // ids[b * ids_stride + tok]
    int B, l, V, ids_stride;
    float one_plus_t, t;      // float32((1+t)), float32(t)
    int top_k; float top_p_thr; int use_top_p;    // thr = float32(1 - top_p)
    uint32_t k0, k1, draw, image_offset;
    float* dbg_masked;        // optional (B, l, V): masked logits for tests
};

__global__ __launch_bounds__(256) void cfg_sample_kernel(SampleArgs a) {
    __shared__ unsigned long long sbuf[NV];        // sort buffer (32 KB); also histogram scratch
    __shared__ unsigned char rm[NV];
    __shared__ double redd[8];
    __shared__ float redf[8];
    __shared__ int sel[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = blockIdx.x, b = blockIdx.y;
    const int V = a.V;
    const float* pc = a.logits + ((size_t)b * a.l + tok) * V;
    const float* pu = a.logits + ((size_t)(a.B + b) * a.l + tok) * V;

    // ---- CFG combine; thread owns chunks v4 = tid + 256*j (4 consecutive vocab entries each)
    float x[VPT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v0 = 4 * (tid + 256 * j);
        if (v0 < V) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(pc + v0), u = *reinterpret_cast<const f32x4*>(pu + v0);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * j + e] = __fsub_rn(__fmul_rn(a.one_plus_t, c[e]), __fmul_rn(a.t, u[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * j + e] = -INFINITY;
        }
    }

    // ---- top-k: exact radix select of the k-th largest key, 4 passes of 8 bits
    if (a.top_k > 0 && a.top_k < V) {
        unsigned int* hist = reinterpret_cast<unsigned int*>(sbuf);
        unsigned int* wsum = hist + 256;
        uint32_t prefix = 0, pmask = 0;
        int k = a.top_k;                               // rank from the top, 1-based
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3);
                if (v < V) {
                    const uint32_t key = f2key(x[i]);
                    if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            // suffix-inclusive count: n_ge[d] = sum_{d' >= d} hist[d'];  thread tid owns digit d = 255 - tid
            const unsigned int mine = hist[255 - tid];
            unsigned int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            unsigned int base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            inc += base;                               // elements with digit >= d
            const unsigned int exc = inc - mine;       // elements with digit > d
            if ((unsigned)k > exc && (unsigned)k <= inc) { sel[0] = 255 - tid; sel[1] = k - (int)exc; }
            __syncthreads();
            prefix |= ((uint32_t)sel[0]) << shift; pmask |= 255u << shift; k = sel[1];
            __syncthreads();
        }
        const float kth = key2f(prefix);
#pragma unroll
        for (int i = 0; i < VPT; ++i) if (x[i] < kth) x[i] = -INFINITY;
    }

    // ---- top-p
    if (a.use_top_p) {
        // Survivors of the top-k filter (finite entries).  At most 1024 of them (top_k = 900 + ties at the k-th value): only they are sorted - compacted into
        // sbuf[0, 1024), padded with keys below every finite key - a quarter of the rows and 55 instead of 78 compare-exchange rounds.  The filtered-out
        // entries are -inf: probability 0, first in the ascending order, removed or not they stay -inf, so dropping them changes neither the sorted
        // probabilities nor their running sum.  More survivors (no top-k, or a large one): the full 4096-row sort.
        int nf = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) nf += (x[i] > -INFINITY) ? 1 : 0;
        int inc = nf;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
        __syncthreads();
        if (lane == 63) sel[wave] = inc;
        __syncthreads();
        int base = inc - nf;
        for (int w = 0; w < wave; ++w) base += sel[w];
        const int total = sel[0] + sel[1] + sel[2] + sel[3];
        __syncthreads();
        if (total > 0 && total <= 1024) {
            constexpr int NC = 1024, CPT = NC / 256;
            for (int p2 = total + tid; p2 < NC; p2 += 256) sbuf[p2] = (unsigned long long)(unsigned)(NV + p2);      // pads: key 0, below every finite key
            int slot = base;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3);
                if (x[i] > -INFINITY) sbuf[slot++] = ((unsigned long long)f2key(x[i]) << 32) | (unsigned)v;
            }
            __syncthreads();
            for (int k = 2; k <= NC; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                    for (int it = 0; it < NC / 512; ++it) {
                        const int p2 = tid + 256 * it;                   // pair index 0..511
                        const int i0 = ((p2 & ~(j - 1)) << 1) | (p2 & (j - 1));
                        const int i1 = i0 | j;
                        const bool up = (i0 & k) == 0;
                        const unsigned long long A = sbuf[i0], Bv = sbuf[i1];
                        if ((A > Bv) == up) { sbuf[i0] = Bv; sbuf[i1] = A; }
                    }
                    __syncthreads();
                }
            }
            // sorted ascending; thread owns positions 4*tid .. 4*tid+3.  softmax over the sorted row, running sum in double as below
            float sv[CPT]; int si[CPT];
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const unsigned long long e = sbuf[CPT * tid + i];
                si[i] = (int)(e & 0xFFFFFFFFu);
                sv[i] = ((e >> 32) == 0ull) ? -INFINITY : key2f((uint32_t)(e >> 32));
            }
            const float mx = key2f((uint32_t)(sbuf[NC - 1] >> 32));
            double part = 0.0;
            float ev[CPT];
#pragma unroll
            for (int i = 0; i < CPT; ++i) { ev[i] = expf(sv[i] - mx); part += (double)ev[i]; }
            const float ssum = (float)block_sum_d(part, redd);
            double loc = 0.0;
#pragma unroll
            for (int i = 0; i < CPT; ++i) { ev[i] = ev[i] / ssum; loc += (double)ev[i]; }
            double dinc = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double n = __shfl_up(dinc, o, 64); if (lane >= o) dinc += n; }
            __syncthreads();
            if (lane == 63) redd[4 + wave] = dinc;
            __syncthreads();
            double dbase = dinc - loc;
            for (int w = 0; w < wave; ++w) dbase += redd[4 + w];
            double run = dbase;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                run += (double)ev[i];
                const bool remove = ((float)run <= a.top_p_thr) && (CPT * tid + i != NC - 1);
                if (si[i] < NV) rm[si[i]] = remove ? 1 : 0;              // pads carry indices >= NV
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3);
                if (x[i] > -INFINITY && rm[v]) x[i] = -INFINITY;
            }
        } else {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3);
            sbuf[v] = (v < V) ? (((unsigned long long)f2key(x[i]) << 32) | (unsigned)v) : (unsigned long long)(unsigned)v;
        }
        __syncthreads();
        for (int k = 2; k <= NV; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                for (int it = 0; it < NV / 512; ++it) {
                    const int p = tid + 256 * it;                    // pair index 0..2047
                    const int i0 = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    const int i1 = i0 | j;
                    const bool up = (i0 & k) == 0;
                    const unsigned long long A = sbuf[i0], Bv = sbuf[i1];
                    if ((A > Bv) == up) { sbuf[i0] = Bv; sbuf[i1] = A; }
                }
                __syncthreads();
            }
        }
        // sorted ascending; thread owns positions 16*tid .. 16*tid+15.  softmax over the sorted row:
        float sv[VPT]; int si[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const unsigned long long e = sbuf[VPT * tid + i];
            si[i] = (int)(e & 0xFFFFFFFFu);
            sv[i] = ((e >> 32) == 0ull) ? -INFINITY : key2f((uint32_t)(e >> 32));   // pads behave as -inf
        }
        const float mx = key2f((uint32_t)(sbuf[NV - 1] >> 32));
        double part = 0.0;
        float ev[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) { ev[i] = expf(sv[i] - mx); part += (double)ev[i]; }
        const float ssum = (float)block_sum_d(part, redd);
        // inclusive scan in double over the sorted probabilities
        double loc = 0.0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { ev[i] = ev[i] / ssum; loc += (double)ev[i]; }
        double dinc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double n = __shfl_up(dinc, o, 64); if (lane >= o) dinc += n; }
        __syncthreads();
        if (lane == 63) redd[4 + wave] = dinc;
        __syncthreads();
        double dbase = dinc - loc;
        for (int w = 0; w < wave; ++w) dbase += redd[4 + w];
        double run = dbase;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            run += (double)ev[i];
            const bool remove = ((float)run <= a.top_p_thr) && (VPT * tid + i != NV - 1);
            rm[si[i]] = remove ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3);
            if (v < V && rm[v]) x[i] = -INFINITY;
        }
        }
    }

    if (a.dbg_masked) {
        float* pd = a.dbg_masked + ((size_t)b * a.l + tok) * V;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3); if (v < V) pd[v] = x[i]; }
    }

    // ---- p = softmax(x); draw = argmax(p / q)
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPT; ++i) mloc = fmaxf(mloc, x[i]);
    const float mx = block_max(mloc, redf);
    double part = 0.0;
    float ev[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) { ev[i] = expf(x[i] - mx); part += (double)ev[i]; }
    const float ssum = (float)block_sum_d(part, redd);

    float best = -1.0f; int besti = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v0 = 4 * (tid + 256 * j);
        if (v0 >= V) continue;
        float qv[4];
        if (a.q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.q + ((size_t)b * a.l + tok) * V + v0);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[e] = t4[e];
        } else {
            uint32_t r4[4];
            philox4x32_10((uint32_t)(v0 >> 2), (uint32_t)tok, a.image_offset + (uint32_t)b, a.draw, a.k0, a.k1, r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[e] = -logf(((float)(r4[e] >> 9) + 0.5f) * 1.1920928955078125e-07f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ratio = (ev[4 * j + e] / ssum) / qv[e];
            if (ratio > best) { best = ratio; besti = v0 + e; }     // ascending v within a thread keeps the first max
        }
    }
    // block argmax with smallest-index tie break
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if (lane == 0) { redf[wave] = best; sel[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        float bb = redf[0]; int bi = sel[0];
        for (int w = 1; w < 4; ++w) if (redf[w] > bb || (redf[w] == bb && sel[w] < bi)) { bb = redf[w]; bi = sel[w]; }
        a.ids[(size_t)b * a.ids_stride + tok] = (long long)bi;
    }
}

int cfg_sample(const float* logits, int B, int l, int V, float one_plus_t, float t, int top_k, int use_top_p, float top_p_thr,
               const float* q, uint64_t seed, uint32_t draw, uint32_t image_offset, long long* ids, int ids_stride, float* dbg_masked,
               hipStream_t stream) {
    SDVAR_CHECK_ARG(logits && ids && B > 0 && l > 0, "cfg_sample: null/empty");
    SDVAR_CHECK_ARG(V > 0 && V <= NV && V % 4 == 0, "cfg_sample: V=%d unsupported (<= %d, multiple of 4)", V, NV);
    SampleArgs a;
    a.logits = logits; a.q = q; a.ids = ids; a.B = B; a.l = l; a.V = V; a.ids_stride = ids_stride;
    a.one_plus_t = one_plus_t; a.t = t; a.top_k = top_k; a.top_p_thr = top_p_thr; a.use_top_p = use_top_p;
    a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32) ^ 0x5D5A17ABu; a.draw = draw; a.image_offset = image_offset;
    a.dbg_masked = dbg_masked;
    hipLaunchKernelGGL(cfg_sample_kernel, dim3(l, B), dim3(256), 0, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ noise_fill
__global__ void noise_fill_kernel(float* q, int B, int l, int V, uint32_t k0, uint32_t k1, uint32_t draw, uint32_t image_offset) {
    const int tok = blockIdx.x, b = blockIdx.y;
    for (int v4 = threadIdx.x; v4 < V / 4; v4 += blockDim.x) {
        uint32_t r4[4];
        philox4x32_10((uint32_t)v4, (uint32_t)tok, image_offset + (uint32_t)b, draw, k0, k1, r4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = -logf(((float)(r4[e] >> 9) + 0.5f) * 1.1920928955078125e-07f);
        *reinterpret_cast<f32x4*>(q + ((size_t)b * l + tok) * V + 4 * v4) = o;
    }
}

int noise_fill(float* q, int B, int l, int V, uint64_t seed, uint32_t draw, uint32_t image_offset, hipStream_t stream) {
    SDVAR_CHECK_ARG(q && V % 4 == 0, "noise_fill: bad args");
    hipLaunchKernelGGL(noise_fill_kernel, dim3(l, B), dim3(256), 0, stream, q, B, l, V, (uint32_t)(seed & 0xFFFFFFFFull),
                       (uint32_t)(seed >> 32) ^ 0x5D5A17ABu, draw, image_offset);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ gumbel_mix
// more_smooth=True (models/var.py:206-208, models/helpers.py:22-36): h = softmax((masked_cfg_logits * (1 + ratio) + g) / tau) @ codebook
// with g = -log(E), E ~ Exp(1) drawn AFTER the multinomial of the same stage (explicit (B, l, V) input or the Philox stream
// at draw | 0x40000000).  `masked` are the CFG logits as the sampler left them (top-k / top-p entries at -inf: the reference's
// sampler masks its argument in place, helpers.py:10,15).  One workgroup per token; fp32 like the reference.
struct GumbelArgs {
    const float* masked;      // (B, l, V)
    const float* e;           // (B, l, V) Exp(1) noise or null
    const float* codebook;    // (V, Cv)
    float* h;                 // (B, l, Cv)
    int B, l, V, Cv;
    float scale, tau;         // float32(1 + ratio), float32(tau)
    uint32_t k0, k1, draw, image_offset;
};

__global__ __launch_bounds__(256) void gumbel_mix_kernel(GumbelArgs a) {
    __shared__ float pv[NV];
    __shared__ float redf[8];
    __shared__ double redd[8];
    __shared__ float part[8][32];
    const int tid = threadIdx.x, tok = blockIdx.x, b = blockIdx.y, V = a.V;
    const size_t row = ((size_t)b * a.l + tok) * V;
    float y[VPT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v0 = 4 * (tid + 256 * j);
        if (v0 < V) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(a.masked + row + v0);
            float ev[4];
            if (a.e) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.e + row + v0);
#pragma unroll
                for (int e = 0; e < 4; ++e) ev[e] = t4[e];
            } else {
                uint32_t r4[4];
                philox4x32_10((uint32_t)(v0 >> 2), (uint32_t)tok, a.image_offset + (uint32_t)b, a.draw, a.k0, a.k1, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) ev[e] = -logf(((float)(r4[e] >> 9) + 0.5f) * 1.1920928955078125e-07f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * j + e] = __fdiv_rn(__fadd_rn(__fmul_rn(m4[e], a.scale), -logf(ev[e])), a.tau);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * j + e] = -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPT; ++i) mloc = fmaxf(mloc, y[i]);
    const float mx = block_max(mloc, redf);
    double ps = 0.0;
    float ev[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) { ev[i] = expf(y[i] - mx); ps += (double)ev[i]; }
    const float ssum = (float)block_sum_d(ps, redd);
#pragma unroll
    for (int i = 0; i < VPT; ++i) { const int v = 4 * (tid + 256 * (i >> 2)) + (i & 3); pv[v] = ev[i] / ssum; }
    __syncthreads();
    // h[c] = sum_v p[v] codebook[v][c]: thread = (slice of V, channel); masked entries have p = 0 exactly
    const int Cv = a.Cv, c = tid % 32, sl = tid / 32;
    for (int c0 = 0; c0 < Cv; c0 += 32) {
        float acc = 0.f;
        if (c0 + c < Cv) for (int v = sl; v < V; v += 8) { const float p = pv[v]; if (p != 0.f) acc = fmaf(p, a.codebook[(size_t)v * Cv + c0 + c], acc); }
        part[sl][c] = acc;
        __syncthreads();
        if (tid < 32 && c0 + tid < Cv) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += part[k][tid];
            a.h[((size_t)b * a.l + tok) * Cv + c0 + tid] = t;
        }
        __syncthreads();
    }
}

int gumbel_mix(const float* masked, int B, int l, int V, float scale, float tau, const float* e, uint64_t seed, uint32_t draw, uint32_t image_offset,
               const float* codebook, int Cv, float* h, hipStream_t stream) {
    SDVAR_CHECK_ARG(masked && codebook && h && B > 0 && l > 0, "gumbel_mix: null/empty");
    SDVAR_CHECK_ARG(V > 0 && V <= NV && V % 4 == 0 && Cv >= 1, "gumbel_mix: V=%d Cv=%d unsupported", V, Cv);
    GumbelArgs a;
    a.masked = masked; a.e = e; a.codebook = codebook; a.h = h; a.B = B; a.l = l; a.V = V; a.Cv = Cv; a.scale = scale; a.tau = tau;
    a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32) ^ 0x5D5A17ABu; a.draw = draw; a.image_offset = image_offset;
    hipLaunchKernelGGL(gumbel_mix_kernel, dim3(l, B), dim3(256), 0, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ verify_accept
// Token rule (`mode`): 0 = draft id == argmax_V target (basic_token_matching, var.py:1199-1203);
// the richer rules sketched in advanced_token_matching (var.py:1229-1243):
//   1 = draft id among the target's top-k (fewer than k vocabulary entries score strictly higher),
//   2 = KL(softmax target || softmax draft) of the token's two CFG distributions <= kl_thr.
// Besides the per-stage match counts the kernel can emit the per-token verdict and the "corrected" id
// (draft id where the rule holds, the target's argmax elsewhere) for token-level partial acceptance.
constexpr int ACC_MAX_CHUNK = 16;
struct AcceptArgs {
    const float* logits;            // (2B, lsum, V) target logits of the chunk
    const long long* draft_ids;     // draft_ids[b * ids_stride + tok_in_chunk]
    int* counts;                    // [0..n) matched per stage, [16] n_accept, [17..17+n) totals
    long long* argmax_out;          // optional (B, lsum)
    int B, lsum, V, ids_stride, n_chunk;
    int qbeg[ACC_MAX_CHUNK + 1];
    float one_plus_t[ACC_MAX_CHUNK], t[ACC_MAX_CHUNK];
    double thr;
    int mode, top_k;
    float kl_thr;
    const float* draft_logits;      // mode 2: stage j at draft_logits + dl_off[j], shaped (2B, l_j, V)
    long long dl_off[ACC_MAX_CHUNK];
    unsigned char* match_out;       // optional (B, lsum)
    long long* corrected_out;       // optional (B, lsum)
};

__global__ __launch_bounds__(256) void verify_match_kernel(AcceptArgs a) {
    __shared__ float redf[8];
    __shared__ int redi[4];
    __shared__ double redd[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = blockIdx.x, b = blockIdx.y, V = a.V;
    int st = 0;
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (tok >= a.qbeg[j]) st = j;
    const float opt = a.one_plus_t[st], tt = a.t[st];
    const float* pc = a.logits + ((size_t)b * a.lsum + tok) * V;
    const float* pu = a.logits + ((size_t)(a.B + b) * a.lsum + tok) * V;
    float x[VPT];
    float best = -INFINITY; int besti = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v0 = 4 * (tid + 256 * j);
        if (v0 < V) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(pc + v0), u = *reinterpret_cast<const f32x4*>(pu + v0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xv = __fsub_rn(__fmul_rn(opt, c[e]), __fmul_rn(tt, u[e]));
                x[4 * j + e] = xv;
                if (xv > best || besti == 0x7FFFFFFF) { best = xv; besti = v0 + e; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * j + e] = -INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) { redf[wave] = best; redi[wave] = besti; }
    __syncthreads();
    float bb = redf[0]; int bi = redi[0];
    for (int w = 1; w < 4; ++w) if (redf[w] > bb || (redf[w] == bb && redi[w] < bi)) { bb = redf[w]; bi = redi[w]; }
    const long long did = a.draft_ids[(size_t)b * a.ids_stride + tok];
    bool match;
    if (a.mode == 1) {              // top-k membership: count entries strictly above the draft token's score
        // the ids may come straight from a caller (SDVAR.advanced_token_matching): an id outside [0, V) matches nothing and is never dereferenced
        const bool valid = did >= 0 && did < (long long)V;            // block-uniform
        const long long sid = valid ? did : 0;
        const float xd = __fsub_rn(__fmul_rn(opt, pc[sid]), __fmul_rn(tt, pu[sid]));
        int above = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) above += (x[i] > xd) ? 1 : 0;
        const double tot = block_sum_d((double)above, redd);
        match = valid && tot < (double)a.top_k;
    } else if (a.mode == 2) {       // KL(p_target || p_draft) over the two CFG distributions of this token
        const int lj = a.qbeg[st + 1] - a.qbeg[st], ti = tok - a.qbeg[st];
        const float* dc = a.draft_logits + a.dl_off[st] + ((size_t)b * lj + ti) * V;
        const float* du = a.draft_logits + a.dl_off[st] + ((size_t)(a.B + b) * lj + ti) * V;
        float xd[VPT];
        float dmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v0 = 4 * (tid + 256 * j);
            if (v0 < V) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(dc + v0), u = *reinterpret_cast<const f32x4*>(du + v0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xd[4 * j + e] = __fsub_rn(__fmul_rn(opt, c[e]), __fmul_rn(tt, u[e])); dmax = fmaxf(dmax, xd[4 * j + e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xd[4 * j + e] = -INFINITY;
            }
        }
        __syncthreads();
        dmax = block_max(dmax, redf + 4);
        double st_ = 0.0, sd_ = 0.0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { st_ += (double)expf(x[i] - bb); sd_ += (double)expf(xd[i] - dmax); }
        const double zt = block_sum_d(st_, redd), zd = block_sum_d(sd_, redd + 4);
        const double lzt = log(zt), lzd = log(zd);
        double kl = 0.0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            if (x[i] == -INFINITY) continue;
            const double lt = (double)(x[i] - bb) - lzt, ld = (double)(xd[i] - dmax) - lzd;
            kl += exp(lt) * (lt - ld);
        }
        __syncthreads();
        kl = block_sum_d(kl, redd);
        match = kl <= (double)a.kl_thr;
    } else {
        match = (long long)bi == did;
    }
    if (tid == 0) {
        if (a.argmax_out) a.argmax_out[(size_t)b * a.lsum + tok] = bi;
        if (a.match_out) a.match_out[(size_t)b * a.lsum + tok] = match ? 1 : 0;
        if (a.corrected_out) a.corrected_out[(size_t)b * a.lsum + tok] = match ? did : (long long)bi;
        if (match) atomicAdd(&a.counts[st], 1);
    }
}

// n_accept = number of leading stages with float32(matched)/float32(total) >= thr (var.py:1203,1217), evaluated the
// way the reference does: a float32 mean promoted to double against the double threshold.
__global__ void accept_scan_kernel(AcceptArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int n = 0; bool alive = true;
    for (int j = 0; j < a.n_chunk; ++j) {
        const int total = a.B * (a.qbeg[j + 1] - a.qbeg[j]);
        a.counts[17 + j] = total;
        const float rate = (float)a.counts[j] / (float)total;
        if (alive && (double)rate >= a.thr) ++n; else alive = false;
    }
    a.counts[16] = n;
}

int verify_accept(const float* logits, int B, int lsum, int V, int n_chunk, const int* qbeg, const float* one_plus_t, const float* t,
                  const long long* draft_ids, int ids_stride, double thr, int mode, int top_k, float kl_thr, const float* draft_logits,
                  const long long* dl_off, int* counts, long long* argmax_out, unsigned char* match_out, long long* corrected_out, hipStream_t stream) {
    SDVAR_CHECK_ARG(logits && draft_ids && counts, "verify_accept: null operand");
    SDVAR_CHECK_ARG(n_chunk >= 1 && n_chunk <= ACC_MAX_CHUNK && V % 4 == 0 && V <= NV, "verify_accept: bad chunk/V");
    SDVAR_CHECK_ARG(mode >= 0 && mode <= 2, "verify_accept: mode %d (0 top-1, 1 top-k membership, 2 KL threshold)", mode);
    SDVAR_CHECK_ARG(mode != 1 || top_k >= 1, "verify_accept: top-k membership needs k >= 1");
    SDVAR_CHECK_ARG(mode != 2 || (draft_logits && dl_off), "verify_accept: the KL rule needs the draft logits of the chunk");
    AcceptArgs a;
    a.logits = logits; a.draft_ids = draft_ids; a.counts = counts; a.argmax_out = argmax_out;
    a.B = B; a.lsum = lsum; a.V = V; a.ids_stride = ids_stride; a.n_chunk = n_chunk; a.thr = thr;
    a.mode = mode; a.top_k = top_k; a.kl_thr = kl_thr; a.draft_logits = draft_logits; a.match_out = match_out; a.corrected_out = corrected_out;
    for (int j = 0; j < n_chunk; ++j) { a.qbeg[j] = qbeg[j]; a.one_plus_t[j] = one_plus_t[j]; a.t[j] = t[j]; a.dl_off[j] = dl_off ? dl_off[j] : 0; }
    a.qbeg[n_chunk] = lsum;
    SDVAR_HIP(hipMemsetAsync(counts, 0, 40 * sizeof(int), stream));
    hipLaunchKernelGGL(verify_match_kernel, dim3(lsum, B), dim3(256), 0, stream, a);
    SDVAR_LAUNCH_CHECK();
    hipLaunchKernelGGL(accept_scan_kernel, dim3(1), dim3(64), 0, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ cfg_combine
// out (B, lsum, V) = (1 + t_j) * cond - t_j * uncond per stage j of a verified chunk, with torch's roundings (var.py:1062-1067): what
// SDVAR.target_verify_batch hands back to a caller that drives the reference's step functions itself.
__global__ __launch_bounds__(256) void cfg_combine_kernel(AcceptArgs a, float* __restrict__ out) {
    const int tok = blockIdx.x, b = blockIdx.y, V = a.V;
    int st = 0;
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (tok >= a.qbeg[j]) st = j;
    const float opt = a.one_plus_t[st], tt = a.t[st];
    const float* pc = a.logits + ((size_t)b * a.lsum + tok) * V;
    const float* pu = a.logits + ((size_t)(a.B + b) * a.lsum + tok) * V;
    float* po = out + ((size_t)b * a.lsum + tok) * V;
    for (int v0 = 4 * threadIdx.x; v0 < V; v0 += 1024) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(pc + v0), u = *reinterpret_cast<const f32x4*>(pu + v0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fsub_rn(__fmul_rn(opt, c[e]), __fmul_rn(tt, u[e]));
        *reinterpret_cast<f32x4*>(po + v0) = o;
    }
}

int cfg_combine(const float* logits, int B, int lsum, int V, int n_chunk, const int* qbeg, const float* one_plus_t, const float* t, float* out, hipStream_t stream) {
    SDVAR_CHECK_ARG(logits && out, "cfg_combine: null operand");
    SDVAR_CHECK_ARG(n_chunk >= 1 && n_chunk <= ACC_MAX_CHUNK && V % 4 == 0 && B >= 1 && lsum >= 1, "cfg_combine: bad chunk/V");
    AcceptArgs a{};
    a.logits = logits; a.B = B; a.lsum = lsum; a.V = V; a.n_chunk = n_chunk;
    for (int j = 0; j < n_chunk; ++j) { a.qbeg[j] = qbeg[j]; a.one_plus_t[j] = one_plus_t[j]; a.t[j] = t[j]; }
    a.qbeg[n_chunk] = lsum;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(lsum, B), dim3(256), 0, stream, a, out);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar
