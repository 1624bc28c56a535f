// VQVAE decoder f_hat -> image (the caller side of the sampler: vae.fhat_to_img at /root/reference/models/vqvae.py:62-63,
// Decoder at models/basic_vae.py:163-226, ResnetBlock :47-73, AttnBlock :76-103, Upsample2x :24-30) as one C-ABI call.
//
// Layouts as in conv.hip: fp32 activations are dense channel-last rows [B H W][C]; GEMM operands are K-blocked bf16x3 planes over
// padded pixel rows (zero frame, guard rows).  Per layer:
//     GroupNorm statistics      gn_partial_kernel + gn_finalize_kernel  (fp32 partial sums per thread, fp64 across threads)
//     GN * gamma + beta, SiLU, optional nearest 2x up-sampling, exact split into planes      prep_planes_kernel (one pass)
//     3x3 / 1x1 convolution + bias (+ residual)                                              conv.hip
//     single-head attention over the H W tokens of the 16x16 levels                          vae_attn_kernel
//     norm_out + SiLU + conv_out (160 -> 3) + clamp                                          convout_partial / convout_gather
// All kernels here are HBM-bound row kernels (16-byte accesses, channel-last rows are contiguous).
#include <math.h>

#include <vector>

#include "../../include/sdvar_hip.h"
#include "common.h"

namespace sdvar {

int conv_weight_planes(const float* w, uint16_t* planes, int Cout, int Cin, int taps, size_t plane_stride, int pfmt, const float* scale, hipStream_t stream);
int weight_scale_f16(const float* w, size_t n, float* sc, hipStream_t stream);
int conv_planes(const uint16_t* X, size_t xps, size_t x_rows, int x_row0, const uint16_t* W, size_t wps, int pfmt, const float* wsi, const float* bias, const float* res,
                float* out, int B, int H, int Wd, int N, int Cin, int taps, float* ws, size_t ws_floats, int force_split, double* gn_part, int* gn_done, int up_phase,
                size_t w_phase_stride, hipStream_t stream);
int upconv_weights(const float* w, float* weff, int Cout, int Cin, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------- layout helpers
// (B, C, H, W) fp32 -> channel-last rows [B H W][C]
__global__ __launch_bounds__(256) void rows_from_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t row = i / C;
        const int x = (int)(row % W), y = (int)((row / W) % H), b = (int)(row / ((size_t)W * H));
        out[i] = in[(((size_t)b * C + c) * H + y) * W + x];
    }
}

// ---------------------------------------------------------------------------------------------------- GroupNorm statistics
// 32 groups, eps 1e-6 (basic_vae.py:20).  Grid (chunks, B), 320 threads: thread = (pixel lane, channel quad); requires (C/4) | 320.
__global__ __launch_bounds__(320) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int C, int H, int W, int rows_per_chunk) {
    __shared__ float sm[2 * 1280];               // [sum | sumsq][pixel lane][channel]: lanes * C = 320 * 4
    const int b = blockIdx.y, chunk = blockIdx.x, nq = C >> 2, npl = 320 / nq;
    const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
    const int y0 = chunk * rows_per_chunk, y1 = min(H, y0 + rows_per_chunk);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    const int npix = (y1 - y0) * W;
    for (int p = pl; p < npix; p += npl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + y0) * W + p) * C + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
    }
    const int NL = npl * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[pl * C + 4 * q + e] = s[e]; sm[NL + pl * C + 4 * q + e] = ss[e]; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x, cpg = C / 32;
        double a = 0.0, a2 = 0.0;
        for (int l = 0; l < npl; ++l)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += (double)sm[l * C + c]; a2 += (double)sm[NL + l * C + c]; }
        double* o = part + (((size_t)b * gridDim.x + chunk) * 32 + g) * 2;
        o[0] = a; o[1] = a2;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats, int nchunk, double count, double eps) {
    __shared__ double sm[2][8][32];
    const int b = blockIdx.x, g = threadIdx.x & 31, sl = threadIdx.x >> 5;      // 8 slices of the chunk list per group, combined in slice order
    double a = 0.0, a2 = 0.0;
    for (int c = sl; c < nchunk; c += 8) { const double* p = part + (((size_t)b * nchunk + c) * 32 + g) * 2; a += p[0]; a2 += p[1]; }
    sm[0][sl][g] = a; sm[1][sl][g] = a2;
    __syncthreads();
    if (sl) return;
    for (int i = 1; i < 8; ++i) { a += sm[0][i][g]; a2 += sm[1][i][g]; }
    const double mean = a / count, var = fmax(a2 / count - mean * mean, 0.0);
    stats[((size_t)b * 32 + g) * 2] = (float)mean;
    stats[((size_t)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + eps));
}

// ---------------------------------------------------------------------------------------------------- planes producer
// in: fp32 rows [B Hi Wi][C] of a (B, C, Hi, Wi) tensor.  out: planes [3][C/32][G + M_out + G][32] of the (B, C, Ho, Wo) tensor,
// Ho = Hi << up.  mode bit 0: GroupNorm (stats, gamma, beta); bit 1: SiLU.  Thread = (output row incl. guards, 8 channels).
// Index arithmetic (round 4): the flat index is 32-bit and the three divisions (by C / 8, W + 2, H + 2) are multiply-high + shift with host-made magic numbers
// (Granlund / Montgomery, the branch-free form) - the 64-bit / and % of the first version cost more vector instructions per thread than the normalisation,
// the SiLU and the split together (the kernel ran at 1.9 TB/s of the ~3 GB a decode moves through it).
struct FastDiv { uint32_t mul, sh; };        // n / d = (((n - t) >> 1) + t) >> sh, t = mulhi(n, mul); d >= 2
static FastDiv fastdiv_make(uint32_t d) {
    FastDiv f{0u, 0u};
    uint32_t l = 31;
    while (!((d >> l) & 1u)) --l;             // floor(log2 d)
    if ((d & (d - 1)) == 0) { f.mul = 0; f.sh = l - 1; return f; }
    const uint64_t num = (uint64_t)1 << (32 + l);
    uint64_t m = num / d;
    const uint64_t rem = num - m * d;
    m += m;
    const uint64_t twice = rem + rem;
    if (twice >= d || twice < rem) m += 1;
    f.mul = (uint32_t)(m + 1); f.sh = l;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f) {
    const uint32_t t = __umulhi(n, f.mul);
    return (((n - t) >> 1) + t) >> f.sh;
}

__global__ __launch_bounds__(256) void prep_planes_kernel(const float* __restrict__ in, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, uint16_t* __restrict__ outp, size_t ops, int B, int C, int Hi, int Wi,
                                                          int up, int mode, int G, int pfmt, FastDiv dc8, FastDiv dw, FastDiv dh, FastDiv dg) {
    const int Ho = Hi << up, Wo = Wi << up, w2o = Wo + 2, h2o = Ho + 2;
    const uint32_t Mo = (uint32_t)B * h2o * w2o, R = Mo + 2u * (uint32_t)G;
    const uint32_t c8n = (uint32_t)C >> 3, cpg = C / 32;
    const uint32_t total = R * c8n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t r = fastdiv(i, dc8), c8 = i - r * c8n;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bool live = false;
        if (r >= (uint32_t)G && r < (uint32_t)G + Mo) {
            const uint32_t row = r - G;
            const uint32_t ry = fastdiv(row, dw), b = fastdiv(ry, dh);
            const int x = (int)(row - ry * w2o) - 1, y = (int)(ry - b * h2o) - 1;
            if (x >= 0 && x < Wo && y >= 0 && y < Ho) {
                live = true;
                const float* p = in + (((size_t)b * Hi + (y >> up)) * Wi + (x >> up)) * C + 8 * c8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
                if (mode & 1) {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 8 * c8), g1 = *reinterpret_cast<const f32x4*>(gamma + 8 * c8 + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + 8 * c8), b1 = *reinterpret_cast<const f32x4*>(beta + 8 * c8 + 4);
                    const float* st = stats + (size_t)b * 64;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t g = cpg == 1 ? 8 * c8 + e : fastdiv(8 * c8 + e, dg);
                        const f32x2p mr = *reinterpret_cast<const f32x2p*>(st + 2 * g);                 // {mean, rstd} of the group
                        v[e] = ((v[e] - mr[0]) * mr[1]) * (e < 4 ? g0[e & 3] : g1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
                    }
                }
                if (mode & 2) {          // SiLU = x sigmoid(x) = x / (1 + 2^(-x log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp each), as the GELU of gemm_f16x2.hip: libm's expf and the
#pragma unroll                          // IEEE division cost ~30 vector instructions per value in a kernel that moves 8 bytes per value
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[e] * -1.4426950408889634f));
                }
            }
        }
        u32x4 a = {0, 0, 0, 0}, bq = {0, 0, 0, 0}, cq = {0, 0, 0, 0};
        const size_t o = ((size_t)(c8 >> 2) * R + r) * 32 + 8 * (c8 & 3);
        if (pfmt == PLANES_F16X2) {          // two fp16 planes (gemm_f16x2.hip): h = fp16(v), l = fp16(v - h)
            if (live) {                   // packed split (common.h): finite values saturate, NaN stays NaN; the clamp itself only where a lane of the wave needs it
                float m = fmaxf(absmax3(v[0], v[1], v[2]), absmax3(v[3], v[4], v[5]));
                m = fmaxf(m, fmaxf(fabsf(v[6]), fabsf(v[7])));
                const bool clampw = __builtin_amdgcn_ballot_w64(m > 65504.0f) != 0ull;          // among the live lanes of the wave
                if (clampw) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = clamp_f16_range(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { uint32_t hw, lw; split2h_pk_raw(v[2 * e], v[2 * e + 1], hw, lw); a[e] = hw; bq[e] = lw; }
            }
            *reinterpret_cast<u32x4*>(outp + o) = a; *reinterpret_cast<u32x4*>(outp + ops + o) = bq;
            continue;
        }
        if (live) split8_packed(v, a, bq, cq);
        *reinterpret_cast<u32x4*>(outp + o) = a; *reinterpret_cast<u32x4*>(outp + ops + o) = bq; *reinterpret_cast<u32x4*>(outp + 2 * ops + o) = cq;
    }
}
static int launch_prep_planes(const float* in, const float* stats, const float* gamma, const float* beta, uint16_t* outp, size_t ops, int B, int C, int Hi, int Wi, int up, int mode,
                              int G, int pfmt, hipStream_t s) {
    const size_t rows = (size_t)B * ((Hi << up) + 2) * ((Wi << up) + 2) + 2 * (size_t)G, total = rows * (C / 8);
    SDVAR_CHECK_ARG(total < ((size_t)1 << 31) && C >= 16 && C % 32 == 0, "vae prep: %zu (row, 8-channel) items exceed the 32-bit index range, or C = %d", total, C);
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(prep_planes_kernel, dim3(grid), dim3(256), 0, s, in, stats, gamma, beta, outp, ops, B, C, Hi, Wi, up, mode, G, pfmt, fastdiv_make((uint32_t)C / 8),
                       fastdiv_make((uint32_t)(Wi << up) + 2), fastdiv_make((uint32_t)(Hi << up) + 2), fastdiv_make(C / 32 >= 2 ? (uint32_t)C / 32 : 2u));
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ---------------------------------------------------------------------------------------------------- AttnBlock core
// qkv: rows [B N][3C] (q | k | v), N = H W tokens per image.  out: rows [B N][C]:
//     out[i] = sum_j softmax_j(q_i . k_j / sqrt(C)) v_j                         basic_vae.py:92-101
// Workgroup = 16 queries of one image, 256 threads, fp32 FMA with 4 x 4 register blocking:
//   scores   thread = (4 queries, 4 keys of a 256-key pass); q and k are staged through LDS in 32-channel slabs, transposed
//            ([channel][query], [channel][key]) so that one ds_read_b128 each feeds 16 FMAs
//   softmax  one wave per 4 queries over the [key][query] score array in LDS
//   output   thread = (4 queries, 4 channels); v rows are read straight from global memory (coalesced), p from LDS
constexpr int VA_QT = 16, VA_SP = VA_QT + 4;     // score rows padded to 20 floats (16-byte aligned rows)
typedef float f32x2v __attribute__((ext_vector_type(2)));
// GPS = true: the probabilities of the workgroup's 16 queries live in a global scratch row block instead of LDS - token maps beyond ~1900 tokens (the 64 x 64
// latent of the 1024^2 ladder: 4096 tokens x 80 bytes = 320 KB).  Every phase of the kernel is separated by __syncthreads(), a workgroup's waves share one CU
// and its L1, so the same code runs on either pointer; this path is about reach, not speed.
// The same attention on the fp32 matrix cores (round 4; v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate - no operand splitting needed at 1.3 GFLOP per
// launch), for N = 256 / 1024 tokens and C % 64 == 0.  Workgroup = 16 queries of one image, 8 waves:
//   scores   wave w owns keys [w N / 8, (w + 1) N / 8): S = Q K^T in 16 x 16 tiles; per 16 channels a lane loads ONE float4 of its query row and one of each key row
//            (channels 16 s + 4 (l >> 4) ..) and feeds four MFMAs - the k index of MFMA j is channel 16 s + 4 g + j, the same permutation on both operands
//   softmax  as in vae_attn_kernel (expf of the shifted scores, sum, scale), one wave per 2 queries, probabilities in LDS [16][N + 4]
//   output   wave w owns 64-channel groups w, w + 8, ...: O = P V; per 16 keys one float4 of P from LDS and four float4 of V rows (channels 4 j .. 4 j + 3 of the group:
//            column tile e of the group holds channels {4 j + e}), sixteen MFMAs; a lane ends with 4 consecutive channels of 4 queries: float4 stores
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int NKT>          // key tiles of 16 per wave = N / 128
__global__ __launch_bounds__(512) void vae_attn_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int N) {
    extern __shared__ __attribute__((aligned(16))) float vps[];      // [16][N + 4]
    const int b = blockIdx.y, q0 = blockIdx.x * 16, tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int SP = N + 4;
    const float* base = qkv + (size_t)b * N * 3 * C;
    const float scale = 1.0f / sqrtf((float)C);
    {
        f32x4m acc[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) acc[t] = f32x4m{0.f, 0.f, 0.f, 0.f};
        const float* pq = base + (size_t)(q0 + j) * 3 * C + 4 * g;
        const float* pk = base + (size_t)(wave * (N / 8) + j) * 3 * C + C + 4 * g;
#pragma unroll 4
        for (int c0 = 0; c0 < C; c0 += 16) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(pq + c0);
            f32x4 kv[NKT];
#pragma unroll
            for (int t = 0; t < NKT; ++t) kv[t] = *reinterpret_cast<const f32x4*>(pk + (size_t)(16 * t) * 3 * C + c0);
#pragma unroll
            for (int t = 0; t < NKT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[e], kv[t][e], acc[t], 0, 0, 0);
        }
        // D: lane holds S[query 4 g + r][key tile column j]
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) vps[(4 * g + r) * SP + wave * (N / 8) + 16 * t + j] = acc[t][r] * scale;
    }
    __syncthreads();
    for (int qi = 2 * wave; qi < 2 * wave + 2; ++qi) {
        float* row = vps + qi * SP;
        float m = -INFINITY;
        for (int k = lane; k < N; k += 64) m = fmaxf(m, row[k]);
        m = wave_max(m);
        float sum = 0.f;
        for (int k = lane; k < N; k += 64) { const float e = expf(row[k] - m); row[k] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int k = lane; k < N; k += 64) row[k] *= inv;
    }
    __syncthreads();
    const float* pp = vps + j * SP + 4 * g;                              // A operand: P[query j][key k0 + 4 g + e]
    for (int grp = wave; grp < C / 64; grp += 8) {
        f32x4m acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = f32x4m{0.f, 0.f, 0.f, 0.f};
        const float* pv = base + 2 * C + 64 * grp + 4 * j;                // B operand: V[key][channels 64 grp + 4 j .. + 3]
#pragma unroll 4
        for (int k0 = 0; k0 < N; k0 += 16) {
            const f32x4 pr = *reinterpret_cast<const f32x4*>(pp + k0);
            f32x4 vv[4];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) vv[mm] = *reinterpret_cast<const f32x4*>(pv + (size_t)(k0 + 4 * g + mm) * 3 * C);
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[mm], vv[mm][e], acc[e], 0, 0, 0);
        }
        // D of column tile e: lane holds O[query 4 g + r][channel 64 grp + 4 j + e]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + 4 * g + r;
            *reinterpret_cast<f32x4*>(out + ((size_t)b * N + qi) * C + 64 * grp + 4 * j) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        }
    }
}

template <bool GPS>
__global__ __launch_bounds__(256) void vae_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int N, float* __restrict__ gps) {
    extern __shared__ __attribute__((aligned(16))) float vsm[];
    float* ks = vsm;                       // [32][256]
    float* qs = vsm + 32 * 256;            // [32][VA_QT]
    float* ps = GPS ? gps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)N * VA_SP : qs + 32 * VA_QT;           // [N][VA_SP]
    const int b = blockIdx.y, q0 = blockIdx.x * VA_QT, tid = threadIdx.x;
    const int tq = tid >> 6, tk = tid & 63;
    const float* base = qkv + (size_t)b * N * 3 * C;
    const float scale = 1.0f / sqrtf((float)C);
    for (int k0 = 0; k0 < N; k0 += 256) {
        f32x2v acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2v{0.f, 0.f}; acc[i][1] = f32x2v{0.f, 0.f}; }
        for (int c0 = 0; c0 < C; c0 += 32) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {           // k slab: thread = key, 8 x float4 along the channels
                const int key = k0 + tid;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (key < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)key * 3 * C + C + c0 + 4 * it);
#pragma unroll
                for (int e = 0; e < 4; ++e) ks[(4 * it + e) * 256 + tid] = v[e];
            }
            if (tid < 128) {                          // q slab
                const int qi = tid >> 3, c4 = tid & 7;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q0 + qi < N) v = *reinterpret_cast<const f32x4*>(base + (size_t)(q0 + qi) * 3 * C + c0 + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) qs[(4 * c4 + e) * VA_QT + qi] = v[e];
            }
            __syncthreads();
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(ks + c * 256 + 4 * tk);
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + c * VA_QT + 4 * tq);
                const f32x2v k01 = {kv[0], kv[1]}, k23 = {kv[2], kv[3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2v qq = {qv[i], qv[i]};
                    acc[i][0] = __builtin_elementwise_fma(qq, k01, acc[i][0]);
                    acc[i][1] = __builtin_elementwise_fma(qq, k23, acc[i][1]);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int key = k0 + 4 * tk + kk;
            if (key < N) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = acc[i][kk >> 1][kk & 1] * scale;
                *reinterpret_cast<f32x4*>(ps + key * VA_SP + 4 * tq) = o;
            }
        }
    }
    __syncthreads();
    {   // softmax over the keys: wave w owns queries 4w .. 4w+3
        const int lane = tid & 63, wave = tid >> 6;
        for (int qi = 4 * wave; qi < 4 * wave + 4; ++qi) {
            float m = -INFINITY;
            for (int j = lane; j < N; j += 64) m = fmaxf(m, ps[j * VA_SP + qi]);
            m = wave_max(m);
            float sum = 0.f;
            for (int j = lane; j < N; j += 64) { const float e = expf(ps[j * VA_SP + qi] - m); ps[j * VA_SP + qi] = e; sum += e; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int j = lane; j < N; j += 64) ps[j * VA_SP + qi] *= inv;
        }
    }
    __syncthreads();
    const int nq4 = C >> 2;                             // channel quads
    for (int item = tid; item < 4 * nq4; item += 256) {
        const int g = item / nq4, cq = item - g * nq4;  // query group, channel quad: consecutive threads read consecutive channels
        f32x2v acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2v{0.f, 0.f}; acc[i][1] = f32x2v{0.f, 0.f}; }
        const float* vp = base + 2 * C + 4 * cq;
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + (size_t)j * 3 * C);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(ps + j * VA_SP + 4 * g);
            const f32x2v v01 = {vv[0], vv[1]}, v23 = {vv[2], vv[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2v pp = {pv[i], pv[i]};
                acc[i][0] = __builtin_elementwise_fma(pp, v01, acc[i][0]);
                acc[i][1] = __builtin_elementwise_fma(pp, v23, acc[i][1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qi = q0 + 4 * g + i;
            if (qi < N) *reinterpret_cast<f32x4*>(out + ((size_t)b * N + qi) * C + 4 * cq) = f32x4{acc[i][0][0], acc[i][0][1], acc[i][1][0], acc[i][1][1]};
        }
    }
}

// ---------------------------------------------------------------------------------------------------- conv_out
// y = conv3x3(silu(GN(x)), w (3, C, 3, 3)) + b, clamped to [-1, 1] (vqvae.py:63).  Pass 1: t[row][tap*3 + o] = sum_c act[row][c] w[o][c][tap]
// (thread = padded row, activations staged through LDS in 32-channel slabs, weights are wave-uniform scalar loads);
// pass 2: out[b][o][y][x] = b[o] + sum_tap t[row + shift(tap)][tap*3 + o].
__global__ __launch_bounds__(256) void convout_partial_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ wt /* [C][28]: k = tap*3+o */,
                                                              float* __restrict__ t, int B, int C, int H, int W) {
    __shared__ float tile[256 * 33];
    const int cpg = C / 32;
    const size_t M = (size_t)B * H * W;
    const size_t r0 = (size_t)blockIdx.x * 256;
    const int tid = threadIdx.x;
    const size_t row = r0 + tid;
    const bool live = row < M;
    const int b = live ? (int)(row / ((size_t)H * W)) : 0;
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
        // coalesced load of rows r0 .. r0+255, channels c0 .. c0+31: thread -> (row = i / 8, 4 channels)
        for (int i = tid; i < 256 * 8; i += 256) {
            const int rr = i >> 3, cq = (i & 7) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r0 + rr < M) v = *reinterpret_cast<const f32x4*>(x + (r0 + rr) * C + c0 + cq);
            tile[rr * 33 + cq] = v[0]; tile[rr * 33 + cq + 1] = v[1]; tile[rr * 33 + cq + 2] = v[2]; tile[rr * 33 + cq + 3] = v[3];
        }
        __syncthreads();
        if (live) {
            for (int cc = 0; cc < 32; ++cc) {
                const int c = c0 + cc, g = c / cpg;
                const float mean = stats[((size_t)b * 32 + g) * 2], rstd = stats[((size_t)b * 32 + g) * 2 + 1];
                float v = ((tile[tid * 33 + cc] - mean) * rstd) * gamma[c] + beta[c];
                v = v / (1.0f + expf(-v));
                const float* w = wt + (size_t)c * 28;
#pragma unroll
                for (int k = 0; k < 27; ++k) acc[k] += v * w[k];
            }
        }
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < 27; ++k) t[row * 28 + k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void convout_gather_kernel(const float* __restrict__ t, const float* __restrict__ bias, float* __restrict__ img, int B, int H, int W) {
    const size_t total = (size_t)B * 3 * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), o = (int)((i / ((size_t)W * H)) % 3), b = (int)(i / ((size_t)3 * W * H));
        float acc = bias[o];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += t[(((size_t)b * H + yy) * W + xx) * 28 + tap * 3 + o];
        }
        img[i] = fminf(fmaxf(acc, -1.0f), 1.0f);
    }
}

// conv_out weight (3, C, 3, 3) -> [C][28] with k = tap * 3 + o
__global__ void convout_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 28) return;
    const int c = i / 28, k = i % 28;
    wt[i] = (k < 27) ? w[((size_t)(k % 3) * C + c) * 9 + k / 3] : 0.f;
}

}  // namespace sdvar

using namespace sdvar;

// ======================================================================================================== decoder object
namespace {

template <typename T>
int vmalloc(T** p, size_t n) {
    *p = nullptr;
    SDVAR_HIP(hipMalloc((void**)p, n * sizeof(T)));
    return SDVAR_OK;
}
#define VAE_TRY(call) do { int rc_ = (call); if (rc_ != SDVAR_OK) return rc_; } while (0)

struct ConvW { uint16_t* wp = nullptr; size_t wps = 0; const float* bias = nullptr; int cin = 0, cout = 0, taps = 0;   // taps = 4: four phase weight sets, NPL wps apart
               float* wsc = nullptr; };                                          // f16x2: device {2^S, 2^-S, scratch, -}, the weight scale of the launch
struct NormW { const float* gamma = nullptr; const float* beta = nullptr; int C = 0; };
struct ResW { NormW n1, n2; ConvW c1, c2, sc; bool has_sc = false; };
struct AttnW { NormW n; ConvW qkv, proj; };
struct Level { std::vector<ResW> blocks; std::vector<AttnW> attns; ConvW up, up9; bool has_up = false; };   // up: 4 phase convs; up9: the plain 3x3 form

}  // namespace

struct sdvar_vae {
    sdvar_vae_desc d;
    int nlev, H0;
    ConvW post_quant, conv_in;
    ResW mid1, mid2; AttnW mid_attn;
    std::vector<Level> levels;            // in execution order (deepest first)
    NormW norm_out; float* wt_out = nullptr; const float* b_out = nullptr; int c_out = 0;
    bool bound = false;
    // workspaces
    float *fa = nullptr, *fb = nullptr, *fc = nullptr, *t27 = nullptr, *ws = nullptr, *stats = nullptr;
    double* part = nullptr;
    uint16_t *p1 = nullptr, *p2 = nullptr;
    size_t f_floats = 0, p1_elems = 0, p2_elems = 0, ws_floats = 0;
    int pfmt = PLANES_F16X2, npl = 2;     // operand plane format of the convolutions (desc.plane_format: 0 / 2 = f16x2, 3 = bf16x3)
    std::vector<void*> owned;
};

static int guard_rows(int W) { return (W + 3 + 15) / 16 * 16; }

// planes of a (B, C, H, W) tensor: elements per plane and rows per channel block
static size_t plane_rows(int B, int H, int W) { return (size_t)B * (H + 2) * (W + 2) + 2 * (size_t)guard_rows(W); }

extern "C" {

int sdvar_vae_create(const sdvar_vae_desc* desc, sdvar_vae_t** out) {
    SDVAR_CHECK_ARG(desc && out, "vae_create: null argument");
    SDVAR_CHECK_ARG(desc->n_mult >= 1 && desc->n_mult <= 8 && desc->num_res_blocks >= 1 && desc->max_batch >= 1 && desc->latent_hw >= 1, "vae_create: bad descriptor");
    SDVAR_CHECK_ARG(desc->z_channels % 32 == 0 && desc->ch % 32 == 0, "vae_create: channel counts must be multiples of 32 (ch=%d z=%d)", desc->ch, desc->z_channels);
    for (int i = 0; i < desc->n_mult; ++i) {
        const int c = desc->ch * desc->ch_mult[i];
        SDVAR_CHECK_ARG(c <= 640 * 2 && 320 % (c / 4) == 0, "vae_create: width %d unsupported by the GroupNorm kernel ((C/4) must divide 320)", c);
    }
    SDVAR_CHECK_ARG(desc->plane_format == 0 || desc->plane_format == PLANES_F16X2 || desc->plane_format == PLANES_BF16X3, "vae_create: plane_format %d (0 / 2 = f16x2, 3 = bf16x3)", desc->plane_format);
    sdvar_vae* v = new sdvar_vae();
    v->d = *desc; v->nlev = desc->n_mult; v->H0 = desc->latent_hw;
    v->pfmt = desc->plane_format == PLANES_BF16X3 ? PLANES_BF16X3 : PLANES_F16X2; v->npl = v->pfmt == PLANES_BF16X3 ? 3 : 2;
    const int B = desc->max_batch;
    // largest fp32 activation / plane tensors over the levels (level lv runs at H0 << (nlev-1-lv) with width ch*mult[lv]; the
    // first block of a level and the up-sampling conv see the previous level's width)
    size_t fmax_ = 0, pmax = 0;
    int cprev = desc->ch * desc->ch_mult[desc->n_mult - 1];
    for (int lv = desc->n_mult - 1; lv >= 0; --lv) {
        const int H = v->H0 << (desc->n_mult - 1 - lv), c = desc->ch * desc->ch_mult[lv];
        const int cm = c > cprev ? c : cprev;
        const size_t M = (size_t)B * H * H;
        const size_t cf = (lv == desc->n_mult - 1) ? (size_t)3 * cm : (size_t)cm;       // qkv rows at the attention level
        if (M * cf > fmax_) fmax_ = M * cf;
        if (plane_rows(B, H, H) * cm > pmax) pmax = plane_rows(B, H, H) * cm;
        cprev = c;
    }
    v->f_floats = fmax_; v->p1_elems = v->npl * pmax; v->p2_elems = v->npl * pmax;
    v->ws_floats = (size_t)64 << 20;
    const int Hl = v->H0 << (desc->n_mult - 1);
    if (vmalloc(&v->fa, v->f_floats) || vmalloc(&v->fb, v->f_floats) || vmalloc(&v->fc, v->f_floats) || vmalloc(&v->p1, v->p1_elems) ||
        vmalloc(&v->p2, v->p2_elems) || vmalloc(&v->ws, v->ws_floats) || vmalloc(&v->t27, (size_t)B * Hl * Hl * 28) ||
        vmalloc(&v->stats, (size_t)B * 64) || vmalloc(&v->part, (size_t)B * ((size_t)Hl * Hl / 256 + 64) * 32 * 2)) {
        sdvar_vae_destroy(v);
        return SDVAR_ERR_HIP;
    }
    *out = v;
    return SDVAR_OK;
}

int sdvar_vae_destroy(sdvar_vae_t* v) {
    if (!v) return SDVAR_OK;
    void* bufs[] = {v->fa, v->fb, v->fc, v->p1, v->p2, v->ws, v->t27, v->stats, v->part, v->wt_out};
    for (void* p : bufs) if (p) (void)hipFree(p);
    for (void* p : v->owned) (void)hipFree(p);
    delete v;
    return SDVAR_OK;
}

}  // extern "C"

namespace {

struct Binder {
    sdvar_vae* v; const float* const* t; int n, pos; hipStream_t s; int rc;
    const float* next() { if (pos >= n) { rc = SDVAR_ERR_ARG; return nullptr; } return t[pos++]; }
    int scale_of(ConvW& c, const float* w, size_t n) {          // f16x2: the power-of-two scale of this launch's weights, computed on the device
        if (v->pfmt != PLANES_F16X2) return SDVAR_OK;
        if (hipMalloc((void**)&c.wsc, 4 * sizeof(float)) != hipSuccess) return SDVAR_ERR_HIP;
        v->owned.push_back(c.wsc);
        return weight_scale_f16(w, n, c.wsc, s);
    }
    void conv(ConvW& c, int cin, int cout, int taps) {
        const float* w = next(); const float* b = next();
        if (rc || !w || !b) { rc = SDVAR_ERR_ARG; return; }
        c.cin = cin; c.cout = cout; c.taps = taps; c.bias = b;
        c.wps = (size_t)taps * cin * cout;
        uint16_t* p = nullptr;
        if (hipMalloc((void**)&p, v->npl * c.wps * sizeof(uint16_t)) != hipSuccess) { rc = SDVAR_ERR_HIP; return; }
        v->owned.push_back(p);
        c.wp = p;
        int r = scale_of(c, w, c.wps);
        if (!r) r = conv_weight_planes(w, p, cout, cin, taps, c.wps, v->pfmt, c.wsc, s);
        if (r) rc = r;
    }
    void upconv(ConvW& c, ConvW& c9, int ch) {          // Upsample2x conv: four 2x2 phase convolutions (conv.hip upconv_weights) + the plain 3x3 form
        const int save = pos;
        conv(c9, ch, ch, 9);
        pos = save;
        const float* w = next(); const float* b = next();
        if (rc || !w || !b) { rc = SDVAR_ERR_ARG; return; }
        c.cin = ch; c.cout = ch; c.taps = 4; c.bias = b;
        c.wps = (size_t)4 * ch * ch;
        uint16_t* p = nullptr; float* weff = nullptr;
        if (hipMalloc((void**)&p, 4 * v->npl * c.wps * sizeof(uint16_t)) != hipSuccess || hipMalloc((void**)&weff, 4 * c.wps * sizeof(float)) != hipSuccess) { rc = SDVAR_ERR_HIP; return; }
        v->owned.push_back(p); v->owned.push_back(weff);
        c.wp = p;
        int r = upconv_weights(w, weff, ch, ch, s);
        if (!r) r = scale_of(c, weff, 4 * c.wps);                  // the four phase kernels run in one launch: one scale
        for (int ph = 0; ph < 4 && !r; ++ph) r = conv_weight_planes(weff + ph * c.wps, p + (size_t)ph * v->npl * c.wps, ch, ch, 4, c.wps, v->pfmt, c.wsc, s);
        if (r) rc = r;
    }
    void norm(NormW& nw, int C) { nw.gamma = next(); nw.beta = next(); nw.C = C; if (!nw.gamma || !nw.beta) rc = SDVAR_ERR_ARG; }
    void res(ResW& r, int cin, int cout) {
        norm(r.n1, cin); conv(r.c1, cin, cout, 9); norm(r.n2, cout); conv(r.c2, cout, cout, 9);
        r.has_sc = cin != cout;
        if (r.has_sc) conv(r.sc, cin, cout, 1);
    }
    void attn(AttnW& a, int C) { norm(a.n, C); conv(a.qkv, C, 3 * C, 1); conv(a.proj, C, C, 1); }
};

}  // namespace

extern "C" {

int sdvar_vae_tensor_count(const sdvar_vae_desc* d) {
    if (!d) return -1;
    int n = 2 + 2;                                         // post_quant_conv, conv_in
    n += 8 + 6 + 8;                                        // mid.block_1, mid.attn_1, mid.block_2
    int cprev = d->ch * d->ch_mult[d->n_mult - 1];
    for (int lv = d->n_mult - 1; lv >= 0; --lv) {
        const int c = d->ch * d->ch_mult[lv];
        for (int i = 0; i <= d->num_res_blocks; ++i) {
            n += 8 + (cprev != c ? 2 : 0);
            cprev = c;
            if (lv == d->n_mult - 1) n += 6;
        }
        if (lv != 0) n += 2;
    }
    return n + 2 + 2;                                      // norm_out, conv_out
}

int sdvar_vae_bind(sdvar_vae_t* v, const float* const* tensors, int32_t n_tensors, void* stream) {
    SDVAR_CHECK_ARG(v && tensors, "vae_bind: null argument");
    SDVAR_CHECK_ARG(n_tensors == sdvar_vae_tensor_count(&v->d), "vae_bind: expected %d tensors, got %d", sdvar_vae_tensor_count(&v->d), n_tensors);
    for (int i = 0; i < n_tensors; ++i) SDVAR_CHECK_ARG(tensors[i], "vae_bind: tensor %d is null", i);
    for (void* p : v->owned) (void)hipFree(p);
    v->owned.clear(); v->levels.clear();
    const sdvar_vae_desc& d = v->d;
    Binder b{v, tensors, n_tensors, 0, (hipStream_t)stream, SDVAR_OK};
    const int z = d.z_channels, ctop = d.ch * d.ch_mult[d.n_mult - 1];
    b.conv(v->post_quant, z, z, 9);
    b.conv(v->conv_in, z, ctop, 9);
    b.res(v->mid1, ctop, ctop); b.attn(v->mid_attn, ctop); b.res(v->mid2, ctop, ctop);
    int cprev = ctop;
    for (int lv = d.n_mult - 1; lv >= 0; --lv) {
        Level L;
        const int c = d.ch * d.ch_mult[lv];
        for (int i = 0; i <= d.num_res_blocks; ++i) {
            ResW r; b.res(r, cprev, c); L.blocks.push_back(r); cprev = c;
            if (lv == d.n_mult - 1) { AttnW a; b.attn(a, c); L.attns.push_back(a); }
        }
        L.has_up = lv != 0;
        if (L.has_up) b.upconv(L.up, L.up9, c);
        v->levels.push_back(L);
    }
    b.norm(v->norm_out, cprev);
    const float* wo = b.next(); v->b_out = b.next(); v->c_out = cprev;
    if (b.rc) { set_error("vae_bind: failed while binding tensor %d", b.pos); return b.rc; }
    if (v->wt_out) { (void)hipFree(v->wt_out); v->wt_out = nullptr; }
    SDVAR_HIP(hipMalloc((void**)&v->wt_out, (size_t)cprev * 28 * sizeof(float)));
    hipLaunchKernelGGL(convout_weight_kernel, dim3((cprev * 28 + 255) / 256), dim3(256), 0, (hipStream_t)stream, wo, v->wt_out, cprev);
    SDVAR_LAUNCH_CHECK();
    v->bound = true;
    return SDVAR_OK;
}

}  // extern "C"

namespace {

struct Runner {
    sdvar_vae* v; int B; hipStream_t s;
    int H = 0;                               // current resolution (square)
    float *x, *h, *t;                        // residual stream, temporary, third buffer

    const float* stats_src = nullptr;        // tensor whose GroupNorm partial sums the last convolution left in v->part ...
    int stats_chunks = 0;                    // ... as this many 256-row chunks per image

    size_t M() const { return (size_t)B * H * H; }
    int stats_of(const float* src, int C) {
        int chunks;
        if (src == stats_src) {
            chunks = stats_chunks;           // fused in the producing convolution's epilogue
        } else {
            const int nch = H < 64 ? H : 64, rpc = (H + nch - 1) / nch;
            chunks = (H + rpc - 1) / rpc;
            hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, B), dim3(320), 0, s, src, v->part, C, H, H, rpc);
            SDVAR_LAUNCH_CHECK();
        }
        stats_src = nullptr;
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, v->part, v->stats, chunks, (double)H * H * (C / 32), 1e-6);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    // planes of (optionally normalised / activated / up-sampled) src -> dst planes; returns via out params the operand geometry
    int prep(const float* src, int C, const NormW* nw, int silu, int up, uint16_t* dst, size_t dst_elems, size_t* ops, size_t* rows, int* G) {
        const int Ho = H << up;
        *G = guard_rows(Ho); *rows = plane_rows(B, Ho, Ho); *ops = *rows * (size_t)C;
        SDVAR_CHECK_ARG(v->npl * *ops <= dst_elems, "vae: plane buffer too small");
        if (nw) VAE_TRY(stats_of(src, C));
        return launch_prep_planes(src, v->stats, nw ? nw->gamma : nullptr, nw ? nw->beta : nullptr, dst, *ops, B, C, H, H, up, (nw ? 1 : 0) | (silu ? 2 : 0), *G, v->pfmt, s);
    }
    int conv(const ConvW& c, const uint16_t* xp, size_t ops, size_t rows, int G, const float* res, float* out) {
        SDVAR_CHECK_ARG(M() * c.cout <= v->f_floats, "vae: activation buffer too small");
        int done = 0;
        VAE_TRY(conv_planes(xp, ops, rows, G, c.wp, c.wps, v->pfmt, c.wsc ? c.wsc + 1 : nullptr, c.bias, res, out, B, H, H, c.cout, c.cin, c.taps, v->ws, v->ws_floats, 0,
                            v->part, &done, -1, 0, s));
        stats_src = done ? out : nullptr; stats_chunks = H * H / 256;
        return SDVAR_OK;
    }
    int resblock(const ResW& r) {        // x <- shortcut(x) + conv2(silu(gn2(conv1(silu(gn1(x))))))          basic_vae.py:62-73
        size_t ops, rows; int G;
        const float* resid = x;
        if (r.has_sc) {
            VAE_TRY(prep(x, r.c1.cin, nullptr, 0, 0, v->p2, v->p2_elems, &ops, &rows, &G));
            VAE_TRY(conv(r.sc, v->p2, ops, rows, G, nullptr, t));
            resid = t;
        }
        VAE_TRY(prep(x, r.c1.cin, &r.n1, 1, 0, v->p1, v->p1_elems, &ops, &rows, &G));
        VAE_TRY(conv(r.c1, v->p1, ops, rows, G, nullptr, h));
        VAE_TRY(prep(h, r.c2.cin, &r.n2, 1, 0, v->p1, v->p1_elems, &ops, &rows, &G));
        if (r.has_sc) { VAE_TRY(conv(r.c2, v->p1, ops, rows, G, resid, t)); float* tmp = x; x = t; t = tmp; }
        else VAE_TRY(conv(r.c2, v->p1, ops, rows, G, resid, x));
        return SDVAR_OK;
    }
    int attnblock(const AttnW& a) {       // x <- x + proj(attn(qkv(gn(x))))                                   basic_vae.py:86-103
        size_t ops, rows; int G;
        const int C = a.n.C;
        VAE_TRY(prep(x, C, &a.n, 0, 0, v->p1, v->p1_elems, &ops, &rows, &G));
        VAE_TRY(conv(a.qkv, v->p1, ops, rows, G, nullptr, h));
        const int N = H * H;
        const size_t lds_fix = ((size_t)32 * 256 + 32 * VA_QT) * sizeof(float), lds = lds_fix + (size_t)N * VA_SP * sizeof(float);
        SDVAR_CHECK_ARG(C % 32 == 0, "vae: attention over %d channels (need a multiple of 32)", C);
        const int qgroups = (N + VA_QT - 1) / VA_QT;
        static const bool no_mfma = getenv("SDVAR_VAE_ATTN_FMA") != nullptr;       // A/B runs: the fp32-FMA kernel
        if (!no_mfma && (N == 256 || N == 1024) && C % 64 == 0) {                  // 16^2 / 32^2 latents: the matrix-core kernel
            const size_t lm = (size_t)16 * (N + 4) * sizeof(float);
            if (N == 256) {
                hipLaunchKernelGGL(vae_attn_mfma_kernel<2>, dim3(N / 16, B), dim3(512), lm, s, h, t, C, N);
            } else {
                SDVAR_HIP(hipFuncSetAttribute((const void*)vae_attn_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm));
                hipLaunchKernelGGL(vae_attn_mfma_kernel<8>, dim3(N / 16, B), dim3(512), lm, s, h, t, C, N);
            }
            SDVAR_LAUNCH_CHECK();
        } else if (lds <= 160 * 1024) {
            SDVAR_HIP(hipFuncSetAttribute((const void*)vae_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(vae_attn_kernel<false>, dim3(qgroups, B), dim3(256), lds, s, h, t, C, N, (float*)nullptr);
            SDVAR_LAUNCH_CHECK();
        } else {            // probabilities in the split-K workspace (idle between the qkv and the proj convolution), as many images per launch as fit
            const size_t per_img = (size_t)qgroups * N * VA_SP;
            const int nb_max = (int)(v->ws_floats / per_img);
            SDVAR_CHECK_ARG(nb_max >= 1, "vae: attention over %d tokens needs %zu floats of scratch per image (workspace: %zu)", N, per_img, v->ws_floats);
            for (int b0 = 0; b0 < B; b0 += nb_max) {
                const int nb = B - b0 < nb_max ? B - b0 : nb_max;
                hipLaunchKernelGGL(vae_attn_kernel<true>, dim3(qgroups, nb), dim3(256), lds_fix, s, h + (size_t)b0 * N * 3 * C, t + (size_t)b0 * N * C, C, N, v->ws);
                SDVAR_LAUNCH_CHECK();
            }
        }
        if (stats_src == t) stats_src = nullptr;
        VAE_TRY(prep(t, C, nullptr, 0, 0, v->p1, v->p1_elems, &ops, &rows, &G));
        VAE_TRY(conv(a.proj, v->p1, ops, rows, G, x, x));
        return SDVAR_OK;
    }
};

}  // namespace

extern "C" {

int sdvar_vae_decode(sdvar_vae_t* v, const float* f_hat, int32_t B, float* img, void* stream) {
    SDVAR_CHECK_ARG(v && f_hat && img, "vae_decode: null argument");
    SDVAR_CHECK_ARG(v->bound, "vae_decode: weights not bound");
    SDVAR_CHECK_ARG(B >= 1 && B <= v->d.max_batch, "vae_decode: batch %d exceeds max_batch %d", B, v->d.max_batch);
    hipStream_t s = (hipStream_t)stream;
    Runner r{v, B, s, v->H0, v->fa, v->fb, v->fc};
    const int z = v->d.z_channels;
    size_t ops, rows; int G;
    {
        const size_t total = r.M() * z;
        hipLaunchKernelGGL(rows_from_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f_hat, r.h, B, z, r.H, r.H);
        SDVAR_LAUNCH_CHECK();
    }
    VAE_TRY(r.prep(r.h, z, nullptr, 0, 0, v->p1, v->p1_elems, &ops, &rows, &G));
    VAE_TRY(r.conv(v->post_quant, v->p1, ops, rows, G, nullptr, r.t));                     // vqvae.py:63 post_quant_conv
    VAE_TRY(r.prep(r.t, z, nullptr, 0, 0, v->p1, v->p1_elems, &ops, &rows, &G));
    VAE_TRY(r.conv(v->conv_in, v->p1, ops, rows, G, nullptr, r.x));                        // basic_vae.py:216 conv_in
    VAE_TRY(r.resblock(v->mid1)); VAE_TRY(r.attnblock(v->mid_attn)); VAE_TRY(r.resblock(v->mid2));
    for (const Level& L : v->levels) {
        for (size_t i = 0; i < L.blocks.size(); ++i) {
            VAE_TRY(r.resblock(L.blocks[i]));
            if (i < L.attns.size()) VAE_TRY(r.attnblock(L.attns[i]));
        }
        if (L.has_up) {                                                                    // Upsample2x: conv(interpolate(x, 2, nearest))
            const long tiles4 = 4 * (long)((r.M() + 255) / 256) * ((L.up.cout + 159) / 160);
            if (tiles4 >= 128) {                                                           // as four 2x2 phase convolutions on the input grid, one launch
                VAE_TRY(r.prep(r.x, L.up.cin, nullptr, 0, 0, v->p1, v->p1_elems, &ops, &rows, &G));
                SDVAR_CHECK_ARG(4 * r.M() * L.up.cout <= v->f_floats, "vae: activation buffer too small");
                int done = 0;
                VAE_TRY(conv_planes(v->p1, ops, rows, G, L.up.wp, L.up.wps, v->pfmt, L.up.wsc ? L.up.wsc + 1 : nullptr, L.up.bias, nullptr, r.h, B, r.H, r.H, L.up.cout,
                                    L.up.cin, 4, nullptr, 0, 0, v->part, &done, 0, v->npl * L.up.wps, s));
                r.stats_src = done ? r.h : nullptr; r.stats_chunks = 4 * (r.H * r.H / 256);
                r.H <<= 1;
            } else {                                                                       // too few tiles: 3x3 on the up-sampled planes, split along K
                VAE_TRY(r.prep(r.x, L.up9.cin, nullptr, 0, 1, v->p1, v->p1_elems, &ops, &rows, &G));
                r.H <<= 1;
                VAE_TRY(r.conv(L.up9, v->p1, ops, rows, G, nullptr, r.h));
            }
            float* tmp = r.x; r.x = r.h; r.h = tmp;
        }
    }
    // norm_out + SiLU + conv_out + clamp
    VAE_TRY(r.stats_of(r.x, v->c_out));
    const size_t M = r.M();
    hipLaunchKernelGGL(convout_partial_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, r.x, v->stats, v->norm_out.gamma, v->norm_out.beta, v->wt_out,
                       v->t27, B, v->c_out, r.H, r.H);
    SDVAR_LAUNCH_CHECK();
    const size_t total = (size_t)B * 3 * r.H * r.H;
    hipLaunchKernelGGL(convout_gather_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, s, v->t27, v->b_out, img, B, r.H, r.H);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

#define SDVAR_TRY_(call) do { int rc_ = (call); if (rc_ != SDVAR_OK) return rc_; } while (0)
/* single operators for the parity tests */
int sdvar_op_conv_weight_planes(const float* w, uint16_t* planes, int32_t Cout, int32_t Cin, int32_t taps, uint64_t plane_stride, int32_t plane_format, float* scale,
                                void* stream) {
    if (plane_format == PLANES_F16X2 && scale) SDVAR_TRY_(weight_scale_f16(w, (size_t)Cout * Cin * taps, scale, (hipStream_t)stream));
    return conv_weight_planes(w, planes, Cout, Cin, taps, (size_t)plane_stride, plane_format, scale, (hipStream_t)stream);
}
int sdvar_op_vae_prep(const float* in, const float* stats, const float* gamma, const float* beta, uint16_t* planes, uint64_t plane_stride, int32_t plane_format, int32_t B,
                      int32_t C, int32_t H, int32_t W, int32_t up, int32_t mode, int32_t guard, void* stream) {
    SDVAR_CHECK_ARG(in && planes && C % 32 == 0 && (!(mode & 1) || (stats && gamma && beta)), "vae_prep: bad arguments");
    return launch_prep_planes(in, stats, gamma, beta, planes, (size_t)plane_stride, B, C, H, W, up, mode, guard, plane_format, (hipStream_t)stream);
}
int sdvar_op_conv_planes(const uint16_t* x_planes, uint64_t x_plane_stride, uint64_t x_rows, int32_t x_row0, const uint16_t* w_planes, uint64_t w_plane_stride,
                         int32_t plane_format, const float* w_scale, const float* bias, const float* res, float* out, int32_t B, int32_t H, int32_t W, int32_t N, int32_t Cin,
                         int32_t taps, float* workspace, uint64_t workspace_floats, int32_t force_split, void* stream) {
    return conv_planes(x_planes, (size_t)x_plane_stride, (size_t)x_rows, x_row0, w_planes, (size_t)w_plane_stride, plane_format, w_scale ? w_scale + 1 : nullptr, bias, res, out,
                       B, H, W, N, Cin, taps, workspace, (size_t)workspace_floats, force_split, nullptr, nullptr, -1, 0, (hipStream_t)stream);
}

}  // extern "C"
