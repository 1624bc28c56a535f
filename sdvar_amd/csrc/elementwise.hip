// Row-wise and gather kernels around the GEMMs (all HBM-bound; one wave64 per row, 16-byte accesses).
//
//   ln_modulate        LN(x; eps=1e-6, no affine) * (1 + scale[r]) + shift[r]     basic_var.py:141,157-158 ; :172-174
//   qk_norm_append     q/k L2-normalise (F.normalize eps 1e-12), q *= exp(min(s_h, ln 100)), append k,v to the
//                      preallocated KV cache at the length cursor                   basic_var.py:101-109
//   silu_rows          SiLU(cond)                                                   basic_var.py:147 (ada_lin[0])
//   prologue           cond = class_emb[label | uncond], x0 = cond + pos_start + lvl_pos[0]   var.py:162-183
//   build_lvl_pos      lvl_embed[lvl(t)] + pos_1LC[t]                               var.py:164
//   embed_next         word_embed(next) + lvl_pos, written to both CFG rows of a (R, ltot, C) chunk input   var.py:186-188
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "common.h"

namespace sdvar {

// A split-K GEMM whose consumer is one of the row kernels below does not run a reduce pass of its own: the consumer sums the
// K-slice slabs (in slice order - deterministic) while it reads its input anyway ("launch-boundary reduce in the next
// kernel's prologue").  `PendingSplitK` describes such an unreduced GEMM result: value[m][n] = sum_s ws[s][m][n] + bias[n].
struct PendingSplitK {
    const float* ws;        // slabs [split][M][N]; null = nothing pending
    const float* bias;      // (N)
    const float* gate;      // gated residual (x += value * gate[row / rows_per_gate]) or null
    int split, rows_per_gate, gate_stride;
};

// ------------------------------------------------------------------------------------------------ ln_modulate
// x (rows, C) ; scale/shift: row r of the CFG batch = row / rows_per_img, element stride `mod_stride` between r's.
// A wave holds one row as LN_MAX_V4 float4 per lane: the kernels are instantiated for 4 / 8 / 12 (C <= 1024 / 2048 / 3072; VAR-d36-s,
// the shared_aln checkpoint of var.py:16-19, has C = 2304) so that narrow models do not pay the registers of wide ones.
constexpr int LN_MAX_C = 3072;

// Which float4 of the row a lane holds in v[i].  PAIR (C % 8 == 0, every model width; round 4): lane l holds the two ADJACENT float4 2 l, 2 l + 1 (+ 128 per pair) - eight consecutive
// values, so that an operand plane receives 16 bytes per lane and store (half the store instructions of the 8-byte form, whole 64-byte K-block segments per 4 lanes);
// otherwise float4 number l + 64 i.
template <bool PAIR>
__device__ __forceinline__ int ln_idx(int lane, int i) { return PAIR ? 2 * lane + (i & 1) + 128 * (i >> 1) : lane + 64 * i; }

// the modulation vectors of a row, float4 number ln_idx(lane, i): loaded FIRST by both kernels (these launches are latency bound at small M: every load
// that does not depend on another must be in flight with it)
template <int LN_MAX_V4, bool PAIR>
__device__ __forceinline__ void ln_load_mod(f32x4* scv, f32x4* shv, int lane, int row, const float* __restrict__ scale, const float* __restrict__ shift,
                                            int C, int rows_per_img, int mod_stride) {
    const int nv = C >> 2;
    const size_t mo = (size_t)(row / rows_per_img) * mod_stride;
    const f32x4* psc = reinterpret_cast<const f32x4*>(scale + mo);
    const f32x4* psh = reinterpret_cast<const f32x4*>(shift + mo);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (ln_idx<PAIR>(lane, i) < nv) { scv[i] = psc[ln_idx<PAIR>(lane, i)]; shv[i] = psh[ln_idx<PAIR>(lane, i)]; }
}

// LayerNorm + modulation of one row held by one wave (v[i] = float4 number lane + 64 i of the row)
template <int LN_MAX_V4, bool PAIR>
__device__ __forceinline__ void ln_row_finish(const f32x4* v, const f32x4* scv, const f32x4* shv, int lane, int row,
                                              float* __restrict__ out, uint16_t* __restrict__ outp, size_t ops, int rows, int C, float eps, int pfmt) {
    const int nv = C >> 2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (ln_idx<PAIR>(lane, i) < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int idx = ln_idx<PAIR>(lane, i);
        if (idx < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; ss += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
    f32x4* po = reinterpret_cast<f32x4*>(out + (size_t)row * C);
    if (PAIR && outp) {                      // eight consecutive values per lane and store: 16 bytes per plane
#pragma unroll
        for (int p = 0; p < LN_MAX_V4 / 2; ++p) {
            const int idx = ln_idx<PAIR>(lane, 2 * p);
            if (idx < nv) {
                float ov[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 sc = scv[2 * p + h], sh = shv[2 * p + h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[4 * h + e] = ((v[2 * p + h][e] - mean) * rstd) * (sc[e] + 1.0f) + sh[e];
                }
                const size_t o = kb_index(row, 4 * idx, rows);
                if (pfmt == PLANES_F16X2) {
                    uint2 h0, l0, h1, l1;
                    split4h_pk(ov, h0, l0); split4h_pk(ov + 4, h1, l1);
                    *reinterpret_cast<u32x4*>(outp + o) = u32x4{h0.x, h0.y, h1.x, h1.y};
                    *reinterpret_cast<u32x4*>(outp + ops + o) = u32x4{l0.x, l0.y, l1.x, l1.y};
                } else {
                    u32x4 a, b, c;
                    split8_packed(ov, a, b, c);
                    *reinterpret_cast<u32x4*>(outp + o) = a; *reinterpret_cast<u32x4*>(outp + ops + o) = b; *reinterpret_cast<u32x4*>(outp + 2 * ops + o) = c;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int idx = ln_idx<PAIR>(lane, i);
        if (idx < nv) {
            const f32x4 sc = scv[i], sh = shv[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((v[i][e] - mean) * rstd) * (sc[e] + 1.0f) + sh[e];
            if (outp) {                      // operand planes of the split-operand GEMMs (gemm_bf16x3.hip / gemm_f16x2.hip)
                const float ov[4] = {o[0], o[1], o[2], o[3]};
                store_planes4(outp, ops, kb_index(row, 4 * idx, rows), ov, pfmt);
            } else {
                po[idx] = o;
            }
        }
    }
}

// x[row] += (sum of K-slice slabs + bias) * gate for float4 number idx of the row (slabs summed in slice order); returns the new value
// DEPTH = slab loads in flight at a time: 16 in the one-workgroup-per-row kernel (latency bound, registers are free), 4 in the one-wave-per-row kernel
// (bandwidth bound at large M: occupancy matters more)
template <int DEPTH>
__device__ __forceinline__ f32x4 pending_residual(const PendingSplitK& pend, f32x4 xv, int row, int idx, int rows, int C) {
    const size_t slab = (size_t)rows * C, o = (size_t)row * C + 4 * idx;
    // bias and gate first: every load of this function is independent of the others, so the whole slab sum costs one or two memory round trips
    // (16 slabs in flight at a time), not one per group of slabs - at M = 16 the row kernel is a chain of such round trips and nothing else
    const f32x4 bb = *reinterpret_cast<const f32x4*>(pend.bias + 4 * idx);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(pend.gate + (size_t)(row / pend.rows_per_gate) * pend.gate_stride + 4 * idx);
    f32x4 acc = *reinterpret_cast<const f32x4*>(pend.ws + o);
    int k = 1;
    for (; k + DEPTH - 1 < pend.split; k += DEPTH) {        // added in slice order (the order is part of the result)
        f32x4 p[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) p[u] = *reinterpret_cast<const f32x4*>(pend.ws + (size_t)(k + u) * slab + o);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] + p[u][e];
    }
    if (k < pend.split) {                         // the remaining 1 .. DEPTH-1 slabs, all in flight together
        f32x4 p[DEPTH - 1];
#pragma unroll
        for (int u = 0; u < DEPTH - 1; ++u) if (k + u < pend.split) p[u] = *reinterpret_cast<const f32x4*>(pend.ws + (size_t)(k + u) * slab + o);
#pragma unroll
        for (int u = 0; u < DEPTH - 1; ++u)
            if (k + u < pend.split) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = acc[e] + p[u][e];
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = xv[e] + (acc[e] + bb[e]) * gg[e];
    return xv;
}

template <int LN_MAX_V4, bool PAIR>
__global__ __launch_bounds__(256) void ln_modulate_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ out, uint16_t* __restrict__ outp,
                                                          size_t ops, int rows, int C, int rows_per_img, int mod_stride, float eps, PendingSplitK pend, int pfmt) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;
    f32x4* px = reinterpret_cast<f32x4*>(x + (size_t)row * C);
    f32x4 v[LN_MAX_V4], scv[LN_MAX_V4], shv[LN_MAX_V4];
    ln_load_mod<LN_MAX_V4, PAIR>(scv, shv, lane, row, scale, shift, C, rows_per_img, mod_stride);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int idx = ln_idx<PAIR>(lane, i);
        if (idx < nv) {
            v[i] = px[idx];
            if (pend.ws) {          // finish the previous block's gated residual, written back
                v[i] = pending_residual<4>(pend, v[i], row, idx, rows, C);
                px[idx] = v[i];
            }
        }
    }
    ln_row_finish<LN_MAX_V4, PAIR>(v, scv, shv, lane, row, out, outp, ops, rows, C, eps, pfmt);
}

// Same result bit for bit, one WORKGROUP per row, for a pending split-K residual at small row counts: with one wave per row
// a 16-row stage has 16 waves on the whole chip summing up to 32 slabs each (37 us measured); here 256 threads share the
// slab sum of one row (phase 1, through LDS) and wave 0 then normalises it exactly as above.
template <int LN_MAX_V4, bool PAIR>
__global__ __launch_bounds__(256) void ln_modulate_row_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ out, uint16_t* __restrict__ outp,
                                                              size_t ops, int rows, int C, int rows_per_img, int mod_stride, float eps, PendingSplitK pend, int pfmt) {
    __shared__ f32x4 vsm[64 * LN_MAX_V4];
    const int tid = threadIdx.x, row = blockIdx.x;
    const int nv = C >> 2;
    f32x4* px = reinterpret_cast<f32x4*>(x + (size_t)row * C);
    f32x4 scv[LN_MAX_V4], shv[LN_MAX_V4];
    if (tid < 64) ln_load_mod<LN_MAX_V4, PAIR>(scv, shv, tid, row, scale, shift, C, rows_per_img, mod_stride);     // in flight with the slab loads below
    for (int idx = tid; idx < nv; idx += 256) {
        const f32x4 nvv = pending_residual<16>(pend, px[idx], row, idx, rows, C);
        px[idx] = nvv;
        vsm[idx] = nvv;
    }
    __syncthreads();
    if (tid >= 64) return;
    f32x4 v[LN_MAX_V4];
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (ln_idx<PAIR>(tid, i) < nv) v[i] = vsm[ln_idx<PAIR>(tid, i)];
    ln_row_finish<LN_MAX_V4, PAIR>(v, scv, shv, tid, row, out, outp, ops, rows, C, eps, pfmt);
}

int ln_modulate(float* x, const float* scale, const float* shift, float* out, uint16_t* outp, size_t ops, int rows, int C, int rows_per_img,
                int mod_stride, const PendingSplitK* pend, int pfmt, hipStream_t stream) {
    PendingSplitK pd{nullptr, nullptr, nullptr, 0, 1, 0};
    if (pend && pend->ws) {
        SDVAR_CHECK_ARG(pend->bias && pend->gate && pend->split >= 1 && pend->rows_per_gate > 0 && pend->gate_stride % 4 == 0, "ln_modulate: bad pending split-K descriptor");
        pd = *pend;
    }
    SDVAR_CHECK_ARG(C % 4 == 0 && C <= LN_MAX_C && rows > 0 && rows_per_img > 0, "ln_modulate: bad shape rows=%d C=%d (C <= %d)", rows, C, LN_MAX_C);
    SDVAR_CHECK_ARG(mod_stride % 4 == 0, "ln_modulate: mod_stride must be a multiple of 4");
    SDVAR_CHECK_ARG(!outp || pfmt == PLANES_BF16X3 || pfmt == PLANES_F16X2, "ln_modulate: plane format %d", pfmt);
    const bool by_row = pd.ws && rows < 1024;
    static const bool pair_off = getenv("SDVAR_LN_PAIR") && atoi(getenv("SDVAR_LN_PAIR")) == 0;          // A/B runs
    const bool pair = !pair_off && C % 8 == 0;
#define SDVAR_LN_LAUNCH2(NV4, PR)                                                                                                              \
    do {                                                                                                                                       \
        if (by_row) hipLaunchKernelGGL((ln_modulate_row_kernel<NV4, PR>), dim3(rows), dim3(256), 0, stream, x, scale, shift, out, outp, ops, rows, C, rows_per_img, mod_stride, 1e-6f, pd, pfmt); \
        else hipLaunchKernelGGL((ln_modulate_kernel<NV4, PR>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, scale, shift, out, outp, ops, rows, C, rows_per_img, mod_stride, 1e-6f, pd, pfmt);  \
    } while (0)
#define SDVAR_LN_LAUNCH(NV4) do { if (pair) SDVAR_LN_LAUNCH2(NV4, true); else SDVAR_LN_LAUNCH2(NV4, false); } while (0)
    if (C <= 1024) SDVAR_LN_LAUNCH(4);
    else if (C <= 2048) SDVAR_LN_LAUNCH(8);
    else SDVAR_LN_LAUNCH(12);
#undef SDVAR_LN_LAUNCH
#undef SDVAR_LN_LAUNCH2
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ qk_norm_append
// qkv (R*l, 3C) with the bias already added.  One wave per (row, head); lane = channel.
// q_out (R, H, l, 64); k_cache / v_cache (R, H, Lmax, 64) written at positions pos0 .. pos0+l-1.
template <typename KV>
__global__ __launch_bounds__(256) void qk_norm_append_kernel(const float* __restrict__ qkv, const float* __restrict__ scale_mul,
                                                             float* __restrict__ q_out, KV* __restrict__ k_cache,
                                                             KV* __restrict__ v_cache, int R, int l, int H, int Lmax, int pos0, PendingSplitK pend) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long long)R * l * H) return;
    const int h = (int)(item % H);
    const long long row = item / H;            // r * l + t
    const int t = (int)(row % l), r = (int)(row / l);
    const int C = H * 64;
    float q, k, v;
    if (pend.ws) {                 // the QKV GEMM left K-slice slabs: qkv[row][col] = sum_s ws[s][row][col] + bias[col]
        const size_t slab = (size_t)R * l * 3 * C, o = (size_t)row * 3 * C + h * 64 + lane;
        q = pend.ws[o]; k = pend.ws[o + C]; v = pend.ws[o + 2 * C];
        for (int s = 1; s < pend.split; ++s) { q += pend.ws[s * slab + o]; k += pend.ws[s * slab + o + C]; v += pend.ws[s * slab + o + 2 * C]; }
        q += pend.bias[h * 64 + lane]; k += pend.bias[C + h * 64 + lane]; v += pend.bias[2 * C + h * 64 + lane];
    } else {
        const float* p = qkv + (size_t)row * 3 * C + h * 64 + lane;
        q = p[0]; k = p[C]; v = p[2 * C];
    }
    // attn_l2_norm=False models (scale_mul == null, basic_var.py:71-72): q and k stay as they are and the softmax scale 0.25 / sqrt(64) = 2^-5 is
    // folded into q (exact)
    const bool l2 = scale_mul != nullptr;
    const float qn = l2 ? fmaxf(sqrtf(wave_sum(q * q)), 1e-12f) : 1.0f;
    const float kn = l2 ? fmaxf(sqrtf(wave_sum(k * k)), 1e-12f) : 1.0f;
    const float sm = l2 ? expf(fminf(scale_mul[h], 4.605170249938965f)) : 0.03125f;    // log(100) as the reference's float32 clamp
    q_out[(((size_t)r * H + h) * l + t) * 64 + lane] = l2 ? (q / qn) * sm : q * sm;
    const size_t c = (((size_t)r * H + h) * Lmax + pos0 + t) * 64 + lane;
    k_cache[c] = (KV)(l2 ? k / kn : k);  // fp16 cache: round-to-nearest-even, like torch's .half()
    v_cache[c] = (KV)v;
}

// Formats 3 / 4 of the cache (attention_f16x2.hip: two fp16 planes per value, or ONE fp16 plane = the fp16 KV cache of BASELINE config P4): K planes AND V planes
// [R][H][NP][Lp][64], one 128-byte row per position (the attention kernel takes its V^T fragments with ds_read_b64_tr_b16: no transposed copy, no LDS pass here).
// Same arithmetic as qk_norm_append_kernel; format 3 holds the fp32 values to 2^-22, format 4 holds fp16(k), fp16(v) rounded to nearest even like torch's .half().
// thread = (position, channel group of 8): the whole append is in flight at once with 16-byte loads (six per thread and slab); the q / k norms are an in-thread
// sum of 8 squares + three shuffle steps inside the 8-lane group.  Used when the QKV launch was split along K (small M) - an unsplit launch finishes q, k, v in
// its own epilogue (gemm_f16x2.hip HEPI_QKV) - and by the op-level tests.
__global__ __launch_bounds__(256) void qk_norm_append_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ scale_mul, float* __restrict__ q_out,
                                                                  uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache, int R, int l, int H, int Lp, int pos0,
                                                                  PendingSplitK pend, int fmt) {
    const int NP = fmt == 3 ? 2 : 1;
    const int tid = threadIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int t = blockIdx.x * 32 + (tid >> 3), cg = tid & 7;
    const bool live = t < l;
    const int C = H * 64;
    const size_t ps = (size_t)Lp * 64;
    const bool l2 = scale_mul != nullptr;            // attn_l2_norm=False: raw q (x 2^-5, the softmax scale) and raw k (basic_var.py:71-72)
    const float sm = l2 ? expf(fminf(scale_mul[h], 4.605170249938965f)) : 0.03125f;
    float q[8], k[8], v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = k[e] = v[e] = 0.f;
    auto add8 = [](float* d, const float* p) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] += a[e]; d[4 + e] += b[e]; }
    };
    if (live) {
        const size_t o = ((size_t)r * l + t) * 3 * C + h * 64 + 8 * cg;
        if (pend.ws) {             // qkv[row][col] = sum_s ws[s][row][col] + bias[col], slabs added in slice order, four slabs (24 16-byte loads) in flight at a time
            const size_t slab = (size_t)R * l * 3 * C;
            int s = 0;
            for (; s + 3 < pend.split; s += 4) {
                f32x4 tt[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const float* p = pend.ws + (size_t)(s + u) * slab + o + (size_t)w * C;
                        tt[u][2 * w] = *reinterpret_cast<const f32x4*>(p); tt[u][2 * w + 1] = *reinterpret_cast<const f32x4*>(p + 4);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        q[e] += tt[u][0][e]; q[4 + e] += tt[u][1][e]; k[e] += tt[u][2][e]; k[4 + e] += tt[u][3][e]; v[e] += tt[u][4][e]; v[4 + e] += tt[u][5][e];
                    }
            }
            for (; s < pend.split; ++s) { const float* p = pend.ws + (size_t)s * slab + o; add8(q, p); add8(k, p + C); add8(v, p + 2 * C); }
            add8(q, pend.bias + h * 64 + 8 * cg); add8(k, pend.bias + C + h * 64 + 8 * cg); add8(v, pend.bias + 2 * C + h * 64 + 8 * cg);
        } else {
            add8(q, qkv + o); add8(k, qkv + o + C); add8(v, qkv + o + 2 * C);
        }
    }
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sq += q[e] * q[e]; sk += k[e] * k[e]; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { sq += __shfl_xor(sq, o, 64); sk += __shfl_xor(sk, o, 64); }
    if (!live) return;
    const float qn = l2 ? fmaxf(sqrtf(sq), 1e-12f) : 1.0f, kn = l2 ? fmaxf(sqrtf(sk), 1e-12f) : 1.0f;
    f32x4 q0, q1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { q0[e] = l2 ? (q[e] / qn) * sm : q[e] * sm; q1[e] = l2 ? (q[4 + e] / qn) * sm : q[4 + e] * sm; }
    float* pq = q_out + (((size_t)r * H + h) * l + t) * 64 + 8 * cg;
    *reinterpret_cast<f32x4*>(pq) = q0; *reinterpret_cast<f32x4*>(pq + 4) = q1;
    const size_t row = ((size_t)r * H + h) * NP * ps + (size_t)(pos0 + t) * 64 + 8 * cg;
    float kn8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) kn8[e] = l2 ? k[e] / kn : k[e];
    uint2 h0, l0, h1, l1;
    split4h_pk(kn8, h0, l0); split4h_pk(kn8 + 4, h1, l1);
    *reinterpret_cast<u32x4*>(k_cache + row) = u32x4{h0.x, h0.y, h1.x, h1.y};
    if (fmt == 3) *reinterpret_cast<u32x4*>(k_cache + row + ps) = u32x4{l0.x, l0.y, l1.x, l1.y};
    split4h_pk(v, h0, l0); split4h_pk(v + 4, h1, l1);
    *reinterpret_cast<u32x4*>(v_cache + row) = u32x4{h0.x, h0.y, h1.x, h1.y};
    if (fmt == 3) *reinterpret_cast<u32x4*>(v_cache + row + ps) = u32x4{l0.x, l0.y, l1.x, l1.y};
}

// Formats 2 / 3 / 4 of the cache (attention_bf16x3.hip: three bf16 planes; attention_f16x2.hip: two fp16 planes, or ONE fp16 plane = the
// fp16 KV cache of BASELINE config P4): K planes [R][H][NP][Lp][64] and V^T planes [R][H][NP][64][Lp] with bits 2 and 3 of the key
// position swapped inside every block of 16 keys.  Same arithmetic as above; format 2 holds the fp32 values exactly, format 3 to
// 2^-22, format 4 holds fp16(k), fp16(v) rounded to nearest even like torch's .half().
// One workgroup per (32-position block of the cache, head, row): the waves normalise their tokens and write q and the K
// planes (128-byte rows), V goes through LDS and leaves as 16-byte runs of 8 cache positions per channel row (transposed
// 2-byte stores cost 2x the whole kernel: 1.15 ms against 0.44 ms per d16 stage-9 call).
__global__ __launch_bounds__(256) void qk_norm_append_planes_kernel(const float* __restrict__ qkv, const float* __restrict__ scale_mul,
                                                                    float* __restrict__ q_out, uint16_t* __restrict__ k_cache,
                                                                    uint16_t* __restrict__ v_cache, int R, int l, int H, int Lp, int pos0, PendingSplitK pend,
                                                                    int fmt, int v_only) {
    __shared__ float vs[32 * 65];
    const int NP = fmt == 2 ? 3 : (fmt == 3 ? 2 : 1);
    const int tid = threadIdx.x;
    const int h = blockIdx.y, r = blockIdx.z;
    const int P0 = (pos0 / 32 + blockIdx.x) * 32;
    const int pb = max(P0, pos0), pe = min(P0 + 32, pos0 + l);
    const int C = H * 64;
    const size_t head = ((size_t)r * H + h) * NP * (size_t)Lp * 64, ps = (size_t)Lp * 64;
    const bool l2 = scale_mul != nullptr;            // attn_l2_norm=False: raw q (x 2^-5, the softmax scale) and raw k (basic_var.py:71-72)
    const float sm = l2 ? expf(fminf(scale_mul[h], 4.605170249938965f)) : 0.03125f;
    // thread = (position P0 + tid / 8, channel group tid % 8: channels 8cg .. 8cg+7): the whole 32-position block is in flight at once with 16-byte
    // loads (six per thread and slab), the q / k norms are an in-thread sum of 8 squares + three shuffle steps inside the 8-lane group
    {
        const int pos = P0 + (tid >> 3), cg = tid & 7;
        const bool live = pos >= pb && pos < pe;
        float q[8], k[8], v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = k[e] = v[e] = 0.f;
        auto add8 = [](float* d, const float* p) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[e] += a[e]; d[4 + e] += b[e]; }
        };
        if (live) {
            const size_t o = ((size_t)r * l + (pos - pos0)) * 3 * C + h * 64 + 8 * cg;
            if (pend.ws) {             // qkv[row][col] = sum_s ws[s][row][col] + bias[col], slabs added in slice order
                // four slabs (24 16-byte loads) in flight at a time, added in slice order: a load-then-add loop pays one memory round trip per slab
                const size_t slab = (size_t)R * l * 3 * C;
                int s = 0;
                for (; s + 3 < pend.split; s += 4) {
                    f32x4 t[4][6];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int w = 0; w < 3; ++w) {
                            const float* p = pend.ws + (size_t)(s + u) * slab + o + (size_t)w * C;
                            t[u][2 * w] = *reinterpret_cast<const f32x4*>(p); t[u][2 * w + 1] = *reinterpret_cast<const f32x4*>(p + 4);
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            q[e] += t[u][0][e]; q[4 + e] += t[u][1][e]; k[e] += t[u][2][e]; k[4 + e] += t[u][3][e]; v[e] += t[u][4][e]; v[4 + e] += t[u][5][e];
                        }
                }
                for (; s < pend.split; ++s) { const float* p = pend.ws + (size_t)s * slab + o; add8(q, p); add8(k, p + C); add8(v, p + 2 * C); }
                add8(q, pend.bias + h * 64 + 8 * cg); add8(k, pend.bias + C + h * 64 + 8 * cg); add8(v, pend.bias + 2 * C + h * 64 + 8 * cg);
            } else {
                if (!v_only) { add8(q, qkv + o); add8(k, qkv + o + C); }
                add8(v, qkv + o + 2 * C);
            }
        }
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sq += q[e] * q[e]; sk += k[e] * k[e]; }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { sq += __shfl_xor(sq, o, 64); sk += __shfl_xor(sk, o, 64); }
        if (live && v_only) {           // q and k left the QKV GEMM's epilogue finished (gemm_f16x2.hip HEPI_QKV): only V^T is written here
#pragma unroll
            for (int e = 0; e < 8; ++e) vs[(pos - P0) * 65 + 8 * cg + e] = v[e];
        } else if (live) {
            const float qn = l2 ? fmaxf(sqrtf(sq), 1e-12f) : 1.0f, kn = l2 ? fmaxf(sqrtf(sk), 1e-12f) : 1.0f;
            f32x4 q0, q1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { q0[e] = l2 ? (q[e] / qn) * sm : q[e] * sm; q1[e] = l2 ? (q[4 + e] / qn) * sm : q[4 + e] * sm; }
            float* pq = q_out + (((size_t)r * H + h) * l + (pos - pos0)) * 64 + 8 * cg;
            *reinterpret_cast<f32x4*>(pq) = q0; *reinterpret_cast<f32x4*>(pq + 4) = q1;
            uint16_t* pk = k_cache + head + (size_t)pos * 64 + 8 * cg;
            float kn8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) kn8[e] = l2 ? k[e] / kn : k[e];
            if (fmt == 2) {
                u32x4 a, b, cc;
                split8_packed(kn8, a, b, cc);
                *reinterpret_cast<u32x4*>(pk) = a; *reinterpret_cast<u32x4*>(pk + ps) = b; *reinterpret_cast<u32x4*>(pk + 2 * ps) = cc;
            } else {
                uint16_t hh[8], ll[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) split2h(kn8[e], hh[e], ll[e]);
                u32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (uint32_t)hh[2 * e] | ((uint32_t)hh[2 * e + 1] << 16); b[e] = (uint32_t)ll[2 * e] | ((uint32_t)ll[2 * e + 1] << 16); }
                *reinterpret_cast<u32x4*>(pk) = a;
                if (fmt == 3) *reinterpret_cast<u32x4*>(pk + ps) = b;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) vs[(pos - P0) * 65 + 8 * cg + e] = v[e];
        }
    }
    __syncthreads();
    // thread -> (channel row d, run c of 8 cache positions): positions P0 + 8c .. +7 hold keys P0 + 16 (c >> 1) + 4 (c & 1) + {0..3, 8..11}
    {
        const int d = tid >> 2, c = tid & 3;
        const int kb = P0 + 16 * (c >> 1) + 4 * (c & 1);
        float v[8];
        int inside = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = kb + (e & 3) + 8 * (e >> 2);
            const bool in = key >= pb && key < pe;
            inside += in;
            v[e] = in ? vs[(key - P0) * 65 + d] : 0.f;
        }
        uint16_t* pv = v_cache + head + (size_t)d * Lp + P0 + 8 * c;
        if (fmt != 2) {             // fp16 planes: h = fp16(v) (+ l = fp16(v - h) for format 3)
            uint16_t hh[8], ll[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split2h(v[e], hh[e], ll[e]);
            if (inside == 8) {
                u32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (uint32_t)hh[2 * e] | ((uint32_t)hh[2 * e + 1] << 16); b[e] = (uint32_t)ll[2 * e] | ((uint32_t)ll[2 * e + 1] << 16); }
                *reinterpret_cast<u32x4*>(pv) = a;
                if (fmt == 3) *reinterpret_cast<u32x4*>(pv + ps) = b;
            } else if (inside) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = kb + (e & 3) + 8 * (e >> 2);
                    if (key >= pb && key < pe) { pv[e] = hh[e]; if (fmt == 3) pv[ps + e] = ll[e]; }
                }
            }
        } else if (inside == 8) {
            u32x4 a, b, cc;
            split8_packed(v, a, b, cc);
            *reinterpret_cast<u32x4*>(pv) = a; *reinterpret_cast<u32x4*>(pv + ps) = b; *reinterpret_cast<u32x4*>(pv + 2 * ps) = cc;
        } else if (inside) {        // run shared with an earlier or later append: only this call's keys may be written
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = kb + (e & 3) + 8 * (e >> 2);
                if (key >= pb && key < pe) {
                    uint16_t v0, v1, v2;
                    split3(v[e], v0, v1, v2);
                    pv[e] = v0; pv[ps + e] = v1; pv[2 * ps + e] = v2;
                }
            }
        }
    }
}

int qk_norm_append(const float* qkv, const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int kv_f16, int R, int l, int H,
                   int Lmax, int pos0, const PendingSplitK* pend, int v_only, hipStream_t stream) {
    PendingSplitK pd{nullptr, nullptr, nullptr, 0, 1, 0};
    if (pend && pend->ws) { SDVAR_CHECK_ARG(pend->bias && pend->split >= 1, "qk_norm_append: bad pending split-K descriptor"); pd = *pend; }
    SDVAR_CHECK_ARG(R > 0 && l > 0 && H > 0 && pos0 >= 0 && pos0 + l <= Lmax, "qk_norm_append: cache overflow pos0=%d l=%d Lmax=%d", pos0, l, Lmax);
    const long long items = (long long)R * l * H;
    SDVAR_CHECK_ARG(kv_f16 >= 0 && kv_f16 <= 4, "qk_norm_append: cache format %d (0 = fp32, 1 = fp16, 2 = bf16x3 planes, 3 = f16x2 planes, 4 = one fp16 plane)", kv_f16);
    if (kv_f16 >= 2) {
        SDVAR_CHECK_ARG(Lmax % 64 == 0, "qk_norm_append: the planes KV formats need Lmax %% 64 == 0 (got %d)", Lmax);
        SDVAR_CHECK_ARG(!v_only, "qk_norm_append: there is no v-only pass any more (an unsplit QKV launch finishes q, k and v in its epilogue)");
        if (kv_f16 == 2) hipLaunchKernelGGL(qk_norm_append_planes_kernel, dim3((unsigned)((pos0 + l + 31) / 32 - pos0 / 32), H, R), dim3(256), 0, stream, qkv, scale_mul, q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, R, l, H, Lmax, pos0, pd, kv_f16, 0);
        else hipLaunchKernelGGL(qk_norm_append_rows_kernel, dim3((unsigned)((l + 31) / 32), H, R), dim3(256), 0, stream, qkv, scale_mul, q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, R, l, H, Lmax, pos0, pd, kv_f16);
    } else if (v_only) { set_error("qk_norm_append: there is no v-only pass"); return SDVAR_ERR_ARG;
    } else if (kv_f16) hipLaunchKernelGGL(qk_norm_append_kernel<__half>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, qkv, scale_mul, q_out, (__half*)k_cache, (__half*)v_cache, R, l, H, Lmax, pos0, pd);
    else hipLaunchKernelGGL(qk_norm_append_kernel<float>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, qkv, scale_mul, q_out, (float*)k_cache, (float*)v_cache, R, l, H, Lmax, pos0, pd);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ f16x2 guard (debug)
// The f16x2 operand format saturates finite activations at +-65504 and loses relative precision below ~1e-3 (gemm_f16x2.hip header).  With the guard
// switched on (sdvar_debug_set_f16x2_guard) every producer of GEMM operand planes in stage_forward is followed by this pass over the HIGH plane it
// wrote: cnt[0] += elements seen, cnt[1] += |h| == 65504 (saturated, or exactly the largest fp16 number), cnt[2] += NaN / Inf, cnt[3] += 0 < |h| < 2^-10
// (the low plane of such a value is subnormal or zero: relative error above 2^-22).  Off by default: one extra read of the plane per producer.
__global__ __launch_bounds__(256) void planes_guard_kernel(const uint16_t* __restrict__ h, size_t n8, unsigned long long* cnt) {
    unsigned int sat = 0, bad = 0, tiny = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(h + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t a = ((w[e >> 1] >> (16 * (e & 1))) & 0x7FFFu);
            sat += a == 0x7BFFu; bad += a >= 0x7C00u; tiny += (a != 0u && a < 0x1400u);
        }
    }
    sat = (unsigned int)wave_sum((float)sat); bad = (unsigned int)wave_sum((float)bad); tiny = (unsigned int)wave_sum((float)tiny);     // < 2^24 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0) {
        if (sat) atomicAdd(cnt + 1, (unsigned long long)sat);
        if (bad) atomicAdd(cnt + 2, (unsigned long long)bad);
        if (tiny) atomicAdd(cnt + 3, (unsigned long long)tiny);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(cnt, (unsigned long long)n8 * 8ull);
}

int planes_guard(const uint16_t* h_plane, size_t n, unsigned long long* cnt, hipStream_t stream) {
    SDVAR_CHECK_ARG(h_plane && cnt && n % 8 == 0 && ((uintptr_t)h_plane % 16) == 0, "planes_guard: plane must be 16-byte aligned with a multiple of 8 elements");
    const size_t n8 = n / 8, blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(planes_guard_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, h_plane, n8, cnt);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.0f + expf(-v)); }
}

int silu_rows(const float* x, float* y, int n, hipStream_t stream) {
    hipLaunchKernelGGL(silu_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, y, n);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// out[r][c] = vec[c] + src[r][c]  (shared adaLN: ada_gss of a block + the shared SiLU-Linear of cond, basic_var.py:153-154)
__global__ void add_row_vector_kernel(const float* __restrict__ src, const float* __restrict__ vec, float* __restrict__ out, int rows, int cols) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols) out[i] = vec[i % cols] + src[i];
}

int add_row_vector(const float* src, const float* vec, float* out, int rows, int cols, hipStream_t stream) {
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(add_row_vector_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, vec, out, rows, cols);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// cond (2B, C) = class_emb[label_b] for rows < B, class_emb[num_classes] for rows >= B - or, with cond_in, the caller's rows (VAR.autoregressive_infer_cfg_sd_helper1
// is handed `sos`, var.py:344-345);  x0 (2B,1,C) = cond + pos_start + lvl_pos[0]
__global__ void prologue_kernel(const long long* __restrict__ labels, const float* __restrict__ cond_in, const float* __restrict__ class_emb, const float* __restrict__ pos_start,
                                const float* __restrict__ lvl_pos, float* __restrict__ cond, float* __restrict__ x0, int B, int C, int num_classes) {
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float e;
    if (cond_in) e = cond_in[(size_t)r * C + c];
    else {
        long long lab = (r < B) ? labels[r] : (long long)num_classes;
        if (lab < 0 || lab > num_classes) lab = num_classes;
        e = class_emb[(size_t)lab * C + c];
    }
    cond[(size_t)r * C + c] = e;
    x0[(size_t)r * C + c] = (e + pos_start[c]) + lvl_pos[c];
}

int prologue(const long long* labels, const float* cond_in, const float* class_emb, const float* pos_start, const float* lvl_pos, float* cond, float* x0,
             int B, int C, int num_classes, hipStream_t stream) {
    hipLaunchKernelGGL(prologue_kernel, dim3((C + 255) / 256, 2 * B), dim3(256), 0, stream, labels, cond_in, class_emb, pos_start, lvl_pos, cond, x0, B, C, num_classes);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// adaLN parameters of a call from the per-class table (api.hip: model_begin): row r of block i = tab[label(r)][i * 6C ..], head row = tab[label(r)][depth * 6C ..]
__global__ __launch_bounds__(256) void ada_gather_kernel(const long long* __restrict__ labels, const float* __restrict__ tab, size_t row_floats, int depth, int C,
                                                         float* __restrict__ ada, size_t blk_stride, float* __restrict__ ada_head, int B, int num_classes) {
    const int r = blockIdx.x, i = blockIdx.y;
    long long lab = (r < B) ? labels[r] : (long long)num_classes;
    if (lab < 0 || lab > num_classes) lab = num_classes;
    const int n4 = (i < depth ? 6 * C : 2 * C) / 4;
    const float4* src = reinterpret_cast<const float4*>(tab + (size_t)lab * row_floats + (size_t)i * 6 * C);
    float4* dst = reinterpret_cast<float4*>(i < depth ? ada + (size_t)i * blk_stride + (size_t)r * 6 * C : ada_head + (size_t)r * 2 * C);
    for (int c = threadIdx.x; c < n4; c += 256) dst[c] = src[c];
}

int ada_gather(const long long* labels, const float* tab, size_t row_floats, int depth, int C, float* ada, size_t blk_stride, float* ada_head, int B, int num_classes,
               hipStream_t stream) {
    hipLaunchKernelGGL(ada_gather_kernel, dim3(2 * B, depth + 1), dim3(256), 0, stream, labels, tab, row_floats, depth, C, ada, blk_stride, ada_head, B, num_classes);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// lvl_pos[t][c] = lvl_embed[stage(t)][c] + pos_1LC[t][c]
__global__ void build_lvl_pos_kernel(const float* __restrict__ lvl_embed, const float* __restrict__ pos, const int* __restrict__ stage_of_tok,
                                     float* __restrict__ out, int L, int C) {
    const int t = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[(size_t)t * C + c] = lvl_embed[(size_t)stage_of_tok[t] * C + c] + pos[(size_t)t * C + c];
}

int build_lvl_pos(const float* lvl_embed, const float* pos, const int* stage_of_tok, float* out, int L, int C, hipStream_t stream) {
    hipLaunchKernelGGL(build_lvl_pos_kernel, dim3((C + 255) / 256, L), dim3(256), 0, stream, lvl_embed, pos, stage_of_tok, out, L, C);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// x[b][t][:] = x[B+b][t][:] = nxt[b][t][0:32] . Ww[:, 0:32]^T + bw + lvl_pos[t0 + t]       (Cvae = 32)
// one workgroup per (token, image): the 32 inputs are broadcast from LDS, each thread owns output channels.
__global__ __launch_bounds__(256) void embed_next_kernel(const float* __restrict__ nxt, const float* __restrict__ Ww, const float* __restrict__ bw,
                                                         const float* __restrict__ lvl_pos, float* __restrict__ x, int B, int l, int C, int t0,
                                                         int ltot, int tok_off) {
    __shared__ float in[32];
    const int t = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x < 32) in[threadIdx.x] = nxt[((size_t)b * l + t) * 32 + threadIdx.x];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const f32x4* w = reinterpret_cast<const f32x4*>(Ww + (size_t)c * 32);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 wv = w[k];
            acc = fmaf(in[4 * k + 0], wv[0], acc); acc = fmaf(in[4 * k + 1], wv[1], acc);
            acc = fmaf(in[4 * k + 2], wv[2], acc); acc = fmaf(in[4 * k + 3], wv[3], acc);
        }
        const float v = (acc + bw[c]) + lvl_pos[(size_t)(t0 + t) * C + c];
        x[((size_t)b * ltot + tok_off + t) * C + c] = v;
        x[((size_t)(B + b) * ltot + tok_off + t) * C + c] = v;
    }
}

int embed_next(const float* nxt, const float* Ww, const float* bw, const float* lvl_pos, float* x, int B, int l, int C, int t0, int ltot, int tok_off,
               hipStream_t stream) {
    SDVAR_CHECK_ARG(B > 0 && l > 0 && tok_off >= 0 && tok_off + l <= ltot, "embed_next: bad placement");
    hipLaunchKernelGGL(embed_next_kernel, dim3(l, B), dim3(256), 0, stream, nxt, Ww, bw, lvl_pos, x, B, l, C, t0, ltot, tok_off);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar
