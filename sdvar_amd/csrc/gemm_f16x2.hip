// fp32-accurate "NT" GEMM on the gfx950 f16 matrix cores with TWO-plane split operands (f16x2):
//     out[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] ),   X ~ Xh + Xl,  W * 2^S ~ Wh + Wl
// Xh = fp16(X) (round to nearest), Xl = fp16(X - Xh): 11 + 11 significand bits, |X - Xh - Xl| <= 2^-22 |X| while Xl is a normal
// fp16 number (|X| >= 2^-3); below that Xl runs into fp16's subnormals, which v_mfma_f32_32x32x16_f16 keeps (measured:
// tools/micro/f16x2_probe.hip), and the error becomes an ABSOLUTE 2^-25.  Three of the four plane products are evaluated
// (Xh.Wh + Xh.Wl + Xl.Wh; every fp16 x fp16 product is exact in fp32; fp32 accumulate); the dropped Xl.Wl term is <= 2^-22
// relative.  Measured against fp64 on random operands at K = 1024 (the probe): max error / rms(out) 2.3e-6, against 3.1e-6 for the
// six-product bf16x3 scheme (gemm_bf16x3.hip) and 2.7e-6 for an fp32 FMA chain - the fp32 accumulation order dominates all three -
// at HALF the matrix-pipe work of bf16x3 (3 x 32 cycles per 16 k) and two thirds of its operand bytes.
// Range: weights are scaled per tensor by an exact power of two at bind time (2^S brings max|W| to (2^12, 2^13], so the low
// plane of a typical weight is a normal fp16 number; the epilogue multiplies by 2^-S); activations are taken as they are and
// SATURATE at +-65504 (the fp16 range the reference's own half-precision attention path assumes, basic_var.py:97,113).  Rows whose
// values are all below ~1e-3 lose relative precision (absolute error stays 2^-25): the bf16x3 mode has no such limit.
//
// Operands are PLANAR and K-blocked: planes[2][K/32][rows][32] fp16 (common.h kb_index); producers (ln_modulate, attention, the fc1
// GELU epilogue) write them directly.  Tiling, split-K, deferral and the hybrid tail split are those of gemm_bf16x3.hip with 2 planes /
// 3 products per k16-step; because a K-step now holds half the matrix work, the 256-row kernel keeps TWO K-steps of LDS-DMA in flight
// (3-stage ring) instead of one.  What differs from gemm_bf16x3.hip: the 32- and 64-row tiles are an LDS-DMA ring too
// (gemm_f16x2_small_kernel), the accumulators are transposed (SDVAR_MFMA3: 16-byte epilogue accesses), the unsplit QKV launch of a
// block finishes q and k in its epilogue (HEPI_QKV), and the cost model is fitted to HBM-cold weights with a deferral-aware split cost.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace sdvar {

unsigned long long* debug_get_gemm_stamps();      // gemm_bf16x3.hip

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
enum { HEPI_BIAS = 0, HEPI_BIAS_GELU_PLANES = 1, HEPI_GATED_RES = 2, HEPI_PARTIAL = 3, HEPI_QKV = 4 };

// Result-corrupting timing switches (SDVAR_GEMM_DBG: skip the in-loop DMA / barrier / fragment reads / reduce launch) exist only in builds made with
// -DSDVAR_TIMING_EXPERIMENTS (make EXTRA=-DSDVAR_TIMING_EXPERIMENTS, tools/micro/gemm_dbg_exp.sh): the product library carries neither the branches
// nor the environment variable.
#ifdef SDVAR_TIMING_EXPERIMENTS
#define SDVAR_DBG(a, bit) ((a).dbg & (bit))
#else
#define SDVAR_DBG(a, bit) 0
#endif

constexpr int HBK = 32;             // k per LDS stage
constexpr int HBN = 128;

// GELU(tanh) (basic_var.py:40): 0.5 x (1 + tanh(u)) = x sigmoid(2u) = x / (1 + 2^(-2u log2 e)) with v_exp_f32 and v_rcp_f32 (1 ulp each, ~2e-7
// relative on the result): libm's tanhf costs ~40 vector instructions per value and the epilogue of a 256 x 128 tile evaluates 64 of them per lane
// with nothing to hide them under (12 us per round of workgroups at M = 4096, fc1)
__device__ __forceinline__ float gelu_tanh_h(float x) {
    // -2 u log2 e = x (C0 + C1 x^2), C0 = -2 log2(e) sqrt(2 / pi), C1 = 0.044715 C0: 7 vector instructions (mul, fma, mul, exp2, add, rcp, mul)
    const float C0 = -2.3022081986f, C1 = -0.10294324f;
    const float w = x * __builtin_fmaf(x * x, C1, C0);
    const float e = __builtin_amdgcn_exp2f(w);                               // exp(-2u); +inf for very negative x: the quotient is then 0
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// HEPI_QKV (the unsplit QKV launch of a transformer block, N = 3 H 64): the q and k thirds leave the epilogue finished - bias, per-head L2 norm, q scale
// (basic_var.py:101-109), q as fp32 (R, H, l, 64), k and v as planes of the KV cache at positions pos0 + t (both row-major [position][64]: the attention kernel
// takes V^T fragments with ds_read_b64_tr_b16, so no transposed copy exists).  Nothing goes to `out`: the launch replaces qk_norm_append altogether
// (round 2 still ran its v-only pass behind this epilogue for a transposed V^T cache layout).
struct QkvEpi {
    const float* scale_mul;      // (H) or null: attn_l2_norm=False (q x 2^-5, raw k)
    float* q_out; uint16_t* k_cache; uint16_t* v_cache;      // v_cache: planes [R][H][NP][Lp][64], row-major like K (attention_f16x2.hip reads V^T through ds_read_b64_tr_b16)
    int l, H, Lp, pos0, fmt;     // tokens per row of the CFG batch, heads, cache rows, first position, cache format (3: two fp16 planes, 4: one)
};

struct GemmHArgs {
    const uint16_t* X; const uint16_t* W;        // K-blocked planes [2][K/32][M][32], [2][K/32][N][32]
    size_t xps, wps;                             // plane strides in elements
    const float* wsi;                            // device scalar 2^-S undoing the weight scale (null: 1)
    const float* bias; float* out; uint16_t* outp; size_t ops;
    const float* res; const float* gate;
    int M, N, K, ldo, ldres, rows_per_gate, gate_stride, split, k_per_split;
    int vec;                 // every pointer of the epilogue 16-byte aligned and every leading dimension a multiple of 4: 16-byte epilogue accesses
    int dbg;                 // timing experiments only (SDVAR_GEMM_DBG, results wrong): bit 0 no DMA inside the K loop, bit 1 no barrier, bit 2 no fragment reads, bit 3 no split-K reduce launch
    unsigned long long* stamps;   // diagnostic (sdvar_debug_set_gemm_stamps; small-M and 128 x 128 kernels): 8 x u64 per workgroup = s_memrealtime (100 MHz) at
                                  // entry / first K-step landed / loop end / exit, then s_memtime (core clock) at the same four points
    QkvEpi qk;               // HEPI_QKV only
    int tile_off, tile_cnt;  // 256-row kernel only: this launch covers tile ids [tile_off, tile_off + tile_cnt) (tile_cnt = 0: all); with
                             // HEPI_PARTIAL the slabs are compact [split][tile_cnt][256][128]
};

// acc += Xl.Wh + Xh.Wl + Xh.Wh (smallest terms first), TRANSPOSED: the W fragment is the MFMA's A operand and the X fragment its B operand, so the
// accumulator holds out^T: lane li = ROW m of the 32-row tile, register r = COLUMN (r & 3) + 8 (r >> 2) + 4 lh of the 32-column tile.  A lane then owns
// four consecutive columns per register group and every epilogue moves 16 bytes (fp32) or 8 bytes (a plane) per access, with one row index - one
// division for the gate row - per lane and tile.  With lane = column (the untransposed product) the 128 x 128 kernel spent 5.5 (bias) / 8 (GELU -> planes) /
// 12-14 us (gated residual) per tile in 4- and 2-byte accesses behind a 23 us K loop at K = 1024 (in-kernel stamps, tools/micro/gemm_stamps_h.py).
#define SDVAR_MFMA3(acc, xh, xl, wh, wl)                                               \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);                \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);                \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0)

// one output element of every epilogue: v = acc * 2^-S + bias, then the epilogue's own arithmetic (the edge / unaligned path of h_store_tile)
template <int EPI>
__device__ __forceinline__ void h_store(const GemmHArgs& a, float* outp, float accv, float wsi, float bv, int m, int n) {
    float v = accv * wsi + bv;
    if (EPI == HEPI_BIAS_GELU_PLANES) {
        uint16_t h, l;
        split2h(gelu_tanh_h(v), h, l);
        const size_t o = kb_index(m, n, a.M);           // the output is the next GEMM's K-blocked operand
        a.outp[o] = h; a.outp[a.ops + o] = l;
    } else {
        if (EPI == HEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n];
        outp[(size_t)m * a.ldo + n] = v;
    }
}

// HEPI_QKV, one lane's share of one (row m, head): `v` = the NT 32-column tiles' 16 values each (bias added), channel of v[j][4 g + e] = 32 j + 8 g + 4 lh + e
// (+ cbase for a wave that owns only the upper half of the head); sq = the head's full sum of squares of the row.  which: 0 q, 1 k.
template <int NT>
__device__ __forceinline__ void qk_store_row(const GemmHArgs& a, const f32x16* v, float sq, int which, int h, int m, int lh, int cbase) {
    if (m >= a.M) return;
    const QkvEpi& e = a.qk;
    const bool l2 = e.scale_mul != nullptr;
    const float nrm = l2 ? fmaxf(sqrtf(sq), 1e-12f) : 1.0f;
    const int r = m / e.l, t = m - r * e.l;
    if (which == 0) {
        const float sm = l2 ? expf(fminf(e.scale_mul[h], 4.605170249938965f)) : 0.03125f;     // log(100) as the reference's float32 clamp
        float* pq = e.q_out + (((size_t)r * e.H + h) * e.l + t) * 64 + cbase + 4 * lh;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x] = l2 ? (v[j][4 * g + x] / nrm) * sm : v[j][4 * g + x] * sm;
                *reinterpret_cast<f32x4*>(pq + 32 * j + 8 * g) = o;
            }
    } else {                   // k (which 1: normalised) or v (which 2: as it is): planes of the cache at position pos0 + t
        const int NP = e.fmt == 3 ? 2 : 1;
        const size_t ps = (size_t)e.Lp * 64;
        const bool nk = l2 && which == 1;
        uint16_t* pk = (which == 1 ? e.k_cache : e.v_cache) + ((size_t)r * e.H + h) * NP * ps + (size_t)(e.pos0 + t) * 64 + cbase + 4 * lh;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float kv[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) kv[x] = nk ? v[j][4 * g + x] / nrm : v[j][4 * g + x];
                uint2 wh, wl;
                split4h_pk(kv, wh, wl);
                *reinterpret_cast<uint2*>(pk + 32 * j + 8 * g) = wh;
                if (e.fmt == 3) *reinterpret_cast<uint2*>(pk + ps + 32 * j + 8 * g) = wl;
            }
    }
}

// acc * 2^-S + bias for one 32-column tile whose first column (of this lane half) is nb; returns the lane's sum of squares
__device__ __forceinline__ float qk_bias_sq(const GemmHArgs& a, const f32x16& acc, f32x16& out, float wsi, int nb) {
    float sq = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 4; ++x) { const float t = acc[4 * g + x] * wsi + bv[x]; out[4 * g + x] = t; sq += t * t; }
    }
    return sq;
}

// One 32 x 32 accumulator tile (transposed layout, SDVAR_MFMA3): this lane's row m, columns nb + 8 g + {0..3} for g = 0..3 (nb already holds the lane
// half's + 4 lh).  a.vec = every pointer 16-byte aligned and every leading dimension a multiple of 4 (checked by the host).
template <int EPI>
__device__ __forceinline__ void h_store_tile(const GemmHArgs& a, float* outp, const f32x16& acc, float wsi, int m, int nb) {
    if (m >= a.M) return;
    const size_t grow = (EPI == HEPI_GATED_RES) ? (size_t)(m / a.rows_per_gate) * a.gate_stride : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        if (n >= a.N) continue;
        if (!a.vec || n + 3 >= a.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < a.N) h_store<EPI>(a, outp, acc[4 * g + e], wsi, (EPI != HEPI_PARTIAL && a.bias) ? a.bias[n + e] : 0.f, m, n + e);
            continue;
        }
        f32x4 v;
        if (EPI != HEPI_PARTIAL && a.bias) {          // acc * 2^-S + bias: the product with a power of two is exact, so the fused form rounds like mul + add
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[4 * g + e], wsi, bv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e] * wsi;
        }
        if (EPI == HEPI_BIAS_GELU_PLANES) {
            float gv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = gelu_tanh_h(v[e]);
            uint2 wh, wl;
            split4h_pk(gv, wh, wl);
            const size_t o = kb_index(m, n, a.M);
            *reinterpret_cast<uint2*>(a.outp + o) = wh; *reinterpret_cast<uint2*>(a.outp + a.ops + o) = wl;
        } else {
            if (EPI == HEPI_GATED_RES) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + (size_t)m * a.ldres + n);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + grow + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e] * gv[e];
            }
            *reinterpret_cast<f32x4*>(outp + (size_t)m * a.ldo + n) = v;
        }
    }
}

// The same tile when the host has checked a.vec and the kernel that every one of the tile's 32 columns is inside N (true for all model shapes: N is a multiple
// of the column tile): no per-group range checks or element-wise fallback, ONE index computation per tile (inside a 32-column tile the K-blocked plane offset and
// the row-major offsets advance by 8 elements per register group), bias / gate / residual through three pointers.  The generic h_store_tile spent more
// instructions on its branches and 64-bit index arithmetic than on the epilogue's arithmetic (8030 instructions per wave in the 256 x 256 kernel's GELU epilogue).
template <int EPI>
__device__ __forceinline__ void h_store_tile_fast(const GemmHArgs& a, float* outp, const f32x16& acc, float wsi, int m, int nb) {
    if (m >= a.M) return;
    const float* pb = (EPI != HEPI_PARTIAL && a.bias) ? a.bias + nb : nullptr;
    if (EPI == HEPI_BIAS_GELU_PLANES) {
        uint16_t* p0 = a.outp + kb_index(m, nb, a.M);
        uint16_t* p1 = p0 + a.ops;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float gv[4];
            const f32x4 bv = pb ? *reinterpret_cast<const f32x4*>(pb + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = gelu_tanh_h(__builtin_fmaf(acc[4 * g + e], wsi, bv[e]));
            uint2 wh, wl;
            split4h_pk(gv, wh, wl);
            *reinterpret_cast<uint2*>(p0 + 8 * g) = wh; *reinterpret_cast<uint2*>(p1 + 8 * g) = wl;
        }
        return;
    }
    float* po = outp + (size_t)m * a.ldo + nb;
    const float* pr = (EPI == HEPI_GATED_RES) ? a.res + (size_t)m * a.ldres + nb : nullptr;
    const float* pg = (EPI == HEPI_GATED_RES) ? a.gate + (size_t)(m / a.rows_per_gate) * a.gate_stride + nb : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
        if (pb) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[4 * g + e], wsi, bv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e] * wsi;
        }
        if (EPI == HEPI_GATED_RES) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(pr + 8 * g), gt = *reinterpret_cast<const f32x4*>(pg + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e] * gt[e];
        }
        *reinterpret_cast<f32x4*>(po + 8 * g) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 128 tile, 8 waves (2 x 4, 64x32 outputs each), K-steps of 32 streamed global -> LDS by the LDS-DMA into a 3-stage ring,
// two K-steps in flight, ONE raw s_barrier per K-step and counted vmcnt waits.
//   stage (32 KB) = 4 sub-arrays [128 rows][64 B]: X planes h, l then W planes h, l; unpadded rows, the 16-byte chunk c of row r is
//   stored at chunk c ^ ((r >> 2) & 3) (applied on the DMA source address and on the ds_read address).
//   Every wave issues 4 DMA instructions per K-step (rows 16w .. 16w+15 of each sub-array).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int H2_STAGE = 4 * 128 * 32;          // fp16 elements per stage (32 KB)
#define SDVAR_LDS_RDH(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")

template <int EPI, int NS>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_v2_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int BM = 128;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + HBN - 1) / HBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * HBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;

    const int drow = 16 * wave + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);
    const int xrow = min(m0 + drow, a.M - 1), wrow = min(n0 + drow, a.N - 1);   // clamped: rows past the edge are never stored
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx = (uint32_t)(xrow * 32 + 8 * dchunk) * 2u, lw = (uint32_t)(wrow * 32 + 8 * dchunk) * 2u;
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32);
    // DMA instruction q (0..3) of K-step t (relative) -> stage t % 3: q = 2p is X plane p, q = 2p + 1 is W plane p
    auto issue_one = [&](int t, int q) {
        uint16_t* st = hsm + (t % (NS == 4 ? 3 : NS)) * H2_STAGE + swave * 512;      // + sub-array * 4096 elements
        const int p = q >> 1;
        if (q & 1) SDVAR_DMA16(lw, bw + ((size_t)t * a.N * 32 + p * a.wps) * 2, SDVAR_LDS_ADDR(st + (2 + p) * 4096));
        else SDVAR_DMA16(lx, bx + ((size_t)t * a.M * 32 + p * a.xps) * 2, SDVAR_LDS_ADDR(st + p * 4096));
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_one(t, q);
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // fragment read offsets (elements) inside a sub-array: row * 32 + 8 * ((2s + lh) ^ ((li >> 2) & 3))
    const int sw = (li >> 2) & 3;
    const int offa0 = (wm * 64 + li) * 32, offb = (wn * 32 + li) * 32;
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);

    unsigned long long vsr[4] = {0, 0, 0, 0}, vsc[4] = {0, 0, 0, 0};
    if (NS == 4) {
        // Software-pipelined variant (template value 4 = "3 stages, pipelined reads"): ONE barrier per K-step, placed where a wave holds every fragment
        // of tile t in registers; behind it tile t+1's first-half fragments are read into the registers the first half of tile t just freed (its
        // second half after the second MFMA group), and the DMA of tile t+3 goes into the stage of tile t.
        f16x8 fa[2][2][2], fb[2][2];
        auto read_half = [&](int t, int sdx) {
            const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(hsm + (t % 3) * H2_STAGE);
            const uint32_t aa = sb + 2 * (offa0 + (sdx ? ch1 : ch0)), ab = sb + 2 * (offb + (sdx ? ch1 : ch0));
            SDVAR_LDS_RDH(fa[sdx][1][0], aa, 8192);  SDVAR_LDS_RDH(fb[sdx][0], ab, 16384); SDVAR_LDS_RDH(fa[sdx][0][0], aa, 0);
            SDVAR_LDS_RDH(fb[sdx][1], ab, 24576);    SDVAR_LDS_RDH(fa[sdx][1][1], aa, 10240); SDVAR_LDS_RDH(fa[sdx][0][1], aa, 2048);
        };
        auto issue3 = [&](int t, int q) {
            uint16_t* st = hsm + (t % 3) * H2_STAGE + swave * 512;
            const int p = q >> 1;
            if (q & 1) SDVAR_DMA16(lw, bw + ((size_t)t * a.N * 32 + p * a.wps) * 2, SDVAR_LDS_ADDR(st + (2 + p) * 4096));
            else SDVAR_DMA16(lx, bx + ((size_t)t * a.M * 32 + p * a.xps) * 2, SDVAR_LDS_ADDR(st + p * 4096));
        };
        if (a.stamps) { vsr[0] = __builtin_amdgcn_s_memrealtime(); vsc[0] = __builtin_amdgcn_s_memtime(); }
        for (int tt = 0; tt < 3 && tt < nk; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) issue3(tt, q);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (a.stamps) { vsr[1] = __builtin_amdgcn_s_memrealtime(); vsc[1] = __builtin_amdgcn_s_memtime(); }
        read_half(0, 0);
        read_half(0, 1);
        for (int t = 0; t < nk; ++t) {
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) { SDVAR_MFMA3(acc[i], fa[0][0][i], fa[0][1][i], fb[0][0], fb[0][1]); }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool more = t + 1 < nk, pf = t + 3 < nk;
            if (more) read_half(t + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                SDVAR_MFMA3(acc[i], fa[1][0][i], fa[1][1][i], fb[1][0], fb[1][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (pf) { issue3(t + 3, 2 * i); issue3(t + 3, 2 * i + 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) read_half(t + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (NS == 6) {
        // Ping-pong variant (template value 6 = "4 stages, the two waves of a SIMD alternate"): the schedule of gemm_f16x2_v4_kernel on this tile.  A K-step is one
        // L segment (12 fragment reads + the 4 DMA instructions of K-step t + 3) and one M segment (12 MFMAs), an s_barrier between segments; waves 4-7 (wm = 1)
        // run one segment behind, so in every slot one wave of a SIMD feeds the matrix pipe while the other reads LDS / issues DMA:
        //     slot:       2t     2t+1    2t+2
        //     waves 0-3:  L(t)   M(t)    L(t+1)
        //     waves 4-7:  M(t-1) L(t)    M(t)
        // Stage (t + 3) % 4 = (t - 1) % 4 is refilled in L(t): its last reader (the late half's L(t-1), slot 2t - 1) is past its lgkmcnt(0) + barrier.
        // K-step t + 1 is read from slot 2t + 2 on: every wave waits for its own share (counted vmcnt: K-steps t + 2, t + 3 stay in flight) at the END of slot
        // 2t + 1 - in M(t) for the early half, in L(t) for the late half - i.e. 4 - 5 slots (~2000 cycles) after it was requested.
        f16x8 fa[2][2][2], fb[2][2];
        const int late = __builtin_amdgcn_readfirstlane(wm);
        auto issue4 = [&](int t) {
            uint16_t* st = hsm + (t & 3) * H2_STAGE + swave * 512;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                SDVAR_DMA16(lx, bx + ((size_t)t * a.M * 32 + p * a.xps) * 2, SDVAR_LDS_ADDR(st + p * 4096));
                SDVAR_DMA16(lw, bw + ((size_t)t * a.N * 32 + p * a.wps) * 2, SDVAR_LDS_ADDR(st + (2 + p) * 4096));
            }
        };
        auto wait_next = [&](int t) {          // K-step t + 1 landed: the K-steps requested after it (at most two) may stay in flight
            const int after = nk - t - 2;
            if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#define SDVAR_H2_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
        for (int tt = 0; tt < 3 && tt < nk; ++tt) issue4(tt);
        wait_next(-1);
        SDVAR_H2_SLOT();
        if (late) SDVAR_H2_SLOT();
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(hsm + (t & 3) * H2_STAGE);
            const uint32_t aa0 = sb + 2 * (offa0 + ch0), aa1 = sb + 2 * (offa0 + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
            SDVAR_LDS_RDH(fa[0][1][0], aa0, 8192);  SDVAR_LDS_RDH(fb[0][0], ab0, 16384); SDVAR_LDS_RDH(fa[0][0][0], aa0, 0);
            SDVAR_LDS_RDH(fb[0][1], ab0, 24576);    SDVAR_LDS_RDH(fa[0][1][1], aa0, 10240); SDVAR_LDS_RDH(fa[0][0][1], aa0, 2048);
            SDVAR_LDS_RDH(fa[1][1][0], aa1, 8192);  SDVAR_LDS_RDH(fb[1][0], ab1, 16384); SDVAR_LDS_RDH(fa[1][0][0], aa1, 0);
            SDVAR_LDS_RDH(fb[1][1], ab1, 24576);    SDVAR_LDS_RDH(fa[1][1][1], aa1, 10240); SDVAR_LDS_RDH(fa[1][0][1], aa1, 2048);
            if (t + 3 < nk) issue4(t + 3);
            if (late) wait_next(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_H2_SLOT();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 2; ++i) { SDVAR_MFMA3(acc[i], fa[s2][0][i], fa[s2][1][i], fb[s2][0], fb[s2][1]); }
            if (!late) wait_next(t);
            SDVAR_H2_SLOT();
        }
        if (!late) SDVAR_H2_SLOT();
#undef SDVAR_H2_SLOT
    } else {
    for (int tt = 0; tt < NS - 1 && tt < nk; ++tt) issue(tt);
    for (int t = 0; t < nk; ++t) {
        // tile t has landed when at most the min(NS - 2, tiles behind it) newest K-steps (4 instructions each) are still in flight
        const int behind = min(NS - 2, nk - 1 - t);
        if (behind >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!SDVAR_DBG(a, 2)) __builtin_amdgcn_s_barrier();
        const bool pf = t + NS - 1 < nk && !SDVAR_DBG(a, 1);
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(hsm + (t % (NS == 4 ? 3 : NS)) * H2_STAGE);
        const uint32_t aa0 = sb + 2 * (offa0 + ch0), aa1 = sb + 2 * (offa0 + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        // fa[s][plane][row tile], fb[s][plane]: X plane p at +8192 p bytes, second 32-row tile at +2048; W plane p at +16384 + 8192 p
        f16x8 fa[2][2][2], fb[2][2];
        if (SDVAR_DBG(a, 4)) {
#pragma unroll
            for (int z = 0; z < 8; ++z) { fa[z >> 2][(z >> 1) & 1][z & 1] = __builtin_bit_cast(f16x8, f32x4{1.f, 2.f, 3.f, (float)z}); }
#pragma unroll
            for (int z = 0; z < 4; ++z) fb[z >> 1][z & 1] = __builtin_bit_cast(f16x8, f32x4{1.f, 2.f, 3.f, (float)z});
        } else {
        SDVAR_LDS_RDH(fa[0][1][0], aa0, 8192);  SDVAR_LDS_RDH(fb[0][0], ab0, 16384); SDVAR_LDS_RDH(fa[0][0][0], aa0, 0);
        SDVAR_LDS_RDH(fb[0][1], ab0, 24576);    SDVAR_LDS_RDH(fa[0][1][1], aa0, 10240); SDVAR_LDS_RDH(fa[0][0][1], aa0, 2048);
        SDVAR_LDS_RDH(fa[1][1][0], aa1, 8192);  SDVAR_LDS_RDH(fb[1][0], ab1, 16384); SDVAR_LDS_RDH(fa[1][0][0], aa1, 0);
        SDVAR_LDS_RDH(fb[1][1], ab1, 24576);    SDVAR_LDS_RDH(fa[1][1][1], aa1, 10240); SDVAR_LDS_RDH(fa[1][0][1], aa1, 2048);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                SDVAR_MFMA3(acc[i], fa[s][0][i], fa[s][1][i], fb[s][0], fb[s][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (pf) issue_one(t + NS - 1, 2 * s + i);     // the 4 DMA instructions of the K-step NS-1 ahead, one behind each MFMA group
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    }

    if (a.stamps) { vsr[2] = __builtin_amdgcn_s_memrealtime(); vsc[2] = __builtin_amdgcn_s_memtime(); }
    const float wsi = a.wsi ? *a.wsi : 1.0f;
    if (EPI == HEPI_QKV) {
        // the 128 columns of a tile lie inside one of the q / k / v thirds (H 64 is a multiple of 128 or the launcher does not pick this epilogue); a head
        // is the 64 columns of the wave pair (wn, wn ^ 1): the row norms are exchanged through LDS (free: the ring's last reads are behind the barrier)
        const int Cq = a.qk.H * 64, which = n0 / Cq, nh = n0 - which * Cq + (wn >> 1) * 64, h = nh >> 6;
        if (which == 2) {          // v: bias added, straight into the cache planes (32 of the head's 64 channels per wave; no norm, nothing to exchange)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 vv;
                (void)qk_bias_sq(a, acc[i], vv, wsi, n0 + wn * 32 + 4 * lh);
                qk_store_row<1>(a, &vv, 0.f, 2, h, m0 + wm * 64 + i * 32 + li, lh, (wn & 1) * 32);
            }
        } else {
            float* ex = reinterpret_cast<float*>(hsm);           // [4 wn][128 rows]
            f32x16 v[2]; float sq[2];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sq[i] = qk_bias_sq(a, acc[i], v[i], wsi, n0 + wn * 32 + 4 * lh);
                sq[i] += __shfl_xor(sq[i], 32, 64);
                if (lh == 0) ex[wn * 128 + wm * 64 + i * 32 + li] = sq[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float tot = (wn & 1) ? ex[(wn ^ 1) * 128 + wm * 64 + i * 32 + li] + sq[i] : sq[i] + ex[(wn ^ 1) * 128 + wm * 64 + i * 32 + li];   // lower half first in both waves
                qk_store_row<1>(a, &v[i], tot, which, h, m0 + wm * 64 + i * 32 + li, lh, (wn & 1) * 32);
            }
        }
    } else {
        float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
        if (a.vec && n0 + HBN <= a.N) {
#pragma unroll
            for (int i = 0; i < 2; ++i) h_store_tile_fast<EPI>(a, outp, acc[i], wsi, m0 + wm * 64 + i * 32 + li, n0 + wn * 32 + 4 * lh);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) h_store_tile<EPI>(a, outp, acc[i], wsi, m0 + wm * 64 + i * 32 + li, n0 + wn * 32 + 4 * lh);
        }
    }
    if (a.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vsr[3] = __builtin_amdgcn_s_memrealtime(); vsc[3] = __builtin_amdgcn_s_memtime();
        unsigned long long* o = a.stamps + 8 * (size_t)blockIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = vsr[i]; o[4 + i] = vsc[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small-tile kernel (32- and 64-row tiles: stages 0-5, the first verify chunks): BM x 128 tile, 4 waves side by side (BM x 32 outputs each), the LDS-DMA
// ring of the 128 x 128 kernel.  These launches stream each weight byte once and do little matrix work: their time is launch + first-operand latency +
// a short K loop, so what counts is bytes in flight and resident waves, not the MFMA schedule.
//   stage = X planes h, l [BM][32] then W planes h, l [128][32] fp16 (20 / 24 KB), chunk swizzle as the 128 x 128 kernel;
//   NS = 3 stages (60 / 72 KB: two workgroups = 8 waves per CU), two K-steps in flight per workgroup; one s_barrier per K-step, counted vmcnt waits;
//   per K-step the 2 (BM / 16 + 8) DMA instructions (16 rows each) are dealt round-robin to the 4 waves: IPS = 5 / 6 each.
// Measured with HBM-cold weights (tools/micro/gemm_cold_mid.py, slab launch alone): against the register-staged kernel it replaces (one K-step in
// flight, two barriers per K-step) M = 64: qkv 8.1 -> 7.0, fc1 8.8 -> 7.4, fc2 10.0 -> 8.0 us; M = 16: 5.8 -> 5.7, 6.4 -> 6.0, 8.7 -> 7.8 us; a 6- / 7-stage
// ring with one workgroup per CU (everything in flight) was no faster at M = 16 and slower from M = 64 (one wave per SIMD).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int NS, int EPI>
__global__ __launch_bounds__(256) void gemm_f16x2_small_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int TM = BM / 32, XB = BM / 16, IPS = (2 * (XB + 8)) / 4, STAGE = 2 * (BM + 128) * 32;
    static_assert((2 * (XB + 8)) % 4 == 0 && (NS - 2) * IPS <= 63, "bad ring");
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + HBN - 1) / HBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int tm = lid % tiles_m, tn = lid / tiles_m;         // the row tiles of one column panel are neighbours: they share the weight slice
    const int m0 = tm * BM, n0 = tn * HBN;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);

    // this wave's DMA instructions j = 0 .. IPS-1 are q = wave + 4 j of the K-step's list: q < 2 XB: X plane q / XB, 16-row block q % XB; else W plane, block
    const char* gb[IPS]; uint32_t vo[IPS], lo[IPS]; size_t sb[IPS];
#pragma unroll
    for (int j = 0; j < IPS; ++j) {
        const int q = wave + 4 * j;
        const bool isx = q < 2 * XB;
        const int qq = isx ? q : q - 2 * XB;
        const int p = isx ? qq / XB : qq / 8, b = isx ? qq % XB : qq % 8;
        const int r16 = b * 16 + (lane >> 2);
        const int grow = isx ? min(m0 + r16, a.M - 1) : min(n0 + r16, a.N - 1);      // clamped: rows past the edge are never stored
        const int ch = (lane & 3) ^ ((r16 >> 2) & 3);
        vo[j] = (uint32_t)(grow * 32 + 8 * ch) * 2u;
        gb[j] = isx ? reinterpret_cast<const char*>(a.X + p * a.xps + (size_t)kt0 * a.M * 32) : reinterpret_cast<const char*>(a.W + p * a.wps + (size_t)kt0 * a.N * 32);
        sb[j] = (size_t)(isx ? a.M : a.N) * 64;
        lo[j] = (uint32_t)((isx ? p * BM * 32 : 2 * BM * 32 + p * 4096) + b * 512) * 2u;
    }
    auto issue = [&](int t) {
        const uint32_t st = SDVAR_LDS_ADDR(hsm + (t % NS) * STAGE);
#pragma unroll
        for (int j = 0; j < IPS; ++j) SDVAR_DMA16(vo[j], gb[j] + (size_t)t * sb[j], st + lo[j]);
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int sw = (li >> 2) & 3;
    const int offa = li * 32, offb = 2 * BM * 32 + (wave * 32 + li) * 32;

    unsigned long long sr[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0};
    if (a.stamps) { sr[0] = __builtin_amdgcn_s_memrealtime(); sc[0] = __builtin_amdgcn_s_memtime(); }
    for (int tt = 0; tt < NS - 1 && tt < nk; ++tt) issue(tt);
    for (int t = 0; t < nk; ++t) {
        // K-step t has landed when at most the min(NS - 2, steps behind it) newest K-steps (IPS instructions each) are still in flight
        switch (min(NS - 2, nk - 1 - t)) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<IPS>(); break;
            case 2: wait_vmcnt<2 * IPS>(); break;
            case 3: wait_vmcnt<3 * IPS>(); break;
            case 4: wait_vmcnt<(NS > 5 ? 4 : 0) * IPS>(); break;
            default: wait_vmcnt<(NS > 6 ? 5 : 0) * IPS>(); break;
        }
        __builtin_amdgcn_s_barrier();                 // every wave's part of K-step t is in LDS, and every wave is done reading the stage of K-step t - 1
        if (a.stamps && t == 0) { sr[1] = __builtin_amdgcn_s_memrealtime(); sc[1] = __builtin_amdgcn_s_memtime(); }
        if (t + NS - 1 < nk) issue(t + NS - 1);       // ... which K-step t + NS - 1 now overwrites
        const uint16_t* st = hsm + (t % NS) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ch = 8 * ((2 * s2 + lh) ^ sw);
            f16x8 fa[2][TM], fb[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fb[p] = *reinterpret_cast<const f16x8*>(st + offb + p * 4096 + ch);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const f16x8*>(st + offa + p * BM * 32 + i * 1024 + ch);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) { SDVAR_MFMA3(acc[i], fa[0][i], fa[1][i], fb[0], fb[1]); }
        }
    }

    if (a.stamps) { sr[2] = __builtin_amdgcn_s_memrealtime(); sc[2] = __builtin_amdgcn_s_memtime(); }
    const float wsi = a.wsi ? *a.wsi : 1.0f;
    if (EPI == HEPI_QKV) {             // as the 128 x 128 kernel: a head is the 64 columns of the wave pair (wave, wave ^ 1)
        const int Cq = a.qk.H * 64, which = n0 / Cq, nh = n0 - which * Cq + (wave >> 1) * 64, h = nh >> 6;
        if (which == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x16 vv;
                (void)qk_bias_sq(a, acc[i], vv, wsi, n0 + wave * 32 + 4 * lh);
                qk_store_row<1>(a, &vv, 0.f, 2, h, m0 + i * 32 + li, lh, (wave & 1) * 32);
            }
        } else {
            float* ex = reinterpret_cast<float*>(hsm);           // [4 waves][BM rows]
            f32x16 v[TM]; float sq[TM];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                sq[i] = qk_bias_sq(a, acc[i], v[i], wsi, n0 + wave * 32 + 4 * lh);
                sq[i] += __shfl_xor(sq[i], 32, 64);
                if (lh == 0) ex[wave * BM + i * 32 + li] = sq[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float other = ex[(wave ^ 1) * BM + i * 32 + li];
                qk_store_row<1>(a, &v[i], (wave & 1) ? other + sq[i] : sq[i] + other, which, h, m0 + i * 32 + li, lh, (wave & 1) * 32);
            }
        }
    } else {
        float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
        if (a.vec && n0 + HBN <= a.N) {
#pragma unroll
            for (int i = 0; i < TM; ++i) h_store_tile_fast<EPI>(a, outp, acc[i], wsi, m0 + i * 32 + li, n0 + wave * 32 + 4 * lh);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) h_store_tile<EPI>(a, outp, acc[i], wsi, m0 + i * 32 + li, n0 + wave * 32 + 4 * lh);
        }
    }
    if (a.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sr[3] = __builtin_amdgcn_s_memrealtime(); sc[3] = __builtin_amdgcn_s_memtime();
        unsigned long long* o = a.stamps + 8 * (size_t)blockIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = sr[i]; o[4 + i] = sc[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Mid-M launches (stages 2-5 and the first verify chunks: M = 144 .. 576): the 64 x 128 tile of the kernel above with the K loop SPLIT OVER THE TWO WAVE GROUPS
// of a 512-thread workgroup, ping-pong ("small_pp").  In gemm_f16x2_small_kernel a K-step is wait -> barrier -> 12 fragment reads -> 12 MFMAs (384 matrix-pipe
// cycles) for the ONE wave a SIMD holds, and the K-step costs ~1050 cycles (profiles/r03_v_midm_ab.log: 16 - 21 us for 32 K-steps whatever the tile): the LDS and
// barrier latency is exposed, not the matrix work.  Here group g (waves 4g .. 4g + 3, one per SIMD) owns the K-steps t = 2k + g, with its own 3-stage LDS-DMA ring
// (2 x 3 x 24 KB = 144 KB, one workgroup per CU) and its own accumulators; the groups run one slot apart -
//     slot:      2k      2k+1     2k+2
//     group 0:   L(k)    M(k)     L(k+1)        L = 12 ds_read_b128 + the 6 DMA instructions of the group's K-step k + 2,  M = 12 MFMAs + the counted vmcnt wait
//     group 1:   M(k-1)  L(k)     M(k)          for the group's K-step k + 1; an s_barrier closes every slot
// so a SIMD's matrix pipe is fed by one group while the other reads LDS: two K-steps per two slots instead of one K-step per ~2.7 slot lengths.  After the loop the
// groups swap one row tile each through LDS (32 KB) - group 0 finishes rows 0-31 of the tile, group 1 rows 32-63 (sum = group 0 + group 1 in both: deterministic) -
// and run the epilogue of gemm_f16x2_small_kernel on it.
template <int BM> constexpr int spp_stage() { return 2 * (BM + 128) * 32; }       // fp16 elements: 24 KB (BM = 64) / 20 KB (BM = 32)

// BM = 64: as described above.  BM = 32 (one 32-row tile per wave): the groups cannot swap row tiles - group 1 hands its partial sums to group 0, whose four waves
// run the epilogue alone (group 1 only keeps the barriers).
template <int BM, int EPI>
__global__ __launch_bounds__(512) void gemm_f16x2_small_pp_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int TM = BM / 32, XB = BM / 16, IPS = (2 * (XB + 8)) / 4, STAGE = spp_stage<BM>();
    static_assert(BM == 32 || BM == 64, "tile");
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + HBN - 1) / HBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int tm = lid % tiles_m, tn = lid / tiles_m;
    const int m0 = tm * BM, n0 = tn * HBN;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6), g = wave8 >> 2, wave = wave8 & 3;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);
    const int n_own = (nk - g + 1) >> 1, n_it = (nk + 1) >> 1;          // K-steps of this group; iterations of the (common) loop = group 0's count

    // the group's DMA instructions j = 0 .. IPS-1 of one K-step are q = wave + 4 j of the list: q < 2 XB: X plane q / XB, 16-row block q % XB; else W plane, block
    const char* gb[IPS]; uint32_t vo[IPS], lo[IPS]; size_t sb[IPS];
#pragma unroll
    for (int j = 0; j < IPS; ++j) {
        const int q = wave + 4 * j;
        const bool isx = q < 2 * XB;
        const int qq = isx ? q : q - 2 * XB;
        const int p = isx ? qq / XB : qq / 8, b = isx ? qq % XB : qq % 8;
        const int r16 = b * 16 + (lane >> 2);
        const int grow = isx ? min(m0 + r16, a.M - 1) : min(n0 + r16, a.N - 1);      // clamped: rows past the edge are never stored
        const int ch = (lane & 3) ^ ((r16 >> 2) & 3);
        vo[j] = (uint32_t)(grow * 32 + 8 * ch) * 2u;
        gb[j] = isx ? reinterpret_cast<const char*>(a.X + p * a.xps + (size_t)(kt0 + g) * a.M * 32) : reinterpret_cast<const char*>(a.W + p * a.wps + (size_t)(kt0 + g) * a.N * 32);
        sb[j] = (size_t)(isx ? a.M : a.N) * 128;                                     // two K-steps on
        lo[j] = (uint32_t)((isx ? p * BM * 32 : 2 * BM * 32 + p * 4096) + b * 512) * 2u;
    }
    uint16_t* ring = hsm + g * 3 * STAGE;
    auto issue = [&](int k) {
        const uint32_t st = SDVAR_LDS_ADDR(ring + (k % 3) * STAGE);
#pragma unroll
        for (int j = 0; j < IPS; ++j) SDVAR_DMA16(vo[j], gb[j] + (size_t)k * sb[j], st + lo[j]);
    };
    auto wait_next = [&](int k) {          // the group's K-step k + 1 landed: its K-step k + 2 (IPS instructions), if requested, may stay in flight
        if (k + 2 < n_own) { if (IPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int sw = (li >> 2) & 3;
    const int offa = li * 32, offb = 2 * BM * 32 + (wave * 32 + li) * 32;
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);
#define SDVAR_SPP_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    if (n_own > 0) issue(0);
    if (n_own > 1) issue(1);
    wait_next(-1);
    SDVAR_SPP_SLOT();
    if (g) SDVAR_SPP_SLOT();
    f16x8 fa[2][2][TM], fb[2][2];          // [k16 half][plane][row tile], [k16 half][plane]
#pragma unroll 1
    for (int k = 0; k < n_it; ++k) {
        const bool act = k < n_own;
        if (act) {
            const uint32_t sbase = (uint32_t)(uintptr_t)(lds_ptr_t)(ring + (k % 3) * STAGE);
            const uint32_t aa0 = sbase + 2 * (offa + ch0), aa1 = sbase + 2 * (offa + ch1), ab0 = sbase + 2 * (offb + ch0), ab1 = sbase + 2 * (offb + ch1);
            // X plane p at + 64 BM p bytes, second 32-row tile at + 2048; W plane p at + 8192 p
            if constexpr (BM == 64) {
                SDVAR_LDS_RDH(fa[0][1][0], aa0, 4096); SDVAR_LDS_RDH(fb[0][0], ab0, 0); SDVAR_LDS_RDH(fa[0][0][0], aa0, 0);
                SDVAR_LDS_RDH(fb[0][1], ab0, 8192);    SDVAR_LDS_RDH(fa[0][1][1], aa0, 6144); SDVAR_LDS_RDH(fa[0][0][1], aa0, 2048);
                SDVAR_LDS_RDH(fa[1][1][0], aa1, 4096); SDVAR_LDS_RDH(fb[1][0], ab1, 0); SDVAR_LDS_RDH(fa[1][0][0], aa1, 0);
                SDVAR_LDS_RDH(fb[1][1], ab1, 8192);    SDVAR_LDS_RDH(fa[1][1][1], aa1, 6144); SDVAR_LDS_RDH(fa[1][0][1], aa1, 2048);
            } else {
                SDVAR_LDS_RDH(fa[0][1][0], aa0, 2048); SDVAR_LDS_RDH(fb[0][0], ab0, 0); SDVAR_LDS_RDH(fa[0][0][0], aa0, 0); SDVAR_LDS_RDH(fb[0][1], ab0, 8192);
                SDVAR_LDS_RDH(fa[1][1][0], aa1, 2048); SDVAR_LDS_RDH(fb[1][0], ab1, 0); SDVAR_LDS_RDH(fa[1][0][0], aa1, 0); SDVAR_LDS_RDH(fb[1][1], ab1, 8192);
            }
            if (k + 2 < n_own) issue(k + 2);          // stage (k + 2) % 3 = (k - 1) % 3: its readers are past L(k - 1)'s lgkmcnt(0) + barrier
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_SPP_SLOT();
        if (act) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i) { SDVAR_MFMA3(acc[i], fa[s2][0][i], fa[s2][1][i], fb[s2][0], fb[s2][1]); }
            wait_next(k);
        }
        SDVAR_SPP_SLOT();
    }
    if (!g) SDVAR_SPP_SLOT();
#undef SDVAR_SPP_SLOT

    float* ex = reinterpret_cast<float*>(hsm);           // [8 waves][16 registers][64 lanes]
    f32x16 mine;
    bool active;
    int row;
    __syncthreads();
    if constexpr (TM == 2) {
        // the groups swap one row tile: group 0 hands over its part of rows 32-63, group 1 its part of rows 0-31; both add in the order group 0 + group 1
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[wave8 * 1024 + r * 64 + lane] = acc[1][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[wave8 * 1024 + r * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        const float* oth = ex + ((1 - g) * 4 + wave) * 1024 + lane;
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = acc[0][r] + oth[r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = oth[r * 64] + acc[1][r];
        }
        active = true; row = m0 + g * 32 + li;
    } else {
        if (g == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[wave * 1024 + r * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        if (g == 0) {
            const float* oth = ex + wave * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = acc[0][r] + oth[r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = 0.f;
        }
        active = (g == 0); row = m0 + li;
    }
    const float wsi = a.wsi ? *a.wsi : 1.0f;
    if (EPI == HEPI_QKV) {             // a head is the 64 columns of the wave pair (wave, wave ^ 1) of one group
        const int Cq = a.qk.H * 64, which = n0 / Cq, nh = n0 - which * Cq + (wave >> 1) * 64, h = nh >> 6;
        f32x16 vv;
        float sq = 0.f;
        if (active) sq = qk_bias_sq(a, mine, vv, wsi, n0 + wave * 32 + 4 * lh);
        if (which == 2) {
            if (active) qk_store_row<1>(a, &vv, 0.f, 2, h, row, lh, (wave & 1) * 32);
        } else {
            float* exq = ex + 8 * 1024;                      // [8 waves][32 rows], behind the swap area
            sq += __shfl_xor(sq, 32, 64);
            if (lh == 0) exq[wave8 * 32 + li] = sq;
            __syncthreads();
            const float other = exq[(wave8 ^ 1) * 32 + li];
            if (active) qk_store_row<1>(a, &vv, (wave & 1) ? other + sq : sq + other, which, h, row, lh, (wave & 1) * 32);
        }
    } else if (active) {
        float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
        if (a.vec && n0 + HBN <= a.N) h_store_tile_fast<EPI>(a, outp, mine, wsi, row, n0 + wave * 32 + 4 * lh);
        else h_store_tile<EPI>(a, outp, mine, wsi, row, n0 + wave * 32 + 4 * lh);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 128 workgroup tile, 8 waves (4 x 2) of 64 x 64 outputs, LDS-DMA into a 3-stage ring (48 KB per stage: X planes [2][256][32]
// then W planes [2][128][32]), two K-steps in flight: a K-step holds 24 MFMAs per wave (768 cycles), less than the DMA latency
// under load, so one K-step of lookahead (what the bf16x3 kernel uses) would expose it.  (Software-pipelining the fragment reads across the
// K-step boundary - next tile's reads under this tile's MFMAs - was measured and changed nothing: the LDS reads are not what is exposed.)
constexpr int H3_STAGE = 2 * (256 + 128) * 32;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_v3_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int BM = 256;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + HBN - 1) / HBN, ntile = tiles_m * tiles_n;
    const int tcnt = a.tile_cnt > 0 ? a.tile_cnt : ntile;
    const int ks = blockIdx.x / tcnt;
    const int lid = a.tile_off + xcd_remap(blockIdx.x - ks * tcnt, tcnt);
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * HBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

    // DMA: wave w fills X rows [32w, 32w+32) (two 16-row groups) and W rows [16w, 16w+16) of both planes: 6 instructions per K-step
    const int r16 = lane >> 2;
    const int xr0 = 32 * wave + r16, xr1 = xr0 + 16, wr = 16 * wave + r16;
    const int cx0 = (lane & 3) ^ ((xr0 >> 2) & 3), cx1 = (lane & 3) ^ ((xr1 >> 2) & 3), cw = (lane & 3) ^ ((wr >> 2) & 3);
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx0 = (uint32_t)(min(m0 + xr0, a.M - 1) * 32 + 8 * cx0) * 2u, lx1 = (uint32_t)(min(m0 + xr1, a.M - 1) * 32 + 8 * cx1) * 2u;
    const uint32_t lw0 = (uint32_t)(min(n0 + wr, a.N - 1) * 32 + 8 * cw) * 2u;
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32);
    // DMA instruction q (0..5) of K-step t -> stage t % 3: plane p = q / 3; q % 3 = 0 / 1: X row groups, 2: W row group
    auto issue_one = [&](int t, int q) {
        uint16_t* st = hsm + (t % 3) * H3_STAGE;
        const int p = q / 3, kind = q % 3;
        const char* ux = bx + ((size_t)t * a.M * 32 + p * a.xps) * 2;          // wave-uniform
        const char* uw = bw + ((size_t)t * a.N * 32 + p * a.wps) * 2;
        if (kind == 0) SDVAR_DMA16(lx0, ux, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024));
        else if (kind == 1) SDVAR_DMA16(lx1, ux, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024 + 512));
        else SDVAR_DMA16(lw0, uw, SDVAR_LDS_ADDR(st + 2 * 8192 + p * 4096 + swave * 512));
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) issue_one(t, q);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int offa = (wm * 64 + li) * 32, offb = 2 * 8192 + (wn * 64 + li) * 32;     // element offsets inside a stage
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);

    issue(0);
    if (nk > 1) issue(1);
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 2 < nk;     // the 6 DMA instructions of K-step t+2 are spread between the MFMA groups below
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(hsm + (t % 3) * H3_STAGE);
        const uint32_t aa0 = sb + 2 * (offa + ch0), aa1 = sb + 2 * (offa + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        // fa[s][plane][row tile], fb[s][plane][col tile]: X plane p at +16384 p bytes, W plane p at +8192 p bytes (behind the X planes), second 32-row tile at +2048
        f16x8 fa[2][2][2], fb[2][2][2];
        SDVAR_LDS_RDH(fa[0][1][0], aa0, 16384); SDVAR_LDS_RDH(fb[0][0][0], ab0, 0);     SDVAR_LDS_RDH(fa[0][0][0], aa0, 0);     SDVAR_LDS_RDH(fb[0][1][0], ab0, 8192);
        SDVAR_LDS_RDH(fb[0][0][1], ab0, 2048);  SDVAR_LDS_RDH(fb[0][1][1], ab0, 10240); SDVAR_LDS_RDH(fa[0][1][1], aa0, 18432); SDVAR_LDS_RDH(fa[0][0][1], aa0, 2048);
        SDVAR_LDS_RDH(fa[1][1][0], aa1, 16384); SDVAR_LDS_RDH(fb[1][0][0], ab1, 0);     SDVAR_LDS_RDH(fa[1][0][0], aa1, 0);     SDVAR_LDS_RDH(fb[1][1][0], ab1, 8192);
        SDVAR_LDS_RDH(fb[1][0][1], ab1, 2048);  SDVAR_LDS_RDH(fb[1][1][1], ab1, 10240); SDVAR_LDS_RDH(fa[1][1][1], aa1, 18432); SDVAR_LDS_RDH(fa[1][0][1], aa1, 2048);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    SDVAR_MFMA3(acc[i][j], fa[s][0][i], fa[s][1][i], fb[s][0][j], fb[s][1][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pf) {
                        const int grp = 4 * s + 2 * i + j;           // 0..7: six DMA instructions over the first six groups
                        if (grp < 6) issue_one(t + 2, grp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }

    const float wsi = a.wsi ? *a.wsi : 1.0f;
    if (EPI == HEPI_QKV) {             // a wave's 64 columns are one head of q, k or v
        const int Cq = a.qk.H * 64, which = n0 / Cq, nh = n0 - which * Cq + wn * 64, h = nh >> 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + li;
            f32x16 v[2];
            float sq = qk_bias_sq(a, acc[i][0], v[0], wsi, n0 + wn * 64 + 4 * lh);
            sq += qk_bias_sq(a, acc[i][1], v[1], wsi, n0 + wn * 64 + 32 + 4 * lh);
            if (which != 2) sq += __shfl_xor(sq, 32, 64);
            qk_store_row<2>(a, v, sq, which, h, m, lh, 0);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
        if (EPI == HEPI_PARTIAL && a.tile_cnt > 0) {         // tail tiles of a hybrid launch: compact slab [256][128] of this (slice, tile), every row stored
            float* slab = a.out + ((size_t)ks * a.tile_cnt + (lid - a.tile_off)) * (256 * 128);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * wsi;
                    *reinterpret_cast<f32x4*>(slab + (wm * 64 + i * 32 + li) * 128 + wn * 64 + j * 32 + 8 * g + 4 * lh) = v;
                }
            continue;
        }
        if (a.vec && n0 + HBN <= a.N) {
#pragma unroll
            for (int i = 0; i < 2; ++i) h_store_tile_fast<EPI>(a, outp, acc[i][j], wsi, m0 + wm * 64 + i * 32 + li, n0 + wn * 64 + j * 32 + 4 * lh);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) h_store_tile<EPI>(a, outp, acc[i][j], wsi, m0 + wm * 64 + i * 32 + li, n0 + wn * 64 + j * 32 + 4 * lh);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 workgroup tile, 8 waves (2 x 4) of 128 (m) x 64 (n) outputs: the large-M kernel for the wide GEMMs (QKV, fc1, head).
// Why: the 256 x 128 kernel pulls 48 KB per K-step through the CU for 24 MFMAs per wave (66 GB/s per CU at the rate it runs: the L2 -> LDS path of
// every CU is busy), reads 0.67 fragments per MFMA, and both waves of a SIMD run the SAME program in phase - they stall on their fragment reads together
// and then compete for the matrix pipe together.  Here a K-step moves 64 KB for 48 MFMAs per wave (a third less operand traffic per flop, 0.5 fragment reads
// per MFMA) and the two waves of every SIMD alternate roles (MI355X_MICROARCH.md "Two waves per SIMD"; the structure of the guide's 256^2 template):
//     slot:        0    1    2    3    4    5    6    7     (one K-step = 8 slots, an s_barrier between slots)
//     waves 0-3:   L0   M0   L1   M1   L2   M2   L3   M3    (wm = 0: the upper 128 rows of the tile)
//     waves 4-7:   M3'  L0   M0   L1   M1   L2   M2   L3    (wm = 1: one slot behind)
//   Mp = 12 MFMAs on one quadrant (64 m x 32 n) of the wave's tile over the K-step's 32 k (3 plane products x 2 k16 x 2 row tiles);
//   L0 = fragment reads X rows 0-63 (8 x b128) + W columns 0-31 (4), L1 = W columns 32-63 (4), L2 = X rows 64-127 (8) + the LDS-DMA of the W slab of
//   K-step t+2 (4 instructions), L3 = the DMA of the X slab of K-step t+2 (4) + the counted vmcnt that retires K-step t+1.
//   While one wave of a SIMD owns the matrix pipe the other one reads LDS / issues DMA, so neither waits for the other's kind of work.
// LDS: 2 stages x 64 KB (X planes [2][256][32] then W planes [2][256][32], rows of 64 B, chunk swizzle of the other kernels).  A slab of stage t % 2 is
// refilled (K-step t+2) as soon as its last reader is past its lgkmcnt(0) + barrier: W after slot 3, X after slot 5 - a whole K-step (64 KB per CU) is always in flight.
constexpr int H4_STAGE = 2 * (256 + 256) * 32;          // fp16 elements per stage (64 KB)

// VAR (DMA placement, A/B): 0 = 4 + 4 instructions in L2 / L3 (a whole K-step of lookahead); 1 = 2 per L segment (X of K-step t+1 in L0 / L1, W of t+2 in L2 / L3);
// 2 = as 1 but issued inside the M segments, one instruction behind the 4th and the 8th MFMA
template <int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_v4_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int BM = 256, BN = 256;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int G = 4, per_group = tiles_m * G;                   // an XCD's run of tile ids covers G column tiles x a run of row tiles
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);

    // DMA: a slab (X or W of one K-step) = 2 planes x 16 row blocks of 16 rows (1 KB each) = 32 instructions, 4 per wave: wave w fills plane w / 4, rows 64 (w % 4) .. + 63
    const int dpl = wave >> 2, drow0 = 64 * (wave & 3) + (lane >> 2);
    const int dch = (lane & 3) ^ ((lane >> 4) & 3);             // chunk swizzle (row >> 2) & 3 of rows drow0 + 16 e: the same for every e
    uint32_t vx[4], vw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vx[e] = (uint32_t)(min(m0 + drow0 + 16 * e, a.M - 1) * 32 + 8 * dch) * 2u;        // clamped: rows past the edge are never stored
        vw[e] = (uint32_t)(min(n0 + drow0 + 16 * e, a.N - 1) * 32 + 8 * dch) * 2u;
    }
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32 + (size_t)dpl * a.xps);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32 + (size_t)dpl * a.wps);
    const uint32_t lds0 = SDVAR_LDS_ADDR(hsm);
    const uint32_t ldx = lds0 + (uint32_t)(dpl * 16384 + (64 * (wave & 3)) * 64), ldw = ldx + 32768u;      // byte address of this wave's first row block, stage 0
    // instructions e0 .. e1-1 of this wave's share of the X / W slab of K-step t
    auto issue_x = [&](int t, int e0, int e1) {
        const char* src = bx + (size_t)t * a.M * 64;
        const uint32_t dst = ldx + (uint32_t)(t & 1) * 65536u;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e >= e0 && e < e1) SDVAR_DMA16(vx[e], src, dst + 1024u * e);
    };
    auto issue_w = [&](int t, int e0, int e1) {
        const char* src = bw + (size_t)t * a.N * 64;
        const uint32_t dst = ldw + (uint32_t)(t & 1) * 65536u;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e >= e0 && e < e1) SDVAR_DMA16(vw[e], src, dst + 1024u * e);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses (bytes, stage 0): row * 64 + 16 * ((2 s + lh) ^ ((li >> 2) & 3)); plane p at + 16384 p, 32-row tile i at + 2048 i
    const int sw = (li >> 2) & 3;
    const uint32_t ax0 = lds0 + (uint32_t)((wm * 128 + li) * 64 + 16 * ((0 + lh) ^ sw)), ax1 = lds0 + (uint32_t)((wm * 128 + li) * 64 + 16 * ((2 + lh) ^ sw));
    const uint32_t aw0 = lds0 + 32768u + (uint32_t)((wn * 64 + li) * 64 + 16 * ((0 + lh) ^ sw)), aw1 = lds0 + 32768u + (uint32_t)((wn * 64 + li) * 64 + 16 * ((2 + lh) ^ sw));

#ifdef SDVAR_V4_STAMPS
    unsigned long long wsr[4] = {0, 0, 0, 0}, wsc[4] = {0, 0, 0, 0};        // s_memrealtime (100 MHz) / s_memtime (core clock) at entry, loop start, loop end, exit
    wsr[0] = __builtin_amdgcn_s_memrealtime(); wsc[0] = __builtin_amdgcn_s_memtime();
#endif
    issue_w(0, 0, 4); issue_x(0, 0, 4);
    if (VAR == 0) {
        if (nk > 1) { issue_w(1, 0, 4); issue_x(1, 0, 4); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (nk > 1) { issue_w(1, 0, 4); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#ifdef SDVAR_V4_STAMPS
    wsr[1] = __builtin_amdgcn_s_memrealtime(); wsc[1] = __builtin_amdgcn_s_memtime();
#endif
    if (wm == 1) __builtin_amdgcn_s_barrier();                  // the lower half of the tile runs one slot behind (wave-uniform branch)

    f16x8 fx[2][2][2], fw[2][2][2];                             // fx[row tile of the half][k16 step][plane], fw[column tile][k16 step][plane]
#ifdef SDVAR_V4_STAMPS          // diagnostic build only: slot boundaries of K-step 8 of waves 0 and 4 of workgroup 0 -> a.stamps[16 (wave / 4) + k]
    unsigned long long stm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SDVAR_H4_STAMP(k) do { if (t == 8) stm[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SDVAR_H4_STAMP(k) do { } while (0)
#endif
#define SDVAR_H4_SLOT(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); SDVAR_H4_STAMP(k); __builtin_amdgcn_sched_barrier(0); } while (0)
    // 12 MFMAs of one quadrant; DMA_A / DMA_B (VAR 2) are issued behind the 4th and the 8th
#define SDVAR_H4_MFMA(I0, J, DMA_A, DMA_B)                                                                                          \
    do {                                                                                                                            \
        SDVAR_MFMA3(acc[(I0)][J], fx[0][0][0], fx[0][0][1], fw[J][0][0], fw[J][0][1]);                                              \
        SDVAR_MFMA3(acc[(I0) + 1][J], fx[1][0][0], fx[1][0][1], fw[J][0][0], fw[J][0][1]);                                          \
        __builtin_amdgcn_sched_barrier(0); if (VAR == 2) { DMA_A; } __builtin_amdgcn_sched_barrier(0);                               \
        SDVAR_MFMA3(acc[(I0)][J], fx[0][1][0], fx[0][1][1], fw[J][1][0], fw[J][1][1]);                                              \
        __builtin_amdgcn_sched_barrier(0); if (VAR == 2) { DMA_B; } __builtin_amdgcn_sched_barrier(0);                               \
        SDVAR_MFMA3(acc[(I0) + 1][J], fx[1][1][0], fx[1][1][1], fw[J][1][0], fw[J][1][1]);                                          \
    } while (0)
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const uint32_t so = (uint32_t)(t & 1) * 65536u;
        const uint32_t x0 = ax0 + so, x1 = ax1 + so, w0 = aw0 + so, w1 = aw1 + so;
        const bool p1 = t + 1 < nk, p2 = t + 2 < nk;
        SDVAR_H4_STAMP(0);
        // ---- L0: X rows 0-63 of the wave's half, W columns 0-31
        SDVAR_LDS_RDH(fx[0][0][0], x0, 0);     SDVAR_LDS_RDH(fx[0][0][1], x0, 16384); SDVAR_LDS_RDH(fw[0][0][0], w0, 0);     SDVAR_LDS_RDH(fw[0][0][1], w0, 16384);
        SDVAR_LDS_RDH(fx[1][0][0], x0, 2048);  SDVAR_LDS_RDH(fx[1][0][1], x0, 18432); SDVAR_LDS_RDH(fx[0][1][0], x1, 0);     SDVAR_LDS_RDH(fx[0][1][1], x1, 16384);
        SDVAR_LDS_RDH(fw[0][1][0], w1, 0);     SDVAR_LDS_RDH(fw[0][1][1], w1, 16384); SDVAR_LDS_RDH(fx[1][1][0], x1, 2048);  SDVAR_LDS_RDH(fx[1][1][1], x1, 18432);
        if (VAR == 1 && p1) issue_x(t + 1, 0, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H4_SLOT(1);
        SDVAR_H4_MFMA(0, 0, if (p1) issue_x(t + 1, 0, 1), if (p1) issue_x(t + 1, 1, 2));                      // M0: rows 0-63 x columns 0-31
        SDVAR_H4_SLOT(2);
        // ---- L1: W columns 32-63
        SDVAR_LDS_RDH(fw[1][0][0], w0, 2048);  SDVAR_LDS_RDH(fw[1][0][1], w0, 18432); SDVAR_LDS_RDH(fw[1][1][0], w1, 2048);  SDVAR_LDS_RDH(fw[1][1][1], w1, 18432);
        if (VAR == 1 && p1) issue_x(t + 1, 2, 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H4_SLOT(3);
        SDVAR_H4_MFMA(0, 1, if (p1) issue_x(t + 1, 2, 3), if (p1) issue_x(t + 1, 3, 4));                      // M1: rows 0-63 x columns 32-63
        SDVAR_H4_SLOT(4);
        // ---- L2: X rows 64-127; every wave is past its last read of this stage's W slab: refill it with K-step t + 2
        SDVAR_LDS_RDH(fx[0][0][0], x0, 4096);  SDVAR_LDS_RDH(fx[0][0][1], x0, 20480); SDVAR_LDS_RDH(fx[1][0][0], x0, 6144);  SDVAR_LDS_RDH(fx[1][0][1], x0, 22528);
        SDVAR_LDS_RDH(fx[0][1][0], x1, 4096);  SDVAR_LDS_RDH(fx[0][1][1], x1, 20480); SDVAR_LDS_RDH(fx[1][1][0], x1, 6144);  SDVAR_LDS_RDH(fx[1][1][1], x1, 22528);
        if (VAR == 0 && p2) issue_w(t + 2, 0, 4);
        if (VAR == 1 && p2) issue_w(t + 2, 0, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H4_SLOT(5);
        SDVAR_H4_MFMA(2, 1, if (p2) issue_w(t + 2, 0, 1), if (p2) issue_w(t + 2, 1, 2));                      // M2: rows 64-127 x columns 32-63
        SDVAR_H4_SLOT(6);
        // ---- L3: the X slab is free too (the other half's L2 was one slot ago); K-step t + 1 must have landed before anybody starts it
        if (VAR == 0) {
            if (p2) { issue_x(t + 2, 0, 4); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (VAR == 1) {
            if (p2) { issue_w(t + 2, 2, 4); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {             // VAR 2: the last two W instructions of K-step t + 2 go out inside M3, behind this wait: in flight are W(t+2)[0..1] only
            if (p2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SDVAR_H4_SLOT(7);
        SDVAR_H4_MFMA(2, 0, if (p2) issue_w(t + 2, 2, 3), if (p2) issue_w(t + 2, 3, 4));                      // M3: rows 64-127 x columns 0-31
        SDVAR_H4_SLOT(8);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                  // matches the extra barrier of the late half
#ifdef SDVAR_V4_STAMPS
    wsr[2] = __builtin_amdgcn_s_memrealtime(); wsc[2] = __builtin_amdgcn_s_memtime();
    if (a.stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {
#pragma unroll
        for (int k = 0; k < 9; ++k) a.stamps[16 * (wave >> 2) + k] = stm[k];
    }
#endif
#undef SDVAR_H4_MFMA
#undef SDVAR_H4_SLOT
#undef SDVAR_H4_STAMP

    const float wsi = a.wsi ? *a.wsi : 1.0f;
    if (EPI == HEPI_QKV) {             // a wave's 64 columns are one head of q, k or v (the thirds are multiples of 64 columns: decided per wave)
        const int Cq = a.qk.H * 64, nw = n0 + wn * 64, which = nw / Cq, h = (nw - which * Cq) >> 6;
        if (nw >= a.N) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + li;
            f32x16 v[2];
            float sq = qk_bias_sq(a, acc[i][0], v[0], wsi, nw + 4 * lh);
            sq += qk_bias_sq(a, acc[i][1], v[1], wsi, nw + 32 + 4 * lh);
            if (which != 2) sq += __shfl_xor(sq, 32, 64);
            qk_store_row<2>(a, v, sq, which, h, m, lh, 0);
        }
        return;
    }
    float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
    if (a.vec && n0 + BN <= a.N) {            // every model shape: aligned operands, N a multiple of the column tile
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) h_store_tile_fast<EPI>(a, outp, acc[i][j], wsi, m0 + wm * 128 + i * 32 + li, n0 + wn * 64 + j * 32 + 4 * lh);
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) h_store_tile<EPI>(a, outp, acc[i][j], wsi, m0 + wm * 128 + i * 32 + li, n0 + wn * 64 + j * 32 + 4 * lh);
    }
#ifdef SDVAR_V4_STAMPS
    if (a.stamps && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wsr[3] = __builtin_amdgcn_s_memrealtime(); wsc[3] = __builtin_amdgcn_s_memtime();
        unsigned long long* o = a.stamps + 32 + (blockIdx.x == 0 ? 0 : 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = wsr[k]; o[4 + k] = wsc[k]; }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The 256 x 256 ping-pong kernel on v_mfma_f32_16x16x32_f16.  The K loop of gemm_f16x2_v4_kernel runs at the matrix pipe's issue rate and at the clock the chip
// holds under that load (1.44 GHz with all 256 CUs busy: profiles/r03_d_gemm_v4_kernel_stamps.log), so the MFMA SHAPE is the in-loop lever left
// (MI355X_MICROARCH.md "DVFS give-back" item 7): the same FLOPs issued as 16x16x32 run 14 % faster on random fp16 operands re-read from LDS
// (tools/micro/mfma_shape_f16.hip, profiles/r03_o_mfma_shape_f16.log: 1704 against 1496 TFLOP/s, 2 waves per SIMD, every CU).
// Same tile, waves, LDS image, DMA schedule and slot structure as v4 (VAR 0); what changes:
//   fragments   one ds_read_b128 = 16 rows x 32 k (lane l: row l & 15, 16-byte chunk l >> 4 of the row's 64 bytes; the chunk swizzle permutes inside the row, a
//               read covers 16 whole rows = 1 KB: conflict-free); X 8 row tiles, W 4 column tiles per wave; no k16 sub-steps
//   products    acc[mt][nt] (f32x4) += Wh Xl + Wl Xh + Wh Xh with W as the A operand: D[n = 4 (l >> 4) + r][m = l & 15] - a lane holds four CONSECUTIVE columns
//               of one row per tile, so every epilogue access is 16 bytes (fp32) / 8 bytes (a plane) as before
//   slots       L0 = X tiles 0-3 (8 reads) + W tiles 0-1 (4); L1 = W tiles 2-3 (4); L2 = X tiles 4-7 (8) + DMA; L3 = DMA + vmcnt; Mq = 4 x 2 tiles x 3 = 24 MFMAs (384 cycles)
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define SDVAR_MFMA3_16(acc, xh, xl, wh, wl)                                            \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);                \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);                \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0)

// four consecutive columns n .. n + 3 of row m (16x16x32 accumulator group): every epilogue; a.vec and n + 3 < N checked by the caller for the 16-byte path
template <int EPI>
__device__ __forceinline__ void h_store4(const GemmHArgs& a, float* outp, const f32x4v& acc, float wsi, int m, int n, bool fast) {
    if (m >= a.M || n >= a.N) return;
    if (!fast) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < a.N) h_store<EPI>(a, outp, acc[e], wsi, (EPI != HEPI_PARTIAL && a.bias) ? a.bias[n + e] : 0.f, m, n + e);
        return;
    }
    f32x4 v;
    if (EPI != HEPI_PARTIAL && a.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[e], wsi, bv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[e] * wsi;
    }
    if (EPI == HEPI_BIAS_GELU_PLANES) {
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = gelu_tanh_h(v[e]);
        uint2 wh, wl;
        split4h_pk(gv, wh, wl);
        uint16_t* p0 = a.outp + kb_index(m, n, a.M);
        *reinterpret_cast<uint2*>(p0) = wh; *reinterpret_cast<uint2*>(p0 + a.ops) = wl;          // (streaming stores were tried here: the 8-byte halves of a 64-byte plane row stop
                                                                                                  //  combining in L2 and fc1 at M = 4096 goes from 90 to 113 us)
        return;
    }
    if (EPI == HEPI_GATED_RES) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + (size_t)m * a.ldres + n);
        const f32x4 gt = *reinterpret_cast<const f32x4*>(a.gate + (size_t)(m / a.rows_per_gate) * a.gate_stride + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e] * gt[e];
    }
    *reinterpret_cast<f32x4*>(outp + (size_t)m * a.ldo + n) = v;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_v5_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int BM = 256, BN = 256;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int G = 4, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);

    // DMA: exactly as gemm_f16x2_v4_kernel
    const int dpl = wave >> 2, drow0 = 64 * (wave & 3) + (lane >> 2);
    const int dch = (lane & 3) ^ ((lane >> 4) & 3);
    uint32_t vx[4], vw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vx[e] = (uint32_t)(min(m0 + drow0 + 16 * e, a.M - 1) * 32 + 8 * dch) * 2u;
        vw[e] = (uint32_t)(min(n0 + drow0 + 16 * e, a.N - 1) * 32 + 8 * dch) * 2u;
    }
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32 + (size_t)dpl * a.xps);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32 + (size_t)dpl * a.wps);
    const uint32_t lds0 = SDVAR_LDS_ADDR(hsm);
    const uint32_t ldx = lds0 + (uint32_t)(dpl * 16384 + (64 * (wave & 3)) * 64), ldw = ldx + 32768u;
    auto issue_x = [&](int t) {
        const char* src = bx + (size_t)t * a.M * 64;
        const uint32_t dst = ldx + (uint32_t)(t & 1) * 65536u;
#pragma unroll
        for (int e = 0; e < 4; ++e) SDVAR_DMA16(vx[e], src, dst + 1024u * e);
    };
    auto issue_w = [&](int t) {
        const char* src = bw + (size_t)t * a.N * 64;
        const uint32_t dst = ldw + (uint32_t)(t & 1) * 65536u;
#pragma unroll
        for (int e = 0; e < 4; ++e) SDVAR_DMA16(vw[e], src, dst + 1024u * e);
    };

    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses (bytes, stage 0): row l15 of the 16-row tile, chunk lq swizzled by (row >> 2) & 3 = (l15 >> 2) & 3; plane p at + 16384 p, tile i at + 1024 i
    const uint32_t fro = (uint32_t)(l15 * 64 + 16 * (lq ^ ((l15 >> 2) & 3)));
    const uint32_t ax = lds0 + (uint32_t)(wm * 128 * 64) + fro, aw = lds0 + 32768u + (uint32_t)(wn * 64 * 64) + fro;

    issue_w(0); issue_x(0);
    if (nk > 1) { issue_w(1); issue_x(1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();

    f16x8 fx[4][2], fw[4][2];                                   // fx[row tile of the half][plane], fw[column tile][plane]
#define SDVAR_H5_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SDVAR_H5_MFMA(I0, J0)                                                                                              \
    do {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) { SDVAR_MFMA3_16(acc[(I0) + i_][(J0) + j_], fx[i_][0], fx[i_][1], fw[(J0) + j_][0], fw[(J0) + j_][1]); } \
    } while (0)
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const uint32_t so = (uint32_t)(t & 1) * 65536u;
        const uint32_t x0 = ax + so, w0 = aw + so;
        const bool p2 = t + 2 < nk;
        // ---- L0: X row tiles 0-3 of the wave's half, W column tiles 0-1
        SDVAR_LDS_RDH(fx[0][0], x0, 0);     SDVAR_LDS_RDH(fx[0][1], x0, 16384); SDVAR_LDS_RDH(fw[0][0], w0, 0);     SDVAR_LDS_RDH(fw[0][1], w0, 16384);
        SDVAR_LDS_RDH(fx[1][0], x0, 1024);  SDVAR_LDS_RDH(fx[1][1], x0, 17408); SDVAR_LDS_RDH(fw[1][0], w0, 1024);  SDVAR_LDS_RDH(fw[1][1], w0, 17408);
        SDVAR_LDS_RDH(fx[2][0], x0, 2048);  SDVAR_LDS_RDH(fx[2][1], x0, 18432); SDVAR_LDS_RDH(fx[3][0], x0, 3072);  SDVAR_LDS_RDH(fx[3][1], x0, 19456);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H5_SLOT();
        SDVAR_H5_MFMA(0, 0);                                                                                  // M0: rows 0-63 x columns 0-31
        SDVAR_H5_SLOT();
        // ---- L1: W column tiles 2-3
        SDVAR_LDS_RDH(fw[2][0], w0, 2048);  SDVAR_LDS_RDH(fw[2][1], w0, 18432); SDVAR_LDS_RDH(fw[3][0], w0, 3072);  SDVAR_LDS_RDH(fw[3][1], w0, 19456);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H5_SLOT();
        SDVAR_H5_MFMA(0, 2);                                                                                  // M1: rows 0-63 x columns 32-63
        SDVAR_H5_SLOT();
        // ---- L2: X row tiles 4-7; the W slab of this stage is free: refill it with K-step t + 2
        SDVAR_LDS_RDH(fx[0][0], x0, 4096);  SDVAR_LDS_RDH(fx[0][1], x0, 20480); SDVAR_LDS_RDH(fx[1][0], x0, 5120);  SDVAR_LDS_RDH(fx[1][1], x0, 21504);
        SDVAR_LDS_RDH(fx[2][0], x0, 6144);  SDVAR_LDS_RDH(fx[2][1], x0, 22528); SDVAR_LDS_RDH(fx[3][0], x0, 7168);  SDVAR_LDS_RDH(fx[3][1], x0, 23552);
        if (p2) issue_w(t + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H5_SLOT();
        SDVAR_H5_MFMA(4, 2);                                                                                  // M2: rows 64-127 x columns 32-63
        SDVAR_H5_SLOT();
        // ---- L3: the X slab is free too; K-step t + 1 must have landed before anybody starts it
        if (p2) { issue_x(t + 2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SDVAR_H5_SLOT();
        SDVAR_H5_MFMA(4, 0);                                                                                  // M3: rows 64-127 x columns 0-31
        SDVAR_H5_SLOT();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef SDVAR_H5_MFMA
#undef SDVAR_H5_SLOT

    const float wsi = a.wsi ? *a.wsi : 1.0f;
    const int mrow = m0 + wm * 128 + l15, ncol = n0 + wn * 64 + 4 * lq;             // + 16 i (row tile), + 16 j (column tile)
    if (EPI == HEPI_QKV) {             // a wave's 64 columns are one head of q, k or v; lane: row mrow + 16 i, columns 16 j + 4 lq + {0..3} of the head
        const QkvEpi& e = a.qk;
        const int Cq = e.H * 64, nw = n0 + wn * 64, which = nw / Cq, h = (nw - which * Cq) >> 6;
        if (nw >= a.N) return;
        const bool l2 = e.scale_mul != nullptr;
        const float sm = (which == 0) ? (l2 ? expf(fminf(e.scale_mul[h], 4.605170249938965f)) : 0.03125f) : 1.0f;
        const int NP = e.fmt == 3 ? 2 : 1;
        const size_t ps = (size_t)e.Lp * 64;
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ncol + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = mrow + 16 * i;
            float v[4][4], sq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 4; ++x) { v[j][x] = acc[i][j][x] * wsi + bv[j][x]; sq += v[j][x] * v[j][x]; }
            if (which != 2 && l2) { sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64); }        // the head's 64 columns: 4 lane quarters x 16
            if (m >= a.M) continue;
            const float nrm = l2 ? fmaxf(sqrtf(sq), 1e-12f) : 1.0f;
            const int r = m / e.l, tt = m - r * e.l;
            if (which == 0) {
                float* pq = e.q_out + (((size_t)r * e.H + h) * e.l + tt) * 64 + 4 * lq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 o;
#pragma unroll
                    for (int x = 0; x < 4; ++x) o[x] = l2 ? (v[j][x] / nrm) * sm : v[j][x] * sm;
                    *reinterpret_cast<f32x4*>(pq + 16 * j) = o;
                }
            } else {
                const bool nkk = l2 && which == 1;
                uint16_t* pk = (which == 1 ? e.k_cache : e.v_cache) + ((size_t)r * e.H + h) * NP * ps + (size_t)(e.pos0 + tt) * 64 + 4 * lq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float kv[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) kv[x] = nkk ? v[j][x] / nrm : v[j][x];
                    uint2 wh, wl;
                    split4h_pk(kv, wh, wl);
                    *reinterpret_cast<uint2*>(pk + 16 * j) = wh;
                    if (e.fmt == 3) *reinterpret_cast<uint2*>(pk + ps + 16 * j) = wl;
                }
            }
        }
        return;
    }
    float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
    const bool fast = a.vec && n0 + BN <= a.N;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) h_store4<EPI>(a, outp, acc[i][j], wsi, mrow + 16 * i, ncol + 16 * j, fast);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 192 tile (round 4): the 256 x 256 kernel above fills whole rounds of 256 workgroups only where N is a multiple of 256 and M / 256 x N / 256 is near 256; the
// QKV launches (N = 3 C: 3072 for d16, 2304 for d12) and d12's fc1 (N = 3072) run 132 - 192 tiles on 256 CUs at M = 2704 / 4096.  With 192 columns per tile N = 3072 gives
// 16 column tiles (256 tiles at M = 4096, 176 at M = 2704) and N = 2304 gives 12 (192 / 132), each with 3/4 of the matrix work.
//   waves   8 x 1: wave w owns rows 32 w .. 32 w + 31 (2 row tiles of 16) x ALL 192 columns (12 column tiles = 3 whole heads: the fused QKV epilogue keeps one head
//           inside a wave); acc[2][12] = 96 VGPRs; W is the A operand (a lane holds 4 consecutive columns of one row), as in the 256 x 256 kernel
//   LDS     X ring of 2 stages (2 planes x 256 rows x 64 B = 32 KB), W ring of 3 stages (2 planes x 192 rows x 64 B = 24 KB): 136 KB.  W of K-step t + 2 is requested in
//           L0(t) - the stage K-step t - 1 left - and X of K-step t + 2 in L1(t): both two K-steps ahead
//   slots   (ping-pong: waves 4-7 run one slot behind waves 0-3)  L0 = 4 X + 12 W fragment reads (W column tiles 0-5) + 3 DMA; M0 = 2 x 6 x 3 = 36 MFMAs;
//           L1 = 12 W fragment reads (tiles 6-11) + 4 DMA + the counted wait for K-step t + 1; M1 = 36 MFMAs
constexpr int H7_XSTAGE = 2 * 256 * 32, H7_WSTAGE = 2 * 192 * 32;          // fp16 elements per X / W stage
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_v7_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    constexpr int BM = 256, BN = 192;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int G = 4, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int late = wave >> 2;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);

    // DMA: waves 0-3 bring plane 0, waves 4-7 plane 1; a wave's instruction = 16 rows x 64 B; X: rows 64 (wave & 3) + 16 e (e < 4); W: rows 48 (wave & 3) + 16 e (e < 3)
    const int dpl = wave >> 2, dr = lane >> 2;
    const int dch = (lane & 3) ^ ((lane >> 4) & 3);
    uint32_t vx[4], vw[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) vx[e] = (uint32_t)(min(m0 + 64 * (wave & 3) + 16 * e + dr, a.M - 1) * 32 + 8 * dch) * 2u;
#pragma unroll
    for (int e = 0; e < 3; ++e) vw[e] = (uint32_t)(min(n0 + 48 * (wave & 3) + 16 * e + dr, a.N - 1) * 32 + 8 * dch) * 2u;
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32 + (size_t)dpl * a.xps);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32 + (size_t)dpl * a.wps);
    const uint32_t lds0 = SDVAR_LDS_ADDR(hsm);
    const uint32_t ldx = lds0 + (uint32_t)(dpl * 16384 + 64 * (wave & 3) * 64);
    const uint32_t ldw = lds0 + 65536u + (uint32_t)(dpl * 12288 + 48 * (wave & 3) * 64);
    auto issue_x = [&](int t) {
        const char* src = bx + (size_t)t * a.M * 64;
        const uint32_t dst = ldx + (uint32_t)(t & 1) * 32768u;
#pragma unroll
        for (int e = 0; e < 4; ++e) SDVAR_DMA16(vx[e], src, dst + 1024u * e);
    };
    auto issue_w = [&](int t) {
        const char* src = bw + (size_t)t * a.N * 64;
        const uint32_t dst = ldw + (uint32_t)(t % 3) * 24576u;
#pragma unroll
        for (int e = 0; e < 3; ++e) SDVAR_DMA16(vw[e], src, dst + 1024u * e);
    };

    f32x4v acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const uint32_t fro = (uint32_t)(l15 * 64 + 16 * (lq ^ ((l15 >> 2) & 3)));
    const uint32_t ax = lds0 + (uint32_t)(wave * 32 * 64) + fro, aw = lds0 + 65536u + fro;

    issue_w(0); issue_x(0);
    if (nk > 1) { issue_w(1); issue_x(1); asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();

    f16x8 fx[2][2], fw[6][2];
#define SDVAR_H7_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SDVAR_H7_MFMA(J0)                                                                                                  \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_)                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) { SDVAR_MFMA3_16(acc[i_][(J0) + j_], fx[i_][0], fx[i_][1], fw[j_][0], fw[j_][1]); } \
    } while (0)
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const uint32_t x0 = ax + (uint32_t)(t & 1) * 32768u, w0 = aw + (uint32_t)(t % 3) * 24576u;
        const bool p2 = t + 2 < nk;
        // ---- L0: the wave's two X row tiles, W column tiles 0-5; W of K-step t + 2 into the stage K-step t - 1 left (every wave is past its L1(t - 1))
        SDVAR_LDS_RDH(fx[0][0], x0, 0);     SDVAR_LDS_RDH(fx[0][1], x0, 16384); SDVAR_LDS_RDH(fw[0][0], w0, 0);     SDVAR_LDS_RDH(fw[0][1], w0, 12288);
        SDVAR_LDS_RDH(fx[1][0], x0, 1024);  SDVAR_LDS_RDH(fx[1][1], x0, 17408); SDVAR_LDS_RDH(fw[1][0], w0, 1024);  SDVAR_LDS_RDH(fw[1][1], w0, 13312);
        SDVAR_LDS_RDH(fw[2][0], w0, 2048);  SDVAR_LDS_RDH(fw[2][1], w0, 14336); SDVAR_LDS_RDH(fw[3][0], w0, 3072);  SDVAR_LDS_RDH(fw[3][1], w0, 15360);
        SDVAR_LDS_RDH(fw[4][0], w0, 4096);  SDVAR_LDS_RDH(fw[4][1], w0, 16384); SDVAR_LDS_RDH(fw[5][0], w0, 5120);  SDVAR_LDS_RDH(fw[5][1], w0, 17408);
        if (p2) issue_w(t + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H7_SLOT();
        SDVAR_H7_MFMA(0);                                                                                     // M0: 32 rows x columns 0-95
        SDVAR_H7_SLOT();
        // ---- L1: W column tiles 6-11; the X stage is free (both halves read it in their L0): refill it with K-step t + 2
        SDVAR_LDS_RDH(fw[0][0], w0, 6144);  SDVAR_LDS_RDH(fw[0][1], w0, 18432); SDVAR_LDS_RDH(fw[1][0], w0, 7168);  SDVAR_LDS_RDH(fw[1][1], w0, 19456);
        SDVAR_LDS_RDH(fw[2][0], w0, 8192);  SDVAR_LDS_RDH(fw[2][1], w0, 20480); SDVAR_LDS_RDH(fw[3][0], w0, 9216);  SDVAR_LDS_RDH(fw[3][1], w0, 21504);
        SDVAR_LDS_RDH(fw[4][0], w0, 10240); SDVAR_LDS_RDH(fw[4][1], w0, 22528); SDVAR_LDS_RDH(fw[5][0], w0, 11264); SDVAR_LDS_RDH(fw[5][1], w0, 23552);
        // K-step t + 1 must have landed before ANY wave starts it: the early half reads it two slots from here, when the late half has just left this segment - so the
        // counted wait stands HERE, not behind M1 (this wave's share; the barriers make it everybody's): the 7 instructions of K-step t + 2 may stay in flight
        if (p2) { issue_x(t + 2); asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_H7_SLOT();
        SDVAR_H7_MFMA(6);                                                                                     // M1: 32 rows x columns 96-191
        SDVAR_H7_SLOT();
    }
    if (!late) __builtin_amdgcn_s_barrier();
#undef SDVAR_H7_MFMA
#undef SDVAR_H7_SLOT

    const float wsi = a.wsi ? *a.wsi : 1.0f;
    const int mrow = m0 + wave * 32 + l15, ncol = n0 + 4 * lq;                        // + 16 i (row tile), + 16 j (column tile)
    if (EPI == HEPI_QKV) {             // the wave's 192 columns are three whole heads of q, k or v (N = 3 H 64, n0 a multiple of 64)
        const QkvEpi& e = a.qk;
        const int Cq = e.H * 64;
        const bool l2 = e.scale_mul != nullptr;
        const int NP = e.fmt == 3 ? 2 : 1;
        const size_t ps = (size_t)e.Lp * 64;
#pragma unroll
        for (int hh = 0; hh < 3; ++hh) {
            const int nw = n0 + 64 * hh;
            if (nw >= a.N) break;
            const int which = nw / Cq, h = (nw - which * Cq) >> 6;
            const float sm = (which == 0) ? (l2 ? expf(fminf(e.scale_mul[h], 4.605170249938965f)) : 0.03125f) : 1.0f;
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nw + 4 * lq + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = mrow + 16 * i;
                float v[4][4], sq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int x = 0; x < 4; ++x) { v[j][x] = acc[i][4 * hh + j][x] * wsi + bv[j][x]; sq += v[j][x] * v[j][x]; }
                if (which != 2 && l2) { sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64); }
                if (m >= a.M) continue;
                const float nrm = l2 ? fmaxf(sqrtf(sq), 1e-12f) : 1.0f;
                const int r = m / e.l, tt = m - r * e.l;
                if (which == 0) {
                    float* pq = e.q_out + (((size_t)r * e.H + h) * e.l + tt) * 64 + 4 * lq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 o;
#pragma unroll
                        for (int x = 0; x < 4; ++x) o[x] = l2 ? (v[j][x] / nrm) * sm : v[j][x] * sm;
                        *reinterpret_cast<f32x4*>(pq + 16 * j) = o;
                    }
                } else {
                    const bool nkk = l2 && which == 1;
                    uint16_t* pk = (which == 1 ? e.k_cache : e.v_cache) + ((size_t)r * e.H + h) * NP * ps + (size_t)(e.pos0 + tt) * 64 + 4 * lq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float kv[4];
#pragma unroll
                        for (int x = 0; x < 4; ++x) kv[x] = nkk ? v[j][x] / nrm : v[j][x];
                        uint2 wh, wl;
                        split4h_pk(kv, wh, wl);
                        *reinterpret_cast<uint2*>(pk + 16 * j) = wh;
                        if (e.fmt == 3) *reinterpret_cast<uint2*>(pk + ps + 16 * j) = wl;
                    }
                }
            }
        }
        return;
    }
    float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
    const bool fast = a.vec && n0 + BN <= a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) h_store4<EPI>(a, outp, acc[i][j], wsi, mrow + 16 * i, ncol + 16 * j, fast);
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny kernel for M <= 80 rows (stages 0 - 1 and the first verify chunk: 16 - 80 CFG rows): these launches are weight streams - 4 - 17 MB of planes read once for
// 0.1 - 0.8 GFLOP - and the LDS-ring kernels above spend their time on everything else: split-K over 6 - 24 workgroups per column panel, slabs, a reduce launch
// or a slab-summing consumer, a DMA ring that never fills (9 - 12 us per launch against a 2 - 3 us stream: profiles/r02_gemm_sweep_cold_full.jsonl).
// Here (cdna_hip_programming.md "GEMV / M <= 16 decode weights": operand streamed once and not shared across waves -> straight to VGPRs, deep unroll, late wait):
//   workgroup = 16 output columns x ALL rows x a K range of at most 8 NW K-steps; NW = 4 waves, wave w takes K-steps w, w + NW, ...
//   every W fragment of a wave (16 columns x 32 k x 2 planes = 2 KB per K-step: one contiguous run of the K-blocked planes, which IS the 16x16x32 A operand) is
//   requested up front - up to 16 loads of 1 KB in flight per wave, ONE HBM round trip per launch; X fragments (L2-resident) likewise, in two batches from MT = 3;
//   acc[mt] += Wh Xl + Wl Xh + Wh Xh on v_mfma_f32_16x16x32_f16; the waves' partial sums meet in LDS (summed in wave order: deterministic) and wave mt finishes row
//   tile mt with the epilogue of the other kernels (h_store4): no slabs and no second launch unless K > 8 NW K-steps (fc2 at NW = 4: split over workgroups).
template <int MT, int NW, int EPI>
__global__ __launch_bounds__(64 * NW) void gemm_f16x2_skinny_kernel(GemmHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hsm[];
    const int tiles_n = (a.N + 15) / 16;
    const int ks = blockIdx.x / tiles_n, tn = blockIdx.x - ks * tiles_n;
    const int n0 = tn * 16;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / HBK - kt0, a.k_per_split);                    // <= 8 NW (host)
    // lane l of a fragment: row l15 of the 16-row tile, k 8 lq .. 8 lq + 7 of the K-step: 16 bytes at ((kt rows + row) 32 + 8 lq) elements of the plane
    const uint16_t* pw = a.W + ((size_t)kt0 * a.N + min(n0 + l15, a.N - 1)) * 32 + 8 * lq;
    const uint16_t* px[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) px[i] = a.X + ((size_t)kt0 * a.M + min(16 * i + l15, a.M - 1)) * 32 + 8 * lq;
    const size_t wstep = (size_t)a.N * 32, xstep = (size_t)a.M * 32;
    f16x8 fw[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int t = wave + s * NW;
        if (t < nk) {
#pragma unroll
            for (int p = 0; p < 2; ++p) fw[s][p] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(pw + (size_t)t * wstep + p * a.wps));
        }
    }
    f32x4v acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    constexpr int XB = MT <= 2 ? 8 : 4;                      // K-steps per batch of X fragments (registers: XB x MT x 2 planes x 4)
#pragma unroll
    for (int b0 = 0; b0 < 8; b0 += XB) {
        f16x8 fx[XB][MT][2];
#pragma unroll
        for (int s = 0; s < XB; ++s) {
            const int t = wave + (b0 + s) * NW;
            if (t < nk) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) fx[s][i][p] = *reinterpret_cast<const f16x8*>(px[i] + (size_t)t * xstep + p * a.xps);
            }
        }
#pragma unroll
        for (int s = 0; s < XB; ++s) {
            const int t = wave + (b0 + s) * NW;
            if (t < nk) {
#pragma unroll
                for (int i = 0; i < MT; ++i) { SDVAR_MFMA3_16(acc[i], fx[s][i][0], fx[s][i][1], fw[b0 + s][0], fw[b0 + s][1]); }
            }
        }
    }
    // cross-wave sum through LDS: red[wave][mt][lane] (16 bytes per lane), then wave mt (mod NW) owns row tile mt
    f32x4v* red = reinterpret_cast<f32x4v*>(hsm);
#pragma unroll
    for (int i = 0; i < MT; ++i) red[(wave * MT + i) * 64 + lane] = acc[i];
    __syncthreads();
    const float wsi = a.wsi ? *a.wsi : 1.0f;
    float* outp = (EPI == HEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
    const bool fast = a.vec && n0 + 16 <= a.N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i % NW != wave) continue;                                      // wave-uniform
        f32x4v sum = red[i * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) { const f32x4v q = red[(w * MT + i) * 64 + lane]; sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2]; sum[3] += q[3]; }
        h_store4<EPI>(a, outp, sum, wsi, 16 * i + l15, n0 + 4 * lq, fast);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-block kernel for M <= 80 rows (round 4): ONE launch per GEMM of a transformer block where stages 0 - 1 ran three - LayerNorm + modulation in the operand
// prologue, the product, and for QKV the per-head finish (bias, q / k L2 norm, q scale, k / v rows into the cache planes: basic_var.py:101-109, 157-158).
//   workgroup = 16 rows x 16 NT columns x ALL of K; 8 waves, wave w takes K-steps w, w + 8, ... (at most KS of them); grid = (N / (16 NT), ceil(M / 16)).
//   W fragments global -> VGPR, all in flight at once (one HBM round trip per launch, as in the skinny kernel above).
//   LN = 1 (QKV, fc1, head; K = C): the X operand is built from the fp32 residual stream.  A lane holds its row's 8 k of every K-step of its wave (row l15, k
//       32 t + 8 lq ..: exactly its MFMA B fragment), the row statistics meet through LDS (sum, then sum of squared deviations: the two-pass form of
//       ln_modulate_kernel; waves summed in order: deterministic), then ((x - mean) rstd)(1 + scale) + shift is split into the two fp16 planes in registers.  The
//       stand-alone ln_modulate launch, its planes in HBM and their re-read are gone; every column block redoes its 16 rows' LayerNorm (16 K values per workgroup).
//   LN = 0 (proj, fc2): X planes from global as before; with KS = 16 a workgroup streams K = 4096 unsplit - no slabs, no pending residual for a later kernel to sum.
//   The waves' partial tiles meet in LDS; wave j finishes column tile j.  HEPI_QKV (NT = 4: the 64 columns of one head of q, k or v): the head's sum of squares
//   goes through LDS once more, then q -> (R, H, l, 64) fp32, k (normalised) / v -> the cache planes at pos0 + t.
struct RowBlkArgs {
    GemmHArgs g;
    const float* x; int ldx;                          // LN = 1: fp32 (M, K) activations
    const float* scale; const float* shift;           // modulation vectors of CFG row m / rows_per_img, stride mod_stride
    int rows_per_img, mod_stride;
    float eps;
};

__device__ __forceinline__ void split8h_pk(const float* v, f16x8& h, f16x8& l) {
    uint2 h0, l0, h1, l1;
    split4h_pk(v, h0, l0); split4h_pk(v + 4, h1, l1);
    const u32x4 hh = {h0.x, h0.y, h1.x, h1.y}, ll = {l0.x, l0.y, l1.x, l1.y};
    h = __builtin_bit_cast(f16x8, hh); l = __builtin_bit_cast(f16x8, ll);
}

template <int NT, int KS, int EPI, int LN>
__global__ __launch_bounds__(512) void gemm_f16x2_rowblk_kernel(RowBlkArgs ra) {
    __shared__ f32x4v red_acc[8 * NT * 64];            // partial tiles of the 8 waves
    __shared__ float red_st[2][8][16];                 // LN: per-wave row sums / squared deviations
    __shared__ float red_sq[NT][16];                   // HEPI_QKV: per column tile sum of squares of a row
    const GemmHArgs& a = ra.g;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * 16;
    const int nk = a.K / HBK;
    const int row = min(m0 + l15, a.M - 1);
    // K-steps of this wave: t = wave + 8 s.  Straight-line code: a K-step past the end is CLAMPED for the loads and its X fragment is zero (a branch per K-step
    // made the compiler wait for each step's loads inside its branch: four serialised round trips, 13 us per launch instead of 7).
    int tc[KS]; bool tv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { const int t = wave + 8 * s; tv[s] = t < nk; tc[s] = tv[s] ? t : nk - 1; }
    f32x4v acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    f16x8 fw[NT][KS][2];
    // W fragments: column n0 + 16 j + l15, k 8 lq .. of K-step t; requested AFTER the short-latency operand loads of the LN prologue (vmcnt counts in order:
    // the prologue then waits for its own loads only while the weight stream stays in flight)
    auto load_w = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint16_t* pw = a.W + (size_t)min(n0 + 16 * j + l15, a.N - 1) * 32 + 8 * lq;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) fw[j][s][p] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(pw + (size_t)tc[s] * a.N * 32 + p * a.wps));
        }
    };
    if (LN) {
        f32x4 xa[KS][2], sa[KS][2], ha[KS][2];
        const float* px = ra.x + (size_t)row * ra.ldx + 8 * lq;
        const size_t mo = (size_t)(row / ra.rows_per_img) * ra.mod_stride + 8 * lq;
#pragma unroll
        for (int s = 0; s < KS; ++s) { xa[s][0] = *reinterpret_cast<const f32x4*>(px + 32 * tc[s]); xa[s][1] = *reinterpret_cast<const f32x4*>(px + 32 * tc[s] + 4); }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sa[s][0] = *reinterpret_cast<const f32x4*>(ra.scale + mo + 32 * tc[s]); sa[s][1] = *reinterpret_cast<const f32x4*>(ra.scale + mo + 32 * tc[s] + 4);
            ha[s][0] = *reinterpret_cast<const f32x4*>(ra.shift + mo + 32 * tc[s]); ha[s][1] = *reinterpret_cast<const f32x4*>(ra.shift + mo + 32 * tc[s] + 4);
        }
        load_w();
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float ps = ((xa[s][0][0] + xa[s][0][1]) + (xa[s][0][2] + xa[s][0][3])) + ((xa[s][1][0] + xa[s][1][1]) + (xa[s][1][2] + xa[s][1][3]));
            sum += tv[s] ? ps : 0.f;
        }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        if (lq == 0) red_st[0][wave][l15] = sum;
        __syncthreads();
        float tot = red_st[0][0][l15];
#pragma unroll
        for (int w = 1; w < 8; ++w) tot += red_st[0][w][l15];
        const float mean = tot / (float)a.K;
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float ps = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xa[s][h][e] - mean; ps += d * d; }
            ss += tv[s] ? ps : 0.f;
        }
        ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
        if (lq == 0) red_st[1][wave][l15] = ss;
        __syncthreads();
        float tot2 = red_st[1][0][l15];
#pragma unroll
        for (int w = 1; w < 8; ++w) tot2 += red_st[1][w][l15];
        const float rstd = 1.0f / sqrtf(tot2 / (float)a.K + ra.eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = ((xa[s][h][e] - mean) * rstd) * (sa[s][h][e] + 1.0f) + ha[s][h][e]; o[4 * h + e] = tv[s] ? v : 0.f; }
            f16x8 xh, xl;
            split8h_pk(o, xh, xl);
#pragma unroll
            for (int j = 0; j < NT; ++j) { SDVAR_MFMA3_16(acc[j], xh, xl, fw[j][s][0], fw[j][s][1]); }
        }
    } else {
        load_w();
        const uint16_t* pxp = a.X + (size_t)row * 32 + 8 * lq;
        constexpr int XB = KS < 8 ? KS : 8;
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int b0 = 0; b0 < KS; b0 += XB) {
            f16x8 fb[XB][2];
#pragma unroll
            for (int s = 0; s < XB; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[s][p] = *reinterpret_cast<const f16x8*>(pxp + (size_t)tc[b0 + s] * a.M * 32 + p * a.xps);
#pragma unroll
            for (int s = 0; s < XB; ++s) {
                const f16x8 xh = tv[b0 + s] ? fb[s][0] : zero8, xl = tv[b0 + s] ? fb[s][1] : zero8;
#pragma unroll
                for (int j = 0; j < NT; ++j) { SDVAR_MFMA3_16(acc[j], xh, xl, fw[j][b0 + s][0], fw[j][b0 + s][1]); }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) red_acc[(wave * NT + j) * 64 + lane] = acc[j];
    __syncthreads();
    // ---- wave j finishes column tile j (summed in wave order)
    const float wsi = a.wsi ? *a.wsi : 1.0f;
    const int m = m0 + l15;
    f32x4v sum4 = {0.f, 0.f, 0.f, 0.f};
    if (wave < NT) {
        sum4 = red_acc[wave * 64 + lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) { const f32x4v q = red_acc[(w * NT + wave) * 64 + lane]; sum4[0] += q[0]; sum4[1] += q[1]; sum4[2] += q[2]; sum4[3] += q[3]; }
    }
    if (EPI == HEPI_QKV) {                 // NT = 4, n0 = 64 x (head of q | k | v): all eight waves reach the barrier
        const QkvEpi& e = a.qk;
        const int cb = n0 >> 6, which = cb / e.H, h = cb - which * e.H;
        const bool l2 = e.scale_mul != nullptr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < NT) {
            const int n = n0 + 16 * wave + 4 * lq;
            const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            float sq = 0.f;
#pragma unroll
            for (int x = 0; x < 4; ++x) { v[x] = sum4[x] * wsi + bv[x]; sq += v[x] * v[x]; }
            sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
            if (lq == 0) red_sq[wave][l15] = sq;
        }
        __syncthreads();
        if (wave >= NT || m >= a.M) return;
        const float sqt = (red_sq[0][l15] + red_sq[1][l15]) + (red_sq[2 % NT][l15] + red_sq[3 % NT][l15]);
        const float nrm = (l2 && which < 2) ? fmaxf(sqrtf(sqt), 1e-12f) : 1.0f;
        const int r = m / e.l, t = m - r * e.l, c = 16 * wave + 4 * lq;
        if (which == 0) {
            const float sm = l2 ? expf(fminf(e.scale_mul[h], 4.605170249938965f)) : 0.03125f;
            f32x4 o;
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] = l2 ? (v[x] / nrm) * sm : v[x] * sm;
            *reinterpret_cast<f32x4*>(e.q_out + (((size_t)r * e.H + h) * e.l + t) * 64 + c) = o;
        } else {
            const int NP = e.fmt == 3 ? 2 : 1;
            const size_t ps = (size_t)e.Lp * 64;
            uint16_t* pk = (which == 1 ? e.k_cache : e.v_cache) + ((size_t)r * e.H + h) * NP * ps + (size_t)(e.pos0 + t) * 64 + c;
            float kv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) kv[x] = (l2 && which == 1) ? v[x] / nrm : v[x];
            uint2 wh, wl;
            split4h_pk(kv, wh, wl);
            *reinterpret_cast<uint2*>(pk) = wh;
            if (e.fmt == 3) *reinterpret_cast<uint2*>(pk + ps) = wl;
        }
        return;
    }
    if (wave >= NT) return;
    constexpr int SEPI = EPI == HEPI_QKV ? HEPI_BIAS : EPI;
    h_store4<SEPI>(a, a.out, sum4, wsi, m, n0 + 16 * wave + 4 * lq, a.vec && n0 + 16 * NT <= a.N);
}

// out = epi( sum_s slab[s] + bias ) for the split-K path; the GELU variant writes planes
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_h_kernel(const float* __restrict__ ws, int split, const float* __restrict__ bias, float* out,
                                                              uint16_t* outp, size_t ops, const float* res, const float* __restrict__ gate, int M, int N,
                                                              int ldo, int ldres, int rows_per_gate, int gate_stride) {
    const int nv = N >> 2;
    const size_t total = (size_t)M * nv, slab = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), n = (int)(i % nv) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + (size_t)m * N + n);
        int s = 1;
        for (; s + 7 < split; s += 8) {               // eight slab loads in flight, added in slice order (one round trip per group, not per slab)
            f32x4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = *reinterpret_cast<const f32x4*>(ws + (size_t)(s + u) * slab + (size_t)m * N + n);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[0] += p[u][0]; acc[1] += p[u][1]; acc[2] += p[u][2]; acc[3] += p[u][3]; }
        }
        if (s < split) {
            f32x4 p[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) if (s + u < split) p[u] = *reinterpret_cast<const f32x4*>(ws + (size_t)(s + u) * slab + (size_t)m * N + n);
#pragma unroll
            for (int u = 0; u < 7; ++u) if (s + u < split) { acc[0] += p[u][0]; acc[1] += p[u][1]; acc[2] += p[u][2]; acc[3] += p[u][3]; }
        }
        uint16_t pl[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (bias ? bias[n + e] : 0.f);
            if (EPI == HEPI_BIAS_GELU_PLANES) { split2h(gelu_tanh_h(v), pl[0][e], pl[1][e]); continue; }
            if (EPI == HEPI_GATED_RES) v = res[(size_t)m * ldres + n + e] + v * gate[(size_t)(m / rows_per_gate) * gate_stride + n + e];
            out[(size_t)m * ldo + n + e] = v;
        }
        if (EPI == HEPI_BIAS_GELU_PLANES) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                uint2 w;
                w.x = (uint32_t)pl[p][0] | ((uint32_t)pl[p][1] << 16); w.y = (uint32_t)pl[p][2] | ((uint32_t)pl[p][3] << 16);
                *reinterpret_cast<uint2*>(outp + p * ops + kb_index(m, n, M)) = w;
            }
        }
    }
}

// Tail tiles of a hybrid launch of the 256-row kernel (launch_h3_hybrid): out = epi( sum_s slab[s][tile] + bias ) for the tiles
// [tile_off, tile_off + tile_cnt); 4 workgroups per tile, the tile id -> (row tile, column tile) map is the kernel's.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_tiles_h_kernel(const float* __restrict__ ws, int split, GemmHArgs a) {
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + HBN - 1) / HBN;
    const int t = blockIdx.x >> 5, part = blockIdx.x & 31, lid = a.tile_off + t;       // 32 workgroups per tile, 8 rows each
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * 256, n0 = tn * HBN;
    const size_t slab = (size_t)a.tile_cnt * (256 * 128);
    {
        const int i = threadIdx.x;                               // 8 rows x 32 float4: one per thread
        const int row = part * 8 + (i >> 5), c4 = (i & 31) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m >= a.M || n >= a.N) return;
        const float* p = ws + (size_t)t * (256 * 128) + row * 128 + c4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(p);
        for (int s2 = 1; s2 < split; ++s2) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + s2 * slab);
            acc[0] += q[0]; acc[1] += q[1]; acc[2] += q[2]; acc[3] += q[3];
        }
        uint16_t pl[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (a.bias ? a.bias[n + e] : 0.f);
            if (EPI == HEPI_BIAS_GELU_PLANES) { split2h(gelu_tanh_h(v), pl[0][e], pl[1][e]); continue; }
            if (EPI == HEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n + e] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n + e];
            a.out[(size_t)m * a.ldo + n + e] = v;
        }
        if (EPI == HEPI_BIAS_GELU_PLANES) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                uint2 w;
                w.x = (uint32_t)pl[pp][0] | ((uint32_t)pl[pp][1] << 16); w.y = (uint32_t)pl[pp][2] | ((uint32_t)pl[pp][3] << 16);
                *reinterpret_cast<uint2*>(a.outp + pp * a.ops + kb_index(m, n, a.M)) = w;
            }
        }
    }
}

// ---- weight scale: 2^S with max|w| * 2^S in (2^12, 2^13]; sc[0] = 2^S, sc[1] = 2^-S (device scalars, no host sync at bind)
__global__ void absmax_kernel(const float* __restrict__ x, size_t n, unsigned int* out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));        // non-negative floats order like their bit patterns
}
__global__ void weight_scale_kernel(const unsigned int* mx, float* sc) {
    const float m = __uint_as_float(*mx);
    int e = 0;
    if (m > 0.f && m < INFINITY) (void)frexpf(m, &e);                     // m = f * 2^e, f in [0.5, 1)
    int S = (m > 0.f && m < INFINITY) ? 13 - e : 0;
    S = S < -24 ? -24 : (S > 40 ? 40 : S);
    sc[0] = ldexpf(1.0f, S); sc[1] = ldexpf(1.0f, -S);
}

int weight_scale_f16(const float* w, size_t n, float* sc, hipStream_t stream) {
    SDVAR_CHECK_ARG(w && sc && n > 0, "weight_scale_f16: null operand");
    unsigned int* mx = reinterpret_cast<unsigned int*>(sc) + 2;           // sc holds 4 floats: scale, 1/scale, scratch, unused
    SDVAR_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int), stream));
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, w, n, mx);
    SDVAR_LAUNCH_CHECK();
    hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1), 0, stream, mx, sc);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// fp32 (rows, cols) row-major -> K-blocked planes [2][cols/32][rows][32] fp16 of x * (*scale) (weights at bind time, tests)
__global__ __launch_bounds__(256) void split_planes_h_kernel(const float* __restrict__ x, uint16_t* __restrict__ p, int rows, int cols, size_t ps, const float* scale) {
    const size_t n4 = (size_t)rows * cols / 4;
    const float sc = scale ? *scale : 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (cols / 4)), k = (int)(i % (cols / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * cols + k);
        uint16_t q[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2h(v[e] * sc, q[0][e], q[1][e]);
        const size_t o = kb_index(row, k, rows);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint2 w;
            w.x = (uint32_t)q[j][0] | ((uint32_t)q[j][1] << 16); w.y = (uint32_t)q[j][2] | ((uint32_t)q[j][3] << 16);
            *reinterpret_cast<uint2*>(p + j * ps + o) = w;
        }
    }
}

int split_planes_f16(const float* x, uint16_t* planes, int rows, int cols, size_t plane_stride, const float* scale, hipStream_t stream) {
    SDVAR_CHECK_ARG(x && planes && rows > 0 && cols > 0 && cols % 32 == 0 && plane_stride % 8 == 0, "split_planes_f16: need cols %% 32 == 0 (rows=%d cols=%d)", rows, cols);
    const size_t blocks = ((size_t)rows * cols / 4 + 255) / 256;
    hipLaunchKernelGGL(split_planes_h_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, planes, rows, cols, plane_stride, scale);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

float* splitk_workspace(size_t* floats);     // gemm.hip: the shared slab workspace

// cost-model constants (see choose_cfg_h).  Fitted to a sweep with HBM-COLD weights (tools/micro/gemm_sweep_cold.sh: rotating weight tensors, as inside a
// model pass; the earlier fit re-read one weight tensor out of the Infinity Cache) in which the launches whose K-slice sum a consumer kernel takes over are
// timed as the slab launch alone and charged the consumer's slab reads instead of a reduce launch:
// `tools/fit_gemm_model.py profiles/r02_gemm_sweep_cold_full.jsonl 192 profiles/r02_gemm_sweep_cold_slab.jsonl`: geometric-mean regret 1.1 %, worst case
// 16 %, over the d12 / d16 shapes incl. gamma = 2 chunks (the constants before this fit: 6 % / 1.44x on the same data)
#define CM_R256 1
#define CM_R128 1
#define CM_R64 2
#define CM_R32 2
#define CM_P256 1.0
#define CM_P64 1.1
#define CM_P32 1.8
#define CM_L1 1.0
#define CM_L2 0.8
#define CM_L3 1.0
#define CM_KOVER 260.0
#define CM_FIX 8000.0
#define CM_FIXBM 80.0
#define CM_RED0 4000.0
#define CM_REDBW 5000.0
#define CM_DEFBW 2000.0     // bytes per cycle at which a deferring consumer (ln_modulate, qk_norm_append) reads the slabs: not fitted
#define CM_K4 2850.0        // 256 x 256 kernel: cost per K-step in the units of the other tiles (3072 matrix-pipe cycles per K-step, at the higher clock the 16x16x32 shape holds)
#define CM_FIX4 50000.0     // ... and its prologue + epilogue + launch, in the units of the other tiles' costs (calibrated on M = 2704 / 4096 / 6800, profiles/r03_gemm_tile_ab.log)

static int g_force_bm_h = 0, g_force_split_h = 0;
void debug_set_gemm_cfg_h(int bm, int split) { g_force_bm_h = bm; g_force_split_h = split; }
// test aid: {row tile of the last gemm_f16x2_nt call of this host thread, its K split, launches that took the hybrid tail split since the last read,
// QKV launches that finished q and k in their epilogue since the last read} - tests assert that the path they mean to cover is the one that ran
static thread_local int g_last_cfg_h[4] = {0, 0, 0, 0};
void debug_get_gemm_cfg_h(int* out) { for (int i = 0; i < 4; ++i) out[i] = g_last_cfg_h[i]; g_last_cfg_h[2] = g_last_cfg_h[3] = 0; }

// same cost model as gemm_bf16x3.hip with half the matrix work per K-step: 3 MFMAs x 32 cycles per 16 k per 32x32 tile
static int g_small_pp = -1;       // SDVAR_GEMM_SMALL_PP (A/B runs): 0 = the 4-wave ring kernel for every small tile, 1 = the K-split ping-pong kernel for 64-row tiles, 2 (default) = for 32-row tiles too
static int small_pp_on() {          // 0 = ring kernel for every small tile, 1 = K-split ping-pong kernel for 64-row tiles, 2 = for 32-row tiles too
    if (g_small_pp < 0) { const char* e = getenv("SDVAR_GEMM_SMALL_PP"); g_small_pp = e ? atoi(e) : 2; }          // default 2: qkv / fc1 at M = 144 / 256 16.7 -> 14.0 us, gemm_small class -2 % (profiles/r03_U_pp32_ab.log)
    return g_small_pp;
}
static inline bool bm_is(int a, int b) { return a == b; }
static void choose_cfg_h(int M, int N, int K, size_t ws_floats, int* bm_out, int* split_out, int* tail_out, bool allow_hybrid, bool deferred) {
    const int nkt = K / HBK, tiles_n = (N + HBN - 1) / HBN;
    double best = 1e30; int bbm = 128, bs = 1, btail = 0;
    const bool spp = small_pp_on() >= 1, spp32 = small_pp_on() >= 2;       // the 64-row tile runs on gemm_f16x2_small_pp_kernel: 144 KB of LDS = ONE workgroup per CU, K-step ~0.8 of the ring kernel's (profiles/r03_x_smallpp_ab.log)
    // per row-tile constants fitted to tools/gemm_bench.py --mode bf16x3 --sweep --dump (tools/fit_gemm_model.py):
    //   resident workgroups per CU, K-step cost factor over the MFMA time, slowdown when 1 / 2 / 3 workgroups share a CU
    const int bms[4] = {256, 128, 64, 32};
    const int resident[4] = {CM_R256, CM_R128, CM_R64, CM_R32};
    const double kfac[4] = {CM_P256, 1.0, CM_P64, CM_P32};
    const double lat[5] = {0.0, CM_L1, CM_L2, CM_L3, 1.0};
    for (int bi = 0; bi < 4; ++bi) {
        const bool ppk = (bm_is(bms[bi], 64) && spp) || (bm_is(bms[bi], 32) && spp32);
        const int bm = bms[bi], res = ppk ? 1 : resident[bi];
        const int tiles = ((M + bm - 1) / bm) * tiles_n;
        const double ktile = 192.0 * (bm / 32) * kfac[bi] * (ppk ? 0.8 : 1.0);              // 3 MFMAs x 32 cycles x 2 k16-steps per 32x32 sub-tile
        for (int split = 1; split <= 32 && split <= nkt / 2; ++split) {
            if (split > 1 && ((size_t)split * M * N > ws_floats || N % 4)) break;
            const int kps = (nkt + split - 1) / split;
            if ((nkt + kps - 1) / kps != split) continue;
            const long blocks = (long)tiles * split;
            const long per_cu = (blocks + 255) / 256;
            const double T = kps * (ktile + CM_KOVER) + CM_FIX + CM_FIXBM * bm;    // + per-K-step sync/refill, prologue + epilogue
            const long full = per_cu / res, rem = per_cu % res;
            const double l_full = (bm == 256) ? 1.0 : lat[res < 4 ? res : 4], l_rem = (bm == 256) ? 1.0 : lat[rem < 4 ? rem : 4];
            double cyc = full * res * T * l_full + (rem ? rem * T * l_rem : 0.0);
            if (split > 1) cyc += deferred ? (double)split * M * N * 4.0 / CM_DEFBW : CM_RED0 + (double)(split + 1) * M * N * 4.0 / CM_REDBW;
            if (cyc < best) { best = cyc; bbm = bm; bs = split; btail = 0; }
        }
        // hybrid for the 256-row tile: the full rounds run unsplit, only the last, partial round is split along K so that it, too,
        // spreads over the CUs (264 tiles = 256 + 8: the 8 cost a whole second round otherwise)
        if (allow_hybrid && bm == 256 && tiles > 256 && tiles % 256 && N % 4 == 0) {
            const long fullr = tiles / 256, remt = tiles % 256;
            const double Tfull = nkt * (ktile + CM_KOVER) + CM_FIX + CM_FIXBM * bm;
            const int cand[7] = {2, 3, 4, 6, 8, 12, 16};
            for (int ci = 0; ci < 7; ++ci) {
                const int ts = cand[ci];
                if (ts > nkt / 2 || (size_t)ts * remt * (256 * 128) > ws_floats) continue;
                const int kps = (nkt + ts - 1) / ts;
                if ((nkt + kps - 1) / kps != ts) continue;
                const long rounds = (remt * ts + 255) / 256;
                // the two extra launches are not free: ~10 us of prologue / slab epilogue / launch latency for the tail kernel, ~6 us for the reduce
                const double cyc = fullr * Tfull + rounds * (kps * (ktile + CM_KOVER) + 20000.0) + 12000.0 + (double)(ts + 1) * remt * (256.0 * 128.0) * 4.0 / CM_REDBW;
                if (cyc < best) { best = cyc; bbm = 256; bs = 1; btail = ts; }
            }
        }
    }
    // the 256 x 256 ping-pong kernel (bm code 512): its K loop runs at the matrix pipe's issue rate (3072 cycles per K-step for twice the tile, in-kernel stamps:
    // tools/micro/gemm_v4_stamps.py) but it needs one workgroup per CU and whole rounds of 256 tiles; unsplit only
    static const bool no_v4 = getenv("SDVAR_GEMM_NO_V4") != nullptr;       // A/B runs only
    if (!no_v4 && N >= 256 && M > 512) {
        const long tiles4 = (long)((M + 255) / 256) * ((N + 255) / 256), rounds = (tiles4 + 255) / 256;
        const double cyc = rounds * (nkt * CM_K4 + CM_FIX4);
        if (cyc < best) { best = cyc; bbm = 512; bs = 1; btail = 0; }
    }
    // the 256 x 192 ping-pong kernel (bm code 768): 3/4 of the 256 x 256 tile's matrix work per K-step and whole rounds where N / 192 x M / 256 fills the chip better
    // than N / 256 does (N = 3 C); same conditions.  Constants: the 256 x 256 kernel's scaled by the MFMA count (72 of 96 per wave and K-step) and
    // the epilogue's share of the fixed part; checked against tools/gemm_bench.py --force on the d12 / d16 shapes (profiles/r04_n_v7_ab.log)
    static const bool no_v7 = getenv("SDVAR_GEMM_NO_V7") != nullptr;       // A/B runs only
    if (!no_v4 && !no_v7 && N >= 192 && N % 64 == 0 && M > 512) {      // a ragged last column tile is fine (N = 4096: 22 tiles; d16 fc1 at M = 2704: 68.7 against 78.0 us)
        const long tiles7 = (long)((M + 255) / 256) * ((N + 191) / 192), rounds = (tiles7 + 255) / 256;
        const double cyc = rounds * (nkt * (0.75 * CM_K4) + 0.85 * CM_FIX4);
        if (cyc < best) { best = cyc; bbm = 768; bs = 1; btail = 0; }
    }
    *bm_out = bbm; *split_out = bs; *tail_out = btail;
}

static thread_local int* g_defer_h = nullptr;     // set per call by gemm_bf16x3_nt; thread-local: host threads may drive different model objects concurrently

static int g_h2_stages = -1;      // variant of the 128 x 128 kernel: 6 (default, round 3) = 4-stage ring, the two waves of a SIMD alternate (1 - 7 % faster than 4: profiles/r03_l_v2pp_ab.log;
                                   // the same kernel on 16x16x32 MFMAs was built, passed the suite and was no faster - this tile is DMA / CU-intake bound, not power bound: profiles/r03_r_v6_ab.log);
                                   // 4 (round-2 default) = 3-stage ring with software-pipelined fragment reads (3-10 % faster than 3 on the shapes
                                   // that use this tile); 3 = 3-stage ring, reads in front of the MFMAs; 2 = 2-stage ring, two workgroups per CU (SDVAR_GEMM_H2_STAGES for A/B runs)

template <int EPI>
static int launch_h2_kernel(const GemmHArgs& a, int grid, hipStream_t stream) {
    if (g_h2_stages < 0) { const char* e = getenv("SDVAR_GEMM_H2_STAGES"); g_h2_stages = (e && atoi(e) == 2) ? 2 : (e && atoi(e) == 3) ? 3 : (e && atoi(e) == 5) ? 5 : (e && atoi(e) == 4) ? 4 : 6; }
    if (g_h2_stages == 5) {        // 5-stage ring (160 KB): four K-steps in flight
        const size_t lds = 5 * (size_t)H2_STAGE * sizeof(uint16_t);
        static LdsOptIn opt_in5;
        SDVAR_LDS_OPT_IN(opt_in5, lds, (const void*)gemm_f16x2_v2_kernel<EPI, 5>);
        hipLaunchKernelGGL((gemm_f16x2_v2_kernel<EPI, 5>), dim3(grid), dim3(512), lds, stream, a);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    if (g_h2_stages == 6) {        // 4 stages (128 KB), ping-pong halves
        const size_t lds = 4 * (size_t)H2_STAGE * sizeof(uint16_t);
        static LdsOptIn opt_in6;
        SDVAR_LDS_OPT_IN(opt_in6, lds, (const void*)gemm_f16x2_v2_kernel<EPI, 6>);
        hipLaunchKernelGGL((gemm_f16x2_v2_kernel<EPI, 6>), dim3(grid), dim3(512), lds, stream, a);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    if (g_h2_stages == 4) {        // 3 stages, software-pipelined fragment reads
        const size_t lds = 3 * (size_t)H2_STAGE * sizeof(uint16_t);
        static LdsOptIn opt_in4;
        SDVAR_LDS_OPT_IN(opt_in4, lds, (const void*)gemm_f16x2_v2_kernel<EPI, 4>);
        hipLaunchKernelGGL((gemm_f16x2_v2_kernel<EPI, 4>), dim3(grid), dim3(512), lds, stream, a);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    if (g_h2_stages == 2) {
        const size_t lds = 2 * (size_t)H2_STAGE * sizeof(uint16_t);      // 64 KB
        hipLaunchKernelGGL((gemm_f16x2_v2_kernel<EPI, 2>), dim3(grid), dim3(512), lds, stream, a);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    const size_t lds = 3 * (size_t)H2_STAGE * sizeof(uint16_t);      // 96 KB
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_v2_kernel<EPI, 3>);
    hipLaunchKernelGGL((gemm_f16x2_v2_kernel<EPI, 3>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

template <int EPI>
static int launch_h3_kernel(const GemmHArgs& a, int grid, hipStream_t stream) {
    const size_t lds = 3 * (size_t)H3_STAGE * sizeof(uint16_t);      // 144 KB
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_v3_kernel<EPI>);
    hipLaunchKernelGGL((gemm_f16x2_v3_kernel<EPI>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

static int g_h4_var = -1;          // which 256 x 256 kernel (SDVAR_GEMM_H4_VAR, A/B runs)
template <int EPI>
static int launch_h4_kernel(const GemmHArgs& a, int grid, hipStream_t stream) {
    const size_t lds = 2 * (size_t)H4_STAGE * sizeof(uint16_t);      // 128 KB
    if (g_h4_var < 0) { const char* e = getenv("SDVAR_GEMM_H4_VAR"); g_h4_var = e ? atoi(e) : 3; if (g_h4_var < 0 || g_h4_var > 3) g_h4_var = 3; }        // 3 (default) = the 16x16x32 kernel (gemm_f16x2_v5_kernel): 8 - 10 % faster than 0 on
                                                                                                                             // qkv / fc1 at M >= 2704 (profiles/r03_p_v5_ab.log); 0 - 2 = the 32x32x16 kernel and its DMA placements
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_v4_kernel<EPI, 0>, (const void*)gemm_f16x2_v4_kernel<EPI, 1>, (const void*)gemm_f16x2_v4_kernel<EPI, 2>, (const void*)gemm_f16x2_v5_kernel<EPI>);
    if (g_h4_var == 3) hipLaunchKernelGGL((gemm_f16x2_v5_kernel<EPI>), dim3(grid), dim3(512), lds, stream, a);
    else if (g_h4_var == 1) hipLaunchKernelGGL((gemm_f16x2_v4_kernel<EPI, 1>), dim3(grid), dim3(512), lds, stream, a);
    else if (g_h4_var == 2) hipLaunchKernelGGL((gemm_f16x2_v4_kernel<EPI, 2>), dim3(grid), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((gemm_f16x2_v4_kernel<EPI, 0>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}
void debug_set_h4_var(int v) { g_h4_var = v; }
void debug_set_h2_stages(int v) { g_h2_stages = v; }
void debug_set_small_pp(int v) { g_small_pp = v; }

static int launch_reduce_h(const GemmHArgs& a, const float* ws, int split, int epi, hipStream_t stream) {
    const size_t total = (size_t)a.M * (a.N / 4);
    const int rgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    dim3 block(256);
    switch (epi) {
        case HEPI_BIAS: hipLaunchKernelGGL(splitk_reduce_h_kernel<HEPI_BIAS>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        case HEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL(splitk_reduce_h_kernel<HEPI_BIAS_GELU_PLANES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        default: hipLaunchKernelGGL(splitk_reduce_h_kernel<HEPI_GATED_RES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

static int launch_h3(GemmHArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + HBN - 1) / HBN);
    const int nkt = a.K / HBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmHArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        int rc = launch_h3_kernel<HEPI_PARTIAL>(p, tiles * split, stream);
        if (rc) return rc;
        if (g_defer_h) { *g_defer_h = split; return SDVAR_OK; }
        return launch_reduce_h(a, ws, split, epi, stream);
    }
    a.split = 1; a.k_per_split = nkt;
    switch (epi) {
        case HEPI_BIAS: return launch_h3_kernel<HEPI_BIAS>(a, tiles, stream);
        case HEPI_BIAS_GELU_PLANES: return launch_h3_kernel<HEPI_BIAS_GELU_PLANES>(a, tiles, stream);
        default: return launch_h3_kernel<HEPI_GATED_RES>(a, tiles, stream);
    }
}

static int launch_h4(GemmHArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const int nkt = a.K / HBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmHArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        int rc = launch_h4_kernel<HEPI_PARTIAL>(p, tiles * split, stream);
        if (rc) return rc;
        if (g_defer_h) { *g_defer_h = split; return SDVAR_OK; }
        return launch_reduce_h(a, ws, split, epi, stream);
    }
    a.split = 1; a.k_per_split = nkt;
    switch (epi) {
        case HEPI_BIAS: return launch_h4_kernel<HEPI_BIAS>(a, tiles, stream);
        case HEPI_BIAS_GELU_PLANES: return launch_h4_kernel<HEPI_BIAS_GELU_PLANES>(a, tiles, stream);
        default: return launch_h4_kernel<HEPI_GATED_RES>(a, tiles, stream);
    }
}

template <int EPI>
static int launch_h7_kernel(const GemmHArgs& a, int grid, hipStream_t stream) {
    const size_t lds = (2 * (size_t)H7_XSTAGE + 3 * (size_t)H7_WSTAGE) * sizeof(uint16_t);      // 136 KB
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_v7_kernel<EPI>);
    hipLaunchKernelGGL((gemm_f16x2_v7_kernel<EPI>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}
static int launch_h7(GemmHArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + 191) / 192);
    const int nkt = a.K / HBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmHArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        int rc = launch_h7_kernel<HEPI_PARTIAL>(p, tiles * split, stream);
        if (rc) return rc;
        if (g_defer_h) { *g_defer_h = split; return SDVAR_OK; }
        return launch_reduce_h(a, ws, split, epi, stream);
    }
    a.split = 1; a.k_per_split = nkt;
    switch (epi) {
        case HEPI_BIAS: return launch_h7_kernel<HEPI_BIAS>(a, tiles, stream);
        case HEPI_BIAS_GELU_PLANES: return launch_h7_kernel<HEPI_BIAS_GELU_PLANES>(a, tiles, stream);
        default: return launch_h7_kernel<HEPI_GATED_RES>(a, tiles, stream);
    }
}

template <int MT, int NW>
static int launch_skinny_mt(GemmHArgs a, int epi, int split, hipStream_t stream) {
    const int tiles_n = (a.N + 15) / 16, nkt = a.K / HBK;
    const size_t lds = (size_t)NW * MT * 64 * 16;
    GemmHArgs p = a;
    p.split = split; p.k_per_split = (nkt + split - 1) / split;
    const dim3 grid(tiles_n * split), block(64 * NW);
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        p.out = ws; p.ldo = a.N;
        hipLaunchKernelGGL((gemm_f16x2_skinny_kernel<MT, NW, HEPI_PARTIAL>), grid, block, lds, stream, p);
        SDVAR_LAUNCH_CHECK();
        if (g_defer_h) { *g_defer_h = split; return SDVAR_OK; }
        return launch_reduce_h(a, ws, split, epi, stream);
    }
    switch (epi) {
        case HEPI_BIAS: hipLaunchKernelGGL((gemm_f16x2_skinny_kernel<MT, NW, HEPI_BIAS>), grid, block, lds, stream, p); break;
        case HEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL((gemm_f16x2_skinny_kernel<MT, NW, HEPI_BIAS_GELU_PLANES>), grid, block, lds, stream, p); break;
        default: hipLaunchKernelGGL((gemm_f16x2_skinny_kernel<MT, NW, HEPI_GATED_RES>), grid, block, lds, stream, p); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}
// M <= 80 rows: NW waves per workgroup and the K split so that a wave streams at most 8 K-steps
template <int NW>
static int launch_skinny(const GemmHArgs& a, int epi, int split, hipStream_t stream) {
    const int mt = (a.M + 15) / 16;
    switch (mt) {
        case 1: return launch_skinny_mt<1, NW>(a, epi, split, stream);
        case 2: return launch_skinny_mt<2, NW>(a, epi, split, stream);
        case 3: return launch_skinny_mt<3, NW>(a, epi, split, stream);
        case 4: return launch_skinny_mt<4, NW>(a, epi, split, stream);
        default: return launch_skinny_mt<5, NW>(a, epi, split, stream);
    }
}

// full rounds unsplit + the partial last round split `tail` ways along K (compact slabs) + a reduce over the tail tiles only
static int launch_h3_hybrid(GemmHArgs a, int epi, int tail, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + HBN - 1) / HBN), full = tiles / 256 * 256, remt = tiles - full;
    const int nkt = a.K / HBK;
    size_t wsf = 0;
    float* ws = splitk_workspace(&wsf);
    if (!ws) return SDVAR_ERR_HIP;
    GemmHArgs f = a;
    f.split = 1; f.k_per_split = nkt; f.tile_off = 0; f.tile_cnt = full;
    int rc;
    switch (epi) {
        case HEPI_BIAS: rc = launch_h3_kernel<HEPI_BIAS>(f, full, stream); break;
        case HEPI_BIAS_GELU_PLANES: rc = launch_h3_kernel<HEPI_BIAS_GELU_PLANES>(f, full, stream); break;
        default: rc = launch_h3_kernel<HEPI_GATED_RES>(f, full, stream); break;
    }
    if (rc) return rc;
    GemmHArgs p = a;
    p.out = ws; p.split = tail; p.k_per_split = (nkt + tail - 1) / tail; p.tile_off = full; p.tile_cnt = remt;
    rc = launch_h3_kernel<HEPI_PARTIAL>(p, remt * tail, stream);
    if (rc) return rc;
    GemmHArgs r = a;
    r.tile_off = full; r.tile_cnt = remt;
    switch (epi) {
        case HEPI_BIAS: hipLaunchKernelGGL(splitk_reduce_tiles_h_kernel<HEPI_BIAS>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
        case HEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL(splitk_reduce_tiles_h_kernel<HEPI_BIAS_GELU_PLANES>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
        default: hipLaunchKernelGGL(splitk_reduce_tiles_h_kernel<HEPI_GATED_RES>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

template <int BM, int NS, int EPI>
static int launch_small_kernel(const GemmHArgs& a, int grid, hipStream_t stream) {
    const size_t lds = (size_t)NS * 2 * (BM + 128) * 32 * sizeof(uint16_t);
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_small_kernel<BM, NS, EPI>);
    hipLaunchKernelGGL((gemm_f16x2_small_kernel<BM, NS, EPI>), dim3(grid), dim3(256), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}
template <int BM, int EPI>
static int launch_small_pp(const GemmHArgs& a, int grid, hipStream_t stream) {
    const size_t lds = (size_t)6 * spp_stage<BM>() * sizeof(uint16_t);          // 144 KB (BM = 64) / 120 KB (BM = 32): one workgroup per CU
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f16x2_small_pp_kernel<BM, EPI>);
    hipLaunchKernelGGL((gemm_f16x2_small_pp_kernel<BM, EPI>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}
template <int BM, int EPI>
static int launch_small_any(const GemmHArgs& a, int grid, hipStream_t stream) {
    if (!a.stamps && (BM == 64 ? small_pp_on() : small_pp_on() >= 2)) return launch_small_pp<BM, EPI>(a, grid, stream);          // gemm_small_pp: 1 = 64-row tiles only, 2 = 32-row tiles too
    return launch_small_kernel<BM, 3, EPI>(a, grid, stream);
}

template <int BM>
static int launch_h(GemmHArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + HBN - 1) / HBN);
    constexpr bool v2 = (BM == 128);
    constexpr int SB = BM == 32 ? 32 : 64;
    const int nkt = a.K / HBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmHArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        int rc = v2 ? launch_h2_kernel<HEPI_PARTIAL>(p, tiles * split, stream) : launch_small_any<SB, HEPI_PARTIAL>(p, tiles * split, stream);
        if (rc) return rc;
        if (g_defer_h) { *g_defer_h = split; return SDVAR_OK; }
        if (SDVAR_DBG(a, 8)) return SDVAR_OK;      // timing experiments: the slab launch alone (what a deferring caller pays)
        return launch_reduce_h(a, ws, split, epi, stream);
    }
    a.split = 1; a.k_per_split = nkt;
    switch (epi) {
        case HEPI_BIAS: return v2 ? launch_h2_kernel<HEPI_BIAS>(a, tiles, stream) : launch_small_any<SB, HEPI_BIAS>(a, tiles, stream);
        case HEPI_BIAS_GELU_PLANES: return v2 ? launch_h2_kernel<HEPI_BIAS_GELU_PLANES>(a, tiles, stream) : launch_small_any<SB, HEPI_BIAS_GELU_PLANES>(a, tiles, stream);
        default: return v2 ? launch_h2_kernel<HEPI_GATED_RES>(a, tiles, stream) : launch_small_any<SB, HEPI_GATED_RES>(a, tiles, stream);
    }
}

// ---- row-block launches (gemm_f16x2_rowblk_kernel): M <= 80 ---------------------------------------------------------------------------
static thread_local bool g_rowblk_floor = false;   // set around the model's query (gemm_f16x2_rowblk_want)
static int g_rowblk = -1;          // SDVAR_ROWBLK (A/B runs): 0 = the round-3 launch sequence (ln_modulate + skinny / small_pp + qk_norm_append) at every M
void debug_set_rowblk(int v) { g_rowblk = v; }
// can the (LayerNorm +) GEMM (+ QKV finish) of this shape run as one row-block launch?  ln: the operand is the fp32 residual stream (K = C)
bool gemm_f16x2_rowblk_ok(int M, int N, int K, int ln, int qkv) {
    if (g_rowblk < 0) { const char* e = getenv("SDVAR_ROWBLK"); g_rowblk = e ? atoi(e) : 1; }
    if (!g_rowblk || g_force_bm_h) return false;                 // a forced tile (tests of the other kernels) keeps the old sequence
    if (M < 1 || M > 80 || K % HBK) return false;
    // the model's own calls (not the op-level tests) come with a floor: below 32 rows the old sequence wins.  A 16-row call is ONE row block: its QKV launch has
    // H x 3 workgroups pulling 256 KB of W each, and with one image per row every workgroup also reads 16 x 8 KB of modulation vectors - a CU takes in only
    // ~45 GB/s of L2 hits / ~20 GB/s of HBM misses, so bytes per workgroup decide (d16 stage 0: 0.84 ms fused against 0.78 ms; stage 1, M = 64: 0.84 against 0.96)
    if (g_rowblk_floor && M < 32) return false;
    if (ln ? K > 1024 : K > 4096) return false;
    return qkv ? N % 64 == 0 : N % 16 == 0;
}

// stage_forward's question: should a call with M rows take the row-block sequence?  (2 forces it at every M <= 80: A/B runs and the tests of the 16-row shapes)
bool gemm_f16x2_rowblk_want(int M, int C, int V, int rows_per_img) {
    g_rowblk_floor = true;
    if (g_rowblk < 0) { const char* e = getenv("SDVAR_ROWBLK"); g_rowblk = e ? atoi(e) : 1; }
    if (g_rowblk == 2) g_rowblk_floor = false;
    // ... and a width floor: at C = 768 (d12) the QKV launch has only 36 x ceil(M / 16) workgroups and fc1 falls between one and two rounds - stage 1 of d12 ran
    // 0.68 ms fused against 0.62 ms (profiles/r04_g_stage_d12.log); at C = 1024 (d16) 0.86 against 0.96
    // ... and at least 4 rows per image (the measured cases: l = 4, 5): with one image per row (stage 0 of a large batch) a row block reads 16 x 8 KB of modulation vectors
    const bool ok = (!g_rowblk_floor || (C >= 1024 && rows_per_img >= 4)) && gemm_f16x2_rowblk_ok(M, 3 * C, C, 1, 1) && gemm_f16x2_rowblk_ok(M, C, 4 * C, 0, 0) && gemm_f16x2_rowblk_ok(M, V, C, 1, 0);
    g_rowblk_floor = false;
    return ok;
}

template <int NT, int KS, int EPI, int LN>
static int launch_rowblk_kernel(const RowBlkArgs& ra, hipStream_t stream) {
    const dim3 grid(ra.g.N / (16 * NT), (ra.g.M + 15) / 16);
    hipLaunchKernelGGL((gemm_f16x2_rowblk_kernel<NT, KS, EPI, LN>), grid, dim3(512), 0, stream, ra);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// x != null: out = epi( (LN(x) (1 + scale) + shift) W^T + bias ), epi = HEPI_BIAS | HEPI_BIAS_GELU_PLANES, or the QKV finish when q_out != null (then N = 3 H 64 and
// nothing is written to `out`); x == null: X planes, epi = HEPI_GATED_RES | HEPI_BIAS.  gemm_f16x2_rowblk_ok(M, N, K, x != null, q_out != null) must hold.
int gemm_f16x2_rowblk(const float* x, int ldx, const float* scale, const float* shift, int rows_per_img, int mod_stride, const uint16_t* X, size_t xps,
                      const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops, int M, int N, int K, int epi,
                      const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride,
                      const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int l, int H, int Lp, int pos0, int kv_fmt, hipStream_t stream) {
    const bool ln = x != nullptr, qkv = q_out != nullptr;
    SDVAR_CHECK_ARG(W && (ln || X) && gemm_f16x2_rowblk_ok(M, N, K, ln, qkv), "gemm_f16x2_rowblk: shape M=%d N=%d K=%d (ln %d, qkv %d) not supported", M, N, K, (int)ln, (int)qkv);
    SDVAR_CHECK_ARG(!ln || (scale && shift && rows_per_img > 0 && ldx >= K && ldx % 4 == 0 && mod_stride % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0),
                    "gemm_f16x2_rowblk: bad LayerNorm operand");
    SDVAR_CHECK_ARG(((uintptr_t)W % 16) == 0 && wps % 8 == 0 && (ln || (((uintptr_t)X % 16) == 0 && xps % 8 == 0)), "gemm_f16x2_rowblk: planes must be 16-byte aligned");
    if (qkv) SDVAR_CHECK_ARG(ln && k_cache && v_cache && l > 0 && H > 0 && N == 3 * H * 64 && M % l == 0 && (kv_fmt == 3 || kv_fmt == 4) && Lp % 8 == 0 && ((uintptr_t)q_out % 16) == 0 &&
                             ((uintptr_t)k_cache % 16) == 0 && ((uintptr_t)v_cache % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0), "gemm_f16x2_rowblk: bad QKV finish arguments");
    else SDVAR_CHECK_ARG(epi == HEPI_BIAS_GELU_PLANES ? (outp != nullptr && ln) : (out != nullptr && ldo >= N && (epi == HEPI_BIAS || (epi == HEPI_GATED_RES && !ln && res && gate && rows_per_gate > 0 && ldres >= N))),
                         "gemm_f16x2_rowblk: epilogue %d not available here", epi);
    RowBlkArgs ra;
    ra.g = GemmHArgs{X, W, xps, wps, wsi, bias, out, outp, ops, res, gate, M, N, K, ldo, ldres, rows_per_gate > 0 ? rows_per_gate : 1, gate_stride, 1, K / HBK, 0, 0, nullptr,
                     QkvEpi{scale_mul, q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, l, H, Lp, pos0, kv_fmt}, 0, 0};
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    ra.g.vec = N % 4 == 0 && al16(bias) && al16(out) && al16(outp) && al16(res) && al16(gate) && ldo % 4 == 0 && ops % 4 == 0 && (epi != HEPI_GATED_RES || (ldres % 4 == 0 && gate_stride % 4 == 0));
    ra.x = x; ra.ldx = ldx; ra.scale = scale; ra.shift = shift; ra.rows_per_img = rows_per_img > 0 ? rows_per_img : 1; ra.mod_stride = mod_stride; ra.eps = 1e-6f;
    static const bool trace = getenv("SDVAR_GEMM_TRACE") != nullptr;
    const int rb = (M + 15) / 16;
    // column tiles per workgroup: bytes a workgroup pulls through its CU (W 16 nt columns + its 16 rows of x + their modulation vectors, all K long) x rounds of 256
    // workgroups - a CU takes in ~45 GB/s of L2 hits, so this product is what a launch costs (the fixed part is the same for every nt)
    int nt = 1;
    {
        const int imgs = (16 + rows_per_img - 1) / (rows_per_img > 0 ? rows_per_img : 1) + 1;       // images a 16-row block touches (upper bound)
        double best = 1e30;
        for (int c = 1; c <= 4; c *= 2) {
            if (N % (16 * c)) break;
            const long blocks = (long)(N / (16 * c)) * rb;
            const double cost = (double)((blocks + 255) / 256) * (16.0 * c + 16.0 + 2.0 * (imgs < 16 ? imgs : 16));
            if (cost < best) { best = cost; nt = c; }
        }
    }
    if (!ln) nt = 1;
    if (trace) fprintf(stderr, "[gemm_f16x2] M=%d N=%d K=%d epi=%d -> rowblk ln=%d qkv=%d nt=%d\n", M, N, K, epi, (int)ln, (int)qkv, qkv ? 4 : nt);
    g_last_cfg_h[0] = 17; g_last_cfg_h[1] = 1;
    if (qkv) return launch_rowblk_kernel<4, 4, HEPI_QKV, 1>(ra, stream);
    if (ln) {
        if (epi == HEPI_BIAS_GELU_PLANES) return nt == 1 ? launch_rowblk_kernel<1, 4, HEPI_BIAS_GELU_PLANES, 1>(ra, stream) : nt == 2 ? launch_rowblk_kernel<2, 4, HEPI_BIAS_GELU_PLANES, 1>(ra, stream)
                                                                                                                                      : launch_rowblk_kernel<4, 4, HEPI_BIAS_GELU_PLANES, 1>(ra, stream);
        return nt == 1 ? launch_rowblk_kernel<1, 4, HEPI_BIAS, 1>(ra, stream) : nt == 2 ? launch_rowblk_kernel<2, 4, HEPI_BIAS, 1>(ra, stream) : launch_rowblk_kernel<4, 4, HEPI_BIAS, 1>(ra, stream);
    }
    // plane operand, K up to 4096 unsplit: 16 K-steps of W per wave in flight (128 VGPRs) leave room for one column tile only
    if (epi == HEPI_GATED_RES) return launch_rowblk_kernel<1, 16, HEPI_GATED_RES, 0>(ra, stream);
    return launch_rowblk_kernel<1, 16, HEPI_BIAS, 0>(ra, stream);
}

// X planes [2][K/32][M][32] (plane stride xps), W planes [2][K/32][N][32] of W * 2^S (plane stride wps), wsi -> 2^-S on the device (null: 1).
// epi: 0 bias -> out fp32; 1 bias + GELU -> outp planes of (M, N) (plane stride ops); 2 gated residual -> out fp32.
// defer: as gemm_bf16x3_nt (the slabs already carry the 2^-S factor).
int gemm_f16x2_nt(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops,
                  int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, int* defer,
                  hipStream_t stream);
static bool g_qkv_fuse_off = false;
void debug_set_qkv_fuse(int on) { g_qkv_fuse_off = !on; }
static thread_local const QkvEpi* g_qkv_epi = nullptr;      // set by gemm_f16x2_qkv around its call of gemm_f16x2_nt
static thread_local int* g_qkv_fused = nullptr;

// The QKV launch of a transformer block: as gemm_f16x2_nt(epi 0, out = the (M, 3 H 64) fp32 qkv buffer, defer), but when the launch comes out UNSPLIT the
// q, k and v are finished in the epilogue (QkvEpi) and *fused = 1: the caller runs no qk_norm_append at all.  Otherwise *fused = 0 and the
// result is in `out` (or in the slabs, *defer > 0) as before.
int gemm_f16x2_qkv(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, int M, int N, int K,
                   const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int l, int H, int Lp, int pos0, int kv_fmt, int* defer, int* fused, hipStream_t stream) {
    SDVAR_CHECK_ARG(fused && defer && q_out && k_cache && v_cache && l > 0 && H > 0 && N == 3 * H * 64 && M % l == 0, "gemm_f16x2_qkv: bad arguments (M=%d N=%d l=%d H=%d)", M, N, l, H);
    *fused = 0;
    static const bool env_off = getenv("SDVAR_NO_QKV_FUSE") != nullptr;      // A/B runs
    const bool off = env_off || g_qkv_fuse_off;
    const QkvEpi e{scale_mul, q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, l, H, Lp, pos0, kv_fmt};
    const bool ok = !off && (kv_fmt == 3 || kv_fmt == 4) && (H * 64) % 128 == 0 && ((uintptr_t)q_out % 16) == 0 && ((uintptr_t)k_cache % 16) == 0 && ((uintptr_t)v_cache % 16) == 0 && Lp % 8 == 0;
    g_qkv_epi = ok ? &e : nullptr; g_qkv_fused = fused;
    const int rc = gemm_f16x2_nt(X, xps, W, wps, wsi, bias, out, ldo, nullptr, 0, M, N, K, HEPI_BIAS, nullptr, 0, nullptr, 1, 0, defer, stream);
    g_qkv_epi = nullptr; g_qkv_fused = nullptr;
    return rc;
}

int gemm_f16x2_nt(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops,
                  int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, int* defer,
                  hipStream_t stream) {
    g_defer_h = defer;
    if (defer) *defer = 0;
    SDVAR_CHECK_ARG(X && W, "gemm_f16x2: null operand");
    SDVAR_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % HBK == 0, "gemm_f16x2: need K %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    SDVAR_CHECK_ARG(epi >= HEPI_BIAS && epi <= HEPI_GATED_RES, "gemm_f16x2: unknown epilogue %d", epi);
    SDVAR_CHECK_ARG(epi == HEPI_BIAS_GELU_PLANES ? (outp != nullptr && N % 4 == 0) : (out != nullptr && ldo >= N), "gemm_f16x2: missing output");
    SDVAR_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && xps % 8 == 0 && wps % 8 == 0, "gemm_f16x2: planes must be 16-byte aligned");
    if (epi == HEPI_GATED_RES) SDVAR_CHECK_ARG(res && gate && rows_per_gate > 0 && ldres >= N, "gemm_f16x2: gated-residual epilogue needs res/gate");
#ifdef SDVAR_TIMING_EXPERIMENTS
    static const int dbg = getenv("SDVAR_GEMM_DBG") ? atoi(getenv("SDVAR_GEMM_DBG")) : 0;
    static const bool warned = (dbg && fprintf(stderr, "[sdvar] SDVAR_GEMM_DBG=%d: timing experiment build, GEMM results are WRONG\n", dbg), true);
    (void)warned;
#else
    const int dbg = 0;
#endif
    GemmHArgs a{X, W, xps, wps, wsi, bias, out, outp, ops, res, gate, M, N, K, ldo, ldres, rows_per_gate > 0 ? rows_per_gate : 1, gate_stride, 1, K / HBK, 0, dbg, debug_get_gemm_stamps(), QkvEpi{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}, 0, 0};
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    a.vec = N % 4 == 0 && al16(bias) && al16(out) && al16(outp) && al16(res) && al16(gate) && ldo % 4 == 0 && ops % 4 == 0 &&
            (epi != HEPI_GATED_RES || (ldres % 4 == 0 && gate_stride % 4 == 0));
    size_t wsf = 0;
    (void)splitk_workspace(&wsf);
    int bm, split, tail = 0;
    static const bool no_hybrid = getenv("SDVAR_GEMM_NO_HYBRID") != nullptr;       // A/B runs only
    // a QKV launch that can finish q and k in its epilogue stays off the hybrid tail split (whose tail tiles go through slabs): the fused epilogue saves more
    const bool qkv = g_qkv_epi != nullptr && a.vec;
    choose_cfg_h(M, N, K, wsf, &bm, &split, &tail, !no_hybrid && !qkv, defer != nullptr);
    if (g_force_bm_h) { bm = g_force_bm_h; tail = 0; }
    if (g_force_bm_h == 256 && g_force_split_h < 0) {        // test aid: force the hybrid tail split -split ways (where the shape has a partial last round)
        const int tiles = ((M + 255) / 256) * ((N + HBN - 1) / HBN), remt = tiles % 256, nkt = K / HBK;
        int ts = -g_force_split_h;
        if (ts > nkt / 2) ts = nkt / 2;
        const int kps = ts > 0 ? (nkt + ts - 1) / ts : nkt;
        if (tiles > 256 && remt && !qkv && N % 4 == 0 && ts >= 2 && (nkt + kps - 1) / kps == ts && (size_t)ts * remt * (256 * 128) <= wsf) { tail = ts; split = 1; }
    }
    static const bool trace = getenv("SDVAR_GEMM_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[gemm_f16x2] M=%d N=%d K=%d epi=%d -> bm=%d split=%d tail=%d\n", M, N, K, epi, bm, split, tail);
    if (g_force_split_h > 0) {
        split = g_force_split_h;
        const int nkt = K / HBK;
        if (split > nkt) split = nkt;
        while (split > 1 && (size_t)split * M * N > wsf) --split;
        const int kps = (nkt + split - 1) / split;
        split = (nkt + kps - 1) / kps;
    }
    g_last_cfg_h[0] = bm; g_last_cfg_h[1] = split; g_last_cfg_h[2] += (bm == 256 && tail > 0) ? 1 : 0; g_last_cfg_h[3] += (qkv && split == 1 && tail == 0) ? 1 : 0;
    // bm code 16: the skinny kernel (M <= 80 rows, N % 16 == 0): chosen for every such shape unless a tile is forced; its own K split (a wave streams <= 8 K-steps)
    static const int skinny_max = getenv("SDVAR_GEMM_SKINNY_MAX") ? atoi(getenv("SDVAR_GEMM_SKINNY_MAX")) : 80;          // A/B runs: 0 switches it off
    if ((g_force_bm_h == 16 || (!g_force_bm_h && M <= skinny_max)) && M <= 80 && N % 16 == 0) {
        const int nkt = K / HBK;
        int sp = (nkt + 31) / 32;                            // 4 waves x 8 K-steps per workgroup (a 16-wave workgroup for K = 4096 is capped at 128 VGPRs and spills)
        if (g_force_bm_h == 16 && g_force_split_h > sp) sp = g_force_split_h;
        while (sp > 1 && (size_t)sp * M * N > wsf) --sp;
        const int kps = (nkt + sp - 1) / sp;
        // automatic choice: only where ONE workgroup streams the whole K (K <= 1024): with a split over workgroups (fc2, K = 4096) the slab path of the ring kernels is
        // as fast or faster (M = 64: 13.5 against 14.9 us, profiles/r03_s_skinny_ab.log); a forced tile (tests) takes any K
        if (kps <= 32 && (sp == 1 || (g_force_bm_h == 16 && N % 4 == 0))) {
            sp = (nkt + kps - 1) / kps;
            g_last_cfg_h[0] = 16; g_last_cfg_h[1] = sp;
            if (trace) fprintf(stderr, "[gemm_f16x2] M=%d N=%d K=%d epi=%d -> skinny split=%d\n", M, N, K, epi, sp);
            return launch_skinny<4>(a, epi, sp, stream);
        }
    }
    if (bm == 16) bm = 32;
    if (qkv && split == 1 && tail == 0) {
        a.qk = *g_qkv_epi; a.split = 1; a.k_per_split = K / HBK;
        *g_qkv_fused = 1;
        if (bm == 768) return launch_h7_kernel<HEPI_QKV>(a, ((M + 255) / 256) * ((N + 191) / 192), stream);
        if (bm == 512) return launch_h4_kernel<HEPI_QKV>(a, ((M + 255) / 256) * ((N + 255) / 256), stream);
        if (bm == 256) return launch_h3_kernel<HEPI_QKV>(a, ((M + 255) / 256) * ((N + HBN - 1) / HBN), stream);
        if (bm == 128) return launch_h2_kernel<HEPI_QKV>(a, ((M + 127) / 128) * ((N + HBN - 1) / HBN), stream);
        if (bm == 64) return launch_small_any<64, HEPI_QKV>(a, ((M + 63) / 64) * ((N + HBN - 1) / HBN), stream);
        return launch_small_any<32, HEPI_QKV>(a, ((M + 31) / 32) * ((N + HBN - 1) / HBN), stream);
    }
    if (bm == 768) return launch_h7(a, epi, split, stream);
    if (bm == 512) return launch_h4(a, epi, split, stream);
    if (bm == 256 && tail > 0) return launch_h3_hybrid(a, epi, tail, stream);
    if (bm == 256) return launch_h3(a, epi, split, stream);
    if (bm == 32) return launch_h<32>(a, epi, split, stream);
    if (bm == 64) return launch_h<64>(a, epi, split, stream);
    return launch_h<128>(a, epi, split, stream);
}

}  // namespace sdvar
