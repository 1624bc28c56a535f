// Verify-attention: softmax(q k^T + block-causal mask) v over the growing KV cache, fp32, flash-style (no score matrix
// in HBM).  Replaces F.scaled_dot_product_attention(q, cat(cached_k, k), cat(cached_v, v), scale=1, attn_mask) at
// /root/reference/models/basic_var.py:107-117 with the mask rows of models/var.py:108-113 derived in-kernel from the
// stage boundaries (never materialised).
//
// Layouts:  q (R, H, l, 64) (already L2-normalised and scaled), kc/vc (R, H, Lmax, 64) fp32 or fp16 (BASELINE config P4:
//           the cache is stored in half precision and widened while staging; all arithmetic stays fp32) with `Ktot` valid keys
//           (prefix + the l keys of this call), out (R, l, H*64) row-major for the projection GEMM.
// Queries of chunk stage j (q index in [qbeg[j], qbeg[j+1])) see keys [0, vis[j]).
//
// Mapping (wave64):  one wave owns 32 queries; workgroup = 4 waves = 128 queries of one (row, head).
//   S^T = K Q^T  with v_mfma_f32_32x32x2_f32: A = K tile (key on the MFMA row), B = Q (query on the lane) so every lane
//   holds 16 scores of ONE query per 32-key sub-tile -> row max/sum are in-register plus one cross-half shuffle.
//   O^T = V^T P^T: A = V^T read from LDS (d on the MFMA row), B = P straight from the score registers (no LDS trip).
//   K/V tiles of 64 keys are streamed HBM -> registers -> LDS (double buffered, coalesced 16-byte loads of contiguous
//   cache rows), K rows padded to 68 floats for conflict-free ds_read_b128.
// Algorithmic bytes per launch: R*H*64*4 * (2*Ktot + 2*l) (K and V read once, Q read, O written).
#include <hip/hip_fp16.h>

#include "common.h"

namespace sdvar {

constexpr int ATT_MAX_CHUNK = 16;
constexpr int KT = 64;              // keys per LDS tile
constexpr int KSTR = 68;            // padded K row (floats)

struct AttnArgs {
    const float* q; const void* kc; const void* vc; float* out;     // kc/vc: fp32 or fp16 (template KVH)
    uint16_t* outp; size_t ops; int pfmt;                           // optional operand planes output instead of `out` (common.h PLANES_*)
    int R, H, l, Lmax, Ktot;
    int n_chunk;
    int qbeg[ATT_MAX_CHUNK + 1];
    int vis[ATT_MAX_CHUNK];
};

template <bool KVH>
__global__ __launch_bounds__(256, 2) void attention_f32_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = KT * KSTR + KT * 64;              // floats per pipeline stage: K tile then V tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int q0 = qt * 128;

    // this lane's query and how many keys it may see
    const int qi_raw = q0 + wave * 32 + li;
    const int qi = min(qi_raw, a.l - 1);
    int vis_q = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (qi >= a.qbeg[j]) vis_q = a.vis[j];
    // keys needed by the workgroup: the last valid query of the block sees the most
    const int q_last = min(q0 + 127, a.l - 1);
    int kend = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (q_last >= a.qbeg[j]) kend = a.vis[j];
    const bool wave_active = (q0 + wave * 32) < a.l;

    // Q fragment: lane (query li, half lh) holds d = 8c + 4lh + e, c = 0..7, e = 0..3
    f32x4 qf[8];
    {
        const float* pq = a.q + (((size_t)r * a.H + h) * a.l + qi) * 64 + 4 * lh;
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4*>(pq + 8 * c);
    }

    // staging: 64 keys x 64 channels per operand.  fp32 cache: 4 float4 per thread (key = tid/16 + 16 i, col = 4 (tid%16));
    // fp16 cache: 2 chunks of 8 halves per thread (key = tid/8 + 32 i, col = 8 (tid%8)), widened to fp32 on the way to LDS.
    const size_t head_off = ((size_t)r * a.H + h) * a.Lmax * 64;
    const float* kbase = reinterpret_cast<const float*>(a.kc) + (KVH ? 0 : head_off);
    const float* vbase = reinterpret_cast<const float*>(a.vc) + (KVH ? 0 : head_off);
    const __half* kbh = reinterpret_cast<const __half*>(a.kc) + (KVH ? head_off : 0);
    const __half* vbh = reinterpret_cast<const __half*>(a.vc) + (KVH ? head_off : 0);
    const int skey = KVH ? (tid >> 3) : (tid >> 4), scol = KVH ? (tid & 7) * 8 : (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto load_tile = [&](int k0) {
        if (KVH) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = k0 + skey + 32 * i;
                if (key < a.Ktot) {
                    const f32x4 hk = *reinterpret_cast<const f32x4*>(kbh + (size_t)key * 64 + scol);   // 8 halves
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(vbh + (size_t)key * 64 + scol);
                    const __half2* pk = reinterpret_cast<const __half2*>(&hk);
                    const __half2* pv = reinterpret_cast<const __half2*>(&hv);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float2 a0 = __half22float2(pk[2 * e]), a1 = __half22float2(pk[2 * e + 1]);
                        const float2 b0 = __half22float2(pv[2 * e]), b1 = __half22float2(pv[2 * e + 1]);
                        rk[2 * i + e] = f32x4{a0.x, a0.y, a1.x, a1.y};
                        rv[2 * i + e] = f32x4{b0.x, b0.y, b1.x, b1.y};
                    }
                } else {
                    rk[2 * i] = rk[2 * i + 1] = rv[2 * i] = rv[2 * i + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = k0 + skey + 16 * i;
                if (key < a.Ktot) {
                    rk[i] = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * 64 + scol);
                    rv[i] = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * 64 + scol);
                } else {
                    rk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (KVH) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    *reinterpret_cast<f32x4*>(smem + buf * STAGE + (skey + 32 * i) * KSTR + scol + 4 * e) = rk[2 * i + e];
                    *reinterpret_cast<f32x4*>(smem + buf * STAGE + KT * KSTR + (skey + 32 * i) * 64 + scol + 4 * e) = rv[2 * i + e];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(smem + buf * STAGE + (skey + 16 * i) * KSTR + scol) = rk[i];
                *reinterpret_cast<f32x4*>(smem + buf * STAGE + KT * KSTR + (skey + 16 * i) * 64 + scol) = rv[i];
            }
        }
    };

    f32x16 o0, o1;                        // O^T accumulators: d = db*32 + (reg&3) + 8*(reg>>2) + 4*lh, column = this query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (kend + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * KT;
        if (t + 1 < ntiles) load_tile(k0 + KT);
        if (wave_active) {
            // LDS reads are hand-placed (inline asm + counted lgkmcnt): the compiler's schedule fetched every fragment right
            // before its MFMA and exposed the LDS latency ~50 times per tile (rocprof: 34 % matrix-pipe use).
            typedef __attribute__((address_space(3))) float* lds_f;
            const uint32_t kb = (uint32_t)(uintptr_t)(lds_f)(smem + buf * STAGE) + 4u * (li * KSTR + 4 * lh);
            const uint32_t vb = (uint32_t)(uintptr_t)(lds_f)(smem + buf * STAGE + KT * KSTR) + 4u * (4 * lh * 64 + li);
            // ---- K fragments of both 32-key sub-tiles (16 x b128), then the 64 score MFMAs
            f32x4 kf[2][8];
#define SDVAR_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
            SDVAR_RD128(kf[0][0], kb, 0);    SDVAR_RD128(kf[0][1], kb, 32);   SDVAR_RD128(kf[0][2], kb, 64);   SDVAR_RD128(kf[0][3], kb, 96);
            SDVAR_RD128(kf[0][4], kb, 128);  SDVAR_RD128(kf[0][5], kb, 160);  SDVAR_RD128(kf[0][6], kb, 192);  SDVAR_RD128(kf[0][7], kb, 224);
            SDVAR_RD128(kf[1][0], kb, 8704); SDVAR_RD128(kf[1][1], kb, 8736); SDVAR_RD128(kf[1][2], kb, 8768); SDVAR_RD128(kf[1][3], kb, 8800);
            SDVAR_RD128(kf[1][4], kb, 8832); SDVAR_RD128(kf[1][5], kb, 8864); SDVAR_RD128(kf[1][6], kb, 8896); SDVAR_RD128(kf[1][7], kb, 8928);
#undef SDVAR_RD128
            f32x16 s[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[sub][i] = 0.f;
                if (sub == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[sub][c][e], qf[c][e], s[sub], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- V operands of the whole tile: 32 x ds_read2st64_b32 (two key rows, 256 B apart, per instruction), issued
            // before the softmax arithmetic so their latency hides under it.  vf[db][sub][i] = V[key(sub, i, lh)][32 db + li].
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 vf[2][2][8];
#define SDVAR_RDV(dst, addr, r0, r1) asm volatile("ds_read2st64_b32 %0, %1 offset0:" #r0 " offset1:" #r1 : "=v"(dst) : "v"(addr) : "memory")
#define SDVAR_RDV_SUB(db, sub, base)                                                                                      \
            SDVAR_RDV(vf[db][sub][0], vb + 128u * db, base + 0, base + 1);   SDVAR_RDV(vf[db][sub][1], vb + 128u * db, base + 2, base + 3);   \
            SDVAR_RDV(vf[db][sub][2], vb + 128u * db, base + 8, base + 9);   SDVAR_RDV(vf[db][sub][3], vb + 128u * db, base + 10, base + 11); \
            SDVAR_RDV(vf[db][sub][4], vb + 128u * db, base + 16, base + 17); SDVAR_RDV(vf[db][sub][5], vb + 128u * db, base + 18, base + 19); \
            SDVAR_RDV(vf[db][sub][6], vb + 128u * db, base + 24, base + 25); SDVAR_RDV(vf[db][sub][7], vb + 128u * db, base + 26, base + 27);
            SDVAR_RDV_SUB(0, 0, 0) SDVAR_RDV_SUB(1, 0, 0) SDVAR_RDV_SUB(0, 1, 32) SDVAR_RDV_SUB(1, 1, 32)
#undef SDVAR_RDV_SUB
#undef SDVAR_RDV
            // ---- mask + online softmax (this lane: one query, keys k0 + sub*32 + (i&3) + 8*(i>>2) + 4*lh)
            float mloc = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    if (key >= vis_q) s[sub][i] = -INFINITY;
                    mloc = fmaxf(mloc, s[sub][i]);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);          // finite from the first tile on (key 0 is always visible)
            // exp(x) = 2^(x log2 e) on v_exp_f32: the exponent argument is (s - m) <= 0, its rounding error |s - m| 2^-24
            // only matters for weights far below the row maximum
            const float L2E = 1.4426950408889634f;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L2E);   // 2^-inf = 0 on the first tile
            float lsum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[sub][i] = __builtin_amdgcn_exp2f((s[sub][i] - m_new) * L2E); lsum += s[sub][i]; }
            lsum += __shfl_xor(lsum, 32, 64);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            // ---- O^T += V^T P^T : step i pairs key (i&3)+8*(i>>2) (half 0) with the same +4 (half 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // i = 4g + e: rows (e, e+1) of group g were fetched together: vf[..][2g + (e >> 1)][e & 1]
                    const int slot = 2 * (i >> 2) + ((i & 3) >> 1), half = i & 1;
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[0][sub][slot][half], s[sub][i], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[1][sub][slot][half], s[sub][i], o1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (wave_active && qi_raw < a.l) {
        const float inv = 1.0f / l_run;
        const size_t obase = ((size_t)r * a.l + qi_raw) * (a.H * 64) + h * 64 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = o0[4 * g + e] * inv; v1[e] = o1[4 * g + e] * inv; }
            if (a.outp) {
                const float u0[4] = {v0[0], v0[1], v0[2], v0[3]}, u1[4] = {v1[0], v1[1], v1[2], v1[3]};
                const int orow = r * a.l + qi_raw, ocol = h * 64 + 4 * lh + 8 * g;      // K-blocked planes of the (R*l, H*64) matrix
                store_planes4(a.outp, a.ops, kb_index(orow, ocol, a.R * a.l), u0, a.pfmt);
                store_planes4(a.outp, a.ops, kb_index(orow, ocol + 32, a.R * a.l), u1, a.pfmt);
            } else {
                *reinterpret_cast<f32x4*>(a.out + obase + 8 * g) = v0;
                *reinterpret_cast<f32x4*>(a.out + obase + 32 + 8 * g) = v1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention under an EXPLICIT additive mask, for the ablation masks of the hand-off sampler (models/var.py:557-578, 777-798: block-wise
// patterns that the stage table of the kernels above cannot express).  bias (l, Ktot) fp32, 0 or -inf, shared by all rows and heads
// (the reference's (1, 1, l, Ktot) attn_bias).  One wave per (query, head, row), lane = channel, any cache format: the sampler calls
// this once per image batch (the prefix prefill), so it is written for clarity, not speed.
__device__ __forceinline__ float kv_elem(const void* base, int fmt, bool is_v, size_t head_elems, int Lmax, int key, int c) {
    // element (key, c) of the (row, head) slice that starts `head_elems` * planes into the cache
    if (fmt == 0) return reinterpret_cast<const float*>(base)[head_elems + (size_t)key * 64 + c];
    if (fmt == 1) return __half2float(reinterpret_cast<const __half*>(base)[head_elems + (size_t)key * 64 + c]);
    const int NP = fmt == 2 ? 3 : (fmt == 3 ? 2 : 1);
    const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + head_elems * NP;
    const size_t ps = (size_t)Lmax * 64;
    size_t o;
    if (is_v && fmt == 2) { const int pos = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1); o = (size_t)c * Lmax + pos; }     // bf16x3 planes: V^T rows, bits 2 and 3 of the key swapped
    else o = (size_t)key * 64 + c;                  // K of every planes format, and V of the fp16-plane formats 3 / 4 (row-major like K)
    float v = 0.f;
    for (int k = NP - 1; k >= 0; --k) {
        const uint16_t u = p[k * ps + o];
        v += (fmt == 2) ? __uint_as_float((uint32_t)u << 16) : (float)__builtin_bit_cast(_Float16, u);
    }
    return v;
}

__global__ __launch_bounds__(256) void attention_masked_kernel(const float* __restrict__ q, const void* kc, const void* vc, int fmt, const float* __restrict__ bias,
                                                               float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lmax, int Ktot) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long long)R * H * l) return;
    const int qi = (int)(item % l), h = (int)((item / l) % H), r = (int)(item / ((long long)l * H));
    const size_t head = ((size_t)r * H + h) * (size_t)Lmax * 64;
    const float qv = q[(((size_t)r * H + h) * l + qi) * 64 + lane];
    float m_run = -INFINITY, l_run = 0.f, o = 0.f;
    for (int key = 0; key < Ktot; ++key) {
        const float b = bias[(size_t)qi * Ktot + key];
        if (b == -INFINITY) continue;                                   // wave-uniform
        const float sc = wave_sum(qv * kv_elem(kc, fmt, false, head, Lmax, key, lane)) + b;
        const float m_new = fmaxf(m_run, sc);
        const float alpha = expf(m_run - m_new), pexp = expf(sc - m_new);
        l_run = l_run * alpha + pexp;
        o = o * alpha + pexp * kv_elem(vc, fmt, true, head, Lmax, key, lane);
        m_run = m_new;
    }
    const float res = o / l_run;
    const int row = r * l + qi, col = h * 64 + lane;
    if (outp) {
        const size_t oo = kb_index(row, col, R * l);
        if (pfmt == PLANES_F16X2) { uint16_t hh, ll; split2h(res, hh, ll); outp[oo] = hh; outp[ops + oo] = ll; }
        else { uint16_t p0, p1, p2; split3(res, p0, p1, p2); outp[oo] = p0; outp[ops + oo] = p1; outp[2 * ops + oo] = p2; }
    } else {
        out[(size_t)row * (H * 64) + col] = res;
    }
}

int attention_masked(const float* q, const void* kc, const void* vc, int fmt, const float* bias, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l,
                     int Lmax, int Ktot, hipStream_t stream) {
    SDVAR_CHECK_ARG(q && kc && vc && bias && (out || outp), "attention_masked: null operand");
    SDVAR_CHECK_ARG(fmt >= 0 && fmt <= 4 && R > 0 && H > 0 && l > 0 && Ktot >= l && Ktot <= Lmax, "attention_masked: fmt=%d l=%d Ktot=%d Lmax=%d", fmt, l, Ktot, Lmax);
    SDVAR_CHECK_ARG(!outp || pfmt == PLANES_BF16X3 || pfmt == PLANES_F16X2, "attention_masked: plane format %d", pfmt);
    const long long items = (long long)R * H * l;
    hipLaunchKernelGGL(attention_masked_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, q, kc, vc, fmt, bias, out, outp, ops, pfmt, R, H, l, Lmax, Ktot);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

int attention_bf16x3(const float* q, const void* kc, const void* vc, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lp,
                     int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream);

int attention_f16x2(const float* q, const void* kc, const void* vc, int nkp, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lp,
                    int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream);

// kv_f16: cache format, 0 = fp32, 1 = fp16 (this file), 2 = bf16x3 planes (attention_bf16x3.hip), 3 = f16x2 planes, 4 = one fp16 plane (attention_f16x2.hip)
int attention_f32(const float* q, const void* kc, const void* vc, int kv_f16, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lmax,
                  int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream) {
    SDVAR_CHECK_ARG(!outp || pfmt == PLANES_BF16X3 || pfmt == PLANES_F16X2, "attention: plane format %d", pfmt);
    if (kv_f16 == 3 || kv_f16 == 4) return attention_f16x2(q, kc, vc, kv_f16 == 3 ? 2 : 1, out, outp, ops, pfmt, R, H, l, Lmax, Ktot, n_chunk, qbeg, vis, stream);
    if (kv_f16 == 2) return attention_bf16x3(q, kc, vc, out, outp, ops, pfmt, R, H, l, Lmax, Ktot, n_chunk, qbeg, vis, stream);
    SDVAR_CHECK_ARG(q && kc && vc && (out || outp), "attention: null operand");
    SDVAR_CHECK_ARG(n_chunk >= 1 && n_chunk <= ATT_MAX_CHUNK, "attention: chunk of %d stages unsupported (max %d)", n_chunk, ATT_MAX_CHUNK);
    SDVAR_CHECK_ARG(R > 0 && H > 0 && l > 0 && Ktot >= l && Ktot <= Lmax, "attention: bad lengths l=%d Ktot=%d Lmax=%d", l, Ktot, Lmax);
    AttnArgs a;
    a.q = q; a.kc = kc; a.vc = vc; a.out = out; a.outp = outp; a.ops = ops; a.pfmt = pfmt; a.R = R; a.H = H; a.l = l; a.Lmax = Lmax; a.Ktot = Ktot; a.n_chunk = n_chunk;
    for (int j = 0; j < n_chunk; ++j) {
        a.qbeg[j] = qbeg[j]; a.vis[j] = vis[j];
        SDVAR_CHECK_ARG(vis[j] >= 1 && vis[j] <= Ktot && (j == 0 ? qbeg[0] == 0 : (qbeg[j] > qbeg[j - 1] && vis[j] >= vis[j - 1])), "attention: bad stage table at %d", j);
    }
    a.qbeg[n_chunk] = l;
    const size_t lds = 2 * (size_t)(KT * KSTR + KT * 64) * sizeof(float);
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)attention_f32_kernel<false>, (const void*)attention_f32_kernel<true>);
    if (kv_f16) hipLaunchKernelGGL(attention_f32_kernel<true>, dim3((l + 127) / 128, H, R), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(attention_f32_kernel<false>, dim3((l + 127) / 128, H, R), dim3(256), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar
