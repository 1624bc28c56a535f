// Verify-attention on the bf16 matrix cores with split operands (same arithmetic contract as gemm_bf16x3.hip: every fp32
// operand is the exact sum of three bf16 planes, six of the nine plane products are evaluated, fp32 accumulate), for the
// "planes" KV-cache format.  Same semantics as attention.hip (basic_var.py:107-117 with the mask rows of var.py:108-113):
//     out = softmax(q k^T + block-causal mask) v,   flash-style over 32-key tiles, scores never leave registers.
//
// Why a second kernel: attention.hip is bound by v_mfma_f32_32x32x2_f32 (16 passes per 2 k); at the verify shapes
// (l = 256..425 queries against <= 680 keys, 256 (row, head) pairs) the cache is re-used by every query of the stage, so the
// kernel is matrix-pipe bound long before HBM.  Splitting K and V ONCE when they are appended (elementwise.hip
// qk_norm_append, format 2) lets this kernel stream ready-made planes with the LDS-DMA and spend 6 x 8 passes per 16 k:
// 2.67x fewer matrix cycles for the same fp32-accurate result.
//
// KV cache, format 2 (Lp = Lmax rounded up to a multiple of 64, zero-initialised by the owner):
//     kc  [R][H][3][Lp][64]  bf16   K planes, one 128-byte row per key
//     vc  [R][H][3][64][Lp]  bf16   V^T planes, one row per channel; inside every block of 16 keys the position of key
//                                   b3 b2 b1 b0 is b2 b3 b1 b0 (bits 2 and 3 swapped), which is the order in which the
//                                   score accumulators of one lane hold their keys - P feeds the second MFMA straight
//                                   from registers and the V^T fragment is one ds_read_b128.
// Mapping (wave64): workgroup = 4 waves = 128 queries of one (row, head), one wave owns 32 queries; 72 KB of LDS, two
// workgroups per CU.
//   S^T = K Q^T : A = K fragment (key on the MFMA row, 16 channels per instruction), B = Q planes (split in the prologue).
//   O^T = V^T P^T: A = V^T fragment (channel on the MFMA row, 16 keys per instruction), B = P split in registers.
//   Tiles of 32 keys x (3 K planes + 3 V^T planes) = 24 KB are DMA'd (global_load_lds, 16 B per lane) into a 3-stage ring;
//   the 16-byte chunk index is XOR-swizzled on the global side ((row >> 1) & 7 for the 128-byte K rows, (row >> 2) & 3 for
//   the 64-byte V^T rows) so the fragment reads are conflict-free.
// Algorithmic bytes per launch: as attention.hip (R*H*64*4*(2*Ktot + 2*l)); the planes make the actual cache traffic 1.5x that.
#include <stdlib.h>

#include "common.h"

namespace sdvar {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int ATTP_MAX_CHUNK = 16;
constexpr int AKT = 32;                 // keys per LDS tile
constexpr int APL = AKT * 64;           // bf16 elements per plane tile (K: 32 keys x 64 channels; V^T: 64 channels x 32 keys)
constexpr int ANST = 3;                 // ring depth
constexpr int ASTAGE = 6 * APL;         // one K tile + one V^T tile, 3 planes each: 24 KB

struct AttnPArgs {
    const float* q; const uint16_t* kc; const uint16_t* vc; float* out;
    uint16_t* outp; size_t ops; int pfmt;       // output planes for the proj GEMM: PLANES_BF16X3 or PLANES_F16X2 (common.h)
    int R, H, l, Lp, Ktot;
    int n_chunk;
    int qbeg[ATTP_MAX_CHUNK + 1];
    int vis[ATTP_MAX_CHUNK];
};

__device__ __forceinline__ void split8(const float* v, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    u32x4 a, b, c;
    split8_packed(v, a, b, c);
    p0 = __builtin_bit_cast(bf16x8, a); p1 = __builtin_bit_cast(bf16x8, b); p2 = __builtin_bit_cast(bf16x8, c);
}

// acc += sum over the six kept plane products of A (planes a[0..2]) and B (planes b[0..2]), smallest terms first
#define SDVAR_MFMA6(acc, a, b)                                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef SDVAR_ATT_STAMPS      // diagnostic build only (tools/micro/att_stamps.py): per-wave cycle totals of the four loop phases
__device__ unsigned long long att_stamps[8 * 4096];
#define SDVAR_STAMP(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define SDVAR_STAMP(var) do { } while (0)
#endif
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Measured on gfx950 (tools/micro/mfma_valu.hip, tools/one_attention.py with pieces compiled out): vector-ALU instructions
// do NOT hide under MFMAs of the same SIMD - not from the same wave, not from a co-resident one (MFMA + 8 v_fma per slot:
// 16.0 -> 23.9 ns; only v_exp_f32 overlaps) - and the LDS array is not the limit (halving the fragment reads per MFMA
// changed nothing).  So the kernel time is MFMA time (18 ns each, 96 per tile and wave) plus the softmax / split
// arithmetic, and the arithmetic is written for instruction count: packed fp32 ops on register pairs, v_max3, one v_perm
// per two bf16 results, masking only in the tiles that straddle a visibility boundary.
__global__ __launch_bounds__(256, 2) void attention_bf16x3_kernel(AttnPArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t att_sm[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int q0 = qt * 128;

    const int qi_raw = q0 + wave * 32 + li;
    const int qi = min(qi_raw, a.l - 1);
    int vis_q = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (qi >= a.qbeg[j]) vis_q = a.vis[j];
    const int q_last = min(q0 + 127, a.l - 1);
    int kend = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (q_last >= a.qbeg[j]) kend = a.vis[j];
    const bool wave_active = (q0 + wave * 32) < a.l;

    // Q planes: B operand of S^T = K Q^T; lane (query li, half lh) holds channels 16c + 8lh .. +7 of step c
    bf16x8 qp[4][3];
    {
        const float* pq = a.q + (((size_t)r * a.H + h) * a.l + qi) * 64 + 8 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(pq + 16 * c), u1 = *reinterpret_cast<const f32x4*>(pq + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = u0[e]; v[4 + e] = u1[e]; }
            split8(v, qp[c][0], qp[c][1], qp[c][2]);
        }
    }

    // DMA, 6 instructions per wave per tile: K plane tile = 32 key rows of 128 B (wave w: rows 8w..8w+7, chunk swizzle
    // (row >> 1) & 7); V^T plane tile = 64 channel rows of 64 B (wave w: rows 16w..16w+15, chunk swizzle (row >> 2) & 3).
    // LDS ring of 3 stages (K planes then V^T planes, 24 KB each); tile t lives in stage t % 3 and is requested two
    // iterations before it is read: one 32-key iteration (~1 us) is shorter than the HBM/MALL latency.
    const size_t head = ((size_t)r * a.H + h) * 3 * (size_t)a.Lp * 64;
    const size_t kps = (size_t)a.Lp * 64;                    // plane stride, both operands
    const int krow = 8 * wave + (lane >> 3), kchunk = (lane & 7) ^ ((krow >> 1) & 7);
    const int vrow = 16 * wave + (lane >> 2), vchunk = (lane & 3) ^ ((vrow >> 2) & 3);
    // loop-invariant 32-bit lane offsets + wave-uniform bases (common.h SDVAR_DMA16): no vector address arithmetic per tile
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lk = (uint32_t)(krow * 64 + 8 * kchunk) * 2u, lv = (uint32_t)(vrow * a.Lp + 8 * vchunk) * 2u;
    const char* const bk = reinterpret_cast<const char*>(a.kc + head);
    const char* const bv = reinterpret_cast<const char*>(a.vc + head);
    auto issue = [&](int t) {
        uint16_t* st = att_sm + (t % ANST) * ASTAGE + swave * 512;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            SDVAR_DMA16(lk, bk + (p * kps + (size_t)t * AKT * 64) * 2, SDVAR_LDS_ADDR(st + p * APL));
            SDVAR_DMA16(lv, bv + (p * kps + (size_t)t * AKT) * 2, SDVAR_LDS_ADDR(st + (3 + p) * APL));
        }
    };

    f32x16 o0, o1;                        // O^T accumulators: d = db*32 + (reg&3) + 8*(reg>>2) + 4*lh, column = this query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    // running maximum in log2 units (M = m log2 e, rounded once per tile so that p and the rescale factor use the same value)
    float M_run = -INFINITY, l_run = 0.f;

    const int swk = (li >> 1) & 7, swv = (li >> 2) & 3;
    const int ntiles = (kend + AKT - 1) / AKT;
    const float L2E = 1.4426950408889634f;
    issue(0);
    if (ntiles > 1) issue(1);
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
    (void)tacc; (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts4; (void)ts5;
    for (int t = 0; t < ntiles; ++t) {
        SDVAR_STAMP(ts0);
        // tile t must have landed; the requests of tile t+1 (issued one iteration ago) may still be in flight
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // ... and every wave is done with tile t-1, whose stage is refilled now
        if (t + 2 < ntiles) issue(t + 2);
        if (!wave_active) continue;
        SDVAR_STAMP(ts1);
        const int k0 = t * AKT;
        const uint16_t* sk = att_sm + (t % ANST) * ASTAGE + li * 64;              // this lane's K row (key li)
        const uint16_t* sv = att_sm + (t % ANST) * ASTAGE + 3 * APL + li * 32;    // this lane's V^T rows (channels li and 32 + li)
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 kf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const bf16x8*>(sk + p * APL + 8 * ((2 * c + lh) ^ swk));
            SDVAR_MFMA6(s, kf, qp[c]);
        }
#ifdef SDVAR_ATT_STAMPS
        asm volatile("v_mov_b32 %0, %0" : "+v"(s[15]) :: "memory");      // forces the score chain to complete
        SDVAR_STAMP(ts2);
#endif
        // V^T fragments of the tile: issued now, consumed after the softmax arithmetic
        bf16x8 vf[2][2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                vf[j][0][p] = *reinterpret_cast<const bf16x8*>(sv + p * APL + 8 * ((2 * j + lh) ^ swv));
                vf[j][1][p] = *reinterpret_cast<const bf16x8*>(sv + p * APL + 1024 + 8 * ((2 * j + lh) ^ swv));
            }
        // ---- mask (only in tiles that reach past some query's visible keys; this lane holds keys k0 + (i&3) + 8*(i>>2) + 4*lh)
        if (__builtin_amdgcn_ballot_w64(k0 + AKT > vis_q) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) if (k0 + (i & 3) + 8 * (i >> 2) + 4 * lh >= vis_q) s[i] = -INFINITY;
        }
        // ---- online softmax in log2 units: p = 2^(s log2e - M)
        float mloc = s[0];                                      // scores are never NaN: plain v_max3_f32, no canonicalisation
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mloc) : "v"(s[0]), "v"(s[1]), "v"(s[2]));
#pragma unroll
        for (int i = 3; i < 15; i += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mloc) : "v"(mloc), "v"(s[i]), "v"(s[i + 1]));
        asm("v_max_f32 %0, %1, %2" : "=v"(mloc) : "v"(mloc), "v"(s[15]));
        {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
            mloc = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));                  // the other half of the query's keys
        }
        const float M_new = fmaxf(M_run, mloc * L2E);          // finite from the first tile on (key 0 is always visible)
        const float alpha = __builtin_amdgcn_exp2f(M_run - M_new);
        M_run = M_new;
        const f32x2 l2e2 = {L2E, L2E}, nM2 = {-M_new, -M_new};
        f32x2 pr[8], ls2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x2 x = __builtin_elementwise_fma(f32x2{s[2 * e], s[2 * e + 1]}, l2e2, nM2);      // v_pk_fma_f32
            pr[e] = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            ls2 += pr[e];
        }
        float lsum = ls2[0] + ls2[1];
        {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
            lsum = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
        }
        l_run = l_run * alpha + lsum;
        // ---- exact 3-way split of the probabilities, two at a time; pair e of step j = registers 8j + 2e, 8j + 2e + 1, which
        // are k-slots 8lh + 2e, +1 of the MFMA, i.e. keys 16j + {0..3, 8..11} + 4lh - the permuted key order of the V^T rows
        u32x4 pw[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 x = pr[4 * j + e];
                const f32x2 r1 = x - __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x) & 0xFFFF0000u);
                const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, r1) & 0xFFFF0000u);
                pw[j][0][e] = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                pw[j][1][e] = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pw[j][2][e] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
#ifdef SDVAR_ATT_STAMPS
        asm volatile("v_mov_b32 %0, %0" : "+v"(pw[1][2][3]) :: "memory");
        asm volatile("v_mov_b32 %0, %0" : "+v"(o1[15]) :: "memory");
        SDVAR_STAMP(ts3);
#endif
        // ---- O^T += V^T P^T
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 pp[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) pp[p] = __builtin_bit_cast(bf16x8, pw[j][p]);
            SDVAR_MFMA6(o0, vf[j][0], pp);
            SDVAR_MFMA6(o1, vf[j][1], pp);
        }
#ifdef SDVAR_ATT_STAMPS
        asm volatile("v_mov_b32 %0, %0" : "+v"(o0[15]) :: "memory");
        asm volatile("v_mov_b32 %0, %0" : "+v"(o1[15]) :: "memory");
        SDVAR_STAMP(ts4);
        tacc[0] += ts1 - ts0; tacc[1] += ts2 - ts1; tacc[2] += ts3 - ts2; tacc[3] += ts4 - ts3; if (t) tacc[4] += ts0 - ts5;
        ts5 = ts4;
#endif
    }
#ifdef SDVAR_ATT_STAMPS
    if (lane == 0) {
        const int wid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
        if (wid < 4096) for (int i = 0; i < 5; ++i) att_stamps[8 * wid + i] = tacc[i];
    }
#endif

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

int attention_bf16x3(const float* q, const void* kc, const void* vc, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lp,
                     int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream) {
    SDVAR_CHECK_ARG(q && kc && vc && (out || outp), "attention: null operand");
    SDVAR_CHECK_ARG(n_chunk >= 1 && n_chunk <= ATTP_MAX_CHUNK, "attention: chunk of %d stages unsupported (max %d)", n_chunk, ATTP_MAX_CHUNK);
    SDVAR_CHECK_ARG(R > 0 && H > 0 && l > 0 && Ktot >= l && Ktot <= Lp, "attention: bad lengths l=%d Ktot=%d Lmax=%d", l, Ktot, Lp);
    SDVAR_CHECK_ARG(Lp % 64 == 0, "attention: the planes KV format needs Lmax %% 64 == 0 (got %d)", Lp);
    AttnPArgs a;
    a.q = q; a.kc = (const uint16_t*)kc; a.vc = (const uint16_t*)vc; a.out = out; a.outp = outp; a.ops = ops; a.pfmt = pfmt;
    a.R = R; a.H = H; a.l = l; a.Lp = Lp; a.Ktot = Ktot; a.n_chunk = n_chunk;
    for (int j = 0; j < n_chunk; ++j) {
        a.qbeg[j] = qbeg[j]; a.vis[j] = vis[j];
        SDVAR_CHECK_ARG(vis[j] >= 1 && vis[j] <= Ktot && (j == 0 ? qbeg[0] == 0 : (qbeg[j] > qbeg[j - 1] && vis[j] >= vis[j - 1])), "attention: bad stage table at %d", j);
    }
    a.qbeg[n_chunk] = l;
    const size_t lds = ANST * (size_t)ASTAGE * sizeof(uint16_t);
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)attention_bf16x3_kernel);
    hipLaunchKernelGGL(attention_bf16x3_kernel, dim3((l + 127) / 128, H, R), dim3(256), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar

#ifdef SDVAR_ATT_STAMPS
extern "C" int sdvar_debug_att_stamps(unsigned long long* host_out, int n_waves) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sdvar::att_stamps), sizeof(unsigned long long) * 8 * (size_t)n_waves);
}
#endif
