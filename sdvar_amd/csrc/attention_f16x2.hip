// Verify-attention on the gfx950 f16 matrix cores with two-plane split operands (the arithmetic contract of gemm_f16x2.hip: an fp32
// value is held as h + l, two fp16 numbers, to 2^-22; three of the four plane products are evaluated, fp32 accumulate), for the
// fp16-plane KV-cache formats.  Same semantics and structure as attention_bf16x3.hip (basic_var.py:107-117 with the mask rows of
// var.py:108-113, flash-style over 32-key tiles, scores never leave registers); what changes is the operand format:
//
//   NKP = 2 (cache format 3, fp32 models in gemm mode f16x2): K and V are split when they are appended (elementwise.hip
//       qk_norm_append, format 3); S^T = Kh Qh + Kh Ql + Kl Qh and O^T = Vh Ph + Vh Pl + Vl Ph: 3 + 3 MFMAs per 16 k instead of the
//       6 + 6 of bf16x3, and the cache is 4 bytes per element - the size of the fp32 cache it stands for (bf16x3 planes: 6).
//   NKP = 1 (cache format 4, BASELINE config P4's fp16 KV cache): the cache holds fp16(k), fp16(v) - exactly what the reference's
//       half-precision cache would hold - so the K / V operands ARE exact fp16 and only Q and P need their low plane: 2 + 2 MFMAs.
//
// KV cache (Lp = Lmax rounded up to a multiple of 64, zero-initialised by the owner):
//     kc  [R][H][NKP][Lp][64]  fp16   K planes, one 128-byte row per key
//     vc  [R][H][NKP][Lp][64]  fp16   V planes, the SAME row-major layout (round 2 kept a transposed, key-permuted V^T copy that only a separate pass
//                                     through LDS could write; now the QKV GEMM's epilogue appends k and v rows alike).  The V^T fragments of O^T = V^T P^T
//                                     are taken from the row-major LDS tile with ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16 lanes a block of
//                                     4 keys x 16 channels comes back transposed, lane = channel, and the 4 key rows are free to choose, so the two reads
//                                     of a fragment fetch exactly the keys {0..3, 8..11} + 4 lh of the k16 step - the order in which a lane's score
//                                     registers hold their keys.
// Mapping (wave64): workgroup = 4 waves = 128 queries of one (row, head); tiles of 32 keys x (NKP K planes + NKP V planes) are
// DMA'd (global_load_lds, 16 B per lane) into a 3-stage ring (16 KB per stage at NKP = 2).  Swizzles (on the DMA source chunk and on the read address):
// K rows chunk ^ ((row >> 1) & 7) (ds_read_b128 by 32 key rows); V rows chunk ^ (4 ((row >> 1) & 1)): the transposed read touches 4 consecutive rows x 64 B
// per 32 lanes and rows r, r + 2 share their banks (128-byte rows), so rows 2, 3 of every 4 swap their 64-byte halves: conflict-free.
// Algorithmic bytes per launch: R*H*64*4*(2*Ktot + 2*l) for NKP = 2 - now also the bytes the cache really holds.
#include <stdlib.h>

#include "common.h"

namespace sdvar {

unsigned long long* debug_get_gemm_stamps();      // gemm_bf16x3.hip: the diagnostic stamp buffer (SDVAR_ATT_STAMPS builds of the 8-wave kernel write slot times to it)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int ATTH_MAX_CHUNK = 16;
constexpr int AKT = 32;                 // keys per LDS tile
constexpr int APL = AKT * 64;           // fp16 elements per plane tile (K: 32 keys x 64 channels; V^T: 64 channels x 32 keys)
constexpr int ANST = 3;                 // ring depth

struct AttnHArgs {
    const float* q; const uint16_t* kc; const uint16_t* vc; float* out;
    uint16_t* outp; size_t ops; int pfmt;       // output planes for the proj GEMM: PLANES_BF16X3 or PLANES_F16X2 (common.h)
    int R, H, l, Lp, Ktot;
    int n_chunk;
    int qbeg[ATTH_MAX_CHUNK + 1];
    int vis[ATTH_MAX_CHUNK];
    unsigned long long* stamps;            // diagnostic builds only (-DSDVAR_ATT_STAMPS)
};

// eight consecutive values -> the two packed fp16x8 words (h = fp16(x), l = fp16(x - h)); values saturate at the fp16 range
__device__ __forceinline__ void split8h(const float* v, f16x8& h, f16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = clamp_f16_range(v[e]);            // NaN stays NaN (common.h)
        const _Float16 hh = (_Float16)x;
        h[e] = hh; l[e] = (_Float16)(x - (float)hh);
    }
}

// One V^T fragment: 8 keys x this lane's channel = two ds_read_b64_tr_b16 (rows `off` .. + 3 and `off` + 8 rows .. + 3 of the addressed 4 x 16 blocks).
// EXEC must be all ones (the gather crosses lanes): the callers keep whole waves active.  The wait is the compiler's (the builtin is counted in lgkmcnt).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 lds_read_tr8(uint32_t addr, int off) {
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(uintptr_t)(addr + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(uintptr_t)(addr + off + 1024));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// acc += A.B over the kept plane products, smallest terms first: a = planes of the cache operand (NKP of them), b = {h, l} of Q or P
template <int NKP>
__device__ __forceinline__ void mfma_planes(f32x16& acc, const f16x8* a, const f16x8* b) {
    if (NKP == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// 16-byte LDS read through the compiler (counted in lgkmcnt by hipcc): addr = LDS byte address
#define SDVAR_LDS_RDP(dst, addr) (dst) = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((uintptr_t)(addr))

// The vector-ALU part of one 32-key tile (both kernels): mask, online softmax with a DEFERRED running maximum, two-plane split of P.
//   s      S^T accumulator of the tile: register i = key k0 + (i & 3) + 8 (i >> 2) + 4 lh of this lane's query
//   M_run  reference maximum in log2 units; it follows a tile's maximum only when some query of the wave sees a score more than 2^ATT_DEFER above it
//          (cdna_hip_programming.md T13): P <= 2^ATT_DEFER stays far inside the fp16 range of its high plane, and the O / l rescale (33 multiplies) leaves
//          almost every tile.  Exact in exact arithmetic: O and l carry the same reference.
//   pp     P^T operand planes of the two k16 steps
constexpr float ATT_DEFER = 6.0f;
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// `mid` runs between the exponentials and the two-plane split of P: the two-slot schedule of the 8-wave kernel issues its V fragment reads there (a wave can have
// 15 LDS requests outstanding: 8 K reads before the arithmetic, 16 V reads in the middle, all landed when the split is done)
template <class Mid = NoMid>
__device__ __forceinline__ void softmax_tile(f32x16& s, int k0, int vis_q, int lh, float& M_run, float& l_run, f32x16& o0, f32x16& o1, f16x8 (&pp)[2][2], Mid mid = Mid()) {
    const float L2E = 1.4426950408889634f;
    if (__builtin_amdgcn_ballot_w64(k0 + AKT > vis_q) != 0) {            // only the tiles that reach past some query's visible keys: a real (wave-uniform) branch
        asm volatile("" ::: "memory");                                   // ... the 32 compare / select instructions are not to be if-converted into every tile
        const int d = vis_q - k0 - 4 * lh;                               // register i holds key k0 + (i & 3) + 8 (i >> 2) + 4 lh: masked from d on
#pragma unroll
        for (int i = 0; i < 16; ++i) if ((i & 3) + 8 * (i >> 2) >= d) s[i] = -INFINITY;
    }
    // The first maximum is the COMPILER's instruction: s comes straight out of the S^T MFMAs, and hipcc pads the matrix-pipe -> vector-ALU read hazard only for
    // instructions it can see (cdna_hip_programming.md 5.7).  With the asm chain directly behind the MFMAs (the mask arithmetic that used to sit in between is a
    // skipped branch now) some lanes read a not yet written accumulator for the maximum - harmless for the quotient (any reference works) but different from
    // launch to launch by ~1e-7, and an overflow risk.  Every later read of s is ordered behind this one (same MFMA results).
    float mloc = fmaxf(s[0], s[1]);
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mloc) : "v"(mloc), "v"(s[2]), "v"(s[3]));
#pragma unroll
    for (int i = 4; i < 16; i += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mloc) : "v"(mloc), "v"(s[i]), "v"(s[i + 1]));
    {
        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
    }
    // deferred running maximum: moved (and O, l rescaled) only when some query of the wave sees a score more than 2^ATT_DEFER above its reference
    const float mt = mloc * L2E;
    if (__builtin_amdgcn_ballot_w64(mt > M_run + ATT_DEFER) != 0) {       // wave-uniform; always taken in tile 0 (M_run = -inf)
        const float M_new = fmaxf(M_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(M_run - M_new);
        M_run = M_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
    const f32x2 l2e2 = {L2E, L2E}, nM2 = {-M_run, -M_run};
    f32x2 pr[8], ls2 = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const f32x2 x = __builtin_elementwise_fma(f32x2{s[2 * e], s[2 * e + 1]}, l2e2, nM2);
        pr[e] = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        ls2 += pr[e];
    }
    float lsum = ls2[0] + ls2[1];
    {
        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        lsum = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
    }
    l_run += lsum;
    mid();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        u32x4 ph, pl;
#if defined(SDVAR_ATTN_P1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t hw; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw) : "v"(pr[4 * j + e][0]), "v"(pr[4 * j + e][1])); ph[e] = hw; pl[e] = 0u; }
#elif !defined(SDVAR_ATTN_FMA_MIX)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                    // p in [0, 2^ATT_DEFER]: inside the fp16 range, no clamp
            uint32_t hw, lw;
            split2h_pk_raw(pr[4 * j + e][0], pr[4 * j + e][1], hw, lw);
            ph[e] = hw; pl[e] = lw;
        }
#else
        // -DSDVAR_ATTN_FMA_MIX (an EXPERIMENT of round 4, measured and not adopted: bit-identical, deterministic, 16 vector instructions fewer per tile and wave - and not a
        // microsecond faster, profiles/r04_r_attn_valu_experiment.log: the slot is not bound by vector issue).
        // h = fp16(p) by the compiler's v_cvt_pk_f16_f32; l = fp16(p - h) in ONE instruction per value: v_fma_mixlo / mixhi_f16 take h as an fp16 source and p as an fp32
        // source (h * -1.0 + p, computed exactly, rounded once: bit-identical to the convert-subtract-convert sequence on 1 M random pairs incl. subnormal low planes,
        // tools/micro/fma_mix_probe.hip) - 3 instead of 5 vector instructions per pair, 16 fewer per tile and wave in a slot whose length IS its vector instruction count
        // (profiles/r04_h_attn_counters.json).  hipcc pads no hazards around asm statements (cdna_hip_programming.md 5.7), so the block carries its own: s_nop 1 in front
        // (the p values come out of v_exp_f32: transcendental-result forwarding), the four mixlo before the four mixhi (a partial write is not read by the next
        // instruction: dst_sel forwarding), s_nop 0 behind (the consumer may be a compiler-made copy).
        {
            uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
            const f32x2 q0 = pr[4 * j], q1 = pr[4 * j + 1], q2 = pr[4 * j + 2], q3 = pr[4 * j + 3];
            {
                const f16x2v c0 = __builtin_convertvector(q0, f16x2v), c1 = __builtin_convertvector(q1, f16x2v), c2 = __builtin_convertvector(q2, f16x2v),
                             c3 = __builtin_convertvector(q3, f16x2v);
                h0 = __builtin_bit_cast(uint32_t, c0); h1 = __builtin_bit_cast(uint32_t, c1); h2 = __builtin_bit_cast(uint32_t, c2); h3 = __builtin_bit_cast(uint32_t, c3);
            }
            asm volatile("s_nop 1\n\t"
                         "v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "s_nop 0"
                         : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                         : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(q0[0]), "v"(q0[1]), "v"(q1[0]), "v"(q1[1]), "v"(q2[0]), "v"(q2[1]), "v"(q3[0]), "v"(q3[1]));
            ph = u32x4{h0, h1, h2, h3}; pl = u32x4{l0, l1, l2, l3};
        }
#endif
        pp[j][0] = __builtin_bit_cast(f16x8, ph); pp[j][1] = __builtin_bit_cast(f16x8, pl);
    }
}

// The softmax / split arithmetic is written for instruction count (vector-ALU work does not hide under MFMAs on gfx950: see
// attention_bf16x3.hip): packed fp32 ops on register pairs, v_max3, masking only in the tiles that straddle a visibility boundary.
// O^T += V^T P^T.  -DSDVAR_ATTN_P1 (an EXPERIMENT, never the product build: tools/micro/attn_p1_exp.sh) drops the low plane of P - 2 instead of 3 products and no
// second conversion pass in the softmax: P then carries 11 significand bits, 2^-11 relative per weight (DESIGN.md section 4a reports what that does to the logits).
template <int NKP>
__device__ __forceinline__ void mfma_pv(f32x16& acc, const f16x8* a, const f16x8* b) {
#ifdef SDVAR_ATTN_P1
    if (NKP == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
#else
    mfma_planes<NKP>(acc, a, b);
#endif
}
template <int NKP>
__global__ __launch_bounds__(256, 2) void attention_f16x2_kernel(AttnHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t att_sm[];
    constexpr int ASTAGE = 2 * NKP * APL;         // one K tile + one V^T tile, NKP planes each

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
    f16x8 qp[4][2];
    {
        const float* pq = a.q + (((size_t)r * a.H + h) * a.l + qi) * 64 + 8 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(pq + 16 * c), u1 = *reinterpret_cast<const f32x4*>(pq + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = u0[e]; v[4 + e] = u1[e]; }
            split8h(v, qp[c][0], qp[c][1]);
        }
    }

    // DMA, 2 NKP instructions per wave per tile: a K / V plane tile = 32 key rows of 128 B (wave w: rows 8w..8w+7; K chunk swizzle (row >> 1) & 7, V chunk
    // swizzle 4 ((row >> 1) & 1)).  LDS ring of 3 stages (K planes then V planes); tile t lives in stage t % 3 and is requested two
    // iterations before it is read: one 32-key iteration (~1 us) is shorter than the HBM/MALL latency.
    const size_t head = ((size_t)r * a.H + h) * NKP * (size_t)a.Lp * 64;
    const size_t kps = (size_t)a.Lp * 64;                    // plane stride, both operands
    const int krow = 8 * wave + (lane >> 3), kchunk = (lane & 7) ^ ((krow >> 1) & 7), vchunk = (lane & 7) ^ (4 * ((krow >> 1) & 1));
    // loop-invariant 32-bit lane offsets + wave-uniform bases (common.h SDVAR_DMA16): no vector address arithmetic per tile
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lk = (uint32_t)(krow * 64 + 8 * kchunk) * 2u, lv = (uint32_t)(krow * 64 + 8 * vchunk) * 2u;
    const char* const bk = reinterpret_cast<const char*>(a.kc + head);
    const char* const bv = reinterpret_cast<const char*>(a.vc + head);
    auto issue = [&](int t) {
        uint16_t* st = att_sm + (t % ANST) * ASTAGE + swave * 512;
#pragma unroll
        for (int p = 0; p < NKP; ++p) {
            SDVAR_DMA16(lk, bk + (p * kps + (size_t)t * AKT * 64) * 2, SDVAR_LDS_ADDR(st + p * APL));
            SDVAR_DMA16(lv, bv + (p * kps + (size_t)t * AKT * 64) * 2, SDVAR_LDS_ADDR(st + (NKP + p) * APL));
        }
    };

    f32x16 o0, o1;                        // O^T accumulators: d = db*32 + (reg&3) + 8*(reg>>2) + 4*lh, column = this query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    // running maximum in log2 units (M = m log2 e, rounded once per tile so that p and the rescale factor use the same value)
    float M_run = -INFINITY, l_run = 0.f;

    const int swk = (li >> 1) & 7;
    // transposed V reads: lane = (lh, channel block cb = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3) supplies the address of key row 4 lh + q (+ 16 j, + 8
    // for the second half of the fragment), channels 32 db + 16 cb + 4 p .. + 3; with the row swizzle the 64-byte half of rows q >= 2 is flipped
    const int vq = (lane >> 2) & 3, vp = lane & 3, vcb = (lane >> 4) & 1;
    const uint32_t voff0 = (uint32_t)((4 * lh + vq) * 128 + (((vq >> 1) & 1) * 64) + vcb * 32 + vp * 8);       // db = 0; db = 1 is ^ 64
    const int ntiles = (kend + AKT - 1) / AKT;
    issue(0);
    if (ntiles > 1) issue(1);
    for (int t = 0; t < ntiles; ++t) {
        // tile t must have landed; the requests of tile t+1 (issued one iteration ago) may still be in flight
        if (t + 1 < ntiles) { if (NKP == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // ... and every wave is done with tile t-1, whose stage is refilled now
        if (t + 2 < ntiles) issue(t + 2);
        if (!wave_active) continue;
        const int k0 = t * AKT;
        const uint16_t* sk = att_sm + (t % ANST) * ASTAGE + li * 64;              // this lane's K row (key li)
        const uint32_t sv0 = SDVAR_LDS_ADDR(att_sm + (t % ANST) * ASTAGE + NKP * APL) + voff0, sv1 = sv0 ^ 64u;     // V tile, channel halves db = 0 / 1
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f16x8 kf[NKP];
#pragma unroll
            for (int p = 0; p < NKP; ++p) kf[p] = *reinterpret_cast<const f16x8*>(sk + p * APL + 8 * ((2 * c + lh) ^ swk));
            mfma_planes<NKP>(s, kf, qp[c]);
        }
        // V^T fragments of the tile (two transposed 8-byte reads each: keys 16 j + 4 lh + {0..3} and + 8): issued now, consumed after the softmax arithmetic
        f16x8 vf[2][2][NKP];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NKP; ++p) {
                vf[j][0][p] = lds_read_tr8(sv0, j * 2048 + p * (APL * 2));
                vf[j][1][p] = lds_read_tr8(sv1, j * 2048 + p * (APL * 2));
            }
        // ---- mask, online softmax (deferred running maximum), two-plane split of P
        f16x8 pp[2][2];
        softmax_tile(s, k0, vis_q, lh, M_run, l_run, o0, o1, pp);
        // ---- O^T += V^T P^T
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            mfma_pv<NKP>(o0, vf[j][0], pp[j]);
            mfma_pv<NKP>(o1, vf[j][1], pp[j]);
        }
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
// The same attention for MORE than 128 queries per (row, head) (stages 8 - 9 of the 256^2 ladder, every large verify chunk): ONE workgroup of 8 waves = 256
// queries streams the (row, head) cache once (the 4-wave kernel above ran two workgroups over the same keys: 1.6x the algorithmic HBM traffic, measured), and
// the two waves of every SIMD alternate between the matrix pipe and the vector ALU - the schedule of gemm_f16x2_v4_kernel (MI355X_MICROARCH.md "Two waves per
// SIMD"): a 32-key tile costs a wave 12 + 12 MFMAs (768 matrix-pipe cycles) AND ~450 vector-ALU cycles of softmax / two-plane split, which one in-order wave
// cannot overlap with its own MFMAs.
//     slot (s_barrier between slots):   4t      4t+1    4t+2    4t+3
//     waves 0-3:                        S(t)    V1(t)   PV(t)   V2(t)
//     waves 4-7 (one slot behind):      V2(t-1) S(t)    V1(t)   PV(t)
//   S(t)  = 12 MFMAs  S^T = K Q^T of tile t (K fragments already in registers)
//   V1(t) = mask, online softmax (running maximum moved only when a score exceeds it by more than 2^6: cdna_hip_programming.md T13), two-plane split of P, O rescale
//   PV(t) = 12 MFMAs  O^T += V^T P^T (V^T fragments already in registers)
//   V2(t) = fragment reads of tile t+1 (K: 8 x b128, V^T: 16 x b64 transposed) + the LDS-DMA of tile t+3 into the stage tile t has just left
// so in every slot one wave of a SIMD feeds the matrix pipe and the other one the vector ALU / LDS.  A tile is completely in registers after V2(t-1): the ring
// (3 stages of 16 KB) holds three tiles in flight.  Tile t+1 must have landed before ANY wave reads it in slot 4t+3: every wave waits for its own share
// (counted vmcnt) at the end of the slot before - PV(t) for the early half, V1(t) for the late half; the waits stand in both segments of both halves.
//
// SCHED = 1 (default since the end of round 3): TWO slots per tile.  tools/micro/valu_rate.hip: ONE wave issues a vector instruction every ~5 cycles, two waves of a SIMD
// together one every ~2.5 (profiles/r03_B_valu_rate.log) - a wave alone in its softmax slot runs the vector ALU at half its rate, and the in-kernel stamps of the
// four-slot schedule show exactly that: ~1050 cycles for the ~110 instructions of V1, plus ~710 for the bare fragment reads of V2, against 384 cycles of MFMA
// in the partner's slot (tile total 3550 cycles per half).  So the slots are re-cut to balance one wave's vector work against BOTH of the partner's MFMA groups:
//     slot (s_barrier between slots):   2t                  2t+1
//     waves 0-3:                        M(t)                V(t)
//     waves 4-7 (one slot behind):      V(t-1)              M(t)
//   M(t) = PV(t-1) then S(t): 24 MFMAs (768 matrix-pipe cycles), every fragment already in registers; then the counted vmcnt wait for tile t+1
//   V(t) = the fragment reads of V(t) and K(t+1) are ISSUED first (they land under the arithmetic), then mask / online softmax / split of tile t, the LDS-DMA of
//          tile t+3 into the stage tile t-1 left (its last reader, the late half's V(t-1), ended with the previous slot), lgkmcnt(0)
// Ring of 4 stages (64 KB): tiles t (V part still unread), t+1, t+2 resident or in flight while t+3 is requested.

template <int NKP, int SCHED>
__global__ __launch_bounds__(512, 2) void attention_f16x2_pp_kernel(AttnHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t att_sm[];
    constexpr int ASTAGE = 2 * NKP * APL;         // one K tile + one V tile, NKP planes each
    constexpr int PP_NST = SCHED == 0 ? 3 : 2 * SCHED + 2;         // ring depth: 3 (four-slot schedule), 4 / 6 / 8 (two-slot schedule); the launcher sizes the LDS the same way

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
    const int q0 = qt * 256;
    const int late = wave >> 2;                                 // waves 4-7 run one slot behind

    const int qi_raw = q0 + wave * 32 + li;
    const int qi = min(qi_raw, a.l - 1);
    int vis_q = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (qi >= a.qbeg[j]) vis_q = a.vis[j];
    const int q_last = min(q0 + 255, a.l - 1);
    int kend = a.vis[0];
#pragma unroll 1
    for (int j = 1; j < a.n_chunk; ++j) if (q_last >= a.qbeg[j]) kend = a.vis[j];
    const bool wave_active = (q0 + wave * 32) < a.l;            // wave-uniform: an inactive wave keeps every barrier and its share of the DMA, nothing else

    f16x8 qp[4][2];
    {
        const float* pq = a.q + (((size_t)r * a.H + h) * a.l + qi) * 64 + 8 * lh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(pq + 16 * c), u1 = *reinterpret_cast<const f32x4*>(pq + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = u0[e]; v[4 + e] = u1[e]; }
            split8h(v, qp[c][0], qp[c][1]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" : "+v"(qp[c][0]), "+v"(qp[c][1])); }      // finished HERE: hipcc would sink the split (and its vmcnt waits, which
    }                                                                                            // drain the DMA queue too) to the first MFMA, behind the tile requests

    // DMA: a tile = NKP K plane tiles + NKP V plane tiles of 32 rows x 128 B (4 instructions each); waves 0-3 bring K rows 8 w .. 8 w + 7 of every plane,
    // waves 4-7 the V rows: NKP instructions per wave and tile
    const size_t head = ((size_t)r * a.H + h) * NKP * (size_t)a.Lp * 64;
    const size_t kps = (size_t)a.Lp * 64;
    const int drow = 8 * (wave & 3) + (lane >> 3);
    const int dchunk = (lane & 7) ^ (late ? 4 * ((drow >> 1) & 1) : ((drow >> 1) & 7));
    const uint32_t ld = (uint32_t)(drow * 64 + 8 * dchunk) * 2u;
    const char* const bsrc = reinterpret_cast<const char*>((late ? a.vc : a.kc) + head);
    const uint32_t lds0 = SDVAR_LDS_ADDR(att_sm);
    auto issue = [&](int t) {
        const uint32_t st = lds0 + (uint32_t)((t % PP_NST) * ASTAGE + (late ? NKP * APL : 0) + (wave & 3) * 512) * 2u;
#pragma unroll
        for (int p = 0; p < NKP; ++p) SDVAR_DMA16(ld, bsrc + (p * kps + (size_t)t * AKT * 64) * 2, st + (uint32_t)(p * APL) * 2u);
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float M_run = -INFINITY, l_run = 0.f;

    const int swk = (li >> 1) & 7;
    const int vq = (lane >> 2) & 3, vp = lane & 3, vcb = (lane >> 4) & 1;
    const uint32_t koff = (uint32_t)(li * 128), voff0 = (uint32_t)(NKP * APL * 2 + (4 * lh + vq) * 128 + (((vq >> 1) & 1) * 64) + vcb * 32 + vp * 8);
    const int ntiles = (kend + AKT - 1) / AKT;

    f16x8 kf[4][NKP], vf[2][2][NKP];
    auto read_k = [&](int t) {                  // K fragments of tile t -> registers
        const uint32_t sb = lds0 + (uint32_t)((t % PP_NST) * ASTAGE) * 2u;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < NKP; ++p) SDVAR_LDS_RDP(kf[c][p], sb + koff + (uint32_t)(p * APL * 2 + 16 * ((2 * c + lh) ^ swk)));
    };
    auto read_v = [&](int t) {                  // V^T fragments of tile t -> registers
        const uint32_t sb = lds0 + (uint32_t)((t % PP_NST) * ASTAGE) * 2u;
        const uint32_t sv0 = sb + voff0, sv1 = sv0 ^ 64u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NKP; ++p) {
                vf[j][0][p] = lds_read_tr8(sv0, j * 2048 + p * (APL * 2));
                vf[j][1][p] = lds_read_tr8(sv1, j * 2048 + p * (APL * 2));
            }
    };
    auto read_tile = [&](int t) { read_k(t); read_v(t); };
#define SDVAR_PP_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef SDVAR_ATT_STAMPS          // slot boundaries of tile 10 (waves 0 and 4 of workgroup (0, 0, 0)) -> stamps[16 (wave / 4) + k]; kernel phases -> stamps[32 ..]
    unsigned long long stm[5] = {0, 0, 0, 0, 0}, wsr[4] = {0, 0, 0, 0}, wsc[4] = {0, 0, 0, 0};
    wsr[0] = __builtin_amdgcn_s_memrealtime(); wsc[0] = __builtin_amdgcn_s_memtime();
#define SDVAR_PP_STAMP(k) do { if (t == 10) stm[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SDVAR_PP_STAMP(k) do { } while (0)
#endif

    if constexpr (SCHED >= 1) {
        // D = request distance in tiles (ring of D + 1 stages): tile t + D is requested in V(t).  SCHED 1: D = 3 (64 KB), 2: D = 5 (96 KB), 3: D = 7 (128 KB)
        constexpr int D = PP_NST - 1;
        auto wait_tiles = [&](int n) {              // of this wave's requests at most the newest n tiles (NKP instructions each) may stay in flight
            switch (n * NKP) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            }
        };
        // requests before the loop: tiles 0 .. D-1 (tile D follows in V(0)); tile 0 must have landed before its K fragments are read
        for (int t = 0; t < D && t < ntiles; ++t) issue(t);
        wait_tiles(min(D, ntiles) - 1);
        SDVAR_PP_SLOT();
        if (late) SDVAR_PP_SLOT();                                  // wave-uniform
        read_k(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f16x8 pp[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { pp[j][0] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; pp[j][1] = pp[j][0]; }
#ifdef SDVAR_ATT_STAMPS
        wsr[1] = __builtin_amdgcn_s_memrealtime(); wsc[1] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
            const int k0 = t * AKT;
            SDVAR_PP_STAMP(0);
            // ---------------- M(t): PV(t-1), S(t)
            f32x16 s;                                   // an inactive wave never reads it
#ifdef SDVAR_ATT_EXP          // timing experiments only (results wrong; tools/micro/attn_lds_exp.sh): bit 0 no V fragment reads, 1 no K fragment reads, 2 no softmax arithmetic, 3 no MFMAs
            if (wave_active && !((SDVAR_ATT_EXP) & 8)) {
#else
            if (wave_active) {
#endif
                if (t > 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        mfma_pv<NKP>(o0, vf[j][0], pp[j]);
                        mfma_pv<NKP>(o1, vf[j][1], pp[j]);
                    }
                }
                // the first product takes a ZERO C operand (an inline constant of the instruction) instead of a zeroed accumulator: the 16 + 7 register moves that cleared
                // s stood in this slot of every tile, and matrix and vector instructions of a SIMD's two waves do not overlap (section 4a of DESIGN.md)
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                s = zero16;
                mfma_planes<NKP>(s, kf[0], qp[0]);
#pragma unroll
                for (int c = 1; c < 4; ++c) mfma_planes<NKP>(s, kf[c], qp[c]);
            }
            if (t + 1 < ntiles) wait_tiles(min(D - 2, ntiles - 2 - t));          // tile t + 1 landed; the tiles requested after it stay in flight
            SDVAR_PP_SLOT(); SDVAR_PP_STAMP(1);
            // ---------------- V(t): reads first, then the arithmetic
#ifdef SDVAR_ATT_EXP
            if (t + 1 < ntiles && !(((SDVAR_ATT_EXP) & 2) && t > 0)) read_k(t + 1);
            if (t + D < ntiles) issue(t + D);
            __builtin_amdgcn_sched_barrier(0);
            auto mid = [&]() { __builtin_amdgcn_sched_barrier(0); if (!(((SDVAR_ATT_EXP) & 1) && t > 0)) read_v(t); __builtin_amdgcn_sched_barrier(0); };
            if (wave_active && !(((SDVAR_ATT_EXP) & 4) && t > 0)) { if ((SDVAR_ATT_EXP) & 8) { for (int i = 0; i < 16; ++i) s[i] = (float)(i + t); } softmax_tile(s, k0, vis_q, lh, M_run, l_run, o0, o1, pp, mid); }
            else mid();
#else
            if (t + 1 < ntiles) read_k(t + 1);
            if (t + D < ntiles) issue(t + D);
            __builtin_amdgcn_sched_barrier(0);
            auto mid = [&]() { __builtin_amdgcn_sched_barrier(0); read_v(t); __builtin_amdgcn_sched_barrier(0); };
            if (wave_active) softmax_tile(s, k0, vis_q, lh, M_run, l_run, o0, o1, pp, mid);
            else mid();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_PP_SLOT(); SDVAR_PP_STAMP(2);
        }
        // ---------------- M(ntiles): the last PV
        if (wave_active) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                mfma_pv<NKP>(o0, vf[j][0], pp[j]);
                mfma_pv<NKP>(o1, vf[j][1], pp[j]);
            }
        }
        SDVAR_PP_SLOT();
        if (!late) SDVAR_PP_SLOT();                                 // matches the late half's extra slot
    } else {
    const int npre = min(PP_NST, ntiles);
    for (int t = 0; t < npre; ++t) issue(t);
    // tile 0 landed: at most the other npre - 1 tiles (NKP instructions each) in flight
    if (npre >= 3) { if (NKP == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else if (npre == 2) { if (NKP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SDVAR_PP_SLOT();
    if (late) SDVAR_PP_SLOT();                                  // wave-uniform
    // V2(-1): fragments of tile 0
    read_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SDVAR_PP_SLOT();

#ifdef SDVAR_ATT_STAMPS
    wsr[1] = __builtin_amdgcn_s_memrealtime(); wsc[1] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * AKT;
        SDVAR_PP_STAMP(0);
        // ---------------- S(t)
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        if (wave_active) {
#pragma unroll
            for (int c = 0; c < 4; ++c) mfma_planes<NKP>(s, kf[c], qp[c]);
        }
        SDVAR_PP_SLOT(); SDVAR_PP_STAMP(1);
        // ---------------- V1(t)
        f16x8 pp[2][2];
        if (wave_active) {
            softmax_tile(s, k0, vis_q, lh, M_run, l_run, o0, o1, pp);
        }
        if (t + 2 < ntiles) { if (NKP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SDVAR_PP_SLOT(); SDVAR_PP_STAMP(2);
        // ---------------- PV(t)
        if (wave_active) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                mfma_pv<NKP>(o0, vf[j][0], pp[j]);
                mfma_pv<NKP>(o1, vf[j][1], pp[j]);
            }
        }
        if (t + 2 < ntiles) { if (NKP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SDVAR_PP_SLOT(); SDVAR_PP_STAMP(3);
        // ---------------- V2(t): tile t+1 into registers, tile t+3 into the stage tile t left one slot ago (both halves are past their reads of it)
        if (t + 1 < ntiles) read_tile(t + 1);
        if (t + PP_NST < ntiles) issue(t + PP_NST);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDVAR_PP_SLOT(); SDVAR_PP_STAMP(4);
    }
    if (!late) SDVAR_PP_SLOT();                                 // matches the late half's extra slot
    }
#ifdef SDVAR_ATT_STAMPS
    wsr[2] = __builtin_amdgcn_s_memrealtime(); wsc[2] = __builtin_amdgcn_s_memtime();
#endif
#undef SDVAR_PP_SLOT
#undef SDVAR_PP_STAMP

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
                const int orow = r * a.l + qi_raw, ocol = h * 64 + 4 * lh + 8 * g;
                store_planes4(a.outp, a.ops, kb_index(orow, ocol, a.R * a.l), u0, a.pfmt);
                store_planes4(a.outp, a.ops, kb_index(orow, ocol + 32, a.R * a.l), u1, a.pfmt);
            } else {
                *reinterpret_cast<f32x4*>(a.out + obase + 8 * g) = v0;
                *reinterpret_cast<f32x4*>(a.out + obase + 32 + 8 * g) = v1;
            }
        }
    }
#ifdef SDVAR_ATT_STAMPS
    if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave == 0 || wave == 4)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wsr[3] = __builtin_amdgcn_s_memrealtime(); wsc[3] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 5; ++k) a.stamps[16 * (wave >> 2) + k] = stm[k];
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.stamps[32 + k] = wsr[k]; a.stamps[36 + k] = wsc[k]; }
        }
    }
#endif
}

static int g_attn_pp_sched = -1;
void debug_set_attn_pp_sched(int v) { g_attn_pp_sched = v; }

// nkp: 2 = cache format 3 (two fp16 planes per value), 1 = cache format 4 (one fp16 plane: the fp16 KV cache)
int attention_f16x2(const float* q, const void* kc, const void* vc, int nkp, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lp,
                    int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream) {
    SDVAR_CHECK_ARG(q && kc && vc && (out || outp), "attention: null operand");
    SDVAR_CHECK_ARG(nkp == 1 || nkp == 2, "attention: %d cache planes", nkp);
    SDVAR_CHECK_ARG(n_chunk >= 1 && n_chunk <= ATTH_MAX_CHUNK, "attention: chunk of %d stages unsupported (max %d)", n_chunk, ATTH_MAX_CHUNK);
    SDVAR_CHECK_ARG(R > 0 && H > 0 && l > 0 && Ktot >= l && Ktot <= Lp, "attention: bad lengths l=%d Ktot=%d Lmax=%d", l, Ktot, Lp);
    SDVAR_CHECK_ARG(Lp % 64 == 0, "attention: the planes KV formats need Lmax %% 64 == 0 (got %d)", Lp);
    AttnHArgs a;
    a.q = q; a.kc = (const uint16_t*)kc; a.vc = (const uint16_t*)vc; a.out = out; a.outp = outp; a.ops = ops; a.pfmt = pfmt;
    a.R = R; a.H = H; a.l = l; a.Lp = Lp; a.Ktot = Ktot; a.n_chunk = n_chunk;
    for (int j = 0; j < n_chunk; ++j) {
        a.qbeg[j] = qbeg[j]; a.vis[j] = vis[j];
        SDVAR_CHECK_ARG(vis[j] >= 1 && vis[j] <= Ktot && (j == 0 ? qbeg[0] == 0 : (qbeg[j] > qbeg[j - 1] && vis[j] >= vis[j - 1])), "attention: bad stage table at %d", j);
    }
    a.qbeg[n_chunk] = l;
    a.stamps = debug_get_gemm_stamps();
    const size_t lds = ANST * (size_t)(2 * nkp * APL) * sizeof(uint16_t);       // 48 KB / 24 KB: under the 64 KB default limit
    static const int pp_min = getenv("SDVAR_ATTN_PP_MIN") ? atoi(getenv("SDVAR_ATTN_PP_MIN")) : 129;      // A/B runs: queries per (row, head) from which the 8-wave kernel runs
    if (g_attn_pp_sched < 0) { const char* e = getenv("SDVAR_ATTN_PP_SCHED"); g_attn_pp_sched = e ? atoi(e) : 1; if (g_attn_pp_sched < 0 || g_attn_pp_sched > 3) g_attn_pp_sched = 1; }
    const int pp_sched = g_attn_pp_sched;          // A/B runs and tests: 0 = the four-slot schedule (S / V1 / PV / V2); 1 (default) / 2 / 3 = two slots, ring of 4 / 6 / 8 stages
    if (l >= pp_min) {
        const size_t ldp = (pp_sched == 0 ? 3 : 2 * pp_sched + 2) * (size_t)(2 * nkp * APL) * sizeof(uint16_t);                // 48 / 64 / 96 / 128 KB (two planes)
        const dim3 grid((l + 255) / 256, H, R);
#define SDVAR_PP_LAUNCH(NK, SC) do { static LdsOptIn oi; SDVAR_LDS_OPT_IN(oi, ldp, (const void*)attention_f16x2_pp_kernel<NK, SC>);                         \
                                     hipLaunchKernelGGL((attention_f16x2_pp_kernel<NK, SC>), grid, dim3(512), ldp, stream, a); } while (0)
        switch (pp_sched) {
            case 0: if (nkp == 2) SDVAR_PP_LAUNCH(2, 0); else SDVAR_PP_LAUNCH(1, 0); break;
            case 2: if (nkp == 2) SDVAR_PP_LAUNCH(2, 2); else SDVAR_PP_LAUNCH(1, 2); break;
            case 3: if (nkp == 2) SDVAR_PP_LAUNCH(2, 3); else SDVAR_PP_LAUNCH(1, 3); break;
            default: if (nkp == 2) SDVAR_PP_LAUNCH(2, 1); else SDVAR_PP_LAUNCH(1, 1); break;
        }
#undef SDVAR_PP_LAUNCH
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    if (nkp == 2) hipLaunchKernelGGL(attention_f16x2_kernel<2>, dim3((l + 127) / 128, H, R), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(attention_f16x2_kernel<1>, dim3((l + 127) / 128, H, R), dim3(256), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar
