// fp32-accurate "NT" GEMM on the gfx950 bf16 matrix cores with split operands (bf16x3):
//     out[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] ),   X = X1 + X2 + X3,  W = W1 + W2 + W3  (exactly)
// where each Xi / Wi is a bf16 tensor holding 8 significand bits of the fp32 value (truncation split, common.h split3):
// 3 x 8 = 24 bits, so the three planes reproduce the fp32 number exactly.  Of the nine plane products the six with
// i + j <= 4 are evaluated with v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in fp32; fp32 accumulate); the
// dropped terms are <= 3 * 2^-24 relative per product.  Measured on random operands (K = 1024, fp64 reference): max error
// 1.2e-6 vs 2.5e-6 for a plain fp32 GEMM - the fp32 accumulation order dominates either way - while the matrix pipe
// spends 6 x 32 cycles per 16 k instead of 8 x 64: 2.67x the throughput of v_mfma_f32_32x32x2_f32 (gemm.hip).
//
// Operands are PLANAR and K-blocked: planes[3][K/32][rows][32] bf16 (common.h kb_index).  Weights are split once at bind time; activations are written as planes
// by their producers (ln_modulate, attention, the fc1 GELU epilogue), so the GEMM itself does no conversion work.
//
// Tiling: BM x 128 x 32 (BM = 128 / 64 / 32), 4 waves, one LDS stage (rows padded to 80 bytes: conflict-free
// ds_read_b128 of the 8-k fragments) refilled from registers that prefetch the next K-step while the MFMAs run;
// tiles are walked in 8-wide column groups inside each XCD's block range so that co-resident workgroups share both
// operand panels in the per-XCD L2.  Split-K and the epilogues are those of gemm.hip.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace sdvar {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { PEPI_BIAS = 0, PEPI_BIAS_GELU_PLANES = 1, PEPI_GATED_RES = 2, PEPI_PARTIAL = 3 };

constexpr int PBK = 32;             // k per LDS stage
constexpr int PROW = 40;            // padded row, in bf16 elements (80 bytes)
constexpr int PBN = 128;

__device__ __forceinline__ float gelu_tanh_p(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x)));
}

struct GemmPArgs {
    const uint16_t* X; const uint16_t* W;        // K-blocked planes [3][K/32][M][32], [3][K/32][N][32]
    size_t xps, wps;                             // plane strides in elements
    const float* bias; float* out; uint16_t* outp; size_t ops;
    const float* res; const float* gate;
    int M, N, K, ldo, ldres, rows_per_gate, gate_stride, split, k_per_split;
    unsigned long long* dbg_stamps;   // diagnostic builds only: per-workgroup s_memtime at entry / loop start / loop end / exit
    int dbg_same_tile;       // timing experiment only: every workgroup streams tile (0,0) (100 % L2 hits, results wrong)
    int tile_off, tile_cnt;  // 256-row kernel only: this launch covers tile ids [tile_off, tile_off + tile_cnt) (tile_cnt = 0: all); with
                             // PEPI_PARTIAL the slabs are compact [split][tile_cnt][256][128]
};

template <int BM, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmPArgs a) {
    constexpr int WM = BM / WAVES_M, WN = PBN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad wave layout");
    constexpr int XCH = (BM * 4 + 255) / 256, WCH = PBN * 4 / 256;   // 16-byte chunks per plane per thread
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[];
    uint16_t* sA = psm;                       // [3][BM][PROW]
    uint16_t* sB = psm + 3 * BM * PROW;       // [3][PBN][PROW]

    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + PBN - 1) / PBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    // 8-wide column groups: lid -> (group g, row tm, column c in group)
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);              // width of the (possibly partial) last group
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * PBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, li = lane & 31, lh = lane >> 5;

    f32x4 rx[3][XCH], rw[3][WCH];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + 256 * i, row = c >> 2, col = (c & 3) * 8;
            const int m = m0 + row;
            const bool ok = (BM * 4 >= 256 || c < BM * 4) && m < a.M;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                rx[p][i] = ok ? *reinterpret_cast<const f32x4*>(a.X + p * a.xps + ((size_t)(k0 >> 5) * a.M + m) * 32 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + 256 * i, row = c >> 2, col = (c & 3) * 8;
            const int n = n0 + row;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                rw[p][i] = (n < a.N) ? *reinterpret_cast<const f32x4*>(a.W + p * a.wps + ((size_t)(k0 >> 5) * a.N + n) * 32 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + 256 * i, row = c >> 2, col = (c & 3) * 8;
            if (BM * 4 >= 256 || c < BM * 4) {
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<f32x4*>(sA + (p * BM + row) * PROW + col) = rx[p][i];
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + 256 * i, row = c >> 2, col = (c & 3) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<f32x4*>(sB + (p * PBN + row) * PROW + col) = rw[p][i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / PBK - kt0, a.k_per_split);
    load_tile(kt0 * PBK);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt0 + kt + 1) * PBK);
        const uint16_t* pa = sA + (wm * WM + li) * PROW + 8 * lh;
        const uint16_t* pb = sB + (wn * WN + li) * PROW + 8 * lh;
#pragma unroll
        for (int s = 0; s < PBK / 16; ++s) {
            bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(pa + (p * BM + i * 32) * PROW + 16 * s);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(pb + (p * PBN + j * 32) * PROW + 16 * s);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest terms first: (1,3) (2,2) (3,1) (1,2) (2,1) (1,1)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                        // every wave is done reading this stage
        if (kt + 1 < nk) { store_tile(); __syncthreads(); }
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + li;
        if (n >= a.N) continue;
        const float bv = (EPI != PEPI_PARTIAL && a.bias) ? a.bias[n] : 0.f;
        float* outp = (EPI == PEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == PEPI_BIAS_GELU_PLANES) {
                    uint16_t p0, p1, p2;
                    split3(gelu_tanh_p(v), p0, p1, p2);
                    const size_t o = kb_index(m, n, a.M);           // the output is the next GEMM's K-blocked operand
                    a.outp[o] = p0; a.outp[a.ops + o] = p1; a.outp[2 * a.ops + o] = p2;
                } else {
                    if (EPI == PEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n];
                    outp[(size_t)m * a.ldo + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2 main-loop structure for the 128x128 tile: 8 waves (2 x 4, 64x32 outputs each), K-steps of 32 streamed global -> LDS
// by the LDS-DMA (global_load_lds_dwordx4, no staging registers) into a 3-stage ring, two K-steps in flight, ONE raw
// s_barrier per K-step and counted vmcnt waits (cdna_hip_programming.md, "Pipelining across barriers").
//   stage (48 KB) = 6 sub-arrays [128 rows][64 B]: X planes 0..2 then W planes 0..2; rows are NOT padded (one DMA
//   instruction writes 16 rows x 64 B contiguously); bank conflicts of the fragment reads are avoided by storing the
//   16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) - applied on the DMA source address and on the ds_read address.
//   Every wave issues 6 DMA instructions per K-step (rows 16w .. 16w+15 of each sub-array).
// Order per K-step t:  vmcnt(6|0) -> s_barrier  (tile t landed for all waves; everyone is done reading tile t-1)
//                      issue tile t+2 into stage (t+2)%3 (the stage tile t-1 occupied)  ->  MFMAs on stage t%3.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
constexpr int V2_STAGE = 6 * 128 * 32;          // bf16 elements per stage (48 KB)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_v2_kernel(GemmPArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[];
    constexpr int BM = 128;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + PBN - 1) / PBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * PBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;

    // DMA source pointers of this lane: row 16*wave + lane/4 of each sub-array, logical chunk (lane%4) ^ ((row>>2)&3)
    const int drow = 16 * wave + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);
    const int xrow = a.dbg_same_tile ? drow : min(m0 + drow, a.M - 1), wrow = a.dbg_same_tile ? drow : min(n0 + drow, a.N - 1);   // clamped: rows past the edge are never stored
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / PBK - kt0, a.k_per_split);
    // K-blocked: K-step t is the slab [t][rows][32].  Lane offsets are loop-invariant; K-step / plane strides stay scalar (common.h SDVAR_DMA16)
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx = (uint32_t)(xrow * 32 + 8 * dchunk) * 2u, lw = (uint32_t)(wrow * 32 + 8 * dchunk) * 2u;
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32);
    // DMA instruction q (0..5) of K-step t (relative) -> stage t % 3: q = 2p is X plane p, q = 2p + 1 is W plane p
    auto issue_one = [&](int t, int q) {
        uint16_t* st = psm + (t % 3) * V2_STAGE + swave * 512;       // + sub-array * 4096 elements
        const int p = q >> 1;
        if (q & 1) SDVAR_DMA16(lw, bw + ((size_t)t * a.N * 32 + p * a.wps) * 2, SDVAR_LDS_ADDR(st + (3 + p) * 4096));
        else SDVAR_DMA16(lx, bx + ((size_t)t * a.M * 32 + p * a.xps) * 2, SDVAR_LDS_ADDR(st + p * 4096));
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) issue_one(t, q);
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

    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.dbg_stamps) ts0 = __builtin_amdgcn_s_memtime();
    issue(0);
    if (nk > 1) issue(1);
    for (int t = 0; t < nk; ++t) {
        if (a.dbg_stamps && t == 1) ts1 = __builtin_amdgcn_s_memtime();
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool pf = (t + 2 < nk) && a.dbg_same_tile != 2;    // (== 2: timing experiment without the DMA stream)
        // LDS byte addresses of this lane's fragments in stage t % 3 (the two k16-steps differ only in the swizzled chunk)
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(psm + (t % 3) * V2_STAGE);
        const uint32_t aa0 = sb + 2 * (offa0 + ch0), aa1 = sb + 2 * (offa0 + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        // Hand-placed LDS reads and waits (the compiler's waitcnt pass falls back to lgkmcnt(0) with 18 reads in flight):
        // step-0 fragments, then step-1 fragments, wait for step 0 only, 12 MFMAs, wait for the rest, 12 MFMAs.  Both
        // waves of a SIMD belong to this workgroup and pass the barrier together, so what is not hidden inside the wave
        // is not hidden at all.
        bf16x8 fa[2][3][2], fb[2][3];
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
        SDVAR_LDS_RD(fa[0][0][0], aa0, 0);     SDVAR_LDS_RD(fb[0][2], ab0, 40960);  SDVAR_LDS_RD(fa[0][1][0], aa0, 8192);
        SDVAR_LDS_RD(fb[0][1], ab0, 32768);    SDVAR_LDS_RD(fa[0][2][0], aa0, 16384); SDVAR_LDS_RD(fb[0][0], ab0, 24576);
        SDVAR_LDS_RD(fa[0][0][1], aa0, 2048);  SDVAR_LDS_RD(fa[0][1][1], aa0, 10240); SDVAR_LDS_RD(fa[0][2][1], aa0, 18432);
        SDVAR_LDS_RD(fa[1][0][0], aa1, 0);     SDVAR_LDS_RD(fb[1][2], ab1, 40960);  SDVAR_LDS_RD(fa[1][1][0], aa1, 8192);
        SDVAR_LDS_RD(fb[1][1], ab1, 32768);    SDVAR_LDS_RD(fa[1][2][0], aa1, 16384); SDVAR_LDS_RD(fb[1][0], ab1, 24576);
        SDVAR_LDS_RD(fa[1][0][1], aa1, 2048);  SDVAR_LDS_RD(fa[1][1][1], aa1, 10240); SDVAR_LDS_RD(fa[1][2][1], aa1, 18432);
#undef SDVAR_LDS_RD
        // The 6 DMA instructions of K-step t+2 are spread between the MFMA groups: issued as one burst right after the
        // barrier they collide with every wave's fragment reads (tools/micro/mfma_lds.hip: 23.1 vs 19.8 ns per MFMA).
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);      // keep the MFMAs behind the wait (rule 18 of the CDNA guide)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][2], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][2][i], fb[s][0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][0], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pf) {
                    if (s == 0) { issue_one(t + 2, 2 * i); issue_one(t + 2, 2 * i + 1); }
                    else issue_one(t + 2, 4 + i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (a.dbg_stamps) ts2 = __builtin_amdgcn_s_memtime();
    const int n = n0 + wn * 32 + li;
    if (n < a.N) {
        const float bv = (EPI != PEPI_PARTIAL && a.bias) ? a.bias[n] : 0.f;
        float* outp = (EPI == PEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = acc[i][r] + bv;
                if (EPI == PEPI_BIAS_GELU_PLANES) {
                    uint16_t p0, p1, p2;
                    split3(gelu_tanh_p(v), p0, p1, p2);
                    const size_t o = kb_index(m, n, a.M);           // the output is the next GEMM's K-blocked operand
                    a.outp[o] = p0; a.outp[a.ops + o] = p1; a.outp[2 * a.ops + o] = p2;
                } else {
                    if (EPI == PEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n];
                    outp[(size_t)m * a.ldo + n] = v;
                }
            }
        }
    }
    if (a.dbg_stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = a.dbg_stamps + 4 * (size_t)blockIdx.x;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3: 256 x 128 workgroup tile, 8 waves (4 x 2) of 64 x 64 outputs, same LDS-DMA scheme with a 2-stage ring (72 KB per
// stage).  Measured on v2 (in-kernel stamps, tools/micro/gemm_stamps.py): 2776 cycles per K-step against 1536 of MFMA
// work, because the LDS array serves 144 KB of fragment reads plus 48 KB of DMA writes per K-step.  64x64 wave tiles
// halve the fragment reads per MFMA and the taller workgroup tile cuts the DMA bytes per flop by a quarter.
constexpr int V3_STAGE = 3 * (256 + 128) * 32;      // bf16 elements per stage: X planes [3][256][32] then W planes [3][128][32]

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_v3_kernel(GemmPArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[];
    constexpr int BM = 256;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + PBN - 1) / PBN, ntile = tiles_m * tiles_n;
    const int tcnt = a.tile_cnt > 0 ? a.tile_cnt : ntile;
    const int ks = blockIdx.x / tcnt;
    const int lid = a.tile_off + xcd_remap(blockIdx.x - ks * tcnt, tcnt);
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * BM, n0 = tn * PBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

    // DMA: wave w fills X rows [32w, 32w+32) (two 16-row groups) and W rows [16w, 16w+16) of every plane: 9 instructions
    const int r16 = lane >> 2;
    const int xr0 = 32 * wave + r16, xr1 = xr0 + 16, wr = 16 * wave + r16;
    const int cx0 = (lane & 3) ^ ((xr0 >> 2) & 3), cx1 = (lane & 3) ^ ((xr1 >> 2) & 3), cw = (lane & 3) ^ ((wr >> 2) & 3);
    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / PBK - kt0, a.k_per_split);
    // Per-lane byte offsets are loop-invariant 32-bit values; everything that moves with the K-step or the plane is wave-uniform and
    // stays in scalar registers (global_load_lds takes "SGPR base + VGPR offset"), and the LDS destination is computed from a scalar
    // wave index: the K-loop carries no vector address arithmetic (vector ALU work does not hide under MFMAs on this chip).
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx0 = (uint32_t)(min(m0 + xr0, a.M - 1) * 32 + 8 * cx0) * 2u, lx1 = (uint32_t)(min(m0 + xr1, a.M - 1) * 32 + 8 * cx1) * 2u;
    const uint32_t lw0 = (uint32_t)(min(n0 + wr, a.N - 1) * 32 + 8 * cw) * 2u;
    const char* const bx = reinterpret_cast<const char*>(a.X + (size_t)kt0 * a.M * 32);
    const char* const bw = reinterpret_cast<const char*>(a.W + (size_t)kt0 * a.N * 32);
    // DMA instruction q (0..8) of K-step t -> stage t & 1: plane p = q / 3; q % 3 = 0 / 1: X row groups, 2: W row group
    auto issue_one = [&](int t, int q) {
        uint16_t* st = psm + (t & 1) * V3_STAGE;
        const int p = q / 3, kind = q % 3;
        const char* ux = bx + ((size_t)t * a.M * 32 + p * a.xps) * 2;          // wave-uniform
        const char* uw = bw + ((size_t)t * a.N * 32 + p * a.wps) * 2;
        if (kind == 0) SDVAR_DMA16(lx0, ux, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024));
        else if (kind == 1) SDVAR_DMA16(lx1, ux, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024 + 512));
        else SDVAR_DMA16(lw0, uw, SDVAR_LDS_ADDR(st + 3 * 8192 + p * 4096 + swave * 512));
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 9; ++q) issue_one(t, q);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int offa = (wm * 64 + li) * 32, offb = 3 * 8192 + (wn * 64 + li) * 32;     // element offsets inside a stage
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);

    issue(0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 1 < nk;     // the 9 DMA instructions of K-step t+1 are spread between the MFMA groups below
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(psm + (t & 1) * V3_STAGE);
        const uint32_t aa0 = sb + 2 * (offa + ch0), aa1 = sb + 2 * (offa + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        bf16x8 fa[2][3][2], fb[2][3][2];
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
        // X plane p at +16384 p bytes, second 32-row tile at +2048; W plane p at +8192 p bytes, second tile at +2048
        SDVAR_LDS_RD(fa[0][0][0], aa0, 0);     SDVAR_LDS_RD(fb[0][2][0], ab0, 16384); SDVAR_LDS_RD(fa[0][1][0], aa0, 16384);
        SDVAR_LDS_RD(fb[0][1][0], ab0, 8192);  SDVAR_LDS_RD(fa[0][2][0], aa0, 32768); SDVAR_LDS_RD(fb[0][0][0], ab0, 0);
        SDVAR_LDS_RD(fb[0][2][1], ab0, 18432); SDVAR_LDS_RD(fb[0][1][1], ab0, 10240); SDVAR_LDS_RD(fb[0][0][1], ab0, 2048);
        SDVAR_LDS_RD(fa[0][0][1], aa0, 2048);  SDVAR_LDS_RD(fa[0][1][1], aa0, 18432); SDVAR_LDS_RD(fa[0][2][1], aa0, 34816);
        SDVAR_LDS_RD(fa[1][0][0], aa1, 0);     SDVAR_LDS_RD(fb[1][2][0], ab1, 16384); SDVAR_LDS_RD(fa[1][1][0], aa1, 16384);
        SDVAR_LDS_RD(fb[1][1][0], ab1, 8192);  SDVAR_LDS_RD(fa[1][2][0], aa1, 32768); SDVAR_LDS_RD(fb[1][0][0], ab1, 0);
        SDVAR_LDS_RD(fb[1][2][1], ab1, 18432); SDVAR_LDS_RD(fb[1][1][1], ab1, 10240); SDVAR_LDS_RD(fb[1][0][1], ab1, 2048);
        SDVAR_LDS_RD(fa[1][0][1], aa1, 2048);  SDVAR_LDS_RD(fa[1][1][1], aa1, 18432); SDVAR_LDS_RD(fa[1][2][1], aa1, 34816);
#undef SDVAR_LDS_RD
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][2][i], fb[s][0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1][i], fb[s][0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0][i], fb[s][0][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pf) {
                        const int grp = 4 * s + 2 * i + j;           // 0..7
                        issue_one(t + 1, grp);
                        if (grp == 7) issue_one(t + 1, 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + li;
        if (n >= a.N) continue;
        const float bv = (EPI != PEPI_PARTIAL && a.bias) ? a.bias[n] : 0.f;
        float* outp = (EPI == PEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
        if (EPI == PEPI_PARTIAL && a.tile_cnt > 0) {         // tail tiles of a hybrid launch: compact slab of this (slice, tile)
            float* slab = a.out + ((size_t)ks * a.tile_cnt + (lid - a.tile_off)) * (256 * 128);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wn * 64 + j * 32 + li] = acc[i][j][r];
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == PEPI_BIAS_GELU_PLANES) {
                    uint16_t p0, p1, p2;
                    split3(gelu_tanh_p(v), p0, p1, p2);
                    const size_t o = kb_index(m, n, a.M);
                    a.outp[o] = p0; a.outp[a.ops + o] = p1; a.outp[2 * a.ops + o] = p2;
                } else {
                    if (EPI == PEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n];
                    outp[(size_t)m * a.ldo + n] = v;
                }
            }
        }
    }
}

// out = epi( sum_s slab[s] + bias ) for the split-K path; the GELU variant writes planes
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_p_kernel(const float* __restrict__ ws, int split, const float* __restrict__ bias, float* out,
                                                              uint16_t* outp, size_t ops, const float* res, const float* __restrict__ gate, int M, int N,
                                                              int ldo, int ldres, int rows_per_gate, int gate_stride) {
    const int nv = N >> 2;
    const size_t total = (size_t)M * nv, slab = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), n = (int)(i % nv) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + (size_t)m * N + n);
        for (int s = 1; s < split; ++s) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(ws + s * slab + (size_t)m * N + n);
            acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3];
        }
        uint16_t pl[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (bias ? bias[n + e] : 0.f);
            if (EPI == PEPI_BIAS_GELU_PLANES) { split3(gelu_tanh_p(v), pl[0][e], pl[1][e], pl[2][e]); continue; }
            if (EPI == PEPI_GATED_RES) v = res[(size_t)m * ldres + n + e] + v * gate[(size_t)(m / rows_per_gate) * gate_stride + n + e];
            out[(size_t)m * ldo + n + e] = v;
        }
        if (EPI == PEPI_BIAS_GELU_PLANES) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint2 w;
                w.x = (uint32_t)pl[p][0] | ((uint32_t)pl[p][1] << 16); w.y = (uint32_t)pl[p][2] | ((uint32_t)pl[p][3] << 16);
                *reinterpret_cast<uint2*>(outp + p * ops + kb_index(m, n, M)) = w;
            }
        }
    }
}

// Tail tiles of a hybrid launch of the 256-row kernel (launch_v3_hybrid): out = epi( sum_s slab[s][tile] + bias ) for the tiles
// [tile_off, tile_off + tile_cnt); 4 workgroups per tile, the tile id -> (row tile, column tile) map is the kernel's.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_tiles_kernel(const float* __restrict__ ws, int split, GemmPArgs a) {
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + PBN - 1) / PBN;
    const int t = blockIdx.x >> 5, part = blockIdx.x & 31, lid = a.tile_off + t;       // 32 workgroups per tile, 8 rows each
    const int G = 8, per_group = tiles_m * G;
    const int g = lid / per_group, rem = lid - g * per_group;
    const int gw = min(G, tiles_n - g * G);
    const int tm = rem / gw, tn = g * G + rem % gw;
    const int m0 = tm * 256, n0 = tn * PBN;
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
        uint16_t pl[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (a.bias ? a.bias[n + e] : 0.f);
            if (EPI == PEPI_BIAS_GELU_PLANES) { split3(gelu_tanh_p(v), pl[0][e], pl[1][e], pl[2][e]); continue; }
            if (EPI == PEPI_GATED_RES) v = a.res[(size_t)m * a.ldres + n + e] + v * a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n + e];
            a.out[(size_t)m * a.ldo + n + e] = v;
        }
        if (EPI == PEPI_BIAS_GELU_PLANES) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                uint2 w;
                w.x = (uint32_t)pl[pp][0] | ((uint32_t)pl[pp][1] << 16); w.y = (uint32_t)pl[pp][2] | ((uint32_t)pl[pp][3] << 16);
                *reinterpret_cast<uint2*>(a.outp + pp * a.ops + kb_index(m, n, a.M)) = w;
            }
        }
    }
}

// fp32 (rows, cols) row-major -> K-blocked planes [3][cols/32][rows][32] bf16 (weights at bind time, tests)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint16_t* __restrict__ p, int rows, int cols, size_t ps) {
    const size_t n4 = (size_t)rows * cols / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (cols / 4)), k = (int)(i % (cols / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * cols + k);
        uint16_t q[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[e], q[0][e], q[1][e], q[2][e]);
        const size_t o = kb_index(row, k, rows);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint2 w;
            w.x = (uint32_t)q[j][0] | ((uint32_t)q[j][1] << 16); w.y = (uint32_t)q[j][2] | ((uint32_t)q[j][3] << 16);
            *reinterpret_cast<uint2*>(p + j * ps + o) = w;
        }
    }
}

int split_planes(const float* x, uint16_t* planes, int rows, int cols, size_t plane_stride, hipStream_t stream) {
    SDVAR_CHECK_ARG(x && planes && rows > 0 && cols > 0 && cols % 32 == 0 && plane_stride % 8 == 0, "split_planes: need cols %% 32 == 0 (rows=%d cols=%d)", rows, cols);
    const size_t blocks = ((size_t)rows * cols / 4 + 255) / 256;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, planes, rows, cols, plane_stride);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

float* splitk_workspace(size_t* floats);     // gemm.hip: the shared slab workspace

// cost-model constants (see choose_cfg_p; tools/fit_gemm_model.py on profiles/r01_e_gemm_sweep_bf16x3.jsonl: geometric-mean
// regret 1.8 %, worst case 21 %, over the d12 / d16 shapes incl. gamma = 2 chunks)
#define CM_R256 1
#define CM_R128 1
#define CM_R64 3
#define CM_R32 3
#define CM_P256 1.1
#define CM_P64 1.3
#define CM_P32 1.3
#define CM_L1 1.2
#define CM_L2 1.0
#define CM_L3 1.0
#define CM_KOVER 260.0
#define CM_FIX 1500.0
#define CM_FIXBM 20.0
#define CM_RED0 2000.0
#define CM_REDBW 5000.0

static unsigned long long* g_dbg_stamps = nullptr;
void debug_set_gemm_stamps(unsigned long long* p) { g_dbg_stamps = p; }
unsigned long long* debug_get_gemm_stamps() { return g_dbg_stamps; }
static int g_force_bm_p = 0, g_force_split_p = 0;
void debug_set_gemm_cfg_p(int bm, int split) { g_force_bm_p = bm; g_force_split_p = split; }

// same cost model as gemm.hip::choose_cfg with this kernel's constants: 6 MFMAs x 32 cycles per 16 k per 32x32 tile
static void choose_cfg_p(int M, int N, int K, size_t ws_floats, int* bm_out, int* split_out, int* tail_out, bool allow_hybrid) {
    const int nkt = K / PBK, tiles_n = (N + PBN - 1) / PBN;
    double best = 1e30; int bbm = 128, bs = 1, btail = 0;
    // per row-tile constants fitted to tools/gemm_bench.py --mode bf16x3 --sweep --dump (tools/fit_gemm_model.py):
    //   resident workgroups per CU, K-step cost factor over the MFMA time, slowdown when 1 / 2 / 3 workgroups share a CU
    const int bms[4] = {256, 128, 64, 32};
    const int resident[4] = {CM_R256, CM_R128, CM_R64, CM_R32};
    const double kfac[4] = {CM_P256, 1.0, CM_P64, CM_P32};
    const double lat[5] = {0.0, CM_L1, CM_L2, CM_L3, 1.0};
    for (int bi = 0; bi < 4; ++bi) {
        const int bm = bms[bi], res = resident[bi];
        const int tiles = ((M + bm - 1) / bm) * tiles_n;
        const double ktile = 384.0 * (bm / 32) * kfac[bi];              // 6 MFMAs x 32 cycles x 2 k16-steps per 32x32 sub-tile
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
            if (split > 1) cyc += CM_RED0 + (double)(split + 1) * M * N * 4.0 / CM_REDBW;
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
    *bm_out = bbm; *split_out = bs; *tail_out = btail;
}

static thread_local int* g_defer = nullptr;     // set per call by gemm_bf16x3_nt; thread-local: host threads may drive different model objects concurrently
static bool g_use_v2 = true;
void debug_set_gemm_v2(int on) { g_use_v2 = on != 0; }

template <int EPI>
static int launch_v2_kernel(const GemmPArgs& a, int grid, hipStream_t stream) {
    const size_t lds = 3 * (size_t)V2_STAGE * sizeof(uint16_t);      // 144 KB
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_bf16x3_v2_kernel<EPI>);
    hipLaunchKernelGGL((gemm_bf16x3_v2_kernel<EPI>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

template <int EPI>
static int launch_v3_kernel(const GemmPArgs& a, int grid, hipStream_t stream) {
    const size_t lds = 2 * (size_t)V3_STAGE * sizeof(uint16_t);      // 144 KB
    static LdsOptIn opt_in;
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_bf16x3_v3_kernel<EPI>);
    hipLaunchKernelGGL((gemm_bf16x3_v3_kernel<EPI>), dim3(grid), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

static int launch_reduce_p(const GemmPArgs& a, const float* ws, int split, int epi, hipStream_t stream) {
    const size_t total = (size_t)a.M * (a.N / 4);
    const int rgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    dim3 block(256);
    switch (epi) {
        case PEPI_BIAS: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_BIAS>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        case PEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_BIAS_GELU_PLANES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        default: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_GATED_RES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

static int launch_v3(GemmPArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + PBN - 1) / PBN);
    const int nkt = a.K / PBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmPArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        int rc = launch_v3_kernel<PEPI_PARTIAL>(p, tiles * split, stream);
        if (rc) return rc;
        if (g_defer) { *g_defer = split; return SDVAR_OK; }
        return launch_reduce_p(a, ws, split, epi, stream);
    }
    a.split = 1; a.k_per_split = nkt;
    switch (epi) {
        case PEPI_BIAS: return launch_v3_kernel<PEPI_BIAS>(a, tiles, stream);
        case PEPI_BIAS_GELU_PLANES: return launch_v3_kernel<PEPI_BIAS_GELU_PLANES>(a, tiles, stream);
        default: return launch_v3_kernel<PEPI_GATED_RES>(a, tiles, stream);
    }
}

// full rounds unsplit + the partial last round split `tail` ways along K (compact slabs) + a reduce over the tail tiles only
static int launch_v3_hybrid(GemmPArgs a, int epi, int tail, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + PBN - 1) / PBN), full = tiles / 256 * 256, remt = tiles - full;
    const int nkt = a.K / PBK;
    size_t wsf = 0;
    float* ws = splitk_workspace(&wsf);
    if (!ws) return SDVAR_ERR_HIP;
    GemmPArgs f = a;
    f.split = 1; f.k_per_split = nkt; f.tile_off = 0; f.tile_cnt = full;
    int rc;
    switch (epi) {
        case PEPI_BIAS: rc = launch_v3_kernel<PEPI_BIAS>(f, full, stream); break;
        case PEPI_BIAS_GELU_PLANES: rc = launch_v3_kernel<PEPI_BIAS_GELU_PLANES>(f, full, stream); break;
        default: rc = launch_v3_kernel<PEPI_GATED_RES>(f, full, stream); break;
    }
    if (rc) return rc;
    GemmPArgs p = a;
    p.out = ws; p.split = tail; p.k_per_split = (nkt + tail - 1) / tail; p.tile_off = full; p.tile_cnt = remt;
    rc = launch_v3_kernel<PEPI_PARTIAL>(p, remt * tail, stream);
    if (rc) return rc;
    GemmPArgs r = a;
    r.tile_off = full; r.tile_cnt = remt;
    switch (epi) {
        case PEPI_BIAS: hipLaunchKernelGGL(splitk_reduce_tiles_kernel<PEPI_BIAS>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
        case PEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL(splitk_reduce_tiles_kernel<PEPI_BIAS_GELU_PLANES>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
        default: hipLaunchKernelGGL(splitk_reduce_tiles_kernel<PEPI_GATED_RES>, dim3(32 * remt), dim3(256), 0, stream, ws, tail, r); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

template <int BM, int WAVES_M, int WAVES_N>
static int launch_p(GemmPArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + PBN - 1) / PBN);
    const size_t lds = 3 * (size_t)(BM + PBN) * PROW * sizeof(uint16_t);
    dim3 block(256);
    const bool v2 = (BM == 128) && g_use_v2;
    const int nkt = a.K / PBK;
    if (split > 1) {
        size_t wsf = 0;
        float* ws = splitk_workspace(&wsf);
        if (!ws) return SDVAR_ERR_HIP;
        GemmPArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        if (v2) { int rc = launch_v2_kernel<PEPI_PARTIAL>(p, tiles * split, stream); if (rc) return rc; }
        else { hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, WAVES_M, WAVES_N, PEPI_PARTIAL>), dim3(tiles * split), block, lds, stream, p); SDVAR_LAUNCH_CHECK(); }
        if (g_defer) { *g_defer = split; return SDVAR_OK; }
        const size_t total = (size_t)a.M * (a.N / 4);
        const int rgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        switch (epi) {
            case PEPI_BIAS: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_BIAS>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
            case PEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_BIAS_GELU_PLANES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
            default: hipLaunchKernelGGL(splitk_reduce_p_kernel<PEPI_GATED_RES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.outp, a.ops, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        }
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    a.split = 1; a.k_per_split = nkt;
    if (v2) {
        switch (epi) {
            case PEPI_BIAS: return launch_v2_kernel<PEPI_BIAS>(a, tiles, stream);
            case PEPI_BIAS_GELU_PLANES: return launch_v2_kernel<PEPI_BIAS_GELU_PLANES>(a, tiles, stream);
            default: return launch_v2_kernel<PEPI_GATED_RES>(a, tiles, stream);
        }
    }
    switch (epi) {
        case PEPI_BIAS: hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, WAVES_M, WAVES_N, PEPI_BIAS>), dim3(tiles), block, lds, stream, a); break;
        case PEPI_BIAS_GELU_PLANES: hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, WAVES_M, WAVES_N, PEPI_BIAS_GELU_PLANES>), dim3(tiles), block, lds, stream, a); break;
        default: hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, WAVES_M, WAVES_N, PEPI_GATED_RES>), dim3(tiles), block, lds, stream, a); break;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// X planes [3][M][K] (plane stride xps), W planes [3][N][K] (plane stride wps).  epi: 0 bias -> out fp32; 1 bias + GELU ->
// outp planes [3][M][N] (plane stride ops); 2 gated residual -> out fp32.
// defer: when non-null and the shape is split along K, the reduce pass is NOT launched; *defer receives the slice count and the
// slabs stay in the shared workspace ([split][M][N]) for the consumer kernel to sum (elementwise.hip PendingSplitK); 0 otherwise.
int gemm_bf16x3_nt(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops,
                   int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, int* defer,
                   hipStream_t stream) {
    g_defer = defer;
    if (defer) *defer = 0;
    SDVAR_CHECK_ARG(X && W, "gemm_bf16x3: null operand");
    SDVAR_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % PBK == 0, "gemm_bf16x3: need K %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    SDVAR_CHECK_ARG(epi >= PEPI_BIAS && epi <= PEPI_GATED_RES, "gemm_bf16x3: unknown epilogue %d", epi);
    SDVAR_CHECK_ARG(epi == PEPI_BIAS_GELU_PLANES ? (outp != nullptr && N % 4 == 0) : (out != nullptr && ldo >= N), "gemm_bf16x3: missing output");
    SDVAR_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && xps % 8 == 0 && wps % 8 == 0, "gemm_bf16x3: planes must be 16-byte aligned");
    if (epi == PEPI_GATED_RES) SDVAR_CHECK_ARG(res && gate && rows_per_gate > 0 && ldres >= N, "gemm_bf16x3: gated-residual epilogue needs res/gate");
#ifdef SDVAR_TIMING_EXPERIMENTS       // results wrong: every workgroup streams tile (0, 0)
    static const int same_tile = getenv("SDVAR_DEBUG_SAME_TILE") ? atoi(getenv("SDVAR_DEBUG_SAME_TILE")) : 0;
#else
    const int same_tile = 0;
#endif
    GemmPArgs a{X, W, xps, wps, bias, out, outp, ops, res, gate, M, N, K, ldo, ldres, rows_per_gate > 0 ? rows_per_gate : 1, gate_stride, 1, K / PBK,
                g_dbg_stamps, same_tile, 0, 0};
    size_t wsf = 0;
    (void)splitk_workspace(&wsf);
    int bm, split, tail = 0;
    static const bool no_hybrid = getenv("SDVAR_GEMM_NO_HYBRID") != nullptr;       // A/B runs only
    choose_cfg_p(M, N, K, wsf, &bm, &split, &tail, !no_hybrid);
    if (g_force_bm_p) { bm = g_force_bm_p; tail = 0; }
    static const bool trace = getenv("SDVAR_GEMM_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[gemm_bf16x3] M=%d N=%d K=%d epi=%d -> bm=%d split=%d tail=%d\n", M, N, K, epi, bm, split, tail);
    if (g_force_split_p) {
        split = g_force_split_p;
        const int nkt = K / PBK;
        if (split > nkt) split = nkt;
        while (split > 1 && (size_t)split * M * N > wsf) --split;
        const int kps = (nkt + split - 1) / split;
        split = (nkt + kps - 1) / kps;
    }
    if (bm == 256 && tail > 0) return launch_v3_hybrid(a, epi, tail, stream);
    if (bm == 256) return launch_v3(a, epi, split, stream);
    if (bm == 32) return launch_p<32, 1, 4>(a, epi, split, stream);
    if (bm == 64) return launch_p<64, 2, 2>(a, epi, split, stream);
    return launch_p<128, 2, 2>(a, epi, split, stream);
}

}  // namespace sdvar
