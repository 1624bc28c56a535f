// 3x3 / 1x1 convolution of the VQVAE decoder (models/basic_vae.py:163-226 behind vqvae.py:62-63 fhat_to_img) as an implicit
// GEMM on the bf16 matrix cores with split operands - the arithmetic contract of gemm_bf16x3.hip (fp32 values held exactly
// as three bf16 planes, six plane products, fp32 accumulate).
//
// Layouts: fp32 activations are dense channel-last rows [M][C], M = B H W, row(b, y, x) = (b H + y) W + x.  The GEMM A operand
// is the same tensor as K-blocked planes [3][C/32][G + Mp + G][32] bf16 over "padded pixel rows": a one-pixel zero frame
// around every image, prow(b, y, x) = (b (H+2) + y + 1) (W+2) + x + 1, Mp = B (H+2) (W+2), plus G >= W+3 zero guard rows on
// both ends.  With that layout the A rows of tap (dy, dx) and channel block cb of a 3x3 convolution are the rows of the
// centre tap shifted by dy (W+2) + dx: every lane of the LDS-DMA computes prow(m) once and adds a wave-uniform offset per
// K-step - no im2col, no bounds tests (the frame is real zeros), and only the B H W real pixels are computed: 256^2 x 8
// images = 2048 row tiles = exactly 8 rounds of the 256 CUs.
// Weights (Cout, Cin, kh, kw) are re-packed once as planes [3][taps Cin / 32][Cout][32] with k = tap Cin + cin.
//
// Kernel: 256 x 160 workgroup tile (every decoder width - 160, 320, 640, 1920 - is a multiple of 160), 8 waves, each
// 32 rows x 160 columns (5 MFMA column tiles: one A fragment feeds 5 x 6 MFMAs), K-step 32, 2-stage LDS-DMA ring of
// 78 KB, XOR-swizzled 64-byte rows (conflict-free ds_read_b128), hand-placed fragment reads with counted lgkmcnt, the
// DMA instructions of the next K-step spread between the MFMA groups (as gemm_bf16x3_v3_kernel).
#include <stdlib.h>

#include "common.h"

// result-corrupting timing switches (SDVAR_CONV_DBG) exist only in -DSDVAR_TIMING_EXPERIMENTS builds (make EXTRA=-DSDVAR_TIMING_EXPERIMENTS)
#ifdef SDVAR_TIMING_EXPERIMENTS
#define SDVAR_CDBG(a, bit) ((a).dbg & (bit))
#else
#define SDVAR_CDBG(a, bit) 0
#endif

namespace sdvar {
static int g_conv_pp_override = -1;
void debug_set_conv_pp(int v) { g_conv_pp_override = v; }


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

enum { CEPI_BIAS = 0, CEPI_BIAS_RES = 1, CEPI_PARTIAL = 2 };
constexpr int CBM = 256, CBN = 160;
constexpr int CSTAGE = 3 * (CBM + CBN) * 32;        // bf16 elements per stage: X planes [3][256][32] then W planes [3][160][32]
constexpr int CSTAGE_H = 2 * (CBM + CBN) * 32;      // f16x2 kernel: X planes [2][256][32] then W planes [2][160][32] (52 KB), 3-stage ring

struct ConvArgs {
    const uint16_t* X; const uint16_t* W;          // planes; X rows include the guards
    size_t xps, wps;                               // plane strides (elements)
    size_t x_rows;                                 // rows per channel block of X (G + M + G)
    int x_row0;                                    // G
    const float* wsi;                              // f16x2 only: device scalar 2^-S undoing the weight scale (gemm_f16x2.hip), null = 1
    const float* bias; const float* res; float* out;
    int M, N, cb, taps, ih, iw, ldo, split, k_per_split;   // cb = Cin / 32; K-steps = taps * cb; (ih, iw): image size, M = B ih iw
    double* gn_part; int cpg;                      // optional GroupNorm partial sums of the OUTPUT: [M/256][32 groups][sum, sumsq], cpg = N / 32
    int up_phase;                                  // >= 0: fused nearest-2x up-sampling (taps = 4): blockIdx.y = output phase (py, px) = (y >> 1, y & 1)
    size_t w_phase_stride;                         //       weight planes of phase p at W + p * w_phase_stride
    int dbg;                                       // timing experiments (SDVAR_CONV_DBG, results wrong): bit 0 = no epilogue, bit 1 = no GroupNorm statistics, bit 2 = no output stores
};

// Shared epilogue of both convolution kernels: (acc * wsi) + bias (+ residual), the up-sampling phase scatter, split-K slabs, and the
// fused GroupNorm statistics of the tile just written.
template <int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[5], float wsi, int m0, int n0, int tm, int ks, int phase, int wave, int li, int lh,
                                              int tid, uint16_t* csm) {
    // epilogue.  The residual is fetched for the whole wave tile first (80 independent loads in flight: the operand
    // fragments are dead by now), then added and stored: interleaved load -> add -> store chains cost 35 us per workgroup.
    float rv[5][16];
    if (EPI == CEPI_BIAS_RES) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int n = n0 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                rv[j][r] = (n < a.N && m < a.M) ? a.res[(size_t)m * a.ldo + n] : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int n = n0 + j * 32 + li;
        if (n >= a.N) continue;
        const float bv = (EPI != CEPI_PARTIAL && a.bias) ? a.bias[n] : 0.f;
        float* outp = (EPI == CEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= a.M) continue;
            float v = acc[j][r] * wsi + bv;
            if (EPI == CEPI_BIAS_RES) v += rv[j][r];
            size_t orow = (size_t)m;
            if (EPI != CEPI_PARTIAL && a.up_phase >= 0) {         // pixel (b, y, x) of the input grid -> (b, 2y + py, 2x + px) of the output grid
                const int hw = a.ih * a.iw, b = m / hw, rem = m - b * hw, y = rem / a.iw, x = rem - y * a.iw;
                orow = ((size_t)(b * 2 * a.ih + 2 * y + (phase >> 1)) * (2 * a.iw) + 2 * x + (phase & 1));
            }
            if (!SDVAR_CDBG(a, 4)) outp[orow * a.ldo + n] = v;
            acc[j][r] = v;
        }
    }
    // GroupNorm statistics of the tile just written (the next layer's norm): per column sum / sum of squares over the 256 rows
    // (all of one image: the host enables this only when H W is a multiple of 256), reduced lane -> wave -> workgroup through
    // LDS in a fixed order, then per group in fp64.  Saves a full read of the activation tensor per normalisation.
    if (EPI != CEPI_PARTIAL && a.gn_part && !SDVAR_CDBG(a, 2)) {
        float* red = reinterpret_cast<float*>(csm);          // [2][8][160]
        __syncthreads();                                     // every wave is done with the operand stages
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float s1 = 0.f, s2 = 0.f;
            if (n0 + j * 32 + li < a.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1 += acc[j][r]; s2 += acc[j][r] * acc[j][r]; }
            }
            const auto w1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
            const auto w2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
            if (lh == 0) {
                red[wave * 160 + j * 32 + li] = __uint_as_float(w1[0]) + __uint_as_float(w1[1]);
                red[1280 + wave * 160 + j * 32 + li] = __uint_as_float(w2[0]) + __uint_as_float(w2[1]);
            }
        }
        __syncthreads();
        const int gl = tid, col0 = gl * a.cpg;
        if (col0 < CBN && n0 + col0 < a.N) {
            double a1 = 0.0, a2 = 0.0;
            for (int w = 0; w < 8; ++w)
                for (int c = 0; c < a.cpg; ++c) { a1 += (double)red[w * 160 + col0 + c]; a2 += (double)red[1280 + w * 160 + col0 + c]; }
            double* o = a.gn_part + ((size_t)(a.up_phase >= 0 ? 4 * tm + phase : tm) * 32 + (n0 + col0) / a.cpg) * 2;
            o[0] = a1; o[1] = a2;
        }
    }
}

// The same epilogue for the accumulators of v_mfma_f32_16x16x32_f16 with W as the A operand (conv_f16x2_kernel<EPI, 2>): acc[mt][nt] (f32x4) holds row
// m0 + 32 wave + 16 mt + (lane & 15), columns n0 + 16 nt + 4 (lane >> 4) .. + 3 - four CONSECUTIVE columns of one pixel row per register group, so bias, residual
// and output move 16 bytes per access (the 32x32x16 layout above: 4 bytes, 80 store instructions per wave against 20) and a lane owns 2 pixel rows, not 16 (two
// index divisions for the up-sampling scatter).  The GroupNorm column sums run over the 16 lanes of a DPP row (4 row_shr adds per value) before the LDS stage.
typedef float cf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float dpp_row_sum16(float v) {          // lane 15 of every 16-lane row ends up with the row's sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));      // row_shr:1, out-of-row lanes read 0
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));      // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));      // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));      // row_shr:8
    return v;
}
template <int EPI>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs& a, cf32x4 (&acc)[2][10], float wsi, int m0, int n0, int tm, int ks, int phase, int wave, int l15, int lq,
                                                int tid, uint16_t* csm) {
    int mrow[2]; size_t orow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wave * 32 + 16 * mt + l15;
        mrow[mt] = m; orow[mt] = (size_t)m;
        if (EPI != CEPI_PARTIAL && a.up_phase >= 0 && m < a.M) {          // pixel (b, y, x) of the input grid -> (b, 2y + py, 2x + px) of the output grid
            const int hw = a.ih * a.iw, b = m / hw, rem = m - b * hw, y = rem / a.iw, x = rem - y * a.iw;
            orow[mt] = ((size_t)(b * 2 * a.ih + 2 * y + (phase >> 1)) * (2 * a.iw) + 2 * x + (phase & 1));
        }
    }
    cf32x4 rv[2][10];
    if (EPI == CEPI_BIAS_RES) {                      // the whole residual tile of the lane in flight first (20 independent 16-byte loads)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 10; ++nt) {
                const int n = n0 + 16 * nt + 4 * lq;
                rv[mt][nt] = (n < a.N && mrow[mt] < a.M) ? *reinterpret_cast<const cf32x4*>(a.res + (size_t)mrow[mt] * a.ldo + n) : cf32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    float* outp = (EPI == CEPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
        const int n = n0 + 16 * nt + 4 * lq;
        if (n >= a.N) continue;                      // N % 4 == 0 (host): a group of four columns is inside or outside as a whole
        const cf32x4 bv = (EPI != CEPI_PARTIAL && a.bias) ? *reinterpret_cast<const cf32x4*>(a.bias + n) : cf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (mrow[mt] >= a.M) continue;
            cf32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[mt][nt][e] * wsi + bv[e]; if (EPI == CEPI_BIAS_RES) v[e] += rv[mt][nt][e]; }
            if (!SDVAR_CDBG(a, 4)) *reinterpret_cast<cf32x4*>(outp + orow[mt] * a.ldo + n) = v;
            acc[mt][nt] = v;
        }
    }
    if (EPI != CEPI_PARTIAL && a.gn_part && !SDVAR_CDBG(a, 2)) {          // GroupNorm statistics of the tile just written (M % 256 == 0 when enabled: every row is real)
        float* red = reinterpret_cast<float*>(csm);          // [2][8][160]
        __syncthreads();                                     // every wave is done with the operand stages
#pragma unroll
        for (int nt = 0; nt < 10; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = n0 + 16 * nt + 4 * lq < a.N;
                const float x0 = in ? acc[0][nt][e] : 0.f, x1 = in ? acc[1][nt][e] : 0.f;
                const float s1 = dpp_row_sum16(x0 + x1), s2 = dpp_row_sum16(x0 * x0 + x1 * x1);
                if (l15 == 15) { red[wave * 160 + 16 * nt + 4 * lq + e] = s1; red[1280 + wave * 160 + 16 * nt + 4 * lq + e] = s2; }
            }
        __syncthreads();
        const int gl = tid, col0 = gl * a.cpg;
        if (col0 < CBN && n0 + col0 < a.N) {
            double a1 = 0.0, a2 = 0.0;
            for (int w = 0; w < 8; ++w)
                for (int c = 0; c < a.cpg; ++c) { a1 += (double)red[w * 160 + col0 + c]; a2 += (double)red[1280 + w * 160 + col0 + c]; }
            double* o = a.gn_part + ((size_t)(a.up_phase >= 0 ? 4 * tm + phase : tm) * 32 + (n0 + col0) / a.cpg) * 2;
            o[0] = a1; o[1] = a2;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void conv_bf16x3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t csm[];
    const int tiles_m = (a.M + CBM - 1) / CBM, tiles_n = (a.N + CBN - 1) / CBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;          // the column tiles of one row tile are neighbours: they share the A panel in L2
    const int m0 = tm * CBM, n0 = tn * CBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int phase = a.up_phase >= 0 ? (int)blockIdx.y : 0;
    const uint16_t* Wp = a.W + (size_t)phase * a.w_phase_stride;

    // DMA: per plane wave w fills X rows [32w, 32w+32) (two 16-row groups) and W rows [16w, 16w+16) (+ [128 + 16w, ..) for w < 2)
    const int r16 = lane >> 2;
    const int xr0 = 32 * wave + r16, xr1 = xr0 + 16, wr0 = 16 * wave + r16, wr1 = 128 + wr0;
    const int cx0 = (lane & 3) ^ ((xr0 >> 2) & 3), cx1 = (lane & 3) ^ ((xr1 >> 2) & 3), cw0 = (lane & 3) ^ ((wr0 >> 2) & 3), cw1 = (lane & 3) ^ ((wr1 >> 2) & 3);
    const int nkt = a.taps * a.cb;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(nkt - kt0, a.k_per_split);
    const int w2 = a.iw + 2;
    auto prow = [&](int m) {                    // dense pixel index -> padded pixel row of the planes
        const int hw = a.ih * a.iw, b = m / hw, rem = m - b * hw, y = rem / a.iw, x = rem - y * a.iw;
        return (size_t)(b * (a.ih + 2) + y + 1) * w2 + x + 1;
    };
    // per-lane byte offsets (loop-invariant, < 4 GB inside one channel block) + wave-uniform bases per K-step (common.h SDVAR_DMA16)
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx0 = (uint32_t)((a.x_row0 + prow(min(m0 + xr0, a.M - 1))) * 32 + 8 * cx0) * 2u;
    const uint32_t lx1 = (uint32_t)((a.x_row0 + prow(min(m0 + xr1, a.M - 1))) * 32 + 8 * cx1) * 2u;
    const uint32_t lw0 = (uint32_t)(min(n0 + wr0, a.N - 1) * 32 + 8 * cw0) * 2u, lw1 = (uint32_t)(min(n0 + wr1, a.N - 1) * 32 + 8 * cw1) * 2u;
    const bool two_w = swave < 2;
    // K-step kb = (tap, channel block): X base = (cblk x_rows + tap shift) rows, W base = kb N rows.  (tap, cblk, tap row, tap column) are
    // advanced incrementally: a division per K-step is emulated on the vector ALU (~200 instructions, which do not hide under the MFMAs).
    const char* ux = nullptr; const char* uw = nullptr;
    const int tw = (a.taps == 9) ? 3 : 2;                       // taps per kernel row (3x3, or the 2x2 phase kernels)
    int sc = kt0 % a.cb, stap = kt0 / a.cb, sty = stap / tw, stx = stap - sty * tw, skb = kt0;        // state of the NEXT step to set up
    auto set_next = [&]() {
        const int shift = (a.taps == 9) ? (sty - 1) * w2 + (stx - 1) : (a.taps == 4) ? (sty - 1 + (phase >> 1)) * w2 + (stx - 1 + (phase & 1)) : 0;
        ux = reinterpret_cast<const char*>(a.X) + ((long long)sc * (long long)a.x_rows + shift) * 64;
        uw = reinterpret_cast<const char*>(Wp) + (size_t)skb * a.N * 64;
        ++skb;
        if (++sc == a.cb) { sc = 0; if (++stx == tw) { stx = 0; ++sty; } }
    };
    // DMA instruction q of the current step -> stage st: q in [0, 9): plane q / 3, kind q % 3 (X group 0, X group 1, W group 0); q in [9, 12): W group 1 of plane q - 9
    auto issue_one = [&](uint16_t* st, int q) {
        if (q < 9) {
            const int p = q / 3, kind = q % 3;
            if (kind == 0) SDVAR_DMA16(lx0, ux + p * a.xps * 2, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024));
            else if (kind == 1) SDVAR_DMA16(lx1, ux + p * a.xps * 2, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024 + 512));
            else SDVAR_DMA16(lw0, uw + p * a.wps * 2, SDVAR_LDS_ADDR(st + 3 * 8192 + p * 5120 + swave * 512));
        } else if (two_w) {
            const int p = q - 9;
            SDVAR_DMA16(lw1, uw + p * a.wps * 2, SDVAR_LDS_ADDR(st + 3 * 8192 + p * 5120 + 4096 + swave * 512));
        }
    };

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int offa = (wave * 32 + li) * 32, offb = 3 * 8192 + li * 32;       // element offsets inside a stage
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);

    set_next();
#pragma unroll
    for (int q = 0; q < 12; ++q) issue_one(csm, q);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 1 < nk;
        if (pf) set_next();
        uint16_t* nst = csm + ((t + 1) & 1) * CSTAGE;
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(csm + (t & 1) * CSTAGE);
        const uint32_t aa0 = sb + 2 * (offa + ch0), aa1 = sb + 2 * (offa + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        bf16x8 fa[2][3], fb[2][3][5];
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
        // X plane p at +16384 p bytes; W plane p at +10240 p bytes, column tile j at +2048 j
#define SDVAR_RD_B(s, ab, p, pb) SDVAR_LDS_RD(fb[s][p][0], ab, pb); SDVAR_LDS_RD(fb[s][p][1], ab, pb + 2048); SDVAR_LDS_RD(fb[s][p][2], ab, pb + 4096); \
                                 SDVAR_LDS_RD(fb[s][p][3], ab, pb + 6144); SDVAR_LDS_RD(fb[s][p][4], ab, pb + 8192)
        // sub-step 0: 18 reads, then the first 12 of sub-step 1 (lgkmcnt is a 4-bit counter: at most 15 may be waited on)
        SDVAR_LDS_RD(fa[0][0], aa0, 0); SDVAR_LDS_RD(fa[0][1], aa0, 16384); SDVAR_LDS_RD(fa[0][2], aa0, 32768);
        SDVAR_RD_B(0, ab0, 2, 20480); SDVAR_RD_B(0, ab0, 1, 10240); SDVAR_RD_B(0, ab0, 0, 0);
        SDVAR_LDS_RD(fa[1][0], aa1, 0); SDVAR_LDS_RD(fa[1][1], aa1, 16384); SDVAR_LDS_RD(fa[1][2], aa1, 32768);
        SDVAR_RD_B(1, ab1, 2, 20480);
        SDVAR_LDS_RD(fb[1][1][0], ab1, 10240); SDVAR_LDS_RD(fb[1][1][1], ab1, 12288); SDVAR_LDS_RD(fb[1][1][2], ab1, 14336); SDVAR_LDS_RD(fb[1][1][3], ab1, 16384);
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (s == 1 && j == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0], fb[s][2][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1], fb[s][1][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][2], fb[s][0][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0], fb[s][1][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][1], fb[s][0][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][0], fb[s][0][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0 && j == 0) {          // the last 6 fragment reads of sub-step 1, behind the first MFMA group
                    SDVAR_LDS_RD(fb[1][1][4], ab1, 18432);
                    SDVAR_RD_B(1, ab1, 0, 0);
                }
                if (pf) {
                    const int grp = 5 * s + j;      // 0..9: 12 DMA instructions of K-step t+1
                    issue_one(nst, grp);
                    if (grp >= 8) issue_one(nst, grp + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef SDVAR_RD_B
#undef SDVAR_LDS_RD
    }

    conv_epilogue<EPI>(a, acc, 1.0f, m0, n0, tm, ks, phase, wave, li, lh, tid, csm);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same implicit GEMM on f16x2 operands (gemm_f16x2.hip: two fp16 planes, three products, weights scaled by 2^S): half the matrix
// work per K-step (30 MFMAs per wave), so the ring has three stages and keeps two K-steps of LDS-DMA in flight.
//   stage = X planes [2][256][32] then W planes [2][160][32]; per K-step wave w issues X rows [32w, 32w+32) of both planes (4 instructions)
//   and W rows [16w, 16w+16) (2), waves 0 and 1 also W rows [128 + 16w, ...) (2 more): the counted vmcnt wait is per wave.
// PP = 1 (default): the two waves of every SIMD alternate between the matrix pipe and LDS / DMA (the schedule of gemm_f16x2_v4_kernel): a K-step is
//     slot:       4t     4t+1    4t+2    4t+3
//     waves 0-3:  L0(t)  M0(t)   L1(t)   M1(t)        Ls = the 12 fragment reads of k16 step s + 4 of the 8 DMA slots of K-step t + 2
//     waves 4-7:  M1(t-1) L0(t)  M0(t)   L1(t)        Ms = its 15 MFMAs (5 column tiles x 3 plane products)
// with an s_barrier between slots.  K-step t + 1 is read from slot 4t + 4 on: every wave waits for its own share (counted vmcnt) at the end of slot 4t + 3.
// PP = 0: the round-2 loop (one barrier per K-step, every wave reads, then multiplies).
template <int EPI, int PP>
__global__ __launch_bounds__(512, 2) void conv_f16x2_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t csm[];
    const int tiles_m = (a.M + CBM - 1) / CBM, tiles_n = (a.N + CBN - 1) / CBN, ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int m0 = tm * CBM, n0 = tn * CBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int phase = a.up_phase >= 0 ? (int)blockIdx.y : 0;
    const uint16_t* Wp = a.W + (size_t)phase * a.w_phase_stride;

    const int r16 = lane >> 2;
    const int xr0 = 32 * wave + r16, xr1 = xr0 + 16, wr0 = 16 * wave + r16, wr1 = 128 + wr0;
    const int cx0 = (lane & 3) ^ ((xr0 >> 2) & 3), cx1 = (lane & 3) ^ ((xr1 >> 2) & 3), cw0 = (lane & 3) ^ ((wr0 >> 2) & 3), cw1 = (lane & 3) ^ ((wr1 >> 2) & 3);
    const int nkt = a.taps * a.cb;
    const int kt0 = ks * a.k_per_split;
    const int nk = min(nkt - kt0, a.k_per_split);
    const int w2 = a.iw + 2;
    auto prow = [&](int m) {
        const int hw = a.ih * a.iw, b = m / hw, rem = m - b * hw, y = rem / a.iw, x = rem - y * a.iw;
        return (size_t)(b * (a.ih + 2) + y + 1) * w2 + x + 1;
    };
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lx0 = (uint32_t)((a.x_row0 + prow(min(m0 + xr0, a.M - 1))) * 32 + 8 * cx0) * 2u;
    const uint32_t lx1 = (uint32_t)((a.x_row0 + prow(min(m0 + xr1, a.M - 1))) * 32 + 8 * cx1) * 2u;
    const uint32_t lw0 = (uint32_t)(min(n0 + wr0, a.N - 1) * 32 + 8 * cw0) * 2u, lw1 = (uint32_t)(min(n0 + wr1, a.N - 1) * 32 + 8 * cw1) * 2u;
    const bool two_w = swave < 2;
    const char* ux = nullptr; const char* uw = nullptr;
    const int tw = (a.taps == 9) ? 3 : 2;
    int sc = kt0 % a.cb, stap = kt0 / a.cb, sty = stap / tw, stx = stap - sty * tw, skb = kt0;
    auto set_next = [&]() {
        const int shift = (a.taps == 9) ? (sty - 1) * w2 + (stx - 1) : (a.taps == 4) ? (sty - 1 + (phase >> 1)) * w2 + (stx - 1 + (phase & 1)) : 0;
        ux = reinterpret_cast<const char*>(a.X) + ((long long)sc * (long long)a.x_rows + shift) * 64;
        uw = reinterpret_cast<const char*>(Wp) + (size_t)skb * a.N * 64;
        ++skb;
        if (++sc == a.cb) { sc = 0; if (++stx == tw) { stx = 0; ++sty; } }
    };
    // DMA instruction q of the step set up last -> stage st: q in [0, 6): plane q / 3, kind q % 3 (X group 0, X group 1, W group 0); q in [6, 8): W group 1 of plane q - 6
    auto issue_one = [&](uint16_t* st, int q) {
        if (q < 6) {
            const int p = q / 3, kind = q % 3;
            if (kind == 0) SDVAR_DMA16(lx0, ux + p * a.xps * 2, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024));
            else if (kind == 1) SDVAR_DMA16(lx1, ux + p * a.xps * 2, SDVAR_LDS_ADDR(st + p * 8192 + swave * 1024 + 512));
            else SDVAR_DMA16(lw0, uw + p * a.wps * 2, SDVAR_LDS_ADDR(st + 2 * 8192 + p * 5120 + swave * 512));
        } else if (two_w) {
            const int p = q - 6;
            SDVAR_DMA16(lw1, uw + p * a.wps * 2, SDVAR_LDS_ADDR(st + 2 * 8192 + p * 5120 + 4096 + swave * 512));
        }
    };

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int sw = (li >> 2) & 3;
    const int offa = (wave * 32 + li) * 32, offb = 2 * 8192 + li * 32;       // element offsets inside a stage
    const int ch0 = 8 * ((0 + lh) ^ sw), ch1 = 8 * ((2 + lh) ^ sw);

    set_next();
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_one(csm, q);
    if (nk > 1) {
        set_next();
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_one(csm + CSTAGE_H, q);
    }
    if constexpr (PP == 2) {
        // the ping-pong schedule of PP = 1 on v_mfma_f32_16x16x32_f16 (the shape the chip holds the higher clock on: profiles/r03_o_mfma_shape_f16.log): a fragment is
        // 16 rows x 32 k (one ds_read_b128: row lane & 15, chunk lane >> 4), so the K-step splits by COLUMN tiles: slot L0 = the 4 X fragments + W tiles 0-4 (14 reads)
        // + DMA slots 0-3, M0 = 2 x 5 x 3 = 30 MFMAs (480 matrix-pipe cycles), L1 = W tiles 5-9 (10 reads) + DMA slots 4-7, M1 = 30 MFMAs.
        const int late = __builtin_amdgcn_readfirstlane(wave >> 2);
        const int l15 = lane & 15, lq = lane >> 4;
        auto wait_next = [&](int t) {
            if (t + 2 < nk) { if (two_w) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#define SDVAR_C_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
#define SDVAR_C_MFMA16(J0)                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
                acc16[i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][0], fx[i][1], acc16[i][(J0) + j], 0, 0, 0);   \
                acc16[i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][1], fx[i][0], acc16[i][(J0) + j], 0, 0, 0);   \
                acc16[i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j][0], fx[i][0], acc16[i][(J0) + j], 0, 0, 0);   \
            }
        cf32x4 acc16[2][10];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 10; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
        // fragment addresses inside a stage (bytes): X plane p at + 16384 p, row tile i at + 1024 i; W planes behind the X planes (+ 32768), plane p at + 10240 p, column tile j at + 1024 j
        const uint32_t fro = (uint32_t)(l15 * 64 + 16 * (lq ^ ((l15 >> 2) & 3)));
        const uint32_t fxo = (uint32_t)(wave * 32 * 64) + fro, fwo = 32768u + fro;
        wait_next(-1);
        SDVAR_C_SLOT();
        if (late) SDVAR_C_SLOT();
        f16x8 fx[2][2], fw[5][2];                // [row tile][plane], [column tile of the half][plane]
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            const bool pf = t + 2 < nk;
            uint16_t* nst = csm + ((t + 2) % 3) * CSTAGE_H;
            const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(csm + (t % 3) * CSTAGE_H);
            const uint32_t xa = sb + fxo, wa = sb + fwo;
            // ---- L0(t)
            SDVAR_LDS_RD(fx[0][0], xa, 0); SDVAR_LDS_RD(fx[0][1], xa, 16384); SDVAR_LDS_RD(fx[1][0], xa, 1024); SDVAR_LDS_RD(fx[1][1], xa, 17408);
            SDVAR_LDS_RD(fw[0][0], wa, 0);    SDVAR_LDS_RD(fw[0][1], wa, 10240); SDVAR_LDS_RD(fw[1][0], wa, 1024); SDVAR_LDS_RD(fw[1][1], wa, 11264);
            SDVAR_LDS_RD(fw[2][0], wa, 2048); SDVAR_LDS_RD(fw[2][1], wa, 12288); SDVAR_LDS_RD(fw[3][0], wa, 3072); SDVAR_LDS_RD(fw[3][1], wa, 13312);
            SDVAR_LDS_RD(fw[4][0], wa, 4096); SDVAR_LDS_RD(fw[4][1], wa, 14336);
            if (pf) {
                set_next();
#pragma unroll
                for (int q = 0; q < 4; ++q) issue_one(nst, q);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_C_SLOT();
            SDVAR_C_MFMA16(0);                   // M0(t): column tiles 0-4
            SDVAR_C_SLOT();
            // ---- L1(t)
            SDVAR_LDS_RD(fw[0][0], wa, 5120); SDVAR_LDS_RD(fw[0][1], wa, 15360); SDVAR_LDS_RD(fw[1][0], wa, 6144); SDVAR_LDS_RD(fw[1][1], wa, 16384);
            SDVAR_LDS_RD(fw[2][0], wa, 7168); SDVAR_LDS_RD(fw[2][1], wa, 17408); SDVAR_LDS_RD(fw[3][0], wa, 8192); SDVAR_LDS_RD(fw[3][1], wa, 18432);
            SDVAR_LDS_RD(fw[4][0], wa, 9216); SDVAR_LDS_RD(fw[4][1], wa, 19456);
            if (pf) {
#pragma unroll
                for (int q = 4; q < 8; ++q) issue_one(nst, q);
            }
            if (late) wait_next(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_C_SLOT();
            SDVAR_C_MFMA16(5);                   // M1(t): column tiles 5-9
            if (!late) wait_next(t);
            SDVAR_C_SLOT();
        }
        if (!late) SDVAR_C_SLOT();
#undef SDVAR_C_MFMA16
#undef SDVAR_LDS_RD
#undef SDVAR_C_SLOT
        if (SDVAR_CDBG(a, 1)) return;
        conv_epilogue16<EPI>(a, acc16, a.wsi ? *a.wsi : 1.0f, m0, n0, tm, ks, phase, wave, l15, lq, tid, csm);
        return;
    } else
    if (PP) {
        const int late = __builtin_amdgcn_readfirstlane(wave >> 2);
        auto wait_next = [&](int t) {          // K-step t + 1 landed: K-step t + 2 (8 instructions for waves 0-1, 6 for the others), if requested, may stay in flight
            if (t + 2 < nk) { if (two_w) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#define SDVAR_C_SLOT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
#define SDVAR_RD_B(ab, p, pb) SDVAR_LDS_RD(fb[p][0], ab, pb); SDVAR_LDS_RD(fb[p][1], ab, pb + 2048); SDVAR_LDS_RD(fb[p][2], ab, pb + 4096); \
                              SDVAR_LDS_RD(fb[p][3], ab, pb + 6144); SDVAR_LDS_RD(fb[p][4], ab, pb + 8192)
#define SDVAR_C_MFMA()                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                 \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[0][j], acc[j], 0, 0, 0);          \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[1][j], acc[j], 0, 0, 0);          \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[0][j], acc[j], 0, 0, 0);          \
        }
        wait_next(-1);
        SDVAR_C_SLOT();
        if (late) SDVAR_C_SLOT();
        f16x8 fa[2], fb[2][5];                  // [plane], [plane][column tile] of the current k16 step
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            const bool pf = t + 2 < nk;
            uint16_t* nst = csm + ((t + 2) % 3) * CSTAGE_H;
            const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(csm + (t % 3) * CSTAGE_H);
            const uint32_t aa0 = sb + 2 * (offa + ch0), aa1 = sb + 2 * (offa + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
            // ---- L0(t)
            SDVAR_LDS_RD(fa[0], aa0, 0); SDVAR_LDS_RD(fa[1], aa0, 16384);
            SDVAR_RD_B(ab0, 0, 0); SDVAR_RD_B(ab0, 1, 10240);
            if (pf) {
                set_next();
#pragma unroll
                for (int q = 0; q < 4; ++q) issue_one(nst, q);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_C_SLOT();
            SDVAR_C_MFMA();                      // M0(t)
            SDVAR_C_SLOT();
            // ---- L1(t)
            SDVAR_LDS_RD(fa[0], aa1, 0); SDVAR_LDS_RD(fa[1], aa1, 16384);
            SDVAR_RD_B(ab1, 0, 0); SDVAR_RD_B(ab1, 1, 10240);
            if (pf) {
#pragma unroll
                for (int q = 4; q < 8; ++q) issue_one(nst, q);
            }
            if (late) wait_next(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDVAR_C_SLOT();
            SDVAR_C_MFMA();                      // M1(t)
            if (!late) wait_next(t);
            SDVAR_C_SLOT();
        }
        if (!late) SDVAR_C_SLOT();
#undef SDVAR_C_MFMA
#undef SDVAR_RD_B
#undef SDVAR_LDS_RD
#undef SDVAR_C_SLOT
    } else
    for (int t = 0; t < nk; ++t) {
        // K-step t has landed when only the newest K-step's instructions (8 for waves 0-1, 6 for the others) are still in flight
        if (t + 1 < nk) { if (two_w) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 2 < nk;
        if (pf) set_next();
        uint16_t* nst = csm + ((t + 2) % 3) * CSTAGE_H;
        const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(csm + (t % 3) * CSTAGE_H);
        const uint32_t aa0 = sb + 2 * (offa + ch0), aa1 = sb + 2 * (offa + ch1), ab0 = sb + 2 * (offb + ch0), ab1 = sb + 2 * (offb + ch1);
        f16x8 fa[2][2], fb[2][2][5];
#define SDVAR_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
        // X plane p at +16384 p bytes; W plane p at +10240 p bytes (behind the X planes), column tile j at +2048 j
#define SDVAR_RD_B(s, ab, p, pb) SDVAR_LDS_RD(fb[s][p][0], ab, pb); SDVAR_LDS_RD(fb[s][p][1], ab, pb + 2048); SDVAR_LDS_RD(fb[s][p][2], ab, pb + 4096); \
                                 SDVAR_LDS_RD(fb[s][p][3], ab, pb + 6144); SDVAR_LDS_RD(fb[s][p][4], ab, pb + 8192)
        SDVAR_LDS_RD(fa[0][0], aa0, 0); SDVAR_LDS_RD(fa[0][1], aa0, 16384);
        SDVAR_RD_B(0, ab0, 0, 0); SDVAR_RD_B(0, ab0, 1, 10240);
        SDVAR_LDS_RD(fa[1][0], aa1, 0); SDVAR_LDS_RD(fa[1][1], aa1, 16384);
        SDVAR_RD_B(1, ab1, 0, 0); SDVAR_RD_B(1, ab1, 1, 10240);
#undef SDVAR_RD_B
#undef SDVAR_LDS_RD
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][1], fb[s][0][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][0], fb[s][1][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][0], fb[s][0][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pf) {
                    const int grp = 5 * s + j;      // 0..9: the 8 DMA slots of K-step t+2 behind the first eight MFMA groups
                    if (grp < 8) issue_one(nst, grp);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (SDVAR_CDBG(a, 1)) return;
    conv_epilogue<EPI>(a, acc, a.wsi ? *a.wsi : 1.0f, m0, n0, tm, ks, phase, wave, li, lh, tid, csm);
}

// out = sum_s slab[s] + bias (+ res)
__global__ __launch_bounds__(256) void conv_reduce_kernel(const float* __restrict__ ws, int split, const float* __restrict__ bias, const float* res,
                                                          float* out, int M, int N) {
    const int nv = N >> 2;
    const size_t total = (size_t)M * nv, slab = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % nv) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + 4 * i);
        for (int s = 1; s < split; ++s) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(ws + s * slab + 4 * i);
            acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3];
        }
        if (bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n); acc[0] += b[0]; acc[1] += b[1]; acc[2] += b[2]; acc[3] += b[3]; }
        if (res) { const f32x4 r = *reinterpret_cast<const f32x4*>(res + 4 * i); acc[0] += r[0]; acc[1] += r[1]; acc[2] += r[2]; acc[3] += r[3]; }
        *reinterpret_cast<f32x4*>(out + 4 * i) = acc;
    }
}

// conv weight (Cout, Cin, taps) fp32 -> planes [3 | 2][taps Cin / 32][Cout][32], k = tap Cin + cin; pfmt 2: the two fp16 planes of w * (*scale)
__global__ __launch_bounds__(256) void conv_weight_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ p, int Cout, int Cin, int taps, size_t ps, int pfmt,
                                                                 const float* scale) {
    const int K = taps * Cin;
    const size_t total = (size_t)Cout * K;
    const float sc = (pfmt == PLANES_F16X2 && scale) ? *scale : 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / K), k = (int)(i % K), tap = k / Cin, ci = k - tap * Cin;
        const float v = w[((size_t)co * Cin + ci) * taps + tap];
        const size_t o = kb_index(co, k, Cout);
        if (pfmt == PLANES_F16X2) {
            uint16_t h, l;
            split2h(v * sc, h, l);
            p[o] = h; p[ps + o] = l;
        } else {
            uint16_t p0, p1, p2;
            split3(v, p0, p1, p2);
            p[o] = p0; p[ps + o] = p1; p[2 * ps + o] = p2;
        }
    }
}

// scale (pfmt 2): device {2^S, 2^-S, ..} from weight_scale_f16 (gemm_f16x2.hip) over the tensor(s) that share one launch; null = unscaled
int conv_weight_planes(const float* w, uint16_t* planes, int Cout, int Cin, int taps, size_t plane_stride, int pfmt, const float* scale, hipStream_t stream) {
    SDVAR_CHECK_ARG(w && planes && Cout > 0 && Cin % 32 == 0 && (taps == 1 || taps == 4 || taps == 9), "conv_weight_planes: Cout=%d Cin=%d taps=%d", Cout, Cin, taps);
    SDVAR_CHECK_ARG(pfmt == PLANES_BF16X3 || pfmt == PLANES_F16X2, "conv_weight_planes: plane format %d", pfmt);
    hipLaunchKernelGGL(conv_weight_planes_kernel, dim3(1024), dim3(256), 0, stream, w, planes, Cout, Cin, taps, plane_stride, pfmt, scale);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// Nearest-2x up-sampling followed by a 3x3 convolution (Upsample2x, basic_vae.py:24-30) = four 2x2 convolutions on the input grid, one per
// output phase (py, px): output pixel (2y + py, 2x + px) reads input rows {y-1, y} (py = 0) or {y, y+1} (py = 1), and the 3x3 taps that land on
// the same input pixel are summed: 16 instead of 36 multiply-adds per output pixel.  weff [4 phases][Cout][Cin][4 taps (ty, tx)].
__global__ __launch_bounds__(256) void upconv_weight_kernel(const float* __restrict__ w, float* __restrict__ weff, int Cout, int Cin) {
    const size_t total = (size_t)4 * Cout * Cin * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i & 3), ty = tap >> 1, tx = tap & 1;
        const size_t oc = (i >> 2) % ((size_t)Cout * Cin);
        const int ph = (int)(i / ((size_t)4 * Cout * Cin)), py = ph >> 1, px = ph & 1;
        // kernel rows (0..2 = dy -1..1) that map to input-row tap ty of phase py
        const int y0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), y1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int x0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), x1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        float acc = 0.f;
        for (int ky = y0; ky <= y1; ++ky)
            for (int kx = x0; kx <= x1; ++kx) acc += w[oc * 9 + ky * 3 + kx];
        weff[i] = acc;
    }
}

int upconv_weights(const float* w, float* weff, int Cout, int Cin, hipStream_t stream) {
    hipLaunchKernelGGL(upconv_weight_kernel, dim3(1024), dim3(256), 0, stream, w, weff, Cout, Cin);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

// split heuristic: fill the 256 CUs (one resident workgroup each) without leaving a mostly empty last round
static int conv_choose_split(int M, int N, int nkt, size_t ws_floats, double kstep_cycles) {
    const long tiles = (long)((M + CBM - 1) / CBM) * ((N + CBN - 1) / CBN);
    double best = 1e30; int bs = 1;
    for (int split = 1; split <= 16 && split <= nkt / 2; ++split) {
        if (split > 1 && ((size_t)split * M * N > ws_floats || N % 4)) break;
        const int kps = (nkt + split - 1) / split;
        if ((nkt + kps - 1) / kps != split) continue;
        const long rounds = (tiles * split + 255) / 256;
        double cyc = (double)rounds * (kps * kstep_cycles + 3000.0);         // bf16x3: 60 MFMAs per wave and K-step, 2 waves per SIMD + DMA/sync (f16x2: 30); prologue + epilogue
        if (split > 1) cyc += 4000.0 + (double)(split + 2) * M * N * 4.0 / 4000.0;
        if (cyc < best) { best = cyc; bs = split; }
    }
    return bs;
}

// out[B H W][N] = conv(X planes of a (B, Cin, H, W) tensor, W planes) + bias (+ res[B H W][N]).  taps = 9: 3x3, pad 1; taps = 1: 1x1.
// up >= 0 (taps = 4, W = four phase weight sets from upconv_weights + conv_weight_planes, w_phase_stride elements apart): the convolution
// of the nearest-2x up-sampled tensor, all four output phases in one launch, written into out[B 2H 2W][N].
// gn_part (optional): receives the GroupNorm partial sums of the output, [B H W / 256][32][2] doubles, when the shape allows the fused
// epilogue (no split-K, H W a multiple of 256, 32 groups that tile the 160-column workgroup tile); *gn_done tells whether it was written.
// pfmt: PLANES_BF16X3 (six bf16 products) or PLANES_F16X2 (three fp16 products; wsi -> 2^-S of the weight scale, or null)
int conv_planes(const uint16_t* X, size_t xps, size_t x_rows, int x_row0, const uint16_t* W, size_t wps, int pfmt, const float* wsi, const float* bias, const float* res,
                float* out, int B, int H, int Wd, int N, int Cin, int taps, float* ws, size_t ws_floats, int force_split, double* gn_part, int* gn_done, int up_phase,
                size_t w_phase_stride, hipStream_t stream) {
    if (gn_done) *gn_done = 0;
    const int M = B * H * Wd, w2 = Wd + 2;
    const size_t Mp = (size_t)B * (H + 2) * w2;
    SDVAR_CHECK_ARG(X && W && out, "conv: null operand");
    SDVAR_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && Cin % 32 == 0 && (taps == 1 || taps == 9 || (taps == 4 && up_phase >= 0)) && (taps == 4 || up_phase < 0), "conv: M=%d N=%d Cin=%d taps=%d", M, N, Cin, taps);
    SDVAR_CHECK_ARG(B > 0 && H > 0 && Wd > 0 && x_row0 >= w2 + 1, "conv: guard rows %d < row pitch %d + 1", x_row0, w2);
    SDVAR_CHECK_ARG(x_rows >= (size_t)x_row0 + Mp + w2 + 1, "conv: plane rows %zu too few", x_rows);
    SDVAR_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && xps % 8 == 0 && wps % 8 == 0, "conv: planes must be 16-byte aligned");
    SDVAR_CHECK_ARG(pfmt == PLANES_BF16X3 || pfmt == PLANES_F16X2, "conv: plane format %d", pfmt);
    const bool F16 = pfmt == PLANES_F16X2;
    ConvArgs a{X, W, xps, wps, x_rows, x_row0, F16 ? wsi : nullptr, bias, res, out, M, N, Cin / 32, taps, H, Wd, N, 1, taps * (Cin / 32), nullptr, 1, up_phase, w_phase_stride, 0};
#ifdef SDVAR_TIMING_EXPERIMENTS
    static const int conv_dbg = getenv("SDVAR_CONV_DBG") ? atoi(getenv("SDVAR_CONV_DBG")) : 0;
    a.dbg = conv_dbg;
#else
    a.dbg = 0;
#endif
    const int nkt = taps * (Cin / 32);
    const int tiles = ((M + CBM - 1) / CBM) * ((N + CBN - 1) / CBN);
    int split = up_phase >= 0 ? 1 : force_split > 0 ? force_split : conv_choose_split(M, N, nkt, ws ? ws_floats : 0, F16 ? 1200.0 : 2100.0);
    if (split > nkt) split = nkt;
    if (split > 1) {
        SDVAR_CHECK_ARG(ws && (size_t)split * M * N <= ws_floats && N % 4 == 0, "conv: split-K workspace too small");
        const int kps = (nkt + split - 1) / split;
        split = (nkt + kps - 1) / kps;
        a.k_per_split = kps; a.split = split;
    }
    const size_t lds = F16 ? 3 * (size_t)CSTAGE_H * sizeof(uint16_t) : 2 * (size_t)CSTAGE * sizeof(uint16_t);      // 156 KB either way
    const char* pp_s = getenv("SDVAR_CONV_PP");          // read per call; sdvar_debug_set_variant("conv_pp", v) overrides it (tests): 0 = the round-2 loop, 1 = ping-pong on 32x32x16, 2 (default) = ping-pong on 16x16x32
    const int pp_env = g_conv_pp_override >= 0 ? g_conv_pp_override : (pp_s ? atoi(pp_s) : 2);
    const int pp = (pp_env == 2 && N % 4 != 0) ? 1 : pp_env;                                         // the 16-byte epilogue needs whole groups of four columns
    static LdsOptIn opt_in, opt_in_h;
    if (F16) SDVAR_LDS_OPT_IN(opt_in_h, lds, (const void*)conv_f16x2_kernel<CEPI_BIAS, 0>, (const void*)conv_f16x2_kernel<CEPI_BIAS_RES, 0>, (const void*)conv_f16x2_kernel<CEPI_PARTIAL, 0>,
                              (const void*)conv_f16x2_kernel<CEPI_BIAS, 1>, (const void*)conv_f16x2_kernel<CEPI_BIAS_RES, 1>, (const void*)conv_f16x2_kernel<CEPI_PARTIAL, 1>,
                              (const void*)conv_f16x2_kernel<CEPI_BIAS, 2>, (const void*)conv_f16x2_kernel<CEPI_BIAS_RES, 2>, (const void*)conv_f16x2_kernel<CEPI_PARTIAL, 2>);
    else SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)conv_bf16x3_kernel<CEPI_BIAS>, (const void*)conv_bf16x3_kernel<CEPI_BIAS_RES>, (const void*)conv_bf16x3_kernel<CEPI_PARTIAL>);
    if (split > 1) {
        ConvArgs p = a;
        p.out = ws;
        if (F16 && pp == 2) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_PARTIAL, 2>), dim3(tiles * split), dim3(512), lds, stream, p);
        else if (F16 && pp) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_PARTIAL, 1>), dim3(tiles * split), dim3(512), lds, stream, p);
        else if (F16) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_PARTIAL, 0>), dim3(tiles * split), dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((conv_bf16x3_kernel<CEPI_PARTIAL>), dim3(tiles * split), dim3(512), lds, stream, p);
        SDVAR_LAUNCH_CHECK();
        const size_t total = (size_t)M * (N / 4);
        const int rgrid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(conv_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, ws, split, bias, res, out, M, N);
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    if (gn_part && N % 32 == 0 && CBN % (N / 32) == 0 && (H * Wd) % CBM == 0) { a.gn_part = gn_part; a.cpg = N / 32; if (gn_done) *gn_done = 1; }
    if (F16) {
        if (res && pp == 2) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS_RES, 2>), dim3(tiles), dim3(512), lds, stream, a);
        else if (res && pp) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS_RES, 1>), dim3(tiles), dim3(512), lds, stream, a);
        else if (res) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS_RES, 0>), dim3(tiles), dim3(512), lds, stream, a);
        else if (pp == 2) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS, 2>), dim3(tiles, up_phase >= 0 ? 4 : 1), dim3(512), lds, stream, a);
        else if (pp) hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS, 1>), dim3(tiles, up_phase >= 0 ? 4 : 1), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((conv_f16x2_kernel<CEPI_BIAS, 0>), dim3(tiles, up_phase >= 0 ? 4 : 1), dim3(512), lds, stream, a);
    } else if (res) hipLaunchKernelGGL((conv_bf16x3_kernel<CEPI_BIAS_RES>), dim3(tiles), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((conv_bf16x3_kernel<CEPI_BIAS>), dim3(tiles, up_phase >= 0 ? 4 : 1), dim3(512), lds, stream, a);
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

}  // namespace sdvar
