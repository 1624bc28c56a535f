// fp32 "NT" GEMM on the gfx950 matrix cores:  out[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] ).
//
// Replaces the reference's F.linear calls on the sampling path (/root/reference/models/basic_var.py:93 QKV, :119 proj,
// :52 fc1/fc2, :156 ada_lin; models/var.py:125 head, :187 word_embed).  The reference computes them in fp32, and the
// parity bar is bit-exact token ids, so the MFMA form used is v_mfma_f32_32x32x2_f32: fp32 in / fp32 accumulate,
// bitwise a k-ordered fmaf chain (MI355X_MICROARCH.md "Matrix cores"), 157 TFLOP/s peak.
//
// Tiling (wave64, 4 waves / 256 threads per workgroup):
//   block tile BM x BN (128x128, 64x128 or 32x128), K-step 32, both operands K-contiguous in HBM;
//   X and W tiles are staged global -> registers -> LDS (double buffered, one barrier per K-step) with rows padded to
//   36 floats so that the per-lane ds_read_b128 of 4 consecutive k is bank-conflict free;
//   the 4 k of one ds_read_b128 feed 4 MFMAs: lane (i, h) supplies k = 8c + 4h + e for MFMA e of chunk c on BOTH
//   operands, i.e. the K order inside a tile is permuted identically for X and W (sum unchanged, order fixed).
//   Output: lane holds column n = lane & 31 of 16 rows -> 128-byte row segments per store instruction.
// Split-K: the sampler's GEMMs have M = 2B * (tokens of the stage) = 16 .. 4096 rows and N as small as C, so most
//   launches have far fewer output tiles than the 256 CUs and a single workgroup walking all of K is latency-bound
//   (measured: 116 us for any grid of 8..200 workgroups at K = 4096).  When tiles < ~256 the K range is cut into
//   `split` slices (grid = tiles x split), each slice stores its raw fp32 partial tile into a workspace slab, and
//   splitk_reduce_kernel sums the slabs in slice order (deterministic, no atomics) and applies the epilogue.
// Epilogues (fused, the reference's elementwise ops around each linear):
//   EPI_BIAS        out = acc + bias
//   EPI_BIAS_GELU   out = gelu_tanh(acc + bias)                         (basic_var.py:40,52)
//   EPI_GATED_RES   out = res + (acc + bias) * gate[row / rows_per_gate] (basic_var.py:157-158: x + f(.)*gamma)
#include "common.h"

namespace sdvar {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_GATED_RES = 2, EPI_PARTIAL = 3 };

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // floats

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))   (nn.GELU(approximate='tanh'))
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

struct GemmArgs {
    const float* X; const float* W; const float* bias; float* out;
    const float* res; const float* gate;
    int M, N, K, ldx, ldo, ldres;
    int rows_per_gate, gate_stride;
    int split, k_per_split;     // split-K: slice ks covers K-tiles [ks*k_per_split, ...) and writes slab ks of `out`
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(GemmArgs a) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;     // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;               // 32x32 MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1, "wave tile too small");
    constexpr int XV = BM / 32, WV = BN / 32;                // float4 loads per thread per K-step

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = (BM + BN) * LDS_STRIDE;           // floats per pipeline stage: X tile then W tile

    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int ntile = tiles_m * tiles_n;
    const int ks = blockIdx.x / ntile;                      // K slice (0 when not split)
    const int lid = xcd_remap(blockIdx.x - ks * ntile, ntile);
    const int tm = lid % tiles_m, tn = lid / tiles_m;       // m fastest: neighbours share the W panel
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    // staging assignment: thread -> (row = tid/8 + 32*i, 4 floats at col 4*(tid%8))
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    f32x4 rx[XV], rw[WV];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int m = m0 + srow + 32 * i;
            rx[i] = (m < a.M) ? *reinterpret_cast<const f32x4*>(a.X + (size_t)m * a.ldx + k0 + scol) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int n = n0 + srow + 32 * i;
            rw[i] = (n < a.N) ? *reinterpret_cast<const f32x4*>(a.W + (size_t)n * a.K + k0 + scol) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XV; ++i) *reinterpret_cast<f32x4*>(smem + buf * STAGE + (srow + 32 * i) * LDS_STRIDE + scol) = rx[i];
#pragma unroll
        for (int i = 0; i < WV; ++i) *reinterpret_cast<f32x4*>(smem + buf * STAGE + (BM + srow + 32 * i) * LDS_STRIDE + scol) = rw[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kt0 = ks * a.k_per_split;
    const int nk = min(a.K / BK - kt0, a.k_per_split);
    load_tile(kt0 * BK);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt0 + kt + 1) * BK);
        const float* px = smem + buf * STAGE + (wm * WM + li) * LDS_STRIDE + 4 * lh;
        const float* pw = smem + buf * STAGE + (BM + wn * WN + li) * LDS_STRIDE + 4 * lh;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(px + i * 32 * LDS_STRIDE + 8 * c);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(pw + j * 32 * LDS_STRIDE + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane owns column n, rows (r&3) + 8*(r>>2) + 4*lh of each 32x32 tile
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + li;
        if (n >= a.N) continue;
        const float bv = (EPI != EPI_PARTIAL && a.bias) ? a.bias[n] : 0.f;
        float* outp = (EPI == EPI_PARTIAL) ? a.out + (size_t)ks * a.M * a.ldo : a.out;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == EPI_BIAS_GELU) v = gelu_tanh(v);
                if (EPI == EPI_GATED_RES) {
                    const float g = a.gate[(size_t)(m / a.rows_per_gate) * a.gate_stride + n];
                    v = a.res[(size_t)m * a.ldres + n] + v * g;
                }
                outp[(size_t)m * a.ldo + n] = v;
            }
        }
    }
}

// out = epi( sum_s ws[s] + bias ), 4 columns per thread, slabs summed in slice order
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int split, const float* __restrict__ bias, float* out,
                                                            const float* res, const float* __restrict__ gate, int M, int N, int ldo, int ldres,
                                                            int rows_per_gate, int gate_stride) {
    const int nv = N >> 2;
    const size_t total = (size_t)M * nv, slab = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), n = (int)(i % nv) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + (size_t)m * N + n);
        for (int s = 1; s < split; ++s) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(ws + s * slab + (size_t)m * N + n);
            acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2]; acc[3] += p[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (bias ? bias[n + e] : 0.f);
            if (EPI == EPI_BIAS_GELU) v = gelu_tanh(v);
            if (EPI == EPI_GATED_RES) v = res[(size_t)m * ldres + n + e] + v * gate[(size_t)(m / rows_per_gate) * gate_stride + n + e];
            out[(size_t)m * ldo + n + e] = v;
        }
    }
}

static thread_local float* g_ws = nullptr;         // split-K slabs, one workspace per host thread (= per model object / stream: sdvar_hip.h)
constexpr size_t WS_FLOATS = (size_t)24 << 20;     // 96 MiB

static thread_local float* g_ws_model = nullptr;   // a model object's own slabs while one of its calls is on the stack (api.hip)
float* set_splitk_workspace(float* p) { float* old = g_ws_model; g_ws_model = p; return old; }
size_t splitk_workspace_floats() { return WS_FLOATS; }

float* splitk_workspace(size_t* floats) {          // shared with gemm_bf16x3.hip
    if (floats) *floats = WS_FLOATS;
    if (g_ws_model) return g_ws_model;
    if (!g_ws && hipMalloc((void**)&g_ws, WS_FLOATS * sizeof(float)) != hipSuccess) { set_error("split-K workspace allocation failed"); return nullptr; }
    return g_ws;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(GemmArgs a, int epi, int split, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 2 * (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
    dim3 block(256);
    static LdsOptIn opt_in;         // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
    SDVAR_LDS_OPT_IN(opt_in, lds, (const void*)gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_BIAS>, (const void*)gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_BIAS_GELU>,
                     (const void*)gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_GATED_RES>, (const void*)gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_PARTIAL>);
    const int nkt = a.K / BK;
    if (split > 1) {
        float* const ws = splitk_workspace(nullptr);          // the calling model's slabs, or this thread's
        if (!ws) return SDVAR_ERR_HIP;
        GemmArgs p = a;
        p.out = ws; p.ldo = a.N; p.split = split; p.k_per_split = (nkt + split - 1) / split;
        hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_PARTIAL>), dim3(tiles * split), block, lds, stream, p);
        SDVAR_LAUNCH_CHECK();
        const size_t total = (size_t)a.M * (a.N / 4);
        const int rgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        switch (epi) {
            case EPI_BIAS: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
            case EPI_BIAS_GELU: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS_GELU>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
            default: hipLaunchKernelGGL(splitk_reduce_kernel<EPI_GATED_RES>, dim3(rgrid), block, 0, stream, ws, split, a.bias, a.out, a.res, a.gate, a.M, a.N, a.ldo, a.ldres, a.rows_per_gate, a.gate_stride); break;
        }
        SDVAR_LAUNCH_CHECK();
        return SDVAR_OK;
    }
    a.split = 1; a.k_per_split = nkt;
    dim3 grid(tiles);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_BIAS>), grid, block, lds, stream, a); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_BIAS_GELU>), grid, block, lds, stream, a); break;
        case EPI_GATED_RES: hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WAVES_M, WAVES_N, EPI_GATED_RES>), grid, block, lds, stream, a); break;
        default: set_error("gemm: unknown epilogue %d", epi); return SDVAR_ERR_ARG;
    }
    SDVAR_LAUNCH_CHECK();
    return SDVAR_OK;
}

static int g_force_bm = 0, g_force_split = 0;     // tools/gemm_bench.py --sweep (sdvar_debug_set_gemm_cfg); 0 = automatic
void debug_set_gemm_cfg(int bm, int split) { g_force_bm = bm; g_force_split = split; }

// (row tile, K slices) as a pure function of the shape - the summation order, hence every output bit, depends only on
// (M, N, K).  Cost model in MFMA cycles per CU, calibrated with tools/gemm_bench.py --sweep on MI355X:
//   a workgroup spends 64 cycles x 16 x (bm/32 x 4 / 4 waves) per K-step on each SIMD; workgroups are dealt evenly over
//   the 256 CUs, co-resident ones share the matrix pipes (so time ~ per-CU sum), a lone workgroup per CU cannot hide
//   its LDS/barrier latency, every workgroup pays a fixed prologue/epilogue, and split > 1 pays the slab round trip.
static void choose_cfg(int M, int N, int K, int* bm_out, int* split_out) {
    const int nkt = K / BK, tiles_n = (N + 127) / 128;
    double best = 1e30; int bbm = 128, bs = 1;
    const int bms[3] = {128, 64, 32};
    const int resident[3] = {2, 2, 3};                                   // workgroups per CU the LDS footprint admits
    const double lat[4] = {0.0, 1.35, 1.08, 1.0};                        // slowdown of n co-resident workgroups' K-step (latency exposed when alone)
    for (int bi = 0; bi < 3; ++bi) {
        const int bm = bms[bi], res = resident[bi];
        const int tiles = ((M + bm - 1) / bm) * tiles_n;
        // MFMA cycles per K-step per workgroup; narrower tiles re-read more LDS/L2 per MFMA (measured +3 % / +12 %)
        const double ktile = 64.0 * 16.0 * (bm / 32) * (bm == 32 ? 1.12 : (bm == 64 ? 1.03 : 1.0));
        for (int split = 1; split <= 32 && split <= nkt / 2; ++split) {
            if (split > 1 && ((size_t)split * M * N > WS_FLOATS || N % 4)) break;
            const int kps = (nkt + split - 1) / split;
            if ((nkt + kps - 1) / kps != split) continue;               // would leave empty trailing slices
            const long blocks = (long)tiles * split;
            const long per_cu = (blocks + 255) / 256;                    // workgroups the busiest CU executes
            const double T = kps * ktile + 2500.0 + 40.0 * bm;           // one workgroup alone on the matrix pipes (+ prologue/epilogue)
            const long full = per_cu / res, rem = per_cu % res;
            double cyc = full * res * T * lat[res] + (rem ? rem * T * lat[rem] : 0.0);
            if (split > 1) cyc += 6000.0 + (double)(split + 1) * M * N * 4.0 / 1800.0;   // reduce launch + slab traffic (~3.8 TB/s at 2.1 GHz)
            if (cyc < best) { best = cyc; bbm = bm; bs = split; }
        }
    }
    *bm_out = bbm; *split_out = bs;
}

// Host entry used by the model code and by the op-level C-ABI.
int gemm_f32_nt(const float* X, int ldx, const float* W, const float* bias, float* out, int ldo, int M, int N, int K, int epi,
                const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, hipStream_t stream) {
    SDVAR_CHECK_ARG(X && W && out, "gemm: null operand");
    SDVAR_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0, "gemm: need K %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    SDVAR_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldo >= N, "gemm: bad leading dims ldx=%d ldo=%d", ldx, ldo);
    SDVAR_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm: operands must be 16-byte aligned");
    SDVAR_CHECK_ARG(epi >= EPI_BIAS && epi <= EPI_GATED_RES, "gemm: unknown epilogue %d", epi);
    if (epi == EPI_GATED_RES) SDVAR_CHECK_ARG(res && gate && rows_per_gate > 0 && ldres >= N, "gemm: gated-residual epilogue needs res/gate");
    GemmArgs a{X, W, bias, out, res, gate, M, N, K, ldx, ldo, ldres, rows_per_gate > 0 ? rows_per_gate : 1, gate_stride, 1, K / BK};
    int bm, split;
    choose_cfg(M, N, K, &bm, &split);
    if (g_force_bm && g_force_bm <= 128) bm = g_force_bm;
    if (g_force_split) {
        split = g_force_split;
        const int nkt = K / BK;
        if (split > nkt) split = nkt;
        while (split > 1 && (size_t)split * M * N > WS_FLOATS) --split;
        const int kps = (nkt + split - 1) / split;
        split = (nkt + kps - 1) / kps;
    }
    if (bm == 32) return launch_cfg<32, 128, 1, 4>(a, epi, split, stream);
    if (bm == 64) return launch_cfg<64, 128, 2, 2>(a, epi, split, stream);
    return launch_cfg<128, 128, 2, 2>(a, epi, split, stream);
}

}  // namespace sdvar
