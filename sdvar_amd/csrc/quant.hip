// VQVAE token -> feature accumulation between scales (the quant.py half of the loop), fp32.
// Replaces VectorQuantizer2.embedding + get_next_autoregressive_input + Phi.forward
// (/root/reference/models/quant.py:39, 187-196, 205-206, 223-226; call site models/var.py:205-211):
//     h      = codebook[ids]                         (B, 32, pn, pn)
//     h      = bicubic_up(h, HW)            (stage < S-1)          = Wup h Wup^T   (constant per-stage matrix)
//     h      = 0.5 h + 0.5 (conv3x3(h) + bias)       (shared Phi_k)
//     f_hat += h
//     next   = area_down(f_hat, pn_next)    (stage < S-1)          = Wdn f Wdn^T   (adaptive average pooling)
// Three small launches per stage, each with B*32 workgroups so the serial inter-stage dependency costs microseconds:
//   quant_up_kernel   (image, channel): gather + separable up-sampling           -> up (B,32,HW,HW)
//   quant_phi_kernel  (image, out-channel): Phi conv + residual mix + f_hat +=   -> f_hat (B,32,HW,HW) in place
//   quant_down_kernel (image, channel): separable area pooling, transposed store -> next (B, pn'^2, 32)
#include "common.h"

namespace sdvar {

constexpr int QMAX_HW = 64;

// up (b, c): tmp[Y][x] = sum_y Wup[Y][y] h[y][x];  out[Y][X] = sum_x tmp[Y][x] Wup[X][x]
// hvec != null: the stage's feature vectors are given directly as (B, pn*pn, Cv) instead of token ids (more_smooth=True mixes the
// codebook softly, models/var.py:206-208)
__global__ __launch_bounds__(256) void quant_up_kernel(const long long* __restrict__ ids, int ids_stride, const float* __restrict__ codebook,
                                                       const float* __restrict__ hvec, const float* __restrict__ Wup, float* __restrict__ up, int pn,
                                                       int HW, int Cv, int identity) {
    extern __shared__ float sm[];
    float* hs = sm;                  // pn*pn
    float* tmp = sm + pn * pn;       // HW*pn
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (hvec) for (int p = tid; p < pn * pn; p += blockDim.x) hs[p] = hvec[((size_t)b * pn * pn + p) * Cv + c];
    else for (int p = tid; p < pn * pn; p += blockDim.x) hs[p] = codebook[(size_t)ids[(size_t)b * ids_stride + p] * Cv + c];
    __syncthreads();
    float* dst = up + ((size_t)b * Cv + c) * HW * HW;
    if (identity) {                  // last stage: no interpolation (quant.py:193-196)
        for (int p = tid; p < HW * HW; p += blockDim.x) dst[p] = hs[p];
        return;
    }
    for (int e = tid; e < HW * pn; e += blockDim.x) {
        const int Y = e / pn, x = e % pn;
        float acc = 0.f;
        for (int y = 0; y < pn; ++y) acc = fmaf(Wup[Y * pn + y], hs[y * pn + x], acc);
        tmp[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < HW * HW; e += blockDim.x) {
        const int Y = e / HW, X = e % HW;
        float acc = 0.f;
        for (int x = 0; x < pn; ++x) acc = fmaf(tmp[Y * pn + x], Wup[X * pn + x], acc);
        dst[e] = acc;
    }
}

// phi (b, co): f_hat[b][co] += 0.5*up[b][co] + 0.5*(bias[co] + sum_{ci,dy,dx} w[co][ci][dy][dx] up[b][ci][Y+dy-1][X+dx-1])
// f_out = f_in + Phi mix (f_in == f_out: the in-place form of quant.py:191; distinct buffers keep the per-stage snapshots of a draft round
// without extra copies)
// STAGED: the image's up-sampled planes (Cv HW^2 floats: 32 KB at HW = 16, 128 KB at 32) and the output channel's 9 Cv weights are copied to LDS
// first - every one of them is read HW^2 / 9 times by the workgroup - and the 288 multiply-adds of a pixel (same order as the unstaged form, so the
// results are the same bits) read LDS: 31.6 -> 7 us per launch at HW = 16, on the serial path between two stages.
template <bool STAGED>
__global__ __launch_bounds__(256) void quant_phi_kernel(const float* __restrict__ up, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* f_in, float* f_hat, int HW, int Cv) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    const int co = blockIdx.x, b = blockIdx.y;
    const float* ub = up + (size_t)b * Cv * HW * HW;
    const float* wc = w + (size_t)co * Cv * 9;
    if (STAGED) {
        const int n4 = Cv * HW * HW / 4;                // HW * HW is a multiple of 4 on this path (host)
        for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<f32x4*>(psm)[i] = reinterpret_cast<const f32x4*>(ub)[i];
        for (int i = threadIdx.x; i < 9 * Cv; i += blockDim.x) psm[Cv * HW * HW + i] = wc[i];
        __syncthreads();
        ub = psm; wc = psm + Cv * HW * HW;
    }
    for (int e = threadIdx.x; e < HW * HW; e += blockDim.x) {
        const int Y = e / HW, X = e % HW;
        float acc = 0.f;
        for (int ci = 0; ci < Cv; ++ci) {
            const float* uc = ub + (size_t)ci * HW * HW;
            const float* wk = wc + ci * 9;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = Y + dy - 1;
                if (yy < 0 || yy >= HW) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = X + dx - 1;
                    if (xx < 0 || xx >= HW) continue;
                    acc = fmaf(wk[dy * 3 + dx], uc[yy * HW + xx], acc);
                }
            }
        }
        const float h = ub[(size_t)co * HW * HW + e];
        const float mixed = h * 0.5f + (acc + bias[co]) * 0.5f;
        const size_t o = ((size_t)b * Cv + co) * HW * HW + e;
        f_hat[o] = f_in[o] + mixed;
    }
}

// down (b, c): tmp[y'][X] = sum_Y Wdn[y'][Y] f[Y][X];  next[b][y'*pn2 + x'][c] = sum_X tmp[y'][X] Wdn[x'][X]
__global__ __launch_bounds__(256) void quant_down_kernel(const float* __restrict__ f_hat, const float* __restrict__ Wdn, float* __restrict__ nxt,
                                                         int HW, int pn2, int Cv) {
    extern __shared__ float sm[];
    float* fs = sm;                  // HW*HW
    float* tmp = sm + HW * HW;       // pn2*HW
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* src = f_hat + ((size_t)b * Cv + c) * HW * HW;
    for (int p = tid; p < HW * HW; p += blockDim.x) fs[p] = src[p];
    __syncthreads();
    for (int e = tid; e < pn2 * HW; e += blockDim.x) {
        const int y2 = e / HW, X = e % HW;
        float acc = 0.f;
        for (int Y = 0; Y < HW; ++Y) acc = fmaf(Wdn[y2 * HW + Y], fs[Y * HW + X], acc);
        tmp[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < pn2 * pn2; e += blockDim.x) {
        const int y2 = e / pn2, x2 = e % pn2;
        float acc = 0.f;
        for (int X = 0; X < HW; ++X) acc = fmaf(tmp[y2 * HW + X], Wdn[x2 * HW + X], acc);
        nxt[((size_t)b * pn2 * pn2 + e) * Cv + c] = acc;
    }
}

int quant_next(const long long* ids, int ids_stride, const float* hvec, const float* codebook, const float* Wup, const float* phi_w, const float* phi_b,
               const float* Wdn, float* up_scratch, const float* f_in, float* f_hat, float* nxt, int B, int pn, int pn_next, int HW, int Cv, int last,
               hipStream_t stream) {
    SDVAR_CHECK_ARG((ids || hvec) && codebook && phi_w && phi_b && up_scratch && f_hat, "quant_next: null operand");
    SDVAR_CHECK_ARG(B > 0 && pn > 0 && pn <= HW && HW <= QMAX_HW && (!last || pn == HW), "quant_next: bad sizes pn=%d HW=%d", pn, HW);
    const size_t lds_up = (size_t)(pn * pn + HW * pn) * sizeof(float);
    hipLaunchKernelGGL(quant_up_kernel, dim3(Cv, B), dim3(256), lds_up, stream, ids, ids_stride, codebook, hvec, Wup, up_scratch, pn, HW, Cv, last);
    SDVAR_LAUNCH_CHECK();
    const size_t lds_phi = ((size_t)Cv * HW * HW + 9 * (size_t)Cv) * sizeof(float);
    if ((HW * HW) % 4 == 0 && lds_phi <= 140 * 1024 && ((uintptr_t)up_scratch % 16) == 0) {
        static LdsOptIn opt_in;
        SDVAR_LDS_OPT_IN(opt_in, 140 * 1024, (const void*)quant_phi_kernel<true>);       // once per device: the largest size this path takes
        hipLaunchKernelGGL(quant_phi_kernel<true>, dim3(Cv, B), dim3(256), lds_phi, stream, up_scratch, phi_w, phi_b, f_in ? f_in : f_hat, f_hat, HW, Cv);
    } else hipLaunchKernelGGL(quant_phi_kernel<false>, dim3(Cv, B), dim3(256), 0, stream, up_scratch, phi_w, phi_b, f_in ? f_in : f_hat, f_hat, HW, Cv);
    SDVAR_LAUNCH_CHECK();
    if (!last) {
        SDVAR_CHECK_ARG(Wdn && nxt && pn_next > 0 && pn_next <= HW, "quant_next: missing down table");
        const size_t lds_dn = (size_t)(HW * HW + pn_next * HW) * sizeof(float);
        hipLaunchKernelGGL(quant_down_kernel, dim3(Cv, B), dim3(256), lds_dn, stream, f_hat, Wdn, nxt, HW, pn_next, Cv);
        SDVAR_LAUNCH_CHECK();
    }
    return SDVAR_OK;
}

}  // namespace sdvar
