/* libsdvar_hip.so - C ABI of the MI355X (gfx950) speculative draft-verify sampler kernels for VAR.
 *
 * The reference (lijrjyan/SDVAR) is pure Python/PyTorch and has no FFI; the boundary it exposes for this path is its
 * Python API.  Each entry point below replaces the op sequence of the reference lines cited on
 * it; sdvar_amd/engine.py is the ctypes binding and sdvar_amd/var.py mirrors VAR / SDVAR on top of it.
 *
 * Conventions: every pointer is a DEVICE pointer unless marked "host"; tensors are dense row-major fp32, ids int64;
 * all work is enqueued on the caller's `stream` (a hipStream_t passed as void*, NULL = default stream) and returns
 * without synchronising; return value 0 = ok, otherwise see sdvar_last_error().  The library never owns caller
 * memory; the KV cache, adaLN table and workspaces it allocates itself are freed by the *_destroy calls.
 * One host thread per model object (same contract as the reference's module-attribute caches, basic_var.py:85-87); different
 * host threads may drive different objects on different streams concurrently (per-thread split-K workspaces, no shared mutable state
 * outside the objects; the sdvar_debug_* / sdvar_prof_* switches are process-wide and meant for single-threaded tools).
 */
#ifndef SDVAR_HIP_H
#define SDVAR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDVAR_ABI_VERSION 4      /* 4 (round 4): sdvar_cfg_combine, sdvar_op_gemm_rowblk, sdvar_debug_set_rowblk; 3 (round 3): the f16-plane KV-cache formats 3 / 4 store V row-major like K; new debug entry points (guard, gemm cfg getter) */
#define SDVAR_MAX_STAGES 16

typedef struct sdvar_model sdvar_model_t;   /* one VAR transformer: weights (borrowed), KV cache, workspaces */
typedef struct sdvar_quant sdvar_quant_t;   /* VectorQuantizer2 inference side: codebook, Phi convs, resample tables */
typedef struct sdvar_vae sdvar_vae_t;       /* VQVAE image decoder (fhat_to_img): conv weights as bf16x3 planes, activation workspaces */

typedef struct {
    int32_t depth;                          /* d: width C = 64 d, heads H = d   (models/__init__.py:26-27) */
    int32_t n_stages;                       /* S */
    int32_t patch_nums[SDVAR_MAX_STAGES];   /* the scale ladder                (models/__init__.py:18) */
    int32_t vocab;                          /* V = 4096 */
    int32_t cvae;                           /* 32 */
    int32_t num_classes;                    /* 1000; class_emb has num_classes + 1 rows (var.py:62) */
    int32_t max_batch;                      /* B; the CFG batch is R = 2B rows (var.py:162,188) */
    int32_t max_chunk_stages;               /* largest number of stages one forward may cover (gamma) */
    int32_t kv_dtype;                       /* KV-cache storage: 0 = fp32 (reference CPU path), 1 = fp16 (BASELINE config P4) */
    int32_t gemm_mode;                      /* 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = bf16x3 split operands on the bf16 MFMA
                                               (fp32-accurate: x = x1+x2+x3 exactly, 6 of 9 plane products, fp32 accumulate), 2 = f16x2 split
                                               operands on the f16 MFMA (x ~ xh + xl to 2^-22, 3 of 4 plane products, weights scaled by a power of
                                               two per tensor, activations saturate at +-65504: csrc/gemm_f16x2.hip) */
} sdvar_model_desc;

int sdvar_abi_version(void);
const char* sdvar_last_error(void);        /* host string, valid until the next failing call on this thread */

/* ---- model object -------------------------------------------------------------------------------------------- */
int sdvar_model_create(const sdvar_model_desc* desc /*host*/, sdvar_model_t** out /*host*/);
int sdvar_model_destroy(sdvar_model_t* m);
/* state_dict tensors of models/var.py:56-78: class_emb (num_classes+1, C), pos_start (1,1,C), pos_1LC (1,L,C),
 * lvl_embed (S,C), word_embed.{weight (C,cvae), bias (C)}.  Builds lvl_pos = lvl_embed[lvl] + pos_1LC (var.py:164). */
int sdvar_model_bind_embed(sdvar_model_t* m, const float* class_emb, const float* pos_start, const float* pos_1LC,
                           const float* lvl_embed, const float* word_w, const float* word_b, void* stream);
/* one AdaLNSelfAttn block (basic_var.py:128-159): ada_lin.1.{weight (6C,C), bias}, attn.mat_qkv.weight (3C,C),
 * attn.q_bias, attn.v_bias, attn.scale_mul_1H11 (H), attn.proj.{weight,bias}, ffn.fc1.{weight (4C,C),bias},
 * ffn.fc2.{weight (C,4C),bias}.  ada_w may be NULL for a shared_aln block (see sdvar_model_bind_shared_aln); scale_mul may be NULL for an
 * attn_l2_norm=False model (basic_var.py:66-72: no q/k normalisation, softmax scale 0.25 / sqrt(64)). */
int sdvar_model_bind_block(sdvar_model_t* m, int32_t block, const float* ada_w, const float* ada_b, const float* qkv_w,
                           const float* q_bias, const float* v_bias, const float* scale_mul, const float* proj_w,
                           const float* proj_b, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                           const float* fc2_b, void* stream);
/* shared_aln=True models (VAR-d36-s; var.py:16-19, 81, 192): shared_ada_lin.1.{weight (6C,C), bias (6C)}; their blocks are bound with
 * ada_w = NULL and ada_b = blocks.i.ada_gss (1,1,6,C)  (basic_var.py:143-144, 153-154). */
int sdvar_model_bind_shared_aln(sdvar_model_t* m, const float* shared_w, const float* shared_b);
/* head_nm.ada_lin.1.{weight (2C,C), bias}, head.{weight (V,C), bias}  (basic_var.py:165-174, var.py:116-117) */
int sdvar_model_bind_head(sdvar_model_t* m, const float* nm_w, const float* nm_b, const float* head_w, const float* head_b, void* stream);

/* Per-call prologue (var.py:162-183, 580-601): cond = class_emb[labels ; uncond], first-token map, adaLN parameters
 * of every block and of the head hoisted out of the stage loop (basic_var.py:156, :173 - cond never changes), KV
 * length cursor reset (basic_var.py:87).  labels: (B) int64. */
int sdvar_model_begin(sdvar_model_t* m, int32_t B, const int64_t* labels, void* stream);
/* The same prologue from the conditioning rows themselves: cond (2B, C) device = `sos` / `cond_BD` as SDVAR.init_param returned it
 * (var.py:580-601) and VAR.autoregressive_infer_cfg_sd_helper1 receives it (var.py:319-345) - no label lookup. */
int sdvar_model_begin_cond(sdvar_model_t* m, int32_t B, const float* cond, void* stream);
/* the tensors SDVAR.init_param returns (var.py:580-601), copied out of the model object: cond (2B,C), lvl_pos (L,C),
 * first-token map (2B,C); any pointer may be NULL */
int sdvar_model_export_prologue(sdvar_model_t* m, float* cond, float* lvl_pos, float* first, void* stream);
/* copy the first-token map (R,1,C) into a chunk input x (R, ltot, C) at token 0 */
int sdvar_model_place_first(sdvar_model_t* m, float* x, int32_t ltot, void* stream);
/* KV-cache cursor: number of valid keys; set_len(n) with n <= current is the rollback after a rejected round. */
int sdvar_kv_len(const sdvar_model_t* m);
int sdvar_kv_set_len(sdvar_model_t* m, int32_t len);
/* Hand-off sampler (var.py:817-824, sd_mask = 0): the target starts at stage `stage` with an EMPTY cache - it never sees the
 * draft's prefix - so cache slot 0 holds the first token of that stage.  Needs kv_len == 0; sdvar_model_begin resets it to stage 0. */
int sdvar_kv_set_origin(sdvar_model_t* m, int32_t stage);
/* next-stage input embedding + CFG duplication (var.py:186-188): nxt (B, l', cvae) -> x rows b and B+b,
 * x[(r*ltot + tok_off + t)*C + :] = word_embed(nxt[b][t]) + lvl_pos[begin(s_next) + t]. */
int sdvar_embed_next(sdvar_model_t* m, const float* nxt, int32_t s_next, float* x, int32_t ltot, int32_t tok_off, void* stream);
/* The same with the lvl_pos rows pos_begin .. pos_begin + l' - 1 instead of the stage's own.  VAR.autoregressive_infer_cfg_sd_helper1 counts
 * `cur_L` from the stage it is resumed at (var.py:352, 369-371, 385, 389: the skipped stages never advance it), so a call resumed at stage c
 * embeds stage s with the rows begin(s) - begin(c); the mirror of that entry point reproduces it through this call. */
int sdvar_embed_next_at(sdvar_model_t* m, const float* nxt, int32_t s_next, int32_t pos_begin, float* x, int32_t ltot, int32_t tok_off, void* stream);
/* All blocks + head over the stages s0 .. s0+n_stages-1 in ONE pass (var.py:195-197; verify chunk var.py:1051-1055
 * with the mask rows of var.py:108-113 derived from the stage table).  x (R, lsum, C) is the input and is CLOBBERED
 * (it is the residual stream); logits (R, lsum, V).  Requires kv_len == begin(s0); appends lsum keys. */
int sdvar_stage_forward(sdvar_model_t* m, float* x, int32_t s0, int32_t n_stages, float* logits, void* stream);
/* The same pass under an EXPLICIT additive attention mask instead of the block-causal rows: bias (lsum, kv_len + lsum) fp32, 0 or -inf, the
 * reference's (1, 1, l, K) attn_bias.  For the ablation masks of the hand-off sampler (var.py:557-578 attn_bias_for_sdmasking /
 * attn_bias_for_block, applied at var.py:777-804). */
int sdvar_stage_forward_masked(sdvar_model_t* m, float* x, int32_t s0, int32_t n_stages, const float* bias, float* logits, void* stream);

/* Final adaLN + vocabulary projection only (VAR.get_logits, var.py:119-125; basic_var.py:172-174) on a residual-stream tensor
 * x (R, l, C) -> logits (R, l, V); x is left untouched.  The hand-off sampler with a prefill mask takes its entry-stage logits from
 * the INPUT token map (var.py:809-811), which is this call. */
int sdvar_head_forward(sdvar_model_t* m, const float* x, int32_t l, float* logits, void* stream);

/* ---- quantizer ----------------------------------------------------------------------------------------------- */
int sdvar_quant_create(int32_t n_stages, const int32_t* patch_nums /*host*/, int32_t cvae, int32_t vocab, int32_t max_batch,
                       int32_t n_phi, sdvar_quant_t** out /*host*/);
int sdvar_quant_destroy(sdvar_quant_t* q);
/* quantize.embedding.weight (V,cvae); the Phi convolutions {weight (cvae,cvae,3,3), bias} as host arrays of n_phi device pointers
 * (quant.py:39, 199-243).  n_phi follows the layout the checkpoint was built with (quant.py:27-32): share_quant_resi >= 2 ->
 * quant_resi.qresi_ls.<k> (PhiPartiallyShared, n_phi = share_quant_resi), 1 -> quant_resi.qresi (PhiShared, n_phi = 1),
 * 0 -> quant_resi.<k> (PhiNonShared, n_phi = n_stages); stage s uses the Phi whose tick is nearest to s / (n_stages - 1). */
int sdvar_quant_bind(sdvar_quant_t* q, const float* codebook, const float* const* phi_w /*host*/, const float* const* phi_b /*host*/);
/* quant.py:187-196 for stage si: f_hat (B,cvae,HW,HW) += Phi(up(codebook[ids])) in place; nxt (B, pn_{si+1}^2, cvae)
 * = area_down(f_hat) (not written for the last stage; may be NULL there).  ids[b*ids_stride + p]. */
int sdvar_quant_next(sdvar_quant_t* q, int32_t si, const int64_t* ids, int32_t ids_stride, float* f_hat, float* nxt, int32_t B, void* stream);
/* the same with separate input and output accumulators: f_out = f_in + Phi(up(codebook[ids])); a draft round keeps one f_hat snapshot
 * per drafted stage (var.py:1013-1022 recomputes them) without copies */
int sdvar_quant_next_from(sdvar_quant_t* q, int32_t si, const int64_t* ids, int32_t ids_stride, const float* f_in, float* f_out, float* nxt,
                          int32_t B, void* stream);
/* the same from explicit feature vectors h (B, pn_si^2, cvae) instead of token ids (more_smooth=True, var.py:206-210) */
int sdvar_quant_next_h(sdvar_quant_t* q, int32_t si, const float* h, float* f_hat, float* nxt, int32_t B, void* stream);
/* more_smooth=True (var.py:206-208 + helpers.py:22-36): h (B,l,cvae) = softmax((masked * (1 + ratio) + g) / tau) @ codebook,
 * g = -log(E), E ~ Exp(1): e_noise (B,l,V) explicit, or NULL for the Philox stream at (seed, draw, image_offset).  `masked_logits`
 * (B,l,V) are the CFG logits as sample_with_top_k_top_p_ leaves them (helpers.py:10,15 mask in place): sdvar_cfg_sample's dbg_masked. */
int sdvar_gumbel_mix(sdvar_quant_t* q, const float* masked_logits, int32_t B, int32_t l, double ratio, double tau, const float* e_noise,
                     uint64_t seed, uint32_t draw, uint32_t image_offset, float* h_out, void* stream);

/* ---- VQVAE decoder: f_hat -> image (the caller side of the sampler) ------------------ */
typedef struct {
    int32_t ch;                             /* base width (160 for vae_ch160v4096z32)          models/vqvae.py:30-33 */
    int32_t z_channels;                     /* Cvae = 32 */
    int32_t n_mult;                         /* number of resolution levels */
    int32_t ch_mult[8];                     /* (1, 1, 2, 2, 4)                                 models/vqvae.py:31 */
    int32_t num_res_blocks;                 /* 2: every decoder level has num_res_blocks + 1 ResnetBlocks (basic_vae.py:196) */
    int32_t max_batch;
    int32_t latent_hw;                      /* side of f_hat: 16 for 256^2 images, 32 for 512^2 */
    int32_t plane_format;                   /* operands of the convolutions: 0 or 2 = f16x2 (two fp16 planes, three MFMA products), 3 = bf16x3 */
} sdvar_vae_desc;
int sdvar_vae_create(const sdvar_vae_desc* desc /*host*/, sdvar_vae_t** out /*host*/);
int sdvar_vae_destroy(sdvar_vae_t* v);
/* number of tensors sdvar_vae_bind expects for this descriptor */
int sdvar_vae_tensor_count(const sdvar_vae_desc* desc /*host*/);
/* Host array of device pointers to the fp32 state_dict tensors, in execution order (weight then bias each):
 * post_quant_conv; decoder.conv_in; decoder.mid.block_1 {norm1, conv1, norm2, conv2}; decoder.mid.attn_1 {norm, qkv,
 * proj_out}; decoder.mid.block_2; then for level = n_mult-1 .. 0: for i = 0 .. num_res_blocks: up.level.block.i {norm1,
 * conv1, norm2, conv2, [nin_shortcut if the width changes]}, [up.level.attn.i at the top level]; [up.level.upsample.conv
 * for level > 0]; decoder.norm_out; decoder.conv_out.   (models/basic_vae.py:163-226)
 * Conv weights are re-packed into bf16x3 planes (owned copies); biases and GroupNorm affine tensors stay borrowed. */
int sdvar_vae_bind(sdvar_vae_t* v, const float* const* tensors /*host*/, int32_t n_tensors, void* stream);
/* vqvae.py:62-63: img (B,3,H,W) = clamp(decoder(post_quant_conv(f_hat (B,Cvae,h,w))), -1, 1), H = h << (n_mult-1) */
int sdvar_vae_decode(sdvar_vae_t* v, const float* f_hat, int32_t B, float* img, void* stream);

/* ---- sampling / acceptance -------------------------------------------------------------------------------------- */
/* var.py:199-202 + helpers.py:6-19: CFG with t = cfg*si/(S-1), top-k, top-p, draw = argmax(p/q).
 * q: explicit Exp(1) noise (B*l, V), or NULL to generate the Philox stream of sdvar_amd/noise.py in-kernel from
 * (seed, draw, image_offset).  ids_out[b*ids_stride + tok].  dbg_masked: optional (B,l,V) masked logits. */
int sdvar_cfg_sample(const float* logits, int32_t B, int32_t l, int32_t V, double t, int32_t top_k, double top_p, const float* q,
                     uint64_t seed, uint32_t draw, uint32_t image_offset, int64_t* ids_out, int32_t ids_stride, float* dbg_masked,
                     void* stream);
/* var.py:1062-1067 + 1199-1222 over a verified chunk: per stage CFG (t[j], host doubles) -> argmax_V -> compare with
 * the draft ids -> matched count; n_accept = leading stages with float32 match rate >= thr.
 * counts (40 x int32, device): [0..16) matched per stage, [16] n_accept, [17..33) tokens per stage. */
int sdvar_verify_accept(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n_stages, const int32_t* stage_lens /*host*/,
                        const double* t /*host*/, const int64_t* draft_ids, int32_t ids_stride, double thr, int32_t* counts,
                        int64_t* argmax_out, void* stream);

/* The richer acceptance rules sketched in SDVAR.advanced_token_matching (var.py:1229-1243).  rule 0: draft id == argmax (as above);
 * 1: draft id among the target's top `match_top_k` (fewer than k entries score strictly higher); 2: KL(softmax target || softmax draft)
 * <= kl_thr, with the draft's raw logits of the chunk in draft_logits (stage j stored as (2B, l_j, V) at element offset 2*B*V*qbeg_j).
 * match_out (B, lsum) u8: per-token verdict; corrected_out (B, lsum): draft id where the rule holds, the target's argmax elsewhere
 * (token-level partial acceptance).  Any of argmax_out / match_out / corrected_out may be NULL. */
int sdvar_verify_accept_ex(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n_stages, const int32_t* stage_lens /*host*/,
                           const double* t /*host*/, const int64_t* draft_ids, int32_t ids_stride, double thr, int32_t rule, int32_t match_top_k,
                           double kl_thr, const float* draft_logits, int32_t* counts, int64_t* argmax_out, uint8_t* match_out,
                           int64_t* corrected_out, void* stream);

/* var.py:1062-1067 alone: out (B, lsum, V) = (1 + t_j) * logits[b] - t_j * logits[B + b] for every stage j of a verified chunk (float32 roundings of torch's
 * scalar ops): the per-stage CFG logits SDVAR.target_verify_batch returns to a caller that drives the reference's step functions itself. */
int sdvar_cfg_combine(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n_stages, const int32_t* stage_lens /*host*/, const double* t /*host*/,
                      float* out, void* stream);

/* ---- single operators (kernel-level parity tests and micro-benchmarks) ---------------------------------------------- */
/* out[M,N] = epi(X[M,K] W[N,K]^T + bias); epi 0 bias, 1 bias+GELU(tanh), 2 res + (.)*gate[row / rows_per_gate] */
int sdvar_op_gemm(const float* X, int32_t ldx, const float* W, const float* bias, float* out, int32_t ldo, int32_t M, int32_t N, int32_t K,
                  int32_t epilogue, const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream);
/* bf16x3 plane tensors are K-blocked: element (row, k) of plane p is at p*plane_stride + ((k/32)*rows + row)*32 + k%32.
 * out (fp32) or out_planes (planes of the (rows, C) result, plane stride in elements) */
int sdvar_op_ln_modulate(const float* x, const float* scale, const float* shift, float* out, uint16_t* out_planes, uint64_t plane_stride,
                         int32_t plane_format /* 3 = bf16x3, 2 = f16x2 */, int32_t rows, int32_t C, int32_t rows_per_img, int32_t mod_stride, void* stream);
/* fp32 (rows, cols) row-major -> three K-blocked bf16 planes of 8 significand bits each, x == p0 + p1 + p2 exactly */
int sdvar_op_split_planes(const float* x, uint16_t* planes, int32_t rows, int32_t cols, uint64_t plane_stride, void* stream);
/* the bf16x3 split-operand GEMM on K-blocked planes of X (M,K) and W (N,K); epi 0 bias -> out, 1 bias+GELU -> out_planes of (M,N), 2 gated residual -> out */
int sdvar_op_gemm_bf16x3(const uint16_t* Xp, uint64_t x_plane_stride, const uint16_t* Wp, uint64_t w_plane_stride, const float* bias, float* out,
                         int32_t ldo, uint16_t* out_planes, uint64_t out_plane_stride, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                         const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream);
/* f16x2 operands: fp32 (rows, cols) -> two K-blocked fp16 planes of x * 2^S; scale (4 device floats, may be NULL = no scaling) receives
 * {2^S, 2^-S, scratch, -} with max|x| 2^S in (2^12, 2^13] (weights); the GEMM takes the same pointer and undoes the scale in its epilogue */
int sdvar_op_split_planes_f16(const float* x, uint16_t* planes, int32_t rows, int32_t cols, uint64_t plane_stride, float* scale, void* stream);
int sdvar_op_gemm_f16x2(const uint16_t* Xp, uint64_t x_plane_stride, const uint16_t* Wp, uint64_t w_plane_stride, const float* w_scale, const float* bias, float* out,
                        int32_t ldo, uint16_t* out_planes, uint64_t out_plane_stride, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                        const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream);
/* Single-operator entry points of the attention path (models/basic_var.py:101-117).  kv_f16 = cache format:
 *   0  fp32   K, V (R, H, Lmax, 64)                    1  fp16, same shape (the reference's half-precision cache)
 *   2  bf16x3 planes: K (R, H, 3, Lmax, 64), V^T (R, H, 3, 64, Lmax) with bits 2 and 3 of the key position swapped inside every 16 keys (attention_bf16x3.hip)
 *   3  f16x2 planes:  K AND V (R, H, 2, Lmax, 64) fp16, high plane then low plane, one 128-byte row per position (gemm mode f16x2, the default)
 *   4  one fp16 plane: K and V (R, H, 1, Lmax, 64) = the fp16 KV cache of BASELINE config P4 in the layout of format 3
 * Formats 2-4 need Lmax % 64 == 0 and a zero-initialised cache (whole 32-key tiles are streamed; rows past the valid keys must be finite). */
int sdvar_op_qk_norm_append(const float* qkv, const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int32_t kv_f16, int32_t R,
                            int32_t l, int32_t H, int32_t Lmax, int32_t pos0, void* stream);
/* q (R,H,l,64), caches in format kv_f16 (above) with Ktot valid keys, out (R,l,H*64) fp32 or out_planes (K-blocked operand planes); queries >= qbeg[j] see keys < vis[j] */
int sdvar_op_attention(const float* q, const void* k_cache, const void* v_cache, int32_t kv_f16, float* out, uint16_t* out_planes, uint64_t plane_stride,
                       int32_t plane_format, int32_t R, int32_t H, int32_t l, int32_t Lmax, int32_t Ktot, int32_t n_stages, const int32_t* qbeg /*host*/,
                       const int32_t* vis /*host*/, void* stream);
int sdvar_op_noise_fill(float* q, int32_t B, int32_t l, int32_t V, uint64_t seed, uint32_t draw, uint32_t image_offset, void* stream);

/* conv weight (Cout, Cin, kh, kw) with kh*kw = taps (1 or 9) -> K-blocked planes [3][taps*Cin/32][Cout][32], k = tap*Cin + cin */
int sdvar_op_conv_weight_planes(const float* w, uint16_t* planes, int32_t Cout, int32_t Cin, int32_t taps, uint64_t plane_stride, int32_t plane_format /* 3 | 2 */,
                                float* scale /* f16x2: receives {2^S, 2^-S, ..} (4 device floats), may be NULL */, void* stream);
/* fp32 channel-last rows [B H W][C] of a (B,C,H,W) tensor -> planes [3][C/32][guard + B(Ho+2)(Wo+2) + guard][32] of the
 * (B,C,H<<up,W<<up) tensor over padded pixel rows (prow(b,y,x) = (b(Ho+2)+y+1)(Wo+2)+x+1, zero frame and guards).
 * mode bit 0: GroupNorm(32 groups) with stats (B,32,{mean,rstd}), gamma, beta; bit 1: SiLU. */
int sdvar_op_vae_prep(const float* in, const float* stats, const float* gamma, const float* beta, uint16_t* planes, uint64_t plane_stride, int32_t plane_format, int32_t B,
                      int32_t C, int32_t H, int32_t W, int32_t up, int32_t mode, int32_t guard, void* stream);
/* out[B H W][N] = conv(x planes of a (B,Cin,H,W) tensor, w planes) + bias (+ res[B H W][N]); taps = 9: 3x3 pad 1, taps = 1: 1x1.  x_row0 =
 * guard rows (>= W+3).  workspace: split-K slabs (may be NULL: no split); force_split > 0 overrides the heuristic. */
int sdvar_op_conv_planes(const uint16_t* x_planes, uint64_t x_plane_stride, uint64_t x_rows, int32_t x_row0, const uint16_t* w_planes, uint64_t w_plane_stride,
                         int32_t plane_format, const float* w_scale, const float* bias, const float* res, float* out, int32_t B, int32_t H, int32_t W, int32_t N,
                         int32_t Cin, int32_t taps, float* workspace, uint64_t workspace_floats, int32_t force_split, void* stream);
/* tuning / test aid (tools/gemm_bench.py --sweep): force the GEMM row tile (32/64/128/256) and K-slice count; 0 = automatic.  f16x2 mode only: bm 512 = the
 * 256 x 256 tile kernel; bm 256 with split = -T forces the hybrid tail split T ways on shapes that have a partial last round. */
int sdvar_debug_set_gemm_cfg(int32_t bm, int32_t split);
/* test aid: 0 = the QKV launch of sdvar_stage_forward never finishes q and k in its epilogue (qk_norm_append does all three), 1 (default) = it does
 * whenever the launch comes out unsplit on the f16x2 planes cache */
int sdvar_debug_set_qkv_fuse(int32_t on);
/* 1 (default): stage_forward calls with 32 .. 80 rows (stage 1, the first verify chunk at B = 8) in GEMM mode f16x2 run five launches per transformer block - LayerNorm +
 * modulation in the operand prologue of the QKV / fc1 / head launch, q / k / v finished by the QKV launch, fc2 unsplit (csrc/gemm_f16x2.hip gemm_f16x2_rowblk_kernel);
 * 0: the launch sequence of every other row count (ln_modulate, GEMM, qk_norm_append: eight launches); 2: five launches at every row count up to 80 (below 32 rows they are SLOWER than the eight: measured).
 * Test / A-B aid (SDVAR_ROWBLK in the environment does the same). */
int sdvar_debug_set_rowblk(int32_t on);
/* The row-block launch alone (M <= 80).  x != NULL: out = epi( (LayerNorm(x; eps 1e-6)(1 + scale[g]) + shift[g]) W^T + bias ), g = row / rows_per_img, K <= 1024, epi 0 (-> out) or
 * 1 (GELU -> out_planes); with q_out != NULL the QKV finish instead (N = 3 H 64; q -> (R, H, l, 64) fp32, k normalised / v -> the cache planes at pos0 + t, kv_fmt 3 | 4) and nothing
 * goes to `out`.  x == NULL: the operand is Xp (K-blocked f16x2 planes, K <= 4096), epi 0 or 2 (gated residual). */
int sdvar_op_gemm_rowblk(const float* x, int32_t ldx, const float* scale, const float* shift, int32_t rows_per_img, int32_t mod_stride, const uint16_t* Xp, uint64_t x_plane_stride,
                         const uint16_t* Wp, uint64_t w_plane_stride, const float* w_scale, const float* bias, float* out, int32_t ldo, uint16_t* out_planes, uint64_t out_plane_stride,
                         int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride,
                         const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int32_t l, int32_t H, int32_t Lp, int32_t pos0, int32_t kv_fmt, void* stream);
/* Select a kernel variant that otherwise only an environment variable (read at first use) selects - tests run the non-default variants in one process.
 * name: "gemm_h4_var" 0..3, "gemm_h2_stages" 2..6, "gemm_small_pp" 0..2, "attn_pp_sched" 0..3, "conv_pp" 0..2; value < 0 restores the environment / default. */
int sdvar_debug_set_variant(const char* name, int32_t value);
/* test aid: out4 = {row tile (32/64/128/256) of the LAST f16x2 GEMM call of this host thread, its K split, number of launches that took the hybrid tail
 * split since the last read, number of QKV launches that finished q and k in their epilogue since the last read}; reading resets the two counters.
 * Tests assert with it that the path they mean to cover is the one that ran. */
int sdvar_debug_get_gemm_cfg(int32_t* out4);
/* f16x2 guard (debug, off by default; mode f16x2 only).  The f16x2 operand format of the default GEMM mode saturates finite activations at +-65504 and loses
 * relative precision below ~1e-3 (the reference computes these GEMMs in fp32: basic_var.py:44-52, 87-119).  With the guard on, every producer of GEMM operand
 * planes inside sdvar_stage_forward (ln_modulate, attention, the fc1 GELU epilogue) is followed by a counting pass over the plane it wrote.
 * sdvar_debug_get_f16x2_guard synchronises the device and returns out4 = {elements seen, saturated (|h| == 65504), NaN / Inf, tiny (0 < |h| < 2^-10)};
 * reset != 0 zeroes the counters.  A non-zero `saturated` or `NaN` count means the run left the range f16x2 is exact in: use gemm_mode bf16x3 (no range limit). */
int sdvar_debug_set_f16x2_guard(int32_t on);
int sdvar_debug_get_f16x2_guard(uint64_t* out4, int32_t reset);
/* diagnostic: per-workgroup stamps of the LDS-DMA GEMM kernels; NULL disables.  bf16x3 kernel: 4 x u64 per workgroup (s_memtime at entry, main loop
 * start, main loop end, exit); f16x2 small-M and 128 x 128 kernels: 8 x u64 (s_memrealtime at entry, first K-step landed, loop end, exit; then s_memtime) */
int sdvar_debug_set_gemm_stamps(uint64_t* stamps);

/* ---- per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg) ----------------------------- */
#define SDVAR_PROF_CLASSES 10  /* 0 gemm with M >= 1024 rows (matrix-pipe regime), 1 attention with more than 36 queries per (row, head) (matrix-pipe bound),
                                  2 ln_modulate, 3 qk_norm_append, 4 sampler, 5 verify, 6 quant, 7 embed/misc, 8 attention with <= 36 queries (stages 0-5: HBM /
                                  latency bound), 9 gemm with M < 1024 rows (stages 0-5, adaLN hoist: weight-streaming / launch-latency regime) */
int sdvar_prof_enable(int32_t on);
/* synchronises the recorded events and accumulates: ms, launches, algorithmic flops, algorithmic bytes per class */
int sdvar_prof_collect(double* ms /*host[SDVAR_PROF_CLASSES]*/, int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SDVAR_HIP_H */
