// C ABI of libsdvar_hip.so (declared in include/sdvar_hip.h): model / quantizer objects, the stage-forward driver that
// strings the kernels together on one stream, and the single-operator entry points used by the parity tests.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "../../include/sdvar_hip.h"
#include "common.h"
#include <string>

namespace sdvar {

// kernels (gemm.hip, elementwise.hip, attention.hip, sampler.hip, quant.hip)
int gemm_f32_nt(const float* X, int ldx, const float* W, const float* bias, float* out, int ldo, int M, int N, int K, int epi,
                const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, hipStream_t stream);
struct PendingSplitK { const float* ws; const float* bias; const float* gate; int split, rows_per_gate, gate_stride; };
int ln_modulate(float* x, const float* scale, const float* shift, float* out, uint16_t* outp, size_t ops, int rows, int C, int rows_per_img, int mod_stride, const PendingSplitK* pend, int pfmt, hipStream_t stream);
float* splitk_workspace(size_t* floats);
float* set_splitk_workspace(float* p);
size_t splitk_workspace_floats();
int qk_norm_append(const float* qkv, const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int kv_f16, int R, int l, int H, int Lmax, int pos0, const PendingSplitK* pend, int v_only, hipStream_t stream);
int gemm_f16x2_qkv(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, int M, int N, int K,
                   const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int l, int H, int Lp, int pos0, int kv_fmt, int* defer, int* fused, hipStream_t stream);
bool gemm_f16x2_rowblk_ok(int M, int N, int K, int ln, int qkv);
bool gemm_f16x2_rowblk_want(int M, int C, int V, int rows_per_img);
int gemm_f16x2_rowblk(const float* x, int ldx, const float* scale, const float* shift, int rows_per_img, int mod_stride, const uint16_t* X, size_t xps,
                      const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops, int M, int N, int K, int epi,
                      const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride,
                      const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int l, int H, int Lp, int pos0, int kv_fmt, hipStream_t stream);
void debug_set_rowblk(int v);
int silu_rows(const float* x, float* y, int n, hipStream_t stream);
int add_row_vector(const float* src, const float* vec, float* out, int rows, int cols, hipStream_t stream);
int ada_gather(const long long* labels, const float* tab, size_t row_floats, int depth, int C, float* ada, size_t blk_stride, float* ada_head, int B, int num_classes,
               hipStream_t stream);
int prologue(const long long* labels, const float* cond_in, const float* class_emb, const float* pos_start, const float* lvl_pos, float* cond, float* x0, int B, int C, int num_classes, hipStream_t stream);
int build_lvl_pos(const float* lvl_embed, const float* pos, const int* stage_of_tok, float* out, int L, int C, hipStream_t stream);
int embed_next(const float* nxt, const float* Ww, const float* bw, const float* lvl_pos, float* x, int B, int l, int C, int t0, int ltot, int tok_off, hipStream_t stream);
int attention_f32(const float* q, const void* kc, const void* vc, int kv_f16, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l, int Lmax, int Ktot, int n_chunk, const int* qbeg, const int* vis, hipStream_t stream);
int gemm_bf16x3_nt(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops,
                   int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, int* defer, hipStream_t stream);
int attention_masked(const float* q, const void* kc, const void* vc, int fmt, const float* bias, float* out, uint16_t* outp, size_t ops, int pfmt, int R, int H, int l,
                     int Lmax, int Ktot, hipStream_t stream);
int split_planes(const float* x, uint16_t* planes, int rows, int cols, size_t plane_stride, hipStream_t stream);
int gemm_f16x2_nt(const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsi, const float* bias, float* out, int ldo, uint16_t* outp, size_t ops,
                  int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride, int* defer, hipStream_t stream);
int split_planes_f16(const float* x, uint16_t* planes, int rows, int cols, size_t plane_stride, const float* scale, hipStream_t stream);
int weight_scale_f16(const float* w, size_t n, float* sc, hipStream_t stream);
void debug_set_gemm_cfg_h(int bm, int split);
void debug_set_gemm_cfg_p(int bm, int split);
void debug_set_gemm_stamps(unsigned long long* p);
void debug_set_qkv_fuse(int on);
void debug_set_h4_var(int v);
void debug_set_h2_stages(int v);
void debug_set_small_pp(int v);
void debug_set_attn_pp_sched(int v);
void debug_set_conv_pp(int v);
void debug_get_gemm_cfg_h(int* out);
int planes_guard(const uint16_t* h_plane, size_t n, unsigned long long* cnt, hipStream_t stream);
int cfg_sample(const float* logits, int B, int l, int V, float one_plus_t, float t, int top_k, int use_top_p, float top_p_thr, const float* q, uint64_t seed,
               uint32_t draw, uint32_t image_offset, long long* ids, int ids_stride, float* dbg_masked, hipStream_t stream);
int noise_fill(float* q, int B, int l, int V, uint64_t seed, uint32_t draw, uint32_t image_offset, hipStream_t stream);
int verify_accept(const float* logits, int B, int lsum, int V, int n_chunk, const int* qbeg, const float* one_plus_t, const float* t, const long long* draft_ids,
                  int ids_stride, double thr, int mode, int top_k, float kl_thr, const float* draft_logits, const long long* dl_off, int* counts,
                  long long* argmax_out, unsigned char* match_out, long long* corrected_out, hipStream_t stream);
int cfg_combine(const float* logits, int B, int lsum, int V, int n_chunk, const int* qbeg, const float* one_plus_t, const float* t, float* out, hipStream_t stream);
int gumbel_mix(const float* masked, int B, int l, int V, float scale, float tau, const float* e, uint64_t seed, uint32_t draw, uint32_t image_offset,
               const float* codebook, int Cv, float* h, hipStream_t stream);
void debug_set_gemm_cfg(int bm, int split);
int quant_next(const long long* ids, int ids_stride, const float* hvec, const float* codebook, const float* Wup, const float* phi_w, const float* phi_b, const float* Wdn,
               float* up_scratch, const float* f_in, float* f_hat, float* nxt, int B, int pn, int pn_next, int HW, int Cv, int last, hipStream_t stream);

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_GATED_RES = 2 };

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- event timing per kernel class -----------------------------------------------------------------------------------
struct ProfRec { int cls; hipEvent_t e0, e1; double flops, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static double g_ms[SDVAR_PROF_CLASSES], g_fl[SDVAR_PROF_CLASSES], g_by[SDVAR_PROF_CLASSES];
static long long g_n[SDVAR_PROF_CLASSES];

struct ProfScope {
    bool on; ProfRec r; hipStream_t s;
    ProfScope(int cls, double flops, double bytes, hipStream_t stream) : on(g_prof_on), s(stream) {
        if (!on) return;
        r.cls = cls; r.flops = flops; r.bytes = bytes;
        (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, s);
        g_prof.push_back(r);
    }
};

// f16x2 guard (debug switch, off by default): see elementwise.hip planes_guard_kernel
static bool g_guard_on = false;
static unsigned long long* g_guard_cnt = nullptr;       // device, 4 counters
static int guard_planes(const uint16_t* h_plane, size_t n, hipStream_t s) {
    if (!g_guard_on || !g_guard_cnt) return SDVAR_OK;
    return planes_guard(h_plane, n, g_guard_cnt, s);
}

struct WsScope {         // routes the split-K slabs of every GEMM below to the model's own workspace for the duration of a call
    float* old;
    explicit WsScope(float* p) : old(set_splitk_workspace(p)) {}
    ~WsScope() { set_splitk_workspace(old); }
};

template <typename T>
static int dmalloc(T** p, size_t n) {
    *p = nullptr;
    SDVAR_HIP(hipMalloc((void**)p, n * sizeof(T)));
    return SDVAR_OK;
}
#define SDVAR_TRY(call) do { int rc_ = (call); if (rc_ != SDVAR_OK) return rc_; } while (0)

struct BlockW {
    const float *ada_w, *ada_b, *qkv_w, *scale_mul, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    float* qkv_bias;     // owned: [q_bias, 0, v_bias]  (basic_var.py:93)
    void *kc, *vc;       // owned KV cache: (Rmax, H, L, 64) fp32 or fp16 (desc.kv_dtype), or the bf16x3 planes of attention_bf16x3.hip (kv_fmt 2)
    uint16_t *qkv_wp, *proj_wp, *fc1_wp, *fc2_wp;   // owned operand planes of the weights (desc.gemm_mode 1: 3 bf16 planes, 2: 2 fp16 planes)
    float* wsc;          // owned, gemm_mode 2: per weight {2^S, 2^-S, scratch, -} x {qkv, proj, fc1, fc2}, the power-of-two weight scales of gemm_f16x2.hip
    bool bound;
};

}  // namespace sdvar

using namespace sdvar;

struct sdvar_model {
    sdvar_model_desc d;
    int C, H, L, Rmax, lmax, S;
    int kv_fmt, Lkv;     // cache format handed to the kernels (0 fp32, 1 fp16, 2 planes) and its row capacity
    const float *shared_w = nullptr, *shared_b = nullptr;   // shared_aln=True: shared_ada_lin.1.{weight (6C,C), bias} (var.py:16-19, 81)
    float* gss_lin = nullptr;  // owned (Rmax, 6C): shared_ada_lin(cond) of the current call
    // owned (num_classes + 1, depth * 6C + 2C) + scratch: the adaLN parameters of EVERY class (ada_lin / head_nm.ada_lin of silu(class_emb)), built by the first
    // sdvar_model_begin after a bind with the GEMMs that call used to run (30 launches streaming 25 MB of fp32 weights each, 1.4 ms per d12 + d16 step); a call
    // gathers its 2B rows.  0.4 GB for d16, 1.4 GB for d30 - HBM the card has.  Not for shared_aln models (one GEMM per call there) nor for sdvar_model_begin_cond.
    float* ada_tab = nullptr;
    bool ada_tab_ready = false;
    float* ws_own = nullptr;   // this model's split-K slabs: two models of one host thread may run on different streams
    int lens[SDVAR_MAX_STAGES], cum[SDVAR_MAX_STAGES];
    // borrowed
    const float *class_emb, *pos_start, *word_w, *word_b, *nm_w, *nm_b, *head_w, *head_b;
    std::vector<BlockW> blk;
    bool embed_bound, head_bound;
    // owned
    float *lvl_pos, *cond, *cond_silu, *x0, *ada, *ada_head;
    float *xn, *qkv, *qbuf, *att, *hid;
    uint16_t *xn_p, *att_p, *hid_p, *head_wp;     // operand planes of the GEMM inputs (desc.gemm_mode >= 1)
    float* head_wsc = nullptr;                    // gemm_mode 2: weight scale of the head
    size_t act_ps;                                // plane stride of xn_p / att_p (elements); hid_p uses 4x
    int* stage_of_tok;
    // run state
    int B, kv_len;
    int kv_origin;       // token position of the cache's first key (0 except in the hand-off sampler: sdvar_kv_set_origin)
    bool begun;
};

struct sdvar_quant {
    int S, Cv, V, maxB, HW, n_phi;
    int pn[SDVAR_MAX_STAGES], phi_of[SDVAR_MAX_STAGES];
    float* Wup[SDVAR_MAX_STAGES];
    float* Wdn[SDVAR_MAX_STAGES];
    std::vector<std::vector<float>> hWup, hWdn;
    float* up_scratch;
    const float* codebook;
    const float* phi_w[SDVAR_MAX_STAGES];
    const float* phi_b[SDVAR_MAX_STAGES];
    bool bound;
};

static inline int begin_of(const sdvar_model* m, int s) { return s == 0 ? 0 : m->cum[s - 1]; }

// the split-operand GEMM of the model's mode: bf16x3 (six bf16 products) or f16x2 (three fp16 products, weights scaled by 2^S: wsc -> {2^S, 2^-S})
static int plane_gemm(const sdvar_model* m, const uint16_t* X, size_t xps, const uint16_t* W, size_t wps, const float* wsc, const float* bias, float* out, int ldo,
                      uint16_t* outp, size_t ops, int M, int N, int K, int epi, const float* res, int ldres, const float* gate, int rows_per_gate, int gate_stride,
                      int* defer, hipStream_t s) {
    if (m->d.gemm_mode == 2) return gemm_f16x2_nt(X, xps, W, wps, wsc + 1, bias, out, ldo, outp, ops, M, N, K, epi, res, ldres, gate, rows_per_gate, gate_stride, defer, s);
    return gemm_bf16x3_nt(X, xps, W, wps, bias, out, ldo, outp, ops, M, N, K, epi, res, ldres, gate, rows_per_gate, gate_stride, defer, s);
}

extern "C" {

int sdvar_abi_version(void) { return SDVAR_ABI_VERSION; }
const char* sdvar_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------- model object
int sdvar_model_create(const sdvar_model_desc* desc, sdvar_model_t** out) {
    SDVAR_CHECK_ARG(desc && out, "model_create: null argument");
    SDVAR_CHECK_ARG(desc->depth >= 1 && desc->depth <= 64, "model_create: depth %d", desc->depth);
    SDVAR_CHECK_ARG(desc->n_stages >= 1 && desc->n_stages <= SDVAR_MAX_STAGES, "model_create: n_stages %d", desc->n_stages);
    SDVAR_CHECK_ARG(desc->vocab > 0 && desc->vocab <= 4096 && desc->vocab % 4 == 0, "model_create: vocab %d", desc->vocab);
    SDVAR_CHECK_ARG(desc->cvae == 32, "model_create: cvae must be 32 (got %d)", desc->cvae);
    SDVAR_CHECK_ARG(desc->max_batch >= 1 && desc->num_classes >= 1, "model_create: batch/classes");
    SDVAR_CHECK_ARG(desc->kv_dtype == 0 || desc->kv_dtype == 1, "model_create: kv_dtype %d (0 = fp32, 1 = fp16)", desc->kv_dtype);
    SDVAR_CHECK_ARG(desc->gemm_mode >= 0 && desc->gemm_mode <= 2, "model_create: gemm_mode %d (0 = fp32 MFMA, 1 = bf16x3, 2 = f16x2 split operands)", desc->gemm_mode);
    SDVAR_CHECK_ARG(desc->max_chunk_stages >= 1 && desc->max_chunk_stages <= SDVAR_MAX_STAGES, "model_create: max_chunk_stages %d", desc->max_chunk_stages);
    sdvar_model* m = new sdvar_model();
    struct Guard {          // every early return below (argument error, out of memory) releases what was allocated so far
        sdvar_model* m; bool keep = false;
        ~Guard() { if (!keep) sdvar_model_destroy(m); }
    } guard{m};
    m->lvl_pos = m->cond = m->cond_silu = m->x0 = m->ada = m->ada_head = m->xn = m->qkv = m->qbuf = m->att = m->hid = nullptr;
    m->xn_p = m->att_p = m->hid_p = m->head_wp = nullptr; m->stage_of_tok = nullptr;
    m->d = *desc;
    m->C = 64 * desc->depth; m->H = desc->depth; m->S = desc->n_stages; m->Rmax = 2 * desc->max_batch;
    int c = 0;
    for (int s = 0; s < m->S; ++s) {
        if (desc->patch_nums[s] < 1 || desc->patch_nums[s] > 64) { set_error("model_create: patch_nums[%d]=%d", s, desc->patch_nums[s]); return SDVAR_ERR_ARG; }
        m->lens[s] = desc->patch_nums[s] * desc->patch_nums[s]; c += m->lens[s]; m->cum[s] = c;
    }
    m->L = c;
    // the split-operand GEMM mode keeps an fp32 cache as exact bf16 planes so that attention runs on the bf16 matrix cores too
    // cache format handed to the kernels: the split-operand GEMM modes keep an fp32 cache as operand planes of the matching matrix-core
    // format (2: three bf16 planes, 3: two fp16 planes) so that attention runs on the matrix cores too; the fp16 cache of config P4 is ONE
    // fp16 plane (4) in the same layout.  SDVAR_ATTN_F32 (A/B runs only) keeps the plain layouts and the fp32-MFMA kernel.
    const bool planes_ok = desc->gemm_mode >= 1 && !getenv("SDVAR_ATTN_F32");
    m->kv_fmt = !planes_ok ? desc->kv_dtype : desc->kv_dtype == 1 ? 4 : (desc->gemm_mode == 2 && !getenv("SDVAR_ATTN_BF16X3")) ? 3 : 2;
    m->Lkv = (m->kv_fmt >= 2) ? (c + 63) / 64 * 64 : c;
    // the largest chunk is a window of max_chunk_stages consecutive stages; stage lengths are non-decreasing
    m->lmax = 0;
    for (int s = 0; s < m->S; ++s) {
        int t = 0;
        for (int j = s; j < m->S && j < s + desc->max_chunk_stages; ++j) t += m->lens[j];
        if (t > m->lmax) m->lmax = t;
    }
    m->blk.resize(desc->depth);
    for (auto& b : m->blk) memset(&b, 0, sizeof(b));
    m->embed_bound = m->head_bound = m->begun = false;
    m->B = 0; m->kv_len = 0; m->kv_origin = 0;
    const size_t C = m->C, R = m->Rmax, M = R * (size_t)m->lmax;
    std::vector<int> sot(m->L);
    for (int s = 0, t = 0; s < m->S; ++s) for (int i = 0; i < m->lens[s]; ++i) sot[t++] = s;
    SDVAR_TRY(dmalloc(&m->stage_of_tok, (size_t)m->L));
    SDVAR_HIP(hipMemcpy(m->stage_of_tok, sot.data(), sizeof(int) * m->L, hipMemcpyHostToDevice));
    SDVAR_TRY(dmalloc(&m->lvl_pos, (size_t)m->L * C));
    SDVAR_TRY(dmalloc(&m->cond, R * C));
    SDVAR_TRY(dmalloc(&m->cond_silu, R * C));
    SDVAR_TRY(dmalloc(&m->x0, R * C));
    SDVAR_TRY(dmalloc(&m->ada, (size_t)desc->depth * R * 6 * C));
    SDVAR_TRY(dmalloc(&m->ada_head, R * 2 * C));
    SDVAR_TRY(dmalloc(&m->xn, M * C));
    SDVAR_TRY(dmalloc(&m->qkv, M * 3 * C));
    SDVAR_TRY(dmalloc(&m->qbuf, M * C));
    SDVAR_TRY(dmalloc(&m->att, M * C));
    m->xn_p = m->att_p = m->hid_p = m->head_wp = nullptr; m->act_ps = M * C;
    const size_t NPL = desc->gemm_mode == 2 ? 2 : 3;                  // planes per operand
    if (desc->gemm_mode >= 1) {
        SDVAR_TRY(dmalloc(&m->xn_p, NPL * M * C));
        SDVAR_TRY(dmalloc(&m->att_p, NPL * M * C));
        SDVAR_TRY(dmalloc(&m->hid_p, NPL * M * 4 * C));
        SDVAR_TRY(dmalloc(&m->head_wp, NPL * (size_t)desc->vocab * C));
        if (desc->gemm_mode == 2) SDVAR_TRY(dmalloc(&m->head_wsc, (size_t)4));
    } else {
        SDVAR_TRY(dmalloc(&m->hid, M * 4 * C));
    }
    for (auto& b : m->blk) {
        if (desc->gemm_mode >= 1) {
            SDVAR_TRY(dmalloc(&b.qkv_wp, NPL * 3 * C * C)); SDVAR_TRY(dmalloc(&b.proj_wp, NPL * C * C));
            SDVAR_TRY(dmalloc(&b.fc1_wp, NPL * 4 * C * C)); SDVAR_TRY(dmalloc(&b.fc2_wp, NPL * 4 * C * C));
            if (desc->gemm_mode == 2) SDVAR_TRY(dmalloc(&b.wsc, (size_t)16));
        }
        SDVAR_TRY(dmalloc(&b.qkv_bias, 3 * C));
        const size_t kvb = R * (size_t)m->H * m->Lkv * 64 * (m->kv_fmt == 2 ? 6 : (m->kv_fmt == 1 || m->kv_fmt == 4) ? 2 : 4);
        SDVAR_HIP(hipMalloc(&b.kc, kvb));
        SDVAR_HIP(hipMalloc(&b.vc, kvb));
        SDVAR_HIP(hipMemset(b.kc, 0, kvb));        // the planes kernel streams whole 64-key tiles: rows past kv_len must be finite
        SDVAR_HIP(hipMemset(b.vc, 0, kvb));
    }
    SDVAR_TRY(dmalloc(&m->ws_own, splitk_workspace_floats()));
    guard.keep = true;
    *out = m;
    return SDVAR_OK;
}

int sdvar_model_destroy(sdvar_model_t* m) {
    if (!m) return SDVAR_OK;
    float* bufs[] = {m->lvl_pos, m->cond, m->cond_silu, m->x0, m->ada, m->ada_head, m->xn, m->qkv, m->qbuf, m->att, m->hid, m->ws_own, m->gss_lin, m->head_wsc, m->ada_tab};
    for (float* p : bufs) if (p) (void)hipFree(p);
    if (m->stage_of_tok) (void)hipFree(m->stage_of_tok);
    uint16_t* pb[] = {m->xn_p, m->att_p, m->hid_p, m->head_wp};
    for (uint16_t* p : pb) if (p) (void)hipFree(p);
    for (auto& b : m->blk) { uint16_t* wp[] = {b.qkv_wp, b.proj_wp, b.fc1_wp, b.fc2_wp}; for (uint16_t* p : wp) if (p) (void)hipFree(p); }
    for (auto& b : m->blk) { if (b.qkv_bias) (void)hipFree(b.qkv_bias); if (b.kc) (void)hipFree(b.kc); if (b.vc) (void)hipFree(b.vc); if (b.wsc) (void)hipFree(b.wsc); }
    delete m;
    return SDVAR_OK;
}

int sdvar_model_bind_embed(sdvar_model_t* m, const float* class_emb, const float* pos_start, const float* pos_1LC, const float* lvl_embed,
                           const float* word_w, const float* word_b, void* stream) {
    SDVAR_CHECK_ARG(m && class_emb && pos_start && pos_1LC && lvl_embed && word_w && word_b, "bind_embed: null argument");
    m->class_emb = class_emb; m->pos_start = pos_start; m->word_w = word_w; m->word_b = word_b;
    SDVAR_TRY(build_lvl_pos(lvl_embed, pos_1LC, m->stage_of_tok, m->lvl_pos, m->L, m->C, (hipStream_t)stream));
    m->embed_bound = true; m->ada_tab_ready = false;
    return SDVAR_OK;
}

int sdvar_model_bind_block(sdvar_model_t* m, int32_t i, const float* ada_w, const float* ada_b, const float* qkv_w, const float* q_bias,
                           const float* v_bias, const float* scale_mul, const float* proj_w, const float* proj_b, const float* fc1_w,
                           const float* fc1_b, const float* fc2_w, const float* fc2_b, void* stream) {
    SDVAR_CHECK_ARG(m && i >= 0 && i < m->d.depth, "bind_block: block index %d", i);
    // ada_w == NULL: a shared_aln block, ada_b is its ada_gss (6C) and sdvar_model_bind_shared_aln supplies the Linear
    // scale_mul == NULL: an attn_l2_norm=False model (basic_var.py:66-72): no q/k normalisation, softmax scale 0.25 / sqrt(head_dim)
    SDVAR_CHECK_ARG(ada_b && qkv_w && q_bias && v_bias && proj_w && proj_b && fc1_w && fc1_b && fc2_w && fc2_b, "bind_block: null tensor");
    BlockW& b = m->blk[i];
    b.ada_w = ada_w; b.ada_b = ada_b; b.qkv_w = qkv_w; b.scale_mul = scale_mul; b.proj_w = proj_w; b.proj_b = proj_b;
    b.fc1_w = fc1_w; b.fc1_b = fc1_b; b.fc2_w = fc2_w; b.fc2_b = fc2_b;
    const size_t C = m->C;
    hipStream_t s = (hipStream_t)stream;
    SDVAR_HIP(hipMemsetAsync(b.qkv_bias, 0, 3 * C * sizeof(float), s));
    SDVAR_HIP(hipMemcpyAsync(b.qkv_bias, q_bias, C * sizeof(float), hipMemcpyDeviceToDevice, s));
    SDVAR_HIP(hipMemcpyAsync(b.qkv_bias + 2 * C, v_bias, C * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (m->d.gemm_mode == 1) {
        const int Ci = m->C;
        SDVAR_TRY(split_planes(qkv_w, b.qkv_wp, 3 * Ci, Ci, 3 * C * C, s)); SDVAR_TRY(split_planes(proj_w, b.proj_wp, Ci, Ci, C * C, s));
        SDVAR_TRY(split_planes(fc1_w, b.fc1_wp, 4 * Ci, Ci, 4 * C * C, s)); SDVAR_TRY(split_planes(fc2_w, b.fc2_wp, Ci, 4 * Ci, 4 * C * C, s));
    } else if (m->d.gemm_mode == 2) {      // per-tensor power-of-two scale (computed on the device, no sync), then the two fp16 planes of w * 2^S
        const int Ci = m->C;
        const float* ws[4] = {qkv_w, proj_w, fc1_w, fc2_w};
        uint16_t* wp[4] = {b.qkv_wp, b.proj_wp, b.fc1_wp, b.fc2_wp};
        const int rows[4] = {3 * Ci, Ci, 4 * Ci, Ci}, cols[4] = {Ci, Ci, Ci, 4 * Ci};
        for (int k = 0; k < 4; ++k) {
            SDVAR_TRY(weight_scale_f16(ws[k], (size_t)rows[k] * cols[k], b.wsc + 4 * k, s));
            SDVAR_TRY(split_planes_f16(ws[k], wp[k], rows[k], cols[k], (size_t)rows[k] * cols[k], b.wsc + 4 * k, s));
        }
    }
    b.bound = true; m->ada_tab_ready = false;
    return SDVAR_OK;
}

int sdvar_model_bind_shared_aln(sdvar_model_t* m, const float* w, const float* b) {
    SDVAR_CHECK_ARG(m && w && b, "bind_shared_aln: null argument");
    m->shared_w = w; m->shared_b = b;
    if (!m->gss_lin) SDVAR_TRY(dmalloc(&m->gss_lin, (size_t)m->Rmax * 6 * m->C));
    return SDVAR_OK;
}

int sdvar_model_bind_head(sdvar_model_t* m, const float* nm_w, const float* nm_b, const float* head_w, const float* head_b, void* stream) {
    SDVAR_CHECK_ARG(m && nm_w && nm_b && head_w && head_b, "bind_head: null argument");
    m->nm_w = nm_w; m->nm_b = nm_b; m->head_w = head_w; m->head_b = head_b; m->head_bound = true; m->ada_tab_ready = false;
    if (m->d.gemm_mode == 1) SDVAR_TRY(split_planes(head_w, m->head_wp, m->d.vocab, m->C, (size_t)m->d.vocab * m->C, (hipStream_t)stream));
    if (m->d.gemm_mode == 2) {
        SDVAR_TRY(weight_scale_f16(head_w, (size_t)m->d.vocab * m->C, m->head_wsc, (hipStream_t)stream));
        SDVAR_TRY(split_planes_f16(head_w, m->head_wp, m->d.vocab, m->C, (size_t)m->d.vocab * m->C, m->head_wsc, (hipStream_t)stream));
    }
    return SDVAR_OK;
}

static int check_bound(const sdvar_model* m) {
    if (!m) { set_error("null model"); return SDVAR_ERR_ARG; }
    if (!m->embed_bound || !m->head_bound) { set_error("model weights not bound (embed=%d head=%d)", (int)m->embed_bound, (int)m->head_bound); return SDVAR_ERR_STATE; }
    for (size_t i = 0; i < m->blk.size(); ++i) {
        if (!m->blk[i].bound) { set_error("block %zu weights not bound", i); return SDVAR_ERR_STATE; }
        if (!m->blk[i].ada_w && !m->shared_w) { set_error("block %zu is a shared_aln block but shared_ada_lin is not bound", i); return SDVAR_ERR_STATE; }
    }
    return SDVAR_OK;
}

static int model_begin_impl(sdvar_model_t* m, int32_t B, const int64_t* labels, const float* cond_in, void* stream);
int sdvar_model_begin(sdvar_model_t* m, int32_t B, const int64_t* labels, void* stream) {
    SDVAR_CHECK_ARG(labels, "model_begin: null labels");
    return model_begin_impl(m, B, labels, nullptr, stream);
}
int sdvar_model_begin_cond(sdvar_model_t* m, int32_t B, const float* cond, void* stream) {
    SDVAR_CHECK_ARG(cond, "model_begin_cond: null cond");
    return model_begin_impl(m, B, nullptr, cond, stream);
}
static int model_begin_impl(sdvar_model_t* m, int32_t B, const int64_t* labels, const float* cond_in, void* stream) {
    SDVAR_TRY(check_bound(m));
    SDVAR_CHECK_ARG(B >= 1 && B <= m->d.max_batch, "model_begin: B=%d (max %d)", B, m->d.max_batch);
    hipStream_t s = (hipStream_t)stream;
    WsScope wsg(m->ws_own);
    const int C = m->C, R = 2 * B;
    m->B = B; m->kv_len = 0; m->kv_origin = 0;
    {
        ProfScope ps(7, 0, 0, s);
        SDVAR_TRY(prologue((const long long*)labels, cond_in, m->class_emb, m->pos_start, m->lvl_pos, m->cond, m->x0, B, C, m->d.num_classes, s));
        SDVAR_TRY(silu_rows(m->cond, m->cond_silu, R * C, s));
    }
    // adaLN parameters of every block: stage-invariant, computed once per call instead of once per stage - or, for label-conditioned calls, once per BIND:
    static const bool tab_off = getenv("SDVAR_ADALN_TABLE") && atoi(getenv("SDVAR_ADALN_TABLE")) == 0;          // A/B runs
    const size_t NC = (size_t)m->d.num_classes + 1, row = (size_t)m->d.depth * 6 * C + 2 * C;
    if (labels && !m->shared_w && !tab_off && NC * (row + C) * sizeof(float) <= ((size_t)8 << 30)) {
        if (!m->ada_tab_ready) {
            if (!m->ada_tab) SDVAR_TRY(dmalloc(&m->ada_tab, NC * (row + C)));
            float* sil = m->ada_tab + NC * row;                  // silu(class_emb), scratch behind the table
            ProfScope ps(9, 2.0 * NC * (double)row * C, 4.0 * (double)row * (C + NC), s);
            SDVAR_TRY(silu_rows(m->class_emb, sil, (int)(NC * C), s));
            for (int i = 0; i < m->d.depth; ++i)
                SDVAR_TRY(gemm_f32_nt(sil, C, m->blk[i].ada_w, m->blk[i].ada_b, m->ada_tab + (size_t)i * 6 * C, (int)row, (int)NC, 6 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
            SDVAR_TRY(gemm_f32_nt(sil, C, m->nm_w, m->nm_b, m->ada_tab + (size_t)m->d.depth * 6 * C, (int)row, (int)NC, 2 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
            SDVAR_HIP(hipStreamSynchronize(s));          // once per bind: later calls may come on other streams and only READ the table
            m->ada_tab_ready = true;
        }
        ProfScope ps(7, 0, 8.0 * R * (double)row, s);
        SDVAR_TRY(ada_gather((const long long*)labels, m->ada_tab, row, m->d.depth, C, m->ada, (size_t)m->Rmax * 6 * C, m->ada_head, B, m->d.num_classes, s));
        m->begun = true;
        return SDVAR_OK;
    }
    if (m->shared_w) {          // shared_aln: one Linear for all blocks (var.py:192), each block adds its ada_gss (basic_var.py:153-154)
        ProfScope ps(9, 2.0 * R * 6.0 * C * C, 4.0 * (6.0 * C * C + R * 7.0 * C), s);
        SDVAR_TRY(gemm_f32_nt(m->cond_silu, C, m->shared_w, m->shared_b, m->gss_lin, 6 * C, R, 6 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
    }
    for (int i = 0; i < m->d.depth; ++i) {
        ProfScope ps(9, 2.0 * R * 6.0 * C * C, 4.0 * (6.0 * C * C + R * 7.0 * C), s);
        float* dst = m->ada + (size_t)i * m->Rmax * 6 * C;
        if (!m->blk[i].ada_w) SDVAR_TRY(add_row_vector(m->gss_lin, m->blk[i].ada_b, dst, R, 6 * C, s));
        else SDVAR_TRY(gemm_f32_nt(m->cond_silu, C, m->blk[i].ada_w, m->blk[i].ada_b, dst, 6 * C, R, 6 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
    }
    {
        ProfScope ps(9, 2.0 * R * 2.0 * C * C, 4.0 * (2.0 * C * C + R * 3.0 * C), s);
        SDVAR_TRY(gemm_f32_nt(m->cond_silu, C, m->nm_w, m->nm_b, m->ada_head, 2 * C, R, 2 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s));
    }
    m->begun = true;
    return SDVAR_OK;
}

int sdvar_model_place_first(sdvar_model_t* m, float* x, int32_t ltot, void* stream) {
    SDVAR_CHECK_ARG(m && m->begun && x && ltot >= 1, "place_first: model not begun or bad args");
    SDVAR_HIP(hipMemcpy2DAsync(x, (size_t)ltot * m->C * sizeof(float), m->x0, (size_t)m->C * sizeof(float), (size_t)m->C * sizeof(float), 2 * m->B,
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SDVAR_OK;
}

int sdvar_model_export_prologue(sdvar_model_t* m, float* cond, float* lvl_pos, float* first, void* stream) {
    SDVAR_CHECK_ARG(m && m->begun, "export_prologue: model not begun");
    hipStream_t s = (hipStream_t)stream;
    const size_t rc = (size_t)2 * m->B * m->C * sizeof(float);
    if (cond) SDVAR_HIP(hipMemcpyAsync(cond, m->cond, rc, hipMemcpyDeviceToDevice, s));
    if (first) SDVAR_HIP(hipMemcpyAsync(first, m->x0, rc, hipMemcpyDeviceToDevice, s));
    if (lvl_pos) SDVAR_HIP(hipMemcpyAsync(lvl_pos, m->lvl_pos, (size_t)m->L * m->C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SDVAR_OK;
}

int sdvar_kv_len(const sdvar_model_t* m) { return m ? m->kv_len : -1; }

int sdvar_kv_set_len(sdvar_model_t* m, int32_t len) {
    SDVAR_CHECK_ARG(m && len >= 0 && len <= m->kv_len, "kv_set_len: %d not in [0, %d] (rollback only)", len, m ? m->kv_len : -1);
    m->kv_len = len;
    return SDVAR_OK;
}

int sdvar_kv_set_origin(sdvar_model_t* m, int32_t stage) {
    SDVAR_CHECK_ARG(m && m->begun && stage >= 0 && stage < m->S, "kv_set_origin: stage %d", stage);
    SDVAR_CHECK_ARG(m->kv_len == 0, "kv_set_origin: the cache holds %d keys (only an empty cache can be re-based)", m->kv_len);
    m->kv_origin = begin_of(m, stage);
    return SDVAR_OK;
}

int sdvar_embed_next_at(sdvar_model_t* m, const float* nxt, int32_t s_next, int32_t pos_begin, float* x, int32_t ltot, int32_t tok_off, void* stream) {
    SDVAR_CHECK_ARG(m && m->begun && nxt && x, "embed_next: model not begun or null");
    SDVAR_CHECK_ARG(s_next >= 1 && s_next < m->S && tok_off >= 0 && tok_off + m->lens[s_next] <= ltot, "embed_next: stage %d off %d ltot %d", s_next, tok_off, ltot);
    SDVAR_CHECK_ARG(pos_begin >= 0 && pos_begin + m->lens[s_next] <= m->L, "embed_next: position rows %d .. %d outside the table (%d)", pos_begin, pos_begin + m->lens[s_next], m->L);
    ProfScope ps(7, 2.0 * m->B * m->lens[s_next] * 32.0 * m->C, 4.0 * m->lens[s_next] * m->C * (2.0 * m->B + 1.0), (hipStream_t)stream);
    return embed_next(nxt, m->word_w, m->word_b, m->lvl_pos, x, m->B, m->lens[s_next], m->C, pos_begin, ltot, tok_off, (hipStream_t)stream);
}

int sdvar_embed_next(sdvar_model_t* m, const float* nxt, int32_t s_next, float* x, int32_t ltot, int32_t tok_off, void* stream) {
    SDVAR_CHECK_ARG(m && s_next >= 1 && s_next < m->S, "embed_next: stage %d", s_next);
    return sdvar_embed_next_at(m, nxt, s_next, begin_of(m, s_next), x, ltot, tok_off, stream);
}

static int stage_forward_impl(sdvar_model_t* m, float* x, int32_t s0, int32_t n, const float* bias, float* logits, void* stream);

int sdvar_stage_forward(sdvar_model_t* m, float* x, int32_t s0, int32_t n, float* logits, void* stream) {
    return stage_forward_impl(m, x, s0, n, nullptr, logits, stream);
}

int sdvar_stage_forward_masked(sdvar_model_t* m, float* x, int32_t s0, int32_t n, const float* bias, float* logits, void* stream) {
    SDVAR_CHECK_ARG(bias, "stage_forward_masked: null mask");
    return stage_forward_impl(m, x, s0, n, bias, logits, stream);
}

static int stage_forward_impl(sdvar_model_t* m, float* x, int32_t s0, int32_t n, const float* bias, float* logits, void* stream) {
    SDVAR_TRY(check_bound(m));
    SDVAR_CHECK_ARG(m->begun && x && logits, "stage_forward: model not begun or null buffers");
    SDVAR_CHECK_ARG(s0 >= 0 && n >= 1 && s0 + n <= m->S && n <= m->d.max_chunk_stages, "stage_forward: stages [%d,%d) invalid (S=%d, max chunk %d)", s0, s0 + n, m->S, m->d.max_chunk_stages);
    if (m->kv_len != begin_of(m, s0) - m->kv_origin) { set_error("stage_forward: KV cache holds %d keys, stage %d needs %d", m->kv_len, s0, begin_of(m, s0) - m->kv_origin); return SDVAR_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    WsScope wsg(m->ws_own);
    const int C = m->C, H = m->H, R = 2 * m->B, V = m->d.vocab;
    int qbeg[SDVAR_MAX_STAGES], vis[SDVAR_MAX_STAGES], lsum = 0;
    double lk = 0;
    for (int j = 0; j < n; ++j) { qbeg[j] = lsum; lsum += m->lens[s0 + j]; vis[j] = m->cum[s0 + j] - m->kv_origin; lk += (double)m->lens[s0 + j] * vis[j]; }
    const int M = R * lsum, Ktot = m->kv_len + lsum;
    const double dM = M, dC = C;
    const int GC = M >= 1024 ? 0 : 9;                                 // profiling class of this call's GEMMs: matrix-pipe regime / weight-streaming + latency regime
    const bool P = m->d.gemm_mode >= 1;                               // split-operand GEMMs: inputs travel as planes
    const int PF = m->d.gemm_mode == 2 ? PLANES_F16X2 : PLANES_BF16X3;
    const size_t ps = (size_t)M * C;                                  // plane stride of this call's (M, C) activations
    const bool G2 = g_guard_on && m->d.gemm_mode == 2;                // f16x2 guard: count saturated / NaN / tiny operand elements behind every producer
    // In bf16x3 mode a split-K GEMM feeding a row kernel leaves its K-slice slabs in the shared workspace and the consumer
    // (qk_norm_append for QKV; the next ln_modulate for the two gated-residual GEMMs) sums them: no reduce launches.
    size_t wsf = 0;
    const float* ws = P ? splitk_workspace(&wsf) : nullptr;
    PendingSplitK pend{nullptr, nullptr, nullptr, 0, 1, 0};           // unreduced gated residual waiting for the next ln_modulate
    int defer = 0;
    int* const dp = &defer;      // every split-K GEMM below leaves its K-slice sum to the row kernel that reads the result next
    // timing experiments only (results wrong): SDVAR_SKIP_CLASS bit 0 ln_modulate, 1 qk_norm_append, 2 attention, 3 fc1, 4 QKV GEMM, 5 proj, 6 fc2 not launched
    // (tools/micro/skip_class_exp.sh: the marginal cost of a kernel class inside the real launch sequence, without a profiler's per-kernel overhead)
#ifdef SDVAR_TIMING_EXPERIMENTS       // make EXTRA=-DSDVAR_TIMING_EXPERIMENTS: never in the product library
    static const int skip = getenv("SDVAR_SKIP_CLASS") ? atoi(getenv("SDVAR_SKIP_CLASS")) : 0;
    static const bool warned = (skip && fprintf(stderr, "[sdvar] SDVAR_SKIP_CLASS=%d: timing experiment build, results are WRONG\n", skip), true);
    (void)warned;
#else
    constexpr int skip = 0;
#endif
    // 32 .. 80 rows in GEMM mode f16x2 (round 4; stage 1 and the first verify chunk at B = 8): FIVE launches per block instead of eight - LayerNorm + modulation run in the operand prologue of the QKV / fc1 / head
    // launch (gemm_f16x2_rowblk_kernel), the QKV launch finishes q, k and v (no qk_norm_append), fc2 streams K = 4C unsplit (no slabs for the next LayerNorm to sum).
    // The f16x2 guard counts the operand planes ln_modulate writes, so a guarded call keeps the old sequence.
    const bool RB = m->d.gemm_mode == 2 && (m->kv_fmt == 3 || m->kv_fmt == 4) && !g_guard_on && skip == 0 && gemm_f16x2_rowblk_want(M, C, V, lsum);
    for (int i = 0; i < m->d.depth && RB; ++i) {
        const BlockW& b = m->blk[i];
        const float* ada = m->ada + (size_t)i * m->Rmax * 6 * C;
        { ProfScope pp(GC, 2 * dM * 3 * dC * dC, 4 * (dM * dC + 3 * dC * dC + 3 * dM * dC), s);
          SDVAR_TRY(gemm_f16x2_rowblk(x, C, ada + 2 * C, ada + 4 * C, lsum, 6 * C, nullptr, 0, b.qkv_wp, (size_t)3 * C * C, b.wsc + 1, b.qkv_bias, nullptr, 0, nullptr, 0, M, 3 * C, C, 0,
                                      nullptr, 0, nullptr, 1, 0, b.scale_mul, m->qbuf, b.kc, b.vc, lsum, H, m->Lkv, m->kv_len, m->kv_fmt, s)); }
        { ProfScope pp(lsum <= 36 ? 8 : 1, 4.0 * R * H * 64.0 * lk, R * H * 64.0 * ((m->d.kv_dtype ? 4.0 : 8.0) * Ktot + 8.0 * lsum), s);
          if (bias) SDVAR_TRY(attention_masked(m->qbuf, b.kc, b.vc, m->kv_fmt, bias, m->att, m->att_p, ps, PF, R, H, lsum, m->Lkv, Ktot, s));
          else SDVAR_TRY(attention_f32(m->qbuf, b.kc, b.vc, m->kv_fmt, m->att, m->att_p, ps, PF, R, H, lsum, m->Lkv, Ktot, n, qbeg, vis, s)); }
        { ProfScope pp(GC, 2 * dM * dC * dC, 4 * (3 * dM * dC + dC * dC), s);          // proj: K = C, the skinny kernel streams it unsplit
          SDVAR_TRY(plane_gemm(m, m->att_p, ps, b.proj_wp, (size_t)C * C, b.wsc + 4, b.proj_b, x, C, nullptr, 0, M, C, C, EPI_GATED_RES, x, C, ada, lsum, 6 * C, nullptr, s)); }
        { ProfScope pp(GC, 2 * dM * 4 * dC * dC, 4 * (dM * dC + 4 * dC * dC + 4 * dM * dC), s);
          SDVAR_TRY(gemm_f16x2_rowblk(x, C, ada + 3 * C, ada + 5 * C, lsum, 6 * C, nullptr, 0, b.fc1_wp, (size_t)4 * C * C, b.wsc + 9, b.fc1_b, nullptr, 0, m->hid_p, 4 * ps, M, 4 * C, C, EPI_BIAS_GELU,
                                      nullptr, 0, nullptr, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, s)); }
        { ProfScope pp(GC, 2 * dM * 4 * dC * dC, 4 * (4 * dM * dC + 4 * dC * dC + 2 * dM * dC), s);
          SDVAR_TRY(gemm_f16x2_rowblk(nullptr, 0, nullptr, nullptr, 1, 0, m->hid_p, 4 * ps, b.fc2_wp, (size_t)4 * C * C, b.wsc + 13, b.fc2_b, x, C, nullptr, 0, M, C, 4 * C, EPI_GATED_RES,
                                      x, C, ada + C, lsum, 6 * C, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, s)); }
    }
    if (RB) {
        ProfScope pp(GC, 2 * dM * dC * V, 4 * (dM * dC + dC * V + dM * V), s);
        SDVAR_TRY(gemm_f16x2_rowblk(x, C, m->ada_head, m->ada_head + C, lsum, 2 * C, nullptr, 0, m->head_wp, (size_t)V * C, m->head_wsc + 1, m->head_b, logits, V, nullptr, 0, M, V, C, EPI_BIAS,
                                    nullptr, 0, nullptr, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, s));
        m->kv_len = Ktot;
        return SDVAR_OK;
    }
    for (int i = 0; i < m->d.depth; ++i) {
        const BlockW& b = m->blk[i];
        const float* ada = m->ada + (size_t)i * m->Rmax * 6 * C;      // (R, 6C): gamma1 gamma2 scale1 scale2 shift1 shift2
        if (!(skip & 1)) { ProfScope pp(2, 8 * dM * dC, (P ? 10 : 8) * dM * dC, s);
          SDVAR_TRY(ln_modulate(x, ada + 2 * C, ada + 4 * C, m->xn, P ? m->xn_p : nullptr, ps, M, C, lsum, 6 * C, &pend, PF, s)); pend.ws = nullptr; }
        if (G2) SDVAR_TRY(guard_planes(m->xn_p, ps, s));
        PendingSplitK pq{nullptr, nullptr, nullptr, 0, 1, 0};
        int qk_fused = 0;        // the QKV launch came out unsplit and finished q, k and v in its epilogue (f16x2 planes cache): no qk_norm_append
        if (!(skip & 16)) { ProfScope pp(GC, 2 * dM * 3 * dC * dC, 4 * (dM * dC + 3 * dC * dC + 3 * dM * dC), s);
          if (P && m->d.gemm_mode == 2 && (m->kv_fmt == 3 || m->kv_fmt == 4)) {
                   SDVAR_TRY(gemm_f16x2_qkv(m->xn_p, ps, b.qkv_wp, (size_t)3 * C * C, b.wsc + 1, b.qkv_bias, m->qkv, 3 * C, M, 3 * C, C, b.scale_mul, m->qbuf, b.kc, b.vc, lsum, H, m->Lkv,
                                            m->kv_len, m->kv_fmt, dp, &qk_fused, s));
                   if (defer) pq = PendingSplitK{ws, b.qkv_bias, nullptr, defer, 1, 0}; }
          else if (P) { SDVAR_TRY(plane_gemm(m, m->xn_p, ps, b.qkv_wp, (size_t)3 * C * C, b.wsc, b.qkv_bias, m->qkv, 3 * C, nullptr, 0, M, 3 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, dp, s));
                   if (defer) pq = PendingSplitK{ws, b.qkv_bias, nullptr, defer, 1, 0}; }
          else SDVAR_TRY(gemm_f32_nt(m->xn, C, b.qkv_w, b.qkv_bias, m->qkv, 3 * C, M, 3 * C, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s)); }
        if (!(skip & 2) && !qk_fused) { ProfScope pp(3, 6 * dM * dC, 4 * 6 * dM * dC, s);      // an unsplit QKV launch has finished q, k and v in its epilogue
          SDVAR_TRY(qk_norm_append(m->qkv, b.scale_mul, m->qbuf, b.kc, b.vc, m->kv_fmt, R, lsum, H, m->Lkv, m->kv_len, &pq, 0, s)); }
        if (!(skip & 4)) { ProfScope pp(lsum <= 36 ? 8 : 1, 4.0 * R * H * 64.0 * lk, R * H * 64.0 * ((m->d.kv_dtype ? 4.0 : 8.0) * Ktot + 8.0 * lsum), s);
          if (bias) SDVAR_TRY(attention_masked(m->qbuf, b.kc, b.vc, m->kv_fmt, bias, m->att, P ? m->att_p : nullptr, ps, PF, R, H, lsum, m->Lkv, Ktot, s));
          else SDVAR_TRY(attention_f32(m->qbuf, b.kc, b.vc, m->kv_fmt, m->att, P ? m->att_p : nullptr, ps, PF, R, H, lsum, m->Lkv, Ktot, n, qbeg, vis, s)); }
        if (G2) SDVAR_TRY(guard_planes(m->att_p, ps, s));
        if (!(skip & 32)) { ProfScope pp(GC, 2 * dM * dC * dC, 4 * (3 * dM * dC + dC * dC), s);
          if (P) { SDVAR_TRY(plane_gemm(m, m->att_p, ps, b.proj_wp, (size_t)C * C, b.wsc + 4, b.proj_b, x, C, nullptr, 0, M, C, C, EPI_GATED_RES, x, C, ada, lsum, 6 * C, dp, s));
                   if (defer) pend = PendingSplitK{ws, b.proj_b, ada, defer, lsum, 6 * C}; }
          else SDVAR_TRY(gemm_f32_nt(m->att, C, b.proj_w, b.proj_b, x, C, M, C, C, EPI_GATED_RES, x, C, ada, lsum, 6 * C, s)); }
        if (!(skip & 1)) { ProfScope pp(2, 8 * dM * dC, (P ? 10 : 8) * dM * dC, s);
          SDVAR_TRY(ln_modulate(x, ada + 3 * C, ada + 5 * C, m->xn, P ? m->xn_p : nullptr, ps, M, C, lsum, 6 * C, &pend, PF, s)); pend.ws = nullptr; }
        if (G2) SDVAR_TRY(guard_planes(m->xn_p, ps, s));
        if (!(skip & 8)) { ProfScope pp(GC, 2 * dM * 4 * dC * dC, 4 * (dM * dC + 4 * dC * dC + 4 * dM * dC), s);
          if (P) SDVAR_TRY(plane_gemm(m, m->xn_p, ps, b.fc1_wp, (size_t)4 * C * C, b.wsc + 8, b.fc1_b, nullptr, 0, m->hid_p, 4 * ps, M, 4 * C, C, EPI_BIAS_GELU, nullptr, 0, nullptr, 0, 0, nullptr, s));
          else SDVAR_TRY(gemm_f32_nt(m->xn, C, b.fc1_w, b.fc1_b, m->hid, 4 * C, M, 4 * C, C, EPI_BIAS_GELU, nullptr, 0, nullptr, 0, 0, s)); }
        if (G2) SDVAR_TRY(guard_planes(m->hid_p, 4 * ps, s));
        if (!(skip & 64)) { ProfScope pp(GC, 2 * dM * 4 * dC * dC, 4 * (4 * dM * dC + 4 * dC * dC + 2 * dM * dC), s);
          if (P) { SDVAR_TRY(plane_gemm(m, m->hid_p, 4 * ps, b.fc2_wp, (size_t)4 * C * C, b.wsc + 12, b.fc2_b, x, C, nullptr, 0, M, C, 4 * C, EPI_GATED_RES, x, C, ada + C, lsum, 6 * C, dp, s));
                   if (defer) pend = PendingSplitK{ws, b.fc2_b, ada + C, defer, lsum, 6 * C}; }
          else SDVAR_TRY(gemm_f32_nt(m->hid, 4 * C, b.fc2_w, b.fc2_b, x, C, M, C, 4 * C, EPI_GATED_RES, x, C, ada + C, lsum, 6 * C, s)); }
    }
    { ProfScope pp(2, 8 * dM * dC, (P ? 10 : 8) * dM * dC, s);      // also finishes the last block's fc2 residual when it was left split
      SDVAR_TRY(ln_modulate(x, m->ada_head, m->ada_head + C, m->xn, P ? m->xn_p : nullptr, ps, M, C, lsum, 2 * C, &pend, PF, s)); pend.ws = nullptr; }
    if (G2) SDVAR_TRY(guard_planes(m->xn_p, ps, s));
    { ProfScope pp(GC, 2 * dM * dC * V, 4 * (dM * dC + dC * V + dM * V), s);
      if (P) SDVAR_TRY(plane_gemm(m, m->xn_p, ps, m->head_wp, (size_t)V * C, m->head_wsc, m->head_b, logits, V, nullptr, 0, M, V, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, nullptr, s));
      else SDVAR_TRY(gemm_f32_nt(m->xn, C, m->head_w, m->head_b, logits, V, M, V, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s)); }
    m->kv_len = Ktot;
    return SDVAR_OK;
}

int sdvar_head_forward(sdvar_model_t* m, const float* x, int32_t l, float* logits, void* stream) {
    SDVAR_TRY(check_bound(m));
    SDVAR_CHECK_ARG(m->begun && x && logits && l >= 1 && l <= m->lmax, "head_forward: model not begun, null buffers or l=%d > %d", l, m->lmax);
    hipStream_t s = (hipStream_t)stream;
    WsScope wsg(m->ws_own);
    const int C = m->C, R = 2 * m->B, V = m->d.vocab, M = R * l;
    const double dM = M, dC = C;
    const int GC = M >= 1024 ? 0 : 9;
    const bool P = m->d.gemm_mode >= 1;
    const int PF = m->d.gemm_mode == 2 ? PLANES_F16X2 : PLANES_BF16X3;
    const size_t ps = (size_t)M * C;
    { ProfScope pp(2, 8 * dM * dC, (P ? 10 : 8) * dM * dC, s);
      SDVAR_TRY(ln_modulate(const_cast<float*>(x), m->ada_head, m->ada_head + C, m->xn, P ? m->xn_p : nullptr, ps, M, C, l, 2 * C, nullptr, PF, s)); }
    { ProfScope pp(GC, 2 * dM * dC * V, 4 * (dM * dC + dC * V + dM * V), s);
      if (P) SDVAR_TRY(plane_gemm(m, m->xn_p, ps, m->head_wp, (size_t)V * C, m->head_wsc, m->head_b, logits, V, nullptr, 0, M, V, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, nullptr, s));
      else SDVAR_TRY(gemm_f32_nt(m->xn, C, m->head_w, m->head_b, logits, V, M, V, C, EPI_BIAS, nullptr, 0, nullptr, 0, 0, s)); }
    return SDVAR_OK;
}

// ---------------------------------------------------------------------------------------------------- quantizer
static void bicubic_row(int n_in, int n_out, int o, float* row) {   // one row of the (n_out x n_in) up-sampling matrix
    const float A = -0.75f;
    const float scale = (float)n_in / (float)n_out;
    const float src = scale * ((float)o + 0.5f) - 0.5f;
    const int i0 = (int)floorf(src);
    const float t = src - (float)i0;
    const float x2 = 1.0f - t;
    const float xs[4] = {t + 1.0f, t, x2, x2 + 1.0f};
    float w[4];
    w[0] = ((A * xs[0] - 5.0f * A) * xs[0] + 8.0f * A) * xs[0] - 4.0f * A;
    w[1] = ((A + 2.0f) * xs[1] - (A + 3.0f)) * xs[1] * xs[1] + 1.0f;
    w[2] = ((A + 2.0f) * xs[2] - (A + 3.0f)) * xs[2] * xs[2] + 1.0f;
    w[3] = ((A * xs[3] - 5.0f * A) * xs[3] + 8.0f * A) * xs[3] - 4.0f * A;
    std::vector<double> acc(n_in, 0.0);
    for (int k = 0; k < 4; ++k) {
        int idx = i0 - 1 + k;
        idx = idx < 0 ? 0 : (idx > n_in - 1 ? n_in - 1 : idx);
        acc[idx] += (double)w[k];
    }
    for (int i = 0; i < n_in; ++i) row[i] = (float)acc[i];
}

int sdvar_quant_create(int32_t S, const int32_t* patch_nums, int32_t cvae, int32_t vocab, int32_t max_batch, int32_t n_phi, sdvar_quant_t** out) {
    SDVAR_CHECK_ARG(out && patch_nums && S >= 1 && S <= SDVAR_MAX_STAGES && cvae >= 1 && vocab >= 1 && max_batch >= 1, "quant_create: bad argument");
    SDVAR_CHECK_ARG(n_phi >= 1 && n_phi <= SDVAR_MAX_STAGES, "quant_create: n_phi %d", n_phi);          // 1 = PhiShared, S = PhiNonShared (quant.py:27-32)
    sdvar_quant* q = new sdvar_quant();
    q->S = S; q->Cv = cvae; q->V = vocab; q->maxB = max_batch; q->n_phi = n_phi; q->bound = false; q->codebook = nullptr;
    for (int s = 0; s < S; ++s) q->pn[s] = patch_nums[s];
    q->HW = patch_nums[S - 1];
    SDVAR_CHECK_ARG(q->HW <= 64, "quant_create: HW %d > 64", q->HW);
    q->hWup.resize(S); q->hWdn.resize(S);
    for (int s = 0; s < S; ++s) {
        // Phi choice: argmin |ticks - s/(S-1)| (quant.py:223-226); ticks = linspace(1/3/K, 1-1/3/K, K) for K == 4 else the 1/2/K form
        // evaluated exactly like numpy.linspace (k*step + start, last tick = stop) so that the tie at s/(S-1) = 7/9
        // between ticks 2 and 3 resolves as in the reference
        int best = 0; double bd = 1e30;
        const double lo = (n_phi == 4) ? 1.0 / 3 / n_phi : 1.0 / 2 / n_phi, hi = 1.0 - lo;
        const double step = (n_phi > 1) ? (hi - lo) / (n_phi - 1) : 0.0;
        const double at = (S > 1) ? (double)s / (S - 1) : 0.0;
        for (int k = 0; k < n_phi; ++k) {
            const double tick = (k == n_phi - 1 && n_phi > 1) ? hi : (double)k * step + lo;
            const double dd = fabs(tick - at);
            if (dd < bd) { bd = dd; best = k; }
        }
        q->phi_of[s] = best;
        q->Wup[s] = q->Wdn[s] = nullptr;
        if (s < S - 1) {
            const int pn = q->pn[s], HW = q->HW, p2 = q->pn[s + 1];
            q->hWup[s].assign((size_t)HW * pn, 0.f);
            for (int o = 0; o < HW; ++o) bicubic_row(pn, HW, o, &q->hWup[s][(size_t)o * pn]);
            q->hWdn[s].assign((size_t)p2 * HW, 0.f);
            for (int o = 0; o < p2; ++o) {
                const int st = (o * HW) / p2, en = ((o + 1) * HW + p2 - 1) / p2;
                for (int i = st; i < en; ++i) q->hWdn[s][(size_t)o * HW + i] = 1.0f / (float)(en - st);
            }
            SDVAR_TRY(dmalloc(&q->Wup[s], q->hWup[s].size()));
            SDVAR_TRY(dmalloc(&q->Wdn[s], q->hWdn[s].size()));
            SDVAR_HIP(hipMemcpy(q->Wup[s], q->hWup[s].data(), q->hWup[s].size() * sizeof(float), hipMemcpyHostToDevice));
            SDVAR_HIP(hipMemcpy(q->Wdn[s], q->hWdn[s].data(), q->hWdn[s].size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    SDVAR_TRY(dmalloc(&q->up_scratch, (size_t)max_batch * cvae * q->HW * q->HW));
    *out = q;
    return SDVAR_OK;
}

int sdvar_quant_destroy(sdvar_quant_t* q) {
    if (!q) return SDVAR_OK;
    for (int s = 0; s < q->S; ++s) { if (q->Wup[s]) (void)hipFree(q->Wup[s]); if (q->Wdn[s]) (void)hipFree(q->Wdn[s]); }
    if (q->up_scratch) (void)hipFree(q->up_scratch);
    delete q;
    return SDVAR_OK;
}

int sdvar_quant_bind(sdvar_quant_t* q, const float* codebook, const float* const* phi_w, const float* const* phi_b) {
    SDVAR_CHECK_ARG(q && codebook && phi_w && phi_b, "quant_bind: null argument");
    for (int k = 0; k < q->n_phi; ++k) { SDVAR_CHECK_ARG(phi_w[k] && phi_b[k], "quant_bind: null phi %d", k); q->phi_w[k] = phi_w[k]; q->phi_b[k] = phi_b[k]; }
    q->codebook = codebook; q->bound = true;
    return SDVAR_OK;
}

static int quant_next_impl(sdvar_quant_t* q, int32_t si, const int64_t* ids, int32_t ids_stride, const float* hvec, const float* f_in, float* f_hat, float* nxt,
                           int32_t B, void* stream) {
    SDVAR_CHECK_ARG(q && q->bound, "quant_next: quantizer not bound");
    SDVAR_CHECK_ARG(si >= 0 && si < q->S && B >= 1 && B <= q->maxB && (ids || hvec) && f_hat, "quant_next: stage %d B %d (max %d)", si, B, q ? q->maxB : 0);
    const int last = (si == q->S - 1), k = q->phi_of[si];
    ProfScope ps(6, 2.0 * B * q->Cv * q->Cv * 9.0 * q->HW * q->HW, 4.0 * B * q->Cv * q->HW * q->HW * 4.0, (hipStream_t)stream);
    return quant_next((const long long*)ids, ids_stride, hvec, q->codebook, q->Wup[si], q->phi_w[k], q->phi_b[k], last ? nullptr : q->Wdn[si], q->up_scratch,
                      f_in, f_hat, nxt, B, q->pn[si], last ? 0 : q->pn[si + 1], q->HW, q->Cv, last, (hipStream_t)stream);
}

int sdvar_quant_next(sdvar_quant_t* q, int32_t si, const int64_t* ids, int32_t ids_stride, float* f_hat, float* nxt, int32_t B, void* stream) {
    return quant_next_impl(q, si, ids, ids_stride, nullptr, nullptr, f_hat, nxt, B, stream);
}

int sdvar_quant_next_from(sdvar_quant_t* q, int32_t si, const int64_t* ids, int32_t ids_stride, const float* f_in, float* f_out, float* nxt, int32_t B, void* stream) {
    SDVAR_CHECK_ARG(f_in && ids, "quant_next_from: null operand");
    return quant_next_impl(q, si, ids, ids_stride, nullptr, f_in, f_out, nxt, B, stream);
}

int sdvar_quant_next_h(sdvar_quant_t* q, int32_t si, const float* h, float* f_hat, float* nxt, int32_t B, void* stream) {
    SDVAR_CHECK_ARG(h, "quant_next_h: null feature vectors");
    return quant_next_impl(q, si, nullptr, 0, h, nullptr, f_hat, nxt, B, stream);
}

int sdvar_gumbel_mix(sdvar_quant_t* q, const float* masked_logits, int32_t B, int32_t l, double ratio, double tau, const float* e_noise, uint64_t seed,
                     uint32_t draw, uint32_t image_offset, float* h_out, void* stream) {
    SDVAR_CHECK_ARG(q && q->bound && masked_logits && h_out, "gumbel_mix: quantizer not bound or null operand");
    SDVAR_CHECK_ARG(tau > 0.0, "gumbel_mix: tau %g", tau);
    // torch evaluates logits.mul(1 + ratio) and the division by tau with the python scalars rounded to float32 (var.py:206-208, helpers.py:27)
    ProfScope ps(4, 2.0 * B * l * q->V * q->Cv, 4.0 * B * l * q->V * (e_noise ? 2.0 : 1.0), (hipStream_t)stream);
    return gumbel_mix(masked_logits, B, l, q->V, (float)(1.0 + ratio), (float)tau, e_noise, seed, draw, image_offset, q->codebook, q->Cv, h_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------- sampling
int sdvar_cfg_sample(const float* logits, int32_t B, int32_t l, int32_t V, double t, int32_t top_k, double top_p, const float* q, uint64_t seed,
                     uint32_t draw, uint32_t image_offset, int64_t* ids_out, int32_t ids_stride, float* dbg_masked, void* stream) {
    // torch evaluates (1+t)*a - t*b with the python scalars rounded to float32 (var.py:199-200); `<= (1 - top_p)` likewise
    ProfScope ps(4, 0, 4.0 * B * l * V * (q ? 3.0 : 2.0), (hipStream_t)stream);
    return cfg_sample(logits, B, l, V, (float)(1.0 + t), (float)t, top_k, top_p > 0.0 ? 1 : 0, (float)(1.0 - top_p), q, seed, draw, image_offset,
                      (long long*)ids_out, ids_stride, dbg_masked, (hipStream_t)stream);
}

int sdvar_verify_accept_ex(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n, const int32_t* stage_lens, const double* t,
                           const int64_t* draft_ids, int32_t ids_stride, double thr, int32_t rule, int32_t match_top_k, double kl_thr,
                           const float* draft_logits, int32_t* counts, int64_t* argmax_out, uint8_t* match_out, int64_t* corrected_out, void* stream) {
    SDVAR_CHECK_ARG(stage_lens && t && n >= 1 && n <= SDVAR_MAX_STAGES, "verify_accept: bad stage table");
    int qbeg[SDVAR_MAX_STAGES]; float opt[SDVAR_MAX_STAGES], tf[SDVAR_MAX_STAGES]; long long dl[SDVAR_MAX_STAGES]; int acc = 0;
    for (int j = 0; j < n; ++j) { qbeg[j] = acc; dl[j] = (long long)2 * B * V * acc; acc += stage_lens[j]; opt[j] = (float)(1.0 + t[j]); tf[j] = (float)t[j]; }
    SDVAR_CHECK_ARG(acc == lsum, "verify_accept: stage lens sum %d != lsum %d", acc, lsum);
    ProfScope ps(5, 0, 4.0 * 2.0 * B * lsum * V * (rule == 2 ? 2.0 : 1.0), (hipStream_t)stream);
    return verify_accept(logits, B, lsum, V, n, qbeg, opt, tf, (const long long*)draft_ids, ids_stride, thr, rule, match_top_k, (float)kl_thr, draft_logits, dl, counts,
                         (long long*)argmax_out, match_out, (long long*)corrected_out, (hipStream_t)stream);
}

int sdvar_verify_accept(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n, const int32_t* stage_lens, const double* t,
                        const int64_t* draft_ids, int32_t ids_stride, double thr, int32_t* counts, int64_t* argmax_out, void* stream) {
    return sdvar_verify_accept_ex(logits, B, lsum, V, n, stage_lens, t, draft_ids, ids_stride, thr, 0, 0, 0.0, nullptr, counts, argmax_out, nullptr, nullptr, stream);
}

int sdvar_cfg_combine(const float* logits, int32_t B, int32_t lsum, int32_t V, int32_t n, const int32_t* stage_lens, const double* t, float* out, void* stream) {
    SDVAR_CHECK_ARG(stage_lens && t && n >= 1 && n <= SDVAR_MAX_STAGES, "cfg_combine: bad stage table");
    int qbeg[SDVAR_MAX_STAGES]; float opt[SDVAR_MAX_STAGES], tf[SDVAR_MAX_STAGES]; int acc = 0;
    for (int j = 0; j < n; ++j) { qbeg[j] = acc; acc += stage_lens[j]; opt[j] = (float)(1.0 + t[j]); tf[j] = (float)t[j]; }
    SDVAR_CHECK_ARG(acc == lsum, "cfg_combine: stage lens sum %d != lsum %d", acc, lsum);
    ProfScope ps(5, 0, 4.0 * 3.0 * B * lsum * V, (hipStream_t)stream);
    return cfg_combine(logits, B, lsum, V, n, qbeg, opt, tf, out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------- single ops
int sdvar_op_gemm(const float* X, int32_t ldx, const float* W, const float* bias, float* out, int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t epi,
                  const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream) {
    ProfScope ps(M >= 1024 ? 0 : 9, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), (hipStream_t)stream);
    return gemm_f32_nt(X, ldx, W, bias, out, ldo, M, N, K, epi, res, ldres, gate, rows_per_gate, gate_stride, (hipStream_t)stream);
}
int sdvar_op_ln_modulate(const float* x, const float* scale, const float* shift, float* out, uint16_t* out_planes, uint64_t plane_stride, int32_t plane_format,
                         int32_t rows, int32_t C, int32_t rows_per_img, int32_t mod_stride, void* stream) {
    SDVAR_CHECK_ARG(out || out_planes, "op_ln_modulate: no output");
    return ln_modulate(const_cast<float*>(x), scale, shift, out, out_planes, (size_t)plane_stride, rows, C, rows_per_img, mod_stride, nullptr, plane_format, (hipStream_t)stream);
}
int sdvar_op_split_planes_f16(const float* x, uint16_t* planes, int32_t rows, int32_t cols, uint64_t plane_stride, float* scale, void* stream) {
    if (scale) SDVAR_TRY(weight_scale_f16(x, (size_t)rows * cols, scale, (hipStream_t)stream));
    return split_planes_f16(x, planes, rows, cols, (size_t)plane_stride, scale, (hipStream_t)stream);
}
int sdvar_op_gemm_f16x2(const uint16_t* Xp, uint64_t x_plane_stride, const uint16_t* Wp, uint64_t w_plane_stride, const float* w_scale, const float* bias, float* out,
                        int32_t ldo, uint16_t* out_planes, uint64_t out_plane_stride, int32_t M, int32_t N, int32_t K, int32_t epi, const float* res, int32_t ldres,
                        const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream) {
    ProfScope ps(M >= 1024 ? 0 : 9, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), (hipStream_t)stream);
    return gemm_f16x2_nt(Xp, (size_t)x_plane_stride, Wp, (size_t)w_plane_stride, w_scale ? w_scale + 1 : nullptr, bias, out, ldo, out_planes, (size_t)out_plane_stride, M, N, K,
                         epi, res, ldres, gate, rows_per_gate, gate_stride, nullptr, (hipStream_t)stream);
}
int sdvar_op_split_planes(const float* x, uint16_t* planes, int32_t rows, int32_t cols, uint64_t plane_stride, void* stream) {
    return split_planes(x, planes, rows, cols, (size_t)plane_stride, (hipStream_t)stream);
}
int sdvar_op_gemm_bf16x3(const uint16_t* Xp, uint64_t x_plane_stride, const uint16_t* Wp, uint64_t w_plane_stride, const float* bias, float* out, int32_t ldo,
                         uint16_t* out_planes, uint64_t out_plane_stride, int32_t M, int32_t N, int32_t K, int32_t epi, const float* res, int32_t ldres,
                         const float* gate, int32_t rows_per_gate, int32_t gate_stride, void* stream) {
    ProfScope ps(M >= 1024 ? 0 : 9, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), (hipStream_t)stream);
    return gemm_bf16x3_nt(Xp, (size_t)x_plane_stride, Wp, (size_t)w_plane_stride, bias, out, ldo, out_planes, (size_t)out_plane_stride, M, N, K, epi, res, ldres,
                          gate, rows_per_gate, gate_stride, nullptr, (hipStream_t)stream);
}
int sdvar_op_qk_norm_append(const float* qkv, const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int32_t kv_f16, int32_t R, int32_t l,
                            int32_t H, int32_t Lmax, int32_t pos0, void* stream) {
    return qk_norm_append(qkv, scale_mul, q_out, k_cache, v_cache, kv_f16, R, l, H, Lmax, pos0, nullptr, 0, (hipStream_t)stream);
}
int sdvar_op_attention(const float* q, const void* kc, const void* vc, int32_t kv_f16, float* out, uint16_t* out_planes, uint64_t plane_stride, int32_t plane_format,
                       int32_t R, int32_t H, int32_t l, int32_t Lmax, int32_t Ktot, int32_t n, const int32_t* qbeg, const int32_t* vis, void* stream) {
    SDVAR_CHECK_ARG(qbeg && vis && n >= 1 && n <= SDVAR_MAX_STAGES, "op_attention: bad stage table");
    double lk = 0;
    for (int j = 0; j < n; ++j) lk += (double)((j + 1 < n ? qbeg[j + 1] : l) - qbeg[j]) * vis[j];
    ProfScope ps(l <= 36 ? 8 : 1, 4.0 * R * H * 64.0 * lk, R * H * 64.0 * ((kv_f16 ? 4.0 : 8.0) * Ktot + 8.0 * l), (hipStream_t)stream);
    return attention_f32(q, kc, vc, kv_f16, out, out_planes, (size_t)plane_stride, plane_format, R, H, l, Lmax, Ktot, n, qbeg, vis, (hipStream_t)stream);
}
int sdvar_op_noise_fill(float* q, int32_t B, int32_t l, int32_t V, uint64_t seed, uint32_t draw, uint32_t image_offset, void* stream) {
    return noise_fill(q, B, l, V, seed, draw, image_offset, (hipStream_t)stream);
}

int sdvar_debug_set_gemm_cfg(int32_t bm, int32_t split) {
    // f16x2 only: bm 512 = the 256 x 256 tile kernel, 768 = the 256 x 192 one; bm 16 = the skinny kernel (M <= 80); bm 256 with split -T = hybrid tail split T ways (the other modes take their nearest tile)
    SDVAR_CHECK_ARG(bm == 0 || bm == 16 || bm == 32 || bm == 64 || bm == 128 || bm == 256 || bm == 512 || bm == 768, "debug_set_gemm_cfg: bm %d", bm);
    SDVAR_CHECK_ARG(split >= -64 && split <= 64 && (split >= 0 || bm == 256), "debug_set_gemm_cfg: split %d", split);
    debug_set_gemm_cfg(bm >= 512 ? 256 : bm == 16 ? 32 : bm, split < 0 ? 0 : split);
    debug_set_gemm_cfg_p(bm >= 512 ? 256 : bm == 16 ? 32 : bm, split < 0 ? 0 : split);
    debug_set_gemm_cfg_h(bm, split);
    return SDVAR_OK;
}

int sdvar_debug_set_qkv_fuse(int32_t on) { debug_set_qkv_fuse(on); return SDVAR_OK; }
int sdvar_debug_set_rowblk(int32_t on) { debug_set_rowblk(on); return SDVAR_OK; }          // 0 off, 1 default (32 .. 80 rows), 2 every call of at most 80 rows
int sdvar_op_gemm_rowblk(const float* x, int32_t ldx, const float* scale, const float* shift, int32_t rows_per_img, int32_t mod_stride, const uint16_t* Xp, uint64_t x_plane_stride,
                         const uint16_t* Wp, uint64_t w_plane_stride, const float* w_scale, const float* bias, float* out, int32_t ldo, uint16_t* out_planes, uint64_t out_plane_stride,
                         int32_t M, int32_t N, int32_t K, int32_t epi, const float* res, int32_t ldres, const float* gate, int32_t rows_per_gate, int32_t gate_stride,
                         const float* scale_mul, float* q_out, void* k_cache, void* v_cache, int32_t l, int32_t H, int32_t Lp, int32_t pos0, int32_t kv_fmt, void* stream) {
    SDVAR_CHECK_ARG(gemm_f16x2_rowblk_ok(M, N, K, x != nullptr, q_out != nullptr), "op_gemm_rowblk: M=%d N=%d K=%d outside the row-block kernel's range (M <= 80; K <= 1024 with a LayerNorm operand, <= 4096 with planes) or the kernel is switched off", M, N, K);
    ProfScope ps(9, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), (hipStream_t)stream);
    return gemm_f16x2_rowblk(x, ldx, scale, shift, rows_per_img, mod_stride, Xp, (size_t)x_plane_stride, Wp, (size_t)w_plane_stride, w_scale ? w_scale + 1 : nullptr, bias, out, ldo, out_planes,
                             (size_t)out_plane_stride, M, N, K, epi, res, ldres, gate, rows_per_gate, gate_stride, scale_mul, q_out, k_cache, v_cache, l, H, Lp, pos0, kv_fmt, (hipStream_t)stream);
}

// Kernel variants that otherwise only an environment variable read at first use selects (A/B runs): tests switch them inside one process.  value < 0 = back to the
// environment / default.
int sdvar_debug_set_variant(const char* name, int32_t value) {
    SDVAR_CHECK_ARG(name, "debug_set_variant: null name");
    const std::string n(name);
    if (n == "gemm_h4_var") { SDVAR_CHECK_ARG(value <= 3, "gemm_h4_var %d", value); debug_set_h4_var(value); }
    else if (n == "gemm_h2_stages") { SDVAR_CHECK_ARG(value < 0 || value == 2 || value == 3 || value == 4 || value == 5 || value == 6, "gemm_h2_stages %d", value); debug_set_h2_stages(value); }
    else if (n == "gemm_small_pp") { SDVAR_CHECK_ARG(value <= 2, "gemm_small_pp %d", value); debug_set_small_pp(value); }
    else if (n == "attn_pp_sched") { SDVAR_CHECK_ARG(value <= 3, "attn_pp_sched %d", value); debug_set_attn_pp_sched(value); }
    else if (n == "conv_pp") { SDVAR_CHECK_ARG(value <= 2, "conv_pp %d", value); debug_set_conv_pp(value); }
    else { set_error("debug_set_variant: unknown variant '%s'", name); return SDVAR_ERR_ARG; }
    return SDVAR_OK;
}

int sdvar_debug_get_gemm_cfg(int32_t* out4) {
    SDVAR_CHECK_ARG(out4, "debug_get_gemm_cfg: null");
    int v[4]; debug_get_gemm_cfg_h(v);
    for (int i = 0; i < 4; ++i) out4[i] = v[i];
    return SDVAR_OK;
}

int sdvar_debug_set_f16x2_guard(int32_t on) {
    if (on && !g_guard_cnt) {
        SDVAR_HIP(hipMalloc((void**)&g_guard_cnt, 4 * sizeof(unsigned long long)));
        SDVAR_HIP(hipMemset(g_guard_cnt, 0, 4 * sizeof(unsigned long long)));
    }
    g_guard_on = on != 0;
    return SDVAR_OK;
}

int sdvar_debug_get_f16x2_guard(uint64_t* out4, int32_t reset) {
    SDVAR_CHECK_ARG(out4, "debug_get_f16x2_guard: null");
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    if (!g_guard_cnt) return SDVAR_OK;
    SDVAR_HIP(hipDeviceSynchronize());
    SDVAR_HIP(hipMemcpy(out4, g_guard_cnt, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) SDVAR_HIP(hipMemset(g_guard_cnt, 0, 4 * sizeof(unsigned long long)));
    return SDVAR_OK;
}

int sdvar_debug_set_gemm_stamps(uint64_t* stamps) { debug_set_gemm_stamps((unsigned long long*)stamps); return SDVAR_OK; }

// ---------------------------------------------------------------------------------------------------- profiling
int sdvar_prof_enable(int32_t on) {
    g_prof_on = on != 0;
    if (on) { for (int i = 0; i < SDVAR_PROF_CLASSES; ++i) { g_ms[i] = g_fl[i] = g_by[i] = 0; g_n[i] = 0; } }
    return SDVAR_OK;
}

int sdvar_prof_collect(double* ms, int64_t* launches, double* flops, double* bytes) {
    for (auto& r : g_prof) {
        SDVAR_HIP(hipEventSynchronize(r.e1));
        float t = 0.f;
        SDVAR_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        g_ms[r.cls] += t; g_n[r.cls] += 1; g_fl[r.cls] += r.flops; g_by[r.cls] += r.bytes;
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    for (int i = 0; i < SDVAR_PROF_CLASSES; ++i) {
        if (ms) ms[i] = g_ms[i];
        if (launches) launches[i] = g_n[i];
        if (flops) flops[i] = g_fl[i];
        if (bytes) bytes[i] = g_by[i];
    }
    return SDVAR_OK;
}

}  // extern "C"
