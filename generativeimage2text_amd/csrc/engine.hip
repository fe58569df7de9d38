// Host side of the GIT engine: weight ingest/repack, workspaces, the forward schedule
// (ViT encode -> decoder prefill over image tokens -> KV-cached decode steps -> device search)
// and the C ABI of include/gitmi.h.
//
// Schedule vs. the reference (SURVEY.md headline facts 2/3): the reference re-runs the visual
// projection and all decoder layers over [image | text] tokens at every decode step and for every
// beam copy.  Image rows never attend to text (mask top-right = -inf) and text is causal, so
// computing the image rows once per image and caching K/V is exact; this engine does that.
#include "../../include/gitmi.h"
#include "abi_common.h"
#include "launchers.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace gitmi;

// Entry points of the measurement build (include/gitmi_experiment.h): schedules that measured slower than the default and
// debug hooks.  The product libraries keep the code paths (they share the generate machinery) but do not export them.
#ifdef GITMI_EXPERIMENT
#include "../../include/gitmi_experiment.h"
#define GITMI_EXP_EXPORT extern "C"
#else
#define GITMI_EXP_EXPORT [[maybe_unused]] static
#endif

// ---------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
int gitmi::fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    float amax = 0.f;                   // max |value| (gitmi_load_tensor; every value is finite)
    size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

struct VitLayerW {
    void *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    // LayerNorm folded into the consumer GEMM (fp16-operand build, kernels_gemm10.hip LNF): row-major f16(W . gamma),
    // beta W^T + bias, column sums of the rounded matrix
    void *wqkv_f = nullptr; float *bqkv_f = nullptr, *cs_qkv = nullptr;   // ln_1
    void *w1_f = nullptr;   float *b1_f = nullptr,   *cs_1 = nullptr;     // ln_2
};
struct DecLayerW {
    void *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *lnag = nullptr, *lnab = nullptr, *lnog = nullptr, *lnob = nullptr;
    // decode chain (bf16 mode): the LayerNorm in front of a GEMM folded into its weights (kernels_dgemm.hip):
    // W' = bf16(W . gamma), folded constant beta W^T + bias, column sums of W'
    // All decode-chain matrices are fragment-major copies (gitmi_common.h frag_offset), rows padded to 16.
    void *wqkv_f = nullptr; float *bqkv_f = nullptr, *cs_qkv = nullptr;   // previous layer's output LayerNorm (layer 0: plain, cs_qkv == nullptr)
    void *w1_f = nullptr;   float *b1_f = nullptr,   *cs_1 = nullptr;     // this layer's attention-output LayerNorm
    void *wo_p = nullptr, *w2_p = nullptr;                                 // plain, packed
    // prefill (image rows, large M): the same two folds in ROW-MAJOR layout for gemm_p8_kernel (layer 0: the visual projection's LayerNorm)
    void *wqkv_pf = nullptr; float *bqkv_pf = nullptr, *cs_qkv_p = nullptr;
    void *w1_pf = nullptr;   float *b1_pf = nullptr,   *cs_1_p = nullptr;
};

struct TimedSpan { hipEvent_t a, b; int tag; double flops; };
enum { TAG_VIT = 0, TAG_PREFILL = 1, TAG_DECODE = 2, TAG_GEMM_VIT = 10, TAG_GEMM_OTHER = 11, TAG_STEP = 20 };

struct gitmi_engine {
    gitmi_config cfg{};
    int device = 0;
    bool f32 = false;
    size_t esz = 2;
    bool finalized = false;
    gitmi_engine* parent = nullptr;   // clone: packed weights are borrowed from this engine
    int attn_impl = 1;          // 1 = MFMA flash kernel for full attention (bf16), 0 = VALU kernel

    std::map<std::string, HostTensor> host_w;
    std::vector<void*> allocs;

    // derived dims: N/gh/gw/H/W describe the CURRENT input resolution (gitmi_set_image_shape); *_nat the stored grid
    int N = 0, gh = 0, gw = 0, H = 0, W = 0, Kp = 0, Kp_pad = 0;
    int N_nat = 0, g_nat = 0, Nmax = 0;
    size_t max_pixels = 0;
    float* pos_var = nullptr;          // [Nmax, D] positional table resized to the current grid
    const float* pos_cur = nullptr;    // pos (native grid) or pos_var

    // packed weights
    void* conv_w = nullptr;
    float *cls = nullptr, *pos = nullptr, *lnpre_g = nullptr, *lnpre_b = nullptr, *lnpost_g = nullptr, *lnpost_b = nullptr;
    std::vector<VitLayerW> vit;
    std::vector<float*> temb;
    void* vp_w = nullptr;
    float *vp_b = nullptr, *vp_lng = nullptr, *vp_lnb = nullptr;
    float *words_f = nullptr, *positions_f = nullptr, *emb_lng = nullptr, *emb_lnb = nullptr;
    std::vector<DecLayerW> dec;
    void* out_w = nullptr;
    float* out_b = nullptr;
    void* out_w_f = nullptr;            // vocabulary head folded with the last layer's output LayerNorm (bf16 mode)
    float *out_b_f = nullptr, *cs_out = nullptr;
    double dec_weight_bytes = 0;

    // ViT workspaces (one frame of max_batch images at a time)
    void *patches = nullptr, *v_h = nullptr, *v_qkv = nullptr, *v_ctx = nullptr, *v_u = nullptr;
    float *patch_out = nullptr, *v_x = nullptr;
    // visual features [B, F*N, vfs]
    void* feats = nullptr;
    // prefill workspaces
    float *p_y = nullptr, *p_hf = nullptr;
    void *p_ht = nullptr, *p_ctx = nullptr, *p_u = nullptr;
    std::vector<void*> img_kv;      // per layer [B*N_img, 3d] (prefill layout)
    std::vector<void*> img_kh, img_vh;   // per layer head-major [B][H][N_img][64] (decode layout)
    // decode workspaces
    float *d_y = nullptr, *d_hf = nullptr, *logits = nullptr;
    void *d_ht = nullptr, *d_qkv = nullptr, *d_ctx = nullptr, *d_u = nullptr;
    std::vector<void*> txt_k, txt_v;   // per layer [R_max, T_max, d]
    int ldl = 0;
    bool skinny = true;                 // bf16 decode steps through the folded-LayerNorm GEMM chain (kernels_dgemm.hip)
    // decode chain workspaces: pre-LayerNorm sums of the two N = d GEMMs of a layer (fp32 + bf16) and their strip partials
    float *xa_f = nullptr, *xo_f = nullptr;
    void *xa_b = nullptr, *xo_b = nullptr;
    float2 *stats_a = nullptr, *stats_o = nullptr;
    // per-step candidate lists [R][nparts][slots] (+ (max, sum exp) per part)
    float* part_val = nullptr; int* part_idx = nullptr; float2* part_lse = nullptr;
    int vocab_cols = 128, vocab_nparts = 1;
    // search
    SearchState ss{};
    int ss_cur = 0, ss_len = 0, ss_minP = 1;
    long long* start_dev = nullptr;     // [max_batch][max_text_len] start tokens of every sentence
    int *plen_dev = nullptr, *img_of_dev = nullptr;
    bool img_identity = true;           // sentence b attends to image b
    // token trie of trie-constrained greedy decoding (gitmi_set_trie; trie_decoder.py) + one cursor per sentence
    int *trie_off = nullptr, *trie_tok = nullptr, *trie_child = nullptr, *trie_cursor = nullptr;
    int trie_nodes = 0;
    bool trie_search = false;           // the current search is GITMI_SEARCH_TRIE
    gitmi_search sample{};              // sampling parameters of the current search (do_sample, top_k, top_p, temperature, seed)
    int attn_dbg = 0, dgemm_dbg = 0;    // timing experiments (GITMI_ATTN_DBG, GITMI_DGEMM_DBG)
    int attn_pw = 0;                    // (sentence, head) pairs per workgroup of the decode attention (GITMI_ATTN_PW; 0 = by policy)
    int attn_nh = 0;                    // waves per (sentence, head) pair of the decode attention (GITMI_ATTN_NH: 1 / 2)
    int attn_ppw = 0;                   // pairs a wave of the packed one-wave decode attention serves one after the other (0 = by policy)
    int attn_stream = -1;               // workgroups of the streaming decode attention (0 = register kernels; -1 = by policy)
    bool shared_device = false;         // gitmi_set_shared_device: other contexts run beside this one
    int dgemm_no_row_walk = -1;         // A/B (GITMI_DGEMM_NO_ROW_WALK=0|1; -1 = by policy)
    int dgemm_strips = -1;              // 16-column strips per workgroup of the wide chain GEMMs at <= 64 rows (1, 2, 4, 6; -1 = by policy)
    int vocab_wgs = -1;                 // workgroups of the vocabulary head (each walks ceil(239 / n) column blocks; 0 = one per block; -1 = by policy)
    int gemm_tall = 1;                  // serving policy: encoder GEMMs always on 256-row tiles (1) or on the modelled height (0; GITMI_GEMM_TALL)
    int dgemm_rows = 0;                 // rows per workgroup of the N = 768 chain GEMMs (GITMI_DGEMM_ROWS: 16 / 32 / 64; 0 = by policy)
    int decode_skip = 0;                // MEASUREMENT BUILDS ONLY (GITMI_EXPERIMENT, GITMI_DECODE_SKIP): launches of the decode chain left
                                        // out -- 1 attention, 2 QKV / FFN1 GEMMs, 4 out-proj / FFN2 GEMMs, 8 vocabulary head (ids are garbage)
    bool use_temb = true;               // add img_temperal_embedding[i] to frame i (the reference does so only for a LIST of frames)
    std::vector<int> plen_host, img_of_host;
    const float* const* frames_dummy = nullptr;

    // state of the current batch
    int cur_B = 0, cur_F = 0, cur_Nimg = 0;
    bool have_feats = false, have_prefill = false;

    // profiling: 1 = eager launches with HIP events around phases, decode steps and every GEMM;
    //            2 = hipGraph replays, the call split into an encode graph and a decode graph with events between them
    //                (what the production path costs: no per-launch host work, no event records inside the chain)
    int profile_mode = 0;
    bool profiling = false;             // profile_mode == 1
    hipGraph_t graph_b = nullptr;
    hipGraphExec_t graph_exec_b = nullptr;
    bool graph_is_split = false;
    // residual streams of the image encoder and the prefill (v_x, p_y, p_hf) stored in fp16 instead of fp32 (bf16 mode
    // only, the default there; GITMI_STREAM_F16=0 keeps fp32): half the bytes of their read-modify-writes at 2^-11 relative
    // rounding.  Measured (profiles/r03_a_bench_f16_*.json, interleaved A/B): encode + prefill 5.31 -> 5.03 ms,
    // 9.48k -> 9.82k captions/s, logit error 0.01118 -> 0.01094, the same 50 of 64 rows identical to the reference.
    bool stream_f16 = false;
    // LayerNorm folding in the encoder and the prefill (round 6; fp16-operand build only: the fp16 stream rows are the
    // consumer GEMM's A operand as they are).  Row partials (sum, sumsq) per 256-column tile: [rows][4], ping-pong for
    // the post-norm prefill (a producer tile reads the previous partials of a row while another tile writes the new ones).
    bool ln_fold = false;
    bool ln_fold_ready = false;         // folded matrices and partial buffers exist (decided at gitmi_create; gitmi_set_ln_fold switches the use)
    float2 *v_part = nullptr, *p_part[2] = {nullptr, nullptr};
    hipEvent_t gev[3] = {nullptr, nullptr, nullptr};
    // serving schedule: this context's image encoder starts only after `enc_after`'s has finished (at most one encoder
    // in flight on the device; decode chains of the other contexts fill in beside it)
    gitmi_engine* enc_after = nullptr;
    std::vector<gitmi_engine*> enc_watchers;   // contexts whose enc_after is this one (they wait on enc_done)
    hipEvent_t enc_done = nullptr;
    double split_encode_ms = 0, split_decode_ms = 0;
    int split_calls = 0, split_steps = 0;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
    double last_decode_step_bytes = 0;
    bool use_graph = true;

    // hipGraph cache for gitmi_generate
    struct GraphKey {
        int B, Q, F, P, kind, k, pn, T, H, W, ragged, ident, temb; double lp;
        int smp, top_k, nh; double top_p, temp, rp; unsigned long long seed;
        bool operator==(const GraphKey& o) const {
            return rp == o.rp && nh == o.nh && smp == o.smp && top_k == o.top_k && top_p == o.top_p && temp == o.temp && seed == o.seed && B == o.B && Q == o.Q && F == o.F && P == o.P && kind == o.kind && k == o.k && pn == o.pn && T == o.T &&
                   H == o.H && W == o.W && ragged == o.ragged && ident == o.ident && temb == o.temb && lp == o.lp;
        }
    };
    bool graph_valid = false;
    GraphKey graph_key{};
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    std::vector<float*> frame_stage;   // engine-owned copies of the input frames (graph inputs)
    long long* out_tokens = nullptr;   // graph outputs, copied to the caller's buffers after the launch
    float* out_lp = nullptr;
    int* out_info = nullptr;
    int* out_sent = nullptr;           // [max_batch][2] per-sentence (length, early) of the last generate
    hipStream_t own_stream = nullptr;  // used when the caller passes the (uncapturable) null stream
    hipEvent_t fence_in = nullptr, fence_out = nullptr;
};

// ---------------------------------------------------------------------------------------
static int dev_alloc(gitmi_engine* e, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HIPCK(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    return 0;
}
template <typename T> static int dev_alloc_t(gitmi_engine* e, T** p, size_t count) {
    return dev_alloc(e, reinterpret_cast<void**>(p), count * sizeof(T));
}

static hipEvent_t get_event(gitmi_engine* e) {
    if (e->event_next == e->event_pool.size()) {
        hipEvent_t ev;
        hipEventCreate(&ev);
        e->event_pool.push_back(ev);
    }
    return e->event_pool[e->event_next++];
}
struct SpanGuard {
    gitmi_engine* e; hipStream_t s; size_t idx; bool on;
    SpanGuard(gitmi_engine* e_, hipStream_t s_, int tag, double flops) : e(e_), s(s_), idx(0), on(e_->profiling) {
        if (!on) return;
        TimedSpan sp{get_event(e), get_event(e), tag, flops};
        hipEventRecord(sp.a, s);
        idx = e->spans.size();
        e->spans.push_back(sp);
    }
    ~SpanGuard() { if (on) hipEventRecord(e->spans[idx].b, s); }
};

// serving policy of the encoder GEMM's tile height (kernels_gemm10.hip: launch_gemm_p8): 256-row tiles whatever the round
// fill, because other contexts' kernels fill the CUs a partial round leaves idle.  gemm_tall (measurement builds): 1 always,
// 0 never (the modelled height, as for a context alone), 2 only for the wide GEMMs (N >= 2048), 3 only for the N < 2048 ones
static bool gemm_tall_tiles(const gitmi_engine* e, int N) {
    if (!e->shared_device) return false;
    return e->gemm_tall == 1 || (e->gemm_tall == 2 && N >= 2048) || (e->gemm_tall == 3 && N < 2048);
}

// GEMM wrapper: C = act(A W^T + bias) (+ res)
static int gemm(gitmi_engine* e, hipStream_t s, const void* A, int lda, const void* W, const float* bias,
                const float* res, int ldr, void* C, int ldc, bool out_f32, int M, int N, int K, int act, int tag) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.res = res; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.act = act;
    g.shared = gemm_tall_tiles(e, N) ? 1 : 0;
    SpanGuard sp(e, s, tag, 2.0 * (double)M * (double)N * (double)K);
    HIPCK(launch_gemm(g, e->f32, out_f32, s));
    return 0;
}

// GEMM whose output (and residual, if any) are rows of a residual stream: fp32, or fp16 with stream_f16
static int gemm_stream(gitmi_engine* e, hipStream_t s, const void* A, int lda, const void* W, const float* bias,
                       const void* res, int ldr, void* C, int ldc, int M, int N, int K, int tag) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.res = (const float*)res; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.act = 0;
    g.out_f16 = e->stream_f16 ? 1 : 0;
    g.shared = gemm_tall_tiles(e, N) ? 1 : 0;
    SpanGuard sp(e, s, tag, 2.0 * (double)M * (double)N * (double)K);
    HIPCK(launch_gemm(g, e->f32, !e->stream_f16, s));
    return 0;
}
// ---- folded LayerNorm (e->ln_fold): what a GEMM needs to know about the LayerNorm in front of it / of its residual
struct LnRef {
    const float2* part = nullptr; int nparts = 0; int D = 0; float eps = 0.f;
    const float* gamma = nullptr; const float* beta = nullptr;       // residual form only
};
// consumer: C = act(LayerNorm(x) W^T + b) with x the raw stream rows, W / bias / colsum the folded set
static int gemm_ln(gitmi_engine* e, hipStream_t s, const void* x, int ldx, const void* Wf, const float* bias_f, const float* colsum,
                   const LnRef& ln, void* C, int ldc, int M, int N, int K, int act, int tag) {
    GemmArgs g{};
    g.A = x; g.W = Wf; g.bias = bias_f; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldc = ldc; g.act = act;
    g.ln_part = ln.part; g.ln_nparts = ln.nparts; g.ln_colsum = colsum; g.ln_inv_d = 1.0f / (float)ln.D; g.ln_eps = ln.eps;
    g.shared = gemm_tall_tiles(e, N) ? 1 : 0;
    SpanGuard sp(e, s, tag, 2.0 * (double)M * (double)N * (double)K);
    HIPCK(launch_gemm(g, false, false, s));
    return 0;
}
// producer: stream rows C = A W^T + b (+ res, or + LayerNorm(res) when res_ln is given) and their row partials
static int gemm_stream_part(gitmi_engine* e, hipStream_t s, const void* A, int lda, const void* W, const float* bias,
                            const void* res, int ldr, const LnRef* res_ln, void* C, int ldc, float2* part_out, int M, int N, int K,
                            int tag) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.res = (const float*)res; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.act = 0;
    g.out_f16 = 1;
    g.part_out = part_out;
    if (res_ln) {
        g.res_part = res_ln->part; g.res_nparts = res_ln->nparts; g.res_gamma = res_ln->gamma; g.res_beta = res_ln->beta;
        g.res_inv_d = 1.0f / (float)res_ln->D; g.res_eps = res_ln->eps;
    }
    g.shared = gemm_tall_tiles(e, N) ? 1 : 0;
    SpanGuard sp(e, s, tag, 2.0 * (double)M * (double)N * (double)K);
    HIPCK(launch_gemm(g, false, false, s));
    return 0;
}
// does a 16-bit GEMM of this shape run on gemm_p8_kernel (the only kernel with the folded epilogues)?
static bool on_p8(const void* A, int lda, const void* W, const void* C, int ldc, int M, int N, int K, bool stream_out) {
    GemmArgs g{};
    g.A = A; g.W = W; g.C = const_cast<void*>(C); g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.out_f16 = stream_out ? 1 : 0;
    return gemm_uses_p8(g, false, false);
}
// LayerNorm of stream rows x -> operand copy y_t (compute dtype) [+ stream copy y_s]
static int ln_stream(gitmi_engine* e, hipStream_t s, const void* x, int ldx, const float* gamma, const float* beta, float eps,
                     void* y_t, int ld_t, void* y_s, int ld_s, int rows, int D) {
    if (e->stream_f16)
        HIPCK(launch_layernorm_s16(x, ldx, gamma, beta, eps, nullptr, y_t, ld_t, false, y_s, ld_s, rows, D, 0, 0, 0, s));
    else
        HIPCK(launch_layernorm((const float*)x, ldx, gamma, beta, eps, nullptr, y_t, ld_t, e->f32, (float*)y_s, ld_s, rows, D,
                               0, 0, 0, s));
    return 0;
}

// ---------------------------------------------------------------------------------------
extern "C" int gitmi_abi_version(void) { return GITMI_ABI_VERSION; }
// 16-bit operand type this library was built for: GITMI_DTYPE_BF16 (libgitmi.so) or GITMI_DTYPE_F16 (libgitmi_f16.so)
extern "C" int gitmi_operand_dtype(void) {
#ifdef GITMI_OPS_F16
    return GITMI_DTYPE_F16;
#else
    return GITMI_DTYPE_BF16;
#endif
}
extern "C" const char* gitmi_last_error(void) { return g_err; }

extern "C" int gitmi_create(const gitmi_config* cfg, int device, gitmi_engine** out) {
    if (!cfg || !out) return fail("gitmi_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("gitmi_create: no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("gitmi_create: device %d out of range (%d devices)", device, ndev);
    HIPCK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("gitmi_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    const gitmi_config& c = *cfg;
    if (c.vit_width % c.vit_heads || c.vit_width / c.vit_heads != 64) return fail("ViT head_dim must be 64");
    if (c.dec_hidden % c.dec_heads || c.dec_hidden / c.dec_heads != 64) return fail("decoder head_dim must be 64");
    if (c.image_size % c.patch) return fail("image_size must be a multiple of patch");
    if (c.max_image_pixels < 0 || c.max_image_tokens < 0) return fail("negative image capacity");
    if (c.vit_width > 1024 || c.dec_hidden > 1024) return fail("hidden sizes above 1024 are not supported");
    if (c.vit_width % 64 || c.dec_hidden % 64 || c.dec_ffn % 64) return fail("hidden sizes must be multiples of 64");
    if (c.max_batch < 1 || c.max_beams < 1 || c.max_beams > 8 || c.max_frames < 1 || c.max_text_len < 2)
        return fail("bad capacity (max_batch>=1, 1<=max_beams<=8, max_frames>=1, max_text_len>=2)");
    if (c.max_text_len > c.max_pos) return fail("max_text_len exceeds max_pos");
    if (c.precision != GITMI_PREC_BF16 && c.precision != GITMI_PREC_F32) return fail("bad precision");

    gitmi_engine* e = new gitmi_engine();
    e->cfg = c;
    e->device = device;
    e->f32 = c.precision == GITMI_PREC_F32;
    e->esz = e->f32 ? 4 : 2;
    e->attn_impl = e->f32 ? 0 : 1;
    e->g_nat = e->gh = e->gw = c.image_size / c.patch;
    e->N_nat = e->N = e->g_nat * e->g_nat + 1;
    e->H = e->W = c.image_size;
    e->Nmax = std::max(e->N_nat, c.max_image_tokens);
    e->max_pixels = std::max((size_t)c.image_size * c.image_size, (size_t)c.max_image_pixels);
    e->Kp = 3 * c.patch * c.patch;
    e->Kp_pad = round_up(e->Kp, 64);
    e->stream_f16 = !e->f32;
#ifdef GITMI_OPS_F16
    e->ln_fold = !e->f32 && c.vit_width % 256 == 0 && c.dec_hidden % 256 == 0;
#endif
#ifdef GITMI_EXPERIMENT
    // measurement builds only (libgitmi_exp.so, `make exp`): kernel-shape overrides and work-skipping switches for A/B runs
    // and timing decompositions.  The product libraries read no environment.
    if (const char* env = getenv("GITMI_ATTN_IMPL")) e->attn_impl = e->f32 ? 0 : atoi(env);
    if (const char* env = getenv("GITMI_GRAPH")) e->use_graph = atoi(env) != 0;
    if (const char* env = getenv("GITMI_ATTN_DBG")) e->attn_dbg = atoi(env);
    if (const char* env = getenv("GITMI_ATTN_PW")) e->attn_pw = atoi(env);
    if (const char* env = getenv("GITMI_ATTN_NH")) e->attn_nh = atoi(env);
    if (const char* env = getenv("GITMI_ATTN_PPW")) e->attn_ppw = atoi(env);
    if (const char* env = getenv("GITMI_ATTN_STREAM")) e->attn_stream = atoi(env);
    if (const char* env = getenv("GITMI_DECODE_SKIP")) e->decode_skip = atoi(env);
    if (const char* env = getenv("GITMI_GEMM_TALL")) e->gemm_tall = atoi(env);
    if (const char* env = getenv("GITMI_DGEMM_ROWS")) e->dgemm_rows = atoi(env);
    if (const char* env = getenv("GITMI_DGEMM_STRIPS")) e->dgemm_strips = atoi(env);
    if (const char* env = getenv("GITMI_DGEMM_NO_ROW_WALK")) e->dgemm_no_row_walk = atoi(env);
    if (const char* env = getenv("GITMI_DGEMM_DBG")) e->dgemm_dbg = atoi(env);
    if (const char* env = getenv("GITMI_VOCAB_WGS")) e->vocab_wgs = atoi(env);
    if (const char* env = getenv("GITMI_SKINNY")) e->skinny = atoi(env) != 0;
    if (const char* env = getenv("GITMI_STREAM_F16")) e->stream_f16 = !e->f32 && atoi(env) != 0;
    if (const char* env = getenv("GITMI_GEMM_IMPL")) set_gemm_impl(atoi(env));
#endif
    e->ln_fold = e->ln_fold && e->stream_f16;
    e->ln_fold_ready = e->ln_fold;
    if (attn_decode_configure() != hipSuccess) { delete e; return fail("hipFuncSetAttribute failed"); }
    *out = e;
    return 0;
}

static void destroy_graph(gitmi_engine* e) {
    if (e->graph_exec) hipGraphExecDestroy(e->graph_exec);
    if (e->graph) hipGraphDestroy(e->graph);
    if (e->graph_exec_b) hipGraphExecDestroy(e->graph_exec_b);
    if (e->graph_b) hipGraphDestroy(e->graph_b);
    e->graph_exec = e->graph_exec_b = nullptr;
    e->graph = e->graph_b = nullptr;
    e->graph_valid = false;
}

extern "C" void gitmi_destroy(gitmi_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    // serving-schedule links: nobody keeps a pointer to a destroyed context
    if (e->enc_after) {
        auto& w = e->enc_after->enc_watchers;
        w.erase(std::remove(w.begin(), w.end(), e), w.end());
    }
    for (gitmi_engine* w : e->enc_watchers) w->enc_after = nullptr;
    destroy_graph(e);
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    if (e->fence_in) hipEventDestroy(e->fence_in);
    if (e->fence_out) hipEventDestroy(e->fence_out);
    for (auto ev : e->gev)
        if (ev) hipEventDestroy(ev);
    if (e->enc_done) hipEventDestroy(e->enc_done);
    for (auto ev : e->event_pool) hipEventDestroy(ev);
    for (void* p : e->allocs) hipFree(p);
    if (e->trie_off) hipFree(e->trie_off);
    if (e->trie_tok) hipFree(e->trie_tok);
    if (e->trie_child) hipFree(e->trie_child);
    delete e;
}

// ---------------------------------------------------------------------------------------
static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; --exp; }
            man &= 0x3ff;
            out = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) out = sign | 0x7f800000u | (man << 13);
    else out = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &out, 4);
    return f;
}

extern "C" int gitmi_load_tensor(gitmi_engine* e, const char* key, const void* data_host, const int64_t* shape,
                                 int ndim, int dtype) {
    if (!e || !key || !data_host || (ndim > 0 && !shape)) return fail("gitmi_load_tensor: null argument");
    if (e->finalized) return fail("gitmi_load_tensor: weights already finalized");
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);          // torch_common.py:95-99 strips DataParallel prefixes
    if (k == "image_encoder.proj") return 0;                   // unused with output_grid=True
    const bool known = k.rfind("image_encoder.", 0) == 0 || k.rfind("textual.", 0) == 0 ||
                       k.rfind("img_temperal_embedding.", 0) == 0;
    if (!known) return fail("gitmi_load_tensor: unknown key '%s'", key);
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    const size_t n = t.numel();
    t.data.resize(n);
    if (dtype == GITMI_DTYPE_F32) memcpy(t.data.data(), data_host, n * 4);
    else if (dtype == GITMI_DTYPE_BF16) {
        const uint16_t* p = (const uint16_t*)data_host;
        for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)p[i] << 16; memcpy(&t.data[i], &u, 4); }
    } else if (dtype == GITMI_DTYPE_F16) {
        const uint16_t* p = (const uint16_t*)data_host;
        for (size_t i = 0; i < n; ++i) t.data[i] = half_to_float(p[i]);
    } else return fail("gitmi_load_tensor: bad dtype %d", dtype);
    // a checkpoint with inf / NaN in it fails here, by name, not as garbage ids later
    float amax = 0.f;
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(t.data[i])) return fail("gitmi_load_tensor: '%s' holds a non-finite value at element %zu", key, i);
        amax = std::max(amax, std::fabs(t.data[i]));
    }
    t.amax = amax;
    e->host_w[k] = std::move(t);
    return 0;
}

static int get_w(gitmi_engine* e, const std::string& key, std::initializer_list<int64_t> shape, const HostTensor** out) {
    auto it = e->host_w.find(key);
    if (it == e->host_w.end()) return fail("missing weight '%s'", key.c_str());
    size_t want = 1;
    for (auto s : shape) want *= (size_t)s;
    if (it->second.numel() != want) return fail("weight '%s' has %zu elements, expected %zu", key.c_str(), it->second.numel(), want);
    *out = &it->second;
    return 0;
}
// fp32 vector / table on device
static int up_f32(gitmi_engine* e, const std::string& key, std::initializer_list<int64_t> shape, float** dst) {
    const HostTensor* t;
    RCK(get_w(e, key, shape, &t));
    RCK(dev_alloc_t(e, dst, t->numel()));
    HIPCK(hipMemcpy(*dst, t->data.data(), t->numel() * 4, hipMemcpyHostToDevice));
    return 0;
}
// a MATRIX that becomes an MFMA operand must fit the operand format: fp16 tops out at 65504 (bf16 and f32 share fp32's exponent)
static int operand_range_check(gitmi_engine* e, const char* what, float amax) {
#ifdef GITMI_OPS_F16
    if (!e->f32 && amax > 65504.f)
        return fail("'%s': max |w| = %g is outside the fp16 operand range (65504): load this checkpoint with precision "
                    "\"bf16\" or \"f32\"", what, (double)amax);
#endif
    (void)e; (void)what; (void)amax;
    return 0;
}
// matrix [rows, K] -> compute dtype [rows, Kpad] written at dst + row_off rows
static int up_mat_into(gitmi_engine* e, const std::string& key, int64_t rows, int K, int Kpad, void* dst, size_t row_off) {
    const HostTensor* t;
    RCK(get_w(e, key, {rows, (int64_t)K}, &t));
    RCK(operand_range_check(e, key.c_str(), t->amax));
    float* tmp = nullptr;
    HIPCK(hipMalloc((void**)&tmp, t->numel() * 4));
    hipError_t err = hipMemcpy(tmp, t->data.data(), t->numel() * 4, hipMemcpyHostToDevice);
    if (err == hipSuccess)
        err = launch_convert_pad(tmp, (char*)dst + row_off * (size_t)Kpad * e->esz, e->f32, (size_t)rows, K, Kpad, 0);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    hipFree(tmp);
    HIPCK(err);
    return 0;
}
static int up_mat(gitmi_engine* e, const std::string& key, int64_t rows, int K, int Kpad, void** dst) {
    RCK(dev_alloc(e, dst, (size_t)rows * Kpad * e->esz));
    return up_mat_into(e, key, rows, K, Kpad, *dst, 0);
}
static int up_f32_into(gitmi_engine* e, const std::string& key, int64_t n, float* dst, size_t off) {
    const HostTensor* t;
    RCK(get_w(e, key, {n}, &t));
    HIPCK(hipMemcpy(dst + off, t->data.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    return 0;
}

static int alloc_workspaces(gitmi_engine* e) {
    const gitmi_config& c = e->cfg;
    const size_t esz = e->esz;
    const int D = c.vit_width, d = c.dec_hidden;
    const size_t Mv = (size_t)c.max_batch * c.max_frames * e->Nmax;  // ViT rows: all frames of a call in one pass
    const size_t Mp = (size_t)c.max_batch * c.max_frames * e->Nmax;  // prefill rows
    // fragment-major operand buffers hold whole 16-row tiles, and the wide chain GEMMs / the vocabulary head load their
    // activations four tiles (64 rows) at a time whatever M is: every row-sized buffer is padded to 64 rows
    const size_t R = (size_t)round_up(c.max_batch * c.max_beams, 64);
    const int T = c.max_text_len;
    RCK(dev_alloc(e, &e->patches, (size_t)c.max_batch * c.max_frames * (e->Nmax - 1) * e->Kp_pad * esz));
    RCK(dev_alloc_t(e, &e->patch_out, (size_t)c.max_batch * c.max_frames * (e->Nmax - 1) * D));
    RCK(dev_alloc_t(e, &e->pos_var, (size_t)e->Nmax * D));
    RCK(dev_alloc_t(e, &e->v_x, Mv * D));
    RCK(dev_alloc(e, &e->v_h, Mv * D * esz));
    RCK(dev_alloc(e, &e->v_qkv, Mv * 3 * D * esz));
    RCK(dev_alloc(e, &e->v_ctx, Mv * D * esz));
    RCK(dev_alloc(e, &e->v_u, Mv * 4 * D * esz));
    RCK(dev_alloc(e, &e->feats, Mp * D * esz));
    if (e->ln_fold_ready) {
        // [row][4] (sum, sumsq) per 256-column tile; slots past the row width stay zero for ever
        RCK(dev_alloc_t(e, &e->v_part, Mv * 4));
        RCK(dev_alloc_t(e, &e->p_part[0], Mp * 4));
        RCK(dev_alloc_t(e, &e->p_part[1], Mp * 4));
        HIPCK(hipMemset(e->v_part, 0, Mv * 4 * sizeof(float2)));
        HIPCK(hipMemset(e->p_part[0], 0, Mp * 4 * sizeof(float2)));
        HIPCK(hipMemset(e->p_part[1], 0, Mp * 4 * sizeof(float2)));
    }
    RCK(dev_alloc_t(e, &e->p_y, Mp * d));
    RCK(dev_alloc_t(e, &e->p_hf, Mp * d));
    RCK(dev_alloc(e, &e->p_ht, Mp * d * esz));
    RCK(dev_alloc(e, &e->p_ctx, Mp * d * esz));
    RCK(dev_alloc(e, &e->p_u, Mp * c.dec_ffn * esz));
    e->img_kv.resize(c.dec_layers);
    e->img_kh.resize(c.dec_layers);
    e->img_vh.resize(c.dec_layers);
    e->txt_k.resize(c.dec_layers);
    e->txt_v.resize(c.dec_layers);
    for (int l = 0; l < c.dec_layers; ++l) {
        RCK(dev_alloc(e, &e->img_kv[l], Mp * 3 * d * esz));
        // decode layout; bf16: per (image, head) keys padded to a multiple of 32 (kernels_attn_decode.hip)
        const size_t Mkv = (size_t)c.max_batch * round_up(c.max_frames * e->Nmax, 32);
        RCK(dev_alloc(e, &e->img_kh[l], Mkv * d * esz));
        RCK(dev_alloc(e, &e->img_vh[l], Mkv * d * esz));
        RCK(dev_alloc(e, &e->txt_k[l], R * T * d * esz));
        RCK(dev_alloc(e, &e->txt_v[l], R * T * d * esz));
    }
    RCK(dev_alloc_t(e, &e->d_y, R * d));
    RCK(dev_alloc_t(e, &e->d_hf, R * d));
    RCK(dev_alloc(e, &e->d_ht, R * d * esz));
    RCK(dev_alloc(e, &e->d_qkv, R * 3 * d * esz));
    RCK(dev_alloc(e, &e->d_ctx, R * d * esz));
    RCK(dev_alloc(e, &e->d_u, R * c.dec_ffn * esz));
    RCK(dev_alloc_t(e, &e->xa_f, R * d));
    RCK(dev_alloc_t(e, &e->xo_f, R * d));
    RCK(dev_alloc(e, &e->xa_b, R * d * 2));
    RCK(dev_alloc(e, &e->xo_b, R * d * 2));
    RCK(dev_alloc_t(e, &e->stats_a, R * (size_t)(d / 16)));
    RCK(dev_alloc_t(e, &e->stats_o, R * (size_t)(d / 16)));
    e->ldl = round_up(c.vocab, 8);
    RCK(dev_alloc_t(e, &e->logits, R * e->ldl));
    // candidate lists of a step: the fused vocabulary head writes one list per (row, 128-column workgroup)
    e->vocab_cols = 128;                  // columns per workgroup of the fused head (239 workgroups for the 30522-token vocabulary)
    // only the bf16 decode chain uses the fused head (finalize_weights turns the chain off for vocabularies above 32768
    // tokens); f32 engines and search-only contexts get ONE list per row from row_topm / sample_rows
    e->vocab_nparts = (e->skinny && !e->f32) ? vocab_parts(c.vocab, e->vocab_cols) : 1;
    RCK(dev_alloc_t(e, &e->part_val, R * (size_t)e->vocab_nparts * 16));
    RCK(dev_alloc_t(e, &e->part_idx, R * (size_t)e->vocab_nparts * 16));
    RCK(dev_alloc_t(e, &e->part_lse, R * (size_t)e->vocab_nparts));
    // search state
    SearchState& s = e->ss;
    for (int i = 0; i < 2; ++i) {
        RCK(dev_alloc_t(e, &s.ids[i], R * T));
        RCK(dev_alloc_t(e, &s.kv_src[i], R * T));
        RCK(dev_alloc_t(e, &s.score[i], R));
    }
    RCK(dev_alloc_t(e, &s.done, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.hyp_n, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.hyp_cnt, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.hyp_worst, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.hyp_score, (size_t)c.max_batch * SS_NHMAX));
    RCK(dev_alloc_t(e, &s.hyp_len, (size_t)c.max_batch * SS_NHMAX));
    RCK(dev_alloc_t(e, &s.hyp_seq, (size_t)c.max_batch * SS_NHMAX));
    RCK(dev_alloc_t(e, &s.hyp_tok, (size_t)c.max_batch * SS_NHMAX * T));
    RCK(dev_alloc_t(e, &s.stop, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.early, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &s.info, 4));
    RCK(dev_alloc_t(e, &s.len_norm, (size_t)T + 1));
    RCK(dev_alloc_t(e, &e->start_dev, (size_t)c.max_batch * T));
    RCK(dev_alloc_t(e, &e->plen_dev, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &e->img_of_dev, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &e->trie_cursor, (size_t)c.max_batch));
    RCK(dev_alloc_t(e, &e->out_tokens, (size_t)c.max_batch * SS_NHMAX * T));
    RCK(dev_alloc_t(e, &e->out_lp, (size_t)c.max_batch * SS_NHMAX));
    RCK(dev_alloc_t(e, &e->out_info, 4));
    RCK(dev_alloc_t(e, &e->out_sent, (size_t)c.max_batch * 2));
    HIPCK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&e->fence_in, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&e->fence_out, hipEventDisableTiming));
    e->frame_stage.resize(c.max_frames);
    for (int f = 0; f < c.max_frames; ++f)
        RCK(dev_alloc_t(e, &e->frame_stage[f], (size_t)c.max_batch * 3 * e->max_pixels));
    return 0;
}

// ---- LayerNorm folding for the decode chain (kernels_dgemm.hip) ------------------------------------------------
// An fp32 value rounded to the 16-bit operand type of this build (round-to-nearest-even, as the device conversions do
// it): the column sums of a folded LayerNorm must be taken over exactly the values the MFMA will see.
#ifdef GITMI_OPS_F16
static inline float bf16_round(float f) { return (float)(_Float16)f; }
#else
static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
#endif
// W [rows, K], bias [rows], LayerNorm (gamma, beta) [K] in front of it  ->  device W' (bf16), folded bias, column sums
// frag: fragment-major packing for the decode chain; else row-major for gemm_p8_kernel
static int fold_layernorm(gitmi_engine* e, const std::vector<float>& W, const std::vector<float>& bias,
                          const std::vector<float>& gamma, const std::vector<float>& beta, int64_t rows, int K,
                          void** Wf, float** bf, float** cs, bool frag = true) {
    std::vector<float> wf((size_t)rows * K), b2((size_t)rows), c2((size_t)rows);
    float amax = 0.f;
    for (int64_t n = 0; n < rows; ++n) {
        double sum = 0.0, cst = bias[n];
        const float* w = &W[(size_t)n * K];
        float* o = &wf[(size_t)n * K];
        for (int k = 0; k < K; ++k) {
            amax = std::max(amax, std::fabs(w[k] * gamma[k]));
            o[k] = bf16_round(w[k] * gamma[k]);
            sum += (double)o[k];
            cst += (double)beta[k] * (double)w[k];
        }
        c2[n] = (float)sum;
        b2[n] = (float)cst;
    }
    RCK(operand_range_check(e, "a matrix with the LayerNorm gain in front of it folded in (W . gamma)", amax));
    const int64_t rows_pad = (rows + 127) / 128 * 128;       // the vocabulary head reads bias / colsum a workgroup (128 columns) at a time
    RCK(dev_alloc(e, Wf, (size_t)rows_pad * K * 2));
    float* tmp = nullptr;
    void* tmp_b = nullptr;
    HIPCK(hipMalloc((void**)&tmp, wf.size() * 4));
    hipError_t err = hipSuccess;
    if (frag) err = hipMalloc(&tmp_b, wf.size() * 2);
    if (err == hipSuccess) err = hipMemcpy(tmp, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = launch_convert_pad(tmp, frag ? tmp_b : *Wf, false, (size_t)rows, K, K, 0);
    if (err == hipSuccess && frag) err = launch_frag_pack(tmp_b, *Wf, (int)rows, (int)rows_pad, K, 0);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    hipFree(tmp);
    if (tmp_b) hipFree(tmp_b);
    HIPCK(err);
    b2.resize((size_t)rows_pad, 0.f);
    c2.resize((size_t)rows_pad, 0.f);
    RCK(dev_alloc_t(e, bf, (size_t)rows_pad));
    RCK(dev_alloc_t(e, cs, (size_t)rows_pad));
    HIPCK(hipMemcpy(*bf, b2.data(), (size_t)rows_pad * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(*cs, c2.data(), (size_t)rows_pad * 4, hipMemcpyHostToDevice));
    return 0;
}
// fragment-major copy of an already packed row-major bf16 matrix [rows, K]
static int pack_frag(gitmi_engine* e, const void* src, int64_t rows, int K, void** dst) {
    const int64_t rows_pad = (rows + 15) / 16 * 16;
    RCK(dev_alloc(e, dst, (size_t)rows_pad * K * 2));
    HIPCK(launch_frag_pack(src, *dst, (int)rows, (int)rows_pad, K, 0));
    HIPCK(hipDeviceSynchronize());
    return 0;
}
static int host_vec(gitmi_engine* e, const std::string& key, size_t n, const std::vector<float>** out) {
    auto it = e->host_w.find(key);
    if (it == e->host_w.end()) return fail("missing weight '%s'", key.c_str());
    if (it->second.numel() != n) return fail("weight '%s' has %zu elements, expected %zu", key.c_str(), it->second.numel(), n);
    *out = &it->second.data;
    return 0;
}
static int fold_decoder(gitmi_engine* e) {
    const gitmi_config& c = e->cfg;
    const int d = c.dec_hidden, f = c.dec_ffn, V = c.vocab;
    const std::string base = "textual.transformer.encoder.layer.";
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string pre = base + std::to_string(i) + ".";
        DecLayerW& L = e->dec[i];
        const std::vector<float>*g, *b, *w, *bi;
        if (i > 0) {      // QKV behind the previous layer's output LayerNorm
            const std::string prev = base + std::to_string(i - 1) + ".";
            RCK(host_vec(e, prev + "output.LayerNorm.weight", d, &g));
            RCK(host_vec(e, prev + "output.LayerNorm.bias", d, &b));
            std::vector<float> wq((size_t)3 * d * d), bq((size_t)3 * d);
            const char* names[3] = {"query", "key", "value"};
            for (int j = 0; j < 3; ++j) {
                RCK(host_vec(e, pre + "attention.self." + names[j] + ".weight", (size_t)d * d, &w));
                RCK(host_vec(e, pre + "attention.self." + names[j] + ".bias", d, &bi));
                std::copy(w->begin(), w->end(), wq.begin() + (size_t)j * d * d);
                std::copy(bi->begin(), bi->end(), bq.begin() + (size_t)j * d);
            }
            RCK(fold_layernorm(e, wq, bq, *g, *b, 3 * d, d, &L.wqkv_f, &L.bqkv_f, &L.cs_qkv));
        } else {          // layer 0 consumes the embedding LayerNorm's output directly: plain weights, packed
            RCK(pack_frag(e, L.wqkv, 3 * d, d, &L.wqkv_f));
            L.bqkv_f = L.bqkv;
        }
        RCK(pack_frag(e, L.wo, d, d, &L.wo_p));
        RCK(pack_frag(e, L.w2, d, f, &L.w2_p));
        RCK(host_vec(e, pre + "attention.output.LayerNorm.weight", d, &g));
        RCK(host_vec(e, pre + "attention.output.LayerNorm.bias", d, &b));
        RCK(host_vec(e, pre + "intermediate.dense.weight", (size_t)f * d, &w));
        RCK(host_vec(e, pre + "intermediate.dense.bias", f, &bi));
        RCK(fold_layernorm(e, *w, *bi, *g, *b, f, d, &L.w1_f, &L.b1_f, &L.cs_1));
    }
    const std::string last = base + std::to_string(c.dec_layers - 1) + ".";
    const std::vector<float>*g, *b, *w, *bi;
    RCK(host_vec(e, last + "output.LayerNorm.weight", d, &g));
    RCK(host_vec(e, last + "output.LayerNorm.bias", d, &b));
    RCK(host_vec(e, "textual.output.weight", (size_t)V * d, &w));
    RCK(host_vec(e, "textual.output.bias", V, &bi));
    RCK(fold_layernorm(e, *w, *bi, *g, *b, V, d, &e->out_w_f, &e->out_b_f, &e->cs_out));
    return 0;
}

// encoder and prefill GEMMs behind a LayerNorm (e->ln_fold): row-major folded copies next to the plain ones (small batches and
// shapes outside gemm_p8_kernel's rules keep the LayerNorm launches and the plain matrices)
static int fold_encoder_prefill(gitmi_engine* e) {
    const gitmi_config& c = e->cfg;
    const int D = c.vit_width, d = c.dec_hidden, f = c.dec_ffn;
    const std::vector<float>*g, *b, *w, *bi;
    for (int i = 0; i < c.vit_layers; ++i) {
        const std::string pre = "image_encoder.transformer.resblocks." + std::to_string(i) + ".";
        VitLayerW& L = e->vit[i];
        RCK(host_vec(e, pre + "ln_1.weight", D, &g));
        RCK(host_vec(e, pre + "ln_1.bias", D, &b));
        RCK(host_vec(e, pre + "attn.in_proj_weight", (size_t)3 * D * D, &w));
        RCK(host_vec(e, pre + "attn.in_proj_bias", (size_t)3 * D, &bi));
        RCK(fold_layernorm(e, *w, *bi, *g, *b, 3 * D, D, &L.wqkv_f, &L.bqkv_f, &L.cs_qkv, false));
        RCK(host_vec(e, pre + "ln_2.weight", D, &g));
        RCK(host_vec(e, pre + "ln_2.bias", D, &b));
        RCK(host_vec(e, pre + "mlp.c_fc.weight", (size_t)4 * D * D, &w));
        RCK(host_vec(e, pre + "mlp.c_fc.bias", (size_t)4 * D, &bi));
        RCK(fold_layernorm(e, *w, *bi, *g, *b, 4 * D, D, &L.w1_f, &L.b1_f, &L.cs_1, false));
    }
    const std::string base = "textual.transformer.encoder.layer.";
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string pre = base + std::to_string(i) + ".";
        DecLayerW& L = e->dec[i];
        if (i == 0) {
            RCK(host_vec(e, "textual.visual_projection.1.weight", d, &g));
            RCK(host_vec(e, "textual.visual_projection.1.bias", d, &b));
        } else {
            const std::string prev = base + std::to_string(i - 1) + ".";
            RCK(host_vec(e, prev + "output.LayerNorm.weight", d, &g));
            RCK(host_vec(e, prev + "output.LayerNorm.bias", d, &b));
        }
        std::vector<float> wq((size_t)3 * d * d), bq((size_t)3 * d);
        const char* names[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            RCK(host_vec(e, pre + "attention.self." + names[j] + ".weight", (size_t)d * d, &w));
            RCK(host_vec(e, pre + "attention.self." + names[j] + ".bias", d, &bi));
            std::copy(w->begin(), w->end(), wq.begin() + (size_t)j * d * d);
            std::copy(bi->begin(), bi->end(), bq.begin() + (size_t)j * d);
        }
        RCK(fold_layernorm(e, wq, bq, *g, *b, 3 * d, d, &L.wqkv_pf, &L.bqkv_pf, &L.cs_qkv_p, false));
        if (i + 1 == c.dec_layers) break;            // the last layer's image rows stop at K / V
        RCK(host_vec(e, pre + "attention.output.LayerNorm.weight", d, &g));
        RCK(host_vec(e, pre + "attention.output.LayerNorm.bias", d, &b));
        RCK(host_vec(e, pre + "intermediate.dense.weight", (size_t)f * d, &w));
        RCK(host_vec(e, pre + "intermediate.dense.bias", f, &bi));
        RCK(fold_layernorm(e, *w, *bi, *g, *b, f, d, &L.w1_pf, &L.b1_pf, &L.cs_1_p, false));
    }
    return 0;
}

extern "C" int gitmi_finalize_weights(gitmi_engine* e) {
    if (!e) return fail("null engine");
    if (e->finalized) return 0;
    HIPCK(hipSetDevice(e->device));
    const gitmi_config& c = e->cfg;
    const int D = c.vit_width, F4 = 4 * D, d = c.dec_hidden, f = c.dec_ffn, V = c.vocab;
    const int64_t p = c.patch;
    // ---- image encoder -----------------------------------------------------------------
    {
        const HostTensor* t;
        RCK(get_w(e, "image_encoder.conv1.weight", {D, 3, p, p}, &t));
        e->host_w["image_encoder.conv1.weight"].shape = {D, (int64_t)e->Kp};
        RCK(up_mat(e, "image_encoder.conv1.weight", D, e->Kp, e->Kp_pad, &e->conv_w));
    }
    RCK(up_f32(e, "image_encoder.class_embedding", {D}, &e->cls));
    RCK(up_f32(e, "image_encoder.positional_embedding", {e->N_nat, D}, &e->pos));
    e->pos_cur = e->pos;
    RCK(up_f32(e, "image_encoder.ln_pre.weight", {D}, &e->lnpre_g));
    RCK(up_f32(e, "image_encoder.ln_pre.bias", {D}, &e->lnpre_b));
    RCK(up_f32(e, "image_encoder.ln_post.weight", {D}, &e->lnpost_g));
    RCK(up_f32(e, "image_encoder.ln_post.bias", {D}, &e->lnpost_b));
    e->vit.resize(c.vit_layers);
    for (int i = 0; i < c.vit_layers; ++i) {
        const std::string pre = "image_encoder.transformer.resblocks." + std::to_string(i) + ".";
        VitLayerW& L = e->vit[i];
        RCK(up_mat(e, pre + "attn.in_proj_weight", 3 * D, D, D, &L.wqkv));
        RCK(up_f32(e, pre + "attn.in_proj_bias", {3 * D}, &L.bqkv));
        RCK(up_mat(e, pre + "attn.out_proj.weight", D, D, D, &L.wo));
        RCK(up_f32(e, pre + "attn.out_proj.bias", {D}, &L.bo));
        RCK(up_f32(e, pre + "ln_1.weight", {D}, &L.ln1g));
        RCK(up_f32(e, pre + "ln_1.bias", {D}, &L.ln1b));
        RCK(up_mat(e, pre + "mlp.c_fc.weight", F4, D, D, &L.w1));
        RCK(up_f32(e, pre + "mlp.c_fc.bias", {F4}, &L.b1));
        RCK(up_mat(e, pre + "mlp.c_proj.weight", D, F4, F4, &L.w2));
        RCK(up_f32(e, pre + "mlp.c_proj.bias", {D}, &L.b2));
        RCK(up_f32(e, pre + "ln_2.weight", {D}, &L.ln2g));
        RCK(up_f32(e, pre + "ln_2.bias", {D}, &L.ln2b));
    }
    e->temb.resize(c.num_frames);
    for (int i = 0; i < c.num_frames; ++i)
        RCK(up_f32(e, "img_temperal_embedding." + std::to_string(i), {1, 1, D}, &e->temb[i]));
    // ---- text decoder ------------------------------------------------------------------
    RCK(up_mat(e, "textual.visual_projection.0.weight", d, D, D, &e->vp_w));
    RCK(up_f32(e, "textual.visual_projection.0.bias", {d}, &e->vp_b));
    RCK(up_f32(e, "textual.visual_projection.1.weight", {d}, &e->vp_lng));
    RCK(up_f32(e, "textual.visual_projection.1.bias", {d}, &e->vp_lnb));
    RCK(up_f32(e, "textual.embedding.words.weight", {V, d}, &e->words_f));
    RCK(up_f32(e, "textual.embedding.positions.weight", {c.max_pos, d}, &e->positions_f));
    RCK(up_f32(e, "textual.embedding.layer_norm.weight", {d}, &e->emb_lng));
    RCK(up_f32(e, "textual.embedding.layer_norm.bias", {d}, &e->emb_lnb));
    e->dec.resize(c.dec_layers);
    double wbytes = 0;
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string pre = "textual.transformer.encoder.layer." + std::to_string(i) + ".";
        DecLayerW& L = e->dec[i];
        // fused [Wq; Wk; Wv] so that one GEMM produces the packed q|k|v rows the attention kernels read
        RCK(dev_alloc(e, &L.wqkv, (size_t)3 * d * d * e->esz));
        RCK(dev_alloc_t(e, &L.bqkv, (size_t)3 * d));
        const char* names[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            RCK(up_mat_into(e, pre + "attention.self." + names[j] + ".weight", d, d, d, L.wqkv, (size_t)j * d));
            RCK(up_f32_into(e, pre + "attention.self." + names[j] + ".bias", d, L.bqkv, (size_t)j * d));
        }
        RCK(up_mat(e, pre + "attention.output.dense.weight", d, d, d, &L.wo));
        RCK(up_f32(e, pre + "attention.output.dense.bias", {d}, &L.bo));
        RCK(up_f32(e, pre + "attention.output.LayerNorm.weight", {d}, &L.lnag));
        RCK(up_f32(e, pre + "attention.output.LayerNorm.bias", {d}, &L.lnab));
        RCK(up_mat(e, pre + "intermediate.dense.weight", f, d, d, &L.w1));
        RCK(up_f32(e, pre + "intermediate.dense.bias", {f}, &L.b1));
        RCK(up_mat(e, pre + "output.dense.weight", d, f, f, &L.w2));
        RCK(up_f32(e, pre + "output.dense.bias", {d}, &L.b2));
        RCK(up_f32(e, pre + "output.LayerNorm.weight", {d}, &L.lnog));
        RCK(up_f32(e, pre + "output.LayerNorm.bias", {d}, &L.lnob));
        wbytes += ((double)4 * d * d + (double)2 * d * f) * e->esz;
    }
    if (e->host_w.find("textual.output.weight") == e->host_w.end())   // tied (decoder.py:503-505)
        e->host_w["textual.output.weight"] = e->host_w["textual.embedding.words.weight"];
    RCK(up_mat(e, "textual.output.weight", V, d, d, &e->out_w));
    RCK(up_f32(e, "textual.output.bias", {V}, &e->out_b));
    wbytes += (double)V * d * e->esz;
    e->dec_weight_bytes = wbytes;
    if (!e->f32 && d % 32 == 0 && f % 32 == 0 && d <= 768 && V <= 32768) RCK(fold_decoder(e));      // the bf16 decode chain (else: generic GEMM + LayerNorm launches)
    else e->skinny = false;
    if (e->ln_fold_ready) RCK(fold_encoder_prefill(e));
    e->host_w.clear();
    RCK(alloc_workspaces(e));
    HIPCK(hipDeviceSynchronize());
    e->finalized = true;
    return 0;
}

// A second execution context on the same device that BORROWS the packed weights of `src` (its own
// workspaces, KV caches, search state, streams and graph).  Lets a server keep several batches in
// flight on different HIP streams: the latency-bound decode steps of one batch overlap the
// MFMA-bound encoder of the next.  `src` must outlive the clone.
extern "C" int gitmi_clone(gitmi_engine* src, gitmi_engine** out) {
    if (!src || !out) return fail("gitmi_clone: null argument");
    if (!src->finalized) return fail("gitmi_clone: source weights not finalized");
    HIPCK(hipSetDevice(src->device));
    gitmi_engine* e = new gitmi_engine();
    e->cfg = src->cfg; e->device = src->device; e->f32 = src->f32; e->esz = src->esz; e->stream_f16 = src->stream_f16; e->ln_fold = src->ln_fold; e->ln_fold_ready = src->ln_fold_ready;
    e->attn_impl = src->attn_impl; e->Kp = src->Kp; e->Kp_pad = src->Kp_pad;
    // a clone starts at the native resolution (its own gitmi_set_image_shape state and resized table)
    e->N_nat = e->N = src->N_nat; e->g_nat = e->gh = e->gw = src->g_nat; e->H = e->W = src->cfg.image_size;
    e->Nmax = src->Nmax; e->max_pixels = src->max_pixels;
    e->use_graph = src->use_graph; e->skinny = src->skinny; e->use_temb = src->use_temb;
    e->attn_dbg = src->attn_dbg; e->dgemm_dbg = src->dgemm_dbg; e->attn_pw = src->attn_pw; e->attn_nh = src->attn_nh; e->attn_ppw = src->attn_ppw; e->attn_stream = src->attn_stream; e->decode_skip = src->decode_skip;
    e->shared_device = src->shared_device; e->gemm_tall = src->gemm_tall; e->dgemm_rows = src->dgemm_rows; e->dgemm_strips = src->dgemm_strips; e->vocab_wgs = src->vocab_wgs; e->dgemm_no_row_walk = src->dgemm_no_row_walk;
    e->parent = src->parent ? src->parent : src;
    e->conv_w = src->conv_w; e->cls = src->cls; e->pos = src->pos; e->pos_cur = src->pos;
    e->lnpre_g = src->lnpre_g; e->lnpre_b = src->lnpre_b; e->lnpost_g = src->lnpost_g; e->lnpost_b = src->lnpost_b;
    e->vit = src->vit; e->temb = src->temb;
    e->vp_w = src->vp_w; e->vp_b = src->vp_b; e->vp_lng = src->vp_lng; e->vp_lnb = src->vp_lnb;
    e->words_f = src->words_f; e->positions_f = src->positions_f; e->emb_lng = src->emb_lng; e->emb_lnb = src->emb_lnb;
    e->dec = src->dec; e->out_w = src->out_w; e->out_b = src->out_b; e->dec_weight_bytes = src->dec_weight_bytes;
    e->out_w_f = src->out_w_f; e->out_b_f = src->out_b_f; e->cs_out = src->cs_out;
    int rc = alloc_workspaces(e);
    if (rc != 0) { gitmi_destroy(e); return rc; }
    HIPCK(hipDeviceSynchronize());
    e->finalized = true;
    *out = e;
    return 0;
}
// ---- input resolution (CLIP/model.py:243-251) ---------------------------------------------
extern "C" int gitmi_set_image_shape(gitmi_engine* e, int H, int W, void* stream) {
    if (!e) return fail("null engine");
    if (!e->finalized) return fail("weights not finalized");
    const gitmi_config& c = e->cfg;
    if (H < c.patch || W < c.patch) return fail("image %dx%d is smaller than one %d-pixel patch", H, W, c.patch);
    const int gh = H / c.patch, gw = W / c.patch;
    if ((size_t)H * W > e->max_pixels)
        return fail("image %dx%d exceeds the max_image_pixels capacity (%zu)", H, W, e->max_pixels);
    if (gh * gw + 1 > e->Nmax)
        return fail("a %dx%d token grid exceeds the max_image_tokens capacity (%d)", gh, gw, e->Nmax);
    if (H == e->H && W == e->W) return 0;
    HIPCK(hipSetDevice(e->device));
    if (gh == e->g_nat && gw == e->g_nat) {
        e->pos_cur = e->pos;
    } else {
        HIPCK(launch_pos_bicubic(e->pos, e->pos_var, e->g_nat, gh, gw, c.vit_width, (hipStream_t)stream));
        e->pos_cur = e->pos_var;
    }
    e->H = H; e->W = W; e->gh = gh; e->gw = gw; e->N = gh * gw + 1;
    e->have_feats = e->have_prefill = false;
    return 0;
}

// ---------------------------------------------------------------------------------------
static int encode_frames_impl(gitmi_engine* e, const float* const* frames, int F, int B, float* feats_out,
                              hipStream_t s) {
    const gitmi_config& c = e->cfg;
    const int D = c.vit_width, N = e->N;
    const int F_eff = c.num_frames > 0 ? std::min(F, c.num_frames) : F;   // zip() truncation, decoder.py:849
    const int Nimg = F_eff * N;
    SpanGuard phase(e, s, TAG_VIT, 0);
    // all frames of the call go through the encoder as ONE batch of F*B images (the reference encodes frame by
    // frame, decoder.py:847; per-image results are identical, the GEMMs just see M = F*B*197 rows)
    const int BI = F_eff * B;                // images in the pass
    const int M = BI * N;
    const int g2 = e->gh * e->gw;
    for (int fr = 0; fr < F_eff; ++fr)
        HIPCK(launch_im2col(frames[fr], (char*)e->patches + (size_t)fr * B * g2 * e->Kp_pad * e->esz, e->f32, B,
                            e->H, e->W, c.patch, e->Kp, e->Kp_pad, s));
    RCK(gemm(e, s, e->patches, e->Kp_pad, e->conv_w, nullptr, nullptr, 0, e->patch_out, D, true, BI * g2, D, e->Kp_pad, 0,
             TAG_GEMM_VIT));
    // LayerNorm folding (fp16-operand build): ln_1 / ln_2 disappear into the GEMMs either side of them when every GEMM of the
    // pass runs on gemm_p8_kernel (more than 512 rows); the stream rows v_x are then the QKV / c_fc GEMMs' A operand as they are
    const bool fold = e->ln_fold && on_p8(e->v_x, D, e->vit[0].wqkv_f, e->v_qkv, 3 * D, M, 3 * D, D, false) &&
                      on_p8(e->v_x, D, e->vit[0].w1_f, e->v_u, 4 * D, M, 4 * D, D, false) &&
                      on_p8(e->v_ctx, D, e->vit[0].wo, e->v_x, D, M, D, D, true) && on_p8(e->v_u, 4 * D, e->vit[0].w2, e->v_x, D, M, D, 4 * D, true);
    LnRef vln;
    vln.part = e->v_part; vln.nparts = D / 256; vln.D = D; vln.eps = 1e-5f;
    HIPCK(launch_vit_assemble_ln(e->patch_out, e->cls, e->pos_cur, e->lnpre_g, e->lnpre_b, 1e-5f, e->v_x, e->stream_f16, BI, N, D,
                                 fold ? e->v_part : nullptr, D / 256, s));
    for (int l = 0; l < c.vit_layers; ++l) {
        const VitLayerW& L = e->vit[l];
        if (fold) {
            RCK(gemm_ln(e, s, e->v_x, D, L.wqkv_f, L.bqkv_f, L.cs_qkv, vln, e->v_qkv, 3 * D, M, 3 * D, D, 0, TAG_GEMM_VIT));
        } else {
            RCK(ln_stream(e, s, e->v_x, D, L.ln1g, L.ln1b, 1e-5f, e->v_h, D, nullptr, 0, M, D));
            RCK(gemm(e, s, e->v_h, D, L.wqkv, L.bqkv, nullptr, 0, e->v_qkv, 3 * D, e->f32, M, 3 * D, D, 0, TAG_GEMM_VIT));
        }
        AttnFullArgs a{};
        a.q = e->v_qkv;
        a.k = (char*)e->v_qkv + (size_t)D * e->esz;
        a.v = (char*)e->v_qkv + (size_t)2 * D * e->esz;
        a.out = e->v_ctx;
        a.ldq = a.ldk = a.ldv = 3 * D;
        a.ldo = D;
        a.N = N; a.H = c.vit_heads; a.scale = 0.125f;
        HIPCK(launch_attn_full(a, BI, e->f32, e->attn_impl, s));
        if (fold) {     // pre-norm blocks: the residual is the raw stream; every producer leaves the partials of its rows
            RCK(gemm_stream_part(e, s, e->v_ctx, D, L.wo, L.bo, e->v_x, D, nullptr, e->v_x, D, e->v_part, M, D, D, TAG_GEMM_VIT));
            RCK(gemm_ln(e, s, e->v_x, D, L.w1_f, L.b1_f, L.cs_1, vln, e->v_u, 4 * D, M, 4 * D, D, 1, TAG_GEMM_VIT));
            RCK(gemm_stream_part(e, s, e->v_u, 4 * D, L.w2, L.b2, e->v_x, D, nullptr, e->v_x, D, l + 1 < c.vit_layers ? e->v_part : nullptr,
                                 M, D, 4 * D, TAG_GEMM_VIT));
            continue;
        }
        RCK(gemm_stream(e, s, e->v_ctx, D, L.wo, L.bo, e->v_x, D, e->v_x, D, M, D, D, TAG_GEMM_VIT));
        RCK(ln_stream(e, s, e->v_x, D, L.ln2g, L.ln2b, 1e-5f, e->v_h, D, nullptr, 0, M, D));
        RCK(gemm(e, s, e->v_h, D, L.w1, L.b1, nullptr, 0, e->v_u, 4 * D, e->f32, M, 4 * D, D, 1, TAG_GEMM_VIT));
        RCK(gemm_stream(e, s, e->v_u, 4 * D, L.w2, L.b2, e->v_x, D, e->v_x, D, M, D, 4 * D, TAG_GEMM_VIT));
    }
    // ln_post (+ temporal embedding of the frame), scattered into the concatenated [B, F*N, D] feature tensor
    for (int fr = 0; fr < F_eff; ++fr) {
        const float* te = (c.num_frames > 0 && e->use_temb) ? e->temb[fr] : nullptr;
        if (e->stream_f16) {
            const char* xs = (const char*)e->v_x + (size_t)fr * B * N * D * 2;      // fp16 rows
            HIPCK(launch_layernorm_s16(xs, D, e->lnpost_g, e->lnpost_b, 1e-5f, te, e->feats, D, false, nullptr, 0, B * N, D, N,
                                       Nimg, fr * N, s));
            if (feats_out)      // parity hook: the fp32 copy of the features comes from a second pass over the same rows
                HIPCK(launch_layernorm_s16(xs, D, e->lnpost_g, e->lnpost_b, 1e-5f, te, feats_out, D, true, nullptr, 0, B * N, D,
                                           N, Nimg, fr * N, s));
        } else {
            HIPCK(launch_layernorm(e->v_x + (size_t)fr * B * N * D, D, e->lnpost_g, e->lnpost_b, 1e-5f, te, e->feats, D, e->f32,
                                   feats_out, D, B * N, D, N, Nimg, fr * N, s));
        }
    }

    e->cur_B = B; e->cur_F = F_eff; e->cur_Nimg = Nimg;
    e->have_feats = true;
    e->have_prefill = false;
    return 0;
}

// image K/V of layer l into the decode layout: head-major (fp32 VALU kernel) or the MFMA operand layouts (bf16)
static int kv_repack(gitmi_engine* e, int l, int B, int Nimg, hipStream_t s) {
    const gitmi_config& c = e->cfg;
    void* kh = e->img_kh[l];
    void* vh = e->img_vh[l];
    if (e->f32) HIPCK(launch_kv_repack(e->img_kv[l], kh, vh, B, Nimg, c.dec_heads, c.dec_hidden, true, s));
    else HIPCK(launch_kv_repack_frag(e->img_kv[l], kh, vh, B, Nimg, round_up(Nimg, 32), c.dec_heads, c.dec_hidden, s));
    return 0;
}

static int prefill_impl(gitmi_engine* e, hipStream_t s) {
    const gitmi_config& c = e->cfg;
    const int d = c.dec_hidden, ffn = c.dec_ffn, D = c.vit_width;
    const int B = e->cur_B, Nimg = e->cur_Nimg, M = B * Nimg;
    SpanGuard phase(e, s, TAG_PREFILL, 0);
    const bool fold = e->ln_fold && on_p8(e->feats, D, e->vp_w, e->p_y, d, M, d, D, true) &&
                      on_p8(e->p_y, d, e->dec[0].wqkv_pf, e->img_kv[0], 3 * d, M, 3 * d, d, false) &&
                      on_p8(e->p_y, d, e->dec[0].wqkv_pf, e->img_kv[0], 3 * d, M, 2 * d, d, false) &&
                      (c.dec_layers < 2 || (on_p8(e->p_y, d, e->dec[0].w1_pf, e->p_u, ffn, M, ffn, d, false) &&
                                            on_p8(e->p_ctx, d, e->dec[0].wo, e->p_y, d, M, d, d, true) &&
                                            on_p8(e->p_u, ffn, e->dec[0].w2, e->p_y, d, M, d, ffn, true)));
    if (fold) {
        // Post-norm layers with every LayerNorm folded: p_y holds the RAW sums (dense + residual) in place, the consumer GEMMs
        // read it as their A operand, and the residual LayerNorm(previous raw row) is rebuilt inside the next producer's
        // epilogue from the partials of the previous producer (ping-pong: tiles of one row run in different workgroups).
        int cur = 0;
        LnRef ln;                   // the LayerNorm that stands between p_y and its consumers right now
        ln.nparts = d / 256; ln.D = d;
        RCK(gemm_stream_part(e, s, e->feats, D, e->vp_w, e->vp_b, nullptr, 0, nullptr, e->p_y, d, e->p_part[cur], M, d, D, TAG_GEMM_OTHER));
        ln.part = e->p_part[cur]; ln.eps = 1e-5f; ln.gamma = e->vp_lng; ln.beta = e->vp_lnb;
        for (int l = 0; l < c.dec_layers; ++l) {
            const DecLayerW& L = e->dec[l];
            if (l + 1 == c.dec_layers) {      // only K and V of the last layer's image rows are ever read
                RCK(gemm_ln(e, s, e->p_y, d, (char*)L.wqkv_pf + (size_t)d * d * e->esz, L.bqkv_pf + d, L.cs_qkv_p + d, ln,
                            (char*)e->img_kv[l] + (size_t)d * e->esz, 3 * d, M, 2 * d, d, 0, TAG_GEMM_OTHER));
                RCK(kv_repack(e, l, B, Nimg, s));
                break;
            }
            RCK(gemm_ln(e, s, e->p_y, d, L.wqkv_pf, L.bqkv_pf, L.cs_qkv_p, ln, e->img_kv[l], 3 * d, M, 3 * d, d, 0, TAG_GEMM_OTHER));
            RCK(kv_repack(e, l, B, Nimg, s));
            AttnFullArgs a{};
            a.q = e->img_kv[l];
            a.k = (char*)e->img_kv[l] + (size_t)d * e->esz;
            a.v = (char*)e->img_kv[l] + (size_t)2 * d * e->esz;
            a.out = e->p_ctx;
            a.ldq = a.ldk = a.ldv = 3 * d;
            a.ldo = d;
            a.N = Nimg; a.H = c.dec_heads; a.scale = 0.125f;
            HIPCK(launch_attn_full(a, B, e->f32, e->attn_impl, s));
            RCK(gemm_stream_part(e, s, e->p_ctx, d, L.wo, L.bo, e->p_y, d, &ln, e->p_y, d, e->p_part[cur ^ 1], M, d, d, TAG_GEMM_OTHER));
            cur ^= 1;
            ln.part = e->p_part[cur]; ln.eps = 1e-12f; ln.gamma = L.lnag; ln.beta = L.lnab;
            RCK(gemm_ln(e, s, e->p_y, d, L.w1_pf, L.b1_pf, L.cs_1_p, ln, e->p_u, ffn, M, ffn, d, 2, TAG_GEMM_OTHER));
            RCK(gemm_stream_part(e, s, e->p_u, ffn, L.w2, L.b2, e->p_y, d, &ln, e->p_y, d, e->p_part[cur ^ 1], M, d, ffn, TAG_GEMM_OTHER));
            cur ^= 1;
            ln.part = e->p_part[cur]; ln.eps = 1e-12f; ln.gamma = L.lnog; ln.beta = L.lnob;
        }
        e->have_prefill = true;
        return 0;
    }
    RCK(gemm_stream(e, s, e->feats, D, e->vp_w, e->vp_b, nullptr, 0, e->p_y, d, M, d, D, TAG_GEMM_OTHER));
    RCK(ln_stream(e, s, e->p_y, d, e->vp_lng, e->vp_lnb, 1e-5f, e->p_ht, d, e->p_hf, d, M, d));
    for (int l = 0; l < c.dec_layers; ++l) {
        const DecLayerW& L = e->dec[l];
        const bool last = l + 1 == c.dec_layers;
        if (!last) {
            RCK(gemm(e, s, e->p_ht, d, L.wqkv, L.bqkv, nullptr, 0, e->img_kv[l], 3 * d, e->f32, M, 3 * d, d, 0, TAG_GEMM_OTHER));
            RCK(kv_repack(e, l, B, Nimg, s));
        } else {
            // the last layer's image-row outputs are never consumed: only its K and V are needed
            RCK(gemm(e, s, e->p_ht, d, (char*)L.wqkv + (size_t)d * d * e->esz, L.bqkv + d, nullptr, 0,
                     (char*)e->img_kv[l] + (size_t)d * e->esz, 3 * d, e->f32, M, 2 * d, d, 0, TAG_GEMM_OTHER));
            RCK(kv_repack(e, l, B, Nimg, s));
            break;
        }
        AttnFullArgs a{};
        a.q = e->img_kv[l];
        a.k = (char*)e->img_kv[l] + (size_t)d * e->esz;
        a.v = (char*)e->img_kv[l] + (size_t)2 * d * e->esz;
        a.out = e->p_ctx;
        a.ldq = a.ldk = a.ldv = 3 * d;
        a.ldo = d;
        a.N = Nimg; a.H = c.dec_heads; a.scale = 0.125f;
        HIPCK(launch_attn_full(a, B, e->f32, e->attn_impl, s));
        RCK(gemm_stream(e, s, e->p_ctx, d, L.wo, L.bo, e->p_hf, d, e->p_y, d, M, d, d, TAG_GEMM_OTHER));
        RCK(ln_stream(e, s, e->p_y, d, L.lnag, L.lnab, 1e-12f, e->p_ht, d, e->p_hf, d, M, d));
        RCK(gemm(e, s, e->p_ht, d, L.w1, L.b1, nullptr, 0, e->p_u, ffn, e->f32, M, ffn, d, 2, TAG_GEMM_OTHER));
        RCK(gemm_stream(e, s, e->p_u, ffn, L.w2, L.b2, e->p_hf, d, e->p_y, d, M, d, ffn, TAG_GEMM_OTHER));
        RCK(ln_stream(e, s, e->p_y, d, L.lnog, L.lnob, 1e-12f, e->p_ht, d, e->p_hf, d, M, d));
    }
    e->have_prefill = true;
    return 0;
}

// work-skipping for timing decompositions exists in measurement builds only: the product libraries cannot be told to
// return wrong answers faster
#ifdef GITMI_EXPERIMENT
#define GITMI_SKIPPED(e, bit) (((e)->decode_skip & (bit)) != 0)
#else
#define GITMI_SKIPPED(e, bit) false
#endif
// ---- decode step ---------------------------------------------------------------------------------------
// bf16: the folded-LayerNorm GEMM chain of kernels_dgemm.hip, 5 launches per layer
//   QKV (LayerNorm of the previous layer folded) -> attention -> out-proj (+ residual, strip partials)
//   -> FFN1 (attention-output LayerNorm folded, erf-GELU) -> FFN2 (+ residual, strip partials)
// and, when the step feeds the search, the vocabulary head with the running top-M / log-sum-exp fused.
// f32 (parity mode): generic GEMM + LayerNorm launches, materialised logits, row_topm.
// Input: the embedded token rows in d_hf (fp32) / d_ht (compute dtype), written by embed_ln or by the previous
// search step.  logits_out != nullptr additionally materialises the logits [R, ldl] (teacher-forced parity hook).
static int dgemm(gitmi_engine* e, hipStream_t s, const DGemmArgs& g_in) {
    DGemmArgs g = g_in;
    g.dbg = e->dgemm_dbg;
    // N = 768 GEMMs of the chain: 64 rows per workgroup (one pass over the weight strip, a quarter of the workgroups) for
    // beam batches -- faster even alone (R = 256: 0.466 -> 0.461 ms per step) -- and whenever other contexts share the
    // device: the launch is 2.8 us longer on its own but closes far fewer CUs to the encoder's GEMM workgroups
    // (profiles/r03_t_bench_lines.txt: greedy 10.34k -> 10.49k, beam-4 6.70k -> 7.00k captions/s in the mixed schedule)
    // wide GEMMs over > 64 rows (beam batches, decode groups): one workgroup per strip walks the row blocks with its weight
    // fragments in registers when other contexts share the device (beam-4: 7.18k -> 7.30k captions/s in the mixed schedule,
    // profiles/r03_zzz_ab_bench_lines.txt); alone, one workgroup per (strip, row block) is 3.5 us faster per launch
    g.no_row_walk = e->dgemm_no_row_walk >= 0 ? e->dgemm_no_row_walk : e->shared_device ? 0 : 1;
    g.strips_per_wg = e->dgemm_strips >= 0 ? e->dgemm_strips : e->shared_device ? 2 : 1;
    // rows per workgroup of the N = 768 chain GEMMs: 16 alone (48 strips x R/16 workgroups, shortest launch); next to other
    // contexts 32 -- half the workgroups for +1 us per launch.  Round 3 took 64 there (a quarter of the workgroups, +5 us:
    // +1.5 % captions/s with the kernels of the time); with the walking vocabulary head and the two-strip wide GEMMs in place
    // 32 gives the same throughput and a 7 % shorter decode step (profiles/r04_k_policy_components_bench_lines.txt:
    // 16 / 32 / 64 rows = 11.18k / 11.24k / 11.20k captions/s at 0.297 / 0.310 / 0.333 ms per step).  > 64 rows (beam
    // batches): 64, faster alone too.
    g.rows_per_wg = e->dgemm_rows > 0 ? e->dgemm_rows : g.M > 64 ? 64 : e->shared_device ? 32 : 16;
    SpanGuard sp(e, s, TAG_GEMM_OTHER, 2.0 * (double)g.M * (double)g.N * (double)g.K);
    HIPCK(launch_dgemm(g, s));
    return 0;
}

static int decode_layers_impl(gitmi_engine* e, const int* kv_src, int ld_ids, int pos, int R, int beams, hipStream_t s) {
    const gitmi_config& c = e->cfg;
    const int d = c.dec_hidden, ffn = c.dec_ffn;
    const int B = R / beams;
    const bool chain = e->skinny && !e->f32;
    const int strips = d / 16;
    const float inv_d = 1.0f / (float)d;
    for (int l = 0; l < c.dec_layers; ++l) {
        const DecLayerW& L = e->dec[l];
        const DecLayerW* Lp = l > 0 ? &e->dec[l - 1] : nullptr;
        if (chain) {
            DGemmArgs q{};
            q.A = (const unsigned short*)(l == 0 ? e->d_ht : e->xo_b); q.lda = d;
            q.W = (const unsigned short*)L.wqkv_f;
            q.bias = L.bqkv_f;
            if (l > 0) { q.colsum = L.cs_qkv; q.stats_in = e->stats_o; q.strips_in = strips; q.inv_d = inv_d; q.eps_in = 1e-12f; }
            q.C = e->d_qkv; q.ldc = 3 * d; q.act = 0; q.M = R; q.N = 3 * d; q.K = d;
            if (!GITMI_SKIPPED(e, 2)) RCK(dgemm(e, s, q));
        } else {
            RCK(gemm(e, s, e->d_ht, d, L.wqkv, L.bqkv, nullptr, 0, e->d_qkv, 3 * d, e->f32, R, 3 * d, d, 0, TAG_GEMM_OTHER));
        }
        AttnDecodeArgs a{};
        a.qkv = e->d_qkv; a.img_k = e->img_kh[l]; a.img_v = e->img_vh[l]; a.txt_k = e->txt_k[l]; a.txt_v = e->txt_v[l]; a.out = e->d_ctx;
        a.kv_src = kv_src; a.ld_src = ld_ids; a.d = d; a.N_img = e->cur_Nimg; a.T_max = c.max_text_len;
        a.img_of = e->img_identity ? nullptr : e->img_of_dev;
        a.pos = pos; a.beams = beams; a.scale = 0.125f;
        a.out_frag = chain ? 1 : 0;
        a.N_pad = round_up(e->cur_Nimg, 32);
        a.dbg = e->attn_dbg;
        a.waves_per_pair = e->attn_nh;
        a.pairs_per_wg = e->attn_pw > 0 ? e->attn_pw : e->shared_device ? 8 : 4;
        a.pairs_per_wave = e->attn_ppw > 0 ? e->attn_ppw : 1;
        // a context that has the device to itself streams the image K/V through LDS rings (kernels_attn_decode.hip: all of a
        // pair's first 36 KiB requested at once, no second memory round trip): 11.7 instead of 15.6 us per launch on 192
        // workgroups.  Same arithmetic as the one-wave register kernel, every fused multiply-add written out in both, so the
        // two agree bit for bit and gitmi_set_shared_device stays bitwise neutral.  Next to other contexts the register kernel
        // packed 8 pairs per workgroup stays: a CU streams ~24 GB/s from HBM whatever the kernel form, and fewer workgroups
        // of the streaming kernel only stretch the launch (profiles/r04_f_*)
        const bool stream_ok = !e->shared_device && a.N_pad <= 8 * 32 && B * c.dec_heads >= 384 && e->attn_nh != 2;
        a.stream_wgs = e->attn_stream >= 0 ? e->attn_stream : stream_ok ? 192 : 0;
        if (e->f32) HIPCK(launch_attn_decode(a, B, c.dec_heads, true, s));
        else if (!GITMI_SKIPPED(e, 1)) HIPCK(launch_attn_decode_mfma(a, B, c.dec_heads, s));
        if (chain) {
            DGemmArgs o{};
            o.A = (const unsigned short*)e->d_ctx; o.lda = d; o.W = (const unsigned short*)L.wo_p; o.bias = L.bo;
            o.res_x = l == 0 ? e->d_hf : e->xo_f;
            if (l > 0) { o.res_stats = e->stats_o; o.res_strips = strips; o.res_gamma = Lp->lnog; o.res_beta = Lp->lnob; o.res_inv_d = inv_d; o.res_eps = 1e-12f; }
            o.x_out = e->xa_f; o.xb_out = (unsigned short*)e->xa_b; o.stats_out = e->stats_a;
            o.M = R; o.N = d; o.K = d;
            if (!GITMI_SKIPPED(e, 4)) RCK(dgemm(e, s, o));
            DGemmArgs f1{};
            f1.A = (const unsigned short*)e->xa_b; f1.lda = d; f1.W = (const unsigned short*)L.w1_f; f1.bias = L.b1_f; f1.colsum = L.cs_1;
            f1.stats_in = e->stats_a; f1.strips_in = strips; f1.inv_d = inv_d; f1.eps_in = 1e-12f;
            f1.C = e->d_u; f1.ldc = ffn; f1.c_frag = 1; f1.act = 2; f1.M = R; f1.N = ffn; f1.K = d;
            if (!GITMI_SKIPPED(e, 2)) RCK(dgemm(e, s, f1));
            DGemmArgs f2{};
            f2.A = (const unsigned short*)e->d_u; f2.lda = ffn; f2.W = (const unsigned short*)L.w2_p; f2.bias = L.b2;
            f2.res_x = e->xa_f; f2.res_stats = e->stats_a; f2.res_strips = strips; f2.res_gamma = L.lnag; f2.res_beta = L.lnab;
            f2.res_inv_d = inv_d; f2.res_eps = 1e-12f;
            f2.x_out = e->xo_f; f2.xb_out = (unsigned short*)e->xo_b; f2.stats_out = e->stats_o;
            f2.M = R; f2.N = d; f2.K = ffn;
            if (!GITMI_SKIPPED(e, 4)) RCK(dgemm(e, s, f2));
        } else {
            RCK(gemm(e, s, e->d_ctx, d, L.wo, L.bo, e->d_hf, d, e->d_y, d, true, R, d, d, 0, TAG_GEMM_OTHER));
            HIPCK(launch_layernorm(e->d_y, d, L.lnag, L.lnab, 1e-12f, nullptr, e->d_ht, d, e->f32, e->d_hf, d, R, d, 0, 0, 0, s));
            RCK(gemm(e, s, e->d_ht, d, L.w1, L.b1, nullptr, 0, e->d_u, ffn, e->f32, R, ffn, d, 2, TAG_GEMM_OTHER));
            RCK(gemm(e, s, e->d_u, ffn, L.w2, L.b2, e->d_hf, d, e->d_y, d, true, R, d, ffn, 0, TAG_GEMM_OTHER));
            HIPCK(launch_layernorm(e->d_y, d, L.lnog, L.lnob, 1e-12f, nullptr, e->d_ht, d, e->f32, e->d_hf, d, R, d, 0, 0, 0, s));
        }
    }
    return 0;
}

static int sample_candidates(gitmi_engine* e, const float* logits, int ldl, int R, int step, hipStream_t s, StepCands* cands);
static int trie_candidates(gitmi_engine* e, const float* logits, int ldl, int cur_len, hipStream_t s, StepCands* cands);
// repetition penalty of the current search (GENERATOR only; 0 and 1 both mean "off")
static float rep_penalty_of(const gitmi_engine* e) {
    const double rp = e->sample.repetition_penalty;
    return (e->ss.kind == GITMI_SEARCH_GENERATOR && rp > 0 && rp != 1.0) ? (float)rp : 0.f;
}

// vocabulary head of the step: candidate lists for the search (and optionally the logits themselves)
static int decode_head_impl(gitmi_engine* e, const int* ids, int ld_ids, int cur_len, int R, int beams, int suppress_kind,
                            int M, float* logits_out, int ldl, hipStream_t s, StepCands* cands) {
    const gitmi_config& c = e->cfg;
    const int d = c.dec_hidden;
    const bool chain = e->skinny && !e->f32;
    cands->part_val = e->part_val; cands->part_idx = e->part_idx; cands->part_lse = e->part_lse;
    const bool sampling = ids != nullptr && e->ss.sampled;
    const bool trie = ids != nullptr && e->trie_search;
    if ((sampling || trie) && !logits_out) { logits_out = e->logits; ldl = e->ldl; }     // the filter / the trie need the whole row
    if (chain) {
        const DecLayerW& L = e->dec[c.dec_layers - 1];
        (void)L;
        VocabArgs v{};
        v.A = (const unsigned short*)e->xo_b; v.lda = d; v.W = (const unsigned short*)e->out_w_f; v.bias = e->out_b_f; v.colsum = e->cs_out;
        v.stats_in = e->stats_o; v.strips_in = d / 16; v.inv_d = 1.0f / (float)d; v.eps_in = 1e-12f;
        v.M = R; v.N = c.vocab; v.K = d; v.cols_per_wg = e->vocab_cols;
        // workgroups of the head: by default one per column block (one HBM round trip; fastest alone); when other contexts
        // share the device, ~60 workgroups that each WALK four column blocks with a rolling weight refill -- the launch is
        // bound by the chip's HBM rate either way, and every CU that holds one of its waves is closed to the image encoder's
        // GEMM workgroups for the whole launch
        // (beam batches, R > 64 rows: one column block per workgroup in either policy -- the walking form re-reads the rows of
        // four row blocks per column block and takes 186 instead of 56 us, and the mix measures the same captions/s with either:
        // profiles/r05_m_beam_head_wgs_ab_bench_lines.txt; the shorter launch takes 0.17 ms off every beam step)
        v.max_wgs = e->vocab_wgs >= 0 ? e->vocab_wgs : (e->shared_device && R <= 64) ? 60 : 0;
        v.ids = ids; v.ld_ids = ld_ids; v.cur_len = cur_len; v.plen = e->plen_dev; v.beams = beams; v.suppress_kind = suppress_kind;
        v.rep_penalty = ids ? rep_penalty_of(e) : 0.f;
        v.part_val = e->part_val; v.part_idx = e->part_idx; v.part_lse = e->part_lse;
        v.logits_out = logits_out; v.ld_logits = ldl;
        {
            SpanGuard sp(e, s, TAG_GEMM_OTHER, 2.0 * (double)R * (double)c.vocab * (double)d);
            if (!GITMI_SKIPPED(e, 8)) HIPCK(launch_vocab_topm(v, M, s));
        }
        cands->nparts = e->vocab_nparts; cands->slots = vocab_mtop_slots(M);
        if (sampling) RCK(sample_candidates(e, logits_out, ldl, R, cur_len, s, cands));
        if (trie) RCK(trie_candidates(e, logits_out, ldl, cur_len, s, cands));
    } else {
        RCK(gemm(e, s, e->d_ht, d, e->out_w, e->out_b, nullptr, 0, e->logits, e->ldl, true, R, c.vocab, d, 0, TAG_GEMM_OTHER));
        cands->nparts = 1; cands->slots = row_topm_slots(M);
        if (sampling) RCK(sample_candidates(e, e->logits, e->ldl, R, cur_len, s, cands));
        else if (trie) RCK(trie_candidates(e, e->logits, e->ldl, cur_len, s, cands));
        else if (ids)
            HIPCK(launch_row_topm(e->logits, e->ldl, c.vocab, ids, ld_ids, cur_len, e->plen_dev, beams, suppress_kind,
                                  rep_penalty_of(e), M, R,
                                  e->part_val, e->part_idx, e->part_lse, s));
        if (logits_out && logits_out != e->logits) HIPCK(launch_copy_f32(e->logits, e->ldl, logits_out, ldl, R, c.vocab, s));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
static int check_ready(gitmi_engine* e) {
    if (!e) return fail("null engine");
    if (!e->finalized) return fail("weights not finalized (call gitmi_finalize_weights)");
    HIPCK(hipSetDevice(e->device));
    return 0;
}

extern "C" int gitmi_encode_frames(gitmi_engine* e, const float* const* frames, int F, int B, float* feats_out,
                                   void* stream) {
    RCK(check_ready(e));
    if (!frames || F < 1 || F > e->cfg.max_frames) return fail("encode_frames: F=%d outside [1,%d]", F, e->cfg.max_frames);
    if (B < 1 || B > e->cfg.max_batch) return fail("encode_frames: B=%d outside [1,%d]", B, e->cfg.max_batch);
    return encode_frames_impl(e, frames, F, B, feats_out, (hipStream_t)stream);
}

extern "C" int gitmi_prefill(gitmi_engine* e, void* stream) {
    RCK(check_ready(e));
    if (!e->have_feats) return fail("prefill: no encoded frames");
    return prefill_impl(e, (hipStream_t)stream);
}

extern "C" int gitmi_step_logits(gitmi_engine* e, const int64_t* tokens, int R, int t, float* logits_out, void* stream) {
    RCK(check_ready(e));
    if (!e->have_feats) return fail("step_logits: no encoded frames");
    hipStream_t s = (hipStream_t)stream;
    if (!e->have_prefill) RCK(prefill_impl(e, s));
    const int B = e->cur_B;
    if (R < B || R % B) return fail("step_logits: R=%d is not a multiple of the encoded batch %d", R, B);
    const int beams = R / B;
    if (beams > e->cfg.max_beams) return fail("step_logits: %d beams exceed max_beams", beams);
    if (t < 1 || t > e->cfg.max_text_len) return fail("step_logits: t=%d outside [1,%d]", t, e->cfg.max_text_len);
    const gitmi_config& c = e->cfg;
    const int T = c.max_text_len;
    e->img_identity = true;
    HIPCK(launch_load_ids((const long long*)tokens, R, t, e->ss.ids[0], e->ss.kv_src[0], T, s));
    // note: load_ids writes rows of length ld = T_max
    StepCands cands{};
    for (int pos = 0; pos < t; ++pos) {
        SpanGuard step(e, s, TAG_STEP, 0);
        HIPCK(launch_embed_ln(e->ss.ids[0], T, pos, e->words_f, e->positions_f, e->emb_lng, e->emb_lnb, 1e-8f, e->d_hf,
                              e->d_ht, e->f32, R, c.dec_hidden, c.vocab, e->skinny && !e->f32, s));
        RCK(decode_layers_impl(e, e->ss.kv_src[0], T, pos, R, beams, s));
        if (pos == t - 1) RCK(decode_head_impl(e, nullptr, T, t, R, beams, 0, 1, logits_out, c.vocab, s, &cands));
    }
    return 0;
}

// ---- search seam -----------------------------------------------------------------------
// `start_dev` / `plen_dev` (/ `img_of_dev`) must already describe the B sentences of the call
static int search_begin_impl(gitmi_engine* e, const gitmi_search* sp, int B, int minP, int maxP, int V, bool ragged,
                             hipStream_t s) {
    const gitmi_config& c = e->cfg;
    if (!sp) return fail("search: null config");
    if (sp->kind != GITMI_SEARCH_AUTOREGRESSIVE && sp->kind != GITMI_SEARCH_GENERATOR && sp->kind != GITMI_SEARCH_TRIE)
        return fail("search: bad kind");
    if (sp->kind == GITMI_SEARCH_TRIE) {
        if (sp->beam_size != 1) return fail("search: TrieAutoRegressiveBeamSearch asserts beam_size == 1 (trie_decoder.py:37)");
        if (!e->trie_off) return fail("search: no trie loaded (gitmi_set_trie)");
        if (sp->do_sample) return fail("search: the trie search has no sampling branch");
    }
    if (B < 1 || B > c.max_batch) return fail("search: B=%d outside [1,%d]", B, c.max_batch);
    if (sp->beam_size < 1 || sp->beam_size > c.max_beams) return fail("search: beam_size %d outside [1,%d]", sp->beam_size, c.max_beams);
    if (sp->per_node_beam_size < 1 && sp->kind != GITMI_SEARCH_TRIE) return fail("search: per_node_beam_size must be >= 1");
    if (sp->kind == GITMI_SEARCH_GENERATOR && sp->per_node_beam_size < 2)
        return fail("search: GeneratorWithBeamSearch requires per_node_beam_size > 1 (decoder.py:1078)");
    if (sp->beam_size * sp->per_node_beam_size > 16) return fail("search: beam_size*per_node_beam_size > 16 unsupported");
    if (sp->max_steps > c.max_text_len) return fail("search: max_steps %d exceeds max_text_len %d", sp->max_steps, c.max_text_len);
    if (minP < 1 || maxP > sp->max_steps) return fail("search: prefix lengths [%d,%d] outside [1,max_steps]", minP, maxP);
    if (sp->kind == GITMI_SEARCH_GENERATOR && !(sp->length_penalty > 0)) return fail("search: length_penalty must be > 0");
    if (sp->repetition_penalty != 0 && sp->repetition_penalty != 1.0) {
        if (sp->kind != GITMI_SEARCH_GENERATOR) return fail("search: repetition_penalty belongs to GeneratorWithBeamSearch (decoder.py:1064)");
        if (!(sp->repetition_penalty >= 1.0)) return fail("search: `repetition_penalty` should be >= 1 (decoder.py:1080)");
        if (sp->max_steps > 1024) return fail("search: repetition_penalty supports histories up to 1024 tokens");
    }
    if (sp->num_keep_best < 0 || sp->num_keep_best > SS_NHMAX) return fail("search: num_keep_best %d outside [1,%d]", sp->num_keep_best, SS_NHMAX);
    if (sp->num_keep_best > 1 && sp->kind != GITMI_SEARCH_GENERATOR)
        return fail("search: num_keep_best belongs to GeneratorWithBeamSearch.search (decoder.py:1087)");
    if (sp->do_sample) {
        if (sp->kind != GITMI_SEARCH_GENERATOR) return fail("search: do_sample is implemented for GeneratorWithBeamSearch (decoder.py:1146-1166) only");
        if (sp->temperature < 0) return fail("search: temperature must be > 0");
        if (V > 32768) return fail("search: sampling supports vocabularies up to 32768 tokens");
    }
    SearchState& st = e->ss;
    e->trie_search = sp->kind == GITMI_SEARCH_TRIE;
    st.B = B; st.k = sp->beam_size; st.pn = e->trie_search ? 1 : sp->per_node_beam_size;
    st.T = sp->max_steps;           // max_length of the search AND the row stride of ids/kv_src/hyp_tok
    // the trie search shares AutoRegressiveBeamSearch's bookkeeping (beam 1): only the candidate selection differs
    st.V = V; st.eos = c.eos; st.kind = e->trie_search ? GITMI_SEARCH_AUTOREGRESSIVE : sp->kind; st.length_penalty = sp->length_penalty;
    st.ragged = ragged ? 1 : 0;
    st.nh = sp->num_keep_best > 1 ? sp->num_keep_best : 1;
    st.sampled = sp->do_sample ? 1 : 0;
    e->sample = *sp;
    st.start = e->start_dev; st.ld_start = c.max_text_len; st.plen = e->plen_dev;
    e->ss_cur = 0; e->ss_len = minP; e->ss_minP = minP;
    HIPCK(launch_search_init(st, s));
    if (e->trie_search) HIPCK(launch_fill_i32(e->trie_cursor, 0, B, s));      // TokenTrie.reset(): every cursor at the root
    return 0;
}

static int search_mtop(const SearchState& st) {
    if (st.sampled) return st.pn;
    return st.kind == GITMI_SEARCH_AUTOREGRESSIVE ? std::max(st.k, st.pn) : st.pn * st.k;
}
// sampling branch: filter + draws from the materialised logits of the step (decoder.py:1146-1166)
static int sample_candidates(gitmi_engine* e, const float* logits, int ldl, int R, int step, hipStream_t s, StepCands* cands) {
    const gitmi_search& sp = e->sample;
    const float temp = sp.temperature > 0 ? (float)sp.temperature : 1.0f;
    HIPCK(launch_sample_rows(logits, ldl, e->ss.V, R, temp, sp.top_k, (float)sp.top_p, e->ss.pn, sp.seed, step,
                             e->part_val, e->part_idx, e->part_lse, nullptr, e->ss.ids[e->ss_cur], e->ss.T, step,
                             rep_penalty_of(e), row_topm_slots(e->ss.pn), s));
    cands->part_val = e->part_val; cands->part_idx = e->part_idx; cands->part_lse = e->part_lse;
    cands->nparts = 1; cands->slots = row_topm_slots(e->ss.pn);
    return 0;
}

// trie-constrained selection on the materialised logits of the step (trie_decoder.py:57-71, 115-158)
static int trie_candidates(gitmi_engine* e, const float* logits, int ldl, int cur_len, hipStream_t s, StepCands* cands) {
    const SearchState& st = e->ss;
    TrieArgs tr{e->trie_off, e->trie_tok, e->trie_child, e->trie_cursor};
    HIPCK(launch_trie_select(logits, ldl, st.V, st.ids[e->ss_cur], st.T, cur_len, e->plen_dev, st.eos, tr, st.B, e->part_val,
                             e->part_idx, e->part_lse, s));
    cands->part_val = e->part_val; cands->part_idx = e->part_idx; cands->part_lse = e->part_lse;
    cands->nparts = 1; cands->slots = 1;
    return 0;
}

static EmbedArgs embed_args(gitmi_engine* e, bool on) {
    EmbedArgs em{};
    if (!on) return em;
    em.words = e->words_f; em.positions = e->positions_f; em.gamma = e->emb_lng; em.beta = e->emb_lnb; em.eps = 1e-8f;
    em.h_f = e->d_hf; em.h_t = e->d_ht; em.D = e->cfg.dec_hidden; em.vocab = e->cfg.vocab;
    em.frag = (e->skinny && !e->f32) ? 1 : 0;
    return em;
}

// one search step on the candidate lists of the current step (+ the embedding of the appended tokens)
static int search_step_impl(gitmi_engine* e, const StepCands& cands, bool embed, hipStream_t s) {
    const SearchState& st = e->ss;
    const int cur_len = e->ss_len;
    if (cur_len >= st.T) return fail("search_advance: sequence already at max_steps");
    HIPCK(launch_search_step(st, e->ss_cur, cur_len, cands, embed_args(e, embed), e->f32, s));
    e->ss_cur ^= 1;
    e->ss_len = cur_len + 1;
    return 0;
}

static int fill_uniform_sentences(gitmi_engine* e, int B, const long long* prefix_dev, int P, hipStream_t s) {
    HIPCK(launch_fill_start(e->start_dev, e->cfg.max_text_len, prefix_dev, 0, 1, e->cfg.sos, B, P, s));
    HIPCK(launch_fill_i32(e->plen_dev, P, B, s));
    e->img_identity = true;
    return 0;
}

// Token trie for GITMI_SEARCH_TRIE (trie_decoder.py:224-257 TokenTrie as CSR): node 0 is the root, the children of node n
// are the edges child_off[n] .. child_off[n + 1] - 1 (token child_tok[e] leads to node child_node[e]).  Host arrays, copied.
// n_nodes == 0 removes the trie.
extern "C" int gitmi_set_trie(gitmi_engine* e, int n_nodes, const int32_t* child_off, const int32_t* child_tok,
                              const int32_t* child_node) {
    RCK(check_ready(e));
    HIPCK(hipDeviceSynchronize());
    destroy_graph(e);                                       // captured launches hold the old pointers
    // the new arrays are built completely before the old ones go: a failed allocation or copy leaves the engine with the
    // trie it had (all three arrays or none), never with offsets that point into freed edge arrays
    int *n_off = nullptr, *n_tok = nullptr, *n_child = nullptr;
    if (n_nodes > 0) {
        if (!child_off || child_off[0] != 0) return fail("set_trie: child_off must start at 0");
        const int n_edges = child_off[n_nodes];
        for (int n = 0; n < n_nodes; ++n)
            if (child_off[n + 1] < child_off[n]) return fail("set_trie: child_off must be non-decreasing");
        if (n_edges > 0 && (!child_tok || !child_node)) return fail("set_trie: null edge arrays");
        for (int i = 0; i < n_edges; ++i)
            if (child_node[i] < 0 || child_node[i] >= n_nodes) return fail("set_trie: edge %d leads to node %d of %d", i, child_node[i], n_nodes);
        hipError_t err = hipMalloc((void**)&n_off, (size_t)(n_nodes + 1) * sizeof(int));
        if (err == hipSuccess) err = hipMalloc((void**)&n_tok, (size_t)std::max(n_edges, 1) * sizeof(int));
        if (err == hipSuccess) err = hipMalloc((void**)&n_child, (size_t)std::max(n_edges, 1) * sizeof(int));
        if (err == hipSuccess) err = hipMemcpy(n_off, child_off, (size_t)(n_nodes + 1) * sizeof(int), hipMemcpyHostToDevice);
        if (err == hipSuccess && n_edges > 0) err = hipMemcpy(n_tok, child_tok, (size_t)n_edges * sizeof(int), hipMemcpyHostToDevice);
        if (err == hipSuccess && n_edges > 0) err = hipMemcpy(n_child, child_node, (size_t)n_edges * sizeof(int), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            if (n_off) hipFree(n_off);
            if (n_tok) hipFree(n_tok);
            if (n_child) hipFree(n_child);
            return fail("set_trie: %s (the previous trie is kept)", hipGetErrorString(err));
        }
    }
    if (e->trie_off) { hipFree(e->trie_off); hipFree(e->trie_tok); hipFree(e->trie_child); }
    e->trie_off = n_off; e->trie_tok = n_tok; e->trie_child = n_child;
    e->trie_nodes = n_nodes > 0 ? n_nodes : 0;
    return 0;
}

extern "C" int gitmi_search_begin(gitmi_engine* e, const gitmi_search* sp, int B, const int64_t* start_host, int P,
                                  int vocab, void* stream) {
    RCK(check_ready(e));
    if (!start_host) return fail("search_begin: null start");
    if (B < 1 || B > e->cfg.max_batch || P < 1 || P > e->cfg.max_text_len) return fail("search_begin: bad B/P");
    if (vocab < 2) return fail("search_begin: bad vocab");
    hipStream_t s = (hipStream_t)stream;
    // start_host is [B, P]: one row per sentence, written into the engine's [B, max_text_len] start table
    HIPCK(hipMemcpy2DAsync(e->start_dev, (size_t)e->cfg.max_text_len * sizeof(long long), start_host,
                           (size_t)P * sizeof(long long), (size_t)P * sizeof(long long), (size_t)B, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));
    HIPCK(launch_fill_i32(e->plen_dev, P, B, s));
    e->img_identity = true;
    return search_begin_impl(e, sp, B, P, P, vocab, false, s);
}

extern "C" int gitmi_search_rows(gitmi_engine* e, int64_t* tokens_out, int* R, int* t, void* stream) {
    RCK(check_ready(e));
    const SearchState& st = e->ss;
    if (R) *R = st.B * st.k;
    if (t) *t = e->ss_len;
    if (tokens_out) HIPCK(launch_search_rows(st, e->ss_cur, e->ss_len, (long long*)tokens_out, (hipStream_t)stream));
    return 0;
}

extern "C" int gitmi_search_advance(gitmi_engine* e, const float* logits, void* stream) {
    RCK(check_ready(e));
    if (!logits) return fail("search_advance: null logits");
    const SearchState& st = e->ss;
    hipStream_t s = (hipStream_t)stream;
    const int M = search_mtop(st), R = st.B * st.k;
    StepCands cands{e->part_val, e->part_idx, e->part_lse, 1, row_topm_slots(M)};
    if (st.sampled) RCK(sample_candidates(e, logits, st.V, R, e->ss_len, s, &cands));
    else if (e->trie_search) RCK(trie_candidates(e, logits, st.V, e->ss_len, s, &cands));
    else
        HIPCK(launch_row_topm(logits, st.V, st.V, st.ids[e->ss_cur], st.T, e->ss_len, e->plen_dev, st.k,
                              st.kind == GITMI_SEARCH_AUTOREGRESSIVE ? 1 : 0, rep_penalty_of(e), M, R, e->part_val,
                              e->part_idx, e->part_lse, s));
    return search_step_impl(e, cands, false, s);
}

extern "C" int gitmi_search_done_count(gitmi_engine* e, int* done_out, void* stream) {
    RCK(check_ready(e));
    if (!done_out) return fail("search_done_count: null argument");
    int h = 0;
    HIPCK(hipMemcpyAsync(&h, e->ss.info, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCK(hipStreamSynchronize((hipStream_t)stream));
    *done_out = h;
    return 0;
}

extern "C" int gitmi_search_finish(gitmi_engine* e, int64_t* tokens_out, float* logprob_out, int32_t* info_out,
                                   void* stream) {
    RCK(check_ready(e));
    const SearchState& st = e->ss;
    HIPCK(launch_search_finish(st, e->ss_cur, e->ss_len, (long long*)tokens_out, logprob_out, info_out, nullptr,
                               (hipStream_t)stream));
    return 0;
}

// ---- the whole hot path ------------------------------------------------------------------
// image encoder + decoder prefill over the image tokens
static int generate_encode(gitmi_engine* e, const float* const* frames, int F, int B, hipStream_t s) {
    RCK(encode_frames_impl(e, frames, F, B, nullptr, s));
    RCK(prefill_impl(e, s));
    return 0;
}

// search over the text positions (teacher-forced prefix positions, then searched ones) + result formatting.
// Q sentences (start_dev / plen_dev / img_of_dev describe them), prefix lengths in [minP, maxP].
static int generate_decode(gitmi_engine* e, int Q, int minP, int maxP, bool ragged, const gitmi_search* sp,
                           long long* tokens_out, float* logprob_out, int32_t* info_out, int32_t* sent_out, hipStream_t s,
                           bool allow_poll) {
    const gitmi_config& c = e->cfg;
    RCK(search_begin_impl(e, sp, Q, minP, maxP, c.vocab, ragged, s));
    const int T = sp->max_steps;
    const SearchState& st = e->ss;
    const int k = sp->beam_size, R = Q * k;
    const int M = search_mtop(st);
    const int suppress = sp->kind != GITMI_SEARCH_GENERATOR ? 1 : 0;
    {
        SpanGuard phase(e, s, TAG_DECODE, 0);
        // position 0 is embedded here; every later position by the search step that appends its token
        HIPCK(launch_embed_ln(st.ids[0], T, 0, e->words_f, e->positions_f, e->emb_lng, e->emb_lnb, 1e-8f, e->d_hf, e->d_ht,
                              e->f32, R, c.dec_hidden, c.vocab, e->skinny && !e->f32, s));
        e->ss_len = 1;
        StepCands cands{e->part_val, e->part_idx, e->part_lse, 1, 1};
        while (e->ss_len < T) {
            const int cur_len = e->ss_len;
            SpanGuard step(e, s, TAG_STEP, 0);
            RCK(decode_layers_impl(e, st.kv_src[e->ss_cur], T, cur_len - 1, R, k, s));
            // steps that only append given prefix tokens to every sentence (VQA question tokens) need no logits
            if (cur_len >= minP)
                RCK(decode_head_impl(e, st.ids[e->ss_cur], T, cur_len, R, k, suppress, M, nullptr, 0, s, &cands));
            RCK(search_step_impl(e, cands, true, s));
            // long step budgets (the shipped default is max_steps=1024, model.py:37): every 8 steps read
            // the device-side count of finished sentences so the loop ends like decoder.py:319 / :1251.
            // Extra steps past that point are idempotent, so polling sparsely is exact.
            if (allow_poll && e->ss_len > maxP && (e->ss_len - maxP) % 8 == 0 && e->ss_len < T) {
                int h[4];
                HIPCK(hipMemcpyAsync(h, st.info, sizeof(h), hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                if (h[0] >= Q) break;
            }
        }
    }
    HIPCK(launch_search_finish(st, e->ss_cur, e->ss_len, tokens_out, logprob_out, info_out, sent_out, s));
    // algorithmic bytes of one decode step: all decoder weights once +
    // per sentence the K/V of every layer (image part shared by beams, text part per beam)
    const double kv = (double)Q * c.dec_layers * 2.0 * ((double)e->cur_Nimg + k * 0.5 * (minP + T)) * c.dec_hidden * e->esz;
    e->last_decode_step_bytes = e->dec_weight_bytes + kv;
    return 0;
}

static int generate_body(gitmi_engine* e, const float* const* frames, int F, int B, int Q, int minP, int maxP, bool ragged,
                         const gitmi_search* sp, long long* tokens_out, float* logprob_out, int32_t* info_out,
                         int32_t* sent_out, hipStream_t s, bool allow_poll) {
    SpanGuard total(e, s, 99, 0);
    RCK(generate_encode(e, frames, F, B, s));
    return generate_decode(e, Q, minP, maxP, ragged, sp, tokens_out, logprob_out, info_out, sent_out, s, allow_poll);
}

// common tail of gitmi_generate / gitmi_generate_prefixed: start_dev / plen_dev / img_of_dev are already enqueued on `s`
static int generate_run(gitmi_engine* e, const float* const* frames, int F, int B, int Q, int minP, int maxP, bool ragged,
                        const gitmi_search* sp, int64_t* tokens_out, float* logprob_out, int32_t* info_out,
                        int32_t* sent_out, hipStream_t s) {
    const gitmi_config& c = e->cfg;
    const bool long_budget = sp->max_steps - minP > 32;
    const bool graph = e->use_graph && !e->profiling && !long_budget;
    if (!graph)
        return generate_body(e, frames, F, B, Q, minP, maxP, ragged, sp, (long long*)tokens_out, logprob_out, info_out,
                             sent_out ? sent_out : e->out_sent, s, long_budget && !e->profiling);

    // ---- hipGraph path: the launch sequence only depends on (B,Q,F,minP,search); inputs and outputs are
    // staged through engine-owned buffers so the captured pointers stay valid across calls.
    // The legacy null stream cannot be captured: run on the engine's own stream, fenced by events.
    hipStream_t x = s ? s : e->own_stream;
    if (x != s) {
        HIPCK(hipEventRecord(e->fence_in, s));
        HIPCK(hipStreamWaitEvent(x, e->fence_in, 0));
    }
    const size_t frame_bytes = (size_t)B * 3 * e->H * e->W * sizeof(float);
    const int F_eff = c.num_frames > 0 ? std::min(F, c.num_frames) : F;
    for (int f = 0; f < F_eff; ++f)
        HIPCK(hipMemcpyAsync(e->frame_stage[f], frames[f], frame_bytes, hipMemcpyDeviceToDevice, x));
    gitmi_engine::GraphKey key{};
    key.B = B; key.Q = Q; key.F = F_eff; key.P = minP; key.kind = sp->kind; key.k = sp->beam_size; key.pn = sp->per_node_beam_size;
    key.T = sp->max_steps; key.H = e->H; key.W = e->W; key.lp = sp->length_penalty;
    key.ragged = ragged ? 1 : 0; key.ident = e->img_identity ? 1 : 0; key.temb = e->use_temb ? 1 : 0;
    key.smp = sp->do_sample; key.top_k = sp->top_k; key.top_p = sp->top_p; key.temp = sp->temperature; key.seed = sp->seed;
    key.rp = sp->repetition_penalty; key.nh = sp->num_keep_best > 1 ? sp->num_keep_best : 1;
    // two graphs (encode + prefill | decode) whenever something has to happen between them: profiling events or the
    // enc_done record other contexts wait for
    const bool split = e->profile_mode == 2 || e->enc_after != nullptr || !e->enc_watchers.empty();
    if (!e->graph_valid || !(key == e->graph_key) || split != e->graph_is_split) {
        destroy_graph(e);
        std::vector<const float*> fp(F_eff);
        for (int f = 0; f < F_eff; ++f) fp[f] = e->frame_stage[f];
        auto capture = [&](int part, hipGraph_t* gr_out) -> int {       // part 0: whole call, 1: encode + prefill, 2: decode
            HIPCK(hipStreamBeginCapture(x, hipStreamCaptureModeThreadLocal));
            int rc = 0;
            if (part == 0) rc = generate_body(e, fp.data(), F_eff, B, Q, minP, maxP, ragged, sp, e->out_tokens, e->out_lp,
                                              e->out_info, e->out_sent, x, false);
            else if (part == 1) rc = generate_encode(e, fp.data(), F_eff, B, x);
            else rc = generate_decode(e, Q, minP, maxP, ragged, sp, e->out_tokens, e->out_lp, e->out_info, e->out_sent, x, false);
            hipGraph_t gr = nullptr;
            hipError_t ce = hipStreamEndCapture(x, &gr);
            if (rc != 0) { if (gr) hipGraphDestroy(gr); return rc; }
            HIPCK(ce);
            *gr_out = gr;
            return 0;
        };
        if (!split) {
            RCK(capture(0, &e->graph));
            HIPCK(hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
        } else {
            RCK(capture(1, &e->graph));
            HIPCK(hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
            RCK(capture(2, &e->graph_b));
            HIPCK(hipGraphInstantiate(&e->graph_exec_b, e->graph_b, nullptr, nullptr, 0));
        }
        e->graph_key = key;
        e->graph_valid = true;
        e->graph_is_split = split;
    } else {
        // host-side mirror of the state generate_body leaves behind
        e->cur_B = B; e->cur_F = F_eff; e->cur_Nimg = F_eff * e->N;
        e->have_feats = e->have_prefill = true;
    }
    if (!split) {
        HIPCK(hipGraphLaunch(e->graph_exec, x));
    } else if (e->profile_mode != 2) {
        if (e->enc_after && e->enc_after->enc_done) HIPCK(hipStreamWaitEvent(x, e->enc_after->enc_done, 0));
        HIPCK(hipGraphLaunch(e->graph_exec, x));
        if (e->enc_done && !e->enc_watchers.empty()) HIPCK(hipEventRecord(e->enc_done, x));
        HIPCK(hipGraphLaunch(e->graph_exec_b, x));
    } else {
        for (auto& ev : e->gev)
            if (!ev) HIPCK(hipEventCreate(&ev));
        HIPCK(hipEventRecord(e->gev[0], x));
        HIPCK(hipGraphLaunch(e->graph_exec, x));
        HIPCK(hipEventRecord(e->gev[1], x));
        HIPCK(hipGraphLaunch(e->graph_exec_b, x));
        HIPCK(hipEventRecord(e->gev[2], x));
        HIPCK(hipStreamSynchronize(x));
        float a = 0, b = 0;
        HIPCK(hipEventElapsedTime(&a, e->gev[0], e->gev[1]));
        HIPCK(hipEventElapsedTime(&b, e->gev[1], e->gev[2]));
        e->split_encode_ms += a; e->split_decode_ms += b; e->split_calls += 1; e->split_steps += sp->max_steps - 1;
    }
    const size_t nout = (size_t)Q * (size_t)(sp->num_keep_best > 1 ? sp->num_keep_best : 1);      // sequences returned
    // the caller's buffers: device memory or PAGE-LOCKED host memory (hipMemcpyDefault: the results then arrive on the host as part
    // of the request itself -- a server reads them after the stream's event without enqueueing anything more, which matters when
    // its other streams keep the device's queues full: a separate small read-back waits milliseconds for a queue slot)
    HIPCK(hipMemcpyAsync(tokens_out, e->out_tokens, nout * sp->max_steps * sizeof(long long), hipMemcpyDefault, x));
    HIPCK(hipMemcpyAsync(logprob_out, e->out_lp, nout * sizeof(float), hipMemcpyDefault, x));
    HIPCK(hipMemcpyAsync(info_out, e->out_info, 4 * sizeof(int), hipMemcpyDefault, x));
    if (sent_out) HIPCK(hipMemcpyAsync(sent_out, e->out_sent, (size_t)Q * 2 * sizeof(int), hipMemcpyDefault, x));
    if (x != s) {
        HIPCK(hipEventRecord(e->fence_out, x));
        HIPCK(hipStreamWaitEvent(s, e->fence_out, 0));
    }
    return 0;
}

extern "C" int gitmi_generate(gitmi_engine* e, const float* const* frames, int F, int B, const int64_t* prefix, int P,
                              const gitmi_search* sp, int64_t* tokens_out, float* logprob_out, int32_t* info_out,
                              void* stream) {
    RCK(check_ready(e));
    const gitmi_config& c = e->cfg;
    if (!frames || !sp || !tokens_out || !logprob_out || !info_out) return fail("generate: null argument");
    if (F < 1 || F > c.max_frames) return fail("generate: F=%d outside [1,%d]", F, c.max_frames);
    if (B < 1 || B > c.max_batch) return fail("generate: B=%d outside [1,%d]", B, c.max_batch);
    if (!prefix) P = 1;
    if (P < 1 || P > c.max_text_len) return fail("generate: prefix length %d outside [1,%d]", P, c.max_text_len);
    if (sp->max_steps < P || sp->max_steps > c.max_text_len) return fail("generate: max_steps %d outside [P,%d]", sp->max_steps, c.max_text_len);
    hipStream_t s = (hipStream_t)stream;
    // start tokens [B, P] on device (shared prefix, or [CLS]) -- filled by a kernel, no host copy
    RCK(fill_uniform_sentences(e, B, (const long long*)prefix, P, s));
    return generate_run(e, frames, F, B, B, P, P, false, sp, tokens_out, logprob_out, info_out, nullptr, s);
}

// Q sentences with their own prefixes over B encoded images (batched VQA: the questions of one image share its K/V).
extern "C" int gitmi_generate_prefixed(gitmi_engine* e, const float* const* frames, int F, int B, const int64_t* prefixes,
                                       int ld_prefix, const int32_t* prefix_len_host, const int32_t* image_of_host, int Q,
                                       const gitmi_search* sp, int64_t* tokens_out, float* logprob_out,
                                       int32_t* sent_out, int32_t* info_out, void* stream) {
    RCK(check_ready(e));
    const gitmi_config& c = e->cfg;
    if (!frames || !sp || !tokens_out || !logprob_out || !info_out || !prefixes || !prefix_len_host)
        return fail("generate_prefixed: null argument");
    if (F < 1 || F > c.max_frames) return fail("generate_prefixed: F=%d outside [1,%d]", F, c.max_frames);
    if (B < 1 || B > c.max_batch) return fail("generate_prefixed: B=%d outside [1,%d]", B, c.max_batch);
    if (Q < 1 || Q > c.max_batch) return fail("generate_prefixed: Q=%d sentences outside [1,%d]", Q, c.max_batch);
    if (!image_of_host && Q != B) return fail("generate_prefixed: without image_of, Q must equal B");
    int minP = 1 << 30, maxP = 0;
    e->plen_host.assign(prefix_len_host, prefix_len_host + Q);
    e->img_of_host.resize(Q);
    bool ident = true;
    for (int q = 0; q < Q; ++q) {
        const int p = prefix_len_host[q];
        if (p < 1 || p > ld_prefix) return fail("generate_prefixed: prefix length %d of sentence %d outside [1,%d]", p, q, ld_prefix);
        minP = std::min(minP, p); maxP = std::max(maxP, p);
        const int im = image_of_host ? image_of_host[q] : q;
        if (im < 0 || im >= B) return fail("generate_prefixed: sentence %d names image %d of %d", q, im, B);
        e->img_of_host[q] = im;
        ident = ident && im == q;
    }
    if (sp->max_steps < maxP || sp->max_steps > c.max_text_len) return fail("generate_prefixed: max_steps %d outside [%d,%d]", sp->max_steps, maxP, c.max_text_len);
    hipStream_t s = (hipStream_t)stream;
    // the (tiny) host tables are staged synchronously: this entry point serves the VQA task loop, not the benchmark
    HIPCK(hipStreamSynchronize(s));
    HIPCK(hipMemcpy(e->plen_dev, e->plen_host.data(), (size_t)Q * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(e->img_of_dev, e->img_of_host.data(), (size_t)Q * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy2D(e->start_dev, (size_t)c.max_text_len * sizeof(long long), prefixes, (size_t)ld_prefix * sizeof(long long),
                      (size_t)maxP * sizeof(long long), (size_t)Q, hipMemcpyDeviceToDevice));
    e->img_identity = ident;
    return generate_run(e, frames, F, B, Q, minP, maxP, true, sp, tokens_out, logprob_out, info_out, sent_out, s);
}

// ---- profiling --------------------------------------------------------------------------
extern "C" int gitmi_profile_enable(gitmi_engine* e, int on) {
    if (!e) return fail("null engine");
    e->profile_mode = on;
    e->profiling = on == 1;
    e->split_encode_ms = e->split_decode_ms = 0; e->split_calls = e->split_steps = 0;
    e->spans.clear();
    e->event_next = 0;
    return 0;
}
// Serving schedule for several contexts on one device: `e`'s image encoder (+ decoder prefill) of a gitmi_generate call
// starts only after the encoder of `after`'s most recently submitted call has finished; the decode steps are not
// ordered.  Chain the contexts in a ring in submission order: at most one MFMA-bound encoder runs at a time and the
// latency-bound decode chains of the other contexts fill in beside it.  after == NULL removes the dependency.
static void unlink_encode_after(gitmi_engine* e) {
    if (!e->enc_after) return;
    auto& w = e->enc_after->enc_watchers;
    w.erase(std::remove(w.begin(), w.end(), e), w.end());
    e->enc_after = nullptr;
}
extern "C" int gitmi_set_encode_after(gitmi_engine* e, gitmi_engine* after) {
    if (!e) return fail("null engine");
    if (after == e) return fail("set_encode_after: a context cannot wait for its own encoder");
    HIPCK(hipSetDevice(e->device));
    unlink_encode_after(e);
    if (after) {
        if (!after->enc_done) HIPCK(hipEventCreateWithFlags(&after->enc_done, hipEventDisableTiming));
        after->enc_watchers.push_back(e);
        e->enc_after = after;
    }
    return 0;
}

// CaptioningModel.forward_one adds img_temperal_embedding[i] only when batch['image'] is a LIST of frames
// (decoder.py:845-857); a bare tensor goes through image_encoder alone, also on a video model.
extern "C" int gitmi_set_temporal_embedding(gitmi_engine* e, int on) {
    if (!e) return fail("null engine");
    if ((on != 0) != e->use_temb) { e->use_temb = on != 0; e->have_feats = e->have_prefill = false; }
    return 0;
}
// Serving policy: other contexts keep the device busy beside this one.  Kernel shapes are then chosen for what they cost
// the device as a whole rather than for their own duration: the encoder GEMMs take the 256-row tile even where it leaves
// a partial round (the idle CUs are filled by the other contexts), the N = 768 GEMMs of the decode chain take 64 rows per
// workgroup, the decode attention packs 8 (sentence, head) pairs per workgroup.  Results are bit-identical either way.
extern "C" int gitmi_set_shared_device(gitmi_engine* e, int on) {
    if (!e) return fail("null engine");
    if ((on != 0) != e->shared_device) {
        HIPCK(hipSetDevice(e->device));
        HIPCK(hipDeviceSynchronize());
        e->shared_device = on != 0;
        destroy_graph(e);
    }
    return 0;
}
// LayerNorm folding of the encoder / prefill passes (fp16-operand build, on by default there): off = every LayerNorm is a launch
// that materialises its output, as in the bf16 build.  Results differ by rounding only (one rounding of the normalised
// rows less with the fold); the switch exists for A/B timing and for the parity tests that hold both forms to the same bound.
extern "C" int gitmi_set_ln_fold(gitmi_engine* e, int on) {
    if (!e) return fail("null engine");
    if (on && !e->ln_fold_ready)
        return fail("gitmi_set_ln_fold: not available (needs the fp16-operand library, the 16-bit engine mode and hidden sizes that are multiples of 256)");
    if ((on != 0) != e->ln_fold) {
        HIPCK(hipSetDevice(e->device));
        HIPCK(hipDeviceSynchronize());
        e->ln_fold = on != 0;
        e->have_feats = e->have_prefill = false;
        destroy_graph(e);
    }
    return 0;
}
extern "C" int gitmi_set_graph(gitmi_engine* e, int on) {
    if (!e) return fail("null engine");
    e->use_graph = on != 0;
    return 0;
}
extern "C" int gitmi_profile_read(gitmi_engine* e, gitmi_profile* out) {
    if (!e || !out) return fail("null argument");
    HIPCK(hipSetDevice(e->device));
    HIPCK(hipDeviceSynchronize());
    memset(out, 0, sizeof(*out));
    if (e->profile_mode == 2) {
        // graph-replay timing: (encode + prefill) graph and decode graph of every call since enable, averaged per call
        const double n = e->split_calls ? (double)e->split_calls : 1.0;
        out->vit_ms = (float)(e->split_encode_ms / n);            // encode + prefill together (one graph)
        out->decode_ms = (float)(e->split_decode_ms / n);
        out->total_ms = out->vit_ms + out->decode_ms;
        out->decode_steps = e->split_calls ? e->split_steps / e->split_calls : 0;
        out->decode_step_ms = e->split_steps ? (float)(e->split_decode_ms / e->split_steps) : 0.f;
        out->decode_step_bytes = e->last_decode_step_bytes;
        e->split_encode_ms = e->split_decode_ms = 0; e->split_calls = e->split_steps = 0;
        return 0;
    }
    double step_ms = 0;
    for (const TimedSpan& sp : e->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
        switch (sp.tag) {
            case TAG_VIT: out->vit_ms += ms; break;
            case TAG_PREFILL: out->prefill_ms += ms; break;
            case TAG_DECODE: out->decode_ms += ms; break;
            case 99: out->total_ms += ms; break;
            case TAG_STEP: step_ms += ms; out->decode_steps += 1; break;
            case TAG_GEMM_VIT:
                out->vit_gemm_ms += ms; out->vit_gemm_launches += 1; out->vit_gemm_flops += sp.flops;
                // fallthrough
            case TAG_GEMM_OTHER:
                out->gemm_ms += ms; out->gemm_launches += 1; out->gemm_flops += sp.flops;
                break;
            default: break;
        }
    }
    out->decode_step_ms = out->decode_steps ? (float)(step_ms / out->decode_steps) : 0.f;
    out->decode_step_bytes = e->last_decode_step_bytes;
    e->spans.clear();
    e->event_next = 0;
    return 0;
}

// ---- error attribution hooks (tools/error_attribution.py): hand the products of a stage from one context to another
// of the SAME model in the other precision, so that a stage can be switched between bf16 and fp32 on its own.
//   stage 1: visual features (image encoder output)        -> dst runs prefill + decode itself
//   stage 2: + the image K/V of every decoder layer (prefill) -> dst runs only the decode steps itself
GITMI_EXP_EXPORT int gitmi_debug_import_stage(gitmi_engine* dst, gitmi_engine* src, int stage, void* stream) {
    RCK(check_ready(dst));
    if (!src || !src->finalized) return fail("debug_import_stage: bad source");
    if (stage != 1 && stage != 2) return fail("debug_import_stage: stage must be 1 or 2");
    const gitmi_config &a = dst->cfg, &b = src->cfg;
    if (a.vit_width != b.vit_width || a.dec_hidden != b.dec_hidden || a.dec_layers != b.dec_layers || a.dec_heads != b.dec_heads)
        return fail("debug_import_stage: the two contexts are different models");
    if (!src->have_feats || (stage == 2 && !src->have_prefill)) return fail("debug_import_stage: the source has not run that stage");
    if (src->cur_B > a.max_batch || src->cur_F > a.max_frames || src->N != dst->N) return fail("debug_import_stage: capacity / resolution mismatch");
    hipStream_t s = (hipStream_t)stream;
    const size_t M = (size_t)src->cur_B * src->cur_Nimg;
    HIPCK(launch_convert(src->feats, src->f32, dst->feats, dst->f32, M * a.vit_width, s));
    dst->cur_B = src->cur_B; dst->cur_F = src->cur_F; dst->cur_Nimg = src->cur_Nimg;
    dst->have_feats = true; dst->have_prefill = false;
    if (stage == 2) {
        for (int l = 0; l < a.dec_layers; ++l) {
            HIPCK(launch_convert(src->img_kv[l], src->f32, dst->img_kv[l], dst->f32, M * 3 * a.dec_hidden, s));
            RCK(kv_repack(dst, l, dst->cur_B, dst->cur_Nimg, s));
        }
        dst->have_prefill = true;
    }
    return 0;
}
// the vocabulary head of `dst` (bf16: the fused, LayerNorm-folded head) applied to the last hidden state that `src`
// (an fp32 context) computed in its most recent gitmi_step_logits over R rows: logits_out fp32 [R, vocab] (device)
GITMI_EXP_EXPORT int gitmi_debug_head_from(gitmi_engine* dst, gitmi_engine* src, int R, float* logits_out, void* stream) {
    RCK(check_ready(dst));
    if (!src || !src->f32 || !logits_out) return fail("debug_head_from: the source must be an fp32 context");
    if (dst->f32 || !dst->skinny) return fail("debug_head_from: the destination must run the bf16 decode chain");
    const gitmi_config& c = dst->cfg;
    if (c.dec_hidden != src->cfg.dec_hidden || c.vocab != src->cfg.vocab) return fail("debug_head_from: different models");
    if (R < 1 || R > c.max_batch * c.max_beams) return fail("debug_head_from: R outside the capacity");
    hipStream_t s = (hipStream_t)stream;
    HIPCK(launch_chain_input(src->d_y, dst->xo_b, dst->stats_o, R, c.dec_hidden, s));   // d_y: pre-LayerNorm sum of the last layer
    StepCands cands{};
    return decode_head_impl(dst, nullptr, c.max_text_len, 1, R, 1, 0, 1, logits_out, c.vocab, s, &cands);
}
