// Unit entry points of the C ABI (include/gitmi.h: gitmi_op_*, gitmi_preprocess_*): ONE launch each on caller-supplied
// buffers -- what the GPU op tests and tools/ call -- plus the kernel-selection hooks of the measurement build.  No engine
// state: everything that needs a gitmi_engine lives in engine.hip.
#include "../../include/gitmi.h"
#include "abi_common.h"
#include "launchers.h"

#include <hip/hip_runtime.h>

using namespace gitmi;

#ifdef GITMI_EXPERIMENT
#include "../../include/gitmi_experiment.h"
#define GITMI_EXP_EXPORT extern "C"
#else
#define GITMI_EXP_EXPORT [[maybe_unused]] static
#endif

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- single-kernel entry points ------------------------------------------------------------
extern "C" int gitmi_op_gemm(const void* A, const void* W, const float* bias, const float* residual, void* C, int M,
                             int N, int K, int lda, int ldc, int in_dtype, int out_dtype, int act, void* stream) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.res = residual; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldc; g.act = act;
    const bool in_f32 = in_dtype == GITMI_DTYPE_F32;
    if ((in_f32 && K % 16) || (!in_f32 && K % 64)) return fail("op_gemm: K must be a multiple of %d", in_f32 ? 16 : 64);
    if (out_dtype == GITMI_DTYPE_F16) {      // fp16 rows out (a branch output / residual-stream rows of the bf16 engine mode)
        if (in_f32) return fail("op_gemm: fp16 output needs bf16 operands");
        g.out_f16 = 1;
    }
    HIPCK(launch_gemm(g, in_f32, out_dtype == GITMI_DTYPE_F32, (hipStream_t)stream));
    return 0;
}
// the folded-LayerNorm forms of the large-M GEMM (kernels_gemm10.hip LNF; fp16-operand library, more than 512 rows), one launch
extern "C" int gitmi_op_gemm_ln(const void* A, const void* W, const float* bias, const float* colsum, const float* ln_part,
                                float ln_eps, const void* residual, const float* res_part, const float* res_gamma,
                                const float* res_beta, float res_eps, void* C, float* part_out, int M, int N, int K, int act,
                                void* stream) {
    GemmArgs g{};
    g.A = A; g.W = W; g.bias = bias; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N; g.act = act;
    if (K % 64) return fail("op_gemm_ln: K must be a multiple of 64");
    if (ln_part) {            // consumer: C = act(LayerNorm(A) W^T + b), A = raw fp16 rows, W / bias / colsum the folded set
        if (!colsum || residual || part_out || res_part) return fail("op_gemm_ln: consumer form takes colsum and no residual / partial output");
        g.ln_part = (const float2*)ln_part; g.ln_nparts = (K + 255) / 256; g.ln_colsum = colsum; g.ln_inv_d = 1.0f / (float)K; g.ln_eps = ln_eps;
    } else {                  // producer: fp16 stream rows + their row partials
        if (!part_out && !res_part) return fail("op_gemm_ln: neither a consumer (ln_part) nor a producer (part_out / res_part)");
        g.out_f16 = 1;
        g.res = (const float*)residual;
        g.part_out = (float2*)part_out;
        if (res_part) {
            g.res_part = (const float2*)res_part; g.res_nparts = (N + 255) / 256; g.res_gamma = res_gamma; g.res_beta = res_beta;
            g.res_inv_d = 1.0f / (float)N; g.res_eps = res_eps;
        }
    }
    if (!gemm_uses_p8(g, false, false))
        return fail("op_gemm_ln: the folded forms exist in gemm_p8_kernel only (more than 512 rows, N %% 256 == 0, K >= 128)");
    if (launch_gemm(g, false, false, (hipStream_t)stream) != hipSuccess)
        return fail("op_gemm_ln: launch refused (the folded forms are built into the fp16-operand library only)");
    return 0;
}
extern "C" int gitmi_op_layernorm(const float* x, const float* gamma, const float* beta, float eps, void* y_t,
                                  float* y_f32, int rows, int D, int out_dtype, void* stream) {
    HIPCK(launch_layernorm(x, D, gamma, beta, eps, nullptr, y_t, D, out_dtype == GITMI_DTYPE_F32, y_f32, D, rows, D, 0, 0,
                           0, (hipStream_t)stream));
    return 0;
}
extern "C" int gitmi_op_attention(const void* qkv, void* out, int B, int N, int H, int dtype, int impl, void* stream) {
    const size_t esz = dtype == GITMI_DTYPE_F32 ? 4 : 2;
    const int D = H * 64;
    AttnFullArgs a{};
    a.q = qkv;
    a.k = (const char*)qkv + (size_t)D * esz;
    a.v = (const char*)qkv + (size_t)2 * D * esz;
    a.out = out;
    a.ldq = a.ldk = a.ldv = 3 * D;
    a.ldo = D;
    a.N = N; a.H = H; a.scale = 0.125f;
    if (attn_decode_configure() != hipSuccess) return fail("configure failed");
    HIPCK(launch_attn_full(a, B, dtype == GITMI_DTYPE_F32, impl, (hipStream_t)stream));
    return 0;
}

// ---- decode-chain kernels (kernels_dgemm.hip), one launch each -----------------------------------------------
#ifdef GITMI_EXPERIMENT
static int g_dgemm_dbg = 0;
GITMI_EXP_EXPORT int gitmi_debug_set_dgemm(int dbg) { g_dgemm_dbg = dbg; return 0; }      // timing bits of kernels_dgemm.hip
#else
static const int g_dgemm_dbg = 0;
#endif
extern "C" int gitmi_op_dgemm(const void* A, const void* W, const float* bias, const float* colsum, const float* stats,
                              int strips, float eps, void* C, int c_frag, int M, int N, int K, int act, int strips_per_wg,
                              void* stream) {
    DGemmArgs g{};
    g.dbg = g_dgemm_dbg;
    g.strips_per_wg = strips_per_wg;
    g.c_frag = c_frag;
    if (c_frag && N % 32) return fail("op_dgemm: a fragment-major output needs N %% 32 == 0");
    g.A = (const unsigned short*)A; g.lda = K; g.W = (const unsigned short*)W; g.bias = bias;
    if (stats) { g.colsum = colsum; g.stats_in = (const float2*)stats; g.strips_in = strips; g.inv_d = 1.0f / (float)K; g.eps_in = eps; }
    g.C = C; g.ldc = N; g.act = act; g.M = M; g.N = N; g.K = K;
    if (K % 32) return fail("op_dgemm: K must be a multiple of 32");
    HIPCK(launch_dgemm(g, (hipStream_t)stream));
    return 0;
}
extern "C" int gitmi_op_dgemm_res(const void* A, const void* W, const float* bias, const float* res_x, const float* res_stats,
                                  int res_strips, const float* res_gamma, const float* res_beta, float res_eps,
                                  float* x_out, void* xb_out, float* stats_out, int M, int N, int K, void* stream) {
    DGemmArgs g{};
    g.dbg = g_dgemm_dbg;
    g.A = (const unsigned short*)A; g.lda = K; g.W = (const unsigned short*)W; g.bias = bias;
    g.res_x = res_x;
    if (res_stats) { g.res_stats = (const float2*)res_stats; g.res_strips = res_strips; g.res_gamma = res_gamma; g.res_beta = res_beta; g.res_inv_d = 1.0f / (float)N; g.res_eps = res_eps; }
    g.x_out = x_out; g.xb_out = (unsigned short*)xb_out; g.stats_out = (float2*)stats_out;
    g.M = M; g.N = N; g.K = K;
    if (K % 32 || N % 16) return fail("op_dgemm_res: need K %% 32 == 0 and N %% 16 == 0");
    HIPCK(launch_dgemm(g, (hipStream_t)stream));
    return 0;
}
extern "C" int gitmi_op_vocab_topm(const void* A, const void* W, const float* bias, const float* colsum, const float* stats,
                                   int strips, float eps, int M, int V, int K, int cols_per_wg, int mtop,
                                   const int* suppress_tok, float* part_val, int* part_idx, float* part_lse,
                                   float* logits_out, int max_wgs, void* stream) {
    VocabArgs v{};
    v.max_wgs = max_wgs;
    v.A = (const unsigned short*)A; v.lda = K; v.W = (const unsigned short*)W; v.bias = bias;
    if (stats) { v.colsum = colsum; v.stats_in = (const float2*)stats; v.strips_in = strips; v.inv_d = 1.0f / (float)K; v.eps_in = eps; }
    v.M = M; v.N = V; v.K = K; v.cols_per_wg = cols_per_wg;
    if (cols_per_wg != 128) return fail("op_vocab_topm: cols_per_wg must be 128");
    // the rule is driven through the search tables in the engine; the unit entry point takes one token per row
    // (ids [M][1], cur_len 1, prefix length 0 => "past the first step")
    static int* zero_plen = nullptr;
    if (suppress_tok) {
        if (!zero_plen) { HIPCK(hipMalloc((void**)&zero_plen, 4096 * sizeof(int))); HIPCK(hipMemset(zero_plen, 0, 4096 * sizeof(int))); }
        if (M > 4096) return fail("op_vocab_topm: at most 4096 rows with suppress_tok");
        v.ids = suppress_tok; v.ld_ids = 1; v.cur_len = 1; v.plen = zero_plen; v.beams = 1; v.suppress_kind = 1;
    }
    v.part_val = part_val; v.part_idx = part_idx; v.part_lse = (float2*)part_lse;
    v.logits_out = logits_out; v.ld_logits = V;
    HIPCK(launch_vocab_topm(v, mtop, (hipStream_t)stream));
    return 0;
}

// one step of the sampling branch on caller-supplied logits [R, V] (decoder.py:1146-1166): filtered logits (optional),
// ndraw draws per row in draw order and their log-probabilities under the filtered softmax
extern "C" int gitmi_op_sample_rows(const float* logits, int R, int V, float temperature, int top_k, float top_p, int ndraw,
                                    uint64_t seed, int step, float* draw_logprob, int* draw_token, float* filtered_out,
                                    void* stream) {
    float2* lse = nullptr;
    HIPCK(hipMalloc((void**)&lse, (size_t)R * sizeof(float2)));
    hipError_t err = launch_sample_rows(logits, V, V, R, temperature, top_k, top_p, ndraw, seed, step, draw_logprob, draw_token,
                                        lse, filtered_out, nullptr, 0, 0, 0.f, ndraw, (hipStream_t)stream);
    if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
    hipFree(lse);
    HIPCK(err);
    return 0;
}

GITMI_EXP_EXPORT int gitmi_debug_set_gemm_impl(int impl) {
    if (!set_gemm_impl(impl)) return fail("debug_set_gemm_impl: unknown selector %d (low byte: -1, 0 or 9)", impl);
    return 0;
}

extern "C" int gitmi_op_attn_decode(const void* qkv, const void* img_k, const void* img_v, void* txt_k, void* txt_v,
                                    const int* kv_src, void* out, int B, int H, int N_img, int T_max, int pos, int beams,
                                    int dtype, int dbg, void* stream) {
    AttnDecodeArgs a{};
    a.qkv = qkv; a.img_k = img_k; a.img_v = img_v; a.txt_k = txt_k; a.txt_v = txt_v; a.out = out;
    a.kv_src = kv_src; a.ld_src = T_max; a.d = H * 64; a.N_img = N_img; a.T_max = T_max; a.pos = pos; a.beams = beams;
    // dbg: bits 0..15 timing experiments of the kernels (measurement builds), bits 16..17 waves per pair (0 = by geometry,
    // 1, 2), bits 18.. workgroups of the streaming kernel (0 = register kernels)
    a.scale = 0.125f; a.dbg = dbg & 0xffff; a.waves_per_pair = (dbg >> 16) & 3; a.stream_wgs = dbg >> 18;
    if (dtype == GITMI_DTYPE_F32) {
        HIPCK(launch_attn_decode(a, B, H, true, (hipStream_t)stream));
        return 0;
    }
    // bf16: img_k / img_v are the MFMA operand layouts written by gitmi_op_kv_repack (keys padded to 32)
    a.N_pad = round_up(N_img, 32);
    HIPCK(launch_attn_decode_mfma(a, B, H, (hipStream_t)stream));
    return 0;
}
// image-row K/V of the prefill ([B*N, 3*H*64] packed q|k|v, bf16) -> the decode layouts of kernels_attn_decode.hip:
// kf, vt: [B][H][round_up(N, 32)][64] each
extern "C" int gitmi_op_kv_repack(const void* qkv_rows, void* kf, void* vt, int B, int N, int H, void* stream) {
    HIPCK(launch_kv_repack_frag(qkv_rows, kf, vt, B, N, round_up(N, 32), H, H * 64, (hipStream_t)stream));
    return 0;
}

// ---- GPU image transform -------------------------------------------------------
extern "C" int gitmi_preprocess_image(const uint8_t* rgb_hwc, int H, int W, int crop, uint8_t* tmp, size_t tmp_bytes,
                                      float* out_chw, void* stream) {
    if (!rgb_hwc || !out_chw || H < 1 || W < 1 || crop < 1) return fail("preprocess: bad argument");
    const int nw = W <= H ? crop : (int)((double)crop * W / H);
    if (nw != W && (!tmp || tmp_bytes < (size_t)H * nw * 3)) return fail("preprocess: workspace must hold H * %d * 3 bytes", nw);
    HIPCK(launch_preprocess(rgb_hwc, H, W, crop, tmp, out_chw, (hipStream_t)stream));
    return 0;
}

// a batch of decoded images in one staging buffer (include/gitmi.h)
extern "C" int gitmi_preprocess_batch(const uint8_t* rgb, size_t rgb_bytes, const int64_t* desc_host, int n, int crop, uint8_t* tmp,
                                      size_t tmp_bytes, float* out, void* stream) {
    if (!rgb || !desc_host || !out || n < 1 || crop < 1 || crop > 4096) return fail("preprocess_batch: bad argument");
    for (int i = 0; i < n; ++i) {
        const int64_t off = desc_host[3 * i], H = desc_host[3 * i + 1], W = desc_host[3 * i + 2];
        if (H < 1 || W < 1 || H > 65535 || W > 65535 || off < 0 || (uint64_t)off + (uint64_t)H * W * 3 > rgb_bytes ||
            (uint64_t)off + (uint64_t)H * W * 3 > 0xffffffffull)
            return fail("preprocess_batch: image %d (offset %lld, %lld x %lld) does not fit the staging buffer of %zu bytes", i,
                        (long long)off, (long long)H, (long long)W, rgb_bytes);
        const double r = W <= H ? (double)H / W : (double)W / H;
        if (r * crop > 65535.0) return fail("preprocess_batch: image %d: aspect ratio too extreme", i);
    }
    const size_t need = preprocess_batch_workspace((const long long*)desc_host, n, crop);
    if (need > 0 && (!tmp || tmp_bytes < need)) return fail("preprocess_batch: workspace must hold %zu bytes", need);
    if (need > 0xffffffffull) return fail("preprocess_batch: batch too large for one call");
    HIPCK(launch_preprocess_batch(rgb, (const long long*)desc_host, n, crop, tmp, out, (hipStream_t)stream));
    return 0;
}

// MinMaxResizeForTest (inference.py:29-64) output: a plain resize to out_h x out_w (no crop) + ToTensor + Normalize.
// The caller computes (out_h, out_w) with the reference's get_size() rule (generativeimage2text_amd/inference.py).
extern "C" int gitmi_preprocess_image_to(const uint8_t* rgb_hwc, int H, int W, int out_h, int out_w, uint8_t* tmp,
                                         size_t tmp_bytes, float* out_chw, void* stream) {
    if (!rgb_hwc || !out_chw || H < 1 || W < 1 || out_h < 1 || out_w < 1) return fail("preprocess: bad argument");
    if (out_w != W && (!tmp || tmp_bytes < (size_t)H * out_w * 3))
        return fail("preprocess: workspace must hold H * %d * 3 bytes", out_w);
    HIPCK(launch_resize_crop_norm(rgb_hwc, H, W, out_h, out_w, 0, 0, out_h, out_w, tmp, out_chw, (hipStream_t)stream));
    return 0;
}
