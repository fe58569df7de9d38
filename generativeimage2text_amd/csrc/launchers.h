// Host-side declarations of the kernel launchers (definitions live next to the kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gitmi {

struct GemmArgs {
    const void* A; const void* W; const float* bias; const float* res; void* C;
    int M, N, K;
    int lda, ldc, ldr;
    int act;
    int tiles_n, nwg;
    int ng;     // XCD tile partition: N split into ng groups, M into 8/ng (kernels_gemm10.hip)
    int dbg;    // measurement builds only (gemm_p8_kernel): 1 no C stores, 2 no epilogue, 4 no MFMA, 8 no loads in loop
    int out_f16; // C and `res` are f16_t rows (residual stream of the bf16 engine mode); bf16 inputs, p8 + generic kernel only
    int shared;  // other contexts run beside this launch (serving schedule): tile choice by FLOP/byte, not by round fill
    // ---- LayerNorm folded into the large-M GEMMs (gemm_p8_kernel, fp16-operand build; kernels_gemm10.hip) ----
    // Row partials are [M][4] float2 (sum, sumsq), slot t = the 256-column tile t of the row (hidden sizes <= 1024; unused slots 0).
    // consumer (QKV / c_fc / FFN1): A = the RAW fp16 residual-stream rows, W = f16(W . gamma), bias = beta W^T + b, and
    //   C = act(rstd_m (A W^T - mean_m colsum) + bias); (mean, rstd) of row m from ln_part[m][0..4)
    const float2* ln_part; int ln_nparts; const float* ln_colsum; float ln_inv_d, ln_eps;
    // producer (N = hidden GEMMs writing stream rows): (sum, sumsq) of the STORED fp16 values of row m over the 256 columns
    //   of tile t -> part_out[m][t]
    float2* part_out;
    // post-norm residual: `res` points at RAW rows, the residual added is LayerNorm(res) rebuilt from res_part / res_gamma / res_beta
    const float2* res_part; int res_nparts; const float* res_gamma; const float* res_beta; float res_inv_d, res_eps;
};
bool gemm_uses_p8(const GemmArgs& g, bool in_f32, bool out_f32);   // would launch_gemm run this shape on gemm_p8_kernel?
hipError_t launch_gemm(const GemmArgs& g, bool in_f32, bool out_f32, hipStream_t s);
hipError_t launch_gemm_p8(GemmArgs g, bool out_f32, hipStream_t s);     // 256x256x64, half-tile pipeline, staggered wave groups
bool gemm_p8_supports(const GemmArgs& g);
int gemm_p8_cost(const GemmArgs& g, int bm);   // modelled launch time (ns) with bm-row tiles: rounds x (prologue + K loop + epilogue)
bool set_gemm_impl(int impl);   // measurement builds: -1 auto, 0 tile kernel only, 9 LDS-DMA kernel wherever it can run (+ dbg bits << 8); false = unknown selector

// ---- decode-step GEMM chain (kernels_dgemm.hip): LayerNorm folded into the consumer, row partials from the producer
struct DGemmArgs {
    const unsigned short* A; const unsigned short* W;   // bf16 [M, K], [N, K], BOTH fragment-major (gitmi_common.h frag_offset)
    const float* bias;          // [N]  bias, or the folded constant  beta W^T + bias
    const float* colsum;        // [N]  sum_k bf16(W gamma)[n][k] when the input LayerNorm is folded, else nullptr
    const float2* stats_in;     // [strips_in][M] (sum, sumsq) strip partials of the folded LayerNorm's input (nullptr: none)
    int strips_in; float inv_d, eps_in;
    void* C;                    // bf16 [M, ldc] output of the QKV / FFN1 form (row-major), or fragment-major [M, N] when c_frag
    int c_frag;
    // N = 768 form: x_out = A W^T + bias + residual (+ bf16 copy + strip partials of x_out)
    const float* res_x;         // fp32 [M, N]: the hidden state, or the raw x the residual LayerNorm is rebuilt from
    const float2* res_stats; int res_strips; const float* res_gamma; const float* res_beta; float res_inv_d, res_eps;
    float* x_out; unsigned short* xb_out; float2* stats_out;
    int M, N, K, lda, ldc, act;
    int dbg;                    // timing experiments only: 1 no activation loads, 2 no weight loads, 4 no MFMA, 8 no epilogue loads
    int rows_per_wg;            // N = 768 form: rows per workgroup (0 / 16 default, 32, 64)
    int strips_per_wg;          // wide form, 33..64 rows: 16-column strips per workgroup, one after the other (1, 2, 4, 6; serving policy: 2)
    int no_row_walk;            // A/B: wide form over > 64 rows as one workgroup per (strip, row block) instead of the row-walking kernel
};
hipError_t launch_dgemm(const DGemmArgs& g, hipStream_t s);

// vocabulary head with the running top-M / log-sum-exp fused: one sorted candidate list per (row, workgroup)
struct VocabArgs {
    const unsigned short* A; int lda; const unsigned short* W; const float* bias; const float* colsum;
    const float2* stats_in; int strips_in; float inv_d, eps_in;
    int M, N, K, cols_per_wg;
    // no-immediate-repeat rule (decoder.py:330): rows of AUTOREGRESSIVE sentences past their first search step
    const int* ids; int ld_ids, cur_len; const int* plen; int beams, suppress_kind;
    float rep_penalty;          // GENERATOR repetition penalty over ids[row][0..cur_len) (decoder.py:1135-1144); 0 or 1: off
    float* part_val; int* part_idx; float2* part_lse;     // [M][column blocks][slots], [M][column blocks] (max, sum exp)
    float* logits_out; int ld_logits;                      // optional full logits (teacher-forced parity hook)
    int max_wgs;                // workgroups of the launch (0: one per column block); fewer = each walks several column blocks
    int nblk;                   // column blocks (set by the launcher)
};
hipError_t launch_vocab_topm(const VocabArgs& g, int mtop, hipStream_t s);
int vocab_parts(int V, int cols_per_wg);
int vocab_mtop_slots(int mtop);

hipError_t launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                            const float* add_after, void* y_t, int ld_t, bool t_is_f32, float* y_f, int ld_f,
                            int rows, int D, int map_n_in, int map_n_out, int map_off, hipStream_t s);
hipError_t launch_im2col(const float* img, void* out, bool out_f32, int B, int H, int W, int p, int K, int Kpad,
                         hipStream_t s);
hipError_t launch_pos_bicubic(const float* pos, float* out, int g, int gh, int gw, int D, hipStream_t s);
hipError_t launch_vit_assemble_ln(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                                  const float* beta, float eps, void* X, bool x_f16, int B, int N, int D, float2* part,
                                  int nparts, hipStream_t s);   // part: row partials for the folded ln_1 of the first block (or nullptr)
hipError_t launch_layernorm_s16(const void* x, int ldx, const float* gamma, const float* beta, float eps,
                                const float* add_after, void* y_t, int ld_t, bool t_is_f32, void* y_s, int ld_s,
                                int rows, int D, int map_n_in, int map_n_out, int map_off, hipStream_t s);
hipError_t launch_embed_ln(const int* ids, int ld_ids, int pos, const float* words, const float* positions,
                           const float* gamma, const float* beta, float eps, float* h_f, void* h_t, bool t_is_f32,
                           int R, int D, int vocab, bool frag, hipStream_t s);
hipError_t launch_frag_pack(const void* src_bf16, void* dst_bf16, int rows, int rows_out, int K, hipStream_t s);
hipError_t launch_convert_pad(const float* src, void* dst, bool dst_f32, size_t rows, int K, int Kpad,
                              hipStream_t s);
hipError_t launch_convert(const void* src, bool src_f32, void* dst, bool dst_f32, size_t n, hipStream_t s);
hipError_t launch_chain_input(const float* x, void* xb_frag, float2* stats, int R, int d, hipStream_t s);
hipError_t launch_copy_f32(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);
hipError_t launch_fill_i32(int* dst, int value, int n, hipStream_t s);

struct AttnFullArgs {
    const void* q; const void* k; const void* v; void* out;
    int ldq, ldk, ldv, ldo;
    int N, H;
    float scale;
};
hipError_t launch_attn_full(const AttnFullArgs& a, int B, bool is_f32, int impl, hipStream_t s);

struct AttnDecodeArgs {
    const void* qkv; const void* img_k; const void* img_v; void* txt_k; void* txt_v; void* out;   // img_k/v head-major [B][H][N_img][64]
    const int* kv_src;
    const int* img_of;   // [B] image whose K/V the sentence attends to (nullptr: sentence b <-> image b)
    int ld_src;
    int d;
    int N_img, T_max, pos, beams;
    int N_pad;           // MFMA kernel (bf16): image keys padded to a multiple of 32; img_k / img_v in the layouts of kv_repack_frag
    int out_frag;        // write `out` in the fragment-major operand layout of the decode chain (bf16)
    float scale;
    int dbg;             // timing experiments: 1 skip image K/V loads, 2 skip scores, 4 skip PV
    int pairs_per_wg;    // MFMA kernel: (sentence, head) pairs per workgroup (1, 2, 4, 8); > 1 packs the launch onto fewer CUs
    int pairs_per_wave;  // one-wave MFMA kernel: pairs a wave serves one after the other (the next pair's first K/V chunk is
                         // requested while the current pair's text keys / output are worked off); 0 / 1 = one
    int stream_wgs;      // > 0: the streaming kernel (K/V through an LDS ring, 4 independent waves per workgroup) on at most this
                         // many workgroups -- each wave walks its share of the pairs; 0: the register kernels above
    int n_pairs;         // set by the launcher
    int waves_per_pair;  // MFMA kernel: 0 / 1 = one wave walks all key steps of a pair (default); 2 = two waves split them (A/B)
};
hipError_t launch_kv_repack(const void* qkv, void* kh, void* vh, int B, int N, int H, int d, bool is_f32, hipStream_t s);
size_t attn_decode_lds_bytes(int beams, int N_img, int pos);
hipError_t launch_attn_decode(const AttnDecodeArgs& a, int B, int H, bool is_f32, hipStream_t s);
// bf16 decode attention on the matrix cores (kernels_attn_decode.hip) and its cache layouts
hipError_t launch_kv_repack_frag(const void* qkv, void* kf, void* vt, int B, int N, int N_pad, int H, int d, hipStream_t s);
hipError_t launch_attn_decode_mfma(const AttnDecodeArgs& a, int B, int H, hipStream_t s);
hipError_t attn_decode_configure();

constexpr int SS_NHMAX = 8;            // num_keep_best supported by the device search
struct SearchState {
    int B, k, pn, T, V, eos, kind;      // B sentences of k beams; T = max_steps = row stride of ids / kv_src / hyp_tok
    int ragged;                         // 1: every sentence stands for its own batch-1 reference call (own prefix)
    int sampled;                        // GENERATOR sampling branch: candidates arrive as per_node draws per beam, in draw order
    double length_penalty;
    double* len_norm;                   // [T_max + 1] ((5 + len) / 6) ** length_penalty, filled by search_init
    const long long* start;             // [B][ld_start] start tokens of every sentence (its prefix, or [CLS])
    int ld_start;
    const int* plen;                    // [B] prefix length of every sentence (>= 1)
    int* ids[2];
    int* kv_src[2];
    float* score[2];
    int* done;
    // GENERATOR: BeamHypotheses of every sentence (decoder.py:1292-1341), nh = num_keep_best slots each
    int nh;                             // hypotheses kept per sentence (1 .. SS_NHMAX)
    int* hyp_n;                         // [B] hypotheses held
    int* hyp_cnt;                       // [B] hypotheses ever added (the next insertion number)
    double* hyp_worst;                  // [B] BeamHypotheses.worst_score (1e9 while empty)
    double* hyp_score;                  // [B][nh]
    int* hyp_len;                       // [B][nh]
    int* hyp_seq;                       // [B][nh] insertion number: the reference's list order (ties of its sorted())
    int* hyp_tok;                       // [B][nh][T]
    int* stop;                          // AUTOREGRESSIVE: cur_len at which every beam of the sentence had ended (0: not yet)
    int* early;                         // AUTOREGRESSIVE, k == 1: the sentence's first prediction was EOS
    int* info;                          // [0] sentences ended / done so far, [2] steps run
};
// candidate lists of a step: per row `nparts` sorted lists of `slots` (logit, token) pairs + (max, sum exp) per part
struct StepCands {
    const float* part_val; const int* part_idx; const float2* part_lse;
    int nparts, slots;
};
// embedding of the tokens a step appends = input of the next decode step (decoder.py:65-78); words == nullptr: skip
struct EmbedArgs {
    const float* words; const float* positions; const float* gamma; const float* beta; float eps;
    float* h_f; void* h_t; int D, vocab;
    int frag;            // h_t in the fragment-major operand layout of the decode chain
};
int row_topm_slots(int M);
// sampling branch (decoder.py:1146-1166, 1343-1375): per row scores / temperature -> top-k / top-p filter (min 2 tokens
// kept) -> `ndraw` draws without replacement (Gumbel-top-k on a counter-based generator) -> their log-probabilities under
// the filtered softmax, in draw order, as one candidate list per row (part_lse = (0, 1): values are log-probs already).
// filtered_out (optional): the filtered logits [R, V] (-inf = removed) for parity checks.
hipError_t launch_sample_rows(const float* logits, int ldl, int V, int R, float temperature, int top_k, float top_p,
                              int ndraw, unsigned long long seed, int step, float* part_val, int* part_idx,
                              float2* part_lse, float* filtered_out, const int* ids, int ld_ids, int cur_len,
                              float rep_penalty, int stride, hipStream_t s);      // stride >= ndraw: entries of a row's list
// token trie on the device (CSR: the children of node n are edges child_off[n] .. child_off[n + 1]) + one cursor per sentence
struct TrieArgs {
    const int* child_off; const int* child_tok; const int* child_node;
    int* cursor;                 // [B] current node of every sentence (0 = root, -1 = unconstrained)
};
// trie-constrained greedy selection on materialised logits [B, ldl] (trie_decoder.py:57-71, 115-158): one candidate
// (log-prob incl. the trie bonus, token) per sentence in the candidate-list format of the search step; moves the cursors
hipError_t launch_trie_select(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len, const int* plen,
                              int eos, const TrieArgs& tr, int B, float* part_val, int* part_idx, float2* part_lse,
                              hipStream_t s);
hipError_t launch_row_topm(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len,
                           const int* plen, int beams, int suppress_kind, float rep_penalty, int M, int R,
                           float* part_val, int* part_idx, float2* part_lse, hipStream_t s);
hipError_t launch_search_step(const SearchState& st, int src, int cur_len, const StepCands& in, const EmbedArgs& em,
                              bool t_is_f32, hipStream_t s);
hipError_t launch_search_init(const SearchState& st, hipStream_t s);
hipError_t launch_search_finish(const SearchState& st, int cur, int cur_len, long long* tokens_out,
                                float* logprob_out, int* info_out, int* sent_out, hipStream_t s);
hipError_t launch_search_rows(const SearchState& st, int cur, int cur_len, long long* out, hipStream_t s);
hipError_t launch_fill_start(long long* start, int ld, const long long* prefix, int ldp, int shared, int sos, int B, int P,
                             hipStream_t s);
hipError_t launch_load_ids(const long long* tokens, int R, int t, int* ids, int* kv_src, int ld, hipStream_t s);

// GPU image transform (Pillow-exact bicubic resize + centre crop + CLIP normalisation)
hipError_t launch_preprocess(const unsigned char* rgb, int H, int W, int crop, unsigned char* tmp, float* out, hipStream_t s);
hipError_t launch_preprocess_batch(const unsigned char* rgb, const long long* desc, int n, int crop, unsigned char* tmp, float* out,
                                   hipStream_t s);
size_t preprocess_batch_workspace(const long long* desc, int n, int crop);
hipError_t launch_resize_crop_norm(const uint8_t* rgb, int H, int W, int nh, int nw, int top, int left, int ch, int cw,
                                   uint8_t* tmp, float* out, hipStream_t s);

}  // namespace gitmi
