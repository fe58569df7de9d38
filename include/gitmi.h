/*
 * gitmi.h -- C ABI of the MI355X-native GIT captioning / VQA inference engine.
 *
 * The reference (microsoft/GenerativeImage2Text) has no FFI / operator layer: its
 * extension seams are Python duck-typed.  This header is the drop-in
 * boundary a maintainer binds instead of those seams; every entry point names the
 * reference interface it replaces (paths relative to
 * /root/reference/generativeimage2text/).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C, no torch types.  All tensor pointers are DEVICE pointers unless the
 *     parameter name ends in _host.  Caller owns inputs/outputs; the engine owns
 *     repacked weights, KV caches and workspaces sized at gitmi_create().
 *   - every call returns 0 on success, non-zero on error; gitmi_last_error() returns
 *     a thread-local message.  Nothing throws across the ABI.
 *   - one engine per device per process; calls on one engine are serialised by the
 *     caller and are asynchronous w.r.t. the host on the hipStream_t passed as
 *     `stream` (a void* so that this header needs no HIP include).
 *   - no CPU fallback exists: without a gfx950 device gitmi_create() fails.
 */
#ifndef GITMI_H_
#define GITMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GITMI_ABI_VERSION 10

/* compute precision of GEMM/attention operands (accumulation, LayerNorm statistics,
 * softmax, residual stream and logits are fp32 in both modes) */
#define GITMI_PREC_BF16 0   /* 16-bit MFMA operands (the type of the library: below) */
#define GITMI_PREC_F32  1   /* f32-input MFMA, exact fp32 -- parity-debug path        */

/* tensor dtypes accepted by gitmi_load_tensor */
#define GITMI_DTYPE_F32  0
#define GITMI_DTYPE_BF16 1
#define GITMI_DTYPE_F16  2

/* search strategies (layers/decoder.py) */
#define GITMI_SEARCH_AUTOREGRESSIVE 0  /* AutoRegressiveBeamSearch   decoder.py:208-440  */
#define GITMI_SEARCH_GENERATOR      1  /* GeneratorWithBeamSearch    decoder.py:1056-1290 */
#define GITMI_SEARCH_TRIE           2  /* TrieAutoRegressiveBeamSearch (beam 1, token trie: gitmi_set_trie)  trie_decoder.py:27-218 */

typedef struct gitmi_engine gitmi_engine;

/* Model + capacity description.  Replaces the hyper-parameters hard-coded in
 * model.py:9-61 (decoder) / CLIP build_model (encoder) / aux_data/models/<m>/parameter.yaml. */
typedef struct gitmi_config {
    int32_t image_size;     /* test_crop_size, 224                                    */
    int32_t patch;          /* 16 (ViT-B/16) | 14 (ViT-L/14)                          */
    int32_t vit_width;      /* 768 | 1024  (== visual_feature_size)                   */
    int32_t vit_layers;     /* 12 | 24                                                */
    int32_t vit_heads;      /* 12 | 16   (head_dim must be 64)                        */
    int32_t dec_hidden;     /* 768                                                    */
    int32_t dec_layers;     /* 6                                                      */
    int32_t dec_heads;      /* 12  (head_dim must be 64)                              */
    int32_t dec_ffn;        /* 3072                                                   */
    int32_t vocab;          /* 30522                                                  */
    int32_t max_pos;        /* 1024 (max_caption_length, model.py:21)                 */
    int32_t num_frames;     /* num_image_with_embedding: temporal embeddings, 0=none  */
    int32_t sos;            /* tokenizer.cls_token_id = 101                           */
    int32_t eos;            /* tokenizer.sep_token_id = 102                           */
    int32_t precision;      /* GITMI_PREC_*                                           */
    int32_t max_batch;      /* capacity: images per call                              */
    int32_t max_beams;      /* capacity: beam_size                                    */
    int32_t max_frames;     /* capacity: frames per sample (>=1)                      */
    int32_t max_text_len;   /* capacity: prefix + generated tokens (KV-cache length)  */
    int32_t max_image_pixels; /* capacity: H*W of an input frame; 0 = image_size^2. Models with
                               * test_respect_ratio_max (MinMaxResizeForTest, inference.py:29-64):
                               * test_crop_size * test_respect_ratio_max               */
    int32_t max_image_tokens; /* capacity: (H/patch)*(W/patch)+1 per frame; 0 = native grid */
} gitmi_config;

/* Replaces the constructor arguments of the two search classes
 * (decoder.py:209-222, 1057-1081). */
typedef struct gitmi_search {
    int32_t kind;                 /* GITMI_SEARCH_*                                   */
    int32_t beam_size;
    int32_t per_node_beam_size;
    int32_t max_steps;            /* TOTAL length incl. [CLS]/prefix (decoder.py:313, 1111) */
    double  length_penalty;       /* GENERATOR only (double: the reference does this math in Python floats) */
    /* sampling branch of GeneratorWithBeamSearch.search (decoder.py:1146-1166): scores / temperature ->
     * top_k_top_p_filtering(min_tokens_to_keep = 2, decoder.py:1343-1375) -> per_node_beam_size draws WITHOUT replacement
     * from the filtered softmax -> log-probabilities of the draws; the beam bookkeeping that follows is the same.
     * torch.multinomial's random stream cannot be reproduced: draws come from a counter-based generator keyed by
     * (seed, step, row, token) through the Gumbel-top-k construction (exactly that sampling distribution). */
    int32_t do_sample;            /* 0 = greedy beam search (default); 1 = sample (GENERATOR only)     */
    int32_t top_k;                /* <= 0: off                                                         */
    double  top_p;                /* >= 1 or <= 0: off                                                 */
    double  temperature;          /* > 0; 1 = unchanged (0 is read as 1)                               */
    uint64_t seed;
    /* GENERATOR only: repetition penalty (decoder.py:1064, 1135-1144) -- before the log-softmax (and before the sampling
     * filter) the raw score of every token already in a row's history (start tokens included) is multiplied by it when
     * negative and divided by it otherwise.  >= 1; 0 and 1 both mean off. */
    double  repetition_penalty;
    /* GENERATOR only: `num_keep_best` of GeneratorWithBeamSearch.search (decoder.py:1087, 1113-1115, 1262-1290) -- every
     * sentence keeps its n best finished hypotheses (BeamHypotheses(n_hyp = n)) and returns all of them, best first:
     * tokens_out becomes [B, n, max_steps], logprob_out [B, n]; sequences the list cannot fill are all EOS with
     * log-prob -1e5, as in the reference.  1 .. 8; 0 is read as 1.  (The reference's `num_return_sequences` -- r independent
     * sentences per image -- is the host mirror's business: r sentences with the same image index, gitmi_generate_prefixed.) */
    int32_t num_keep_best;
    int32_t reserved_;
} gitmi_search;

/* per-phase device timings (ms) of the last profiled gitmi_generate(), see gitmi_profile_enable */
typedef struct gitmi_profile {
    float    vit_ms, prefill_ms, decode_ms, total_ms;
    float    gemm_ms;             /* sum over every GEMM launch of the call            */
    int32_t  gemm_launches;
    double   gemm_flops;          /* algorithmic 2*M*N*K summed over those launches    */
    float    vit_gemm_ms;         /* GEMM launches of the image encoder only           */
    int32_t  vit_gemm_launches;
    double   vit_gemm_flops;
    float    decode_step_ms;      /* average over executed decode steps                */
    int32_t  decode_steps;
    double   decode_step_bytes;   /* algorithmic bytes per step: weights + KV read     */
} gitmi_profile;

/* ---- lifecycle ------------------------------------------------------------------- */
int  gitmi_abi_version(void);
/* The 16-bit operand type of the library's fast mode (GITMI_PREC_BF16 of gitmi_config.precision means "the 16-bit operand
 * mode of this build"): GITMI_DTYPE_BF16 for libgitmi.so, GITMI_DTYPE_F16 for libgitmi_f16.so -- the benchmarked build since round 6 --,
 * the same sources built with -DGITMI_OPS_F16 (IEEE fp16 operands on v_mfma_f32_16x16x32_f16: same rate, 3 more
 * mantissa bits; 16-bit operands handed to the gitmi_op_* entry points are then fp16 too). */
int  gitmi_operand_dtype(void);
const char* gitmi_last_error(void);
/* replaces get_git_model(tokenizer, param) + model.cuda()  (model.py:9-61, inference.py:83-87) */
int  gitmi_create(const gitmi_config* cfg, int device, gitmi_engine** out);
void gitmi_destroy(gitmi_engine* e);

/* ---- checkpoint ingest:  replaces load_state_dict(model, ckpt) (torch_common.py:93-145).
 * `key` is the reference state-dict key, e.g.
 * "image_encoder.transformer.resblocks.3.attn.in_proj_weight".  `data_host` is a HOST
 * pointer to a dense row-major tensor. Unknown keys return an error; "image_encoder.proj"
 * is accepted and ignored (unused when output_grid=True, CLIP/model.py:263-268). */
int  gitmi_load_tensor(gitmi_engine* e, const char* key, const void* data_host,
                       const int64_t* shape, int ndim, int dtype);
/* repack for the device: fused QKV, K padding, compute-dtype copies; ties
 * textual.output.weight to embedding.words.weight if it was not loaded (decoder.py:503-505) */
int  gitmi_finalize_weights(gitmi_engine* e);

/* second execution context on the same device that borrows the packed weights of `src` (own
 * workspaces / KV caches / search state / graph): several batches in flight on different streams.
 * `src` must outlive the clone; destroy clones with gitmi_destroy. */
int  gitmi_clone(gitmi_engine* src, gitmi_engine** out);

/* serving schedule for several contexts of one device: the image encoder (+ decoder prefill) of `e`'s gitmi_generate
 * calls starts only after the encoder of `after`'s most recently submitted call has finished (decode steps are not
 * ordered).  Chain context i (in submission order) after context i - c: at most c MFMA-bound encoders run at a time while
 * the latency-bound decode chains of the other contexts fill in beside them (measured best on MI355X: 4 contexts, c = 2;
 * bench.py --encoder-chains).  after = NULL: no dependency (the context goes back to one hipGraph per call once no other
 * context waits for it either).  Destroying either context of a link removes the link. */
int  gitmi_set_encode_after(gitmi_engine* e, gitmi_engine* after);

/* serving policy: tell a context that other contexts keep the device busy beside it.  Kernel shapes are then chosen
 * for what a launch costs the device as a whole, not for its own duration: the image-encoder GEMMs always take the
 * 256x256 tile (a partial round's idle CUs are filled by the other contexts; measured +2.2 % captions/s in the mixed
 * schedule, -3 % for a context alone), the N = 768 GEMMs of the decode chain take 32 rows per workgroup (64 for beam
 * batches: beam-4 +4.5 %), the decode attention packs 8 instead of 4 (sentence, head) pairs per workgroup (+0.8 %), the wide
 * chain GEMMs take two 16-column strips per workgroup one after the other (+1.3 %) and, for beam batches, walk their row
 * blocks in one workgroup per weight strip (beam-4 +1.7 %), the vocabulary head runs on ~60 workgroups that each walk four
 * column blocks (+1 %).
 * Results are bit-identical either way.  Clones inherit the setting of their source at clone time. */
int  gitmi_set_shared_device(gitmi_engine* e, int on);
/* (new, ABI 9) LayerNorm folding in the image encoder and the prefill -- replaces the LayerNorm modules of
 * layers/CLIP/model.py:161-168,189-202 (ln_1 / ln_2) and layers/bert/modeling_bert.py:171-178,243-250 + layers/decoder.py:35
 * (post-norm LayerNorms over the image rows) for passes of more than 512 rows in the fp16-operand library: the producer
 * GEMM leaves (sum, sumsq) per row, the consumer GEMM reads the raw stream rows and applies the normalisation in its
 * epilogue.  on = 1 is the default where it is available; on = 0 runs one LayerNorm launch per module instead (what
 * libgitmi.so, the f32 mode and small batches always do).  Fails if on = 1 is asked of an engine that cannot fold. */
int  gitmi_set_ln_fold(gitmi_engine* e, int on);

/* ---- input resolution of the following encode/generate calls (default: image_size x image_size).
 * Replaces the run-time branch of VisualTransformer.forward for inputs that are not the native
 * resolution (CLIP/model.py:243-251): the token grid becomes (H / patch) x (W / patch) -- the stride-patch
 * convolution ignores the H % patch / W % patch remainder -- and the positional table is resized to it with
 * torch's bicubic (align_corners=False) kernel, class row kept.  H*W and the token count must fit the
 * max_image_pixels / max_image_tokens capacities given at gitmi_create(). */
int  gitmi_set_image_shape(gitmi_engine* e, int H, int W, void* stream);

/* ---- image encoder: replaces model.image_encoder(x) + the multi-frame branch of
 * CaptioningModel.forward_one (CLIP/model.py:240-274, decoder.py:845-857).
 * frames: F device pointers to fp32 [B,3,H,W] (H, W as set by gitmi_set_image_shape).  The visual features
 * [B, F*N, vit_width] stay resident in the engine; feats_out (optional, fp32) receives a copy. */
int  gitmi_encode_frames(gitmi_engine* e, const float* const* frames, int F, int B,
                         float* feats_out, void* stream);

/* batch['image'] given as a LIST of frames (on = 1, default) or as a bare tensor (on = 0): the reference adds
 * img_temperal_embedding[i] to frame i only in the list case (decoder.py:845-857). */
int  gitmi_set_temporal_embedding(gitmi_engine* e, int on);

/* ---- decoder prefill over image tokens (visual_projection + image rows of all layers);
 * builds the per-image K/V cache.  Mathematically the image part of
 * TransformerDecoderTextualHead.forward (decoder.py:521-600), computed once per image. */
int  gitmi_prefill(gitmi_engine* e, void* stream);

/* ---- teacher-forced step == the reference's `step` callable
 * (CaptioningModel.decoding_step, decoder.py:1013-1054): tokens int64 [R,t], R = B*beams with
 * rows of one image contiguous; writes fp32 next-token logits [R, vocab]. */
int  gitmi_step_logits(gitmi_engine* e, const int64_t* tokens, int R, int t,
                       float* logits_out, void* stream);

/* ---- whole hot path: replaces model({'image':..., 'prefix':...})
 * (CaptioningModel.forward/infer, decoder.py:838-877, 977-1011) incl. the search loop.
 *   prefix       : int64 [P] starting with [CLS], or NULL for captioning (P ignored);
 *                  the reference requires B==1 with a prefix (decoder.py:988) -- here a
 *                  single prefix is shared by all B images.
 *   tokens_out, logprob_out, info_out (and sent_out of gitmi_generate_prefixed): device buffers OR page-locked host buffers
 *                  (hipHostMalloc / a pinned torch tensor): with host buffers the results are on the host when the stream reaches
 *                  the end of the call, no read-back has to be enqueued afterwards.
 *   tokens_out   : int64 [B, max_steps]; sequences INCLUDE the start tokens, EOS-padded.
 *   logprob_out  : fp32 [B]  (AUTOREGRESSIVE: sum/num_valid, decoder.py:429-438;
 *                              GENERATOR: length-normalised score, decoder.py:1310-1320)
 *   info_out     : int32 [4] = { seq_len, early_all_eos, steps_run, nonfinite }
 *                  seq_len = length of the tensor the reference returns (AUTOREGRESSIVE stops
 *                  when every beam ended, decoder.py:319; GENERATOR always max_steps);
 *                  early_all_eos=1 is the first-step early return of decoder.py:279-291;
 *                  steps_run = text positions appended (max_steps - 1 unless a long-budget call stopped early);
 *                  nonfinite = returned sequences whose log-prob is inf / NaN: an operand left the range of the 16-bit
 *                  operand format (fp16: 65504) somewhere upstream -- the ids of such a call are meaningless and the
 *                  caller must treat it as an error (the Python binding raises).  Weights are range-checked when they
 *                  are loaded (gitmi_load_tensor / gitmi_finalize_weights fail by name). */
int  gitmi_generate(gitmi_engine* e, const float* const* frames, int F, int B,
                    const int64_t* prefix, int P, const gitmi_search* search,
                    int64_t* tokens_out, float* logprob_out, int32_t* info_out, void* stream);

/* ---- batched VQA: Q sentences with their OWN prefixes over B images.  The reference answers one question per
 * model call (decoder.py:984-989 asserts a single prefix; inference.py:172-199 loops); here the questions of one
 * image share its encoded K/V and questions of different lengths share every decode step (all sentences sit at the
 * same text position; a sentence still inside its prefix just appends the given token).  Each sentence gets exactly
 * what its own batch-1 reference call returns.
 *   prefixes        : int64 [Q, ld_prefix] DEVICE, row q = its tokens incl. [CLS] (entries past its length ignored)
 *   prefix_len_host : int32 [Q] HOST, 1 <= len <= ld_prefix
 *   image_of_host   : int32 [Q] HOST, image index of every sentence, or NULL (Q == B, sentence q <-> image q)
 *   tokens_out      : int64 [Q, max_steps], logprob_out fp32 [Q] as in gitmi_generate
 *   sent_out        : int32 [Q, 2] = (length of the sequence returned for this sentence, its early-return flag) or NULL */
int  gitmi_generate_prefixed(gitmi_engine* e, const float* const* frames, int F, int B,
                             const int64_t* prefixes, int ld_prefix, const int32_t* prefix_len_host,
                             const int32_t* image_of_host, int Q, const gitmi_search* search,
                             int64_t* tokens_out, float* logprob_out, int32_t* sent_out, int32_t* info_out,
                             void* stream);

/* ---- search with caller-supplied logits: the seam decoder.search(start, step)
 * (decoder.py:224-231, 1083-1092) for scripted-step parity tests of the device search.
 * begin -> [ next_input -> (caller computes logits) -> advance ]* -> finish. */
/* ---- token trie of GITMI_SEARCH_TRIE: replaces TokenTrie (trie_decoder.py:224-257) as CSR arrays on the host (copied):
 * node 0 is the root; the children of node n are the edges child_off[n] .. child_off[n+1]-1, edge i = (token
 * child_tok[i] -> node child_node[i]), children in insertion order.  At every search step the log-probabilities of the
 * cursor's child tokens are raised by (max - min + 1) of the step's logits before the top-1 (trie_decoder.py:57-71,
 * 115-158) and the cursor follows the choice.  EVERY sentence of a call has its own cursor, i.e. behaves like its own
 * batch-1 reference call (the reference moves one cursor with row 0's choice).  n_nodes == 0 removes the trie. */
int  gitmi_set_trie(gitmi_engine* e, int n_nodes, const int32_t* child_off, const int32_t* child_tok,
                    const int32_t* child_node);

int  gitmi_search_begin(gitmi_engine* e, const gitmi_search* search, int B,
                        const int64_t* start_host, int P, int vocab, void* stream);
/* current rows the `step` callable would receive: int64 [R, cur_len]; returns cur_len via *t */
int  gitmi_search_rows(gitmi_engine* e, int64_t* tokens_out, int* R, int* t, void* stream);
int  gitmi_search_advance(gitmi_engine* e, const float* logits, void* stream);
/* sentences that need no further step (synchronises `stream`): GENERATOR -- sentences whose BeamHypotheses are done
 * (the reference leaves its loop at all(done), decoder.py:1251); AUTOREGRESSIVE -- sentences whose beams all ended with
 * EOS (decoder.py:319).  Steps past *done_out == B are idempotent; a host loop uses this to stop calling `step`. */
int  gitmi_search_done_count(gitmi_engine* e, int* done_out, void* stream);
int  gitmi_search_finish(gitmi_engine* e, int64_t* tokens_out, float* logprob_out,
                         int32_t* info_out, void* stream);

/* ---- profiling ---------------------------------------------------------------------- */
/* on = 1: eager launches with HIP events around phases, decode steps and every GEMM launch (per-kernel durations);
 * on = 2: hipGraph replays with the call split into an (encode + prefill) graph and a decode graph and events
 *         between them -- the production launch path, timed: vit_ms = encode + prefill, decode_ms, decode_step_ms;
 * on = 0: off. */
int  gitmi_profile_enable(gitmi_engine* e, int on);
int  gitmi_profile_read(gitmi_engine* e, gitmi_profile* out);   /* synchronises */
/* use hipGraph replay for gitmi_generate (default 1 unless profiling) */
int  gitmi_set_graph(gitmi_engine* e, int on);

/* ---- single-kernel entry points (unit parity tests through the C ABI).
 * dtype arguments are GITMI_DTYPE_F32 / GITMI_DTYPE_BF16. ------------------------------- */
/* C[M,N] = act(A[M,K] * W[N,K]^T + bias) (+ residual);  act: 0 none, 1 QuickGELU, 2 erf-GELU.
 * A, W in `in_dtype`; bias/residual fp32 (may be NULL); C in `out_dtype`. K % 64 == 0. */
int  gitmi_op_gemm(const void* A, const void* W, const float* bias, const float* residual,
                   void* C, int M, int N, int K, int lda, int ldc, int in_dtype, int out_dtype,
                   int act, void* stream);
/* (ABI 10) the folded-LayerNorm forms of the large-M GEMM, one launch (libgitmi_f16.so only; M > 512, N % 256 == 0, K % 64 == 0):
 *  consumer (ln_part != NULL): C fp16 [M,N] = act(LayerNorm_K(A) W0^T + b0) computed from the RAW fp16 rows A [M,K] with the
 *    folded set W = f16(W0 . gamma) [N,K], bias = beta W0^T + b0, colsum[n] = sum_k W[n][k], ln_part = float2 [M][4]: (sum, sumsq)
 *    of row m over its 256-column tiles (unused slots 0);
 *  producer (ln_part == NULL): fp16 stream rows C [M,N] = A W^T + bias + r, r = residual rows (fp16 [M,N], may be NULL) or
 *    LayerNorm_N(residual rows) rebuilt from res_part / res_gamma / res_beta; part_out = float2 [M][4] of the rows as stored
 *    (N <= 1024).  This is what replaces the LayerNorm modules between the GEMMs of CLIP/model.py:189-202 and
 *    modeling_bert.py:171-178, 243-250 inside the engine (gitmi_set_ln_fold). */
int  gitmi_op_gemm_ln(const void* A, const void* W, const float* bias, const float* colsum, const float* ln_part,
                      float ln_eps, const void* residual, const float* res_part, const float* res_gamma,
                      const float* res_beta, float res_eps, void* C, float* part_out, int M, int N, int K, int act,
                      void* stream);
/* y = LayerNorm(x) (biased variance, eps), x fp32 [rows, D]; y_t (in out_dtype) and/or y_f32 */
int  gitmi_op_layernorm(const float* x, const float* gamma, const float* beta, float eps,
                        void* y_t, float* y_f32, int rows, int D, int out_dtype, void* stream);

/* full (unmasked) multi-head attention over packed qkv [B*N, 3*D] (q|k|v, head h = cols h*64..);
 * out [B*N, D].  impl: 0 = reference VALU kernel, 1 = MFMA flash kernel (bf16 only). */
int  gitmi_op_attention(const void* qkv, void* out, int B, int N, int H, int dtype, int impl,
                        void* stream);

/* decode-step GEMM chain (kernels_dgemm.hip; bf16 A [M,K], W [N,K], K % 32 == 0).  BOTH operands are FRAGMENT-MAJOR:
 * 16-row x 32-k tiles in MFMA operand order, element (row, k) at
 *     (((row/16)*(K/32) + k/32)*64 + ((k%32)/8)*16 + row%16)*8 + k%8,   rows padded to a multiple of 16
 * (weights: zero rows; bias / colsum of the vocabulary head padded to a multiple of cols_per_wg), so that every
 * fragment load of a wave is one contiguous 1-KiB read.  The engine repacks weights once and its kernels write
 * activations in this order; xb_out and (with c_frag) C come out fragment-major too.  The post-norm BERT layer
 * (modeling_bert.py:171-178, 243-250) runs WITHOUT LayerNorm launches: the N = hidden GEMMs emit the pre-LayerNorm
 * sum and per-16-column-strip row partials (sum, sum of squares), the consumer GEMM folds the LayerNorm.
 * stats layout: fp32 [strips][M][2].
 *   gitmi_op_dgemm     : C bf16 [M,N] = act( LN_fold(A) W^T + bias ); stats == NULL: plain A W^T + bias.
 *                        With stats: W must be bf16(W . gamma), bias = beta W^T + b, colsum[n] = sum_k W'[n][k].
 *   gitmi_op_dgemm_res : x = A W^T + bias + r,  r = res_x (res_stats == NULL) or LayerNorm(res_x; res_gamma, res_beta)
 *                        rebuilt from res_stats; writes x fp32, its bf16 copy and stats_out [N/16][M][2].  N % 16 == 0.
 *                        strips_per_wg (QKV / FFN1 form, 33..64 rows): 16-column strips a workgroup computes one after
 *                        the other -- 0 / 1 (one workgroup per strip), 2, 4 or 6; results do not depend on it. */
int  gitmi_op_dgemm(const void* A, const void* W, const float* bias, const float* colsum, const float* stats,
                    int strips, float eps, void* C, int c_frag, int M, int N, int K, int act, int strips_per_wg,
                    void* stream);
int  gitmi_op_dgemm_res(const void* A, const void* W, const float* bias, const float* res_x, const float* res_stats,
                        int res_strips, const float* res_gamma, const float* res_beta, float res_eps,
                        float* x_out, void* xb_out, float* stats_out, int M, int N, int K, void* stream);
/* vocabulary head with the search's top-M and log-softmax statistics fused (replaces decoder.py:1054 + the
 * log_softmax/topk of :265-271, 358-366, 1169-1175): per COLUMN BLOCK of cols_per_wg = 128 columns and row, a sorted list
 * of `slots` = {1,2,4,8,16} >= mtop (logit, token) pairs + (max, sum exp):
 * part_val/part_idx [M][ceil(V/128)][slots], part_lse [..][2].  suppress_tok int32 [M] (or NULL): that
 * token's logit counts as -10000 (decoder.py:330).  logits_out fp32 [M,V] optional.  K <= 768.
 * max_wgs: workgroups of the launch -- 0 = one per column block; n > 0 = at most n, each WALKING its column blocks with a
 * rolling refill of the weight registers (at most 8 blocks per workgroup); results do not depend on it.
 * A rows must be readable up to the next multiple of 64, W rows / bias / colsum up to the next multiple of 128. */
int  gitmi_op_vocab_topm(const void* A, const void* W, const float* bias, const float* colsum, const float* stats,
                         int strips, float eps, int M, int V, int K, int cols_per_wg, int mtop,
                         const int* suppress_tok, float* part_val, int* part_idx, float* part_lse,
                         float* logits_out, int max_wgs, void* stream);

/* decode attention for one new text position (unit parity / timing): qkv [R,3d] (R = B*beams), text caches
 * [R][T_max][d] (position `pos` is appended), kv_src int32 [R][T_max], out [R,d].
 *   fp32 : image K/V head-major [B][H][N_img][64] (scalar kernel);
 *   bf16 : image K/V in the matrix-core operand layouts written by gitmi_op_kv_repack (keys padded to 32).
 * dbg: 0, or timing-experiment bits of the fp32 kernel (results undefined). */
int  gitmi_op_attn_decode(const void* qkv, const void* img_k, const void* img_v, void* txt_k, void* txt_v,
                          const int* kv_src, void* out, int B, int H, int N_img, int T_max, int pos, int beams,
                          int dtype, int dbg, void* stream);
/* bf16 image-row K/V of the decoder prefill ([B*N, 3*H*64] packed q|k|v) -> decode layouts kf / vt, each
 * [B][H][round_up(N,32)][64]: K fragment-major (a wave's MFMA operand is one contiguous 1-KiB read), V transposed with
 * the key slots ordered so that softmax probabilities feed the P V product without leaving their registers. */
int  gitmi_op_kv_repack(const void* qkv_rows, void* kf, void* vt, int B, int N, int H, void* stream);

/* GPU-side image transform == get_image_transform(param) of the reference (inference.py:111-132):
 * Resize(crop, BICUBIC) -> CenterCrop(crop) -> ToTensor -> Normalize(CLIP mean/std), bit-exact with the
 * PIL/torchvision pipeline (Pillow's 8.22 fixed-point resampler is reproduced).  rgb_hwc: uint8 [H,W,3] on
 * the device (a decoded image); tmp: device workspace of >= H * W_resized * 3 bytes; out_chw: fp32 [3,crop,crop]. */
int  gitmi_preprocess_image(const uint8_t* rgb_hwc, int H, int W, int crop, uint8_t* tmp, size_t tmp_bytes,
                            float* out_chw, void* stream);
/* the same transform for a BATCH: n decoded images of any sizes lie in ONE device buffer `rgb` (rgb_bytes long; one upload per
 * batch), desc_host is a HOST array int64 [n][3] = (byte offset of image i, H, W).  One launch pair per 24 images instead of one
 * per image: at ~10k images/s the per-image form is bound by the submitting host thread, not by the device.
 * tmp: device workspace of >= sum_i H_i * W_resized_i * 3 bytes (+ 64 per image); out: fp32 [n, 3, crop, crop].
 * Bit-identical to n calls of gitmi_preprocess_image. */
int  gitmi_preprocess_batch(const uint8_t* rgb, size_t rgb_bytes, const int64_t* desc_host, int n, int crop, uint8_t* tmp,
                            size_t tmp_bytes, float* out, void* stream);
/* same arithmetic for MinMaxResizeForTest (inference.py:29-64, the test_respect_ratio_max models): resize to
 * out_h x out_w (the caller applies get_size()), no crop, ToTensor, Normalize -> fp32 [3, out_h, out_w].
 * tmp: uint8 workspace of H * out_w * 3 bytes. */
int  gitmi_preprocess_image_to(const uint8_t* rgb_hwc, int H, int W, int out_h, int out_w, uint8_t* tmp,
                               size_t tmp_bytes, float* out_chw, void* stream);

/* one step of the sampling branch of GeneratorWithBeamSearch.search on caller-supplied fp32 logits [R, V]
 * (decoder.py:1146-1166, 1343-1375): filtered_out (optional) receives top_k_top_p_filtering(logits / temperature,
 * min_tokens_to_keep = 2) with -inf for removed tokens; draw_token / draw_logprob [R, ndraw] the draws (without
 * replacement, in draw order) and their log-probabilities under the filtered softmax.  Synchronises the stream. */
int  gitmi_op_sample_rows(const float* logits, int R, int V, float temperature, int top_k, float top_p, int ndraw,
                          uint64_t seed, int step, float* draw_logprob, int* draw_token, float* filtered_out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GITMI_H_ */
