/* gitmi_experiment.h -- entry points of the MEASUREMENT build only (libgitmi_exp.so: the product sources compiled with
 * -DGITMI_EXPERIMENT).  libgitmi.so / libgitmi_f16.so do not export them.
 *
 * What lives here and why:
 *   - schedules that were built, are bit-identical to gitmi_generate and MEASURED SLOWER than the default mixed schedule on
 *     MI355X: the two-submission split of a call (bench.py --phased) and decode groups
 *     (bench.py --decode-group; rounds 3 and 4: 9.7-10.6k against 10.1-10.9k captions/s).  They stay buildable and tested so
 *     that the measurement can be repeated, not as product API;
 *   - debug hooks that exchange stage products between contexts (tools/error_attribution.py) or force a GEMM variant
 *     (tools/gemm_bench.py, tests of the forced tile heights).
 * The measurement build also reads GITMI_* environment overrides at gitmi_create (kernel shapes, work-skipping switches for
 * timing decompositions); the product libraries read no environment. */
#ifndef GITMI_EXPERIMENT_H_
#define GITMI_EXPERIMENT_H_
#include "gitmi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* gitmi_generate as TWO submissions with the same arguments: the image encoder + decoder prefill of the call, then its
 * search over the text positions + results.  The reference has no such seam (one model(batch) call does both,
 * decoder.py:838-877, 977-1011); it exists for servers that order the halves of several contexts themselves -- e.g. the
 * MFMA-bound encoders of a group of batches first, their latency-bound decode chains side by side afterwards
 * (bench.py --phased).  gitmi_generate_decode must follow the gitmi_generate_encode of the same call on the same context;
 * results are identical to one gitmi_generate call. */
int  gitmi_generate_encode(gitmi_engine* e, const float* const* frames, int F, int B,
                           const int64_t* prefix, int P, const gitmi_search* search, void* stream);
int  gitmi_generate_decode(gitmi_engine* e, int F, int B, const int64_t* prefix, int P, const gitmi_search* search,
                           int64_t* tokens_out, float* logprob_out, int32_t* info_out, void* stream);

/* ---- decode groups (ABI 6): ONE decode chain for the requests of several contexts.
 * The reference decodes every batch on its own (decoder.py:313-417); on the device the decode chain of a 64-image batch
 * is 19 x 32 dependent, latency-bound launches that read every decoder weight once per step, whatever the row count.
 * A GROUP context (gitmi_clone_sized with max_batch = the sum of its members') owns the image K/V cache; MEMBER contexts
 * (gitmi_set_decode_group) run image encoder + prefill of their own requests as before -- each on its stream, as soon as
 * its request arrives -- and write their K/V into the group's cache at `image_offset`; gitmi_group_decode then searches
 * over the first B images of the cache in one chain: rows = the members' rows, weights streamed once per step, a quarter
 * / half of the launches per caption.  Captions do not depend on their batch neighbours, so every request gets exactly
 * what its own gitmi_generate call returns (same kernels, same per-row arithmetic).
 *   ordering is the engine's: a member's K/V repack (a small graph of its own behind its prefill) waits for the group's
 *   previous gitmi_group_decode, and gitmi_group_decode waits for the members covering images [0, B).  The caller only
 *   keeps HOST order: the members' gitmi_generate_encode calls of a round, then the group's decode, then the next round
 *   (a member's next gitmi_generate_encode before a gitmi_group_decode covering its previous request was submitted is
 *   refused: no event could order it).
 *   A member accepts gitmi_generate_encode only (search / prefix arguments as for the group's decode).
 * gitmi_set_decode_group(member, NULL, 0) detaches; destroying either context removes the link. */
int  gitmi_clone_sized(gitmi_engine* src, int max_batch, gitmi_engine** out);
int  gitmi_set_decode_group(gitmi_engine* member, gitmi_engine* group, int image_offset);
int  gitmi_group_decode(gitmi_engine* group, int F, int B, const int64_t* prefix, int P, const gitmi_search* search,
                        int64_t* tokens_out, float* logprob_out, int32_t* info_out, void* stream);

/* ---- error attribution hooks (tools/error_attribution.py; not part of the serving path).  Two contexts of the SAME
 * model in different precisions: import_stage hands the products of the image encoder (stage 1) or of encoder + prefill
 * (stage 2: the image K/V of every decoder layer) from `src` to `dst`, converted, so that gitmi_step_logits on `dst`
 * continues from there; head_from applies dst's (bf16, fused) vocabulary head to the last hidden state of src's (fp32)
 * most recent gitmi_step_logits over R rows -> logits_out fp32 [R, vocab] on the device. */
int  gitmi_debug_import_stage(gitmi_engine* dst, gitmi_engine* src, int stage, void* stream);
int  gitmi_debug_head_from(gitmi_engine* dst, gitmi_engine* src, int R, float* logits_out, void* stream);

/* kernel selection for A/B measurements: -1 auto (default), 0 first-generation GEMM only,
 * 1 force the direct-to-LDS GEMM wherever its constraints hold */
int  gitmi_debug_set_gemm_impl(int impl);

#ifdef __cplusplus
}
#endif
#endif /* GITMI_EXPERIMENT_H_ */
