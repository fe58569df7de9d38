/* gitmi_experiment.h -- entry points of the MEASUREMENT build only (libgitmi_exp.so: the product sources compiled with
 * -DGITMI_EXPERIMENT, `make exp`).  libgitmi.so / libgitmi_f16.so do not export them.
 *
 * What lives here: debug hooks that exchange stage products between contexts (tools/error_attribution.py), force a GEMM
 * variant (tools/gemm_bench.py, tests of the forced tile heights) or set the timing bits of the decode-chain GEMMs
 * (tools/dgemm_bench.py).  The measurement build also reads GITMI_* environment overrides at gitmi_create (kernel shapes,
 * work-skipping switches for timing decompositions); the product libraries read no environment.
 *
 * Removed in round 5 (history and docs/LAB_NOTEBOOK.md keep them): the two-submission split of a call
 * (gitmi_generate_encode / gitmi_generate_decode, bench.py --phased) and decode groups (gitmi_clone_sized,
 * gitmi_set_decode_group, gitmi_group_decode) -- bit-identical to gitmi_generate and measured slower than the default mixed
 * schedule three rounds running (9.1-10.6k against 10.9k captions/s). */
#ifndef GITMI_EXPERIMENT_H_
#define GITMI_EXPERIMENT_H_
#include "gitmi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- error attribution hooks (tools/error_attribution.py; not part of the serving path).  Two contexts of the SAME
 * model in different precisions: import_stage hands the products of the image encoder (stage 1) or of encoder + prefill
 * (stage 2: the image K/V of every decoder layer) from `src` to `dst`, converted, so that gitmi_step_logits on `dst`
 * continues from there; head_from applies dst's (bf16, fused) vocabulary head to the last hidden state of src's (fp32)
 * most recent gitmi_step_logits over R rows -> logits_out fp32 [R, vocab] on the device. */
int  gitmi_debug_import_stage(gitmi_engine* dst, gitmi_engine* src, int stage, void* stream);
int  gitmi_debug_head_from(gitmi_engine* dst, gitmi_engine* src, int R, float* logits_out, void* stream);

/* encoder-GEMM selection for A/B measurements (process-wide): low byte -1 auto (default) | 0 register-staged tile kernel
 * only | 9 the LDS-DMA kernel wherever its shape rules hold; bits 8.. = variant / timing bits of the selected kernel
 * (tools/gemm_bench.py lists them: 64 / 128 / 16384 / 32768 / 65536 force the 192- / 256- / 160- / 224- / 128-row tile ...).
 * Any other low byte is refused (returns non-zero, selection unchanged). */
int  gitmi_debug_set_gemm_impl(int impl);

/* timing bits of the decode-chain GEMMs (kernels_dgemm.hip; tools/dgemm_bench.py) for gitmi_op_dgemm / gitmi_op_dgemm_res */
int  gitmi_debug_set_dgemm(int dbg);

#ifdef __cplusplus
}
#endif
#endif /* GITMI_EXPERIMENT_H_ */
