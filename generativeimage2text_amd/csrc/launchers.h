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
    int ng;     // XCD tile partition: N split into ng groups, M into 8/ng (kernels_gemm3.hip)
    int dbg;    // timing experiments only (gemm_ring_kernel): 1 no C stores, 2 no epilogue, 4 no MFMA, 8 no loads in loop
};
hipError_t launch_gemm(const GemmArgs& g, bool in_f32, bool out_f32, hipStream_t s);
// second-generation bf16 kernel (direct-to-LDS staging, swizzled LDS, LDS-staged epilogue)
bool gemm_dlds_supported(const GemmArgs& g, bool in_f32, bool out_f32);
hipError_t launch_gemm_dlds(GemmArgs g, bool out_f32, hipStream_t s);
hipError_t launch_gemm_ring(GemmArgs g, bool out_f32, hipStream_t s);   // 256x128 tile, 3-stage ring
hipError_t launch_gemm_ring256(GemmArgs g, bool out_f32, hipStream_t s); // 256x256x32, 8 waves of 128x64, 4-stage ring
hipError_t launch_gemm_p8(GemmArgs g, bool out_f32, hipStream_t s);     // 256x256x64, half-tile pipeline, staggered wave groups
bool gemm_p8_supports(const GemmArgs& g);
int gemm_p8_cost(const GemmArgs& g, int mh);   // rounds x relative tile time of the 256-row (mh=128) / 192-row (96) tile
void set_gemm_impl(int impl);   // -1 auto, 0 first-generation kernel only, 1 force direct-to-LDS kernel

// weight-streaming GEMM for decode (bf16 operands): C or fp32 partial slabs [S][M][N]
struct SkinnyArgs {
    const void* A; const void* W; const float* bias; const float* res; void* C; float* partial;
    int M, N, K;
    int lda, ldc, ldr;
    int act;
    int S;      // K slices across workgroups (1 = fused epilogue, >1 = partial slabs)
};
hipError_t launch_skinny_gemm(SkinnyArgs g, bool out_f32, int NT, hipStream_t s);
hipError_t launch_splitk_ln(const float* partial, int S, const float* bias, const float* res, const float* gamma,
                            const float* beta, float eps, float* y_f, void* y_t, int rows, int D, hipStream_t s);

hipError_t launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                            const float* add_after, void* y_t, int ld_t, bool t_is_f32, float* y_f, int ld_f,
                            int rows, int D, int map_n_in, int map_n_out, int map_off, hipStream_t s);
hipError_t launch_im2col(const float* img, void* out, bool out_f32, int B, int H, int W, int p, int K, int Kpad,
                         hipStream_t s);
hipError_t launch_pos_bicubic(const float* pos, float* out, int g, int gh, int gw, int D, hipStream_t s);
hipError_t launch_vit_assemble_ln(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                                  const float* beta, float eps, float* X, int B, int N, int D, hipStream_t s);
hipError_t launch_embed_ln(const int* ids, int ld_ids, int pos, const float* words, const float* positions,
                           const float* gamma, const float* beta, float eps, float* h_f, void* h_t, bool t_is_f32,
                           int R, int D, int vocab, hipStream_t s);
hipError_t launch_convert_pad(const float* src, void* dst, bool dst_f32, size_t rows, int K, int Kpad,
                              hipStream_t s);
hipError_t launch_copy_f32(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);

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
    int ld_src;
    int d;
    int N_img, T_max, pos, beams;
    float scale;
    int dbg;             // timing experiments: 1 skip image K/V loads, 2 skip scores, 4 skip PV
};
hipError_t launch_kv_repack(const void* qkv, void* kh, void* vh, int B, int N, int H, int d, bool is_f32, hipStream_t s);
size_t attn_decode_lds_bytes(int beams, int N_img, int pos);
hipError_t launch_attn_decode(const AttnDecodeArgs& a, int B, int H, bool is_f32, hipStream_t s);
hipError_t attn_decode_configure();

struct SearchState {
    int B, k, pn, P, T, V, eos, kind;
    double length_penalty;
    int* ids[2];
    int* kv_src[2];
    float* score[2];
    float* cand_val;
    int* cand_idx;
    int* done;
    int* hyp_n;
    double* hyp_score;
    int* hyp_len;
    int* hyp_tok;
    int* info;
};
hipError_t launch_row_topm(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len, int eos,
                           int suppress_last, int force_eos, int M, int R, float* cand_val, int* cand_idx,
                           hipStream_t s);
hipError_t launch_s1_advance(const SearchState& st, int src, int cur_len, int first, int M, hipStream_t s);
hipError_t launch_s2_advance(const SearchState& st, int src, int cur_len, int M, hipStream_t s);
hipError_t launch_search_init(const SearchState& st, const long long* start_dev, hipStream_t s);
hipError_t launch_search_finish(const SearchState& st, int cur, int cur_len, long long* tokens_out,
                                float* logprob_out, int* info_out, hipStream_t s);
hipError_t launch_search_rows(const SearchState& st, int cur, int cur_len, long long* out, hipStream_t s);
hipError_t launch_fill_start(long long* start, const long long* prefix, int sos, int B, int P, hipStream_t s);
hipError_t launch_load_ids(const long long* tokens, int R, int t, int* ids, int* kv_src, int ld, hipStream_t s);

// GPU image transform (Pillow-exact bicubic resize + centre crop + CLIP normalisation)
hipError_t launch_preprocess(const unsigned char* rgb, int H, int W, int crop, unsigned char* tmp, float* out, hipStream_t s);
hipError_t launch_resize_crop_norm(const uint8_t* rgb, int H, int W, int nh, int nw, int top, int left, int ch, int cw,
                                   uint8_t* tmp, float* out, hipStream_t s);

}  // namespace gitmi
