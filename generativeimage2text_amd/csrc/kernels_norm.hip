// Row-wise / elementwise kernels of the GIT engine (HBM-bound, one wave per row):
//   LayerNorm                      CLIP/model.py:161-168, decoder.py:35,60, modeling_bert.py:168,241
//   patch extraction (im2col)      CLIP/model.py:242  (conv k=s=p, no bias == per-patch dot product)
//   class token + positional add + ln_pre        CLIP/model.py:252-257
//   word + position embedding + LayerNorm(1e-8)  decoder.py:65-78
//   f32 -> compute-dtype weight repack
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

// Output-row remap: out_row = (row / n_in) * n_out + off + row % n_in.
// Used to scatter per-frame encoder outputs into the concatenated [B, F*N, D] visual
// feature tensor (decoder.py:847-851) without a separate torch.cat pass.
struct RowMap {
    int n_in, n_out, off;
};
__device__ __forceinline__ size_t map_row(const RowMap& m, int row) {
    return (size_t)(row / m.n_in) * m.n_out + m.off + row % m.n_in;
}

constexpr int LN_MAXV = 16;   // supports D <= 1024 with one wave per row

// y = (x - mean) * rsqrt(var + eps) * gamma + beta (+ add_after[D]);  statistics in fp32,
// biased variance from centred values (matches at::native layer_norm numerics class).
// TS = element type of the residual stream: x and the optional y_f copy (float, or f16_t in bf16 engine mode)
template <typename TOut, typename TS = float>
__global__ __launch_bounds__(256) void layernorm_kernel(const TS* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ add_after,
                                                        TOut* __restrict__ y_t, int ld_t,
                                                        TS* __restrict__ y_f, int ld_f, int rows, int D,
                                                        RowMap map) {
    // one wave per row, 16-byte accesses: lane handles columns (lane + 64*i)*4 .. +4  (D % 4 == 0, D <= 1024)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TS* xr = x + (size_t)row * ldx;
    constexpr int NV = LN_MAXV / 4;
    f32x4_t v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = c < D ? ld4s(xr + c) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const size_t orow = map_row(map, row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(gamma + c);
            const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(beta + c);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * g4[r] + b4[r];
            if (add_after) {
                const f32x4_t a4 = *reinterpret_cast<const f32x4_t*>(add_after + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += a4[r];
            }
            if (y_t) {
                if constexpr (sizeof(TOut) == 4) {
                    *reinterpret_cast<f32x4_t*>(y_t + orow * ld_t + c) = f32x4_t{o[0], o[1], o[2], o[3]};
                } else {
                    uint2 t;
                    t.x = pack2bf(o[0], o[1]);
                    t.y = pack2bf(o[2], o[3]);
                    *reinterpret_cast<uint2*>(y_t + orow * ld_t + c) = t;
                }
            }
            if (y_f) st4s(y_f + orow * ld_f + c, f32x4_t{o[0], o[1], o[2], o[3]});
        }
    }
}

// The same LayerNorm for the wide rows of the bf16 engine mode (D = 768 / 1024: D % 256 == 0), one row per HALF wave:
// a lane owns NCH chunks of 8 consecutive elements (chunk lane32 + 32 i), i.e. 16-byte loads of fp16 stream rows and
// 16-byte bf16 stores -- the one-wave-per-row kernel above moves 8 bytes per lane and instruction and reached 2.8 TB/s on
// the 36 launches per encode + prefill pass (13.8 us each for 12608 x 768; profiles/r03_a_solo_graph_kernel_stats.txt).
// Statistics as above (fp32, centred second pass); only the order of the partial sums differs.
template <typename TS, int NCH>
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const TS* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             const float* __restrict__ add_after, bf16_t* __restrict__ y_t,
                                                             int ld_t, TS* __restrict__ y_f, int ld_f, int rows, RowMap map) {
    constexpr int D = NCH * 256;
    const int l32 = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const TS* xr = x + (size_t)row * ldx;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (l32 + 32 * i) * 8;
        if constexpr (sizeof(TS) == 2) {
            const u32x4_t u = *reinterpret_cast<const u32x4_t*>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) unpack2h(u[e], v[i][2 * e], v[i][2 * e + 1]);
        } else {
            ld8(reinterpret_cast<const float*>(xr) + c, v[i]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)D + eps);
    const size_t orow = map_row(map, row);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (l32 + 32 * i) * 8;
        float g8[8], b8[8], o[8];
        ld8(gamma + c, g8);
        ld8(beta + c, b8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g8[e] + b8[e];
        if (add_after) {
            float a8[8];
            ld8(add_after + c, a8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += a8[e];
        }
        if (y_t) st8(y_t + orow * ld_t + c, o);
        if (y_f) {
            if constexpr (sizeof(TS) == 2) {
                u32x4_t u;
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = pack2h(o[2 * e], o[2 * e + 1]);
                *reinterpret_cast<u32x4_t*>(y_f + orow * ld_f + c) = u;
            } else {
                st8(reinterpret_cast<float*>(y_f) + orow * ld_f + c, o);
            }
        }
    }
}

// bf16 operand rows out, D = 768 / 1024, 16-byte aligned rows: the wide kernel; false: not applicable
template <typename TS>
static bool launch_layernorm_wide(const TS* x, int ldx, const float* gamma, const float* beta, float eps, const float* add_after,
                                  bf16_t* y_t, int ld_t, TS* y_f, int ld_f, int rows, int D, RowMap m, hipStream_t s) {
    if ((D != 768 && D != 1024) || (ldx & 7) || (ld_t & 7) || (y_f && (ld_f & 7))) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y_t) & 15) || (reinterpret_cast<uintptr_t>(y_f) & 15) ||
        (reinterpret_cast<uintptr_t>(gamma) & 15) || (reinterpret_cast<uintptr_t>(beta) & 15) ||
        (reinterpret_cast<uintptr_t>(add_after) & 15))
        return false;
    dim3 grid((rows + 7) / 8), block(256);
    if (D == 768) hipLaunchKernelGGL((layernorm_wide_kernel<TS, 3>), grid, block, 0, s, x, ldx, gamma, beta, eps, add_after, y_t, ld_t, y_f, ld_f, rows, m);
    else hipLaunchKernelGGL((layernorm_wide_kernel<TS, 4>), grid, block, 0, s, x, ldx, gamma, beta, eps, add_after, y_t, ld_t, y_f, ld_f, rows, m);
    return true;
}

// patches[(b*gh*gw + gy*gw + gx), c*p*p + ky*p + kx] = img[b, c, gy*p + ky, gx*p + kx]; zero pad to Kpad.
// img is [B, C, H, W]; gh = H / p, gw = W / p (floor): like the stride-p convolution of CLIP/model.py:242 the
// H % p bottom rows and W % p right columns are never read.
template <typename TOut>
__global__ void im2col_kernel(const float* __restrict__ img, TOut* __restrict__ out, int B, int C, int H, int W,
                              int p, int gh, int gw, int K, int Kpad) {
    const size_t total = (size_t)B * gh * gw * Kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const size_t prow = i / Kpad;
        float val = 0.f;
        if (k < K) {
            const int kx = k % p, ky = (k / p) % p, c = k / (p * p);
            const int gx = (int)(prow % gw), gy = (int)((prow / gw) % gh), b = (int)(prow / ((size_t)gh * gw));
            val = img[(((size_t)b * C + c) * H + gy * p + ky) * W + gx * p + kx];
        }
        st<TOut>(out + i, val);
    }
}

// The same for 16-pixel patches and bf16 output (ViT-B/16, the benchmark path), eight consecutive kx per thread: two
// 16-byte reads of an image row, one 16-byte store.  Thread order = output order, so stores are row-contiguous.
__global__ __launch_bounds__(256) void im2col_p16_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C,
                                                         int H, int W, int gh, int gw, int Kpad) {
    const int K8 = (C * 256) >> 3;                                   // 8-element groups per patch row (K = C*16*16)
    const size_t total = (size_t)B * gh * gw * K8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % K8);
        const size_t prow = i / K8;
        const int kx0 = (k8 & 1) * 8, ky = (k8 >> 1) & 15, c = k8 >> 5;
        const int gx = (int)(prow % gw), gy = (int)((prow / gw) % gh), b = (int)(prow / ((size_t)gh * gw));
        const float* src = img + (((size_t)b * C + c) * H + gy * 16 + ky) * W + gx * 16 + kx0;
        float v[8];
        ld8(src, v);
        st8(out + prow * Kpad + (size_t)k8 * 8, v);
    }
}

// Positional embedding for a gh x gw token grid from the stored g x g one (CLIP/model.py:243-251:
// F.interpolate(mode='bicubic', align_corners=False)): source coordinate (o + 0.5) * in/out - 0.5, cubic
// convolution taps (A = -0.75) at floor-1 .. floor+2 with border-clamped indices, x first, then y; the class
// row 0 is copied.  pos [g*g + 1, D] -> out [gh*gw + 1, D].  One thread per output element.
__device__ __forceinline__ void cubic_taps(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, u = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ void pos_bicubic_kernel(const float* __restrict__ pos, float* __restrict__ out, int g, int gh, int gw, int D) {
    const size_t total = ((size_t)gh * gw + 1) * D;
    const float sy = (float)g / (float)gh, sx = (float)g / (float)gw;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int n = (int)(i / D);
        if (n == 0) { out[i] = pos[c]; continue; }
        const int oy = (n - 1) / gw, ox = (n - 1) % gw;
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float y0f = floorf(fy), x0f = floorf(fx);
        float wy[4], wx[4];
        cubic_taps(fy - y0f, wy);
        cubic_taps(fx - x0f, wx);
        const int y0 = (int)y0f, x0 = (int)x0f;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = y0 - 1 + a;
            yy = yy < 0 ? 0 : (yy > g - 1 ? g - 1 : yy);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int xx = x0 - 1 + b;
                xx = xx < 0 ? 0 : (xx > g - 1 ? g - 1 : xx);
                row += wx[b] * pos[((size_t)1 + (size_t)yy * g + xx) * D + c];
            }
            acc += wy[a] * row;
        }
        out[i] = acc;
    }
}

// token row (b, n): n == 0 ? class_embedding : patch_out[b*g2 + n - 1];  + positional[n];  ln_pre.
// Output: fp32 residual stream X[B*N, D].
template <typename TS>
__global__ __launch_bounds__(256) void vit_assemble_ln_kernel(const float* __restrict__ patch_out,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              TS* __restrict__ X, int B, int N, int D,
                                                              float2* __restrict__ part, int nparts) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * N) return;
    const int b = row / N, n = row % N;
    const float* src = n == 0 ? cls : patch_out + ((size_t)b * (N - 1) + (n - 1)) * D;
    const float* pr = pos + (size_t)n * D;
    float v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? src[c] + pr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) st<TS>(X + (size_t)row * D + c, (v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
    // folded LayerNorm of the first block (kernels_gemm10.hip LNF): (sum, sumsq) of the row AS STORED in slot 0 of the row's
    // four column-tile slots, zeros in the others
    if (part) {
        float sx = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float o = (float)(TS)((v[i] - mean) * rstd * gamma[c] + beta[c]);
                sx += o;
                sq = fmaf(o, o, sq);
            }
        }
        sx = wave_sum(sx);
        sq = wave_sum(sq);
        if (lane < 4) part[(size_t)row * 4 + lane] = lane == 0 ? float2{sx, sq} : float2{0.f, 0.f};
    }
}

// x = words[ids[r, pos]] + positions[pos];  LayerNorm(eps);  -> fp32 hidden + compute-dtype hidden.
// One 256-thread workgroup per row, one float4 per thread (D <= 1024, D % 4 == 0): a single round of loads.
template <typename TOut>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ ids, int ld_ids, int pos,
                                                       const float* __restrict__ words,
                                                       const float* __restrict__ positions,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps,
                                                       float* __restrict__ h_f, TOut* __restrict__ h_t, int R,
                                                       int D, int vocab, int frag) {
    __shared__ float s_part[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    int tok = ids[(size_t)row * ld_ids + pos];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const int c = tid * 4;
    const bool on = c < D;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, g4 = a, b4 = a;
    if (on) {
        const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(words + (size_t)tok * D + c);
        const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(positions + (size_t)pos * D + c);
        g4 = *reinterpret_cast<const f32x4_t*>(gamma + c);
        b4 = *reinterpret_cast<const f32x4_t*>(beta + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = w4[r] + p4[r];
    }
    float sum = wave_sum(a[0] + a[1] + a[2] + a[3]);
    if (lane == 0) s_part[wave] = sum;
    __syncthreads();
    const float mean = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)D;
    float q = 0.f;
    if (on) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = a[r] - mean; q += d * d; }
    }
    q = wave_sum(q);
    if (lane == 0) s_part[4 + wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((s_part[4] + s_part[5] + s_part[6] + s_part[7]) / (float)D + eps);
    if (on) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (a[r] - mean) * rstd * g4[r] + b4[r];
        *reinterpret_cast<f32x4_t*>(h_f + (size_t)row * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
        if constexpr (sizeof(TOut) == 4) {
            *reinterpret_cast<f32x4_t*>(h_t + (size_t)row * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
        } else {
            uint2 t;
            t.x = pack2bf(o[0], o[1]);
            t.y = pack2bf(o[2], o[3]);
            // operand of the decode chain's first GEMM: fragment-major (kernels_dgemm.hip)
            const size_t off = frag ? frag_offset(row, c, D >> 5) : (size_t)row * D + c;
            *reinterpret_cast<uint2*>(h_t + off) = t;
        }
    }
}

// dst[r, 0..Kpad) = convert(src[r, 0..K)), zero padded
template <typename TOut>
__global__ void convert_pad_kernel(const float* __restrict__ src, TOut* __restrict__ dst, size_t rows, int K,
                                   int Kpad) {
    const size_t total = rows * (size_t)Kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const size_t r = i / Kpad;
        st<TOut>(dst + i, k < K ? src[r * K + k] : 0.f);
    }
}

// element-wise dtype conversion (debug / attribution hooks: tensors handed from an fp32 context to a bf16 one and back)
template <typename TIn, typename TOut>
__global__ void convert_any_kernel(const TIn* __restrict__ src, TOut* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        st<TOut>(dst + i, ld<TIn>(src + i));
}

// fp32 rows x [R, d] -> what a decode-chain N = d GEMM leaves for its consumer (kernels_dgemm.hip): the bf16 copy in the
// fragment-major operand layout and the (sum, sum of squares) partials of every 16-column strip, [d/16][R]
__global__ void chain_input_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb, float2* __restrict__ stats, int R, int d) {
    const int strips = d >> 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * strips; i += gridDim.x * blockDim.x) {
        const int r = i / strips, sp = i % strips;
        float sum = 0.f, sq = 0.f;
        for (int c = 0; c < 16; ++c) {
            const float v = x[(size_t)r * d + sp * 16 + c];
            sum += v; sq += v * v;
            xb[frag_offset(r, sp * 16 + c, d >> 5)] = f2bf(v);
        }
        stats[(size_t)sp * R + r] = float2{sum, sq};
    }
}

__global__ void copy_f32_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols, c = i % cols;
        dst[r * ldd + c] = src[r * lds + c];
    }
}

// row-major bf16 [rows, K] -> fragment-major [ceil16(rows_out), K] (gitmi_common.h frag_offset); rows >= `rows` are zero
__global__ void frag_pack_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int rows_out, int K) {
    const int ksteps = K >> 5;
    const size_t total = (size_t)rows_out * (K >> 3);                 // 8-element (16-byte) groups
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // destination-linear index: tile, lane, so that writes are contiguous
        const size_t tile = i >> 6;
        const int lane = (int)(i & 63);
        const int rt = (int)(tile / ksteps), ks = (int)(tile % ksteps);
        const int row = rt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (row < rows) v = *reinterpret_cast<const u32x4_t*>(src + (size_t)row * K + k);
        *reinterpret_cast<u32x4_t*>(dst + i * 8) = v;
    }
}

__global__ void fill_i32_kernel(int* __restrict__ dst, int value, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = value;
}

// ---- host launchers ------------------------------------------------------------------
static inline int grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    return (int)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}

hipError_t launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                            const float* add_after, void* y_t, int ld_t, bool t_is_f32, float* y_f, int ld_f,
                            int rows, int D, int map_n_in, int map_n_out, int map_off, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (D > 64 * LN_MAXV || (D & 3) || (ldx & 3) || (ld_t & 3) || (y_f && (ld_f & 3))) return hipErrorInvalidValue;
    RowMap m{map_n_in > 0 ? map_n_in : rows, map_n_in > 0 ? map_n_out : rows, map_off};
    dim3 grid((rows + 3) / 4), block(256);
    if (t_is_f32)
        hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, x, ldx, gamma, beta, eps, add_after,
                           (float*)y_t, ld_t, y_f, ld_f, rows, D, m);
    else if (y_t && launch_layernorm_wide<float>(x, ldx, gamma, beta, eps, add_after, (bf16_t*)y_t, ld_t, y_f, ld_f, rows, D, m, s))
        ;
    else
        hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, s, x, ldx, gamma, beta, eps, add_after,
                           (bf16_t*)y_t, ld_t, y_f, ld_f, rows, D, m);
    return hipGetLastError();
}

hipError_t launch_im2col(const float* img, void* out, bool out_f32, int B, int H, int W, int p, int K, int Kpad,
                         hipStream_t s) {
    const int gh = H / p, gw = W / p;
    const size_t total = (size_t)B * gh * gw * Kpad;
    if (out_f32)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, img, (float*)out, B,
                           3, H, W, p, gh, gw, K, Kpad);
    else if (p == 16 && K == 3 * 256 && Kpad == K && (W & 3) == 0 && ((uintptr_t)img & 15) == 0)
        hipLaunchKernelGGL(im2col_p16_kernel, dim3(grid_for(total / 8, 256)), dim3(256), 0, s, img, (bf16_t*)out, B, 3, H, W,
                           gh, gw, Kpad);
    else
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, img, (bf16_t*)out,
                           B, 3, H, W, p, gh, gw, K, Kpad);
    return hipGetLastError();
}

hipError_t launch_pos_bicubic(const float* pos, float* out, int g, int gh, int gw, int D, hipStream_t s) {
    const size_t total = ((size_t)gh * gw + 1) * D;
    hipLaunchKernelGGL(pos_bicubic_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, pos, out, g, gh, gw, D);
    return hipGetLastError();
}

hipError_t launch_vit_assemble_ln(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                                  const float* beta, float eps, void* X, bool x_f16, int B, int N, int D, float2* part,
                                  int nparts, hipStream_t s) {
    if (D > 64 * LN_MAXV || (part && (!x_f16 || nparts < 1 || nparts > 64))) return hipErrorInvalidValue;
    if (x_f16)
        hipLaunchKernelGGL(vit_assemble_ln_kernel<f16_t>, dim3((B * N + 3) / 4), dim3(256), 0, s, patch_out, cls, pos, gamma,
                           beta, eps, (f16_t*)X, B, N, D, part, nparts);
    else
        hipLaunchKernelGGL(vit_assemble_ln_kernel<float>, dim3((B * N + 3) / 4), dim3(256), 0, s, patch_out, cls, pos, gamma,
                           beta, eps, (float*)X, B, N, D, (float2*)nullptr, 0);
    return hipGetLastError();
}

// LayerNorm over a residual stream stored in fp16 (bf16 engine mode): x and the optional stream copy y_s are f16_t,
// the operand copy y_t is bf16 (t_is_f32: fp32 -- the parity hook that hands the features back); add_after / row remap as in
// launch_layernorm.
hipError_t launch_layernorm_s16(const void* x, int ldx, const float* gamma, const float* beta, float eps,
                                const float* add_after, void* y_t, int ld_t, bool t_is_f32, void* y_s, int ld_s,
                                int rows, int D, int map_n_in, int map_n_out, int map_off, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (D > 64 * LN_MAXV || (D & 3) || (ldx & 3) || (ld_t & 3) || (y_s && (ld_s & 3))) return hipErrorInvalidValue;
    RowMap m{map_n_in > 0 ? map_n_in : rows, map_n_in > 0 ? map_n_out : rows, map_off};
    dim3 grid((rows + 3) / 4), block(256);
    if (t_is_f32)
        hipLaunchKernelGGL((layernorm_kernel<float, f16_t>), grid, block, 0, s, (const f16_t*)x, ldx, gamma, beta, eps,
                           add_after, (float*)y_t, ld_t, (f16_t*)y_s, ld_s, rows, D, m);
    else if (y_t && launch_layernorm_wide<f16_t>((const f16_t*)x, ldx, gamma, beta, eps, add_after, (bf16_t*)y_t, ld_t, (f16_t*)y_s, ld_s,
                                                 rows, D, m, s))
        ;
    else
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, f16_t>), grid, block, 0, s, (const f16_t*)x, ldx, gamma, beta, eps,
                           add_after, (bf16_t*)y_t, ld_t, (f16_t*)y_s, ld_s, rows, D, m);
    return hipGetLastError();
}

hipError_t launch_embed_ln(const int* ids, int ld_ids, int pos, const float* words, const float* positions,
                           const float* gamma, const float* beta, float eps, float* h_f, void* h_t, bool t_is_f32,
                           int R, int D, int vocab, bool frag, hipStream_t s) {
    if (D > 1024 || (D & 3) || (frag && (t_is_f32 || (D & 31)))) return hipErrorInvalidValue;
    dim3 grid(R), block(256);
    if (t_is_f32)
        hipLaunchKernelGGL(embed_ln_kernel<float>, grid, block, 0, s, ids, ld_ids, pos, words, positions, gamma,
                           beta, eps, h_f, (float*)h_t, R, D, vocab, 0);
    else
        hipLaunchKernelGGL(embed_ln_kernel<bf16_t>, grid, block, 0, s, ids, ld_ids, pos, words, positions, gamma,
                           beta, eps, h_f, (bf16_t*)h_t, R, D, vocab, frag ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_frag_pack(const void* src_bf16, void* dst_bf16, int rows, int rows_out, int K, hipStream_t s) {
    if (K % 32 || rows_out % 16 || rows_out < rows) return hipErrorInvalidValue;
    const size_t total = (size_t)rows_out * (K >> 3);
    hipLaunchKernelGGL(frag_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const bf16_t*)src_bf16,
                       (bf16_t*)dst_bf16, rows, rows_out, K);
    return hipGetLastError();
}

hipError_t launch_convert_pad(const float* src, void* dst, bool dst_f32, size_t rows, int K, int Kpad,
                              hipStream_t s) {
    const size_t total = rows * (size_t)Kpad;
    if (total == 0) return hipSuccess;
    if (dst_f32)
        hipLaunchKernelGGL(convert_pad_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, (float*)dst,
                           rows, K, Kpad);
    else
        hipLaunchKernelGGL(convert_pad_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, src,
                           (bf16_t*)dst, rows, K, Kpad);
    return hipGetLastError();
}

hipError_t launch_convert(const void* src, bool src_f32, void* dst, bool dst_f32, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const dim3 g(grid_for(n, 256)), b(256);
    if (src_f32 && dst_f32) hipLaunchKernelGGL((convert_any_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
    else if (src_f32) hipLaunchKernelGGL((convert_any_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (dst_f32) hipLaunchKernelGGL((convert_any_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n);
    else hipLaunchKernelGGL((convert_any_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    return hipGetLastError();
}

hipError_t launch_chain_input(const float* x, void* xb, float2* stats, int R, int d, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if (d % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_input_kernel, dim3(grid_for((size_t)R * (d >> 4), 256)), dim3(256), 0, s, x, (bf16_t*)xb, stats, R, d);
    return hipGetLastError();
}

hipError_t launch_fill_i32(int* dst, int value, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
}

hipError_t launch_copy_f32(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    return hipGetLastError();
}

}  // namespace gitmi
