// GEMM kernels for gfx950 (MI355X):  C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])
//
// Replaces every nn.Linear / the patch-embed conv of the reference hot path
// (CLIP/model.py:179-185,242; modeling_bert.py:105-107,165,222,237; decoder.py:32,503):
// in the reference these are implicit cuBLAS/cuDNN calls, here one MFMA kernel family
// with the bias / QuickGELU / erf-GELU / residual epilogues fused.
//
// Layout: A (activations) and W (nn.Linear weight) are both K-contiguous, which is exactly
// what the MFMA operands want.  The MFMA is issued "swapped": operand A of the instruction
// is a 16-row slice of W, operand B a 16-row slice of the activations, so the accumulator
// tile is C^T[n][m]: each lane ends up with 4 CONSECUTIVE n for one m and the epilogue
// stores 8/16 contiguous bytes per lane.
//
//   bf16 path : v_mfma_f32_16x16x32_bf16, BK = 64, fp32 accumulate.
//   f32  path : v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain), BK = 16.
//
// Block = 256 threads = 4 waves (2 along M x 2 along N), tile BM x BN (128x128 default,
// 64-row variants for skinny decode GEMMs), register-staged global->LDS prefetch of the
// next K tile while the current one feeds the MFMAs, padded LDS rows (conflict-free
// ds_read_b128), XCD-aware block-id remap so that blocks sharing an activation panel
// sit on one XCD's L2.
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

template <typename TIn> struct GemmTraits;
template <> struct GemmTraits<bf16_t> {
    static constexpr int BK = 64;      // elements per K tile
    static constexpr int LDK = 72;     // padded LDS row (144 B: 16 rows hit 16 distinct 16-B slots)
    static constexpr int CHUNK = 8;    // elements per 16-byte chunk
};
template <> struct GemmTraits<float> {
    static constexpr int BK = 16;
    static constexpr int LDK = 20;     // 80 B rows
    static constexpr int CHUNK = 4;
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    // bijective XCD-aware remap: blocks dispatched round-robin over 8 XCDs get contiguous tiles
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut>
__device__ __forceinline__ void store4(TOut* p, const float (&v)[4], bool vec);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float (&v)[4], bool vec) {
    if (vec) {
        f32x4_t t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(p) = t;
    } else {
        p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
    }
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4], bool vec) {
    if (vec) {
        uint2 t;
        t.x = pack2bf(v[0], v[1]);
        t.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    } else {
        p[0] = f2bf(v[0]); p[1] = f2bf(v[1]); p[2] = f2bf(v[2]); p[3] = f2bf(v[3]);
    }
}

template <>
__device__ __forceinline__ void store4<f16_t>(f16_t* p, const float (&v)[4], bool vec) {
    if (vec) {
        uint2 t;
        t.x = pack2h(v[0], v[1]);
        t.y = pack2h(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    } else {
        p[0] = (f16_t)v[0]; p[1] = (f16_t)v[1]; p[2] = (f16_t)v[2]; p[3] = (f16_t)v[3];
    }
}

// TOut = f16_t: C and the residual are rows of the fp16 residual stream (GemmArgs::out_f16)
template <typename TIn, typename TOut, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    using TR = GemmTraits<TIn>;
    constexpr int BK = TR::BK, LDK = TR::LDK, CH = TR::CHUNK;
    constexpr int CPR = BK / CH;                 // 16-B chunks per tile row (8 bf16 / 4 f32)
    constexpr int A_IT = BM * CPR / 256;         // chunks per thread for the activation tile
    constexpr int W_IT = BN * CPR / 256;
    constexpr int TM = BM / 2 / 16;              // 16x16 tiles per wave along M
    constexpr int TN = BN / 2 / 16;
    static_assert(A_IT >= 1 && W_IT >= 1, "tile too small for 256 threads");

    __shared__ __attribute__((aligned(16))) TIn smem[(BM + BN) * LDK];
    TIn* As = smem;
    TIn* Ws = smem + BM * LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int swz = xcd_remap(blockIdx.x, g.nwg);
    const int tile_n = swz % g.tiles_n;
    const int tile_m = swz / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const TIn* __restrict__ A = reinterpret_cast<const TIn*>(g.A);
    const TIn* __restrict__ W = reinterpret_cast<const TIn*>(g.W);

    // per-thread global source rows (clamped: out-of-range rows re-read the last valid row,
    // their results are never stored)
    const int c_col = (tid % CPR) * CH;
    const int c_row = tid / CPR;                 // 0 .. 256/CPR-1
    constexpr int ROWS_PER_IT = 256 / CPR;
    const TIn* a_src[A_IT];
    const TIn* w_src[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int r = m0 + c_row + i * ROWS_PER_IT;
        r = r < g.M ? r : g.M - 1;
        a_src[i] = A + (size_t)r * g.lda + c_col;
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        int r = n0 + c_row + i * ROWS_PER_IT;
        r = r < g.N ? r : g.N - 1;
        w_src[i] = W + (size_t)r * g.K + c_col;
    }

    u32x4_t a_reg[A_IT], w_reg[W_IT];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a_reg[i] = *reinterpret_cast<const u32x4_t*>(a_src[i] + k0);
#pragma unroll
        for (int i = 0; i < W_IT; ++i) w_reg[i] = *reinterpret_cast<const u32x4_t*>(w_src[i] + k0);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<u32x4_t*>(As + (c_row + i * ROWS_PER_IT) * LDK + c_col) = a_reg[i];
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            *reinterpret_cast<u32x4_t*>(Ws + (c_row + i * ROWS_PER_IT) * LDK + c_col) = w_reg[i];
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    load_tile(0);
    store_tile();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);

        if constexpr (sizeof(TIn) == 2) {
            // two 32-deep MFMA steps per 64-deep tile
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t wf[TN], af[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const bf16x8_t*>(
                        Ws + (wn * (BN / 2) + j * 16 + l15) * LDK + kk * 32 + lg * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8_t*>(
                        As + (wm * (BM / 2) + i * 16 + l15) * LDK + kk * 32 + lg * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
            }
        } else {
            // four 4-deep f32 MFMA steps per 16-deep tile (exact fp32 fmaf chain)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float wf[TN], af[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = reinterpret_cast<const float*>(Ws)[(wn * (BN / 2) + j * 16 + l15) * LDK + kk * 4 + lg];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = reinterpret_cast<const float*>(As)[(wm * (BM / 2) + i * 16 + l15) * LDK + kk * 4 + lg];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[j][i], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds C^T[n = .. + lg*4 + r][m = .. + l15] --------------------
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);
    const bool vec_c = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
    const bool vec_r = g.res != nullptr && (g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + l15;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lg * 4;
            if (n >= g.N) continue;
            const bool full = n + 3 < g.N;
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (g.bias) {
                if (full) {   // unguarded loads: hipcc keeps them in flight together
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += g.bias[n + r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.N) v[r] += g.bias[n + r];
                }
            }
            if (g.act != GITMI_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
            }
            if (g.res) {
                if constexpr (std::is_same<TOut, f16_t>::value) {
                    const f16_t* rp = reinterpret_cast<const f16_t*>(g.res) + (size_t)m * g.ldr + n;
                    if (full && vec_r) {                     // 8-byte aligned: ldr % 4 == 0 and a 16-byte aligned base
                        const f32x4_t t = ld4s(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += t[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N) v[r] += (float)rp[r];
                    }
                } else {
                    const float* rp = g.res + (size_t)m * g.ldr + n;
                    if (full && vec_r) {
                        f32x4_t t = *reinterpret_cast<const f32x4_t*>(rp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += t[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.N) v[r] += rp[r];
                    }
                }
            }
            TOut* cp = C + (size_t)m * g.ldc + n;
            if (full) {
                store4<TOut>(cp, v, vec_c);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) st<TOut>(cp + r, v[r]);
            }
        }
    }
}

// ---- host launcher -------------------------------------------------------------------
template <typename TIn, typename TOut, int BM, int BN>
static hipError_t launch_gemm_t(GemmArgs g, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.nwg = tiles_m * g.tiles_n;
    hipLaunchKernelGGL((gemm_kernel<TIn, TOut, BM, BN>), dim3(g.nwg), dim3(256), 0, s, g);
    return hipGetLastError();
}

template <typename TIn, typename TOut>
static hipError_t launch_gemm_tiles(const GemmArgs& g, hipStream_t s) {
    // skinny M (decode steps: M = rows of the beam batch): smaller tiles -> more workgroups
    // streaming the weight matrix concurrently.
    if (g.M <= 64) return launch_gemm_t<TIn, TOut, 64, 64>(g, s);
    if (g.M <= 512 || (long)((g.M + 127) / 128) * ((g.N + 127) / 128) < 256)
        return launch_gemm_t<TIn, TOut, 64, 128>(g, s);
    return launch_gemm_t<TIn, TOut, 128, 128>(g, s);
}

static int g_gemm_impl = -1;
static int g_gemm_dbg = 0;
bool set_gemm_impl(int impl) {
    int dbg = 0;
    if (impl >= 0) { dbg = impl >> 8; impl &= 0xff; }
    if (impl != -1 && impl != 0 && impl != 9) return false;      // unknown selector: refused, state unchanged
    g_gemm_dbg = dbg;
    g_gemm_impl = impl;
    return true;
}

// what the LDS-DMA kernel (kernels_gemm10.hip) needs of the operands besides its shape rules (gemm_p8_supports): bf16/fp16
// operands, 16-byte aligned rows and bases
static bool gemm_fast_operands(const GemmArgs& g, bool in_f32, bool out_f32) {
    if (in_f32) return false;
    if (g.K % 64 != 0 || g.N % 8 != 0 || g.lda % 8 != 0) return false;
    if (g.ldc % (out_f32 ? 4 : 8) != 0) return false;
    if (g.res && g.ldr % (g.out_f16 ? 8 : 4) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C) |
         reinterpret_cast<uintptr_t>(g.res)) & 15)
        return false;
    return true;
}

// in_f32/out_f32: element types of A,W and of C.  Two kernels: the 256x256x64 half-tile LDS-DMA pipeline
// (kernels_gemm10.hip) for every 16-bit GEMM with more than 512 rows that meets its shape rules -- all of the image encoder
// and the decoder prefill of every BASELINE configuration -- and the register-staged tile kernel of this file for the rest
// (fp32 parity mode, small batches, odd shapes).  The three generations in between (128x128 direct-to-LDS, 256x128 and
// 256x256x32 rings) were removed in round 4: no BASELINE configuration reached them any more.
// impl (measurement builds: gitmi_debug_set_gemm_impl): -1 auto | 0 tile kernel only | 9 LDS-DMA kernel wherever it can run
bool gemm_uses_p8(const GemmArgs& g, bool in_f32, bool out_f32) {
    const bool fast_ok = g_gemm_impl != 0 && gemm_fast_operands(g, in_f32, out_f32) && gemm_p8_supports(g);
    return fast_ok && (g.M > 512 || g_gemm_impl == 9);
}

hipError_t launch_gemm(const GemmArgs& g_in, bool in_f32, bool out_f32, hipStream_t s) {
    GemmArgs g = g_in;
    g.dbg = g_gemm_dbg;
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.out_f16 && (in_f32 || out_f32 || g.K % 64 != 0)) return hipErrorInvalidValue;     // fp16 residual-stream rows: bf16 engine mode
    if (gemm_uses_p8(g, in_f32, out_f32)) return launch_gemm_p8(g, out_f32, s);
    if (g.ln_part || g.part_out || g.res_part) return hipErrorInvalidValue;                 // folded LayerNorm: gemm_p8_kernel only
    if (g.out_f16) return launch_gemm_tiles<bf16_t, f16_t>(g, s);
    if (in_f32) {
        if (g.K % 16 != 0) return hipErrorInvalidValue;
        return out_f32 ? launch_gemm_tiles<float, float>(g, s) : launch_gemm_tiles<float, bf16_t>(g, s);
    }
    if (g.K % 64 != 0) return hipErrorInvalidValue;
    return out_f32 ? launch_gemm_tiles<bf16_t, float>(g, s) : launch_gemm_tiles<bf16_t, bf16_t>(g, s);
}

}  // namespace gitmi
