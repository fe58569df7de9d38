// Weight-streaming GEMM for the decode steps (M = rows of the beam batch, 64..256) on gfx950.
//
// A decode step multiplies a handful of rows by every decoder weight matrix once: the work is
// HBM-bound weight streaming (131.8 MB of bf16 weights per step), and with a
// 128x128-tile GEMM an N=768 projection would occupy 6 of 256 CUs.  This kernel instead
//   * gives every workgroup a 64-row x (16*NT)-column output strip and a K slice,
//   * splits that K slice again over the 4 waves of the workgroup,
//   * feeds v_mfma_f32_16x16x32_bf16 straight from global memory: the weight fragment
//     (16 rows x 32 k, 16 B per lane) is read exactly once from HBM, the activation fragments come
//     out of L2 (the activation matrix is <= 400 KB) -- no LDS staging, no barrier in the K loop,
//     up to 24 independent 16-byte loads in flight per lane,
//   * reduces the 4 waves' accumulators through LDS, and either applies the fused epilogue
//     (bias / activation / residual) or, for split-K across workgroups, writes fp32 partial
//     slabs that `splitk_ln_kernel` sums in a FIXED order together with bias + residual +
//     LayerNorm (post-norm BERT layer, modeling_bert.py:171-178, 243-250) -- deterministic, no atomics.
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

template <typename TOut>
__device__ __forceinline__ void sk_store4(TOut* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void sk_store4<float>(float* p, const float (&v)[4]) {
    f32x4_t t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4_t*>(p) = t;
}
template <>
__device__ __forceinline__ void sk_store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = pack2bf(v[0], v[1]);
    t.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}

// grid = (ceil(N / (16*NT)), ceil(M / 64), S); block = 256
template <typename TOut, int NT>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs g) {
    // cross-wave exchange: a wave publishes only the 3 m-tiles it does NOT finish itself -> 12 KiB per n-tile,
    // which lets a decode workgroup share a CU with a 144-KiB ring-GEMM workgroup of another stream
    __shared__ __attribute__((aligned(16))) f32x4_t red[4][3][64];   // [writer wave][foreign m-tile][lane], reused per n-tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * 64, slice = blockIdx.z;

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // this wave's k-steps (32 wide): contiguous range of the block's slice
    const int ksteps = g.K >> 5;
    const int parts = 4 * g.S;
    const int per = (ksteps + parts - 1) / parts;
    const int kb = (slice * 4 + wave) * per;
    const int ke = min(kb + per, ksteps);

    const bf16_t* wp[NT];
    const bf16_t* xp[4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + j * 16 + l15;
        n = n < g.N ? n : g.N - 1;
        wp[j] = W + (size_t)n * g.K + lg * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 16 + l15;
        m = m < g.M ? m : g.M - 1;
        xp[i] = X + (size_t)m * g.lda + lg * 8;
    }

    f32x4_t acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // bias of the columns this lane will finish (wave w finishes m-tile w of every n-tile): fetched together
    // with the first operand loads so the epilogue has no dependent memory round trip of its own
    float bias_pf[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + lg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = n + r < g.N ? n + r : g.N - 1;
            bias_pf[j][r] = (g.bias && g.S == 1) ? g.bias[nn] : 0.f;
        }
    }

    // k-steps in chunks whose loads are all issued before the first MFMA (6, then 2, then 1)
    auto chunk = [&](auto UC, int k0) {
        constexpr int U = decltype(UC)::value;
        bf16x8_t wf[U][NT], xf[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[u][j] = *reinterpret_cast<const bf16x8_t*>(wp[j] + (size_t)(k0 + u) * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[u][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)(k0 + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][j], xf[u][i], acc[j][i], 0, 0, 0);
    };
    int k = kb;
    for (; k + 6 <= ke; k += 6) chunk(std::integral_constant<int, 6>{}, k);
    for (; k + 2 <= ke; k += 2) chunk(std::integral_constant<int, 2>{}, k);
    for (; k < ke; ++k) chunk(std::integral_constant<int, 1>{}, k);

    // ---- cross-wave reduction, one n-tile at a time: wave w finishes m-tile w ----------------------
    const int i = wave;
    const int m = m0 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j > 0) __syncthreads();                  // the exchange buffer of the previous n-tile has been consumed
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            if (ii != wave) red[wave][ii < wave ? ii : ii - 1][lane] = acc[j][ii];
        __syncthreads();
        f32x4_t s = acc[j][0];                       // own partial of tile `wave` (static-index select)
#pragma unroll
        for (int ii = 1; ii < 4; ++ii)
            if (ii == wave) s = acc[j][ii];
        // fixed summation order over the writer waves (own partial takes its place in that order)
        f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4_t t = w == wave ? s : red[w][i < w ? i : i - 1][lane];
            tot[0] += t[0]; tot[1] += t[1]; tot[2] += t[2]; tot[3] += t[3];
        }
        s = tot;
        const int n = n0 + j * 16 + lg * 4;
        if (m >= g.M || n >= g.N) continue;
        float v[4] = {s[0], s[1], s[2], s[3]};
        if (g.S > 1) {
            // raw fp32 partial slab [S][M][N]; bias/residual/LayerNorm happen in splitk_ln_kernel
            float* pp = g.partial + ((size_t)slice * g.M + m) * g.N + n;
            if (n + 3 < g.N && (g.N & 3) == 0) sk_store4<float>(pp, v);
            else
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) pp[r] = v[r];
            continue;
        }
        const bool full = n + 3 < g.N;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias_pf[j][r];
        if (g.act != GITMI_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
        }
        if (g.res) {
            if (full) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += g.res[(size_t)m * g.ldr + n + r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) v[r] += g.res[(size_t)m * g.ldr + n + r];
            }
        }
        TOut* cp = reinterpret_cast<TOut*>(g.C) + (size_t)m * g.ldc + n;
        if (full && (g.ldc & 3) == 0) sk_store4<TOut>(cp, v);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) st<TOut>(cp + r, v[r]);
    }
}

// y = LayerNorm(sum_s partial[s] + bias + residual);  one 256-thread workgroup per row, one float4 per
// thread (D <= 1024, D % 4 == 0), every load of the row issued before the first add.
// Writes the fp32 hidden state (next residual) and its bf16 copy (next GEMM operand).
template <int S>
__global__ __launch_bounds__(256) void splitk_ln_kernel(const float* __restrict__ partial,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ y_f, bf16_t* __restrict__ y_t, int rows,
                                                        int D) {
    __shared__ float s_part[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int c = tid * 4;
    const bool on = c < D;
    f32x4_t p[S];
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, g4 = a, b4 = a;
    if (on) {
#pragma unroll
        for (int s = 0; s < S; ++s) p[s] = *reinterpret_cast<const f32x4_t*>(partial + ((size_t)s * rows + row) * D + c);
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(bias + c);
        const f32x4_t rr = *reinterpret_cast<const f32x4_t*>(res + (size_t)row * D + c);
        g4 = *reinterpret_cast<const f32x4_t*>(gamma + c);
        b4 = *reinterpret_cast<const f32x4_t*>(beta + c);
        a = p[0];
#pragma unroll
        for (int s = 1; s < S; ++s) { a[0] += p[s][0]; a[1] += p[s][1]; a[2] += p[s][2]; a[3] += p[s][3]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] += bb[r] + rr[r];
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
        sk_store4<float>(y_f + (size_t)row * D + c, o);
        sk_store4<bf16_t>(y_t + (size_t)row * D + c, o);
    }
}

// ---- host launchers ------------------------------------------------------------------
hipError_t launch_skinny_gemm(SkinnyArgs g, bool out_f32, int NT, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K % 32 != 0 || g.S < 1) return hipErrorInvalidValue;
    if (g.S > 1 && g.partial == nullptr) return hipErrorInvalidValue;
    dim3 grid((g.N + 16 * NT - 1) / (16 * NT), (g.M + 63) / 64, g.S), block(256);
    if (NT == 1) {
        if (out_f32) hipLaunchKernelGGL((skinny_gemm_kernel<float, 1>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_gemm_kernel<bf16_t, 1>), grid, block, 0, s, g);
    } else if (NT == 2) {
        if (out_f32) hipLaunchKernelGGL((skinny_gemm_kernel<float, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((skinny_gemm_kernel<bf16_t, 2>), grid, block, 0, s, g);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_splitk_ln(const float* partial, int S, const float* bias, const float* res, const float* gamma,
                            const float* beta, float eps, float* y_f, void* y_t, int rows, int D, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (D > 1024 || (D & 3) || S < 1 || S > 8) return hipErrorInvalidValue;
#define GITMI_SKLN(SS)                                                                                           \
    hipLaunchKernelGGL(splitk_ln_kernel<SS>, dim3(rows), dim3(256), 0, s, partial, bias, res, gamma, beta, eps, y_f, \
                       (bf16_t*)y_t, rows, D)
    switch (S) {
        case 1: GITMI_SKLN(1); break;
        case 2: GITMI_SKLN(2); break;
        case 3: GITMI_SKLN(3); break;
        case 4: GITMI_SKLN(4); break;
        case 5: GITMI_SKLN(5); break;
        case 6: GITMI_SKLN(6); break;
        case 7: GITMI_SKLN(7); break;
        default: GITMI_SKLN(8); break;
    }
#undef GITMI_SKLN
    return hipGetLastError();
}

}  // namespace gitmi
