// bf16 GEMM for the large-M phases (ViT encoder, decoder prefill) -- second generation.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 64 == 0, N % 8 == 0
//
// Differences to kernels_gemm.hip (kept for fp32 and odd shapes):
//   * operands go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction), no
//     register staging and no ds_write pass; two LDS stages, ONE barrier per 64-deep K step, the
//     loads of step t+1 are in flight while step t feeds the MFMAs;
//   * LDS rows are 128 B (64 bf16) with the 16-byte chunk index XOR-swizzled by (row & 7); since the
//     LDS image of a direct load is lane-linear, the swizzle is applied to the per-lane SOURCE
//     address and again on the fragment read (same involution);
//   * the epilogue goes through LDS so that every lane stores 16 contiguous bytes and a wave
//     stores whole 256/512-byte row segments (the fp32 residual is read the same way).
// Tile 128x128x64, 256 threads = 2x2 waves of 64x64 (4x4 MFMA 16x16x32 tiles), swapped MFMA
// orientation (accumulator = C^T) as in kernels_gemm.hip, XCD-aware tile order.
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;      // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // 64 KiB

__device__ __forceinline__ int xcd_remap2(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

typedef __attribute__((address_space(3))) void lds_void_t;

template <typename TOut>
__global__ __launch_bounds__(256, 2) void gemm_dlds_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int swz = xcd_remap2(blockIdx.x, g.nwg);
    const int tile_n = swz % g.tiles_n;
    const int tile_m = swz / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // ---- direct-to-LDS staging: wave w, instruction q covers tile rows (w*4+q)*8 .. +8 ------
    // lane -> (row = lane>>3, LDS slot = lane&7) holds global chunk (slot ^ row)
    const int ld_r = lane >> 3;
    const int ld_c = ((lane & 7) ^ ld_r) * 8;
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int r = m0 + (wave * 4 + q) * 8 + ld_r;
        r = r < g.M ? r : g.M - 1;
        a_src[q] = A + (size_t)r * g.lda + ld_c;
        int n = n0 + (wave * 4 + q) * 8 + ld_r;
        n = n < g.N ? n : g.N - 1;
        w_src[q] = W + (size_t)n * g.K + ld_c;
    }
    auto issue = [&](int kt, int stage) {
        unsigned char* base = smem + stage * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(a_src[q] + kt * BK), (lds_void_t*)(base + q * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(w_src[q] + kt * BK),
                                             (lds_void_t*)(base + BM * BK * 2 + q * 1024), 16, 0, 0);
    };

    // ---- fragment addressing: row*128 B + ((kk*4 + lg) ^ (row & 7)) * 16 B -------------------
    const int sw0 = ((0 * 4 + lg) ^ (l15 & 7)) * 16;
    const int sw1 = ((1 * 4 + lg) ^ (l15 & 7)) * 16;
    const int a_row_off = (wm * 64 + l15) * 128;
    const int w_row_off = BM * BK * 2 + (wn * 64 + l15) * 128;

    f32x4_t acc[4][4];   // [j: n-tile][i: m-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    issue(0, 0);
    __syncthreads();   // LDS-DMA pending -> the barrier's fence drains vmcnt(0)

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const unsigned char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk == 0 ? sw0 : sw1;
            bf16x8_t wf[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8_t*>(sb + w_row_off + j * 16 * 128 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_row_off + i * 16 * 128 + sw);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
        }
        __syncthreads();   // (a) next stage landed (own loads drained before the barrier) (b) this stage free
    }

    // ---- epilogue through LDS, two 64-row halves ----------------------------------------------
    constexpr int EPS = sizeof(TOut) == 4 ? 132 : 136;           // padded row stride (elements)
    constexpr int CPR = BN * (int)sizeof(TOut) / 16;             // 16-byte chunks per row
    constexpr int EPC = 16 / (int)sizeof(TOut);                  // elements per chunk
    TOut* ep = reinterpret_cast<TOut*>(smem);
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = wn * 64 + j * 16 + lg * 4;           // column inside the tile
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (g.bias && n0 + nl + 3 < g.N) {      // N % 8 == 0: one unconditional 16-byte load
                    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + nl);
                    bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[j][i][r] + bv[r], g.act);
                    TOut* p = ep + (i * 16 + l15) * EPS + nl;
                    if constexpr (sizeof(TOut) == 4) {
                        *reinterpret_cast<f32x4_t*>(p) = f32x4_t{v[0], v[1], v[2], v[3]};
                    } else {
                        uint2 t;
                        t.x = pack2bf(v[0], v[1]);
                        t.y = pack2bf(v[2], v[3]);
                        *reinterpret_cast<uint2*>(p) = t;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 64 * CPR / 256; ++q) {
            const int chunk = tid + q * 256;
            const int row = chunk / CPR, cc = chunk % CPR;
            const int m = m0 + half * 64 + row;
            const int n = n0 + cc * EPC;
            if (m < g.M && n < g.N) {       // N % 8 == 0: a chunk is entirely inside or outside
                if constexpr (sizeof(TOut) == 4) {
                    f32x4_t v = *reinterpret_cast<const f32x4_t*>(ep + row * EPS + cc * EPC);
                    if (g.res) {
                        const f32x4_t rr = *reinterpret_cast<const f32x4_t*>(g.res + (size_t)m * g.ldr + n);
                        v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
                    }
                    *reinterpret_cast<f32x4_t*>(C + (size_t)m * g.ldc + n) = v;
                } else {
                    u32x4_t v = *reinterpret_cast<const u32x4_t*>(ep + row * EPS + cc * EPC);
                    if (g.res) {
                        const float* rp = g.res + (size_t)m * g.ldr + n;
                        const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(rp);
                        const f32x4_t r1 = *reinterpret_cast<const f32x4_t*>(rp + 4);
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unpack2op(v[e], f[2 * e], f[2 * e + 1]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { f[e] += r0[e]; f[4 + e] += r1[e]; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack2bf(f[2 * e], f[2 * e + 1]);
                    }
                    *reinterpret_cast<u32x4_t*>(C + (size_t)m * g.ldc + n) = v;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

bool gemm_dlds_supported(const GemmArgs& g, bool in_f32, bool out_f32) {
    if (in_f32) return false;
    if (g.K % BK != 0 || g.N % 8 != 0 || g.lda % 8 != 0) return false;
    if (g.ldc % (out_f32 ? 4 : 8) != 0) return false;
    if (g.res && g.ldr % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C) |
         reinterpret_cast<uintptr_t>(g.res)) & 15)
        return false;
    return true;
}

hipError_t launch_gemm_dlds(GemmArgs g, bool out_f32, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.nwg = tiles_m * g.tiles_n;
    if (out_f32) hipLaunchKernelGGL(gemm_dlds_kernel<float>, dim3(g.nwg), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_dlds_kernel<bf16_t>, dim3(g.nwg), dim3(256), 0, s, g);
    return hipGetLastError();
}

}  // namespace gitmi
