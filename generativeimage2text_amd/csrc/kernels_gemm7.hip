// bf16 GEMM for the large-M phases -- seventh generation: 256x256x32 tile, eight waves of 128x64, 4-stage ring.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 32 == 0, N % 8 == 0
//
// A 256x256 tile needs 128 FLOP per operand byte fed from L2 (the 256x128 ring of kernels_gemm3.hip: 85), which is
// what bounds these kernels.  BK = 32 keeps four 32-KiB stages (three K steps in flight) inside 128 KiB.  Per K step
// and wave: 4 global_load_lds_dwordx4, 12 ds_read_b128, 32 MFMAs, a counted s_waitcnt vmcnt + raw s_barrier.
// Now the fallback for narrow outputs when the half-tile pipeline of kernels_gemm10.hip does not apply.
//   * LDS image: 64-byte rows, four per 256-byte bank row; 16-byte chunk c of row r lives at
//         r*64 + (c ^ ((-(r>>2)) & 3))*16
//     so that the 16 rows of a ds_read_b128 lane group hit 16 distinct bank slots; applied on the
//     per-lane SOURCE address of the direct-to-LDS loads and again on the fragment read;
//   * epilogue straight from the accumulators (C^T layout: 4 consecutive columns per lane): bias
//     prefetched before the main loop, residual read as unconditional 16-byte loads per column group.
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int A_BYTES = BM * BK * 2;                 // 16 KiB
constexpr int W_BYTES = BN * BK * 2;                 // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 32 KiB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;      // 128 KiB

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int xcd_remap7(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut, int ACT>
__global__ __launch_bounds__(512) void gemm_ring256_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;     // 2 (M) x 4 (N) waves of 128 x 64
    const int l15 = lane & 15, lg = lane >> 4;

    const int swz = xcd_remap7(blockIdx.x, g.nwg);
    const int tile_n = swz % g.tiles_n;
    const int tile_m = swz / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // ---- staging sources: a wave instruction fills 1 KiB = 16 tile rows of 64 B --------------------
    // lane -> row lane>>2, LDS slot lane&3 holds global chunk slot ^ t, t = (-(row>>2)) & 3
    const int ld_r = lane >> 2;
    const int ld_c = ((lane & 3) ^ ((4 - (lane >> 4)) & 3)) * 8;
    const bf16_t* a_src[2];
    const bf16_t* w_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int r = m0 + (wave * 2 + q) * 16 + ld_r;
        r = r < g.M ? r : g.M - 1;
        a_src[q] = A + (size_t)r * g.lda + ld_c;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int n = n0 + (wave * 2 + q) * 16 + ld_r;
        n = n < g.N ? n : g.N - 1;
        w_src[q] = W + (size_t)n * g.K + ld_c;
    }
    auto issue = [&](int kt, int stage) {
        unsigned char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(a_src[q] + kt * BK),
                                             (lds_void_t*)(base + (wave * 2 + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(w_src[q] + kt * BK),
                                             (lds_void_t*)(base + A_BYTES + (wave * 2 + q) * 1024), 16, 0, 0);
    };

    // ---- fragment addressing: row*64 + ((lg ^ t) * 16), t = (-(l15>>2)) & 3 -------------------------
    const int frag = l15 * 64 + ((lg ^ ((4 - (l15 >> 2)) & 3)) * 16);
    const int a_off = wm * 128 * 64 + frag;
    const int w_off = A_BYTES + wn * 64 * 64 + frag;

    f32x4_t acc[4][8];   // [j: n-tile][i: m-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    issue(0, 0);
    // bias for this lane's 4x4 output columns: four UNCONDITIONAL 16-byte loads, issued before the main loop
    f32x4_t bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int n = n0 + wn * 64 + j * 16 + lg * 4;
        n = n + 3 < g.N ? n : 0;
        bias4[j] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    // steps in flight: min(nk,3); the oldest must have landed (4 loads per step and wave)
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // loads of step kt+3 go into the stage consumed in step kt-1 (all waves passed that barrier)
        if (kt + 3 < nk) issue(kt + 3, (stage + 3) & 3);
        const unsigned char* sb = smem + stage * STAGE_BYTES;
        {
            bf16x8_t wf[4], af[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sb + w_off + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_off + i * 16 * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
        }
        // this wave's loads of step kt+1 have landed; steps kt+2 and kt+3 may stay in flight
        if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = (stage + 1) & 3;
    }

    // ---- epilogue straight from the accumulators ----------------------------------------------------
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + lg * 4;
        f32x4_t rr[8];
        if (g.res) {
            const int nc = n < g.N ? n : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int m = m0 + wm * 128 + i * 16 + l15;
                m = m < g.M ? m : g.M - 1;
                rr[i] = *reinterpret_cast<const f32x4_t*>(g.res + (size_t)m * g.ldr + nc);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + i * 16 + l15;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[j][i][r] + bias4[j][r]);
            if (g.res) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rr[i][r];
            }
            if (m < g.M && n < g.N) {
                if constexpr (sizeof(TOut) == 4) {
                    *reinterpret_cast<f32x4_t*>(C + (size_t)m * g.ldc + n) = f32x4_t{v[0], v[1], v[2], v[3]};
                } else {
                    uint2 t;
                    t.x = pack2bf(v[0], v[1]);
                    t.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(C + (size_t)m * g.ldc + n) = t;
                }
            }
        }
    }
}

}  // namespace

template <typename TOut>
static void launch_ring256_t(const GemmArgs& g, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_ring256_kernel<TOut, GITMI_ACT_QUICKGELU>), dim3(g.nwg), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_ring256_kernel<TOut, GITMI_ACT_GELU_ERF>), dim3(g.nwg), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_ring256_kernel<TOut, GITMI_ACT_NONE>), dim3(g.nwg), dim3(512), 0, s, g); break;
    }
}

hipError_t launch_gemm_ring256(GemmArgs g, bool out_f32, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.nwg = tiles_m * g.tiles_n;
    if (out_f32) launch_ring256_t<float>(g, s);
    else launch_ring256_t<bf16_t>(g, s);
    return hipGetLastError();
}

}  // namespace gitmi
