// bf16 GEMM for the large-M phases -- sixth generation: 256x128x32 tile, FOUR waves of 128x64, 3-stage ring, two workgroups per CU.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 32 == 0, N % 8 == 0
//
// (variant of kernels_gemm5.hip: 4 waves per workgroup, each owning 128x64 = 8x4 MFMA tiles, so a K step
// needs 12 ds_read_b128 per 32 MFMAs instead of 8 per 16, and the two co-resident workgroups of a CU --
// one wave each per SIMD -- run decoupled: one computes while the other waits on its barrier.)
// kernels_gemm3.hip keeps one 144-KiB workgroup per CU, so a tile's ramp-up (two stages of HBM
// latency) and its epilogue are fully exposed -- with K = 768 they cost as much as the 12-step main
// loop.  Here a K step is 32 deep: a stage is 24 KiB, the 3-stage ring 72 KiB, and two workgroups
// (16 waves) share a CU: while one sits in its prologue, epilogue, barrier or counted wait the other
// one feeds the MFMA pipe.  Per step and wave: 3 global_load_lds_dwordx4, 8 ds_read_b128, 16 MFMAs,
// `s_waitcnt vmcnt(3)` + raw s_barrier.
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

constexpr int BM = 256, BN = 128, BK = 32;
constexpr int A_BYTES = BM * BK * 2;                 // 16 KiB
constexpr int W_BYTES = BN * BK * 2;                 //  8 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 24 KiB
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;      // 72 KiB

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int xcd_remap6(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_ring32w_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;     // 2 (M) x 2 (N) waves of 128 x 64
    const int l15 = lane & 15, lg = lane >> 4;

    const int swz = xcd_remap6(blockIdx.x, g.nwg);
    const int tile_n = swz % g.tiles_n;
    const int tile_m = swz / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // ---- staging sources: a wave instruction fills 1 KiB = 16 tile rows of 64 B --------------------
    // lane -> row lane>>2, LDS slot lane&3 holds global chunk slot ^ t, t = (-(row>>2)) & 3
    const int ld_r = lane >> 2;
    const int ld_c = ((lane & 3) ^ ((4 - (lane >> 4)) & 3)) * 8;
    const bf16_t* a_src[4];
    const bf16_t* w_src[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int r = m0 + (wave * 4 + q) * 16 + ld_r;
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
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(a_src[q] + kt * BK),
                                             (lds_void_t*)(base + (wave * 4 + q) * 1024), 16, 0, 0);
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
    // bias for this lane's 4x4 output columns: four UNCONDITIONAL 16-byte loads, issued before the main
    // loop (per-element guarded loads make hipcc branch + wait per element: 16 serial L2 round trips)
    f32x4_t bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int n = n0 + wn * 64 + j * 16 + lg * 4;
        n = n + 3 < g.N ? n : 0;                       // N % 8 == 0: a 4-group is entirely in or out
        bias4[j] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (nk > 1) {
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        int nxt2 = stage + 2;
        nxt2 = nxt2 >= NSTAGE ? nxt2 - NSTAGE : nxt2;
        if (kt + 2 < nk) issue(kt + 2, nxt2);
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
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        // this wave's loads of step kt+1 have landed; the ones just issued (kt+2) may stay in flight
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
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
static void launch_ring32w_t(const GemmArgs& g, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_ring32w_kernel<TOut, GITMI_ACT_QUICKGELU>), dim3(g.nwg), dim3(256), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_ring32w_kernel<TOut, GITMI_ACT_GELU_ERF>), dim3(g.nwg), dim3(256), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_ring32w_kernel<TOut, GITMI_ACT_NONE>), dim3(g.nwg), dim3(256), 0, s, g); break;
    }
}

hipError_t launch_gemm_ring32w(GemmArgs g, bool out_f32, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.nwg = tiles_m * g.tiles_n;
    if (out_f32) launch_ring32w_t<float>(g, s);
    else launch_ring32w_t<bf16_t>(g, s);
    return hipGetLastError();
}

}  // namespace gitmi
