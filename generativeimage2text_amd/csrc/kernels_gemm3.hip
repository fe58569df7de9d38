// bf16 GEMM for the large-M phases -- third generation: 256x128x64 tile, 8 waves, 3 LDS stages.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 64 == 0, N % 8 == 0
//
//   * 512 threads = 8 waves as 4 (M) x 2 (N), each wave a 64x64 output (4x4 MFMA 16x16x32 tiles,
//     swapped orientation: accumulator = C^T, see kernels_gemm.hip);
//   * operands stream HBM -> LDS with global_load_lds_dwordx4 into a 3-deep ring (3 x 48 KiB): the
//     loads of K step t+2 are issued at the top of step t, and the step ends with a COUNTED
//     `s_waitcnt vmcnt(6)` (this wave's 6 loads of step t+1 have landed, the 6 of step t+2 stay in
//     flight) followed by a raw s_barrier -- no vmcnt(0) drain in the main loop;
//   * LDS image: 128-byte rows paired into 256-byte bank rows; 16-byte chunk c of row r lives at
//         (r>>1)*256 + ((r&1) ^ ((r>>3)&1))*128 + (c ^ ((r>>1)&7))*16
//     which makes every ds_read_b128 lane group hit 16 distinct bank slots.  A direct-to-LDS load
//     writes lane-linearly, so the permutation is applied to each lane's SOURCE address (and again
//     on the fragment read);
//   * epilogue staged through LDS: bias/activation in registers, then 16-byte row-contiguous
//     stores (and 16/32-byte row-contiguous residual reads).
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;                 // 32 KiB
constexpr int W_BYTES = BN * BK * 2;                 // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 48 KiB
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;      // 144 KiB

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int xcd_remap3(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut, int ACT>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- tile of this workgroup: 2-D partition of the tile grid over the 8 XCDs (private 4-MiB L2 each).
    // Workgroup b is dispatched to XCD b % 8 (observed; only speed depends on it).  XCD x owns N group
    // x % ng and M group x / ng, and walks its sub-grid N-fastest: its slice of W (<= ~2.5 MB) stays L2
    // resident while activation panels stream through, instead of every XCD re-reading all of W for
    // every 256-row panel (measured: 220 MB fetched for 24 MB of operands with the 1-D split).
    int tile_m, tile_n;
    {
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int ng = g.ng, mg = 8 / ng;
        const int gn = x % ng, gm = x / ng;
        const int tiles_m = (g.M + BM - 1) / BM;
        const int n_lo = gn * g.tiles_n / ng, n_hi = (gn + 1) * g.tiles_n / ng;
        const int m_lo = gm * tiles_m / mg, m_hi = (gm + 1) * tiles_m / mg;
        const int nn = n_hi - n_lo;
        if (nn <= 0 || idx >= nn * (m_hi - m_lo)) return;      // surplus workgroup of an uneven split
        tile_m = m_lo + idx / nn;
        tile_n = n_lo + idx % nn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // ---- staging sources: a wave instruction fills 1 KiB = 4 bank rows = 8 tile rows ---------
    // lane -> bank row Rl = lane>>4, half hi = (lane>>3)&1, slot lo = lane&7; group parity p = q&1
    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    const bf16_t* a_src[4];
    const bf16_t* w_src[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = q & 1;
        int r = m0 + (wave * 4 + q) * 8 + 2 * Rl + (hi ^ p);
        r = r < g.M ? r : g.M - 1;
        a_src[q] = A + (size_t)r * g.lda + (lo ^ (p * 4 + Rl)) * 8;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = q & 1;
        int n = n0 + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ p);
        n = n < g.N ? n : g.N - 1;
        w_src[q] = W + (size_t)n * g.K + (lo ^ (p * 4 + Rl)) * 8;
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

    // ---- fragment addressing -----------------------------------------------------------------
    const int rowpart = (l15 >> 1) * 256 + ((l15 & 1) ^ ((l15 >> 3) & 1)) * 128;
    const int x = (l15 >> 1) & 7;
    const int ch0 = ((0 * 4 + lg) ^ x) * 16;
    const int ch1 = ((1 * 4 + lg) ^ x) * 16;
    const int a_off = wm * 64 * 128 + rowpart;
    const int w_off = A_BYTES + wn * 64 * 128 + rowpart;

    f32x4_t acc[4][4];   // [j: n-tile][i: m-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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
        if (kt + 2 < nk && !(g.dbg & 8)) issue(kt + 2, nxt2);
        const unsigned char* sb = smem + stage * STAGE_BYTES;
        if (!(g.dbg & 4))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = kk == 0 ? ch0 : ch1;
            bf16x8_t wf[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8_t*>(sb + w_off + j * 16 * 128 + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_off + i * 16 * 128 + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
        }
        // this wave's loads of step kt+1 have landed; the ones just issued (kt+2) may stay in flight
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }

    // ---- epilogue through LDS (whole 256x128 tile) --------------------------------------------
    constexpr int EPS = sizeof(TOut) == 4 ? 132 : 136;           // padded row stride (elements)
    constexpr int CPR = BN * (int)sizeof(TOut) / 16;             // 16-byte chunks per row
    constexpr int EPC = 16 / (int)sizeof(TOut);                  // elements per chunk
    static_assert(BM * EPS * sizeof(TOut) <= LDS_BYTES, "epilogue tile does not fit");
    TOut* ep = reinterpret_cast<TOut*>(smem);
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);
    if (g.dbg & 2) {
        if (acc[0][0][0] == 12345.678f) C[0] = (TOut)0;   // keep the accumulators live
        return;
    }
    if (!(g.dbg & 16))
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + lg * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[j][i][r] + bias4[j][r]);
            TOut* p = ep + (wm * 64 + i * 16 + l15) * EPS + nl;
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
    __syncthreads();
    if (g.dbg & 32) {
        if (acc[0][0][0] == 12345.678f) C[0] = (TOut)0;
        return;
    }
#pragma unroll 4
    for (int q = 0; q < BM * CPR / 512; ++q) {
        const int chunk = tid + q * 512;
        const int row = chunk / CPR, cc = chunk % CPR;
        const int m = m0 + row;
        const int n = n0 + cc * EPC;
        if (m < g.M && n < g.N && !(g.dbg & 1)) {
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
}

}  // namespace

template <typename TOut>
static void launch_ring_t(const GemmArgs& g, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_ring_kernel<TOut, GITMI_ACT_QUICKGELU>), dim3(g.nwg), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_ring_kernel<TOut, GITMI_ACT_GELU_ERF>), dim3(g.nwg), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_ring_kernel<TOut, GITMI_ACT_NONE>), dim3(g.nwg), dim3(512), 0, s, g); break;
    }
}

hipError_t launch_gemm_ring(GemmArgs g, bool out_f32, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    // N groups: smallest power of two that brings an XCD's share of W under ~2.5 MB
    int ng = 1;
    const double wbytes = (double)g.N * g.K * 2.0;
    while (ng < 8 && wbytes / ng > 2.5e6 && ng * 2 <= g.tiles_n && 8 / (ng * 2) <= tiles_m) ng *= 2;
    if (8 / ng > tiles_m) ng = 8;                     // very few M panels: split N only
    if (ng > g.tiles_n) ng = 1;
    g.ng = ng;
    const int mg = 8 / ng;
    int max_cnt = 0;
    for (int x = 0; x < 8; ++x) {
        const int gn = x % ng, gm = x / ng;
        const int nn = (gn + 1) * g.tiles_n / ng - gn * g.tiles_n / ng;
        const int mm = (gm + 1) * tiles_m / mg - gm * tiles_m / mg;
        max_cnt = nn * mm > max_cnt ? nn * mm : max_cnt;
    }
    g.nwg = 8 * max_cnt;
    if (out_f32) launch_ring_t<float>(g, s);
    else launch_ring_t<bf16_t>(g, s);
    return hipGetLastError();
}

}  // namespace gitmi
