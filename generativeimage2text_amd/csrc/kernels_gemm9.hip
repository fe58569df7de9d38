// bf16 GEMM for the large-M phases -- ninth generation: PERSISTENT ring (second attempt).
//
// One workgroup per CU walks its tiles (b, b+G, ...).  Inside a tile the loop is exactly the 3-stage ring of
// kernels_gemm3.hip; the only difference is that the loads of K steps nk and nk+1 -- i.e. steps 0 and 1 of the
// NEXT tile -- are issued during the last two steps of the current one, so a tile starts with its first two stages
// in flight instead of an exposed HBM round trip, and the epilogue (straight from the accumulators, no LDS) runs
// while they land.
//
// (tile, LDS image, counted waits as in the third generation:)
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

__device__ __forceinline__ int xcd_remap9(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut, int ACT>
__global__ __launch_bounds__(512) void gemm_pring2_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lg = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    const int nk = g.K / BK;
    const int my_tiles = (g.nwg - 1 - b) / G + 1;        // host guarantees b < nwg

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);

    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    // element offsets of this lane's staging sources for a tile (32-bit: the matrices are < 4G elements)
    struct Src { unsigned a[4]; unsigned w[2]; int m0, n0; };
    auto tile_src = [&](int it) {
        Src t;
        const int swz = xcd_remap9(b + it * G, g.nwg);
        t.m0 = (swz / g.tiles_n) * BM;
        t.n0 = (swz % g.tiles_n) * BN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = q & 1;
            int r = t.m0 + (wave * 4 + q) * 8 + 2 * Rl + (hi ^ p);
            r = r < g.M ? r : g.M - 1;
            t.a[q] = (unsigned)r * (unsigned)g.lda + (lo ^ (p * 4 + Rl)) * 8;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = q & 1;
            int n = t.n0 + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ p);
            n = n < g.N ? n : g.N - 1;
            t.w[q] = (unsigned)n * (unsigned)g.K + (lo ^ (p * 4 + Rl)) * 8;
        }
        return t;
    };
    auto issue = [&](const Src& t, int kt, int stage) {
        unsigned char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(A + t.a[q] + kt * BK),
                                             (lds_void_t*)(base + (wave * 4 + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(W + t.w[q] + kt * BK),
                                             (lds_void_t*)(base + A_BYTES + (wave * 2 + q) * 1024), 16, 0, 0);
    };

    const int rowpart = (l15 >> 1) * 256 + ((l15 & 1) ^ ((l15 >> 3) & 1)) * 128;
    const int x = (l15 >> 1) & 7;
    const int ch0 = ((0 * 4 + lg) ^ x) * 16;
    const int ch1 = ((1 * 4 + lg) ^ x) * 16;
    const int a_off = wm * 64 * 128 + rowpart;
    const int w_off = A_BYTES + wn * 64 * 128 + rowpart;

    f32x4_t acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    Src cur = tile_src(0);
    // prologue: the first two K steps of the stream (a 1-step tile takes its second one from the next tile)
    issue(cur, 0, 0);
    int in_flight = 1;
    if (nk > 1) { issue(cur, 1, 1); in_flight = 2; }
    else if (my_tiles > 1) { const Src n1 = tile_src(1); issue(n1, 0, 1); in_flight = 2; }
    if (in_flight == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int stage = 0;
    for (int it = 0; it < my_tiles; ++it) {
        const bool has_next = it + 1 < my_tiles;
        Src nxt = cur;
        if (has_next) nxt = tile_src(it + 1);
        // bias of this lane's columns: unconditional 16-byte loads at the top of the tile, consumed in its epilogue
        f32x4_t bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = cur.n0 + wn * 64 + j * 16 + lg * 4;
            n = n + 3 < g.N ? n : 0;
            bias4[j] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        for (int kt = 0; kt < nk; ++kt) {
            int nxt2 = stage + 2;
            nxt2 = nxt2 >= NSTAGE ? nxt2 - NSTAGE : nxt2;
            // stream step +2: same tile, or step (kt+2-nk) of the next tile
            bool issued = true;
            if (kt + 2 < nk) issue(cur, kt + 2, nxt2);
            else if (has_next && kt + 2 - nk < nk) issue(nxt, kt + 2 - nk, nxt2);
            else issued = false;
            const unsigned char* sb = smem + stage * STAGE_BYTES;
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
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
            }
            // everything older than the loads issued in this step has landed (next step's stage in particular)
            if (issued) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        }
        // ---- epilogue of tile `it` straight from the accumulators (the ring already carries the next tile) ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = cur.n0 + wn * 64 + j * 16 + lg * 4;
            f32x4_t rr[4];
            if (g.res) {
                const int nc = n < g.N ? n : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int m = cur.m0 + wm * 64 + i * 16 + l15;
                    m = m < g.M ? m : g.M - 1;
                    rr[i] = *reinterpret_cast<const f32x4_t*>(g.res + (size_t)m * g.ldr + nc);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = cur.m0 + wm * 64 + i * 16 + l15;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[j][i][r] + bias4[j][r]);
                acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
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
        cur = nxt;
    }
}

}  // namespace

template <typename TOut>
static void launch_pring2_t(const GemmArgs& g, int grid, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_pring2_kernel<TOut, GITMI_ACT_QUICKGELU>), dim3(grid), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_pring2_kernel<TOut, GITMI_ACT_GELU_ERF>), dim3(grid), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_pring2_kernel<TOut, GITMI_ACT_NONE>), dim3(grid), dim3(512), 0, s, g); break;
    }
}

hipError_t launch_gemm_pring2(GemmArgs g, bool out_f32, hipStream_t s) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    g.nwg = tiles_m * g.tiles_n;
    const int grid = g.nwg < n_cu ? g.nwg : n_cu;
    if (out_f32) launch_pring2_t<float>(g, grid, s);
    else launch_pring2_t<bf16_t>(g, grid, s);
    return hipGetLastError();
}

}  // namespace gitmi
