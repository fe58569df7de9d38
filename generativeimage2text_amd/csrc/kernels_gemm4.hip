// bf16 GEMM for the large-M phases -- fourth generation: PERSISTENT 3-stage ring.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 64 == 0, N % 8 == 0
//
// Same 256x128x64 tile / 8 waves / LDS image / counted-vmcnt ring as kernels_gemm3.hip, but one
// workgroup per CU walks a list of tiles and treats (tile, k-step) pairs as ONE stream of K steps:
// the loads of stream step s+2 are issued during step s regardless of tile boundaries, so a new
// tile starts with its first two stages already in LDS (no per-tile ramp-up; for K=768 that ramp
// was as long as the 12-step main loop), and the epilogue of tile i overlaps the loads of tile i+1.
//   * the epilogue stores straight from the accumulators (C^T layout: 4 consecutive columns per
//     lane, 8/16-byte stores), which leaves the whole ring free for the next tile's stages;
//   * the bias of a tile arrives through the same LDS-DMA path (one global_load_lds_dword per wave
//     with the tile's first stage, two 2-KiB slots alternating by tile), so the main loop contains
//     no register-destination global load -- hipcc would otherwise drain vmcnt(0) around it;
//   * every wait is counted: at the end of step s the wave waits for everything but the loads it
//     issued for step s+2 (6, or 7 when that step opened a tile with a bias).
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;                 // 32 KiB
constexpr int W_BYTES = BN * BK * 2;                 // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 48 KiB
constexpr int NSTAGE = 3;
constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;     // 144 KiB
constexpr int BIAS_SLOT = 8 * 256;                   // 8 waves x (64 lanes x 4 B)
constexpr int LDS_BYTES = RING_BYTES + 2 * BIAS_SLOT;

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int xcd_remap4(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TOut, int ACT>
__global__ __launch_bounds__(512) void gemm_pring_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int l15 = lane & 15, lg = lane >> 4;
    const int G = gridDim.x, b = blockIdx.x;
    const int nk = g.K / BK;
    const int my_tiles = (g.nwg - 1 - b) / G + 1;        // host guarantees b < nwg
    const int S = my_tiles * nk;                         // stream length in K steps
    const bool has_bias = g.bias != nullptr;

    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

    // ---- load cursor ---------------------------------------------------------------------------
    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    unsigned a_src[4];     // element offsets into A / W / bias (32-bit: fewer live registers than pointers)
    unsigned w_src[2];
    unsigned b_src = 0;
    int l_it = 0, l_kt = 0;
    auto set_load_tile = [&](int it) {
        const int swz = xcd_remap4(b + it * G, g.nwg);
        const int m0 = (swz / g.tiles_n) * BM, n0 = (swz % g.tiles_n) * BN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = q & 1;
            int r = m0 + (wave * 4 + q) * 8 + 2 * Rl + (hi ^ p);
            r = r < g.M ? r : g.M - 1;
            a_src[q] = (unsigned)r * (unsigned)g.lda + (lo ^ (p * 4 + Rl)) * 8;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = q & 1;
            int n = n0 + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ p);
            n = n < g.N ? n : g.N - 1;
            w_src[q] = (unsigned)n * (unsigned)g.K + (lo ^ (p * 4 + Rl)) * 8;
        }
        if (has_bias) {
            int n = n0 + wave * 16 + l15;                 // lanes 16..63 re-read the same 16 floats
            n = n < g.N ? n : g.N - 1;
            b_src = (unsigned)n;
        }
    };
    // issues the loads of the cursor's step into `stage`; returns the number of VMEM ops issued
    auto issue = [&](int stage) -> int {
        unsigned char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(A + a_src[q] + l_kt * BK),
                                             (lds_void_t*)(base + (wave * 4 + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(W + w_src[q] + l_kt * BK),
                                             (lds_void_t*)(base + A_BYTES + (wave * 2 + q) * 1024), 16, 0, 0);
        int ops = 6;
        if (has_bias && l_kt == 0) {
            __builtin_amdgcn_global_load_lds((const void*)(g.bias + b_src),
                                             (lds_void_t*)(smem + RING_BYTES + (l_it & 1) * BIAS_SLOT + wave * 256), 4, 0, 0);
            ops = 7;
        }
        if (++l_kt == nk) {
            l_kt = 0;
            ++l_it;
            if (l_it < my_tiles) set_load_tile(l_it);
        }
        return ops;
    };

    // ---- fragment addressing (same LDS image as kernels_gemm3.hip) -------------------------------
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

    // ---- compute cursor ------------------------------------------------------------------------
    int c_it = 0, c_kt = 0;
    int cm0, cn0;
    {
        const int swz = xcd_remap4(b, g.nwg);
        cm0 = (swz / g.tiles_n) * BM;
        cn0 = (swz % g.tiles_n) * BN;
    }

    set_load_tile(0);
    issue(0);
    if (S > 1) {
        const int ops1 = issue(1);
        if (ops1 == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);
    int stage = 0;
    for (int s = 0; s < S; ++s) {
        // loads of stream step s+2 go into the stage consumed in step s-1 (every wave has passed that
        // step's barrier).  Normally issued first thing; in a tile's last step they are issued AFTER the
        // epilogue's memory traffic so that nothing younger than them is outstanding at the counted wait.
        const bool epi = c_kt == nk - 1;
        int nxt2 = stage + 2;
        nxt2 = nxt2 >= NSTAGE ? nxt2 - NSTAGE : nxt2;
        int ops2 = 0;
        if (!epi && s + 2 < S) ops2 = issue(nxt2);
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

        if (epi) {
            // ---- tile finished: epilogue straight from the accumulators ---------------------------
            const unsigned char* bslot = smem + RING_BYTES + (c_it & 1) * BIAS_SLOT;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = cn0 + wn * 64 + j * 16 + lg * 4;
                // residual of this column group: 4 UNCONDITIONAL 16-byte loads from clamped addresses, in
                // flight together (a guarded load per element makes hipcc wait vmcnt(0) once per element)
                f32x4_t rr[4];
                if (g.res) {
                    const int nc = n < g.N ? n : 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int m = cm0 + wm * 64 + i * 16 + l15;
                        m = m < g.M ? m : g.M - 1;
                        rr[i] = *reinterpret_cast<const f32x4_t*>(g.res + (size_t)m * g.ldr + nc);
                    }
                }
                f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
                if (has_bias) b4 = *reinterpret_cast<const f32x4_t*>(bslot + (wn * 4 + j) * 256 + lg * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = cm0 + wm * 64 + i * 16 + l15;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[j][i][r] + b4[r]);
                    acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (g.res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[i][r];
                    }
                    if (m < g.M && n < g.N) {                 // N % 8 == 0: a 4-group is entirely in or out
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
            c_kt = 0;
            ++c_it;
            if (c_it < my_tiles) {
                const int swz = xcd_remap4(b + c_it * G, g.nwg);
                cm0 = (swz / g.tiles_n) * BM;
                cn0 = (swz % g.tiles_n) * BN;
            }
        } else {
            ++c_kt;
        }

        if (epi && s + 2 < S) ops2 = issue(nxt2);
        if (ops2 == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (ops2 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
}

template <typename TOut>
void launch_pring_t(const GemmArgs& g, int grid, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_pring_kernel<TOut, GITMI_ACT_QUICKGELU>), dim3(grid), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_pring_kernel<TOut, GITMI_ACT_GELU_ERF>), dim3(grid), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_pring_kernel<TOut, GITMI_ACT_NONE>), dim3(grid), dim3(512), 0, s, g); break;
    }
}

}  // namespace

hipError_t launch_gemm_pring(GemmArgs g, bool out_f32, hipStream_t s) {
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
    const int grid = g.nwg < n_cu ? g.nwg : n_cu;      // one 144-KiB workgroup per CU
    if (out_f32) launch_pring_t<float>(g, grid, s);
    else launch_pring_t<bf16_t>(g, grid, s);
    return hipGetLastError();
}

}  // namespace gitmi
