// bf16 GEMM for the large-M phases -- eleventh generation ("p9"): the 256x256x64 half-tile pipeline of
// kernels_gemm10.hip as a PERSISTENT kernel.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N])     16-bit output (bf16 operand rows, or fp16 rows of a branch output
//                                                 that the next LayerNorm kernel adds to the residual stream)
//   K % 64 == 0, K >= 256, N % 256 == 0
//
// What the measurements of rounds 1-2 left on the table for the K = 768 shapes of the image encoder:
// ~7 us of prologue + epilogue per 12-us K loop, a K loop that runs 35-50 % slower with every CU busy because only
// 64 KiB of operand loads fit in flight against ~1.3 us of loaded L2 latency, and whole rounds of tiles that start and
// drain together.  This kernel changes three things and nothing about the MFMA / LDS-read schedule itself:
//
//   * ONE workgroup per CU walks a list of tiles (the XCD's tile list of kernels_gemm10.hip, strided by the workgroups
//     of the XCD).  The half-tile loads form ONE stream over all (tile, K tile) pairs of the workgroup: while the last
//     K tiles of a tile are multiplied the first K tiles of the next tile are already in flight -- no prologue per tile.
//   * The LDS is a RING of S = 10 half-tile slots (all 160 KiB) instead of two K-tile buffers: a half tile is issued
//     D = S - 2 = 8 phases before the phase that reads it and confirmed c = S - 4 = 6 phases after its issue
//     (`s_waitcnt vmcnt(12)`), i.e. six half tiles = 96 KiB are in flight per CU instead of four.
//       half tile h = 4 u + j of K-tile u (j: 0 = A rows 0.., 1 = W rows 0.., 2 = W rows 128.., 3 = A rows MH..) lives in
//       slot h % S, is read in phase 4u (j = 0, 1), 4u + 1 (j = 2), 4u + 2 (j = 3) and its slot is refilled by
//       half tile h + S, issued in phase h + S - D = h + 2 >= (read phase) + 2: the two-phase rule of the staggered
//       wave groups (kernels_gemm10.hip).
//   * No epilogue phase: quadrant (qm, qn) of the accumulators is complete after ITS last MFMA group -- (0,0) after
//     phase 1 of the tile's last K tile, (0,1) after phase 2, (1,1) after phase 3, (1,0) after phase 4 -- and is written
//     out (bias, activation, 16-bit pack, 8-byte stores straight from the registers; no LDS, no barrier) in the READ
//     segment of the following phase, while the other wave group of the workgroup issues MFMAs.  The stores drain
//     under the next tile's K loop.
//
// VMEM bookkeeping: `vmcnt` counts LDS-DMA loads, the (inline-asm) bias loads and the epilogue stores together.  A counted
// wait `vmcnt(N)` proves that load L has landed iff at least N loads were issued after L (loads return in order among
// themselves; stores only make the wait stricter) -- N = 12 = the loads of the six most recent phases.  The bias of a
// tile is requested with inline-asm loads in the tile's first phase and first used >= 12 phases later (K >= 256), by
// which time at least 24 younger loads have been waited for: no wait of its own (an ordinary load would make hipcc
// drain the whole DMA queue at its first use, cdna_hip_programming.md section 5 trap 4b).
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

namespace {

constexpr int BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;             // 16 KiB: one half tile (128 rows x 64 k, bf16)

typedef __attribute__((address_space(3))) void lds_void_t;

#define P9_BARRIER()                           \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

template <int N> __device__ __forceinline__ void p9_wait_vm() {
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else static_assert(N == 0, "unsupported count");
}

template <typename TOut> __device__ __forceinline__ uint32_t pack2_16(float lo, float hi) {
    if constexpr (std::is_same<TOut, f16_t>::value) return pack2h(lo, hi);
    else return pack2bf(lo, hi);
}

// 16 bytes from global memory by inline asm (invisible to hipcc's vmcnt bookkeeping, see the header): SGPR base + 32-bit
// per-lane byte offset + immediate, so that no 64-bit per-lane pointer has to live across the K loop
template <int IMM> __device__ __forceinline__ f32x4_t asm_load_f32x4(const float* base, uint32_t voff) {
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "n"(IMM) : "memory");
    return v;
}

// MH = rows of an activation half tile: 128 -> 256x256 tile, 96 -> 192x256 tile (kernels_gemm10.hip)
// S  = ring slots (10: all 160 KiB; 8: the 128 KiB / four-half-tiles-in-flight schedule of kernels_gemm10.hip, for A/B)
// DBG (measurement builds, tools/gemm_p9_diag.py): 1 no output stores, 2 no quadrant output at all (results are wrong)
template <typename TOut, int ACT, int MH, int S, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_p9_kernel(GemmArgs g) {
    constexpr int BM = 2 * MH, MI = MH / 32;
    constexpr int D = S - 2;                           // issue distance in phases
    constexpr int WAITN = 2 * (S - 4);                 // loads of the c = S - 4 most recent phases
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * HALF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- tile list of this workgroup: the XCD's (M-major, N-fastest) chunk of kernels_gemm10.hip, strided by the
    // workgroups of the XCD, so that the workgroups of an XCD always work on neighbouring tiles (shared panels in its L2)
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int ng = g.ng, mg = 8 / ng;
    const int gn = xcd % ng, gm = xcd / ng;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int n_lo = gn * g.tiles_n / ng, n_hi = (gn + 1) * g.tiles_n / ng;
    const int nn = n_hi - n_lo;
    const int tg = tiles_m * nn;
    const int lo_t = gm * tg / mg, hi_t = (gm + 1) * tg / mg;
    const int cnt = hi_t - lo_t;
    if (widx >= cnt) return;                           // surplus workgroup (whole workgroup: no barrier is ever reached)
    const int n_mine = (cnt - widx + wpx - 1) / wpx;
    const int nk = g.K / BK;                           // >= 4 (launcher)

    const char* __restrict__ Ab = reinterpret_cast<const char*>(g.A);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(g.W);
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);

    // ---- staging sources of the ISSUE cursor's tile (kernels_gemm10.hip): this wave owns pieces 2*wave + q of a half tile
    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    uint32_t a_off[2][2], w_off[2][2];                 // byte offsets [half][q]
    auto tile_coords = [&](int ordinal, int& m0, int& n0) {
        const int L = lo_t + widx + ordinal * wpx;
        m0 = (L / nn) * BM;
        n0 = (n_lo + L % nn) * BN;
    };
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rr = h * MH + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ q);
                const int ch = lo ^ (q * 4 + Rl);
                int m = m0 + rr;
                m = m < g.M ? m : g.M - 1;
                a_off[h][q] = ((uint32_t)m * (uint32_t)g.lda + ch * 8) * 2u;
                const int rn = h * 128 + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ q);
                w_off[h][q] = ((uint32_t)(n0 + rn) * (uint32_t)g.K + ch * 8) * 2u;
            }
    };
    // issue cursor: next half tile to request = (tile ordinal is_t, K tile is_kt, j = is_j) -> slot is_slot
    int is_t = 0, is_kt = 0, is_slot = 0;
    {
        int m0, n0;
        tile_coords(0, m0, n0);
        set_sources(m0, n0);
    }
    // J (compile time: runtime-indexed register arrays would go to scratch): 0 = A half 0, 1 = W half 0, 2 = W half 1,
    // 3 = A half 1 -- the half tile the stream requests next; the cursor moves on after J == 3
    auto issue_next = [&](auto j_c) {
        constexpr int J = decltype(j_c)::value;
        constexpr bool isw = J == 1 || J == 2;
        constexpr int half = J >> 1;
        const char* src = (isw ? Wb : Ab) + (size_t)is_kt * (BK * 2);
        unsigned char* dst = smem + is_slot * HALF_BYTES + wave * 2048;
        const uint32_t o0 = isw ? w_off[half][0] : a_off[half][0];
        const uint32_t o1 = isw ? w_off[half][1] : a_off[half][1];
        __builtin_amdgcn_global_load_lds((const void*)(src + o0), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(src + o1), (lds_void_t*)(dst + 1024), 16, 0, 0);
        is_slot = is_slot + 1 == S ? 0 : is_slot + 1;
        if constexpr (J == 3) {
            if (++is_kt == nk) {
                is_kt = 0;
                if (++is_t < n_mine) {
                    int m0, n0;
                    tile_coords(is_t, m0, n0);
                    set_sources(m0, n0);
                }
            }
        }
    };
    // one phase's request + the counted wait that confirms the half tiles of the NEXT phase.  Phase p (0..3) of a K tile
    // requests the half tile D phases ahead: j = (p + D) % 4
    auto issue_and_wait = [&](auto p_c) {
        constexpr int J = (decltype(p_c)::value + D) & 3;
        if (is_t < n_mine) { issue_next(std::integral_constant<int, J>{}); p9_wait_vm<WAITN>(); }
        else p9_wait_vm<0>();                           // tail of the workgroup's stream: nothing left to request
    };

    // ---- fragment addressing (bank-conflict free image, kernels_gemm3.hip) ---------------------------
    const int rowpart = (l15 >> 1) * 256 + ((l15 & 1) ^ ((l15 >> 3) & 1)) * 128;
    const int x7 = (l15 >> 1) & 7;
    const int ch0 = ((0 * 4 + lg) ^ x7) * 16;
    const int ch1 = ((1 * 4 + lg) ^ x7) * 16;
    const int a_rd = grp * (MH / 2) * 128 + rowpart;              // + slot base + i*2048 + ch
    const int w_rd = wc * 32 * 128 + rowpart;                     // + slot base + j*2048 + ch

    f32x4_t acc[2][2][2][MI];  // [qm][qn][j: n-frag][i: m-frag]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[a][b][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    bf16x8_t af[MI][2], wf0[2][2], wf1[2][2];

    auto read_a = [&](int slot) {
        const unsigned char* sb = smem + slot * HALF_BYTES + a_rd;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8_t*>(sb + i * 2048 + ch0);
            af[i][1] = *reinterpret_cast<const bf16x8_t*>(sb + i * 2048 + ch1);
        }
    };
    auto read_w = [&](int slot, bf16x8_t (&wf)[2][2]) {
        const unsigned char* sb = smem + slot * HALF_BYTES + w_rd;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf[j][0] = *reinterpret_cast<const bf16x8_t*>(sb + j * 2048 + ch0);
            wf[j][1] = *reinterpret_cast<const bf16x8_t*>(sb + j * 2048 + ch1);
        }
    };
    auto mma = [&](f32x4_t (&c)[2][MI], const bf16x8_t (&wf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    c[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[i][kk], c[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- output of one accumulator quadrant of tile (em0, en0), then reset of the quadrant.
    // acc[qm][qn][j][i][r] is C[em0 + qm*MH + grp*MH/2 + i*16 + l15][en0 + qn*128 + wc*32 + j*16 + lg*4 + r]
    f32x4_t bias4[2][2];                               // [qn][j] bias of the COMPUTE cursor's tile
    int em0 = 0, en0 = 0;                              // tile whose quadrants are being written
    auto load_bias = [&](int n0) {
        if (g.bias) {
            const float* base = g.bias + n0;                                 // wave-uniform
            const uint32_t voff = (uint32_t)(wc * 32 + lg * 4) * 4u;
            bias4[0][0] = asm_load_f32x4<0>(base, voff);
            bias4[0][1] = asm_load_f32x4<64>(base, voff);
            bias4[1][0] = asm_load_f32x4<512>(base, voff);
            bias4[1][1] = asm_load_f32x4<576>(base, voff);
        } else {
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int j = 0; j < 2; ++j) bias4[qn][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto flush_quadrant = [&](auto qm_c, auto qn_c) {
        constexpr int qm = decltype(qm_c)::value, qn = decltype(qn_c)::value;
        if constexpr (DBG & 2) return;
        const int col0 = en0 + qn * 128 + wc * 32 + lg * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = em0 + qm * MH + grp * (MH / 2) + i * 16 + l15;
            TOut* cp = C + (size_t)m * g.ldc + col0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[qm][qn][j][i][r] + bias4[qn][j][r]);
                uint2 t2;
                t2.x = pack2_16<TOut>(v[0], v[1]);
                t2.y = pack2_16<TOut>(v[2], v[3]);
                if constexpr (DBG & 1) asm volatile("" ::"v"(t2.x), "v"(t2.y));
                else if (m < g.M) *reinterpret_cast<uint2*>(cp + j * 16) = t2;
                acc[qm][qn][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    static_assert(D == 6 || D == 8, "prologue written for S = 8 / 10");

    // ---- prologue: the first D half tiles; the first two (A0, W0 of K tile 0) confirmed --------------------------
    // 4 * nk * n_mine >= 16 > D half tiles exist
    issue_next(I0{}); issue_next(I1{}); issue_next(I2{}); issue_next(I3{});
    issue_next(I0{}); issue_next(I1{});
    if constexpr (D == 8) { issue_next(I2{}); issue_next(I3{}); }
    p9_wait_vm<WAITN>();                                // 2 D loads issued, <= 2 (D - 2) outstanding: half tiles 0, 1 landed
    P9_BARRIER();
    if (grp == 1) P9_BARRIER();                         // group 1 runs one barrier behind

    int rd_slot = 0;                                    // slot of A0 of the compute cursor's K tile
    bool pend10 = false;                                // quadrant (1,0) of the previous tile still to be written
    auto slot_add = [&](int s, int d) { const int t = s + d; return t >= S ? t - S : t; };

    for (int ct = 0; ct < n_mine; ++ct) {
        int cm0, cn0;
        tile_coords(ct, cm0, cn0);
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt == nk - 1;
            // ---- P1: A0, W0 -> quadrant (0,0)
            read_w(slot_add(rd_slot, 1), wf0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(rd_slot);
            issue_and_wait(I0{});
            if (kt == 0) {
                if (pend10) { flush_quadrant(I1{}, I0{}); pend10 = false; }     // still the previous tile's coordinates / bias
                em0 = cm0; en0 = cn0;
                load_bias(cn0);
            }
            P9_BARRIER();
            mma(acc[0][0], wf0);
            P9_BARRIER();
            // ---- P2: W1 -> quadrant (0,1)
            read_w(slot_add(rd_slot, 2), wf1);
            issue_and_wait(I1{});
            if (last) flush_quadrant(I0{}, I0{});
            P9_BARRIER();
            mma(acc[0][1], wf1);
            P9_BARRIER();
            // ---- P3: A1 -> quadrant (1,1)
            read_a(slot_add(rd_slot, 3));
            issue_and_wait(I2{});
            if (last) flush_quadrant(I0{}, I1{});
            P9_BARRIER();
            mma(acc[1][1], wf1);
            P9_BARRIER();
            // ---- P4: quadrant (1,0)
            issue_and_wait(I3{});
            if (last) flush_quadrant(I1{}, I1{});
            P9_BARRIER();
            mma(acc[1][0], wf0);
            P9_BARRIER();
            rd_slot = slot_add(rd_slot, 4);
            if (last) pend10 = true;
        }
    }
    flush_quadrant(I1{}, I0{});                          // the last tile's fourth quadrant
    if (grp == 0) P9_BARRIER();
}

}  // namespace

bool gemm_p9_supports(const GemmArgs& g) {
    return g.K % BK == 0 && g.K >= 4 * BK && g.N % BN == 0 && g.res == nullptr && g.ldc % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(g.C) & 7) == 0 && (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) &&
           (double)g.M * g.lda * 2.0 < 4.0e9 && (double)g.N * g.K * 2.0 < 4.0e9;
}

// tiles of an XCD under the partition of kernels_gemm10.hip (ng N-groups x 8/ng M-groups)
static void p9_plan(const GemmArgs& g, int mh, int* ng_out, int* max_cnt_out) {
    const int tiles_m = (g.M + 2 * mh - 1) / (2 * mh), tiles_n = g.N / BN;
    int best_ng = 1, best_cnt = 1 << 30, best_panels = 1 << 30;
    for (int ng = 1; ng <= 8 && ng <= tiles_n; ng *= 2) {
        const int mg = 8 / ng;
        int max_cnt = 0, max_nn = 0;
        for (int x = 0; x < 8; ++x) {
            const int gn = x % ng, gm = x / ng;
            const int nn = (gn + 1) * tiles_n / ng - gn * tiles_n / ng;
            const int tg = tiles_m * nn;
            const int cnt = (gm + 1) * tg / mg - gm * tg / mg;
            max_cnt = cnt > max_cnt ? cnt : max_cnt;
            max_nn = nn > max_nn ? nn : max_nn;
        }
        const int rounds = (max_cnt + 31) / 32, best_rounds = (best_cnt + 31) / 32;
        const int panels = (32 + max_nn - 1) / max_nn + (max_nn < 32 ? max_nn : 32);
        if (rounds < best_rounds || (rounds == best_rounds && panels < best_panels)) {
            best_ng = ng; best_cnt = max_cnt; best_panels = panels;
        }
    }
    *ng_out = best_ng;
    *max_cnt_out = best_cnt;
}

template <typename TOut, int MH, int S>
static void launch_p9_t(const GemmArgs& g, int nwg, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_p9_kernel<TOut, GITMI_ACT_QUICKGELU, MH, S>), dim3(nwg), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_p9_kernel<TOut, GITMI_ACT_GELU_ERF, MH, S>), dim3(nwg), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_p9_kernel<TOut, GITMI_ACT_NONE, MH, S>), dim3(nwg), dim3(512), 0, s, g); break;
    }
}

// dbg bits (A/B, tests): 64 / 128 force the 192- / 256-row tile; 512 the 8-slot ring (128 KiB, four half tiles in flight);
// bits 12.. : workgroups per XCD (default 32 = one per CU); 1 / 2 (bf16, no activation, 256-row tile, 10 slots only):
// measurement builds without output stores / without any quadrant output
hipError_t launch_gemm_p9(GemmArgs g, hipStream_t s) {
    // the 192-row tile also on a tie when an activation is fused: the 256-row tile + activation temporaries does not fit
    // the register file without scratch, and hipcc waits vmcnt(0) for its scratch reloads -- a drained DMA queue per tile
    const int c96 = gemm_p8_cost(g, 96), c128 = gemm_p8_cost(g, 128);
    int mh = (c96 < c128 || (c96 == c128 && g.act != GITMI_ACT_NONE)) ? 96 : 128;
    if (g.dbg & 64) mh = 96;
    if (g.dbg & 128) mh = 128;
    const bool ring8 = (g.dbg & 512) != 0;
    int wpx = (g.dbg >> 12) & 63;
    if (wpx <= 0 || wpx > 32) wpx = 32;
    g.tiles_n = g.N / BN;
    int max_cnt = 0;
    p9_plan(g, mh, &g.ng, &max_cnt);
    if (max_cnt < wpx) wpx = max_cnt;
    const int nwg = 8 * wpx;
    g.nwg = nwg;
    const bool f16 = g.out_f16 != 0;
    if ((g.dbg & 3) && !f16 && g.act == GITMI_ACT_NONE && mh == 128 && !ring8) {        // measurement builds
        if (g.dbg & 2) hipLaunchKernelGGL((gemm_p9_kernel<bf16_t, GITMI_ACT_NONE, 128, 10, 2>), dim3(nwg), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_p9_kernel<bf16_t, GITMI_ACT_NONE, 128, 10, 1>), dim3(nwg), dim3(512), 0, s, g);
        return hipGetLastError();
    }
#define GITMI_P9(MHH, SS)                                             \
    do {                                                              \
        if (f16) launch_p9_t<f16_t, MHH, SS>(g, nwg, s);              \
        else launch_p9_t<bf16_t, MHH, SS>(g, nwg, s);                 \
    } while (0)
    if (mh == 96) { if (ring8) GITMI_P9(96, 8); else GITMI_P9(96, 10); }
    else { if (ring8) GITMI_P9(128, 8); else GITMI_P9(128, 10); }
#undef GITMI_P9
    return hipGetLastError();
}

}  // namespace gitmi
