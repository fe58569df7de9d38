// bf16 GEMM for the large-M phases -- eleventh generation ("p10"): the 256x256x64 half-tile LDS-DMA pipeline of
// kernels_gemm10.hip as a PERSISTENT workgroup whose K loop streams across tile boundaries.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N])      16-bit outputs, no residual;  K % 128 == 0, K >= 256, N % 256 == 0
//
// Why (profiles/r05_b_gemm_probe_clock_and_sections.txt: the workgroups' own s_memtime / s_memrealtime stamps): a 256-row
// tile of a K = 768 GEMM spends 3 400 cycles between its entry and its first MFMA, 30 650 in the K loop (2 554 per K tile,
// whatever the number of concurrent tiles -- the chip answers load with its CLOCK: 2.39 GHz for 9 tiles, 1.9 GHz for 450,
// 1.59 GHz when all 256 CUs sit in their K loops at once) and 10 600 in the epilogue, 6 800 of them waiting for the
// acknowledgement of its own stores before it may end.  With one tile per workgroup the next workgroup of the CU cannot start
// before that, and pays the 3 400 cycles again.  Here a workgroup walks a list of tiles:
//
//   * the half-tile stream never stops at a tile boundary: the issue slots the last two K tiles of a tile leave empty
//     request the first six half tiles of the NEXT tile (exactly the steady-state pattern with another source pointer),
//     and the two remaining half tiles of its second K tile follow as soon as the last fragment reads have retired;
//   * the epilogue does not touch the pipeline's LDS: accumulators leave through a separate 32-KiB slab (64 rows x 256
//     columns, 16-byte chunks XOR-swizzled by the row so that neither the 8-byte fragment writes nor the 16-byte row reads
//     conflict) in four passes, while the next tile's eight half tiles land in the 128 KiB beside it;
//   * the output stores are issued AFTER those loads.  vmcnt retires in order on gfx9, so the first K tiles of the next tile
//     wait with `vmcnt(S + n)` (S = stores of the epilogue, a compile-time constant) for loads that are OLDER than the
//     stores: the first wait that needs the stores' acknowledgement is six phases (about 2 us) into the next tile.  (Round 3's
//     persistent kernel requested the next tile's loads after the stores and stalled 7 us per tile on them.)
//   * a tile with rows beyond M predicates stores away, so its S is not the constant: it drains `vmcnt(0)` after its
//     epilogue (one M tile in fifty).
//
// Everything else -- LDS image, fragment addressing, phases, wave groups staggered by a barrier, bias / activation
// arithmetic -- is kernels_gemm10.hip's, and the outputs are bit-identical to it (same K order per output element).
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

namespace p10 {

constexpr int BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;             // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;            // 64 KiB: A0 A1 B0 B1
constexpr int PIPE_BYTES = 2 * BUF_BYTES;            // 128 KiB
constexpr int SLAB_BYTES = 64 * 512;                 // 32 KiB epilogue slab: 64 rows x 256 16-bit columns
constexpr int SLOT_A0 = 0, SLOT_B0 = 2 * HALF_BYTES;

typedef __attribute__((address_space(3))) void lds_void_t;

#define P10_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// orders this workgroup's LDS traffic only (no vmcnt: loads and stores stay in flight across it)
#define P10_LDS_BARRIER()                                  \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)

template <int N> __device__ __forceinline__ void wait_vmc() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

// DBG (measurement builds): 1 no global stores, 32 every tile drains its stores before the next K loop starts (the A/B of
// the store ordering: what the persistent form is worth without it)
template <int ACT, int DBG, int MH, int MH1>
static __global__ __launch_bounds__(512) void gemm_p10_kernel(GemmArgs g) {
    static_assert(MH == 128 && MH1 % 32 == 0 && MH1 <= MH, "first half tile 128 rows");
    constexpr int BM = MH + MH1, MI = MH / 32, MI1 = MH1 / 32;
    constexpr int S = 2 * (MI + MI1);                  // output stores a wave issues per tile (16-byte rows of the slab)
    __shared__ __attribute__((aligned(16))) unsigned char smem[PIPE_BYTES + SLAB_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- tile list of this workgroup: the XCD's chunk of its group's (M-major, N-fastest) list, every W-th tile
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int ng = g.ng, mg = 8 / ng;
    const int gn = x % ng, gm = x / ng;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int n_lo = gn * g.tiles_n / ng, n_hi = (gn + 1) * g.tiles_n / ng;
    const int nn = n_hi - n_lo;
    const int tg = tiles_m * nn;
    const int lo_t = gm * tg / mg, hi_t = (gm + 1) * tg / mg;
    int L = lo_t + idx;
    if (L >= hi_t) return;

    const char* __restrict__ Ab = reinterpret_cast<const char*>(g.A);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(g.W);
    bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(g.C);

    // ---- staging sources (kernels_gemm10.hip): this wave owns pieces 2*wave + q of every half tile
    // (lane-derived terms are recomputed from an opaque copy of the lane id wherever a tile's offsets or its epilogue
    // addresses are formed: hoisted out of the tile loop they would stay live across every K loop, and the kernel has no
    // registers to spare -- 128 accumulators + 64 fragment registers per lane)
    uint32_t a_off[2][2], w_off[2][2];          // byte offsets [half][q] of the tile whose half tiles are being requested
    auto offsets = [&](int tile, uint32_t (&ao)[2][2], uint32_t (&wo)[2][2]) {
        const int m0 = (tile / nn) * BM, n0 = (n_lo + tile % nn) * BN;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int Rl = ln >> 4, hi = (ln >> 3) & 1, lo = ln & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rr = h * MH + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ q);
                const int ch = lo ^ (q * 4 + Rl);
                int m = m0 + rr;
                m = m < g.M ? m : g.M - 1;
                ao[h][q] = ((uint32_t)m * (uint32_t)g.lda + ch * 8) * 2u;
                const int rn = h * 128 + (wave * 2 + q) * 8 + 2 * Rl + (hi ^ q);
                wo[h][q] = ((uint32_t)(n0 + rn) * (uint32_t)g.K + ch * 8) * 2u;
            }
    };
    auto issue = [&](int isw, int half, int kt) {
        const char* src = (isw ? Wb : Ab) + (size_t)kt * (BK * 2);
        unsigned char* dst = smem + (kt & 1) * BUF_BYTES + (isw ? SLOT_B0 : SLOT_A0) + half * HALF_BYTES + wave * 2048;
        const uint32_t o0 = isw ? w_off[half][0] : a_off[half][0];
        const uint32_t o1 = isw ? w_off[half][1] : a_off[half][1];
        __builtin_amdgcn_global_load_lds((const void*)(src + o0), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(src + o1), (lds_void_t*)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment addressing
    const int rowpart = (l15 >> 1) * 256 + ((l15 & 1) ^ ((l15 >> 3) & 1)) * 128;
    const int x7 = (l15 >> 1) & 7;
    const int ch0 = ((0 * 4 + lg) ^ x7) * 16;
    const int ch1 = ((1 * 4 + lg) ^ x7) * 16;
    const int a_rd = grp * (MH / 2) * 128 + rowpart;
    const int a_rd1 = grp * (MH1 / 2) * 128 + rowpart + HALF_BYTES;
    const int w_rd = SLOT_B0 + wc * 32 * 128 + rowpart;

    f32x4_t acc[2][2][2][MI];  // [qm][qn][j][i]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[a][b][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    bf16x8_t af[MI][2], wf0[2][2], wf1[2][2];
    constexpr std::integral_constant<int, 0> H0{};
    constexpr std::integral_constant<int, 1> H1{};
    auto read_a = [&](const unsigned char* sb, auto half_c) {
        constexpr int half = decltype(half_c)::value;
        const int base = half ? a_rd1 : a_rd;
#pragma unroll
        for (int i = 0; i < (half ? MI1 : MI); ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8_t*>(sb + base + i * 2048 + ch0);
            af[i][1] = *reinterpret_cast<const bf16x8_t*>(sb + base + i * 2048 + ch1);
        }
    };
    auto read_w = [&](const unsigned char* sb, int half, bf16x8_t (&wf)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf[j][0] = *reinterpret_cast<const bf16x8_t*>(sb + w_rd + half * HALF_BYTES + j * 2048 + ch0);
            wf[j][1] = *reinterpret_cast<const bf16x8_t*>(sb + w_rd + half * HALF_BYTES + j * 2048 + ch1);
        }
    };
    auto mma = [&](f32x4_t (&c)[2][MI], const bf16x8_t (&wf)[2][2], auto half_c) {
        constexpr int MIq = decltype(half_c)::value ? MI1 : MI;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MIq; ++i)
                    c[j][i] = mfma16(wf[j][kk], af[i][kk], c[j][i]);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K tile = four phases (kernels_gemm10.hip): P1 reads B0, A0 and requests B1 of the next K tile; P2 reads B1, requests
    // A1 of the next; P3 reads A1, requests A0 of the K tile after next; P4 requests B0 of that one.  The workgroup's K tiles
    // form ONE stream across its tiles, so "next" may belong to the next tile (t + 1 >= nk: its K tile t + 1 - nk, through
    // that tile's offsets, which replace the current ones between P2 and P3 of K tile nk - 2).  A tile's first K tile finds
    // its B1 / A1 requests already made (the prologue / the previous tile's last phases and the two requests behind its K
    // loop), so P1 and P2 of t = 0 request nothing.  The last tile of the list requests its own first K tiles again
    // instead of nothing (never read: the steady-state vmcnt arithmetic then needs no drain modes).
    // Waits: `vmcnt(8)` = all but the last four half tiles, except behind an epilogue, whose S stores sit in the queue
    // between the tile's first eight half tiles (older) and what its K loop requests (younger):
    //        t = 0:  P1 10 + S   P2 8 + S   P3 8 + S   P4 8 + S        t = 1:  P1 8 + S   P2 8 + S   P3 8   P4 8
    // (first tile of the list: S = 0).  Every wait names the half tile the NEXT phase reads; the first one that needs the
    // stores' acknowledgement is P3 of t = 1, six phases into the tile.
    // PAR = t & 1 = the LDS buffer, a compile-time value at every call site (nk is even): left to the run time, the
    // fragment addresses of both buffers are formed ahead of the tile loop and spilled.
    const int nk = g.K / BK;                                       // even, >= 4 (launcher)
    bool first = true;                                             // no epilogue's stores in the queue yet
    int next_tile = 0;
    auto wait_phase = [&](int t, int p) {
        constexpr int SV = (DBG & 32) ? 0 : S;
        if (t == 0 && p == 1) {
            if (first) wait_vmc<10>(); else wait_vmc<SV + 10>();
        } else if (!first && (t == 0 || (t == 1 && p <= 2))) {
            wait_vmc<SV + 8>();
        } else {
            wait_vmc<8>();
        }
    };
    auto request = [&](int isw, int half, int kt) { issue(isw, half, kt >= nk ? kt - nk : kt); };
    auto ktile = [&](auto par_c, int t) {
        constexpr int PAR = decltype(par_c)::value;
        const unsigned char* sb = smem + PAR * BUF_BYTES;
        // ---- P1
        read_w(sb, 0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(sb, H0);
        if (t != 0) request(1, 1, t + 1);
        wait_phase(t, 1);
        P10_BARRIER();
        mma(acc[0][0], wf0, H0);
        P10_BARRIER();
        // ---- P2
        read_w(sb, 1, wf1);
        if (t != 0) request(0, 1, t + 1);
        wait_phase(t, 2);
        P10_BARRIER();
        mma(acc[0][1], wf1, H0);
        P10_BARRIER();
        // ---- P3
        read_a(sb, H1);
        if (t == nk - 2) offsets(next_tile, a_off, w_off);       // the current tile's last request is behind us
        request(0, 0, t + 2);
        wait_phase(t, 3);
        P10_BARRIER();
        mma(acc[1][1], wf1, H1);
        P10_BARRIER();
        // ---- P4
        request(1, 0, t + 2);
        wait_phase(t, 4);
        P10_BARRIER();
        mma(acc[1][0], wf0, H1);
        P10_BARRIER();
    };

    unsigned char* slab = smem + PIPE_BYTES;

    // ---- first tile: its first two K tiles are requested whole
    offsets(L, a_off, w_off);
    issue(0, 0, 0); issue(1, 0, 0); issue(1, 1, 0); issue(0, 1, 0); issue(0, 0, 1); issue(1, 0, 1); issue(1, 1, 1); issue(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");              // A0(0), B0(0) of this wave have landed
    P10_BARRIER();
    for (;;) {
        const int Ln = L + W;
        const bool has_next = Ln < hi_t;
        next_tile = has_next ? Ln : L;
        const int tile_m = L / nn, tile_n = n_lo + L % nn;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        if (grp == 1) P10_BARRIER();                               // group 1 runs one barrier behind
        for (int t = 0; t < nk; t += 2) {
            ktile(H0, t);
            ktile(H1, t + 1);
        }
        if (grp == 0) P10_BARRIER();                               // both groups have retired their last fragment reads
        issue(1, 1, 1);                                            // the rest of the next tile's second K tile
        issue(0, 1, 1);

        // ---- epilogue through the slab: pass (qm, ih) = rows  qm*MH + grp*MHq/2 + (2 ih + f)*16 + [0,16)  for f = 0, 1
        // bias of the tile's columns through the SCALAR cache: the 16 columns of a (qn, j) fragment are wave-uniform, a lane
        // keeps the four of its lane group.  A vector load here would sit in the vmcnt queue between the next tile's loads and
        // this tile's stores, and hipcc waits vmcnt(0) -- the whole prefetch -- before the first use of an ordinary load's
        // result while LDS-DMA loads are in flight.
        int te = tid;
        asm volatile("" : "+v"(te));
        const int l15 = te & 15, lg = (te >> 4) & 3;                // shadow the kernel-scope copies: recomputed per tile
        f32x4_t bias4[2][2];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bias4[qn][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (g.bias) {
                    const float* bp = g.bias + n0 + qn * 128 + wc * 32 + j * 16;          // wave-uniform
                    f32x4_t s0, s1, s2, s3;
                    asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                                 "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                                 : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "s"(bp) : "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias4[qn][j][r] = lg == 0 ? s0[r] : lg == 1 ? s1[r] : lg == 2 ? s2[r] : s3[r];
                }
            }
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                const int MIq = qm ? MI1 : MI, MHq = qm ? MH1 : MH;       // constants once the loops are unrolled
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int i = 2 * ih + f;
                    if (i >= MIq) continue;
                    const int R = grp * 32 + f * 16 + l15;                    // slab row
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[qm][qn][j][i][r] + bias4[qn][j][r]);
                            const int chunk = qn * 16 + wc * 4 + j * 2 + (lg >> 1);
                            uint2 t2;
                            t2.x = pack2bf(v[0], v[1]);
                            t2.y = pack2bf(v[2], v[3]);
                            *reinterpret_cast<uint2*>(slab + R * 512 + ((chunk ^ (R & 15)) << 4) + (lg & 1) * 8) = t2;
                        }
                }
                P10_LDS_BARRIER();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = q & 1, gq = q >> 1;                         // rows q*16 .. : (group gq, fragment f)
                    if (2 * ih + f >= MIq) continue;
                    const int R = q * 16 + (te >> 5), cc = te & 31;
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(slab + R * 512 + ((cc ^ (R & 15)) << 4));
                    const int m = m0 + qm * MH + gq * (MHq / 2) + (2 * ih + f) * 16 + (te >> 5);
                    bf16_t* cp = C + (size_t)m * g.ldc + n0 + cc * 8;
                    if (m < g.M && !(DBG & 1))
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cp), "v"(v) : "memory");
                }
                P10_LDS_BARRIER();
            }
        if (!has_next) break;
        // a tile with rows beyond M (or an ablation) did not issue exactly S stores: drain, so that the counted waits of
        // the next tile's first K tiles -- which assume S stores behind the loads -- can only be too strict, never too lax
        if (m0 + BM > g.M || (DBG & 1) || (DBG & 32)) wait_vmc<0>();
        zero_acc();
        L = Ln;
        first = false;
    }
    wait_vmc<0>();                                                 // the last tile's unread requests
}

}  // namespace p10

bool gemm_p10_supports(const GemmArgs& g, bool out_f32) {
    return !out_f32 && !g.out_f16 && g.res == nullptr && g.K % (2 * p10::BK) == 0 && g.K >= 4 * p10::BK && g.N % p10::BN == 0 &&
           (double)g.M * g.lda * 2.0 < 4.0e9 && (double)g.N * g.K * 2.0 < 4.0e9;
}

template <int DBG, int MH1>
static void launch_p10_act(const GemmArgs& g, hipStream_t s) {
    switch (g.act) {
        case GITMI_ACT_QUICKGELU: hipLaunchKernelGGL((p10::gemm_p10_kernel<GITMI_ACT_QUICKGELU, DBG, 128, MH1>), dim3(g.nwg), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF: hipLaunchKernelGGL((p10::gemm_p10_kernel<GITMI_ACT_GELU_ERF, DBG, 128, MH1>), dim3(g.nwg), dim3(512), 0, s, g); break;
        default: hipLaunchKernelGGL((p10::gemm_p10_kernel<GITMI_ACT_NONE, DBG, 128, MH1>), dim3(g.nwg), dim3(512), 0, s, g); break;
    }
}

// bm: 256 or 224 rows; g.ng / g.tiles_n / the XCD's tile count come from the one-tile kernel's planner (p8_plan)
hipError_t launch_gemm_p10(GemmArgs g, int bm, int max_cnt, hipStream_t s) {
    g.nwg = 8 * (max_cnt < 32 ? max_cnt : 32);
    const int dbg = g.dbg;
    g.dbg = 0;
    if (bm == 224) {
        if (dbg == 1) launch_p10_act<1, 96>(g, s);
        else if (dbg == 32) launch_p10_act<32, 96>(g, s);
        else launch_p10_act<0, 96>(g, s);
    } else {
        if (dbg == 1) launch_p10_act<1, 128>(g, s);
        else if (dbg == 32) launch_p10_act<32, 128>(g, s);
        else launch_p10_act<0, 128>(g, s);
    }
    return hipGetLastError();
}

}  // namespace gitmi
