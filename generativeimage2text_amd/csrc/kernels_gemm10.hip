// bf16 GEMM for the large-M phases -- tenth generation: 256x256x64 tile, 8 waves, half-tile granular LDS-DMA
// pipeline in four phases per K tile, the two wave groups of a workgroup staggered by one barrier.
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N])      K % 64 == 0, K >= 128, N % 256 == 0
//
// Why: a 256x128 tile (the ring kernels of rounds 1-2, removed in round 4) needs one operand byte from L2 per 85 FLOP
// and saturates near 12 TB/s of operand feed.  A 256x256 tile needs one byte per 128 FLOP; the 128 KiB of LDS
// that leaves room for only two K tiles are recycled at HALF-TILE granularity so that four half tiles
// (64 KiB) are always in flight.
//
//   * 512 threads = 8 waves.  wave = 4*grp + wc.  A wave owns 128x64 of the output as FOUR quadrants of
//     64x32:  rows  qm*128 + grp*64 + [0,64),  columns  qn*128 + wc*32 + [0,32)   (qm, qn in {0,1}),
//     so that quadrant (qm,qn) of EVERY wave reads activation half qm (rows qm*128..) and weight half qn.
//     MFMA 16x16x32 bf16 in swapped orientation (accumulator = C^T, see kernels_gemm.hip).
//   * LDS: 2 K tiles x [A0 | A1 | B0 | B1], each half tile = 128 rows x 128 B = 16 KiB in the bank-conflict
//     free image below; a wave fills 2 KiB of every half tile (2 global_load_lds_dwordx4).
//     LDS image: 128-byte rows paired into 256-byte bank rows; 16-byte chunk c of row r lives at
//         (r>>1)*256 + ((r&1) ^ ((r>>3)&1))*128 + (c ^ ((r>>1)&7))*16
//     which makes every ds_read_b128 lane group hit 16 distinct bank slots.  A direct-to-LDS load writes lane-linearly,
//     so the permutation is applied to each lane's SOURCE address (and again on the fragment read).
//   * K tile t, phase p = 1..4, each phase = [ds_reads, one half-tile prefetch, counted vmcnt] s_barrier
//     [16 MFMAs] s_barrier:
//         P1: read B0,A0 (12 ds_read_b128)   prefetch B1(t+1)   MFMA quadrant (0,0)
//         P2: read B1    (4)                 prefetch A1(t+1)   MFMA quadrant (0,1)
//         P3: read A1    (8)                 prefetch A0(t+2)   MFMA quadrant (1,1)
//         P4: -                              prefetch B0(t+2)   MFMA quadrant (1,0)
//     A half tile is refilled no earlier than two phases after the phase that read it (the other wave group
//     runs one barrier behind), and `s_waitcnt vmcnt(8)` at the end of a read segment confirms the half tile
//     issued four phases ago -- the one the NEXT phase reads.  Group 1 executes one extra barrier up front:
//     while one group issues MFMAs (s_setprio 1) the other one's ds_reads and LDS-DMA issues are in flight.
//   * epilogue: bf16 outputs go through LDS, one 128-row slab at a time (16-byte row-contiguous stores); fp32
//     outputs (the N = 768 GEMMs with the fp32 residual stream) leave the accumulators directly: a lane holds four
//     consecutive columns of one row, so its residual read and its store are 16-byte accesses and a wave instruction
//     covers 16 rows x 64 B; every residual load of an MH-row half of the tile is requested before the first is consumed.
//     No LDS round trip and no barriers, so the waves of a tile drain independently.  Measured (profiles/r02_g_epi_ab.txt):
//     isolated launches are not faster (a quarter wave touches 16 rows: 4x the requests of the staged copy loop; the
//     phase is bound by the chip-wide read-modify-write of the residual stream anyway), the whole pass in situ is 1.4 %.
//   * LayerNorm folding (round 6, template parameter LNF, fp16-operand build): the LayerNorm modules between these GEMMs
//     (CLIP/model.py:189-202 ln_1 / ln_2; modeling_bert.py:171-178, 243-250 over the image rows) live in the epilogues --
//     a PRODUCER of stream rows leaves (sum, sumsq) per row and 256-column tile, a CONSUMER multiplies the raw rows by
//     W . gamma and applies rstd (acc - mean colsum) + (beta W^T + b).  See the comment at the kernel; K loop unchanged.
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

namespace {

constexpr int BN = 256, BK = 64;                    // BM = 2 * MH, MH = 128 (256x256 tile) or 96 (192x256)
constexpr int HALF_BYTES = 128 * BK * 2;             // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;            // 64 KiB: A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF_BYTES;             // 128 KiB
constexpr int SLOT_A0 = 0, SLOT_A1 = HALF_BYTES, SLOT_B0 = 2 * HALF_BYTES, SLOT_B1 = 3 * HALF_BYTES;

typedef __attribute__((address_space(3))) void lds_void_t;

#define P8_BARRIER()                           \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// a + b rounded on its own: never fused with the multiply that produced `a` (keeps act(x) + residual bitwise equal to
// the kernels that pass act(x) through memory first)
__device__ __forceinline__ float add_unfused(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// Sums of two values over each 32-lane half of a wave, valid in lanes 16..31 / 48..63: five v_add_f32 with a DPP source per
// value (two quad permutes, two row rotations, then row 0's / row 2's total broadcast into the row above it) -- no LDS
// traffic, no v_mov per step (update_dpp + add compiled to v_mov + v_mov_dpp + s_nop + v_add per step: profiles/r06_i_*).  The two
// chains are interleaved; a DPP read needs two wait states after the VALU write of its source (the other chain's add + s_nop 0).
// The summation order is fixed.
__device__ __forceinline__ void half_wave_sum2_hi(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"                  // the compiler does not know these are DPP reads: its own writes of a / b may be 1 instruction old
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}

// epilogue barrier: orders this workgroup's LDS traffic only.  __syncthreads() would also wait for every outstanding
// global store of the wave (vmcnt(0)) -- the slab just written out -- before the next slab may even be staged.
#define P8_LDS_BARRIER()                                   \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)

// 16-bit output types: bf16_t (operands of the next GEMM) and f16_t (the residual stream in bf16 engine mode)
template <typename T> __device__ __forceinline__ uint32_t pack2o(float lo, float hi) {
    if constexpr (std::is_same<T, f16_t>::value) return pack2h(lo, hi);
    else return pack2bf(lo, hi);
}
template <typename T> __device__ __forceinline__ void unpack2o(uint32_t u, float& lo, float& hi) {
    if constexpr (std::is_same<T, f16_t>::value) unpack2h(u, lo, hi);
    else unpack2op(u, lo, hi);
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// DBG (measurement builds only): 1 no global stores / residual reads, 2 no epilogue, 4 no ds_reads / MFMAs, 8 no loads,
// 16 time stamps (tools/gemm_probe.hip): wave 0 writes s_memtime / s_memrealtime at entry, K-loop start, K-loop end and exit
// to ((uint64_t*)g.res)[blockIdx.x * 8 ..] -- 16-bit outputs without a residual only
// MH = rows of an activation half tile: 128 -> 256x256 tile; 96 -> 192x256 tile (3 instead of 4 row fragments per
// quadrant), used when it fills more CUs in a single round (N = 768: 198 instead of 150 workgroups).  The A slots keep
// their 16 KiB; with MH = 96 the last four 1-KiB pieces of a slot are loaded (clamped rows) but never read, so every
// wave still issues two loads per half tile and the vmcnt arithmetic is unchanged.
// MH1 (round 5) = rows of the SECOND activation half when it is shorter than the first: tiles of 224 (128 + 96), 160 (96 + 64)
// rows ... -- any multiple of 32.  The tile height is what decides how many workgroups a launch has, and on 256 CUs
// that decides the rounds: the N = 768 GEMMs are 150 tiles of 256 rows (59 % of the CUs for one long round) but 237 tiles
// of 160 rows (one round at 0.625 of the length).  Quadrants (1, *) then carry MH1 / 32 row fragments instead of MH / 32.
// EPI (fp32 outputs only): 1 direct epilogue from the accumulators, 0 the LDS-staged one (A/B: dbg bit 256)
// LNF (round 6, fp16-operand build): LayerNorm folded into the GEMMs either side of it.  1 = CONSUMER: A is the raw fp16 stream,
// W carries the gain, and the epilogue applies rstd_m (acc - mean_m colsum_n) + folded bias from the row partials the producer left.
// 2 = PRODUCER (fp16 stream rows out): per row and 256-column tile the (sum, sumsq) of the values as stored, and -- post-norm
// layers -- the residual rebuilt as LayerNorm(raw rows) from the previous partials.  No normalised tensor is materialised and the
// encoder / prefill passes lose their LayerNorm launches (36 -> 1 per request on GIT_BASE).
template <typename TOut, int ACT, int DBG = 0, int MH = 128, int EPI = 1, int MH1 = MH, int LNF = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs g) {
    static_assert(LNF == 0 || sizeof(TOut) == 2, "LayerNorm folding: 16-bit outputs only");
    static_assert(MH % 32 == 0 && MH1 % 32 == 0 && MH1 <= MH && MH <= 128, "half tiles: multiples of 32 rows, second <= first");
    constexpr int BM = MH + MH1, MI = MH / 32, MI1 = MH1 / 32;   // row fragments of a quadrant of half 0 / half 1
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    uint64_t ts[8];
    auto stamp = [&](int i) {
        if constexpr ((DBG & 16) != 0) {
            __builtin_amdgcn_sched_barrier(0);
            ts[2 * i] = __builtin_amdgcn_s_memtime();
            ts[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stamps_out = [&]() {
        if constexpr ((DBG & 16) != 0) {
            if (tid == 0) {
                uint64_t* o = reinterpret_cast<uint64_t*>(const_cast<float*>(g.res)) + (size_t)blockIdx.x * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = ts[i];
            }
        }
    };
    stamp(0);

    // ---- tile of this workgroup.  Workgroup b runs on XCD b % 8 (observed; only speed depends on it).  The N tiles
    // are cut into ng groups; the 8/ng XCDs of a group share its (M-major, N-fastest) tile list in contiguous,
    // equally long chunks, so that no XCD needs an extra round of its 32 CUs because of an uneven rectangular split
    // (450 tiles: 57 per XCD instead of 65 on one of them) while its tiles still share activation / weight panels.
    int tile_m, tile_n;
    {
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int ng = g.ng, mg = 8 / ng;
        const int gn = x % ng, gm = x / ng;
        const int tiles_m = (g.M + BM - 1) / BM;
        const int n_lo = gn * g.tiles_n / ng, n_hi = (gn + 1) * g.tiles_n / ng;
        const int nn = n_hi - n_lo;
        const int tg = tiles_m * nn;
        const int lo_t = gm * tg / mg, hi_t = (gm + 1) * tg / mg;
        if (idx >= hi_t - lo_t) return;                         // surplus workgroup
        const int L = lo_t + idx;
        tile_m = L / nn;
        tile_n = n_lo + L % nn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const char* __restrict__ Ab = reinterpret_cast<const char*>(g.A);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(g.W);

    // ---- staging sources.  A wave instruction fills 1 KiB = 4 bank rows = 8 tile rows; this wave owns
    // pieces P = 2*wave + q (q = 0,1) of every half tile.  lane -> bank row Rl, half hi, slot lo.
    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    uint32_t a_off[2][2], w_off[2][2];                  // byte offsets [half][q]
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
    // half tile `half` (0/1) of operand `isw` for K tile kt -> slot of buffer kt & 1
    auto issue = [&](int isw, int half, int kt) {
        if constexpr (DBG & 8) return;
        const char* src = (isw ? Wb : Ab) + (size_t)kt * (BK * 2);
        unsigned char* dst = smem + (kt & 1) * BUF_BYTES + (isw ? SLOT_B0 : SLOT_A0) + half * HALF_BYTES + wave * 2048;
        const uint32_t o0 = isw ? w_off[half][0] : a_off[half][0];
        const uint32_t o1 = isw ? w_off[half][1] : a_off[half][1];
        __builtin_amdgcn_global_load_lds((const void*)(src + o0), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(src + o1), (lds_void_t*)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment addressing (bank-conflict free image, see the header) -------------------------------
    const int rowpart = (l15 >> 1) * 256 + ((l15 & 1) ^ ((l15 >> 3) & 1)) * 128;
    const int x7 = (l15 >> 1) & 7;
    const int ch0 = ((0 * 4 + lg) ^ x7) * 16;
    const int ch1 = ((1 * 4 + lg) ^ x7) * 16;
    const int a_rd = grp * (MH / 2) * 128 + rowpart;                     // + half*HALF_BYTES + i*2048 + ch  (half 0)
    const int a_rd1 = grp * (MH1 / 2) * 128 + rowpart + HALF_BYTES;      //                     + i*2048 + ch  (half 1)
    const int w_rd = SLOT_B0 + wc * 32 * 128 + rowpart;            // + half*HALF_BYTES + j*2048 + ch

    f32x4_t acc[2][2][2][MI];  // [qm][qn][j: n-frag][i: m-frag]; zeroed right before the K loop (after the LNF 1 statistics
                               // have been reduced: their raw partials and the accumulators are never live together)

    bf16x8_t af[MI][2], wf0[2][2], wf1[2][2];

    auto read_a = [&](const unsigned char* sb, auto half_c) {
        if constexpr (DBG & 4) return;
        constexpr int half = decltype(half_c)::value;
        const int base = half ? a_rd1 : a_rd;
#pragma unroll
        for (int i = 0; i < (half ? MI1 : MI); ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8_t*>(sb + base + i * 2048 + ch0);
            af[i][1] = *reinterpret_cast<const bf16x8_t*>(sb + base + i * 2048 + ch1);
        }
    };
    constexpr std::integral_constant<int, 0> H0{};
    constexpr std::integral_constant<int, 1> H1{};
    auto read_w = [&](const unsigned char* sb, int half, bf16x8_t (&wf)[2][2]) {
        if constexpr (DBG & 4) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf[j][0] = *reinterpret_cast<const bf16x8_t*>(sb + w_rd + half * HALF_BYTES + j * 2048 + ch0);
            wf[j][1] = *reinterpret_cast<const bf16x8_t*>(sb + w_rd + half * HALF_BYTES + j * 2048 + ch1);
        }
    };
    auto mma = [&](f32x4_t (&c)[2][MI], const bf16x8_t (&wf)[2][2], auto half_c) {
        if constexpr (DBG & 4) return;
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

    // MODE 0: steady state (t <= nk-3)   MODE 1: t == nk-2   MODE 2: t == nk-1
    auto ktile = [&](auto mode_c, int t) {
        constexpr int MODE = decltype(mode_c)::value;
        const unsigned char* sb = smem + (t & 1) * BUF_BYTES;
        // ---- P1
        read_w(sb, 0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(sb, H0);
        if constexpr (MODE <= 1) { issue(1, 1, t + 1); wait_vm<8>(); } else { wait_vm<2>(); }
        P8_BARRIER();
        mma(acc[0][0], wf0, H0);
        P8_BARRIER();
        // ---- P2
        read_w(sb, 1, wf1);
        if constexpr (MODE <= 1) { issue(0, 1, t + 1); wait_vm<8>(); } else { wait_vm<0>(); }
        P8_BARRIER();
        mma(acc[0][1], wf1, H0);
        P8_BARRIER();
        // ---- P3
        read_a(sb, H1);
        if constexpr (MODE == 0) { issue(0, 0, t + 2); wait_vm<8>(); }
        P8_BARRIER();
        mma(acc[1][1], wf1, H1);
        P8_BARRIER();
        // ---- P4
        if constexpr (MODE == 0) { issue(1, 0, t + 2); wait_vm<8>(); }
        else if constexpr (MODE == 1) { wait_vm<4>(); }
        P8_BARRIER();
        mma(acc[1][0], wf0, H1);
        P8_BARRIER();
    };

    // LNF 1: (mean, rstd) of this lane's 2 x MI output rows.  The row partials ([row][4] (sum, sumsq), 32 B per row) travel the
    // way the tiles do: ONE direct-to-LDS load per wave, issued before the first half tiles -- wave (grp, wc) fetches the 2 x 16
    // rows of fragment i = wc of its group's rows -- into the A1 slot of buffer 1, which the pipeline does not refill before
    // phase 2 of K tile 0, two barriers after the last read below.  Being the oldest request it has landed when the
    // prologue's vmcnt(8) is over (a register load would make the compiler wait for vmcnt(0): every prefetch in flight).  After
    // the barrier lane group lg reduces rows (qm, i = lg) and the lane groups exchange (mean, rstd) by ds_bpermute (crossbar
    // only): 4 x MI registers through the K loop, ~40 instructions in front of it.
    float lmean[2][MI], lrstd[2][MI];
    unsigned char* const ln_lds = smem + BUF_BYTES + SLOT_A1;
    if constexpr (LNF == 1) {
        const int r = lane >> 1, qm = r >> 4, l = r & 15;                        // 32 rows x two 16-byte halves
        const int MHq = qm ? MH1 : MH, MIq = qm ? MI1 : MI;
        const int ii = wc < MIq ? wc : MIq - 1;
        const int m = m0 + qm * MH + grp * (MHq / 2) + ii * 16 + l;
        const int mc = m < g.M ? m : g.M - 1;
        const char* src = reinterpret_cast<const char*>(g.ln_part) + (size_t)mc * 32 + (lane & 1) * 16;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_t*)(ln_lds + wave * 1024), 16, 0, 0);
    }
    auto ln_rows = [&]() {
        if constexpr (LNF == 1) {
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                const int MIq = qm ? MI1 : MI;
                const int ii = lg < MIq ? lg : MIq - 1;
                const unsigned char* pr = ln_lds + (grp * 4 + ii) * 1024 + (qm * 16 + l15) * 32;
                const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(pr);
                const f32x4_t p1 = *reinterpret_cast<const f32x4_t*>(pr + 16);
                const float sx = (p0[0] + p0[2]) + (p1[0] + p1[2]);
                const float sq = (p0[1] + p0[3]) + (p1[1] + p1[3]);
                const float mean = sx * g.ln_inv_d;
                const float rstd = rsqrtf(fmaxf(sq * g.ln_inv_d - mean * mean, 0.f) + g.ln_eps);
#pragma unroll
                for (int i = 0; i < MIq; ++i) {
                    lmean[qm][i] = __shfl(mean, i * 16 + l15, 64);
                    lrstd[qm][i] = __shfl(rstd, i * 16 + l15, 64);
                    // pin the results HERE (first use is in the epilogue: the compiler would sink the arithmetic there)
                    asm volatile("" : "+v"(lmean[qm][i]), "+v"(lrstd[qm][i]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nk = g.K / BK;                                       // >= 2 (launcher)
    issue(0, 0, 0); issue(1, 0, 0); issue(1, 1, 0); issue(0, 1, 0); issue(0, 0, 1); issue(1, 0, 1);
    wait_vm<8>();                                                  // A0(0), B0(0) of this wave have landed (and, LNF 1, its row partials)
    P8_BARRIER();
    ln_rows();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[a][b][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (grp == 1) P8_BARRIER();                                    // group 1 runs one barrier behind
    stamp(1);
    for (int t = 0; t < nk - 2; ++t) ktile(std::integral_constant<int, 0>{}, t);
    ktile(std::integral_constant<int, 1>{}, nk - 2);
    ktile(std::integral_constant<int, 2>{}, nk - 1);
    if (grp == 0) P8_BARRIER();
    stamp(2);

    // ---- epilogue through LDS: slab = MH rows (qm) x WCOL columns ---------------------------------
    constexpr int NQN = sizeof(TOut) == 2 ? 2 : 1;                 // weight halves per slab
    constexpr int WCOL = 128 * NQN;
    constexpr int EPC = 16 / (int)sizeof(TOut);                    // elements per 16-byte chunk
    constexpr int EPS = WCOL + EPC;                                // padded row stride (elements)
    constexpr int CPR = WCOL / EPC;                                // chunks per row
    static_assert(MH * EPS * sizeof(TOut) <= LDS_BYTES && (MH * CPR) % 512 == 0 && (MH1 * CPR) % 512 == 0, "epilogue slab does not fit");
    TOut* ep = reinterpret_cast<TOut*>(smem);
    TOut* __restrict__ C = reinterpret_cast<TOut*>(g.C);

    if constexpr (DBG & 2) {
        float sum = 0.f;                                 // keeps every accumulator live
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) sum += acc[a][b][j][i][0] + acc[a][b][j][i][1] + acc[a][b][j][i][2] + acc[a][b][j][i][3];
        if (sum == 12345.678f) C[0] = (TOut)0;
        stamp(3);
        stamps_out();
        return;
    }
    f32x4_t bias4[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bias4[qn][j] = g.bias ? *reinterpret_cast<const f32x4_t*>(g.bias + n0 + qn * 128 + wc * 32 + j * 16 + lg * 4)
                                  : f32x4_t{0.f, 0.f, 0.f, 0.f};

    // folded LayerNorm of the A rows (LNF 1): column sums of this lane's columns ((mean, rstd) of its rows: taken before the K loop)
    f32x4_t cs4[2][2];
    if constexpr (LNF == 1) {
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                cs4[qn][j] = *reinterpret_cast<const f32x4_t*>(g.ln_colsum + n0 + qn * 128 + wc * 32 + j * 16 + lg * 4);
    }
    auto ln_apply = [&](float a, int qm, int qn, int j, int i, int r) -> float {
        if constexpr (LNF == 1) return lrstd[qm][i] * fmaf(-lmean[qm][i], cs4[qn][j][r], a);
        else return a;
    };

    if constexpr (sizeof(TOut) == 4 && EPI == 1) {
        // ---- direct fp32 epilogue.  acc[qm][qn][j][i][r] is C[m0 + qm*MH + grp*MH/2 + i*16 + l15][n0 + qn*128 +
        // wc*32 + j*16 + lg*4 + r]; same arithmetic order as the staged path (act(acc + bias), then + residual), so
        // the two are bitwise equal.  res may alias C (in-place residual stream): every element is read and written
        // by the same lane, the read first.
        float* Cf = reinterpret_cast<float*>(g.C);
        const float* R = g.res;
        const int col0 = n0 + wc * 32 + lg * 4;
        const bool has_res = R != nullptr && !(DBG & 1);
#pragma unroll
        for (int qm = 0; qm < 2; ++qm) {
            const int MIq = qm ? MI1 : MI, MHq = qm ? MH1 : MH;           // constants once the loop is unrolled
            f32x4_t rr[MI][2][2];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < MIq; ++i) {
                    const int m = m0 + qm * MH + grp * (MHq / 2) + i * 16 + l15;
                    const int mc = m < g.M ? m : g.M - 1;                      // clamped rows are loaded, never stored
                    const float* rp = R + (size_t)mc * g.ldr + col0;
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            rr[i][qn][j] = *reinterpret_cast<const f32x4_t*>(rp + qn * 128 + j * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < MIq; ++i) {
                const int m = m0 + qm * MH + grp * (MHq / 2) + i * 16 + l15;
                float* cp = Cf + (size_t)m * g.ldc + col0;
                const bool ok = m < g.M && !(DBG & 1);
#pragma unroll
                for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4_t v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[qm][qn][j][i][r] + bias4[qn][j][r]);
                        if (has_res) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = add_unfused(v[r], rr[i][qn][j][r]);
                        }
                        if (ok) *reinterpret_cast<f32x4_t*>(cp + qn * 128 + j * 16) = v;
                    }
            }
        }
        return;
    }

    float rgm[8], rbt[8];               // LNF 2, post-norm residual: gain / bias of this thread's 8 columns (the same in every slab row)
    if constexpr (LNF == 2) {
        if (g.res_part) {
            ld8(g.res_gamma + n0 + (tid & 31) * 8, rgm);
            ld8(g.res_beta + n0 + (tid & 31) * 8, rbt);
        }
    }
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int s = 0; s < 2 / NQN; ++s) {
            const int MIq = qm ? MI1 : MI, MHq = qm ? MH1 : MH;           // constants once the loops are unrolled
#pragma unroll
            for (int u = 0; u < NQN; ++u) {
                const int qn = s * NQN + u;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nl = u * 128 + wc * 32 + j * 16 + lg * 4;
#pragma unroll
                    for (int i = 0; i < MIq; ++i) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(ln_apply(acc[qm][qn][j][i][r], qm, qn, j, i, r) + bias4[qn][j][r]);
                        TOut* p = ep + (grp * (MHq / 2) + i * 16 + l15) * EPS + nl;
                        if constexpr (sizeof(TOut) == 4) {
                            *reinterpret_cast<f32x4_t*>(p) = f32x4_t{v[0], v[1], v[2], v[3]};
                        } else {
                            uint2 t2;
                            t2.x = pack2o<TOut>(v[0], v[1]);
                            t2.y = pack2o<TOut>(v[2], v[3]);
                            *reinterpret_cast<uint2*>(p) = t2;
                        }
                    }
                }
            }
            P8_LDS_BARRIER();
#pragma unroll 4
            for (int q = 0; q < MHq * CPR / 512; ++q) {
                const int chunk = tid + q * 512;
                const int row = chunk / CPR, cc = chunk % CPR;
                const int m = m0 + qm * MH + row;
                const int n = n0 + s * WCOL + cc * EPC;
                if (m < g.M && !(DBG & 1)) {
                    if constexpr (sizeof(TOut) == 4) {
                        f32x4_t v = *reinterpret_cast<const f32x4_t*>(ep + row * EPS + cc * EPC);
                        if (g.res && !(DBG & 16)) {
                            const f32x4_t rr = *reinterpret_cast<const f32x4_t*>(g.res + (size_t)m * g.ldr + n);
                            v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
                        }
                        *reinterpret_cast<f32x4_t*>(C + (size_t)m * g.ldc + n) = v;
                    } else if constexpr (LNF == 2) {
                        // handled below (every lane takes part in the row reductions, rows past M included)
                    } else {
                        u32x4_t v = *reinterpret_cast<const u32x4_t*>(ep + row * EPS + cc * EPC);
                        if (g.res && !(DBG & 16)) {
                            float f[8], rs[8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) unpack2o<TOut>(v[e], f[2 * e], f[2 * e + 1]);
                            if constexpr (std::is_same<TOut, f16_t>::value) {
                                // the residual IS the fp16 stream (g.res points at f16 rows; may alias C: same thread reads, then writes)
                                const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const f16_t*>(g.res) + (size_t)m * g.ldr + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) unpack2h(rr[e], rs[2 * e], rs[2 * e + 1]);
                            } else {
                                const float* rp = g.res + (size_t)m * g.ldr + n;
                                const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(rp);
                                const f32x4_t r1 = *reinterpret_cast<const f32x4_t*>(rp + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { rs[e] = r0[e]; rs[4 + e] = r1[e]; }
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += rs[e];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = pack2o<TOut>(f[2 * e], f[2 * e + 1]);
                        }
                        // sc1: write-through WITHOUT keeping the line in this XCD's L2 (MI355X_MICROARCH.md, store flavours).  The
                        // 58 - 78 MB a launch writes are never re-read by it; left in the L2 they evict the weight panels every
                        // tile of the XCD re-reads.  Measured (profiles/r03_r_*): 57.6 -> 56.5 us per encoder launch, +0.6 %
                        // captions/s; dbg 512 = plain stores (A/B)
                        TOut* cp = C + (size_t)m * g.ldc + n;
                        if (g.dbg & 512)
                            *reinterpret_cast<u32x4_t*>(cp) = v;
                        else
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cp), "v"(v) : "memory");
                    }
                }
                if constexpr (LNF == 2) {
                    // fp16 stream rows with their row partials: a row of the slab is the 32 chunks of one HALF wave
                    static_assert(LNF != 2 || CPR == 32, "one stream row per half wave");
                    const bool ok = m < g.M;
                    const int mc = ok ? m : g.M - 1;
                    u32x4_t v = *reinterpret_cast<const u32x4_t*>(ep + row * EPS + cc * EPC);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) unpack2h(v[e], f[2 * e], f[2 * e + 1]);
                    if (g.res) {
                        float rs[8];
                        const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const f16_t*>(g.res) + (size_t)mc * g.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) unpack2h(rr[e], rs[2 * e], rs[2 * e + 1]);
                        if (g.res_part) {       // post-norm layer: the residual is LayerNorm(raw row), rebuilt here
                            const f32x4_t* pp = reinterpret_cast<const f32x4_t*>(g.res_part + (size_t)mc * 4);
                            const f32x4_t p0 = pp[0], p1 = pp[1];
                            const float mean = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * g.res_inv_d;
                            const float rstd = rsqrtf(fmaxf(((p0[1] + p0[3]) + (p1[1] + p1[3])) * g.res_inv_d - mean * mean, 0.f) + g.res_eps);
#pragma unroll
                            for (int e = 0; e < 8; ++e) rs[e] = fmaf((rs[e] - mean) * rstd, rgm[e], rbt[e]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += rs[e];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack2h(f[2 * e], f[2 * e + 1]);
                    }
                    if (g.part_out) {           // statistics of the values AS STORED (what the consumer's MFMA will read)
                        typedef __attribute__((ext_vector_type(2))) float f32x2_t;      // pairs: v_pk_add_f32 / v_pk_fma_f32
                        f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x0, x1;
                            unpack2h(v[e], x0, x1);
                            const f32x2_t x = {x0, x1};
                            s2 += x;
                            q2 = __builtin_elementwise_fma(x, x, q2);
                        }
                        float sx = s2[0] + s2[1], sq = q2[0] + q2[1];
                        half_wave_sum2_hi(sx, sq);
                        if (ok && cc == 16 && !(DBG & 1)) g.part_out[(size_t)m * 4 + tile_n] = float2{sx, sq};
                    }
                    if (ok && !(DBG & 1)) {
                        TOut* cp = C + (size_t)m * g.ldc + n;
                        // sc1 as for every 16-bit output; plain stores (the rows are the next GEMM's A operand) measured the same:
                        // 53.3 vs 53.6 us for the consumer (profiles/r06_j_*)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cp), "v"(v) : "memory");
                    }
                }
            }
            P8_LDS_BARRIER();
        }
    if constexpr ((DBG & 16) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stamp includes the store acknowledgements
        stamp(3);
        stamps_out();
    }
}

}  // namespace

template <typename TOut, int MH, int EPI = 1, int MH1 = MH, int LNF = 0>
static void launch_p8_t(const GemmArgs& g, hipStream_t s) {
    if constexpr (LNF == 2) {       // stream rows: no activation (and no instantiation of the activation variants)
        hipLaunchKernelGGL((gemm_p8_kernel<TOut, GITMI_ACT_NONE, 0, MH, EPI, MH1, LNF>), dim3(g.nwg), dim3(512), 0, s, g);
        return;
    } else {
    if (g.dbg && LNF == 0) {      // measurement builds (tools/gemm_dbg.py); act is ignored
        if constexpr (MH == 128 && MH1 == 128) {
            switch (g.dbg) {
                case 1: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 1>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 2: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 2>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 6: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 6>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 10: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 10>), dim3(g.nwg), dim3(512), 0, s, g); return;
#ifdef GITMI_PROBE      // tools/gemm_probe.hip: the same variants with time stamps
                case 16: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 16>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 17: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 17>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 18: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 18>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 22: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 22>), dim3(g.nwg), dim3(512), 0, s, g); return;
                case 26: hipLaunchKernelGGL((gemm_p8_kernel<TOut, 0, 26>), dim3(g.nwg), dim3(512), 0, s, g); return;
#endif
                default: break;
            }
        }
    }
    switch (g.act) {
        case GITMI_ACT_QUICKGELU:
            hipLaunchKernelGGL((gemm_p8_kernel<TOut, GITMI_ACT_QUICKGELU, 0, MH, EPI, MH1, LNF>), dim3(g.nwg), dim3(512), 0, s, g); break;
        case GITMI_ACT_GELU_ERF:
            hipLaunchKernelGGL((gemm_p8_kernel<TOut, GITMI_ACT_GELU_ERF, 0, MH, EPI, MH1, LNF>), dim3(g.nwg), dim3(512), 0, s, g); break;
        default:
            hipLaunchKernelGGL((gemm_p8_kernel<TOut, GITMI_ACT_NONE, 0, MH, EPI, MH1, LNF>), dim3(g.nwg), dim3(512), 0, s, g); break;
    }
    }
}

bool gemm_p8_supports(const GemmArgs& g) {
    return g.K % BK == 0 && g.K >= 2 * BK && g.N % BN == 0 &&
           (double)g.M * g.lda * 2.0 < 4.0e9 && (double)g.N * g.K * 2.0 < 4.0e9;
}

// Partition of the tile grid over the 8 XCDs for tile height bm: the N tiles are cut into ng groups, the 8/ng XCDs
// of a group share its (M-major, N-fastest) tile list in equal chunks.  Returns the rounds an XCD's 32 CUs need (one
// workgroup per CU) for the best ng: fewest rounds first (33 tiles on one XCD cost a whole extra round), then the
// fewest distinct operand panels among the 32 tiles an XCD runs concurrently -- ceil(32/nn) activation panels + nn
// weight panels for a group nn tiles wide -- i.e. the best L2 sharing (8192^3: ng = 8 -> 12 panels, 1.49 PFLOP/s;
// ng = 1 -> 33 panels, 1.05 PFLOP/s).
static int p8_xcd_count(const GemmArgs& g, int bm, int ng, int* max_nn_out) {
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = g.N / BN, mg = 8 / ng;
    int max_cnt = 0, max_nn = 0;
    for (int x = 0; x < 8; ++x) {
        const int gn = x % ng, gm = x / ng;
        const int nn = (gn + 1) * tiles_n / ng - gn * tiles_n / ng;
        const int tg = tiles_m * nn;
        const int cnt = (gm + 1) * tg / mg - gm * tg / mg;
        max_cnt = cnt > max_cnt ? cnt : max_cnt;
        max_nn = nn > max_nn ? nn : max_nn;
    }
    if (max_nn_out) *max_nn_out = max_nn;
    return max_cnt;
}
static int p8_plan(const GemmArgs& g, int bm, int* ng_out, int* max_cnt_out) {
    const int tiles_n = g.N / BN;
    int best_ng = 1, best_rounds = 1 << 30, best_cnt = 0, best_panels = 1 << 30;
    for (int ng = 1; ng <= 8 && ng <= tiles_n; ng *= 2) {
        int max_nn = 0;
        const int max_cnt = p8_xcd_count(g, bm, ng, &max_nn);
        const int rounds = (max_cnt + 31) / 32;
        const int panels = (32 + max_nn - 1) / max_nn + (max_nn < 32 ? max_nn : 32);
        if (rounds < best_rounds || (rounds == best_rounds && panels < best_panels)) {
            best_ng = ng; best_rounds = rounds; best_cnt = max_cnt; best_panels = panels;
        }
    }
    if (ng_out) *ng_out = best_ng;
    if (max_cnt_out) *max_cnt_out = best_cnt;
    return best_rounds;
}

// Time model of a launch with bm-row tiles, in ns: rounds x (prologue + K loop + epilogue of one tile).  Constants from the
// workgroups' own time stamps and from the height A/B (profiles/r05_b_gemm_probe_clock_and_sections.txt, r05_c_gemm_heights.txt):
// 1.45 us from entry to the first MFMA; 1.45 us per K tile of a 256-row tile with every CU busy, of which 35 % do not shrink
// with the rows (the weight half tiles, barriers and waits of a K tile are the same for every height); 4.7 us of epilogue
// for 256 rows of 16-bit output.  What it decides is how a launch quantises into rounds of 256 workgroups: N = 768 at
// M = 12 608 is 150 tiles of 256 rows (59 % of the CUs, one long round: 28.2 / 69.8 us for K = 768 / 3072) or 237 tiles of
// 160 rows (93 %: 22.1 / 55.7 us); N = 3072 is 600 tiles of 256 rows (three rounds, the last 34 % full: 80.3 us) or 684 of
// 224 rows (three shorter rounds: 73.2 us).  The model ranks the heights as the A/B measured them on all twelve shapes of
// the three BASELINE models except where two heights are within 3 %.
int gemm_p8_cost(const GemmArgs& g, int bm) {
    const int rounds = p8_plan(g, bm, nullptr, nullptr);
    const int ktile = 1450 * (90 + 166 * bm / 256) / 256;          // 1450 ns x (0.35 + 0.65 bm / 256)
    return rounds * (1450 + (g.K / BK) * ktile + 4700 * bm / 256);
}

template <int MH, int MH1>
static void launch_p8_height(const GemmArgs& g, bool out_f32, bool staged, hipStream_t s) {
#ifdef GITMI_OPS_F16        // LayerNorm folding: the stream rows ARE operands only when the operand type is fp16
    if (g.ln_part) { launch_p8_t<bf16_t, MH, 1, MH1, 1>(g, s); return; }
    if (g.part_out || g.res_part) { launch_p8_t<f16_t, MH, 1, MH1, 2>(g, s); return; }
#endif
    if (g.out_f16) launch_p8_t<f16_t, MH, 1, MH1>(g, s);
    else if (!out_f32) launch_p8_t<bf16_t, MH, 1, MH1>(g, s);
    else if (staged) launch_p8_t<float, MH, 0, MH1>(g, s);
    else launch_p8_t<float, MH, 1, MH1>(g, s);
}

// g.out_f16: C (and the residual, if any) are f16_t rows -- the residual stream of the bf16 engine mode
hipError_t launch_gemm_p8(GemmArgs g, bool out_f32, hipStream_t s) {
    if (g.ln_part || g.part_out || g.res_part) {
#ifndef GITMI_OPS_F16
        return hipErrorInvalidValue;
#endif
        // consumer: 16-bit operand rows out, no residual; producer: fp16 stream rows out
        if (g.ln_part && (out_f32 || g.out_f16 || g.res || !g.ln_colsum || g.ln_nparts < 1 || g.part_out || g.res_part)) return hipErrorInvalidValue;
        if ((g.part_out || g.res_part) && (!g.out_f16 || out_f32 || g.act != 0 || g.N > 1024)) return hipErrorInvalidValue;
        if (g.res_part && (!g.res || !g.res_gamma || !g.res_beta || g.res_nparts < 1)) return hipErrorInvalidValue;
    }
    // tile height: the one with the lowest modelled launch time among 256 / 224 / 192 / 160 / 128 rows (ties: the taller tile,
    // it moves fewer operand bytes per FLOP).  dbg bits force a height (tests, A/B): 64 -> 192, 128 -> 256, 16384 -> 160,
    // 32768 -> 224, 65536 -> 128.
    // g.shared (several contexts keep the device busy: gitmi_set_shared_device): always the 256-row tile -- the CUs a
    // partial round leaves idle are filled by the other contexts' kernels, so the tile with the better FLOP/byte wins
    // (measured in the mixed schedule, profiles/r03_m_*: 10.05k -> 10.27k captions/s, while the same choice is 3 % slower
    // for a context that has the device to itself)
    static const int heights[5] = {256, 224, 192, 160, 128};
    int bm = 256;
    if (!g.shared) {
        int best = gemm_p8_cost(g, 256);
        for (int h = 1; h < 5; ++h) {
            const int c = gemm_p8_cost(g, heights[h]);
            if (c < best) { best = c; bm = heights[h]; }
        }
    }
    if (g.dbg & 64) bm = 192;
    if (g.dbg & 128) bm = 256;
    if (g.dbg & 16384) bm = 160;
    if (g.dbg & 32768) bm = 224;
    if (g.dbg & 65536) bm = 128;
    g.dbg &= ~(64 | 128 | 16384 | 32768 | 65536);
    g.tiles_n = g.N / BN;
    int max_cnt = 0;
    p8_plan(g, bm, &g.ng, &max_cnt);
    if (g.dbg & (1024 | 2048 | 4096 | 8192)) {           // A/B (tools/gemm_bench.py): force the XCD partition ng = 1 / 2 / 4 / 8
        const int ng = (g.dbg & 1024) ? 1 : (g.dbg & 2048) ? 2 : (g.dbg & 4096) ? 4 : 8;
        g.dbg &= ~(1024 | 2048 | 4096 | 8192);
        if (ng <= g.tiles_n) { g.ng = ng; max_cnt = p8_xcd_count(g, bm, ng, nullptr); }
    }
    g.nwg = 8 * max_cnt;
    const bool staged = (g.dbg & 256) != 0;              // A/B: fp32 outputs through the LDS-staged epilogue
    g.dbg &= ~256;
    switch (bm) {
        case 224: launch_p8_height<128, 96>(g, out_f32, staged, s); break;
        case 192: launch_p8_height<96, 96>(g, out_f32, staged, s); break;
        case 160: launch_p8_height<96, 64>(g, out_f32, staged, s); break;
        case 128: launch_p8_height<64, 64>(g, out_f32, staged, s); break;
        default: launch_p8_height<128, 128>(g, out_f32, staged, s); break;
    }
    return hipGetLastError();
}

}  // namespace gitmi
