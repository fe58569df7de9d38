// EXPERIMENT (measurement build only: the whole file compiles to nothing without -DGITMI_EXPERIMENT).
//
// The question VERDICT r03 item 3(iii) / DESIGN "Next" left open: would a FOUR-wave workgroup with 128x128 wave tiles on
// v_mfma_f32_32x32x16_bf16 beat the eight-wave 256x256x64 kernel of kernels_gemm10.hip ("p8")?  On paper it reads two
// thirds of p8's LDS bytes per FLOP (a wave reads 128 + 128 operand rows per K tile for a 128x128 tile; p8's wave reads
// 128 + 64 for 128x64), issues half the MFMA instructions, and needs ONE workgroup barrier per K tile instead of eight.
//
//   C[M,N] (bf16) = act(A[M,K] * W[N,K]^T + bias[N])        K % 64 == 0, K >= 128, N % 256 == 0     (no residual: timing kernel)
//
//   * 256 threads = 4 waves, one per SIMD, wave (wr, wc) owns rows wr*128.. and columns wc*128.. of the 256x256 tile as
//     4 x 4 MFMA tiles of 32x32 (swapped orientation: operand A = 32 weight rows, operand B = 32 activation rows, so a
//     lane's accumulator registers hold 4 CONSECUTIVE n of one m).  256 accumulator registers per lane (AGPRs).
//   * LDS: the same two K-tile buffers of [A0 | A1 | B0 | B1] half tiles in the same bank-conflict-free image as p8, filled by
//     the same LDS-DMA pieces (a wave issues 4 of the 16 one-KiB pieces of every half tile).  A 32-row fragment read
//     (lane -> row lane & 31, 16-byte chunk 2*ks + (lane >> 5)) hits 16 distinct bank slots per ds_read_b128 lane group in
//     that image too.
//   * ONE wave per SIMD cannot hide a ds_read behind another wave's MFMAs, so the K loop is software-pipelined inside the
//     wave: the operands of k-step s+1 are requested before the 16 MFMAs of k-step s are issued, and the K-tile hand-over
//     (counted vmcnt for the wave's own pieces of tile t+1, the workgroup barrier, the LDS-DMA issue of tile t+2 into the
//     buffer just drained, the first operand reads of tile t+1) sits in front of the LAST k-step's MFMAs of tile t.
//   * Epilogue straight from the accumulators (8-byte bf16 stores); enough for a timing comparison on the bf16-output
//     shapes (QKV, c_fc).
#include "gitmi_common.h"
#include "launchers.h"

#ifdef GITMI_EXPERIMENT
namespace gitmi {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void q4_lds_void_t;

constexpr int Q4_BK = 64;
constexpr int Q4_HALF = 128 * Q4_BK * 2;              // 16 KiB: 128 rows x 128 B
constexpr int Q4_BUF = 4 * Q4_HALF;                   // A0 A1 B0 B1
constexpr int Q4_LDS = 2 * Q4_BUF;                    // 128 KiB

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
#ifdef GITMI_OPS_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

template <int ACT>
__global__ __launch_bounds__(256) void gemm_q4_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[Q4_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // tile of this workgroup: the XCD partition of p8 (kernels_gemm10.hip)
    int tile_m, tile_n;
    {
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int ng = g.ng, mg = 8 / ng;
        const int gn = x % ng, gm = x / ng;
        const int tiles_m = (g.M + 255) / 256;
        const int n_lo = gn * g.tiles_n / ng, n_hi = (gn + 1) * g.tiles_n / ng;
        const int nn = n_hi - n_lo;
        const int tg = tiles_m * nn;
        const int lo_t = gm * tg / mg, hi_t = (gm + 1) * tg / mg;
        if (idx >= hi_t - lo_t) return;
        const int L = lo_t + idx;
        tile_m = L / nn;
        tile_n = n_lo + L % nn;
    }
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const char* __restrict__ Ab = reinterpret_cast<const char*>(g.A);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(g.W);

    // staging: a wave instruction fills 1 KiB = 4 bank rows = 8 tile rows; wave w owns pieces P = 4*w + q (q = 0..3) of every
    // half tile.  lane -> bank row Rl, half hi, slot lo (the image of kernels_gemm10.hip)
    const int Rl = lane >> 4, hi = (lane >> 3) & 1, lo = lane & 7;
    uint32_t a_off[2][4], w_off[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int P = wave * 4 + q;
            const int rr = h * 128 + P * 8 + 2 * Rl + (hi ^ (P & 1));
            const int ch = lo ^ ((P & 1) * 4 + Rl);
            int m = m0 + rr;
            m = m < g.M ? m : g.M - 1;
            a_off[h][q] = ((uint32_t)m * (uint32_t)g.lda + ch * 8) * 2u;
            w_off[h][q] = ((uint32_t)(n0 + rr) * (uint32_t)g.K + ch * 8) * 2u;
        }
    auto issue_tile = [&](int kt) {                    // all four half tiles of K tile kt -> buffer kt & 1 (16 LDS-DMA pieces)
        unsigned char* buf = smem + (kt & 1) * Q4_BUF;
#pragma unroll
        for (int isw = 0; isw < 2; ++isw)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const char* src = (isw ? Wb : Ab) + (size_t)kt * (Q4_BK * 2);
                unsigned char* dst = buf + (isw * 2 + h) * Q4_HALF + wave * 4096;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_global_load_lds((const void*)(src + (isw ? w_off[h][q] : a_off[h][q])),
                                                     (q4_lds_void_t*)(dst + q * 1024), 16, 0, 0);
            }
    };

    // fragment addressing: row r of a half tile, 16-byte chunk c -> (r>>1)*256 + ((r&1)^((r>>3)&1))*128 + (c ^ ((r>>1)&7))*16
    const int l31 = lane & 31, kh = lane >> 5;
    int row_off[4];                                    // the lane's row of fragment i (rows i*32 + l31 of the half tile)
    int row_x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 32 + l31;
        row_off[i] = (r >> 1) * 256 + ((r & 1) ^ ((r >> 3) & 1)) * 128;
        row_x[i] = (r >> 1) & 7;
    }
    const int a_base = wr * Q4_HALF, w_base = (2 + wc) * Q4_HALF;

    f32x16_t acc[4][4];                                // [j: n fragment][i: m fragment]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    bf16x8_t af[2][4], wf[2][4];                       // operand double buffer [parity of the k-step][fragment]
    auto read_ops = [&](const unsigned char* sb, int ks, int par) {
        const int c = 2 * ks + kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[par][i] = *reinterpret_cast<const bf16x8_t*>(sb + a_base + row_off[i] + ((c ^ row_x[i]) << 4));
            wf[par][i] = *reinterpret_cast<const bf16x8_t*>(sb + w_base + row_off[i] + ((c ^ row_x[i]) << 4));
        }
    };
    auto mma = [&](int par) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma32(wf[par][j], af[par][i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
    };
#define Q4_FENCE() __builtin_amdgcn_sched_barrier(0)

    const int nk = g.K / Q4_BK;                        // >= 2 (launcher)
    issue_tile(0);
    issue_tile(1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // this wave's 16 pieces of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_ops(smem, 0, 0);
    for (int t = 0; t < nk; ++t) {
        const unsigned char* sb = smem + (t & 1) * Q4_BUF;
        // k-steps 0..2: operands of the next k-step travel while this one's 16 MFMAs issue
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            Q4_FENCE();
            read_ops(sb, ks + 1, (ks + 1) & 1);
            Q4_FENCE();
            mma(ks & 1);
        }
        // k-step 3 operands (parity 1) are in registers once lgkmcnt drains; then the hand-over to tile t + 1
        Q4_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // every read of buffer t & 1 has returned
        if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // own pieces of tile t + 1 (issued one tile ago)
            __builtin_amdgcn_s_barrier();                                  // tile t + 1 complete; buffer t & 1 drained by all
            asm volatile("" ::: "memory");
            if (t + 2 < nk) issue_tile(t + 2);                             // into buffer t & 1
            read_ops(smem + ((t + 1) & 1) * Q4_BUF, 0, 0);
        }
        Q4_FENCE();
        mma(1);
    }
#undef Q4_FENCE

    // ---- epilogue from the accumulators.  D[n][m]: lane holds m = lane & 31, n = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 128 + i * 32 + l31;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wc * 128 + j * 32 + 8 * q + 4 * kh;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act_t<ACT>(acc[j][i][4 * q + r] + (g.bias ? g.bias[n + r] : 0.f));
                uint2 o;
                o.x = pack2bf(v[0], v[1]);
                o.y = pack2bf(v[2], v[3]);
                *reinterpret_cast<uint2*>(C + (size_t)m * g.ldc + n) = o;
            }
        }
    }
}

}  // namespace

// same partition rule as p8 (256-row tiles); bf16 output, no residual
hipError_t launch_gemm_q4(GemmArgs g, hipStream_t s) {
    if (g.res || g.out_f16 || g.K % 64 || g.K < 128 || g.N % 256) return hipErrorInvalidValue;
    const int tiles_m = (g.M + 255) / 256;
    g.tiles_n = g.N / 256;
    int best_ng = 1, best_rounds = 1 << 30, best_cnt = 0, best_panels = 1 << 30;
    for (int ng = 1; ng <= 8 && ng <= g.tiles_n; ng *= 2) {
        const int mg = 8 / ng;
        int max_cnt = 0, max_nn = 0;
        for (int x = 0; x < 8; ++x) {
            const int gn = x % ng, gm = x / ng;
            const int nn = (gn + 1) * g.tiles_n / ng - gn * g.tiles_n / ng;
            const int tg = tiles_m * nn;
            const int cnt = (gm + 1) * tg / mg - gm * tg / mg;
            max_cnt = cnt > max_cnt ? cnt : max_cnt;
            max_nn = nn > max_nn ? nn : max_nn;
        }
        const int rounds = (max_cnt + 31) / 32;
        const int panels = (32 + max_nn - 1) / max_nn + (max_nn < 32 ? max_nn : 32);
        if (rounds < best_rounds || (rounds == best_rounds && panels < best_panels)) {
            best_ng = ng; best_rounds = rounds; best_cnt = max_cnt; best_panels = panels;
        }
    }
    g.ng = best_ng;
    const dim3 grid(8 * best_cnt);
    switch (g.act) {
        case GITMI_ACT_QUICKGELU: hipLaunchKernelGGL((gemm_q4_kernel<GITMI_ACT_QUICKGELU>), grid, dim3(256), 0, s, g); break;
        case GITMI_ACT_GELU_ERF: hipLaunchKernelGGL((gemm_q4_kernel<GITMI_ACT_GELU_ERF>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_q4_kernel<GITMI_ACT_NONE>), grid, dim3(256), 0, s, g); break;
    }
    return hipGetLastError();
}

}  // namespace gitmi
#endif  // GITMI_EXPERIMENT
