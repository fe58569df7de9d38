// Decode-step GEMM family (bf16, gfx950): the 5-launches-per-layer chain of the KV-cached decode step.
//
// A decode step multiplies R = B*beams rows (64..256) by every decoder matrix once.  Each launch is a short
// dependent stage (one HBM round trip + a few hundred MFMAs), so what matters is the NUMBER of stages and that no
// stage does redundant row work.  Round 1 ran 7 launches per layer (QKV, attention, out-proj split-K, sum+LayerNorm,
// FFN1, FFN2 split-K, sum+LayerNorm); the two LayerNorm launches are gone here:
//
//   * the N = 768 GEMMs (BertSelfOutput / BertOutput dense, modeling_bert.py:171-178, 243-250) write the PRE-LayerNorm
//     sum x = A W^T + bias + residual (fp32 + a bf16 copy) and, per 16-column strip, the row partials (sum x, sum x^2)
//     of their own columns -- no split-K slabs, no LayerNorm launch;
//   * the next GEMM (QKV / FFN1 / vocabulary head) consumes bf16(x) directly as its MFMA operand with the LayerNorm
//     FOLDED into weights and epilogue:
//         LN(x) W^T + b  =  rstd * ( x (W . gamma)^T  -  mean * colsum(W . gamma) )  +  ( beta W^T + b )
//     (W' = bf16(W . gamma), colsum over the bf16-rounded W', the constant in fp32 -- all prepared at weight
//     finalisation).  mean / rstd of a row come from the 48 strip partials (fixed summation order);
//   * the residual of a post-norm layer is the NORMALISED hidden state: the N = 768 epilogue rebuilds it on the fly
//     from the previous raw x and its strip partials, so the normalised tensor is never materialised either.
//
// Kernel shape (all variants): a workgroup owns 16*MT rows x one 16-column strip, the K range is split over its NW
// waves, MFMA fragments come straight from global memory (weights are read once, activations from L2; no LDS staging,
// all loads of a chunk in flight before the first MFMA), partial accumulators are exchanged through LDS and summed
// in a FIXED order -- deterministic, batch-invariant (a row's result does not depend on which rows share its tile).
//
// Operand layout: BOTH operands are fragment-major (gitmi_common.h frag_offset): every fragment load of a wave is one
// contiguous 1-KiB read.  Weights are repacked once at finalisation; activations are WRITTEN in that order by their
// producers (the N = 768 epilogue, the FFN1 epilogue, decode attention, the embedding), so the chain never transposes.
// Row-major fragment gathers (16 rows x 64 B per instruction) cost 2.5 us of a 6.3 us QKV launch (profiles/r02_a).
//
// The vocabulary head (`vocab_topm_kernel`) keeps its 64-row activation fragments in registers and sweeps 128
// columns per workgroup (weights double-buffered), applies the no-repeat rule (decoder.py:330), and keeps a running
// per-row top-M and log-sum-exp in registers: logits never reach HBM (decoder.py:1054, 1169-1175 fused); what is
// written is one sorted candidate list + (max, sum-exp) per (row, workgroup), merged by the search kernel.
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>
#include <utility>

namespace gitmi {

// ---- row statistics from strip partials --------------------------------------------------------------
// stats[strip][row] = (sum, sum of squares) over the strip's 16 columns.  The 4 lanes that share a row (lane>>4 = 0..3)
// each take strips lg, lg+4, ... and combine with two shuffles; every lane ends with (mean, rstd) of its row.
constexpr int STRIP_SLOTS = 16;          // per lane: up to 64 strips = 1024 columns

struct RowStatLoads {
    float2 v[STRIP_SLOTS];
};
__device__ __forceinline__ void stats_issue(RowStatLoads& L, const float2* __restrict__ stats, int strips, int M,
                                            int row, int lg) {
#pragma unroll
    for (int u = 0; u < STRIP_SLOTS; ++u) {
        const int sidx = lg + 4 * u;
        L.v[u] = sidx < strips ? stats[(size_t)sidx * M + row] : float2{0.f, 0.f};
    }
}
// branch-free form (clamped strip index, zero-selected): no control flow around the loads
__device__ __forceinline__ void stats_issue_nb(RowStatLoads& L, const float2* __restrict__ stats, int strips, int M,
                                               int row, int lg) {
#pragma unroll
    for (int u = 0; u < STRIP_SLOTS; ++u) {
        const int sidx = lg + 4 * u;
        const float2 v = stats[(size_t)min(sidx, strips - 1) * M + row];
        L.v[u] = sidx < strips ? v : float2{0.f, 0.f};
    }
}
__device__ __forceinline__ void stats_finish(const RowStatLoads& L, float inv_d, float eps, float& mean, float& rstd) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < STRIP_SLOTS; ++u) { s += L.v[u].x; q += L.v[u].y; }
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    mean = s * inv_d;
    const float var = fmaxf(q * inv_d - mean * mean, 0.f);
    rstd = rsqrtf(var + eps);
}

enum { DEPI_BF16 = 0, DEPI_RES = 1 };

// grid = (ceil(N/16), ceil(M/(16*MT))); block = 64*NW
template <int MT, int NW, int EPI>
__global__ __launch_bounds__(64 * NW) void dgemm_kernel(DGemmArgs g) {
    static_assert(NW >= MT, "wave i finishes row tile i: a workgroup needs at least MT waves");
    __shared__ __attribute__((aligned(16))) f32x4_t red[NW][MT][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);

    const bf16_t* __restrict__ X = g.A;
    const bf16_t* __restrict__ W = g.W;

    // the tile this wave finishes (if any) and the row/columns this lane finishes in it
    const bool finisher = wave < MT;
    const int fi = wave;                                  // m-tile finished by this wave
    const int fm_raw = m0 + fi * 16 + l15;
    const int fm = fm_raw < g.M ? fm_raw : g.M - 1;
    const int fn = n0 + lg * 4;

    // ---- early, latency-hiding loads of everything the epilogue needs ------------------------------------
    RowStatLoads sl;
    bool have_stats = false;
    float4 ep_a = {0.f, 0.f, 0.f, 0.f}, ep_b = ep_a, ep_c = ep_a, ep_d = ep_a;
    auto ld4 = [&](const float* p, int n) {
        float4 r;
        r.x = p[n < g.N ? n : g.N - 1];
        r.y = p[n + 1 < g.N ? n + 1 : g.N - 1];
        r.z = p[n + 2 < g.N ? n + 2 : g.N - 1];
        r.w = p[n + 3 < g.N ? n + 3 : g.N - 1];
        return r;
    };
    if (finisher && !(g.dbg & 8)) {
        ep_a = ld4(g.bias, fn);
        if constexpr (EPI == DEPI_BF16) {
            if (g.stats_in) {
                stats_issue(sl, g.stats_in, g.strips_in, g.M, fm, lg);
                have_stats = true;
                ep_b = ld4(g.colsum, fn);
            }
        } else {
            // residual source (always 16-byte aligned: N % 16 == 0 for this epilogue)
            ep_b = *reinterpret_cast<const float4*>(g.res_x + (size_t)fm * g.N + fn);
            if (g.res_stats) {
                stats_issue(sl, g.res_stats, g.res_strips, g.M, fm, lg);
                have_stats = true;
                ep_c = *reinterpret_cast<const float4*>(g.res_gamma + fn);
                ep_d = *reinterpret_cast<const float4*>(g.res_beta + fn);
            }
        }
    }

    // ---- K range of this wave, operand pointers ----------------------------------------------------------
    const int ksteps = g.K >> 5;
    const int per = (ksteps + NW - 1) / NW;
    const int kb = wave * per;
    const int ke = min(kb + per, ksteps);

    // fragment-major operands: tile (row tile, k-step) is 512 elements, this lane's 8 at lane*8
    const bf16_t* wp = W + frag_tile(blockIdx.x, 0, ksteps, lane);
    const bf16_t* xp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xp[i] = X + frag_tile(blockIdx.y * MT + i, 0, ksteps, lane);
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](auto UC, int k0) {
        constexpr int U = decltype(UC)::value;
        bf16x8_t wf[U], xf[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g.dbg & 16) wf[u] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)(k0 + u) * 512);      // A/B: cacheable weight loads
            else if (!(g.dbg & 2)) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)(k0 + u) * 512));
            else wf[u] = bf16x8_t{1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (!(g.dbg & 1)) xf[u][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)(k0 + u) * 512);
                else xf[u][i] = bf16x8_t{1, 1, 1, 1, 1, 1, 1, 1};
            }
        }
        if (g.dbg & 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                asm volatile("" ::"v"(wf[u]));
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(xf[u][i]));
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i] = mfma16(wf[u], xf[u][i], acc[i]);
    };
    constexpr int UBIG = MT >= 4 ? 6 : 12;       // <= 30 sixteen-byte loads in flight per lane
    int k = kb;
    for (; k + UBIG <= ke; k += UBIG) chunk(std::integral_constant<int, UBIG>{}, k);
    for (; k + 2 <= ke; k += 2) chunk(std::integral_constant<int, 2>{}, k);
    for (; k < ke; ++k) chunk(std::integral_constant<int, 1>{}, k);

    // ---- cross-wave reduction: every wave publishes all its tiles; wave i sums tile i over the writers 0..NW-1 --
#pragma unroll
    for (int i = 0; i < MT; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (!finisher) return;
    f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const f32x4_t t = red[w][fi][lane];
        tot[0] += t[0]; tot[1] += t[1]; tot[2] += t[2]; tot[3] += t[3];
    }
    float v[4] = {tot[0], tot[1], tot[2], tot[3]};
    const float biasv[4] = {ep_a.x, ep_a.y, ep_a.z, ep_a.w};

    if constexpr (EPI == DEPI_BF16) {
        if (have_stats) {
            float mean, rstd;
            stats_finish(sl, g.inv_d, g.eps_in, mean, rstd);
            const float cs[4] = {ep_b.x, ep_b.y, ep_b.z, ep_b.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * cs[r]) + biasv[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += biasv[r];
        }
        if (g.act != GITMI_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
        }
        if (fm_raw >= g.M || fn >= g.N) return;
        if (g.c_frag) {      // operand of the next chain GEMM (N % 32 == 0): 4 consecutive columns = 8 contiguous bytes
            uint2 t;
            t.x = pack2bf(v[0], v[1]);
            t.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + frag_offset(fm, fn, g.N >> 5)) = t;
            return;
        }
        bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (size_t)fm * g.ldc + fn;
        if (fn + 3 < g.N && (g.ldc & 3) == 0) {
            uint2 t;
            t.x = pack2bf(v[0], v[1]);
            t.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(cp) = t;
        } else {
            for (int r = 0; r < 4; ++r)
                if (fn + r < g.N) cp[r] = f2bf(v[r]);
        }
    } else {
        // x = A W^T + bias + residual, residual = LayerNorm_prev(x_prev) rebuilt from its strip partials
        float res[4] = {ep_b.x, ep_b.y, ep_b.z, ep_b.w};
        if (have_stats) {
            float mean, rstd;
            stats_finish(sl, g.res_inv_d, g.res_eps, mean, rstd);
            const float gm[4] = {ep_c.x, ep_c.y, ep_c.z, ep_c.w}, bt[4] = {ep_d.x, ep_d.y, ep_d.z, ep_d.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) res[r] = (res[r] - mean) * rstd * gm[r] + bt[r];
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] += biasv[r] + res[r];
            s += v[r];
            q += v[r] * v[r];
        }
        s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
        if (fm_raw >= g.M) return;
        *reinterpret_cast<f32x4_t*>(g.x_out + (size_t)fm * g.N + fn) = f32x4_t{v[0], v[1], v[2], v[3]};
        uint2 t;
        t.x = pack2bf(v[0], v[1]);
        t.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(g.xb_out + frag_offset(fm, fn, g.N >> 5)) = t;      // fragment-major operand copy
        if (lg == 0) g.stats_out[(size_t)blockIdx.x * g.M + fm] = float2{s, q};
    }
}

// ---- wide form over MORE than 64 rows (beam batches, decode groups): the workgroup walks the row blocks ------------
// grid = (ceil(N/16), 1); block = 256.  Same arithmetic per row as dgemm_kernel<4, 4, DEPI_BF16> (K split over 4 waves,
// partials summed in wave order), so results are bit-identical; the wave's slice of the weight strip (K/4 <= 6 k-steps)
// is loaded ONCE and stays in registers while the workgroup walks the 64-row blocks -- with one workgroup per (strip,
// row block) a 256-row beam batch streamed every weight four times (9.1 us against 5.7 us for 64 rows).
constexpr int DW_KS = 6;        // k-steps per wave held in registers: K <= 4 * 6 * 32 = 768

__global__ __launch_bounds__(256) void dgemm_wide_rows_kernel(DGemmArgs g) {
    __shared__ __attribute__((aligned(16))) f32x4_t red[2][4][4][64];      // double-buffered: one barrier per row block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16, fn = n0 + lg * 4;
    const bf16_t* __restrict__ X = g.A;
    const int ksteps = g.K >> 5;
    const int per = (ksteps + 3) / 4;
    const int kb = wave * per, ke = min(kb + per, ksteps);

    auto ld4 = [&](const float* p, int n) {
        float4 r;
        r.x = p[n < g.N ? n : g.N - 1];
        r.y = p[n + 1 < g.N ? n + 1 : g.N - 1];
        r.z = p[n + 2 < g.N ? n + 2 : g.N - 1];
        r.w = p[n + 3 < g.N ? n + 3 : g.N - 1];
        return r;
    };
    // per-column constants and this wave's weight fragments: once per workgroup
    const float4 bias4 = ld4(g.bias, fn);
    const float4 cs4 = g.stats_in ? ld4(g.colsum, fn) : float4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wp = g.W + frag_tile(blockIdx.x, 0, ksteps, lane);
    bf16x8_t wf[DW_KS];
#pragma unroll
    for (int u = 0; u < DW_KS; ++u)
        wf[u] = kb + u < ke ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)(kb + u) * 512))
                            : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};

    const int nrb = (g.M + 63) >> 6;
    for (int rb = 0; rb < nrb; ++rb) {
        const int m0 = rb * 64;
        const int fm_raw = m0 + wave * 16 + l15;                 // wave i finishes row tile i of the block
        const int fm = fm_raw < g.M ? fm_raw : g.M - 1;
        RowStatLoads sl;
        if (g.stats_in) stats_issue(sl, g.stats_in, g.strips_in, g.M, fm, lg);
        bf16x8_t xf[DW_KS][4];
#pragma unroll
        for (int u = 0; u < DW_KS; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xf[u][i] = kb + u < ke ? *reinterpret_cast<const bf16x8_t*>(X + frag_tile(rb * 4 + i, kb + u, ksteps, lane))
                                       : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        f32x4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < DW_KS; ++u) {
            if (kb + u < ke) {                                   // uniform per wave; keeps the MFMA sequence of dgemm_kernel
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16(wf[u], xf[u][i], acc[i]);
            }
        }
        f32x4_t(*rd)[4][64] = red[rb & 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) rd[wave][i][lane] = acc[i];
        __syncthreads();
        f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4_t t = rd[w][wave][lane];
            tot[0] += t[0]; tot[1] += t[1]; tot[2] += t[2]; tot[3] += t[3];
        }
        float v[4] = {tot[0], tot[1], tot[2], tot[3]};
        const float biasv[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
        if (g.stats_in) {
            float mean, rstd;
            stats_finish(sl, g.inv_d, g.eps_in, mean, rstd);
            const float cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * cs[r]) + biasv[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += biasv[r];
        }
        if (g.act != GITMI_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
        }
        if (fm_raw >= g.M || fn >= g.N) continue;
        if (g.c_frag) {
            uint2 t;
            t.x = pack2bf(v[0], v[1]);
            t.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + frag_offset(fm, fn, g.N >> 5)) = t;
            continue;
        }
        bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (size_t)fm * g.ldc + fn;
        if (fn + 3 < g.N && (g.ldc & 3) == 0) {
            uint2 t;
            t.x = pack2bf(v[0], v[1]);
            t.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(cp) = t;
        } else {
            for (int r = 0; r < 4; ++r)
                if (fn + r < g.N) cp[r] = f2bf(v[r]);
        }
    }
}

// ---- wide form, <= 64 rows, NST adjacent 16-column strips per workgroup ----------------------------------------------
// grid = (ceil(strips / NST), 1); block = 256.  Per element the arithmetic of dgemm_kernel<4, 4, DEPI_BF16> (K split over 4
// waves, partials summed in wave order): bit-identical results.  The four waves load the activation fragments of their K
// slice ONCE and the weight fragments of all NST strips up front (one memory round trip, as before), then run the strips
// back to back: 1/NST of the workgroups / resident waves and of the activation traffic for ~1 us more per extra strip --
// what a decode launch costs the image encoders running beside it is its resident waves x time.
template <int NST>
__global__ __launch_bounds__(256) void dgemm_wide_strips_kernel(DGemmArgs g) {
    __shared__ __attribute__((aligned(16))) f32x4_t red[NST][4][4][64];      // [strip][wave][row tile][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nstrips = (g.N + 15) >> 4;
    const bf16_t* __restrict__ X = g.A;
    const int ksteps = g.K >> 5;
    const int per = (ksteps + 3) / 4;
    const int kb = wave * per, ke = min(kb + per, ksteps);
    const int fm_raw = wave * 16 + l15;                       // wave i finishes row tile i
    const int fm = fm_raw < g.M ? fm_raw : g.M - 1;

    auto ld4 = [&](const float* p, int n) {
        float4 r;
        r.x = p[n < g.N ? n : g.N - 1];
        r.y = p[n + 1 < g.N ? n + 1 : g.N - 1];
        r.z = p[n + 2 < g.N ? n + 2 : g.N - 1];
        r.w = p[n + 3 < g.N ? n + 3 : g.N - 1];
        return r;
    };
    RowStatLoads sl;
    if (g.stats_in) stats_issue(sl, g.stats_in, g.strips_in, g.M, fm, lg);
    float4 bias4[NST], cs4[NST];
    bf16x8_t wf[NST][DW_KS];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int strip = min((int)blockIdx.x * NST + t, nstrips - 1);
        const int fn = strip * 16 + lg * 4;
        bias4[t] = ld4(g.bias, fn);
        cs4[t] = g.stats_in ? ld4(g.colsum, fn) : float4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* wp = g.W + frag_tile(strip, 0, ksteps, lane);
#pragma unroll
        for (int u = 0; u < DW_KS; ++u)
            wf[t][u] = kb + u < ke ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)(kb + u) * 512))
                                   : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
    bf16x8_t xf[DW_KS][4];
#pragma unroll
    for (int u = 0; u < DW_KS; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xf[u][i] = kb + u < ke ? *reinterpret_cast<const bf16x8_t*>(X + frag_tile(i, kb + u, ksteps, lane))
                                   : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        f32x4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < DW_KS; ++u) {
            if (kb + u < ke) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16(wf[t][u], xf[u][i], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[t][wave][i][lane] = acc[i];
    }
    __syncthreads();
    float mean = 0.f, rstd = 1.f;
    if (g.stats_in) stats_finish(sl, g.inv_d, g.eps_in, mean, rstd);
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int strip = (int)blockIdx.x * NST + t;
        if (strip >= nstrips) break;
        const int fn = strip * 16 + lg * 4;
        f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4_t p = red[t][w][wave][lane];
            tot[0] += p[0]; tot[1] += p[1]; tot[2] += p[2]; tot[3] += p[3];
        }
        float v[4] = {tot[0], tot[1], tot[2], tot[3]};
        const float biasv[4] = {bias4[t].x, bias4[t].y, bias4[t].z, bias4[t].w};
        if (g.stats_in) {
            const float cs[4] = {cs4[t].x, cs4[t].y, cs4[t].z, cs4[t].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * cs[r]) + biasv[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += biasv[r];
        }
        if (g.act != GITMI_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
        }
        if (fm_raw >= g.M || fn >= g.N) continue;
        if (g.c_frag) {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + frag_offset(fm, fn, g.N >> 5)) = o;
            continue;
        }
        bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (size_t)fm * g.ldc + fn;
        if (fn + 3 < g.N && (g.ldc & 3) == 0) {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(cp) = o;
        } else {
            for (int r = 0; r < 4; ++r)
                if (fn + r < g.N) cp[r] = f2bf(v[r]);
        }
    }
}

// ---- vocabulary head + running top-M / log-sum-exp ---------------------------------------------------------
// grid = min(column blocks, max_wgs); block = 256 (K split over 4 waves, K <= 768).  A column block = NS 16-column strips.
// A workgroup WALKS its column blocks (block b, b + gridDim.x, ...) with every weight fragment of a block in registers
// (NS*VKS 16-byte loads per lane = up to 96 KiB per workgroup): as soon as the last row block has consumed strip st of the
// current column block, the registers of that strip are refilled with strip st of the NEXT block, so a workgroup always has
// a whole block of weights in flight and never waits a full memory latency after its first block.  Why walk: with one
// workgroup per column block (round 3: 239 workgroups, one HBM round trip, 21 us) the launch is bound by the chip's HBM
// rate while EVERY CU that holds one of its 256-register waves is closed to the image encoder's GEMM workgroups for the
// whole 21 us; ~60 walking workgroups stream the same 47 MB at the same rate and leave the other CUs alone.  Results do
// not depend on the grid: a (row, column block) pair produces the same candidate list whichever workgroup computes it.
constexpr int VKS = 6;      // k-steps of 32 per wave held in registers
constexpr int VOC_NS = 8;   // 16-column strips per column block
constexpr int VOC_MAX_BLOCKS = 8;      // column blocks a workgroup may walk (bias / colsum of all of them sit in LDS)

// Orders this workgroup's LDS traffic only.  __syncthreads() also fences global memory: with the refill loads of the next
// column block in flight that would be an s_waitcnt vmcnt(0) per strip -- one memory latency per strip instead of none.
#define VOC_LDS_BARRIER()                                  \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)

// MT = 4 row tiles (64 rows, the padding of every activation buffer) per row block and NS = 8 strips (128 columns) per
// column block are fixed.  MODE selects what the column-block loop may contain:
//   VOC_GREEDY  one row block (<= 64 rows), K = 768, no logits output, no repetition penalty: activations and row
//               statistics stay in registers, the candidate lists are parked in LDS until the walk is over, and the loop
//               contains NOTHING but MFMAs, the LDS exchange and the rolling refill.  On gfx9 loads and stores share
//               `vmcnt` and complete out of order with respect to each other, so a single global store inside the loop
//               (or a branch around a load) turns every counted `s_waitcnt vmcnt(42)` into `vmcnt(0)` -- one exposed
//               memory latency per strip.
//   VOC_BEAM    several row blocks, K = 768: the rows' activations are re-read (from the L2) per (column block, row
//               block), which drains the queue once per row block anyway; the refill runs behind the last row block.
//   VOC_GENERIC any K <= 768, logits output, repetition penalty: conditional loads, stores in the loop.
enum { VOC_GENERIC = 0, VOC_BEAM = 1, VOC_GREEDY = 2 };

// x into a descending list tv (ties: the older entry stays ahead; -inf fill), in PLACE and without a dependency chain: every
// position decides for itself from the eight comparisons -- position j takes its left neighbour if x goes in somewhere
// before it, x if x goes in exactly here, and keeps its entry otherwise.  The same list the bubble insertion
// (`if (x > tv[M-1]) { tv[M-1] = x; swap upwards while larger }`) leaves, bit for bit, in 5 M independent instructions
// instead of a 7-deep chain of dependent compare-and-swaps behind a branch that some lane of the wave always takes: with one
// wave per SIMD (256 registers) nothing hides that chain's latency -- 24 of the beam-search head's 65 us
// (profiles/r05_l_vocab_head_beam_decomposition.txt).
template <int M>
__device__ __forceinline__ void topm_insert(float (&tv)[M], int (&ti)[M], float x, int idx) {
    bool c[M];
#pragma unroll
    for (int j = 0; j < M; ++j) c[j] = x > tv[j];
#pragma unroll
    for (int j = M - 1; j >= 1; --j) {
        tv[j] = c[j - 1] ? tv[j - 1] : (c[j] ? x : tv[j]);
        ti[j] = c[j - 1] ? ti[j - 1] : (c[j] ? idx : ti[j]);
    }
    tv[0] = c[0] ? x : tv[0];
    ti[0] = c[0] ? idx : ti[0];
}

// f(integral_constant<int, I>) for I = B .. N-1, fully unrolled with I a compile-time constant in the body
template <int B, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < N) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, N>(f);
    }
}

// ABL (tools/probe/vocab_probe.hip only; 0 in every library): 1 no top-M insertion, 2 no log-sum-exp update, 4 no merge rounds
// of the four lanes of a row, 8 the row block's activations are not re-read, 16 no MFMAs -- timing decomposition, wrong results
template <int MTOP, int MODE, int ABL = 0>
__global__ __launch_bounds__(256) void vocab_topm_kernel(VocabArgs g) {
    constexpr int MT = 4, NS = VOC_NS;
    constexpr bool FULLK = MODE != VOC_GENERIC, ONE_RB = MODE == VOC_GREEDY, PARK = MODE == VOC_GREEDY;
    // ONE LDS object (a second __shared__ array makes hipcc drain the vector-memory queue before LDS reads): the exchange
    // buffers (double-buffered: one barrier per strip), bias / column sum of every column this workgroup walks, and
    // (VOC_GREEDY) the parked candidate lists [walked block][row][2 + 2*MTOP]
    constexpr int RED_F4 = 2 * 4 * MT * 64;
    constexpr int COLC_F = VOC_MAX_BLOCKS * 16 * NS * 2;
    constexpr int OUT_W = 2 + 2 * MTOP;
    constexpr int PARK_F = PARK ? VOC_MAX_BLOCKS * 64 * OUT_W : 0;
    __shared__ __attribute__((aligned(16))) f32x4_t smem[RED_F4 + (COLC_F + PARK_F) / 4];
    f32x4_t(*red)[4][MT][64] = reinterpret_cast<f32x4_t(*)[4][MT][64]>(smem);
    float* colc = reinterpret_cast<float*>(smem + RED_F4);   // [walked block][0: bias, 1: colsum][16*NS]
    float* park = colc + COLC_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool fold = g.stats_in != nullptr;
    const int nblk = g.nblk;                                 // column blocks of 16*NS columns (launcher)
    // bias and colsum are padded to a multiple of 16*NS entries; a workgroup walks <= VOC_MAX_BLOCKS blocks (launcher)
    for (int i = tid; i < VOC_MAX_BLOCKS * 16 * NS; i += 256) {
        const int w = i / (16 * NS), c = i % (16 * NS);
        const int blk = (int)blockIdx.x + w * (int)gridDim.x;
        if (blk < nblk) {
            colc[(w * 2 + 0) * (16 * NS) + c] = g.bias[blk * (16 * NS) + c];
            colc[(w * 2 + 1) * (16 * NS) + c] = fold ? g.colsum[blk * (16 * NS) + c] : 0.f;
        }
    }

    const int ksteps = g.K >> 5;
    const int per = (ksteps + 3) / 4;
    const int kb = wave * per;
    const int ks = max(0, min(kb + per, ksteps) - kb);        // <= VKS (checked by the launcher); == VKS when FULLK

    // weight fragments of this wave's K range, one column block.  The packed matrix is zero-padded to a multiple of 128
    // rows (fold_layernorm / gitmi_op_vocab_topm): every strip of every column block is readable.
    bf16x8_t wf[NS][VKS];
    const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto fill_strip = [&](auto stc, int blk) {                // strip st of column block blk < nblk
        constexpr int st = decltype(stc)::value;
        const bf16_t* wp = g.W + frag_tile(blk * NS + st, kb, ksteps, lane);
#pragma unroll
        for (int u = 0; u < VKS; ++u) {
            if (FULLK || u < ks) wf[st][u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(wp + (size_t)u * 512));
            else wf[st][u] = zero8;
        }
    };
    auto for_strips = [&](auto&& f) { static_for<0, NS>(f); };
    for_strips([&](auto stc) { fill_strip(stc, (int)blockIdx.x); });

    const int nrb = ONE_RB ? 1 : (g.M + 16 * MT - 1) / (16 * MT);
    bf16x8_t xf[VKS][MT];
    float mean = 0.f, rstd = 1.f;
    int last_tok = -1;
    auto load_rows = [&](int rb) {            // everything that depends on the row block only (wave i finishes row tile i)
        const int fm_raw = rb * (16 * MT) + wave * 16 + l15;
        const int fm = fm_raw < g.M ? fm_raw : g.M - 1;
        RowStatLoads sl;
        last_tok = -1;
        if (fold) stats_issue_nb(sl, g.stats_in, g.strips_in, g.M, fm, lg);
        if (g.ids) {
            const int sent = fm / g.beams;
            const bool suppress = g.suppress_kind && g.cur_len > g.plen[sent];      // decoder.py:330 (not on a sentence's first step)
            if (suppress) last_tok = g.ids[(size_t)fm * g.ld_ids + g.cur_len - 1];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bf16_t* xp = g.A + frag_tile(rb * MT + i, kb, ksteps, lane);
#pragma unroll
            for (int u = 0; u < VKS; ++u) {
                if (FULLK || u < ks) xf[u][i] = *reinterpret_cast<const bf16x8_t*>(xp + (size_t)u * 512);
                else xf[u][i] = zero8;
            }
        }
        mean = 0.f; rstd = 1.f;
        if (fold) stats_finish(sl, g.inv_d, g.eps_in, mean, rstd);
    };
    if constexpr (ONE_RB) {
        load_rows(0);
        // The compiler must KNOW these loads have landed before the loop: its wait counts at the top of the loop body are
        // computed for the merged (entry + back edge) state, and with the activation loads -- the youngest of the prologue
        // -- still pending there, the first strip of EVERY column block would wait for (almost) all 48 refill loads.
#pragma unroll
        for (int u = 0; u < VKS; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(xf[u][i]));
    }

    const bool pen = MODE == VOC_GENERIC && g.ids != nullptr && g.rep_penalty != 0.f && g.rep_penalty != 1.f;
    int xbuf = 0;      // exchange buffer of the next strip: alternates across row and column blocks (an odd strip count -- the
                       // vocabulary tail -- must not reuse buffer 0 for the last strip of one pass and the first of the next)
    VOC_LDS_BARRIER();                                       // colc is complete
    // One column block.  REFILL (every block but the workgroup's last) is a COMPILE-TIME property of the call: a branch
    // around the refill loads -- even a workgroup-uniform one -- makes hipcc's wait-count bookkeeping age the older loads
    // at every join, and the counted waits of the rolling refill collapse to vmcnt(0..5).  For the same reason every block
    // computes all NS strips: past the vocabulary the packed weights, bias and column sums are zero and the columns are
    // skipped in the candidate update, so the tail block needs no per-strip branch.
    auto column_block = [&](auto refill_c, const int blk, const int walked) {
        constexpr bool REFILL = decltype(refill_c)::value;
        const int c0 = blk * (16 * NS);
        const int nxt = blk + (int)gridDim.x;
        const float* cb = colc + walked * 2 * (16 * NS);
        for (int rb = 0; rb < nrb; ++rb) {
            const bool last_rb = rb == nrb - 1;
            const int fm_raw = rb * (16 * MT) + wave * 16 + l15;
            const int fm = fm_raw < g.M ? fm_raw : g.M - 1;
            if constexpr (!ONE_RB) { if (!(ABL & 8) || rb == 0) load_rows(rb); }
            // repetition penalty: which of this lane's columns (bit st*4 + r <-> column c0 + st*16 + lg*4 + r) are in the row's history
            unsigned int pen_mask = 0u;
            if constexpr (MODE == VOC_GENERIC) {
                if (pen) {
                    for (int s2 = 0; s2 < g.cur_len; ++s2) {
                        const int rel = g.ids[(size_t)fm * g.ld_ids + s2] - c0;
                        if (rel >= 0 && rel < 16 * NS && ((rel >> 2) & 3) == lg) pen_mask |= 1u << ((rel >> 4) * 4 + (rel & 3));
                    }
                }
            }
            // running per-lane state: sorted top-MTOP of its 4 columns per strip, online log-sum-exp
            float tv[MTOP];
            int ti[MTOP];
#pragma unroll
            for (int j = 0; j < MTOP; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
            float mx = -INFINITY, sm = 0.f;

            for_strips([&](auto stc) {
                constexpr int st = decltype(stc)::value;
                {
                    f32x4_t acc[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < VKS; ++u)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            if constexpr (ABL & 16) { asm volatile("" ::"v"(wf[st][u]), "v"(xf[u][i])); acc[i][0] += 1.f; }
                            else acc[i] = mfma16(wf[st][u], xf[u][i], acc[i]);
                        }
                    const int buf = xbuf;
                    xbuf ^= 1;
#pragma unroll
                    for (int i = 0; i < MT; ++i) red[buf][wave][i][lane] = acc[i];
                    VOC_LDS_BARRIER();
                    f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f32x4_t t = red[buf][w][wave][lane];
                        tot[0] += t[0]; tot[1] += t[1]; tot[2] += t[2]; tot[3] += t[3];
                    }
                    const int n = c0 + st * 16 + lg * 4;
                    float v[4] = {tot[0], tot[1], tot[2], tot[3]};
                    const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(cb + st * 16 + lg * 4);
                    const f32x4_t cc = *reinterpret_cast<const f32x4_t*>(cb + 16 * NS + st * 16 + lg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (fold) v[r] = rstd * (v[r] - mean * cc[r]) + bb[r];
                        else v[r] += bb[r];
                    }
                    if constexpr (MODE == VOC_GENERIC) {
                        if (g.logits_out && fm_raw < g.M) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n + r < g.N) g.logits_out[(size_t)fm * g.ld_logits + n + r] = v[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = v[r];
                        const int idx = n + r;
                        if (idx >= g.N) continue;
                        if constexpr (MODE == VOC_GENERIC) {
                            if ((pen_mask >> (st * 4 + r)) & 1u) x = rep_penalize(x, g.rep_penalty);
                        }
                        if (idx == last_tok) x = -10000.f;
                        if constexpr (ABL & 2) { asm volatile("" ::"v"(x)); }
                        else {
                            // online log-sum-exp, both arms of `x > mx` evaluated and selected (the arithmetic of
                            //   if (x > mx) { sm = sm * exp(mx - x) + 1; mx = x; } else sm += exp(x - mx);
                            // with ONE exponential and no divergent branch)
                            const bool up = x > mx;
                            const float e = fast_exp(up ? mx - x : x - mx);
                            sm = up ? sm * e + 1.f : sm + e;
                            mx = up ? x : mx;
                        }
                        if constexpr (ABL & 1) { asm volatile("" ::"v"(x)); tv[0] = fmaxf(tv[0], x); }
                        else if constexpr (MTOP == 1) {
                            if (x > tv[0]) { tv[0] = x; ti[0] = idx; }
                        } else topm_insert<MTOP>(tv, ti, x, idx);
                    }
                }
                // strip st of this column block is done for the last row block: its registers take the next block's strip
                if constexpr (REFILL) {
                    if (ONE_RB || last_rb) fill_strip(stc, nxt);
                }
            });
            // ---- merge the 4 lanes of a row (equal lane & 15): log-sum-exp, then MTOP rounds of 4-way arg-max ----------
            float bm = fmaxf(mx, __shfl_xor(mx, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            float part = mx == -INFINITY ? 0.f : sm * fast_exp(mx - bm);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            const size_t slot = (size_t)fm * nblk + blk;
            const bool writer = lg == 0 && fm_raw < g.M;
            float* pk = park + ((size_t)walked * 64 + wave * 16 + l15) * OUT_W;      // VOC_GREEDY: this lane's parked list
            if (writer) {
                if constexpr (PARK) { pk[0] = bm; pk[1] = part; }
                else g.part_lse[slot] = float2{bm, part};
            }
#pragma unroll
            for (int round = 0; round < ((ABL & 4) ? 1 : MTOP); ++round) {
                float v = tv[0];
                int id = ti[0];
                int who = lg;
#pragma unroll
                for (int o = 16; o < 64; o <<= 1) {
                    const float ov = __shfl_xor(v, o, 64);
                    const int oi = __shfl_xor(id, o, 64);
                    const int ow = __shfl_xor(who, o, 64);
                    if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; who = ow; }
                }
                if (writer) {
                    if constexpr (PARK) { pk[2 + round] = v; pk[2 + MTOP + round] = __int_as_float(id); }
                    else {
                        g.part_val[slot * MTOP + round] = v;
                        g.part_idx[slot * MTOP + round] = id;
                    }
                }
                if (who == lg) {                                   // pop the winner's head (static shifts: no dynamic register index)
#pragma unroll
                    for (int j = 0; j + 1 < MTOP; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
                    tv[MTOP - 1] = -INFINITY; ti[MTOP - 1] = 0x7fffffff;
                }
            }
        }
    };
    const int nwalk = (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;      // blockIdx.x < nblk (launcher)
    for (int w = 0; w + 1 < nwalk; ++w) column_block(std::true_type{}, (int)blockIdx.x + w * (int)gridDim.x, w);
    column_block(std::false_type{}, (int)blockIdx.x + (nwalk - 1) * (int)gridDim.x, nwalk - 1);
    const int walked = nwalk;
    if constexpr (PARK) {
        // the walk is over, nothing is in flight: every writer lane stores the lists it parked (it reads its own LDS words)
        const int fm = wave * 16 + l15;
        if (lg == 0 && fm < g.M) {
            for (int w = 0; w < walked; ++w) {
                const float* pk = park + ((size_t)w * 64 + fm) * OUT_W;
                const size_t slot = (size_t)fm * nblk + (blockIdx.x + (size_t)w * gridDim.x);
                g.part_lse[slot] = float2{pk[0], pk[1]};
#pragma unroll
                for (int round = 0; round < MTOP; ++round) {
                    g.part_val[slot * MTOP + round] = pk[2 + round];
                    g.part_idx[slot * MTOP + round] = __float_as_int(pk[2 + MTOP + round]);
                }
            }
        }
    }
}

// ---- host launchers ------------------------------------------------------------------
template <int EPI>
static hipError_t launch_dgemm_t(const DGemmArgs& g, hipStream_t s) {
    // rows per workgroup: all of a <=64-row batch share one pass over the weight strip when the strip count alone
    // fills the chip (N >= 2304); the N = 768 GEMMs use 16-row blocks so that 48 strips x R/16 workgroups do
    const bool wide = g.N >= 1536;
    const int big_k = g.K >= 2048;
    if (EPI == DEPI_RES || !wide) {
        // g.rows_per_wg: 16 (default: 48 strips x R/16 workgroups, each re-reading its weight strip from the L2) / 32 / 64
        // (one pass over the strip for 64 rows: a quarter of the workgroups and 40 % less L2 traffic, a longer launch)
        const int mt = g.rows_per_wg >= 64 && g.M > 32 ? 4 : g.rows_per_wg >= 32 && g.M > 16 ? 2 : 1;
        dim3 grid((g.N + 15) / 16, (g.M + 16 * mt - 1) / (16 * mt));
        if (mt == 4) {
            if (big_k) hipLaunchKernelGGL((dgemm_kernel<4, 8, EPI>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((dgemm_kernel<4, 4, EPI>), grid, dim3(256), 0, s, g);
        } else if (mt == 2) {
            if (big_k) hipLaunchKernelGGL((dgemm_kernel<2, 8, EPI>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((dgemm_kernel<2, 4, EPI>), grid, dim3(256), 0, s, g);
        } else if (big_k) hipLaunchKernelGGL((dgemm_kernel<1, 8, EPI>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((dgemm_kernel<1, 4, EPI>), grid, dim3(256), 0, s, g);
    } else if (g.M <= 16) {
        hipLaunchKernelGGL((dgemm_kernel<1, 4, EPI>), dim3((g.N + 15) / 16, 1), dim3(256), 0, s, g);
    } else if (g.M <= 32) {
        hipLaunchKernelGGL((dgemm_kernel<2, 4, EPI>), dim3((g.N + 15) / 16, 1), dim3(256), 0, s, g);
    } else if (EPI == DEPI_BF16 && g.M <= 64 && (g.K >> 5) <= 4 * DW_KS && g.strips_per_wg >= 2) {
        const int nst = g.strips_per_wg >= 6 ? 6 : g.strips_per_wg >= 4 ? 4 : 2;                        // strips per workgroup
        const dim3 grid(((g.N + 15) / 16 + nst - 1) / nst, 1);
        if (nst == 6) hipLaunchKernelGGL(dgemm_wide_strips_kernel<6>, grid, dim3(256), 0, s, g);
        else if (nst == 4) hipLaunchKernelGGL(dgemm_wide_strips_kernel<4>, grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL(dgemm_wide_strips_kernel<2>, grid, dim3(256), 0, s, g);
    } else if (EPI == DEPI_BF16 && g.M > 64 && (g.K >> 5) <= 4 * DW_KS && !g.dbg && !g.no_row_walk) {
        hipLaunchKernelGGL(dgemm_wide_rows_kernel, dim3((g.N + 15) / 16, 1), dim3(256), 0, s, g);       // weights once for all row blocks
    } else {
        hipLaunchKernelGGL((dgemm_kernel<4, 4, EPI>), dim3((g.N + 15) / 16, (g.M + 63) / 64), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

hipError_t launch_dgemm(const DGemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K % 32 != 0 || !g.bias) return hipErrorInvalidValue;
    if (g.stats_in && (!g.colsum || g.strips_in > 4 * STRIP_SLOTS)) return hipErrorInvalidValue;
    if (g.x_out) {
        if ((g.N & 15) || !g.xb_out || !g.stats_out || !g.res_x) return hipErrorInvalidValue;
        if (g.res_stats && (g.res_strips > 4 * STRIP_SLOTS || !g.res_gamma || !g.res_beta)) return hipErrorInvalidValue;
        return launch_dgemm_t<DEPI_RES>(g, s);
    }
    if (!g.C) return hipErrorInvalidValue;
    return launch_dgemm_t<DEPI_BF16>(g, s);
}

int vocab_parts(int V, int cols_per_wg) { return (V + cols_per_wg - 1) / cols_per_wg; }
int vocab_mtop_slots(int mtop) { return mtop <= 1 ? 1 : mtop <= 2 ? 2 : mtop <= 4 ? 4 : mtop <= 8 ? 8 : 16; }

template <int MTOP>
static hipError_t launch_vocab_m(const VocabArgs& g_in, hipStream_t s) {
    VocabArgs g = g_in;
    g.nblk = vocab_parts(g.N, 16 * VOC_NS);
    int nwg = g.max_wgs > 0 && g.max_wgs < g.nblk ? g.max_wgs : g.nblk;              // each workgroup walks nblk / nwg column blocks
    if ((g.nblk + nwg - 1) / nwg > VOC_MAX_BLOCKS) nwg = (g.nblk + VOC_MAX_BLOCKS - 1) / VOC_MAX_BLOCKS;
    const bool pen = g.ids != nullptr && g.rep_penalty != 0.f && g.rep_penalty != 1.f;
    const bool fast = (g.K >> 5) == 4 * VKS && !g.logits_out && !pen;
    if (!fast) hipLaunchKernelGGL((vocab_topm_kernel<MTOP, VOC_GENERIC>), dim3(nwg), dim3(256), 0, s, g);
    else if (g.M <= 64) hipLaunchKernelGGL((vocab_topm_kernel<MTOP, VOC_GREEDY>), dim3(nwg), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((vocab_topm_kernel<MTOP, VOC_BEAM>), dim3(nwg), dim3(256), 0, s, g);     // walks the row blocks too
    return hipGetLastError();
}

// 128 columns per column block (cols_per_wg, kept in the argument list as a check); bias / colsum / the packed weight rows
// must be readable up to the next multiple of 128; activations up to the next multiple of 64 rows
hipError_t launch_vocab_topm(const VocabArgs& g, int mtop, hipStream_t s) {
    if (g.M <= 0) return hipSuccess;
    if (g.K % 32 != 0 || (g.K >> 5) > 4 * VKS || g.cols_per_wg != 16 * VOC_NS || mtop < 1 || mtop > 16)
        return hipErrorInvalidValue;
    if (g.stats_in && (!g.colsum || g.strips_in > 4 * STRIP_SLOTS)) return hipErrorInvalidValue;
    if (mtop <= 1) return launch_vocab_m<1>(g, s);
    if (mtop <= 2) return launch_vocab_m<2>(g, s);
    if (mtop <= 4) return launch_vocab_m<4>(g, s);
    if (mtop <= 8) return launch_vocab_m<8>(g, s);
    return launch_vocab_m<16>(g, s);
}

}  // namespace gitmi
