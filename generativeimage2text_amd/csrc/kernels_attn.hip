// Attention kernels of the GIT engine (head_dim is 64 everywhere: ViT-B/16, ViT-L/14, decoder).
//
//  attn_full_valu   : unmasked multi-head attention, fp32 math on the vector ALUs, any dtype.
//                     Reference / exact-precision path and the checker for the MFMA kernel.
//  attn_full_mfma   : same contract, bf16 MFMA flash kernel (swapped QK^T so that softmax
//                     statistics are lane-local, online softmax over 64-key tiles).
//  attn_decode      : one new text position per row against [shared image K/V | per-beam text K/V]
//                     with beam indirection (no KV copies when beams are re-ordered).
//
// Replaces nn.MultiheadAttention in CLIP/model.py:189-197 and BertSelfAttention/qk2attn in
// modeling_bert.py:41-47,122-159.  The additive block mask of decoder.py:111-149
// (image->image 0, image->text -inf, text->image 0, text->text causal) is never materialised:
// image rows run the unmasked kernel over image keys only, text rows see all image keys plus
// text keys <= their own position.
#include "gitmi_common.h"
#include "launchers.h"
#include <type_traits>

namespace gitmi {

constexpr int HD = 64;

// ---------------------------------------------------------------------------------------
// VALU kernel: block = 4 waves, each wave owns 4 query rows; K/V tiles of 64 keys in LDS (fp32).
// grid = (ceil(N/16), H, B)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_full_valu_kernel(AttnFullArgs a) {
    __shared__ float Ks[64][HD + 1];
    __shared__ float Vs[64][HD + 1];
    __shared__ float qs[16][HD];
    __shared__ float ps[16][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 16;
    const T* Q = reinterpret_cast<const T*>(a.q);
    const T* K = reinterpret_cast<const T*>(a.k);
    const T* V = reinterpret_cast<const T*>(a.v);
    T* O = reinterpret_cast<T*>(a.out);
    const size_t base_row = (size_t)b * a.N;

    for (int i = tid; i < 16 * HD; i += 256) {
        const int qi = i / HD, d = i % HD;
        const int qr = q0 + qi;
        qs[qi][d] = qr < a.N ? ld<T>(Q + (base_row + qr) * a.ldq + h * HD + d) * a.scale : 0.f;
    }

    float m_run[4], l_run[4], o_acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m_run[j] = -INFINITY; l_run[j] = 0.f; o_acc[j] = 0.f; }

    for (int kt = 0; kt < a.N; kt += 64) {
        __syncthreads();   // previous tile fully consumed (also orders the qs fill on the first pass)
        for (int i = tid; i < 64 * HD; i += 256) {
            const int key = i / HD, d = i % HD;
            const int kr = kt + key;
            float kv = 0.f, vv = 0.f;
            if (kr < a.N) {
                kv = ld<T>(K + (base_row + kr) * a.ldk + h * HD + d);
                vv = ld<T>(V + (base_row + kr) * a.ldv + h * HD + d);
            }
            Ks[key][d] = kv;
            Vs[key][d] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qi = wave * 4 + j;
            // lane = key
            float s = 0.f;
#pragma unroll 16
            for (int d = 0; d < HD; ++d) s += qs[qi][d] * Ks[lane][d];
            if (kt + lane >= a.N) s = -INFINITY;
            const float m_new = fmaxf(m_run[j], wave_max(s));
            const float p = __expf(s - m_new);             // exp(-inf) = 0 for masked keys
            const float alpha = __expf(m_run[j] - m_new);  // first tile: exp(-inf) = 0
            l_run[j] = l_run[j] * alpha + wave_sum(p);
            m_run[j] = m_new;
            ps[qi][lane] = p;
            // wave-private row of ps: written and read by this wave only, LDS ops are in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // lane = dim
            float o = o_acc[j] * alpha;
#pragma unroll 16
            for (int key = 0; key < 64; ++key) o += ps[qi][key] * Vs[key][lane];
            o_acc[j] = o;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qr = q0 + wave * 4 + j;
        if (qr < a.N) st<T>(O + (base_row + qr) * a.ldo + h * HD + lane, o_acc[j] / l_run[j]);
    }
}

// ---------------------------------------------------------------------------------------
// MFMA flash kernel (bf16).  Block = 4 waves, up to 256 query rows (4 tiles of 16 per wave), key tiles of 64.
//   S^T[key][query] = K_tile . Q^T      (MFMA A = K rows from LDS, B = Q rows from registers)
//   O^T[dim][query] += V^T_tile . P^T   (MFMA A = V^T rows from LDS, B = P^T from registers)
// In both products the lane's column is its query (lane & 15), so row max / row sum / the
// rescale factor never leave the lane group {l, l^16, l^32, l^48}.
// The 32-deep contraction of the second product enumerates keys in the order the first
// product's accumulators already hold them: slot j<4 -> key 4g+j, slot j>=4 -> key 16+4g+(j-4)
// (g = lane>>4) inside each 32-key half tile; V^T is read with the same permutation.
// grid = (ceil(N/256), H, B)
// ---------------------------------------------------------------------------------------
constexpr int FA_LDK = HD + 8;    // K tile row stride (bf16): 144 B, conflict-free b128 reads
constexpr int FA_LDV = 64 + 8;    // V^T tile row stride (keys)

__global__ __launch_bounds__(256) void attn_full_mfma_kernel(AttnFullArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * FA_LDK];   // [key][dim]
    __shared__ __attribute__((aligned(16))) bf16_t Vt[HD * FA_LDV];   // [dim][key]

    // One workgroup = up to 256 queries of one (image, head): wave w owns the 16-query tiles w, w+4, w+8, w+12
    // of the chunk, so each 64-key K/V tile is fetched from HBM and transposed into LDS ONCE for all of them
    // (a workgroup per 64 queries re-read K/V four times: measured 174 MB fetched for 77 MB of operands).
    constexpr int QT = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qbase = blockIdx.x * 256;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out);
    const size_t base_row = (size_t)b * a.N;

    // Q^T operands (B of the first product): lane = query l15, dims lg*8 + 32*s .. +8
    bf16x8_t qf[QT][2];
    f32x4_t o_acc[QT][4];   // O^T[dim = dt*16 + lg*4 + r][query = l15]
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qr = qbase + (t * 4 + wave) * 16 + l15;
        qr = qr < a.N ? qr : a.N - 1;
        const bf16_t* qp = Q + (base_row + qr) * a.ldq + h * HD + lg * 8;
        qf[t][0] = *reinterpret_cast<const bf16x8_t*>(qp);
        qf[t][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_acc[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
    }

    // cooperative tile load mapping: 64 rows x 8 chunks(16 B) = 512 chunks, 2 per thread
    const int ld_row = tid >> 3;          // 0..31 (+32)
    const int ld_c = (tid & 7) * 8;       // dim offset of the chunk

    for (int kt = 0; kt < a.N; kt += 64) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int key = ld_row + it * 32;
            int kr = kt + key;
            kr = kr < a.N ? kr : a.N - 1;          // clamped rows are masked below
            const u32x4_t kc = *reinterpret_cast<const u32x4_t*>(K + (base_row + kr) * a.ldk + h * HD + ld_c);
            const u32x4_t vc = *reinterpret_cast<const u32x4_t*>(V + (base_row + kr) * a.ldv + h * HD + ld_c);
            *reinterpret_cast<u32x4_t*>(Ks + key * FA_LDK + ld_c) = kc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {          // transpose V into [dim][key]
                Vt[(ld_c + 2 * e) * FA_LDV + key] = (bf16_t)(vc[e] & 0xffffu);
                Vt[(ld_c + 2 * e + 1) * FA_LDV + key] = (bf16_t)(vc[e] >> 16);
            }
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (qbase + (t * 4 + wave) * 16 >= a.N) continue;        // wave-uniform: this query tile is empty
            // ---- S^T tile: 4 key sub-tiles of 16 ---------------------------------------
            f32x4_t s_acc[4];
#pragma unroll
            for (int st_ = 0; st_ < 4; ++st_) {
                s_acc[st_] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (st_ * 16 + l15) * FA_LDK + ks * 32 + lg * 8);
                    s_acc[st_] = mfma16(kf, qf[t][ks], s_acc[st_]);
                }
            }
            // lane holds keys kt + st*16 + lg*4 + r for its query
            float tmax = -INFINITY;
#pragma unroll
            for (int st_ = 0; st_ < 4; ++st_)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + st_ * 16 + lg * 4 + r;
                    const float sv = key < a.N ? s_acc[st_][r] * a.scale : -INFINITY;
                    s_acc[st_][r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run[t], tmax);
            const float alpha = fast_exp(m_run[t] - m_new);
            float psum = 0.f;
#pragma unroll
            for (int st_ = 0; st_ < 4; ++st_)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp(s_acc[st_][r] - m_new);
                    s_acc[st_][r] = p;
                    psum += p;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run[t] = l_run[t] * alpha + psum;
            m_run[t] = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o_acc[t][dt][r] *= alpha;

            // ---- O^T += V^T . P^T, two 32-key halves ------------------------------------
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                // P^T operand: slots 0-3 <- sub-tile 2*hf, slots 4-7 <- sub-tile 2*hf+1
                bf16x8_t pf;
                {
                    union { bf16x8_t v; uint32_t u[4]; } pk;
                    pk.u[0] = pack2bf(s_acc[2 * hf][0], s_acc[2 * hf][1]);
                    pk.u[1] = pack2bf(s_acc[2 * hf][2], s_acc[2 * hf][3]);
                    pk.u[2] = pack2bf(s_acc[2 * hf + 1][0], s_acc[2 * hf + 1][1]);
                    pk.u[3] = pack2bf(s_acc[2 * hf + 1][2], s_acc[2 * hf + 1][3]);
                    pf = pk.v;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    // V^T operand: row = dim dt*16 + l15, keys hf*32 + {4lg..4lg+3, 16+4lg..16+4lg+3}
                    union { bf16x8_t v; uint2 h2[2]; } vf;
                    const bf16_t* vp = Vt + (dt * 16 + l15) * FA_LDV + hf * 32 + lg * 4;
                    vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
                    vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 16);
                    o_acc[t][dt] = mfma16(vf.v, pf, o_acc[t][dt]);
                }
            }
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qr = qbase + (t * 4 + wave) * 16 + l15;
        if (qr < a.N) {
            const float inv = 1.0f / l_run[t];
            bf16_t* op = O + (base_row + qr) * a.ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 tt;
                tt.x = pack2bf(o_acc[t][dt][0] * inv, o_acc[t][dt][1] * inv);
                tt.y = pack2bf(o_acc[t][dt][2] * inv, o_acc[t][dt][3] * inv);
                *reinterpret_cast<uint2*>(op + dt * 16 + lg * 4) = tt;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Single-pass MFMA kernel for SHORT sequences (N <= 16*NSUB, NSUB = 13 for the 197 tokens of ViT-B/16 at 224 px,
// 17 for the 257 of ViT-L/14): grid = (H, B), one workgroup per (image, head).
// The whole K ([key][dim]) and V^T ([dim][key]) of the head are staged in LDS ONCE (60 / 75 KiB, one
// __syncthreads in the kernel); a wave then takes two 16-query tiles at a time through
//     S^T = K . Q^T  over ALL keys (accumulators stay in registers: 2 x NSUB x 4 floats per lane),
//     one max / exp2 / sum per lane (no running rescale: every key is present), O^T = V^T . P^T.
// Against the 64-key flash kernel above: no per-tile barriers and load round trips, no online rescaling
// (half the VALU work per score), and only ceil(N/16) key sub-tiles instead of ceil(N/64)*4.
// Same operand conventions as attn_full_mfma_kernel (swapped products, permuted key order inside a 32-key block).
// ---------------------------------------------------------------------------------------
template <int NSUB>
__global__ __launch_bounds__(256, 2) void attn_full_mfma_short_kernel(AttnFullArgs a) {
    constexpr int NK = NSUB * 16;               // padded keys of the first product
    constexpr int NB = (NSUB + 1) / 2;          // 32-key blocks of the second product
    constexpr int LDV = NB * 32 + 8;            // V^T row stride (bf16): 16 B x odd -> conflict-free b64 reads
    static_assert((LDV / 8) % 2 == 1, "V^T row stride must be an odd multiple of 16 bytes");
    __shared__ __attribute__((aligned(16))) bf16_t Ks[NK * FA_LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[HD * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.v);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out);
    const size_t base_row = (size_t)b * a.N;

    // ---- stage K: 16-byte chunks, rows >= N are copies of the last row (masked below) ------------
#pragma unroll
    for (int it = 0; it < (NK * 8 + 255) / 256; ++it) {
        int idx = tid + it * 256;
        const bool ok = idx < NK * 8;
        idx = ok ? idx : NK * 8 - 1;
        const int key = idx >> 3, c = idx & 7;
        const int kr = key < a.N ? key : a.N - 1;
        const u32x4_t kc = *reinterpret_cast<const u32x4_t*>(K + (base_row + kr) * a.ldk + h * HD + c * 8);
        if (ok) *reinterpret_cast<u32x4_t*>(Ks + key * FA_LDK + c * 8) = kc;
    }
    // ---- stage V^T: a task = 4 consecutive keys x 8 dims -> eight 8-byte stores; keys >= N are zeros ----
#pragma unroll
    for (int it = 0; it < (NB * 8 * 8 + 255) / 256; ++it) {
        int task = tid + it * 256;
        const bool ok = task < NB * 8 * 8;
        task = ok ? task : NB * 8 * 8 - 1;
        const int kg = task % (NB * 8), c = task / (NB * 8);     // key group fastest: conflict-free LDS stores
        u32x4_t vc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kg * 4 + e;
            const int kr = key < a.N ? key : a.N - 1;
            vc[e] = *reinterpret_cast<const u32x4_t*>(V + (base_row + kr) * a.ldv + h * HD + c * 8);
            if (key >= a.N) vc[e] = u32x4_t{0u, 0u, 0u, 0u};
        }
        if (ok) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {                        // dims c*8 + 2w, c*8 + 2w + 1
                uint2 even, odd;
                even.x = (vc[0][w] & 0xffffu) | (vc[1][w] << 16);
                even.y = (vc[2][w] & 0xffffu) | (vc[3][w] << 16);
                odd.x = (vc[0][w] >> 16) | (vc[1][w] & 0xffff0000u);
                odd.y = (vc[2][w] >> 16) | (vc[3][w] & 0xffff0000u);
                *reinterpret_cast<uint2*>(Vt + (c * 8 + 2 * w) * LDV + kg * 4) = even;
                *reinterpret_cast<uint2*>(Vt + (c * 8 + 2 * w + 1) * LDV + kg * 4) = odd;
            }
        }
    }
    __syncthreads();

    const int nqt = (a.N + 15) >> 4;
    const float c2 = a.scale * 1.4426950408889634f;           // exp(scale * (s - m)) = exp2(c2 * s - c2 * m)

    auto process = [&](auto nt_c, int t0) {
        constexpr int NT = decltype(nt_c)::value;
        bf16x8_t qf[NT][2];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            int qr = (t0 + 4 * u) * 16 + l15;
            qr = qr < a.N ? qr : a.N - 1;
            const bf16_t* qp = Q + (base_row + qr) * a.ldq + h * HD + lg * 8;
            qf[u][0] = *reinterpret_cast<const bf16x8_t*>(qp);
            qf[u][1] = *reinterpret_cast<const bf16x8_t*>(qp + 32);
        }
        // ---- S^T over all keys --------------------------------------------------------------
        f32x4_t sacc[NT][NSUB];
#pragma unroll
        for (int st_ = 0; st_ < NSUB; ++st_) {
            const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t*>(Ks + (st_ * 16 + l15) * FA_LDK + lg * 8);
            const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t*>(Ks + (st_ * 16 + l15) * FA_LDK + 32 + lg * 8);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                sacc[u][st_] = mfma16(kf0, qf[u][0], f32x4_t{0.f, 0.f, 0.f, 0.f});
                sacc[u][st_] = mfma16(kf1, qf[u][1], sacc[u][st_]);
            }
            // many sub-tiles (GIT_LARGE: 17): keep the scheduler from hoisting every K fragment load to the top of the
            // unrolled loop -- with 136 score registers live that spilled
            if constexpr (NSUB > 13) { if (st_ % 4 == 3) __builtin_amdgcn_sched_barrier(0); }
        }
        // ---- softmax: the lane holds keys st*16 + lg*4 + r of its query; only the last sub-tile can hold keys >= N
        float inv_l[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((NSUB - 1) * 16 + lg * 4 + r >= a.N) sacc[u][NSUB - 1][r] = -INFINITY;
            float m = -INFINITY;
#pragma unroll
            for (int st_ = 0; st_ < NSUB; ++st_)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, sacc[u][st_][r]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float mc = m * c2;
            float psum = 0.f;
#pragma unroll
            for (int st_ = 0; st_ < NSUB; ++st_)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[u][st_][r] * c2 - mc);
                    sacc[u][st_][r] = p;
                    psum += p;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            inv_l[u] = 1.0f / psum;
        }
        // ---- P packed to bf16 right away: halves the registers that stay live across the second product (the 17-sub-tile
        // instantiation of GIT_LARGE's 257 tokens spilled 36 B/lane with fp32 scores held until their block's turn)
        uint32_t pk2[NT][NSUB][2];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int st_ = 0; st_ < NSUB; ++st_) {
                pk2[u][st_][0] = pack2bf(sacc[u][st_][0], sacc[u][st_][1]);
                pk2[u][st_][1] = pack2bf(sacc[u][st_][2], sacc[u][st_][3]);
            }
        // ---- O^T = V^T . P^T ------------------------------------------------------------------
        f32x4_t o_acc[NT][4];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o_acc[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            bf16x8_t pf[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                union { bf16x8_t v; uint32_t w[4]; } pk;
                pk.w[0] = pk2[u][2 * blk][0];
                pk.w[1] = pk2[u][2 * blk][1];
                const bool has1 = 2 * blk + 1 < NSUB;              // odd NSUB: the last block has one sub-tile
                const int s1 = has1 ? 2 * blk + 1 : 0;
                pk.w[2] = has1 ? pk2[u][s1][0] : 0u;
                pk.w[3] = has1 ? pk2[u][s1][1] : 0u;
                pf[u] = pk.v;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                union { bf16x8_t v; uint2 h2[2]; } vf;
                const bf16_t* vp = Vt + (dt * 16 + l15) * LDV + blk * 32 + lg * 4;
                vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
                vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 16);
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    o_acc[u][dt] = mfma16(vf.v, pf[u], o_acc[u][dt]);
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int qr = (t0 + 4 * u) * 16 + l15;
            if (qr < a.N) {
                bf16_t* op = O + (base_row + qr) * a.ldo + h * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    uint2 tt;
                    tt.x = pack2bf(o_acc[u][dt][0] * inv_l[u], o_acc[u][dt][1] * inv_l[u]);
                    tt.y = pack2bf(o_acc[u][dt][2] * inv_l[u], o_acc[u][dt][3] * inv_l[u]);
                    *reinterpret_cast<uint2*>(op + dt * 16 + lg * 4) = tt;
                }
            }
        }
    };

    // wave w owns query tiles w, w+4, w+8, ...; two at a time
    for (int t0 = wave; t0 < nqt; t0 += 8) {
        if (t0 + 4 < nqt) process(std::integral_constant<int, 2>{}, t0);
        else process(std::integral_constant<int, 1>{}, t0);
    }
}

// Image K/V cache in HEAD-MAJOR layout: kh/vh[b][h][n][64].  The prefill GEMM writes q|k|v token-major
// ([B*N, 3d], what the prefill attention wants); a decode workgroup (image b, head h) would then touch one
// 128-byte slice per 4.6-KB token row -- a new DRAM page per access (measured 2.4 TB/s).  Repacked once per
// generate, every decode step streams two contiguous 25-KB runs per workgroup instead.
template <typename T>
__global__ void kv_repack_kernel(const T* __restrict__ qkv, T* __restrict__ kh, T* __restrict__ vh, int B, int N, int H,
                                 int d) {
    constexpr int CH = 16 / (int)sizeof(T);                // elements per 16-byte chunk
    constexpr int CPH = HD / CH;                           // chunks per head row (8 bf16 / 16 f32)
    const size_t per = (size_t)B * H * N * CPH;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 2 * per; i += (size_t)gridDim.x * blockDim.x) {
        const int kv = i >= per;
        size_t r = kv ? i - per : i;
        const int c = (int)(r % CPH); r /= CPH;
        const int n = (int)(r % N); r /= N;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        const T* src = qkv + ((size_t)b * N + n) * 3 * d + (kv ? 2 : 1) * d + h * HD + c * CH;
        T* dst = (kv ? vh : kh) + (((size_t)b * H + h) * N + n) * HD + c * CH;
        *reinterpret_cast<u32x4_t*>(dst) = *reinterpret_cast<const u32x4_t*>(src);
    }
}

// 16 bytes-or-32 of one key/value row slice kept raw in registers until it is consumed
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    u32x4_t r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const u32x4_t*>(p); }
    __device__ __forceinline__ void load_nt(const bf16_t* p) { r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<u32x4_t*>(p) = r; }
    __device__ __forceinline__ void zero() { r = u32x4_t{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unpack2op(r[i], v[2 * i], v[2 * i + 1]);
        }
    }
};
template <> struct Raw8<float> {
    f32x4_t a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4_t*>(p);
        b = *reinterpret_cast<const f32x4_t*>(p + 4);
    }
    __device__ __forceinline__ void load_nt(const float* p) {
        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
        b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + 4));
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<f32x4_t*>(p) = a;
        *reinterpret_cast<f32x4_t*>(p + 4) = b;
    }
    __device__ __forceinline__ void zero() { a = f32x4_t{0.f, 0.f, 0.f, 0.f}; b = a; }
    __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
};

// Flash-style decode attention, ONE barrier: every 8-lane group owns a subset of the keys, keeps (max, sum,
// 64-dim partial output) per beam in registers -- scores never touch LDS, there is no separate softmax pass --
// the 8 groups of a wave are merged with shuffles and the 4 waves through a 1-KB-per-beam LDS exchange.
// All global loads (q, image K/V of the first 256 keys, text K/V through the beam indirection) are issued
// before the first use; the kernel is a single memory round trip + ~1 us of arithmetic.
// PF / TI: image keys per 8-lane group per chunk / text items per group whose loads are issued up front (8 / 5: one
// memory round trip for 197 image keys and short texts, 144 VGPRs at one beam; 4 / 1 would need 76)
template <typename T, int KB, int PF = 8, int TI = 5>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeArgs a) {
    __shared__ float part[4][KB][HD + 2];      // per wave and beam: o[64], m, l

    const int tid = threadIdx.x;
    const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x;
    const int k = a.beams;                       // k <= KB
    const T* QKV = reinterpret_cast<const T*>(a.qkv);
    const T* IMGK = reinterpret_cast<const T*>(a.img_k);
    const T* IMGV = reinterpret_cast<const T*>(a.img_v);
    T* TK = reinterpret_cast<T*>(a.txt_k);
    T* TV = reinterpret_cast<T*>(a.txt_v);
    T* O = reinterpret_cast<T*>(a.out);
    const int ld3 = 3 * a.d;
    const int row0 = b * k;
    const int grp = tid >> 3, sub = tid & 7;    // 32 groups of 8 lanes; a group reads one 64-dim row at a time
    const int wave = tid >> 6;

    const int bi = a.img_of ? a.img_of[b] : b;                             // sentence -> image (several questions per image)
    const T* kbase = IMGK + ((size_t)bi * H + h) * a.N_img * HD + sub * 8;  // head-major: contiguous per (image, h)
    const T* vbase = IMGV + ((size_t)bi * H + h) * a.N_img * HD + sub * 8;
    const int nt = a.pos + 1;

    // ---- issue every load up front -------------------------------------------------------------------
    int t_j[TI], t_s[TI], t_row[TI];
#pragma unroll
    for (int u = 0; u < TI; ++u) {
        const int it = grp + 32 * u;
        t_j[u] = it < k * nt ? it / nt : -1;
        t_s[u] = it < k * nt ? it % nt : 0;
        // one beam: histories are never re-ordered, the cache row is the row itself (no dependent index load)
        if constexpr (KB == 1) t_row[u] = row0;
        else t_row[u] = (t_j[u] >= 0 && t_s[u] != a.pos) ? a.kv_src[(size_t)(row0 + t_j[u]) * a.ld_src + t_s[u]] : 0;
    }
    Raw8<T> qr[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        if (j < k) qr[j].load(QKV + (size_t)(row0 + j) * ld3 + h * HD + sub * 8);
        else qr[j].zero();
    }
    Raw8<T> kr[PF], vr[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int n = grp + 32 * u;
        if (n < a.N_img && !(a.dbg & 1)) { kr[u].load_nt(kbase + (size_t)n * HD); vr[u].load_nt(vbase + (size_t)n * HD); }   // streamed once per step
        else { kr[u].zero(); vr[u].zero(); }
    }
    Raw8<T> tk[TI], tv[TI];
#pragma unroll
    for (int u = 0; u < TI; ++u) {
        if (t_j[u] < 0) { tk[u].zero(); tv[u].zero(); }
        else if (t_s[u] == a.pos) {
            const T* src = QKV + (size_t)(row0 + t_j[u]) * ld3 + a.d + h * HD + sub * 8;
            tk[u].load(src);
            tv[u].load(src + a.d);
        } else {
            const size_t off = ((size_t)t_row[u] * a.T_max + t_s[u]) * a.d + h * HD + sub * 8;
            tk[u].load(TK + off);
            tv[u].load(TV + off);
        }
    }
    // append this position's K/V of every beam to the text cache (16-byte copies by the first k*8 threads)
    if (tid < k * 8) {
        const int j = tid >> 3;
        const T* src = QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8;
        const size_t dst = ((size_t)(row0 + j) * a.T_max + a.pos) * a.d + h * HD + sub * 8;
        Raw8<T> ck, cv;
        ck.load(src);
        cv.load(src + a.d);
        ck.store(TK + dst);
        cv.store(TV + dst);
    }

    float q[KB][8], m[KB], l[KB], o[KB][8];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        qr[j].get(q[j]);
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { q[j][e] *= a.scale; o[j][e] = 0.f; }
    }
    auto dot8 = [&](const float (&x)[8], const float (&y)[8]) {
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) p += x[e] * y[e];
        p += __shfl_xor(p, 1, 64);
        p += __shfl_xor(p, 2, 64);
        p += __shfl_xor(p, 4, 64);
        return p;                                   // all 8 lanes of the group hold the 64-dim dot product
    };

    // ---- image keys: chunks of 256 keys; two passes inside a chunk (max, then exp/accumulate) -------------
    for (int base = 0; base < a.N_img; base += 32 * PF) {
        if (base > 0) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int n = base + grp + 32 * u;
                if (n < a.N_img) { kr[u].load_nt(kbase + (size_t)n * HD); vr[u].load_nt(vbase + (size_t)n * HD); }
                else { kr[u].zero(); vr[u].zero(); }
            }
        }
        float sc[KB][PF];
        float cm[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) cm[j] = -INFINITY;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const bool valid = base + grp + 32 * u < a.N_img;
            float kv[8];
            kr[u].get(kv);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const float sv = valid ? dot8(q[j], kv) : -INFINITY;
                sc[j][u] = sv;
                cm[j] = fmaxf(cm[j], sv);
            }
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (cm[j] == -INFINITY) continue;           // this group has no key in the chunk
            const float mn = fmaxf(m[j], cm[j]);
            const float al = fast_exp(m[j] - mn);       // exp(-inf) = 0 on the first chunk
            l[j] *= al;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[j][e] *= al;
            m[j] = mn;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            float vv[8];
            vr[u].get(vv);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const float p = sc[j][u] == -INFINITY ? 0.f : fast_exp(sc[j][u] - m[j]);
                l[j] += p;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[j][e] += p * vv[e];
            }
        }
    }
    // ---- text keys of this group (beam-specific): online update -------------------------------------------
    auto text_item = [&](int jj, const float (&kv)[8], const float (&vv)[8]) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j == jj) {                              // uniform within the 8-lane group
                const float sv = dot8(q[j], kv);
                const float mn = fmaxf(m[j], sv);
                const float al = fast_exp(m[j] - mn);
                const float p = fast_exp(sv - mn);
                l[j] = l[j] * al + p;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[j][e] = o[j][e] * al + p * vv[e];
                m[j] = mn;
            }
        }
    };
#pragma unroll
    for (int u = 0; u < TI; ++u) {
        if (t_j[u] >= 0) {
            float kv[8], vv[8];
            tk[u].get(kv);
            tv[u].get(vv);
            text_item(t_j[u], kv, vv);
        }
    }
    for (int it = grp + 32 * TI; it < k * nt; it += 32) {   // long texts: dependent-load path
        const int j = it / nt, sidx = it % nt;
        float kv[8], vv[8];
        if (sidx == a.pos) {
            ld8(QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8, kv);
            ld8(QKV + (size_t)(row0 + j) * ld3 + 2 * a.d + h * HD + sub * 8, vv);
        } else {
            const int srow = KB == 1 ? row0 : a.kv_src[(size_t)(row0 + j) * a.ld_src + sidx];
            ld8(TK + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, kv);
            ld8(TV + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, vv);
        }
        text_item(j, kv, vv);
    }

    // ---- merge the 8 groups of the wave (lanes with equal `sub`), then the 4 waves through LDS ---------------
#pragma unroll
    for (int j = 0; j < KB; ++j) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m[j], off, 64);
            const float l2 = __shfl_xor(l[j], off, 64);
            const float mn = fmaxf(m[j], m2);
            const float a1 = m[j] == -INFINITY ? 0.f : fast_exp(m[j] - mn);
            const float a2 = m2 == -INFINITY ? 0.f : fast_exp(m2 - mn);
            l[j] = l[j] * a1 + l2 * a2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o2 = __shfl_xor(o[j][e], off, 64);
                o[j][e] = o[j][e] * a1 + o2 * a2;
            }
            m[j] = mn;
        }
        if ((tid & 63) < 8 && j < k) {                  // lanes 0..7 of the wave hold the wave's partial
#pragma unroll
            for (int e = 0; e < 8; ++e) part[wave][j][sub * 8 + e] = o[j][e];
            if (sub == 0) { part[wave][j][HD] = m[j]; part[wave][j][HD + 1] = l[j]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < k * HD; i += 256) {
        const int j = i / HD, dd = i % HD;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, part[w][j][HD]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = part[w][j][HD];
            const float aw = mw == -INFINITY ? 0.f : fast_exp(mw - mm);
            num += aw * part[w][j][dd];
            den += aw * part[w][j][HD + 1];
        }
        if constexpr (sizeof(T) == 2) {
            if (a.out_frag) { O[frag_offset(row0 + j, h * HD + dd, a.d >> 5)] = f2bf(num / den); continue; }
        }
        st<T>(O + (size_t)(row0 + j) * a.d + h * HD + dd, num / den);
    }
}

// ---- host launchers ------------------------------------------------------------------
hipError_t launch_attn_full(const AttnFullArgs& a, int B, bool is_f32, int impl, hipStream_t s) {
    if (B <= 0 || a.N <= 0) return hipSuccess;
    if (impl == 1) {
        if (is_f32) return hipErrorInvalidValue;
        const int nsub = (a.N + 15) / 16;
        if (nsub == 13) hipLaunchKernelGGL(attn_full_mfma_short_kernel<13>, dim3(a.H, B), dim3(256), 0, s, a);
        else if (nsub == 17) hipLaunchKernelGGL(attn_full_mfma_short_kernel<17>, dim3(a.H, B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(attn_full_mfma_kernel, dim3((a.N + 255) / 256, a.H, B), dim3(256), 0, s, a);
    } else if (impl == 2) {                        // the 64-key flash kernel, forced (A/B and tests)
        if (is_f32) return hipErrorInvalidValue;
        hipLaunchKernelGGL(attn_full_mfma_kernel, dim3((a.N + 255) / 256, a.H, B), dim3(256), 0, s, a);
    } else if (is_f32) {
        hipLaunchKernelGGL(attn_full_valu_kernel<float>, dim3((a.N + 15) / 16, a.H, B), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(attn_full_valu_kernel<bf16_t>, dim3((a.N + 15) / 16, a.H, B), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_kv_repack(const void* qkv, void* kh, void* vh, int B, int N, int H, int d, bool is_f32, hipStream_t s) {
    if (B <= 0 || N <= 0) return hipSuccess;
    const size_t total = 2 * (size_t)B * H * N * (is_f32 ? 16 : 8);
    size_t grid = (total + 255) / 256;
    grid = grid > 8192 ? 8192 : grid;
    if (is_f32) hipLaunchKernelGGL(kv_repack_kernel<float>, dim3((int)grid), dim3(256), 0, s, (const float*)qkv, (float*)kh,
                                   (float*)vh, B, N, H, d);
    else hipLaunchKernelGGL(kv_repack_kernel<bf16_t>, dim3((int)grid), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)kh,
                            (bf16_t*)vh, B, N, H, d);
    return hipGetLastError();
}

hipError_t attn_decode_configure() { return hipSuccess; }   // (no dynamic LDS any more)

size_t attn_decode_lds_bytes(int beams, int N_img, int pos) {
    (void)N_img; (void)pos;
    return sizeof(float) * 4 * (size_t)beams * (HD + 2);
}

template <typename T>
static void launch_attn_decode_t(const AttnDecodeArgs& a, int B, int H, hipStream_t s) {
    if (a.beams <= 1) hipLaunchKernelGGL((attn_decode_kernel<T, 1>), dim3(H, B), dim3(256), 0, s, a);
    else if (a.beams <= 2) hipLaunchKernelGGL((attn_decode_kernel<T, 2>), dim3(H, B), dim3(256), 0, s, a);
    else if (a.beams <= 4) hipLaunchKernelGGL((attn_decode_kernel<T, 4>), dim3(H, B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_decode_kernel<T, 8>), dim3(H, B), dim3(256), 0, s, a);
}

hipError_t launch_attn_decode(const AttnDecodeArgs& a, int B, int H, bool is_f32, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (a.beams > 8) return hipErrorInvalidValue;
    if (is_f32) launch_attn_decode_t<float>(a, B, H, s);
    else launch_attn_decode_t<bf16_t>(a, B, H, s);
    return hipGetLastError();
}

}  // namespace gitmi
