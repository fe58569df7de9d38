// Decode attention on the matrix cores (bf16): one new text position per beam row against
// [image K/V shared by all beams of a sentence | that beam's text K/V] -- BertSelfAttention for the appended row
// (modeling_bert.py:41-47, 122-159) with the image rows' K/V cached once per image.
//
// The image part is the bulk (197 keys at 224 px, 1182 for 6 video frames) and identical for every beam, so it runs as
// two tiny MFMA products per 32-key step,  S = Q K^T  (16 beam rows x 32 keys, K = 64 dims)  and  O += P V  (16 rows x
// 64 dims, K = 32 keys), with the softmax in between in registers.  The scalar formulation this replaces spent
// ~1500 VALU instructions per wave on dot products and shuffles: 6.6 of its 8.9 us per launch were arithmetic, not
// memory (profiles/r02_a_decode_kernel_ablation.txt, dbg1).
//
// Cache layouts (written once per generate by kv_repack_frag_kernel), per (image, head), keys padded to 32 with zeros:
//   K  : fragment-major [key tile of 16][dim step of 32][lane][8]: a wave's K operand of one MFMA is ONE contiguous
//        1-KiB read (lane = (dim%32)/8*16 + key%16 holds 8 consecutive dims of its key);
//   V^T: [key step of 32][dim tile of 16][lane][8], lane = lg*16 + dim%16; the 8 contraction slots of lane group lg hold
//        keys  32s + lg*4 + {0,1,2,3}  and  32s + 16 + lg*4 + {0,1,2,3}  -- exactly the 8 scores that lane already holds
//        in the accumulators of the two S tiles of the step, so P goes from the S accumulators into the P V operand
//        WITHOUT any cross-lane movement or LDS round trip.
// A workgroup serves two heads with two waves each; a wave keeps (max, sum, O) per beam row for its half of the keys.
// The text keys differ per beam (histories are re-ordered by index, kv_src) and are few (<= max_text_len): they keep the
// 8-lanes-per-key scalar path and are folded into the wave's partial before the two halves are combined.
#include "gitmi_common.h"
#include "launchers.h"
#include <algorithm>

namespace gitmi {

static constexpr int HD = 64;

// grid = (key steps, H, B); block = 256.  qkv: prefill layout [B*N, 3d] (q|k|v); one workgroup repacks the K and V of
// one (image, head, 32-key step).
__global__ __launch_bounds__(256) void kv_repack_frag_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kf,
                                                             bf16_t* __restrict__ vt, int N, int Np, int H, int d) {
    __shared__ bf16_t vs[32][HD + 8];          // the step's V tile [key][dim], padded rows
    const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int nsteps = Np >> 5;
    bf16_t* kdst = kf + ((size_t)b * H + h) * Np * HD;
    bf16_t* vdst = vt + ((size_t)b * H + h) * Np * HD;
    // 32 keys x 64 dims = 256 chunks of 8: thread -> (key, chunk)
    const int key = tid >> 3, ch = tid & 7;
    const int n = s * 32 + key;
    u32x4_t kv = {0u, 0u, 0u, 0u}, vv = kv;
    if (n < N) {
        const bf16_t* src = qkv + ((size_t)b * N + n) * 3 * d + d + h * HD + ch * 8;
        kv = *reinterpret_cast<const u32x4_t*>(src);
        vv = *reinterpret_cast<const u32x4_t*>(src + d);
    }
    // K: fragment-major, rows = keys, 2 dim steps: element (key n, dim c*8..) -> tile n/16, step c/4, lane (c%4)*16 + n%16
    *reinterpret_cast<u32x4_t*>(kdst + frag_tile(n >> 4, ch >> 2, 2, (ch & 3) * 16 + (n & 15))) = kv;
    *reinterpret_cast<u32x4_t*>(&vs[key][ch * 8]) = vv;
    __syncthreads();
    // V^T: 4 dim tiles x 64 lanes chunks of 8 keys (permuted slots, see header): thread -> (dim tile, lane)
    const int dt = tid >> 6, lane = tid & 63, lg = lane >> 4, dim = dt * 16 + (lane & 15);
    bf16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = vs[(e >> 2) * 16 + lg * 4 + (e & 3)][dim];
    u32x4_t ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (uint32_t)o[2 * e] | ((uint32_t)o[2 * e + 1] << 16);
    *reinterpret_cast<u32x4_t*>(vdst + (((size_t)s * 4 + dt) * 64 + lane) * 8) = ov;
}

__device__ __forceinline__ void ld8bf(const bf16_t* p, float (&v)[8]) {
    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unpack2op(r[i], v[2 * i], v[2 * i + 1]);
    }
}

// grid = (H, sentences); block = 128 = the 2 waves of one head (768 workgroups at B = 64: three per CU, evenly -- pairing
// two heads per workgroup left half the CUs with twice the K/V to ingest).  The two waves split the key steps (even / odd)
// and its text items; each wave requests ALL of its image K/V fragments (up to 4 steps = 32 sixteen-byte loads per
// lane) in its first instructions, computes every score tile, does ONE max / exp pass and then the P V products; the two
// halves meet once in LDS.  KB >= beams (1, 2, 4 or 8); TI: text items per 8-lane group loaded up front.
constexpr int ACS = 4;          // key steps per chunk and wave

// PW: (sentence, head) pairs per workgroup (a wave needs 214 registers, so 8 waves fill a CU).  1 spreads a launch over
// every CU (fastest on an idle device); next to the image encoder of another context every CU that holds even one of
// these waves is closed to a GEMM workgroup (8 waves x 232 registers, 128 KiB LDS) until the wave retires, so the launcher
// packs the one-wave kernel (launch_attn_decode_mfma).
// NH: waves per pair.  2 = the two waves of a head split its key steps (one memory round trip each); 1 = ONE wave walks
// all key steps in chunks of ACS (two round trips at 197 image keys): half the resident waves for ~1.3x the time.
// LOOP (packed one-wave form only): a wave serves a.pairs_per_wave pairs one after the other, the next pair's first K/V
// chunk requested while the current pair's merge and output are worked off.  A separate instantiation: the loop-carried
// K/V registers cost the 512-thread form a few spilled registers, which the single-pair form must not pay.
template <int KB, int TI = 3, int PW = 1, int NH = 2, bool LOOP = false>
__global__ __launch_bounds__(64 * NH * PW) void attn_decode_mfma_kernel(AttnDecodeArgs a) {
    // every fused multiply-add of the softmax bookkeeping is written out (fmaf): with contraction left to the compiler the
    // a*b + c*d updates fuse differently per instantiation, and results must not depend on the packing or the kernel form
#pragma clang fp contract(off)
    __shared__ float part[PW][NH][KB][HD + 2];    // [pair][half][beam]: o[64], m, l
    constexpr int PT = 64 * NH;                   // threads of a pair

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int hp = wave / NH, half = wave % NH;    // pair of the workgroup, half of the head's keys
    const int H = a.d / HD;
    const int k = a.beams;                       // k <= KB <= 8 < 16 MFMA rows
    const bf16_t* QKV = reinterpret_cast<const bf16_t*>(a.qkv);
    bf16_t* TK = reinterpret_cast<bf16_t*>(a.txt_k);
    bf16_t* TV = reinterpret_cast<bf16_t*>(a.txt_v);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out);
    const int ld3 = 3 * a.d;
    const int Np = a.N_pad, nsteps = Np >> 5;
    // a wave serves `reps` pairs one after the other (launcher: 1 unless the one-wave kernel is packed further): pair index
    // of repetition r.  PW == 1 keeps the (head, sentence) grid of the two-wave kernel.
    const int reps = LOOP ? a.pairs_per_wave : 1;
    auto pair_of = [&](int r) {
        if constexpr (NH == 2) return (int)(blockIdx.y * H + blockIdx.x);      // grid = (H, sentences)
        else return ((int)blockIdx.x * reps + r) * PW + hp;
    };
    struct PairKV { const bf16_t* Kf; const bf16_t* Vt; int nimg_steps; };
    auto kv_of = [&](int pair) {
        const bool on = PW == 1 || pair < a.n_pairs;
        const int h = on ? pair % H : 0, b = on ? pair / H : 0;
        const int bi = a.img_of ? a.img_of[b] : b;   // sentence -> image (several questions per image)
        PairKV r;
        r.Kf = reinterpret_cast<const bf16_t*>(a.img_k) + ((size_t)bi * H + h) * Np * HD;
        r.Vt = reinterpret_cast<const bf16_t*>(a.img_v) + ((size_t)bi * H + h) * Np * HD;
        r.nimg_steps = (!on || (a.dbg & 1)) ? 0 : nsteps;
        return r;
    };

    // ---- image K/V of a chunk of key steps: requested before anything else (the longest latency) ---------------
    bf16x8_t kq[ACS][2][2], vq[ACS][4];
    auto load_chunk = [&](const PairKV& p, int s0) {              // steps s0, s0 + NH, ... (this half's parity)
#pragma unroll
        for (int c = 0; c < ACS; ++c) {
            const int s = s0 + NH * c;
            const bool on = s < p.nimg_steps && !(a.dbg & 8);
            const bf16_t* kp = p.Kf + frag_tile(2 * s, 0, 2, lane);            // the step's 4 KiB of K, then of V^T: contiguous
            const bf16_t* vp = p.Vt + ((size_t)s * 4 * 64 + lane) * 8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ds = 0; ds < 2; ++ds) {
                    const bf16x8_t* q = reinterpret_cast<const bf16x8_t*>(kp + (t * 2 + ds) * 512);
                    kq[c][t][ds] = !on ? bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0} : (a.dbg & 16) ? *q : __builtin_nontemporal_load(q);
                }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t* q = reinterpret_cast<const bf16x8_t*>(vp + dt * 512);
                vq[c][dt] = !on ? bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0} : (a.dbg & 16) ? *q : __builtin_nontemporal_load(q);
            }
        }
    };
    PairKV cur = kv_of(pair_of(0));
    load_chunk(cur, half);

  for (int rep = 0; rep < reps; ++rep) {
    const int pair = pair_of(rep);
    const bool head_on = PW == 1 || pair < a.n_pairs;
    const int h = head_on ? pair % H : 0, b = head_on ? pair / H : 0;
    const int row0 = b * k;
    const int nimg_steps = cur.nimg_steps;
    if constexpr (LOOP) {
        if (rep > 0 && (a.dbg & 32)) load_chunk(cur, half);
    }

    // ---- text items: the 16 eight-lane groups of the head's two waves share them ----------------------------------
    const int grp = (tid % PT) >> 3, sub = tid & 7;      // 8 * NH eight-lane groups per pair
    const int nt = a.pos + 1;
    int t_j[TI], t_s[TI];
    u32x4_t tkr[TI], tvr[TI];
#pragma unroll
    for (int u = 0; u < TI; ++u) {
        const int it = grp + 8 * NH * u;
        t_j[u] = (head_on && it < k * nt) ? it / nt : -1;
        t_s[u] = it < k * nt ? it % nt : 0;
        tkr[u] = u32x4_t{0u, 0u, 0u, 0u};
        tvr[u] = tkr[u];
        if (t_j[u] >= 0) {
            if (t_s[u] == a.pos) {
                const bf16_t* src = QKV + (size_t)(row0 + t_j[u]) * ld3 + a.d + h * HD + sub * 8;
                tkr[u] = *reinterpret_cast<const u32x4_t*>(src);
                tvr[u] = *reinterpret_cast<const u32x4_t*>(src + a.d);
            } else {
                // one beam: histories are never re-ordered, the cache row is the row itself (no dependent index load)
                const int srow = KB == 1 ? row0 : a.kv_src[(size_t)(row0 + t_j[u]) * a.ld_src + t_s[u]];
                const size_t off = ((size_t)srow * a.T_max + t_s[u]) * a.d + h * HD + sub * 8;
                tkr[u] = *reinterpret_cast<const u32x4_t*>(TK + off);
                tvr[u] = *reinterpret_cast<const u32x4_t*>(TV + off);
            }
        }
    }
    // append this position's K/V of every beam to the text cache (16-byte copies by the first k*8 threads of the head)
    if (head_on && (tid % PT) < k * 8) {
        const int j = (tid % PT) >> 3;
        const bf16_t* src = QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8;
        const size_t dst = ((size_t)(row0 + j) * a.T_max + a.pos) * a.d + h * HD + sub * 8;
        *reinterpret_cast<u32x4_t*>(TK + dst) = *reinterpret_cast<const u32x4_t*>(src);
        *reinterpret_cast<u32x4_t*>(TV + dst) = *reinterpret_cast<const u32x4_t*>(src + a.d);
    }

    // ---- image part on the matrix cores -------------------------------------------------------------------------
    // Q operand: lane (row = l15, lg) holds dims ds*32 + lg*8 .. +8 of beam row l15, pre-scaled by 1/8 (exact in bf16)
    bf16x8_t qf[2];
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
        float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (head_on && l15 < k) ld8bf(QKV + (size_t)(row0 + l15) * ld3 + h * HD + ds * 32 + lg * 8, qv);
        u32x4_t t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = pack2bf(qv[2 * e] * a.scale, qv[2 * e + 1] * a.scale);
        qf[ds] = __builtin_bit_cast(bf16x8_t, t);
    }
    float m_i = -INFINITY, l_i = 0.f;            // running max / (per-lane partial) sum of beam row l15
    f32x4_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int s0 = half; s0 < nimg_steps; s0 += NH * ACS) {
        if (s0 != half) load_chunk(cur, s0);               // later chunks (long image sequences: video, VQA resolutions)
        // every score tile of the chunk: lane (row l15, lg) holds keys 32s + t*16 + lg*4 + r
        f32x4_t sc[ACS][2];
        float cm = -INFINITY;
#pragma unroll
        for (int c = 0; c < ACS; ++c) {
            const int s = s0 + NH * c;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sc[c][t] = mfma16(kq[c][t][0], qf[0], f32x4_t{0.f, 0.f, 0.f, 0.f});
                sc[c][t] = mfma16(kq[c][t][1], qf[1], sc[c][t]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s * 32 + t * 16 + lg * 4 + r >= a.N_img || s >= nimg_steps) sc[c][t][r] = -INFINITY;   // padded keys / steps
                    cm = fmaxf(cm, sc[c][t][r]);
                }
        }
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float mn = fmaxf(m_i, cm);                   // the chunk's first step holds >= 1 real key: mn is finite
        if (s0 != half) {
            const float al = fast_exp(m_i - mn);
            l_i *= al;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { oacc[dt][0] *= al; oacc[dt][1] *= al; oacc[dt][2] *= al; oacc[dt][3] *= al; }
        }
        m_i = mn;
#pragma unroll
        for (int c = 0; c < ACS; ++c) {
            float p[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[t * 4 + r] = fast_exp(sc[c][t][r] - mn);    // exp(-inf) = 0 for padded keys
                    l_i += p[t * 4 + r];
                }
            u32x4_t pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = pack2bf(p[2 * e], p[2 * e + 1]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pp);
            if (a.dbg & 4) {      // timing experiment: no P V product
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) asm volatile("" ::"v"(vq[c][dt]));
                asm volatile("" ::"v"(pf));
                continue;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma16(vq[c][dt], pf, oacc[dt]);
        }
    }
    l_i += __shfl_xor(l_i, 16, 64);
    l_i += __shfl_xor(l_i, 32, 64);
    // ---- text keys (beam-specific): 8 lanes per key, online update -------------------------------------------------
    float q[KB][8], m[KB], l[KB], o[KB][8];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { q[j][e] = 0.f; o[j][e] = 0.f; }
        if (head_on && j < k) {
            ld8bf(QKV + (size_t)(row0 + j) * ld3 + h * HD + sub * 8, q[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[j][e] *= a.scale;
        }
    }
    auto dot8 = [&](const float (&x)[8], const float (&y)[8]) {
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) p = fmaf(x[e], y[e], p);
        p += __shfl_xor(p, 1, 64);
        p += __shfl_xor(p, 2, 64);
        p += __shfl_xor(p, 4, 64);
        return p;                                   // all 8 lanes of the group hold the 64-dim dot product
    };
    auto text_item = [&](int jj, const float (&kv)[8], const float (&vv)[8]) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (j == jj) {                              // uniform within the 8-lane group
                const float sv = dot8(q[j], kv);
                const float mn = fmaxf(m[j], sv);
                const float al = fast_exp(m[j] - mn);
                const float p = fast_exp(sv - mn);
                l[j] = fmaf(l[j], al, p);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[j][e] = fmaf(o[j][e], al, p * vv[e]);
                m[j] = mn;
            }
        }
    };
    auto unpack = [&](const u32x4_t& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unpack2op(r[i], v[2 * i], v[2 * i + 1]);
        }
    };
#pragma unroll
    for (int u = 0; u < TI; ++u) {
        if (t_j[u] >= 0 && !(a.dbg & 2)) {
            float kv[8], vv[8];
            unpack(tkr[u], kv);
            unpack(tvr[u], vv);
            text_item(t_j[u], kv, vv);
        }
    }
    if (head_on) {
        for (int it = grp + 8 * NH * TI; it < k * nt; it += 8 * NH) {   // long texts: dependent-load path
            const int j = it / nt, sidx = it % nt;
            float kv[8], vv[8];
            if (sidx == a.pos) {
                ld8bf(QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8, kv);
                ld8bf(QKV + (size_t)(row0 + j) * ld3 + 2 * a.d + h * HD + sub * 8, vv);
            } else {
                const int srow = KB == 1 ? row0 : a.kv_src[(size_t)(row0 + j) * a.ld_src + sidx];
                ld8bf(TK + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, kv);
                ld8bf(TV + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, vv);
            }
            text_item(j, kv, vv);
        }
    }
    // the K/V registers (and the text items) are free: the first chunk of the wave's NEXT pair travels while this pair's
    // merge and output are worked off (requested any earlier -- behind the image part -- the 512-thread form spills)
    if constexpr (LOOP) {
        if (rep + 1 < reps) {
            cur = kv_of(pair_of(rep + 1));
            if (!(a.dbg & 32)) load_chunk(cur, half);      // dbg 32 (A/B): no look-ahead, the chunk is requested at the top of the pair
        }
    }
    // merge the 8 groups of the wave (lanes with equal `sub`): lanes 0..7 end up with the wave's text partial (m, l, o[8])
#pragma unroll
    for (int j = 0; j < KB; ++j) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m[j], off, 64);
            const float l2 = __shfl_xor(l[j], off, 64);
            const float mn = fmaxf(m[j], m2);
            const float a1 = m[j] == -INFINITY ? 0.f : fast_exp(m[j] - mn);
            const float a2 = m2 == -INFINITY ? 0.f : fast_exp(m2 - mn);
            l[j] = fmaf(l[j], a1, l2 * a2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o2 = __shfl_xor(o[j][e], off, 64);
                o[j][e] = fmaf(o[j][e], a1, o2 * a2);
            }
            m[j] = mn;
        }
    }
    // ---- this wave's partial per beam row = its image keys + its text items, published to LDS ----------------------
    // image part: lane (row l15 < k, lg) holds dims dt*16 + lg*4 + r; text part: lanes 0..7 hold dims sub*8 + e of row j
    if (l15 < k) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[hp][half][l15][dt * 16 + lg * 4 + r] = oacc[dt][r];
        if (lg == 0) { part[hp][half][l15][HD] = m_i; part[hp][half][l15][HD + 1] = l_i; }
    }
    // NH == 1: a wave reads back only what it wrote itself (LDS operations of a wave complete in order): no workgroup barrier
    // -- a __syncthreads() here would also wait for the next pair's K/V loads
    if constexpr (NH > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    // fold the text partial of this wave into its slot (lanes 0..7, one beam row at a time), then combine the halves
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        if (lane < 8 && j < k) {
            const float mi = part[hp][half][j][HD], li = part[hp][half][j][HD + 1];
            const float mn = fmaxf(mi, m[j]);
            const float a1 = mi == -INFINITY ? 0.f : fast_exp(mi - mn);
            const float a2 = m[j] == -INFINITY ? 0.f : fast_exp(m[j] - mn);
#pragma unroll
            for (int e = 0; e < 8; ++e) part[hp][half][j][sub * 8 + e] = fmaf(part[hp][half][j][sub * 8 + e], a1, o[j][e] * a2);
            if (sub == 0) { part[hp][half][j][HD] = mn; part[hp][half][j][HD + 1] = fmaf(li, a1, l[j] * a2); }
        }
    }
    if constexpr (NH > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (head_on)
    for (int i = tid % PT; i < k * HD; i += PT) {
        const int j = i / HD, dd = i % HD;
        if constexpr (NH == 1) {
            const float r1 = part[hp][0][j][dd] / part[hp][0][j][HD + 1];
            if (a.out_frag) O[frag_offset(row0 + j, h * HD + dd, a.d >> 5)] = f2bf(r1);
            else O[(size_t)(row0 + j) * a.d + h * HD + dd] = f2bf(r1);
            continue;
        }
        constexpr int H1 = NH - 1;
        const float m0 = part[hp][0][j][HD], m1 = part[hp][H1][j][HD];
        const float mm = fmaxf(m0, m1);
        const float a0 = m0 == -INFINITY ? 0.f : fast_exp(m0 - mm);
        const float a1 = m1 == -INFINITY ? 0.f : fast_exp(m1 - mm);
        const float num = fmaf(a0, part[hp][0][j][dd], a1 * part[hp][H1][j][dd]);
        const float den = fmaf(a0, part[hp][0][j][HD + 1], a1 * part[hp][H1][j][HD + 1]);
        const float r = num / den;
        if (a.out_frag) O[frag_offset(row0 + j, h * HD + dd, a.d >> 5)] = f2bf(r);
        else O[(size_t)(row0 + j) * a.d + h * HD + dd] = f2bf(r);
    }
    if constexpr (NH == 1) __builtin_amdgcn_wave_barrier();      // the next pair reuses this wave's LDS slot
  }
}

// ---- streaming form of the one-wave kernel -----------------------------------------------------------------------
// The kernels above keep a pair's K/V chunk in REGISTERS (128 of a wave's 215): a wave can only ask for its next chunk when
// the registers are free, so every pair costs two exposed memory round trips and a CU never has more than its resident
// waves' chunks in flight -- 24 GB/s per CU, 96 CUs for 17.8 us per launch, while every one of those CUs is closed to the
// image encoder's GEMM workgroups.  Here the K/V of a wave's pairs STREAM through an LDS ring: 4-KiB slots (the K or the
// V^T of one 32-key step -- contiguous in the cache layouts above) are fetched by LDS-DMA (global_load_lds, lane-linear:
// the fragment-major layout lands so that lane l's operand of fragment f is the 16 bytes at f * 1 KiB + l * 16) in exactly
// the order the wave consumes them,
//        pair 0: K0 K1 K2 K3 | V0 V1 V2 V3 | K4 K5 K6 | V4 V5 V6      pair 1: ...
// RING slots ahead of the consumer, across chunk AND pair boundaries: the stream never drains between pairs, text keys,
// merge and output of one pair are worked off while the next pair's slots land.  A wave holds ~100 registers instead of
// 215, a workgroup is 4 waves x RING x 4 KiB of LDS = one per CU.
// The arithmetic is the one-wave kernel's, operation for operation (same chunks of ACS key steps, same max / rescale
// order, every fused multiply-add of the bookkeeping written out with `fp contract(off)` + explicit fmaf in BOTH kernels), so
// the two agree BIT FOR BIT over whole decodes: the engine selects between them by policy (engine.hip: streaming for a
// context alone with >= 384 (sentence, head) pairs and <= 256 padded keys, the register form otherwise), and
// gitmi_set_shared_device's promise of identical results rests on it.  Guards: tests/test_gpu_ops.py (streaming == register
// kernel on the unit entry) and tests/test_gpu_policy.py::test_shared_device_policy_is_bitwise_neutral (features, ids and
// log-probs of the benchmark geometry under both policies, bf16 / f16 / f32 builds).
// Ordering rules (MI355X_MICROARCH.md, LDS-DMA): a ds_read sees a DMA's bytes only after the issuing wave's counted vmcnt
// -- loads complete in order, so `vmcnt(4 * slots issued after this one)` retires the slot (other vector-memory operations
// issued in between only make that wait stricter); a slot is refilled only after the ds_reads that emptied it have
// returned (lgkmcnt(0)).  Rings are private to a wave: no workgroup barrier anywhere.
typedef __attribute__((address_space(3))) void attn_lds_void_t;

template <int N> __device__ __forceinline__ void attn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// younger slots still in flight are allowed to stay in flight: 4 LDS-DMA instructions per slot
__device__ __forceinline__ void attn_wait_slots_ahead(int ahead) {
    switch (ahead) {
        case 0: attn_wait_vm<0>(); break;
        case 1: attn_wait_vm<4>(); break;
        case 2: attn_wait_vm<8>(); break;
        case 3: attn_wait_vm<12>(); break;
        case 4: attn_wait_vm<16>(); break;
        case 5: attn_wait_vm<20>(); break;
        case 6: attn_wait_vm<24>(); break;
        case 7: attn_wait_vm<28>(); break;
        case 8: attn_wait_vm<32>(); break;
        case 9: attn_wait_vm<36>(); break;
        case 10: attn_wait_vm<40>(); break;
        default: attn_wait_vm<44>(); break;       // ahead >= 11: waiting for fewer outstanding operations is always safe
    }
}

constexpr int AS_WAVES = 4;                    // waves (independent streams) per workgroup
constexpr int AS_SLOT = 4096;                  // bytes: K or V^T of one 32-key step of one (image, head)

template <int KB, int TI, int RING>
__global__ __launch_bounds__(64 * AS_WAVES) void attn_decode_stream_kernel(AttnDecodeArgs a) {
#pragma clang fp contract(off)
    static_assert(RING >= 2 && RING <= 12, "ring depth");
    __shared__ __attribute__((aligned(16))) unsigned char ring_mem[AS_WAVES][RING * AS_SLOT];
    __shared__ float part[AS_WAVES][KB][HD + 2];   // [wave][beam]: o[64], m, l

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = a.d / HD;
    const int k = a.beams;
    const bf16_t* QKV = reinterpret_cast<const bf16_t*>(a.qkv);
    bf16_t* TK = reinterpret_cast<bf16_t*>(a.txt_k);
    bf16_t* TV = reinterpret_cast<bf16_t*>(a.txt_v);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.out);
    const int ld3 = 3 * a.d;
    const int Np = a.N_pad, nsteps = Np >> 5;
    unsigned char* ring = ring_mem[wave];

    // pairs of this wave: w, w + W, w + 2W, ...  (W = waves of the launch)
    const int W = (int)gridDim.x * AS_WAVES, w0 = (int)blockIdx.x * AS_WAVES + wave;
    const int npw = w0 < a.n_pairs ? (a.n_pairs - 1 - w0) / W + 1 : 0;
    const int slots_per_pair = 2 * nsteps;
    const int total_slots = npw * slots_per_pair;

    // ---- producer: the next slot of the stream -> ring position issued % RING.  All state is wave-uniform.
    int issued = 0;                      // slots requested so far
    int p_ord = 0, p_s0 = 0, p_v = 0, p_c = 0;     // pair ordinal, chunk start step, 0 = K / 1 = V^T, step within the chunk
    const bf16_t* p_K = nullptr; const bf16_t* p_V = nullptr;
    auto producer_pair = [&]() {
        const int pair = w0 + p_ord * W;
        const int h = pair % H, b = pair / H;
        const int bi = a.img_of ? a.img_of[b] : b;
        p_K = reinterpret_cast<const bf16_t*>(a.img_k) + ((size_t)bi * H + h) * Np * HD;
        p_V = reinterpret_cast<const bf16_t*>(a.img_v) + ((size_t)bi * H + h) * Np * HD;
    };
    if (npw > 0) producer_pair();
    auto issue_slot = [&]() {
        if (issued >= total_slots) return;
        const bf16_t* src = (p_v ? p_V : p_K) + (size_t)(p_s0 + p_c) * (AS_SLOT / 2) + lane * 8;
        unsigned char* dst = ring + (issued % RING) * AS_SLOT;
#pragma unroll
        for (int f = 0; f < 4; ++f)
            __builtin_amdgcn_global_load_lds((const void*)(src + f * 512), (attn_lds_void_t*)(dst + f * 1024), 16, 0, 2);   // aux 2 = nt
        ++issued;
        const int nsc = min(ACS, nsteps - p_s0);
        if (++p_c == nsc) {
            p_c = 0;
            if (p_v == 0) p_v = 1;
            else {
                p_v = 0; p_s0 += ACS;
                if (p_s0 >= nsteps) { p_s0 = 0; ++p_ord; if (p_ord < npw) producer_pair(); }
            }
        }
    };
    for (int i = 0; i < RING; ++i) issue_slot();

    int consumed = 0;                    // slots read out of the ring so far
    // wait for the oldest unread slot, return its ring address
    auto slot_ready = [&]() -> const unsigned char* {
        const int ahead = issued - consumed - 1;
        if (ahead == RING - 1) attn_wait_vm<4 * (RING - 1)>();        // steady state
        else attn_wait_slots_ahead(ahead);
        return ring + (consumed % RING) * AS_SLOT;
    };
    // the slot's fragments are in registers (the caller has used them): free it and keep the stream RING slots ahead
    auto slot_done = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++consumed;
        issue_slot();
    };

    for (int ord = 0; ord < npw; ++ord) {
        const int pair = w0 + ord * W;
        const int h = pair % H, b = pair / H;
        const int row0 = b * k;

        // ---- text items: the 8 eight-lane groups of the wave share them -------------------------------------------
        const int grp = lane >> 3, sub = lane & 7;
        const int nt = a.pos + 1;
        int t_j[TI], t_s[TI];
        u32x4_t tkr[TI], tvr[TI];
#pragma unroll
        for (int u = 0; u < TI; ++u) {
            const int it = grp + 8 * u;
            t_j[u] = it < k * nt ? it / nt : -1;
            t_s[u] = it < k * nt ? it % nt : 0;
            tkr[u] = u32x4_t{0u, 0u, 0u, 0u};
            tvr[u] = tkr[u];
            if (t_j[u] >= 0) {
                if (t_s[u] == a.pos) {
                    const bf16_t* src = QKV + (size_t)(row0 + t_j[u]) * ld3 + a.d + h * HD + sub * 8;
                    tkr[u] = *reinterpret_cast<const u32x4_t*>(src);
                    tvr[u] = *reinterpret_cast<const u32x4_t*>(src + a.d);
                } else {
                    const int srow = KB == 1 ? row0 : a.kv_src[(size_t)(row0 + t_j[u]) * a.ld_src + t_s[u]];
                    const size_t off = ((size_t)srow * a.T_max + t_s[u]) * a.d + h * HD + sub * 8;
                    tkr[u] = *reinterpret_cast<const u32x4_t*>(TK + off);
                    tvr[u] = *reinterpret_cast<const u32x4_t*>(TV + off);
                }
            }
        }
        // append this position's K/V of every beam to the text cache
        if (lane < k * 8) {
            const int j = lane >> 3;
            const bf16_t* src = QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8;
            const size_t dst = ((size_t)(row0 + j) * a.T_max + a.pos) * a.d + h * HD + sub * 8;
            *reinterpret_cast<u32x4_t*>(TK + dst) = *reinterpret_cast<const u32x4_t*>(src);
            *reinterpret_cast<u32x4_t*>(TV + dst) = *reinterpret_cast<const u32x4_t*>(src + a.d);
        }

        // ---- image part on the matrix cores: operands from the ring ------------------------------------------------
        bf16x8_t qf[2];
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (l15 < k) ld8bf(QKV + (size_t)(row0 + l15) * ld3 + h * HD + ds * 32 + lg * 8, qv);
            u32x4_t t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = pack2bf(qv[2 * e] * a.scale, qv[2 * e + 1] * a.scale);
            qf[ds] = __builtin_bit_cast(bf16x8_t, t);
        }
        float m_i = -INFINITY, l_i = 0.f;
        f32x4_t oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        for (int s0 = 0; s0 < nsteps; s0 += ACS) {
            f32x4_t sc[ACS][2];
            float cm = -INFINITY;
#pragma unroll
            for (int c = 0; c < ACS; ++c) {
                const int s = s0 + c;
                if (s < nsteps) {                                              // wave-uniform
                    const unsigned char* sl = slot_ready();
                    bf16x8_t kq[2][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int ds = 0; ds < 2; ++ds)
                            kq[t][ds] = *reinterpret_cast<const bf16x8_t*>(sl + (t * 2 + ds) * 1024 + lane * 16);
                    __builtin_amdgcn_sched_barrier(0);       // all four ds_reads in flight before the first MFMA waits for one
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        sc[c][t] = mfma16(kq[t][0], qf[0], f32x4_t{0.f, 0.f, 0.f, 0.f});
                        sc[c][t] = mfma16(kq[t][1], qf[1], sc[c][t]);
                    }
                    slot_done();
                } else {
                    sc[c][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    sc[c][1] = sc[c][0];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (s * 32 + t * 16 + lg * 4 + r >= a.N_img || s >= nsteps) sc[c][t][r] = -INFINITY;   // padded keys / steps
                        cm = fmaxf(cm, sc[c][t][r]);
                    }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float mn = fmaxf(m_i, cm);
            if (s0 != 0) {
                const float al = fast_exp(m_i - mn);
                l_i *= al;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { oacc[dt][0] *= al; oacc[dt][1] *= al; oacc[dt][2] *= al; oacc[dt][3] *= al; }
            }
            m_i = mn;
#pragma unroll
            for (int c = 0; c < ACS; ++c) {
                float p[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[t * 4 + r] = fast_exp(sc[c][t][r] - mn);    // exp(-inf) = 0 for padded keys
                        l_i += p[t * 4 + r];
                    }
                u32x4_t pp;
#pragma unroll
                for (int e = 0; e < 4; ++e) pp[e] = pack2bf(p[2 * e], p[2 * e + 1]);
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pp);
                if (s0 + c < nsteps) {
                    const unsigned char* sl = slot_ready();
                    bf16x8_t vq[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) vq[dt] = *reinterpret_cast<const bf16x8_t*>(sl + dt * 1024 + lane * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma16(vq[dt], pf, oacc[dt]);
                    slot_done();
                } else {
                    // the register kernels multiply an all-zero V^T step by P = 0 here: adds +0 to every accumulator
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma16(bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}, pf, oacc[dt]);
                }
            }
        }
        l_i += __shfl_xor(l_i, 16, 64);
        l_i += __shfl_xor(l_i, 32, 64);
        // ---- text keys (beam-specific): 8 lanes per key, online update ---------------------------------------------
        float q[KB][8], m[KB], l[KB], o[KB][8];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            m[j] = -INFINITY;
            l[j] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { q[j][e] = 0.f; o[j][e] = 0.f; }
            if (j < k) {
                ld8bf(QKV + (size_t)(row0 + j) * ld3 + h * HD + sub * 8, q[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) q[j][e] *= a.scale;
            }
        }
        auto dot8 = [&](const float (&x)[8], const float (&y)[8]) {
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) p = fmaf(x[e], y[e], p);
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            return p;
        };
        auto text_item = [&](int jj, const float (&kv)[8], const float (&vv)[8]) {
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                if (j == jj) {
                    const float sv = dot8(q[j], kv);
                    const float mn = fmaxf(m[j], sv);
                    const float al = fast_exp(m[j] - mn);
                    const float p = fast_exp(sv - mn);
                    l[j] = fmaf(l[j], al, p);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[j][e] = fmaf(o[j][e], al, p * vv[e]);
                    m[j] = mn;
                }
            }
        };
        auto unpack = [&](const u32x4_t& r, float (&v)[8]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) unpack2op(r[i], v[2 * i], v[2 * i + 1]);
        };
#pragma unroll
        for (int u = 0; u < TI; ++u) {
            if (t_j[u] >= 0) {
                float kv[8], vv[8];
                unpack(tkr[u], kv);
                unpack(tvr[u], vv);
                text_item(t_j[u], kv, vv);
            }
        }
        for (int it = grp + 8 * TI; it < k * nt; it += 8) {                  // long texts: dependent-load path
            const int j = it / nt, sidx = it % nt;
            float kv[8], vv[8];
            if (sidx == a.pos) {
                ld8bf(QKV + (size_t)(row0 + j) * ld3 + a.d + h * HD + sub * 8, kv);
                ld8bf(QKV + (size_t)(row0 + j) * ld3 + 2 * a.d + h * HD + sub * 8, vv);
            } else {
                const int srow = KB == 1 ? row0 : a.kv_src[(size_t)(row0 + j) * a.ld_src + sidx];
                ld8bf(TK + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, kv);
                ld8bf(TV + ((size_t)srow * a.T_max + sidx) * a.d + h * HD + sub * 8, vv);
            }
            text_item(j, kv, vv);
        }
        // merge the 8 groups of the wave (lanes with equal `sub`): lanes 0..7 end up with the wave's text partial
#pragma unroll
        for (int j = 0; j < KB; ++j) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                const float m2 = __shfl_xor(m[j], off, 64);
                const float l2 = __shfl_xor(l[j], off, 64);
                const float mn = fmaxf(m[j], m2);
                const float a1 = m[j] == -INFINITY ? 0.f : fast_exp(m[j] - mn);
                const float a2 = m2 == -INFINITY ? 0.f : fast_exp(m2 - mn);
                l[j] = fmaf(l[j], a1, l2 * a2);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float o2 = __shfl_xor(o[j][e], off, 64);
                    o[j][e] = fmaf(o[j][e], a1, o2 * a2);
                }
                m[j] = mn;
            }
        }
        // ---- the wave's partial per beam row = image keys + text items, through its own LDS slot -------------------
        if (l15 < k) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][l15][dt * 16 + lg * 4 + r] = oacc[dt][r];
            if (lg == 0) { part[wave][l15][HD] = m_i; part[wave][l15][HD + 1] = l_i; }
        }
        __builtin_amdgcn_wave_barrier();       // a wave reads back only what it wrote itself (its LDS operations complete in order)
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (lane < 8 && j < k) {
                const float mi = part[wave][j][HD], li = part[wave][j][HD + 1];
                const float mn = fmaxf(mi, m[j]);
                const float a1 = mi == -INFINITY ? 0.f : fast_exp(mi - mn);
                const float a2 = m[j] == -INFINITY ? 0.f : fast_exp(m[j] - mn);
#pragma unroll
                for (int e = 0; e < 8; ++e) part[wave][j][sub * 8 + e] = fmaf(part[wave][j][sub * 8 + e], a1, o[j][e] * a2);
                if (sub == 0) { part[wave][j][HD] = mn; part[wave][j][HD + 1] = fmaf(li, a1, l[j] * a2); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < k * HD; i += 64) {
            const int j = i / HD, dd = i % HD;
            const float r1 = part[wave][j][dd] / part[wave][j][HD + 1];
            if (a.out_frag) O[frag_offset(row0 + j, h * HD + dd, a.d >> 5)] = f2bf(r1);
            else O[(size_t)(row0 + j) * a.d + h * HD + dd] = f2bf(r1);
        }
        __builtin_amdgcn_wave_barrier();       // the next pair reuses this wave's LDS slot
    }
}

// ---- host launchers ------------------------------------------------------------------
hipError_t launch_kv_repack_frag(const void* qkv, void* kf, void* vt, int B, int N, int N_pad, int H, int d, hipStream_t s) {
    if (B <= 0 || N <= 0) return hipSuccess;
    if (N_pad % 32 || N_pad < N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kv_repack_frag_kernel, dim3(N_pad / 32, H, B), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)kf,
                       (bf16_t*)vt, N, N_pad, H, d);
    return hipGetLastError();
}

hipError_t launch_attn_decode_mfma(const AttnDecodeArgs& a, int B, int H, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (a.beams > 8 || a.N_pad % 32 || a.N_pad < a.N_img || a.N_img < 1) return hipErrorInvalidValue;
    AttnDecodeArgs p = a;
    p.n_pairs = B * H;
    // Which kernel: the GEOMETRY decides, so a model has ONE attention arithmetic (the two kernels differ in the last bits:
    // one partial per row instead of two).
    //  * ONE wave per (sentence, head) pair when the image keys fit two chunks (<= 8 key steps: one 224-pixel image).  Against
    //    two waves per pair it is 1.9 us slower per launch on an idle device (two memory round trips instead of one) and
    //    keeps half as many 214-register waves resident: +1.3 % captions/s in the mixed schedule, where every CU that holds
    //    one of these waves is closed to the image encoder's GEMM workgroups (profiles/r03_v_bench_lines.txt).
    //  * TWO waves per pair for long key sequences (6 video frames, a 480 x 640 VQA image: 37 steps would be 10 round trips
    //    in one wave, 5 in two; GIT_BASE_VATEX bs = 16: 1.78k captions/s with one wave, 1.90k with two).
    // waves_per_pair 1 / 2 (GITMI_ATTN_NH) force one of them for A/B.
    if (a.stream_wgs > 0) {
        // streaming kernel (the solo policy's choice, engine.hip; also the unit entry): workgroups = min(stream_wgs, pairs / 4)
        const int nwg = std::max(1, std::min(a.stream_wgs, (p.n_pairs + AS_WAVES - 1) / AS_WAVES));
        if (a.beams <= 1) hipLaunchKernelGGL((attn_decode_stream_kernel<1, 3, 9>), dim3(nwg), dim3(64 * AS_WAVES), 0, s, p);
        else if (a.beams <= 2) hipLaunchKernelGGL((attn_decode_stream_kernel<2, 3, 9>), dim3(nwg), dim3(64 * AS_WAVES), 0, s, p);
        else if (a.beams <= 4) hipLaunchKernelGGL((attn_decode_stream_kernel<4, 3, 9>), dim3(nwg), dim3(64 * AS_WAVES), 0, s, p);
        else hipLaunchKernelGGL((attn_decode_stream_kernel<8, 3, 8>), dim3(nwg), dim3(64 * AS_WAVES), 0, s, p);
        return hipGetLastError();
    }
    const bool one_wave = a.waves_per_pair == 1 || (a.waves_per_pair != 2 && a.N_pad <= 8 * 32);
    if (!one_wave) {
        const dim3 grid(H, B);
        if (a.beams <= 1) hipLaunchKernelGGL((attn_decode_mfma_kernel<1>), grid, dim3(128), 0, s, p);
        else if (a.beams <= 2) hipLaunchKernelGGL((attn_decode_mfma_kernel<2>), grid, dim3(128), 0, s, p);
        else if (a.beams <= 4) hipLaunchKernelGGL((attn_decode_mfma_kernel<4>), grid, dim3(128), 0, s, p);
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<8>), grid, dim3(128), 0, s, p);
        return hipGetLastError();
    }
    // pairs per workgroup (same per-wave arithmetic whatever the packing): 4 by default, 8 = a full CU under the serving
    // policy (10.58k -> 10.67k captions/s in the mixed schedule, +11 us per decode step alone; profiles/r03_y_*) -- but
    // never fewer than 96 workgroups: a small batch packed onto a handful of CUs streams its K/V through too few of them
    int pw = a.pairs_per_wg >= 8 ? 8 : a.pairs_per_wg >= 4 || a.pairs_per_wg <= 0 ? 4 : a.pairs_per_wg >= 2 ? 2 : 1;
    while (pw > 1 && p.n_pairs / pw < 96) pw >>= 1;
    // pairs per WAVE (one after the other, the same per-pair arithmetic): only on top of full workgroups, and never fewer
    // than 24 workgroups
    int ppw = pw == 8 && a.pairs_per_wave > 1 ? a.pairs_per_wave : 1;
    while (ppw > 1 && p.n_pairs / (pw * ppw) < 24) --ppw;
    p.pairs_per_wave = ppw;
    const int per_wg = pw * ppw;
#define GITMI_ATTN1(KBV)                                                                                                   \
    do {                                                                                                                   \
        if (pw == 8 && ppw > 1) hipLaunchKernelGGL((attn_decode_mfma_kernel<KBV, 3, 8, 1, true>), dim3((p.n_pairs + per_wg - 1) / per_wg), dim3(512), 0, s, p); \
        else if (pw == 8) hipLaunchKernelGGL((attn_decode_mfma_kernel<KBV, 3, 8, 1>), dim3((p.n_pairs + 7) / 8), dim3(512), 0, s, p); \
        else if (pw == 4) hipLaunchKernelGGL((attn_decode_mfma_kernel<KBV, 3, 4, 1>), dim3((p.n_pairs + 3) / 4), dim3(256), 0, s, p); \
        else if (pw == 2) hipLaunchKernelGGL((attn_decode_mfma_kernel<KBV, 3, 2, 1>), dim3((p.n_pairs + 1) / 2), dim3(128), 0, s, p); \
        else hipLaunchKernelGGL((attn_decode_mfma_kernel<KBV, 3, 1, 1>), dim3(p.n_pairs), dim3(64), 0, s, p);                         \
    } while (0)
    if (a.beams <= 1) GITMI_ATTN1(1);
    else if (a.beams <= 2) GITMI_ATTN1(2);
    else if (a.beams <= 4) GITMI_ATTN1(4);
    else { if (pw == 8) { pw = 4; ppw = 1; p.pairs_per_wave = 1; } GITMI_ATTN1(8); }       // 8 beams: 269 registers, one wave per SIMD: 4 pairs fill a CU
#undef GITMI_ATTN1
    return hipGetLastError();
}

}  // namespace gitmi
