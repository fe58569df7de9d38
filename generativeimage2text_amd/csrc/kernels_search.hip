// Device-side search: the whole bookkeeping of AutoRegressiveBeamSearch.search
// (decoder.py:224-440) and GeneratorWithBeamSearch.search + BeamHypotheses
// (decoder.py:1083-1341) runs on the device, so the decode loop has no host<->device
// synchronisation at all (the reference does ~10^3 .item() syncs per step at B=64, k=4).
//
//   row_topm     : (fp32 parity path / caller-supplied logits) per row: optional "no immediate repeat"
//                  (-10000 on the last token's logit, decoder.py:330), online max / sum-exp and the M best
//                  (logit, token) pairs, sorted -- written in the same "partial list" format the fused
//                  vocabulary head of the bf16 path emits (kernels_dgemm.hip), with one part per row.
//   search_step  : ONE workgroup per sentence and step: merges the row's partial candidate lists into the top-M
//                  log-probabilities (log-softmax = logit - logsumexp), then runs the step of the sentence's
//                  search class, re-orders the beam histories, and finally embeds the k chosen tokens
//                  (word + position + LayerNorm, decoder.py:65-78) for the next decode step -- selection,
//                  beam bookkeeping and the next step's embedding are one launch.
//                    AUTOREGRESSIVE: decoder.py:257-298 (first step), 313-417
//                    GENERATOR     : decoder.py:1169-1232 + BeamHypotheses 1292-1341 (n_hyp = num_keep_best <= SS_NHMAX)
//   finish       : select outputs.
//
// Sentences may carry their own prefix (VQA questions of different lengths in one batch; the reference runs them
// one at a time, decoder.py:984-989): while cur_len < plen[b] the step simply appends the given token.  All
// sentences sit at the same text position in every step, so no padding or position shifting is involved.
//
// Beams are re-ordered by index only: ids are gathered, and kv_src[row][pos] (the cache row
// that holds the K/V of text position pos for this row's history) is gathered with them.
#include "gitmi_common.h"
#include "launchers.h"

namespace gitmi {

// ---------------------------------------------------------------------------------------
template <int MMAX, int NT>
__global__ __launch_bounds__(NT) void row_topm_kernel(const float* __restrict__ logits, int ldl, int V,
                                                       const int* __restrict__ ids, int ld_ids, int cur_len,
                                                       const int* __restrict__ plen, int beams, int suppress_kind,
                                                       float rep_penalty, float* __restrict__ part_val,
                                                       int* __restrict__ part_idx, float2* __restrict__ part_lse) {
    constexpr int NW = NT / 64;
    __shared__ int s_hist[HIST_SLOTS];
    __shared__ float s_val[NT * MMAX];
    __shared__ int s_idx[NT * MMAX];
    __shared__ float s_red[2 * NW];
    __shared__ int s_redi[2 * NW];
    __shared__ int s_owner;

    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * ldl;
    const bool suppress = suppress_kind && cur_len > plen[r / beams];
    const int last = suppress ? ids[(size_t)r * ld_ids + cur_len - 1] : -1;
    float* cv = part_val + (size_t)r * MMAX;
    int* ci = part_idx + (size_t)r * MMAX;
    // repetition penalty (GENERATOR, decoder.py:1135-1144): the tokens of this row's history, as a set in LDS
    const bool pen = ids != nullptr && rep_penalty != 0.f && rep_penalty != 1.f;
    if (pen) {
        for (int i = tid; i < HIST_SLOTS; i += NT) s_hist[i] = -1;
        __syncthreads();
        for (int s = tid; s < cur_len; s += NT) hist_insert(s_hist, ids[(size_t)r * ld_ids + s]);
        __syncthreads();
    }

    float tv[MMAX];
    int ti[MMAX];
#pragma unroll
    for (int j = 0; j < MMAX; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    float mx = -INFINITY, sm = 0.f;
    auto feed = [&](float v, int i) {
        if (pen && hist_contains(s_hist, i)) v = rep_penalize(v, rep_penalty);
        if (i == last) v = -10000.f;
        // online log-sum-exp
        if (v > mx) { sm = sm * __expf(mx - v) + 1.f; mx = v; }
        else sm += __expf(v - mx);
        if (v > tv[MMAX - 1]) {
            tv[MMAX - 1] = v; ti[MMAX - 1] = i;
#pragma unroll
            for (int j = MMAX - 1; j > 0; --j) {
                if (tv[j] > tv[j - 1]) {
                    const float a = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = a;
                    const int c = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = c;
                }
            }
        }
    };
    if ((ldl & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        // 16-byte loads, four of them in flight per thread before the first compare
        const int nchunk = V >> 2;
        const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(x);
        int c = tid;
        for (; c + 3 * NT < nchunk; c += 4 * NT) {
            f32x4_t q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = x4[c + u * NT];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) feed(q[u][r], (c + u * NT) * 4 + r);
        }
        for (; c < nchunk; c += NT) {
            const f32x4_t q = x4[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) feed(q[r], c * 4 + r);
        }
        for (int i = (nchunk << 2) + tid; i < V; i += NT) feed(x[i], i);
    } else {
        for (int i = tid; i < V; i += NT) feed(x[i], i);
    }
    // block max / sum-exp
    float bmx = wave_max(mx);
    if (lane == 0) s_red[wave] = bmx;
    __syncthreads();
    bmx = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) bmx = fmaxf(bmx, s_red[w]);
    float part = mx == -INFINITY ? 0.f : sm * __expf(mx - bmx);
    part = wave_sum(part);
    if (lane == 0) s_red[NW + wave] = part;
#pragma unroll
    for (int j = 0; j < MMAX; ++j) { s_val[tid * MMAX + j] = tv[j]; s_idx[tid * MMAX + j] = ti[j]; }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += s_red[NW + w];
    if (tid == 0) part_lse[r] = float2{bmx, tot};
    __syncthreads();

    // MMAX rounds of block arg-max over the heads of the per-thread sorted lists
    int head = 0;
    for (int round = 0; round < MMAX; ++round) {
        float v = head < MMAX ? s_val[tid * MMAX + head] : -INFINITY;
        int id = head < MMAX ? s_idx[tid * MMAX + head] : 0x7fffffff;
        int who = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(id, o, 64);
            const int ow = __shfl_xor(who, o, 64);
            if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; who = ow; }
        }
        if (lane == 0) { s_red[wave] = v; s_redi[wave] = id; s_redi[NW + wave] = who; }
        __syncthreads();
        if (tid == 0) {
            float bv = s_red[0]; int bi = s_redi[0], bw = s_redi[NW];
#pragma unroll
            for (int w = 1; w < NW; ++w)
                if (s_red[w] > bv || (s_red[w] == bv && s_redi[w] < bi)) { bv = s_red[w]; bi = s_redi[w]; bw = s_redi[NW + w]; }
            cv[round] = bv;
            ci[round] = bi;
            s_owner = bw;
        }
        __syncthreads();
        if (tid == s_owner) ++head;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Sampling branch of GeneratorWithBeamSearch.search (decoder.py:1146-1166) for one row per workgroup:
//   x = logits / temperature
//   top_k_top_p_filtering(x, top_k, top_p, min_tokens_to_keep = 2)   (decoder.py:1343-1375)
//       top-k : remove x < (k-th largest x), k = min(max(top_k, 2), V)
//       top-p : in descending order, remove token i when the cumulative softmax mass of the tokens BEFORE it exceeds
//               top_p (the first token that crosses the threshold is kept); the flags of the first two are cleared
//               BEFORE the shift by one (decoder.py:1364-1369), so the first THREE positions always survive
//   draws     : `ndraw` tokens without replacement from softmax(filtered) -- Gumbel-top-k: the ndraw largest
//               log p_j + G_j, G_j = -log(-log u_j), u_j from a counter-based hash of (seed, step, row, j), in that order
//   output    : log_softmax(filtered)[draw]
// Both filters are thresholds on x, found by bit-wise bisection over an order-preserving integer image of the floats
// (32 counting / mass passes over the row held in registers) -- no sort of the 30522 logits.
constexpr int SMP_NT = 1024, SMP_PER = 32;          // up to 32768 tokens per row

__device__ __forceinline__ unsigned int f2key(float f) {      // monotone: a < b  <=>  key(a) < key(b)
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <typename T> __device__ __forceinline__ T block_sum_1024(T v, T* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    T t = 0;
#pragma unroll
    for (int w = 0; w < SMP_NT / 64; ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(SMP_NT) void sample_rows_kernel(const float* __restrict__ logits, int ldl, int V,
                                                             float inv_temp, int top_k, float top_p, int ndraw,
                                                             unsigned int seed_lo, unsigned int seed_hi, int step,
                                                             float* __restrict__ part_val, int* __restrict__ part_idx,
                                                             float2* __restrict__ part_lse, float* __restrict__ filtered_out,
                                                             const int* __restrict__ ids, int ld_ids, int cur_len,
                                                             float rep_penalty, int stride) {
    __shared__ int s_hist[HIST_SLOTS];
    __shared__ float sh_f[SMP_NT / 64];
    __shared__ int sh_i[SMP_NT / 64];
    __shared__ float s_bv[SMP_NT / 64];
    __shared__ int s_bi[SMP_NT / 64];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * ldl;
    // repetition penalty on the raw scores, before the temperature (decoder.py:1135-1149)
    const bool pen = ids != nullptr && rep_penalty != 0.f && rep_penalty != 1.f;
    if (pen) {
        for (int i = tid; i < HIST_SLOTS; i += SMP_NT) s_hist[i] = -1;
        __syncthreads();
        for (int s = tid; s < cur_len; s += SMP_NT) hist_insert(s_hist, ids[(size_t)r * ld_ids + s]);
        __syncthreads();
    }
    float xv[SMP_PER];
    unsigned int key[SMP_PER];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SMP_PER; ++i) {
        const int j = tid + i * SMP_NT;
        float raw = j < V ? x[j] : 0.f;
        if (pen && j < V && hist_contains(s_hist, j)) raw = rep_penalize(raw, rep_penalty);
        xv[i] = j < V ? raw * inv_temp : -INFINITY;
        key[i] = j < V ? f2key(xv[i]) : 0u;
        mx = fmaxf(mx, xv[i]);
    }
    // ---- top-k threshold: the largest key T with count(key >= T) >= kk  (= key of the kk-th largest) ------------------
    unsigned int thr = 0u;                           // keep key >= thr
    const int kk = top_k > 0 ? min(max(top_k, 2), V) : 0;
    if (kk > 0) {
        unsigned int T = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned int cand = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < SMP_PER; ++i) c += (key[i] >= cand && key[i] != 0u) ? 1 : 0;
            if (block_sum_1024<int>(c, sh_i) >= kk) T = cand;
        }
        thr = T;
    }
    // ---- softmax of what survived top-k ---------------------------------------------------------------------------
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) sh_f[wave] = mx;
    __syncthreads();
    float bmx = sh_f[0];
#pragma unroll
    for (int w = 1; w < SMP_NT / 64; ++w) bmx = fmaxf(bmx, sh_f[w]);
    float ev[SMP_PER];
    float es = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_PER; ++i) {
        ev[i] = (key[i] != 0u && key[i] >= thr) ? __expf(xv[i] - bmx) : 0.f;
        es += ev[i];
    }
    const float tot_k = block_sum_1024<float>(es, sh_f);
    // ---- top-p threshold: keep token i iff the mass of the tokens ranked strictly before it is <= top_p ---------------
    // (plus the first two).  P(T) := mass(key > T) <= top_p * total is monotone in T; T* = largest T with P false; keep key > T*.
    if (top_p > 0.f && top_p < 1.f) {
        const float lim = top_p * tot_k;
        unsigned int T = 0u;
        bool any_false = false;
        {
            float m0 = 0.f;
#pragma unroll
            for (int i = 0; i < SMP_PER; ++i) m0 += key[i] > 0u ? ev[i] : 0.f;
            any_false = block_sum_1024<float>(m0, sh_f) > lim;           // P(0) false?
        }
        if (any_false) {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned int cand = T | (1u << bit);
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < SMP_PER; ++i) m += key[i] > cand ? ev[i] : 0.f;
                if (block_sum_1024<float>(m, sh_f) > lim) T = cand;      // P(cand) false
            }
            unsigned int thr_p = T + 1u;                                   // keep key > T*
            // min_tokens_to_keep = 2, cleared before the shift: never above the key of the THIRD largest survivor
            unsigned int T2 = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned int cand = T2 | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < SMP_PER; ++i) c += (key[i] >= cand && key[i] >= thr && key[i] != 0u) ? 1 : 0;
                if (block_sum_1024<int>(c, sh_i) >= 3) T2 = cand;
            }
            thr_p = min(thr_p, T2);
            thr = max(thr, thr_p);
        }
    }
    // ---- filtered distribution: log-sum-exp over the kept tokens ------------------------------------------------------
    float ks = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_PER; ++i) {
        const bool keep = key[i] != 0u && key[i] >= thr;
        if (!keep) ev[i] = 0.f;
        ks += ev[i];
        if (filtered_out) {
            const int j = tid + i * SMP_NT;
            if (j < V) filtered_out[(size_t)r * V + j] = keep ? xv[i] : -INFINITY;
        }
    }
    const float lse = bmx + logf(block_sum_1024<float>(ks, sh_f));
    // ---- ndraw draws without replacement: the largest (log p + Gumbel) keys, one block arg-max per draw ---------------
    float gk[SMP_PER];
#pragma unroll
    for (int i = 0; i < SMP_PER; ++i) {
        const int j = tid + i * SMP_NT;
        if (ev[i] > 0.f) {
            unsigned int h = mix32(seed_lo ^ mix32((unsigned int)j + 0x9e3779b9u * (unsigned int)(step + 1)));
            h = mix32(h ^ seed_hi ^ (0x85ebca6bu * (unsigned int)(r + 1)));
            const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1)
            gk[i] = (xv[i] - lse) - logf(-logf(u));
        } else {
            gk[i] = -INFINITY;
        }
    }
    if (tid == 0) part_lse[r] = float2{0.f, 1.f};            // candidate values are log-probabilities already
    for (int d = 0; d < ndraw; ++d) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SMP_PER; ++i)
            if (gk[i] > bv) { bv = gk[i]; bi = tid + i * SMP_NT; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_bv[wave] = bv; s_bi[wave] = bi; }
        __syncthreads();
        bv = s_bv[0]; bi = s_bi[0];
#pragma unroll
        for (int w = 1; w < SMP_NT / 64; ++w)
            if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < bi)) { bv = s_bv[w]; bi = s_bi[w]; }
        // the owner of the winner publishes its log-probability and retires the token
#pragma unroll
        for (int i = 0; i < SMP_PER; ++i) {
            if (tid + i * SMP_NT == bi) {
                part_val[(size_t)r * stride + d] = xv[i] - lse;
                part_idx[(size_t)r * stride + d] = bi;
                gk[i] = -INFINITY;
            }
        }
        if (bi == 0x7fffffff && tid == 0) {       // fewer kept tokens than draws (cannot happen with min_tokens_to_keep = 2 <= ndraw ...)
            part_val[(size_t)r * stride + d] = -INFINITY;
            part_idx[(size_t)r * stride + d] = 0;
        }
    }
    // list entries beyond the draws (the search step reads lists of 1 / 2 / 4 / 8 / 16 entries)
    for (int d = ndraw + tid; d < stride; d += SMP_NT) {
        part_val[(size_t)r * stride + d] = -INFINITY;
        part_idx[(size_t)r * stride + d] = 0x7fffffff;
    }
}

// ---------------------------------------------------------------------------------------
// Trie-constrained greedy step (TrieAutoRegressiveBeamSearch.search, trie_decoder.py:57-71, 115-158), one workgroup per
// sentence (beam_size == 1).  On the step's logits x (the last token's logit set to -10000 after the first step, :121):
//     lp    = log_softmax(x)                                  computed as (x - max) - log(sum exp(x - max)), like torch
//     bonus = (max x - min x) + 1                             (:64, :151; -10000 is part of the min after the first step)
//     lp[t] += bonus  for every child token t of the sentence's trie cursor
//     (value, token) = top-1 of lp;  the cursor moves to the chosen child
// and the pair is handed to the search step as a one-entry candidate list whose value already is a log-probability
// (part_lse = (0, 1)); the step adds it to the running sum exactly like AutoRegressiveBeamSearch (:170-176).  Every
// sentence has its OWN cursor: it behaves like its own batch-1 reference call (the reference moves one cursor with
// row 0's choice and asserts as soon as row 0 has ended while another row has not).  A choice outside the trie -- the
// reference would assert in TokenTrie.move -- leaves the sentence unconstrained from then on (cursor -1).
constexpr int TRIE_NT = 1024;

__global__ __launch_bounds__(TRIE_NT) void trie_select_kernel(const float* __restrict__ logits, int ldl, int V,
                                                             const int* __restrict__ ids, int ld_ids, int cur_len,
                                                             const int* __restrict__ plen, int eos, TrieArgs tr,
                                                             float* __restrict__ part_val, int* __restrict__ part_idx,
                                                             float2* __restrict__ part_lse) {
    __shared__ float s_f[TRIE_NT / 64];
    __shared__ float s_g[TRIE_NT / 64];
    __shared__ int s_i[TRIE_NT / 64];
    __shared__ int s_j[TRIE_NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = TRIE_NT / 64;
    const int P = plen[b];
    if (cur_len < P) return;                                      // still inside the sentence's prefix: nothing to select
    const bool first = cur_len == P;
    const int last = ids[(size_t)b * ld_ids + cur_len - 1];
    if (!first && last == eos) {                                  // ended: the search step forces EOS at log-prob 0 (:138-142)
        if (tid == 0) { part_val[b] = 0.f; part_idx[b] = eos; part_lse[b] = float2{0.f, 1.f}; }
        return;
    }
    const float* x = logits + (size_t)b * ldl;
    const int sup = first ? -1 : last;
    // ---- max (lowest index on ties), min
    float mx = -INFINITY, mn = INFINITY;
    int am = 0x7fffffff;
    for (int i = tid; i < V; i += TRIE_NT) {
        const float v = i == sup ? -10000.f : x[i];
        if (v > mx) { mx = v; am = i; }
        mn = fminf(mn, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    if (lane == 0) { s_f[wave] = mx; s_i[wave] = am; s_g[wave] = mn; }
    __syncthreads();
    mx = s_f[0]; am = s_i[0]; mn = s_g[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        if (s_f[w] > mx || (s_f[w] == mx && s_i[w] < am)) { mx = s_f[w]; am = s_i[w]; }
        mn = fminf(mn, s_g[w]);
    }
    __syncthreads();
    // ---- log-sum-exp
    float sm = 0.f;
    for (int i = tid; i < V; i += TRIE_NT) sm += __expf((i == sup ? -10000.f : x[i]) - mx);
    sm = wave_sum(sm);
    if (lane == 0) s_f[wave] = sm;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += s_f[w];
    __syncthreads();
    const float logsum = logf(tot);
    const float bonus = (mx - mn) + 1.0f;
    // ---- best child of the cursor
    const int node = tr.cursor[b];
    const int e0 = node >= 0 ? tr.child_off[node] : 0, e1 = node >= 0 ? tr.child_off[node + 1] : 0;
    float bv = -INFINITY;
    int bt = 0x7fffffff, be = -1;
    for (int e = e0 + tid; e < e1; e += TRIE_NT) {
        const int t = tr.child_tok[e];
        if (t < 0 || t >= V) continue;
        const float xv = t == sup ? -10000.f : x[t];
        const float v = ((xv - mx) - logsum) + bonus;
        if (v > bv || (v == bv && t < bt)) { bv = v; bt = t; be = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int ot = __shfl_xor(bt, o, 64), oe = __shfl_xor(be, o, 64);
        if (ov > bv || (ov == bv && ot < bt)) { bv = ov; bt = ot; be = oe; }
    }
    if (lane == 0) { s_f[wave] = bv; s_i[wave] = bt; s_j[wave] = be; }
    __syncthreads();
    if (tid == 0) {
        bv = s_f[0]; bt = s_i[0]; be = s_j[0];
        for (int w = 1; w < NW; ++w)
            if (s_f[w] > bv || (s_f[w] == bv && s_i[w] < bt)) { bv = s_f[w]; bt = s_i[w]; be = s_j[w]; }
        const float plain = (mx - mx) - logsum;                   // log-prob of the unconstrained arg-max
        const bool take_valid = be >= 0 && (bv > plain || (bv == plain && bt <= am));
        part_val[b] = take_valid ? bv : plain;
        part_idx[b] = take_valid ? bt : am;
        part_lse[b] = float2{0.f, 1.f};
        tr.cursor[b] = take_valid ? tr.child_node[be] : -1;
    }
}

// ---------------------------------------------------------------------------------------
// Host arithmetic of the reference is Python double -> double here (decoder.py:1310-1341).
__device__ __forceinline__ double length_norm(int len, double alpha) {
    return pow(5.0 + (double)len, alpha) / pow(6.0, alpha);
}

constexpr int SS_KMAX = 8;        // beams per sentence
constexpr int SS_CMAX = 16;       // candidates per row (beam_size * per_node_beam_size <= 16)

__device__ __forceinline__ int ids_at(const SearchState& st, int buf, int r, int s) { return st.ids[buf][(size_t)r * st.T + s]; }

// grid = B (one workgroup per sentence), block = 256.  cur_len = tokens currently in ids (before appending).
// SLOTS = entries per partial list (compile time: the per-lane lists live in registers and are popped by static shifts)
template <typename TOut, int SLOTS>
__global__ __launch_bounds__(256) void search_step_kernel(SearchState st, int src, int cur_len, StepCands in,
                                                          EmbedArgs em) {
    __shared__ float c_val[SS_KMAX][SS_CMAX];  // merged top-M log-probabilities per beam row
    __shared__ int c_idx[SS_KMAX][SS_CMAX];
    __shared__ int sel_src[SS_KMAX], sel_word[SS_KMAX];
    __shared__ float sel_score[SS_KMAX];
    __shared__ int s_nadd, s_add_slot[SS_CMAX * 2], s_add_row[SS_CMAX * 2];   // GENERATOR: hypotheses added this step (slot <- row)
    // Per-sentence state of the search, staged by many threads at once (round 6): the bookkeeping of phase B is ONE thread
    // walking a chain of dependent reads -- from global memory that was ~45 exposed round trips of a beam-4 step (36 us per
    // launch); from LDS they cost a few cycles each.  What phase B changes is written back by the same thread.
    __shared__ float s_score[SS_KMAX];
    __shared__ int s_last[SS_KMAX];
    __shared__ int s_done, s_hyp_n, s_hyp_cnt, s_stop;
    __shared__ double s_hyp_worst, s_len_cur, s_len_last;
    __shared__ double s_hscore[SS_NHMAX];
    __shared__ int s_hseq[SS_NHMAX];
    __shared__ float s_part[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int k = st.k, pn = st.pn, T = st.T;
    const int dst = src ^ 1;
    const int P = st.plen[b];
    const bool forced = cur_len < P;                       // still inside this sentence's prefix
    const bool first = cur_len == P;                       // first search step of this sentence
    const int M = st.kind == 0 ? (first ? k : pn) : (st.sampled ? pn : pn * k);   // candidates needed per row

    if (tid == 0) s_nadd = 0;
    {   // staging loads: one thread per value, all in flight together with the candidate lists of phase A
        const int t2 = tid - 64;                           // wave 1 (its lanes are idle until their row's lists arrive anyway)
        if (t2 >= 0 && t2 < k) {
            s_score[t2] = st.score[src][b * k + t2];
            s_last[t2] = cur_len > 0 ? ids_at(st, src, b * k + t2, cur_len - 1) : -1;
        }
        if (t2 == 8) s_done = st.done[b];
        if (t2 == 9) s_hyp_n = st.hyp_n[b];
        if (t2 == 10) s_hyp_cnt = st.hyp_cnt[b];
        if (t2 == 11) s_stop = st.stop[b];
        if (t2 == 12) s_hyp_worst = st.hyp_worst[b];
        if (t2 == 13) s_len_cur = st.len_norm[cur_len];
        if (t2 == 14) s_len_last = st.len_norm[T - 1];
        if (t2 >= 16 && t2 < 16 + st.nh) {
            s_hscore[t2 - 16] = st.hyp_score[(size_t)b * st.nh + t2 - 16];
            s_hseq[t2 - 16] = st.hyp_seq[(size_t)b * st.nh + t2 - 16];
        }
    }

    // ---- phase A: merged top-M log-probabilities of every beam row of the sentence.  ONE WAVE PER ROW (rows j = wave,
    // wave + 4): a lane holds the sorted partial lists of parts lane, lane + 64, lane + 128, lane + 192 in registers, a
    // round = lane-local best head -> wave arg-max (value, then lower token) -> the owner pops its head.  No workgroup
    // barrier inside (round 2: the four beams of a sentence were merged one after the other with three __syncthreads
    // per round: 63 us per step for beam 4; profiles/r02_d_beam4_kernel_stats.txt).
    if (!forced) {
        for (int j = wave; j < k; j += 4) {
            const int r = b * k + j;
            if (st.kind == 0 && first && j > 0) break;     // the first step expands beam 0 only (decoder.py:257-271)
            const int last = ids_at(st, src, r, cur_len - 1);
            if (st.kind == 0 && !first && last == st.eos) {
                // one-hot distribution on EOS (decoder.py:300-310, 347-351): log-prob 0, everything else -inf
                if (lane < M) {
                    c_val[j][lane] = lane == 0 ? 0.f : -INFINITY;
                    c_idx[j][lane] = lane == 0 ? st.eos : (lane - 1 < st.eos ? lane - 1 : lane);
                }
                continue;
            }
            // this lane's lists + the log-sum-exp over the parts (fixed order: part index -> lane tree)
            float hv[4][SLOTS];
            int hi[4][SLOTS];
            float pm = -INFINITY, ps = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = lane + 64 * q;
                const bool ok = p < in.nparts;
                if (ok) {
                    const float2 ml = in.part_lse[(size_t)r * in.nparts + p];
                    if (ml.x > pm) { ps = ps * __expf(pm - ml.x) + ml.y; pm = ml.x; }
                    else if (ml.x != -INFINITY) ps += ml.y * __expf(ml.x - pm);
                }
                const size_t base = ((size_t)r * in.nparts + (ok ? p : 0)) * SLOTS;
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) {
                    hv[q][e] = ok ? in.part_val[base + e] : -INFINITY;
                    hi[q][e] = ok ? in.part_idx[base + e] : 0x7fffffff;
                }
            }
            const float bm = wave_max(pm);
            const float part = wave_sum(pm == -INFINITY ? 0.f : ps * __expf(pm - bm));
            const float lse = bm + logf(part);
            for (int round = 0; round < M; ++round) {
                float v = hv[0][0];
                int id = hi[0][0], who = 0;
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (hv[q][0] > v || (hv[q][0] == v && hi[q][0] < id)) { v = hv[q][0]; id = hi[q][0]; who = q; }
                int own = lane;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(v, o, 64);
                    const int oi = __shfl_xor(id, o, 64);
                    const int ow = __shfl_xor(own, o, 64);
                    if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; own = ow; }
                }
                if (lane == 0) {
                    c_val[j][round] = v - lse;
                    c_idx[j][round] = id == 0x7fffffff ? 0 : id;
                }
                if (own == lane) {                          // pop the winner's head (static shifts)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q == who) {
#pragma unroll
                            for (int e = 0; e + 1 < SLOTS; ++e) { hv[q][e] = hv[q][e + 1]; hi[q][e] = hi[q][e + 1]; }
                            hv[q][SLOTS - 1] = -INFINITY; hi[q][SLOTS - 1] = 0x7fffffff;
                        }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase B: the search step of this sentence (one thread; k <= 8, <= 16 candidates per row) ---------------
    if (tid == 0) {
        if (forced) {
            const int word = (int)st.start[(size_t)b * st.ld_start + cur_len];
            for (int j = 0; j < k; ++j) { sel_src[j] = b * k + j; sel_word[j] = word; sel_score[j] = st.score[src][b * k + j]; }
        } else if (st.kind == 0) {
            // ---- AutoRegressiveBeamSearch -------------------------------------------------------------------
            if (!first && s_stop == 0) {
                // decoder.py:319: the loop stops once every beam's last token is EOS (later steps are idempotent)
                int all_eos = 1;
                for (int j = 0; j < k; ++j)
                    if (s_last[j] != st.eos) all_eos = 0;
                if (all_eos) { st.stop[b] = cur_len; atomicAdd(&st.info[0], 1); }
            }
            if (first) {
                for (int j = 0; j < k; ++j) {
                    sel_src[j] = b * k; sel_word[j] = c_idx[0][j]; sel_score[j] = c_val[0][j];
                }
                // decoder.py:279-291: every first prediction is EOS (k == 1) -> early return
                if (k == 1) st.early[b] = c_idx[0][0] == st.eos ? 1 : 0;
            } else {
                // top-k (sorted) of the k*pn summed candidates, ties -> lower candidate index
                unsigned long long used = 0ull;
                for (int j = 0; j < k; ++j) {
                    float best = 0.f; int bc = -1;
                    for (int c = 0; c < k * pn; ++c) {
                        if (used >> c & 1ull) continue;
                        const float v = c_val[c / pn][c % pn] + s_score[c / pn];
                        if (bc < 0 || v > best) { best = v; bc = c; }
                    }
                    used |= 1ull << bc;
                    sel_src[j] = b * k + bc / pn; sel_word[j] = c_idx[bc / pn][bc % pn]; sel_score[j] = best;
                }
            }
        } else {
            // ---- GeneratorWithBeamSearch ----------------------------------------------------------------------
            const int V = st.V;
            const int ncand = pn * k;              // "2k"
            // merge: each row's list is sorted by log-prob; adding the row's beam score keeps the order
            int headp[SS_KMAX];
            for (int j = 0; j < k; ++j) headp[j] = 0;
            float n_score[SS_CMAX * 2]; int n_beam[SS_CMAX * 2], n_word[SS_CMAX * 2];
            float n_max = -INFINITY;
            for (int c = 0; c < ncand; ++c) {
                if (st.sampled) {
                    // sampling branch (decoder.py:1155-1166): the per_node draws of beam 0, then of beam 1, ... in draw order.
                    // Word and score of candidate c come from beam row c / pn, but the reference attaches it to beam
                    // c % k: its beam offsets are arange(k) * V tiled per_node times over the flattened [k, pn] draws
                    // (beam_indices.repeat(batch, per_node_beam_size), decoder.py:1161-1164).  Reproduced as is.
                    const int j = c / pn, d = c % pn;
                    n_score[c] = c_val[j][d] + s_score[j]; n_beam[c] = c % k; n_word[c] = c_idx[j][d];
                } else {
                    float best = 0.f; int bj = -1; long long bflat = 0;
                    for (int j = 0; j < k; ++j) {
                        if (headp[j] >= M) continue;
                        const float v = c_val[j][headp[j]] + s_score[j];
                        const long long flat = (long long)j * V + c_idx[j][headp[j]];
                        if (bj < 0 || v > best || (v == best && flat < bflat)) { best = v; bj = j; bflat = flat; }
                    }
                    n_score[c] = best; n_beam[c] = bj; n_word[c] = c_idx[bj][headp[bj]];
                    headp[bj]++;
                }
                n_max = fmaxf(n_max, n_score[c]);
            }
            const int nh = st.nh;
            double* h_score = st.hyp_score + (size_t)b * nh;
            int* h_len = st.hyp_len + (size_t)b * nh;
            int* h_seq = st.hyp_seq + (size_t)b * nh;
            const int was_done = s_done;
            int is_done = was_done;
            int hyp_n = s_hyp_n, hyp_cnt = s_hyp_cnt;
            double hyp_worst = s_hyp_worst;
            if (!is_done && hyp_n >= nh) {
                // BeamHypotheses.is_done(max next score); self.max_length = max_length - 1
                is_done = hyp_worst >= (double)n_max / s_len_last;
            }
            if (is_done && !was_done) { atomicAdd(&st.info[0], 1); st.done[b] = is_done; }
            int nb = 0;
            if (!is_done) {
                for (int c = 0; c < ncand && nb < k; ++c) {
                    const int word = n_word[c];
                    if (word == st.eos || cur_len + 1 == T) {
                        // hyps.add(input_ids[row, :cur_len], score)
                        // BeamHypotheses.add (decoder.py:1316-1328): keep the nh best; a full list drops its lowest score
                        // (sorted() on (score, list index): the earliest-added of equal scores) and worst_score becomes the
                        // lowest score that is left
                        const double sc = (double)n_score[c] / s_len_cur;
                        const int n = hyp_n;
                        if (n < nh || sc > hyp_worst) {
                            int slot = n;
                            if (n < nh) {
                                hyp_n = n + 1;
                                hyp_worst = sc < hyp_worst ? sc : hyp_worst;
                            } else {
                                slot = 0;
                                for (int i = 1; i < nh; ++i)
                                    if (s_hscore[i] < s_hscore[slot] || (s_hscore[i] == s_hscore[slot] && s_hseq[i] < s_hseq[slot])) slot = i;
                                double w = sc;
                                for (int i = 0; i < nh; ++i)
                                    if (i != slot && s_hscore[i] < w) w = s_hscore[i];
                                hyp_worst = w;
                            }
                            s_hscore[slot] = sc; s_hseq[slot] = hyp_cnt;
                            h_score[slot] = sc;
                            h_len[slot] = cur_len;
                            h_seq[slot] = hyp_cnt++;
                            s_add_slot[s_nadd] = slot; s_add_row[s_nadd] = b * k + n_beam[c];
                            ++s_nadd;
                        }
                    } else {
                        sel_score[nb] = n_score[c]; sel_word[nb] = word; sel_src[nb] = b * k + n_beam[c];
                        ++nb;
                    }
                }
                if (hyp_n != s_hyp_n) st.hyp_n[b] = hyp_n;
                if (hyp_cnt != s_hyp_cnt) st.hyp_cnt[b] = hyp_cnt;
                if (hyp_worst != s_hyp_worst) st.hyp_worst[b] = hyp_worst;
            }
            if (nb < k) {
                // done sentence, or every candidate finished (cur_len + 1 == max_length): pad the batch with
                // (score 0, EOS, row 0 of the call)  -- decoder.py:1189, 1219-1220.  A sentence of a batched ragged call
                // stands for its own batch-1 reference call, whose row 0 is the sentence's own first row.
                const int pad_row = st.ragged ? b * k : 0;
                for (int j = 0; j < k; ++j) { sel_score[j] = 0.f; sel_word[j] = st.eos; sel_src[j] = pad_row; }
            }
        }
        if (b == 0) st.info[2] += 1;
    }
    __syncthreads();

    // ---- phase C: histories of the new beams (all threads), new token, scores -------------------------------------
    for (int a = 0; a < s_nadd; ++a) {           // in the order of the adds: a slot refilled twice in one step keeps the later history
        int* ht = st.hyp_tok + ((size_t)b * st.nh + s_add_slot[a]) * T;
        for (int s = tid; s < cur_len; s += 256) ht[s] = ids_at(st, src, s_add_row[a], s);
    }
    // all k rows in one sweep (one round trip, not k: the compiler cannot move row j + 1's loads above row j's stores)
    for (int idx = tid; idx < k * cur_len; idx += 256) {
        const int j = idx / cur_len, s = idx - j * cur_len;
        const int r = b * k + j, rs = sel_src[j];
        st.ids[dst][(size_t)r * T + s] = st.ids[src][(size_t)rs * T + s];
        st.kv_src[dst][(size_t)r * T + s] = st.kv_src[src][(size_t)rs * T + s];
    }
    if (tid < k) {
        const int r = b * k + tid;
        st.ids[dst][(size_t)r * T + cur_len] = sel_word[tid];
        st.kv_src[dst][(size_t)r * T + cur_len] = r;
        st.score[dst][r] = sel_score[tid];
    }

    // ---- phase D: embedding + LayerNorm of the chosen tokens = input of the next decode step -----------------------
    // One WAVE per beam row (rows j = wave, wave + 4), all rows at once: a lane owns the float4 chunks lane, lane + 64, ...
    // of its row.  (Until round 6 the 256 threads took the rows one after the other, one exposed embedding-row fetch and three
    // barriers each.)  The sums are formed exactly as before -- the partial of chunk group i is what wave i's wave_sum used to
    // give, the groups are added in order -- so the rows are bit-identical to the earlier form and to embed_ln_kernel's.
    if (em.words == nullptr || cur_len + 1 >= T) return;
    const int D = em.D;
    if (k == 1) {
        // one row: the whole workgroup on it, one float4 per thread (a single wave would carry four reduction trees in a
        // row: measured 8.8 -> 10.1 us per greedy step)
        const int r = b;
        int tok = sel_word[0];
        tok = tok < 0 ? 0 : (tok >= em.vocab ? em.vocab - 1 : tok);
        const int c = tid * 4;
        const bool on = c < D;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, g4 = a, b4 = a;
        if (on) {
            const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(em.words + (size_t)tok * D + c);
            const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(em.positions + (size_t)cur_len * D + c);
            g4 = *reinterpret_cast<const f32x4_t*>(em.gamma + c);
            b4 = *reinterpret_cast<const f32x4_t*>(em.beta + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = w4[q] + p4[q];
        }
        const float sum = wave_sum(a[0] + a[1] + a[2] + a[3]);
        if (lane == 0) s_part[wave] = sum;
        __syncthreads();
        const float mean = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)D;
        float q2 = 0.f;
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float d = a[q] - mean; q2 += d * d; }
        }
        q2 = wave_sum(q2);
        if (lane == 0) s_part[4 + wave] = q2;
        __syncthreads();
        const float rstd = rsqrtf((s_part[4] + s_part[5] + s_part[6] + s_part[7]) / (float)D + em.eps);
        if (on) {
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (a[q] - mean) * rstd * g4[q] + b4[q];
            *reinterpret_cast<f32x4_t*>(em.h_f + (size_t)r * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
            TOut* ht = reinterpret_cast<TOut*>(em.h_t);
            if constexpr (sizeof(TOut) == 4) {
                *reinterpret_cast<f32x4_t*>(ht + (size_t)r * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
            } else {
                uint2 t;
                t.x = pack2bf(o[0], o[1]);
                t.y = pack2bf(o[2], o[3]);
                const size_t off = em.frag ? frag_offset(r, c, D >> 5) : (size_t)r * D + c;
                *reinterpret_cast<uint2*>(ht + off) = t;
            }
        }
        return;
    }
    for (int j = wave; j < k; j += 4) {
        const int r = b * k + j;
        int tok = sel_word[j];
        tok = tok < 0 ? 0 : (tok >= em.vocab ? em.vocab - 1 : tok);
        f32x4_t a[4], g4[4], b4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            a[i] = g4[i] = b4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(em.words + (size_t)tok * D + c);
                const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(em.positions + (size_t)cur_len * D + c);
                g4[i] = *reinterpret_cast<const f32x4_t*>(em.gamma + c);
                b4[i] = *reinterpret_cast<const f32x4_t*>(em.beta + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[i][q] = w4[q] + p4[q];
            }
        }
        float part[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] = wave_sum(a[i][0] + a[i][1] + a[i][2] + a[i][3]);
        const float mean = (part[0] + part[1] + part[2] + part[3]) / (float)D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float q2 = 0.f;
            if ((lane + 64 * i) * 4 < D) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float d = a[i][q] - mean; q2 += d * d; }
            }
            part[i] = wave_sum(q2);
        }
        const float rstd = rsqrtf((part[0] + part[1] + part[2] + part[3]) / (float)D + em.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (a[i][q] - mean) * rstd * g4[i][q] + b4[i][q];
                *reinterpret_cast<f32x4_t*>(em.h_f + (size_t)r * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
                TOut* ht = reinterpret_cast<TOut*>(em.h_t);
                if constexpr (sizeof(TOut) == 4) {
                    *reinterpret_cast<f32x4_t*>(ht + (size_t)r * D + c) = f32x4_t{o[0], o[1], o[2], o[3]};
                } else {
                    uint2 t;
                    t.x = pack2bf(o[0], o[1]);
                    t.y = pack2bf(o[2], o[3]);
                    const size_t off = em.frag ? frag_offset(r, c, D >> 5) : (size_t)r * D + c;
                    *reinterpret_cast<uint2*>(ht + off) = t;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
__global__ void search_init_kernel(SearchState st) {
    const int R = st.B * st.k;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * st.T; i += gridDim.x * blockDim.x) {
        const int r = i / st.T, s = i % st.T;
        const int b = r / st.k;
        st.ids[0][i] = s < st.plen[b] ? (int)st.start[(size_t)b * st.ld_start + s] : st.eos;
        st.kv_src[0][i] = r;
    }
    if (blockIdx.x == 0) {
        for (int r = threadIdx.x; r < R; r += blockDim.x) {
            // decoder.py:1118-1120: only beam 0 is live at the start (GENERATOR); AUTOREGRESSIVE
            // takes its first step from beam 0 explicitly.
            st.score[0][r] = (st.kind == 1 && r % st.k > 0) ? -1e9f : 0.f;
        }
        for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
            st.done[b] = 0; st.hyp_n[b] = 0; st.hyp_cnt[b] = 0; st.hyp_worst[b] = 1e9; st.stop[b] = 0; st.early[b] = 0;
        }
        if (threadIdx.x < 4) st.info[threadIdx.x] = 0;
        // BeamHypotheses length norm ((5 + len) / 6) ** length_penalty for every length, once per search: the step kernel's
        // single bookkeeping thread otherwise evaluates several double-precision pow() per step
        for (int len = threadIdx.x; len <= st.T; len += blockDim.x) st.len_norm[len] = length_norm(len, st.length_penalty);
    }
}

// AUTOREGRESSIVE: best beam (index 0), log-prob / num_valid (decoder.py:429-438)
// GENERATOR     : best hypothesis + EOS, EOS padded; -1e5 when none (decoder.py:1264-1290)
// sent_out[b] = (length of the sequence the reference returns for this sentence alone, early-return flag)
__global__ void search_finish_kernel(SearchState st, int cur, int cur_len, long long* __restrict__ tokens_out,
                                     float* __restrict__ logprob_out, int* __restrict__ info_out,
                                     int* __restrict__ sent_out) {
    __shared__ int s_max_stop, s_all_stop, s_all_early;
    if (threadIdx.x == 0) { s_max_stop = 0; s_all_stop = 1; s_all_early = 1; }
    __syncthreads();
    for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
        const int nout = st.kind == 0 ? 1 : st.nh;             // GENERATOR: num_keep_best sequences per sentence
        long long* out = tokens_out + (size_t)b * nout * st.T;
        int L = st.T, early = 0;
        if (st.kind == 0) {
            const int r = b * st.k;
            const int stop = st.stop[b];
            L = stop > 0 ? stop : cur_len;                      // length the reference returns for this sentence
            early = st.k == 1 ? st.early[b] : 0;
            int non_eos = 0, any_eos = 0;
            for (int s = 0; s < st.T; ++s) {
                const int tok = s < L ? st.ids[cur][(size_t)r * st.T + s] : st.eos;
                out[s] = tok;
                if (s < L) { if (tok != st.eos) ++non_eos; else any_eos = 1; }
            }
            int nv = non_eos + any_eos - st.plen[b];
            nv = nv < 1 ? 1 : nv;
            logprob_out[b] = st.score[cur][r] / (float)nv;
            if (sent_out) sent_out[2 * b + 1] = early;
            if (stop > 0) atomicMax(&s_max_stop, stop); else s_all_stop = 0;
            if (!early) s_all_early = 0;
        } else {
            // decoder.py:1264-1290: the hypotheses by descending score (torch.topk over the list; equal scores in list
            // order), each followed by EOS and EOS-padded; sequences the list cannot fill are all EOS with log-prob -1e5
            const int nh = st.nh, have = st.hyp_n[b];
            const double* h_score = st.hyp_score + (size_t)b * nh;
            const int* h_seq = st.hyp_seq + (size_t)b * nh;
            unsigned taken = 0u;
            for (int i = 0; i < nh; ++i) {
                int best = -1;
                for (int j = 0; j < have; ++j) {
                    if ((taken >> j) & 1u) continue;
                    if (best < 0 || h_score[j] > h_score[best] || (h_score[j] == h_score[best] && h_seq[j] < h_seq[best])) best = j;
                }
                const int n = best >= 0 ? st.hyp_len[(size_t)b * nh + best] : 0;
                const int* ht = st.hyp_tok + ((size_t)b * nh + (best >= 0 ? best : 0)) * st.T;
                for (int s = 0; s < st.T; ++s) out[(size_t)i * st.T + s] = s < n ? ht[s] : st.eos;
                logprob_out[(size_t)b * nh + i] = best >= 0 ? (float)h_score[best] : -1e5f;
                if (best >= 0) taken |= 1u << best;
            }
            if (sent_out) sent_out[2 * b + 1] = 0;
        }
        if (sent_out) sent_out[2 * b] = L;
    }
    __syncthreads();
    // (decoder.py:279-291, the first-step early return, hands back the raw first log-prob: with the sequence
    //  [prefix, EOS] num_valid is 1, so the normalised value above is already that number)
    if (threadIdx.x == 0) {
        info_out[0] = st.kind == 0 ? (s_all_stop ? s_max_stop : cur_len) : st.T;
        info_out[1] = (st.kind == 0 && st.k == 1) ? s_all_early : 0;
        info_out[2] = st.info[2];
        // sequences whose log-prob is not finite: an operand left the range of the 16-bit format somewhere upstream (an inf in
        // any activation turns the LayerNorm statistics, the softmax and the log-sum-exp of its sentence into NaN) -- the host
        // side raises instead of returning garbage ids
        const int nout = st.kind == 0 ? 1 : st.nh;
        int bad = 0;
        for (int i = 0; i < st.B * nout; ++i) bad += !isfinite(logprob_out[i]);
        info_out[3] = bad;
    }
}

__global__ void search_rows_kernel(SearchState st, int cur, int cur_len, long long* __restrict__ out) {
    const int R = st.B * st.k;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * cur_len; i += gridDim.x * blockDim.x) {
        const int r = i / cur_len, s = i % cur_len;
        out[i] = st.ids[cur][(size_t)r * st.T + s];
    }
}

// teacher-forced rows for gitmi_step_logits: ids <- tokens, identity KV indirection
__global__ void load_ids_kernel(const long long* __restrict__ tokens, int R, int t, int* __restrict__ ids,
                                int* __restrict__ kv_src, int ld) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * ld; i += gridDim.x * blockDim.x) {
        const int r = i / ld, s = i % ld;
        ids[i] = s < t ? (int)tokens[(size_t)r * t + s] : 0;
        kv_src[i] = r;
    }
}

// start tokens [B, ld]: row b = its own prefix (prefixes [B, ldp], length plen[b]), or the shared prefix, or [CLS]
// (decoder.py:979-989) -- no host round trip
__global__ void fill_start_kernel(long long* __restrict__ start, int ld, const long long* __restrict__ prefix, int ldp,
                                  int shared, int sos, int B, int P) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * P; i += gridDim.x * blockDim.x) {
        const int b = i / P, s = i % P;
        start[(size_t)b * ld + s] = prefix ? prefix[(shared ? 0 : (size_t)b * ldp) + s] : (long long)sos;
    }
}

// ---- host launchers ------------------------------------------------------------------
hipError_t launch_fill_start(long long* start, int ld, const long long* prefix, int ldp, int shared, int sos, int B, int P,
                             hipStream_t s) {
    hipLaunchKernelGGL(fill_start_kernel, dim3(8), dim3(256), 0, s, start, ld, prefix, ldp, shared, sos, B, P);
    return hipGetLastError();
}

hipError_t launch_load_ids(const long long* tokens, int R, int t, int* ids, int* kv_src, int ld, hipStream_t s) {
    hipLaunchKernelGGL(load_ids_kernel, dim3(64), dim3(256), 0, s, tokens, R, t, ids, kv_src, ld);
    return hipGetLastError();
}

int row_topm_slots(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : M <= 8 ? 8 : 16; }

hipError_t launch_row_topm(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len,
                           const int* plen, int beams, int suppress_kind, float rep_penalty, int M, int R,
                           float* part_val, int* part_idx, float2* part_lse, hipStream_t s) {
    if (M < 1 || M > 16) return hipErrorInvalidValue;
    if (rep_penalty != 0.f && rep_penalty != 1.f && ids && cur_len > HIST_SLOTS / 2) return hipErrorInvalidValue;
    // one workgroup per row; more threads per row when the per-thread candidate list is short (LDS: NT*MMAX*8 B)
#define GITMI_TOPM(MM, NTT)                                                                                         \
    hipLaunchKernelGGL((row_topm_kernel<MM, NTT>), dim3(R), dim3(NTT), 0, s, logits, ldl, V, ids, ld_ids, cur_len, plen, \
                       beams, suppress_kind, rep_penalty, part_val, part_idx, part_lse)
    if (M <= 1) GITMI_TOPM(1, 1024);
    else if (M <= 2) GITMI_TOPM(2, 1024);
    else if (M <= 4) GITMI_TOPM(4, 1024);
    else if (M <= 8) GITMI_TOPM(8, 512);
    else GITMI_TOPM(16, 256);
#undef GITMI_TOPM
    return hipGetLastError();
}

hipError_t launch_sample_rows(const float* logits, int ldl, int V, int R, float temperature, int top_k, float top_p,
                              int ndraw, unsigned long long seed, int step, float* part_val, int* part_idx,
                              float2* part_lse, float* filtered_out, const int* ids, int ld_ids, int cur_len,
                              float rep_penalty, int stride, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if (stride < ndraw) return hipErrorInvalidValue;
    if (rep_penalty != 0.f && rep_penalty != 1.f && ids && cur_len > HIST_SLOTS / 2) return hipErrorInvalidValue;
    if (V > SMP_NT * SMP_PER || V < 2 || ndraw < 1 || ndraw > SS_CMAX || !(temperature > 0.f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_rows_kernel, dim3(R), dim3(SMP_NT), 0, s, logits, ldl, V, 1.0f / temperature, top_k, top_p, ndraw,
                       (unsigned int)(seed & 0xffffffffu), (unsigned int)(seed >> 32), step, part_val, part_idx, part_lse,
                       filtered_out, ids, ld_ids, cur_len, rep_penalty, stride);
    return hipGetLastError();
}

hipError_t launch_trie_select(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len, const int* plen,
                              int eos, const TrieArgs& tr, int B, float* part_val, int* part_idx, float2* part_lse,
                              hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!tr.child_off || !tr.child_tok || !tr.child_node || !tr.cursor || V < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(trie_select_kernel, dim3(B), dim3(TRIE_NT), 0, s, logits, ldl, V, ids, ld_ids, cur_len, plen, eos, tr,
                       part_val, part_idx, part_lse);
    return hipGetLastError();
}

hipError_t launch_search_step(const SearchState& st, int src, int cur_len, const StepCands& in, const EmbedArgs& em,
                              bool t_is_f32, hipStream_t s) {
    if (st.k > SS_KMAX || in.slots < 1 || in.slots > SS_CMAX || in.nparts < 1 || in.nparts > 256) return hipErrorInvalidValue;
    if (st.k * st.pn > SS_CMAX) return hipErrorInvalidValue;
    if (em.words && (em.D > 1024 || (em.D & 3))) return hipErrorInvalidValue;
#define GITMI_SSTEP(SL)                                                                                               \
    do {                                                                                                                \
        if (t_is_f32) hipLaunchKernelGGL((search_step_kernel<float, SL>), dim3(st.B), dim3(256), 0, s, st, src, cur_len, in, em); \
        else hipLaunchKernelGGL((search_step_kernel<bf16_t, SL>), dim3(st.B), dim3(256), 0, s, st, src, cur_len, in, em);          \
    } while (0)
    switch (in.slots) {
        case 1: GITMI_SSTEP(1); break;
        case 2: GITMI_SSTEP(2); break;
        case 4: GITMI_SSTEP(4); break;
        case 8: GITMI_SSTEP(8); break;
        case 16: GITMI_SSTEP(16); break;
        default: return hipErrorInvalidValue;       // list lengths are 1, 2, 4, 8, 16 (vocab_mtop_slots / row_topm_slots) or pn
    }
#undef GITMI_SSTEP
    return hipGetLastError();
}
hipError_t launch_search_init(const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_init_kernel, dim3(16), dim3(256), 0, s, st);
    return hipGetLastError();
}
hipError_t launch_search_finish(const SearchState& st, int cur, int cur_len, long long* tokens_out,
                                float* logprob_out, int* info_out, int* sent_out, hipStream_t s) {
    hipLaunchKernelGGL(search_finish_kernel, dim3(1), dim3(256), 0, s, st, cur, cur_len, tokens_out, logprob_out,
                       info_out, sent_out);
    return hipGetLastError();
}
hipError_t launch_search_rows(const SearchState& st, int cur, int cur_len, long long* out, hipStream_t s) {
    hipLaunchKernelGGL(search_rows_kernel, dim3(64), dim3(256), 0, s, st, cur, cur_len, out);
    return hipGetLastError();
}

}  // namespace gitmi
