// Device-side search: the whole bookkeeping of AutoRegressiveBeamSearch.search
// (decoder.py:224-440) and GeneratorWithBeamSearch.search + BeamHypotheses
// (decoder.py:1083-1341) runs in two small kernels per decode step, so the decode loop has
// no host<->device synchronisation at all (the reference does ~10^3 .item() syncs per step
// at B=64, k=4).
//
//   row_topm   : per row r of the beam batch: optional "no immediate repeat" (-10000 on the
//                last token's logit, decoder.py:330), optional forced-EOS rows
//                (decoder.py:347-351), log-softmax (online max / sum-exp) and the M best
//                (log-prob, token) pairs, sorted.  One block per row, one pass over the logits.
//   s1_advance : AutoRegressiveBeamSearch step for every image (top-k over k*per_node
//                candidates, beam gather) + "all beams ended" detection.
//   s2_advance : GeneratorWithBeamSearch step for every image: merge the k sorted candidate
//                lists into the top-2k of the flattened [k*V] axis, then the per-sentence loop
//                of decoder.py:1184-1222 (is_done, hypotheses, next beam, padding rule).
//   finish     : select outputs.
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
                                                       int eos, int suppress_last, int force_eos, int M,
                                                       float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    constexpr int NW = NT / 64;
    __shared__ float s_val[NT * MMAX];
    __shared__ int s_idx[NT * MMAX];
    __shared__ float s_red[2 * NW];
    __shared__ int s_redi[2 * NW];
    __shared__ int s_owner;

    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * ldl;
    const int last = ids[(size_t)r * ld_ids + cur_len - 1];
    float* cv = cand_val + (size_t)r * M;
    int* ci = cand_idx + (size_t)r * M;

    if (force_eos && last == eos) {
        // one-hot distribution on EOS (decoder.py:300-310, 347-351): log-prob 0, everything else -inf
        if (tid < M) {
            cv[tid] = tid == 0 ? 0.f : -INFINITY;
            ci[tid] = tid == 0 ? eos : (tid - 1 < eos ? tid - 1 : tid);
        }
        return;
    }

    float tv[MMAX];
    int ti[MMAX];
#pragma unroll
    for (int j = 0; j < MMAX; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    float mx = -INFINITY, sm = 0.f;
    auto feed = [&](float v, int i) {
        if (suppress_last && i == last) v = -10000.f;
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
    // block log-sum-exp
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
    const float lse = bmx + logf(tot);
    __syncthreads();

    // M rounds of block arg-max over the heads of the 256 sorted per-thread lists
    int head = 0;
    for (int round = 0; round < M; ++round) {
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
            cv[round] = bv - lse;
            ci[round] = bi == 0x7fffffff ? 0 : bi;
            s_owner = bw;
        }
        __syncthreads();
        if (tid == s_owner) ++head;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// AutoRegressiveBeamSearch.  One thread per image, single block.
//   first == 1 : decoder.py:257-298 (top-k of beam 0's distribution, all beams share the prefix)
//   first == 0 : decoder.py:313-417
// cur_len = number of tokens currently in ids (before appending).
// ---------------------------------------------------------------------------------------
__global__ void s1_advance_kernel(SearchState st, int src, int cur_len, int first, int M) {
    __shared__ int s_all_eos;
    const int dst = src ^ 1;
    const int R = st.B * st.k;
    if (threadIdx.x == 0) s_all_eos = 1;
    __syncthreads();
    if (!first) {
        // decoder.py:319: stop when every beam's last token is EOS (the step is idempotent after that)
        int ok = 1;
        for (int r = threadIdx.x; r < R; r += blockDim.x)
            if (st.ids[src][(size_t)r * st.T + cur_len - 1] != st.eos) ok = 0;
        if (!ok) s_all_eos = 0;
        __syncthreads();
        if (threadIdx.x == 0 && s_all_eos && st.info[0] == 0) st.info[0] = cur_len;
    }
    for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
        const int k = st.k, pn = st.pn;
        int chosen_src[16];
        if (first) {
            const int r0 = b * k;
            for (int j = 0; j < k; ++j) {
                const int r = r0 + j;
                for (int s = 0; s < cur_len; ++s) {
                    st.ids[dst][(size_t)r * st.T + s] = st.ids[src][(size_t)r0 * st.T + s];
                    st.kv_src[dst][(size_t)r * st.T + s] = st.kv_src[src][(size_t)r0 * st.T + s];
                }
                st.ids[dst][(size_t)r * st.T + cur_len] = st.cand_idx[(size_t)r0 * M + j];
                st.kv_src[dst][(size_t)r * st.T + cur_len] = r;
                st.score[dst][r] = st.cand_val[(size_t)r0 * M + j];
            }
        } else {
            // top-k (sorted) of the k*pn summed candidates, ties -> lower candidate index
            unsigned long long used = 0ull;
            for (int j = 0; j < k; ++j) {
                float best = 0.f; int bc = -1;
                for (int c = 0; c < k * pn; ++c) {
                    if (used >> c & 1ull) continue;
                    const int r = b * k + c / pn;
                    const float v = st.cand_val[(size_t)r * M + c % pn] + st.score[src][r];
                    if (bc < 0 || v > best) { best = v; bc = c; }
                }
                used |= 1ull << bc;
                const int rs = b * k + bc / pn;
                const int r = b * k + j;
                chosen_src[j] = rs;
                st.score[dst][r] = best;
                st.ids[dst][(size_t)r * st.T + cur_len] = st.cand_idx[(size_t)rs * M + bc % pn];
                st.kv_src[dst][(size_t)r * st.T + cur_len] = r;
            }
            for (int j = 0; j < k; ++j) {
                const int r = b * k + j, rs = chosen_src[j];
                for (int s = 0; s < cur_len; ++s) {
                    st.ids[dst][(size_t)r * st.T + s] = st.ids[src][(size_t)rs * st.T + s];
                    st.kv_src[dst][(size_t)r * st.T + s] = st.kv_src[src][(size_t)rs * st.T + s];
                }
            }
        }
    }
    if (first && st.k == 1) {
        // decoder.py:279-291: every first prediction is EOS -> early return
        __syncthreads();
        int ok = 1;
        for (int b = threadIdx.x; b < st.B; b += blockDim.x)
            if (st.cand_idx[(size_t)b * M] != st.eos) ok = 0;
        if (!ok) s_all_eos = 0;
        __syncthreads();
        if (threadIdx.x == 0 && s_all_eos) st.info[1] = 1;
    }
    if (threadIdx.x == 0) st.info[2] += 1;
}

// ---------------------------------------------------------------------------------------
// GeneratorWithBeamSearch step (decoder.py:1169-1232) + BeamHypotheses (1292-1341, n_hyp = 1).
// Host arithmetic of the reference is Python double -> double here.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double length_norm(int len, double alpha) {
    return pow(5.0 + (double)len, alpha) / pow(6.0, alpha);
}

__global__ void s2_advance_kernel(SearchState st, int src, int cur_len, int M) {
    const int dst = src ^ 1;
    for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
        const int k = st.k, V = st.V;
        const int ncand = st.pn * k;              // "2k"
        // merge: each row's list is sorted by log-prob; adding the row's beam score keeps the order
        int headp[16];
        for (int j = 0; j < k; ++j) headp[j] = 0;
        float n_score[32]; int n_beam[32], n_word[32];
        for (int c = 0; c < ncand; ++c) {
            float best = 0.f; int bj = -1; long long bflat = 0;
            for (int j = 0; j < k; ++j) {
                if (headp[j] >= M) continue;
                const int r = b * k + j;
                const float v = st.cand_val[(size_t)r * M + headp[j]] + st.score[src][r];
                const long long flat = (long long)j * V + st.cand_idx[(size_t)r * M + headp[j]];
                if (bj < 0 || v > best || (v == best && flat < bflat)) { best = v; bj = j; bflat = flat; }
            }
            n_score[c] = best; n_beam[c] = bj; n_word[c] = st.cand_idx[(size_t)(b * k + bj) * M + headp[bj]];
            headp[bj]++;
        }
        // ---- per-sentence loop ---------------------------------------------------------
        int is_done = st.done[b];
        if (!is_done && st.hyp_n[b] >= 1) {
            // BeamHypotheses.is_done(max next score); self.max_length = max_length - 1
            is_done = st.hyp_score[b] >= (double)n_score[0] / length_norm(st.T - 1, st.length_penalty);
        }
        st.done[b] = is_done;
        int nb = 0;
        int sel_src[16], sel_word[16]; float sel_score[16];
        if (!is_done) {
            for (int c = 0; c < ncand && nb < k; ++c) {
                const int word = n_word[c];
                if (word == st.eos || cur_len + 1 == st.T) {
                    // hyps.add(input_ids[row, :cur_len], score)
                    const double sc = (double)n_score[c] / length_norm(cur_len, st.length_penalty);
                    if (st.hyp_n[b] < 1 || sc > st.hyp_score[b]) {
                        st.hyp_n[b] = 1;
                        st.hyp_score[b] = sc;
                        st.hyp_len[b] = cur_len;
                        const int r = b * k + n_beam[c];
                        for (int s = 0; s < cur_len; ++s) st.hyp_tok[(size_t)b * st.T + s] = st.ids[src][(size_t)r * st.T + s];
                    }
                } else {
                    sel_score[nb] = n_score[c]; sel_word[nb] = word; sel_src[nb] = b * k + n_beam[c];
                    ++nb;
                }
            }
        }
        if (nb < k) {
            // done sentence, or every candidate finished (cur_len + 1 == max_length): pad the batch
            // with (score 0, EOS, global row 0)  -- decoder.py:1189, 1219-1220
            for (int j = 0; j < k; ++j) { sel_score[j] = 0.f; sel_word[j] = st.eos; sel_src[j] = 0; }
        }
        for (int j = 0; j < k; ++j) {
            const int r = b * k + j, rs = sel_src[j];
            for (int s = 0; s < cur_len; ++s) {
                st.ids[dst][(size_t)r * st.T + s] = st.ids[src][(size_t)rs * st.T + s];
                st.kv_src[dst][(size_t)r * st.T + s] = st.kv_src[src][(size_t)rs * st.T + s];
            }
            st.ids[dst][(size_t)r * st.T + cur_len] = sel_word[j];
            st.kv_src[dst][(size_t)r * st.T + cur_len] = r;
            st.score[dst][r] = sel_score[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int all = 1;
        for (int b = 0; b < st.B; ++b)
            if (!st.done[b]) all = 0;
        st.info[3] = all;          // decoder.py:1251: the host loop may stop polling here
        st.info[2] += 1;
    }
}

// ---------------------------------------------------------------------------------------
__global__ void search_init_kernel(SearchState st, const long long* __restrict__ start /*[B][P]*/) {
    const int R = st.B * st.k;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int b = r / st.k, j = r % st.k;
        for (int s = 0; s < st.T; ++s) {
            st.ids[0][(size_t)r * st.T + s] = s < st.P ? (int)start[(size_t)b * st.P + s] : st.eos;
            st.kv_src[0][(size_t)r * st.T + s] = r;
        }
        // decoder.py:1118-1120: only beam 0 is live at the start (GENERATOR); AUTOREGRESSIVE
        // takes its first step from beam 0 explicitly.
        st.score[0][r] = (st.kind == 1 && j > 0) ? -1e9f : 0.f;
    }
    for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
        st.done[b] = 0; st.hyp_n[b] = 0; st.hyp_score[b] = 0.0; st.hyp_len[b] = 0;
    }
    if (threadIdx.x < 4) st.info[threadIdx.x] = 0;
}

// AUTOREGRESSIVE: best beam (index 0), log-prob / num_valid (decoder.py:429-438)
// GENERATOR     : best hypothesis + EOS, EOS padded; -1e5 when none (decoder.py:1264-1290)
__global__ void search_finish_kernel(SearchState st, int cur, int cur_len, long long* __restrict__ tokens_out,
                                     float* __restrict__ logprob_out, int* __restrict__ info_out) {
    for (int b = threadIdx.x; b < st.B; b += blockDim.x) {
        long long* out = tokens_out + (size_t)b * st.T;
        if (st.kind == 0) {
            const int r = b * st.k;
            const int stop = st.info[0];
            const int L = stop > 0 ? stop : cur_len;          // length the reference returns
            int non_eos = 0, any_eos = 0;
            for (int s = 0; s < st.T; ++s) {
                const int tok = s < L ? st.ids[cur][(size_t)r * st.T + s] : st.eos;
                out[s] = tok;
                if (s < L) { if (tok != st.eos) ++non_eos; else any_eos = 1; }
            }
            if (st.info[1]) {
                logprob_out[b] = st.score[cur][r];              // early return: raw first log-prob
            } else {
                int nv = non_eos + any_eos - st.P;
                nv = nv < 1 ? 1 : nv;
                logprob_out[b] = st.score[cur][r] / (float)nv;
            }
        } else {
            const int n = st.hyp_n[b] > 0 ? st.hyp_len[b] : 0;
            for (int s = 0; s < st.T; ++s) out[s] = s < n ? st.hyp_tok[(size_t)b * st.T + s] : st.eos;
            logprob_out[b] = st.hyp_n[b] > 0 ? (float)st.hyp_score[b] : -1e5f;
        }
    }
    if (threadIdx.x == 0) {
        const int stop = st.info[0];
        info_out[0] = st.kind == 0 ? (stop > 0 ? stop : cur_len) : st.T;
        info_out[1] = st.info[1];
        info_out[2] = st.info[2];
        info_out[3] = 0;
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

// start tokens [B,P]: the shared prefix (device pointer) or [CLS] (decoder.py:979-989) -- no host round trip
__global__ void fill_start_kernel(long long* __restrict__ start, const long long* __restrict__ prefix, int sos, int B,
                                  int P) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * P; i += gridDim.x * blockDim.x)
        start[i] = prefix ? prefix[i % P] : (long long)sos;
}

// ---- host launchers ------------------------------------------------------------------
hipError_t launch_fill_start(long long* start, const long long* prefix, int sos, int B, int P, hipStream_t s) {
    hipLaunchKernelGGL(fill_start_kernel, dim3(8), dim3(256), 0, s, start, prefix, sos, B, P);
    return hipGetLastError();
}

hipError_t launch_load_ids(const long long* tokens, int R, int t, int* ids, int* kv_src, int ld, hipStream_t s) {
    hipLaunchKernelGGL(load_ids_kernel, dim3(64), dim3(256), 0, s, tokens, R, t, ids, kv_src, ld);
    return hipGetLastError();
}

hipError_t launch_row_topm(const float* logits, int ldl, int V, const int* ids, int ld_ids, int cur_len, int eos,
                           int suppress_last, int force_eos, int M, int R, float* cand_val, int* cand_idx,
                           hipStream_t s) {
    if (M < 1 || M > 16) return hipErrorInvalidValue;
    // one workgroup per row; more threads per row when the per-thread candidate list is short (LDS: NT*MMAX*8 B)
#define GITMI_TOPM(MM, NTT)                                                                                         \
    hipLaunchKernelGGL((row_topm_kernel<MM, NTT>), dim3(R), dim3(NTT), 0, s, logits, ldl, V, ids, ld_ids, cur_len, eos, \
                       suppress_last, force_eos, M, cand_val, cand_idx)
    if (M <= 1) GITMI_TOPM(1, 1024);
    else if (M <= 2) GITMI_TOPM(2, 1024);
    else if (M <= 4) GITMI_TOPM(4, 1024);
    else if (M <= 8) GITMI_TOPM(8, 512);
    else GITMI_TOPM(16, 256);
#undef GITMI_TOPM
    return hipGetLastError();
}

hipError_t launch_s1_advance(const SearchState& st, int src, int cur_len, int first, int M, hipStream_t s) {
    hipLaunchKernelGGL(s1_advance_kernel, dim3(1), dim3(256), 0, s, st, src, cur_len, first, M);
    return hipGetLastError();
}
hipError_t launch_s2_advance(const SearchState& st, int src, int cur_len, int M, hipStream_t s) {
    hipLaunchKernelGGL(s2_advance_kernel, dim3(1), dim3(256), 0, s, st, src, cur_len, M);
    return hipGetLastError();
}
hipError_t launch_search_init(const SearchState& st, const long long* start_dev, hipStream_t s) {
    hipLaunchKernelGGL(search_init_kernel, dim3(1), dim3(256), 0, s, st, start_dev);
    return hipGetLastError();
}
hipError_t launch_search_finish(const SearchState& st, int cur, int cur_len, long long* tokens_out,
                                float* logprob_out, int* info_out, hipStream_t s) {
    hipLaunchKernelGGL(search_finish_kernel, dim3(1), dim3(256), 0, s, st, cur, cur_len, tokens_out, logprob_out,
                       info_out);
    return hipGetLastError();
}
hipError_t launch_search_rows(const SearchState& st, int cur, int cur_len, long long* out, hipStream_t s) {
    hipLaunchKernelGGL(search_rows_kernel, dim3(64), dim3(256), 0, s, st, cur, cur_len, out);
    return hipGetLastError();
}

}  // namespace gitmi
