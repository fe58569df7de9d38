// Standalone timing decomposition of the fused vocabulary head (kernels_dgemm.hip: vocab_topm_kernel) in its beam-search form:
// 256 rows (64 sentences x 4 beams) x 30 522 columns x K = 768, top-8 lists per (row, 128-column block), LayerNorm folded.
// Variants are compile-time ablations (ABL): what the launch costs without the list insertion, the log-sum-exp, the 4-lane
// merge, the per-row-block activation re-read, the MFMAs.  Random operands; results of the ablated variants are wrong by design.
#include "kernels_dgemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned short f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

template <int MTOP, int MODE, int ABL>
static void run(const char* label, gitmi::VocabArgs g, int nwg, hipEvent_t e0, hipEvent_t e1) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gitmi::vocab_topm_kernel<MTOP, MODE, ABL>), dim3(nwg), dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gitmi::vocab_topm_kernel<MTOP, MODE, ABL>), dim3(nwg), dim3(256), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("   %-58s %3d WGs  %7.1f us\n", label, nwg, ms * 1e3 / 20);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int V = 30522, K = 768, Vp = (V + 127) / 128 * 128, nblk = Vp / 128;
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned short> hW((size_t)Vp * K), hA((size_t)256 * K);
    for (auto& v : hW) v = f2bf_host(nd(rng) * 0.02f);
    for (auto& v : hA) v = f2bf_host(nd(rng));
    std::vector<float> hb(Vp), hc(Vp);
    for (auto& v : hb) v = nd(rng) * 0.01f;
    for (auto& v : hc) v = nd(rng) * 0.1f;
    std::vector<float2> hs((size_t)48 * 256, float2{0.f, 16.f});
    unsigned short *dW, *dA; float *db, *dc, *dval; int* didx; float2 *ds, *dlse;
    CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&db, Vp * 4)); CK(hipMalloc(&dc, Vp * 4));
    CK(hipMalloc(&ds, hs.size() * 8)); CK(hipMalloc(&dval, (size_t)256 * nblk * 8 * 4)); CK(hipMalloc(&didx, (size_t)256 * nblk * 8 * 4));
    CK(hipMalloc(&dlse, (size_t)256 * nblk * 8));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Vp * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), Vp * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    gitmi::VocabArgs g{};
    g.A = dA; g.lda = K; g.W = dW; g.bias = db; g.colsum = dc; g.stats_in = ds; g.strips_in = 48; g.inv_d = 1.f / K; g.eps_in = 1e-12f;
    g.N = V; g.K = K; g.cols_per_wg = 128; g.beams = 4; g.part_val = dval; g.part_idx = didx; g.part_lse = dlse; g.nblk = nblk;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    using namespace gitmi;
    for (int nwg : {239, 60}) {
        printf("== beam search form: 256 rows, top-8 per (row, column block), %d workgroups (%s)\n", nwg, nwg == 239 ? "one column block each: a context alone" : "each walks four column blocks: serving policy");
        g.M = 256;
        run<8, VOC_BEAM, 0>("full", g, nwg, e0, e1);
        run<8, VOC_BEAM, 1>("no top-8 insertion (a running max instead)", g, nwg, e0, e1);
        run<8, VOC_BEAM, 2>("no log-sum-exp", g, nwg, e0, e1);
        run<8, VOC_BEAM, 4>("one merge round instead of eight", g, nwg, e0, e1);
        run<8, VOC_BEAM, 7>("no insertion, no log-sum-exp, one merge round", g, nwg, e0, e1);
        run<8, VOC_BEAM, 8>("activations read once, not per row block", g, nwg, e0, e1);
        run<8, VOC_BEAM, 15>("neither", g, nwg, e0, e1);
        run<8, VOC_BEAM, 31>("and no MFMAs (loads, LDS exchange, barriers)", g, nwg, e0, e1);
        run<2, VOC_BEAM, 0>("full with top-2 lists", g, nwg, e0, e1);
        run<1, VOC_BEAM, 0>("full with top-1 lists", g, nwg, e0, e1);
        g.M = 64;
        run<1, VOC_GREEDY, 0>("greedy form, 64 rows, top-1", g, nwg, e0, e1);
    }
    return 0;
}
