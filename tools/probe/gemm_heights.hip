// Standalone A/B of the encoder GEMM's tile heights (256 / 224 / 192 / 160 / 128 rows) on the shapes of the three BASELINE
// models: every height must produce the bits of the 256-row tile (same K order per output element); event timings, 20
// launches each, interleaved over 3 rounds (median).  tools/probe/build.sh builds it; no Python, no torch.
#define GITMI_PROBE 1
#include "kernels_gemm10.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned short f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

struct Shape { const char* name; int M, N, K; int stream; int act; };   // stream: f16 residual-stream output + residual

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::vector<Shape> shapes = {
        {"base.qkv", 12608, 2304, 768, 0, 0}, {"base.out", 12608, 768, 768, 1, 0}, {"base.c_fc", 12608, 3072, 768, 0, 1},
        {"base.c_proj", 12608, 768, 3072, 1, 0},
        {"large.qkv", 8224, 3072, 1024, 0, 0}, {"large.out", 8224, 1024, 1024, 1, 0}, {"large.c_fc", 8224, 4096, 1024, 0, 1},
        {"large.c_proj", 8224, 1024, 4096, 1, 0},
        {"vatex.qkv", 18912, 2304, 768, 0, 0}, {"vatex.out", 18912, 768, 768, 1, 0}, {"vatex.c_fc", 18912, 3072, 768, 0, 1},
        {"vatex.c_proj", 18912, 768, 3072, 1, 0},
        {"short_k", 12608, 2304, 256, 0, 0}, {"ragged", 5000, 2304, 384, 0, 2}};
    const char* only = getenv("PROBE_SHAPES");
    const int heights[5] = {256, 224, 192, 160, 128};
    const int bits[5] = {128, 32768, 64, 16384, 65536};
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        if (only && !strstr(only, sh.name)) continue;
        const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        std::vector<unsigned short> hA(na), hW(nw), hR(nc);
        for (auto& v : hA) v = f2bf_host(nd(rng));
        const float ws = 1.0f / sqrtf((float)sh.K);
        for (auto& v : hW) v = f2bf_host(nd(rng) * ws);
        for (auto& v : hR) { _Float16 h = (_Float16)nd(rng); memcpy(&v, &h, 2); }
        std::vector<float> hb(sh.N);
        for (auto& v : hb) v = nd(rng);
        unsigned short *dA, *dW, *dC, *dR, *dRef; float* db;
        CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dW, nw * 2)); CK(hipMalloc(&dC, nc * 2)); CK(hipMalloc(&dR, nc * 2)); CK(hipMalloc(&dRef, nc * 2));
        CK(hipMalloc(&db, sh.N * 4));
        CK(hipMemcpy(dA, hA.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, hR.data(), nc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        gitmi::GemmArgs g{};
        g.A = dA; g.W = dW; g.bias = db; g.C = dC; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldc = sh.N; g.ldr = sh.N;
        g.act = sh.act; g.shared = 1; g.out_f16 = sh.stream; g.res = sh.stream ? reinterpret_cast<const float*>(dR) : nullptr;
        printf("== %-12s M=%d N=%d K=%d %s\n", sh.name, sh.M, sh.N, sh.K, sh.stream ? "f16 stream out + residual" : (sh.act ? "bf16 out, QuickGELU" : "bf16 out"));
        std::vector<unsigned short> ref(nc), got(nc);
        std::vector<double> best(5, 1e30);
        std::vector<std::vector<double>> runs(5);
        for (int h = 0; h < 5; ++h) {
            g.dbg = bits[h]; g.C = h == 0 ? dRef : dC;
            CK(hipMemset(g.C, 0xff, nc * 2));
            CK(gitmi::launch_gemm_p8(g, false, 0));
            CK(hipDeviceSynchronize());
            if (h == 0) CK(hipMemcpy(ref.data(), dRef, nc * 2, hipMemcpyDeviceToHost));
            else {
                CK(hipMemcpy(got.data(), dC, nc * 2, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < nc; ++i) bad += got[i] != ref[i];
                if (bad) printf("   !! height %d: %zu of %zu outputs differ from the 256-row tile\n", heights[h], bad, nc);
            }
        }
        g.C = dC;
        for (int round = 0; round < 3; ++round)
            for (int h = 0; h < 5; ++h) {
                g.dbg = bits[h];
                for (int i = 0; i < 2; ++i) CK(gitmi::launch_gemm_p8(g, false, 0));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i) CK(gitmi::launch_gemm_p8(g, false, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                runs[h].push_back(ms * 1e3 / 20);
            }
        const double flops = 2.0 * sh.M * sh.N * sh.K;
        for (int h = 0; h < 5; ++h) {
            std::sort(runs[h].begin(), runs[h].end());
            const double us = runs[h][1];
            gitmi::GemmArgs q = g; q.dbg = 0;
            const int tiles = ((sh.M + heights[h] - 1) / heights[h]) * (sh.N / 256);
            printf("   %3d rows: %4d tiles  %7.1f us  %6.1f TFLOP/s  %.3f of 2.5 PF   (model %6.1f us)\n", heights[h], tiles, us, flops / us / 1e6,
                   flops / us / 1e6 / 2500.0, gitmi::gemm_p8_cost(q, heights[h]) * 1e-3);
        }
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dR)); CK(hipFree(dRef)); CK(hipFree(db));
    }
    return 0;
}
