// Which MFMA shape draws less power per FLOP on random operands?  Register-resident MFMA loops, no memory traffic in the loop:
// 256 CUs x 8 waves, each wave 128 accumulator registers (the encoder GEMM's budget), operands N(0,1) bf16 or all zero.
// Reports TFLOP/s and the clock (s_memtime / s_memrealtime) for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// SHAPE 0: 32 accumulator tiles of 16x16 (4 regs each), 8 A x 4 B fragments per round = 32 MFMAs of 16x16x32 (8192 MACs each)
// SHAPE 1: 8 accumulator tiles of 32x32 (16 regs each), 4 A x 2 B fragments per round = 8 MFMAs of 32x32x16 (16384 MACs each... x2 rounds)
template <int SHAPE, int ORDER = 0>
__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, uint64_t* __restrict__ ts, int iters) {
    const int tid = threadIdx.x;
    bf16x8_t a[8], b[4];
    const uint4* p = src + ((size_t)blockIdx.x * 512 + tid) * 12;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint4 v = p[i]; a[i] = __builtin_bit_cast(bf16x8_t, v); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint4 v = p[8 + i]; b[i] = __builtin_bit_cast(bf16x8_t, v); }
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if constexpr (ORDER == 0) {          // second operand stationary over 4 MFMAs, first changes every MFMA
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            } else if constexpr (ORDER == 1) {   // first operand stationary over 8 MFMAs (the encoder GEMM's order: weights stationary over the row fragments)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            } else {                             // both operands change with every MFMA
#pragma unroll
                for (int d = 0; d < 8; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[(d + j) & 7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[(d + j) & 7], acc[(d + j) & 7][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h)        // two k-halves: the same operand bytes as SHAPE 0 per iteration (8 A + 4 B registers x 4)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2 * j + h], a[2 * i + h], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (sum == 12345.678f) out[0] = sum;
    if (tid == 0) { ts[blockIdx.x * 2] = c1 - c0; ts[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nwg = 256, iters = 4000;
    const size_t n16 = (size_t)nwg * 512 * 12;
    std::vector<uint4> h(n16);
    uint4* d; float* dout; uint64_t* dts;
    CK(hipMalloc(&d, n16 * 16)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dts, nwg * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int fill = 0; fill < 2; ++fill) {
        unsigned short* hs = reinterpret_cast<unsigned short*>(h.data());
        for (size_t i = 0; i < n16 * 8; ++i) {
            float f = fill ? 0.f : nd(rng) * 0.05f;
            uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); hs[i] = (unsigned short)(u >> 16);
        }
        CK(hipMemcpy(d, h.data(), n16 * 16, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep)
            for (int shape = 0; shape < 4; ++shape) {
                auto launch = [&]() {
                    if (shape == 0) hipLaunchKernelGGL((mfma_loop<0, 0>), dim3(nwg), dim3(512), 0, 0, d, dout, dts, iters);
                    else if (shape == 1) hipLaunchKernelGGL((mfma_loop<1, 0>), dim3(nwg), dim3(512), 0, 0, d, dout, dts, iters);
                    else if (shape == 2) hipLaunchKernelGGL((mfma_loop<0, 1>), dim3(nwg), dim3(512), 0, 0, d, dout, dts, iters);
                    else hipLaunchKernelGGL((mfma_loop<0, 2>), dim3(nwg), dim3(512), 0, 0, d, dout, dts, iters);
                };
                for (int w = 0; w < 2; ++w) launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int w = 0; w < 5; ++w) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<uint64_t> t(nwg * 2);
                CK(hipMemcpy(t.data(), dts, nwg * 16, hipMemcpyDeviceToHost));
                double cyc = 0, rt = 0;
                for (int b = 0; b < nwg; ++b) { cyc += (double)t[2 * b]; rt += (double)t[2 * b + 1] * 10.0; }
                const double flops = 2.0 * nwg * 8 * (double)iters * 32 * 8192.0;        // per launch: both shapes 32 x 8192 MACs per wave-iteration
                printf("%-9s operands  %-26s %8.1f TFLOP/s   clock %.3f GHz   %.1f cycles per 8192-MAC unit per SIMD\n", fill ? "ALL-ZERO" : "N(0,.05)",
                       shape == 1 ? "32x32x16" : shape == 0 ? "16x16x32 B stationary x4" : shape == 2 ? "16x16x32 A stationary x8" : "16x16x32 both change", flops / (ms / 5 * 1e-3) / 1e12, cyc / rt,
                       cyc / nwg / ((double)iters * 32) / 2.0);
            }
    }
    return 0;
}
