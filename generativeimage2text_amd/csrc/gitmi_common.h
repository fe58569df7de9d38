// Device-side helpers shared by all gfx950 kernels of the GIT engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define GITMI_WAVE 64

// ---- the 16-bit OPERAND type of the fast engine mode.  Default: bfloat16 (the benchmarked mode).  With -DGITMI_OPS_F16
// the very same kernels are built for IEEE fp16 operands (libgitmi_f16.so: `v_mfma_f32_16x16x32_f16` runs at the bf16
// rate and carries 3 more mantissa bits; every operand of this path -- LayerNorm outputs, attention probabilities,
// weights of N(0, 0.02) / width^-0.5 scale -- is far inside fp16's range).  `bf16_t` stays the name of the raw 16-bit
// storage type in both builds; pack2bf / f2bf / bf2f / unpack2op convert to and from the OPERAND encoding of the build.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#ifdef GITMI_OPS_F16
#define GITMI_OPERAND_NAME "f16"
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    const _Float16 b = (_Float16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack2op(uint32_t u, float& lo, float& hi) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, u);
    lo = (float)v[0];
    hi = (float)v[1];
}
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
#else
#define GITMI_OPERAND_NAME "bf16"
// bf16 <-> f32: gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN-safe); a (__bf16) cast is
// what makes hipcc emit it -- no branches, one instruction per pair
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack2op(uint32_t u, float& lo, float& hi) {
    lo = __uint_as_float(u << 16);
    hi = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif

// ---- fp16: storage type of the residual stream in bf16 engine mode (GITMI_STREAM_F16): half the bytes of fp32 at
// 2^-11 relative rounding -- tools/residual_precision_study.py: +0.002 max feature error, a bf16 stream costs 3x
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack2h(uint32_t u, float& lo, float& hi) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, u);
    lo = (float)v[0];
    hi = (float)v[1];
}
// 4 consecutive elements of a residual-stream row (16-byte aligned fp32 / 8-byte aligned fp16)
__device__ __forceinline__ f32x4_t ld4s(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ f32x4_t ld4s(const f16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    float a, b, c, d;
    unpack2h(u.x, a, b);
    unpack2h(u.y, c, d);
    return f32x4_t{a, b, c, d};
}
__device__ __forceinline__ void st4s(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void st4s(f16_t* p, f32x4_t v) {
    uint2 u;
    u.x = pack2h(v[0], v[1]);
    u.y = pack2h(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}

// ---- typed element access: T is float (exact path) or bf16_t (fast path) ----------
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ float ld<f16_t>(const f16_t* p) { return (float)*p; }
template <> __device__ __forceinline__ void st<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }

// load 8 consecutive elements as floats (16-byte aligned for bf16, 32 for float)
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    u32x4_t r = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2op(r[i], v[2 * i], v[2 * i + 1]);
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
    f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    u32x4_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2bf(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4_t*>(p) = r;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    f32x4_t a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4_t*>(p) = a;
    *reinterpret_cast<f32x4_t*>(p + 4) = b;
}

// ---- 64-lane wave reductions -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- fragment-major operand layout of the decode chain (kernels_dgemm.hip) -------------------------------------
// A [rows, K] bf16 matrix is stored as 16-row x 32-k tiles, each tile in MFMA 16x16x32 operand order: lane = (k%32)/8*16 + row%16
// holds 8 consecutive k.  A wave's fragment load is then ONE contiguous 1-KiB read (8 full cache lines) instead of
// 16 rows x 64 B gathered from 16 different lines.  ksteps = K / 32.
__device__ __host__ __forceinline__ size_t frag_offset(int row, int k, int ksteps) {
    return (((size_t)(row >> 4) * ksteps + (k >> 5)) * 64 + (size_t)(((k & 31) >> 3) * 16 + (row & 15))) * 8 + (k & 7);
}
// start of tile (row tile rt, k-step ks) for a given lane
__device__ __forceinline__ size_t frag_tile(int rt, int ks, int ksteps, int lane) {
    return (((size_t)rt * ksteps + ks) * 64 + lane) * 8;
}

// ---- repetition penalty of GeneratorWithBeamSearch.search (decoder.py:1135-1144): the raw score of every token that is
// already in a row's history is multiplied (score < 0) or divided (score >= 0) by the penalty before the log-softmax.
__device__ __forceinline__ float rep_penalize(float x, float rp) { return x < 0.f ? x * rp : x / rp; }
// the history of a row as an open-addressing set in LDS (<= HIST_SLOTS / 2 tokens; empty slot = -1)
constexpr int HIST_SLOTS = 2048;
__device__ __forceinline__ unsigned int hist_hash(int tok) { return ((unsigned int)tok * 2654435761u) >> 21; }   // 11 bits
__device__ __forceinline__ void hist_insert(int* tbl, int tok) {
    unsigned int h = hist_hash(tok);
    for (;;) {
        const int old = atomicCAS(&tbl[h], -1, tok);
        if (old == -1 || old == tok) return;
        h = (h + 1) & (HIST_SLOTS - 1);
    }
}
__device__ __forceinline__ bool hist_contains(const int* tbl, int tok) {
    unsigned int h = hist_hash(tok);
    for (;;) {
        const int v = tbl[h];
        if (v == tok) return true;
        if (v == -1) return false;
        h = (h + 1) & (HIST_SLOTS - 1);
    }
}

// activation codes shared with the host
#define GITMI_ACT_NONE 0
#define GITMI_ACT_QUICKGELU 1   // x * sigmoid(1.702 x)            CLIP/model.py:171-173
#define GITMI_ACT_GELU_ERF 2    // 0.5 x (1 + erf(x / sqrt 2))     bert/activations.py:15-22

// v_exp_f32 / v_rcp_f32 directly (1 ulp class): expf()/division expand to ~20 instructions with branches
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float quick_gelu(float x) { return x * fast_rcp(1.0f + fast_exp(-1.702f * x)); }

// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), branch-free:
//   erf(z) = sign(z) * (1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-z^2)),  t = 1 / (1 + p |z|)
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = fast_rcp(1.0f + 0.3275911f * az);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    poly *= t;
    const float e = 1.0f - poly * fast_exp(-az * az);
    return 0.5f * x * (1.0f + copysignf(e, z));
}

template <int ACT> __device__ __forceinline__ float apply_act_t(float x) {
    if constexpr (ACT == GITMI_ACT_QUICKGELU) return quick_gelu(x);
    else if constexpr (ACT == GITMI_ACT_GELU_ERF) return gelu_erf(x);
    else return x;
}
__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == GITMI_ACT_QUICKGELU) return quick_gelu(x);
    if (act == GITMI_ACT_GELU_ERF) return gelu_erf(x);
    return x;
}
