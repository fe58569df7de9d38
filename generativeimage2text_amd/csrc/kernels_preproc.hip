// GPU-side image transform of the inference tasks (reference inference.py:111-132):
//   Resize(crop, BICUBIC) -> CenterCrop(crop) -> ToTensor -> Normalize(CLIP mean/std)
// The reference runs it on PIL images, i.e. the resize is Pillow's 8-bit resampler
// (libImaging/Resample.c): separable, horizontal pass first, coefficients in 8.22 fixed point, every
// pass rounded and clipped to uint8.  This file reproduces that arithmetic bit for bit:
//   * the coefficient tables are computed on the host in double precision with Pillow's formulas
//     (precompute_coeffs / normalize_coeffs_8bpc) and cached on the device per (in, out) size;
//   * resize_h_kernel: one thread per output pixel of the horizontal pass (3 channels);
//   * resize_v_crop_norm_kernel: vertical pass, evaluated only inside the centre crop, fused with
//     /255, -mean, /std (IEEE fp32 division, same operation order as torchvision) and the HWC->CHW transpose.
// At ~8k images/s per GPU the host-side PIL resize (milliseconds per image) is the end-to-end
// bottleneck of the reference pipeline; JPEG decoding stays on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace gitmi {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in: [H, W_in, 3] -> out: [H, W_out, 3]
__global__ void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W_in, int W_out,
                                const int* __restrict__ kk, const int* __restrict__ bounds, int ksize) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (xo >= W_out) return;
    const int xmin = bounds[2 * xo], xmax = bounds[2 * xo + 1];
    const int* k = kk + (size_t)xo * ksize;
    const uint8_t* row = in + ((size_t)y * W_in + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int c = k[x];
        s0 += row[3 * x] * c;
        s1 += row[3 * x + 1] * c;
        s2 += row[3 * x + 2] * c;
    }
    uint8_t* o = out + ((size_t)y * W_out + xo) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// in: [H_in, W, 3] (after the horizontal pass) -> out fp32 [3, ch, cw] for the crop window at (top, left)
__global__ void resize_v_crop_norm_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int H_in, int W,
                                          int ch, int cw, int top, int left, const int* __restrict__ kk,
                                          const int* __restrict__ bounds, int ksize, int identity_v, float m0, float m1,
                                          float m2, float d0, float d1, float d2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cw) return;
    const int yo = y + top, xi = x + left;
    int p0, p1, p2;
    if (identity_v) {                               // Pillow skips a pass whose size does not change
        const uint8_t* px = in + ((size_t)yo * W + xi) * 3;
        p0 = px[0]; p1 = px[1]; p2 = px[2];
    } else {
        const int ymin = bounds[2 * yo], ymax = bounds[2 * yo + 1];
        const int* k = kk + (size_t)yo * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < ymax; ++t) {
            const uint8_t* px = in + ((size_t)(ymin + t) * W + xi) * 3;
            const int c = k[t];
            s0 += px[0] * c;
            s1 += px[1] * c;
            s2 += px[2] * c;
        }
        p0 = clip8(s0); p1 = clip8(s1); p2 = clip8(s2);
    }
    const size_t plane = (size_t)ch * cw, o = (size_t)y * cw + x;
    out[o] = ((float)p0 / 255.0f - m0) / d0;
    out[plane + o] = ((float)p1 / 255.0f - m1) / d1;
    out[2 * plane + o] = ((float)p2 / 255.0f - m2) / d2;
}

double bicubic(double x) {
    const double a = -0.5;
    if (x < 0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Coeffs { int* kk = nullptr; int* bounds = nullptr; int ksize = 0; };
std::mutex g_mu;
std::map<std::pair<int, int>, Coeffs> g_cache;

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the full input range, BICUBIC (support 2)
hipError_t get_coeffs(int in_size, int out_size, Coeffs* out) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find({in_size, out_size});
    if (it != g_cache.end()) { *out = it->second; return hipSuccess; }
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    std::vector<int> kk((size_t)out_size * ksize, 0), bounds((size_t)out_size * 2);
    std::vector<double> pre(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic((x + xmin - center + 0.5) * ss);
            pre[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) pre[x] /= ww;
            const double v = pre[x] * (double)(1 << PRECISION_BITS);
            kk[(size_t)xx * ksize + x] = pre[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    Coeffs c;
    c.ksize = ksize;
    hipError_t e = hipMalloc((void**)&c.kk, kk.size() * sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMalloc((void**)&c.bounds, bounds.size() * sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemcpy(c.kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    e = hipMemcpy(c.bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    g_cache[{in_size, out_size}] = c;
    *out = c;
    return hipSuccess;
}

}  // namespace

// Pillow resize of rgb uint8 [H, W, 3] to nh x nw (BICUBIC), crop window (top, left, ch, cw) of the result,
// ToTensor + Normalize -> out fp32 [3, ch, cw].  tmp: uint8 workspace of at least H * nw * 3 bytes (unused if nw == W).
hipError_t launch_resize_crop_norm(const uint8_t* rgb, int H, int W, int nh, int nw, int top, int left, int ch, int cw,
                                   uint8_t* tmp, float* out, hipStream_t s) {
    const uint8_t* hsrc = rgb;
    if (nw != W) {
        Coeffs chz;
        hipError_t e = get_coeffs(W, nw, &chz);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(resize_h_kernel, dim3((nw + 127) / 128, H), dim3(128), 0, s, rgb, tmp, H, W, nw, chz.kk, chz.bounds,
                           chz.ksize);
        hsrc = tmp;
    }
    Coeffs cv;
    const int identity_v = nh == H;
    if (!identity_v) {
        hipError_t e = get_coeffs(H, nh, &cv);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(resize_v_crop_norm_kernel, dim3((cw + 127) / 128, ch), dim3(128), 0, s, hsrc, out, H, nw, ch, cw, top,
                       left, cv.kk, cv.bounds, cv.ksize, identity_v, 0.48145466f, 0.4578275f, 0.40821073f, 0.26862954f,
                       0.26130258f, 0.27577711f);
    return hipGetLastError();
}

// Resize(crop, BICUBIC) -> CenterCrop(crop) -> ToTensor -> Normalize (inference.py:118-131)
hipError_t launch_preprocess(const uint8_t* rgb, int H, int W, int crop, uint8_t* tmp, float* out, hipStream_t s) {
    // torchvision Resize(int): shorter side -> crop, the other int(crop * long / short)
    int nw, nh;
    if (W <= H) { nw = crop; nh = (int)((double)crop * H / W); }
    else { nw = (int)((double)crop * W / H); nh = crop; }
    // CenterCrop offsets: int(round((size - crop) / 2.0)) with Python's round-half-even
    auto pyround = [](double v) { return (int)std::nearbyint(v); };
    const int left = pyround((nw - crop) / 2.0), top = pyround((nh - crop) / 2.0);
    return launch_resize_crop_norm(rgb, H, W, nh, nw, top, left, crop, crop, tmp, out, s);
}

}  // namespace gitmi
