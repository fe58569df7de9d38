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

// ---- a BATCH of decoded images of any sizes in one staging buffer -> [n, 3, crop, crop] ---------------------------------------
// At serving rates (~10k images/s per GPU) the per-image form costs the host more than the device: two launches, two
// allocations and one upload per image on the submitting thread.  Here the decoded images of a batch lie behind each other in ONE
// buffer (one upload), and the two passes run as ONE launch pair per chunk of PRE_CHUNK images: the per-image geometry and
// coefficient tables travel in the kernel arguments (no descriptor upload, nothing whose lifetime the caller must manage).
// Same arithmetic, instruction for instruction, as the per-image kernels above.
namespace {
constexpr int PRE_CHUNK = 24;
struct ImgDesc {
    const int* kk_h; const int* b_h; const int* kk_v; const int* b_v;
    unsigned int src_off, tmp_off;                     // bytes into the staging buffer / the workspace
    unsigned short H, W, nw, top, left, ks_h, ks_v, flags;   // flags: 1 = no horizontal pass, 2 = no vertical pass
};
struct ImgChunk { ImgDesc d[PRE_CHUNK]; };

__global__ void resize_h_batch_kernel(const uint8_t* __restrict__ base, uint8_t* __restrict__ tmp, ImgChunk c) {
    const ImgDesc& d = c.d[blockIdx.z];
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if ((d.flags & 1) || xo >= d.nw || y >= d.H) return;
    const int xmin = d.b_h[2 * xo], xmax = d.b_h[2 * xo + 1];
    const int* k = d.kk_h + (size_t)xo * d.ks_h;
    const uint8_t* row = base + d.src_off + ((size_t)y * d.W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int cf = k[x];
        s0 += row[3 * x] * cf;
        s1 += row[3 * x + 1] * cf;
        s2 += row[3 * x + 2] * cf;
    }
    uint8_t* o = tmp + d.tmp_off + ((size_t)y * d.nw + xo) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

__global__ void resize_v_crop_norm_batch_kernel(const uint8_t* __restrict__ base, const uint8_t* __restrict__ tmp,
                                                float* __restrict__ out, int crop, ImgChunk c, float m0, float m1, float m2,
                                                float d0, float d1, float d2) {
    const ImgDesc& d = c.d[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= crop) return;
    const uint8_t* in = (d.flags & 1) ? base + d.src_off : tmp + d.tmp_off;      // Pillow skips a pass whose size does not change
    const int yo = y + d.top, xi = x + d.left, W = d.nw;
    int p0, p1, p2;
    if (d.flags & 2) {
        const uint8_t* px = in + ((size_t)yo * W + xi) * 3;
        p0 = px[0]; p1 = px[1]; p2 = px[2];
    } else {
        const int ymin = d.b_v[2 * yo], ymax = d.b_v[2 * yo + 1];
        const int* k = d.kk_v + (size_t)yo * d.ks_v;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < ymax; ++t) {
            const uint8_t* px = in + ((size_t)(ymin + t) * W + xi) * 3;
            const int cf = k[t];
            s0 += px[0] * cf;
            s1 += px[1] * cf;
            s2 += px[2] * cf;
        }
        p0 = clip8(s0); p1 = clip8(s1); p2 = clip8(s2);
    }
    const size_t plane = (size_t)crop * crop, o = (size_t)blockIdx.z * 3 * plane + (size_t)y * crop + x;
    out[o] = ((float)p0 / 255.0f - m0) / d0;
    out[plane + o] = ((float)p1 / 255.0f - m1) / d1;
    out[2 * plane + o] = ((float)p2 / 255.0f - m2) / d2;
}
}  // namespace

// workspace bytes the horizontal passes of a batch need: sum of H_i * nw_i * 3 (64-byte aligned each)
size_t preprocess_batch_workspace(const long long* desc, int n, int crop) {
    size_t t = 0;
    for (int i = 0; i < n; ++i) {
        const int H = (int)desc[3 * i + 1], W = (int)desc[3 * i + 2];
        const int nw = W <= H ? crop : (int)((double)crop * W / H);
        if (nw != W) t += ((size_t)H * nw * 3 + 63) / 64 * 64;
    }
    return t;
}

// desc: HOST int64 [n][3] = (byte offset of image i in `rgb`, H, W).  out: fp32 [n, 3, crop, crop].
hipError_t launch_preprocess_batch(const uint8_t* rgb, const long long* desc, int n, int crop, uint8_t* tmp, float* out,
                                   hipStream_t s) {
    auto pyround = [](double v) { return (int)std::nearbyint(v); };
    size_t tmp_off = 0;
    for (int c0 = 0; c0 < n; c0 += PRE_CHUNK) {
        const int nc = n - c0 < PRE_CHUNK ? n - c0 : PRE_CHUNK;
        ImgChunk ch;
        int maxH = 1, maxnw = 1;
        bool any_h = false;
        for (int i = 0; i < nc; ++i) {
            const long long off = desc[3 * (c0 + i)];
            const int H = (int)desc[3 * (c0 + i) + 1], W = (int)desc[3 * (c0 + i) + 2];
            int nw, nh;
            if (W <= H) { nw = crop; nh = (int)((double)crop * H / W); }
            else { nw = (int)((double)crop * W / H); nh = crop; }
            ImgDesc& d = ch.d[i];
            d = ImgDesc{};
            d.src_off = (unsigned int)off;
            d.H = (unsigned short)H; d.W = (unsigned short)W; d.nw = (unsigned short)nw;
            d.left = (unsigned short)pyround((nw - crop) / 2.0);
            d.top = (unsigned short)pyround((nh - crop) / 2.0);
            if (nw != W) {
                Coeffs k;
                hipError_t e = get_coeffs(W, nw, &k);
                if (e != hipSuccess) return e;
                d.kk_h = k.kk; d.b_h = k.bounds; d.ks_h = (unsigned short)k.ksize;
                d.tmp_off = (unsigned int)tmp_off;
                tmp_off += ((size_t)H * nw * 3 + 63) / 64 * 64;
                any_h = true;
                if (H > maxH) maxH = H;
                if (nw > maxnw) maxnw = nw;
            } else d.flags |= 1;
            if (nh != H) {
                Coeffs k;
                hipError_t e = get_coeffs(H, nh, &k);
                if (e != hipSuccess) return e;
                d.kk_v = k.kk; d.b_v = k.bounds; d.ks_v = (unsigned short)k.ksize;
            } else d.flags |= 2;
        }
        if (any_h)
            hipLaunchKernelGGL(resize_h_batch_kernel, dim3((maxnw + 127) / 128, maxH, nc), dim3(128), 0, s, rgb, tmp, ch);
        hipLaunchKernelGGL(resize_v_crop_norm_batch_kernel, dim3((crop + 127) / 128, crop, nc), dim3(128), 0, s, rgb, tmp,
                           out + (size_t)c0 * 3 * crop * crop, crop, ch, 0.48145466f, 0.4578275f, 0.40821073f, 0.26862954f,
                           0.26130258f, 0.27577711f);
    }
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
