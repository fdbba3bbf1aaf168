// HBM-bound layout / pooling kernels (gfx950): image ingest, max-pool, SPP pyramid, upsample, slice copy.
// All are pure streaming kernels: 16-byte vector accesses along the NHWC channel axis, grid-stride-free
// (one thread per 16-byte chunk), nothing staged in LDS because there is no cross-thread reuse that the
// L2 does not already capture (pool windows overlap along W/H inside one XCD's L2 working set).
#include "y3_common.h"

#include <stdlib.h>

namespace {

template <typename T> struct Vec {  // 16 bytes of T
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

template <typename T> Y3_DEV T lowest();
template <> Y3_DEV f16_t lowest<f16_t>() { return (f16_t)(-65504.0f); }
template <> Y3_DEV bf16_t lowest<bf16_t>() { return (bf16_t)(-3.38e38f); }
template <> Y3_DEV float lowest<float>() { return -3.402823466e38f; }

template <typename TI> Y3_DEV float load_as_f32(const TI* p) { return to_f32<TI>(*p); }
template <> Y3_DEV float load_as_f32<unsigned char>(const unsigned char* p) { return (float)*p; }

// NCHW TI -> NHWC TO with channel zero-padding; one thread per output pixel (reads are coalesced along W
// per channel plane, the write is one or more 16-byte stores per pixel).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const TI* __restrict__ src, int n, int c, int h, int w, float scale,
                                                             TO* __restrict__ dst, int cpad, int pitch) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long hw = (long long)h * w;
    if (idx >= (long long)n * hw) return;
    const int b = (int)(idx / hw);
    const long long pix = idx - (long long)b * hw;
    TO* o = dst + idx * pitch;
    const TI* s = src + (long long)b * c * hw + pix;
    for (int c0 = 0; c0 < cpad; c0 += Vec<TO>::N) {
        Vec<TO> out;
#pragma unroll
        for (int q = 0; q < Vec<TO>::N; ++q) {
            const int ci = c0 + q;
            // torch: x.to(dtype) first, then true-divide in that dtype (val.py:358-359); 1 -> plain cast
            float v = ci < c ? load_as_f32<TI>(s + (long long)ci * hw) : 0.0f;
            if (scale != 1.0f) v = rt<TO>(v) / scale;  // `scale` is the divisor
            out.v[q] = from_f32<TO>(v);
        }
        *(Vec<TO>*)(o + c0) = out;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, int n, int h, int w, int c, int pitch, T* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // over NCHW output, w fastest
    const long long total = (long long)n * c * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w);
    long long t = idx / w;
    const int y = (int)(t % h);
    t /= h;
    const int ci = (int)(t % c);
    const int b = (int)(t / c);
    dst[idx] = src[((long long)(b * h + y) * w + x) * pitch + ci];
}

// MaxPool2d(k, s, p) (+ optional zero-pad right/bottom first).  One thread per (output pixel, 16-byte chunk).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, T* __restrict__ y, int Ho,
                                                        int Wo, int ypitch, int k, int s, int pad, int zr, int zb) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * Ho * Wo * cv;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * V;
    long long t = idx / cv;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[V];
#pragma unroll
    for (int q = 0; q < V; ++q) m[q] = -INFINITY;
    const int Hz = H + zb, Wz = W + zr;  // zero-padded extent
    for (int kh = 0; kh < k; ++kh) {
        const int hi = ho * s - pad + kh;
        if (hi < 0 || hi >= Hz) continue;
        for (int kw = 0; kw < k; ++kw) {
            const int wi = wo * s - pad + kw;
            if (wi < 0 || wi >= Wz) continue;
            if (hi < H && wi < W) {
                const Vec<T> v = *(const Vec<T>*)(x + ((long long)(n * H + hi) * W + wi) * xpitch + c0);
#pragma unroll
                for (int q = 0; q < V; ++q) m[q] = fmaxf(m[q], to_f32<T>(v.v[q]));
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q) m[q] = fmaxf(m[q], 0.0f);  // explicit zero padding cell
            }
        }
    }
    Vec<T> o;
#pragma unroll
    for (int q = 0; q < V; ++q) o.v[q] = from_f32<T>(m[q]);
    *(Vec<T>*)(y + ((long long)(n * Ho + ho) * Wo + wo) * ypitch + c0) = o;
}

// SPP pyramid, k = 5 / 9 / 13, stride 1, "same" (-inf) padding, all three windows in one pass over the
// 13x13 neighbourhood (the 5- and 9-windows are its centred sub-windows), written to three channel slices.
// SPP pyramid through LDS: max-pool is separable and 5 -> 9 -> 13 cascades (pool5 of pool5 = pool9, pool5 of pool9 = pool13,
// with the -inf padding of nn.MaxPool2d reproduced by skipping out-of-image taps), so one block keeps an image's H x W plane of
// CG 16-byte channel groups in LDS and makes three (row max5, column max5) passes: 30 LDS reads per element instead of the 169
// global loads of the direct kernel below (0.28 ms for 32 x 20 x 20 x 512, 4 % of the yolov3-spp forward).
template <typename T>
__global__ __launch_bounds__(256) void spp_lds_kernel(const T* __restrict__ x, int H, int W, int C, int xpitch, T* __restrict__ y, int ypitch, int CG) {
    constexpr int V = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char spp_smem[];
    const int P = H * W;
    Vec<T>* A = (Vec<T>*)spp_smem;
    Vec<T>* B = A + (size_t)P * CG;
    const int groups = C / (V * CG);
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG, PL = 256 / CG;
    const int c0 = (g * CG + cg) * V;
    const T* xin = x + (long long)n * P * xpitch + c0;
    T* yout = y + (long long)n * P * ypitch + c0;
    if (pl < PL)
        for (int p = pl; p < P; p += PL) A[p * CG + cg] = *(const Vec<T>*)(xin + (long long)p * xpitch);
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        if (pl < PL)
            for (int p = pl; p < P; p += PL) {   // rows: B = max over w-2 .. w+2 of A
                const int h = p / W, w = p - h * W;
                const int lo = w - 2 < 0 ? 0 : w - 2, hi = w + 2 >= W ? W - 1 : w + 2;
                Vec<T> m = A[(h * W + lo) * CG + cg];
                for (int q = lo + 1; q <= hi; ++q) {
                    const Vec<T> v = A[(h * W + q) * CG + cg];
#pragma unroll
                    for (int e = 0; e < V; ++e) m.v[e] = to_f32<T>(v.v[e]) > to_f32<T>(m.v[e]) ? v.v[e] : m.v[e];
                }
                B[p * CG + cg] = m;
            }
        __syncthreads();
        if (pl < PL)
            for (int p = pl; p < P; p += PL) {   // columns: A = max over h-2 .. h+2 of B, and the stage's output slice
                const int h = p / W, w = p - h * W;
                const int lo = h - 2 < 0 ? 0 : h - 2, hi = h + 2 >= H ? H - 1 : h + 2;
                Vec<T> m = B[(lo * W + w) * CG + cg];
                for (int q = lo + 1; q <= hi; ++q) {
                    const Vec<T> v = B[(q * W + w) * CG + cg];
#pragma unroll
                    for (int e = 0; e < V; ++e) m.v[e] = to_f32<T>(v.v[e]) > to_f32<T>(m.v[e]) ? v.v[e] : m.v[e];
                }
                A[p * CG + cg] = m;
                *(Vec<T>*)(yout + (long long)p * ypitch + stage * C) = m;
            }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void spp_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, T* __restrict__ y, int ypitch) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * H * W * cv;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * V;
    long long t = idx / cv;
    const int wo = (int)(t % W);
    t /= W;
    const int ho = (int)(t % H);
    const int n = (int)(t / H);
    float m5[V], m9[V], m13[V];
#pragma unroll
    for (int q = 0; q < V; ++q) m5[q] = m9[q] = m13[q] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int hi = ho + dy;
        if (hi < 0 || hi >= H) continue;
        const int ay = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int wi = wo + dx;
            if (wi < 0 || wi >= W) continue;
            const int ax = dx < 0 ? -dx : dx;
            const int r = ay > ax ? ay : ax;
            const Vec<T> v = *(const Vec<T>*)(x + ((long long)(n * H + hi) * W + wi) * xpitch + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const float f = to_f32<T>(v.v[q]);
                m13[q] = fmaxf(m13[q], f);
                if (r <= 4) m9[q] = fmaxf(m9[q], f);
                if (r <= 2) m5[q] = fmaxf(m5[q], f);
            }
        }
    }
    Vec<T> o5, o9, o13;
#pragma unroll
    for (int q = 0; q < V; ++q) {
        o5.v[q] = from_f32<T>(m5[q]);
        o9.v[q] = from_f32<T>(m9[q]);
        o13.v[q] = from_f32<T>(m13[q]);
    }
    T* o = y + ((long long)(n * H + ho) * W + wo) * ypitch + c0;
    *(Vec<T>*)(o) = o5;
    *(Vec<T>*)(o + C) = o9;
    *(Vec<T>*)(o + 2 * C) = o13;
}

// y[n, ho, wo, :] = x[n, ho/up, wo/up, :]  (up = 1 -> slice copy, up = 2 -> nearest x2)
template <typename T>
__global__ __launch_bounds__(256) void resample_copy_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, T* __restrict__ y, int ypitch,
                                                              int up) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const int Ho = H * up, Wo = W * up;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * Ho * Wo * cv;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * V;
    long long t = idx / cv;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    *(Vec<T>*)(y + ((long long)(n * Ho + ho) * Wo + wo) * ypitch + c0) =
        *(const Vec<T>*)(x + ((long long)(n * H + ho / up) * W + wo / up) * xpitch + c0);
}

bool vec_ok(const y3_tensor* t, int esz) {
    const int v = 16 / esz;
    return (t->c % v) == 0 && (t->pitch % v) == 0 && (((uintptr_t)t->data) & 15) == 0;
}
int esize(int dtype) { return dtype == Y3_F32 ? 4 : 2; }
unsigned nblk(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

#define Y3_DISPATCH_FLOAT(dtype, EXPR)                                  \
    switch (dtype) {                                                    \
        case Y3_F16: { typedef f16_t T; EXPR; } break;                  \
        case Y3_BF16: { typedef bf16_t T; EXPR; } break;                \
        case Y3_F32: { typedef float T; EXPR; } break;                  \
        default: Y3_FAIL("bad dtype %d", (int)(dtype));                 \
    }

template <typename TI> static int ingest(const void* src, int n, int c, int h, int w, float scale, int out_dtype, const y3_tensor* out, hipStream_t st) {
    const long long total = (long long)n * h * w;
    Y3_DISPATCH_FLOAT(out_dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, T>), dim3(nblk(total)), dim3(256), 0, st, (const TI*)src, n, c, h, w, scale,
                                                    (T*)out->data, out->c, out->pitch));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_nchw_to_nhwc(const void* src, int32_t src_dtype, int32_t n, int32_t c, int32_t h, int32_t w, float scale, int32_t out_dtype,
                               const y3_tensor* out, void* stream) {
    if (!src || !out) Y3_FAIL("y3_nchw_to_nhwc: null argument");
    if (out->n != n || out->h != h || out->w != w || out->c < c) Y3_FAIL("y3_nchw_to_nhwc: output shape mismatch");
    if (!vec_ok(out, esize(out_dtype))) Y3_FAIL("y3_nchw_to_nhwc: output must be 16-byte aligned / channel-padded");
    hipStream_t st = (hipStream_t)stream;
    switch (src_dtype) {
        case Y3_U8: return ingest<unsigned char>(src, n, c, h, w, scale, out_dtype, out, st);
        case Y3_F16: return ingest<f16_t>(src, n, c, h, w, scale, out_dtype, out, st);
        case Y3_BF16: return ingest<bf16_t>(src, n, c, h, w, scale, out_dtype, out, st);
        case Y3_F32: return ingest<float>(src, n, c, h, w, scale, out_dtype, out, st);
    }
    Y3_FAIL("y3_nchw_to_nhwc: bad source dtype %d", src_dtype);
}

extern "C" int y3_nhwc_to_nchw(const y3_tensor* src, int32_t dtype, void* dst, void* stream) {
    if (!src || !dst) Y3_FAIL("y3_nhwc_to_nchw: null argument");
    const long long total = (long long)src->n * src->c * src->h * src->w;
    Y3_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src->data, src->n,
                                                src->h, src->w, src->c, src->pitch, (T*)dst));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_maxpool2d(const y3_tensor* x, const y3_tensor* y, int32_t dtype, int32_t k, int32_t stride, int32_t pad, int32_t zr, int32_t zb,
                            void* stream) {
    if (!x || !y) Y3_FAIL("y3_maxpool2d: null argument");
    const int Ho = (x->h + zb + 2 * pad - k) / stride + 1, Wo = (x->w + zr + 2 * pad - k) / stride + 1;
    if (y->n != x->n || y->h != Ho || y->w != Wo || y->c != x->c) Y3_FAIL("y3_maxpool2d: output is (%d,%d,%d,%d), expected (%d,%d,%d,%d)", y->n, y->h, y->w, y->c, x->n, Ho, Wo, x->c);
    if (!vec_ok(x, esize(dtype)) || !vec_ok(y, esize(dtype))) Y3_FAIL("y3_maxpool2d: alignment");
    const long long total = (long long)x->n * Ho * Wo * (x->c / (16 / esize(dtype)));
    Y3_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((maxpool_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x->data, x->n, x->h, x->w,
                                                x->c, x->pitch, (T*)y->data, Ho, Wo, y->pitch, k, stride, pad, zr, zb));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_spp_pyramid(const y3_tensor* x, const y3_tensor* y, int32_t dtype, void* stream) {
    if (!x || !y) Y3_FAIL("y3_spp_pyramid: null argument");
    if (y->n != x->n || y->h != x->h || y->w != x->w || y->c != 3 * x->c) Y3_FAIL("y3_spp_pyramid: output slice must be 3x the input channels");
    if (!vec_ok(x, esize(dtype)) || !vec_ok(y, esize(dtype))) Y3_FAIL("y3_spp_pyramid: alignment");
    {   // LDS path: two H x W planes of CG 16-byte channel groups must fit 64 KiB (CG = 4 -> 64-byte runs per pixel)
        const int vecs = x->c / (16 / esize(dtype));
        const long long P = (long long)x->h * x->w;
        int CG = 4;
        while (CG > 1 && (vecs % CG || 2 * P * CG * 16 > 65536)) CG >>= 1;
        const bool direct = y3_knob(Y3K_SPP_DIRECT) != 0;   // knob "spp_direct" (A/B: pools without the LDS pyramid)
        if (!direct && 2 * P * CG * 16 <= 65536 && (long long)x->n * (vecs / CG) < 0x7fffffffLL) {
            const size_t lds = (size_t)(2 * P * CG * 16);
            Y3_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((spp_lds_kernel<T>), dim3((unsigned)(x->n * (vecs / CG))), dim3(256), lds, (hipStream_t)stream, (const T*)x->data, x->h, x->w,
                                                        x->c, x->pitch, (T*)y->data, y->pitch, CG));
            Y3_CHECK_LAUNCH();
            return 0;
        }
    }
    const long long total = (long long)x->n * x->h * x->w * (x->c / (16 / esize(dtype)));
    Y3_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((spp_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x->data, x->n, x->h, x->w, x->c,
                                                x->pitch, (T*)y->data, y->pitch));
    Y3_CHECK_LAUNCH();
    return 0;
}

static int resample(const y3_tensor* x, const y3_tensor* y, int dtype, int up, void* stream, const char* who) {
    if (!x || !y) Y3_FAIL("%s: null argument", who);
    if (y->n != x->n || y->h != x->h * up || y->w != x->w * up || y->c != x->c) Y3_FAIL("%s: shape mismatch", who);
    if (!vec_ok(x, esize(dtype)) || !vec_ok(y, esize(dtype))) Y3_FAIL("%s: alignment", who);
    const long long total = (long long)y->n * y->h * y->w * (x->c / (16 / esize(dtype)));
    Y3_DISPATCH_FLOAT(dtype, hipLaunchKernelGGL((resample_copy_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x->data, x->n, x->h,
                                                x->w, x->c, x->pitch, (T*)y->data, y->pitch, up));
    Y3_CHECK_LAUNCH();
    return 0;
}
extern "C" int y3_upsample2x(const y3_tensor* x, const y3_tensor* y, int32_t dtype, void* stream) { return resample(x, y, dtype, 2, stream, "y3_upsample2x"); }
extern "C" int y3_copy_slice(const y3_tensor* x, const y3_tensor* y, int32_t dtype, void* stream) { return resample(x, y, dtype, 1, stream, "y3_copy_slice"); }
