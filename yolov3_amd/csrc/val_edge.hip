// Input and output edges of the validation / detection loop on gfx950.  Input: letterbox (below).
// Output: what the reference does per image in Python right after non_max_suppression --
//   * scale_boxes + clip_boxes   (reference utils/general.py:613-626, upstream ultralytics.utils.ops.clip_boxes; callers val.py:397,403
//     and detect.py:223): undo the letterbox gain / padding and clamp to the native image, and
//   * process_batch              (reference val.py:147-188, upstream ultralytics.utils.metrics.box_iou): the (detections x IoU
//     thresholds) "correct" matrix behind mAP --
// for the whole batch in one launch each, reading the batched NMS output (bs, max_det, 6) + counts where it lies.
// fp32 arithmetic in the reference's operation order (built with -ffp-contract=off; IEEE division as torch's CPU kernels).
#include "y3_common.h"

namespace {

// rows[img][r][0..3] = clip((xyxy - pad) / gain); params[img] = {gain, pad_x, pad_y, w0, h0}
__global__ __launch_bounds__(256) void scale_boxes_kernel(float* __restrict__ rows, long long img_stride, int row_stride, const int* __restrict__ counts, int bs, int max_rows,
                                                            const float* __restrict__ params) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)bs * max_rows) return;
    const int img = (int)(idx / max_rows), r = (int)(idx - (long long)img * max_rows);
    if (counts && r >= counts[img]) return;
    const float gain = params[img * 5], px = params[img * 5 + 1], py = params[img * 5 + 2], w0 = params[img * 5 + 3], h0 = params[img * 5 + 4];
    float* b = rows + img * img_stride + (long long)r * row_stride;
    // boxes[..., [0, 2]] -= pad[0]; boxes[..., [1, 3]] -= pad[1]; boxes[..., :4] /= gain; clamp_(0, w0 | h0)
    const float x1 = (b[0] - px) / gain, y1 = (b[1] - py) / gain, x2 = (b[2] - px) / gain, y2 = (b[3] - py) / gain;
    b[0] = fminf(fmaxf(x1, 0.0f), w0);
    b[1] = fminf(fmaxf(y1, 0.0f), h0);
    b[2] = fminf(fmaxf(x2, 0.0f), w0);
    b[3] = fminf(fmaxf(y2, 0.0f), h0);
}

// upstream box_iou(labels, detections)[l][d]: inter / (area_l + area_d - inter + eps), eps = 1e-7
Y3_DEV float pair_iou(const float* lb, const float* dt) {
    const float iw = fmaxf(fminf(lb[2], dt[2]) - fmaxf(lb[0], dt[0]), 0.0f);
    const float ih = fmaxf(fminf(lb[3], dt[3]) - fmaxf(lb[1], dt[1]), 0.0f);
    const float inter = iw * ih;
    const float a1 = (lb[2] - lb[0]) * (lb[3] - lb[1]), a2 = (dt[2] - dt[0]) * (dt[3] - dt[1]);
    return inter / (a1 + a2 - inter + 1e-7f);
}

// ---- input edge: letterbox ---------------------------------------------------------------------------------------------
// reference utils/augmentations.py:104-134 with auto=False (the call of models/common.py:866): cv2.resize(INTER_LINEAR) to
// new_unpad, cv2.copyMakeBorder(constant 114) to (H1, W1), then the BHWC -> BCHW transpose of :867 -- one pass, one thread per
// output pixel.  cv2 is an un-vendored dependency (absent here: "parity unpinned"): the 8-bit INTER_LINEAR path is restated
// from OpenCV's resize.cpp (HResizeLinear / VResizeLinear<uchar, int, short>): source coordinate f = (d + 0.5) * scale - 0.5
// in double, cast to float; coefficients saturate_cast<short>(w * 2048) (round half to even); horizontal taps in int32, vertical
// blend ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; columns left of 0 / right of w0 - 1 take the border pixel
// with weight 2048, rows are clamped with their fractional weights kept.
struct LetterboxArgs {
    const unsigned char* src;
    unsigned char* dst;      // image `index` of the (n, 3, H1, W1) batch
    int h0, w0, cs, H1, W1, nh, nw, top, left, color;
    double scale_x, scale_y; // 1 / (nw / w0), 1 / (nh / h0) as cv2 computes them
};
Y3_DEV void lb_coef(int d, double scale, int size, bool clamp_frac, int& s, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp_frac) {   // x axis: outside columns read one pixel with full weight
        if (s < 0) { f = 0.0f; s = 0; }
        if (s >= size - 1) { f = 0.0f; s = size - 1; }
    }
    c0 = (int)(short)__float2int_rn((1.0f - f) * 2048.0f);
    c1 = (int)(short)__float2int_rn(f * 2048.0f);
}
__global__ __launch_bounds__(256) void letterbox_u8_kernel(const LetterboxArgs p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.W1 || y >= p.H1) return;
    const long long plane = (long long)p.H1 * p.W1, o = (long long)y * p.W1 + x;
    const int dx = x - p.left, dy = y - p.top;
    if (dx < 0 || dx >= p.nw || dy < 0 || dy >= p.nh) {
        for (int c = 0; c < 3; ++c) p.dst[c * plane + o] = (unsigned char)p.color;
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    lb_coef(dx, p.scale_x, p.w0, true, sx, a0, a1);
    lb_coef(dy, p.scale_y, p.h0, false, sy, b0, b1);
    const int sx1 = sx + 1 < p.w0 ? sx + 1 : sx;   // weight 0 whenever it would fall outside
    const int r0 = sy < 0 ? 0 : (sy > p.h0 - 1 ? p.h0 - 1 : sy), r1 = sy + 1 < 0 ? 0 : (sy + 1 > p.h0 - 1 ? p.h0 - 1 : sy + 1);
    const unsigned char* q0 = p.src + ((long long)r0 * p.w0) * p.cs;
    const unsigned char* q1 = p.src + ((long long)r1 * p.w0) * p.cs;
    for (int c = 0; c < 3; ++c) {
        const int h0v = (int)q0[sx * p.cs + c] * a0 + (int)q0[sx1 * p.cs + c] * a1;
        const int h1v = (int)q1[sx * p.cs + c] * a0 + (int)q1[sx1 * p.cs + c] * a1;
        const int v = (((b0 * (h0v >> 4)) >> 16) + ((b1 * (h1v >> 4)) >> 16) + 2) >> 2;
        p.dst[c * plane + o] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// One block per image.  The reference, per threshold t: pairs (label, detection) with IoU >= t and equal class, sorted by IoU
// descending; every detection keeps its best label (first np.unique), then every label keeps the LOWEST-INDEX detection among
// those (second np.unique on the detection-ordered list; the re-sort by IoU in between is commented out, val.py:185).
// A detection's best label does not depend on t (it is the class-matched label of maximal IoU, valid while that IoU >= t), so:
//   correct[d][t] = biou[d] >= t  and no d' < d with the same best label has biou[d'] >= t.
// Exact IoU ties between two labels of one detection resolve to the larger label index (numpy's reversed argsort for <= 16
// candidate pairs; undefined in the reference beyond that).
constexpr int MATCH_CAP = 4096;
__global__ __launch_bounds__(256) void match_detections_kernel(const float* __restrict__ dets, long long img_stride, int row_stride, const int* __restrict__ counts, int max_det,
                                                                 const float* __restrict__ labels, const int* __restrict__ offs, const float* __restrict__ iouv, int niou,
                                                                 unsigned char* __restrict__ correct) {
    __shared__ int bl[MATCH_CAP];
    __shared__ float biou[MATCH_CAP];
    const int img = blockIdx.x;
    int n = counts ? counts[img] : max_det;
    if (n > max_det) n = max_det;
    const int l0 = offs[img], l1 = offs[img + 1];
    const float* D = dets + img * img_stride;
    for (int d = threadIdx.x; d < n; d += 256) {
        const float* dt = D + (long long)d * row_stride;
        const float cls = dt[5];
        int best = -1;
        float bv = -1.0f;
        for (int l = l0; l < l1; ++l) {
            const float* lb = labels + (long long)l * 5;
            if (lb[0] != cls) continue;
            const float v = pair_iou(lb + 1, dt);
            if (v >= bv) { bv = v; best = l; }
        }
        bl[d] = best;
        biou[d] = bv;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < max_det; d += 256) {
        unsigned char* out = correct + ((long long)img * max_det + d) * niou;
        if (d >= n || bl[d] < 0) {
            for (int t = 0; t < niou; ++t) out[t] = 0;
            continue;
        }
        const int b = bl[d];
        const float v = biou[d];
        float earlier = -1.0f;   // best IoU among lower-index detections that chose the same label
        for (int e = 0; e < d; ++e)
            if (bl[e] == b) earlier = fmaxf(earlier, biou[e]);
        for (int t = 0; t < niou; ++t) {
            const float thr = iouv[t];
            out[t] = (unsigned char)((v >= thr) && !(earlier >= thr));
        }
    }
}

// ---- test-time augmentation edges (reference models/yolo.py:239-276 _forward_augment) ---------------------------------------------
// scale_img (upstream ultralytics.utils.torch_utils.scale_img, un-vendored; restated in oracle/upstream.py): the batch, optionally mirrored left-right
// (`x.flip(3)`, models/yolo.py:246), resized with F.interpolate(mode="bilinear", align_corners=False) to (ih, iw) = (int(h ratio), int(w ratio)) and padded
// on the right / bottom with 0.447 to the next multiple of the largest stride.  Source index of output o: max(0, fma(in / out, o + 0.5, -0.5)), fp32 weights,
// the four products summed in torch's order (row pairs first), one rounding to T at the end.  NCHW in, NCHW out (the model's ingest kernel reads NCHW).
struct ScaleImgArgs {
    const void* src;
    void* dst;
    int planes, h, w, ih, iw, oh, ow, flip;
    float rh, rw, pad;
};
template <typename T> __global__ __launch_bounds__(256) void scale_img_kernel(const ScaleImgArgs a) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= a.ow || oy >= a.oh) return;
    const bool inside = ox < a.iw && oy < a.ih;
    int y0 = 0, x0 = 0, y1 = 0, x1 = 0;
    float ly = 0.f, lx = 0.f;
    if (inside) {
        // scale * (o + 0.5) - 0.5 as ONE fused multiply-add, like torch's kernels (the file is built with -ffp-contract=off; two roundings move the
        // weights by an ulp of the index: 1.7e-6 on the goldens instead of 1.2e-7)
        const float sy = fmaxf(fmaf(a.rh, (float)oy + 0.5f, -0.5f), 0.0f), sx = fmaxf(fmaf(a.rw, (float)ox + 0.5f, -0.5f), 0.0f);
        y0 = (int)sy; x0 = (int)sx;
        y1 = y0 + (y0 < a.h - 1 ? 1 : 0);
        x1 = x0 + (x0 < a.w - 1 ? 1 : 0);
        ly = sy - (float)y0; lx = sx - (float)x0;
        if (a.flip) { x0 = a.w - 1 - x0; x1 = a.w - 1 - x1; }   // the mirrored image's column j is column w - 1 - j of the source
    }
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const T* src = (const T*)a.src;
    T* dst = (T*)a.dst;
    for (int pl = blockIdx.z; pl < a.planes; pl += gridDim.z) {
        float v = a.pad;
        if (inside) {
            const T* s = src + (long long)pl * a.h * a.w;
            const float v00 = to_f32<T>(s[(long long)y0 * a.w + x0]), v01 = to_f32<T>(s[(long long)y0 * a.w + x1]);
            const float v10 = to_f32<T>(s[(long long)y1 * a.w + x0]), v11 = to_f32<T>(s[(long long)y1 * a.w + x1]);
            v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        }
        dst[((long long)pl * a.oh + oy) * a.ow + ox] = from_f32<T>(v);
    }
}

// _descale_pred + the row selection of _clip_augmented (models/yolo.py:253-276) in one pass: rows [row0, row0 + nrows) of every image of one scale's
// decoded prediction (bs, src_rows, no) land at rows [dst_row0, ...) of the concatenated (bs, dst_rows, no) result with xywh / scale and, for a mirrored
// pass, x = img_w - x (flip 3) or y = img_h - y (flip 2); every step rounded to T like the in-place tensor ops of the reference.
template <typename T> __global__ __launch_bounds__(256) void descale_pred_kernel(const T* __restrict__ src, int src_rows, int no, int row0, int nrows, float scale, int flip, float img_h,
                                                                                   float img_w, T* __restrict__ dst, int dst_rows, int dst_row0, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % no);
    const long long rr = idx / no;
    const int r = (int)(rr % nrows), img = (int)(rr / nrows);
    float v = to_f32<T>(src[((long long)img * src_rows + row0 + r) * no + col]);
    if (col < 4) {
        v = rt<T>(v / scale);
        if (flip == 3 && col == 0) v = rt<T>(img_w - v);
        if (flip == 2 && col == 1) v = rt<T>(img_h - v);
    }
    dst[((long long)img * dst_rows + dst_row0 + r) * no + col] = from_f32<T>(v);
}

}  // namespace

extern "C" int y3_scale_boxes(float* rows, int64_t img_stride, int32_t row_stride, const int32_t* counts, int32_t bs, int32_t max_rows, const float* params, void* stream) {
    if (!rows || !params) Y3_FAIL("y3_scale_boxes: null argument");
    if (bs < 0 || max_rows < 0 || row_stride < 4) Y3_FAIL("y3_scale_boxes: bad geometry (bs %d, rows %d, row stride %d)", bs, max_rows, row_stride);
    const long long total = (long long)bs * max_rows;
    if (total == 0) return 0;
    hipLaunchKernelGGL(scale_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, (long long)img_stride, row_stride, counts, bs, max_rows, params);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_match_detections(const float* dets, int64_t img_stride, int32_t row_stride, const int32_t* counts, int32_t bs, int32_t max_det, const float* labels,
                                   const int32_t* label_offsets, const float* iouv, int32_t niou, uint8_t* correct, void* stream) {
    if (!dets || !label_offsets || !iouv || !correct) Y3_FAIL("y3_match_detections: null argument");
    if (bs < 0 || niou < 1 || row_stride < 6) Y3_FAIL("y3_match_detections: bad geometry (bs %d, niou %d, row stride %d)", bs, niou, row_stride);
    if (max_det < 0 || max_det > MATCH_CAP) Y3_FAIL("y3_match_detections: max_det %d unsupported (max %d)", max_det, MATCH_CAP);
    if (bs == 0 || max_det == 0) return 0;
    hipLaunchKernelGGL(match_detections_kernel, dim3((unsigned)bs), dim3(256), 0, (hipStream_t)stream, dets, (long long)img_stride, row_stride, counts, max_det, labels, label_offsets, iouv,
                       niou, correct);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_letterbox_u8(const uint8_t* src, int32_t h0, int32_t w0, int32_t cs, uint8_t* dst_batch, int32_t index, int32_t H1, int32_t W1, int32_t new_h, int32_t new_w,
                               int32_t top, int32_t left, int32_t color, void* stream) {
    if (!src || !dst_batch) Y3_FAIL("y3_letterbox_u8: null argument");
    if (h0 < 1 || w0 < 1 || cs < 3 || H1 < 1 || W1 < 1 || new_h < 1 || new_w < 1 || index < 0) Y3_FAIL("y3_letterbox_u8: bad geometry");
    if (top < 0 || left < 0 || top + new_h > H1 || left + new_w > W1) Y3_FAIL("y3_letterbox_u8: the resized image (%dx%d at %d,%d) does not fit %dx%d", new_w, new_h, left, top, W1, H1);
    LetterboxArgs a;
    a.src = src;
    a.dst = dst_batch + (size_t)index * 3 * H1 * W1;
    a.h0 = h0; a.w0 = w0; a.cs = cs; a.H1 = H1; a.W1 = W1; a.nh = new_h; a.nw = new_w; a.top = top; a.left = left; a.color = color;
    a.scale_x = 1.0 / ((double)new_w / (double)w0);
    a.scale_y = 1.0 / ((double)new_h / (double)h0);
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3((unsigned)((W1 + 63) / 64), (unsigned)((H1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_scale_img(const void* src, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t flip_lr, float pad_value,
                            void* dst, void* stream) {
    if (!src || !dst) Y3_FAIL("y3_scale_img: null argument");
    if (n < 0 || c < 1 || h < 1 || w < 1 || ih < 1 || iw < 1 || oh < ih || ow < iw) Y3_FAIL("y3_scale_img: bad geometry (%dx%dx%dx%d -> %dx%d in %dx%d)", n, c, h, w, ih, iw, oh, ow);
    if ((long long)n * c > 0x7fffffffLL) Y3_FAIL("y3_scale_img: too many planes");
    if (n == 0) return 0;
    ScaleImgArgs a;
    a.src = src; a.dst = dst; a.planes = n * c; a.h = h; a.w = w; a.ih = ih; a.iw = iw; a.oh = oh; a.ow = ow; a.flip = flip_lr ? 1 : 0;
    a.rh = (float)h / (float)ih;   // torch's area_pixel_compute_scale without a scale factor: input size / output size
    a.rw = (float)w / (float)iw;
    a.pad = pad_value;
    const dim3 grid((unsigned)((ow + 63) / 64), (unsigned)((oh + 3) / 4), (unsigned)(a.planes < 64 ? a.planes : 64));
    switch (dtype) {
        case Y3_F32: hipLaunchKernelGGL(scale_img_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
        case Y3_F16: hipLaunchKernelGGL(scale_img_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
        case Y3_BF16: hipLaunchKernelGGL(scale_img_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
        default: Y3_FAIL("y3_scale_img: unsupported dtype %d", dtype);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_descale_pred(const void* src, int32_t dtype, int32_t bs, int32_t src_rows, int32_t no, int32_t row0, int32_t nrows, float scale, int32_t flip, float img_h, float img_w,
                               void* dst, int32_t dst_rows, int32_t dst_row0, void* stream) {
    if (!src || !dst) Y3_FAIL("y3_descale_pred: null argument");
    if (bs < 0 || no < 5 || row0 < 0 || nrows < 0 || row0 + nrows > src_rows || dst_row0 < 0 || dst_row0 + nrows > dst_rows) Y3_FAIL("y3_descale_pred: bad geometry");
    if (!(scale > 0.0f) || (flip != 0 && flip != 2 && flip != 3)) Y3_FAIL("y3_descale_pred: scale %g / flip %d unsupported (flip: 0 none, 2 up-down, 3 left-right)", (double)scale, flip);
    const long long total = (long long)bs * nrows * no;
    if (total == 0) return 0;
    const dim3 grid((unsigned)((total + 255) / 256));
    switch (dtype) {
        case Y3_F32: hipLaunchKernelGGL(descale_pred_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, src_rows, no, row0, nrows, scale, flip, img_h, img_w, (float*)dst, dst_rows, dst_row0, total); break;
        case Y3_F16: hipLaunchKernelGGL(descale_pred_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, src_rows, no, row0, nrows, scale, flip, img_h, img_w, (_Float16*)dst, dst_rows, dst_row0, total); break;
        case Y3_BF16: hipLaunchKernelGGL(descale_pred_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_rows, no, row0, nrows, scale, flip, img_h, img_w, (bf16_t*)dst, dst_rows, dst_row0, total); break;
        default: Y3_FAIL("y3_descale_pred: unsupported dtype %d", dtype);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}
