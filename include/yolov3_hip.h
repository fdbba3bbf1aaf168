/* yolov3_hip.h -- C ABI of libyolov3_hip.so (MI355X / gfx950 only).
 *
 * The reference (ultralytics/yolov3) is 100 % Python and has no FFI of its own:
 * every op on its detection hot path is an ATen call.  This header is the boundary a maintainer
 * binds instead of those ATen calls; each entry point names the reference call site it replaces.
 * The Python host side (yolov3_amd/) mirrors the reference's own signatures
 * (DetectionModel/Detect.forward, non_max_suppression, ComputeLoss) on top of these symbols.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (the host side
 *    allocates through torch's caching allocator so it is stream-ordered); no hidden hipMalloc,
 *    no host synchronisation inside any call; `stream` is a hipStream_t passed as void*.
 *  - activations are NHWC ("pixel-major"): element (n,h,w,c) at data[((n*H+h)*W+w)*pitch + c];
 *    `pitch` >= c lets a tensor be a channel slice of a wider buffer (zero-copy Concat).
 *  - return 0 on success, negative on error; y3_last_error() gives the thread-local message.
 */
#ifndef YOLOV3_HIP_H
#define YOLOV3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3_ABI_VERSION 5

typedef enum { Y3_F16 = 0, Y3_BF16 = 1, Y3_F32 = 2, Y3_U8 = 3 } y3_dtype;
typedef enum { Y3_ACT_NONE = 0, Y3_ACT_SILU = 1 } y3_act;
typedef enum { Y3_ALGO_AUTO = 0, Y3_ALGO_MFMA = 1, Y3_ALGO_DIRECT = 2 } y3_algo;

/* NHWC view. `data` already points at channel 0 of the slice. */
typedef struct {
    void* data;
    int32_t n, h, w, c;
    int32_t pitch; /* elements between consecutive pixels (>= c) */
} y3_tensor;

/* Fused Conv2d(k in {1,3}, stride in {1,2}, pad=k/2, groups=1) + bias + activation (+ residual).
 * Replaces reference models/common.py:75 / :81 (Conv.forward / forward_fuse: conv -> bn -> SiLU),
 * the residual add of models/common.py:165 (Bottleneck.forward), the nn.Upsample(x2, nearest) of
 * models/yolov3.yaml:43,51 (upsample2x != 0: every output pixel is written to its 2x2 block of the
 * (2*Ho, 2*Wo) destination) and torch.cat of models/common.py:428 (write into a channel slice). */
typedef struct {
    int32_t dtype;      /* y3_dtype of x / w / residual / y (F16, BF16, F32) */
    int32_t ksize;      /* 1 or 3 */
    int32_t stride;     /* 1 or 2 */
    int32_t act;        /* y3_act */
    int32_t upsample2x; /* 0 / 1 */
    int32_t algo;       /* y3_algo; AUTO = MFMA for F16/BF16, DIRECT for F32 */
    int32_t cin;        /* logical input channels of the packed filter (x->c must equal it) */
    int32_t cout;       /* output channels stored (multiple of 8; pad filters with zero rows) */
    int32_t in_dilation; /* 0/1 = plain; 2 = x is a virtual zero-interleaved (2h x 2w) image: data-gradient of a
                            stride-2 conv = stride-1 conv of the dilated output gradient with the flipped, transposed
                            filter (y3_pack_filter_dgrad); y's height/width then give the gradient's size */
    int64_t filter_elems; /* elements of `dtype` the caller's packed bank holds (ABI 4).  Non-zero: a bank shorter than y3_packed_filter_elems(cout, cin, ksize)
                             is refused -- e.g. one sized rows x Kpad by the ABI-1 rule, without the fragment-ordered second copy the persistent 3x3 kernel
                             reads behind the row-major one.  0 = unchecked: the caller vouches for the size */
} y3_conv_desc;

int y3_abi_version(void);
const char* y3_last_error(void);

/* Run-time tuning knobs: every A/B hook of the library in one table (no counterpart in the reference; ATen's equivalents are
 * torch.backends.cudnn.benchmark and friends).  Defaults are the measured-best settings.  Keys: "conv" (0 per-shape dispatch, 2 / 4 / 5 /
 * 6 / 15 force a tile variant), "conv_ahead" (3 / 2: K-steps the LDS-DMA requests run ahead), "bn_nt_bytes" (threshold of
 * the non-temporal BatchNorm passes), "wgrad" (0 per-shape, 2 128-tile, 3 256-tile, 4 direct fp32), "wgrad_xcd" (0..3), "dgrad_quad"
 * (1 / 0), "spp_direct" (0 / 1), "conv_v10" (1 auto, 0 off, 2 every eligible shape), "v10_mp" / "v10_blocks" (0 auto; widest wave tile / blocks per filter tile of csrc/conv_v10.h: tests), "v10_half" (2 auto, 0 one block per CU, 1 two half-size blocks per CU), "v10_ksplit" (1 small launches with a workspace, 0 never, 2 every eligible launch) / "v10_slices" (0 auto; forced slice count: tests), "v10_group" (1 the blocks of a filter tile on one XCD take the tiles of their common pixel range round-robin, 0 contiguous runs), "tile_xcd" (1 XCD-grouped tile ids in the persistent tile loops of csrc/stem.hip, 0 dispatch order), "conv_1x1s" (1 the persistent 1x1 kernel with register-resident filters of csrc/conv_1x1s.h on the HBM-bound 1x1 layers, 0 the tile kernels, 2 also small launches), "wgrad_strip" / "conv_strip" (1 the strip-walking kernels of csrc/wgrad_strip.h / conv_strip.h on the
 * small-channel 3x3 layers, 0 off, 2 also small launches, N > 2: N rows per block -- tests).  The environment variable Y3_TUNE="key=value,key=value" is read once when the library is first used.
 * Process-wide, not stream-ordered: set a knob before enqueueing the launches it should affect. */
int y3_tune_set(const char* key, int64_t value); /* 0, or -1 for an unknown key */
int64_t y3_tune_get(const char* key);            /* INT64_MIN for an unknown key */
void y3_tune_reset(void);                        /* defaults + Y3_TUNE again */

/* number of elements (of `dtype`) of a packed filter bank for (cout, cin, ksize).  ALWAYS size banks with this call: a 3x3 bank with cout % 256 == 0 and
 * cin % 32 == 0 is TWO copies -- the row-major one every kernel reads, followed by a fragment-ordered one (per 64-row wave and K-step a contiguous 4 KiB block
 * of four MFMA A fragments) that the persistent 3x3 kernel loads straight into registers (csrc/conv_v10.h, csrc/y3_common.h::y3_frag_index).  Every packer
 * below (y3_pack_filter, _dgrad, _pair, _jobs) writes both (ABI version 2). */
size_t y3_packed_filter_elems(int32_t cout, int32_t cin, int32_t ksize);
/* OIHW fp32 (cout_src x cin_src x k x k) -> packed [cout_pad][k*k*cin_pad (+K pad)] in `dtype`;
 * `cout`/`cin` are the padded logical sizes used by y3_conv2d_fwd (>= the source sizes; pad = 0).
 * Replaces the layout the reference gets from nn.Conv2d.weight (models/common.py:68). */
int y3_pack_filter(const float* w_oihw, int32_t cout_src, int32_t cin_src, int32_t ksize, int32_t cout, int32_t cin,
                   int32_t dtype, void* packed, void* stream);

int y3_conv2d_fwd(const y3_conv_desc* desc, const y3_tensor* x, const void* packed_filter, const float* bias,
                  const y3_tensor* residual /* may be NULL */, const y3_tensor* y, void* stream);
/* The same call with a caller-owned scratch buffer of y3_conv_workspace_bytes() bytes (256-byte aligned, used by one stream at a time): unlocks the K-split
 * form of the persistent 3x3 kernel for SMALL launches of the 3x3 / stride-1 layers with cout % 256 == 0 (models/common.py:150-165 Bottleneck.cv2, the 3x3
 * convs of the head; batch 1-4 at 640x640) -- the channel blocks of every tile are cut into slices, every (tile, slice) writes fp32 partial sums into the
 * workspace and a second launch adds them in slice order and applies bias / SiLU / residual / statistics.  Results are deterministic (fixed split, fixed
 * summation order).  Launches the form does not cover run exactly as y3_conv2d_fwd. */
size_t y3_conv_workspace_bytes(void);
int y3_conv2d_fwd_ws(const y3_conv_desc* desc, const y3_tensor* x, const void* packed_filter, const float* bias,
                     const y3_tensor* residual /* may be NULL */, const y3_tensor* y, void* workspace, size_t workspace_bytes,
                     void* stream);
/* Which kernel variant the dispatcher picks for this problem ("v10", "v10h", "v10k", "s1x1", "v6", "v3_bk64_128x128", "strip", ..., "direct"); launches nothing.
 * The parity tests assert it so that a tolerance is always attached to the kernel that actually ran. */
int y3_conv2d_fwd_variant(const y3_conv_desc* desc, const y3_tensor* x, const y3_tensor* y, int32_t has_residual,
                          size_t workspace_bytes, char* name, size_t name_capacity);
/* The tile plan of the persistent 3x3 kernel (csrc/conv_v10.h: "v10", "v10h", "v10k") for this problem; launches nothing.  One record of four int32 per tile of ONE filter
 * tile -- (block of the filter tile, tile of that block, first 32-pixel column block, column blocks) -- computed by the functions the kernel itself runs, so a host test can
 * check that the tiles cover the pixel axis exactly once and that the blocks sharing an XCD work on neighbouring tiles (knob "v10_group").  `records` may be NULL
 * (count only); *column_blocks = ceil(N Ho Wo / 32), *group_blocks = blocks per interleave group.  Fails when the dispatcher picks another kernel for the problem. */
int y3_conv_v10_tiles(const y3_conv_desc* desc, const y3_tensor* x, const y3_tensor* y, size_t workspace_bytes, int32_t* records,
                      int64_t capacity, int64_t* n_tiles, int32_t* column_blocks, int32_t* group_blocks);
/* Training forward of a 1x1 convolution whose INPUT is another layer's Conv + BatchNorm + activation (+ shortcut) output (reference models/common.py:75 Conv.forward inside
 * Bottleneck.forward, :165): `u_in` is that layer's pre-BatchNorm tensor; the launch computes y_in = act(in_scale * u_in + in_shift) (+ in_shortcut) on the way in -- the
 * arithmetic of y3_bn_act_fwd --, stores it to `y_in` once (the other consumers read it) and convolves it: y = conv1x1(y_in) + bias with statistics rows like
 * y3_conv2d_fwd_stats.  Replaces the y3_bn_act_fwd launch of the producing layer and this layer's read of y_in.  Only the HBM-bound Bottleneck.cv1 shapes
 * (cin / cout 64 / 32, 128 / 64, 256 / 128); y3_conv2d_fwd_bnin_rows returns the statistics rows, or -1 when the shape is not covered (the caller keeps the two launches). */
int64_t y3_conv2d_fwd_bnin_rows(const y3_conv_desc* desc, const y3_tensor* u_in, const y3_tensor* y_in, const y3_tensor* y, int32_t has_shortcut);
int y3_conv2d_fwd_bnin_stats(const y3_conv_desc* desc, const y3_tensor* u_in, const float* in_scale, const float* in_shift, int32_t in_act,
                             const y3_tensor* in_shortcut /* may be NULL */, const y3_tensor* y_in, const void* packed_filter, const float* bias, const y3_tensor* y,
                             float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* stream);
/* Name of the variant the last convolution / data-gradient call of the calling thread launched ("v3_quad": the four output-parity
 * classes of y3_conv2d_dgrad_s2 in one launch). */
int y3_conv_last_variant(char* name, size_t name_cap);

/* Stem convolution: the first layer `Conv(ch<=4, 32|64, 3, 1)` (reference models/yolov3.yaml:16, models/common.py:57-81) computed
 * straight from the caller's NCHW image, fused with the ingest (`im.half(); im /= 255`, val.py:354-360): no NHWC copy of the
 * image is made.  `packed` comes from y3_pack_filter_stem ([cout_pad32][3][16] in `dtype`, BN folded by the caller),
 * y is the NHWC output view (n, h, w, cout), stride 1 / pad 1 only; src_dtype u8 / f16 / bf16 / f32, compute f16 / bf16. */
size_t y3_packed_filter_stem_elems(int32_t cout);
int y3_pack_filter_stem(const float* w_oihw, int32_t cout_src, int32_t cin_src, int32_t cout, int32_t dtype, void* packed, void* stream);
int y3_stem_conv_fwd(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                     const float* bias, int32_t dtype, int32_t act, const y3_tensor* y, void* stream);
/* Training form of the first layer (models/common.py:75 `act(bn(conv(x)))` with batch statistics): the same launch also writes one row
 * of (sum, sum of squares) per filter of the STORED values per block into stat_rows ([rows][y->c][2] floats; rows =
 * y3_stem_conv_stats_rows(n, h, w), reported in *n_rows) for y3_bn_finalize_rows -- no separate reduction pass over the output. */
int64_t y3_stem_conv_stats_rows(int32_t n, int32_t h, int32_t w);
int y3_stem_conv_fwd_stats(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                           const float* bias, int32_t dtype, int32_t act, const y3_tensor* y, float* stat_rows, int64_t capacity_rows, int64_t* n_rows,
                           void* stream);
/* Layer 0 of the training step by recomputation (round 6; models/common.py:75 `act(bn(conv(x)))`, models/yolov3.yaml:16): the pre-BatchNorm output of the first layer --
 * 64 B per input pixel, the largest tensor of the step -- is never written.  y3_stem_conv_stats_only: the statistics rows of y3_stem_conv_fwd_stats without the
 * store (`shape` = (n, h, w, filters); its data pointer is not used).  y3_stem_conv_fwd_bn: y = act(scale * u + shift) with u = the conv output rounded to the
 * compute dtype, recomputed from the image -- bit for bit what y3_bn_act_fwd writes from a stored u. */
int y3_stem_conv_stats_only(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                            int32_t dtype, const y3_tensor* shape, float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* stream);
int y3_stem_conv_fwd_bn(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                        const float* scale, const float* shift, int32_t act, int32_t dtype, const y3_tensor* y, void* stream);
/* Layers 0 + 1 of yolov3 / yolov3-spp in one kernel (models/yolov3.yaml:16-17: Conv(3,32,3,1) -> Conv(32,64,3,2)): layer 0's
 * output (the largest tensor of the network, single consumer) stays in LDS.  packed0 / bias0: 32 filters in the stem format
 * (y3_pack_filter_stem); packed1 / bias1: 64 filters over 32 channels in the generic format (y3_pack_filter); y: NHWC
 * (n, (h-1)/2+1, (w-1)/2+1, 64). */
int y3_stem_pair_fwd(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor,
                     const void* packed0, const float* bias0, int32_t act0, const void* packed1, const float* bias1, int32_t act1,
                     int32_t dtype, const y3_tensor* y, void* stream);
/* Bottleneck(C, C), C = 64 or 128 (models/yolov3.yaml:18,20: layers 2 and 4) in one kernel: y = [x +] cv2(cv1(x)) with cv1 = 1x1 conv
 * C -> C/2 (+ bias1, act1) and cv2 = 3x3 stride-1 conv C/2 -> C (+ bias2, act2); the C/2-channel intermediate (rounded to the storage
 * dtype as the two-launch form stores it) stays in LDS and x is read once.  packed1 / packed2: generic banks (y3_pack_filter) of
 * C/2 x (1x1xC) and C x (3x3xC/2).  x, y: NHWC (n, h, w, C) views, not aliased. */
int y3_bneck_pair_fwd(const y3_tensor* x, const void* packed1, const float* bias1, int32_t act1, const void* packed2, const float* bias2,
                      int32_t act2, int32_t add_residual, int32_t dtype, const y3_tensor* y, void* stream);

/* NCHW (u8 / f16 / bf16 / f32) image batch -> NHWC `out_dtype`: cast, then true-divide by `divisor`
 * (1.0 = none) in the output dtype, channels zero-padded to out->c.
 * Replaces `im.half()/float(); im /= 255` of reference val.py:354-360 / train.py:380 when src is u8. */
int y3_nchw_to_nhwc(const void* src, int32_t src_dtype, int32_t n, int32_t c, int32_t h, int32_t w, float divisor,
                    int32_t out_dtype, const y3_tensor* out, void* stream);
/* NHWC -> contiguous NCHW of the same dtype (debug / feature export). */
int y3_nhwc_to_nchw(const y3_tensor* src, int32_t dtype, void* dst, void* stream);

/* MaxPool2d(k, stride, pad) with -inf padding; optional zero padding on the right/bottom edge first
 * (zpad_r, zpad_b) = nn.ZeroPad2d([0,zpad_r,0,zpad_b]).  Replaces reference models/yolov3-tiny.yaml:21-32
 * (nn.MaxPool2d / nn.ZeroPad2d) and one branch of SPP (models/common.py:287-290). */
int y3_maxpool2d(const y3_tensor* x, const y3_tensor* y, int32_t dtype, int32_t k, int32_t stride, int32_t pad,
                 int32_t zpad_r, int32_t zpad_b, void* stream);
/* SPP pyramid: y[..., c:2c]=mp5(x), [2c:3c]=mp9(x), [3c:4c]=mp13(x) in ONE pass; x may alias y[..., 0:c].
 * Replaces reference models/common.py:287-290 (3x max_pool2d + cat). */
int y3_spp_pyramid(const y3_tensor* x, const y3_tensor* y3c /* slice starting at channel c, 3c wide */, int32_t dtype,
                   void* stream);
/* nearest x2 upsample / channel-slice copy (fallbacks when not fused into a conv epilogue):
 * reference models/yolov3.yaml:43 (nn.Upsample) and models/common.py:428 (torch.cat). */
int y3_upsample2x(const y3_tensor* x, const y3_tensor* y, int32_t dtype, void* stream);
int y3_copy_slice(const y3_tensor* x, const y3_tensor* y, int32_t dtype, void* stream);

/* Detect head post-conv: reference models/yolo.py:98 (view/permute -> raw (bs,na,ny,nx,no)) and
 * :104-108 (sigmoid, xy=(s*2+grid)*stride, wh=(s*2)^2*anchor_grid, cat, view) for ONE level.
 * head: NHWC (bs,ny,nx, >= na*no) conv output; raw (may be NULL): contiguous (bs,na,ny,nx,no);
 * z (may be NULL): rows [row_offset, row_offset+na*ny*nx) of a contiguous (bs,total_rows,no) tensor.
 * anchors_px: na*2 floats on the HOST (anchor w,h in pixels = Detect.anchors[i]*stride[i]).
 * Arithmetic rounds after every op in `dtype`, as torch does for half tensors. */
int y3_detect_decode(const y3_tensor* head, int32_t dtype, int32_t na, int32_t no, const float* anchors_px,
                     float stride, void* raw, void* z, int64_t row_offset, int64_t total_rows, void* stream);

/* Batched non_max_suppression: reference utils/general.py:630-750 incl. torchvision.ops.nms (:733).
 * pred: contiguous (bs, n_rows, 5+nc) of `dtype`.  Output: out_rows (bs, max_det, 6) fp32
 * [x1,y1,x2,y2,conf,cls] in descending-score order (score ties keep the reference's nonzero order, i.e.
 * torch.sort(stable=True)), out_counts (bs) int32.  classes: optional device int32 list.
 * Work space from y3_nms_workspace_bytes().  No host sync: counts stay on device. */
typedef struct {
    double iou_thres;           /* compared in double, as torchvision's CPU kernel does */
    float conf_thres;           /* rounded to `dtype` before the compares, as torch does for a python scalar */
    int32_t multi_label, agnostic;
    int32_t max_det, max_nms;   /* reference: 300 / 30000 */
    float max_wh;               /* reference: 7680 */
    int32_t n_classes_filter;   /* 0 = no class filter */
} y3_nms_params;
/* capacity = candidate slots for the whole batch; <= 0 picks the default (bs*16384, clamped to the
 * worst case bs*n_rows*nc).  Pass the same capacity to both calls. */
size_t y3_nms_workspace_bytes(int32_t bs, int32_t n_rows, int32_t nc, const y3_nms_params* p, int64_t capacity);
/* out_status (device, 2 x int32): [0] = 1 if the candidate capacity overflowed (results invalid: call
 * again with a workspace sized for capacity >= out_status[1]); [1] = total candidates found. */
int y3_nms(const void* pred, int32_t dtype, int32_t bs, int32_t n_rows, int32_t nc, const y3_nms_params* p,
           const int32_t* classes, float* out_rows, int32_t* out_counts, int32_t* out_status, int64_t capacity,
           void* workspace, size_t workspace_bytes, void* stream);

/* Output edge after NMS, batched (csrc/val_edge.hip).
 * y3_scale_boxes: reference utils/general.py:613-626 scale_boxes + upstream clip_boxes (callers val.py:397,403, detect.py:223),
 * in place on fp32 xyxy rows: row r of image i starts at rows + i*img_stride + r*row_stride (floats; row_stride >= 4 so both the
 * (bs, max_det, 6) NMS output and an (n, 4) view work).  counts (DEVICE, may be NULL = max_rows rows per image); params: DEVICE
 * (bs, 5) fp32 {gain, pad_x, pad_y, w0, h0} computed by the caller exactly as the reference does (ratio_pad or the shapes).
 * y3_match_detections: reference val.py:147-188 process_batch with upstream box_iou, for every image: dets as above
 * ([x1,y1,x2,y2,conf,cls], row_stride >= 6), labels DEVICE (nl, 5) fp32 [cls,x1,y1,x2,y2] grouped by image through
 * label_offsets (DEVICE, bs+1 int32), iouv DEVICE (niou) fp32; correct: DEVICE (bs, max_det, niou) bytes (0/1), rows beyond
 * counts[i] are 0.  max_det <= 4096. */
/* Input edge: reference utils/augmentations.py:104-134 letterbox(auto=False) -- cv2.resize(INTER_LINEAR) to (new_h, new_w),
 * cv2.copyMakeBorder(color) to (H1, W1) -- fused with the HWC -> CHW transpose of models/common.py:867, one image per call.
 * src: DEVICE (h0, w0, cs) uint8, cs >= 3 interleaved channels (the first 3 are used); dst_batch: DEVICE (n, 3, H1, W1) uint8,
 * image `index` is written.  new_h/new_w/top/left are computed by the caller exactly as the reference does (Python round()). */
int y3_letterbox_u8(const uint8_t* src, int32_t h0, int32_t w0, int32_t cs, uint8_t* dst_batch, int32_t index, int32_t H1,
                    int32_t W1, int32_t new_h, int32_t new_w, int32_t top, int32_t left, int32_t color, void* stream);
/* Test-time augmentation edges: reference models/yolo.py:239-276 (_forward_augment, _descale_pred, _clip_augmented).
 * y3_scale_img: upstream ultralytics.utils.torch_utils.scale_img (un-vendored) fused with the `x.flip(3)` of models/yolo.py:246 -- NCHW (n, c, h, w)
 * -> NCHW (n, c, oh, ow): the (optionally left-right mirrored) batch resized like F.interpolate(size=(ih, iw), mode="bilinear",
 * align_corners=False), pad_value (the reference's 0.447) right / below up to (oh, ow).  The caller computes ih = int(h ratio), iw = int(w ratio),
 * oh = ceil(h ratio / gs) gs, ow likewise, as the reference does.  src != dst.
 * y3_descale_pred: rows [row0, row0 + nrows) of every image of a decoded prediction (bs, src_rows, no) -> rows [dst_row0, ...) of the concatenated
 * (bs, dst_rows, no) result, xywh / scale, flip 3: x = img_w - x, flip 2: y = img_h - y (0: none); every step rounded to dtype like the reference's
 * in-place tensor ops.  The row windows are _clip_augmented's. */
int y3_scale_img(const void* src, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                 int32_t flip_lr, float pad_value, void* dst, void* stream);
int y3_descale_pred(const void* src, int32_t dtype, int32_t bs, int32_t src_rows, int32_t no, int32_t row0, int32_t nrows, float scale,
                    int32_t flip, float img_h, float img_w, void* dst, int32_t dst_rows, int32_t dst_row0, void* stream);
int y3_scale_boxes(float* rows, int64_t img_stride, int32_t row_stride, const int32_t* counts, int32_t bs, int32_t max_rows,
                   const float* params, void* stream);
int y3_match_detections(const float* dets, int64_t img_stride, int32_t row_stride, const int32_t* counts, int32_t bs,
                        int32_t max_det, const float* labels, const int32_t* label_offsets, const float* iouv, int32_t niou,
                        uint8_t* correct, void* stream);

/* ComputeLoss: reference utils/loss.py:98-244 (build_targets :183-244, __call__ :131-181, criteria :104-129,
 * FocalLoss :31-63 when fl_gamma > 0) with upstream bbox_iou(CIoU) and smooth_bce.
 * preds: HOST array of nl DEVICE pointers, level i is contiguous (bs, na, ny[i], nx[i], nc+5) of `dtype`;
 * targets: DEVICE (nt, 6) fp32 [img, cls, x, y, w, h] normalised; anchors in grid units (Detect.anchors).
 * y3_loss_fwd writes out4 (DEVICE) = [ (lbox+lobj+lcls)*bs, lbox, lobj, lcls ] (gains applied, :176-181).
 * y3_loss_bwd (same params / preds / targets / workspace as the preceding fwd) overwrites grads[i] (same
 * shape and dtype as preds[i]) with d(out4[0]) / d preds[i] * grad_out[0] (grad_out: DEVICE scalar or NULL = 1).
 * Duplicate matches of one cell: tobj takes the LAST match in the reference's list order (CPU index_put), box/cls
 * gradients accumulate.  A target whose image index is outside [0, bs) or whose class is outside [0, nc) (the reference raises an
 * IndexError on the host) is skipped and turns out4 into NaN: the step fails loudly instead of reading out of bounds.  gr = 1 (the reference's value); autobalance: y3_loss_level_obj below. */
typedef struct {
    int32_t nl, na, nc, bs;
    int32_t ny[5], nx[5];
    float anchors[50]; /* [nl][na][2] */
    float balance[5];  /* reference: {4, 1, 0.4} for nl == 3, else first nl of {4, 1, .25, .06, .02}  (:122) */
    float anchor_t;    /* hyp['anchor_t'] */
    float box_gain, obj_gain, cls_gain; /* hyp['box'], hyp['obj'], hyp['cls'] (already scaled, train.py:327-329) */
    float cls_pw, obj_pw;               /* BCE pos_weight */
    float cp, cn;                       /* smooth_bce(label_smoothing) */
    float fl_gamma;                     /* 0 = plain BCE */
    int32_t sort_obj_iou;               /* ComputeLoss.sort_obj_iou (:101,156-158): a cell matched by several targets keeps its LARGEST iou as objectness target
                                           (the reference writes tobj in ascending-iou order); 0 = the last match in list order wins, the reference's default (ABI 3) */
} y3_loss_params;
size_t y3_loss_workspace_bytes(const y3_loss_params* p, int32_t nt);
int y3_loss_fwd(const y3_loss_params* p, int32_t dtype, const void* const* preds, const float* targets, int32_t nt,
                float* out4, void* workspace, size_t workspace_bytes, void* stream);
int y3_loss_bwd(const y3_loss_params* p, int32_t dtype, const void* const* preds, const float* targets, int32_t nt,
                const float* grad_out, void* const* grads, void* workspace, size_t workspace_bytes, void* stream);
/* ComputeLoss(autobalance=True), reference utils/loss.py:171-175: the per-level objectness losses `obji` of the forward that filled
 * `workspace` (same params / dtype / nt), nl device floats; the caller reads them back and moves its balance weights */
int y3_loss_level_obj(const y3_loss_params* p, int32_t dtype, int32_t nt, void* workspace, size_t workspace_bytes, float* obj_levels,
                      void* stream);

/* ---------------------------------------------------------------------------------------------- training
 * Train-mode `Conv` = act(bn(conv(x))) with BATCH statistics (reference models/common.py:75; BN eps 1e-3 / momentum
 * 0.03 set at models/yolo.py:229) and the autograd of the graph, decomposed as:
 *   forward : u = y3_conv2d_fwd(x)            (act NONE, zero bias; u is kept for the backward)
 *             y3_bn_stats(u) -> y3_bn_finalize (or y3_bn_stats_finalize) -> y = y3_bn_act_fwd(u [, residual])
 *   backward: du = y3_bn_act_bwd(u, dy)       (also gives dgamma, dbeta)
 *             dW = y3_conv2d_wgrad(x, du)     (fp32 OIHW, the layout of nn.Conv2d.weight.grad)
 *             dx (+)= y3_conv2d_fwd(du, filter packed by y3_pack_filter_dgrad [, residual = dx, in_dilation = stride])
 * `sums` is a caller-owned scratch of Y3_BN_SCRATCH_DOUBLES(C) doubles (totals + per-block partial rows; reductions
 * are atomics-free and deterministic); all per-channel vectors are DEVICE fp32. */
#define Y3_BN_SCRATCH_DOUBLES(C) ((size_t)(1 + 512) * 2 * (size_t)(C))
int y3_bn_stats(const y3_tensor* u, int32_t dtype, double* sums, void* stream);
int y3_bn_finalize(const double* sums, int64_t count, int32_t channels, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean /* updated in place, may be NULL */, float* running_var,
                   float* scale, float* shift, float* mean, float* invstd, void* stream);
/* Training forward with the BatchNorm statistics taken in the conv epilogue (f16/bf16 MFMA path, no residual / upsample):
 * y3_conv2d_fwd plus, per (pixel tile, pixel wave) of the dispatched tile variant, one row [cout][2] fp32 of (sum, sum of
 * squares) of the STORED output values in stat_rows; *n_rows = rows written (query: y3_conv2d_fwd_stats_rows, -1 on error).
 * y3_bn_finalize_rows sums the rows in fp64 (fixed order; `sums` is a Y3_BN_SCRATCH_DOUBLES(C) scratch, totals land in its first
 * 2*C entries) and finalizes like y3_bn_finalize (count = n*h*w). */
int64_t y3_conv2d_fwd_stats_rows(const y3_conv_desc* desc, const y3_tensor* x, const y3_tensor* y);
int y3_conv2d_fwd_stats(const y3_conv_desc* desc, const y3_tensor* x, const void* packed_filter, const float* bias,
                        const y3_tensor* y, float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* stream);
/* the same with the scratch buffer of y3_conv2d_fwd_ws (the persistent kernel writes 4 rows per pixel tile: query with the same size) */
int64_t y3_conv2d_fwd_stats_rows_ws(const y3_conv_desc* desc, const y3_tensor* x, const y3_tensor* y, size_t workspace_bytes);
int y3_conv2d_fwd_stats_ws(const y3_conv_desc* desc, const y3_tensor* x, const void* packed_filter, const float* bias,
                           const y3_tensor* y, float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* workspace,
                           size_t workspace_bytes, void* stream);
int y3_bn_finalize_rows(const float* stat_rows, int64_t n_rows, int64_t count, int32_t channels, double* sums, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                        float* shift, float* mean, float* invstd, void* stream);
/* y3_bn_stats followed by y3_bn_finalize (count = n*h*w of u) with the partial-row sum and the finalize in one launch */
int y3_bn_stats_finalize(const y3_tensor* u, int32_t dtype, double* sums, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean /* may be NULL */, float* running_var, float* scale, float* shift,
                         float* mean, float* invstd, void* stream);
int y3_bn_act_fwd(const y3_tensor* u, const float* scale, const float* shift, const y3_tensor* residual /* may be NULL */,
                  const y3_tensor* y, int32_t dtype, int32_t act, void* stream);
int y3_bn_act_bwd(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean,
                  const float* invstd, int32_t dtype, int32_t act, double* sums, const y3_tensor* du,
                  float* dgamma /* may be NULL */, float* dbeta /* may be NULL */, void* stream);
/* y3_bn_act_bwd for a unit with a residual input (`out = act(bn(conv(x))) + residual`, Bottleneck with shortcut): also writes
 * (gres_accumulate = 0) or accumulates (1) dy into the residual's gradient on the pass that reads dy anyway. */
int y3_bn_act_bwd_res(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean,
                      const float* invstd, int32_t dtype, int32_t act, double* sums, const y3_tensor* du, float* dgamma,
                      float* dbeta, const y3_tensor* gres, int32_t gres_accumulate, void* stream);
/* SyncBatchNorm (reference train.py:270-272 `--sync-bn`: torch.nn.SyncBatchNorm.convert_sync_batchnorm before DDP).  The host side all-reduces a few
 * fp64 words per layer between these calls (yolov3_amd/train_engine.py); everything stays on the device:
 *   forward : y3_bn_sum_rows (rows of a y3_conv2d_fwd_stats launch -> sums[0 .. 2C) = per-channel (sum, sum of squares); y3_bn_stats is the form that
 *             reads u) -> all-reduce(sums[0 .. 2C), element count) -> y3_bn_finalize_devcount (y3_bn_finalize with the count read from the device).
 *   backward: y3_bn_act_bwd_reduce (this rank's totals of (dz, dz xhat) in sums[0 .. 2C), dgamma / dbeta from them -- local, the gradient exchange
 *             averages them like every other parameter gradient -- and the local means in sums[2C .. 4C)) -> the host overwrites sums[2C .. 4C)
 *             with all-reduced totals / all-reduced count -> y3_bn_act_bwd_apply (du, and the residual's gradient as y3_bn_act_bwd_res).
 * y3_bn_act_bwd == reduce + apply back to back. */
int y3_bn_sum_rows(const float* stat_rows, int64_t n_rows, int32_t C, double* sums, void* stream);
int y3_bn_finalize_devcount(const double* sums, const double* count_dev, int32_t C, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                            float* invstd, void* stream);
int y3_bn_act_bwd_reduce(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean,
                         const float* invstd, int32_t dtype, int32_t act, double* sums, float* dgamma /* may be NULL */,
                         float* dbeta /* may be NULL */, void* stream);
int y3_bn_act_bwd_apply(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean,
                        const float* invstd, int32_t dtype, int32_t act, const double* sums, const y3_tensor* du,
                        const y3_tensor* gres /* may be NULL */, int32_t gres_accumulate, void* stream);
/* Backward of the first layer (Conv(3, 32, 3, 1) + BatchNorm + act; no data gradient) in two passes over (u, dy) instead of three
 * plus a write: the reduction of y3_bn_act_bwd (totals into `sums`, dgamma, dbeta), then ONE kernel that applies the BatchNorm /
 * activation backward, rounds du to the storage dtype as y3_bn_act_bwd would have stored it, and accumulates
 * dw_oihw[32][cin][3][3] (fp32, the layout of nn.Conv2d.weight.grad) against the source image -- du never reaches memory.
 * x_nchw / src_dtype / divisor: the image exactly as y3_stem_conv_fwd received it.  32 filters, cin <= 3, f16 / bf16.
 * workspace: y3_stem_bn_bwd_wgrad_workspace_bytes() bytes (one fp32 partial tile per persistent block; summed in block order). */
size_t y3_stem_bn_bwd_wgrad_workspace_bytes(void);
int y3_stem_bn_bwd_wgrad(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const y3_tensor* u,
                         const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                         int32_t act, double* sums, float* dgamma /* may be NULL */, float* dbeta /* may be NULL */, float* dw_oihw,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same backward when the forward did not store u (y3_stem_conv_stats_only + y3_stem_conv_fwd_bn): both passes rebuild the tile's u from the image patch
 * they stage anyway (the forward's MFMAs on the forward's operands: the same bits).  packed0: the stem-packed filters the forward used (y3_pack_filter_stem). */
int y3_stem_bn_bwd_wgrad_recompute(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed0,
                                   const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                   int32_t act, double* sums, float* dgamma /* may be NULL */, float* dbeta /* may be NULL */, float* dw_oihw,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* OIHW fp32 -> filter bank of the data-gradient convolution: `cin` filters over (kh, kw, cout) with flipped taps. */
int y3_pack_filter_dgrad(const float* w_oihw, int32_t cout_src, int32_t cin_src, int32_t ksize, int32_t cout, int32_t cin,
                         int32_t dtype, void* packed, void* stream);
/* y3_pack_filter (forward bank) and y3_pack_filter_dgrad (data-gradient bank) of one layer in one launch (f16/bf16). */
int y3_pack_filter_pair(const float* w_oihw, int32_t cout_src, int32_t cin_src, int32_t ksize, int32_t cout, int32_t cin,
                        int32_t dtype, void* packed_fwd, void* packed_dgrad, void* stream);
/* y3_pack_filter_pair for MANY layers in one launch (the training step re-packs every layer's banks each step).  `jobs` is a DEVICE
 * array; job i occupies blocks [first_block, first_block + y3_pack_job_blocks(...)) of a grid of total_blocks 256-thread blocks, the
 * jobs laid out back to back in array order.  packed_fwd / packed_dgrad may be NULL (that bank is not wanted).  f16 / bf16.
 * Only the elements that come from a weight are written (source-indexed 32 x 32 tiles, round 5): ZERO-FILL a bank once when it is allocated -- its row / K
 * padding is never touched afterwards. */
typedef struct y3_pack_job {
    const float* w;          /* OIHW fp32 weights (nn.Conv2d.weight) */
    void* packed_fwd;        /* y3_packed_filter_elems(cout, cin, ksize) elements, or NULL */
    void* packed_dgrad;      /* y3_packed_filter_elems(cin, cout, ksize) elements, or NULL */
    int32_t cout_src, cin_src, ksize, cout, cin;
    int32_t first_block;
} y3_pack_job;
int64_t y3_pack_job_blocks(int32_t ksize, int32_t cout, int32_t cin, int32_t want_fwd, int32_t want_dgrad);
int y3_pack_filter_jobs(const y3_pack_job* jobs_device, int32_t n_jobs, int64_t total_blocks, int32_t dtype, void* stream);
/* Data gradient of a 3x3 stride-2 pad-1 conv without multiplying the zero taps of the dilated form: four output-parity
 * classes, each a small stride-1 conv of du (1, 2, 2 and 4 taps) with its own filter bank, written to every second
 * pixel of gx (+= residual when given; residual may alias gx).  f16/bf16 only. */
size_t y3_packed_filter_dgrad_s2_elems(int32_t cout, int32_t cin);
int y3_pack_filter_dgrad_s2(const float* w_oihw, int32_t cout_src, int32_t cin_src, int32_t cout, int32_t cin, int32_t dtype,
                            void* packed4, void* stream);
int y3_conv2d_dgrad_s2(int32_t dtype, const y3_tensor* du, const void* packed4, const y3_tensor* residual /* may be NULL */,
                       const y3_tensor* gx, void* stream);
/* filter gradient (and optional bias gradient = per-channel sum of du) of the conv described by `desc`
 * (dtype, ksize, stride, cin, cout = padded sizes of x / du); dw is (cout_real, cin_real, k, k) fp32, overwritten. */
size_t y3_conv2d_wgrad_workspace_bytes(const y3_conv_desc* desc, const y3_tensor* x);
/* dry run: the geometry y3_conv2d_wgrad launches for (desc, x) -- tile edge (128 / 256; 0 = direct fp32 kernel), number of pixel
 * slices (split-K over n*ho*wo) and whether a slice's tiles are grouped per XCD (knob "wgrad_xcd"); tile 3 = the 3x3 strip kernel (csrc/wgrad_strip.h),
 * slices = its persistent blocks */
int y3_conv2d_wgrad_plan(const y3_conv_desc* desc, const y3_tensor* x, int32_t* tile, int64_t* slices, int32_t* xcd_grouped);
int y3_conv2d_wgrad(const y3_conv_desc* desc, const y3_tensor* x, const y3_tensor* du, int32_t cout_real, int32_t cin_real,
                    float* dw_oihw, float* dbias /* may be NULL */, void* workspace, size_t workspace_bytes, void* stream);
/* backward of nn.Upsample(x2, nearest) / nn.MaxPool2d (+ZeroPad2d) / Detect's view+permute (models/yolo.py:98) */
int y3_upsample2x_bwd(const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t accumulate, void* stream);
int y3_maxpool2d_bwd(const y3_tensor* x, const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t k, int32_t stride,
                     int32_t pad, int32_t zpad_r, int32_t zpad_b, int32_t accumulate, void* stream);
/* The max-pool backward with a byte of workspace per OUTPUT element (y3_maxpool2d_bwd_workspace_bytes): pass 1 records each window's first maximum, pass 2 looks
 * the k^2 windows of an input element up -- k^2 byte loads per element where y3_maxpool2d_bwd scans k^2 windows of k^2 elements (SPP's 13 x 13 pool: 210 ms -> well
 * under a millisecond at batch 64).  Same sums in the same order.  Without a (large enough) workspace, or k > 15, it runs y3_maxpool2d_bwd. */
size_t y3_maxpool2d_bwd_workspace_bytes(const y3_tensor* x, int32_t k, int32_t stride, int32_t pad, int32_t zpad_r, int32_t zpad_b);
int y3_maxpool2d_bwd_ws(const y3_tensor* x, const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t k, int32_t stride, int32_t pad, int32_t zpad_r,
                        int32_t zpad_b, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
int y3_detect_raw_bwd(const void* graw, int32_t dtype, int32_t bs, int32_t na, int32_t ny, int32_t nx, int32_t no,
                      const y3_tensor* ghead, void* stream);

/* Fused optimizer step: GradScaler.unscale_ + inf check, clip_grad_norm_(max_norm), SGD(nesterov)
 * with per-tensor lr / weight decay (reference utils/torch_utils.py:207-237: three parameter groups) and the ModelEMA
 * lerp (train.py:414-422), for ALL tensors in three launches.  `tensor_table`: DEVICE array of n_tensors records
 *   { float* param; const float* grad; float* momentum_buf; float* ema (or NULL); int64 numel; float lr, weight_decay;
 *     int32 first_chunk; int32 pad; }   (y3_sgd_tensor_record_bytes() == 56; chunks of 16384 elements)
 * scratch: n_chunks + 2 floats; scratch[0] = unclipped gradient norm, scratch[1] = clip coefficient after the call.
 * found_inf (device int32) = 1 when a gradient was inf/nan, in which case nothing is updated (GradScaler.step). */
size_t y3_sgd_tensor_record_bytes(void);
int y3_sgd_step(const void* tensor_table, int32_t n_tensors, int32_t n_chunks, float inv_scale, float max_norm, float momentum,
                int32_t nesterov, int32_t first_step, float ema_decay, float* scratch, int32_t* found_inf, void* stream);
/* The same step with the loss scale read from DEVICE memory (1 float), and torch.cuda.amp.GradScaler.update() (reference
 * train.py:345 `scaler = GradScaler(enabled=amp)`, :416-417 `scaler.step(optimizer); scaler.update()`; ATen _amp_update_scale_):
 * found_inf -> scale *= backoff_factor, tracker = 0; otherwise ++tracker and at growth_interval: scale *= growth_factor (if finite),
 * tracker = 0.  No host synchronisation anywhere. */
int y3_sgd_step_dynamic(const void* tensor_table, int32_t n_tensors, int32_t n_chunks, const float* loss_scale, float max_norm,
                        float momentum, int32_t nesterov, int32_t first_step, float ema_decay, float* scratch, int32_t* found_inf,
                        void* stream);
int y3_loss_scale_update(float* loss_scale, int32_t* growth_tracker, const int32_t* found_inf, float growth_factor,
                         float backoff_factor, int32_t growth_interval, void* stream);
/* The owner's step of the two-phase gradient exchange that replaces a bucket's all-reduce on a fully connected xGMI mesh (reference: DDP's gradient
   averaging, utils/torch_utils.py:60-72 smart_DDP / train.py:411): `parts` holds n_parts contributions of n floats each ([n_parts][n], what the all-to-all
   of the shards delivered); out[i] = (parts[0][i] + ... + parts[n_parts - 1][i]) * scale, added in that order.  `out` may alias none of `parts`. */
int y3_shard_mean(const float* parts, int32_t n_parts, int64_t n, float scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV3_HIP_H */
