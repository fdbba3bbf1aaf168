// Stem convolution: the network's first layer (3x3, stride 1, pad 1, <= 4 input channels, 32 or 64 filters) computed
// straight from the caller's NCHW image -- models/common.py:57-81 `Conv(3, 32, 3, 1)` (models/yolov3.yaml:16) together
// with the `im.half()` / `im /= 255` ingest of val.py:354-360.  It replaces y3_nchw_to_nhwc + the generic implicit-GEMM launch
// for that layer: the generic path stages K = 9 taps x 8 padded channels = 72 -> 96 and spends its time in per-block
// prologue / epilogue (0.43 ms at 2.4 TB/s for 640x640 bs 32); here a block keeps a 10 x 68 pixel patch of the image in
// LDS as 4-channel pixels, a 32-pixel MFMA tile needs three 16-byte fragment reads (one per filter row: 4 pixels x 4
// channels = 16 k-values, the 4th pixel meets zero weights) and three v_mfma_f32_32x32x16, and the 64-byte NHWC rows of
// consecutive pixels leave as fully contiguous 1 KiB stores.
#include "../../include/yolov3_hip.h"
#include "y3_common.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace {

struct StemArgs {
    const void* x;      // (N, Cin, H, W) source image
    const void* w;      // packed filters [cout_pad32][3 (kh)][16 (kw*4 + c)]
    const float* bias;  // cout floats or null
    void* y;            // NHWC output view
    int N, Cin, H, W;
    int ypitch, Cout;
    int act;
    int tiles_w, tiles_h;
    float divisor;
    float* stats;       // optional: row blockIdx.x of [Cout][2] floats = (sum, sum of squares) of the block's STORED values (training: BatchNorm
                        // batch statistics without a separate pass over the 64 B/pixel output)
    // round 6, layer 0 of the training step by recomputation (the 64 B/pixel pre-BatchNorm tensor is never written):
    //   y == nullptr: statistics only -- the rows are those of the values a store would have written, nothing is stored;
    //   scale != nullptr: what is stored is act(scale u + shift) of the rounded conv output u, rounded again -- bit for bit what bn_act_fwd_kernel writes from the stored u
    const float* scale;
    const float* shift;
    int act_bn;
};

constexpr int TR = 8, TW = 64;   // output rows x columns per block
constexpr int PR = TR + 2;       // patch rows (one halo row above and below)
constexpr int PW = TW + 4;       // patch columns: one halo column left, one right, and the 4th pixel of the widest fragment read

template <typename T> struct Mfma16;
template <> struct Mfma16<f16_t> {
    typedef f16x8 frag;
    static Y3_DEV f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<bf16_t> {
    typedef bf16x8 frag;
    static Y3_DEV f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

template <typename S> Y3_DEV float src_f32(S v) { return (float)v; }

template <typename T, typename S, int MC>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemArgs p) {
    typedef typename Mfma16<T>::frag frag;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CH = MC * 4;         // 16-byte chunks per output pixel
    constexpr int RB = CH * 16;        // bytes per output pixel (MC * 32 filters)
    constexpr int PPI = 64 / CH;       // pixels per store instruction
    constexpr int NI = 32 / PPI;
    __shared__ __attribute__((aligned(16))) unsigned char patch[PR * PW * 8];
    __shared__ __attribute__((aligned(16))) unsigned char slices[4 * 32 * RB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;   // (XCD-grouped tile ids as in stem_pair_kernel were measured on this non-persistent grid: 3 % slower at batch 64, profiles/r04_tile_xcd_ab.txt)
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h;
    const int n = b / p.tiles_h;
    const int row0 = th * TR, col0 = tw * TW;

    // ---- image patch -> LDS as 4-channel pixels (channel 3 = 0), zero outside the image ----
    const S* __restrict__ xs = (const S*)p.x + (long long)n * p.Cin * p.H * p.W;
    for (int e = tid; e < PR * PW; e += 256) {
        const int pr = e / PW, pc = e - pr * PW;
        const int gh = row0 + pr - 1, gw = col0 + pc - 1;
        vec4 v = {(T)0.0f, (T)0.0f, (T)0.0f, (T)0.0f};
        if ((unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W) {
            const long long o = (long long)gh * p.W + gw;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < p.Cin) v[c] = from_f32<T>(src_f32<S>(xs[(long long)c * p.H * p.W + o]) / p.divisor);
            if (p.Cin > 3) v[3] = from_f32<T>(src_f32<S>(xs[3ll * p.H * p.W + o]) / p.divisor);
        }
        *(vec4*)(patch + e * 8) = v;
    }

    // ---- filters: lane (filter = lane & 31, fk = lane >> 5) holds k = 8 fk + j of each filter row ----
    const int frow = lane & 31, fk = lane >> 5;
    frag af[MC][3];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) af[a][kh] = *(const frag*)((const T*)p.w + ((a * 32 + frow) * 3 + kh) * 16 + fk * 8);
    f32x4 bz[MC][4];
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cb = a * 32 + 8 * g + 4 * fk;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bz[a][g] = (p.bias && cb + 4 <= p.Cout) ? *(const f32x4*)(p.bias + cb) : z;
        }
    __syncthreads();

    unsigned char* wl = slices + wv * (32 * RB);
    T* __restrict__ yg = (T*)p.y;
    const int rp = lane / CH, ch = lane % CH;
    const bool want_stats = p.stats != nullptr;   // kernel-uniform
    const bool bn_apply = p.scale != nullptr;     // kernel-uniform
    f32x2 bsc[4], bsh[4];
    if (bn_apply && ch * 8 + 8 <= p.Cout) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bsc[q] = f32x2{p.scale[ch * 8 + 2 * q], p.scale[ch * 8 + 2 * q + 1]};
            bsh[q] = f32x2{p.shift[ch * 8 + 2 * q], p.shift[ch * 8 + 2 * q + 1]};
        }
    }
    float st0[8], st1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) st0[q] = st1[q] = 0.0f;
    for (int t = wv; t < TR * TW / 32; t += 4) {
        const int trow = t / (TW / 32), j0 = (t % (TW / 32)) * 32;
        f32x16 acc[MC];
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[a][4 * g + q] = bz[a][g][q];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            // pixel j0 + frow, taps kw = 2 fk and 2 fk + 1 (+ channel pad): 16 bytes at an 8-byte aligned address
            const unsigned char* src = patch + (((trow + kh) * PW + j0 + frow + 2 * fk) * 8);
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 raw;
            raw[0] = *(const unsigned long long*)src;
            raw[1] = *(const unsigned long long*)(src + 8);
            const frag bf = __builtin_bit_cast(frag, raw);
#pragma unroll
            for (int a = 0; a < MC; ++a) acc[a] = Mfma16<T>::run(af[a][kh], bf, acc[a]);
        }
        // ---- activation, pairs of filters rounded to T, lane-half swap (8 consecutive filters per lane), transpose through the wave's slice ----
        if (p.act == Y3_ACT_SILU) {
#pragma unroll
            for (int a = 0; a < MC; ++a) silu_vec<f32x16, 16>(acc[a]);
        }
#pragma unroll
        for (int a = 0; a < MC; ++a)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x4 ov;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(acc[a][8 * gp + 2 * h], acc[a][8 * gp + 2 * h + 1]),
                                                                     pack2<T>(acc[a][8 * gp + 4 + 2 * h], acc[a][8 * gp + 4 + 2 * h + 1]), false, false);
                    ov[h] = (unsigned)sw[0];
                    ov[2 + h] = (unsigned)sw[1];
                }
                const int chunk = a * 4 + gp * 2 + fk;
                *(u32x4*)(wl + frow * RB + ((chunk ^ (frow & (CH - 1))) << 4)) = ov;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int grow = row0 + trow;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pl = i * PPI + rp;
            const frag ov = *(const frag*)(wl + pl * RB + ((ch ^ (pl & (CH - 1))) << 4));
            const int gcol = col0 + j0 + pl;
            if (grow < p.H && gcol < p.W && ch * 8 + 8 <= p.Cout) {
                if (want_stats) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const float f = to_f32<T>(ov[q]); st0[q] += f; st1[q] += f * f; }
                }
                if (yg) {
                    frag out = ov;
                    if (bn_apply) {
                        u32x4 w4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 z = y3_bn_act2(f32x2{to_f32<T>(ov[2 * q]), to_f32<T>(ov[2 * q + 1])}, bsc[q], bsh[q], p.act_bn == Y3_ACT_SILU, false, f32x2{0.0f, 0.0f});
                            w4[q] = pack2<T>(z[0], z[1]);
                        }
                        out = __builtin_bit_cast(frag, w4);
                    }
                    *(frag*)(yg + ((long long)(n * p.H + grow) * p.W + gcol) * p.ypitch + ch * 8) = out;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next tile
    }
    if (want_stats) {
        // lanes rp * CH + ch hold partial sums of the same 8 filters: butterfly over the rp bits, then the four waves through LDS
#pragma unroll
        for (int off = CH; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < 8; ++q) { st0[q] += __shfl_xor(st0[q], off, 64); st1[q] += __shfl_xor(st1[q], off, 64); }
        __syncthreads();   // every wave is done with the patch: its first 2 KiB become the reduction buffer
        float* red = (float*)patch;
        if (rp == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { red[(wv * CH + ch) * 16 + 2 * q] = st0[q]; red[(wv * CH + ch) * 16 + 2 * q + 1] = st1[q]; }
        }
        __syncthreads();
        if (tid < CH * 16 && (tid >> 4) * 8 + 8 <= p.Cout) {   // thread = (channel group, filter of the group, sum / sum of squares)
            const float v = red[tid] + red[CH * 16 + tid] + red[2 * CH * 16 + tid] + red[3 * CH * 16 + tid];   // fixed order: deterministic
            p.stats[(long long)blockIdx.x * p.Cout * 2 + tid] = v;
        }
    }
}

// ---- layers 0 + 1 in one kernel --------------------------------------------------------------------------------------------
// models/yolov3.yaml:16-17: Conv(3, 32, 3, 1) -> Conv(32, 64, 3, 2).  Layer 0's output is the largest tensor of the network
// (64 B per input pixel: 839 MB at 640x640 batch 32) and has exactly one consumer; written and read back it costs 1.7 GB of
// HBM traffic around 0.5 GB of useful input + output (stem 0.29 ms + layer 1 0.33 ms of a 7.4 ms forward).  Here a block owns a
// 4 x 32 tile of LAYER-1 output pixels: it stages the 11 x 67 image patch as the stem kernel does, computes the 9 x 65 layer-0
// pixels the tile needs (bias + SiLU, rounded to T exactly as the stored tensor would be; positions outside the image are
// layer 1's zero padding) into LDS as [pixel][32 channels] rows, and runs layer 1 as an implicit GEMM out of LDS: wave (filter
// tile of 32, row pair) keeps its 18 filter fragments (9 taps x 32 channels) in registers for the block's whole life
// (persistent blocks, tiles in a grid-stride loop) and reads one 16-byte pixel fragment per MFMA.  Layer 0 is recomputed for
// the one-pixel overlap between tiles (585 pixels for 512 fresh ones: 14 % of a layer that is 1 % of the network's FLOPs).
struct PairArgs {
    const void* x;      // (N, Cin, H, W) source image
    const void* w0;     // stem-packed layer-0 filters [32][3][16]
    const float* b0;    // 32 floats
    const void* w1;     // generic packed layer-1 bank [>= 64 rows][kpad1], k = (kh * 3 + kw) * 32 + ci
    const float* b1;    // 64 floats
    void* y;            // NHWC (N, Ho, Wo, 64) view
    int N, Cin, H, W, Ho, Wo, ypitch, act0, act1, kpad1;
    int tiles_w, tiles_h, n_tiles;
    int xcd;            // 1: block b starts at tile xcd_remap(b) (knob "tile_xcd")
    float divisor;
#ifdef Y3_TIMELINE
    unsigned long long* tl;
#endif
};
#ifdef Y3_TIMELINE   // debug build (tools/stem_probe.py): per-wave tick sums of the phases of a tile
static unsigned long long* g_pair_tl = nullptr;
#define SP_T(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define SP_T(i) do { } while (0)
#endif
constexpr int QR = 4, QW = 32;                    // layer-1 output rows x columns per tile
constexpr int R0 = 2 * QR + 1, C0 = 2 * QW + 1;   // layer-0 pixels needed: 9 x 65
constexpr int XR = R0 + 2, XW = C0 + 3;           // image patch: one halo row/column each side + the 4th pixel of the widest fragment read
// layer-0 pixels in LDS: 64 bytes (32 channels) each; a region row holds its 33 even columns, then its 32 odd columns, so the
// stride-2 fragment reads of layer 1 (columns 2 c + kw) walk CONSECUTIVE 64-byte pixels of one parity; the four 16-byte channel
// chunks of pixel i sit at chunk ^ ((i >> 2) & 3): the 16 lanes of a ds_read_b128 pass then touch 16 different 16-byte bank slots
constexpr int L0E = C0 / 2 + 1, L0ROW = C0 * 64;  // even pixels per row, bytes per region row

template <typename T, typename S>
__global__ __launch_bounds__(256, 2) void stem_pair_kernel(const PairArgs p) {
    typedef typename Mfma16<T>::frag frag;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int PATCH_BYTES = XR * XW * 8, SLICE_BYTES = 4 * 32 * 64;
    constexpr int SCR = PATCH_BYTES > SLICE_BYTES ? PATCH_BYTES : SLICE_BYTES;   // the output slices reuse the patch (dead after layer 0)
    __shared__ __attribute__((aligned(16))) unsigned char scratch[SCR];
    __shared__ __attribute__((aligned(16))) unsigned char l0buf[R0 * L0ROW];
    __shared__ __attribute__((aligned(16))) unsigned char cw0[32 * 48 * 2];
    __shared__ __attribute__((aligned(16))) float cb0[32], cb1[64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = lane >> 5;
    const int wc = wv >> 1, wp = wv & 1;   // layer 1: filter tile, row pair

    // ---- per-block constants.  Layer-0 filters (3 KB) and both bias vectors live in LDS and are re-read where they are used (44 VGPRs
    // if held for the block's life; the LDS pipe has room, the register file does not) ----
    for (int i = tid; i < 32 * 48 * 2 / 16; i += 256) ((f32x4*)cw0)[i] = ((const f32x4*)p.w0)[i];
    if (tid < 8) ((f32x4*)cb0)[tid] = ((const f32x4*)p.b0)[tid];
    else if (tid < 24) ((f32x4*)cb1)[tid - 8] = ((const f32x4*)p.b1)[tid - 8];
    frag a1f[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) a1f[t] = *(const frag*)((const T*)p.w1 + (long long)(wc * 32 + frow) * p.kpad1 + t * 16 + fk * 8);

    // The kernel is bound by VALU issue (SiLU = two quarter-rate transcendentals per value; profiles/r02_stem_pair.md), so everything a
    // lane can know before the tile loop is computed here once: its patch pixels, and for each of its layer-0 MFMA tiles the patch
    // read offset, the l0buf write offset and the region coordinates.
    constexpr int NPX = (XR * XW + 255) / 256;
    int prc[NPX];            // patch pixel (row << 8 | column) of this thread's j-th element, -1 = none
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int e = tid + j * 256;
        const int pr = e / XW;
        prc[j] = e < XR * XW ? (pr << 8) | (e - pr * XW) : -1;
    }
    constexpr int NT0 = (R0 * C0 + 31) / 32, NJ0 = (NT0 + 3) / 4;   // layer-0 MFMA tiles per region, per wave
    int l0rd[NJ0], l0wr[NJ0], l0rc[NJ0];   // l0wr < 0: the lane's pixel lies beyond the region (last tile)
#pragma unroll
    for (int j = 0; j < NJ0; ++j) {
        int q = (wv + 4 * j) * 32 + frow;
        const bool live = q < R0 * C0;
        if (!live) q = R0 * C0 - 1;
        const int r = q / C0, c = q - r * C0, ci = c >> 1;
        l0rd[j] = (r * XW + c + 2 * fk) * 8;
        l0wr[j] = live ? r * L0ROW + ((c & 1) ? L0E * 64 : 0) + ci * 64 + ((fk ^ ((ci >> 2) & 3)) << 4) : -1;
        l0rc[j] = (r << 8) | c;
    }
    const int hw = p.H * p.W;

    T* __restrict__ yg = (T*)p.y;
    // image patch of a tile: every thread owns up to 3 patch pixels (x 3-4 channels).  The loads of tile i+1 are issued before
    // layer 1 of tile i runs and land in registers behind its MFMAs (a block's phases are a serial chain: without the prefetch every
    // tile started with a full HBM round trip, 11 us per tile)
    S raw[NPX][4];
    bool rin[NPX];
    auto fetch = [&](int tile) {
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int gh0 = 2 * th * QR - 2, gw0 = 2 * tw * QW - 2;   // image coordinates of patch pixel (0, 0)
        const S* __restrict__ xs = (const S*)p.x + (long long)n * p.Cin * hw;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int gh = gh0 + (prc[j] >> 8), gw = gw0 + (prc[j] & 255);
            rin[j] = prc[j] >= 0 && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            if (rin[j]) {
                const unsigned o = (unsigned)(gh * p.W + gw);   // channel planes of one image: < 2^31 elements (checked on the host)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < p.Cin) raw[j][c] = xs[o + (unsigned)(c * hw)];
            }
        }
    };
    const bool plain = std::is_same<S, T>::value && p.divisor == 1.0f;   // the source already holds T values: no convert / divide / convert
    auto stash = [&]() {   // registers -> LDS as 4-channel pixels (channel 3 = 0 unless Cin = 4), zero outside the image
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int e = tid + j * 256;
            if (e < XR * XW) {
                vec4 v = {(T)0.0f, (T)0.0f, (T)0.0f, (T)0.0f};
                if (rin[j]) {
                    bool done = false;
                    if constexpr (std::is_same<S, T>::value) {
                        if (plain) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (c < p.Cin) v[c] = raw[j][c];
                            done = true;
                        }
                    }
                    if (!done) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < p.Cin) v[c] = from_f32<T>(src_f32<S>(raw[j][c]) / p.divisor);
                    }
                }
                *(vec4*)(scratch + e * 8) = v;
            }
        }
    };
    // the tiles of a round (gridDim.x consecutive ids) are spread so that every XCD walks 64 NEIGHBOURING tiles (6.4 tile rows of the image): the halo rows and the
    // 128-byte lines that horizontally adjacent patches share are then fetched once per XCD and round instead of once per tile (dispatch order puts tile t on XCD t % 8:
    // PMC 280 MB read per batch-32 launch for a 79 MB image)
    const int first = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    if (first < p.n_tiles) fetch(first);
#ifdef Y3_TIMELINE
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = first; tile < p.n_tiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int oh0 = th * QR, ow0 = tw * QW;           // layer-1 tile origin
        const int gh0 = 2 * oh0 - 1, gw0 = 2 * ow0 - 1;   // layer-0 (= image) coordinates of region pixel (0, 0)
        const bool border = gh0 < 0 || gw0 < 0 || gh0 + R0 > p.H || gw0 + C0 > p.W;   // some region pixels are layer 1's zero padding

        stash();   // this tile's patch (requested one tile ago)
        SP_T(0);
        __syncthreads();
        SP_T(1);

        // ---- layer 0 on the 9 x 65 region (flattened, 32 pixels per MFMA tile) -> l0buf[pixel][32 channels] ----
#pragma unroll
        for (int j = 0; j < NJ0; ++j) {
            if (wv + 4 * j >= NT0) break;
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * g + e] = cb0[8 * g + 4 * fk + e];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned char* src = scratch + l0rd[j] + kh * (XW * 8);
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 rw;
                rw[0] = *(const unsigned long long*)src;
                rw[1] = *(const unsigned long long*)(src + 8);
                acc = Mfma16<T>::run(*(const frag*)(cw0 + ((frow * 3 + kh) * 16 + fk * 8) * 2), __builtin_bit_cast(frag, rw), acc);
            }
            if (p.act0 == Y3_ACT_SILU) silu_vec<f32x16, 16>(acc);
            bool keep = l0wr[j] >= 0;
            bool inside = true;
            if (border) inside = (unsigned)(gh0 + (l0rc[j] >> 8)) < (unsigned)p.H && (unsigned)(gw0 + (l0rc[j] & 255)) < (unsigned)p.W;
            // round pairs of consecutive filters to T (one v_cvt_pk per pair), THEN swap the register halves between the two lane halves:
            // a lane ends with 8 consecutive filters of its pixel per 16-byte chunk
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x4 ov;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned a = pack2<T>(acc[8 * gp + 2 * h], acc[8 * gp + 2 * h + 1]);
                    const unsigned b = pack2<T>(acc[8 * gp + 4 + 2 * h], acc[8 * gp + 4 + 2 * h + 1]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                    ov[h] = inside ? (unsigned)sw[0] : 0u;
                    ov[2 + h] = inside ? (unsigned)sw[1] : 0u;
                }
                if (keep) *(u32x4*)(l0buf + (l0wr[j] ^ (gp << 5))) = ov;
            }
        }
        SP_T(2);
        __syncthreads();
        SP_T(3);
        if (tile + (int)gridDim.x < p.n_tiles) fetch(tile + gridDim.x);   // next tile's image loads fly under layer 1

        // ---- layer 1: D[32 filters of tile wc][32 columns] for rows 2 wp and 2 wp + 1, K = 9 taps x 32 channels out of l0buf ----
        f32x16 acc1[2];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[b2][4 * g + e] = cb1[wc * 32 + 8 * g + 4 * fk + e];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * (tap / 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int lr = 2 * wp + b2;
                    const int ci = frow + (kw >> 1);
                    const frag bf = *(const frag*)(l0buf + (2 * lr + kh) * L0ROW + ((kw & 1) ? L0E * 64 : 0) + ci * 64 + (((ks * 2 + fk) ^ ((ci >> 2) & 3)) << 4));
                    acc1[b2] = Mfma16<T>::run(a1f[tap * 2 + ks], bf, acc1[b2]);
                }
        }
        SP_T(4);
        // ---- activation, 8 consecutive filters per lane, transpose through the wave's slice, 64-byte runs per pixel ----
        unsigned char* wl = scratch + wv * (32 * 64);
        const int rp = lane >> 2, ch = lane & 3;
        if (p.act1 == Y3_ACT_SILU) {
            silu_vec<f32x16, 16>(acc1[0]);
            silu_vec<f32x16, 16>(acc1[1]);
        }
        const long long ybase = ((long long)(n * p.Ho + oh0 + 2 * wp) * p.Wo + ow0) * p.ypitch + wc * 32 + ch * 8;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x4 ov;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned a = pack2<T>(acc1[b2][8 * gp + 2 * h], acc1[b2][8 * gp + 2 * h + 1]);
                    const unsigned b = pack2<T>(acc1[b2][8 * gp + 4 + 2 * h], acc1[b2][8 * gp + 4 + 2 * h + 1]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                    ov[h] = (unsigned)sw[0];
                    ov[2 + h] = (unsigned)sw[1];
                }
                const int chunk = gp * 2 + fk;
                *(u32x4*)(wl + frow * 64 + ((chunk ^ (frow & 3)) << 4)) = ov;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int oh = oh0 + 2 * wp + b2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = i * 16 + rp;
                const u32x4 ov = *(const u32x4*)(wl + pl * 64 + ((ch ^ (pl & 3)) << 4));
                if (oh < p.Ho && ow0 + pl < p.Wo) *(u32x4*)(yg + ybase + (long long)(b2 * p.Wo + pl) * p.ypitch) = ov;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next row
        }
        SP_T(5);
        __syncthreads();   // scratch (slices) and l0buf are rewritten by the next tile
        SP_T(6);
    }
#ifdef Y3_TIMELINE
    if (p.tl && blockIdx.x < 64 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.tl[(blockIdx.x * 4 + wv) * 8 + i] = tsum[i];
    }
#endif
}

// ---- Bottleneck(C, C), C = 64 or 128, in one kernel --------------------------------------------------------------------------------
// models/yolov3.yaml:18,20 (layers 2 and 4): x + cv2(cv1(x)), cv1 = Conv(C, C/2, 1, 1), cv2 = Conv(C/2, C, 3, 1), on the 320x320 / 160x160
// maps.  As two launches the C/2-channel intermediate is written and read back and x is read twice (cv1's input, cv2's residual):
// layer 2 moved 1.68 GB of HBM traffic for 0.84 GB of input + output (0.13 + 0.28 ms of a 7 ms forward at batch 32).
// Here a block owns an 8 x 32 tile of output pixels: the 10 x 34 pixel patch of x goes to LDS once by LDS-DMA (the next tile's patch
// streams in under this tile's cv2), cv1 (+ bias + SiLU, rounded to T as the stored tensor would be, zero outside the image = cv2's
// padding) runs on the 340 patch pixels into LDS, cv2 reads its nine taps out of LDS with the wave's 9 * C/32 filter fragments
// resident in registers (the layer-1 half of stem_pair with stride 1), and the residual is the centre of the patch, read out of LDS (round 3).
// Waves: (C/32 filter tiles of cv2) x (two groups of four output rows) = 4 (C = 64, two blocks per CU) or 8 (C = 128, one block per CU).
struct BneckArgs {
    const void* x;      // NHWC (N, H, W, C)
    const void* w1;     // generic packed bank of cv1 [>= C/2 rows][kpad1], k = ci
    const float* b1;    // C/2 floats
    const void* w2;     // generic packed bank of cv2 [>= C rows][kpad2], k = (kh * 3 + kw) * C/2 + ci
    const float* b2;    // C floats
    void* y;            // NHWC (N, H, W, C)
    int N, H, W, xpitch, ypitch, act1, act2, add, kpad1, kpad2;
    int tiles_w, tiles_h, n_tiles;
    int xcd;            // 1: block b starts at tile xcd_remap(b) (knob "tile_xcd")
    unsigned x_bytes;   // extent of x for the buffer descriptor of the LDS-DMA loads
};
constexpr int BC = 32, RC = BC + 2;               // output columns per tile, columns of the cv1 region
constexpr int bneck_rows(int c) { return c == 64 ? 8 : 4; }   // output rows per tile: 8 x 32 at C = 64; 4 x 32 at C = 128, where a wave's 36 resident
                                                              // filter fragments (144 VGPRs) leave room for two output rows of accumulators, not four
constexpr int L1PITCH = 40;                       // cv1-output pixels per region row in LDS: 40 keeps the row term of the chunk swizzle key a pure
                                                  // XOR on the k-step index (see c2key below)

template <typename F, int... Is> Y3_DEV void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> Y3_DEV void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <typename T, int C>
__global__ __launch_bounds__(C * 4, C == 64 ? 2 : 1) void bneck_pair_kernel(const BneckArgs p) {
    typedef typename Mfma16<T>::frag frag;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int BR = bneck_rows(C), RW = BR / 2;   // output rows per tile / per wave
    constexpr int RR = BR + 2, RPX = RR * RC;        // cv1 region: 10 x 34 = 340 or 6 x 34 = 204 pixels
    constexpr int NT1 = (RPX + 31) / 32;             // cv1 MFMA pixel tiles per region: 11 or 7
    constexpr int CM = C / 2;                     // channels of the intermediate
    constexpr int FT = C / 32;                    // filter tiles of cv2 = waves along the filters
    constexpr int NWV = 2 * FT;
    constexpr int XB = C * 2, XCHK = XB / 16;     // bytes / 16-byte chunks per x pixel: 128 / 8 or 256 / 16
    constexpr int LB = CM * 2, LCHK = LB / 16;    // the same for the intermediate: 64 / 4 or 128 / 8
    constexpr int K1S = C / 16, M1 = CM / 32;     // cv1: k-steps, filter tiles
    constexpr int KT = CM / 16;                   // cv2: k-steps per tap
    constexpr int K2S = 9 * KT;                   // cv2: k-steps = resident filter fragments per wave
    constexpr int XPIECES = (RPX * XCHK + 63) / 64;   // 1 KiB LDS-DMA pieces of the x patch (the last one partly empty)
    constexpr int XJ = (XPIECES + NWV - 1) / NWV;
    constexpr int NJ1 = (NT1 + NWV - 1) / NWV;    // cv1 pixel tiles per wave
    constexpr int QSTEP = 32 * NWV;               // pixel distance between a wave's consecutive cv1 tiles: 128 = 3 rows + 26, 256 = 7 rows + 18
    // swizzles (16 slots of 16 B per 256-B bank row): x pixel q: chunk ^ xkey(q); intermediate pixel (r, c): chunk ^ lkey(c) ^ (row parity term)
    constexpr int XKS = XCHK == 8 ? 1 : 0;        // xkey(q) = (q >> XKS) & (XCHK - 1)
    constexpr int LKS = LCHK == 4 ? 2 : 1;        // lkey(c) = (c >> LKS) & (LCHK - 1);  ((r * 40 + c) >> LKS) & (LCHK - 1) = lkey(c) ^ ((r & 1) * LCHK / 2)
    __shared__ __attribute__((aligned(16))) unsigned char xbuf[XPIECES * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char l1buf[RR * L1PITCH * LB];
    __shared__ __attribute__((aligned(16))) unsigned char slices[NWV * 32 * 64];
    __shared__ __attribute__((aligned(16))) float cb1[CM], cb2[C];
    // cv1's filters: at C = 128 (one block per CU: LDS to spare, no VGPR to spare beside cv2's 36 resident fragments) a swizzled copy in LDS,
    // row r's 16-byte chunk c at c ^ (r & 15); at C = 64 (two blocks per CU fill the LDS) re-read from L1 / L2 per tile
    constexpr bool W1_LDS = C == 128;
    __shared__ __attribute__((aligned(16))) unsigned char cw1[W1_LDS ? CM * C * 2 : 16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = lane >> 5;
    const int wc = wv >> 1, wp = wv & 1;   // cv2: filter tile, group of RW output rows

    if (tid < CM / 4) ((f32x4*)cb1)[tid] = ((const f32x4*)p.b1)[tid];
    else if (tid < CM / 4 + C / 4) ((f32x4*)cb2)[tid - CM / 4] = ((const f32x4*)p.b2)[tid - CM / 4];
    if constexpr (W1_LDS) {
        for (int i = tid; i < CM * XCHK; i += 64 * NWV) {
            const int r = i / XCHK, c = i % XCHK;
            *(u32x4*)(cw1 + r * XB + ((c ^ (r & 15)) << 4)) = *(const u32x4*)((const T*)p.w1 + (long long)r * p.kpad1 + c * 8);
        }
    }
    frag a2f[K2S];
#pragma unroll
    for (int t = 0; t < K2S; ++t) a2f[t] = *(const frag*)((const T*)p.w2 + (long long)(wc * 32 + frow) * p.kpad2 + t * 16 + fk * 8);

    // ---- per-lane constants (tile independent; everything that advances with the piece / pixel-tile index is recomputed incrementally
    // per tile: kept in registers for the block's life such tables were spilled) ----
    // x patch by LDS-DMA (`buffer_load ... lds`: the wave's 64 lanes fill 64 consecutive 16-byte slots): piece wv + NWV j = 64 / XCHK pixels,
    // lane -> pixel q = (64 / XCHK) wv + lane / XCHK + 32 j, physical chunk lane % XCHK = logical chunk ^ xkey(q) -- the key does not depend on
    // j -- and the swizzle is applied on the SOURCE address
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    // cv1: MFMA pixel tile wv + NWV j, pixel q = 32 wv + frow + QSTEP j: xkey(q) does not depend on j either
    const int c1q0 = 32 * wv + frow;
    const int c1r0 = c1q0 / RC, c1c0 = c1q0 - c1r0 * RC;
    const int c1rd0 = c1q0 * XB + ((fk ^ ((c1q0 >> XKS) & (XCHK - 1))) << 4);              // + QSTEP * XB * j; k-step ks: ^ (32 ks)
    // cv2: lane's column frow + kw, k-chunk fk of an EVEN region row: k-step ks of the tap is ^ (32 ks), odd rows ^ (16 * LCHK / 2) more
    int c2rd[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) c2rd[kw] = (frow + kw) * LB + ((fk ^ (((frow + kw) >> LKS) & (LCHK - 1))) << 4);

    T* __restrict__ yg = (T*)p.y;
    auto fetch = [&](int tile) {   // the tile's 10 x 34 pixel patch of x -> xbuf, zeros outside the image (out-of-range offsets read 0)
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int gh0 = th * BR - 1, gw0 = tw * BC - 1;
        int lane_f = lane;
        asm volatile("" : "+v"(lane_f));   // the lane constants below are recomputed per call (a handful of VALU instructions), not carried
        const int xq0 = (64 / XCHK) * wv + lane_f / XCHK;                                        // < 32 < RC: region row 0, column xq0
        const unsigned xsrc = (unsigned)(((lane_f % XCHK) ^ ((xq0 >> XKS) & (XCHK - 1))) * 16);   // byte offset of the chunk within the pixel
        int r = 0, c = xq0;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            if (wv + NWV * j >= XPIECES) break;
            const int gh = gh0 + r, gw = gw0 + c;
            const bool ok = xq0 + 32 * j < RPX && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)(((n * p.H + gh) * p.W + gw) * p.xpitch) * 2u + xsrc : 0xffffffffu;
            // by inline asm: issued through the builtin, the compiler waits for the pieces (vmcnt(0)) in front of LDS accesses that follow.
            // The counted wait is the explicit vmcnt(0) at the top of the next tile.
            const unsigned ldsa = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)xbuf + (unsigned)((wv + NWV * j) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ldsa), "v"(off), "s"(rs_x) : "memory");
            c += 32;
            if (c >= RC) { c -= RC; r += 1; }
        }
    };
    const int first = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;   // XCD-grouped tile ids: see stem_pair_kernel
    if (first < p.n_tiles) fetch(first);
    for (int tile = first; tile < p.n_tiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int oh0 = th * BR, ow0 = tw * BC;
        const int gh0 = oh0 - 1, gw0 = ow0 - 1;   // image coordinates of region pixel (0, 0)
        const bool border = gh0 < 0 || gw0 < 0 || gh0 + RR > p.H || gw0 + RC > p.W;

        frag a1g[W1_LDS ? 1 : K1S * M1];   // C = 64: cv1's filter fragments, requested here so that they arrive with the patch
        if constexpr (!W1_LDS) {
#pragma unroll
            for (int ks = 0; ks < K1S; ++ks)
#pragma unroll
                for (int m = 0; m < M1; ++m) a1g[ks * M1 + m] = *(const frag*)((const T*)p.w1 + (long long)(m * 32 + frow) * p.kpad1 + ks * 16 + fk * 8);
        }
        // this wave's pieces of the patch (requested one tile ago) have landed.  Through the builtin, not inline asm: the compiler then also
        // knows that ITS loads (the resident filter fragments) are complete and puts no counted waits for them into the K loop of cv2,
        // where they would wait for the next tile's pieces, issued just before
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __syncthreads();

        // ---- cv1 (1 x 1, K = C) on the region -> l1buf.  The wave's pixel tiles share the filter fragments of a k-step (re-read from L1 / L2 per
        // tile: 16 KB at C = 128 -- in registers for the block's life they would not leave room for cv2's) ----
        {
            f32x16 acc[NJ1][M1];
#pragma unroll
            for (int j = 0; j < NJ1; ++j)
#pragma unroll
                for (int m = 0; m < M1; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][m][4 * g + e] = cb1[m * 32 + 8 * g + 4 * fk + e];
#pragma unroll
            for (int ks = 0; ks < K1S; ++ks) {
                frag af[M1];
#pragma unroll
                for (int m = 0; m < M1; ++m) {
                    if constexpr (W1_LDS) af[m] = *(const frag*)(cw1 + (m * 32 + frow) * XB + (((2 * ks + fk) ^ (frow & 15)) << 4));
                    else af[m] = a1g[ks * M1 + m];
                }
#pragma unroll
                for (int j = 0; j < NJ1; ++j) {
                    if (wv + NWV * j >= NT1) break;
                    const frag bf = *(const frag*)(xbuf + ((c1rd0 + j * (QSTEP * XB)) ^ (ks << 5)));
#pragma unroll
                    for (int m = 0; m < M1; ++m) acc[j][m] = Mfma16<T>::run(af[m], bf, acc[j][m]);
                }
            }
            int r1 = c1r0, c1 = c1c0;
#pragma unroll
            for (int j = 0; j < NJ1; ++j) {
                if (wv + NWV * j >= NT1) break;
                const bool live = c1q0 + QSTEP * j < RPX;   // the last pixel tile is partly beyond the region: those lanes multiplied whatever follows the patch in LDS and store nothing
                bool inside = true;
                if (border) inside = (unsigned)(gh0 + r1) < (unsigned)p.H && (unsigned)(gw0 + c1) < (unsigned)p.W;
                const int wr = (r1 * L1PITCH + c1) * LB + ((fk ^ ((c1 >> LKS) & (LCHK - 1)) ^ ((r1 & 1) * (LCHK / 2))) << 4);   // chunk 4 m + 2 gp + fk -> ^ (64 m + 32 gp)
#pragma unroll
                for (int m = 0; m < M1; ++m) {
                    if (p.act1 == Y3_ACT_SILU) silu_vec<f32x16, 16>(acc[j][m]);
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        u32x4 ov;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(acc[j][m][8 * gp + 2 * h], acc[j][m][8 * gp + 2 * h + 1]),
                                                                             pack2<T>(acc[j][m][8 * gp + 4 + 2 * h], acc[j][m][8 * gp + 4 + 2 * h + 1]), false, false);
                            ov[h] = inside ? (unsigned)sw[0] : 0u;
                            ov[2 + h] = inside ? (unsigned)sw[1] : 0u;
                        }
                        if (live) *(u32x4*)(l1buf + (wr ^ ((m * 4 + gp * 2) << 4))) = ov;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                c1 += QSTEP % RC;
                r1 += QSTEP / RC;
                if (c1 >= RC) { c1 -= RC; r1 += 1; }
            }
        }
        // (round 3) the residual x[pixel][this wave's 32 channels] is the centre of the patch that is still in xbuf: taken from there, in the store
        // layout of the epilogue, BEFORE the barrier behind which the next tile's patch overwrites it -- the second read of x from HBM / L2
        // (PMC: 2.3 x the algorithmic input per launch in round 2) is gone; 2 RW registers live across cv2
        u32x4 xres[2 * RW];
        if (p.add) {
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
            const int rp_r = lane_r >> 2, lc = wc * 4 + (lane_r & 3);   // pixel within a group of 16, logical 16-byte chunk of the pixel
#pragma unroll
            for (int b2 = 0; b2 < RW; ++b2)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = (RW * wp + b2 + 1) * RC + (i * 16 + rp_r + 1);   // region pixel of output (row RW wp + b2, column 16 i + rp)
                    xres[b2 * 2 + i] = *(const u32x4*)(xbuf + q * XB + ((lc ^ ((q >> XKS) & (XCHK - 1))) << 4));
                }
        }
        __syncthreads();
        // xbuf is dead: the next tile's patch streams into it under this tile's MFMAs
        {
            int nt = tile + (int)gridDim.x;
            asm volatile("" : "+s"(nt));   // the piece addresses are computed HERE: scheduled to the top of the tile they lived in scratch until now,
                                           // and every reload's vmcnt(0) waited for the previous piece -- eleven serial round trips per tile
            if (nt < p.n_tiles) fetch(nt);
        }
        // ---- cv2: D[32 filters of tile wc][32 columns] for output rows RW wp .. RW wp + RW - 1, K = 9 taps x C/2 channels out of l1buf ----
        f32x16 acc2[RW];
#pragma unroll
        for (int b2 = 0; b2 < RW; ++b2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[b2][4 * g + e] = cb2[wc * 32 + 8 * g + 4 * fk + e];
        const unsigned char* l1w = l1buf + (RW * wp) * (L1PITCH * LB);
        // the RW fragments of k-step s + 1 are read while the RW MFMAs of step s run; scheduling fences keep it at that (left alone the
        // compiler hoisted dozens of reads and spilled the resident filter fragments to make room)
        auto rd = [&](auto step_c, frag (&bf)[RW]) {   // step = KT tap + ks
            constexpr int tap = decltype(step_c)::value / KT, ks = decltype(step_c)::value % KT, kh = tap / 3, kw = tap - 3 * (tap / 3);
#pragma unroll
            for (int b2 = 0; b2 < RW; ++b2) {
                const int rr = b2 + kh;   // region row minus RW wp (even): odd rows hold k-chunk c where even rows hold c ^ (LCHK / 2)
                bf[b2] = *(const frag*)(l1w + rr * (L1PITCH * LB) + (c2rd[kw] ^ (((2 * ks) ^ ((rr & 1) * (LCHK / 2))) << 4)));
            }
        };
        auto mm = [&](auto step_c, const frag (&bf)[RW]) {
#pragma unroll
            for (int b2 = 0; b2 < RW; ++b2) acc2[b2] = Mfma16<T>::run(a2f[decltype(step_c)::value], bf[b2], acc2[b2]);
        };
        frag bA[RW], bB[RW];
        rd(std::integral_constant<int, 0>{}, bA);
        __builtin_amdgcn_sched_barrier(0);
        sfor<K2S / 2 - 1>([&](auto i_c) {   // two k-steps per trip: the fragment buffers swap roles
            constexpr int s0 = 2 * decltype(i_c)::value;
            rd(std::integral_constant<int, s0 + 1>{}, bB);
            mm(std::integral_constant<int, s0>{}, bA);
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, s0 + 2>{}, bA);
            mm(std::integral_constant<int, s0 + 1>{}, bB);
            __builtin_amdgcn_sched_barrier(0);
        });
        rd(std::integral_constant<int, K2S - 1>{}, bB);
        mm(std::integral_constant<int, K2S - 2>{}, bA);
        __builtin_amdgcn_sched_barrier(0);
        mm(std::integral_constant<int, K2S - 1>{}, bB);
        __builtin_amdgcn_sched_barrier(0);
        // ---- activation, 8 consecutive filters per lane, transpose through the wave's slice, residual, 64-byte runs per pixel ----
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));   // (the epilogue's addresses are computed after the K loop, not carried through it)
        const int rp = lane_o >> 2, ch = lane_o & 3;
        unsigned char* wl = slices + wv * (32 * 64);
        if (p.act2 == Y3_ACT_SILU) {
#pragma unroll
            for (int b2 = 0; b2 < RW; ++b2) {
                silu_vec<f32x16, 16>(acc2[b2]);
                __builtin_amdgcn_sched_barrier(0);   // one accumulator at a time: 16 temporaries, not 64
            }
        }
        const long long ybase = ((long long)(n * p.H + oh0 + RW * wp) * p.W + ow0) * p.ypitch + wc * 32 + ch * 8;
#pragma unroll
        for (int b2 = 0; b2 < RW; ++b2) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x4 ov;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(acc2[b2][8 * gp + 2 * h], acc2[b2][8 * gp + 2 * h + 1]),
                                                                     pack2<T>(acc2[b2][8 * gp + 4 + 2 * h], acc2[b2][8 * gp + 4 + 2 * h + 1]), false, false);
                    ov[h] = (unsigned)sw[0];
                    ov[2 + h] = (unsigned)sw[1];
                }
                const int chunk = gp * 2 + fk;
                *(u32x4*)(wl + frow * 64 + ((chunk ^ (frow & 3)) << 4)) = ov;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int orow = RW * wp + b2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = i * 16 + rp;
                frag ov = *(const frag*)(wl + pl * 64 + ((ch ^ (pl & 3)) << 4));
                if (p.add) {   // x + cv2(cv1(x)): fp32 sum of the two stored values, rounded once
                    const frag xr = __builtin_bit_cast(frag, xres[b2 * 2 + i]);
                    u32x4 sum;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) sum[e >> 1] = pack2<T>(to_f32<T>(ov[e]) + to_f32<T>(xr[e]), to_f32<T>(ov[e + 1]) + to_f32<T>(xr[e + 1]));
                    ov = __builtin_bit_cast(frag, sum);
                }
                if (oh0 + orow < p.H && ow0 + pl < p.W) *(frag*)(yg + ybase + (long long)(b2 * p.W + pl) * p.ypitch) = ov;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next row
        }
        __syncthreads();   // l1buf and the slices are rewritten by the next tile
    }
}

template <typename T>
__global__ void pack_stem_kernel(const float* __restrict__ src, int cout_src, int cin_src, int rows, T* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * 48) return;
    const int k = idx % 16, kh = (idx / 16) % 3, co = idx / 48;
    const int kw = k >> 2, c = k & 3;
    float v = 0.0f;
    if (co < cout_src && kw < 3 && c < cin_src) v = src[(((long long)co * cin_src + c) * 3 + kh) * 3 + kw];
    dst[idx] = from_f32<T>(v);
}

template <typename T, typename S> int launch_stem(const StemArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)((long long)a.tiles_w * a.tiles_h * a.N));
    if (a.Cout <= 32)
        hipLaunchKernelGGL((stem_conv_kernel<T, S, 1>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((stem_conv_kernel<T, S, 2>), grid, dim3(256), 0, st, a);
    Y3_CHECK_LAUNCH();
    return 0;
}

template <typename T> int dispatch_src(const StemArgs& a, int sdt, hipStream_t st) {
    switch (sdt) {
        case Y3_F16: return launch_stem<T, f16_t>(a, st);
        case Y3_BF16: return launch_stem<T, bf16_t>(a, st);
        case Y3_F32: return launch_stem<T, float>(a, st);
        case Y3_U8: return launch_stem<T, unsigned char>(a, st);
    }
    Y3_FAIL("y3_stem_conv_fwd: bad source dtype %d", sdt);
}

}  // namespace

extern "C" size_t y3_packed_filter_stem_elems(int32_t cout) { return (size_t)((cout + 31) / 32) * 32 * 48; }

extern "C" int y3_pack_filter_stem(const float* w, int32_t cout_src, int32_t cin_src, int32_t cout, int32_t dtype, void* packed, void* stream) {
    if (!w || !packed) Y3_FAIL("y3_pack_filter_stem: null pointer");
    if (cin_src < 1 || cin_src > 4 || cout < cout_src) Y3_FAIL("y3_pack_filter_stem: needs 1..4 input channels and cout >= cout_src");
    const int rows = (cout + 31) / 32 * 32;
    const dim3 grid((unsigned)((rows * 48 + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: hipLaunchKernelGGL((pack_stem_kernel<f16_t>), grid, dim3(256), 0, st, w, cout_src, cin_src, rows, (f16_t*)packed); break;
        case Y3_BF16: hipLaunchKernelGGL((pack_stem_kernel<bf16_t>), grid, dim3(256), 0, st, w, cout_src, cin_src, rows, (bf16_t*)packed); break;
        default: Y3_FAIL("y3_pack_filter_stem: f16/bf16 only");
    }
    Y3_CHECK_LAUNCH();
    return 0;
}

static int stem_conv_impl(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed, const float* bias,
                          int32_t dtype, int32_t act, const y3_tensor* y, float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* stream, const char* who,
                          bool store = true, const float* bn_scale = nullptr, const float* bn_shift = nullptr, int32_t bn_act = Y3_ACT_NONE) {
    if (!x_nchw || !packed || !y || (store && !y->data)) Y3_FAIL("%s: null argument", who);
    if (cin < 1 || cin > 4) Y3_FAIL("%s: %d input channels (1..4 supported)", who, cin);
    if (y->n != n || y->h != h || y->w != w) Y3_FAIL("%s: output must be (%d,%d,%d,*) for a stride-1 pad-1 3x3", who, n, h, w);
    if ((y->c % 8) || y->c > 64 || (y->pitch % 8) || (store && ((uintptr_t)y->data & 15)) || ((uintptr_t)packed & 15)) Y3_FAIL("%s: 8..64 filters (multiple of 8), 16-byte aligned views", who);
    if (!(divisor > 0.0f)) Y3_FAIL("%s: divisor must be positive", who);
    if ((long long)n * h * w > 0x7fffffffLL) Y3_FAIL("%s: too many pixels", who);
    StemArgs a;
    a.x = x_nchw; a.w = packed; a.bias = bias; a.y = store ? y->data : nullptr;
    a.scale = bn_scale; a.shift = bn_shift; a.act_bn = bn_act;
    a.N = n; a.Cin = cin; a.H = h; a.W = w; a.ypitch = y->pitch; a.Cout = y->c; a.act = act;
    a.tiles_w = (w + TW - 1) / TW; a.tiles_h = (h + TR - 1) / TR;
    a.divisor = divisor;
    a.stats = stat_rows;
    const long long rows = (long long)a.tiles_w * a.tiles_h * n;
    if (stat_rows) {
        if (rows > capacity_rows) Y3_FAIL("%s: %lld statistics rows, capacity %lld", who, rows, (long long)capacity_rows);
        if (n_rows) *n_rows = rows;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: return dispatch_src<f16_t>(a, src_dtype, st);
        case Y3_BF16: return dispatch_src<bf16_t>(a, src_dtype, st);
    }
    Y3_FAIL("%s: f16/bf16 compute only", who);
}

extern "C" int y3_stem_conv_fwd(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                                const float* bias, int32_t dtype, int32_t act, const y3_tensor* y, void* stream) {
    return stem_conv_impl(x_nchw, src_dtype, n, cin, h, w, divisor, packed, bias, dtype, act, y, nullptr, 0, nullptr, stream, "y3_stem_conv_fwd");
}

extern "C" int64_t y3_stem_conv_stats_rows(int32_t n, int32_t h, int32_t w) { return (int64_t)((w + TW - 1) / TW) * ((h + TR - 1) / TR) * n; }

extern "C" int y3_stem_conv_fwd_stats(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed,
                                      const float* bias, int32_t dtype, int32_t act, const y3_tensor* y, float* stat_rows, int64_t capacity_rows, int64_t* n_rows,
                                      void* stream) {
    if (!stat_rows || !n_rows) Y3_FAIL("y3_stem_conv_fwd_stats: null statistics buffer");
    return stem_conv_impl(x_nchw, src_dtype, n, cin, h, w, divisor, packed, bias, dtype, act, y, stat_rows, capacity_rows, n_rows, stream, "y3_stem_conv_fwd_stats");
}

// statistics of layer 0's conv output WITHOUT writing it (round 6): `shape` gives (n, h, w, filters); its data pointer is not used
extern "C" int y3_stem_conv_stats_only(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed, int32_t dtype,
                                       const y3_tensor* shape, float* stat_rows, int64_t capacity_rows, int64_t* n_rows, void* stream) {
    if (!stat_rows || !n_rows) Y3_FAIL("y3_stem_conv_stats_only: null statistics buffer");
    return stem_conv_impl(x_nchw, src_dtype, n, cin, h, w, divisor, packed, nullptr, dtype, Y3_ACT_NONE, shape, stat_rows, capacity_rows, n_rows, stream, "y3_stem_conv_stats_only", false);
}

// y = act(scale u + shift) with u = the rounded conv output recomputed from the image: layer 0's normalised activation in one pass over the image
extern "C" int y3_stem_conv_fwd_bn(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed, const float* scale,
                                   const float* shift, int32_t act, int32_t dtype, const y3_tensor* y, void* stream) {
    if (!scale || !shift) Y3_FAIL("y3_stem_conv_fwd_bn: null scale / shift");
    return stem_conv_impl(x_nchw, src_dtype, n, cin, h, w, divisor, packed, nullptr, dtype, Y3_ACT_NONE, y, nullptr, 0, nullptr, stream, "y3_stem_conv_fwd_bn", true, scale, shift, act);
}

namespace {
template <typename T> int dispatch_pair(const PairArgs& a, int sdt, long long tiles, hipStream_t st) {
    // persistent: 2 blocks per CU, tiles in a grid-stride loop.  (Three blocks per CU -- 168 VGPRs, the per-lane tables spilled -- and a
    // staggered start of the blocks that share a CU were measured and dropped: the kernel is bound by VALU issue, profiles/r02_stem_pair.md)
    const int cap = 2 * y3_cu_count();
    const int blocks = tiles < cap ? (int)tiles : cap;
    switch (sdt) {
        case Y3_F16: hipLaunchKernelGGL((stem_pair_kernel<T, f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case Y3_BF16: hipLaunchKernelGGL((stem_pair_kernel<T, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case Y3_F32: hipLaunchKernelGGL((stem_pair_kernel<T, float>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        case Y3_U8: hipLaunchKernelGGL((stem_pair_kernel<T, unsigned char>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
        default: Y3_FAIL("y3_stem_pair_fwd: bad source dtype %d", sdt);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}
}  // namespace

#ifdef Y3_TIMELINE
extern "C" void y3_debug_pair_timeline(void* buf) { g_pair_tl = (unsigned long long*)buf; }
#endif

extern "C" int y3_stem_pair_fwd(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed0, const float* bias0,
                                int32_t act0, const void* packed1, const float* bias1, int32_t act1, int32_t dtype, const y3_tensor* y, void* stream) {
    if (!x_nchw || !packed0 || !bias0 || !packed1 || !bias1 || !y || !y->data) Y3_FAIL("y3_stem_pair_fwd: null argument");
    if (cin < 1 || cin > 4) Y3_FAIL("y3_stem_pair_fwd: %d input channels (1..4 supported)", cin);
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;   // 3x3, stride 2, pad 1
    if (y->n != n || y->h != Ho || y->w != Wo || y->c != 64) Y3_FAIL("y3_stem_pair_fwd: output must be (%d,%d,%d,64)", n, Ho, Wo);
    if ((y->pitch % 8) || ((uintptr_t)y->data & 15) || ((uintptr_t)packed0 & 15) || ((uintptr_t)packed1 & 15) || ((uintptr_t)bias0 & 15) || ((uintptr_t)bias1 & 15))
        Y3_FAIL("y3_stem_pair_fwd: 16-byte aligned views");
    if (!(divisor > 0.0f)) Y3_FAIL("y3_stem_pair_fwd: divisor must be positive");
    if ((long long)n * h * w > 0x7fffffffLL) Y3_FAIL("y3_stem_pair_fwd: too many pixels");
    PairArgs a;
    a.x = x_nchw; a.w0 = packed0; a.b0 = bias0; a.w1 = packed1; a.b1 = bias1; a.y = y->data;
    a.N = n; a.Cin = cin; a.H = h; a.W = w; a.Ho = Ho; a.Wo = Wo; a.ypitch = y->pitch; a.act0 = act0; a.act1 = act1;
    a.kpad1 = y3_filter_kpad(32, 3);
    a.tiles_w = (Wo + QW - 1) / QW; a.tiles_h = (Ho + QR - 1) / QR;
    const long long tiles = (long long)a.tiles_w * a.tiles_h * n;
    if (tiles > 0x7fffffffLL) Y3_FAIL("y3_stem_pair_fwd: too many tiles");
    a.n_tiles = (int)tiles;
    a.xcd = y3_knob(Y3K_TILE_XCD) != 0;
    a.divisor = divisor;
#ifdef Y3_TIMELINE
    a.tl = g_pair_tl;
#endif
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: return dispatch_pair<f16_t>(a, src_dtype, tiles, st);
        case Y3_BF16: return dispatch_pair<bf16_t>(a, src_dtype, tiles, st);
    }
    Y3_FAIL("y3_stem_pair_fwd: f16/bf16 compute only");
}

extern "C" int y3_bneck_pair_fwd(const y3_tensor* x, const void* packed1, const float* bias1, int32_t act1, const void* packed2, const float* bias2, int32_t act2,
                                 int32_t add_residual, int32_t dtype, const y3_tensor* y, void* stream) {
    if (!x || !x->data || !packed1 || !bias1 || !packed2 || !bias2 || !y || !y->data) Y3_FAIL("y3_bneck_pair_fwd: null argument");
    const int c = x->c;
    if ((c != 64 && c != 128) || y->c != c) Y3_FAIL("y3_bneck_pair_fwd: Bottleneck(64, 64) or Bottleneck(128, 128) (cv1 C -> C/2 1x1, cv2 C/2 -> C 3x3), got %d -> %d channels", x->c, y->c);
    if (y->n != x->n || y->h != x->h || y->w != x->w) Y3_FAIL("y3_bneck_pair_fwd: output shape");
    if ((x->pitch % 8) || (y->pitch % 8) || ((uintptr_t)x->data & 15) || ((uintptr_t)y->data & 15) || ((uintptr_t)packed1 & 15) || ((uintptr_t)packed2 & 15) ||
        ((uintptr_t)bias1 & 15) || ((uintptr_t)bias2 & 15))
        Y3_FAIL("y3_bneck_pair_fwd: 16-byte aligned views");
    if (x->data == y->data) Y3_FAIL("y3_bneck_pair_fwd: in-place operation is not supported (tiles read their neighbours' input)");
    BneckArgs a;
    a.x = x->data; a.w1 = packed1; a.b1 = bias1; a.w2 = packed2; a.b2 = bias2; a.y = y->data;
    a.N = x->n; a.H = x->h; a.W = x->w; a.xpitch = x->pitch; a.ypitch = y->pitch; a.act1 = act1; a.act2 = act2; a.add = add_residual ? 1 : 0;
    a.kpad1 = y3_filter_kpad(c, 1); a.kpad2 = y3_filter_kpad(c / 2, 3);
    a.tiles_w = (x->w + BC - 1) / BC; a.tiles_h = (x->h + bneck_rows(c) - 1) / bneck_rows(c);
    const long long tiles = (long long)a.tiles_w * a.tiles_h * x->n;
    if (tiles > 0x7fffffffLL || (long long)x->n * x->h * x->w > 0x7fffffffLL) Y3_FAIL("y3_bneck_pair_fwd: too many pixels");
    a.n_tiles = (int)tiles;
    a.xcd = y3_knob(Y3K_TILE_XCD) != 0;
    const long long xb = (((long long)x->n * x->h * x->w - 1) * x->pitch + x->c) * 2;
    if (xb >= 0x7fffffffLL) Y3_FAIL("y3_bneck_pair_fwd: input beyond the 2 GiB reach of a buffer descriptor (split the batch)");
    a.x_bytes = (unsigned)xb;
    const int cap = (c == 64 ? 2 : 1) * y3_cu_count();   // persistent: 2 blocks of 4 waves / 1 block of 8 waves per CU
    const int blocks = tiles < cap ? (int)tiles : cap;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != Y3_F16 && dtype != Y3_BF16) Y3_FAIL("y3_bneck_pair_fwd: f16/bf16 only");
    if (c == 64) {
        if (dtype == Y3_F16) hipLaunchKernelGGL((bneck_pair_kernel<f16_t, 64>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bneck_pair_kernel<bf16_t, 64>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        if (dtype == Y3_F16) hipLaunchKernelGGL((bneck_pair_kernel<f16_t, 128>), dim3((unsigned)blocks), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bneck_pair_kernel<bf16_t, 128>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}
