// Detect decode + batched non_max_suppression for gfx950.  Built with -ffp-contract=off: every float op here
// must round exactly like the reference's CPU path (no fused multiply-adds).
//
// Decode   : reference models/yolo.py:98-110.  HBM-bound streaming kernel.
// NMS      : reference utils/general.py:630-750 + torchvision.ops.nms.  Pipeline (no host sync anywhere):
//   1 count    wave per 64 anchor rows: obj filter by ballot, per-row label count (multi-label / best class)
//   2 scan     nms_scan_kernel (block per image; nms_sort = 0: rocprim exclusive scan) of the per-row counts -> ordered (nonzero-order) emission slots
//   3 emit     candidates: xyxy box (input-dtype rounding), fp32 score, class; key1 = [img | ~score | ordinal]
//   4 order    nms_sort_kernel, one block per image (round 6; knob nms_sort = 0: steps 4-6 of rounds 1-5 -- rocprim radix sort by key1, nms_rank_kernel, rocprim radix
//              sort by key2 = [img | class | rank]): stable counting passes by the score bits -> per-image descending-score order (ties keep nonzero order), cut at
//              max_nms, class-offset boxes (+cls*max_wh, fp32), one more stable pass by class -> per (image, class) segments in score order.  Boxes of different
//              classes cannot overlap after the class offset when every |coord| < max_wh/2, so greedy NMS factorises per class; an image that violates the bound
//              (or agnostic mode) is ONE segment.
//   5 greedy   block per (image, class): visit in order; a kept box suppresses later ones with
//              inter/(a_i+a_j-inter) > thr (strict, compared in double like torchvision's CPU kernel);
//              IoU rows are evaluated lazily for kept boxes only; stop after max_det kept per segment.
//   6 gather   block per image: kept flags back in score order, first max_det rows -> (bs, max_det, 6) fp32.
#include "y3_common.h"

#include <rocprim/rocprim.hpp>

namespace {

// ------------------------------------------------------------------------------------------------ decode
template <typename T>
Y3_DEV float decode_one(float v, int o, int x, int y, float stride, float aw, float ah);

template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(const T* __restrict__ head, int bs, int ny, int nx, int pitch, int na, int no, float aw0, float ah0,
                                                       float aw1, float ah1, float aw2, float ah2, float aw3, float ah3, float aw4, float ah4, float stride,
                                                       T* __restrict__ raw, T* __restrict__ z, long long row_offset, long long total_rows) {
    const int nch = na * no;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)bs * ny * nx * nch;
    if (idx >= total) return;
    const int ch = (int)(idx % nch);
    long long t = idx / nch;
    const int x = (int)(t % nx);
    t /= nx;
    const int y = (int)(t % ny);
    const int b = (int)(t / ny);
    const int a = ch / no, o = ch - a * no;
    const T hv = head[((long long)(b * ny + y) * nx + x) * pitch + ch];
    const long long cell = ((long long)(b * na + a) * ny + y) * nx + x;  // (bs,na,ny,nx) index
    if (raw) raw[cell * no + o] = hv;
    if (z) {
        const float aw = a == 0 ? aw0 : a == 1 ? aw1 : a == 2 ? aw2 : a == 3 ? aw3 : aw4;
        const float ah = a == 0 ? ah0 : a == 1 ? ah1 : a == 2 ? ah2 : a == 3 ? ah3 : ah4;
        const float r = decode_one<T>(to_f32<T>(hv), o, x, y, stride, aw, ah);
        const long long row = row_offset + ((long long)a * ny + y) * nx + x;
        z[((long long)b * total_rows + row) * no + o] = from_f32<T>(r);
    }
}

// One decoded element: the reference's op order with a round-to-T after every op (models/yolo.py:104-108 run in T).
template <typename T>
Y3_DEV float decode_one(float v, int o, int x, int y, float stride, float aw, float ah) {
    const float s = rt<T>(1.0f / (1.0f + expf(-v)));
    if (o < 2) {
        const float g = rt<T>((o == 0 ? (float)x : (float)y) - 0.5f);
        return rt<T>(rt<T>(rt<T>(s * 2.0f) + g) * rt<T>(stride));
    }
    if (o < 4) {
        const float d = rt<T>(s * 2.0f);
        return rt<T>(rt<T>(d * d) * (o == 2 ? aw : ah));
    }
    return s;
}

// 2-byte dtypes, 16-byte aligned blocks: within one (image, anchor) block both outputs are ONE contiguous run of ny*nx*no
// elements (raw: (bs,na,ny,nx,no); z: rows [row_offset + a*ny*nx, +ny*nx) of (bs,total_rows,no)), so a thread produces 8
// consecutive elements = one 16-byte store per output; the matching head elements (pixel p, channel a*no + o) are at most
// two pixels' runs.  The element-per-thread kernel above ran at 1.1 TB/s (2-byte accesses, three 64-bit divisions each).
template <typename T>
__global__ __launch_bounds__(256) void decode_vec_kernel(const T* __restrict__ head, int bs, int ny, int nx, int pitch, int na, int no, float aw0, float ah0,
                                                           float aw1, float ah1, float aw2, float ah2, float aw3, float ah3, float aw4, float ah4, float stride,
                                                           T* __restrict__ raw, T* __restrict__ z, long long row_offset, long long total_rows) {
    const int P = ny * nx;
    const int chunks = P * no / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)bs * na * chunks) return;
    const int ba = (int)(idx / chunks), chunk = (int)(idx - (long long)ba * chunks);
    const int b = ba / na, a = ba - b * na;
    const int f0 = chunk * 8;
    int pix = f0 / no, o = f0 - pix * no;
    int y = pix / nx, x = pix - y * nx;
    const float aw = a == 0 ? aw0 : a == 1 ? aw1 : a == 2 ? aw2 : a == 3 ? aw3 : aw4;
    const float ah = a == 0 ? ah0 : a == 1 ? ah1 : a == 2 ? ah2 : a == 3 ? ah3 : ah4;
    const T* hp = head + ((long long)b * P + pix) * pitch + a * no;
    typedef T vec8 __attribute__((ext_vector_type(8)));
    vec8 rv, zv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const T hv = hp[o];
        rv[e] = hv;
        zv[e] = from_f32<T>(decode_one<T>(to_f32<T>(hv), o, x, y, stride, aw, ah));
        if (++o == no) {   // next pixel (no >= 8: at most once per thread)
            o = 0;
            hp += pitch;
            if (++x == nx) { x = 0; ++y; }
        }
    }
    const long long f = (long long)ba * P * no + f0;
    if (raw) *(vec8*)(raw + f) = rv;
    if (z) *(vec8*)(z + ((long long)b * total_rows + row_offset + (long long)a * P) * no + f0) = zv;
}

// ------------------------------------------------------------------------------------------------ NMS
constexpr int IMG_BITS = 11, ORD_BITS = 21, CLS_BITS = 12, RANK_BITS = 15;
constexpr unsigned long long KEY_UNUSED = ~0ull;

struct NmsWs {  // device pointers carved from the caller's workspace
    int* row_count;                // bs*n_rows (+1)
    int* row_off;                  // bs*n_rows + 1 (exclusive scan, last = total)
    unsigned* img_maxabs;          // bs  (float bits of max |coord| among the image's candidates)
    unsigned* img_total;           // bs * 64: candidates of the image, spread over 64 slots by wave id (count pass; own scan) -- one slot per image serialised 1575 atomics on one address
    unsigned long long* key_a;     // cap
    unsigned long long* key_b;     // cap
    unsigned* val_a;               // cap
    unsigned* val_b;               // cap
    unsigned long long* key_c;     // cap  (sorted key2)
    unsigned* val_c;               // cap  (sorted-1 positions in segment order)
    float4* cbox;                  // cap  emitted xyxy (no class offset)
    float* cscore;                 // cap
    int* ccls;                     // cap
    float4* sbox;                  // cap  class-offset boxes in sorted-1 order
    unsigned char* keep;           // cap  indexed by sorted-1 position
    void* tmp;                     // rocprim temp storage
    size_t tmp_bytes;
    long long cap;
    int own_sort;                  // 1: nms_sort_kernel orders the candidates (key_a / key_b / key_c / val_a are its scratch); 0: the rocPRIM sorts on key_a / val_a
};

template <typename T> Y3_DEV bool gt_thr(float v, float thr_T) { return v > thr_T; }

Y3_DEV bool class_allowed(int c, const int* __restrict__ classes, int ncf) {
    if (ncf == 0) return true;
    bool ok = false;
    for (int i = 0; i < ncf; ++i) ok |= (classes[i] == c);
    return ok;
}

// One wave per 64 consecutive anchor rows of one image.  MODE 0 = count, MODE 1 = emit.
// The rows that pass the objectness test are walked in row order.  multi_label with nc <= 128 (the benchmarked load: ~13 of a wave's 64 rows pass) takes them
// NMS_ROW_BATCH at a time: the class scores (two loads per lane and row) and the box of every row of the batch are requested before the first ballot, so a wave waits
// for memory once per batch instead of twice per row -- the one-row loop was latency-bound (75 / 123 us for the two passes over 137 MB).  Same order, same arithmetic.
constexpr int NMS_ROW_BATCH = 8;
template <typename T, int MODE>
__global__ __launch_bounds__(256) void nms_candidates_kernel(const T* __restrict__ pred, int bs, int n_rows, int nc, float thr, int multi_label,
                                                               const int* __restrict__ classes, int ncf, NmsWs ws, int* __restrict__ status, int ord_shift) {
    const int lane = threadIdx.x & 63;
    const int chunks_per_img = (n_rows + 63) / 64;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_id >= (long long)bs * chunks_per_img) return;
    const int img = (int)(wave_id / chunks_per_img);
    const int r0 = (int)(wave_id % chunks_per_img) * 64;
    const int no = nc + 5;
    const T* __restrict__ base = pred + (long long)img * n_rows * no;

    const int myrow = r0 + lane;
    float obj_l = 0.0f;
    if (myrow < n_rows) obj_l = to_f32<T>(base[(long long)myrow * no + 4]);
    const bool pass_l = (myrow < n_rows) && (obj_l > thr);
    unsigned long long mask = __ballot(pass_l);
    int mycount = 0;
    int myoff = 0;
    if (MODE == 1 && myrow < n_rows) myoff = ws.row_off[(long long)img * n_rows + myrow];
    const long long img_off0 = MODE == 1 ? (long long)ws.row_off[(long long)img * n_rows] : 0;
    float maxabs = 0.0f;

    if (multi_label && nc <= 128) {
        const bool in0 = lane < nc, in1 = 64 + lane < nc;
        const bool ok0 = in0 && class_allowed(lane, classes, ncf), ok1 = in1 && class_allowed(64 + lane, classes, ncf);
        while (mask) {
            int rr[NMS_ROW_BATCH];
            float v0[NMS_ROW_BATCH], v1[NMS_ROW_BATCH], bx[NMS_ROW_BATCH][4];
#pragma unroll
            for (int q = 0; q < NMS_ROW_BATCH; ++q) {   // (wave-uniform: the mask is a ballot)
                rr[q] = mask ? __builtin_ctzll(mask) : -1;
                mask &= mask - 1;                        // (0 stays 0)
            }
#pragma unroll
            for (int q = 0; q < NMS_ROW_BATCH; ++q) {
                // no branch around a load (a predicated load ends in its own s_waitcnt vmcnt(0): the batch would be serial again): an unused slot re-reads the
                // batch's first row, a lane beyond nc the row's first class -- valid addresses, values dropped by in0 / in1 / rr below
                const T* __restrict__ rp = base + (long long)(r0 + (rr[q] >= 0 ? rr[q] : rr[0])) * no;
                v0[q] = to_f32<T>(rp[5 + (in0 ? lane : 0)]);
                v1[q] = to_f32<T>(rp[5 + (in1 ? 64 + lane : 0)]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bx[q][i] = MODE == 1 ? to_f32<T>(rp[i]) : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < NMS_ROW_BATCH; ++q) {
                if (rr[q] < 0) break;
                const int r = rr[q];
                const float obj = __shfl(obj_l, r);
                int cnt = 0, off = 0;
                float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 1) {
                    off = __shfl(myoff, r);
                    const float hw = rt<T>(bx[q][2] / 2.0f), hh = rt<T>(bx[q][3] / 2.0f);   // xywh2xyxy in the input dtype (utils/general.py:705)
                    box = make_float4(rt<T>(bx[q][0] - hw), rt<T>(bx[q][1] - hh), rt<T>(bx[q][0] + hw), rt<T>(bx[q][1] + hh));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && nc <= 64) break;
                    const int c = h * 64 + lane;
                    const float conf = (h ? in1 : in0) ? rt<T>((h ? v1[q] : v0[q]) * obj) : 0.0f;   // x[:, 5:] *= x[:, 4:5] in the input dtype  (:702)
                    const bool ok = (h ? ok1 : ok0) && (conf > thr);
                    const unsigned long long m = __ballot(ok);
                    if (MODE == 1 && ok) {
                        const int k = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                        const long long g = (long long)off + k;
                        if (g < ws.cap) {
                            const unsigned ord = (unsigned)((g - img_off0) >> ord_shift);
                            ws.cbox[g] = box;
                            ws.cscore[g] = conf;
                            ws.ccls[g] = c;
                            if (!ws.own_sort) {
                                ws.key_a[g] = ((unsigned long long)img << (32 + ORD_BITS)) | ((unsigned long long)(~__float_as_uint(conf)) << ORD_BITS) |
                                              (unsigned long long)(ord & ((1u << ORD_BITS) - 1u));
                                ws.val_a[g] = (unsigned)g;
                            }
                        } else {
                            status[0] = 1;
                        }
                    }
                    cnt += __builtin_popcountll(m);
                }
                if (MODE == 1 && cnt > 0) {
                    const float m4 = fmaxf(fmaxf(fabsf(box.x), fabsf(box.y)), fmaxf(fabsf(box.z), fabsf(box.w)));
                    maxabs = (m4 == m4) ? fmaxf(maxabs, m4) : INFINITY;  // NaN -> forbid class partitioning
                }
                if (lane == r) mycount = cnt;
            }
        }
    }
    while (mask) {   // (every other case: one row at a time)
        const int r = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int row = r0 + r;
        const T* __restrict__ rp = base + (long long)row * no;
        const float obj = __shfl(obj_l, r);
        int cnt = 0;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        int off = 0;
        if (MODE == 1) {
            off = __shfl(myoff, r);
            // xywh2xyxy in the input dtype: half = wh/2 (rounded), xy -/+ half (rounded)   utils/general.py:705
            const float cx = to_f32<T>(rp[0]), cy = to_f32<T>(rp[1]);
            const float hw = rt<T>(to_f32<T>(rp[2]) / 2.0f), hh = rt<T>(to_f32<T>(rp[3]) / 2.0f);
            box = make_float4(rt<T>(cx - hw), rt<T>(cy - hh), rt<T>(cx + hw), rt<T>(cy + hh));
        }
        if (multi_label) {
            for (int c0 = 0; c0 < nc; c0 += 64) {
                const int c = c0 + lane;
                float conf = 0.0f;
                bool ok = false;
                if (c < nc) {
                    conf = rt<T>(to_f32<T>(rp[5 + c]) * obj);  // x[:, 5:] *= x[:, 4:5] in the input dtype  (:702)
                    ok = (conf > thr) && class_allowed(c, classes, ncf);
                }
                const unsigned long long m = __ballot(ok);
                if (MODE == 1 && ok) {
                    const int k = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    const long long g = (long long)off + k;
                    if (g < ws.cap) {
                        const unsigned ord = (unsigned)((g - img_off0) >> ord_shift);
                        ws.cbox[g] = box;
                        ws.cscore[g] = conf;
                        ws.ccls[g] = c;
                        if (!ws.own_sort) {
                            ws.key_a[g] = ((unsigned long long)img << (32 + ORD_BITS)) | ((unsigned long long)(~__float_as_uint(conf)) << ORD_BITS) |
                                          (unsigned long long)(ord & ((1u << ORD_BITS) - 1u));
                            ws.val_a[g] = (unsigned)g;
                        }
                    } else {
                        status[0] = 1;
                    }
                }
                cnt += __builtin_popcountll(m);
            }
        } else {
            // best class: max over classes, first index on ties (:713), then conf > thr (:714)
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = lane; c < nc; c += 64) {
                const float conf = rt<T>(to_f32<T>(rp[5 + c]) * obj);
                if (conf > best) { best = conf; bi = c; }
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                const float ob = __shfl_xor(best, s);
                const int oi = __shfl_xor(bi, s);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const bool ok = (best > thr) && class_allowed(bi, classes, ncf);
            cnt = ok ? 1 : 0;
            if (MODE == 1 && ok && lane == 0) {
                const long long g = off;
                if (g < ws.cap) {
                    const unsigned ord = (unsigned)((g - img_off0) >> ord_shift);
                    ws.cbox[g] = box;
                    ws.cscore[g] = best;
                    ws.ccls[g] = bi;
                    if (!ws.own_sort) {
                        ws.key_a[g] = ((unsigned long long)img << (32 + ORD_BITS)) | ((unsigned long long)(~__float_as_uint(best)) << ORD_BITS) |
                                      (unsigned long long)(ord & ((1u << ORD_BITS) - 1u));
                        ws.val_a[g] = (unsigned)g;
                    }
                } else {
                    status[0] = 1;
                }
            }
        }
        if (MODE == 1 && cnt > 0) {
            const float m4 = fmaxf(fmaxf(fabsf(box.x), fabsf(box.y)), fmaxf(fabsf(box.z), fabsf(box.w)));
            maxabs = (m4 == m4) ? fmaxf(maxabs, m4) : INFINITY;  // NaN -> forbid class partitioning
        }
        if (lane == r) mycount = cnt;
    }
    if (MODE == 0) {
        if (myrow < n_rows) ws.row_count[(long long)img * n_rows + myrow] = mycount;
        if (ws.own_sort) {   // the image's total for nms_scan_kernel (integer adds: any order, same sum)
            int t = mycount;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
            if (lane == 0 && t > 0) atomicAdd(&ws.img_total[img * 64 + (int)(wave_id & 63)], (unsigned)t);
        }
    } else {
        if (lane == 0 && maxabs > 0.0f) atomicMax(&ws.img_maxabs[img], __float_as_uint(maxabs));
    }
}

// exclusive scan of the per-row label counts -> emission slots (round 6; rounds 1-5: rocprim::exclusive_scan, two launches).  One block per image: the image's first
// slot = the totals of the images before it (the count pass adds every wave's count to one of its image's 64 total slots), then per chunk of 1024 x 33 rows: counts
// to LDS with coalesced loads, every thread scans ITS 33 consecutive rows there (odd stride: no bank conflicts), one block scan of the threads' sums, offsets back
// through LDS with coalesced stores.  row_off[bs * n_rows] = the number of candidates of the batch.
constexpr int SCAN_PER = 33, SCAN_CHUNK = 1024 * SCAN_PER;
__global__ __launch_bounds__(1024) void nms_scan_kernel(NmsWs ws, int bs, int n_rows) {
    __shared__ unsigned buf[SCAN_CHUNK];
    __shared__ unsigned wsum[16];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto block_sum_scan = [&](unsigned v, unsigned& excl, unsigned& total) {   // exclusive prefix of v over the block's threads, and the block's sum
        unsigned inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        __syncthreads();   // (wsum free; buf written)
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned t = wsum[w];
            if (w < wv) before += t;
            all += t;
        }
        excl = before + inc - v;
        total = all;
    };
    unsigned prev = 0;
    for (int j = tid; j < img * 64; j += 1024) prev += ws.img_total[j];
    unsigned e0, carry;
    block_sum_scan(prev, e0, carry);
    const int* __restrict__ cnt = ws.row_count + (long long)img * n_rows;
    int* __restrict__ off = ws.row_off + (long long)img * n_rows;
    for (int c0 = 0; c0 < n_rows; c0 += SCAN_CHUNK) {
        const int n = n_rows - c0 < SCAN_CHUNK ? n_rows - c0 : SCAN_CHUNK;
        __syncthreads();   // (buf free)
        for (int i = tid; i < n; i += 1024) buf[i] = (unsigned)cnt[c0 + i];
        __syncthreads();
        const int r0 = tid * SCAN_PER, r1 = r0 + SCAN_PER < n ? r0 + SCAN_PER : n;
        unsigned mine = 0;
        for (int r = r0; r < r1; ++r) mine += buf[r];
        unsigned excl, total;
        block_sum_scan(mine, excl, total);
        unsigned run = carry + excl;
        for (int r = r0; r < r1; ++r) {
            const unsigned c = buf[r];
            buf[r] = run;
            run += c;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 1024) off[c0 + i] = (int)buf[i];
        carry += total;
    }
    if (img == bs - 1 && tid == 0) ws.row_off[(long long)bs * n_rows] = (int)carry;
}

// sorted-1 order -> per-image rank, max_nms cut, class-offset boxes, key2
__global__ __launch_bounds__(256) void nms_rank_kernel(NmsWs ws, int bs, int n_rows, int max_nms, float max_wh, int agnostic, const unsigned long long* __restrict__ key1,
                                                        const unsigned* __restrict__ idx1, unsigned long long* __restrict__ key2, unsigned* __restrict__ val2,
                                                        int* __restrict__ status) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ws.cap) return;
    const long long total = ws.row_off[(long long)bs * n_rows];
    if (i == 0) status[1] = (int)(total > 0x7fffffffLL ? 0x7fffffff : total);
    unsigned long long k2 = KEY_UNUSED;
    ws.keep[i] = 0;
    if (i < total && key1[i] != KEY_UNUSED) {
        const int img = (int)(key1[i] >> (32 + ORD_BITS));
        const long long start = ws.row_off[(long long)img * n_rows];
        const long long rank = i - start;
        if (rank < max_nms) {
            const unsigned g = idx1[i];
            const int cls = ws.ccls[g];
            const float4 b = ws.cbox[g];
            const bool part = !agnostic && (__uint_as_float(ws.img_maxabs[img]) < max_wh * 0.5f);
            const float c = (float)cls * (agnostic ? 0.0f : max_wh);  // x[:, 5:6] * (0 if agnostic else max_wh)  (:731)
            ws.sbox[i] = make_float4(b.x + c, b.y + c, b.z + c, b.w + c);
            const unsigned seg = part ? (unsigned)cls : 0u;
            k2 = ((unsigned long long)img << (CLS_BITS + RANK_BITS)) | ((unsigned long long)seg << RANK_BITS) | (unsigned long long)rank;
        }
    }
    key2[i] = k2;
    val2[i] = (unsigned)i;
}

// ---- the two orderings of the pipeline by ONE launch (round 6; steps 4-6 of the list at the top) ----------------------------------------------------
// Rounds 1-5 ran two device-wide rocprim::radix_sort_pairs over the whole candidate capacity (64-bit keys; at ~0.4 M elements rocPRIM takes its merge path: a block
// sort + eight merge launches per sort, ~0.42 of the 0.72 ms NMS leg).  Neither ordering is device-wide: the candidates of an image are contiguous and already in
// nonzero order (the emission slots come from a scan), the first ordering is "stable by descending score inside the image", the second "stable by class inside the
// image's first max_nms".  One block of 16 waves per image does both as LSD counting passes over its own range:
//   * a pass = (A) per-wave digit histograms in LDS, (B) exclusive prefix over (digit, wave) -- wave w's elements of digit d go behind those of waves < w --,
//     (C) every wave walks ITS contiguous slice in order, 64 elements at a time: the lanes holding the same digit find each other with one ballot per digit bit
//     (peers), a lane's slot is the wave's running count of the digit + the peers below it, the highest peer advances the count.  Stable by construction, no
//     atomics with a result, no ordinal in the key;
//   * score: the bits of conf that can differ in the tensor dtype (a value rounded to f16 / bf16 has 18 significant bits below the sign: two 9-bit passes;
//     fp32: four 8-bit passes), complemented, so ascending digits = descending score;
//   * class: one pass of ceil(log2(nseg)) <= 9 bits (two for more than 512 segments) over the image's first max_nms positions, together with what nms_rank_kernel
//     did (class-offset boxes in score order, cleared keep flags).  An image that cannot be partitioned (agnostic, or a coordinate beyond max_wh / 2) is ONE segment.
// The block reads what it wrote in the previous pass through agent-scope loads (L2): no reliance on how this CU's vector L1 treats its own stores.
constexpr int SORT_WAVES = 16, SORT_NB = 512;
struct SortLds {
    unsigned hist[SORT_WAVES * SORT_NB];   // [wave][digit]: counts (A), then the wave's running offset inside the digit (B, C)
    unsigned tot[SORT_NB];                 // per digit: count, then the digit's first slot
};
Y3_DEV unsigned ld_l2(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename Load, typename Digit, typename Put>
Y3_DEV void counting_pass(SortLds& L, const int cnt, const int bits, Load load, Digit digit, Put put) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = 1 << bits;
    for (int i = tid; i < SORT_WAVES * nb; i += SORT_WAVES * 64) L.hist[(i >> bits) * SORT_NB + (i & (nb - 1))] = 0u;
    __syncthreads();
    const int per = (((cnt + SORT_WAVES - 1) / SORT_WAVES) + 63) & ~63;   // a wave's slice: whole groups of 64, consecutive elements
    const int lo = wv * per, hi = cnt < lo + per ? cnt : lo + per;
    // (A)
    for (int c0 = lo; c0 < hi; c0 += 256) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // the loads of four groups in flight together (an index beyond the slice re-reads its first element: dropped below)
            const int r = c0 + 64 * u + lane;
            e[u] = load(r < hi ? r : lo);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = c0 + 64 * u + lane;
            if (r < hi) atomicAdd(&L.hist[wv * SORT_NB + digit(e[u].x)], 1u);
        }
    }
    __syncthreads();
    // (B)
    if (tid < nb) {
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const unsigned t = L.hist[w * SORT_NB + tid];
            L.hist[w * SORT_NB + tid] = run;
            run += t;
        }
        L.tot[tid] = run;
    }
    __syncthreads();
    if (wv == 0) {   // exclusive prefix of the digit totals: 8 consecutive digits per lane
        unsigned v[SORT_NB / 64], sum = 0;
#pragma unroll
        for (int k = 0; k < SORT_NB / 64; ++k) {
            const int d = lane * (SORT_NB / 64) + k;
            v[k] = d < nb ? L.tot[d] : 0u;
            sum += v[k];
        }
        unsigned inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        unsigned run = inc - sum;
#pragma unroll
        for (int k = 0; k < SORT_NB / 64; ++k) {
            const int d = lane * (SORT_NB / 64) + k;
            if (d < nb) L.tot[d] = run;
            run += v[k];
        }
    }
    __syncthreads();
    // (C)
    volatile unsigned* H = L.hist + wv * SORT_NB;   // (volatile: group u + 1 reads what group u wrote, in program order)
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c0 = lo; c0 < hi; c0 += 256) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = c0 + 64 * u + lane;
            e[u] = load(r < hi ? r : lo);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + 64 * u >= hi) break;   // (uniform)
            const int r = c0 + 64 * u + lane;
            const bool valid = r < hi;
            const unsigned d = digit(e[u].x);
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < bits; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const unsigned base = H[d];
                put(L.tot[d] + base + (unsigned)__builtin_popcountll(peers & below), e[u].x, e[u].y);
                if ((peers >> lane) == 1ull) H[d] = base + (unsigned)__builtin_popcountll(peers);   // the highest lane of the digit
            }
        }
    }
    __syncthreads();   // the slots are written (workgroup scope); L is free
}

__global__ __launch_bounds__(SORT_WAVES * 64) void nms_sort_kernel(NmsWs ws, int bs, int n_rows, int max_nms, float max_wh, int agnostic, int nseg, int sbit0, int sbits,
                                                                     int n_sp, int* __restrict__ status) {
    __shared__ SortLds L;
    const int img = blockIdx.x, tid = threadIdx.x;
    const long long total = ws.row_off[(long long)bs * n_rows];
    if (img == 0 && tid == 0) status[1] = (int)(total > 0x7fffffffLL ? 0x7fffffff : total);
    const long long start = ws.row_off[(long long)img * n_rows];
    long long cnt_ll = (long long)ws.row_off[(long long)(img + 1) * n_rows] - start;
    if (start + cnt_ll > ws.cap) cnt_ll = ws.cap - start > 0 ? ws.cap - start : 0;   // (overflow: status[0] is set by the emit pass, the caller grows the workspace)
    const int cnt = (int)cnt_ll;
    if (cnt <= 0) return;   // (uniform)
    // scratch, 32-bit views: key_a = X0 | X1, key_b = X2 | X3, key_c = X4 (the segment id per slot of the final order) | X5
    unsigned* const X0 = (unsigned*)ws.key_a + start, * const X1 = (unsigned*)ws.key_a + ws.cap + start;
    unsigned* const X2 = (unsigned*)ws.key_b + start, * const X3 = (unsigned*)ws.key_b + ws.cap + start;
    unsigned* const X4 = (unsigned*)ws.key_c + start, * const X5 = (unsigned*)ws.key_c + ws.cap + start;
    unsigned* const VA = ws.val_a + start, * const VB = ws.val_b + start, * const VC = ws.val_c + start;
    const float* __restrict__ score = ws.cscore + start;

    // ---- per-image descending-score order (ties keep nonzero order): VB[rank] = candidate slot
    const unsigned dmask = (1u << sbits) - 1u;
    for (int p = 0; p < n_sp; ++p) {
        const int shift = sbit0 + p * sbits;
        const bool last = p == n_sp - 1;
        const unsigned* kin = p == 1 ? X0 : p == 2 ? X2 : X5;
        const unsigned* vin = p == 1 ? X1 : p == 2 ? X3 : VA;
        unsigned* kout = p == 0 ? X0 : p == 1 ? X2 : X5;
        unsigned* vout = last ? VB : p == 0 ? X1 : p == 1 ? X3 : VA;
        counting_pass(
            L, cnt, sbits,
            [&](int r) { return p == 0 ? make_uint2(~__float_as_uint(score[r]), (unsigned)(start + r)) : make_uint2(ld_l2(kin + r), ld_l2(vin + r)); },
            [&](unsigned k) { return (k >> shift) & dmask; },
            [&](unsigned pos, unsigned k, unsigned v) {
                if (!last) kout[pos] = k;
                vout[pos] = v;
            });
    }

    // ---- the first max_nms of the image: class-offset boxes in score order, keep flags (nms_rank_kernel), then per-(image, segment) runs in score order
    const int cntm = cnt < max_nms ? cnt : max_nms;
    const bool part = !agnostic && nseg > 1 && (__uint_as_float(ws.img_maxabs[img]) < max_wh * 0.5f);
    for (int r = tid; r < cntm; r += SORT_WAVES * 64) {
        const unsigned g = ld_l2(VB + r);
        const float4 b = ws.cbox[g];
        const float c = (float)ws.ccls[g] * (agnostic ? 0.0f : max_wh);   // x[:, 5:6] * (0 if agnostic else max_wh)  (:731)
        ws.sbox[start + r] = make_float4(b.x + c, b.y + c, b.z + c, b.w + c);
        ws.keep[start + r] = 0;
        if (!part) { VC[r] = (unsigned)(start + r); X4[r] = 0u; }
    }
    if (!part) return;   // (uniform)
    int cbits = 1;
    while ((1 << cbits) < nseg) ++cbits;
    auto seg_of = [&](int r) { return make_uint2((unsigned)ws.ccls[ld_l2(VB + r)], (unsigned)(start + r)); };
    if (cbits <= 9) {
        counting_pass(L, cntm, cbits, seg_of, [&](unsigned k) { return k; }, [&](unsigned pos, unsigned k, unsigned v) { X4[pos] = k; VC[pos] = v; });
    } else {
        counting_pass(L, cntm, 9, seg_of, [&](unsigned k) { return k & 511u; }, [&](unsigned pos, unsigned k, unsigned v) { X0[pos] = k; X1[pos] = v; });
        counting_pass(
            L, cntm, cbits - 9, [&](int r) { return make_uint2(ld_l2(X0 + r), ld_l2(X1 + r)); }, [&](unsigned k) { return k >> 9; },
            [&](unsigned pos, unsigned k, unsigned v) { X4[pos] = k; VC[pos] = v; });
    }
}

Y3_DEV long long lower_bound_u64(const unsigned long long* __restrict__ a, long long n, unsigned long long v) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

Y3_DEV long long lower_bound_u32(const unsigned* __restrict__ a, long long n, unsigned v) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// greedy NMS of one (image, segment); positions refer to sorted-1 order (ws.sbox / ws.keep).
//
// The answer is torchvision's CPU loop (visit the boxes in score order; a box that no KEPT earlier box overlaps by more than the threshold is kept and suppresses
// later ones; stop after max_det kept).  Round 1-4 ran that loop one kept box at a time -- three block barriers and a pass over the remaining boxes per kept box:
// ~1.2 us each, 0.4 ms for a batch whose boxes rarely suppress each other (the model's own predictions: up to max_det kept per class, profiles/r05_bench_kernel_stats_b.md).
// Now the segment is walked in blocks of 64 boxes:
//   1. the block's 64 x 64 suppression bits (bit j of row i: box i would suppress box j > i) are computed by all 256 threads at once (16 pairs each);
//   2. ONE wave resolves the block sequentially out of registers (row i lives in lane i; v_readlane, no memory in the chain): box i is kept unless an earlier kept
//      box -- of an earlier block (bits in `removed`) or of this one -- suppressed it; the kept count stops the walk at max_det exactly where the loop would;
//   3. the block's kept boxes (<= 64, in LDS) suppress the boxes of all later blocks in parallel.
// The same IoU expression on the same operands decides every bit (inter / (area_i + area_j - inter) as a float, compared with the double threshold), so the
// kept set is the loop's, bit for bit; four barriers per 64 boxes instead of three per kept box.
__global__ __launch_bounds__(256) void nms_greedy_kernel(NmsWs ws, int nseg, double iou_thr, int max_det, int max_nms, const unsigned long long* __restrict__ key2,
                                                          const unsigned* __restrict__ pos2, const unsigned* __restrict__ seg_sorted, int n_rows) {
    __shared__ unsigned removed[(1 << RANK_BITS) / 32];  // 32768 bits = 4 KiB
    __shared__ float4 blk[64];                            // the current block's boxes
    __shared__ unsigned long long srow[64];               // their suppression rows inside the block
    __shared__ unsigned long long s_keepm;                // which of them were kept
    __shared__ int s_kept;                                // kept so far in the segment
    const int img = blockIdx.x / nseg, seg = blockIdx.x % nseg;
    long long lo, hi;
    if (seg_sorted) {   // nms_sort_kernel's order: the image's first max_nms slots hold ascending segment ids
        const long long start = ws.row_off[(long long)img * n_rows];
        long long cnt = (long long)ws.row_off[(long long)(img + 1) * n_rows] - start;
        if (start + cnt > ws.cap) cnt = ws.cap - start > 0 ? ws.cap - start : 0;
        if (cnt > max_nms) cnt = max_nms;
        lo = start + lower_bound_u32(seg_sorted + start, cnt, (unsigned)seg);
        hi = start + lower_bound_u32(seg_sorted + start, cnt, (unsigned)seg + 1u);
    } else {
        const unsigned long long kbase = ((unsigned long long)img << (CLS_BITS + RANK_BITS)) | ((unsigned long long)seg << RANK_BITS);
        lo = lower_bound_u64(key2, ws.cap, kbase);
        hi = lower_bound_u64(key2, ws.cap, kbase + (1ull << RANK_BITS));
    }
    const int n = (int)(hi - lo);
    if (n <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int w = tid; w < 2 * ((n + 63) / 64); w += 256) removed[w] = 0u;   // whole 64-box blocks: the walk below reads both words of a block
    if (tid == 0) s_kept = 0;
    const unsigned* __restrict__ pos = pos2 + lo;
    auto suppresses = [&](const float4 bi, const float iarea, const float4 bj) {   // torchvision's test, operand for operand
        const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
        const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
        float w = xx2 - xx1, h = yy2 - yy1;
        w = w > 0.0f ? w : 0.0f;
        h = h > 0.0f ? h : 0.0f;
        const float inter = w * h;
        const float jarea = (bj.z - bj.x) * (bj.w - bj.y);
        const float ovr = inter / (iarea + jarea - inter);
        return (double)ovr > iou_thr;
    };
    for (int base = 0; base < n; base += 64) {
        const int cnt = n - base < 64 ? n - base : 64;
        __syncthreads();   // `removed` as the blocks before this one left it (first trip: zeroed); blk / srow free
        const unsigned long long rem_in = (unsigned long long)removed[base >> 5] | ((unsigned long long)removed[(base >> 5) + 1] << 32);   // (uniform; the words of the last block are zeroed whole, bits beyond n stay 0)
        const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
        if ((rem_in & valid) == valid) continue;   // every box of the block is already suppressed (uniform)
        if (tid < 64) {
            if (tid < cnt) blk[tid] = ws.sbox[pos[base + tid]];
            srow[tid] = 0ull;
        }
        __syncthreads();
        // 1. suppression rows: lane i of wave w tests box i against boxes 16 w .. 16 w + 15 of the block (only j > i matters)
        if (lane < cnt) {
            const float4 bi = blk[lane];
            const float iarea = (bi.z - bi.x) * (bi.w - bi.y);
            unsigned long long bits = 0ull;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int j = wv * 16 + q;
                if (j > lane && j < cnt && suppresses(bi, iarea, blk[j])) bits |= 1ull << j;
            }
            if (bits) atomicOr(&srow[lane], bits);
        }
        __syncthreads();
        // 2. sequential resolve by wave 0, out of registers
        if (wv == 0) {
            const unsigned long long row = srow[lane];
            const int row_lo = (int)(unsigned)row, row_hi = (int)(unsigned)(row >> 32);
            unsigned long long rem = rem_in, keepm = 0ull;
            int kept = s_kept;
            for (int i = 0; i < cnt && kept < max_det; ++i) {
                if (!((rem >> i) & 1ull)) {
                    keepm |= 1ull << i;
                    ++kept;
                    rem |= (unsigned long long)(unsigned)__builtin_amdgcn_readlane(row_lo, i) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(row_hi, i) << 32);
                }
            }
            if ((keepm >> lane) & 1ull) ws.keep[pos[base + lane]] = 1;
            if (lane == 0) { s_keepm = keepm; s_kept = kept; }
        }
        __syncthreads();
        const unsigned long long keepm = s_keepm;
        if (s_kept >= max_det) break;   // (uniform) later boxes can never be kept: the loop stops here too
        // 3. the kept boxes of this block against every box behind the block
        if (keepm)
            for (int j = base + 64 + tid; j < n; j += 256) {
                if (removed[j >> 5] & (1u << (j & 31))) continue;
                const float4 bj = ws.sbox[pos[j]];
                unsigned long long km = keepm;
                while (km) {
                    const int i = __builtin_ctzll(km);
                    km &= km - 1ull;
                    const float4 bi = blk[i];
                    if (suppresses(bi, (bi.z - bi.x) * (bi.w - bi.y), bj)) { atomicOr(&removed[j >> 5], 1u << (j & 31)); break; }
                }
            }
    }
}

// per image: kept candidates in descending-score order, first max_det -> out rows
__global__ __launch_bounds__(256) void nms_gather_kernel(NmsWs ws, int n_rows, int max_det, int max_nms, const unsigned* __restrict__ idx1,
                                                          float* __restrict__ out_rows, int* __restrict__ out_counts) {
    __shared__ int wave_cnt[4];
    __shared__ int s_base;
    const int img = blockIdx.x;
    const long long start = ws.row_off[(long long)img * n_rows];
    long long cnt = (long long)ws.row_off[(long long)(img + 1) * n_rows] - start;
    if (start + cnt > ws.cap) cnt = ws.cap - start > 0 ? ws.cap - start : 0;
    if (cnt > max_nms) cnt = max_nms;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < cnt; c0 += 256) {
        const long long i = c0 + tid;
        const bool k = (i < cnt) && ws.keep[start + i];
        const unsigned long long m = __ballot(k);
        if (lane == 0) wave_cnt[wv] = __builtin_popcountll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wv; ++w) before += wave_cnt[w];
        const int slot = before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (k && slot < max_det) {
            const unsigned g = idx1[start + i];
            const float4 b = ws.cbox[g];
            float* o = out_rows + ((long long)img * max_det + slot) * 6;
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w;
            o[4] = ws.cscore[g];
            o[5] = (float)ws.ccls[g];
        }
        __syncthreads();
        if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
        if (s_base >= max_det) break;
    }
    if (tid == 0) out_counts[img] = s_base < max_det ? s_base : max_det;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

long long worst_capacity(int bs, int n_rows, int nc, const y3_nms_params* p) {
    return (long long)bs * n_rows * ((p->multi_label && nc > 1) ? nc : 1);
}
long long default_capacity(int bs, int n_rows, int nc, const y3_nms_params* p) {
    const long long worst = worst_capacity(bs, n_rows, nc, p);
    long long cap = (long long)bs * 16384;
    if (cap > worst) cap = worst;
    if (cap < 1024) cap = 1024;
    return cap;
}

size_t temp_bytes(long long cap, size_t scan_n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)cap, 0u,
                                    64u, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b, (int*)nullptr, (int*)nullptr, 0, scan_n + 1, rocprim::plus<int>(), (hipStream_t)0);
    return align_up(a > b ? a : b);
}

size_t carve(NmsWs& ws, unsigned char* base, int bs, long long cap, size_t scan_n) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    ws.cap = cap;
    ws.row_count = (int*)take((scan_n + 1) * sizeof(int));
    ws.row_off = (int*)take((scan_n + 1) * sizeof(int));
    ws.img_maxabs = (unsigned*)take((size_t)65 * bs * sizeof(unsigned));
    ws.img_total = ws.img_maxabs ? ws.img_maxabs + bs : nullptr;
    ws.key_a = (unsigned long long*)take((size_t)cap * 8);
    ws.key_b = (unsigned long long*)take((size_t)cap * 8);
    ws.key_c = (unsigned long long*)take((size_t)cap * 8);
    ws.val_a = (unsigned*)take((size_t)cap * 4);
    ws.val_b = (unsigned*)take((size_t)cap * 4);
    ws.val_c = (unsigned*)take((size_t)cap * 4);
    ws.cbox = (float4*)take((size_t)cap * 16);
    ws.cscore = (float*)take((size_t)cap * 4);
    ws.ccls = (int*)take((size_t)cap * 4);
    ws.sbox = (float4*)take((size_t)cap * 16);
    ws.keep = (unsigned char*)take((size_t)cap);
    ws.tmp_bytes = temp_bytes(cap, scan_n);
    ws.tmp = take(ws.tmp_bytes);
    return off;
}

template <typename T>
int run_nms(const void* pred, int bs, int n_rows, int nc, const y3_nms_params* p, const int* classes, float* out_rows, int* out_counts, int* out_status,
            long long capacity, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const size_t scan_n = (size_t)bs * n_rows;
    long long cap = capacity > 0 ? capacity : default_capacity(bs, n_rows, nc, p);
    const long long worst = worst_capacity(bs, n_rows, nc, p);
    if (cap > worst) cap = worst;
    if (cap < 1024) cap = 1024;
    NmsWs ws;
    const size_t need = carve(ws, nullptr, bs, cap, scan_n);
    if (need > workspace_bytes) Y3_FAIL("y3_nms: workspace too small (%zu bytes given, %zu needed for capacity %lld)", workspace_bytes, need, cap);
    if (cap > 0x7ffffff0LL) Y3_FAIL("y3_nms: capacity too large");
    carve(ws, (unsigned char*)workspace, bs, cap, scan_n);

    const float thr = (float)(T)(p->conf_thres);              // python scalar -> tensor dtype before the compare
    const int multi = (p->multi_label && nc > 1) ? 1 : 0;       // multi_label &= nc > 1   (utils/general.py:677)
    const int ncf = classes ? p->n_classes_filter : 0;
    // tie-break ordinal (nonzero order) must fit ORD_BITS; coarser buckets fall back to the radix sort's stability
    int ord_shift = 0;
    while ((((long long)n_rows * (multi ? nc : 1)) >> ord_shift) >= (1ll << ORD_BITS)) ++ord_shift;
    const bool one_seg = p->agnostic != 0;
    const int nseg = one_seg ? 1 : nc;

    Y3_HIP(hipMemsetAsync(out_status, 0, 2 * sizeof(int), st));
    Y3_HIP(hipMemsetAsync(ws.img_maxabs, 0, (size_t)65 * bs * sizeof(unsigned), st));   // (+ img_total)
    ws.own_sort = y3_knob(Y3K_NMS_SORT) != 0 ? 1 : 0;
    if (!ws.own_sort) Y3_HIP(hipMemsetAsync(ws.key_a, 0xff, (size_t)cap * 8, st));
    Y3_HIP(hipMemsetAsync(ws.row_count + scan_n, 0, sizeof(int), st));

    const long long waves = (long long)bs * ((n_rows + 63) / 64);
    const unsigned cblocks = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((nms_candidates_kernel<T, 0>), dim3(cblocks), dim3(256), 0, st, (const T*)pred, bs, n_rows, nc, thr, multi, classes, ncf, ws, out_status, ord_shift);
    Y3_CHECK_LAUNCH();
    size_t tb = ws.tmp_bytes;
    if (ws.own_sort) {
        hipLaunchKernelGGL(nms_scan_kernel, dim3((unsigned)bs), dim3(1024), 0, st, ws, bs, n_rows);
        Y3_CHECK_LAUNCH();
    } else if (rocprim::exclusive_scan(ws.tmp, tb, ws.row_count, ws.row_off, 0, scan_n + 1, rocprim::plus<int>(), st) != hipSuccess) Y3_FAIL("y3_nms: scan failed");
    hipLaunchKernelGGL((nms_candidates_kernel<T, 1>), dim3(cblocks), dim3(256), 0, st, (const T*)pred, bs, n_rows, nc, thr, multi, classes, ncf, ws, out_status, ord_shift);
    Y3_CHECK_LAUNCH();
    if (ws.own_sort) {
        // conf is a value of T: below the sign an f16 / bf16 value has 18 bits that can differ (fp32 bits 13 .. 30), an fp32 value all of them
        const bool narrow = sizeof(T) == 2;
        hipLaunchKernelGGL(nms_sort_kernel, dim3((unsigned)bs), dim3(SORT_WAVES * 64), 0, st, ws, bs, n_rows, p->max_nms, p->max_wh, p->agnostic, nseg, narrow ? 13 : 0,
                           narrow ? 9 : 8, narrow ? 2 : 4, out_status);
        Y3_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_greedy_kernel, dim3((unsigned)(bs * nseg)), dim3(256), 0, st, ws, nseg, p->iou_thres, p->max_det, p->max_nms,
                           (const unsigned long long*)nullptr, ws.val_c, (const unsigned*)ws.key_c, n_rows);
        Y3_CHECK_LAUNCH();
        hipLaunchKernelGGL(nms_gather_kernel, dim3((unsigned)bs), dim3(256), 0, st, ws, n_rows, p->max_det, p->max_nms, ws.val_b, out_rows, out_counts);
        Y3_CHECK_LAUNCH();
        return 0;
    }
    tb = ws.tmp_bytes;
    if (rocprim::radix_sort_pairs(ws.tmp, tb, ws.key_a, ws.key_b, ws.val_a, ws.val_b, (size_t)cap, 0u, 64u, st) != hipSuccess) Y3_FAIL("y3_nms: sort-1 failed");
    // (key_b, val_b) = per-image descending-score order.  key_a / val_a are free again: reuse for key2 / positions.
    const unsigned rblocks = (unsigned)((cap + 255) / 256);
    hipLaunchKernelGGL(nms_rank_kernel, dim3(rblocks), dim3(256), 0, st, ws, bs, n_rows, p->max_nms, p->max_wh, p->agnostic, ws.key_b, ws.val_b, ws.key_a, ws.val_a,
                       out_status);
    Y3_CHECK_LAUNCH();
    tb = ws.tmp_bytes;
    if (rocprim::radix_sort_pairs(ws.tmp, tb, ws.key_a, ws.key_c, ws.val_a, ws.val_c, (size_t)cap, 0u, (unsigned)(IMG_BITS + CLS_BITS + RANK_BITS), st) != hipSuccess)
        Y3_FAIL("y3_nms: sort-2 failed");
    hipLaunchKernelGGL(nms_greedy_kernel, dim3((unsigned)(bs * nseg)), dim3(256), 0, st, ws, nseg, p->iou_thres, p->max_det, p->max_nms, ws.key_c, ws.val_c,
                       (const unsigned*)nullptr, n_rows);
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_gather_kernel, dim3((unsigned)bs), dim3(256), 0, st, ws, n_rows, p->max_det, p->max_nms, ws.val_b, out_rows, out_counts);
    Y3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" size_t y3_nms_workspace_bytes(int32_t bs, int32_t n_rows, int32_t nc, const y3_nms_params* p, int64_t capacity) {
    if (!p || bs <= 0 || n_rows <= 0 || nc <= 0) return 0;
    long long cap = capacity > 0 ? capacity : default_capacity(bs, n_rows, nc, p);
    const long long worst = worst_capacity(bs, n_rows, nc, p);
    if (cap > worst) cap = worst;
    if (cap < 1024) cap = 1024;
    NmsWs ws;
    return carve(ws, nullptr, bs, cap, (size_t)bs * n_rows);
}

extern "C" int y3_nms(const void* pred, int32_t dtype, int32_t bs, int32_t n_rows, int32_t nc, const y3_nms_params* p, const int32_t* classes, float* out_rows,
                      int32_t* out_counts, int32_t* out_status, int64_t capacity, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !p || !out_rows || !out_counts || !out_status || !workspace) Y3_FAIL("y3_nms: null argument");
    if (!(p->conf_thres >= 0.0f && p->conf_thres <= 1.0f)) Y3_FAIL("Invalid Confidence threshold %g, valid values are between 0.0 and 1.0", (double)p->conf_thres);
    if (!(p->iou_thres >= 0.0 && p->iou_thres <= 1.0)) Y3_FAIL("Invalid IoU %g, valid values are between 0.0 and 1.0", p->iou_thres);
    if (bs <= 0 || bs >= (1 << IMG_BITS) - 1) Y3_FAIL("y3_nms: batch size %d unsupported (max %d)", bs, (1 << IMG_BITS) - 2);
    if (nc <= 0 || nc >= (1 << CLS_BITS)) Y3_FAIL("y3_nms: class count %d unsupported", nc);
    if (p->max_nms <= 0 || p->max_nms > (1 << RANK_BITS) - 1) Y3_FAIL("y3_nms: max_nms %d unsupported (max %d)", p->max_nms, (1 << RANK_BITS) - 1);
    if ((long long)bs * n_rows >= 0x7fffffffLL) Y3_FAIL("y3_nms: too many rows");
    if (p->max_det <= 0) Y3_FAIL("y3_nms: max_det must be positive");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case Y3_F16: return run_nms<f16_t>(pred, bs, n_rows, nc, p, classes, out_rows, out_counts, out_status, capacity, workspace, workspace_bytes, st);
        case Y3_BF16: return run_nms<bf16_t>(pred, bs, n_rows, nc, p, classes, out_rows, out_counts, out_status, capacity, workspace, workspace_bytes, st);
        case Y3_F32: return run_nms<float>(pred, bs, n_rows, nc, p, classes, out_rows, out_counts, out_status, capacity, workspace, workspace_bytes, st);
    }
    Y3_FAIL("y3_nms: bad dtype %d", dtype);
}

extern "C" int y3_detect_decode(const y3_tensor* head, int32_t dtype, int32_t na, int32_t no, const float* anchors_px, float stride, void* raw, void* z,
                                int64_t row_offset, int64_t total_rows, void* stream) {
    if (!head || !anchors_px) Y3_FAIL("y3_detect_decode: null argument");
    if (na < 1 || na > 5) Y3_FAIL("y3_detect_decode: na=%d unsupported (1..5)", na);
    if (head->c < na * no) Y3_FAIL("y3_detect_decode: head has %d channels, needs %d", head->c, na * no);
    float a[10] = {0};
    for (int i = 0; i < na * 2; ++i) a[i] = anchors_px[i];
    const long long total = (long long)head->n * head->h * head->w * na * no;
    hipStream_t st = (hipStream_t)stream;
    const long long P = (long long)head->h * head->w;
    const bool vec = dtype != Y3_F32 && no >= 8 && (P * no) % 8 == 0 && (total_rows * no) % 8 == 0 && (row_offset * no) % 8 == 0 &&
                     !((uintptr_t)raw & 15) && !((uintptr_t)z & 15) && P * no < 0x7fffffffLL;
    if (vec) {
        const dim3 vgrid((unsigned)((total / 8 + 255) / 256));
#define Y3_DECODE_V(T)                                                                                                                                           \
    hipLaunchKernelGGL((decode_vec_kernel<T>), vgrid, dim3(256), 0, st, (const T*)head->data, head->n, head->h, head->w, head->pitch, na, no, a[0], a[1], a[2], a[3], \
                       a[4], a[5], a[6], a[7], a[8], a[9], stride, (T*)raw, (T*)z, (long long)row_offset, (long long)total_rows)
        if (dtype == Y3_F16) Y3_DECODE_V(f16_t); else Y3_DECODE_V(bf16_t);
#undef Y3_DECODE_V
        Y3_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)((total + 255) / 256));
#define Y3_DECODE(T)                                                                                                                                        \
    hipLaunchKernelGGL((decode_kernel<T>), grid, dim3(256), 0, st, (const T*)head->data, head->n, head->h, head->w, head->pitch, na, no, a[0], a[1], a[2], a[3], \
                       a[4], a[5], a[6], a[7], a[8], a[9], stride, (T*)raw, (T*)z, (long long)row_offset, (long long)total_rows)
    switch (dtype) {
        case Y3_F16: Y3_DECODE(f16_t); break;
        case Y3_BF16: Y3_DECODE(bf16_t); break;
        case Y3_F32: Y3_DECODE(float); break;
        default: Y3_FAIL("y3_detect_decode: bad dtype %d", dtype);
    }
#undef Y3_DECODE
    Y3_CHECK_LAUNCH();
    return 0;
}
