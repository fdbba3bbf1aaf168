// Training-side kernels (gfx950): batch-statistics BatchNorm + SiLU forward/backward, filter gradients,
// data-gradient filter packing, and the backward of the parameter-free layers.
//
// Reference semantics: `Conv.forward` in train mode = act(bn(conv(x))) with nn.BatchNorm2d(eps 1e-3, momentum 0.03)
// using BATCH statistics (models/common.py:75, models/yolo.py:229); autograd of conv2d / batch_norm / silu /
// upsample_nearest2d / cat / max_pool2d.  The data gradient of a convolution reuses the forward
// implicit-GEMM kernel (conv.hip) on the output gradient with the flipped+transposed filter bank packed here.
//
// Statistics are accumulated in fp64 (per-thread partials -> LDS tree -> one partial row per block -> fixed-order sum):
// E[x^2]-E[x]^2 in fp32 would not hold the 1e-4 parity bar; no atomics, so results are run-to-run deterministic.
#include "y3_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

template <typename T> struct V16 {  // one 16-byte vector of T
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

// 16-byte vector access with the cache policy as a template flag: NT = `nt` loads / stores.  Measured on MI355X (tools/lab/bn_lab.hip,
// profiles/r02_bn_lab_sweep*.txt): on tensors that no cache level can hold (>= 128 MB here) the elementwise BatchNorm passes gain
// 10-30 % from non-temporal STORES (fwd 5.5 -> 7.1 TB/s on 210 MB) and the two-read passes another 5-10 % from non-temporal LOADS;
// on tensors that fit the 256 MB Infinity Cache `nt` loads lose 10-15 %, so the host picks per launch by the tensor's bytes.
typedef unsigned int y3_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, typename T> Y3_DEV V16<T> ldv(const T* p) {
    const y3_u32x4 r = NT ? __builtin_nontemporal_load((const y3_u32x4*)p) : *(const y3_u32x4*)p;
    return __builtin_bit_cast(V16<T>, r);
}
template <bool NT, typename T> Y3_DEV void stv(T* p, const V16<T>& v) {
    const y3_u32x4 r = __builtin_bit_cast(y3_u32x4, v);
    if (NT) __builtin_nontemporal_store(r, (y3_u32x4*)p); else *(y3_u32x4*)p = r;
}
// tensors at least this large take the non-temporal forms of the elementwise passes (knob "bn_nt_bytes", default 64 MiB: below the
// Infinity-Cache size the plain forms win alone, profiles/r02_bn_lab_sweep*.txt; inside the two-stream train step 64 MiB measured 55.04 / 54.99 ms against 55.25 / 55.23 at
// 128 MiB and 55.29 / 55.12 at 32, profiles/r06_train_knob_sweep.txt); the tests lower it to run the forms on small tensors
#define Y3_NT_BYTES y3_knob(Y3K_BN_NT_BYTES)

Y3_DEV float silu_grad(float z, float s) { return s + z * s * (1.0f - s); }  // d silu(z)/dz with s = sigmoid(z)

// The elementwise BN kernels work on PAIRS of channels: the multiplies / adds / fmas are packed fp32 (v_pk_*: two values per lane and
// issue), the activation is a template parameter (tested per element at run time it left one uniform branch per element and nothing
// for the scheduler to fill the transcendental latency with) and 16-bit results are rounded two at a time (v_cvt_pk).  The
// backward reduction was bound by VALU issue, not HBM: 22 VALU + 2 transcendental instructions per element.
template <typename T> Y3_DEV f32x2 ld2(const V16<T>& x, int q) { return f32x2{to_f32<T>(x.v[q]), to_f32<T>(x.v[q + 1])}; }
template <typename T> Y3_DEV void st2(V16<T>& o, int q, f32x2 v) {
    if constexpr (sizeof(T) == 2) {
        const unsigned w = pack2<T>(v[0], v[1]);
        __builtin_memcpy(&o.v[q], &w, 4);
    } else {
        o.v[q] = v[0];
        o.v[q + 1] = v[1];
    }
}
// sigmoid2 / silu_grad2: y3_common.h (shared with the data-gradient epilogue of conv.hip)
// v_exp_f32 + v_rcp_f32 (1-2 ulp each): the elementwise BN kernels were VALU-bound on expf() + an IEEE divide per element
Y3_DEV float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

unsigned nblk(long long total) { return (unsigned)((total + 255) / 256); }
int esize(int dtype) { return dtype == Y3_F32 ? 4 : 2; }
bool vec_ok(const y3_tensor* t, int esz) {
    const int v = 16 / esz;
    return (t->c % v) == 0 && (t->pitch % v) == 0 && (((uintptr_t)t->data) & 15) == 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Per-channel reductions over N*H*W.  MODE 0: (sum u, sum u^2).  MODE 1: (sum dz, sum dz*xhat) for the BN+act backward.
// Thread t owns channel vector (t % CG) on pixel lane (t / CG); CG = C / V (<= 256).
template <typename T, int MODE, bool SILU = false, bool NTL = false>
__global__ __launch_bounds__(256) void channel_reduce_kernel(const T* __restrict__ u, int upitch, const T* __restrict__ dy, int dpitch, long long M, int C,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               double* __restrict__ sums) {
    constexpr int V = V16<T>::N;
    constexpr int RP = 258;               // plane pitch in doubles: 516 dwords = 4 banks apart, so the 2 V planes one channel group's entries sit in never share a bank
    __shared__ double red[RP * 2 * V];    // 33 KiB (16-bit T) / 16.5 KiB (fp32)
    const int CG = C / V;
    const int PL = 256 / CG;
    const int tid = threadIdx.x;
    const int cg = tid % CG, pl = tid / CG;
    double a0[V], a1[V];
#pragma unroll
    for (int q = 0; q < V; ++q) a0[q] = a1[q] = 0.0;
    if (pl < PL) {
        f32x2 sc[V / 2], sh[V / 2], mu[V / 2], is[V / 2];
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < V; q += 2) {
                const int c = cg * V + q;
                sc[q / 2] = f32x2{scale[c], scale[c + 1]}; sh[q / 2] = f32x2{shift[c], shift[c + 1]};
                mu[q / 2] = f32x2{mean[c], mean[c + 1]}; is[q / 2] = f32x2{invstd[c], invstd[c + 1]};
            }
        }
        // fp32 partials over runs of 8 pixels (relative error ~1e-6 per run), flushed into the fp64 accumulators:
        // keeps the fp64 VALU work at 1/8 of the element count
        f32x2 f0[V / 2], f1[V / 2];
#pragma unroll
        for (int q = 0; q < V / 2; ++q) f0[q] = f1[q] = f32x2{0.0f, 0.0f};
        // 4 pixels per trip with all loads issued before the arithmetic (one 16-byte load in flight per thread left the
        // reductions at 1.5-1.8 TB/s), and the NEXT trip's loads issued before this trip's arithmetic (two register sets: with
        // 8 waves per CU the sigmoid / fp64 work of MODE 1 otherwise runs behind an idle memory pipe: 3.4 TB/s);
        // fp32 partials are flushed into the fp64 accumulators every 2 trips = 8 pixels
        const long long stride = (long long)gridDim.x * PL;
        auto fetch = [&](long long m0, V16<T> (&xs)[4], V16<T> (&gs)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long m = m0 + j * stride;
                if (m < M) {
                    xs[j] = ldv<NTL, T>(u + m * upitch + cg * V);
                    if (MODE == 1) gs[j] = ldv<NTL, T>(dy + m * dpitch + cg * V);
                }
            }
        };
        auto accumulate = [&](long long m0, const V16<T> (&xs)[4], const V16<T> (&gs)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (m0 + j * stride >= M) break;
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < V / 2; ++q) { const f32x2 f = ld2<T>(xs[j], 2 * q); f0[q] += f; f1[q] += f * f; }
                } else {
#pragma unroll
                    for (int q = 0; q < V / 2; ++q) {
                        const f32x2 uf = ld2<T>(xs[j], 2 * q);
                        f32x2 dz = ld2<T>(gs[j], 2 * q);
                        if (SILU) {
                            const f32x2 z = uf * sc[q] + sh[q];
                            dz *= silu_grad2(z, sigmoid2(z));
                        }
                        const f32x2 xh = (uf - mu[q]) * is[q];
                        f0[q] += dz;
                        f1[q] += dz * xh;
                    }
                }
            }
        };
        auto flush = [&]() {
#pragma unroll
            for (int q = 0; q < V; ++q) { a0[q] += (double)f0[q / 2][q & 1]; a1[q] += (double)f1[q / 2][q & 1]; }
#pragma unroll
            for (int q = 0; q < V / 2; ++q) f0[q] = f1[q] = f32x2{0.0f, 0.0f};
        };
        V16<T> xa[4], ga[4], xb[4], gb[4];
        long long m0 = (long long)blockIdx.x * PL + pl;
        fetch(m0, xa, ga);
        while (m0 < M) {
            fetch(m0 + 4 * stride, xb, gb);
            accumulate(m0, xa, ga);
            m0 += 4 * stride;
            if (m0 >= M) break;
            fetch(m0 + 4 * stride, xa, ga);
            accumulate(m0, xb, gb);
            m0 += 4 * stride;
            flush();
        }
#pragma unroll
        for (int q = 0; q < V; ++q) { a0[q] += (double)f0[q / 2][q & 1]; a1[q] += (double)f1[q / 2][q & 1]; }
    }
    // reduce over pixel lanes: every thread drops its 2 V sums into LDS ([entry][thread]: conflict-free), ONE barrier, then thread j adds the PL pixel lanes of entry j
    // (channel j / 2, sum j & 1) in lane order -- the same order, hence the same bits, as the per-channel rounds this replaces (2 V barriers and a serial PL-long
    // loop each: ~10 us at the end of every block, and every block of these one-round grids ends at the same time -- a quarter of a 45 us launch on the 80 x 80 maps)
#pragma unroll
    for (int q = 0; q < V; ++q) {
        red[(2 * q) * RP + tid] = a0[q];
        red[(2 * q + 1) * RP + tid] = a1[q];
    }
    __syncthreads();
    double* part = sums + (size_t)(1 + blockIdx.x) * 2 * C;   // row 0 = totals, rows 1.. = per-block partials
    for (int j = tid; j < 2 * C; j += 256) {
        const int c = j >> 1, cgj = c / V, q = c - cgj * V;
        const double* src = red + (2 * q + (j & 1)) * RP + cgj;
        double sacc = 0.0;
        for (int k = 0; k < PL; ++k) sacc += src[k * CG];
        part[j] = sacc;
    }
}

// totals[j] = sum over the per-block partial rows: one block per 16 entries, 16 row-lanes each, fixed tree (deterministic).
// The consumers of the totals ride along (one launch instead of two or three ~5 us launches per conv unit and pass):
//   MODE 1: + BatchNorm finalize (mean / biased var -> scale, shift, running statistics) for the block's 8 channels
//   MODE 2: + (dbeta, dgamma) of the BN backward        MODE 3: + fp32 copy of the even entries (bias gradient)
struct BnFinalizeArgs {
    double count;
    const double* count_dev;   // SyncBatchNorm: the all-reduced element count, read on the device (overrides `count` when set)
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* rmean;
    float* rvar;
    float* scale;
    float* shift;
    float* mean;
    float* invstd;
};
Y3_DEV void bn_finalize_channel(int c, double s0, double s1, const BnFinalizeArgs& fa) {
    BnFinalizeArgs f = fa;
    if (f.count_dev) f.count = *f.count_dev;
    const double mu = s0 / f.count;
    double var = s1 / f.count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)f.eps));
    const float g = f.gamma ? f.gamma[c] : 1.0f, b = f.beta ? f.beta[c] : 0.0f;
    f.mean[c] = (float)mu;
    f.invstd[c] = is;
    f.scale[c] = g * is;
    f.shift[c] = b - (float)mu * g * is;
    if (f.rmean) f.rmean[c] = (1.0f - f.momentum) * f.rmean[c] + f.momentum * (float)mu;
    if (f.rvar) f.rvar[c] = (1.0f - f.momentum) * f.rvar[c] + f.momentum * (float)(f.count > 1.0 ? var * f.count / (f.count - 1.0) : var);
}
// TIN = double: rows 1.. of `sums` (the reduction kernels' partial rows); TIN = float: the rows the conv epilogue wrote (`part`)
template <int MODE, typename TIN = double>
__global__ __launch_bounds__(256) void reduce_partials_kernel(double* __restrict__ sums, int n2c, int nblocks, BnFinalizeArgs f, float* __restrict__ o0, float* __restrict__ o1,
                                                                int c_out, const TIN* __restrict__ part = nullptr) {
    __shared__ double red[256];
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    const int rl = threadIdx.x >> 4;
    double a = 0.0;
    if (j < n2c) {
        // 8 independent loads in flight per trip, added in row order (same sum as the plain loop: the launch is a chain of up to
        // 32 dependent-latency round trips otherwise -- 11-13 us for a kernel that moves a few hundred KB, 288 times per step)
        const TIN* src = std::is_same<TIN, double>::value ? (const TIN*)(sums + n2c) : part;
        int b = rl;
        for (; b + 7 * 16 < nblocks; b += 8 * 16) {
            TIN v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(b + q * 16) * n2c + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) a += (double)v[q];
        }
        for (; b < nblocks; b += 16) a += (double)src[(size_t)b * n2c + j];
    }
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
        if (rl < s) red[threadIdx.x] += red[threadIdx.x + s * 16];
        __syncthreads();
    }
    if (rl == 0 && j < n2c) {
        sums[j] = red[threadIdx.x];
        if (MODE != 0 && (j & 1) == 0) {   // entries (2c, 2c+1) of channel c sit in neighbouring lanes of row-lane 0
            const int c = j >> 1;
            const double s0 = red[threadIdx.x], s1 = red[threadIdx.x + 1];
            if (MODE == 1) bn_finalize_channel(c, s0, s1, f);
            if (MODE == 2) {
                if (o0) o0[c] = (float)s0;
                if (o1) o1[c] = (float)s1;
                // means for the apply pass in partial row 0 (this block is the only reader of its 16 columns and is done with them): the
                // apply kernel then needs no fp64 division per thread
                if (f.count > 0.0) { sums[n2c + j] = s0 / f.count; sums[n2c + j + 1] = s1 / f.count; }
            }
            if (MODE == 3) { if (c < c_out) o0[c] = (float)s0; }
        }
    }
}

// sums -> mean / biased var -> (scale, shift) of the normalisation, running-stat update (momentum, unbiased var)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int C, BnFinalizeArgs f) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    bn_finalize_channel(c, sums[c * 2], sums[c * 2 + 1], f);
}

// y = act(u*scale + shift) (+ residual).  Thread (cg, pl) keeps its 8 channels' scale/shift in registers and walks
// pixels pl, pl+PL*grid, ...: one 16-byte load and store per pixel, no per-element index arithmetic.
template <typename T, bool SILU, bool NTS = false>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ u, int upitch, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const T* __restrict__ res, int rpitch, T* __restrict__ y, int ypitch, long long M, int C) {
    constexpr int V = V16<T>::N;
    const int CG = C / V, PL = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f32x2 sc[V / 2], sh[V / 2];
#pragma unroll
    for (int q = 0; q < V; q += 2) { sc[q / 2] = f32x2{scale[cg * V + q], scale[cg * V + q + 1]}; sh[q / 2] = f32x2{shift[cg * V + q], shift[cg * V + q + 1]}; }
    // (a 4-pixel unroll with all loads issued first measured slower here: 6.2 -> 6.9 ms per batch-64 step)
    for (long long m = (long long)blockIdx.x * PL + pl; m < M; m += (long long)gridDim.x * PL) {
        const V16<T> x = *(const V16<T>*)(u + m * upitch + cg * V);
        V16<T> r;
        if (res) r = *(const V16<T>*)(res + m * rpitch + cg * V);
        V16<T> o;
#pragma unroll
        for (int q = 0; q < V / 2; ++q) {
            const f32x2 z = y3_bn_act2(ld2<T>(x, 2 * q), sc[q], sh[q], SILU, res != nullptr, res ? ld2<T>(r, 2 * q) : f32x2{0.0f, 0.0f});   // (shared with conv_1x1s.h)
            st2<T>(o, 2 * q, z);
        }
        stv<NTS, T>(y + m * ypitch + cg * V, o);
    }
}

// du = gamma*invstd * (dz - mean(dz) - xhat*mean(dz*xhat));   dz = dy * act'(z).  Same thread mapping as the forward.
// `means` = (mean dz, mean dz*xhat) per channel as fp64 (row 1 of the reduction scratch / second half of the epilogue totals)
template <typename T, bool SILU, bool NT = false>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ u, int upitch, const T* __restrict__ dy, int dpitch,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const double* __restrict__ means,
                                                                 T* __restrict__ du, int opitch, long long M, int C, T* __restrict__ gres, int gpitch,
                                                                 int gres_acc) {
    constexpr int V = V16<T>::N;
    const int CG = C / V, PL = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f32x2 sc[V / 2], sh[V / 2], mu[V / 2], is[V / 2], m0[V / 2], m1[V / 2];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const int c = cg * V + q;
        sc[q / 2][q & 1] = scale[c]; sh[q / 2][q & 1] = shift[c]; mu[q / 2][q & 1] = mean[c]; is[q / 2][q & 1] = invstd[c];
        m0[q / 2][q & 1] = (float)means[c * 2];
        m1[q / 2][q & 1] = (float)means[c * 2 + 1];
    }
    for (long long m = (long long)blockIdx.x * PL + pl; m < M; m += (long long)gridDim.x * PL) {
        const V16<T> x = ldv<NT, T>(u + m * upitch + cg * V);
        const V16<T> g = ldv<NT, T>(dy + m * dpitch + cg * V);
        V16<T> o;
#pragma unroll
        for (int q = 0; q < V / 2; ++q) {
            const f32x2 uf = ld2<T>(x, 2 * q);
            f32x2 dz = ld2<T>(g, 2 * q);
            if (SILU) {
                const f32x2 z = uf * sc[q] + sh[q];
                dz *= silu_grad2(z, sigmoid2(z));
            }
            const f32x2 xh = (uf - mu[q]) * is[q];
            st2<T>(o, 2 * q, sc[q] * (dz - m0[q] - xh * m1[q]));  // scale = gamma * invstd
        }
        stv<NT, T>(du + m * opitch + cg * V, o);
        if (gres) {   // out = act(bn(conv)) + residual: the residual's gradient (+)= dy, on the pass that reads dy anyway
            V16<T> r = g;
            if (gres_acc) {
                const V16<T> old = *(const V16<T>*)(gres + m * gpitch + cg * V);
#pragma unroll
                for (int q = 0; q < V / 2; ++q) st2<T>(r, 2 * q, ld2<T>(g, 2 * q) + ld2<T>(old, 2 * q));
            }
            *(V16<T>*)(gres + m * gpitch + cg * V) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient, direct form: one thread per dW[co][ci][kh][kw] (OIHW fp32, the layout of nn.Conv2d.weight.grad),
// serial over the pixels of a slice, slices combined with atomicAdd.  Exact-order-free fp32; used for the fp32 parity
// path and as the cross-check of the MFMA kernel.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const T* __restrict__ x, int N, int H, int W, int Cin, int xpitch, const T* __restrict__ du, int Ho,
                                                             int Wo, int Cout, int dpitch, int ks, int stride, int pad, int cin_real, int cout_real,
                                                             float* __restrict__ dw, int mslices) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)cout_real * cin_real * ks * ks;
    if (idx >= total) return;
    const int kw = (int)(idx % ks);
    long long t = idx / ks;
    const int kh = (int)(t % ks);
    t /= ks;
    const int ci = (int)(t % cin_real);
    const int co = (int)(t / cin_real);
    const long long M = (long long)N * Ho * Wo;
    const long long per = (M + mslices - 1) / mslices;
    const long long m0 = (long long)blockIdx.y * per, m1 = m0 + per < M ? m0 + per : M;
    float acc = 0.0f;
    for (long long m = m0; m < m1; ++m) {
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (long long)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
        if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
        acc = fmaf(to_f32<T>(du[m * dpitch + co]), to_f32<T>(x[((long long)(n * H + hi) * W + wi) * xpitch + ci]), acc);
    }
    atomicAdd(&dw[idx], acc);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient on MFMA:  dW[co][(tap,ci)] = sum_m du[m][co] * x[m@tap][ci]  -- a GEMM whose reduction axis (pixels)
// is the SLOW axis of both NHWC operands.  Each loader thread therefore fetches 4 consecutive pixels x 8 channels
// (4 x 16 B, coalesced along channels) and writes them TRANSPOSED into LDS as 8 x ds_write_b64 (4 pixels of one
// channel each), so that MFMA fragments (8 consecutive pixels of one channel) are again plain ds_read_b128.
// K-step 64 pixels, LDS rows padded 128 -> 144 B: conflict-free fragment reads and transposed writes; 2-deep
// register prefetch (counted vmcnt) so a K-step never waits for HBM/L2.
// Tile 128 co x 128 (tap,ci) columns, 4 waves x (64 x 64), pixels split over gridDim.y slices; every block stores
// its fp32 tile into the caller's workspace and wgrad_reduce_kernel sums the slices in a fixed order (deterministic,
// no atomics) into the OIHW gradient.  Threads 0-127 stage du, 128-255 stage x.
struct WgradArgs {
    const void* x;
    const void* du;
    float* dw;
    float* part;   // [slice][tile][128 cols][128 rows] fp32 partial tiles
    int N, H, W, Cin, xpitch, Ho, Wo, Cout, dpitch, ks, stride, pad, cin_real, cout_real;
    long long M;
    int per_slice;  // pixels per slice, multiple of 64
    int n_nt;       // column tiles
    unsigned x_bytes, du_bytes;
    y3_divisor dv_hw, dv_w;   // Ho*Wo, Wo
    int step_n, step_q, step_r;   // a K-step of the 256-tile kernel (32 pixels) as (images, rows, columns): 32 = (step_n*Ho + step_q)*Wo + step_r
    int xcd_group;                // wgrad_block: the tiles of a pixel slice back to back on one XCD
};

// (tile, slice) of a block of the split-K filter-gradient grids (grid = tiles x slices).  The hardware hands consecutive workgroups to
// consecutive XCDs, so with tile = blockIdx.x the column tiles of ONE pixel slice -- which read the same du rows and overlapping x
// rows -- land on different XCDs and each L2 fetches the slice again.  xcd_group: workgroups are taken in runs of 8 x tiles; inside a
// run the 8 XCDs get one slice each and walk its tiles back to back, so a slice's operands are fetched into one L2 once.
Y3_DEV void wgrad_block(int xcd_group, int& tile, int& slice) {
    tile = blockIdx.x;
    slice = blockIdx.y;
    if (!xcd_group) return;
    const int tiles = gridDim.x, slices = gridDim.y;
    const long long L = (long long)blockIdx.y * tiles + blockIdx.x;
    const long long run = 8ll * tiles;
    const int g = (int)(L / run), r = (int)(L - (long long)g * run);
    const int left = slices - g * 8;          // slices of this run (8, fewer in the last one)
    const int w = left < 8 ? left : 8;
    slice = g * 8 + r % w;
    tile = r / w;
}

// Transposing LDS read as inline asm.  Through the builtin the compiler treats the read as possibly aliasing every pending
// `buffer_load ... lds` and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of each K-step: the tile requested a
// moment earlier is then awaited before any MFMA is issued (no load/compute overlap inside a wave -- seen in the ISA of the
// first version of these kernels).  As asm the read is opaque: the kernels order DMA vs. reads themselves (counted vmcnt +
// barrier) and wait for the read data with explicit lgkmcnt(0) statements that carry the fragments as operands.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <int OFF> Y3_DEV s16x4_t lds_read_tr16(unsigned addr) {
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <typename F> Y3_DEV void lds_wait4(F& a, F& b, F& c, F& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory"); }
template <typename F> Y3_DEV void lds_wait6(F& a, F& b, F& c, F& d, F& e, F& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "memory");
}

// ---- wgrad, LDS-DMA + transposing LDS reads (gfx950 ds_read_b64_tr_b16) -------------------------------------------------
// 128 filters x 128 (tap, channel) columns per block, K-step = 64 pixels, partial tiles per pixel slice; the operands are staged in
// their NATURAL layout: a K-step of du is 64 rows of 128
// filters, a K-step of the im2col'd x is 64 rows of 128 columns, both pixel-major exactly as NHWC stores them, so
// `buffer_load ... lds` fills the stage buffers directly (no VGPR round trip, no 32 pack + 16 ds_write_b64 per thread and
// K-step as in the register-transposing kernel of round 1, removed in round 3).  The MFMA wants the pixel (= reduction) index contiguous per lane:
// ds_read_b64_tr_b16 delivers exactly that -- a 16-lane group reads a [4 pixels][16 channels] block and lane i gets
// channel i's 4 pixels; two such reads make one 8-k fragment.  Bank conflicts are avoided with an XOR on the 16-byte slot
// index, physical = logical ^ 4 (row & 3), applied on the DMA source side (the 4 rows of a read group land on 4
// different quarter-rows = all 64 banks once per 32-lane pass).
// CO32 = 32-filter MFMA tiles of the block's filter tile: 4 = 128 filters (2 x 2 waves of 64 x 64), 2 = 64 filters and 1 = 32 filters
// (4 waves side by side along the columns, 64 / 32 filters x 32 columns each).  The narrow forms serve the layers with <= 64 / <= 32
// filters (the 640x640 and 320x320 maps of yolov3): with the 128-filter tile their launches multiplied 75 % / 50 % zero rows and were
// MFMA-bound on padding (layer 0 at batch 64: 0.86 TFLOP executed for 0.05 useful, 1.09 ms against a 0.35 ms HBM floor).
template <typename T, int CO32>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BKP = 64, ROWB = 256, TILE = BKP * ROWB, STAGE = 2 * TILE;   // 16 KiB per operand and stage
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int AM = CO32 >= 2 ? 2 : 1;      // filter tiles per wave
    constexpr int WC = CO32 / AM;              // waves along the filters
    constexpr int WN = 4 / WC;                 // waves along the columns
    constexpr int BN = 4 / WN;                 // column tiles per wave
    const int wc = wv / WN, wn = wv % WN;
    int tile_id, slice_id;
    wgrad_block(p.xcd_group, tile_id, slice_id);
    const int ct = tile_id / p.n_nt, nt = tile_id % p.n_nt;
    const long long m_begin = (long long)slice_id * p.per_slice;
    long long m_end = m_begin + p.per_slice;
    if (m_end > p.M) m_end = p.M;
    if (m_begin >= m_end) return;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.du, 0, (int)p.du_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // staging role: wave w fills rows 16w .. 16w+15 of both operands, 4 rows (one 1 KiB piece) per instruction;
    // lane -> (row within the piece = lane / 16, physical slot = lane % 16); row & 3 is the same for all of a lane's rows
    const int prow = lane >> 4, pslot = lane & 15;
    const int lslot = pslot ^ (4 * prow);
    const int a_ch = ct * 128 + lslot * 8;                 // filter group this lane copies from du
    const bool a_ok = a_ch < p.Cout && lslot * 8 < CO32 * 32;
    const int ncol = nt * 128 + lslot * 8;                 // im2col column group this lane copies from x
    const int b_tap = ncol / p.Cin, b_ci = ncol - b_tap * p.Cin;
    const int b_kh = b_tap / p.ks, b_kw = b_tap - b_kh * p.ks;
    const bool b_ok = b_tap < p.ks * p.ks;

    // pixel cursors of the lane's four rows (rows 16 w + 4 j + prow of a K-step), advanced incrementally by one K-step of 64 pixels: output
    // row / column (for the wraps), input row / column of the lane's tap, byte offsets of its 16 bytes in du and x -- no division and no
    // multiplication per K-step (round 3; the stateless form recomputed two multiply-shift divisions and five multiplies per row and K-step)
    int cm[4], cho[4], cwo[4], chi[4], cwi[4];
    unsigned aoffv[4], xlin[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long m = m_begin + wv * 16 + j * 4 + prow;
        const int mm = (int)(m < p.M ? m : p.M - 1);
        cm[j] = (int)m;
        const int n0 = y3_fdiv(mm, p.dv_hw);
        const int rem = mm - n0 * (p.Ho * p.Wo);
        cho[j] = y3_fdiv(rem, p.dv_w);
        cwo[j] = rem - cho[j] * p.Wo;
        chi[j] = cho[j] * p.stride - p.pad + b_kh;
        cwi[j] = cwo[j] * p.stride - p.pad + b_kw;
        aoffv[j] = ((unsigned)cm[j] * (unsigned)p.dpitch + (unsigned)a_ch) * 2u;
        xlin[j] = (unsigned)(((n0 * p.H + chi[j]) * p.W + cwi[j]) * p.xpitch + b_ci) * 2u;   // meaningful only while (chi, cwi) is inside the image
    }
    const int m_end_i = (int)m_end;
    const int st_n = y3_fdiv(BKP, p.dv_hw), st_rem = BKP - st_n * (p.Ho * p.Wo);          // 64 pixels = (st_n images, st_q rows, st_r columns)
    const int st_q = y3_fdiv(st_rem, p.dv_w), st_r = st_rem - st_q * p.Wo;
    const int xp2 = p.xpitch * 2;
    const unsigned a_step = (unsigned)(BKP * p.dpitch * 2);
    const int wi_step = st_r * p.stride, wi_wrap = p.Wo * p.stride;
    const int hi_step = st_q * p.stride, hi_wrap = p.Ho * p.stride;
    const unsigned x_step = (unsigned)(((st_n * p.H + st_q * p.stride) * p.W + st_r * p.stride) * xp2);
    const unsigned x_wrapw = (unsigned)((p.stride * p.W - p.Wo * p.stride) * xp2);
    const unsigned x_wraph = (unsigned)((p.H - p.Ho * p.stride) * p.W * xp2);

    auto dma = [&](int it, int stage) {   // K-steps are requested strictly in order (it = 0, 1, 2, ...): the cursors advance by one K-step per call
        (void)it;
        unsigned char* al = smem + stage * STAGE;
        unsigned char* bl = al + TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = cm[j] < m_end_i;
            const unsigned aoff = (live && a_ok) ? aoffv[j] : OOB;
            const bool inb = live && b_ok && (unsigned)chi[j] < (unsigned)p.H && (unsigned)cwi[j] < (unsigned)p.W;
            const unsigned boff = inb ? xlin[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr_t)(al + (wv * 16 + j * 4) * ROWB), 16, aoff, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bl + (wv * 16 + j * 4) * ROWB), 16, boff, 0, 0, 0);
            cm[j] += BKP;
            aoffv[j] += a_step;
            cwo[j] += st_r;
            cho[j] += st_q;
            const bool ww = cwo[j] >= p.Wo;
            cwo[j] -= ww ? p.Wo : 0;
            cho[j] += ww ? 1 : 0;
            const bool wh = cho[j] >= p.Ho;   // st_q <= Ho - 1: one wrap is enough
            cho[j] -= wh ? p.Ho : 0;
            cwi[j] += wi_step - (ww ? wi_wrap : 0);
            chi[j] += hi_step + (ww ? p.stride : 0) - (wh ? hi_wrap : 0);
            xlin[j] += x_step + (ww ? x_wrapw : 0u) + (wh ? x_wraph : 0u);
        }
    };

    f32x16 acc[AM][BN];
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < BN; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

    // fragment role: 16-lane group g = lane >> 4: channel block 16 (g & 1) of the 32-wide MFMA tile, k-group g >> 1;
    // lane i of the group reads pixel row (i >> 2), channels 4 (i & 3) .. +3 of the block.  The row's swizzle key k & 3 is
    // (i >> 2) for every fragment of the lane (all other row terms are multiples of 4), so the per-lane LDS address of MFMA
    // tile t32 is one VGPR and (stage, kk, t) are immediate offsets.
    const int gi = lane & 15, gg = lane >> 4;
    const int krow0 = (gg >> 1) * 8 + (gi >> 2);            // + 16 kk + 4 t
    const int chan0 = (gg & 1) * 16 + 4 * (gi & 3);         // + 32 (tile index) within the 128-wide operand tile
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned fa[AM], fb[BN];
#pragma unroll
    for (int i = 0; i < AM; ++i) {
        const int cha = (wc * AM + i) * 32 + chan0;
        fa[i] = lds0 + krow0 * ROWB + (((cha >> 3) ^ (4 * (krow0 & 3))) << 4) + (cha & 4) * 2;
    }
#pragma unroll
    for (int i = 0; i < BN; ++i) {
        const int chb = (wn * BN + i) * 32 + chan0;
        fb[i] = lds0 + TILE + krow0 * ROWB + (((chb >> 3) ^ (4 * (krow0 & 3))) << 4) + (chb & 4) * 2;
    }
    auto tr_frag = [&](unsigned base, auto stage_kk) -> frag {   // stage_kk: integral_constant<int, stage * 4 + kk>
        constexpr int SK = decltype(stage_kk)::value;
        constexpr int OFF = (SK >> 2) * STAGE + (SK & 3) * 16 * ROWB;
        const s16x4_t v0 = lds_read_tr16<OFF>(base), v1 = lds_read_tr16<OFF + 4 * ROWB>(base);
        const s16x8_t r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(frag, r);
    };
    auto mma = [&](const frag (&af)[AM], const frag (&bf)[BN]) {
#pragma unroll
        for (int a = 0; a < AM; ++a)
#pragma unroll
            for (int b = 0; b < BN; ++b) {
                if constexpr (std::is_same<T, f16_t>::value) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
    };
    // the reads of k-substep kk+1 are in flight under the MFMAs of kk; the lgkmcnt(0) behind the MFMAs finds them landed
    auto compute = [&](auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        frag a0[AM], b0[BN], a1[AM], b1[BN];
        auto rd = [&](auto kk_c, frag (&af)[AM], frag (&bf)[BN]) {
            constexpr int SK = ST * 4 + decltype(kk_c)::value;
#pragma unroll
            for (int i = 0; i < AM; ++i) af[i] = tr_frag(fa[i], std::integral_constant<int, SK>{});
#pragma unroll
            for (int i = 0; i < BN; ++i) bf[i] = tr_frag(fb[i], std::integral_constant<int, SK>{});
        };
        auto landed = [&](frag (&af)[AM], frag (&bf)[BN]) {   // the fragments as operands of the wait: nothing consumes them earlier
            if constexpr (AM == 2 && BN == 2) lds_wait4(af[0], af[1], bf[0], bf[1]);
            else if constexpr (AM == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(bf[0]) : : "memory");
        };
        rd(std::integral_constant<int, 0>{}, a0, b0);
        landed(a0, b0);
        rd(std::integral_constant<int, 1>{}, a1, b1);
        mma(a0, b0);
        landed(a1, b1);
        rd(std::integral_constant<int, 2>{}, a0, b0);
        mma(a1, b1);
        landed(a0, b0);
        rd(std::integral_constant<int, 3>{}, a1, b1);
        mma(a0, b0);
        landed(a1, b1);
        mma(a1, b1);
    };

    const int steps = (int)((m_end - m_begin + BKP - 1) / BKP);
    dma(0, 0);
    for (int it = 0; it < steps; it += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // step `it` has landed for every wave; the other stage is no longer being read
        if (it + 1 < steps) dma(it + 1, 1);
        compute(std::integral_constant<int, 0>{});
        if (it + 1 >= steps) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 2 < steps) dma(it + 2, 0);
        compute(std::integral_constant<int, 1>{});
    }

    // D[row = co][col = n] -> partial tile [n][co] (each lane owns 4 consecutive co: one 16-byte store)
    const int frow = lane & 31, fk = lane >> 5;
    float* tile = p.part + ((size_t)slice_id * gridDim.x + tile_id) * (128 * 128);
#pragma unroll
    for (int b = 0; b < BN; ++b) {
        const int nl = (wn * BN + b) * 32 + frow;
#pragma unroll
        for (int a = 0; a < AM; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wc * AM + a) * 32 + 8 * g + 4 * fk;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(tile + nl * 128 + col) = v;
            }
    }
#endif
}

// ---- wgrad, 256 filters x 256 (tap, channel) columns per block: the conv v6 schedule on the filter-gradient GEMM ---------------
// The 128x128 kernel above reaches ~560 TFLOP/s on the K >= 1152 layers (32 launches, 14 ms of a batch-64 step) against 800-1060
// for the forward kernels on the same FLOPs: per MFMA it stages twice the bytes and issues twice the fragment reads of a
// 256x256 tile, recomputes every row's pixel decomposition in each K-step, and all four waves drain their loads at one barrier.
// Here: 8 waves (64 filters x 128 columns each: 12 transposed fragments for 16 MFMAs per 16 pixels), K-step = 32 pixels, rows of
// 256 channels (512 B) staged in the natural NHWC layout by `buffer_load ... lds` with the XOR slot swizzle of the kernel above
// (it only touches slot bits 2-3, so the bank argument is unchanged), FOUR stages, the two wave halves one barrier interval apart
// (MEM = request tile t+2 + transposed fragment reads of tile t | MMA = 16 MFMAs under s_setprio), counted vmcnt(4), and an
// incremental (image, row, column) cursor per staged row instead of two multiply-shift divisions per row and K-step.
template <typename T>
__global__ __launch_bounds__(512, 2) void wgrad_big_kernel(const WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BKP = 32, ROWB = 512, TILE = BKP * ROWB, STAGE = 2 * TILE, NST = 4;   // 16 KiB per operand and stage, 128 KiB in all
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv >> 1, wn = wv & 1;      // 4 x 2 waves: 64 filters x 128 columns each
    int tile_id, slice_id;
    wgrad_block(p.xcd_group, tile_id, slice_id);
    const int ct = tile_id / p.n_nt, nt = tile_id % p.n_nt;
    const long long m_begin = (long long)slice_id * p.per_slice;
    long long m_end = m_begin + p.per_slice;
    if (m_end > p.M) m_end = p.M;
    if (m_begin >= m_end) return;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.du, 0, (int)p.du_bytes, 0x00020000);
    constexpr unsigned OOB = 0xffffffffu;

    // staging role: wave w fills rows 4w .. 4w+3 of both operands, 2 rows (one 1 KiB piece) per instruction;
    // lane -> (row within the piece = lane / 32, physical slot = lane % 32); piece j holds rows 4w + 2j, 4w + 2j + 1
    const int prow = lane >> 5, pslot = lane & 31;
    int a_ch[2], b_ci[2], b_kh[2], b_kw[2];
    bool a_ok[2], b_ok[2];
    // pixel cursor of the lane's row in piece j: flat output index, output row / column (for the wraps), and -- advanced INCREMENTALLY, no
    // multiplication per K-step (round 3: the per-step `v_mul_lo_u32` / `v_mad_u64_u32` of the offset arithmetic cost more issue time than the
    // requests they fed) -- the input row / column of the lane's tap and the byte offsets of its 16 bytes in du and x
    int cm[2], cho[2], cwo[2], chi[2], cwi[2];
    unsigned aoffv[2], xlin[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 2 * j + prow;                    // row & 3 (4w is a multiple of 4)
        const int lslot = pslot ^ (4 * row);
        a_ch[j] = ct * 256 + lslot * 8;
        a_ok[j] = a_ch[j] < p.Cout;
        const int ncol = nt * 256 + lslot * 8;
        const int tap = ncol / p.Cin;
        b_ci[j] = ncol - tap * p.Cin;
        b_kh[j] = tap / p.ks;
        b_kw[j] = tap - b_kh[j] * p.ks;
        b_ok[j] = tap < p.ks * p.ks;
        const long long m = m_begin + wv * 4 + row;
        const int mm = (int)(m < p.M ? m : p.M - 1);
        cm[j] = (int)m;
        const int n0 = y3_fdiv(mm, p.dv_hw);
        const int rem = mm - n0 * (p.Ho * p.Wo);
        cho[j] = y3_fdiv(rem, p.dv_w);
        cwo[j] = rem - cho[j] * p.Wo;
        chi[j] = cho[j] * p.stride - p.pad + b_kh[j];
        cwi[j] = cwo[j] * p.stride - p.pad + b_kw[j];
        aoffv[j] = ((unsigned)cm[j] * (unsigned)p.dpitch + (unsigned)a_ch[j]) * 2u;
        xlin[j] = (unsigned)(((n0 * p.H + chi[j]) * p.W + cwi[j]) * p.xpitch + b_ci[j]) * 2u;   // meaningful only while (chi, cwi) is inside the image
    }
    const int m_end_i = (int)m_end;
    // what a K-step of 32 pixels = (step_n images, step_q rows, step_r columns) adds, and what a column / row wrap of the cursor adds on top
    const int xp2 = p.xpitch * 2;
    const unsigned a_step = (unsigned)(BKP * p.dpitch * 2);
    const int wi_step = p.step_r * p.stride, wi_wrap = p.Wo * p.stride;
    const int hi_step = p.step_q * p.stride, hi_wrap = p.Ho * p.stride;
    const unsigned x_step = (unsigned)(((p.step_n * p.H + p.step_q * p.stride) * p.W + p.step_r * p.stride) * xp2);
    const unsigned x_wrapw = (unsigned)((p.stride * p.W - p.Wo * p.stride) * xp2);            // column cursor back by Wo, row cursor forward by one
    const unsigned x_wraph = (unsigned)((p.H - p.Ho * p.stride) * p.W * xp2);                 // row cursor back by Ho, image forward by one

    auto dma = [&](int stage) {   // K-steps are requested strictly in order: the cursors advance by 32 pixels per call
        unsigned char* al = smem + stage * STAGE;
        unsigned char* bl = al + TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool live = cm[j] < m_end_i;
            const unsigned aoff = (live && a_ok[j]) ? aoffv[j] : OOB;
            const bool inb = live && b_ok[j] && (unsigned)chi[j] < (unsigned)p.H && (unsigned)cwi[j] < (unsigned)p.W;
            const unsigned boff = inb ? xlin[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr_t)(al + (wv * 4 + j * 2) * ROWB), 16, aoff, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bl + (wv * 4 + j * 2) * ROWB), 16, boff, 0, 0, 0);
            cm[j] += BKP;
            aoffv[j] += a_step;
            cwo[j] += p.step_r;
            cho[j] += p.step_q;
            const bool ww = cwo[j] >= p.Wo;
            cwo[j] -= ww ? p.Wo : 0;
            cho[j] += ww ? 1 : 0;
            const bool wh = cho[j] >= p.Ho;   // step_q <= Ho - 1: one wrap is enough
            cho[j] -= wh ? p.Ho : 0;
            cwi[j] += wi_step - (ww ? wi_wrap : 0);
            chi[j] += hi_step + (ww ? p.stride : 0) - (wh ? hi_wrap : 0);
            xlin[j] += x_step + (ww ? x_wrapw : 0u) + (wh ? x_wraph : 0u);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

    // fragment role (see wgrad_dma_kernel): 16-lane group g = lane >> 4 reads channel block 16 (g & 1) of a 32-wide MFMA tile for
    // k-group g >> 1; lane i of the group addresses pixel row (i >> 2), channels 4 (i & 3) .. +3 and receives channel i's 4 pixels.
    // One address VGPR per MFMA tile of the wave (2 filter tiles, 4 column tiles); (kk, t) are immediate offsets, the stage base
    // is added per K-step (4 x 32 KiB does not fit the 16-bit offset field).
    const int gi = lane & 15, gg = lane >> 4;
    const int krow0 = (gg >> 1) * 8 + (gi >> 2);            // + 16 kk + 4 t
    const int chan0 = (gg & 1) * 16 + 4 * (gi & 3);         // + 32 (tile index) within the 256-wide operand tile
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned fa[2], fb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = (wc * 2 + i) * 32 + chan0;
        fa[i] = lds0 + krow0 * ROWB + (((ch >> 3) ^ (4 * (krow0 & 3))) << 4) + (ch & 4) * 2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = (wn * 4 + i) * 32 + chan0;
        fb[i] = lds0 + TILE + krow0 * ROWB + (((ch >> 3) ^ (4 * (krow0 & 3))) << 4) + (ch & 4) * 2;
    }
    auto tr_frag = [&](unsigned addr, auto kk_c) -> frag {
        constexpr int OFF = decltype(kk_c)::value * 16 * ROWB;
        const s16x4_t v0 = lds_read_tr16<OFF>(addr), v1 = lds_read_tr16<OFF + 4 * ROWB>(addr);
        const s16x8_t r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(frag, r);
    };
    auto load_frags = [&](int stage, auto kk_c, frag (&af)[2], frag (&bf)[4]) {
        const unsigned sb = (unsigned)(stage * STAGE);
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = tr_frag(fa[a] + sb, kk_c);
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = tr_frag(fb[b] + sb, kk_c);
    };
    auto mma = [&](const frag (&af)[2], const frag (&bf)[4]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (std::is_same<T, f16_t>::value) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
    };

    const int steps = (m_end_i - (int)m_begin + BKP - 1) / BKP;
    const int half = wv >> 2;   // 0: leading half, 1: trailing half (one barrier interval behind)
    // a wave whose 128 columns lie beyond the last (tap, channel) column multiplies nothing: 9 * 128 = 1152 columns are 4.5 tiles, so on the 128 -> 256 layers the
    // `wn = 1` waves of the fifth column tile -- 10 % of the launch's MFMAs -- only stage their share of the operands and keep the barriers (round 4: the chip is
    // power-bound under these kernels, matrix work that multiplies zeros costs time)
    const bool idle = nt * 256 + wn * 128 >= p.ks * p.ks * p.Cin;
    dma(0);
    if (steps > 1) dma(1);
    if (steps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // tile 0 is visible to everyone
    if (half) __builtin_amdgcn_s_barrier();   // stagger
    for (int it = 0; it < steps; ++it) {
        // ---- MEM(it): request tile it+2, read the fragments of tile it, retire this wave's pieces of tile it+1 ----
        const bool more = it + 2 < steps;
        if (more) dma((it + 2) & 3);   // requests first: issuing them behind the 24 fragment reads measured +0.85 ms per batch-64 step
        frag a0[2], b0[4], a1[2], b1[4];
        if (!idle) {
            load_frags(it & 3, std::integral_constant<int, 0>{}, a0, b0);
            load_frags(it & 3, std::integral_constant<int, 1>{}, a1, b1);
        }
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- MMA(it) ----
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!idle) {
            lds_wait6(a0[0], a0[1], b0[0], b0[1], b0[2], b0[3]);   // the fragment reads landed while the wave sat at the barrier
            lds_wait6(a1[0], a1[1], b1[0], b1[1], b1[2], b1[3]);
            mma(a0, b0);
            mma(a1, b1);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!half) __builtin_amdgcn_s_barrier();  // re-align: every wave has passed 2 steps + 2 barriers

    // D[row = co][col = n] -> partial tile [n][co] (each lane owns 4 consecutive co: one 16-byte store)
    if (idle) return;   // (the reduce never reads columns beyond the last tap)
    const int frow = lane & 31, fk = lane >> 5;
    float* tile = p.part + ((size_t)slice_id * gridDim.x + tile_id) * (256 * 256);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int nl = (wn * 4 + b) * 32 + frow;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wc * 2 + a) * 32 + 8 * g + 4 * fk;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(tile + nl * 256 + col) = v;
            }
    }
#endif
}

// dW (OIHW fp32) = sum over slices of the partial tiles.  Threads follow the partial layout (co fastest) so every
// slice is read fully coalesced; the scattered 4-byte OIHW write happens once per element.  Fixed summation order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int tiles, int n_nt, int slices, int Cin, int ks, int cin_real, int cout_real,
                                                             float* __restrict__ dw, int tsh) {   // tile edge = 1 << tsh (128 or 256)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;   // element of the tile array: (tile, n_local, co_local)
    const int ts = 1 << tsh;
    if (e >= (long long)tiles << (2 * tsh)) return;
    const int col = (int)(e & (ts - 1)), nl = (int)((e >> tsh) & (ts - 1)), tile = (int)(e >> (2 * tsh));
    const int ct = tile / n_nt, nt = tile - ct * n_nt;
    const int co = ct * ts + col, n = nt * ts + nl;
    const int tap = n / Cin, ci = n - tap * Cin;
    if (co >= cout_real || ci >= cin_real || tap >= ks * ks) return;
    const size_t stride = (size_t)tiles << (2 * tsh);
    float a = 0.0f;
#pragma unroll 4
    for (int s = 0; s < slices; ++s) a += part[(size_t)s * stride + e];
    const int kh = tap / ks, kw = tap - kh * ks;
    dw[(((long long)co * cin_real + ci) * ks + kh) * ks + kw] = a;
}

// The same sum, four consecutive filters per thread (one 16-byte non-temporal load per slice: the partials are read once) with up to 8
// slices in flight.  The one-element kernel above was latency-bound -- 4-byte loads, two dependent batches of 4 per thread: 104 us for the
// 151 MB of the 512 -> 1024 layer (1.45 TB/s).  Same summation order per element, so the result is bit-identical.
// G slice groups per block (round 6): a launch with few units and many slices (64 -> 32 1x1: 4096 units x 512 slices) ran as 16 blocks of threads that each walked
// every slice -- 0.1-0.4 ms for a few tens of MB.  The 256 threads of a block are 256 / G units x G groups: group g sums ITS run of consecutive slices in order, the
// groups' sums meet in LDS and are added in group order by the threads of group 0.  The order depends on (slices, G) only, G on the launch's shape only: the same bits
// from run to run (G = 1: the order of rounds 2-5).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ part, int tiles, int n_nt, int slices, int Cin, int ks, int cin_real, int cout_real,
                                                              float* __restrict__ dw, int tsh, int G) {
    __shared__ f32x4 red[256];
    const int U = 256 / G, u = (int)threadIdx.x % U, g = (int)threadIdx.x / U;
    const long long e = ((long long)blockIdx.x * U + u) * 4;
    const int ts = 1 << tsh;
    const bool in_range = e < (long long)tiles << (2 * tsh);
    const int col = (int)(e & (ts - 1)), nl = (int)((e >> tsh) & (ts - 1)), tile = (int)(e >> (2 * tsh));
    const int ct = tile / n_nt, nt = tile - ct * n_nt;
    const int co = ct * ts + col, n = nt * ts + nl;
    const int tap = n / Cin, ci = n - tap * Cin;
    const bool live = in_range && !(co >= cout_real || ci >= cin_real || tap >= ks * ks);
    const size_t stride = (size_t)tiles << (2 * tsh);
    f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const int per = (slices + G - 1) / G;
        int s = g * per;
        const int s_end = s + per < slices ? s + per : slices;
        const float* src = part + e;
        for (; s + 8 <= s_end; s += 8) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(s + q) * stride));
#pragma unroll
            for (int q = 0; q < 8; ++q) a += v[q];
        }
        if (s < s_end) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (s + q < s_end) v[q] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(s + q) * stride));
#pragma unroll
            for (int q = 0; q < 8; ++q) if (s + q < s_end) a += v[q];
        }
    }
    if (G > 1) {
        red[threadIdx.x] = a;
        __syncthreads();
        if (g != 0) return;
        for (int q = 1; q < G; ++q) a += red[q * U + u];
    }
    if (!live) return;
    const int kk = ks * ks;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (co + q < cout_real) dw[((long long)(co + q) * cin_real + ci) * kk + tap] = a[q];
}

// slice groups per block for a slab sum of `units` 16-byte units over `slices` slabs: enough threads to fill the chip, at least 8 slabs per group
static int slab_sum_groups(long long units, long long slices) {
    int G = 1;
    while (G < 32 && units * G < 65536 && slices >= 16LL * G) G *= 2;
    return G;
}

// per-channel sum of an NHWC tensor into fp32 (bias gradient of the Detect convs)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ g, int pitch, long long M, int C, float* __restrict__ out) {
    const int c = blockIdx.x;
    __shared__ float red[256];
    float a = 0.0f;
    for (long long m = threadIdx.x; m < M; m += 256) a += to_f32<T>(g[m * pitch + c]);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// OIHW fp32 -> data-gradient filter bank: rows = cin, K = (kh', kw', cout) with the taps flipped
template <typename T>
__global__ void pack_filter_dgrad_kernel(const float* __restrict__ src, int cout_src, int cin_src, int ks, int cout, int rows, int kpad, T* __restrict__ dst, int frag) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)rows * kpad;
    if (idx >= total) return;
    const int k = (int)(idx % kpad);
    const int ci = (int)(idx / kpad);
    float v = 0.0f;
    if (ci < cin_src && k < ks * ks * cout) {
        const int tap = k / cout, co = k - tap * cout;
        const int kh = ks - 1 - tap / ks, kw = ks - 1 - tap % ks;
        if (co < cout_src) v = src[(((long long)co * cin_src + ci) * ks + kh) * ks + kw];
    }
    dst[idx] = from_f32<T>(v);
    if (frag && k < 9 * cout) dst[total + y3_frag_index(ci, k, cout)] = from_f32<T>(v);   // the copy conv_v10.h reads (y3_common.h)
}

// both filter banks of a training step from the fp32 master weights in one launch: the forward bank [cout][kh][kw][cin] and the
// data-gradient bank [cin][kh'][kw'][cout] (flipped taps) -- 145 ~7 us pack launches per step become 75
template <typename T>
__global__ void pack_filter_pair_kernel(const float* __restrict__ src, int cout_src, int cin_src, int ks, int cout, int cin, int rows_f, int kpad_f, int rows_d, int kpad_d,
                                        T* __restrict__ dst_f, T* __restrict__ dst_d) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long long)rows_f * kpad_f) {
        const int k = (int)(idx % kpad_f), co = (int)(idx / kpad_f);
        float v = 0.0f;
        if (co < cout_src && k < ks * ks * cin) {
            const int tap = k / cin, ci = k - tap * cin;
            const int kh = tap / ks, kw = tap - kh * ks;
            if (ci < cin_src) v = src[(((long long)co * cin_src + ci) * ks + kh) * ks + kw];
        }
        dst_f[idx] = from_f32<T>(v);
        if (y3_filter_has_frag(cout, cin, ks) && k < 9 * cin) dst_f[(long long)rows_f * kpad_f + y3_frag_index(co, k, cin)] = from_f32<T>(v);   // the copies conv_v10.h reads
    }
    if (idx < (long long)rows_d * kpad_d) {
        const int k = (int)(idx % kpad_d), ci = (int)(idx / kpad_d);
        float v = 0.0f;
        if (ci < cin_src && k < ks * ks * cout) {
            const int tap = k / cout, co = k - tap * cout;
            const int kh = ks - 1 - tap / ks, kw = ks - 1 - tap % ks;
            if (co < cout_src) v = src[(((long long)co * cin_src + ci) * ks + kh) * ks + kw];
        }
        dst_d[idx] = from_f32<T>(v);
        if (y3_filter_has_frag(cin, cout, ks) && k < 9 * cout) dst_d[(long long)rows_d * kpad_d + y3_frag_index(ci, k, cout)] = from_f32<T>(v);
    }
}

// Every layer's banks in ONE launch: the training step re-packs all filters each step (the fp32 master weights moved); 75 pack launches of 4-40 us (0.8 ms per
// batch-64 step, most of it launch-to-launch latency) became one in round 3.  Round 5: SOURCE-indexed tiles.  The destination-indexed form (a thread per bank element,
// like the single-layer packers above) gathered the weights with a stride of 9 floats for the forward bank and of 9 Cin floats for the data-gradient bank -- 1.6 GB of
// reads for 248 MB of weights, 0.54 ms.  Now a block owns a 32-filter x 32-channel tile: it reads the tile's 32 x (32 k k) floats as 32 contiguous runs, keeps them as T
// in LDS ([tap][filter][channel]) and writes the four destinations -- forward bank (32 consecutive channels of a (filter, tap) = 64 bytes), data-gradient bank (32
// consecutive filters of a (channel, flipped tap)), and the fragment-ordered copies of either -- from there.  ONLY the elements that come from a weight are written: row
// padding (filters / channels beyond the source's) and K padding stay what they are, so the caller zero-fills a bank ONCE when it allocates it (ops.PackJobs does).
// A block finds its job by a binary search over the jobs' first block (wave-uniform scalar loads).
template <typename T>
__global__ __launch_bounds__(256) void pack_filter_jobs_kernel(const y3_pack_job* __restrict__ jobs, int n_jobs) {
    __shared__ T tile[9][32][34];   // (34: the transposed read of the data-gradient pass walks the filter index: 17 dwords apart)
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const y3_pack_job j = jobs[lo];
    const float* __restrict__ src = j.w;
    const int ks = j.ksize, KK = ks * ks, cin = j.cin, cout = j.cout;
    const int n_tci = (cin + 31) / 32;
    const int t = (int)blockIdx.x - j.first_block;
    const int co0 = (t / n_tci) * 32, ci0 = (t % n_tci) * 32;
    if (co0 >= j.cout_src || ci0 >= j.cin_src) return;   // a tile of padding only (uniform)
    const int tid = threadIdx.x;
    for (int e = tid; e < 32 * 32 * KK; e += 256) {
        const int co_l = e / (32 * KK), r = e - co_l * (32 * KK);
        const int ci_l = r / KK, tap = r - ci_l * KK;
        const int co = co0 + co_l, ci = ci0 + ci_l;
        float v = 0.0f;
        if (co < j.cout_src && ci < j.cin_src) v = src[((long long)co * j.cin_src + ci) * KK + tap];
        tile[tap][co_l][ci_l] = from_f32<T>(v);
    }
    __syncthreads();
    const int rows_f = (cout + 127) / 128 * 128, kpad_f = (KK * cin + 63) / 64 * 64;
    const int rows_d = (cin + 127) / 128 * 128, kpad_d = (KK * cout + 63) / 64 * 64;
    if (j.packed_fwd) {
        T* __restrict__ dst = (T*)j.packed_fwd;
        const bool frag = y3_filter_has_frag(cout, cin, ks);
        for (int e = tid; e < 32 * 32 * KK; e += 256) {
            const int ci_l = e & 31, co_l = (e >> 5) & 31, tap = e >> 10;
            const int co = co0 + co_l, ci = ci0 + ci_l;
            if (co < j.cout_src && ci < j.cin_src) {
                const T v = tile[tap][co_l][ci_l];
                const int k = tap * cin + ci;
                dst[(long long)co * kpad_f + k] = v;
                if (frag) dst[(long long)rows_f * kpad_f + y3_frag_index(co, k, cin)] = v;   // the copy conv_v10.h reads (y3_common.h)
            }
        }
    }
    if (j.packed_dgrad) {
        T* __restrict__ dst = (T*)j.packed_dgrad;
        const bool frag = y3_filter_has_frag(cin, cout, ks);
        for (int e = tid; e < 32 * 32 * KK; e += 256) {
            const int co_l = e & 31, ci_l = (e >> 5) & 31, tap = e >> 10;
            const int co = co0 + co_l, ci = ci0 + ci_l;
            if (co < j.cout_src && ci < j.cin_src) {
                const T v = tile[tap][co_l][ci_l];
                const int k = (KK - 1 - tap) * cout + co;   // flipped tap (kh, kw) -> (ks - 1 - kh, ks - 1 - kw)
                dst[(long long)ci * kpad_d + k] = v;
                if (frag) dst[(long long)rows_d * kpad_d + y3_frag_index(ci, k, cout)] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward of nearest x2 upsampling: dx[h,w] (+)= sum of the 2x2 block of dy
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, int N, int H, int W, int C, int dpitch, T* __restrict__ dx, int xpitch,
                                                               int accumulate) {
    constexpr int V = V16<T>::N;
    const int cv = C / V;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H * W * cv) return;
    const int c0 = (int)(idx % cv) * V;
    long long t = idx / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float a[V];
    T* o = dx + ((long long)(n * H + h) * W + w) * xpitch + c0;
    if (accumulate) {
        const V16<T> prev = *(const V16<T>*)o;
#pragma unroll
        for (int q = 0; q < V; ++q) a[q] = to_f32<T>(prev.v[q]);
    } else {
#pragma unroll
        for (int q = 0; q < V; ++q) a[q] = 0.0f;
    }
    for (int dy_ = 0; dy_ < 2; ++dy_)
        for (int dx_ = 0; dx_ < 2; ++dx_) {
            const V16<T> g = *(const V16<T>*)(dy + ((long long)(n * 2 * H + 2 * h + dy_) * (2 * W) + 2 * w + dx_) * dpitch + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) a[q] += to_f32<T>(g.v[q]);
        }
    V16<T> out;
#pragma unroll
    for (int q = 0; q < V; ++q) out.v[q] = from_f32<T>(a[q]);
    *(V16<T>*)o = out;
}

// backward of MaxPool2d(k, s, p) (+ right/bottom zero pad): each output's gradient goes to the FIRST maximal input of
// its window (row-major scan, as ATen's max_pool2d_with_indices does); one thread per input element gathers.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, const T* __restrict__ dy, int Ho, int Wo,
                                                            int dpitch, T* __restrict__ dx, int gpitch, int k, int s, int pad, int zr, int zb, int accumulate) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H * W * C) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float mine = to_f32<T>(x[((long long)(n * H + h) * W + w) * xpitch + c]);
    float g = 0.0f;
    // outputs whose window contains (h, w)
    for (int ho = (h + pad - k + s) / s > 0 ? (h + pad - k + s) / s : 0; ho < Ho && ho * s - pad <= h; ++ho) {
        for (int wo = (w + pad - k + s) / s > 0 ? (w + pad - k + s) / s : 0; wo < Wo && wo * s - pad <= w; ++wo) {
            // is (h,w) the first maximum of window (ho,wo)?
            bool first = true;
            for (int kh = 0; kh < k && first; ++kh) {
                const int hi = ho * s - pad + kh;
                if (hi < 0 || hi >= H + zb) continue;
                for (int kw = 0; kw < k; ++kw) {
                    const int wi = wo * s - pad + kw;
                    if (wi < 0 || wi >= W + zr) continue;
                    const float v = (hi < H && wi < W) ? to_f32<T>(x[((long long)(n * H + hi) * W + wi) * xpitch + c]) : 0.0f;
                    const bool before = (hi < h) || (hi == h && wi < w);
                    if (v > mine || (v == mine && before)) { first = false; break; }
                }
            }
            if (first) g += to_f32<T>(dy[((long long)(n * Ho + ho) * Wo + wo) * dpitch + c]);
        }
    }
    T* o = dx + ((long long)(n * H + h) * W + w) * gpitch + c;
    *o = from_f32<T>(accumulate ? to_f32<T>(*o) + g : g);
}

// The same backward in two passes over a byte per output (round 6).  The gather above tests, for every input element and every window that contains it, whether the
// element is the window's first maximum by scanning the window: k^4 comparisons per element -- 28 561 at k = 13, 210 ms for the three pools of yolov3-spp at batch 64
// (the whole yolov3 step is 56 ms).  Here pass 1 scans each window ONCE and records where its first maximum is (kh * k + kw, row-major scan with a strict `>`: ATen's
// max_pool2d_with_indices), pass 2 lets every input element look its k^2 windows up: k^2 loads of a byte instead of k^2 window scans.  Same windows in the same
// (ho, wo) order: the sums are the gather's bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, int Ho, int Wo, int k, int s, int pad, int zr,
                                                               int zb, unsigned char* __restrict__ idx) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)N * Ho * Wo * C) return;
    const int c = (int)(id % C);
    long long t = id / C;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best = 0.0f;
    int bi = 255;   // 255 = no position inside the (zero-padded) input: such a window hands its gradient to nobody
    for (int kh = 0; kh < k; ++kh) {
        const int hi = ho * s - pad + kh;
        if (hi < 0 || hi >= H + zb) continue;
        for (int kw = 0; kw < k; ++kw) {
            const int wi = wo * s - pad + kw;
            if (wi < 0 || wi >= W + zr) continue;
            const float v = (hi < H && wi < W) ? to_f32<T>(x[((long long)(n * H + hi) * W + wi) * xpitch + c]) : 0.0f;   // (the ZeroPad2d region takes part in the maximum)
            if (bi == 255 || v > best) { best = v; bi = kh * k + kw; }
        }
    }
    idx[id] = (unsigned char)bi;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_indexed_kernel(const unsigned char* __restrict__ idx, int N, int H, int W, int C, const T* __restrict__ dy, int Ho, int Wo,
                                                                    int dpitch, T* __restrict__ dx, int gpitch, int k, int s, int pad, int accumulate) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)N * H * W * C) return;
    const int c = (int)(id % C);
    long long t = id / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float g = 0.0f;
    for (int ho = (h + pad - k + s) / s > 0 ? (h + pad - k + s) / s : 0; ho < Ho && ho * s - pad <= h; ++ho) {
        const int kh = h - (ho * s - pad);
        for (int wo = (w + pad - k + s) / s > 0 ? (w + pad - k + s) / s : 0; wo < Wo && wo * s - pad <= w; ++wo) {
            const int kw = w - (wo * s - pad);
            const long long o = ((long long)(n * Ho + ho) * Wo + wo);
            if (idx[o * C + c] == kh * k + kw) g += to_f32<T>(dy[o * dpitch + c]);
        }
    }
    T* o = dx + ((long long)(n * H + h) * W + w) * gpitch + c;
    *o = from_f32<T>(accumulate ? to_f32<T>(*o) + g : g);
}

// the two passes with one 16-byte vector of channels per thread (C, the pitches and the pointers allow it: every layer of the yolov3 family): the element-per-thread
// kernels move 2 bytes per lane and access -- the six 2 x 2 pools of yolov3-tiny ran at ~1 TB/s on maps of up to 839 MB (8.5 ms of a 16.4 ms batch-64 step)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_vec_kernel(const T* __restrict__ x, int N, int H, int W, int C, int xpitch, int Ho, int Wo, int k, int s, int pad, int zr,
                                                                   int zb, unsigned char* __restrict__ idx) {
    constexpr int V = V16<T>::N;
    const int CG = C / V;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)N * Ho * Wo * CG) return;
    const int cg = (int)(id % CG);
    long long t = id / CG;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best[V];
    int bi[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { best[q] = 0.0f; bi[q] = 255; }
    for (int kh = 0; kh < k; ++kh) {
        const int hi = ho * s - pad + kh;
        if (hi < 0 || hi >= H + zb) continue;
        for (int kw = 0; kw < k; ++kw) {
            const int wi = wo * s - pad + kw;
            if (wi < 0 || wi >= W + zr) continue;
            V16<T> v;
            const bool in = hi < H && wi < W;
            if (in) v = ldv<false, T>(x + ((long long)(n * H + hi) * W + wi) * xpitch + cg * V);
            const int code = kh * k + kw;
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const float f = in ? to_f32<T>(v.v[q]) : 0.0f;
                const bool take = bi[q] == 255 || f > best[q];
                best[q] = take ? f : best[q];
                bi[q] = take ? code : bi[q];
            }
        }
    }
    unsigned char* o = idx + ((long long)(n * Ho + ho) * Wo + wo) * C + cg * V;
#pragma unroll
    for (int q = 0; q < V; q += 4) *(unsigned*)(o + q) = (unsigned)bi[q] | ((unsigned)bi[q + 1] << 8) | ((unsigned)bi[q + 2] << 16) | ((unsigned)bi[q + 3] << 24);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_indexed_vec_kernel(const unsigned char* __restrict__ idx, int N, int H, int W, int C, const T* __restrict__ dy, int Ho, int Wo,
                                                                        int dpitch, T* __restrict__ dx, int gpitch, int k, int s, int pad, int accumulate) {
    constexpr int V = V16<T>::N;
    const int CG = C / V;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)N * H * W * CG) return;
    const int cg = (int)(id % CG);
    long long t = id / CG;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float g[V];
#pragma unroll
    for (int q = 0; q < V; ++q) g[q] = 0.0f;
    for (int ho = (h + pad - k + s) / s > 0 ? (h + pad - k + s) / s : 0; ho < Ho && ho * s - pad <= h; ++ho) {
        const int kh = h - (ho * s - pad);
        for (int wo = (w + pad - k + s) / s > 0 ? (w + pad - k + s) / s : 0; wo < Wo && wo * s - pad <= w; ++wo) {
            const unsigned code = (unsigned)(kh * k + w - (wo * s - pad));
            const long long o = ((long long)(n * Ho + ho) * Wo + wo);
            const unsigned char* ip = idx + o * C + cg * V;
            unsigned iw[V / 4];
#pragma unroll
            for (int q = 0; q < V / 4; ++q) iw[q] = *(const unsigned*)(ip + 4 * q);
            bool any = false;
#pragma unroll
            for (int q = 0; q < V; ++q) any |= ((iw[q / 4] >> (8 * (q & 3))) & 255u) == code;
            if (!any) continue;   // (most windows of a 13 x 13 pool have their maximum elsewhere: skip the gradient load)
            const V16<T> d = ldv<false, T>(dy + o * dpitch + cg * V);
#pragma unroll
            for (int q = 0; q < V; ++q)
                if (((iw[q / 4] >> (8 * (q & 3))) & 255u) == code) g[q] += to_f32<T>(d.v[q]);
        }
    }
    T* o = dx + ((long long)(n * H + h) * W + w) * gpitch + cg * V;
    V16<T> r;
    if (accumulate) {
        const V16<T> old = ldv<false, T>(o);
#pragma unroll
        for (int q = 0; q < V; ++q) r.v[q] = from_f32<T>(to_f32<T>(old.v[q]) + g[q]);
    } else {
#pragma unroll
        for (int q = 0; q < V; ++q) r.v[q] = from_f32<T>(g[q]);
    }
    stv<false, T>(o, r);
}

// raw-output gradient (bs, na, ny, nx, no) -> head-conv output gradient NHWC (bs, ny, nx, cpad), pad channels = 0
// V = 16 bytes of consecutive head channels per thread: channels of one anchor are consecutive in graw, the (anchor, output) cursor is
// advanced by hand and the pixel decomposed once per thread (a thread per element -- three 64-bit divisions and a 2-byte store each --
// moved the 2 x 210 MB of the 80x80 level at batch 64 at 1.1 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void detect_raw_bwd_kernel(const T* __restrict__ graw, int bs, int na, int ny, int nx, int no, T* __restrict__ ghead, int pitch,
                                                               int cpad) {
    constexpr int V = 16 / (int)sizeof(T);
    const int groups = (cpad + V - 1) / V;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)bs * ny * nx * groups) return;
    const int g = (int)(idx % groups);
    const long long pix = idx / groups;
    const int x = (int)(pix % nx);
    const long long t = pix / nx;
    const int y = (int)(t % ny);
    const int b = (int)(t / ny);
    const int ch0 = g * V;
    int a = ch0 / no, o = ch0 - a * no;
    const long long plane = (long long)ny * nx * no;
    const T* src = graw + ((long long)b * na * ny + y) * nx * no + (long long)x * no;   // + a * plane + o
    T v[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        v[q] = (ch0 + q < na * no) ? src[a * plane + o] : from_f32<T>(0.0f);
        if (++o == no) { o = 0; ++a; }
    }
    T* dst = ghead + pix * pitch + ch0;
    if (ch0 + V <= cpad && (pitch % V) == 0 && ((uintptr_t)ghead & 15) == 0) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 raw;
        __builtin_memcpy(&raw, v, 16);
        *(u4*)dst = raw;
    } else {
        for (int q = 0; q < V && ch0 + q < cpad; ++q) dst[q] = v[q];
    }
}


// ---- layer 0 backward in two passes instead of three ---------------------------------------------------------------------
// Conv(3, 32, 3, 1) has no data gradient: its du (= BatchNorm + SiLU backward of dy, 64 B per 640x640 pixel) has ONE consumer, the
// filter gradient -- 864 MACs per pixel.  Written by the apply pass and read back by the generic wgrad kernel it cost 3.4 GB of HBM
// traffic and a launch that multiplied mostly zero padding (1.0 ms at batch 64).  Here the apply arithmetic (same formulas as
// bn_act_bwd_apply_kernel) runs on a 4 x 64 pixel tile, du is rounded to T exactly as the stored tensor was and dropped into LDS
// channel-major, the image patch is staged as three column-shifted planar copies, and dW[32 filters][27 (tap, channel)] accumulates in
// ONE 32x32 MFMA tile per wave over the block's tiles (persistent blocks; one fp32 partial tile per block, summed in block order).
struct StemBwdArgs {
    const void* x;      // (N, Cin, H, W) source image
    const void* u;      // NHWC conv output of layer 0 (32 channels)
    const void* dy;     // NHWC gradient of the activation output
    int upitch, dpitch;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    const double* sums; // totals (sum dz, sum dz*xhat) per channel, interleaved
    double count;
    float* part;        // [gridDim.x][32 * 32]
    int N, Cin, H, W, tiles_w, tiles_h, n_tiles;
    float divisor;
};
constexpr int SB_TR = 4, SB_TW = 64;
constexpr int SB_ROWB = SB_TR * SB_TW * 2 + 16;   // one channel's 256 pixels + 16 bytes: the 32 rows of an A-fragment read start in 16 different bank slots

template <typename S> Y3_DEV float stem_src_f32(S v) { return (float)v; }

template <typename T, typename S, bool SILU>
__global__ __launch_bounds__(256, 4) void stem_bn_bwd_wgrad_kernel(const StemBwdArgs p) {
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    __shared__ __attribute__((aligned(16))) unsigned char duT[32 * SB_ROWB];
    __shared__ __attribute__((aligned(16))) T patch[3 * 3 * (SB_TR + 2) * SB_TW];   // [kw][c][row][col]: column j of copy kw = image column col0 + j + kw - 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = tid & 3;
    f32x2 sc[4], sh[4], mu[4], is[4], m0[4], m1[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = cg * 8 + q;
        sc[q / 2][q & 1] = p.scale[c]; sh[q / 2][q & 1] = p.shift[c]; mu[q / 2][q & 1] = p.mean[c]; is[q / 2][q & 1] = p.invstd[c];
        m0[q / 2][q & 1] = (float)(p.sums[c * 2] / p.count);
        m1[q / 2][q & 1] = (float)(p.sums[c * 2 + 1] / p.count);
    }
    // MFMA roles: D[filter][n] += A[filter][pixel] * B[pixel][n], n = tap * Cin + channel (< 9 Cin <= 27)
    const int fn = lane & 31, kg = lane >> 5;
    const int b_tap = fn / p.Cin, b_c = fn - b_tap * p.Cin;
    const bool b_ok = fn < 9 * p.Cin;
    const int b_kh = b_tap / 3, b_kw = b_tap - 3 * b_kh;
    const int b_off = b_ok ? (((b_kw * 3 + b_c) * (SB_TR + 2) + wv + b_kh) * SB_TW + 8 * kg) : 0;   // + 16 s; wave wv owns tile row wv
    const int a_off = fn * SB_ROWB + (wv * SB_TW + 8 * kg) * 2;                                        // + 32 s bytes
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    const int hw = p.H * p.W;
    const T* __restrict__ ug = (const T*)p.u;
    const T* __restrict__ dg = (const T*)p.dy;

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int row0 = th * SB_TR, col0 = tw * SB_TW;
        // ---- du of the tile, two horizontally adjacent pixels x 8 channels per thread and trip, channel-major into LDS ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pi = (tid >> 2) + 64 * it;          // pixel pair of the tile
            const int r = pi >> 5, pj = pi & 31;
            const int gh = row0 + r, gw = col0 + 2 * pj;
            const bool v0 = gh < p.H && gw < p.W, v1 = gh < p.H && gw + 1 < p.W;
            const long long pix = (long long)(n * p.H + gh) * p.W + gw;
            V16<T> x0, x1, g0, g1;
            if (v0) { x0 = *(const V16<T>*)(ug + pix * p.upitch + cg * 8); g0 = *(const V16<T>*)(dg + pix * p.dpitch + cg * 8); }
            if (v1) { x1 = *(const V16<T>*)(ug + (pix + 1) * p.upitch + cg * 8); g1 = *(const V16<T>*)(dg + (pix + 1) * p.dpitch + cg * 8); }
            float d0[8], d1[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
                if (v0) {
                    const f32x2 uf = ld2<T>(x0, 2 * q);
                    f32x2 dz = ld2<T>(g0, 2 * q);
                    if (SILU) { const f32x2 z = uf * sc[q] + sh[q]; dz *= silu_grad2(z, sigmoid2(z)); }
                    r0 = sc[q] * (dz - m0[q] - ((uf - mu[q]) * is[q]) * m1[q]);
                }
                if (v1) {
                    const f32x2 uf = ld2<T>(x1, 2 * q);
                    f32x2 dz = ld2<T>(g1, 2 * q);
                    if (SILU) { const f32x2 z = uf * sc[q] + sh[q]; dz *= silu_grad2(z, sigmoid2(z)); }
                    r1 = sc[q] * (dz - m0[q] - ((uf - mu[q]) * is[q]) * m1[q]);
                }
                d0[2 * q] = r0[0]; d0[2 * q + 1] = r0[1]; d1[2 * q] = r1[0]; d1[2 * q + 1] = r1[1];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *(unsigned*)(duT + (cg * 8 + q) * SB_ROWB + (r * SB_TW + 2 * pj) * 2) = pack2<T>(d0[q], d1[q]);
        }
        // ---- image patch, rounded to T as the forward rounded it: rows row0 - 1 .. row0 + 4, columns col0 - 1 .. col0 + 64; every element is
        // loaded once and stored into the (up to three) column-shifted copies it belongs to ----
        const S* __restrict__ xs = (const S*)p.x + (long long)n * p.Cin * hw;
        for (int e = tid; e < p.Cin * (SB_TR + 2) * (SB_TW + 2); e += 256) {
            const int jj = e % (SB_TW + 2);
            int t = e / (SB_TW + 2);
            const int r = t % (SB_TR + 2), c = t / (SB_TR + 2);
            const int gh = row0 + r - 1, gw = col0 + jj - 1;
            float v = 0.0f;
            if ((unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W) v = stem_src_f32<S>(xs[c * hw + gh * p.W + gw]) / p.divisor;
            const T tv = from_f32<T>(v);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int j = jj - kw;
                if ((unsigned)j < (unsigned)SB_TW) patch[((kw * 3 + c) * (SB_TR + 2) + r) * SB_TW + j] = tv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const frag a = *(const frag*)(duT + a_off + 32 * s4);
            frag bf = *(const frag*)(patch + b_off + 16 * s4);
            if (!b_ok) {
#pragma unroll
                for (int q = 0; q < 8; ++q) bf[q] = (T)0.0f;
            }
            if constexpr (std::is_same<T, f16_t>::value) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bf, acc, 0, 0, 0);
        }
        __syncthreads();   // the tile's LDS is rewritten by the next trip
    }
    // ---- the four waves' tiles -> one partial tile per block: D[filter = 8 g + 4 kg + e][n = fn] ----
    float* red = (float*)duT;   // 4 x 1024 floats
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wv * 1024 + (8 * g + 4 * kg + e) * 32 + fn] = acc[4 * g + e];
    __syncthreads();
    float* out = p.part + (long long)blockIdx.x * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = tid + 256 * i;
        out[o] = (red[o] + red[1024 + o]) + (red[2048 + o] + red[3072 + o]);
    }
}

// ---- layer 0 backward by RECOMPUTATION (round 6): u -- the 64 B/pixel pre-BatchNorm output of layer 0, 1.68 GB at batch 64 -- is not stored by the forward any more
// (stem.hip: statistics-only pass + one pass that writes act(bn(u))).  Both backward passes rebuild the tile's u from the image patch they stage anyway: the patch goes
// to LDS a second time as 4-channel pixels (the B operand of stem_conv_kernel), each wave multiplies its tile row (2 x 3 MFMAs with the stem-packed filters held in
// registers) and writes the ROUNDED values -- the same MFMAs on the same operands in the same order as the forward, hence the same bits the stored tensor had -- as
// [pixel][32 channels] rows into LDS, where the elementwise code reads them instead of global memory.
//   REDUCE = true : (sum dz, sum dz xhat) per channel -> one partial row per block (rows 1 .. of `sums`, as channel_reduce_kernel writes them)
//   REDUCE = false: du = BatchNorm + activation backward, dW += du (x) image patch (as stem_bn_bwd_wgrad_kernel)
struct StemRecArgs {
    StemBwdArgs b;
    const void* w0;     // stem-packed layer-0 filters [32][3][16]
    double* part_rows;  // REDUCE: sums + 64 (row 0 = totals)
};
constexpr int SR_PW = SB_TW + 4;   // 4-channel patch columns: one halo column each side + the 4th pixel of the widest fragment read (+ 1 spare)

template <typename T, typename S, bool SILU, bool REDUCE>
__global__ __launch_bounds__(256, 2) void stem_bwd_recompute_kernel(const StemRecArgs q) {
    const StemBwdArgs& p = q.b;
    typedef typename std::conditional<std::is_same<T, f16_t>::value, f16x8, bf16x8>::type frag;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char duT[REDUCE ? 16 : 32 * SB_ROWB];
    __shared__ __attribute__((aligned(16))) T patch[REDUCE ? 8 : 3 * 3 * (SB_TR + 2) * SB_TW];   // [kw][c][row][col] (the filter gradient's B operand)
    __shared__ __attribute__((aligned(16))) unsigned char patch4[(SB_TR + 2) * SR_PW * 8];        // [row][col][4 channels] (conv0's B operand)
    __shared__ __attribute__((aligned(16))) unsigned char utile[SB_TR * SB_TW * 64];             // u of the tile: [pixel][32 channels], 16-byte chunks XOR-swizzled by pixel
    __shared__ double red[REDUCE ? 4 * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = tid & 3;
    f32x2 sc[4], sh[4], mu[4], is[4], m0[4], m1[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        sc[e / 2][e & 1] = p.scale[c]; sh[e / 2][e & 1] = p.shift[c]; mu[e / 2][e & 1] = p.mean[c]; is[e / 2][e & 1] = p.invstd[c];
        if (!REDUCE) {
            m0[e / 2][e & 1] = (float)(p.sums[c * 2] / p.count);
            m1[e / 2][e & 1] = (float)(p.sums[c * 2 + 1] / p.count);
        }
    }
    const int fn = lane & 31, kg = lane >> 5;
    // conv0: lane (filter = fn, fk = kg) holds k = 8 fk + j of each filter row (stem_conv_kernel)
    frag af[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) af[kh] = *(const frag*)((const T*)q.w0 + (fn * 3 + kh) * 16 + kg * 8);
    // filter gradient roles (stem_bn_bwd_wgrad_kernel)
    const int b_tap = fn / p.Cin, b_c = fn - b_tap * p.Cin;
    const bool b_ok = fn < 9 * p.Cin;
    const int b_kh = b_tap / 3, b_kw = b_tap - 3 * b_kh;
    const int b_off = b_ok ? (((b_kw * 3 + b_c) * (SB_TR + 2) + wv + b_kh) * SB_TW + 8 * kg) : 0;
    const int a_off = fn * SB_ROWB + (wv * SB_TW + 8 * kg) * 2;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    double a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.0;
    const int hw = p.H * p.W;
    const T* __restrict__ dg = (const T*)p.dy;
    for (int e = tid; e < (SB_TR + 2) * SR_PW * 2; e += 256) ((unsigned*)patch4)[e] = 0u;   // channel 3 and the two spare columns stay zero for the block's life

    // A block's phases are a serial chain (patch -> conv0 -> elementwise -> filter gradient): with the loads issued where they are used every tile started with a full
    // HBM round trip and the first cut ran at 2.2 TB/s (1.50 ms for the reduction pass against 0.69 ms of the pass over the stored tensor).  Everything a tile reads
    // from memory -- its image patch elements and its four 16-byte pieces of dy per thread -- is therefore requested one tile AHEAD into registers (stem_pair_kernel's
    // scheme): the loads of tile i + 1 fly under the arithmetic of tile i.
    constexpr int NEL = (3 * (SB_TR + 2) * (SB_TW + 2) + 255) / 256;   // patch elements per thread (Cin <= 3)
    int pe_c[NEL], pe_r[NEL], pe_j[NEL];
#pragma unroll
    for (int k = 0; k < NEL; ++k) {
        const int e = tid + k * 256;
        const int jj = e % (SB_TW + 2);
        const int t = e / (SB_TW + 2);
        pe_r[k] = t % (SB_TR + 2);
        pe_c[k] = e < p.Cin * (SB_TR + 2) * (SB_TW + 2) ? t / (SB_TR + 2) : -1;
        pe_j[k] = jj;
    }
    S raw[NEL];
    bool rin[NEL];
    V16<T> gq[4];
    bool gv[4];
    auto fetch = [&](int tile) {
        int b = tile;
        const int tw = b % p.tiles_w; b /= p.tiles_w;
        const int th = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int row0 = th * SB_TR, col0 = tw * SB_TW;
        const S* __restrict__ xs = (const S*)p.x + (long long)n * p.Cin * hw;
#pragma unroll
        for (int k = 0; k < NEL; ++k) {
            const int gh = row0 + pe_r[k] - 1, gw = col0 + pe_j[k] - 1;
            rin[k] = pe_c[k] >= 0 && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            if (rin[k]) raw[k] = xs[pe_c[k] * hw + gh * p.W + gw];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pi = (tid >> 2) + 64 * it;
            const int r = pi >> 5, pj = pi & 31;
            const int gh = row0 + r, gw = col0 + 2 * pj;
            const long long pix = (long long)(n * p.H + gh) * p.W + gw;
            gv[2 * it] = gh < p.H && gw < p.W;
            gv[2 * it + 1] = gh < p.H && gw + 1 < p.W;
            if (gv[2 * it]) gq[2 * it] = *(const V16<T>*)(dg + pix * p.dpitch + cg * 8);
            if (gv[2 * it + 1]) gq[2 * it + 1] = *(const V16<T>*)(dg + (pix + 1) * p.dpitch + cg * 8);
        }
    };
    if ((int)blockIdx.x < p.n_tiles) fetch(blockIdx.x);
    __syncthreads();

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        // ---- this tile's image patch (requested one tile ago), rounded to T as the forward rounded it, into both layouts ----
#pragma unroll
        for (int k = 0; k < NEL; ++k) {
            if (pe_c[k] < 0) continue;
            const T tv = from_f32<T>(rin[k] ? stem_src_f32<S>(raw[k]) / p.divisor : 0.0f);
            *(T*)(patch4 + ((pe_r[k] * SR_PW + pe_j[k]) * 4 + pe_c[k]) * 2) = tv;
            if (!REDUCE) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int j = pe_j[k] - kw;
                    if ((unsigned)j < (unsigned)SB_TW) patch[((kw * 3 + pe_c[k]) * (SB_TR + 2) + pe_r[k]) * SB_TW + j] = tv;
                }
            }
        }
        V16<T> gc[4];
        bool vc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { gc[k] = gq[k]; vc[k] = gv[k]; }
        __syncthreads();
        if (tile + (int)gridDim.x < p.n_tiles) fetch(tile + gridDim.x);   // the next tile's loads fly under this tile's arithmetic
        // ---- u of the tile: wave wv owns tile row wv, two 32-pixel column blocks ----
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x16 ua;
#pragma unroll
            for (int e = 0; e < 16; ++e) ua[e] = 0.0f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned char* src = patch4 + (((wv + kh) * SR_PW + jb * 32 + fn + 2 * kg) * 8);
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 rw;
                rw[0] = *(const unsigned long long*)src;
                rw[1] = *(const unsigned long long*)(src + 8);
                const frag bf = __builtin_bit_cast(frag, rw);
                if constexpr (std::is_same<T, f16_t>::value) ua = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kh], bf, ua, 0, 0, 0);
                else ua = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh], bf, ua, 0, 0, 0);
            }
            const int pix = wv * SB_TW + jb * 32 + fn;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x4 ov;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(pack2<T>(ua[8 * gp + 2 * h], ua[8 * gp + 2 * h + 1]), pack2<T>(ua[8 * gp + 4 + 2 * h], ua[8 * gp + 4 + 2 * h + 1]), false, false);
                    ov[h] = (unsigned)sw[0];
                    ov[2 + h] = (unsigned)sw[1];
                }
                const int chunk = gp * 2 + kg;
                *(u32x4*)(utile + pix * 64 + ((chunk ^ (pix & 3)) << 4)) = ov;
            }
        }
        __syncthreads();
        // ---- the elementwise part: two horizontally adjacent pixels x 8 channels per thread and trip ----
        f32x2 f0[4], f1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f0[e] = f1[e] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pi = (tid >> 2) + 64 * it;          // pixel pair of the tile
            const int r = pi >> 5, pj = pi & 31;
            const bool v0 = vc[2 * it], v1 = vc[2 * it + 1];
            const int lp = r * SB_TW + 2 * pj;
            const V16<T> x0 = *(const V16<T>*)(utile + lp * 64 + ((cg ^ (lp & 3)) << 4));
            const V16<T> x1 = *(const V16<T>*)(utile + (lp + 1) * 64 + ((cg ^ ((lp + 1) & 3)) << 4));
            const V16<T> g0 = gc[2 * it], g1 = gc[2 * it + 1];
            float d0[8], d1[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
                if (v0) {
                    const f32x2 uf = ld2<T>(x0, 2 * e);
                    f32x2 dz = ld2<T>(g0, 2 * e);
                    if (SILU) { const f32x2 z = uf * sc[e] + sh[e]; dz *= silu_grad2(z, sigmoid2(z)); }
                    const f32x2 xh = (uf - mu[e]) * is[e];
                    if (REDUCE) { f0[e] += dz; f1[e] += dz * xh; }
                    else r0 = sc[e] * (dz - m0[e] - xh * m1[e]);
                }
                if (v1) {
                    const f32x2 uf = ld2<T>(x1, 2 * e);
                    f32x2 dz = ld2<T>(g1, 2 * e);
                    if (SILU) { const f32x2 z = uf * sc[e] + sh[e]; dz *= silu_grad2(z, sigmoid2(z)); }
                    const f32x2 xh = (uf - mu[e]) * is[e];
                    if (REDUCE) { f0[e] += dz; f1[e] += dz * xh; }
                    else r1 = sc[e] * (dz - m0[e] - xh * m1[e]);
                }
                d0[2 * e] = r0[0]; d0[2 * e + 1] = r0[1]; d1[2 * e] = r1[0]; d1[2 * e + 1] = r1[1];
            }
            if (!REDUCE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) *(unsigned*)(duT + (cg * 8 + e) * SB_ROWB + (r * SB_TW + 2 * pj) * 2) = pack2<T>(d0[e], d1[e]);
            }
        }
        if (REDUCE) {   // fp32 over the tile's four pixels per thread, fp64 across tiles (channel_reduce_kernel's scheme)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += (double)f0[e / 2][e & 1]; a1[e] += (double)f1[e / 2][e & 1]; }
            __syncthreads();   // utile / patch4 are rewritten by the next trip
            continue;
        }
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const frag a = *(const frag*)(duT + a_off + 32 * s4);
            frag bf = *(const frag*)(patch + b_off + 16 * s4);
            if (!b_ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bf[e] = (T)0.0f;
            }
            if constexpr (std::is_same<T, f16_t>::value) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bf, acc, 0, 0, 0);
        }
        __syncthreads();   // the tile's LDS is rewritten by the next trip
    }
    if constexpr (REDUCE) {
        // lanes with the same cg (lane & 3) hold partial sums of the same 8 channels: butterfly over the other lane bits, then the four waves in wave order
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += __shfl_xor(a0[e], off, 64); a1[e] += __shfl_xor(a1[e], off, 64); }
        if (lane < 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wv * 64 + (cg * 8 + e) * 2] = a0[e]; red[wv * 64 + (cg * 8 + e) * 2 + 1] = a1[e]; }
        }
        __syncthreads();
        if (tid < 64) q.part_rows[(size_t)blockIdx.x * 64 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    } else {
        // ---- the four waves' tiles -> one partial tile per block: D[filter = 8 g + 4 kg + e][n = fn] ----
        float* redf = (float*)duT;   // 4 x 1024 floats
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) redf[wv * 1024 + (8 * g + 4 * kg + e) * 32 + fn] = acc[4 * g + e];
        __syncthreads();
        float* out = p.part + (long long)blockIdx.x * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = tid + 256 * i;
            out[o] = (redf[o] + redf[1024 + o]) + (redf[2048 + o] + redf[3072 + o]);
        }
    }
}

// dW[co][c][kh][kw] = sum over the blocks' partial tiles [co][n = (kh * 3 + kw) * Cin + c], block order (deterministic)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, int cin, int cout, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cout * cin * 9) return;
    const int kw = idx % 3, kh = (idx / 3) % 3, c = (idx / 9) % cin, co = idx / (9 * cin);
    const int o = co * 32 + (kh * 3 + kw) * cin + c;
    float a = 0.0f;
    int b = 0;
    for (; b + 7 < nblocks; b += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = part[(long long)(b + q) * 1024 + o];
#pragma unroll
        for (int q = 0; q < 8; ++q) a += v[q];
    }
    for (; b < nblocks; ++b) a += part[(long long)b * 1024 + o];
    dw[idx] = a;
}

}  // namespace

#define Y3_DISPATCH_T(dtype, EXPR)                            \
    switch (dtype) {                                          \
        case Y3_F16: { typedef f16_t T; EXPR; } break;        \
        case Y3_BF16: { typedef bf16_t T; EXPR; } break;      \
        case Y3_F32: { typedef float T; EXPR; } break;        \
        default: Y3_FAIL("bad dtype %d", (int)(dtype));       \
    }

static int reduce_geometry(int C, int esz, long long M, unsigned& grid) {
    const int V = 16 / esz;
    if (C % V) Y3_FAIL("channel count %d must be a multiple of %d", C, V);
    const int CG = C / V;
    if (CG > 256) Y3_FAIL("channel count %d too large for the per-channel reduction (max %d)", C, 256 * V);
    const int PL = 256 / CG;
    long long g = (M + (long long)PL * 16 - 1) / ((long long)PL * 16);
    if (g > Y3_BN_PARTIAL_ROWS) g = Y3_BN_PARTIAL_ROWS;  // one partial row of 2*C doubles per block, summed by reduce_partials_kernel
    if (g < 1) g = 1;
    grid = (unsigned)g;
    return 0;
}

// streaming kernels: 4 pixels per thread.  (Round 1 capped the grid at 8192 blocks; measured with tools/lab/bn_lab.hip the uncapped
// grid -- every thread makes its 4 trips and leaves -- streams 839 MB tensors at 6.0-6.3 TB/s instead of 4.9-5.5.)
static int elementwise_geometry(int C, int esz, long long M, unsigned& grid) {
    const int V = 16 / esz;
    if (C % V) Y3_FAIL("channel count %d must be a multiple of %d", C, V);
    const int CG = C / V;
    if (CG > 256) Y3_FAIL("channel count %d too large (max %d)", C, 256 * V);
    const int PL = 256 / CG;
    long long g = (M + (long long)PL * 4 - 1) / ((long long)PL * 4);
    const long long cap = 1ll << 20;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    grid = (unsigned)g;
    return 0;
}

static int bn_stats_launch(const y3_tensor* u, int32_t dtype, double* sums, hipStream_t st, unsigned& grid) {
    if (!vec_ok(u, esize(dtype))) Y3_FAIL("y3_bn_stats: alignment");
    const long long M = (long long)u->n * u->h * u->w;
    if (reduce_geometry(u->c, esize(dtype), M, grid)) return -1;
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((channel_reduce_kernel<T, 0>), dim3(grid), dim3(256), 0, st, (const T*)u->data, u->pitch, (const T*)nullptr, 0, M, u->c,
                                            (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_bn_stats(const y3_tensor* u, int32_t dtype, double* sums, void* stream) {
    if (!u || !sums) Y3_FAIL("y3_bn_stats: null argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned grid;
    if (bn_stats_launch(u, dtype, sums, st, grid)) return -1;
    hipLaunchKernelGGL(reduce_partials_kernel<0>, dim3((2 * u->c + 15) / 16), dim3(256), 0, st, sums, 2 * u->c, (int)grid, BnFinalizeArgs{}, (float*)nullptr, (float*)nullptr, 0);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_bn_finalize(const double* sums, int64_t count, int32_t C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (!sums || !scale || !shift || !mean || !invstd || count <= 0) Y3_FAIL("y3_bn_finalize: bad argument");
    const BnFinalizeArgs f{(double)count, nullptr, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, f);
    Y3_CHECK_LAUNCH();
    return 0;
}

// SyncBatchNorm form (reference train.py:270-272 `--sync-bn`): `sums` holds the totals all-reduced over the ranks, count_dev the all-reduced element count
// (one fp64 on the device: no host round trip between the collective and the normalisation)
extern "C" int y3_bn_finalize_devcount(const double* sums, const double* count_dev, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (!sums || !count_dev || !scale || !shift || !mean || !invstd) Y3_FAIL("y3_bn_finalize_devcount: bad argument");
    const BnFinalizeArgs f{0.0, count_dev, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, f);
    Y3_CHECK_LAUNCH();
    return 0;
}

// first level of the row sum when a launch wrote many rows (the 320x320 layers at batch 64 write 51200): block b adds rows
// [b*per, (b+1)*per) into fp64 partial row 1+b of `sums`, all 2C entries, coalesced along the row
__global__ __launch_bounds__(256) void stat_rows_to_partials_kernel(const float* __restrict__ rows, long long n_rows, int n2c, int per, double* __restrict__ sums) {
    const long long r0 = (long long)blockIdx.x * per;
    long long r1 = r0 + per;
    if (r1 > n_rows) r1 = n_rows;
    for (int j = threadIdx.x; j < n2c; j += 256) {
        double a = 0.0;
        long long r = r0;
        for (; r + 7 < r1; r += 8) {   // 8 loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = rows[(r + q) * n2c + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) a += (double)v[q];
        }
        for (; r < r1; ++r) a += (double)rows[r * n2c + j];
        sums[(size_t)(1 + blockIdx.x) * n2c + j] = a;
    }
}

// BatchNorm finalize from the (sum, sum of squares) rows a y3_conv2d_fwd_stats launch wrote: fixed-order fp64 sum over the rows
// + finalize in one launch (replaces the statistics pass over u)
extern "C" int y3_bn_finalize_rows(const float* stat_rows, int64_t n_rows, int64_t count, int32_t C, double* sums, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (!stat_rows || !sums || !scale || !shift || !mean || !invstd || count <= 0 || n_rows <= 0 || n_rows > 0x7fffffffLL) Y3_FAIL("y3_bn_finalize_rows: bad argument");
    const BnFinalizeArgs f{(double)count, nullptr, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd};
    hipStream_t st = (hipStream_t)stream;
    if (n_rows > Y3_BN_PARTIAL_ROWS) {   // `sums` is a Y3_BN_SCRATCH_DOUBLES(C) buffer: two levels through its partial rows
        const int per = (int)((n_rows + Y3_BN_PARTIAL_ROWS - 1) / Y3_BN_PARTIAL_ROWS);
        const int blocks = (int)((n_rows + per - 1) / per);
        hipLaunchKernelGGL(stat_rows_to_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, st, stat_rows, (long long)n_rows, 2 * C, per, sums);
        Y3_CHECK_LAUNCH();
        hipLaunchKernelGGL((reduce_partials_kernel<1, double>), dim3((2 * C + 15) / 16), dim3(256), 0, st, sums, 2 * C, blocks, f, (float*)nullptr, (float*)nullptr, 0, (const double*)nullptr);
    } else {
        hipLaunchKernelGGL((reduce_partials_kernel<1, float>), dim3((2 * C + 15) / 16), dim3(256), 0, st, sums, 2 * C, (int)n_rows, f, (float*)nullptr, (float*)nullptr, 0, stat_rows);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}

// the row sum of y3_bn_finalize_rows without the finalize: totals (sum, sum of squares) per channel in sums[0 .. 2C) -- what a SyncBatchNorm forward
// all-reduces before y3_bn_finalize_devcount
extern "C" int y3_bn_sum_rows(const float* stat_rows, int64_t n_rows, int32_t C, double* sums, void* stream) {
    if (!stat_rows || !sums || n_rows <= 0 || n_rows > 0x7fffffffLL || C < 1) Y3_FAIL("y3_bn_sum_rows: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n_rows > Y3_BN_PARTIAL_ROWS) {
        const int per = (int)((n_rows + Y3_BN_PARTIAL_ROWS - 1) / Y3_BN_PARTIAL_ROWS);
        const int blocks = (int)((n_rows + per - 1) / per);
        hipLaunchKernelGGL(stat_rows_to_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, st, stat_rows, (long long)n_rows, 2 * C, per, sums);
        Y3_CHECK_LAUNCH();
        hipLaunchKernelGGL((reduce_partials_kernel<0, double>), dim3((2 * C + 15) / 16), dim3(256), 0, st, sums, 2 * C, blocks, BnFinalizeArgs{}, (float*)nullptr, (float*)nullptr, 0, (const double*)nullptr);
    } else {
        hipLaunchKernelGGL((reduce_partials_kernel<0, float>), dim3((2 * C + 15) / 16), dim3(256), 0, st, sums, 2 * C, (int)n_rows, BnFinalizeArgs{}, (float*)nullptr, (float*)nullptr, 0, stat_rows);
    }
    Y3_CHECK_LAUNCH();
    return 0;
}

// y3_bn_stats + y3_bn_finalize in two launches (the partial-row sum and the finalize share one kernel)
extern "C" int y3_bn_stats_finalize(const y3_tensor* u, int32_t dtype, double* sums, const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (!u || !sums || !scale || !shift || !mean || !invstd) Y3_FAIL("y3_bn_stats_finalize: null argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned grid;
    if (bn_stats_launch(u, dtype, sums, st, grid)) return -1;
    const long long M = (long long)u->n * u->h * u->w;
    if (M <= 0) Y3_FAIL("y3_bn_stats_finalize: empty tensor");
    const BnFinalizeArgs f{(double)M, nullptr, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd};
    hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3((2 * u->c + 15) / 16), dim3(256), 0, st, sums, 2 * u->c, (int)grid, f, (float*)nullptr, (float*)nullptr, 0);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_bn_act_fwd(const y3_tensor* u, const float* scale, const float* shift, const y3_tensor* residual, const y3_tensor* y, int32_t dtype, int32_t act,
                             void* stream) {
    if (!u || !scale || !shift || !y) Y3_FAIL("y3_bn_act_fwd: null argument");
    if (y->n != u->n || y->h != u->h || y->w != u->w || y->c != u->c) Y3_FAIL("y3_bn_act_fwd: shape mismatch");
    if (residual && (residual->n != u->n || residual->h != u->h || residual->w != u->w || residual->c != u->c)) Y3_FAIL("y3_bn_act_fwd: residual shape mismatch");
    const int esz = esize(dtype);
    if (!vec_ok(u, esz) || !vec_ok(y, esz) || (residual && !vec_ok(residual, esz))) Y3_FAIL("y3_bn_act_fwd: alignment");
    const long long M = (long long)u->n * u->h * u->w;
    unsigned egrid;
    if (elementwise_geometry(u->c, esz, M, egrid)) return -1;
    const bool nt = M * u->c * esz >= Y3_NT_BYTES;
#define Y3_BN_FWD(SILU, NTS) hipLaunchKernelGGL((bn_act_fwd_kernel<T, SILU, NTS>), dim3(egrid), dim3(256), 0, (hipStream_t)stream, (const T*)u->data, u->pitch, scale, shift, \
                                          residual ? (const T*)residual->data : (const T*)nullptr, residual ? residual->pitch : 0, (T*)y->data, y->pitch, M, u->c)
    Y3_DISPATCH_T(dtype, if (act == Y3_ACT_SILU) { if (nt) Y3_BN_FWD(true, true); else Y3_BN_FWD(true, false); } else { if (nt) Y3_BN_FWD(false, true); else Y3_BN_FWD(false, false); });
#undef Y3_BN_FWD
    Y3_CHECK_LAUNCH();
    return 0;
}

static int bn_act_bwd_check(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                            const double* sums) {
    if (!u || !dy || !scale || !shift || !mean || !invstd || !sums) Y3_FAIL("y3_bn_act_bwd: null argument");
    if (dy->n != u->n || dy->h != u->h || dy->w != u->w || dy->c != u->c) Y3_FAIL("y3_bn_act_bwd: shape mismatch");
    const int esz = esize(dtype);
    if (!vec_ok(u, esz) || !vec_ok(dy, esz)) Y3_FAIL("y3_bn_act_bwd: alignment");
    return 0;
}

// phase 1: per-channel sums of (dz, dz * xhat) over this rank's pixels -> sums[0 .. 2C) (totals), dbeta / dgamma, and the two means the apply pass
// reads in partial row 0 (sums[2C .. 4C))
static int bn_act_bwd_reduce_impl(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                  int32_t act, double* sums, float* dgamma, float* dbeta, void* stream) {
    if (bn_act_bwd_check(u, dy, scale, shift, mean, invstd, dtype, sums)) return -1;
    const int esz = esize(dtype);
    const long long M = (long long)u->n * u->h * u->w;
    unsigned grid;
    if (reduce_geometry(u->c, esz, M, grid)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool nt = M * u->c * esz >= Y3_NT_BYTES;
#define Y3_BN_RED(SILU, NTL) hipLaunchKernelGGL((channel_reduce_kernel<T, 1, SILU, NTL>), dim3(grid), dim3(256), 0, st, (const T*)u->data, u->pitch, (const T*)dy->data, dy->pitch, M, \
                                          u->c, scale, shift, mean, invstd, sums)
    // (the reduction's operands are read again by the apply pass: plain loads keep what fits in the Infinity Cache)
    Y3_DISPATCH_T(dtype, if (act == Y3_ACT_SILU) { if (nt) Y3_BN_RED(true, true); else Y3_BN_RED(true, false); } else { if (nt) Y3_BN_RED(false, true); else Y3_BN_RED(false, false); });
#undef Y3_BN_RED
    Y3_CHECK_LAUNCH();
    // partial rows -> totals + (dbeta, dgamma) + the means the apply pass reads (partial row 0) in the same launch
    BnFinalizeArgs fm{};
    fm.count = (double)M;
    hipLaunchKernelGGL(reduce_partials_kernel<2>, dim3((2 * u->c + 15) / 16), dim3(256), 0, st, sums, 2 * u->c, (int)grid, fm, dbeta, dgamma, 0);
    Y3_CHECK_LAUNCH();
    return 0;
}

// phase 2: du = gamma invstd (dz - mean(dz) - xhat mean(dz xhat)) with the means in sums[2C .. 4C) (the caller's all-reduced ones under SyncBatchNorm)
static int bn_act_bwd_apply_impl(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                 int32_t act, const double* sums, const y3_tensor* du, const y3_tensor* gres, int32_t gres_accumulate, void* stream) {
    if (bn_act_bwd_check(u, dy, scale, shift, mean, invstd, dtype, sums)) return -1;
    if (!du) Y3_FAIL("y3_bn_act_bwd: null argument");
    if (du->c != u->c || du->h != u->h) Y3_FAIL("y3_bn_act_bwd: shape mismatch");
    const int esz = esize(dtype);
    if (!vec_ok(du, esz)) Y3_FAIL("y3_bn_act_bwd: alignment");
    if (gres && (gres->n != u->n || gres->h != u->h || gres->w != u->w || gres->c != u->c || !vec_ok(gres, esz))) Y3_FAIL("y3_bn_act_bwd: residual gradient shape / alignment");
    if (gres && gres->data == dy->data) Y3_FAIL("y3_bn_act_bwd: the residual gradient must not alias dy");
    const long long M = (long long)u->n * u->h * u->w;
    hipStream_t st = (hipStream_t)stream;
    const bool nt = M * u->c * esz >= Y3_NT_BYTES;
    unsigned egrid;
    if (elementwise_geometry(u->c, esz, M, egrid)) return -1;
#define Y3_BN_APPLY(SILU, NT) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, SILU, NT>), dim3(egrid), dim3(256), 0, st, (const T*)u->data, u->pitch, (const T*)dy->data, \
                                            dy->pitch, scale, shift, mean, invstd, (const double*)sums + 2 * u->c, (T*)du->data, du->pitch, M, u->c, \
                                            gres ? (T*)gres->data : (T*)nullptr, gres ? gres->pitch : 0, gres_accumulate)
    Y3_DISPATCH_T(dtype, if (act == Y3_ACT_SILU) { if (nt) Y3_BN_APPLY(true, true); else Y3_BN_APPLY(true, false); } else { if (nt) Y3_BN_APPLY(false, true); else Y3_BN_APPLY(false, false); });
#undef Y3_BN_APPLY
    Y3_CHECK_LAUNCH();
    return 0;
}

static int bn_act_bwd_impl(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                           int32_t act, double* sums, const y3_tensor* du, float* dgamma, float* dbeta, const y3_tensor* gres, int32_t gres_accumulate, void* stream) {
    if (!du) Y3_FAIL("y3_bn_act_bwd: null argument");
    if (bn_act_bwd_reduce_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, dgamma, dbeta, stream)) return -1;
    return bn_act_bwd_apply_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, du, gres, gres_accumulate, stream);
}

// the two phases on their own: a SyncBatchNorm backward all-reduces the totals between them and writes the global means into sums[2C .. 4C)
extern "C" int y3_bn_act_bwd_reduce(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                    int32_t act, double* sums, float* dgamma, float* dbeta, void* stream) {
    return bn_act_bwd_reduce_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, dgamma, dbeta, stream);
}
extern "C" int y3_bn_act_bwd_apply(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                   int32_t act, const double* sums, const y3_tensor* du, const y3_tensor* gres, int32_t gres_accumulate, void* stream) {
    return bn_act_bwd_apply_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, du, gres, gres_accumulate, stream);
}

extern "C" int y3_bn_act_bwd(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                             int32_t act, double* sums, const y3_tensor* du, float* dgamma, float* dbeta, void* stream) {
    return bn_act_bwd_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, du, dgamma, dbeta, nullptr, 0, stream);
}

// ... and the gradient of the residual input of `out = act(bn(conv(x))) + residual` (reference models/common.py:165): gres (+)= dy
extern "C" int y3_bn_act_bwd_res(const y3_tensor* u, const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype,
                                 int32_t act, double* sums, const y3_tensor* du, float* dgamma, float* dbeta, const y3_tensor* gres, int32_t gres_accumulate, void* stream) {
    if (!gres) Y3_FAIL("y3_bn_act_bwd_res: null residual gradient");
    return bn_act_bwd_impl(u, dy, scale, shift, mean, invstd, dtype, act, sums, du, dgamma, dbeta, gres, gres_accumulate, stream);
}

// Layer 0 (no data gradient): BatchNorm + activation backward and the filter gradient without materialising du.
// Pass 1 = the reduction of y3_bn_act_bwd (totals, dgamma, dbeta); pass 2 = stem_bn_bwd_wgrad_kernel + the block-order sum.
extern "C" size_t y3_stem_bn_bwd_wgrad_workspace_bytes(void) { return (size_t)1024 * 1024 * sizeof(float); }

extern "C" int y3_stem_bn_bwd_wgrad(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const y3_tensor* u,
                                    const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype, int32_t act,
                                    double* sums, float* dgamma, float* dbeta, float* dw_oihw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_nchw || !u || !dy || !scale || !shift || !mean || !invstd || !sums || !dw_oihw || !workspace) Y3_FAIL("y3_stem_bn_bwd_wgrad: null argument");
    if (cin < 1 || cin > 3) Y3_FAIL("y3_stem_bn_bwd_wgrad: %d input channels (1..3: the 9 * cin filter taps share one 32-wide MFMA tile)", cin);
    if (u->c != 32 || dy->c != 32) Y3_FAIL("y3_stem_bn_bwd_wgrad: 32 filters (layer 0 of yolov3 / yolov3-spp), got %d", u->c);
    if (u->n != n || u->h != h || u->w != w || dy->n != n || dy->h != h || dy->w != w) Y3_FAIL("y3_stem_bn_bwd_wgrad: shape mismatch");
    if (dtype != Y3_F16 && dtype != Y3_BF16) Y3_FAIL("y3_stem_bn_bwd_wgrad: f16/bf16 only");
    if (!vec_ok(u, 2) || !vec_ok(dy, 2)) Y3_FAIL("y3_stem_bn_bwd_wgrad: alignment");
    if (!(divisor > 0.0f)) Y3_FAIL("y3_stem_bn_bwd_wgrad: divisor must be positive");
    if (workspace_bytes < y3_stem_bn_bwd_wgrad_workspace_bytes() || (((uintptr_t)workspace) & 15)) Y3_FAIL("y3_stem_bn_bwd_wgrad: workspace too small / unaligned");
    const long long M = (long long)n * h * w;
    if (M > 0x7fffffffLL / 4) Y3_FAIL("y3_stem_bn_bwd_wgrad: too many pixels");
    hipStream_t st = (hipStream_t)stream;
    unsigned grid;
    if (reduce_geometry(32, 2, M, grid)) return -1;
    const bool nt = M * 32 * 2 >= Y3_NT_BYTES;
#define Y3_BN_RED(SILU, NTL) hipLaunchKernelGGL((channel_reduce_kernel<T, 1, SILU, NTL>), dim3(grid), dim3(256), 0, st, (const T*)u->data, u->pitch, (const T*)dy->data, dy->pitch, M, \
                                          32, scale, shift, mean, invstd, sums)
    Y3_DISPATCH_T(dtype, if (act == Y3_ACT_SILU) { if (nt) Y3_BN_RED(true, true); else Y3_BN_RED(true, false); } else { if (nt) Y3_BN_RED(false, true); else Y3_BN_RED(false, false); });
#undef Y3_BN_RED
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel<2>, dim3((2 * 32 + 15) / 16), dim3(256), 0, st, sums, 2 * 32, (int)grid, BnFinalizeArgs{}, dbeta, dgamma, 0);
    Y3_CHECK_LAUNCH();
    StemBwdArgs a;
    a.x = x_nchw; a.u = u->data; a.dy = dy->data; a.upitch = u->pitch; a.dpitch = dy->pitch;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.sums = sums; a.count = (double)M;
    a.part = (float*)workspace;
    a.N = n; a.Cin = cin; a.H = h; a.W = w;
    a.tiles_w = (w + SB_TW - 1) / SB_TW; a.tiles_h = (h + SB_TR - 1) / SB_TR;
    const long long tiles = (long long)a.tiles_w * a.tiles_h * n;
    a.n_tiles = (int)tiles;
    a.divisor = divisor;
    const int cap = 4 * y3_cu_count() < 1024 ? 4 * y3_cu_count() : 1024;   // persistent: 4 blocks per CU (the workspace holds 1024 partial tiles)
    const int blocks = tiles < cap ? (int)tiles : cap;
#define Y3_SB(TT, SS) do { if (act == Y3_ACT_SILU) hipLaunchKernelGGL((stem_bn_bwd_wgrad_kernel<TT, SS, true>), dim3(blocks), dim3(256), 0, st, a); \
                           else hipLaunchKernelGGL((stem_bn_bwd_wgrad_kernel<TT, SS, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
#define Y3_SB_SRC(TT) switch (src_dtype) { case Y3_F16: Y3_SB(TT, f16_t); break; case Y3_BF16: Y3_SB(TT, bf16_t); break; case Y3_F32: Y3_SB(TT, float); break; \
                                            case Y3_U8: Y3_SB(TT, unsigned char); break; default: Y3_FAIL("y3_stem_bn_bwd_wgrad: bad source dtype %d", src_dtype); }
    if (dtype == Y3_F16) { Y3_SB_SRC(f16_t) } else { Y3_SB_SRC(bf16_t) }
#undef Y3_SB_SRC
#undef Y3_SB
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((32 * cin * 9 + 255) / 256), dim3(256), 0, st, (const float*)workspace, blocks, cin, 32, dw_oihw);
    Y3_CHECK_LAUNCH();
    return 0;
}

// Layer 0 backward when the forward did not store u (y3_stem_conv_stats_only + y3_stem_conv_fwd_bn): both passes recompute it from the image (stem_bwd_recompute_kernel)
extern "C" int y3_stem_bn_bwd_wgrad_recompute(const void* x_nchw, int32_t src_dtype, int32_t n, int32_t cin, int32_t h, int32_t w, float divisor, const void* packed0,
                                              const y3_tensor* dy, const float* scale, const float* shift, const float* mean, const float* invstd, int32_t dtype, int32_t act,
                                              double* sums, float* dgamma, float* dbeta, float* dw_oihw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_nchw || !packed0 || !dy || !scale || !shift || !mean || !invstd || !sums || !dw_oihw || !workspace) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: null argument");
    if (cin < 1 || cin > 3) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: %d input channels (1..3)", cin);
    if (dy->c != 32 || dy->n != n || dy->h != h || dy->w != w) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: the gradient must be (%d,%d,%d,32)", n, h, w);
    if (dtype != Y3_F16 && dtype != Y3_BF16) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: f16/bf16 only");
    if (!vec_ok(dy, 2) || (((uintptr_t)packed0) & 15)) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: alignment");
    if (!(divisor > 0.0f)) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: divisor must be positive");
    if (workspace_bytes < y3_stem_bn_bwd_wgrad_workspace_bytes() || (((uintptr_t)workspace) & 15)) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: workspace too small / unaligned");
    const long long M = (long long)n * h * w;
    if (M > 0x7fffffffLL / 4) Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: too many pixels");
    hipStream_t st = (hipStream_t)stream;
    StemRecArgs a;
    memset(&a, 0, sizeof(a));
    a.b.x = x_nchw; a.b.u = nullptr; a.b.dy = dy->data; a.b.upitch = 0; a.b.dpitch = dy->pitch;
    a.b.scale = scale; a.b.shift = shift; a.b.mean = mean; a.b.invstd = invstd; a.b.sums = sums; a.b.count = (double)M;
    a.b.part = (float*)workspace;
    a.b.N = n; a.b.Cin = cin; a.b.H = h; a.b.W = w;
    a.b.tiles_w = (w + SB_TW - 1) / SB_TW; a.b.tiles_h = (h + SB_TR - 1) / SB_TR;
    const long long tiles = (long long)a.b.tiles_w * a.b.tiles_h * n;
    a.b.n_tiles = (int)tiles;
    a.b.divisor = divisor;
    a.w0 = packed0;
    a.part_rows = sums + 64;
    const int cap_r = 2 * y3_cu_count() < Y3_BN_PARTIAL_ROWS ? 2 * y3_cu_count() : Y3_BN_PARTIAL_ROWS;   // the scratch holds Y3_BN_PARTIAL_ROWS partial rows
    const int blocks_r = tiles < cap_r ? (int)tiles : cap_r;
    const int cap = 2 * y3_cu_count() < 1024 ? 2 * y3_cu_count() : 1024;   // persistent: 2 blocks per CU (the one-tile-ahead registers put the kernel at ~230 VGPRs; the workspace holds 1024 partial tiles)
    const int blocks = tiles < cap ? (int)tiles : cap;
#define Y3_SR(TT, SS, RED, NB) do { if (act == Y3_ACT_SILU) hipLaunchKernelGGL((stem_bwd_recompute_kernel<TT, SS, true, RED>), dim3(NB), dim3(256), 0, st, a); \
                                    else hipLaunchKernelGGL((stem_bwd_recompute_kernel<TT, SS, false, RED>), dim3(NB), dim3(256), 0, st, a); } while (0)
#define Y3_SR_SRC(TT, RED, NB) switch (src_dtype) { case Y3_F16: Y3_SR(TT, f16_t, RED, NB); break; case Y3_BF16: Y3_SR(TT, bf16_t, RED, NB); break; case Y3_F32: Y3_SR(TT, float, RED, NB); break; \
                                                   case Y3_U8: Y3_SR(TT, unsigned char, RED, NB); break; default: Y3_FAIL("y3_stem_bn_bwd_wgrad_recompute: bad source dtype %d", src_dtype); }
    if (dtype == Y3_F16) { Y3_SR_SRC(f16_t, true, blocks_r) } else { Y3_SR_SRC(bf16_t, true, blocks_r) }
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel<2>, dim3((2 * 32 + 15) / 16), dim3(256), 0, st, sums, 2 * 32, blocks_r, BnFinalizeArgs{}, dbeta, dgamma, 0);
    Y3_CHECK_LAUNCH();
    if (dtype == Y3_F16) { Y3_SR_SRC(f16_t, false, blocks) } else { Y3_SR_SRC(bf16_t, false, blocks) }
#undef Y3_SR_SRC
#undef Y3_SR
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((32 * cin * 9 + 255) / 256), dim3(256), 0, st, (const float*)workspace, blocks, cin, 32, dw_oihw);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_pack_filter_dgrad(const float* w, int32_t cout_src, int32_t cin_src, int32_t ks, int32_t cout, int32_t cin, int32_t dtype, void* packed,
                                    void* stream) {
    if (!w || !packed) Y3_FAIL("y3_pack_filter_dgrad: null pointer");
    if (cout < cout_src || cin < cin_src || (cout % 8) != 0 || (cin % 8) != 0) Y3_FAIL("y3_pack_filter_dgrad: bad padded sizes");
    // the data-gradient convolution has `cin` filters of ks*ks*cout taps: same packed geometry with the roles swapped
    const int rows = y3_filter_rows(cin), kpad = y3_filter_kpad(cout, ks);
    const long long total = (long long)rows * kpad;
    hipStream_t st = (hipStream_t)stream;
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((pack_filter_dgrad_kernel<T>), dim3(nblk(total)), dim3(256), 0, st, w, cout_src, cin_src, ks, cout, rows, kpad, (T*)packed,
                                    y3_filter_has_frag(cin, cout, ks) ? 1 : 0));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_pack_filter_pair(const float* w, int32_t cout_src, int32_t cin_src, int32_t ks, int32_t cout, int32_t cin, int32_t dtype, void* packed_fwd, void* packed_dgrad,
                                   void* stream) {
    if (!w || !packed_fwd || !packed_dgrad) Y3_FAIL("y3_pack_filter_pair: null pointer");
    if (cout < cout_src || cin < cin_src || (cout % 8) != 0 || (cin % 8) != 0) Y3_FAIL("y3_pack_filter_pair: bad padded sizes");
    if (dtype != Y3_F16 && dtype != Y3_BF16) Y3_FAIL("y3_pack_filter_pair: f16/bf16 only");
    const int rows_f = y3_filter_rows(cout), kpad_f = y3_filter_kpad(cin, ks), rows_d = y3_filter_rows(cin), kpad_d = y3_filter_kpad(cout, ks);
    const long long tf = (long long)rows_f * kpad_f, td = (long long)rows_d * kpad_d;
    const long long total = tf > td ? tf : td;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == Y3_F16) hipLaunchKernelGGL((pack_filter_pair_kernel<f16_t>), dim3(nblk(total)), dim3(256), 0, st, w, cout_src, cin_src, ks, cout, cin, rows_f, kpad_f, rows_d, kpad_d, (f16_t*)packed_fwd, (f16_t*)packed_dgrad);
    else hipLaunchKernelGGL((pack_filter_pair_kernel<bf16_t>), dim3(nblk(total)), dim3(256), 0, st, w, cout_src, cin_src, ks, cout, cin, rows_f, kpad_f, rows_d, kpad_d, (bf16_t*)packed_fwd, (bf16_t*)packed_dgrad);
    Y3_CHECK_LAUNCH();
    return 0;
}

// blocks a job of y3_pack_filter_jobs occupies (the caller lays the jobs out back to back: first_block = running sum)
extern "C" int64_t y3_pack_job_blocks(int32_t ksize, int32_t cout, int32_t cin, int32_t want_fwd, int32_t want_dgrad) {
    (void)ksize; (void)want_fwd; (void)want_dgrad;
    return (int64_t)((cout + 31) / 32) * ((cin + 31) / 32);   // one block per 32-filter x 32-channel tile of the weights
}

extern "C" int y3_pack_filter_jobs(const y3_pack_job* jobs_device, int32_t n_jobs, int64_t total_blocks, int32_t dtype, void* stream) {
    if (!jobs_device || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) Y3_FAIL("y3_pack_filter_jobs: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == Y3_F16) hipLaunchKernelGGL((pack_filter_jobs_kernel<f16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st, jobs_device, n_jobs);
    else if (dtype == Y3_BF16) hipLaunchKernelGGL((pack_filter_jobs_kernel<bf16_t>), dim3((unsigned)total_blocks), dim3(256), 0, st, jobs_device, n_jobs);
    else Y3_FAIL("y3_pack_filter_jobs: f16/bf16 only");
    Y3_CHECK_LAUNCH();
    return 0;
}

// knob "wgrad": 0 per shape; 2 the 128x128 kernel; 3 the 256x256 kernel whenever the shape allows it (tests); 4 the direct fp32 kernel
static int wgrad_mode() { return (int)y3_knob(Y3K_WGRAD); }
// 256x256 tiles (one 8-wave block per CU): whole 256-filter tiles, a long reduction and enough columns to fill the tile
static bool wgrad_use_big(const y3_conv_desc* d, long long M) {
    const int mode = wgrad_mode();
    if (d->dtype == Y3_F32 || mode == 2 || mode == 4) return false;
    const bool shape_ok = (d->cout % 256) == 0 && d->ksize * d->ksize * d->cin >= 1024 && M >= 256;
    if (mode == 3) return shape_ok;
    return shape_ok && d->ksize * d->ksize * d->cin >= 1152 && M >= 16384;
}
#include "wgrad_strip.h"
#include "wgrad_patch.h"

static void wgrad_geometry(const y3_conv_desc* d, long long M, int& n_ct, int& n_nt, long long& slices, long long& per, int& tsh) {
    if (wgrad_use_big(d, M)) {
        tsh = 8;
        n_ct = y3_ceil_div(d->cout, 256);
        n_nt = y3_ceil_div(d->ksize * d->ksize * d->cin, 256);
        const long long tiles = (long long)n_ct * n_nt;
        // one block per CU: pick the slice count with the fewest (rounds of 256 blocks) x (pixels per block + a fixed
        // prologue/epilogue cost of ~8 K-steps); every slice writes a 256 KiB partial tile, so ties go to fewer slices
        long long smax = M / 256;   // at least 8 K-steps (of 32 pixels) per block
        if (smax < 1) smax = 1;
        if (smax > 512) smax = 512;
        long long best = 1, best_cost = -1;
        for (long long sl = 1; sl <= smax; ++sl) {
            const long long rounds = (tiles * sl + 255) / 256;
            const long long len = ((M + sl - 1) / sl + 31) / 32 * 32 + 256;
            const long long cost = rounds * len;
            if (best_cost < 0 || cost < best_cost) { best = sl; best_cost = cost; }
        }
        slices = best;
        per = (M + slices - 1) / slices;
        per = (per + 31) / 32 * 32;
        slices = (M + per - 1) / per;
        return;
    }
    tsh = 7;
    n_ct = y3_ceil_div(d->cout, 128);
    n_nt = y3_ceil_div(d->ksize * d->ksize * d->cin, 128);
    const long long tiles = (long long)n_ct * n_nt;
    const long long want_blocks = y3_knob(Y3K_WGRAD_BLOCKS) > 0 ? y3_knob(Y3K_WGRAD_BLOCKS) : 512;
    slices = (want_blocks + tiles - 1) / tiles;                // knob "wgrad_blocks": one round of resident blocks (2 per CU); 1024 = two rounds wrote twice the slabs for no gain (profiles/r06_wgrad_blocks_ab.txt)
    const long long max_slices = (M + 511) / 512;              // at least 8 K-steps (of 64 pixels) per block
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    per = (M + slices - 1) / slices;
    per = (per + 63) / 64 * 64;
    slices = (M + per - 1) / per;
}

static int wgrad_xcd_grouped(const y3_conv_desc* d, long long tiles, long long slices, int tsh) {
    const int xcd_mode = (int)y3_knob(Y3K_WGRAD_XCD);
    const bool narrow = tsh == 7 && d->cout <= 64;
    return (tiles > 1 && slices > 1 && (narrow ? xcd_mode >= 3 : (tsh == 8 ? xcd_mode >= 2 : xcd_mode >= 1))) ? 1 : 0;
}

// geometry y3_conv2d_wgrad would launch for (desc, x): tile edge (128 / 256; 0 = the direct fp32 kernel), pixel slices, and whether a
// slice's tiles are placed back to back on one XCD -- so that tests can assert WHICH form a shape exercises
extern "C" int y3_conv2d_wgrad_plan(const y3_conv_desc* d, const y3_tensor* x, int32_t* tile, int64_t* slices_out, int32_t* xcd_grouped) {
    if (!d || !x || !tile || !slices_out || !xcd_grouped) Y3_FAIL("y3_conv2d_wgrad_plan: null argument");
    const int pad = d->ksize / 2;
    const int Ho = (x->h + 2 * pad - d->ksize) / d->stride + 1, Wo = (x->w + 2 * pad - d->ksize) / d->stride + 1;
    const long long M = (long long)x->n * Ho * Wo;
    const long long xb = (((long long)x->n * x->h * x->w - 1) * x->pitch + x->c) * 2, db_ = ((M - 1) * d->cout + d->cout) * 2;
    if (d->dtype == Y3_F32 || wgrad_mode() == 4 || xb >= 0x7fffffffLL || db_ >= 0x7fffffffLL) {
        *tile = 0; *slices_out = 0; *xcd_grouped = 0;
        return 0;
    }
    StripPlan sp;
    if (strip_plan(d, x->n, x->h, x->w, d->cout, d->cin, false, sp)) {   // (the query has no real channel counts: the padded ones, which is what these layers have)
        *tile = 3; *slices_out = sp.blocks; *xcd_grouped = 0;             // tile 3 = the 3x3 strip kernel (wgrad_strip.h)
        return 0;
    }
    PatchPlan pp;
    if (patch_plan(d, x->n, x->h, x->w, pp)) {
        *tile = 4; *slices_out = pp.slices; *xcd_grouped = 1;             // tile 4 = the padded-position kernel (wgrad_patch.h)
        return 0;
    }
    int n_ct, n_nt, tsh;
    long long slices, per;
    wgrad_geometry(d, M, n_ct, n_nt, slices, per, tsh);
    *tile = 1 << tsh;
    *slices_out = slices;
    *xcd_grouped = wgrad_xcd_grouped(d, (long long)n_ct * n_nt, slices, tsh);
    return 0;
}

extern "C" size_t y3_conv2d_wgrad_workspace_bytes(const y3_conv_desc* d, const y3_tensor* x) {
    if (!d || !x || d->dtype == Y3_F32) return 256;
    const int pad = d->ksize / 2;
    const int Ho = (x->h + 2 * pad - d->ksize) / d->stride + 1, Wo = (x->w + 2 * pad - d->ksize) / d->stride + 1;
    int n_ct, n_nt, tsh;
    long long slices, per;
    wgrad_geometry(d, (long long)x->n * Ho * Wo, n_ct, n_nt, slices, per, tsh);
    size_t need = ((size_t)slices * n_ct * n_nt * sizeof(float)) << (2 * tsh);
    StripPlan sp;
    if (strip_plan(d, x->n, x->h, x->w, d->cout, d->cin, false, sp) && sp.ws_bytes > need) need = sp.ws_bytes;
    PatchPlan pp;
    if (patch_plan(d, x->n, x->h, x->w, pp) && pp.ws_bytes > need) need = pp.ws_bytes;
    return need;
}

extern "C" int y3_conv2d_wgrad(const y3_conv_desc* d, const y3_tensor* x, const y3_tensor* du, int32_t cout_real, int32_t cin_real, float* dw_oihw, float* dbias,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x || !du || !dw_oihw) Y3_FAIL("y3_conv2d_wgrad: null argument");
    if (x->c != d->cin || du->c != d->cout) Y3_FAIL("y3_conv2d_wgrad: channel mismatch");
    const int pad = d->ksize / 2;
    const int Ho = (x->h + 2 * pad - d->ksize) / d->stride + 1, Wo = (x->w + 2 * pad - d->ksize) / d->stride + 1;
    if (du->n != x->n || du->h != Ho || du->w != Wo) Y3_FAIL("y3_conv2d_wgrad: gradient is (%d,%d,%d), expected (%d,%d,%d)", du->n, du->h, du->w, x->n, Ho, Wo);
    if (cout_real > d->cout || cin_real > d->cin) Y3_FAIL("y3_conv2d_wgrad: real sizes exceed padded sizes");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)cout_real * cin_real * d->ksize * d->ksize;
    const long long M = (long long)x->n * Ho * Wo;
    const bool force_direct = wgrad_mode() == 4;
    const long long xb = (((long long)x->n * x->h * x->w - 1) * x->pitch + x->c) * 2, db_ = ((M - 1) * du->pitch + du->c) * 2;
    StripPlan sp;
    if (!force_direct && xb < 0x7fffffffLL && db_ < 0x7fffffffLL && strip_plan(d, x->n, x->h, x->w, cout_real, cin_real, dbias != nullptr, sp)) {
        if (!workspace || workspace_bytes < sp.ws_bytes || (((uintptr_t)workspace) & 15)) Y3_FAIL("y3_conv2d_wgrad: workspace too small or not 16-byte aligned");
        StripArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x->data; a.du = du->data; a.part = (float*)workspace;
        a.N = x->n; a.H = x->h; a.W = x->w; a.xpitch = x->pitch; a.Ho = Ho; a.Wo = Wo; a.dpitch = du->pitch;
        a.x_bytes = (unsigned)xb; a.du_bytes = (unsigned)db_;
        a.strips = sp.strips; a.T = sp.T; a.per = sp.per;
        a.dv_ho = y3_make_divisor(Ho); a.dv_strips = y3_make_divisor(sp.strips);
        if (d->dtype == Y3_F16) launch_strip_t<f16_t>(d, a, sp.blocks, st); else launch_strip_t<bf16_t>(d, a, sp.blocks, st);
        Y3_CHECK_LAUNCH();
        {
            const long long units = (long long)9 * d->cin * d->cout / 4;
            const int G = slab_sum_groups(units, sp.blocks);
            hipLaunchKernelGGL(wgrad_strip_reduce_kernel, dim3((unsigned)((units * G + 255) / 256)), dim3(256), 0, st, (const float*)workspace, sp.blocks, d->cin, d->cout, dw_oihw, G);
        }
        Y3_CHECK_LAUNCH();
        return 0;
    }
    PatchPlan pp;
    if (!force_direct && xb < 0x7fffffffLL && db_ < 0x7fffffffLL && patch_plan(d, x->n, x->h, x->w, pp)) {
        if (!workspace || workspace_bytes < pp.ws_bytes || (((uintptr_t)workspace) & 15)) Y3_FAIL("y3_conv2d_wgrad: workspace too small or not 16-byte aligned");
        if (launch_patch(d, x, du, cout_real, cin_real, dw_oihw, workspace, pp, (unsigned)xb, (unsigned)db_, st)) return -1;
    } else if (d->dtype != Y3_F32 && !force_direct && xb < 0x7fffffffLL && db_ < 0x7fffffffLL) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x->data; a.du = du->data; a.dw = dw_oihw; a.part = (float*)workspace;
        a.N = x->n; a.H = x->h; a.W = x->w; a.Cin = d->cin; a.xpitch = x->pitch; a.Ho = Ho; a.Wo = Wo; a.Cout = d->cout; a.dpitch = du->pitch;
        a.ks = d->ksize; a.stride = d->stride; a.pad = pad; a.cin_real = cin_real; a.cout_real = cout_real; a.M = M;
        a.x_bytes = (unsigned)xb; a.du_bytes = (unsigned)db_;
        a.dv_hw = y3_make_divisor(Ho * Wo); a.dv_w = y3_make_divisor(Wo);
        a.step_n = 32 / (Ho * Wo); a.step_q = (32 % (Ho * Wo)) / Wo; a.step_r = (32 % (Ho * Wo)) % Wo;
        if (M > 0x7fffffffLL) Y3_FAIL("y3_conv2d_wgrad: too many pixels");
        int n_ct, tsh;
        long long slices, per;
        wgrad_geometry(d, M, n_ct, a.n_nt, slices, per, tsh);
        const long long tiles = (long long)n_ct * a.n_nt;
        if (!workspace || workspace_bytes < (((size_t)slices * tiles * sizeof(float)) << (2 * tsh))) Y3_FAIL("y3_conv2d_wgrad: workspace too small");
        a.per_slice = (int)per;
        const dim3 grid((unsigned)tiles, (unsigned)slices);
        // wgrad_block: a slice's tiles back to back on one XCD.  Measured at batch 64 (profiles/r02_wgrad_xcd.txt): 64 -> 128 layers 0.60 -> 0.48 and
        // 0.52 -> 0.45 ms, the 256-tile kernel slightly better, but the narrow (<= 64-filter, 3 column tiles) launches of the 320x320 maps lose
        // 10-15 %, so those keep the dispatch order.  Knob "wgrad_xcd" = 0: never; 1: 128-tile kernels only; 2 (default): + the 256-tile kernel; 3: all
        a.xcd_group = wgrad_xcd_grouped(d, tiles, slices, tsh);
        if (tsh == 8) {
            if (d->dtype == Y3_F16) hipLaunchKernelGGL((wgrad_big_kernel<f16_t>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((wgrad_big_kernel<bf16_t>), grid, dim3(512), 0, st, a);
        } else {
            const int co32 = d->cout <= 32 ? 1 : (d->cout <= 64 ? 2 : 4);   // narrow filter tiles for the <= 64-filter layers (one filter tile, n_ct == 1)
            if (d->dtype == Y3_F16) {
                if (co32 == 1) hipLaunchKernelGGL((wgrad_dma_kernel<f16_t, 1>), grid, dim3(256), 0, st, a);
                else if (co32 == 2) hipLaunchKernelGGL((wgrad_dma_kernel<f16_t, 2>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((wgrad_dma_kernel<f16_t, 4>), grid, dim3(256), 0, st, a);
            } else {
                if (co32 == 1) hipLaunchKernelGGL((wgrad_dma_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
                else if (co32 == 2) hipLaunchKernelGGL((wgrad_dma_kernel<bf16_t, 2>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((wgrad_dma_kernel<bf16_t, 4>), grid, dim3(256), 0, st, a);
            }
        }
        Y3_CHECK_LAUNCH();
        if ((((uintptr_t)workspace) & 15) == 0)   // (the one-element kernel serves workspaces that are not 16-byte aligned)
        {
            const long long units = (tiles << (2 * tsh)) / 4;
            const int G = slab_sum_groups(units, slices);
            hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((units * G + 255) / 256)), dim3(256), 0, st, (const float*)workspace, (int)tiles, a.n_nt, (int)slices, d->cin,
                               d->ksize, cin_real, cout_real, dw_oihw, tsh, G);
        }
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk(tiles << (2 * tsh))), dim3(256), 0, st, (const float*)workspace, (int)tiles, a.n_nt, (int)slices, d->cin, d->ksize, cin_real,
                               cout_real, dw_oihw, tsh);
        Y3_CHECK_LAUNCH();
    } else {
        Y3_HIP(hipMemsetAsync(dw_oihw, 0, (size_t)total * sizeof(float), st));
        // enough pixel slices to fill the machine when the filter is small
        long long want = (256LL * 8 * 256 + total - 1) / total;
        if (want > M / 64) want = M / 64;
        if (want < 1) want = 1;
        if (want > 4096) want = 4096;
        const dim3 grid(nblk(total), (unsigned)want);
        Y3_DISPATCH_T(d->dtype, hipLaunchKernelGGL((wgrad_direct_kernel<T>), grid, dim3(256), 0, st, (const T*)x->data, x->n, x->h, x->w, d->cin, x->pitch,
                                                   (const T*)du->data, Ho, Wo, d->cout, du->pitch, d->ksize, d->stride, pad, cin_real, cout_real, dw_oihw, (int)want));
        Y3_CHECK_LAUNCH();
    }
    if (dbias) {
        // bias gradient = per-channel sum of du.  The filter-gradient workspace is idle again (stream order): it holds the
        // per-block partial rows of the BN reduction kernel, summed in a fixed order (one strided block per channel ran
        // at 0.35 ms per head on batch 64)
        const int esz = esize(d->dtype);
        unsigned grid = 0;
        const size_t row_bytes = (size_t)2 * d->cout * sizeof(double);
        if (vec_ok(du, esz) && d->cout / (16 / esz) <= 256 && workspace && workspace_bytes >= 3 * row_bytes && (((uintptr_t)workspace) & 7) == 0) {
            if (reduce_geometry(d->cout, esz, M, grid)) return -1;
            const size_t fit = workspace_bytes / row_bytes - 1;
            if (grid > fit) grid = (unsigned)fit;
        }
        if (grid) {
            double* sums = (double*)workspace;
            Y3_DISPATCH_T(d->dtype, hipLaunchKernelGGL((channel_reduce_kernel<T, 0>), dim3(grid), dim3(256), 0, st, (const T*)du->data, du->pitch, (const T*)nullptr, 0, M, d->cout,
                                                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums));
            Y3_CHECK_LAUNCH();
            hipLaunchKernelGGL(reduce_partials_kernel<3>, dim3((2 * d->cout + 15) / 16), dim3(256), 0, st, sums, 2 * d->cout, (int)grid, BnFinalizeArgs{}, dbias, (float*)nullptr,
                               cout_real);
            Y3_CHECK_LAUNCH();
        } else {
            Y3_DISPATCH_T(d->dtype, hipLaunchKernelGGL((channel_sum_kernel<T>), dim3((unsigned)cout_real), dim3(256), 0, st, (const T*)du->data, du->pitch, M, d->cout, dbias));
            Y3_CHECK_LAUNCH();
        }
    }
    return 0;
}

extern "C" int y3_upsample2x_bwd(const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t accumulate, void* stream) {
    if (!dy || !dx) Y3_FAIL("y3_upsample2x_bwd: null argument");
    if (dy->n != dx->n || dy->h != 2 * dx->h || dy->w != 2 * dx->w || dy->c != dx->c) Y3_FAIL("y3_upsample2x_bwd: shape mismatch");
    const int esz = esize(dtype);
    if (!vec_ok(dy, esz) || !vec_ok(dx, esz)) Y3_FAIL("y3_upsample2x_bwd: alignment");
    const long long total = (long long)dx->n * dx->h * dx->w * (dx->c / (16 / esz));
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2x_bwd_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)dy->data, dx->n, dx->h, dx->w,
                                            dx->c, dy->pitch, (T*)dx->data, dx->pitch, accumulate));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_maxpool2d_bwd(const y3_tensor* x, const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t k, int32_t stride, int32_t pad, int32_t zr,
                                int32_t zb, int32_t accumulate, void* stream) {
    if (!x || !dy || !dx) Y3_FAIL("y3_maxpool2d_bwd: null argument");
    const int Ho = (x->h + zb + 2 * pad - k) / stride + 1, Wo = (x->w + zr + 2 * pad - k) / stride + 1;
    if (dy->h != Ho || dy->w != Wo || dy->c != x->c || dx->h != x->h || dx->w != x->w || dx->c != x->c) Y3_FAIL("y3_maxpool2d_bwd: shape mismatch");
    const long long total = (long long)x->n * x->h * x->w * x->c;
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x->data, x->n, x->h, x->w, x->c,
                                            x->pitch, (const T*)dy->data, Ho, Wo, dy->pitch, (T*)dx->data, dx->pitch, k, stride, pad, zr, zb, accumulate));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t y3_maxpool2d_bwd_workspace_bytes(const y3_tensor* x, int32_t k, int32_t stride, int32_t pad, int32_t zr, int32_t zb) {
    if (!x || k < 1 || k > 15 || stride < 1) return 0;
    const int Ho = (x->h + zb + 2 * pad - k) / stride + 1, Wo = (x->w + zr + 2 * pad - k) / stride + 1;
    return Ho > 0 && Wo > 0 ? (size_t)x->n * Ho * Wo * x->c : 0;
}

extern "C" int y3_maxpool2d_bwd_ws(const y3_tensor* x, const y3_tensor* dy, const y3_tensor* dx, int32_t dtype, int32_t k, int32_t stride, int32_t pad, int32_t zr,
                                   int32_t zb, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dx) Y3_FAIL("y3_maxpool2d_bwd_ws: null argument");
    const size_t need = y3_maxpool2d_bwd_workspace_bytes(x, k, stride, pad, zr, zb);
    if (!workspace || !need || workspace_bytes < need) return y3_maxpool2d_bwd(x, dy, dx, dtype, k, stride, pad, zr, zb, accumulate, stream);   // (k > 15: an index does not fit a byte)
    const int Ho = (x->h + zb + 2 * pad - k) / stride + 1, Wo = (x->w + zr + 2 * pad - k) / stride + 1;
    if (dy->h != Ho || dy->w != Wo || dy->c != x->c || dx->h != x->h || dx->w != x->w || dx->c != x->c) Y3_FAIL("y3_maxpool2d_bwd_ws: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int esz = esize(dtype);
    if (vec_ok(x, esz) && vec_ok(dy, esz) && vec_ok(dx, esz) && (((uintptr_t)workspace) & 3) == 0) {
        const int vv = 16 / esz;
        Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_argmax_vec_kernel<T>), dim3(nblk((long long)need / vv)), dim3(256), 0, st, (const T*)x->data, x->n, x->h, x->w, x->c, x->pitch,
                                                Ho, Wo, k, stride, pad, zr, zb, (unsigned char*)workspace));
        Y3_CHECK_LAUNCH();
        const long long tot = (long long)x->n * x->h * x->w * (x->c / vv);
        Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_indexed_vec_kernel<T>), dim3(nblk(tot)), dim3(256), 0, st, (const unsigned char*)workspace, x->n, x->h, x->w, x->c,
                                                (const T*)dy->data, Ho, Wo, dy->pitch, (T*)dx->data, dx->pitch, k, stride, pad, accumulate));
        Y3_CHECK_LAUNCH();
        return 0;
    }
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_argmax_kernel<T>), dim3(nblk((long long)need)), dim3(256), 0, st, (const T*)x->data, x->n, x->h, x->w, x->c, x->pitch, Ho, Wo,
                                            k, stride, pad, zr, zb, (unsigned char*)workspace));
    Y3_CHECK_LAUNCH();
    const long long total = (long long)x->n * x->h * x->w * x->c;
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_indexed_kernel<T>), dim3(nblk(total)), dim3(256), 0, st, (const unsigned char*)workspace, x->n, x->h, x->w, x->c,
                                            (const T*)dy->data, Ho, Wo, dy->pitch, (T*)dx->data, dx->pitch, k, stride, pad, accumulate));
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_detect_raw_bwd(const void* graw, int32_t dtype, int32_t bs, int32_t na, int32_t ny, int32_t nx, int32_t no, const y3_tensor* ghead, void* stream) {
    if (!graw || !ghead) Y3_FAIL("y3_detect_raw_bwd: null argument");
    if (ghead->n != bs || ghead->h != ny || ghead->w != nx || ghead->c < na * no) Y3_FAIL("y3_detect_raw_bwd: shape mismatch");
    const int vec = 16 / esize(dtype);
    const long long total = (long long)bs * ny * nx * ((ghead->c + vec - 1) / vec);
    Y3_DISPATCH_T(dtype, hipLaunchKernelGGL((detect_raw_bwd_kernel<T>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, (const T*)graw, bs, na, ny, nx, no,
                                            (T*)ghead->data, ghead->pitch, ghead->c));
    Y3_CHECK_LAUNCH();
    return 0;
}
