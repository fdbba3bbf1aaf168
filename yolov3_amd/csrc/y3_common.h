// Shared device/host helpers for libyolov3_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/yolov3_hip.h"

#define Y3_DEV __device__ __forceinline__

typedef _Float16 f16_t;
typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- error plumbing -------------------------------------------------------------------------
void y3_set_error(const char* fmt, ...);
#define Y3_FAIL(...)               \
    do {                           \
        y3_set_error(__VA_ARGS__); \
        return -1;                 \
    } while (0)
#define Y3_CHECK_LAUNCH()                                                     \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) Y3_FAIL("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
    } while (0)

#define Y3_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) Y3_FAIL("%s:%d %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    } while (0)

// ---- run-time tuning knobs -------------------------------------------------------------------
// The A/B hooks of the library live in ONE table (api.cpp) instead of getenv() calls on the launch paths: defaults are the
// measured-best settings, the environment variable Y3_TUNE ("key=value,key=value") overrides them once at load, and
// y3_tune_set / y3_tune_get / y3_tune_reset (include/yolov3_hip.h) change them at run time (tests, tools/*_lab.py).
#include "y3_knobs.h"

// ---- scalar conversions ----------------------------------------------------------------------
// Bijective remap of the hardware block id (workgroup i runs on XCD i % 8) so that consecutive LOGICAL ids share an XCD: XCD x owns the ids
// [x * nb / 8, (x + 1) * nb / 8) -- neighbouring tiles then meet in one L2 (conv.hip, conv_v10.h, the persistent tile loops of stem.hip)
Y3_DEV int xcd_remap(int b, int nb) {
    const int xcd = b & 7, i = b >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

template <typename T> Y3_DEV float to_f32(T v);
template <> Y3_DEV float to_f32<f16_t>(f16_t v) { return (float)v; }
template <> Y3_DEV float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <> Y3_DEV float to_f32<float>(float v) { return v; }
template <typename T> Y3_DEV T from_f32(float v);
template <> Y3_DEV f16_t from_f32<f16_t>(float v) { return (f16_t)v; }   // round-to-nearest-even
template <> Y3_DEV bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; } // round-to-nearest-even
template <> Y3_DEV float from_f32<float>(float v) { return v; }

// two floats -> one register of two T, round-to-nearest-even: a single v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 on gfx950 (element by
// element the compiler emits v_cvt + v_cvt + v_pack for f16)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> Y3_DEV unsigned pack2(float a, float b);
template <> Y3_DEV unsigned pack2<f16_t>(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}
template <> Y3_DEV unsigned pack2<bf16_t>(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// round a float through T (what torch does after every elementwise op on a half tensor)
template <typename T> Y3_DEV float rt(float v) { return to_f32<T>(from_f32<T>(v)); }

Y3_DEV float silu_f32(float v) { return v / (1.0f + __expf(-v)); }

// SiLU on a whole MFMA accumulator in straight-line code: all the v_exp_f32, then all the v_rcp_f32 (quarter-rate transcendentals,
// two per value: the floor of this op), the rest as packed fp32 (v_pk_mul_f32 / v_pk_add_f32, two values per lane and issue).
// Written per value under a run-time `if (act == SILU)` the compiler kept one uniform branch per value PAIR: basic blocks of two
// dependent exp -> add -> rcp -> mul chains with nothing to fill the transcendental latency.
template <typename V, int N> Y3_DEV void silu_vec(V& a) {
    V e = a * -1.44269504088896f;
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
    e = e + 1.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
    a = a * e;
}

// packed-pair sigmoid / SiLU derivative (the elementwise BatchNorm kernels of train.hip and the BatchNorm-backward statistics of the
// data-gradient epilogue in conv.hip): v_exp_f32 + v_rcp_f32 per value, the rest packed fp32
Y3_DEV f32x2 sigmoid2(f32x2 z) {
    f32x2 e = z * -1.44269504088896f;
    e[0] = __builtin_amdgcn_exp2f(e[0]);
    e[1] = __builtin_amdgcn_exp2f(e[1]);
    e = e + 1.0f;
    e[0] = __builtin_amdgcn_rcpf(e[0]);
    e[1] = __builtin_amdgcn_rcpf(e[1]);
    return e;
}
Y3_DEV f32x2 silu_grad2(f32x2 z, f32x2 s) { return s + z * s * (1.0f - s); }   // d silu(z)/dz with s = sigmoid(z)
// y = act(scale u + shift) (+ shortcut) on a channel pair: ONE definition for the normalise pass (train.hip::bn_act_fwd_kernel) and for the same arithmetic applied on the way
// into a 1x1 consumer (conv_1x1s.h) -- explicit fused multiply-adds and no contraction across the shortcut's add, so both places round alike whatever the optimiser prefers
Y3_DEV f32x2 y3_bn_act2(f32x2 u, f32x2 sc, f32x2 sh, bool silu, bool has_res, f32x2 r) {
#pragma clang fp contract(off)
    f32x2 z = {__builtin_fmaf(u[0], sc[0], sh[0]), __builtin_fmaf(u[1], sc[1], sh[1])};
    if (silu) z = z * sigmoid2(z);
    if (has_res) z = z + r;
    return z;
}

// rows of per-block partial sums behind the 2*C totals of y3_bn_stats / y3_bn_act_bwd scratch buffers
#define Y3_BN_PARTIAL_ROWS 512   // (2048 rows measured slower: the partial-row sum grows faster than the reduction gains)

// q = n / d for 0 <= n < 2^31 as umulhi(n, mul) >> (sh - 1); mul == 0 encodes d == 1.  Host side fills (mul, sh) once per launch.
struct y3_divisor {
    unsigned mul, sh;
};
static inline y3_divisor y3_make_divisor(int d) {
    y3_divisor r;
    if (d <= 1) { r.mul = 0; r.sh = 1; return r; }
    unsigned s = 0;
    while ((1ll << s) < d) ++s;
    r.mul = (unsigned)(((1ull << (31 + s)) / (unsigned long long)d) + 1ull);
    r.sh = s;
    return r;
}
Y3_DEV int y3_fdiv(int n, y3_divisor d) { return d.mul ? (int)(__umulhi((unsigned)n, d.mul) >> (d.sh - 1)) : n; }

static inline int y3_ceil_div(int a, int b) { return (a + b - 1) / b; }
// compute units of the CURRENT device (persistent grids and tile plans are sized from it: 256 on a whole MI355X, fewer on a partition).  Cached per device
// ordinal: a process that drives several devices or partitions of different size (CPX / NPS modes) gets each one's own count
static inline int y3_cu_count() {
    constexpr int MAXD = 64;
    static int cache[MAXD] = {0};   // 0 = not asked yet (a racing first call stores the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev < MAXD && cache[dev] > 0) return cache[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev < MAXD) cache[dev] = cus;
    return cus;
}
static inline size_t y3_round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// packed filter geometry (see conv.hip): rows padded to 128 filters, K padded to 64
static inline int y3_filter_rows(int cout) { return (int)y3_round_up((size_t)cout, 128); }
static inline int y3_filter_kpad(int cin, int ksize) { return (int)y3_round_up((size_t)ksize * ksize * cin, 64); }
// 3x3 banks with rows % 256 == 0 and channels % 32 == 0 carry a second copy behind the row-major one, in the order conv_v10.h consumes it: per
// (256-row filter tile, 64-row wave, K-step) one 4 KiB block of four 1 KiB MFMA A fragments j = 2 kk + a (k-substep kk, rows 32 a .. 32 a + 31), lane = 32 fk + row
// holding k-group 16 kk + 8 fk .. + 7 of the K-step's 32 channels; K-steps in loop order (channel block, tap).  Every packer writes both copies.
__host__ __device__ static inline bool y3_filter_has_frag(int rows, int channels, int ksize) { return ksize == 3 && rows > 0 && (rows % 256) == 0 && channels > 0 && (channels % 32) == 0; }
__host__ __device__ static inline long long y3_frag_index(int row, int k, int channels) {
    const int tap = k / channels, ci = k - tap * channels;
    const int cb = ci >> 5, kk = (ci >> 4) & 1, fk = (ci >> 3) & 1, e = ci & 7;
    const int tw = row >> 6, a = (row >> 5) & 1, fr = row & 31;   // tw = 4 * filter tile + wave
    const int nk = 9 * (channels >> 5);
    return ((((long long)tw * nk + cb * 9 + tap) * 4 + (kk * 2 + a)) * 64 + fk * 32 + fr) * 8 + e;
}
