// Fused optimizer step for gfx950: GradScaler.unscale_ + inf/nan check, clip_grad_norm_ (global L2), SGD with Nesterov
// momentum and per-tensor weight decay / lr, and the ModelEMA lerp -- two launches for the whole model instead of the
// reference's ~5 full passes over 248 MB and hundreds of foreach launches (reference train.py:414-422,
// utils/torch_utils.py:207-237 smart_optimizer, upstream ModelEMA.update).  HBM-bound streaming: one read of grad,
// read+write of param, momentum buffer and EMA.  No host synchronisation: the clip coefficient and the found-inf flag
// stay on the device (a step with inf/nan gradients is skipped, like GradScaler.step).
#include "y3_common.h"

namespace {

constexpr int CHUNK = 16384;  // elements per block

struct OptTensor {            // one entry per parameter tensor, DEVICE memory, built by the host mirror
    float* param;
    const float* grad;
    float* mom;               // momentum buffer (zero-initialised before the first step)
    float* ema;               // may be null
    long long numel;
    float lr, weight_decay;
    int first_chunk;          // prefix sum of chunk counts
    int pad;
};

__device__ int find_tensor(const OptTensor* __restrict__ t, int n, int chunk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// pass 1: sum of squares of the (unscaled) gradients + non-finite detection; one partial per block
__global__ __launch_bounds__(256) void grad_norm_kernel(const OptTensor* __restrict__ t, int n, float inv_scale, const float* __restrict__ scale_dev,
                                                          float* __restrict__ partial, int* __restrict__ found_inf) {
    __shared__ float red[256];
    if (scale_dev) inv_scale *= 1.0f / scale_dev[0];   // dynamic loss scale (GradScaler): lives on the device, no host round trip
    const int ti = find_tensor(t, n, blockIdx.x);
    const OptTensor T = t[ti];
    const long long base = (long long)(blockIdx.x - T.first_chunk) * CHUNK;
    float a = 0.0f;
    bool bad = false;
    for (int i = threadIdx.x; i < CHUNK; i += 256) {
        const long long e = base + i;
        if (e < T.numel) {
            const float g = T.grad[e] * inv_scale;
            bad |= !(fabsf(g) <= 3.402823466e38f);
            a += g * g;
        }
    }
    red[threadIdx.x] = a;
    if (bad) *found_inf = 1;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// total norm (fixed-order sum of the partials) -> clip coefficient  min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int nchunks, float max_norm, float* __restrict__ out /* [norm, coef] */) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) a += (double)partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[0] = norm;
        float c = max_norm > 0.0f ? max_norm / (norm + 1e-6f) : 1.0f;
        out[1] = c < 1.0f ? c : 1.0f;
    }
}

// pass 2: p, buf, ema update (torch.optim.SGD semantics: g += wd*p; buf = first ? g : mu*buf + g; g = nesterov ? g + mu*buf : buf)
__global__ __launch_bounds__(256) void sgd_update_kernel(const OptTensor* __restrict__ t, int n, float inv_scale, const float* __restrict__ scale_dev,
                                                           const float* __restrict__ clip, const int* __restrict__ found_inf, float momentum, int nesterov, int first_step,
                                                           float ema_decay) {
    if (*found_inf) return;  // GradScaler.step skips the update when any gradient is inf/nan
    if (scale_dev) inv_scale *= 1.0f / scale_dev[0];
    const int ti = find_tensor(t, n, blockIdx.x);
    const OptTensor T = t[ti];
    const long long base = (long long)(blockIdx.x - T.first_chunk) * CHUNK;
    const float gs = inv_scale * clip[1];
    for (int i = threadIdx.x; i < CHUNK; i += 256) {
        const long long e = base + i;
        if (e >= T.numel) break;
        float p = T.param[e];
        float g = T.grad[e] * gs;
        if (T.weight_decay != 0.0f) g += T.weight_decay * p;
        float b = first_step ? g : momentum * T.mom[e] + g;
        T.mom[e] = b;
        g = nesterov ? g + momentum * b : b;
        p -= T.lr * g;
        T.param[e] = p;
        if (T.ema) T.ema[e] = ema_decay * T.ema[e] + (1.0f - ema_decay) * p;
    }
}

// torch.cuda.amp.GradScaler.update() (reference train.py:345,416-417; ATen _amp_update_scale_): on a step that found inf/nan the scale
// is multiplied by backoff_factor and the growth counter reset; otherwise the counter advances and every growth_interval clean steps
// the scale is multiplied by growth_factor (unless that overflows fp32).  All on the device.
__global__ void loss_scale_update_kernel(float* __restrict__ scale, int* __restrict__ growth_tracker, const int* __restrict__ found_inf, float growth_factor,
                                         float backoff_factor, int growth_interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*found_inf) {
        *scale = *scale * backoff_factor;
        *growth_tracker = 0;
    } else {
        const int successful = *growth_tracker + 1;
        if (successful == growth_interval) {
            const float grown = *scale * growth_factor;
            if (fabsf(grown) <= 3.402823466e38f) *scale = grown;
            *growth_tracker = 0;
        } else {
            *growth_tracker = successful;
        }
    }
}

// the owner's step of the two-phase gradient exchange (parallel.GradBuckets, exchange = "direct"): out[i] = (parts[0][i] + parts[1][i] + ... + parts[P-1][i]) * scale,
// the P contributions added in rank order (every rank receives this one sum: replicas stay bit-identical), one pass, one launch
__global__ __launch_bounds__(256) void shard_mean_kernel(const float* __restrict__ parts, int P, long long n, float scale, float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n && (((uintptr_t)parts | (uintptr_t)out) & 15) == 0 && (n & 3) == 0) {
        f32x4 a = *(const f32x4*)(parts + i);
        for (int r = 1; r < P; ++r) a += *(const f32x4*)(parts + (long long)r * n + i);
        *(f32x4*)(out + i) = a * scale;
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) {
        float a = parts[j];
        for (int r = 1; r < P; ++r) a += parts[(long long)r * n + j];
        out[j] = a * scale;
    }
}

}  // namespace

extern "C" int y3_shard_mean(const float* parts, int32_t n_parts, int64_t n, float scale, float* out, void* stream) {
    if (!parts || !out || n_parts < 1 || n < 0) Y3_FAIL("y3_shard_mean: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(shard_mean_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, parts, n_parts, (long long)n, scale, out);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t y3_sgd_tensor_record_bytes(void) { return sizeof(OptTensor); }

static int sgd_step_impl(const void* tensor_table, int32_t n_tensors, int32_t n_chunks, float inv_scale, const float* scale_dev, float max_norm, float momentum,
                         int32_t nesterov, int32_t first_step, float ema_decay, float* scratch, int32_t* found_inf, void* stream) {
    if (!tensor_table || !scratch || !found_inf || n_tensors <= 0 || n_chunks <= 0) Y3_FAIL("y3_sgd_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const OptTensor* t = (const OptTensor*)tensor_table;
    Y3_HIP(hipMemsetAsync(found_inf, 0, sizeof(int), st));
    hipLaunchKernelGGL(grad_norm_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, t, n_tensors, inv_scale, scale_dev, scratch + 2, found_inf);
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, (const float*)(scratch + 2), n_chunks, max_norm, scratch);
    Y3_CHECK_LAUNCH();
    hipLaunchKernelGGL(sgd_update_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, t, n_tensors, inv_scale, scale_dev, (const float*)scratch, (const int*)found_inf, momentum,
                       nesterov, first_step, ema_decay);
    Y3_CHECK_LAUNCH();
    return 0;
}

extern "C" int y3_sgd_step(const void* tensor_table, int32_t n_tensors, int32_t n_chunks, float inv_scale, float max_norm, float momentum, int32_t nesterov,
                           int32_t first_step, float ema_decay, float* scratch /* n_chunks + 2 floats */, int32_t* found_inf, void* stream) {
    return sgd_step_impl(tensor_table, n_tensors, n_chunks, inv_scale, nullptr, max_norm, momentum, nesterov, first_step, ema_decay, scratch, found_inf, stream);
}

extern "C" int y3_sgd_step_dynamic(const void* tensor_table, int32_t n_tensors, int32_t n_chunks, const float* loss_scale, float max_norm, float momentum, int32_t nesterov,
                                   int32_t first_step, float ema_decay, float* scratch, int32_t* found_inf, void* stream) {
    if (!loss_scale) Y3_FAIL("y3_sgd_step_dynamic: null loss scale");
    return sgd_step_impl(tensor_table, n_tensors, n_chunks, 1.0f, loss_scale, max_norm, momentum, nesterov, first_step, ema_decay, scratch, found_inf, stream);
}

extern "C" int y3_loss_scale_update(float* loss_scale, int32_t* growth_tracker, const int32_t* found_inf, float growth_factor, float backoff_factor, int32_t growth_interval,
                                    void* stream) {
    if (!loss_scale || !growth_tracker || !found_inf || growth_interval < 1) Y3_FAIL("y3_loss_scale_update: bad argument");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_scale, growth_tracker, found_inf, growth_factor, backoff_factor, growth_interval);
    Y3_CHECK_LAUNCH();
    return 0;
}
