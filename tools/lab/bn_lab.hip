// bn_lab: bandwidth sweep of the elementwise BatchNorm passes of the training step (standalone, no torch).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/bn_lab.hip -o tools/lab/bn_lab && tools/lab/bn_lab
// For each tensor shape of the yolov3 batch-64 step: a float4 copy and a float4 read as calibration, then the forward pass
// y = silu(scale*u + shift) and the backward apply (2 reads + 1 write) in several loop / grid / cache-policy forms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ h8 ld(const h8* p) { if (NT) return __builtin_nontemporal_load(p); return *p; }
template <bool NT> __device__ __forceinline__ void st(h8* p, h8 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ h8 bn_silu(h8 x, const f2* sc, const f2* sh) {
    h8 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f2 z = f2{(float)x[2 * q], (float)x[2 * q + 1]} * sc[q] + sh[q];
        f2 e = z * -1.44269504088896f;
        e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
        e = e + 1.0f;
        e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
        z = z * e;
        o[2 * q] = (_Float16)z[0]; o[2 * q + 1] = (_Float16)z[1];
    }
    return o;
}

// production mapping: thread (cg, pl) keeps its 8 channels' scale/shift and walks pixels pl + k*PL*grid; U pixels per trip, loads first
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fwd_loop(const h8* __restrict__ u, const float* __restrict__ scale, const float* __restrict__ shift, h8* __restrict__ y, long long M, int CG) {
    const int PL = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; }
    const long long stride = (long long)gridDim.x * PL;
    for (long long m = (long long)blockIdx.x * PL + pl; m < M; m += U * stride) {
        h8 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * stride < M) x[j] = ld<NTL>(u + (m + j * stride) * CG + cg);
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * stride < M) st<NTS>(y + (m + j * stride) * CG + cg, bn_silu(x[j], sc, sh));
    }
}
// contiguous chunks: block b owns pixels [b*chunk, (b+1)*chunk): a block streams one contiguous range (DRAM page locality)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fwd_chunk(const h8* __restrict__ u, const float* __restrict__ scale, const float* __restrict__ shift, h8* __restrict__ y, long long M, int CG, long long chunk) {
    const int PL = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; }
    const long long m0 = (long long)blockIdx.x * chunk;
    long long m1 = m0 + chunk; if (m1 > M) m1 = M;
    for (long long m = m0 + pl; m < m1; m += U * PL) {
        h8 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * PL < m1) x[j] = ld<NTL>(u + (m + j * PL) * CG + cg);
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * PL < m1) st<NTS>(y + (m + j * PL) * CG + cg, bn_silu(x[j], sc, sh));
    }
}
// one thread per 16-byte vector, no loop
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void fwd_flat(const h8* __restrict__ u, const float* __restrict__ scale, const float* __restrict__ shift, h8* __restrict__ y, long long NV, int CG) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NV) return;
    const int cg = (int)(i % CG);
    f2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; }
    st<NTS>(y + i, bn_silu(ld<NTL>(u + i), sc, sh));
}
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ a, f4* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ a, float* __restrict__ out, long long n) {
    f4 s = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.0f;
}
// backward apply: du = sc * (dy*silu'(z) - m0 - xhat*m1); two reads, one write
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void apply_loop(const h8* __restrict__ u, const h8* __restrict__ dy, const float* __restrict__ scale, const float* __restrict__ shift, h8* __restrict__ du, long long M, int CG) {
    const int PL = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; }
    const long long stride = (long long)gridDim.x * PL;
    for (long long m = (long long)blockIdx.x * PL + pl; m < M; m += U * stride) {
        h8 x[U], g[U];
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * stride < M) { x[j] = ld<NTL>(u + (m + j * stride) * CG + cg); g[j] = ld<NTL>(dy + (m + j * stride) * CG + cg); }
#pragma unroll
        for (int j = 0; j < U; ++j) if (m + j * stride < M) {
            h8 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 uf = f2{(float)x[j][2 * q], (float)x[j][2 * q + 1]};
                f2 dz = f2{(float)g[j][2 * q], (float)g[j][2 * q + 1]};
                const f2 z = uf * sc[q] + sh[q];
                f2 e = z * -1.44269504088896f;
                e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
                e = e + 1.0f;
                e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
                dz *= e + z * e * (1.0f - e);
                const f2 r = sc[q] * (dz - 0.01f - (uf - sh[q]) * sc[q] * 0.02f);
                o[2 * q] = (_Float16)r[0]; o[2 * q + 1] = (_Float16)r[1];
            }
            st<NTS>(du + (m + j * stride) * CG + cg, o);
        }
    }
}


// flat forms of the backward apply (one thread per 16-byte vector, no loop)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void apply_flat(const h8* __restrict__ u, const h8* __restrict__ dy, const float* __restrict__ scale, const float* __restrict__ shift, h8* __restrict__ du, long long NV, int CG) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NV) return;
    const int cg = (int)(i % CG);
    f2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; }
    const h8 x = ld<NTL>(u + i), g = ld<NTL>(dy + i);
    h8 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f2 uf = f2{(float)x[2 * q], (float)x[2 * q + 1]};
        f2 dz = f2{(float)g[2 * q], (float)g[2 * q + 1]};
        const f2 z = uf * sc[q] + sh[q];
        f2 e = z * -1.44269504088896f;
        e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
        e = e + 1.0f;
        e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
        dz *= e + z * e * (1.0f - e);
        const f2 r = sc[q] * (dz - 0.01f - (uf - sh[q]) * sc[q] * 0.02f);
        o[2 * q] = (_Float16)r[0]; o[2 * q + 1] = (_Float16)r[1];
    }
    st<NTS>(du + i, o);
}
// backward reduction (sum g, sum g*xhat per channel): thread (cg, pl) walks pixels, U pixels per trip, the NEXT trip's loads in flight under this
// trip's arithmetic (PF) as in train.hip; NT threads per block; per-thread fp32 partials written out (the block reduction is not what is measured)
template <int U, bool PF, int NT, bool NTL>
__global__ __launch_bounds__(NT) void reduce_loop(const h8* __restrict__ u, const h8* __restrict__ dy, const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out, long long M, int CG) {
    const int PL = NT / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    if (pl >= PL) return;
    f2 sc[4], sh[4], a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[q] = f2{scale[cg * 8 + 2 * q], scale[cg * 8 + 2 * q + 1]}; sh[q] = f2{shift[cg * 8 + 2 * q], shift[cg * 8 + 2 * q + 1]}; a0[q] = a1[q] = f2{0.0f, 0.0f}; }
    const long long stride = (long long)gridDim.x * PL;
    auto fetch = [&](long long m0, h8 (&xs)[U], h8 (&gs)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) if (m0 + j * stride < M) { xs[j] = ld<NTL>(u + (m0 + j * stride) * CG + cg); gs[j] = ld<NTL>(dy + (m0 + j * stride) * CG + cg); }
    };
    auto acc = [&](long long m0, const h8 (&xs)[U], const h8 (&gs)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (m0 + j * stride >= M) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2 uf = f2{(float)xs[j][2 * q], (float)xs[j][2 * q + 1]};
                f2 dz = f2{(float)gs[j][2 * q], (float)gs[j][2 * q + 1]};
                const f2 z = uf * sc[q] + sh[q];
                f2 e = z * -1.44269504088896f;
                e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
                e = e + 1.0f;
                e[0] = __builtin_amdgcn_rcpf(e[0]); e[1] = __builtin_amdgcn_rcpf(e[1]);
                dz *= e + z * e * (1.0f - e);
                a0[q] += dz;
                a1[q] += dz * ((uf - sh[q]) * sc[q]);
            }
        }
    };
    long long m0 = (long long)blockIdx.x * PL + pl;
    if (PF) {
        h8 xa[U], ga[U], xb[U], gb[U];
        fetch(m0, xa, ga);
        while (m0 < M) {
            fetch(m0 + U * stride, xb, gb);
            acc(m0, xa, ga);
            m0 += U * stride;
            if (m0 >= M) break;
            fetch(m0 + U * stride, xa, ga);
            acc(m0, xb, gb);
            m0 += U * stride;
        }
    } else {
        h8 xa[U], ga[U];
        for (; m0 < M; m0 += U * stride) { fetch(m0, xa, ga); acc(m0, xa, ga); }
    }
    float* o = out + ((long long)blockIdx.x * NT + threadIdx.x) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) { o[4 * q] = a0[q][0]; o[4 * q + 1] = a0[q][1]; o[4 * q + 2] = a1[q][0]; o[4 * q + 3] = a1[q][1]; }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <typename F> double run(F f, int reps = 8) {
        f(); f();
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        return ms / reps * 1e-3;
    }
};

int main() {
    struct Shape { long long M; int C; const char* name; };
    const Shape shapes[] = {{6553600, 64, "L1 320x320x64"}, {6553600, 32, "L2.cv1 320x320x32"}, {1638400, 128, "L3 160x160x128"}, {409600, 256, "L5 80x80x256"}, {102400, 512, "L7 40x40x512"}, {25600, 1024, "L9 20x20x1024"}};
    Timer T;
    float *scale, *shift, *out;
    CK(hipMalloc(&scale, 4096)); CK(hipMalloc(&shift, 4096)); CK(hipMalloc(&out, 64));
    std::vector<float> hs(1024, 1.01f);
    CK(hipMemcpy(scale, hs.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(shift, hs.data(), 4096, hipMemcpyHostToDevice));
    for (const Shape& s : shapes) {
        const size_t bytes = (size_t)s.M * s.C * 2;
        h8 *u, *dy, *y;
        CK(hipMalloc(&u, bytes)); CK(hipMalloc(&dy, bytes)); CK(hipMalloc(&y, bytes));
        CK(hipMemset(u, 0x11, bytes)); CK(hipMemset(dy, 0x12, bytes)); CK(hipMemset(y, 0, bytes));
        const int CG = s.C / 8, PL = 256 / CG;
        const long long NV = (long long)s.M * CG;
        printf("== %s  (%.1f MB per tensor)\n", s.name, bytes / 1e6);
        auto rep = [&](const char* name, double t, double nbytes) { printf("  %-44s %8.1f us  %6.2f TB/s\n", name, t * 1e6, nbytes / t / 1e12); };
        for (int g : {2048, 8192, 32768}) {
            char nm[96];
            snprintf(nm, sizeof nm, "copy float4 grid %d", g);
            rep(nm, T.run([&] { hipLaunchKernelGGL(copy_k, dim3(g), dim3(256), 0, 0, (const f4*)u, (f4*)y, (long long)(bytes / 16)); }), 2.0 * bytes);
            snprintf(nm, sizeof nm, "read float4 grid %d", g);
            rep(nm, T.run([&] { hipLaunchKernelGGL(read_k, dim3(g), dim3(256), 0, 0, (const f4*)u, out, (long long)(bytes / 16)); }), 1.0 * bytes);
        }
        auto grid_for = [&](long long cap, int per) { long long g = (s.M + (long long)PL * per - 1) / ((long long)PL * per); if (g > cap) g = cap; if (g < 1) g = 1; return (unsigned)g; };
#define FWD_LOOP(U, NTL, NTS, CAP) do { char nm[96]; snprintf(nm, sizeof nm, "fwd loop U%d ntl%d nts%d cap %d", U, NTL, NTS, CAP); const unsigned g = grid_for(CAP, 4); \
        rep(nm, T.run([&] { hipLaunchKernelGGL((fwd_loop<U, NTL, NTS>), dim3(g), dim3(256), 0, 0, u, scale, shift, y, s.M, CG); }), 2.0 * bytes); } while (0)
        FWD_LOOP(1, false, false, 8192);   // production
        FWD_LOOP(1, false, true, 8192);
        FWD_LOOP(1, false, true, 32768);
        FWD_LOOP(1, false, true, 1 << 20);
        {
            const unsigned g = (unsigned)((NV + 255) / 256);
            rep("fwd flat", T.run([&] { hipLaunchKernelGGL((fwd_flat<false, false>), dim3(g), dim3(256), 0, 0, u, scale, shift, y, NV, CG); }), 2.0 * bytes);
            rep("fwd flat nts", T.run([&] { hipLaunchKernelGGL((fwd_flat<false, true>), dim3(g), dim3(256), 0, 0, u, scale, shift, y, NV, CG); }), 2.0 * bytes);
            rep("fwd flat ntl", T.run([&] { hipLaunchKernelGGL((fwd_flat<true, false>), dim3(g), dim3(256), 0, 0, u, scale, shift, y, NV, CG); }), 2.0 * bytes);
            rep("fwd flat ntl nts", T.run([&] { hipLaunchKernelGGL((fwd_flat<true, true>), dim3(g), dim3(256), 0, 0, u, scale, shift, y, NV, CG); }), 2.0 * bytes);
            rep("apply flat", T.run([&] { hipLaunchKernelGGL((apply_flat<false, false>), dim3(g), dim3(256), 0, 0, u, dy, scale, shift, y, NV, CG); }), 3.0 * bytes);
            rep("apply flat nts", T.run([&] { hipLaunchKernelGGL((apply_flat<false, true>), dim3(g), dim3(256), 0, 0, u, dy, scale, shift, y, NV, CG); }), 3.0 * bytes);
            rep("apply flat ntl nts", T.run([&] { hipLaunchKernelGGL((apply_flat<true, true>), dim3(g), dim3(256), 0, 0, u, dy, scale, shift, y, NV, CG); }), 3.0 * bytes);
        }
#define APPLY(U, NTL, NTS, CAP) do { char nm[96]; snprintf(nm, sizeof nm, "apply loop U%d ntl%d nts%d cap %d", U, NTL, NTS, CAP); const unsigned g = grid_for(CAP, 4); \
        rep(nm, T.run([&] { hipLaunchKernelGGL((apply_loop<U, NTL, NTS>), dim3(g), dim3(256), 0, 0, u, dy, scale, shift, y, s.M, CG); }), 3.0 * bytes); } while (0)
        APPLY(1, false, false, 8192);   // production
        APPLY(1, false, true, 8192);
        APPLY(1, false, true, 32768);
        APPLY(1, false, true, 1 << 20);
        APPLY(1, false, false, 1 << 20);
        float* red;
        CK(hipMalloc(&red, (size_t)8192 * 1024 * 16 * 4));
#define REDUCE(U, PF, NT, NTL, GRID) do { char nm[96]; snprintf(nm, sizeof nm, "reduce U%d pf%d threads %d ntl%d grid %d", U, PF, NT, NTL, GRID); \
        long long g = (s.M + (long long)(NT / CG) * 16 - 1) / ((long long)(NT / CG) * 16); if (g > GRID) g = GRID; if (g < 1) g = 1; \
        rep(nm, T.run([&] { hipLaunchKernelGGL((reduce_loop<U, PF, NT, NTL>), dim3((unsigned)g), dim3(NT), 0, 0, u, dy, scale, shift, red, s.M, CG); }), 2.0 * bytes); } while (0)
        REDUCE(4, true, 256, false, 512);    // production
        REDUCE(4, true, 256, true, 512);
        REDUCE(4, true, 512, false, 512);
        REDUCE(4, true, 1024, false, 512);
        REDUCE(4, true, 256, false, 1024);
        REDUCE(4, true, 256, false, 2048);
        REDUCE(4, true, 256, false, 4096);
        REDUCE(2, true, 256, false, 2048);
        REDUCE(4, false, 256, false, 2048);
        REDUCE(4, false, 256, false, 8192);
        REDUCE(2, false, 256, false, 8192);
        REDUCE(1, false, 256, false, 8192);
        REDUCE(8, false, 256, false, 512);
        REDUCE(8, false, 512, false, 512);
        REDUCE(4, true, 512, false, 1024);
        CK(hipFree(red));
        CK(hipFree(u)); CK(hipFree(dy)); CK(hipFree(y));
    }
    return 0;
}
