// Does VALU work of one wave hide under the MFMAs of another wave of the same SIMD on this chip?  (round 6 probe; hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe)
// Blocks of 512 threads = 8 waves = 2 per SIMD, one block per CU.  Mode bits per wave pair: waves 0-3 run role A, waves 4-7 role B.
//   role 1: a chain-free stream of v_mfma_f32_32x32x16_f16 (4 independent accumulators)      role 2: v_exp_f32 + v_rcp_f32 + 3 packed ops per value (the SiLU epilogue mix)
//   role 0: idle
// Reports the time of (A, B) = (1,0), (0,2), (1,2), (1,1), (2,2) for equal per-wave iteration counts: if (1,2) ~ max((1,0),(0,2)) the pipes overlap; if ~ sum they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void role_mfma(int iters, float* out) {
    h8 a[4], b[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) { a[k][i] = (_Float16)(threadIdx.x * 0.001f + i + k); b[k][i] = (_Float16)(i * 0.5f - k); }   // (distinct operands: no CSE of the four streams)
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], c3, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__device__ __forceinline__ void role_valu(int iters, float* out) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float e = __builtin_amdgcn_exp2f(v[i] * -1.44269504f);
            e = __builtin_amdgcn_rcpf(e + 1.0f);
            v[i] = v[i] * e + 0.25f;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// role 3: plain full-rate VALU (dependent v_fma_f32 chains, 16 values x 10 per iteration: about the issue time of role 2's 16 values) -- is it the transcendental unit or any VALU?
__device__ __forceinline__ void role_fma(int iters, float* out) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int k = 0; k < 10; ++k) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f * k);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// role 4: LDS reads (ds_read_b128, conflict-free), 16 per iteration
__device__ __forceinline__ void role_lds(int iters, float* out) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 buf[1024];
    buf[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f};
    buf[threadIdx.x + 512] = f4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f4 t;
            const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&buf[(threadIdx.x + 64 * i + it) & 1023];
            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(8)");
            acc += 0.0f;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
// roles 5 / 6: the SAME wave interleaves: per iteration four MFMAs and, between them, 16 independent v_fma_f32 (5) or 4 v_exp_f32 (6) -- 16 issue cycles of other work per
// 32-cycle MFMA.  If the time stays at role 1's, a wave's own independent instructions do run in the shadow of its MFMAs.
template <int KIND> __device__ __forceinline__ void role_mix(int iters, float* out) {
    h8 a[4], b[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) { a[k][i] = (_Float16)(threadIdx.x * 0.001f + i + k); b[k][i] = (_Float16)(i * 0.5f - k); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    auto side = [&](int q) {
        if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * q + i] = __builtin_fmaf(v[4 * q + i], 0.999f, 0.001f);
        } else {
            v[4 * q] = __builtin_amdgcn_exp2f(v[4 * q] * 0.5f);
        }
    };
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0); side(0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c1, 0, 0, 0); side(1);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], c2, 0, 0, 0); side(2);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], c3, 0, 0, 0); side(3);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// role 7: role 1 behind s_setprio 3 (the other wave of the SIMD stays at priority 0): does the issue arbiter let the MFMA wave in when it is due?
__device__ __forceinline__ void role_mfma_prio(int iters, float* out) {
    __builtin_amdgcn_s_setprio(3);
    role_mfma(iters, out);
    __builtin_amdgcn_s_setprio(0);
}
__global__ __launch_bounds__(512, 1) void probe(int roleA, int roleB, int itA, int itB, float* out) {
    const int wv = threadIdx.x >> 6;
    const int role = wv < 4 ? roleA : roleB;
    const int it = role == 1 ? itA : itB;   // (itA: MFMA iterations, itB: VALU iterations, whichever slot runs the role)
    if (role == 1) role_mfma(it, out);
    else if (role == 2) role_valu(it, out);
    else if (role == 3) role_fma(it, out);
    else if (role == 4) role_lds(it, out);
    else if (role == 5) role_mix<5>(itA, out);
    else if (role == 7) role_mfma_prio(itA, out);
    else if (role == 6) role_mix<6>(itA, out);
}
int main(int argc, char** argv) {
    const int itM = argc > 1 ? atoi(argv[1]) : 20000;   // x 4 MFMAs of 32 cycles = 128 cycles per iteration
    const int itV = argc > 2 ? atoi(argv[2]) : 1400;    // x 16 values x ~44 cycles = ~700 cycles per iteration
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int modes[8][2] = {{1, 0}, {7, 0}, {0, 3}, {1, 3}, {7, 3}, {0, 2}, {7, 2}, {7, 4}};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 8; ++m) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, modes[m][0], modes[m][1], itM, itV, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, modes[m][0], modes[m][1], itM, itV, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves = ((modes[m][0] == 1 || modes[m][0] >= 5) ? 4.0 : 0) + ((modes[m][1] == 1 || modes[m][1] >= 5) ? 4.0 : 0);
            const double tf = waves * 256.0 * itM * 4.0 * (2.0 * 32 * 32 * 16) / (ms / 10 * 1e-3) / 1e12;
            printf("A=%d B=%d : %.3f ms per launch   MFMA %.0f TF\n", modes[m][0], modes[m][1], ms / 10, tf);
        }
    return 0;
}
