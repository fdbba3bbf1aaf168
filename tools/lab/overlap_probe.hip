// Does VALU work of one wave hide under the MFMAs of another wave of the same SIMD on this chip?  (round 6 probe; hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe)
// Blocks of 512 threads = 8 waves = 2 per SIMD, one block per CU.  Mode bits per wave pair: waves 0-3 run role A, waves 4-7 role B.
//   role 1: a chain-free stream of v_mfma_f32_32x32x16_f16 (4 independent accumulators)      role 2: v_exp_f32 + v_rcp_f32 + 3 packed ops per value (the SiLU epilogue mix)
//   role 0: idle
// Reports the time of (A, B) = (1,0), (0,2), (1,2), (1,1), (2,2) for equal per-wave iteration counts: if (1,2) ~ max((1,0),(0,2)) the pipes overlap; if ~ sum they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void role_mfma(int iters, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
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
__global__ __launch_bounds__(512, 1) void probe(int roleA, int roleB, int itA, int itB, float* out) {
    const int wv = threadIdx.x >> 6;
    const int role = wv < 4 ? roleA : roleB;
    const int it = wv < 4 ? itA : itB;
    if (role == 1) role_mfma(it, out);
    else if (role == 2) role_valu(it, out);
}
int main(int argc, char** argv) {
    const int itM = argc > 1 ? atoi(argv[1]) : 20000;   // x 4 MFMAs of 32 cycles = 128 cycles per iteration
    const int itV = argc > 2 ? atoi(argv[2]) : 1400;    // x 16 values x ~44 cycles = ~700 cycles per iteration
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int modes[6][2] = {{1, 0}, {0, 2}, {1, 2}, {1, 1}, {2, 2}, {1, 2}};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 6; ++m) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, modes[m][0], modes[m][1], itM, itV, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, modes[m][0], modes[m][1], itM, itV, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves = (modes[m][0] == 1 ? 4.0 : 0) + (modes[m][1] == 1 ? 4.0 : 0);
            const double tf = waves * 256.0 * itM * 4.0 * (2.0 * 32 * 32 * 16) / (ms / 10 * 1e-3) / 1e12;
            printf("A=%d B=%d : %.3f ms per launch   MFMA %.0f TF\n", modes[m][0], modes[m][1], ms / 10, tf);
        }
    return 0;
}
