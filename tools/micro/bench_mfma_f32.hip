// Issue rate of the exact-fp32 MFMA forms on gfx950, one wave per SIMD (the parity mode's sweeps) and several (its GEMMs):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bench_mfma_f32.hip -o tools/micro/bin/bench_mfma_f32
// Prints shader cycles per instruction and the FLOP / cycle / SIMD they amount to (64 = the 157 TF/s figure at 2.4 GHz, 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int blocks : {1, 256}) {
            long long c;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms16 = 0.f, ms32 = 0.f;
            hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 0.5f); hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 0.5f);
            hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms16, e0, e1);
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per16 = (double)c / (iters * 8 * 4);
            hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 0.5f); hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 0.5f);
            hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms32, e0, e1);
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per32 = (double)c / (iters * 8 * 2);
            const int wps = threads / 256;
            printf("%4d threads (%d wave/SIMD) x %3d blocks: 16x16x4 f32 %.1f cycles per wave-instruction (%.1f FLOP/clk/SIMD), 32x32x2 f32 %.1f (%.1f)\n",
                   threads, wps, blocks, per16, 2048.0 * wps / per16, per32, 4096.0 * wps / per32);
            printf("      by the clock: 16x16x4 %.3f ms = %.1f TF/s, 32x32x2 %.3f ms = %.1f TF/s\n", ms16,
                   (double)blocks * (threads / 64) * iters * 32.0 * 2048.0 / ms16 / 1e9, ms32, (double)blocks * (threads / 64) * iters * 16.0 * 4096.0 / ms32 / 1e9);
        }
    }
    return 0;
}
