// Standalone micro-benchmark for the Speller's per-step skinny-M contraction (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I automatic-speech-recognition_amd/csrc tools/micro/bench_skinny.hip -o /tmp/bench_skinny
#include "gemm.hip"
#include "common.hip"
#include <vector>
#include <cstdio>

// ---- candidate: grid (ct, mt), one 16-row tile per workgroup, A fp32 or bf16 ------------------
template <bool ABF>
__global__ __launch_bounds__(512, 1) void skinny2_kernel(const void* __restrict__ Av, int lda, int M, int K,
                                                         const u16x8_t* __restrict__ Bp, int KS, int N,
                                                         float* __restrict__ C, int ldc, const float* __restrict__ bias) {
    constexpr int NW = 8;
    __shared__ float red[NW][64][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ct = blockIdx.x, mt = blockIdx.y;
    const int KSW = (KS + NW - 1) / NW;
    const int ks0 = w * KSW, ks1 = min(KS, ks0 + KSW);
    const u16x8_t* bp = Bp + (size_t)ct * KS * 64 + lane;
    int row = mt * 16 + c;
    if (row >= M) row = M - 1;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 8;
    for (int ks = ks0; ks < ks1; ks += UN) {
        u16x8_t bv[UN], av[UN];
        float4 a0[UN], a1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = ks + u;
            const bool on = kk < ks1;
            bv[u] = on ? bp[(size_t)kk * 64] : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            const bool ka = on && (kk * 32 + g * 8 + 8 <= K);
            if (ABF) {
                const unsigned short* ap = (const unsigned short*)Av + (long long)row * lda + g * 8;
                av[u] = ka ? *reinterpret_cast<const u16x8_t*>(ap + kk * 32) : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            } else {
                const float* ap = (const float*)Av + (long long)row * lda + g * 8;
                a0[u] = ka ? *reinterpret_cast<const float4*>(ap + kk * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                a1[u] = ka ? *reinterpret_cast<const float4*>(ap + kk * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ABF) {
                uint4 pk;
                pk.x = f2bf2(a0[u].x, a0[u].y); pk.y = f2bf2(a0[u].z, a0[u].w);
                pk.z = f2bf2(a1[u].x, a1[u].y); pk.w = f2bf2(a1[u].z, a1[u].w);
                av[u] = __builtin_bit_cast(u16x8_t, pk);
            }
            acc = mfma_bf16_16x16x32(av[u], bv[u], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][lane][r] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int r16 = tid >> 4, c16 = tid & 15;
        const int l2 = (r16 >> 2) * 16 + c16, reg = r16 & 3;
        const int orow = mt * 16 + r16, col = ct * 16 + c16;
        if (orow < M && col < N) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += red[ww][l2][reg];
            if (bias) v += bias[col];
            C[(long long)orow * ldc + col] = v;
        }
    }
}

__global__ void writer_kernel(float* A, unsigned short* Ab, int n, float v) {   // 48 WGs rewrite the operand
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        A[i] = v + (float)(i & 63) * 1e-3f;
        Ab[i] = f2bf(v + (float)(i & 63) * 1e-3f);
    }
}
__global__ void polluter_kernel(const float4* src, float* sink, int n4) {      // 48 WGs stream 16 MB
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int M = 48;
    struct Shape { int K, N; } shapes[] = {{1152, 2048}, {2048, 1152}, {512, 128}};
    float* pol; hipMalloc(&pol, 16 << 20); hipMemset(pol, 0, 16 << 20);
    float* sink; hipMalloc(&sink, 64);
    for (auto s : shapes) {
        const int K = s.K, N = s.N, KS = cdiv(K, 32), nct = cdiv(N, 16);
        float *A, *W, *C, *C2; unsigned short* Ab; void* packed;
        hipMalloc(&A, M * K * 4); hipMalloc(&Ab, M * K * 2); hipMalloc(&W, (size_t)K * N * 4);
        hipMalloc(&C, M * N * 4); hipMalloc(&C2, M * N * 4);
        std::vector<float> hw((size_t)K * N);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&packed, las_skinny_pack_bytes(K, N));
        las_skinny_pack(W, N, K, N, 0, packed, 0);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int IT = 300;
        for (int mode = 0; mode < 2; ++mode) {          // 0: back-to-back, 1: writer + polluter between
            float base = 0.f;
            for (int v = -1; v < 4; ++v) {
                if (mode == 0 && v == -1) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0, 0);
                    for (int it = 0; it < IT; ++it) {
                        if (mode == 1) {
                            hipLaunchKernelGGL(writer_kernel, dim3(48), dim3(1024), 0, 0, A, Ab, M * K, (float)it * 1e-3f);
                            hipLaunchKernelGGL(polluter_kernel, dim3(48), dim3(1024), 0, 0, (const float4*)pol, sink, (16 << 20) / 16);
                        }
                        const u16x8_t* Bp = (const u16x8_t*)packed;
                        if (v == 0) las_skinny_gemm(A, K, M, K, packed, N, C, N, nullptr, 0);
                        if (v == 1) hipLaunchKernelGGL(skinny2_kernel<false>, dim3(nct, 3), dim3(512), 0, 0, A, K, M, K, Bp, KS, N, C2, N, nullptr);
                        if (v == 2) hipLaunchKernelGGL(skinny2_kernel<true>, dim3(nct, 3), dim3(512), 0, 0, Ab, K, M, K, Bp, KS, N, C2, N, nullptr);
                        if (v == 3) hipLaunchKernelGGL(skinny2_kernel<true>, dim3(nct, 1), dim3(512), 0, 0, Ab, K, 16, K, Bp, KS, N, C2, N, nullptr);
                    }
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep == 1) {
                        if (v == -1) base = ms;
                        printf("K=%d N=%d mode=%d v=%d : %.2f us/iter (net %.2f)\n", K, N, mode, v, ms * 1e3f / IT, (ms - base) * 1e3f / IT);
                    }
                }
            }
        }
        // correctness of the candidates vs the current kernel
        hipLaunchKernelGGL(writer_kernel, dim3(48), dim3(1024), 0, 0, A, Ab, M * K, 0.25f);
        las_skinny_gemm(A, K, M, K, packed, N, C, N, nullptr, 0);
        hipLaunchKernelGGL(skinny2_kernel<false>, dim3(nct, 3), dim3(512), 0, 0, A, K, M, K, (const u16x8_t*)packed, KS, N, C2, N, nullptr);
        std::vector<float> h1(M * N), h2(M * N);
        hipMemcpy(h1.data(), C, M * N * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), C2, M * N * 4, hipMemcpyDeviceToHost);
        double md = 0; for (int i = 0; i < M * N; ++i) md = fmax(md, fabs(h1[i] - h2[i]));
        printf("K=%d N=%d max|v0-v1| = %g (C[5]=%g)\n", K, N, md, h1[5]);
        hipFree(A); hipFree(Ab); hipFree(W); hipFree(C); hipFree(C2); hipFree(packed);
    }
    return 0;
}
