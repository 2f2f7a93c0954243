// Per-CU ingest rate of ONE workgroup per CU pulling L2 / MALL-resident data with 16-byte loads (a wave-instruction = 1 KB contiguous), as the
// LSTM cell kernels of the beam search do: GB/s per CU against the number of waves (8 / 16), the loads a thread keeps in flight (D) and how
// (BURST: issue D, wait for all, repeat -- the cell kernel's chunk pipeline; ROLL: re-issue each load as soon as it is consumed), for data
// private to the workgroup (MALL-served after the first pass) or shared by the 4 workgroups of an XCD that own the same rows (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int D, bool ROLL>
__global__ __launch_bounds__(NT) void ingest(const uint4* __restrict__ src, float* sink, int n_loads, size_t wg_stride, int share) {
    // blockIdx -> XCD = blockIdx % 8; `share` workgroups of an XCD read the same region
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const size_t region = (size_t)xcd * (gridDim.x >> 3) + (share > 1 ? slot / share * share : slot);
    const uint4* p = src + region * wg_stride + threadIdx.x;
    float acc = 0.f;
    uint4 v[D];
    if (ROLL) {
#pragma unroll
        for (int u = 0; u < D; ++u) v[u] = p[(size_t)u * NT];
        for (int i = D; i < n_loads; i += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
                v[u] = p[(size_t)(i + u) * NT];
            }
        }
#pragma unroll
        for (int u = 0; u < D; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
    } else {
        for (int i = 0; i < n_loads; i += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) v[u] = p[(size_t)(i + u) * NT];
#pragma unroll
            for (int u = 0; u < D; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
            __syncthreads();
        }
    }
    if (acc == 1.2345f) sink[0] = acc;
}
template <int NT, int D, bool ROLL>
void run(const uint4* src, float* sink, int nwg, int share, size_t bytes_wg) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_loads = (int)(bytes_wg / (NT * 16)) / D * D;
    const size_t stride = bytes_wg / 16;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((ingest<NT, D, ROLL>), dim3(nwg), dim3(NT), 0, 0, src, sink, n_loads, stride, share);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3 / 20, kb = (double)n_loads * NT * 16 / 1024;
    printf("wgs=%3d threads=%4d D=%2d %s share=%d: %6.0f KB per workgroup in %6.2f us = %5.1f GB/s per CU (%.2f TB/s chip)\n", nwg, NT, D,
           ROLL ? "roll " : "burst", share, kb, us, kb * 1024 / us / 1e3, kb * 1024 * nwg / us / 1e6);
}
int main() {
    const size_t bytes_wg = 640 * 1024;
    uint4* src; float* sink;
    hipMalloc(&src, bytes_wg * 256); hipMemset(src, 1, bytes_wg * 256); hipMalloc(&sink, 64);
    for (int nwg : {256, 96}) for (int share : {1, 4}) {
        run<512, 8, false>(src, sink, nwg, share, bytes_wg);
        run<512, 24, false>(src, sink, nwg, share, bytes_wg);
        run<512, 8, true>(src, sink, nwg, share, bytes_wg);
        run<512, 24, true>(src, sink, nwg, share, bytes_wg);
        run<1024, 8, false>(src, sink, nwg, share, bytes_wg);
        run<1024, 8, true>(src, sink, nwg, share, bytes_wg);
        run<1024, 20, true>(src, sink, nwg, share, bytes_wg);
        run<256, 24, true>(src, sink, nwg, share, bytes_wg);
    }
    return 0;
}
