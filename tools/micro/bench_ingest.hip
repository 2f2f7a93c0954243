// How fast can ONE workgroup (1024 threads) pull its ~336 KB of per-step operands, (a) re-launched every step, (b) looping inside one launch?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void ingest(const uint4* __restrict__ src, float* sink, int iters, int per_thread) {
    const uint4* p = src + (size_t)blockIdx.x * 1024 * 32 + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint4 v[21];
#pragma unroll
        for (int u = 0; u < 21; ++u) v[u] = u < per_thread ? p[(size_t)u * 1024] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 21; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
        __syncthreads();
        asm volatile("" ::: "memory");
    }
    if (acc == 1.2345f) sink[0] = acc;
}
int main() {
    uint4* src; float* sink;
    const int nwg = 48 * 4;
    hipMalloc(&src, (size_t)nwg * 1024 * 32 * 16); hipMemset(src, 1, (size_t)nwg * 1024 * 32 * 16); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg : {48, 96, 192}) for (int per : {21, 10, 5}) {
        for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
            const int IT = 200;
            hipEventRecord(e0, 0);
            if (mode == 0) for (int i = 0; i < IT; ++i) hipLaunchKernelGGL(ingest, dim3(wg), dim3(1024), 0, 0, src, sink, 1, per);
            else hipLaunchKernelGGL(ingest, dim3(wg), dim3(1024), 0, 0, src, sink, IT, per);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("wgs=%3d bytes/wg=%3d KB %s : %.2f us/iter\n", wg, per * 16, mode ? "in-kernel loop" : "relaunch      ", ms * 1e3f / IT);
        }
    }
    return 0;
}
