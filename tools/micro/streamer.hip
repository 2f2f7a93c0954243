// A stand-in for "GEMMs next to a sweep": workgroups stream a large buffer through their XCD's L2, only on the XCDs of `mask`
// (workgroup id % 8 = XCD; the others exit at once).  tools/probe_xcd_interference.py launches it beside a BPTT sweep.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/streamer.hip -o tools/micro/bin/libstreamer.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ buf, size_t n16, unsigned mask, int iters, uint4* sink) {
    const int xcd = blockIdx.x & 7;
    if (!((mask >> xcd) & 1u)) return;
    const size_t per = n16 / gridDim.x, lo = per * blockIdx.x;
    uint4 acc = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it)
        for (size_t i = lo + threadIdx.x; i < lo + per; i += 256 * 4) {
            const uint4 a = buf[i], b = i + 256 < lo + per ? buf[i + 256] : a, c = i + 512 < lo + per ? buf[i + 512] : a, d = i + 768 < lo + per ? buf[i + 768] : a;
            acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y;
        }
    if (acc.x == 0x12345u && acc.y == 77u) sink[threadIdx.x] = acc;
}
extern "C" int streamer_launch(const void* buf, size_t bytes, unsigned mask, int iters, void* sink, void* stream) {
    hipLaunchKernelGGL(stream_kernel, dim3(8 * 64), dim3(256), 0, (hipStream_t)stream, (const uint4*)buf, bytes / 16, mask, iters, (uint4*)sink);
    return (int)hipGetLastError();
}
