// One-hop latency of the cluster exchange: workgroups A (block 0) and B (block 8, same XCD) bounce a tagged 16-byte granule through
// the XCD's L2 (sc0 store, sc1 load) N times; cycles per hop = round trip / 2.  Variants of the consumer's polling loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I automatic-speech-recognition_amd/csrc tools/micro/bench_hop.hip -o /tmp/bench_hop
#include "las_common.h"
#include <cstdio>
__device__ int g_fail;
template <int MODE, int SLEEP>
__device__ __forceinline__ u32x4_t wait_granule(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag) {
    if (MODE == 1) __builtin_amdgcn_s_sleep(SLEEP);                  // delayed first poll
    u32x4_t v = granule16_load(rs, off);
    int budget = g_fail ? 1 : (1 << 16);
    if (MODE == 2) {                                                 // two loads in flight, half a poll interval apart
        u32x4_t v2 = granule16_load(rs, off);
        for (;;) {
            if (v.x == tag && v.w == tag) return v;
            if (--budget == 0) { g_fail = 1; return v; }
            v = v2; v2 = granule16_load(rs, off);
        }
    }
    while (!(v.x == tag && v.w == tag)) {
        if (--budget == 0) { g_fail = 1; return v; }
        if (MODE == 4) { asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15"); } else if (MODE != 3) __builtin_amdgcn_s_sleep(MODE == 1 ? 1 : SLEEP);
        v = granule16_load(rs, off);
    }
    return v;
}
template <int MODE, int SLEEP>
__global__ __launch_bounds__(256) void hop_kernel(unsigned long long* buf, int N, int local, long long* out, int partner) {
    const int who = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
    if (who < 0) return;
    const __amdgpu_buffer_rsrc_t rs = granule_rsrc(buf);
    const unsigned mine = (unsigned)(who * 256 + threadIdx.x) * 16u, theirs = (unsigned)((1 - who) * 256 + threadIdx.x) * 16u;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int i = 1; i <= N; ++i) {
        const unsigned slot = (unsigned)(i & 1) * 2 * 256 * 16u;
        if (who == 0) {
            granule16_store(rs, slot + mine, (unsigned)i, acc, acc, local != 0);
            acc += wait_granule<MODE, SLEEP>(rs, slot + theirs, (unsigned)i).y;
        } else {
            acc += wait_granule<MODE, SLEEP>(rs, slot + theirs, (unsigned)i).y;
            granule16_store(rs, slot + mine, (unsigned)i, acc, acc, local != 0);
        }
        __syncthreads();      // like the sweep: the whole workgroup proceeds together
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { out[who * 2] = t1 - t0; out[who * 2 + 1] = (long long)g_fail * 1000 + (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }
}
template <int MODE, int SLEEP>
static void run(const char* name, int local, int partner, int threads) {
    unsigned long long* buf; long long* out;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64);
    const int N = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(buf, 0, 1 << 20); { int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_fail), &z, 4); }
        hipLaunchKernelGGL((hop_kernel<MODE, SLEEP>), dim3(partner + 1), dim3(threads), 0, 0, buf, N, local, out, partner);
        hipDeviceSynchronize();
    }
    long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("%-44s local=%d partner=blk%-3d thr=%d : %6.0f cycles per hop\n", name, local, partner, threads, (double)h[0] / N / 2); printf("      xcc/fail: A %lld B %lld\n", h[1], h[3]); fflush(stdout);
    hipFree(buf); hipFree(out);
}
int main() {
    for (int thr : {256}) {
        run<0, 1>("poll: load, s_sleep 1, reload", 1, 8, thr);
        run<0, 1>("poll: load, s_sleep 1, reload", 1, 8, thr);
        run<1, 1>("first poll after s_sleep 1", 1, 8, thr);
        run<1, 2>("first poll after s_sleep 2", 1, 8, thr);
        run<1, 3>("first poll after s_sleep 3", 1, 8, thr);
        run<1, 4>("first poll after s_sleep 4", 1, 8, thr);
        run<1, 5>("first poll after s_sleep 5", 1, 8, thr);
        run<1, 6>("first poll after s_sleep 6", 1, 8, thr);
        run<1, 8>("first poll after s_sleep 8", 1, 8, thr);
        run<1, 10>("first poll after s_sleep 10", 1, 8, thr);
        run<1, 12>("first poll after s_sleep 12", 1, 8, thr);
        run<4, 0>("tight reload, s_nop spacing", 1, 8, thr);
    }
    return 0;
}
