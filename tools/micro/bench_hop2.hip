// Which publish instruction makes a granule visible to another CU of the same XCD fastest?  Same ping-pong as bench_hop.hip
// (consumer: sc1 load, s_sleep 1, reload), producer variants: store cache-policy bits, 64-bit atomic swap.
#include "las_common.h"
#include <cstdio>
__device__ int g_fail;
template <int LAUX>
__device__ __forceinline__ u32x2_t wait8(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag) {
    u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, LAUX);
    int budget = g_fail ? 1 : (1 << 16);
    while (v.x != tag) {
        if (--budget == 0) { g_fail = 1; return v; }
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, LAUX);
    }
    return v;
}
template <int KIND, int LAUX>
__device__ __forceinline__ void publish(__amdgpu_buffer_rsrc_t rs, unsigned long long* buf, unsigned off, unsigned tag, unsigned val) {
    const u32x2_t v = {tag, val};
    if (KIND < 32) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, KIND);
    else if (KIND == 32) __hip_atomic_exchange(buf + off / 8, ((unsigned long long)val << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (KIND == 33) { __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else if (KIND == 34) { __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 1); asm volatile("buffer_wbl2 sc0" ::: "memory"); }
}
template <int KIND, int LAUX>
__global__ __launch_bounds__(256) void hop_kernel(unsigned long long* buf, int N, long long* out, int partner) {
    const int who = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
    if (who < 0) return;
    const __amdgpu_buffer_rsrc_t rs = granule_rsrc(buf);
    const unsigned mine = (unsigned)(who * 256 + threadIdx.x) * 8u, theirs = (unsigned)((1 - who) * 256 + threadIdx.x) * 8u;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int i = 1; i <= N; ++i) {
        const unsigned slot = (unsigned)(i & 1) * 2 * 256 * 8u;
        if (who == 0) {
            publish<KIND, LAUX>(rs, buf, slot + mine, (unsigned)i, acc);
            acc += wait8<LAUX>(rs, slot + theirs, (unsigned)i).y;
        } else {
            acc += wait8<LAUX>(rs, slot + theirs, (unsigned)i).y;
            publish<KIND, LAUX>(rs, buf, slot + mine, (unsigned)i, acc);
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { out[who * 2] = t1 - t0; out[who * 2 + 1] = (long long)g_fail * 1000 + (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }
}
template <int KIND, int LAUX>
static void run(const char* name, int partner, int threads) {
    unsigned long long* buf; long long* out;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64);
    const int N = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(buf, 0, 1 << 20); { int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_fail), &z, 4); }
        hipLaunchKernelGGL((hop_kernel<KIND, LAUX>), dim3(partner + 1), dim3(threads), 0, 0, buf, N, out, partner);
        hipDeviceSynchronize();
    }
    long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("%-52s partner=blk%-3d thr=%d : %6.0f cycles per hop   (xcc/fail A %lld B %lld)\n", name, partner, threads, (double)h[0] / N / 2, h[1], h[3]);
    fflush(stdout);
    hipFree(buf); hipFree(out);
}
int main() {
    const int thr = 256;
    run<1, 16>("store sc0            | load sc1", 8, thr);
    run<0, 16>("store plain          | load sc1", 8, thr);
    run<2, 16>("store nt             | load sc1", 8, thr);
    run<3, 16>("store sc0 nt         | load sc1", 8, thr);
    run<16, 16>("store sc1            | load sc1", 8, thr);
    run<17, 16>("store sc0 sc1        | load sc1", 8, thr);
    run<19, 16>("store sc0 sc1 nt     | load sc1", 8, thr);
    run<32, 16>("atomic swap (agent)  | load sc1", 8, thr);
    run<33, 16>("store sc0 + vmcnt(0) | load sc1", 8, thr);
    run<34, 16>("store sc0 + wbl2 sc0 | load sc1", 8, thr);
    run<1, 17>("store sc0            | load sc0 sc1", 8, thr);
    run<1, 1>("store sc0            | load sc0", 8, thr);
    run<1, 18>("store sc0            | load sc1 nt", 8, thr);
    run<32, 16>("atomic swap (agent)  | load sc1, other XCD", 1, thr);
    run<17, 16>("store sc0 sc1        | load sc1, other XCD", 1, thr);
    return 0;
}
