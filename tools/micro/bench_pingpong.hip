// Two persistent kernels (48 "row" workgroups, 384 "product" workgroups) hand a decode step back and forth through
// agent-scope counters: what does one step cost compared with two kernel launches per step?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define RLX __ATOMIC_RELAXED
#define AGT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ bool wait_count(const unsigned* p, unsigned want, int* err) {
    int budget = 1 << 20;
    while (__hip_atomic_load(p, RLX, AGT) < want) {
        if (--budget == 0) { *err = 1; return false; }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}
__device__ __forceinline__ void busy_us(float us) {
    const u64 t0 = wall_clock64();
    while ((float)(wall_clock64() - t0) < us * 100.f) {}
}
// row kernel: 1024 threads per utterance
__global__ __launch_bounds__(1024) void row_kernel(unsigned* rflag, unsigned* gflag, const float* gates, uint2* xbf, int U, int B, float work_us, int* err) {
    const int b = blockIdx.x, mt = b >> 4, tid = threadIdx.x;
    __shared__ int ok;
    float acc = 0.f;
    for (int t = 0; t < U; ++t) {
        if (t > 0) {
            if (tid == 0) ok = wait_count(gflag + (t - 1) * 4 + mt, 128u, err);
            __syncthreads();
            const float* gp = gates + ((size_t)(t - 1) * B + b) * 2048;
            acc += __builtin_nontemporal_load(gp + tid) + __builtin_nontemporal_load(gp + 1024 + tid);   // 8 KB row
        }
        if (tid == 0) busy_us(work_us);
        __syncthreads();
        // publish the 2304-byte input row (288 x 8 bytes) write-through, drain, count
        if (tid < 288) __hip_atomic_store((u64*)(xbf + ((size_t)(t & 1) * B + b) * 288 + tid), ((u64)t << 32) | (unsigned)acc, RLX, AGT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(rflag + t * 4 + mt, 1u, RLX, AGT);
    }
    if (acc == 1.2345f) err[1] = 1;
}
// product kernel: 512 threads per (column tile, row tile)
__global__ __launch_bounds__(512) void prod_kernel(unsigned* rflag, unsigned* gflag, float* gates, const uint2* xbf, const uint4* W, int U, int B, float work_us, int* err) {
    const int ct = blockIdx.x, mt = blockIdx.y, tid = threadIdx.x;
    __shared__ int ok;
    uint4 w[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = W[((size_t)ct * 5 + i) * 512 + tid];      // resident weight fragments
    unsigned acc = w[0].x ^ w[1].y ^ w[2].z ^ w[3].w ^ w[4].x;
    for (int t = 0; t < U; ++t) {
        if (tid == 0) ok = wait_count(rflag + t * 4 + mt, 16u, err);
        __syncthreads();
        // 16 rows x 288 granules = 4608 x 8 B: 9 per thread
        const u64* xp = (const u64*)(xbf + ((size_t)(t & 1) * B + mt * 16) * 288);
        u64 v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = __hip_atomic_load(xp + i * 512 + tid, RLX, AGT);
#pragma unroll
        for (int i = 0; i < 9; ++i) acc ^= (unsigned)v[i];
        if (tid == 0) busy_us(work_us);
        __syncthreads();
        if (tid < 256) {
            float* gp = gates + ((size_t)t * B + mt * 16 + (tid >> 4)) * 2048 + ct * 16 + (tid & 15);
            __hip_atomic_store((unsigned*)gp, acc, RLX, AGT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(gflag + t * 4 + mt, 1u, RLX, AGT);
    }
}
int main() {
    const int U = 191, B = 48;
    unsigned *rflag, *gflag; float* gates; uint2* xbf; uint4* W; int* err;
    hipMalloc(&rflag, U * 4 * 4); hipMalloc(&gflag, U * 4 * 4);
    hipMalloc(&gates, (size_t)U * B * 2048 * 4); hipMemset(gates, 0, (size_t)U * B * 2048 * 4);
    hipMalloc(&xbf, (size_t)2 * B * 288 * 8); hipMalloc(&W, (size_t)128 * 5 * 512 * 16); hipMemset(W, 1, (size_t)128 * 5 * 512 * 16);
    hipMalloc(&err, 64); hipMemset(err, 0, 64);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, eb; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb);
    for (float rw : {0.f, 5.f}) for (float pw : {0.f, 1.f}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(rflag, 0, U * 16, sa); hipMemsetAsync(gflag, 0, U * 16, sa);
            hipEventRecord(e0, sa);
            hipStreamWaitEvent(sb, e0, 0);
            hipLaunchKernelGGL(row_kernel, dim3(B), dim3(1024), 0, sa, rflag, gflag, gates, xbf, U, B, rw, err);
            hipLaunchKernelGGL(prod_kernel, dim3(128, 3), dim3(512), 0, sb, rflag, gflag, gates, xbf, W, U, B, pw, err);
            hipEventRecord(eb, sb);
            hipStreamWaitEvent(sa, eb, 0);
            hipEventRecord(e1, sa);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int herr[2]; hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost);
            if (rep == 2) printf("row work %.0f us, product work %.0f us: %.2f us/step  (err %d)\n", rw, pw, ms * 1e3f / U, herr[0]);
        }
    }
    return 0;
}
