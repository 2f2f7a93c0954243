// common.hip -- version / error plumbing of the C ABI.
#include "las_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void las_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int las_version(void) { return LAS_HIP_ABI_VERSION; }
extern "C" const char* las_last_error(void) { return g_err; }

// compute units of the current device (init-once attribute cache; partitioned / CU-masked devices report fewer than 256)
int las_device_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    return cus;
}

// Stream-ordered wait on a device word (bounded): everything enqueued behind it on `stream` starts only once *word == value
// or max_us microseconds have passed.  A scheduling aid, never a correctness dependency: the host uses it to keep the
// weight-gradient GEMMs of the side stream off the machine until the next recurrent sweep (LAS_SEQ_ANNOUNCE) is resident.
__global__ __launch_bounds__(64) void wait_word_kernel(const int* word, int value, long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();                      // 100 MHz
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        if (wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" int las_wait_word(const int* word, int value, int max_us, void* stream) {
    LAS_ARG(word && max_us >= 0, "las_wait_word: bad arguments");
    hipLaunchKernelGGL(wait_word_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, value, (long long)max_us * 100);
    LAS_LAUNCHED();
    return 0;
}

// The same for the sweeps' announcement word (LAS_SEQ_ANNOUNCE numbers run cyclically through 1..1023): passes as soon as the word HAS
// REACHED n, i.e. also when a later sweep has announced itself meanwhile -- a hold that is enqueued late (its stream was busy) must
// not sit out its whole bound because the number it waits for has come and gone.
__global__ __launch_bounds__(64) void wait_announce_kernel(const int* word, int n, long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();                      // 100 MHz
    for (;;) {
        const int w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w >= 1 && w <= 1023 && ((w - n + 1023) % 1023) < 512) break;
        if (wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" int las_wait_announce(const int* word, int n, int max_us, void* stream) {
    LAS_ARG(word && max_us >= 0 && n >= 1 && n <= 1023, "las_wait_announce: bad arguments");
    hipLaunchKernelGGL(wait_announce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, n, (long long)max_us * 100);
    LAS_LAUNCHED();
    return 0;
}

// stream-ordered store of a device word (the completion flag behind a chunk of work that another, already running, kernel waits for)
__global__ __launch_bounds__(64) void set_word_kernel(int* word, int value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// A resident "foreign" kernel (diagnostics, round 5): `n` workgroups of 256 threads that hold their compute-unit slots -- `lds` bytes of LDS, 32 or
// 64 VGPRs per lane: the footprint of a collective's channel -- until *stop != 0 (bounded by max_ms on the constant 100 MHz clock).
// Workgroup L lands on XCD L % 8, so n = 8 is one per XCD.  resident[0] counts the workgroups that have started.
template <int VGPRS>
__global__ __launch_bounds__(256) void occupy_kernel(const int* stop, int* resident, long long max_ticks, int lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char occ_lds[];
    if (VGPRS > 32) asm volatile("; the wave holds 64 VGPRs" ::: "v63");
    else            asm volatile("; the wave holds 32 VGPRs" ::: "v31");
    if (threadIdx.x == 0) { if (lds > 0) occ_lds[0] = 1; atomicAdd(resident, 1); }
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(64);
}
extern "C" int las_occupy(const int* stop, int* resident, int n, int lds, int vgprs, int max_ms, void* stream) {
    LAS_ARG(stop && resident && n > 0 && n <= 1024 && lds >= 0 && lds <= 64 * 1024 && max_ms > 0 && max_ms <= 60000, "las_occupy: bad arguments");
    const long long ticks = (long long)max_ms * 100000;
    if (vgprs > 32) hipLaunchKernelGGL(occupy_kernel<64>, dim3(n), dim3(256), lds, (hipStream_t)stream, stop, resident, ticks, lds);
    else            hipLaunchKernelGGL(occupy_kernel<32>, dim3(n), dim3(256), lds, (hipStream_t)stream, stop, resident, ticks, lds);
    LAS_LAUNCHED();
    return 0;
}

// Stream-ordered wait until EVERY one of n device words has reached `need` (the progress words of a BPTT sweep that publishes how far its
// d(pre-activation) has reached memory: las_rnn_seq_bwd_db_progress).  A CORRECTNESS dependency, unlike las_wait_announce: work enqueued
// behind it reads what the words vouch for, so a time-out (max_us on the 100 MHz clock) stores `code` into status[0] -- the step is then
// invalid and las_clip_adam skips it.
__global__ __launch_bounds__(256) void wait_words_min_kernel(const int* words, int n, int need, long long max_ticks, int* status, int code) {
    const long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < n; i += 256) {
        while (__hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if (wall_clock64() - t0 > max_ticks) { if (status) status[0] = code; return; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
extern "C" int las_wait_words_min(const int* words, int n, int need, int max_us, int* status, int code, void* stream) {
    LAS_ARG(words && n > 0 && max_us > 0, "las_wait_words_min: bad arguments");
    hipLaunchKernelGGL(wait_words_min_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, words, n, need, (long long)max_us * 100, status, code);
    LAS_LAUNCHED();
    return 0;
}

extern "C" int las_set_word(int* word, int value, void* stream) {
    LAS_ARG(word, "las_set_word: null pointer");
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, value);
    LAS_LAUNCHED();
    return 0;
}
