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

// Are the workgroups of a 1-D grid dealt to the XCDs round-robin by their id (id % 8 = XCD: SPX mode on an eight-XCD part)?  Probed once per process:
// 64 workgroups report their XCC_ID.  Kernels that keep a group of workgroups on one XCD by their ids use this to choose XCD-scope hand-overs
// (served by that XCD's L2) over device-scope ones (write-through; correct anywhere).
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu);       // HW_REG_XCC_ID[3:0]
}
int las_xcd_round_robin() {
    static int ok = [] {
        int* d = nullptr;
        int h[64];
        if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 0;
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(64), dim3(64), 0, 0, d);
        const bool got = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!got) return 0;
        for (int j = 0; j < 64; ++j) if (h[j] != h[j & 7]) return 0;
        for (int a = 0; a < 8; ++a) for (int b = a + 1; b < 8; ++b) if (h[a] == h[b]) return 0;
        return 1;
    }();
    return ok;
}

extern "C" int las_dev_xcd_round_robin() { return las_xcd_round_robin(); }     // (diagnostics: tools/probe_wide_stamps.py prints it)

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

// ------------------------------------------------------------------------------------------------
// Input dropout of a bidirectional layer (reference las/layers.py:37-47: fw_cell and bw_cell each sit in their own
// DropoutWrapper(input_keep_prob = 1 - rate), i.e. the two directions see INDEPENDENT Bernoulli masks of the same input, scaled by
// 1 / keep).  Until round 5 this was two torch.nn.functional.dropout launches + a zero-pad + a bf16 cast per direction; here the two
// operand blocks of the directions' x-projections leave ONE launch, already in the product's layout (bf16, K padded with zero columns to a
// multiple of `kpad`), and the masks are never stored: element i of direction d keeps its value iff the 16-bit slice of
// splitmix64(seed, d, i / 4) that belongs to it is below keep x 65536 -- the backward launch regenerates them from the same counter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long drop_bits(unsigned long long seed, int dir, unsigned long long quad) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (2 * quad + (unsigned long long)dir + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dropout_pair_fwd_kernel(const TI* __restrict__ x, long long rows, int K, int ldx, TO* __restrict__ yf,
                                                               TO* __restrict__ yb, int ldy, unsigned long long seed, unsigned thr, float scale) {
    const long long quads_per_row = ldy / 4, total = rows * quads_per_row;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const long long r = q / quads_per_row;
        const int c0 = (int)(q - r * quads_per_row) * 4;
        const unsigned long long quad = (unsigned long long)r * ((K + 3) / 4) + (c0 >> 2);       // (counter over the UNPADDED block: the mask of an element does not depend on the padding)
        const unsigned long long bf = drop_bits(seed, 0, quad), bb = drop_bits(seed, 1, quad);
        float vf[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            float v = 0.f;
            if (c < K) {
                if (sizeof(TI) == 2) v = bf2f(((const unsigned short*)x)[r * ldx + c]);
                else v = ((const float*)x)[r * ldx + c];
            }
            vf[e] = ((unsigned)(bf >> (16 * e)) & 0xffffu) < thr ? v * scale : 0.f;
            vb[e] = ((unsigned)(bb >> (16 * e)) & 0xffffu) < thr ? v * scale : 0.f;
        }
        if (sizeof(TO) == 2) {
            reinterpret_cast<uint2*>((unsigned short*)yf + r * ldy + c0)[0] = make_uint2(f2bf2(vf[0], vf[1]), f2bf2(vf[2], vf[3]));
            reinterpret_cast<uint2*>((unsigned short*)yb + r * ldy + c0)[0] = make_uint2(f2bf2(vb[0], vb[1]), f2bf2(vb[2], vb[3]));
        } else {
            reinterpret_cast<float4*>((float*)yf + r * ldy + c0)[0] = make_float4(vf[0], vf[1], vf[2], vf[3]);
            reinterpret_cast<float4*>((float*)yb + r * ldy + c0)[0] = make_float4(vb[0], vb[1], vb[2], vb[3]);
        }
    }
}
// dx[r, c] = scale (m_fw[r, c] gf[r, c] + m_bw[r, c] gb[r, c])
template <typename TG, typename TO>
__global__ __launch_bounds__(256) void dropout_pair_bwd_kernel(const TG* __restrict__ gf, const TG* __restrict__ gb, int ldg, long long rows, int K,
                                                               TO* __restrict__ dx, int lddx, unsigned long long seed, unsigned thr, float scale) {
    const long long quads_per_row = (K + 3) / 4, total = rows * quads_per_row;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const long long r = q / quads_per_row;
        const int c0 = (int)(q - r * quads_per_row) * 4;
        const unsigned long long bf = drop_bits(seed, 0, (unsigned long long)q), bb = drop_bits(seed, 1, (unsigned long long)q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            if (c >= K) break;
            float a, b;
            if (sizeof(TG) == 2) { a = bf2f(((const unsigned short*)gf)[r * ldg + c]); b = bf2f(((const unsigned short*)gb)[r * ldg + c]); }
            else { a = ((const float*)gf)[r * ldg + c]; b = ((const float*)gb)[r * ldg + c]; }
            const float v = ((((unsigned)(bf >> (16 * e)) & 0xffffu) < thr ? a : 0.f) + (((unsigned)(bb >> (16 * e)) & 0xffffu) < thr ? b : 0.f)) * scale;
            if (sizeof(TO) == 2) ((unsigned short*)dx)[r * lddx + c] = f2bf(v);
            else ((float*)dx)[r * lddx + c] = v;
        }
    }
}
static unsigned drop_threshold(float keep) {
    const double t = (double)keep * 65536.0 + 0.5;
    return t >= 65536.0 ? 65536u : (t <= 0.0 ? 0u : (unsigned)t);
}
extern "C" int las_dropout_pair_fwd(const void* x, int x_dt, long long rows, int K, int ldx, void* y_fw, void* y_bw, int y_dt, int ldy,
                                    float keep, unsigned long long seed, void* stream) {
    LAS_ARG(x && y_fw && y_bw && rows >= 0 && K > 0 && ldx >= K && ldy >= K && (ldy % 4) == 0, "las_dropout_pair_fwd: bad arguments (ldy must be a multiple of 4 and >= K)");
    LAS_ARG(keep > 0.f && keep <= 1.f, "las_dropout_pair_fwd: keep probability %g outside (0, 1]", keep);
    if (rows == 0) return 0;
    int nb = cdiv(rows * (ldy / 4), 256);
    if (nb > 8192) nb = 8192;
    hipStream_t st = (hipStream_t)stream;
    const unsigned thr = drop_threshold(keep);
    const float scale = 1.0f / keep;
    typedef unsigned short u16;
#define DF(TI, TO) hipLaunchKernelGGL((dropout_pair_fwd_kernel<TI, TO>), dim3(nb), dim3(256), 0, st, (const TI*)x, rows, K, ldx, (TO*)y_fw, (TO*)y_bw, ldy, seed, thr, scale)
    if (x_dt == LAS_DT_BF16) { if (y_dt == LAS_DT_BF16) DF(u16, u16); else DF(u16, float); }
    else { if (y_dt == LAS_DT_BF16) DF(float, u16); else DF(float, float); }
#undef DF
    LAS_LAUNCHED();
    return 0;
}
extern "C" int las_dropout_pair_bwd(const void* g_fw, const void* g_bw, int g_dt, int ldg, long long rows, int K, void* dx, int dx_dt, int lddx,
                                    float keep, unsigned long long seed, void* stream) {
    LAS_ARG(g_fw && g_bw && dx && rows >= 0 && K > 0 && ldg >= K && lddx >= K, "las_dropout_pair_bwd: bad arguments");
    LAS_ARG(keep > 0.f && keep <= 1.f, "las_dropout_pair_bwd: keep probability %g outside (0, 1]", keep);
    if (rows == 0) return 0;
    int nb = cdiv(rows * ((K + 3) / 4), 256);
    if (nb > 8192) nb = 8192;
    hipStream_t st = (hipStream_t)stream;
    const unsigned thr = drop_threshold(keep);
    const float scale = 1.0f / keep;
    typedef unsigned short u16;
#define DB(TG, TO) hipLaunchKernelGGL((dropout_pair_bwd_kernel<TG, TO>), dim3(nb), dim3(256), 0, st, (const TG*)g_fw, (const TG*)g_bw, ldg, rows, K, (TO*)dx, lddx, seed, thr, scale)
    if (g_dt == LAS_DT_BF16) { if (dx_dt == LAS_DT_BF16) DB(u16, u16); else DB(u16, float); }
    else { if (dx_dt == LAS_DT_BF16) DB(float, u16); else DB(float, float); }
#undef DB
    LAS_LAUNCHED();
    return 0;
}
