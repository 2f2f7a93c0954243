// las_common.h -- shared device/host helpers for liblas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/las_hip.h"

// gfx950 only: v_permlane32_swap / v_permlane16_swap, v_cvt_pk_bf16_f32, v_mfma_f32_16x16x32_bf16 and ds_read_b64_tr_b16 are
// used unconditionally -- there is no other target and no fallback path to keep in step with
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liblas_hip.so is written for gfx950 (MI355X / CDNA4) only"
#endif

// ---- error reporting (thread-local message, negative return codes) ----------------------------
void las_set_error(const char* fmt, ...);
int las_device_cus();   // compute units of the current device (cached)
int las_xcd_round_robin();   // 1: workgroup id % 8 = XCD for 1-D grids on this device (probed once)

#define LAS_ARG(cond, ...)                                   \
    do { if (!(cond)) { las_set_error(__VA_ARGS__); return -1; } } while (0)
#define LAS_HIP(expr)                                        \
    do { hipError_t e__ = (expr);                            \
         if (e__ != hipSuccess) { las_set_error("%s -> %s", #expr, hipGetErrorString(e__)); \
                                  return (int)e__; } } while (0)
#ifdef LAS_DEBUG_SYNC   // `make dbg`: every launch is waited for, a faulting kernel is reported with the file and line of its launch
#define LAS_LAUNCHED()                                       \
    do { hipError_t e__ = hipGetLastError();                 \
         if (e__ == hipSuccess) e__ = hipDeviceSynchronize(); \
         if (e__ != hipSuccess) { las_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
                                  return (int)e__; } } while (0)
#else
#define LAS_LAUNCHED() LAS_HIP(hipGetLastError())
#endif

// ---- vector types for MFMA --------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// fp32 -> bf16 bits, round to nearest even (finite inputs; NaN stays NaN-ish).
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// two fp32 -> packed bf16x2 (lo in bits 0-15), RNE, one v_cvt_pk_bf16_f32 (gfx950)
__device__ __forceinline__ unsigned int f2bf2(float lo, float hi) {
    unsigned int r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// Touch every 64-byte line of the kernel's argument segment at entry (one scalar load per line, results unused).  The compiler loads
// arguments lazily, in dependent batches (a batch, a branch, the next batch ...), and the first touch of a line is a miss all the way to
// the segment's memory (~0.5-0.7 us each, r5 stamps: 2.1 us from a workgroup's entry to its first vector load with a 600-byte argument
// struct); with the lines requested together up front the later batches hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned d;
    // (one block, ending in its own wait: the compiler does not know that an asm's scalar load is still in flight and would hand the
    //  destination register to somebody else; the wait is the round trip the kernel's first argument batch waits for anyway)
    if (BYTES <= 256)
        asm volatile("s_load_dword %0, %1, 0x0\n s_load_dword %0, %1, 0x40\n s_load_dword %0, %1, 0x80\n s_load_dword %0, %1, 0xc0\n s_waitcnt lgkmcnt(0)"
                     : "=&s"(d) : "s"(kp) : "memory");
    else if (BYTES <= 512)
        asm volatile("s_load_dword %0, %1, 0x0\n s_load_dword %0, %1, 0x40\n s_load_dword %0, %1, 0x80\n s_load_dword %0, %1, 0xc0\n"
                     "s_load_dword %0, %1, 0x100\n s_load_dword %0, %1, 0x140\n s_load_dword %0, %1, 0x180\n s_load_dword %0, %1, 0x1c0\n s_waitcnt lgkmcnt(0)"
                     : "=&s"(d) : "s"(kp) : "memory");
    else
        asm volatile("s_load_dword %0, %1, 0x0\n s_load_dword %0, %1, 0x40\n s_load_dword %0, %1, 0x80\n s_load_dword %0, %1, 0xc0\n"
                     "s_load_dword %0, %1, 0x100\n s_load_dword %0, %1, 0x140\n s_load_dword %0, %1, 0x180\n s_load_dword %0, %1, 0x1c0\n"
                     "s_load_dword %0, %1, 0x200\n s_load_dword %0, %1, 0x240\n s_load_dword %0, %1, 0x280\n s_load_dword %0, %1, 0x2c0\n s_waitcnt lgkmcnt(0)"
                     : "=&s"(d) : "s"(kp) : "memory");
    static_assert(BYTES <= 768, "kernarg_warm: argument struct larger than the lines it touches");
}

// D = A(16x32 bf16) . B(32x16 bf16) + C.  Lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15],
// C/D[(l>>4)*4+r][l&15]  (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x4_t mfma_bf16_16x16x32(u16x8_t a, u16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ---- activations --------------------------------------------------------------------------------
// accurate forms (fp32 parity mode)
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_acc(float x) { return tanhf(x); }
// fast forms (speed mode): one v_exp_f32 + one v_rcp_f32 (both ~1 ulp), no correctly-rounded division sequence
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

template <bool FAST> __device__ __forceinline__ float sigm(float x) { return FAST ? sigmoid_fast(x) : sigmoid_acc(x); }
template <bool FAST> __device__ __forceinline__ float tanhx(float x) { return FAST ? tanh_fast(x) : tanh_acc(x); }

// ---- wave / block reductions (wave = 64) --------------------------------------------------------
// a lane's value through the VALU's data-parallel path (DPP) instead of a ds_bpermute round trip through the LDS queue: row rotations
// (0x120 + n: lane i of a 16-lane row reads lane (i + n) % 16) and quad permutations (four 2-bit selectors)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_f<0xB1>(v); }       // quad_perm [1,0,3,2] == __shfl_xor(v, 1)
// v[i] + v[i ^ 32] / v[i ^ 16] in every lane without the LDS: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half (the odd
// rows) of one register with the lower half (the even rows) of another -- of two copies of v that leaves (lower, lower) and (upper, upper),
// whose sum is the xor step's sum in every lane (a + b == b + a): bit-identical to v + __shfl_xor(v, 32 / 16)
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the xor partner's value itself (not a combination): of the swap's two results one is the lane's own half, the other the partner's
__device__ __forceinline__ unsigned xor32_get(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xor16_get(unsigned v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
}
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// The xor butterfly's steps 32 and 16 cross the rows (lane swaps, below); after them the four rows hold the same 16 values, and the steps 8, 4,
// 2, 1 are row rotations: after the step with distance d every lane equals its partner at distance d (a op b == b op a), so what a
// rotation by d / 2 brings is what the xor partner holds -- bit-identical to the all-xor form, four LDS round trips fewer.
// PRECONDITION of wave_sum / wave_max / block_argmax and the xor*_ helpers: EVERY lane of the wave is active.  The lane swaps skip inactive
// lanes and a DPP read of a disabled lane returns 0 (bound_ctrl): under divergence the result would be silently different.  Every caller
// reaches them on uniform control flow (lanes without a value contribute the operation's identity instead of branching around the call).
__device__ __forceinline__ float wave_sum(float v) {
    v = xor32_sum(v); v = xor16_sum(v);
    v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); v += dpp_f<0x121>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor32_max(v); v = xor16_max(v);
    v = fmaxf(v, dpp_f<0x128>(v)); v = fmaxf(v, dpp_f<0x124>(v)); v = fmaxf(v, dpp_f<0x122>(v)); v = fmaxf(v, dpp_f<0x121>(v));
    return v;
}
// block-wide sum for blockDim.x == NT (multiple of 64); red must hold NT/64 floats. All threads get the result.
template <int NT> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
template <int NT> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain vmcnt, so global loads
// issued earlier (prefetch) stay in flight across it.  Cross-wave data must travel through LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int NT> __device__ __forceinline__ float block_sum_lds(float v, float* red) {
    v = wave_sum(v);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
template <int NT> __device__ __forceinline__ float block_max_lds(float v, float* red) {
    v = wave_max(v);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}

// ---- cross-CU exchange granules -----------------------------------------------------------------
// 16-byte form: two granules {tag, a} {b, tag} written by ONE store and read by ONE load.  Each 8-byte half carries its
// own tag, so the pair is valid even if the 16 bytes are not delivered atomically (the consumer checks both tags).
// Buffer intrinsics so that the compiler tracks vmcnt for the loads; aux bit 0 = sc0, bit 4 = sc1.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t granule_rsrc(unsigned long long* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);        // wave-uniform base, raw addressing
}
__device__ __forceinline__ void granule16_store(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned tag, unsigned a, unsigned b,
                                                bool local) {
    const u32x4_t v = {tag, a, b, tag};
    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 1);          // sc0: stays in the XCD's L2
    else       __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 17);         // sc0 sc1: write-through
}
__device__ __forceinline__ u32x4_t granule16_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);                 // sc1: bypass L1
}

// 8-byte form {tag, a}: one aligned 64-bit access, delivered atomically
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void granule8_store(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned tag, unsigned a, bool local) {
    const u32x2_t v = {tag, a};
    if (local) __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte_off, 0, 1);
    else       __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte_off, 0, 17);
}
__device__ __forceinline__ u32x2_t granule8_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

extern "C" int las_colsum_dt(const void* X, int dtype, int rows, int cols, int ldx, float beta, float* out, void* ws,
                             size_t ws_bytes, void* stream);

// ---- skinny-M contraction (gemm.hip): C[M<=64, N] = A[M,K] . B + bias with B pre-packed to bf16 MFMA
// fragments.  Used by the Speller's per-step cell products, where M = batch rows and the weights are re-read
// every step: packing once per call makes every weight load a 1 KiB coalesced wave access.
size_t las_skinny_pack_bytes(int K, int N);
int las_skinny_pack(const float* W, int ldw, int K, int N, int transposed, void* packed, hipStream_t st);
int las_skinny_gemm(const float* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc,
                    const float* bias, hipStream_t st);
int las_skinny_gemm_bf16(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc,
                         const float* bias, hipStream_t st);
int las_skinny_gemm_bf16_tanh(const unsigned short* A, int lda, int M, int K, const void* packed, int N, const float* bias, float* eh, int ldh,
                              unsigned short* b0, int ld0, unsigned short* b1, int ld1, hipStream_t st);
int las_skinny_gemm_bf16_tanh_bwd(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, int c0, int D,
                                  const float* h, int ldh, const float* sa, int lda_, const float* sb, int ldb, int vlast, float* gp, int ldg,
                                  unsigned short* gb, int ldgb, hipStream_t st);
int las_skinny_gemm_bf16_lstm_bwd(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, int c0, int D,
                                  const float* sa, int lda_, const float* sb, int ldb, int vlast, float* gp, int ldg, unsigned short* gb, int ldgb,
                                  const float* c, const float* cp, float* dC, int ldc_, hipStream_t st);
int las_skinny_lstm_bf16(const unsigned short* A, int lda, int M, int K, const void* packed, int D, const float* bias, float fb, const float* cprev,
                         float* c, float* h, float* gates, unsigned short* b0, int ld0, unsigned short* b1, int ld1, hipStream_t st);
bool las_skinny_ok(int M, int K, int N, int lda, const void* A);

// ---- one LSTM cell step for a block of rows in one launch (loss_opt.hip, C entry las_lstm_cell_rows): z = [x ; h] . kernel + bias from
// las_skinny_pack fragments (Wx: [I, 4H], Wh: [H, 4H]; either part may be absent), TF gate order i, j, f, o, then the gate math.
// x fp32 or bf16 (x_bf16); one-hot input: x = NULL + ids / id_shift / xrows.  fast: the Speller's approximated transcendentals;
// gates_out (optional [M, 4H]): the ACTIVATED gates, as the Speller's backward pass reads them.
typedef las_lstm_cell_args LstmCellLaunch;            // (the public struct of include/las_hip.h)
int las_lstm_cell_rows_launch(const LstmCellLaunch& a, hipStream_t st);
int las_lstm_cell_check(const LstmCellLaunch& a);
// two independent cell steps as the two problems of ONE grid (a: fast + bf16 x = the Speller's; b: exact + fp32 / one-hot = the LM's)
int las_lstm_cell_rows_launch2(const LstmCellLaunch& a, const LstmCellLaunch& b, hipStream_t st);
