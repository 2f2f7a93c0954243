// gemm_kk.hip -- the dependency-chain contractions of the Listener in speed mode (K1 / K3):
//     C[M,N] = act( A[M,K] . B[N,K]^T + bias )          A, B bf16 with the CONTRACTION index contiguous, C bf16 or fp32
//
// Replaces, for the activations that now live in HBM as bf16, the fp32-operand path of gemm.hip whose loader converted
// fp32 -> bf16 on every tile (2x the algorithmic bytes and a per-CU L1-bandwidth bound).  Every chain
// product of the pyramidal listener has this shape once the weights are kept as bf16 shadows in both orientations:
//   x-projection   gates = x . [W_ih_fw | W_ih_bw]     (reference las/layers.py:31,49-53)   B = shadow of W^T  [2GH, I]
//   dense (+tanh)  y = tanh(out . Wd + b)              (las/layers.py:71-74, :89-93)        B = shadow of Wd^T [2H, 2H|4H]
//   their input gradients  dX = dY . W^T                                                    B = shadow of W    [K_in, N_out]
// Structure (cdna_hip_programming.md section 5): 128 x 128 x 64 tile, 4 waves (2 x 2), each wave 64 x 64 = 4 x 4
// v_mfma_f32_16x16x32_bf16 tiles; both operand tiles go HBM -> LDS with global_load_lds (16 B per lane, no VGPR round
// trip), double buffered with ONE barrier per k-tile (the next tile's DMA is in flight under the current tile's MFMAs);
// LDS rows are 128 B, the 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7 on the SOURCE address and on the
// fragment read (rule 21), which makes the ds_read_b128 fragment loads conflict free.  The MFMA operands are swapped
// (D = B_frag . A_frag) so that a lane ends up with 4 CONSECUTIVE output columns of one row: 8-byte (bf16) / 16-byte
// (fp32) stores, bias as one float4 per tile.
#include "las_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gconst_void_t;

struct KKArgs {
    const unsigned short* A; long long lda;     // [M, K] bf16, row pitch lda elements (multiple of 8)
    const unsigned short* B; long long ldb;     // [N, K] bf16
    void* C; long long ldc;                     // [M, N] bf16 or fp32
    const float* bias;                          // [N] or null
    const unsigned short* ymul; long long ldy;  // optional [M, N] bf16: C *= 1 - y^2 (the tanh gradient of the layer that produced A's consumer)
    int M, N, K, act, out_bf16;
    // optional frame selection ("row map"): logical row m of A and C is frame t of utterance b in a [B, T, *] tensor, with
    // b = m / fn, r = m % fn, t = r < nlo ? lo0 + r : hi0 + (r - nlo), fn = nlo + nhi (0 = identity).  Lets the x-projection of a
    // recurrent layer be computed in time chunks, both ends of the sequence first (las_gemm_kk_frames).
    int fT, lo0, nlo, hi0, fn;
};
__device__ __forceinline__ long long kk_row(const KKArgs& g, int m) {
    if (g.fn == 0) return m;
    const int b = m / g.fn, r = m - b * g.fn;
    return (long long)b * g.fT + (r < g.nlo ? g.lo0 + r : g.hi0 + (r - g.nlo));
}

constexpr int KK_BK = 64;
// Two tilings: 128 x 128 (4 waves as 2 x 2, wave tile 64 x 64, 2 workgroups per CU) and 256 x 256 (8 waves as 4 x 2, wave tile
// 64 x 128, one workgroup per CU).  With every CU streaming operand tiles the chip delivers ~13 bytes per clock and CU into LDS
// (MI355X_MICROARCH.md, "prologue HBM burst"; measured here: 2.0 GB of tile fills in 232 us = 8.6 TB/s for the x-projection),
// so the tall products are bound by tile fills, not by MFMA (27 % busy): the large tile moves half the bytes per flop.
template <int WM, int TN> struct KKCfg {
    static constexpr int NW = WM * 2, BM = WM * 64, BN = 2 * TN * 16;
    static constexpr int TILE_A = BM * KK_BK * 2, TILE_B = BN * KK_BK * 2, STAGE = TILE_A + TILE_B, LDS = 2 * STAGE;
    static_assert(BM / NW == 32 && BN / NW == 32, "every wave stages 32 rows of each operand tile");
};

template <int WM, int TN>
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void gemm_kk_kernel(KKArgs g) {
    using Cfg = KKCfg<WM, TN>;
    constexpr int KK_BM = Cfg::BM, KK_BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform (LDS-DMA base)
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile order (workgroup L runs on XCD L % 8, each XCD has its own L2): all column tiles of one row block go
    // to the same XCD back to back, so the 128-row slab of A is fetched into that L2 once
    const int nx = (g.N + KK_BN - 1) / KK_BN, ny = (g.M + KK_BM - 1) / KK_BM;
    const int L = blockIdx.x, xcd = L & 7, li = L >> 3;
    const int by = xcd + 8 * (li / nx), bx = li % nx;
    if (by >= ny) return;
    const int m0 = by * KK_BM, n0 = bx * KK_BN;

    // ---- LDS-DMA sources: wave w stages rows [w*32, w*32+32) of both tiles, 8 rows (1 KiB) per instruction; lane l of
    // piece q covers row w*32 + q*8 + (l >> 3), LDS position l & 7, i.e. global chunk (l & 7) ^ ((row >> 1) & 7)
    const unsigned short* pa[4];
    const unsigned short* pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = w * 32 + q * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        const int ra = min(m0 + row, g.M - 1), rb = min(n0 + row, g.N - 1);     // edge tiles: clamped, results never stored
        pa[q] = g.A + kk_row(g, ra) * g.lda + ch * 8;
        pb[q] = g.B + (long long)rb * g.ldb + ch * 8;
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        unsigned char* ab = smem + buf * Cfg::STAGE + w * 32 * 128;
        unsigned char* bb = smem + buf * Cfg::STAGE + Cfg::TILE_A + w * 32 * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((gconst_void_t*)pa[q], (lds_void_t*)(ab + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gconst_void_t*)pb[q], (lds_void_t*)(bb + q * 1024), 16, 0, 0);
            pa[q] += KK_BK; pb[q] += KK_BK;
        }
    };
    // ---- fragment reads: row (lane & 15) of a 16-row MFMA tile, k-chunk s*4 + (lane >> 4), swizzled like the source
    const int sw = (lane & 15) >> 1;
    const int fo0 = (lane & 15) * 128 + (((0 + (lane >> 4)) ^ sw) << 4);
    const int fo1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ sw) << 4);

    f32x4_t acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / KK_BK;
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile kt has landed
        __builtin_amdgcn_s_barrier();                         // ... everybody's has; everybody is done reading tile kt-1
#ifndef KK_ABL
#define KK_ABL 0                                              // timing experiments: 1 = no MFMAs, 2 = no tile fills after the first two
#endif
        if (kt + 1 < nk && !((KK_ABL & 2) && kt >= 1)) stage((kt + 1) & 1);   // flies under this tile's MFMAs
        if (KK_ABL & 1) continue;
        const unsigned char* As = smem + (kt & 1) * Cfg::STAGE + wm * 64 * 128;
        const unsigned char* Bs = smem + (kt & 1) * Cfg::STAGE + Cfg::TILE_A + wn * (TN * 16) * 128;
        // second k-half's 4 + TN fragment reads are issued behind the first MFMAs of the first half and land under the rest of them
        // (left to itself the scheduler puts every read directly in front of its MFMAs, behind an lgkmcnt(0))
        u16x8_t a[2][4], b[2][TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const u16x8_t*>(As + i * 2048 + fo0);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const u16x8_t*>(Bs + j * 2048 + fo0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[0][j] = mfma_bf16_16x16x32(b[0][j], a[0][0], acc[0][j]);       // swapped operands: D[n][m]
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[1][i] = *reinterpret_cast<const u16x8_t*>(As + i * 2048 + fo1);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[1][j] = *reinterpret_cast<const u16x8_t*>(Bs + j * 2048 + fo1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(b[0][j], a[0][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(b[1][j], a[1][i], acc[i][j]);
    }
    // ---- epilogue: lane holds C[m = tile row (lane & 15)][n = 4 consecutive columns (lane >> 4)*4 + r]
    const bool do_tanh = g.act == LAS_ACT_TANH;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 16) + j * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;                                  // N % 4 == 0
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) b4 = *reinterpret_cast<const float4*>(g.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = m0 + wm * 64 + i * 16 + (lane & 15);
            if (ml >= g.M) continue;
            const long long m = kk_row(g, ml);
            float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
            if (do_tanh) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
            if (g.ymul) {          // fused Tanh gradient: dX = dY * (1 - Y*Y), Y at the same [m, n]
                const uint2 yy = *reinterpret_cast<const uint2*>(g.ymul + m * g.ldy + n);
                const float y0 = __uint_as_float(yy.x << 16), y1 = __uint_as_float(yy.x & 0xffff0000u);
                const float y2 = __uint_as_float(yy.y << 16), y3 = __uint_as_float(yy.y & 0xffff0000u);
                v0 *= 1.f - y0 * y0; v1 *= 1.f - y1 * y1; v2 *= 1.f - y2 * y2; v3 *= 1.f - y3 * y3;
            }
            if (g.out_bf16) {
                uint2 pk;
                pk.x = f2bf2(v0, v1); pk.y = f2bf2(v2, v3);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + m * g.ldc + n) = pk;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + m * g.ldc + n) = make_float4(v0, v1, v2, v3);
            }
        }
    }
}

extern "C" int las_gemm_kk_tanhgrad(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                                    void* C, int c_dtype, long long ldc, const float* bias, int act, const void* y, long long ldy, void* stream);

extern "C" int las_gemm_kk(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                           void* C, int c_dtype, long long ldc, const float* bias, int act, void* stream) {
    return las_gemm_kk_tanhgrad(M, N, K, A, lda, B, ldb, C, c_dtype, ldc, bias, act, nullptr, 0, stream);
}

static int gemm_kk_impl(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C, int c_dtype, long long ldc,
                        const float* bias, int act, const void* y, long long ldy, int fT, int lo0, int nlo, int hi0, int nhi, void* stream);
static bool g_kk_big = true;
#ifdef LAS_DEV   // development builds only (make prof): A/B switch, not part of the shipping library
extern "C" void las_dev_gemm_kk_big(int on) { g_kk_big = on != 0; }        // development switch (A/B measurements)
#endif

extern "C" int las_gemm_kk_tanhgrad(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                                    void* C, int c_dtype, long long ldc, const float* bias, int act, const void* y, long long ldy, void* stream) {
    return gemm_kk_impl(M, N, K, A, lda, B, ldb, C, c_dtype, ldc, bias, act, y, ldy, 0, 0, 0, 0, 0, stream);
}

// A, C are [nb, T, *] tensors; only the frames [lo0, lo0 + nlo) and [hi0, hi0 + nhi) of every utterance are computed
extern "C" int las_gemm_kk_frames(int nb, int T, int lo0, int nlo, int hi0, int nhi, int N, int K, const void* A, long long lda,
                                  const void* B, long long ldb, void* C, int c_dtype, long long ldc, const float* bias, int act,
                                  const void* y, long long ldy, void* stream) {
    LAS_ARG(nb > 0 && T > 0 && nlo >= 0 && nhi >= 0 && nlo + nhi > 0 && lo0 >= 0 && lo0 + nlo <= T && hi0 >= 0 && hi0 + nhi <= T,
            "las_gemm_kk_frames: bad frame ranges");
    return gemm_kk_impl(nb * (nlo + nhi), N, K, A, lda, B, ldb, C, c_dtype, ldc, bias, act, y, ldy, T, lo0, nlo, hi0, nhi, stream);
}

static int gemm_kk_impl(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C, int c_dtype, long long ldc,
                        const float* bias, int act, const void* y, long long ldy, int fT, int lo0, int nlo, int hi0, int nhi, void* stream) {
    LAS_ARG(!y || (ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 7) == 0), "las_gemm_kk: y must be 8-byte aligned with a pitch that is a multiple of 4 and >= N");
    LAS_ARG(A && B && C, "las_gemm_kk: null operand");
    LAS_ARG(M > 0 && N > 0 && K > 0, "las_gemm_kk: bad dims M=%d N=%d K=%d", M, N, K);
    LAS_ARG(K % KK_BK == 0, "las_gemm_kk: K=%d must be a multiple of %d (pad the operands with zero columns)", K, KK_BK);
    LAS_ARG(N % 4 == 0, "las_gemm_kk: N=%d must be a multiple of 4", N);
    LAS_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "las_gemm_kk: row pitches must cover K / N and be multiples of 8");
    LAS_ARG(c_dtype == LAS_DT_F32 || c_dtype == LAS_DT_BF16, "las_gemm_kk: bad output type %d", c_dtype);
    LAS_ARG((ldc % 4) == 0, "las_gemm_kk: ldc must be a multiple of 4");
    LAS_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "las_gemm_kk: operands must be 16-byte aligned");
    LAS_ARG(act == LAS_ACT_NONE || act == LAS_ACT_TANH, "las_gemm_kk: bad act %d", act);
    static int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kk_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, KKCfg<2, 4>::LDS) |
                      (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kk_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, KKCfg<4, 8>::LDS);
    if (attr != 0) { las_set_error("hipFuncSetAttribute(gemm_kk) failed: %d", attr); return attr; }
    KKArgs g;
    g.A = (const unsigned short*)A; g.lda = lda; g.B = (const unsigned short*)B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.M = M; g.N = N; g.K = K; g.act = act; g.out_bf16 = c_dtype == LAS_DT_BF16;
    g.ymul = (const unsigned short*)y; g.ldy = ldy;
    g.fT = fT; g.lo0 = lo0; g.nlo = nlo; g.hi0 = hi0; g.fn = fT ? nlo + nhi : 0;
    // the large tile where it still fills the chip (>= 3/4 of a round of 256 workgroups) and N has no ragged 256-column edge
    constexpr int lds_big = KKCfg<4, 8>::LDS, lds_small = KKCfg<2, 4>::LDS;
    const bool big = g_kk_big && N % 256 == 0 && (long long)cdiv(M, 256) * (N / 256) >= 192;
    if (big) {
        const int nx = N / 256, ny = cdiv(M, 256);
        hipLaunchKernelGGL((gemm_kk_kernel<4, 8>), dim3(nx * ((ny + 7) / 8 * 8)), dim3(512), lds_big, (hipStream_t)stream, g);
    } else {
        const int nx = cdiv(N, 128), ny = cdiv(M, 128);
        hipLaunchKernelGGL((gemm_kk_kernel<2, 4>), dim3(nx * ((ny + 7) / 8 * 8)), dim3(256), lds_small, (hipStream_t)stream, g);
    }
    LAS_LAUNCHED();
    return 0;
}
