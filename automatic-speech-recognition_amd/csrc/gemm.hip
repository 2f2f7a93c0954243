// gemm.hip -- K1/K3/K4 dense contractions + their gradients for the LAS hot path (gfx950).
//
// Two arithmetic modes behind one entry point (include/las_hip.h: las_gemm):
//   LAS_PREC_F32  : exact fp32 FMA chains on the VALU (64x64x16 LDS tile, 4x4 per thread).
//   LAS_PREC_BF16 : operands rounded to bf16 while they are staged into LDS, fp32 accumulation on
//                   the matrix cores (v_mfma_f32_16x16x32_bf16), 64-wide waves, WM x WN waves per
//                   workgroup, TM x TN MFMA tiles per wave.
// Both take fp32 tensors with arbitrary (row,k) strides so the same kernel serves  x.W  (NN),
// dY.W^T (NT) and X^T.dY (TN), with a fused bias + tanh epilogue, batching (blockIdx.z) and a
// deterministic split-K (partials to workspace, fixed-order reduce) for the tall-K weight gradients.
#include "las_common.h"
#include <type_traits>

struct GemmArgs {
    int M, N, K;
    float alpha, beta;
    const float* A; long long rsA, ksA, strideA;   // op(A)(m,k) = A[m*rsA + k*ksA]
    const float* B; long long rsB, ksB, strideB;   // op(B)(k,n) = B[n*rsB + k*ksB]
    float* C; int ldc; long long strideC;
    const float* bias; int act;
    int mask_period, mask_skip;                    // zero contraction rows k with k % period == skip
    int vecA, vecB;                                // 16-byte vector loads legal
    int splitk, kchunk;                            // split-K: blockIdx.z = split, K range [z*kchunk, ..)
    float* partial;                                // [splitk][M][N] when splitk > 1
    int in_bf16;                                   // A and B are bf16 in HBM (speed-mode activations / gradients)
    int zgroup;                                    // > 0: 1-D grid, all tiles of one k-chunk / batch entry on one XCD (count of entries)
};

__device__ __forceinline__ float apply_act(float v, int act) { return act == LAS_ACT_TANH ? tanhf(v) : v; }

// ------------------------------------------------------------------------------------------------
// generic tile loader: ROWS x 32 (bf16 path) of a strided fp32 operand into registers
// ------------------------------------------------------------------------------------------------
template <int ROWS, int NT>
struct TileRegs { float4 v[(ROWS * 8 + NT - 1) / NT]; };

// Branch-free: every element is loaded from a clamped address and selected afterwards, so that all loads of a tile are in flight
// together (the round-4 form branched per chunk and per tail element, and every join waited for its loads: a k-tile's chunks were
// that many memory round trips in a row -- the K = 39 feature projection ran at 19 % of the fp32 matrix peak).
template <int ROWS, int NT>
__device__ __forceinline__ void tile_gload(TileRegs<ROWS, NT>& r, const float* __restrict__ X, long long rs,
                                           long long ks, int row0, int k0, int R, int Kend, int vec,
                                           int mperiod, int mskip) {
    constexpr int NCH = ROWS * 8;
    constexpr int RQ = ROWS / 4;
    // `vec` (16-byte loads legal: base, leading dimension and batch stride multiples of 4 floats) with an extent that is a multiple of 4: a chunk is
    // whole or empty, ONE 16-byte load from a clamped address (the per-element form had replaced the float4 path for every caller)
    const bool v4k = vec && (Kend & 3) == 0 && (k0 & 3) == 0, v4r = vec && (R & 3) == 0 && (row0 & 3) == 0;
#pragma unroll
    for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
        const int c = threadIdx.x + i * NT;
        const bool live = (NCH % NT == 0) || c < NCH;
        float e[4];
        if (ks == 1 && v4k) {
            const int row = c >> 3, kq = (c & 7) * 4;
            const int gr = row0 + row, gk = k0 + kq;
            const bool on = live && gr < R && gk < Kend;
            const float4 q = *reinterpret_cast<const float4*>(X + (long long)(on ? gr : 0) * rs + (on ? gk : 0));
            e[0] = on ? q.x : 0.f; e[1] = on ? q.y : 0.f; e[2] = on ? q.z : 0.f; e[3] = on ? q.w : 0.f;
        } else if (ks != 1 && v4r) {
            const int k = c / RQ, rq = (c % RQ) * 4;
            const int gr = row0 + rq, gk = k0 + k;
            const bool on = live && gk < Kend && !(mperiod > 0 && (gk % mperiod) == mskip) && gr < R;
            const float4 q = *reinterpret_cast<const float4*>(X + (long long)(on ? gk : 0) * ks + (on ? gr : 0));
            e[0] = on ? q.x : 0.f; e[1] = on ? q.y : 0.f; e[2] = on ? q.z : 0.f; e[3] = on ? q.w : 0.f;
        } else if (ks == 1) {  // contraction index contiguous in memory
            const int row = c >> 3, kq = (c & 7) * 4;
            const int gr = row0 + row, gk = k0 + kq;
            const float* p = X + (long long)(gr < R ? gr : 0) * rs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = live && gr < R && gk + j < Kend;
                const float v = p[on ? gk + j : 0];
                e[j] = on ? v : 0.f;
            }
        } else {        // row index contiguous in memory (rs == 1)
            const int k = c / RQ, rq = (c % RQ) * 4;
            const int gr = row0 + rq, gk = k0 + k;
            const bool kon = live && gk < Kend && !(mperiod > 0 && (gk % mperiod) == mskip);
            const float* p = X + (long long)(kon ? gk : 0) * ks;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = kon && gr + j < R;
                const float v = p[on ? gr + j : 0];
                e[j] = on ? v : 0.f;
            }
        }
        r.v[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

constexpr int LDK = 40;  // bf16 elements per LDS row: 32 + 8 pad (80 B, keeps 16-B alignment)

template <int ROWS, int NT>
__device__ __forceinline__ void tile_sstore(unsigned short* S, const TileRegs<ROWS, NT>& r, long long ks) {
    constexpr int NCH = ROWS * 8;
    constexpr int RQ = ROWS / 4;
#pragma unroll
    for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < NCH) {
            const float4 v = r.v[i];
            if (ks == 1) {
                const int row = c >> 3, kq = (c & 7) * 4;
                uint2 pk;
                pk.x = f2bf2(v.x, v.y); pk.y = f2bf2(v.z, v.w);
                *reinterpret_cast<uint2*>(&S[row * LDK + kq]) = pk;
            } else {
                const int k = c / RQ, rq = (c % RQ) * 4;
                const unsigned int p01 = f2bf2(v.x, v.y), p23 = f2bf2(v.z, v.w);
                S[(rq + 0) * LDK + k] = (unsigned short)(p01 & 0xffffu);
                S[(rq + 1) * LDK + k] = (unsigned short)(p01 >> 16);
                S[(rq + 2) * LDK + k] = (unsigned short)(p23 & 0xffffu);
                S[(rq + 3) * LDK + k] = (unsigned short)(p23 >> 16);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = WM * WN * 64;
    __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * LDK];
    unsigned short* As = lds;
    unsigned short* Bs = lds + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = g.A;
    const float* B = g.B;
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = blockIdx.z * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)blockIdx.z * g.strideA;
        B += (long long)blockIdx.z * g.strideB;
        C += (long long)blockIdx.z * g.strideC;
    }

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    TileRegs<BM, NT> ra;
    TileRegs<BN, NT> rb;
    if (kbeg < kend) {
        tile_gload<BM, NT>(ra, A, g.rsA, g.ksA, m0, kbeg, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
        tile_gload<BN, NT>(rb, B, g.rsB, g.ksB, n0, kbeg, g.N, kend, g.vecB, 0, 0);
    }
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();
        tile_sstore<BM, NT>(As, ra, g.ksA);
        tile_sstore<BN, NT>(Bs, rb, g.ksB);
        __syncthreads();
        if (k0 + 32 < kend) {  // next tile's global loads fly under this tile's MFMAs
            tile_gload<BM, NT>(ra, A, g.rsA, g.ksA, m0, k0 + 32, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
            tile_gload<BN, NT>(rb, B, g.rsB, g.ksB, n0, k0 + 32, g.N, kend, g.vecB, 0, 0);
        }
        u16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = *reinterpret_cast<const u16x8_t*>(&As[((wm * TM + i) * 16 + (lane & 15)) * LDK + (lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const u16x8_t*>(&Bs[((wn * TN + j) * 16 + (lane & 15)) * LDK + (lane >> 4) * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
    }

    // epilogue: lane holds rows (lane>>4)*4 + r, column lane&15 of each 16x16 tile
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * TM + i) * 16 + (lane >> 4) * 4 + r;
                if (row < g.M && col < g.N) {
                    if (g.splitk > 1) {
                        g.partial[((long long)blockIdx.z * g.M + row) * g.N + col] = acc[i][j][r];
                    } else {
                        float v = g.alpha * acc[i][j][r];
                        if (g.bias) v += g.bias[col];
                        float* cp = C + (long long)row * g.ldc + col;
                        if (g.beta != 0.f) v += g.beta * (*cp);
                        *cp = apply_act(v, g.act);
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, branch-free fast path: 16-byte loads legal on both operands, contiguity known at
// compile time (AKC/BKC: contraction index contiguous in memory), rows clamped instead of predicated
// (edge tiles compute throw-away rows), k tail zero-filled by a select.  LDS is double buffered: one
// barrier per k-tile, the next tile's global loads are issued before the MFMAs of the current one.
// ------------------------------------------------------------------------------------------------
// TI = float (operands converted to bf16 while they are staged) or unsigned short (operands already bf16 in HBM:
// speed-mode activations / gradients -- half the bytes, no conversion)
template <int ROWS, int NT, bool KC, typename TI>
struct FastRegs {
    // KC: one 4-element chunk (4 consecutive k).  !KC: a 4(rows) x 4(k) micro-tile per chunk = 4 chunks along rows.
    static constexpr int N = KC ? (ROWS * 8 + NT - 1) / NT : 4 * ((ROWS * 2 + NT - 1) / NT);
    typename std::conditional<std::is_same<TI, float>::value, float4, uint2>::type v[N];
};

__device__ __forceinline__ void ld4(float4& d, const float* p, bool on) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    d = on ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void ld4(uint2& d, const unsigned short* p, bool on) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    d = on ? v : make_uint2(0u, 0u);
}
__device__ __forceinline__ void zero4(float4& d) { d = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero4(uint2& d) { d = make_uint2(0u, 0u); }

template <int ROWS, int NT, bool KC, typename TI>
__device__ __forceinline__ void fast_gload(FastRegs<ROWS, NT, KC, TI>& r, const TI* __restrict__ X, long long rs, long long ks,
                                           int row0, int k0, int R, int Kend) {
    if (KC) {
        constexpr int NCH = ROWS * 8;
#pragma unroll
        for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
            const int c = threadIdx.x + i * NT;
            if (NCH % NT != 0 && c >= NCH) { zero4(r.v[i]); continue; }
            const int row = c >> 3, kq = (c & 7) * 4;
            const int gr = min(row0 + row, R - 1), gk = k0 + kq;
            const bool on = gk + 3 < Kend;                         // K % 4 == 0 on this path
            ld4(r.v[i], X + (long long)gr * rs + (on ? gk : 0), on);
        }
    } else {
        constexpr int NCH = ROWS * 2, RQ = ROWS / 4;
#pragma unroll
        for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
            const int c = threadIdx.x + i * NT;
            const bool live = (NCH % NT == 0) || c < NCH;
            const int kq = (c / RQ) * 4, rq = (c % RQ) * 4;
            const int gr = min(row0 + rq, R - 4);                  // R % 4 == 0 on this path
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int gk = k0 + kq + kk;
                const bool on = live && gk < Kend;
                ld4(r.v[i * 4 + kk], X + (long long)(on ? gk : 0) * ks + gr, on);
            }
        }
    }
}

// 4 consecutive k of one row -> 8 bytes of bf16 in LDS
__device__ __forceinline__ uint2 pack_k4(const float4& v) { uint2 pk; pk.x = f2bf2(v.x, v.y); pk.y = f2bf2(v.z, v.w); return pk; }
__device__ __forceinline__ uint2 pack_k4(const uint2& v) { return v; }
// register micro-transpose of a 4(rows) x 4(k) tile held as 4 row-chunks (one per k): row j gets its 4 consecutive k
__device__ __forceinline__ void transpose4(const float4& k0v, const float4& k1v, const float4& k2v, const float4& k3v,
                                           uint2& p0, uint2& p1, uint2& p2, uint2& p3) {
    p0.x = f2bf2(k0v.x, k1v.x); p0.y = f2bf2(k2v.x, k3v.x);
    p1.x = f2bf2(k0v.y, k1v.y); p1.y = f2bf2(k2v.y, k3v.y);
    p2.x = f2bf2(k0v.z, k1v.z); p2.y = f2bf2(k2v.z, k3v.z);
    p3.x = f2bf2(k0v.w, k1v.w); p3.y = f2bf2(k2v.w, k3v.w);
}
__device__ __forceinline__ void transpose4(const uint2& k0v, const uint2& k1v, const uint2& k2v, const uint2& k3v,
                                           uint2& p0, uint2& p1, uint2& p2, uint2& p3) {
    // k?v.x = rows (0,1), k?v.y = rows (2,3) of column k, 16 bits each
    p0.x = (k0v.x & 0xffffu) | (k1v.x << 16);         p0.y = (k2v.x & 0xffffu) | (k3v.x << 16);
    p1.x = (k0v.x >> 16) | (k1v.x & 0xffff0000u);     p1.y = (k2v.x >> 16) | (k3v.x & 0xffff0000u);
    p2.x = (k0v.y & 0xffffu) | (k1v.y << 16);         p2.y = (k2v.y & 0xffffu) | (k3v.y << 16);
    p3.x = (k0v.y >> 16) | (k1v.y & 0xffff0000u);     p3.y = (k2v.y >> 16) | (k3v.y & 0xffff0000u);
}

template <int ROWS, int NT, bool KC, typename TI>
__device__ __forceinline__ void fast_sstore(unsigned short* S, const FastRegs<ROWS, NT, KC, TI>& r) {
    if (KC) {
        constexpr int NCH = ROWS * 8;
#pragma unroll
        for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
            const int c = threadIdx.x + i * NT;
            if (NCH % NT != 0 && c >= NCH) continue;
            const int row = c >> 3, kq = (c & 7) * 4;
            *reinterpret_cast<uint2*>(&S[row * LDK + kq]) = pack_k4(r.v[i]);
        }
    } else {
        constexpr int NCH = ROWS * 2, RQ = ROWS / 4;
#pragma unroll
        for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
            const int c = threadIdx.x + i * NT;
            if (NCH % NT != 0 && c >= NCH) continue;
            const int kq = (c / RQ) * 4, rq = (c % RQ) * 4;
            uint2 p0, p1, p2, p3;
            transpose4(r.v[i * 4 + 0], r.v[i * 4 + 1], r.v[i * 4 + 2], r.v[i * 4 + 3], p0, p1, p2, p3);
            *reinterpret_cast<uint2*>(&S[(rq + 0) * LDK + kq]) = p0;
            *reinterpret_cast<uint2*>(&S[(rq + 1) * LDK + kq]) = p1;
            *reinterpret_cast<uint2*>(&S[(rq + 2) * LDK + kq]) = p2;
            *reinterpret_cast<uint2*>(&S[(rq + 3) * LDK + kq]) = p3;
        }
    }
}

template <int WM, int WN, int TM, int TN, bool AKC, bool BKC, typename TI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_fast_kernel(GemmArgs g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = WM * WN * 64;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * (BM + BN) * LDK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    // XCD-aware tile order: workgroup L runs on XCD L % 8 (each XCD has its own L2).  All column tiles of one row block
    // go to the same XCD back to back, so the row block of A is fetched into that L2 once instead of once per XCD.
    // (Only for tall outputs: with fewer than a few row blocks per XCD the plain 2-D order fills the chip better.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.zgroup) {
        // split-K / batched contractions with few output tiles (the weight gradients): 1-D grid, ALL tiles of one k-chunk (or
        // batch entry) on ONE XCD and next to each other in dispatch order, so that the chunk's operand rows are fetched into that
        // XCD's L2 once and shared by its tiles instead of every tile pulling them from HBM (the 3-D order spread them over 8 L2s)
        const int nx = (g.N + BN - 1) / BN, ny = (g.M + BM - 1) / BM, nt = nx * ny;
        const int L = blockIdx.x, xcd = L & 7, li = L >> 3;
        bz = xcd + 8 * (li / nt);
        if (bz >= g.zgroup) return;
        const int t = li % nt;
        by = t / nx; bx = t % nx;
    } else if (gridDim.y == 1 && g.M > BM) {
        const int nx = (g.N + BN - 1) / BN, ny = (g.M + BM - 1) / BM;
        const int L = blockIdx.x, xcd = L & 7, li = L >> 3;
        by = xcd + 8 * (li / nx); bx = li % nx;
        if (by >= ny) return;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const TI* A = reinterpret_cast<const TI*>(g.A);
    const TI* B = reinterpret_cast<const TI*>(g.B);
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = bz * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)bz * g.strideA;
        B += (long long)bz * g.strideB;
        C += (long long)bz * g.strideC;
    }
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    FastRegs<BM, NT, AKC, TI> ra;
    FastRegs<BN, NT, BKC, TI> rb;
    fast_gload<BM, NT, AKC, TI>(ra, A, g.rsA, g.ksA, m0, kbeg, g.M, kend);
    fast_gload<BN, NT, BKC, TI>(rb, B, g.rsB, g.ksB, n0, kbeg, g.N, kend);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        unsigned short* As = lds + buf * (BM + BN) * LDK;
        unsigned short* Bs = As + BM * LDK;
        fast_sstore<BM, NT, AKC, TI>(As, ra);
        fast_sstore<BN, NT, BKC, TI>(Bs, rb);
        __syncthreads();                       // tile visible; the other buffer is free (its readers passed the previous barrier)
        if (k0 + 32 < kend) {
            fast_gload<BM, NT, AKC, TI>(ra, A, g.rsA, g.ksA, m0, k0 + 32, g.M, kend);
            fast_gload<BN, NT, BKC, TI>(rb, B, g.rsB, g.ksB, n0, k0 + 32, g.N, kend);
        }
        u16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = *reinterpret_cast<const u16x8_t*>(&As[((wm * TM + i) * 16 + (lane & 15)) * LDK + (lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const u16x8_t*>(&Bs[((wn * TN + j) * 16 + (lane & 15)) * LDK + (lane >> 4) * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
        buf ^= 1;
    }
    const bool has_bias = g.bias != nullptr, has_beta = g.beta != 0.f, do_tanh = g.act == LAS_ACT_TANH;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 16 + (lane & 15);
            const float bcol = (has_bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * TM + i) * 16 + (lane >> 4) * 4 + r;
                if (row < g.M && col < g.N) {
                    if (g.splitk > 1) {
                        g.partial[((long long)bz * g.M + row) * g.N + col] = acc[i][j][r];
                    } else {
                        float v = g.alpha * acc[i][j][r] + bcol;
                        float* cp = C + (long long)row * g.ldc + col;
                        if (has_beta) v += g.beta * (*cp);
                        *cp = do_tanh ? tanh_fast(v) : v;
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// TN contraction with bf16 operands that are BOTH k-strided (the weight gradients  dW = X^T . dZ : contraction over frames,
// rows of X and dZ are frames).  The generic fast kernel above turns the [k][m] tiles into [m][k] LDS rows with a 4 x 4
// register transpose and four 8-byte LDS stores per lane -- rows 4 apart are 80 dwords apart, 16 mod 32, so every one of those
// stores is 16-way bank conflicted, and the ablation builds (GEMM_ABL) showed the stores bound the kernel: 190 us for dW_ih of
// a T = 1274 layer, 189 without the global loads, 77 without loads and stores.  Here the tiles stay [k][m] in LDS (16-byte
// loads, 16-byte conflict-free stores, no VALU work) and the transposition happens in the LDS READ: ds_read_b64_tr_b16 hands
// lane i of a 16-lane group column i of a [4 k][16 m] block (tools/micro/probe_tr.hip), two of them are one MFMA fragment.
// 128 x 128 x 32 tiles, 4 waves (2 x 2), rows of 256 + 32 bytes: the four k-rows of a block land in four different bank octets.
// ------------------------------------------------------------------------------------------------
constexpr int TR_PITCH = 288;                                        // bytes per LDS k-row (128 elements + 16 pad)
constexpr int TR_TILE = 32 * TR_PITCH;                               // one operand tile
typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_v4s lds_v4s_t;

__device__ __forceinline__ u16x8_t tr_frag(const unsigned char* tile, int col0, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned char* p = tile + (g * 8 + (l15 >> 2)) * TR_PITCH + (col0 + (l15 & 3) * 4) * 2;
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(__attribute__((address_space(3))) unsigned char*)p);
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(__attribute__((address_space(3))) unsigned char*)(p + 4 * TR_PITCH));
    u16x8_t f;
    f[0] = (unsigned short)lo[0]; f[1] = (unsigned short)lo[1]; f[2] = (unsigned short)lo[2]; f[3] = (unsigned short)lo[3];
    f[4] = (unsigned short)hi[0]; f[5] = (unsigned short)hi[1]; f[6] = (unsigned short)hi[2]; f[7] = (unsigned short)hi[3];
    return f;
}

// BM = 128: waves 2 x 2, wave tile 64 x 64.  BM = 64 (a first-layer dW_ih: 40 input features): waves 1 x 4, wave tile 64 x 32; the A
// tile's columns beyond M (a multiple of 8) are neither read nor stored.
template <int BM>
__global__ __launch_bounds__(256, 2) void gemm_tn_tr_kernel(GemmArgs g) {
    constexpr int BN = 128, WN = BM == 128 ? 2 : 4, TNF = BM == 128 ? 4 : 2;
    constexpr int APR = BM / 8, APASS = APR / 8, AROWS = 256 / APR;      // A staging: 16-byte pieces per k-row, passes, k-rows per pass
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TR_TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.zgroup) {                                                  // all tiles of one k-chunk / batch entry on one XCD (see the fast kernel)
        const int nx = g.N / BN, ny = (g.M + BM - 1) / BM, nt = nx * ny;
        const int L = blockIdx.x, xcd = L & 7, li = L >> 3;
        bz = xcd + 8 * (li / nt);
        if (bz >= g.zgroup) return;
        const int t = li % nt;
        by = t / nx; bx = t % nx;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* B = reinterpret_cast<const unsigned short*>(g.B);
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = bz * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)bz * g.strideA;
        B += (long long)bz * g.strideB;
        C += (long long)bz * g.strideC;
    }
    f32x4_t acc[4][TNF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // staging: a piece is 16 bytes = 8 columns of one k-row; B: piece tid + 256 u (u = 0, 1) of k-row (tid >> 4) + 16 u;
    // A: the same at BM = 128, one piece of k-row tid >> 3 at BM = 64
    const int pk = tid >> 4, pc = (tid & 15) * 8;
    const int ak = tid / APR, ac = (tid % APR) * 8;
    const unsigned short* pa = A + (long long)(kbeg + ak) * g.ksA + m0 + ac;
    const unsigned short* pb = B + (long long)(kbeg + pk) * g.ksB + n0 + pc;
    const long long saR = (long long)AROWS * g.ksA, sb16 = 16 * g.ksB, sa32 = 32 * g.ksA, sb32 = 32 * g.ksB;
    const int so = pk * TR_PITCH + pc * 2, soa = ak * TR_PITCH + ac * 2;
    u32x4_t ra[APASS], rb[2];
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    auto gload = [&](int k0) __attribute__((always_inline)) {
        // (default cache policy: non-temporal loads lose the L2 sharing between the tiles of a k-chunk -- 124 -> 155 us)
#pragma unroll
        for (int u = 0; u < APASS; ++u) {
            const bool on = k0 + ak + AROWS * u < kend && (BM == 128 || ac < g.M);     // rows past the contraction range (and, at BM = 64,
                                                                                       // columns past M: a multiple of 8) read as zero
            const u32x4_t va = *reinterpret_cast<const u32x4_t*>(on ? pa + u * saR : A);
            ra[u] = on ? va : zero;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = k0 + pk + 16 * u < kend;
            const u32x4_t vb = *reinterpret_cast<const u32x4_t*>(on ? pb + u * sb16 : B);
            rb[u] = on ? vb : zero;
        }
        pa += sa32; pb += sb32;
    };
    gload(kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        unsigned char* As = lds + buf * 2 * TR_TILE;
        unsigned char* Bs = As + TR_TILE;
#pragma unroll
        for (int u = 0; u < APASS; ++u) *reinterpret_cast<u32x4_t*>(As + soa + u * AROWS * TR_PITCH) = ra[u];
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<u32x4_t*>(Bs + so + u * 16 * TR_PITCH) = rb[u];
        __syncthreads();                       // tile visible; the other buffer is free (its readers passed the previous barrier)
        if (k0 + 32 < kend) gload(k0 + 32);
        u16x8_t a[4], b[TNF];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = tr_frag(As, wm * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < TNF; ++j) b[j] = tr_frag(Bs, wn * (TNF * 16) + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TNF; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
        buf ^= 1;
    }
    const bool has_bias = g.bias != nullptr, has_beta = g.beta != 0.f, do_tanh = g.act == LAS_ACT_TANH;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNF; ++j) {
            const int col = n0 + (wn * TNF + j) * 16 + (lane & 15);
            const float bcol = has_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * 4 + i) * 16 + (lane >> 4) * 4 + r;
                if (BM == 64 && row >= g.M) continue;
                if (g.splitk > 1) {
                    g.partial[((long long)bz * g.M + row) * g.N + col] = acc[i][j][r];
                } else {
                    float v = g.alpha * acc[i][j][r] + bcol;
                    float* cp = C + (long long)row * g.ldc + col;
                    if (has_beta) v += g.beta * (*cp);
                    *cp = do_tanh ? tanh_fast(v) : v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Both weight gradients of one direction of a recurrent layer in ONE pass over d(pre-activation) (round 4):
//     dW[0 : I, :]     += X^T . dZ                      (dW_ih: contraction over all B T frames)
//     dW[I : I + H, :] += sum_b sum_t h_prev(b, t)^T . dZ(b, t)     (dW_hh: h_prev = the layer's own output one step back in sweep order)
// i.e. dW = [X | H_prev]^T . dZ with a VIRTUAL left operand: row blocks below `nb1` read X, the others read the layer output shifted by
// one frame (out[b, t - 1] forward direction, out[b, t + 1] backward direction; the frame without a predecessor contributes zero).  Until
// round 3 these were two products -- a flat split-K TN product for dW_ih and a per-utterance batched one for dW_hh whose 48 partial
// [H, G H] tiles went through HBM and a column-sum kernel -- each reading all of dZ (125 MB per direction at T = 1274): 1.8 GB per train
// step of side-stream traffic next to the BPTT sweeps.  Same tile machinery as gemm_tn_tr_kernel<128> (k-major LDS tiles, transposing
// LDS reads), all row blocks of one k-chunk on ONE XCD so that the dZ rows enter that L2 once.
// ------------------------------------------------------------------------------------------------
struct WgradDir {
    const unsigned short* X;                       // layer input [K, ldx] bf16 (the direction's own copy under input dropout)
    const unsigned short* O; int shift;            // this direction's column block of the layer output; h_prev = frame t + shift (-1 / +1)
    const unsigned short* Z;                       // this direction's column block of d(pre-activation) [K, ldz] bf16
    float* dW;                                     // [I + H, N] fp32, accumulated
    int t0;                                        // frame window (WIN kernels): virtual row k' = b nf + i is frame t0 + i of utterance b
};
struct WgradArgs {
    WgradDir d[2]; int ndir;
    int ldx, M1;                                   // M1 = columns of X that exist (multiple of 8)
    int ldo; long long obs;                        // layer output: row pitch, batch stride
    int ldz;
    int T, K, N, nb1, nb2;                         // frames per utterance, K = B T (WIN: B nf), N = G H, row blocks from X / from the output
    float invT;
    int nf; float invnf;                           // WIN: frames per utterance inside the window
    int splitk, kchunk;
    float* partial;                                // [ndir][splitk][(nb1 + nb2) * 128][N]
};

// (Measured, round 4: two or three k-steps requested ahead of the one in the matrix cores -- 16 staging registers each -- do NOT make it
//  faster: 150 / 163 vs 144 us for a direction of the bottom layer.  The kernel is not waiting for memory: per k-step and CU the
//  vector-memory path (32 KB), the LDS (64 KB of fragment reads + 32 KB of tile writes) and the matrix cores (2 x 512 clocks) are each
//  25-50 % busy and overlap poorly with two workgroups per CU; four per CU (twice the k-chunks) is what helps the two-direction launch.)
// WIN: only a WINDOW of nf frames per utterance is contracted (las_wgrad_ih_hh_window, round 5: the bottom layer's weight gradients follow
// the BPTT sweep that is still producing dZ, window by window, instead of waiting for its end) -- the virtual row k' = b nf + i maps to frame
// t0 + i of utterance b for all three operands; WIN = false is the whole-sequence kernel, unchanged.
template <bool WIN>
__global__ __launch_bounds__(256, 4) void wgrad_tn_tr_kernel(WgradArgs g) {      // (4 waves per SIMD: 128 VGPRs -- at 130 the k-chunks of a tile run three to a CU instead of four)
    constexpr int BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TR_TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int nx = g.N / BN, ny = g.nb1 + g.nb2, nt = nx * ny, ntd = nt * g.ndir;
    const int L = blockIdx.x, xcd = L & 7, li = L >> 3;
    const int bz = xcd + 8 * (li / ntd);
    if (bz >= g.splitk) return;
    const int td = li % ntd, dir = td / nt, t_ = td - dir * nt, by = t_ / nx, bx = t_ % nx;   // both directions of a k-chunk on one XCD: X enters that L2 once
    const WgradDir gd = g.d[dir];
    const int n0 = bx * BN;
    const bool hsrc = by >= g.nb1;
    const int m0 = (hsrc ? by - g.nb1 : by) * BM;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int pk = tid >> 4, pc = (tid & 15) * 8;                   // a piece is 16 bytes = 8 columns of one k-row: piece tid + 256 u of k-row pk + 16 u
    const unsigned short* pb = gd.Z + (long long)(kbeg + pk) * g.ldz + n0 + pc;
    const long long sb16 = 16LL * g.ldz, sb32 = 32LL * g.ldz;
    const int so = pk * TR_PITCH + pc * 2;
    u32x4_t ra[2], rb[2];
    bool oa[2], ob[2];                                              // (the zeroing select waits for the load: applied when the piece is written to LDS, an iteration later)
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + pk + 16 * u;
            bool on = k < kend;
            const unsigned short* src;
            if (WIN) {
                int b = (int)((float)k * g.invnf);                   // k = b nf + i  (k < 2^24: the float quotient is off by at most one)
                if (b * g.nf > k) --b;
                if ((b + 1) * g.nf <= k) ++b;
                const int tf = gd.t0 + k - b * g.nf;                 // the frame
                const long long kr = (long long)b * g.T + tf;        // its row in the [B T, .] operands
                const bool onb = on;
                if (!hsrc) {
                    on = on && m0 + pc < g.M1;
                    src = gd.X + (on ? kr : 0LL) * g.ldx + m0 + pc;
                } else {
                    const int tp = tf + gd.shift;
                    on = on && tp >= 0 && tp < g.T;
                    src = gd.O + (on ? (long long)b * g.obs + (long long)tp * g.ldo : 0LL) + m0 + pc;
                }
                const u32x4_t va = *reinterpret_cast<const u32x4_t*>(on ? src : gd.Z);
                ra[u] = va; oa[u] = on;
                const u32x4_t vb = *reinterpret_cast<const u32x4_t*>(onb ? gd.Z + kr * g.ldz + n0 + pc : gd.Z);
                rb[u] = vb; ob[u] = onb;
                continue;
            }
            if (!hsrc) {
                on = on && m0 + pc < g.M1;
                src = gd.X + (long long)(on ? k : 0) * g.ldx + m0 + pc;
            } else {
                int b = (int)((float)k * g.invT);                    // k = b T + t  (k < 2^24: the float quotient is off by at most one)
                if (b * g.T > k) --b;
                if ((b + 1) * g.T <= k) ++b;
                const int tp = k - b * g.T + gd.shift;
                on = on && tp >= 0 && tp < g.T;
                src = gd.O + (on ? (long long)b * g.obs + (long long)tp * g.ldo : 0LL) + m0 + pc;
            }
            const u32x4_t va = *reinterpret_cast<const u32x4_t*>(on ? src : gd.Z);
            ra[u] = va; oa[u] = on;
            const bool onb = k < kend;
            const u32x4_t vb = *reinterpret_cast<const u32x4_t*>(onb ? pb + u * sb16 : gd.Z);
            rb[u] = vb; ob[u] = onb;
        }
        pb += sb32;
    };
    gload(kbeg);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        unsigned char* As = lds + buf * 2 * TR_TILE;
        unsigned char* Bs = As + TR_TILE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<u32x4_t*>(As + so + u * 16 * TR_PITCH) = oa[u] ? ra[u] : zero;
            *reinterpret_cast<u32x4_t*>(Bs + so + u * 16 * TR_PITCH) = ob[u] ? rb[u] : zero;
        }
        __syncthreads();
        gload(k0 + 32);                                             // unconditional (behind the last k-step every piece is off: one dummy address) -- behind an
                                                                    // `if` the compiler ends the branch in s_waitcnt vmcnt(0) and the prefetch lands before the MFMAs
        u16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = tr_frag(As, wm * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = tr_frag(Bs, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
        buf ^= 1;
    }
    float* P = g.partial + (((long long)dir * g.splitk + bz) * ny + by) * BM * g.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + (wn * 4 + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(long long)((wm * 4 + i) * 16 + (lane >> 4) * 4 + r) * g.N + col] = acc[i][j][r];
        }
}

// fixed-order reduction of the k-chunks: virtual row r of block row `by` -> dW row (X blocks: r < I; output blocks: I + r)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs g, int I, int H) {
    const int ny = g.nb1 + g.nb2;
    const long long per = (long long)ny * 128 * g.N, total = per * g.ndir;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int dir = (int)(idx / per);
        const long long e = idx - dir * per;
        const int vr = (int)(e / g.N), col = (int)(e % g.N), by = vr >> 7, r = vr & 127;
        int row;
        if (by < g.nb1) { row = by * 128 + r; if (row >= I) continue; }
        else { row = (by - g.nb1) * 128 + r; if (row >= H) continue; row += I; }
        const float* P = g.partial + (long long)dir * g.splitk * per + e;
        float s_ = 0.f;
        for (int z = 0; z < g.splitk; ++z) s_ += P[(long long)z * per];
        g.d[dir].dW[(long long)row * g.N + col] += s_;
    }
}

static int wgrad_split(int I, int H, int GH, long long K, int ndir, int* kchunk) {
    const int ny = cdiv(I, 128) + H / 128, tiles = ny * (GH / 128) * ndir;
    const int wgs = ndir == 2 ? 1024 : 512;       // workgroups a launch aims at: two per CU; four for the two-direction launch (383 -> 260 us at the bottom layer)
    int s = (wgs + tiles - 1) / tiles;
    if (s > K / 512) s = (int)(K / 512);
    if (s < 1) s = 1;
    const int kc = (int)(((K + s - 1) / s + 31) / 32 * 32);
    if (kchunk) *kchunk = kc;
    return (int)((K + kc - 1) / kc);
}
extern "C" size_t las_wgrad_ih_hh_workspace_bytes(int I, int H, int GH, int B, int T, int ndir) {
    if (I <= 0 || H <= 0 || GH <= 0 || B <= 0 || T <= 0 || ndir < 1 || ndir > 2) return 0;
    const int ny = cdiv(I, 128) + H / 128;
    int s = wgrad_split(I, H, GH, (long long)B * T, ndir, nullptr);
    if (s < wgrad_split(I, H, GH, (long long)B * T, 1, nullptr)) s = wgrad_split(I, H, GH, (long long)B * T, 1, nullptr);
    return (size_t)ndir * s * ny * 128 * GH * sizeof(float);
}

// dir = 0 / 1: one direction (dW = that direction's gradient, dW2 ignored); dir = 2: BOTH in one launch (dW = forward, dW2 = backward
// direction, X2 = the backward direction's input copy or NULL = X).  out / dZ: the [.., 2 H] / [.., 2 G H] tensors of both directions.
static int wgrad_impl(const void* X, const void* X2, int ldx, int I, const void* out, int ld_out, long long out_bstride, const void* dZ, int lddz,
                      int B, int T, int H, int GH, int dir, float* dW, float* dW2, void* ws, size_t ws_bytes, void* stream,
                      bool win, int t0_fw, int t0_bw, int nf, int max_wgs) {
    LAS_ARG(X && out && dZ && dW && ws && dir >= 0 && dir <= 2 && (dir < 2 || dW2), "las_wgrad_ih_hh: null pointer / bad direction");
    LAS_ARG(B > 0 && T > 0 && I > 0 && H > 0 && H % 128 == 0 && GH % 128 == 0, "las_wgrad_ih_hh: needs H and G H multiples of 128 (I=%d H=%d GH=%d)", I, H, GH);
    LAS_ARG((long long)B * T < (1 << 24), "las_wgrad_ih_hh: B T must stay below 2^24");
    LAS_ARG(ldx % 8 == 0 && ldx >= (I + 7) / 8 * 8 && ld_out % 8 == 0 && lddz % 8 == 0 && out_bstride % 8 == 0 &&
            (((uintptr_t)X | (uintptr_t)X2 | (uintptr_t)out | (uintptr_t)dZ) & 15) == 0, "las_wgrad_ih_hh: operands must be 16-byte aligned with pitches that are multiples of 8");
    LAS_ARG(!win || (nf > 0 && t0_fw >= 0 && t0_bw >= 0 && t0_fw + nf <= T && t0_bw + nf <= T), "las_wgrad_ih_hh_window: bad frame window");
    WgradArgs g;
    g.ndir = dir == 2 ? 2 : 1;
    for (int i = 0; i < g.ndir; ++i) {
        const int d = dir == 2 ? i : dir;
        g.d[i].X = (const unsigned short*)((d == 1 && X2) ? X2 : X);
        g.d[i].O = (const unsigned short*)out + (size_t)d * H; g.d[i].shift = d ? 1 : -1;
        g.d[i].Z = (const unsigned short*)dZ + (size_t)d * GH;
        g.d[i].dW = (dir == 2 && i == 1) ? dW2 : dW;
        g.d[i].t0 = d ? t0_bw : t0_fw;
    }
    if (g.ndir == 1) g.d[1] = g.d[0];
    g.ldx = ldx; g.M1 = (I + 7) / 8 * 8; g.ldo = ld_out; g.obs = out_bstride; g.ldz = lddz;
    g.T = T; g.K = win ? B * nf : B * T; g.N = GH; g.nb1 = cdiv(I, 128); g.nb2 = H / 128;
    g.invT = 1.0f / (float)T;
    g.nf = win ? nf : T; g.invnf = 1.0f / (float)g.nf;
    const int ny = g.nb1 + g.nb2, nt = ny * (GH / 128) * g.ndir;
    g.splitk = wgrad_split(I, H, GH, g.K, g.ndir, &g.kchunk);
    if (win && max_wgs > 0 && nt * g.splitk > max_wgs) {      // a window that runs BESIDE a sweep: few workgroups, long k-chunks
        int s_ = max_wgs / nt;
        if (s_ < 1) s_ = 1;
        g.kchunk = (int)((((long long)g.K + s_ - 1) / s_ + 31) / 32 * 32);
        g.splitk = (g.K + g.kchunk - 1) / g.kchunk;
    }
    g.partial = (float*)ws;
    LAS_ARG(ws_bytes >= (size_t)g.ndir * g.splitk * ny * 128 * GH * sizeof(float), "las_wgrad_ih_hh: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (win) hipLaunchKernelGGL(wgrad_tn_tr_kernel<true>, dim3(nt * ((g.splitk + 7) / 8 * 8)), dim3(256), 0, st, g);
    else     hipLaunchKernelGGL(wgrad_tn_tr_kernel<false>, dim3(nt * ((g.splitk + 7) / 8 * 8)), dim3(256), 0, st, g);
    LAS_LAUNCHED();
    int nb = cdiv((long long)g.ndir * ny * 128 * GH, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, st, g, I, H);
    LAS_LAUNCHED();
    return 0;
}
extern "C" int las_wgrad_ih_hh(const void* X, const void* X2, int ldx, int I, const void* out, int ld_out, long long out_bstride, const void* dZ, int lddz,
                               int B, int T, int H, int GH, int dir, float* dW, float* dW2, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_impl(X, X2, ldx, I, out, ld_out, out_bstride, dZ, lddz, B, T, H, GH, dir, dW, dW2, ws, ws_bytes, stream, false, 0, 0, 0, 0);
}
extern "C" int las_wgrad_ih_hh_window(const void* X, const void* X2, int ldx, int I, const void* out, int ld_out, long long out_bstride, const void* dZ, int lddz,
                                      int B, int T, int H, int GH, int dir, int t0_fw, int t0_bw, int nframes, int max_workgroups,
                                      float* dW, float* dW2, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_impl(X, X2, ldx, I, out, ld_out, out_bstride, dZ, lddz, B, T, H, GH, dir, dW, dW2, ws, ws_bytes, stream, true, t0_fw, t0_bw, nframes, max_workgroups);
}

static bool g_tn_tr_on = true;
#ifdef LAS_DEV   // development builds only (make prof): A/B switch, not part of the shipping library
extern "C" void las_dev_gemm_tn_tr(int on) { g_tn_tr_on = on != 0; }       // development switch (A/B measurements)
#endif
static bool g_f32_valu = false;      // parity-mode products through the round-1 VALU tile loop instead of the exact-fp32 MFMA kernel
#ifdef LAS_DEV   // development builds only (make prof): A/B switch, not part of the shipping library
extern "C" void las_dev_gemm_f32_valu(int on) { g_f32_valu = on != 0; }
#endif
static bool g_f32_fast_ld = true;    // the exact-fp32 kernels' branch-free tile loader (tile_gload_f32fast)
#ifdef LAS_DEV   // development builds only (make prof): A/B switch, not part of the shipping library
extern "C" void las_dev_gemm_f32_fast_ld(int on) { g_f32_fast_ld = on != 0; }
#endif
static bool g_zgroup_on = true;
#ifdef LAS_DEV   // development builds only (make prof): A/B switch, not part of the shipping library
extern "C" void las_dev_gemm_zgroup(int on) { g_zgroup_on = on != 0; }     // development switch (A/B measurements)
#endif

template <int WM, int WN, int TM, int TN, typename TI>
static void launch_fast_t(const GemmArgs& g, int zdim, hipStream_t st) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const int nx = cdiv(g.N, BN), ny = cdiv(g.M, BM);
    const bool xcd_order = ny >= 64 && zdim == 1;        // tall output: 1-D grid, row blocks padded to 8, XCD-aware order
    dim3 grid(xcd_order ? nx * ((ny + 7) / 8 * 8) : nx, xcd_order ? 1 : ny, zdim), blk(WM * WN * 64);
    GemmArgs gz = g;
    gz.zgroup = 0;
    if (zdim > 1 && nx * ny <= 64 && g_zgroup_on) {      // few tiles per k-chunk / batch entry: group them per XCD (see the kernel)
        gz.zgroup = zdim;
        grid = dim3(nx * ny * ((zdim + 7) / 8 * 8), 1, 1);
    }
    const GemmArgs& g_ = gz;
    const bool akc = g.ksA == 1, bkc = g.ksB == 1;
    if (akc && bkc)       hipLaunchKernelGGL((gemm_bf16_fast_kernel<WM, WN, TM, TN, true, true, TI>), grid, blk, 0, st, g_);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16_fast_kernel<WM, WN, TM, TN, true, false, TI>), grid, blk, 0, st, g_);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf16_fast_kernel<WM, WN, TM, TN, false, true, TI>), grid, blk, 0, st, g_);
    else                  hipLaunchKernelGGL((gemm_bf16_fast_kernel<WM, WN, TM, TN, false, false, TI>), grid, blk, 0, st, g_);
}
template <int WM, int WN, int TM, int TN>
static void launch_fast(const GemmArgs& g, int zdim, hipStream_t st) {
    if (g.in_bf16) launch_fast_t<WM, WN, TM, TN, unsigned short>(g, zdim, st);
    else           launch_fast_t<WM, WN, TM, TN, float>(g, zdim, st);
}

// ------------------------------------------------------------------------------------------------
// exact-fp32 MFMA kernel (parity mode, round 4).  v_mfma_f32_16x16x4_f32 is a k-ordered fp32 fma chain -- one rounding per
// product, bitwise what a per-thread v_fmac loop gives -- at the VALU's peak rate but from one operand register per lane and
// with the VALU free for addressing: the 64x64x16 VALU tile loop below it replaced ran at ~10 % of that peak.  Same strided
// operand description as the other kernels (NN / NT / TN, batching, contraction mask, deterministic split-K, fused bias /
// beta / tanh epilogue), the guarded generic loader, register double buffering: the next k-tile's global loads fly under the
// 32 x TM x TN MFMAs of the current one.  LDS tiles are k-major ([32][BM + 20] fp32: a lane's A operand is row k0 + (lane >> 4),
// column m0 + (lane & 15)).
// ------------------------------------------------------------------------------------------------
// Pitch ROWS + 20 (20 mod 64 banks): the k-contiguous operand's staging stores -- four scalar stores per 16-byte chunk, rows (c >> 3),
// k (c & 7) * 4 + e -- hit 32 banks per wave instead of 8 (pitch ROWS + 16 made the reads conflict-free but these stores 8-way
// conflicted: 1024 LDS cycles per k-tile and workgroup next to 4096 cycles of MFMA); the operand reads stay within 1.25-way.
template <int ROWS>
struct F32Lds { static constexpr int PITCH = ROWS + 20; };

template <int ROWS, int NT>
__device__ __forceinline__ void tile_sstore_f32(float* S, const TileRegs<ROWS, NT>& r, long long ks) {
    constexpr int NCH = ROWS * 8, RQ = ROWS / 4, LP = F32Lds<ROWS>::PITCH;
#pragma unroll
    for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < NCH) {
            const float4 v = r.v[i];
            if (ks == 1) {                           // the chunk runs along k: four rows of the k-major tile
                const int row = c >> 3, kq = (c & 7) * 4;
                S[(kq + 0) * LP + row] = v.x; S[(kq + 1) * LP + row] = v.y; S[(kq + 2) * LP + row] = v.z; S[(kq + 3) * LP + row] = v.w;
            } else {                                 // the chunk runs along the row index: one 16-byte store
                const int k = c / RQ, rq = (c % RQ) * 4;
                *reinterpret_cast<float4*>(&S[k * LP + rq]) = v;
            }
        }
    }
}

// Branch-free form of tile_gload for the exact-fp32 kernels (same register layout, same zero fill, so the arithmetic is unchanged):
// 16-byte loads legal, whole chunks inside or outside the operand (K % 4 == 0 for a k-contiguous operand, R % 4 == 0 for a row-
// contiguous one), no contraction mask.  The generic loader's per-chunk branches each end in s_waitcnt vmcnt(0) -- a k-tile's eight
// chunk loads were eight round trips in a row, 44-57 % of the fp32 matrix peak (round 4's r4_f32_gemm_shapes.txt); here they are
// all in flight before the first is needed.  KC: contraction index contiguous in memory.
template <int ROWS, int NT, bool KC>
__device__ __forceinline__ void tile_gload_f32fast(TileRegs<ROWS, NT>& r, const float* __restrict__ X, long long rs, long long ks,
                                                   int row0, int k0, int R, int Kend) {
    constexpr int NCH = ROWS * 8, RQ = ROWS / 4;
#pragma unroll
    for (int i = 0; i < (NCH + NT - 1) / NT; ++i) {
        const int c = threadIdx.x + i * NT;
        if (NCH % NT != 0 && c >= NCH) { r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        if (KC) {
            const int row = c >> 3, kq = (c & 7) * 4;
            const int gr = row0 + row, gk = k0 + kq;
            const bool on = gr < R && gk < Kend;
            ld4(r.v[i], X + (on ? (long long)gr * rs + gk : 0ll), on);
        } else {
            const int k = c / RQ, rq = (c % RQ) * 4;
            const int gr = row0 + rq, gk = k0 + k;
            const bool on = gk < Kend && gr < R;
            ld4(r.v[i], X + (on ? (long long)gk * ks + gr : 0ll), on);
        }
    }
}
#ifndef LAS_SKINNY_KSLICE
#define LAS_SKINNY_KSLICE 256     // the skinny fp32 product's K slices: K / this many, capped by 512 / column blocks
#endif
#ifndef LAS_MF32_PIPE
#define LAS_MF32_PIPE 1
#endif
template <int ROWS, int NT, int LD, bool IS_A>
__device__ __forceinline__ void tile_gload_f32(TileRegs<ROWS, NT>& r, const float* __restrict__ X, long long rs, long long ks,
                                               int row0, int k0, int R, int Kend, int vec, int mperiod, int mskip) {
    if (LD == 0) tile_gload<ROWS, NT>(r, X, rs, ks, row0, k0, R, Kend, vec, mperiod, mskip);
    else if (((LD - 1) >> (IS_A ? 0 : 1)) & 1) tile_gload_f32fast<ROWS, NT, true>(r, X, rs, ks, row0, k0, R, Kend);
    else tile_gload_f32fast<ROWS, NT, false>(r, X, rs, ks, row0, k0, R, Kend);
}

// LD: 0 = generic loader; 1 + (A k-contiguous) + 2 (B k-contiguous) = branch-free loader (las_gemm_dt checks the conditions)
template <int TM, int TN, int LD = 0>
__global__ __launch_bounds__(256) void gemm_mf32_kernel(GemmArgs g) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16, NT = 256, PA = F32Lds<BM>::PITCH, PB = F32Lds<BN>::PITCH;
    __shared__ __attribute__((aligned(16))) float lds[32 * (PA + PB)];
    float* As = lds;
    float* Bs = lds + 32 * PA;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = g.A;
    const float* B = g.B;
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = blockIdx.z * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)blockIdx.z * g.strideA;
        B += (long long)blockIdx.z * g.strideB;
        C += (long long)blockIdx.z * g.strideC;
    }
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    TileRegs<BM, NT> ra;
    TileRegs<BN, NT> rb;
    if (kbeg < kend) {
        tile_gload_f32<BM, NT, LD, true>(ra, A, g.rsA, g.ksA, m0, kbeg, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
        tile_gload_f32<BN, NT, LD, false>(rb, B, g.rsB, g.ksB, n0, kbeg, g.N, kend, g.vecB, 0, 0);
    }
#ifndef LAS_MF32_ABL
#define LAS_MF32_ABL 0       // timing experiments: 1 = no global loads after the first tile, 2 = no LDS stores / barriers after the first tile
#endif
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        if (!(LAS_MF32_ABL & 2) || k0 == kbeg) {
        __syncthreads();
        tile_sstore_f32<BM, NT>(As, ra, g.ksA);
        tile_sstore_f32<BN, NT>(Bs, rb, g.ksB);
        __syncthreads();
        }
        if (k0 + 32 < kend && !(LAS_MF32_ABL & 1)) {
            tile_gload_f32<BM, NT, LD, true>(ra, A, g.rsA, g.ksA, m0, k0 + 32, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
            tile_gload_f32<BN, NT, LD, false>(rb, B, g.rsB, g.ksB, n0, k0 + 32, g.N, kend, g.vecB, 0, 0);
        }
        const float* ap = As + lk * PA + wm * TM * 16 + li;
        const float* bp = Bs + lk * PB + wn * TN * 16 + li;
#if LAS_MF32_PIPE
        // the operands of k-step ks + 1 are read while the MFMAs of k-step ks run (two register sets)
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 16];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 16];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {             // (k beyond kend was loaded as zeros: whole k-steps always)
            if (ks + 1 < 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(ks + 1) & 1][i] = ap[(ks + 1) * 4 * PA + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(ks + 1) & 1][j] = bp[(ks + 1) * 4 * PB + j * 16];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0);   // (C^T tiles: see the epilogue)
            __builtin_amdgcn_sched_barrier(0);       // keep the next k-step's reads in front of this k-step's MFMAs
        }
#else
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {             // (k beyond kend was loaded as zeros: whole k-steps always)
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ap[ks * 4 * PA + i * 16];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bp[ks * 4 * PB + j * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
#endif
    }
    // epilogue.  The products above are issued with the operands swapped (B fragment first): a tile comes out TRANSPOSED, lane (li, lk)
    // holds row li and the four CONSECUTIVE columns 4 lk .. 4 lk + 3 -- one 16-byte store per tile and lane instead of four 4-byte stores
    // to four rows (round 5; every dot product is the same k-ordered fma chain: bit-identical)
    const bool cvec = (g.ldc % 4) == 0 && (((uintptr_t)C) & 15) == 0 && (g.N % 4) == 0;
    const bool pvec = (g.N % 4) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = m0 + (wm * TM + i) * 16 + li, col0 = n0 + (wn * TN + j) * 16 + lk * 4;
            if (row >= g.M || col0 >= g.N) continue;
            if (g.splitk > 1) {
                float* pp = g.partial + ((long long)blockIdx.z * g.M + row) * g.N + col0;
                if (pvec) *reinterpret_cast<float4*>(pp) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col0 + r < g.N) pp[r] = acc[i][j][r];
                }
                continue;
            }
            float* cp = C + (long long)row * g.ldc + col0;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = g.alpha * acc[i][j][r];
                if (g.bias && col0 + r < g.N) v[r] += g.bias[col0 + r];
            }
            if (cvec) {
                if (g.beta != 0.f) {
                    const float4 c4 = *reinterpret_cast<const float4*>(cp);
                    v[0] += g.beta * c4.x; v[1] += g.beta * c4.y; v[2] += g.beta * c4.z; v[3] += g.beta * c4.w;
                }
                *reinterpret_cast<float4*>(cp) = make_float4(apply_act(v[0], g.act), apply_act(v[1], g.act), apply_act(v[2], g.act), apply_act(v[3], g.act));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (col0 + r < g.N) {
                        if (g.beta != 0.f) v[r] += g.beta * cp[r];
                        cp[r] = apply_act(v[r], g.act);
                    }
                }
            }
        }
}

// Skinny form, M <= 64 (the parity mode's per-decode-step products: [B rows] x [1152] x [2048] and its transpose, 382 of them per
// B = 48 train step -- 129 us each through the 64 x 64 tile above, 50 of the parity step's 97 ms: 32 workgroups, each walking
// K in 32-wide stages whose loads it waits for).  Here a workgroup owns 64 x 32 outputs and a stage is 96 k: nine 16-byte
// loads per thread in flight, the four waves split the stage's k-steps (6 each), their partial tiles meet in LDS in fixed order;
// with scratch the contraction is also cut into K slices over blockIdx.z (las_gemm_dt), reduced in fixed order afterwards.
template <int TMS, int LD = 0>
__global__ __launch_bounds__(256) void gemm_mf32_skinny_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 32, NQ = 3, BK = 32 * NQ, NT = 256, PA = F32Lds<BM>::PITCH, PB = F32Lds<BN>::PITCH;
    __shared__ __attribute__((aligned(16))) float lds[BK * (PA + PB)];          // 51 KB
    float* As = lds;
    float* Bs = lds + BK * PA;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = g.A;
    const float* B = g.B;
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {                          // K slices over blockIdx.z: 64 column blocks alone leave three quarters of the chip idle
        kbeg = blockIdx.z * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)blockIdx.z * g.strideA;
        B += (long long)blockIdx.z * g.strideB;
        C += (long long)blockIdx.z * g.strideC;
    }
    f32x4_t acc[TMS][2];
#pragma unroll
    for (int i = 0; i < TMS; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    TileRegs<BM, NT> ra[NQ];
    TileRegs<BN, NT> rb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        tile_gload_f32<BM, NT, LD, true>(ra[q], A, g.rsA, g.ksA, m0, kbeg + 32 * q, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
        tile_gload_f32<BN, NT, LD, false>(rb[q], B, g.rsB, g.ksB, n0, kbeg + 32 * q, g.N, kend, g.vecB, 0, 0);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            tile_sstore_f32<BM, NT>(As + q * 32 * PA, ra[q], g.ksA);
            tile_sstore_f32<BN, NT>(Bs + q * 32 * PB, rb[q], g.ksB);
        }
        __syncthreads();
        if (k0 + BK < kend) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                tile_gload_f32<BM, NT, LD, true>(ra[q], A, g.rsA, g.ksA, m0, k0 + BK + 32 * q, g.M, kend, g.vecA, g.mask_period, g.mask_skip);
                tile_gload_f32<BN, NT, LD, false>(rb[q], B, g.rsB, g.ksB, n0, k0 + BK + 32 * q, g.N, kend, g.vecB, 0, 0);
            }
        }
        const float* ap = As + (w * (BK / 4) + lk) * PA + li;
        const float* bp = Bs + (w * (BK / 4) + lk) * PB + li;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            float a[TMS], b[2];
#pragma unroll
            for (int i = 0; i < TMS; ++i) a[i] = ap[ks * 4 * PA + i * 16];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = bp[ks * 4 * PB + j * 16];
#pragma unroll
            for (int i = 0; i < TMS; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = lds;                                                            // [wave][tile i][tile j][16 x 16]
#pragma unroll
    for (int i = 0; i < TMS; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w * TMS + i) * 2 + j) * 256 + (lk * 4 + r) * 16 + li] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < TMS * 2; ++e) {
        const int idx = tid + 256 * e, rl = idx >> 5, cl = idx & 31;             // 16 TMS rows x 32 columns
        const int o = ((rl >> 4) * 2 + (cl >> 4)) * 256 + (rl & 15) * 16 + (cl & 15);
        const float sum = ((red[o] + red[TMS * 512 + o]) + red[2 * TMS * 512 + o]) + red[3 * TMS * 512 + o];
        const int row = m0 + rl, col = n0 + cl;
        if (row < g.M && col < g.N && g.splitk > 1) {
            g.partial[((long long)blockIdx.z * g.M + row) * g.N + col] = sum;
        } else if (row < g.M && col < g.N) {
            float v = g.alpha * sum;
            if (g.bias) v += g.bias[col];
            float* cp = C + (long long)row * g.ldc + col;
            if (g.beta != 0.f) v += g.beta * (*cp);
            *cp = apply_act(v, g.act);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 VALU kernel (parity mode)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = g.A;
    const float* B = g.B;
    float* C = g.C;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = blockIdx.z * g.kchunk;
        kend = min(g.K, kbeg + g.kchunk);
    } else {
        A += (long long)blockIdx.z * g.strideA;
        B += (long long)blockIdx.z * g.strideB;
        C += (long long)blockIdx.z * g.strideC;
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (g.ksA == 1) { k = tid & 15; m = (tid >> 4) + 16 * i; }
            else            { m = tid & 63; k = (tid >> 6) + 4 * i; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < kend && !(g.mask_period > 0 && (gk % g.mask_period) == g.mask_skip))
                v = A[(long long)gm * g.rsA + (long long)gk * g.ksA];
            As[k][m] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int n, k;
            if (g.ksB == 1) { k = tid & 15; n = (tid >> 4) + 16 * i; }
            else            { n = tid & 63; k = (tid >> 6) + 4 * i; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.f;
            if (gn < g.N && gk < kend) v = B[(long long)gn * g.rsB + (long long)gk * g.ksB];
            Bs[k][n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = m0 + ty * 4 + i, col = n0 + tx * 4 + j;
            if (row < g.M && col < g.N) {
                if (g.splitk > 1) {
                    g.partial[((long long)blockIdx.z * g.M + row) * g.N + col] = acc[i][j];
                } else {
                    float v = g.alpha * acc[i][j];
                    if (g.bias) v += g.bias[col];
                    float* cp = C + (long long)row * g.ldc + col;
                    if (g.beta != 0.f) v += g.beta * (*cp);
                    *cp = apply_act(v, g.act);
                }
            }
        }
}

// fixed-order reduction of the split-K partials + epilogue
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
    const long long total = (long long)g.M * g.N;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int row = (int)(idx / g.N), col = (int)(idx % g.N);
        float s = 0.f;
        for (int z = 0; z < g.splitk; ++z) s += g.partial[(long long)z * total + idx];
        float v = g.alpha * s;
        if (g.bias) v += g.bias[col];
        float* cp = g.C + (long long)row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * (*cp);
        *cp = apply_act(v, g.act);
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_bf16(const GemmArgs& g, int zdim, hipStream_t st) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), zdim);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN>), grid, dim3(WM * WN * 64), 0, st, g);
    return 0;
}

template <int TM, int TN>
static void launch_mf32(const GemmArgs& g, int ld, dim3 grid, hipStream_t st) {
    switch (ld) {
        case 1: hipLaunchKernelGGL((gemm_mf32_kernel<TM, TN, 1>), grid, dim3(256), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_mf32_kernel<TM, TN, 2>), grid, dim3(256), 0, st, g); break;
        case 3: hipLaunchKernelGGL((gemm_mf32_kernel<TM, TN, 3>), grid, dim3(256), 0, st, g); break;
        case 4: hipLaunchKernelGGL((gemm_mf32_kernel<TM, TN, 4>), grid, dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_mf32_kernel<TM, TN, 0>), grid, dim3(256), 0, st, g); break;
    }
}
template <int TMS>
static void launch_mf32_skinny_t(const GemmArgs& g, int ld, dim3 grid, hipStream_t st) {
    switch (ld) {
        case 1: hipLaunchKernelGGL((gemm_mf32_skinny_kernel<TMS, 1>), grid, dim3(256), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_mf32_skinny_kernel<TMS, 2>), grid, dim3(256), 0, st, g); break;
        case 3: hipLaunchKernelGGL((gemm_mf32_skinny_kernel<TMS, 3>), grid, dim3(256), 0, st, g); break;
        case 4: hipLaunchKernelGGL((gemm_mf32_skinny_kernel<TMS, 4>), grid, dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_mf32_skinny_kernel<TMS, 0>), grid, dim3(256), 0, st, g); break;
    }
}
static void launch_mf32_skinny(const GemmArgs& g, int ld, int row_tiles, dim3 grid, hipStream_t st) {
    switch (row_tiles) {
        case 1: launch_mf32_skinny_t<1>(g, ld, grid, st); break;
        case 2: launch_mf32_skinny_t<2>(g, ld, grid, st); break;
        case 3: launch_mf32_skinny_t<3>(g, ld, grid, st); break;
        default: launch_mf32_skinny_t<4>(g, ld, grid, st); break;
    }
}

// Scratch the deterministic split-K of las_gemm would use for this product if it could have all it wants (never more than
// LAS_GEMM_WS_CAP: beyond that the split degree is cut to fit, with any workspace): tile counts as in las_gemm_dt below.
extern "C" size_t las_gemm_workspace_bytes(int prec, int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || batch != 1) return 0;
    const int b = (M < 128 || N < 128) ? 64 : 128;        // (both precisions; the bf16 48-row tile has one row block like the 64-row one)
    const long long tiles = (long long)cdiv(M, b) * cdiv(N, b);
    int s = 1;
    if (K >= 2048 && tiles < 256) {                        // the tall-contraction rule
        const int want = (int)((512 + tiles - 1) / tiles), maxs = K / 512;
        s = want < maxs ? want : maxs;
    }
    if (s <= 1 && prec == LAS_PREC_F32 && M <= 64 && K >= 512) {       // the parity mode's skinny products: K slices (las_gemm_dt)
        s = K / LAS_SKINNY_KSLICE;
        if (s > 512 / cdiv(N, 32)) s = 512 / cdiv(N, 32);
    }
    if (s <= 1) return 0;
    const size_t need = (size_t)s * M * N * sizeof(float);
    return need < LAS_GEMM_WS_CAP ? need : LAS_GEMM_WS_CAP;
}

extern "C" int las_gemm(int prec, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                        int lda, long long strideA, const float* B, int ldb, long long strideB, float beta,
                        float* C, int ldc, long long strideC, const float* bias, int act, int batch,
                        int a_mask_period, int a_mask_skip, void* ws, size_t ws_bytes, void* stream) {
    return las_gemm_dt(prec, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, LAS_DT_F32, beta, C, ldc, strideC, bias,
                       act, batch, a_mask_period, a_mask_skip, ws, ws_bytes, stream);
}

extern "C" int las_gemm_dt(int prec, int transA, int transB, int M, int N, int K, float alpha, const void* Av,
                           int lda, long long strideA, const void* Bv, int ldb, long long strideB, int in_dtype, float beta,
                           float* C, int ldc, long long strideC, const float* bias, int act, int batch,
                           int a_mask_period, int a_mask_skip, void* ws, size_t ws_bytes, void* stream) {
    const float* A = (const float*)Av;
    const float* B = (const float*)Bv;
    LAS_ARG(in_dtype == LAS_DT_F32 || (in_dtype == LAS_DT_BF16 && prec == LAS_PREC_BF16), "las_gemm: bf16 operands need LAS_PREC_BF16");
    LAS_ARG(prec == LAS_PREC_F32 || prec == LAS_PREC_BF16, "las_gemm: bad prec %d", prec);
    LAS_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "las_gemm: bad dims M=%d N=%d K=%d batch=%d", M, N, K, batch);
    LAS_ARG(A && B && C, "las_gemm: null operand");
    LAS_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "las_gemm: leading dimension too small");
    LAS_ARG(a_mask_period == 0 || transA == 1, "las_gemm: a_mask_period needs transA=1");
    LAS_ARG(act == LAS_ACT_NONE || act == LAS_ACT_TANH, "las_gemm: bad act %d", act);
    if (M == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
    g.A = A; g.B = B; g.C = C; g.ldc = ldc;
    g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
    g.rsA = transA ? 1 : lda;  g.ksA = transA ? lda : 1;
    g.rsB = transB ? ldb : 1;  g.ksB = transB ? 1 : ldb;
    g.bias = bias; g.act = act;
    g.mask_period = a_mask_period; g.mask_skip = a_mask_skip;
    const uintptr_t amask = in_dtype == LAS_DT_BF16 ? 7 : 15;       // 4-element chunks: 8 bytes of bf16 / 16 bytes of fp32
    g.vecA = ((lda % 4) == 0) && (((uintptr_t)A & amask) == 0) && ((strideA % 4) == 0);
    g.vecB = ((ldb % 4) == 0) && (((uintptr_t)B & amask) == 0) && ((strideB % 4) == 0);
    g.splitk = 1; g.kchunk = K; g.partial = nullptr;
    g.in_bf16 = in_dtype == LAS_DT_BF16;

    // tile configuration
    int BM, BN;
    int cfg;
    // branch-free fast path: 16-byte loads legal, k (or row) counts multiples of 4, no contraction mask
    const bool fastA = g.vecA && (g.ksA == 1 ? (K % 4 == 0) : (M % 4 == 0 && M >= 4));
    const bool fastB = g.vecB && (g.ksB == 1 ? (K % 4 == 0) : (N % 4 == 0 && N >= 4));
    const bool fast_ok = prec == LAS_PREC_BF16 && fastA && fastB && a_mask_period == 0 && K > 0;
    // exact-fp32 kernels: the same conditions select their branch-free loader (1 + A k-contiguous + 2 B k-contiguous; 0 = generic)
    const int f32_ld = (g_f32_fast_ld && fastA && fastB && a_mask_period == 0 && K > 0) ? 1 + (g.ksA == 1 ? 1 : 0) + (g.ksB == 1 ? 2 : 0) : 0;
    if (prec == LAS_PREC_F32) { cfg = 0; BM = BN = (M < 128 || N < 128) ? 64 : 128; }    // exact-fp32 MFMA tiles
    else if (M <= 48 && !(fast_ok && K >= 4096)) { cfg = 3; BM = 48; BN = 64; }   // tall contractions: 64-row fast tiles win
    else if (M < 128 || N < 128) { cfg = 2; BM = 64; BN = 64; }
    else                      { cfg = 1; BM = 128; BN = 128; }
    const long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN);

    // deterministic split-K for tall contractions that would leave most of the 256 CUs idle
    if (batch == 1 && ws && K >= 2048 && tiles < 256) {
        int want = (int)((512 + tiles - 1) / tiles);
        int maxs = K / 512;
        int s = want < maxs ? want : maxs;
        while (s > 1 && (size_t)s * M * N * sizeof(float) > ws_bytes) --s;
        if (s > 1) {
            int kchunk = ((K + s - 1) / s + 31) / 32 * 32;
            s = (K + kchunk - 1) / kchunk;
            g.splitk = s; g.kchunk = kchunk; g.partial = (float*)ws;
        }
    }
    // parity mode, skinny products (the Speller's per-step cell products): 64 / 36 column blocks of a latency-bound K walk -> K slices
    if (prec == LAS_PREC_F32 && !g_f32_valu && M <= 64 && batch == 1 && ws && g.splitk == 1 && K >= 512) {
        const int blocks = cdiv(N, 32);
        int s = K / LAS_SKINNY_KSLICE, want = 512 / blocks;
        if (s > want) s = want;
        while (s > 1 && (size_t)s * M * N * sizeof(float) > ws_bytes) --s;
        if (s > 1) {
            const int kchunk = ((K + s - 1) / s + 31) / 32 * 32;
            g.splitk = (K + kchunk - 1) / kchunk; g.kchunk = kchunk; g.partial = (float*)ws;
        }
    }
    const int zdim = g.splitk > 1 ? g.splitk : batch;

    if (K == 0 && g.splitk == 1) {
        // empty contraction: C = act(beta*C + bias); run the kernel with no k-tiles
    }
    LAS_ARG(!g.in_bf16 || (fast_ok && cfg != 3) || (fast_ok && M <= 48),
            "las_gemm: bf16 operands are served by the branch-free path only (aligned pitches, K / row counts multiples of 4, no mask)");
    if (g.in_bf16 && cfg == 3) { cfg = 2; BM = 64; BN = 64; }
    // weight-gradient form (both operands k-strided bf16, whole 128 x 128 tiles, 16-byte aligned rows): LDS-transposing kernel
    const bool tn_al = g_tn_tr_on && fast_ok && g.in_bf16 && g.ksA != 1 && g.ksB != 1 && N % 128 == 0 &&
                       lda % 8 == 0 && ldb % 8 == 0 && strideA % 8 == 0 && strideB % 8 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
    const bool tn_tr = tn_al && cfg == 1 && M % 128 == 0;
    const bool tn_tr64 = tn_al && cfg == 2 && M <= 64 && M % 8 == 0 && batch == 1;     // (a first layer's dW_ih; one row block)
    if (fast_ok && (cfg == 1 || cfg == 2)) {
        if (tn_tr64) {
            const int nx = N / 128;
            GemmArgs gz = g;
            gz.zgroup = 0;
            dim3 grid(nx, 1, zdim);
            if (zdim > 1 && nx <= 64 && g_zgroup_on) {
                gz.zgroup = zdim;
                grid = dim3(nx * ((zdim + 7) / 8 * 8), 1, 1);
            }
            hipLaunchKernelGGL(gemm_tn_tr_kernel<64>, grid, dim3(256), 0, st, gz);
        }
        else if (tn_tr) {
            const int nx = N / 128, ny = M / 128;
            GemmArgs gz = g;
            gz.zgroup = 0;
            dim3 grid(nx, ny, zdim);
            if (zdim > 1 && nx * ny <= 64 && g_zgroup_on) {
                gz.zgroup = zdim;
                grid = dim3(nx * ny * ((zdim + 7) / 8 * 8), 1, 1);
            }
            hipLaunchKernelGGL(gemm_tn_tr_kernel<128>, grid, dim3(256), 0, st, gz);
        }
        else if (cfg == 1) launch_fast<2, 2, 4, 4>(g, zdim, st);
        else          launch_fast<2, 2, 2, 2>(g, zdim, st);
        LAS_LAUNCHED();
        if (g.splitk > 1) {
            int nb = cdiv((long long)M * N, 256);
            if (nb > 2048) nb = 2048;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, g);
            LAS_LAUNCHED();
        }
        return 0;
    }
    switch (cfg) {
        case 0: {
            dim3 grid(cdiv(N, BN), cdiv(M, BM), zdim);
            if (g_f32_valu) hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(N, 64), cdiv(M, 64), zdim), dim3(256), 0, st, g);
            else if (M <= 64) launch_mf32_skinny(g, f32_ld, cdiv(M, 16), dim3(cdiv(N, 32), 1, zdim), st);
            else if (BM == 128) launch_mf32<4, 4>(g, f32_ld, grid, st);
            else launch_mf32<2, 2>(g, f32_ld, grid, st);
        } break;
        case 1: launch_bf16<2, 2, 4, 4>(g, zdim, st); break;
        case 2: launch_bf16<2, 2, 2, 2>(g, zdim, st); break;
        default: launch_bf16<1, 4, 3, 1>(g, zdim, st); break;
    }
    LAS_LAUNCHED();
    if (g.splitk > 1) {
        int nb = cdiv((long long)M * N, 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, g);
        LAS_LAUNCHED();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// column sums (BiasAdd gradient), deterministic two-stage
// ------------------------------------------------------------------------------------------------
constexpr int CS_SPLITS = 64;

template <typename T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_f(T* p, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

// bf16 input with 16-byte alignment: a thread owns 8 adjacent columns (one 16-byte load per row), a block 512 columns x 4 row lanes.
// direct (nsplit == 1): out = beta * out + sum, no second pass.
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const unsigned short* __restrict__ X, int rows, int cols, int ldx,
                                                            float* __restrict__ part, int nsplit, float beta, float* __restrict__ out) {
    const int c8 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int rl = threadIdx.x >> 6;
    const int per = (rows + nsplit - 1) / nsplit;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c8 < cols)
        for (int r = r0 + rl; r < r1; r += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(X + (long long)r * ldx + c8);
            s[0] += __uint_as_float(v.x << 16); s[1] += __uint_as_float(v.x & 0xffff0000u);
            s[2] += __uint_as_float(v.y << 16); s[3] += __uint_as_float(v.y & 0xffff0000u);
            s[4] += __uint_as_float(v.z << 16); s[5] += __uint_as_float(v.z & 0xffff0000u);
            s[6] += __uint_as_float(v.w << 16); s[7] += __uint_as_float(v.w & 0xffff0000u);
        }
    __shared__ float red[4][64][9];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][threadIdx.x & 63][e] = s[e];
    __syncthreads();
    if (rl == 0 && c8 < cols) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = red[0][threadIdx.x][e] + red[1][threadIdx.x][e] + red[2][threadIdx.x][e] + red[3][threadIdx.x][e];
            if (nsplit == 1) out[c8 + e] = (beta != 0.f ? beta * out[c8 + e] : 0.f) + v;
            else part[(long long)blockIdx.y * cols + c8 + e] = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, int rows, int cols, int ldx,
                                                             float* __restrict__ part, int nsplit) {
    // block: 64 columns x 4 row lanes; blockIdx.y = row split
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int per = (rows + nsplit - 1) / nsplit;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + rl; r < r1; r += 4) s += ld_f<T>(X + (long long)r * ldx + c);
    __shared__ float red[4][64];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols)
        part[(long long)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// rows < 512 (the per-utterance dW_hh partials: rows = batch): one pass, out = beta * out + column sum
template <typename T>
__global__ __launch_bounds__(256) void colsum_direct_kernel(const T* __restrict__ X, int rows, int cols, int ldx, float beta, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += ld_f<T>(X + (long long)r * ldx + c);
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int cols, int nsplit, float beta,
                                                           float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += part[(long long)z * cols + c];
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
}

extern "C" size_t las_colsum_workspace_bytes(int cols) { return (size_t)CS_SPLITS * cols * sizeof(float); }

extern "C" int las_colsum_dt(const void* X, int dtype, int rows, int cols, int ldx, float beta, float* out, void* ws,
                             size_t ws_bytes, void* stream) {
    LAS_ARG(X && out && rows >= 0 && cols > 0 && ldx >= cols, "las_colsum: bad arguments");
    LAS_ARG(dtype == LAS_DT_F32 || dtype == LAS_DT_BF16, "las_colsum: bad dtype %d", dtype);
    LAS_ARG(ws && ws_bytes >= las_colsum_workspace_bytes(cols), "las_colsum: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nsplit = rows >= CS_SPLITS * 8 ? CS_SPLITS : 1;
    if (nsplit == 1) {                 // few rows: one launch
        if (dtype == LAS_DT_BF16) hipLaunchKernelGGL(colsum_direct_kernel<unsigned short>, dim3(cdiv(cols, 256)), dim3(256), 0, st,
                                                      (const unsigned short*)X, rows, cols, ldx, beta, out);
        else hipLaunchKernelGGL(colsum_direct_kernel<float>, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const float*)X, rows, cols, ldx, beta, out);
        LAS_LAUNCHED();
        return 0;
    }
    if (dtype == LAS_DT_BF16 && cols % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)X & 15) == 0) {
        hipLaunchKernelGGL(colsum_bf16x8_kernel, dim3(cdiv(cols, 512), nsplit), dim3(256), 0, st, (const unsigned short*)X, rows, cols, ldx,
                           (float*)ws, nsplit, beta, out);
        LAS_LAUNCHED();
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const float*)ws, cols, nsplit, beta, out);
        LAS_LAUNCHED();
        return 0;
    }
    if (dtype == LAS_DT_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<unsigned short>, dim3(cdiv(cols, 64), nsplit), dim3(256), 0, st,
                           (const unsigned short*)X, rows, cols, ldx, (float*)ws, nsplit);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(cdiv(cols, 64), nsplit), dim3(256), 0, st, (const float*)X, rows, cols, ldx,
                           (float*)ws, nsplit);
    LAS_LAUNCHED();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const float*)ws, cols, nsplit,
                       beta, out);
    LAS_LAUNCHED();
    return 0;
}

extern "C" int las_colsum(const float* X, int rows, int cols, int ldx, float beta, float* out, void* ws,
                          size_t ws_bytes, void* stream) {
    return las_colsum_dt(X, LAS_DT_F32, rows, cols, ldx, beta, out, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// tanh gradient
// ------------------------------------------------------------------------------------------------
template <typename TY, typename TD, typename TX>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const TY* __restrict__ Y, int ldy, const TD* __restrict__ dY,
                                                       int lddy, TX* __restrict__ dX, int lddx, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / cols;
        const int c = (int)(idx % cols);
        const float y = ld_f<TY>(Y + r * ldy + c);
        st_f<TX>(dX + r * lddx + c, ld_f<TD>(dY + r * lddy + c) * (1.f - y * y));
    }
}

// dX = dY * (1 - Y*Y); each tensor fp32 or bf16 (y_dt / dy_dt / dx_dt)
extern "C" int las_tanh_bwd_dt(const void* Y, int y_dt, int ldy, const void* dY, int dy_dt, int lddy, void* dX, int dx_dt, int lddx,
                               int rows, int cols, void* stream) {
    LAS_ARG(Y && dY && dX && rows >= 0 && cols >= 0, "las_tanh_bwd: bad arguments");
    if (rows == 0 || cols == 0) return 0;
    int nb = cdiv((long long)rows * cols, 256);
    if (nb > 4096) nb = 4096;
    hipStream_t st = (hipStream_t)stream;
    typedef unsigned short u16;
#define TB(TY, TD, TX) hipLaunchKernelGGL((tanh_bwd_kernel<TY, TD, TX>), dim3(nb), dim3(256), 0, st, (const TY*)Y, ldy, (const TD*)dY, lddy, (TX*)dX, lddx, rows, cols)
    const int key = (y_dt == LAS_DT_BF16 ? 4 : 0) | (dy_dt == LAS_DT_BF16 ? 2 : 0) | (dx_dt == LAS_DT_BF16 ? 1 : 0);
    switch (key) {
        case 0: TB(float, float, float); break;
        case 1: TB(float, float, u16); break;
        case 2: TB(float, u16, float); break;
        case 3: TB(float, u16, u16); break;
        case 4: TB(u16, float, float); break;
        case 5: TB(u16, float, u16); break;
        case 6: TB(u16, u16, float); break;
        default: TB(u16, u16, u16); break;
    }
#undef TB
    LAS_LAUNCHED();
    return 0;
}

extern "C" int las_tanh_bwd(const float* Y, int ldy, const float* dY, int lddy, float* dX, int lddx, int rows,
                            int cols, void* stream) {
    return las_tanh_bwd_dt(Y, LAS_DT_F32, ldy, dY, LAS_DT_F32, lddy, dX, LAS_DT_F32, lddx, rows, cols, stream);
}

// ------------------------------------------------------------------------------------------------
// skinny-M contraction with pre-packed bf16 weights (see las_common.h)
// fragment (ct, ks): lane l holds B[ks*32 + 8*(l>>4) + e][ct*16 + (l&15)], e = 0..7   (zero past K / N)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skinny_pack_kernel(const float* __restrict__ W, int ldw, int K, int N, int transposed,
                                                          int KS, unsigned short* __restrict__ out, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long f = idx >> 9;
        const int ks = (int)(f % KS), ct = (int)(f / KS);
        const int k = ks * 32 + (lane >> 4) * 8 + e, n = ct * 16 + (lane & 15);
        float v = 0.f;
        if (k < K && n < N) v = transposed ? W[(long long)n * ldw + k] : W[(long long)k * ldw + n];
        out[idx] = f2bf(v);
    }
}

size_t las_skinny_pack_bytes(int K, int N) { return (size_t)cdiv(N, 16) * cdiv(K, 32) * 1024; }

int las_skinny_pack(const float* W, int ldw, int K, int N, int transposed, void* packed, hipStream_t st) {
    const int KS = cdiv(K, 32);
    const long long total = (long long)cdiv(N, 16) * KS * 512;
    int nb = cdiv(total, 256 * 8);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(skinny_pack_kernel, dim3(nb), dim3(256), 0, st, W, ldw, K, N, transposed, KS, (unsigned short*)packed, total);
    LAS_LAUNCHED();
    return 0;
}

bool las_skinny_ok(int M, int K, int N, int lda, const void* A) {
    return M >= 1 && M <= 1024 && (K % 8) == 0 && (lda % 4) == 0 && (((uintptr_t)A) & 15) == 0 && N >= 1;
}

// Grid (column tile, 16-row tile): every workgroup reads ONE 16-row slab of A (fp32, or bf16 written by the producer
// kernel) and one column tile of fragments -- 37+37 KB at K=1152 instead of 221+37 KB for the all-rows form above.
// 8 waves split K with every load in flight at once, then an LDS reduction of the 8 partial tiles.
// EPI (round 6, the wide Speller path's tanh cells): the epilogue is the BasicRNNCell itself -- h = tanh(acc + bias) -- written as fp32 to
// `eh` (the saved state the gradient loop reads) and as bf16 into up to two operand rows of the NEXT products (the layer above's [x ; h]
// row, the query projection's state row): the gate-math launch between two dependent products disappears.
struct SkinnyEpi { float* eh; int ldh; unsigned short* b0; int ld0; unsigned short* b1; int ld1; };
// EPI = 2 (round 6, the wide path's reverse loop with tanh cells): the columns [c0, c0 + D) of the product are one layer's share of d h, and the
// epilogue is that layer's gate gradient -- d h = the product's value, the recurrent gradient `sa` and the attention's `sb`, summed in the order of
// wide_cell_bwd_kernel (vlast: sa + sb + value, the top layer behind d s = dq . Ws^T; else value + sa + sb, a lower layer behind the layer
// above's product), times 1 - h^2 -> fp32 into the saved gates, bf16 into the operand row of the layer's own product.  The other columns are
// stored as usual.  The gate launch between two dependent products disappears.
struct SkinnyEpiB { const float* h; int ldh; const float* sa; int lda; const float* sb; int ldb; int vlast; int c0; int D; float* gp; int ldg;
                    unsigned short* gb; int ldgb;
                    // EPI = 3, the LSTM cell's gate gradient (wide_cell_bwd_kernel<LSTM>: gp holds the ACTIVATED gates [i | j | f | o] of the step and
                    // receives their pre-activation gradients; c / cp: the cell state after / before the step; dC: the carried cell gradient, updated)
                    const float* c; const float* cp; float* dC; int ldc_; };
struct SkinnyEpiNone {};
// NT (round 6): column tiles per workgroup -- 2 when one tile each would be more workgroups than the device has CUs (the LSTM-sized products of the
// wide path: 432 workgroups of 131 + 131 KB): the A slab is read once for both tiles and the grid fits the machine in one round.  Same sums.
template <bool ABF, int EPI = 0, class EpiT = SkinnyEpiNone, int NT = 1>
__global__ __launch_bounds__(512, 1) void skinny_rows_kernel(const void* __restrict__ Av, int lda, int M, int K,
                                                             const u16x8_t* __restrict__ Bp, int KS, int N,
                                                             float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                             int accumulate, EpiT ep = EpiT{}) {
    constexpr int NW = 8;
    __shared__ float red[NW][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ct0 = blockIdx.x * NT, mt = blockIdx.y, nct = (N + 15) >> 4;
    const int KSW = (KS + NW - 1) / NW;
    const int ks0 = w * KSW, ks1 = min(KS, ks0 + KSW);
    const u16x8_t* bp = Bp + (size_t)ct0 * KS * 64 + lane;
    int row = mt * 16 + c;
    if (row >= M) row = M - 1;                     // padded rows compute garbage that is never stored
    f32x4_t acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 8;
    for (int ks = ks0; ks < ks1; ks += UN) {
        u16x8_t bv[UN][NT], av[UN];
        float4 a0[UN], a1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = ks + u;
            const bool on = kk < ks1;
#pragma unroll
            for (int q = 0; q < NT; ++q)
                bv[u][q] = (on && ct0 + q < nct) ? bp[((size_t)q * KS + kk) * 64] : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            const bool ka = on && (kk * 32 + g * 8 + 8 <= K);
            if (ABF) {
                const unsigned short* ap = (const unsigned short*)Av + (long long)row * lda + g * 8;
                av[u] = ka ? *reinterpret_cast<const u16x8_t*>(ap + kk * 32) : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            } else {
                const float* ap = (const float*)Av + (long long)row * lda + g * 8;
                a0[u] = ka ? *reinterpret_cast<const float4*>(ap + kk * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                a1[u] = ka ? *reinterpret_cast<const float4*>(ap + kk * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ABF) {
                uint4 pk;
                pk.x = f2bf2(a0[u].x, a0[u].y); pk.y = f2bf2(a0[u].z, a0[u].w);
                pk.z = f2bf2(a1[u].x, a1[u].y); pk.w = f2bf2(a1[u].z, a1[u].w);
                av[u] = __builtin_bit_cast(u16x8_t, pk);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = mfma_bf16_16x16x32(av[u], bv[u][q], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][q][lane][r] = acc[q][r];
    __syncthreads();
    for (int i = tid; i < 256 * NT; i += 512) {
        const int q = i >> 8, t8 = i & 255;
        const int r16 = t8 >> 4, c16 = t8 & 15;
        const int l2 = (r16 >> 2) * 16 + c16, reg = r16 & 3;
        const int orow = mt * 16 + r16, col = (ct0 + q) * 16 + c16;
        if (orow < M && col < N) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += red[ww][q][l2][reg];
            if (bias) v += bias[col];
            bool done = false;
            if constexpr (EPI == 2) {
                const SkinnyEpiB& eb = ep;
                if (col >= eb.c0 && col < eb.c0 + eb.D) {
                    const int d = col - eb.c0;
                    float dh;
                    if (eb.vlast) { dh = eb.sa[(long long)orow * eb.lda + d]; if (eb.sb) dh += eb.sb[(long long)orow * eb.ldb + d]; dh += v; }
                    else { dh = v; if (eb.sa) dh += eb.sa[(long long)orow * eb.lda + d]; if (eb.sb) dh += eb.sb[(long long)orow * eb.ldb + d]; }
                    const float h = eb.h[(long long)orow * eb.ldh + d];
                    const float dp = dh * (1.f - h * h);
                    eb.gp[(long long)orow * eb.ldg + d] = dp;
                    eb.gb[(long long)orow * eb.ldgb + d] = f2bf(dp);
                    if (C) C[(long long)orow * ldc + col] = v;
                    done = true;
                }
            }
            if constexpr (EPI == 3) {
                const SkinnyEpiB& eb = ep;
                if (col >= eb.c0 && col < eb.c0 + eb.D) {
                    const int d = col - eb.c0, D = eb.D;
                    float dh;
                    if (eb.vlast) { dh = eb.sa[(long long)orow * eb.lda + d]; if (eb.sb) dh += eb.sb[(long long)orow * eb.ldb + d]; dh += v; }
                    else { dh = v; if (eb.sa) dh += eb.sa[(long long)orow * eb.lda + d]; if (eb.sb) dh += eb.sb[(long long)orow * eb.ldb + d]; }
                    float* gp = eb.gp + (long long)orow * eb.ldg;
                    const float gi = gp[d], gj = gp[D + d], gf = gp[2 * D + d], go = gp[3 * D + d];
                    const float cc = eb.c[(long long)orow * eb.ldc_ + d], cp = eb.cp[(long long)orow * eb.ldc_ + d];
                    const float tc = tanh_fast(cc);
                    float* dCr = eb.dC + (long long)orow * eb.ldc_;
                    const float dc = dCr[d] + dh * go * (1.f - tc * tc);
                    dCr[d] = dc * gf;
                    const float di = dc * gj * gi * (1.f - gi), dj = dc * gi * (1.f - gj * gj);
                    const float df = dc * cp * gf * (1.f - gf), dO = dh * tc * go * (1.f - go);
                    gp[d] = di; gp[D + d] = dj; gp[2 * D + d] = df; gp[3 * D + d] = dO;
                    unsigned short* gb = eb.gb + (long long)orow * eb.ldgb;
                    gb[d] = f2bf(di); gb[D + d] = f2bf(dj); gb[2 * D + d] = f2bf(df); gb[3 * D + d] = f2bf(dO);
                    if (C) C[(long long)orow * ldc + col] = v;
                    done = true;
                }
            }
            if constexpr (EPI == 1) {
                const SkinnyEpi& epi = ep;
                const float h = tanh_fast(v);
                epi.eh[(long long)orow * epi.ldh + col] = h;
                const unsigned short hb = f2bf(h);
                if (epi.b0) epi.b0[(long long)orow * epi.ld0 + col] = hb;
                if (epi.b1) epi.b1[(long long)orow * epi.ld1 + col] = hb;
                done = true;
            }
            if (!done) {
                if (accumulate) v += C[(long long)orow * ldc + col];
                C[(long long)orow * ldc + col] = v;
            }
        }
    }
}

// ---- the LSTM cell of the wide Speller path's forward chain as ONE launch (round 6): z = bf16(A) . packed + bias for the FOUR gate column tiles of
// 16 hidden units (TF's kernel layout [K, i | j | f | o]: column tile g D / 16 + ut of the same las_skinny_pack fragments -- no re-packing), then the
// gate math of wide_pointwise_fwd_kernel / wide_state_kernel: the activated gates (what the gradient loop reads), c, h in fp32 and h in bf16 into up to
// two operand rows of the products that read it next.  Grid (D / 16, 16-row tiles); 8 waves split K exactly as skinny_rows_kernel does and the partial
// tiles are summed in the same order, so the pre-activations are the bits of the product-then-gate-launch form.  The A slab is read once for four
// column tiles instead of four times (run.sh sizes, B = 48: 71 MB through the CUs per product instead of 113 MB), and the gate launch is gone.
struct SkinnyLstm { const float* bias; float fb; const float* cprev; float* c; float* h; float* gates; unsigned short* b0; int ld0; unsigned short* b1; int ld1; };
__global__ __launch_bounds__(512, 1) void skinny_lstm_kernel(const unsigned short* __restrict__ A, int lda, int M, int K, const u16x8_t* __restrict__ Bp,
                                                             int KS, int D, SkinnyLstm e) {
    constexpr int NW = 8;
    __shared__ float red[NW][4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int ut = blockIdx.x, mt = blockIdx.y, nut = D >> 4;
    const int KSW = (KS + NW - 1) / NW;
    const int ks0 = w * KSW, ks1 = min(KS, ks0 + KSW);
    int row = mt * 16 + c;
    if (row >= M) row = M - 1;                     // padded rows compute garbage that is never stored
    f32x4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 4;
    const unsigned short* ap = A + (long long)row * lda + g * 8;
    for (int ks = ks0; ks < ks1; ks += UN) {
        u16x8_t bv[UN][4], av[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = ks + u;
            const bool on = kk < ks1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[u][q] = on ? Bp[((size_t)(q * nut + ut) * KS + kk) * 64 + lane] : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            const bool ka = on && (kk * 32 + g * 8 + 8 <= K);
            av[u] = ka ? *reinterpret_cast<const u16x8_t*>(ap + kk * 32) : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma_bf16_16x16x32(av[u], bv[u][q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][q][lane][r] = acc[q][r];
    __syncthreads();
    if (tid < 256) {
        const int r16 = tid >> 4, c16 = tid & 15;
        const int l2 = (r16 >> 2) * 16 + c16, reg = r16 & 3;
        const int orow = mt * 16 + r16, d = ut * 16 + c16;
        if (orow < M) {
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) v += red[ww][q][l2][reg];
                if (e.bias) v += e.bias[q * D + d];
                z[q] = v;
            }
            const float gi = sigmoid_fast(z[0]), gj = tanh_fast(z[1]);
            const float gf = sigmoid_fast(z[2] + e.fb), go = sigmoid_fast(z[3]);
            const float cc = e.cprev[(long long)orow * D + d] * gf + gi * gj;
            const float h = tanh_fast(cc) * go;
            float* gp = e.gates + (long long)orow * 4 * D;
            gp[d] = gi; gp[D + d] = gj; gp[2 * D + d] = gf; gp[3 * D + d] = go;
            e.c[(long long)orow * D + d] = cc;
            e.h[(long long)orow * D + d] = h;
            const unsigned short hb = f2bf(h);
            if (e.b0) e.b0[(long long)orow * e.ld0 + d] = hb;
            if (e.b1) e.b1[(long long)orow * e.ld1 + d] = hb;
        }
    }
}
// A [M, K] bf16 (lda), packed = las_skinny_pack of the TF kernel [K, 4 D]; cprev / c / h [M, D], gates [M, 4 D] (activated i | j | f | o), b0 / b1 optional
int las_skinny_lstm_bf16(const unsigned short* A, int lda, int M, int K, const void* packed, int D, const float* bias, float fb, const float* cprev,
                         float* c, float* h, float* gates, unsigned short* b0, int ld0, unsigned short* b1, int ld1, hipStream_t st) {
    const int KS = cdiv(K, 32), MT = cdiv(M, 16);
    SkinnyLstm e{bias, fb, cprev, c, h, gates, b0, ld0, b1, ld1};
    hipLaunchKernelGGL(skinny_lstm_kernel, dim3(D / 16, MT), dim3(512), 0, st, A, lda, M, K, reinterpret_cast<const u16x8_t*>(packed), KS, D, e);
    LAS_LAUNCHED();
    return 0;
}

static int skinny_nt(int nct, int MT) { return ((long long)nct * MT > las_device_cus() && nct >= 2) ? 2 : 1; }     // column tiles per workgroup

int las_skinny_gemm(const float* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, const float* bias,
                    hipStream_t st) {
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    hipLaunchKernelGGL(skinny_rows_kernel<false>, dim3(nct, MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                       reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, bias, 0);
    LAS_LAUNCHED();
    return 0;
}

// A already rounded to bf16 by its producer (lda in elements, multiple of 8; A 16-byte aligned)
int las_skinny_gemm_bf16(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc,
                         const float* bias, hipStream_t st) {
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    if (skinny_nt(nct, MT) == 2)
        hipLaunchKernelGGL((skinny_rows_kernel<true, 0, SkinnyEpiNone, 2>), dim3(cdiv(nct, 2), MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                           reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, bias, 0);
    else
        hipLaunchKernelGGL(skinny_rows_kernel<true>, dim3(nct, MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                           reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, bias, 0);
    LAS_LAUNCHED();
    return 0;
}


// h = tanh(bf16(A) . packed + bias) -> eh [M, N] fp32 (ldh), and its bf16 copy into b0 / b1 (either may be null; element (r, c) at r ld + c)
int las_skinny_gemm_bf16_tanh(const unsigned short* A, int lda, int M, int K, const void* packed, int N, const float* bias, float* eh, int ldh,
                              unsigned short* b0, int ld0, unsigned short* b1, int ld1, hipStream_t st) {
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    SkinnyEpi e{eh, ldh, b0, ld0, b1, ld1};
    hipLaunchKernelGGL((skinny_rows_kernel<true, 1, SkinnyEpi>), dim3(nct, MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                       reinterpret_cast<const u16x8_t*>(packed), KS, N, (float*)nullptr, 0, bias, 0, e);
    LAS_LAUNCHED();
    return 0;
}

// C[M, N] = bf16(A) . packed (stored), and for the columns [c0, c0 + D) the tanh cell's gate gradient as the epilogue (SkinnyEpiB)
// ... and the LSTM cell's (SkinnyEpiB: EPI = 3): gp = the step's activated gates [M, 4 D] in / their pre-activation gradients out, gb = the bf16 operand row
// [M, 4 D], c / cp = the cell state after / before the step [M, D], dC = the carried cell-state gradient [M, D] (updated in place)
int las_skinny_gemm_bf16_lstm_bwd(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, int c0, int D,
                                  const float* sa, int lda_, const float* sb, int ldb, int vlast, float* gp, int ldg, unsigned short* gb, int ldgb,
                                  const float* c, const float* cp, float* dC, int ldc_, hipStream_t st) {
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    SkinnyEpiB e{nullptr, 0, sa, lda_, sb, ldb, vlast, c0, D, gp, ldg, gb, ldgb, c, cp, dC, ldc_};
    if (skinny_nt(nct, MT) == 2)
        hipLaunchKernelGGL((skinny_rows_kernel<true, 3, SkinnyEpiB, 2>), dim3(cdiv(nct, 2), MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                           reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, (const float*)nullptr, 0, e);
    else
    hipLaunchKernelGGL((skinny_rows_kernel<true, 3, SkinnyEpiB>), dim3(nct, MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                       reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, (const float*)nullptr, 0, e);
    LAS_LAUNCHED();
    return 0;
}

int las_skinny_gemm_bf16_tanh_bwd(const unsigned short* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, int c0, int D,
                                  const float* h, int ldh, const float* sa, int lda_, const float* sb, int ldb, int vlast, float* gp, int ldg,
                                  unsigned short* gb, int ldgb, hipStream_t st) {
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    SkinnyEpiB e{h, ldh, sa, lda_, sb, ldb, vlast, c0, D, gp, ldg, gb, ldgb, nullptr, nullptr, nullptr, 0};
    if (skinny_nt(nct, MT) == 2)
        hipLaunchKernelGGL((skinny_rows_kernel<true, 2, SkinnyEpiB, 2>), dim3(cdiv(nct, 2), MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                           reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, (const float*)nullptr, 0, e);
    else
    hipLaunchKernelGGL((skinny_rows_kernel<true, 2, SkinnyEpiB>), dim3(nct, MT), dim3(512), 0, st, (const void*)A, lda, M, K,
                       reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, (const float*)nullptr, 0, e);
    LAS_LAUNCHED();
    return 0;
}

// ---- C ABI of the skinny-M product (include/las_hip.h): weights packed once, then C[M,N] (+)= bf16(A[M,K]) . bf16(W) + bias per call
extern "C" size_t las_gemm_skinny_pack_bytes(int K, int N) { return K > 0 && N > 0 ? las_skinny_pack_bytes(K, N) : 0; }

extern "C" int las_gemm_skinny_pack(const float* W, int ldw, int K, int N, void* packed, void* stream) {
    LAS_ARG(W && packed && K > 0 && N > 0 && ldw >= N, "las_gemm_skinny_pack: bad arguments");
    LAS_ARG((K % 8) == 0, "las_gemm_skinny_pack: K must be a multiple of 8 (got %d)", K);
    return las_skinny_pack(W, ldw, K, N, 0, packed, (hipStream_t)stream);
}

extern "C" int las_gemm_skinny(const float* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, const float* bias,
                               int accumulate, void* stream) {
    LAS_ARG(A && packed && C && N > 0 && ldc >= N, "las_gemm_skinny: bad arguments");
    LAS_ARG(las_skinny_ok(M, K, N, lda, A), "las_gemm_skinny: needs 1 <= M <= 1024, K %% 8 == 0, lda %% 4 == 0, A 16-byte aligned");
    const int KS = cdiv(K, 32), nct = cdiv(N, 16), MT = cdiv(M, 16);
    hipLaunchKernelGGL(skinny_rows_kernel<false>, dim3(nct, MT), dim3(512), 0, (hipStream_t)stream, (const void*)A, lda, M, K,
                       reinterpret_cast<const u16x8_t*>(packed), KS, N, C, ldc, bias, accumulate ? 1 : 0);
    LAS_LAUNCHED();
    return 0;
}
