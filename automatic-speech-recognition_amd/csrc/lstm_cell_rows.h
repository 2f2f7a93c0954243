// lstm_cell_rows.h -- the device body of las_lstm_cell_rows (one LSTM cell step for a block of 32 rows x 16 units), shared by the kernels of
// loss_opt.hip (the cell alone, two cells as one grid) and speller.hip (round 5: the LM's first layer as extra workgroups of the beam
// search's attention-row launch).  reference lang/char_rnn_model.py:57-66 (BasicLSTMCell gate math), las/beam_search.py:109-116.
#pragma once
#include "las_common.h"
#include <type_traits>

constexpr int LC_KC = 512, LC_LD = LC_KC + 8;      // K chunk staged per pass; LDS row stride in bf16 (16-byte reads of 16 rows hit 64 distinct banks)

// FAST: the Speller's approximated transcendentals (its cell in a beam-search step, see las_common.h); XBF: x is already bf16
// PIPE: the software pipeline over the K chunks (190-200 VGPRs: one workgroup per CU); without it 92 VGPRs, two workgroups per CU --
// what the pair kernel wants: its two problems then run side by side on every CU and hide each other's round trips
template <bool FAST, bool XBF, bool PIPE>
__device__ __forceinline__ void lstm_cell_rows_body(const LstmCellLaunch& a, unsigned short* As, float (&gates)[2][4][64][4], const int ub,
                                                    const int row0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = lane >> 4, c = lane & 15;
    const int gt = w & 3, rt = w >> 2;
    const int H = a.H, ct = gt * (H >> 4) + ub;                      // this wave's column tile of the [K, 4H] kernel
    if (ub * 16 >= H || row0 >= a.M) return;                         // (the pair kernel's grid covers the larger of two problems)
    // operands of the gate math at the END of the launch, requested here: bias, the row's c_{t-1} and the index of its x row were three
    // dependent round trips behind the product (bias -> index -> x row + c), ~1.3 us of a 11-17 us launch (round 5)
    const int er_ = tid >> 4, eu_ = tid & 15, erow_ = min(row0 + er_, a.M - 1), eunit_ = ub * 16 + eu_;
    float pb_[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pb_[g] = a.bias[g * H + eunit_];
    const float pc_ = a.c_prev[(size_t)erow_ * H + eunit_];
    const int pid_ = a.xrows ? a.ids[erow_] : 0;
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (!PIPE) {
        for (int part = 0; part < 2; ++part) {
            const void* src = part ? (const void*)a.h : a.x;
            if (!src) continue;
            const bool sbf = XBF && part == 0;
            const int ld = part ? a.ldh : a.ldx, Kp = part ? H : a.I, KS = Kp >> 5;
            const u16x8_t* bp = reinterpret_cast<const u16x8_t*>(part ? a.Wh : a.Wx) + (size_t)ct * KS * 64 + lane;
            for (int k0 = 0; k0 < Kp; k0 += LC_KC) {
                const int kc = min(LC_KC, Kp - k0), nks = kc >> 5;
                u16x8_t bv[LC_KC / 32];                               // this chunk's weight fragments first (all in flight), then the rows
#pragma unroll
                for (int u = 0; u < LC_KC / 32; ++u) bv[u] = bp[(size_t)min((k0 >> 5) + u, KS - 1) * 64];
                __syncthreads();                                      // the previous chunk's readers are done
                // thread -> (row, 16-byte piece) with COMPILE-TIME pieces per row (a chunk shorter than LC_KC leaves lanes idle): the mapping
                // by `idx / pieces-of-this-chunk` was an integer division per load and per store -- 1,300 instructions per thread and chunk
                if (sbf) {
                    const int q = tid & 63, rb = tid >> 6;
                    if (q < (kc >> 3)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = rb + j * 8, row = min(row0 + r, a.M - 1);
                            *reinterpret_cast<uint4*>(As + r * LC_LD + q * 8) =
                                *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + (size_t)row * ld + k0 + q * 8);
                        }
                    }
                } else {
                    const int q = tid & 127, rb = tid >> 7;
                    if (q < (kc >> 2)) {
#pragma unroll 4                                                      // (four loads in flight: with all eight the pair kernel passes 128 VGPRs and its two problems no longer share a CU)
                        for (int j = 0; j < 8; ++j) {
                            const int r = rb + j * 4, row = min(row0 + r, a.M - 1);
                            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (size_t)row * ld + k0 + q * 4);
                            uint2 pk; pk.x = f2bf2(v.x, v.y); pk.y = f2bf2(v.z, v.w);
                            *reinterpret_cast<uint2*>(As + r * LC_LD + q * 4) = pk;
                        }
                    }
                }
                __syncthreads();
                const unsigned short* ar = As + (rt * 16 + c) * LC_LD + g4 * 8;
#pragma unroll
                for (int u = 0; u < LC_KC / 32; ++u)
                    if (u < nks) acc = mfma_bf16_16x16x32(*reinterpret_cast<const u16x8_t*>(ar + u * 32), bv[u], acc);
            }
        }
    } else {
        // K = [x ; h] in chunks of <= LC_KC that do not straddle the two parts.  Software pipeline: while chunk c is multiplied, the weight
        // fragments and the rows of chunk c + 1 are already on their way (one round trip to L2 / HBM per chunk was ~1.5 us of a 3-chunk call)
        const int KSx = a.x ? a.I >> 5 : 0, KSh = a.h ? H >> 5 : 0;
        const int ncx = (KSx + 15) >> 4, nch = ncx + ((KSh + 15) >> 4);
        u16x8_t bv[LC_KC / 32], bn[LC_KC / 32];
        float4 ra[8];                                                     // fp32 rows: 32 x 512 floats / 512 threads; bf16 rows use ra[0..3] as uint4
        auto chunk = [&](const int ci, const void*& src, int& ld, int& k0, int& kc, const u16x8_t*& bp, bool& sbf) {
            const bool hp = ci >= ncx;
            const int cj = hp ? ci - ncx : ci, Kp = hp ? H : a.I, KS = Kp >> 5;
            src = hp ? (const void*)a.h : a.x; ld = hp ? a.ldh : a.ldx; k0 = cj * LC_KC; kc = min(LC_KC, Kp - k0);
            bp = reinterpret_cast<const u16x8_t*>(hp ? a.Wh : a.Wx) + ((size_t)ct * KS + (k0 >> 5)) * 64 + lane;
            sbf = XBF && !hp;
        };
        auto load_b = [&](const int ci, u16x8_t (&dst)[LC_KC / 32]) {
            const void* src; int ld, k0, kc; const u16x8_t* bp; bool sbf;
            chunk(ci, src, ld, k0, kc, bp, sbf);
            const int nks = kc >> 5;
    #pragma unroll
            for (int u = 0; u < LC_KC / 32; ++u) dst[u] = bp[(size_t)(u < nks ? u : nks - 1) * 64];
        };
        auto load_a = [&](const int ci) {
            const void* src; int ld, k0, kc; const u16x8_t* bp; bool sbf;
            chunk(ci, src, ld, k0, kc, bp, sbf);
            // thread -> (row, 16-byte piece), pieces per row fixed at compile time (see the unpipelined path); idle lanes of a short chunk
            // re-read piece 0 of their row
            if (sbf) {
                const int q0 = tid & 63, q = q0 < (kc >> 3) ? q0 : 0, rb = tid >> 6;
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = min(row0 + rb + j * 8, a.M - 1);
                    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + (size_t)row * ld + k0 + q * 8);
                    ra[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            } else {
                const int q0 = tid & 127, q = q0 < (kc >> 2) ? q0 : 0, rb = tid >> 7;
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = min(row0 + rb + j * 4, a.M - 1);
                    ra[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (size_t)row * ld + k0 + q * 4);
                }
            }
        };
        auto store_a = [&](const int ci) {
            const void* src; int ld, k0, kc; const u16x8_t* bp; bool sbf;
            chunk(ci, src, ld, k0, kc, bp, sbf);
            if (sbf) {
                const int q = tid & 63, rb = tid >> 6;
                if (q < (kc >> 3)) {
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<uint4*>(As + (rb + j * 8) * LC_LD + q * 8) =
                            make_uint4(__float_as_uint(ra[j].x), __float_as_uint(ra[j].y), __float_as_uint(ra[j].z), __float_as_uint(ra[j].w));
                }
            } else {
                const int q = tid & 127, rb = tid >> 7;
                if (q < (kc >> 2)) {
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        uint2 pk; pk.x = f2bf2(ra[j].x, ra[j].y); pk.y = f2bf2(ra[j].z, ra[j].w);
                        *reinterpret_cast<uint2*>(As + (rb + j * 4) * LC_LD + q * 4) = pk;
                    }
                }
            }
        };
        auto multiply = [&](const int ci) {
            const void* src; int ld, k0, kc; const u16x8_t* bp; bool sbf;
            chunk(ci, src, ld, k0, kc, bp, sbf);
            const int nks = kc >> 5;
            const unsigned short* ar = As + (rt * 16 + c) * LC_LD + g4 * 8;
    #pragma unroll
            for (int u = 0; u < LC_KC / 32; ++u)
                if (u < nks) acc = mfma_bf16_16x16x32(*reinterpret_cast<const u16x8_t*>(ar + u * 32), bv[u], acc);
        };
        load_b(0, bv);
        load_a(0);
        // every chunk but the last: its successor's loads are issued UNCONDITIONALLY (behind an `if (ci + 1 < nch)` the compiler ends the
        // branch in s_waitcnt vmcnt(0): the prefetch was complete before the first MFMA of the chunk it was meant to hide behind)
        for (int ci = 0; ci + 1 < nch; ++ci) {
            __syncthreads();                                              // the previous chunk's readers are done
            store_a(ci);
            load_b(ci + 1, bn);
            load_a(ci + 1);
            __syncthreads();
            multiply(ci);
    #pragma unroll
            for (int u = 0; u < LC_KC / 32; ++u) bv[u] = bn[u];
        }
        __syncthreads();
        store_a(nch - 1);
        __syncthreads();
        multiply(nch - 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gates[rt][gt][lane][r] = acc[r];
    __syncthreads();
    // gate math: thread = (row of the block, unit of the block); MFMA C layout: element (row r16, col u) sits in lane (r16 / 4) * 16 + u, register r16 % 4
    const int r = tid >> 4, u = tid & 15, row = row0 + r;
    if (row >= a.M) return;
    const int l2 = ((r & 15) >> 2) * 16 + u, reg = r & 3, unit = ub * 16 + u;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = gates[r >> 4][g][l2][reg] + pb_[g];
    if (a.xrows) {
        int id = pid_ - a.id_shift;
        if (id < 0) id = 0;
        const float* xr = a.xrows + (size_t)id * 4 * H + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += xr[g * H];
    }
    const float gi = sigm<FAST>(z[0]), gj = tanhx<FAST>(z[1]), gf = sigm<FAST>(z[2] + a.fb), go = sigm<FAST>(z[3]);
    const size_t o = (size_t)row * H + unit;
    const float cn = fmaf(pc_, gf, gi * gj);          // (spelled out: left to the compiler, which of the two products is fused depends on the surrounding code -- the 32-row and the 128-row body must agree bit for bit)
    a.c_out[o] = cn;
    a.h_out[o] = tanhx<FAST>(cn) * go;
    if (a.gates_out) {                                   // the activated gates, as the Speller's backward pass reads them
        float* gp = a.gates_out + (size_t)row * 4 * H + unit;
        gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
    }
}


// ------------------------------------------------------------------------------------------------
// Round 5: the same cell step for MANY rows (a beam search over 64 utterances x beam 16 = 1024 hypothesis rows: decode.py's default batch).
// The 32-row workgroups above are latency-bound -- three dependent load / stage / multiply phases per workgroup -- and at M = 1024 a launch
// is four ROUNDS of them on the 256 CUs: 37-39 us per cell launch, 116-180 TFLOP/s (r5 trace, profiles/r5_decode_b64_kernels.txt).  Here a
// workgroup owns 128 rows x 16 units, so M = 1024 is ONE round: 8 waves = 4 gates x 2 row halves, a wave multiplies four 16-row tiles
// against its gate's fragments (the B operand is loaded once for 4x the rows), K chunks of 256 through a 67 KB LDS tile, next chunk's rows
// and fragments in flight while the current one is multiplied.  Same arithmetic, element for element: bf16 operands, fp32 accumulation in
// k order, gate math of the body above.
// ------------------------------------------------------------------------------------------------
#ifndef LB_STAMP
#define LB_STAMP 0
#endif
#ifndef LB_ABL
#define LB_ABL 0          // timing experiments (make ablf F=loss_opt D=-DLB_ABL=..): 1 no MFMAs, 2 no row loads after the first chunk, 4 no fragment loads after the first, 8 LDS-only barriers
#endif
constexpr int LB_KC = 128, LB_LD = LB_KC + 8;
// ROWS per workgroup: 128, 64 or 32 (the launcher picks the largest that still gives the 256 CUs a workgroup each: 1024 rows -> 128, 512 -> 64,
// 256 -> 32).  LDS: two row tiles of ROWS x (128 + 8) bf16 (chunk c is multiplied from one while chunk c + 1 is written to the other: one
// barrier per chunk); the gate exchange [2][ROWS / 32][4][64][4] floats reuses the first
constexpr int lb_lds_bytes(int rows) { return 2 * rows * LB_LD * 2; }

// ABF: the rows (x and / or h) are bf16; otherwise fp32.  One element type per launch: a type switch inside the chunk loop is a branch
// around loads, and the compiler ends every such branch in s_waitcnt vmcnt(0) -- which is what the first version of this body did at the end
// of EVERY chunk's prefetch (r5 ISA: the "pipelined" loads of chunk c + 1 were complete before the first MFMA of chunk c; a chunk took load
// time + multiply time, 2.7 / 3.6 us per 256 columns of K).  Here every load of the loop is unconditional (chunks behind the last are the
// last one again, K steps and pieces behind a short chunk's end re-read its first), K moves in chunks of 128 through TWO register stages:
// the rows of chunk c + 2 are requested when chunk c's have been written to LDS, its weight fragments when chunk c has been multiplied.
template <bool FAST, bool ABF, int ROWS>
__device__ __forceinline__ void lstm_cell_rows_big_body(const LstmCellLaunch& a, unsigned char* smem, const int ub, const int row0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g4 = lane >> 4, c = lane & 15;
    static_assert(ROWS == 32 || ROWS == 64 || ROWS == 128, "rows per workgroup");
    constexpr int RTW = ROWS / 32, LB_TILE = ROWS * LB_LD;             // 16-row tiles per wave (8 waves = 4 gates x 2 row halves); bf16 elements of a staged tile
    const int gt = w & 3, rh = w >> 2;
    const int H = a.H, ct = gt * (H >> 4) + ub;
    if (ub * 16 >= H || row0 >= a.M) return;
#if LB_STAMP
    // timing build (make ablf F=loss_opt D=-DLB_STAMP=1; tools/bench_cell_rows.py STAMP=1): gates_out is a stamp buffer [workgroup][16] of
    // 100 MHz ticks -- 0 entry, 1 first loads issued, 2 + c chunk c multiplied (c < 10), 12 gates exchanged, 13 done
    long long* lbs_ = reinterpret_cast<long long*>(a.gates_out) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
#define LBS(k) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0 && a.gates_out) lbs_[k] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LBS(k)
#endif
    LBS(0);
    // operands of the gate math at the end of the launch, requested here (see lstm_cell_rows_body): the thread's unit is the same for its
    // RTW elements (rows tid / 16 + 32 k)
    const int eunit_ = ub * 16 + (tid & 15);
    float pb_[4], pc_[RTW];
    int pid_[RTW];
#pragma unroll
    for (int g = 0; g < 4; ++g) pb_[g] = a.bias[g * H + eunit_];
#pragma unroll
    for (int k = 0; k < RTW; ++k) {
        const int erow = min(row0 + ((tid + k * 512) >> 4), a.M - 1);
        pc_[k] = a.c_prev[(size_t)erow * H + eunit_];
        pid_[k] = a.xrows ? a.ids[erow] : 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);
    f32x4_t acc[RTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int KSx = a.x ? a.I >> 5 : 0, KSh = a.h ? H >> 5 : 0;
    constexpr int NKS = LB_KC / 32;                                       // k-steps per chunk
    const int ncx = (KSx + NKS - 1) / NKS, nch = ncx + (KSh + NKS - 1) / NKS;
    constexpr int NRA = ABF ? ROWS / 32 : ROWS / 16;                      // 16-byte pieces of a chunk's rows per thread
    constexpr int PPR = ABF ? LB_KC / 8 : LB_KC / 4;                      // 16-byte pieces per row and chunk
    constexpr int RPP = 512 / PPR;                                        // rows per pass of the 512 threads
    u16x8_t bq[2][NKS];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));       // (a native vector: HIP's uint4 struct stayed in scratch)
    u32x4_t ra[2][NRA];
    const int aq0 = tid & (PPR - 1), arb = tid / PPR;
    // chunk ci (clamped to the last one): source rows, leading dimension, first column, columns (0 behind the last chunk), weight fragments
    auto chunk = [&](const int ci_, const void*& src, int& ld, int& k0, int& kc, const u16x8_t*& bp) {
        const int ci = ci_ < nch ? ci_ : nch - 1;
        const bool hp = ci >= ncx;
        const int cj = hp ? ci - ncx : ci, Kp = hp ? H : a.I, KS = Kp >> 5;
        src = hp ? (const void*)a.h : a.x; ld = hp ? a.ldh : a.ldx; k0 = cj * LB_KC; kc = min(LB_KC, Kp - k0);
        bp = reinterpret_cast<const u16x8_t*>(hp ? a.Wh : a.Wx) + ((size_t)ct * KS + (k0 >> 5)) * 64 + lane;
        if (ci_ >= nch) kc = 0;
    };
    auto load_b = [&](auto SI, const int ci) {
        constexpr int S = decltype(SI)::value;
        const void* src; int ld, k0, kc; const u16x8_t* bp;
        chunk(ci, src, ld, k0, kc, bp);
        const int nks = kc >> 5;
#pragma unroll
        for (int u = 0; u < NKS; ++u) bq[S][u] = bp[(size_t)(u < nks ? u : 0) * 64];
    };
    auto load_a = [&](auto SI, const int ci) {
        constexpr int S = decltype(SI)::value;
        const void* src; int ld, k0, kc; const u16x8_t* bp;
        chunk(ci, src, ld, k0, kc, bp);
        const int q = aq0 < (ABF ? kc >> 3 : kc >> 2) ? aq0 : 0;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(src) + ((size_t)k0 * (ABF ? 2 : 4) + (size_t)q * 16);
        const size_t ldb = (size_t)ld * (ABF ? 2 : 4);
#pragma unroll
        for (int j = 0; j < NRA; ++j) {
            const int row = min(row0 + arb + j * RPP, a.M - 1);
            ra[S][j] = *reinterpret_cast<const u32x4_t*>(base + (size_t)row * ldb);
        }
    };
    auto store_a = [&](auto SI, const int ci) {
        constexpr int S = decltype(SI)::value;
        const void* src; int ld, k0, kc; const u16x8_t* bp;
        chunk(ci, src, ld, k0, kc, bp);
        if (aq0 < (ABF ? kc >> 3 : kc >> 2)) {
#pragma unroll
            for (int j = 0; j < NRA; ++j) {
                if (ABF) {
                    *reinterpret_cast<u32x4_t*>(As + S * LB_TILE + (arb + j * RPP) * LB_LD + aq0 * 8) = ra[S][j];
                } else {
                    uint2 pk;
                    pk.x = f2bf2(__uint_as_float(ra[S][j][0]), __uint_as_float(ra[S][j][1]));
                    pk.y = f2bf2(__uint_as_float(ra[S][j][2]), __uint_as_float(ra[S][j][3]));
                    *reinterpret_cast<uint2*>(As + S * LB_TILE + (arb + j * RPP) * LB_LD + aq0 * 4) = pk;
                }
            }
        }
    };
    auto multiply = [&](auto SI, const int ci) {
        constexpr int S = decltype(SI)::value;
        const void* src; int ld, k0, kc; const u16x8_t* bp;
        chunk(ci, src, ld, k0, kc, bp);
        const int nks = kc >> 5;
        const unsigned short* ar = As + S * LB_TILE + (rh * (ROWS / 2) + c) * LB_LD + g4 * 8;
#pragma unroll
        for (int u = 0; u < NKS; ++u) {
            if (u < nks && !(LB_ABL & 1)) {
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
                    acc[rt] = mfma_bf16_16x16x32(*reinterpret_cast<const u16x8_t*>(ar + rt * 16 * LB_LD + u * 32), bq[S][u], acc[rt]);
            }
        }
    };
    // chunk ci sits in tile S (written before the last barrier); this half multiplies it and writes chunk ci + 1 to the OTHER tile, whose
    // readers (chunk ci - 1) passed the last barrier too
    auto half = [&](auto SI, auto SN, const int ci) {
        store_a(SN, ci + 1);
        load_a(SN, ci + 3);
        multiply(SI, ci);
        load_b(SI, ci + 2);
        __syncthreads();
        if (ci < 10) LBS(2 + ci);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_a(S0{}, 0); load_b(S0{}, 0);
    load_a(S1{}, 1); load_b(S1{}, 1);
    LBS(1);
    store_a(S0{}, 0);
    load_a(S0{}, 2);
    __syncthreads();
    for (int ci = 0; ci < nch; ci += 2) {
        half(S0{}, S1{}, ci);
        half(S1{}, S0{}, ci + 1);                                     // (behind the last chunk: nothing stored, nothing multiplied)
    }
    __syncthreads();                                                  // every wave is done reading the row tile: its space becomes the gate exchange
    float* gx = reinterpret_cast<float*>(smem);                       // [rh 2][rt RTW][gate 4][lane 64][4]
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) *reinterpret_cast<f32x4_t*>(gx + ((((rh * RTW + rt) * 4 + gt) * 64 + lane) << 2)) = acc[rt];
    __syncthreads();
    LBS(12);
    // gate math: element (row r of the ROWS, unit u of the 16); MFMA C layout: (row r16, col u) of a tile sits in lane (r16 / 4) * 16 + u, register r16 % 4
#pragma unroll
    for (int k = 0; k < RTW; ++k) {
        const int idx = tid + k * 512, r = idx >> 4, u = idx & 15, row = row0 + r;
        if (row >= a.M) continue;
        const int r16 = r & 15, l2 = (r16 >> 2) * 16 + u, reg = r16 & 3, unit = ub * 16 + u, tile = (r / (ROWS / 2)) * RTW + ((r >> 4) % RTW);
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = gx[(((tile * 4 + g) * 64 + l2) << 2) + reg] + pb_[g];
        if (a.xrows) {
            int id = pid_[k] - a.id_shift;
            if (id < 0) id = 0;
            const float* xr = a.xrows + (size_t)id * 4 * H + unit;
#pragma unroll
            for (int g = 0; g < 4; ++g) z[g] += xr[g * H];
        }
        const float gi = sigm<FAST>(z[0]), gj = tanhx<FAST>(z[1]), gf = sigm<FAST>(z[2] + a.fb), go = sigm<FAST>(z[3]);
        const size_t o = (size_t)row * H + unit;
        const float cn = fmaf(pc_[k], gf, gi * gj);
        a.c_out[o] = cn;
        const float hn = tanhx<FAST>(cn) * go;
        a.h_out[o] = hn;
        if (a.h_out_bf16) reinterpret_cast<unsigned short*>(a.h_out_bf16)[o] = (unsigned short)(f2bf2(hn, 0.f) & 0xffffu);   // (the staging's own conversion)
        if (a.gates_out && !LB_STAMP) {
            float* gp = a.gates_out + (size_t)row * 4 * H + unit;
            gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
        }
    }
    LBS(13);
#undef LBS
}
