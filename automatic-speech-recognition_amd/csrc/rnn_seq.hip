// rnn_seq.hip -- K2 / K2b: the recurrent sweep of one bidirectional layer, forward and BPTT.
//
// Replaces tf.nn.bidirectional_dynamic_rnn's per-step while-loop (reference las/layers.py:49-53,
// cell las/layers.py:31) -- ~6 tiny TF kernels per time step -- by ONE persistent launch per layer:
// a workgroup owns (direction, batch tile) for all T steps, h/c never leave the CU (LDS + VGPRs),
// the input projection x.W_ih+b was hoisted into one big K1 GEMM and arrives through `gates`.
//
//   LAS_PREC_F32 : fp32 FMA chains on the VALU, any H (parity mode).  Thread = hidden unit,
//                  8 batch rows per workgroup, h broadcast from LDS, W_hh streamed from L2.
//   LAS_PREC_BF16: v_mfma_f32_16x16x32_bf16.  16 batch rows per workgroup (one MFMA M-tile),
//                  4 waves x 64 hidden units; every wave owns all G gates of its units so the gate
//                  nonlinearity is lane-local on the accumulator layout.  W_hh is pre-packed into
//                  MFMA B-fragment order (1 KiB contiguous per wave-load); as many fragments as fit
//                  stay resident in LDS for the whole sweep, the rest stream from L2 each step.
//                  Next step's x-projection is prefetched under the current step's MFMAs.
// No grid barrier, no inter-workgroup traffic: directions and batch tiles are independent.
#include "las_common.h"
#include <stdlib.h>

#ifndef LAS_ABL
#define LAS_ABL 0      // development: bit mask of parts of the forward sweep to leave out (timing experiments, tools/abl_rnn.py)
#endif
#include "rnn_seq_args.h"

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would stall
// every recurrent step on the completion of that step's global stores and of the NEXT step's x-projection
// prefetch (cdna_hip_programming.md section 5, "Pipelining across barriers").  Cross-wave data here travels
// through LDS exclusively; global traffic is per-lane private within a sweep.

// ------------------------------------------------------------------------------------------------
// fp32 VALU kernels
// ------------------------------------------------------------------------------------------------
constexpr int F32_BT = 8;    // batch rows per workgroup
constexpr int F32_UPT = 4;   // hidden units per thread (H <= 1024)

template <int CELL>
__global__ __launch_bounds__(256) void rnn_seq_fwd_f32_kernel(RnnArgs a) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    constexpr int BT = F32_BT, UPT = F32_UPT;
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [H][BT]
    const int tid = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * BT;
    const int H = a.H, T = a.T, B = a.B, GH = G * H;
    const float* __restrict__ W = a.whh[dir];
    float cst[UPT][BT];
#pragma unroll
    for (int ui = 0; ui < UPT; ++ui)
#pragma unroll
        for (int r = 0; r < BT; ++r) cst[ui][r] = 0.f;
    for (int i = tid; i < H * BT; i += 256) hs[i] = 0.f;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        float hnew[UPT][BT];
#pragma unroll
        for (int ui = 0; ui < UPT; ++ui) {
            const int u = tid + ui * 256;
#pragma unroll
            for (int r = 0; r < BT; ++r) hnew[ui][r] = 0.f;
            if (u < H) {
                float acc[G][BT];
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int r = 0; r < BT; ++r) acc[q][r] = 0.f;
                for (int k = 0; k < H; ++k) {
                    float w[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) w[q] = W[(long long)k * a.ldw + q * H + u];
                    const float4 h0 = *reinterpret_cast<const float4*>(&hs[k * BT]);
                    const float4 h1 = *reinterpret_cast<const float4*>(&hs[k * BT + 4]);
                    const float hk[BT] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int q = 0; q < G; ++q)
#pragma unroll
                        for (int r = 0; r < BT; ++r) acc[q][r] = fmaf(hk[r], w[q], acc[q][r]);
                }
#pragma unroll
                for (int r = 0; r < BT; ++r) {
                    const int b = b0 + r;
                    if (b < B) {
                        const long long fr = ((long long)b * T + t) * 2 + dir;
                        float* gp = a.gates + fr * GH + u;
                        float h;
                        if (CELL == LAS_CELL_LSTM) {
                            const float gi = sigmoid_acc(acc[0][r] + gp[0]);
                            const float gj = tanh_acc(acc[G > 1 ? 1 : 0][r] + gp[H]);
                            const float gf = sigmoid_acc(acc[G > 2 ? 2 : 0][r] + gp[2 * H] + a.fb);
                            const float go = sigmoid_acc(acc[G > 3 ? 3 : 0][r] + gp[3 * H]);
                            const float c = cst[ui][r] * gf + gi * gj;
                            h = tanh_acc(c) * go;
                            cst[ui][r] = c;
                            gp[0] = gi; gp[H] = gj; gp[2 * H] = gf; gp[3 * H] = go;
                            a.cstate[fr * H + u] = c;
                        } else {
                            h = tanh_acc(acc[0][r] + gp[0]);
                        }
                        a.out[(long long)b * a.obs + (long long)t * a.ld_out + dir * H + u] = h;
                        hnew[ui][r] = h;
                    }
                }
            }
        }
        __syncthreads();  // every thread is done reading h_{t-1}
#pragma unroll
        for (int ui = 0; ui < UPT; ++ui) {
            const int u = tid + ui * 256;
            if (u < H) {
                *reinterpret_cast<float4*>(&hs[u * BT]) = make_float4(hnew[ui][0], hnew[ui][1], hnew[ui][2], hnew[ui][3]);
                *reinterpret_cast<float4*>(&hs[u * BT + 4]) = make_float4(hnew[ui][4], hnew[ui][5], hnew[ui][6], hnew[ui][7]);
            }
        }
        __syncthreads();
    }
}

// BPTT.  a.wpack = W_hh^T as fp32 [2][G*H][H] (coalesced over the output unit).
template <int CELL>
__global__ __launch_bounds__(256) void rnn_seq_bwd_f32_kernel(RnnArgs a) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    constexpr int BT = F32_BT, UPT = F32_UPT;
    extern __shared__ __attribute__((aligned(16))) float dps[];  // [G*H][BT]
    const int tid = threadIdx.x, dir = blockIdx.y, b0 = blockIdx.x * BT;
    const int H = a.H, T = a.T, B = a.B, GH = G * H;
    const float* __restrict__ WT = reinterpret_cast<const float*>(a.wpack) + (long long)dir * GH * H;
    float dhr[UPT][BT], dcc[UPT][BT];
#pragma unroll
    for (int ui = 0; ui < UPT; ++ui)
#pragma unroll
        for (int r = 0; r < BT; ++r) { dhr[ui][r] = 0.f; dcc[ui][r] = 0.f; }

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;          // reverse of the forward order
        const int tp = dir ? t + 1 : t - 1;         // the step that ran before t in the forward sweep
        const bool hasp = tp >= 0 && tp < T;
#pragma unroll
        for (int ui = 0; ui < UPT; ++ui) {
            const int u = tid + ui * 256;
            if (u < H) {
#pragma unroll
                for (int r = 0; r < BT; ++r) {
                    const int b = b0 + r;
                    float dz[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) dz[q] = 0.f;
                    if (b < B) {
                        const long long fr = ((long long)b * T + t) * 2 + dir;
                        const float dh = a.dout[(long long)b * a.dobs + (long long)t * a.ld_dout + dir * H + u] + dhr[ui][r];
                        float* gp = a.gates + fr * GH + u;
                        if (CELL == LAS_CELL_LSTM) {
                            const float gi = gp[0], gj = gp[H], gf = gp[2 * H], go = gp[3 * H];
                            const float c = a.cstate[fr * H + u];
                            const float cp = hasp ? a.cstate[(((long long)b * T + tp) * 2 + dir) * H + u] : 0.f;
                            const float tc = tanh_acc(c);
                            const float dc = dcc[ui][r] + dh * go * (1.f - tc * tc);
                            dcc[ui][r] = dc * gf;
                            dz[0] = dc * gj * gi * (1.f - gi);
                            dz[G > 1 ? 1 : 0] = dc * gi * (1.f - gj * gj);
                            dz[G > 2 ? 2 : 0] = dc * cp * gf * (1.f - gf);
                            dz[G > 3 ? 3 : 0] = dh * tc * go * (1.f - go);
                        } else {
                            const float h = a.out[(long long)b * a.obs + (long long)t * a.ld_out + dir * H + u];
                            dz[0] = dh * (1.f - h * h);
                        }
#pragma unroll
                        for (int q = 0; q < G; ++q) gp[q * H] = dz[q];
                    }
#pragma unroll
                    for (int q = 0; q < G; ++q) dps[(q * H + u) * BT + r] = dz[q];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ui = 0; ui < UPT; ++ui) {
            const int u = tid + ui * 256;
            if (u < H) {
                float acc[BT];
#pragma unroll
                for (int r = 0; r < BT; ++r) acc[r] = 0.f;
                for (int col = 0; col < GH; ++col) {
                    const float w = WT[(long long)col * H + u];
                    const float4 d0 = *reinterpret_cast<const float4*>(&dps[col * BT]);
                    const float4 d1 = *reinterpret_cast<const float4*>(&dps[col * BT + 4]);
                    acc[0] = fmaf(d0.x, w, acc[0]); acc[1] = fmaf(d0.y, w, acc[1]);
                    acc[2] = fmaf(d0.z, w, acc[2]); acc[3] = fmaf(d0.w, w, acc[3]);
                    acc[4] = fmaf(d1.x, w, acc[4]); acc[5] = fmaf(d1.y, w, acc[5]);
                    acc[6] = fmaf(d1.z, w, acc[6]); acc[7] = fmaf(d1.w, w, acc[7]);
                }
#pragma unroll
                for (int r = 0; r < BT; ++r) dhr[ui][r] = acc[r];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void transpose_whh_kernel(const float* W0, const float* W1, int ldw, int H, int GH,
                                                            float* out) {
    const long long total = 2LL * GH * H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int u = (int)(idx % H);
        const int col = (int)((idx / H) % GH);
        const int dir = (int)(idx / ((long long)H * GH));
        const float* W = dir ? W1 : W0;
        out[idx] = W[(long long)u * ldw + col];
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernels
// ------------------------------------------------------------------------------------------------
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- cluster exchange ---------------------------------------------------------------------------------
// With P > 1 the hidden units of one (direction, batch tile) are split over P workgroups (= P CUs), each
// keeping ITS slice of W_hh resident on chip for the whole sweep (streaming W_hh into one CU tops out at
// ~25-90 GB/s per CU and would cost 4-15 us per step).  Every step each member needs all H values of
// h_t (fwd) / all G*H values of d(pre-activation) (bwd): members publish their slice as 8-byte granules
// {tag = step+1, payload = 2 x bf16} with ONE relaxed agent-scope (sc1, write-through) store per granule,
// and read the others' granules with relaxed agent-scope loads until the tag matches -- the data IS the
// flag, no fence, placement independent (cdna_hip_programming.md G16, recipe R2).  Two slots alternate
// (a member can be at most one step ahead of the slowest reader).  The buffer is zeroed by a memset node
// before every launch; spins are bounded and report through err[0].
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
// Activations of the speed mode live in HBM as bf16 (SURVEY 8(d) algorithmic bytes): x-projections / saved gates, cell
// states, h and their gradients.  BfPtr keeps the kernels' element-wise addressing readable: p[i] reads a bf16 element as
// float, p[i] = v rounds (RNE) and stores it; explicit global address space (global_load/store_short, never flat).
typedef __attribute__((address_space(1))) unsigned short gio;
struct BfRef {
    gio* p;
    __device__ __forceinline__ operator float() const { return bf2f(*p); }
    __device__ __forceinline__ void operator=(float v) const { *p = f2bf(v); }
};
struct BfPtr {
    gio* p;
    __device__ __forceinline__ BfRef operator[](long long i) const { return BfRef{p + i}; }
    __device__ __forceinline__ BfPtr operator+(long long o) const { return BfPtr{p + o}; }
    __device__ __forceinline__ BfPtr& operator+=(long long o) { p += o; return *this; }
};
#define GF(p) (BfPtr{(gio*)(p)})
#define GCF(p) (BfPtr{(gio*)(p)})
typedef __attribute__((address_space(1))) float gfloat;          // fp32 globals (helper waves' LDS rings stay fp32)

// `local` = every member of this cluster runs on the same XCD (verified at kernel start, cluster_same_xcd): the
// granule then only has to reach that XCD's L2, so a workgroup-scope (sc0) store is enough and the consumers' sc1
// loads are served by the L2 instead of the fabric (-0.4 us per dependent step).  Otherwise: agent-scope
// write-through store, correct under any placement.
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned tag, unsigned val, bool local) {
    const unsigned long long v = ((unsigned long long)tag << 32) | val;
    if (local) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Fetch this thread's N granules (all loads in flight at once), then re-poll only the stale ones.
template <int N, int P, int GPM>
__device__ __forceinline__ void gather_granules(unsigned long long (&xv)[N], const unsigned long long* xslot, int pm, int tid,
                                                unsigned tag, int& errflag, int spin) {
    constexpr int PER = GPM / 256;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int m = (pm + 1 + n / PER) % P;
        xv[n] = __hip_atomic_load(xslot + (size_t)m * GPM + tid + (n % PER) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int budget = errflag ? 1 : spin;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int n = 0; n < N; ++n) ok &= (unsigned)(xv[n] >> 32) == tag;
        if (ok) break;
        if (--budget <= 0) { errflag = 1; break; }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if ((unsigned)(xv[n] >> 32) != tag) {
                const int m = (pm + 1 + n / PER) % P;
                xv[n] = __hip_atomic_load(xslot + (size_t)m * GPM + tid + (n % PER) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int CELL, int UT, int P>
struct RnnCfg {
    static constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    static constexpr int H = UT * 64, GH = G * H;
    static constexpr int UTP = UT / P;            // 16-unit tiles owned by one wave of one member
    static constexpr int UPM = H / P;             // hidden units owned by one member
    static constexpr int KS = H / 32;             // k-steps of the forward product (K = H)
    static constexpr int NFW = G * UTP * KS;      // B fragments per wave, forward
    static constexpr int LDH = H + 16;            // bf16 row pitch of the h tile: 8 dwords mod 64 banks -> the ds_read_b128 A-fragment reads of a lane group (8 or 16 rows x 2 k-chunks) hit 64 distinct banks (H + 8 was 2-way conflicted with 8-row tiles)
    static constexpr int HS_BYTES = 2 * 16 * LDH * 2;
    // W_hh slice placement: VGPR/AGPRs first (one wave per SIMD owns 512 registers per lane; an MFMA B operand
    // that already sits in a register costs no LDS cycle and no latency), LDS for the remainder.
    static constexpr int RCAP = 64;               // fragments (x4 registers) a wave may pin
    static constexpr int RF = cmin(NFW, RCAP);
    static constexpr int LF = cmin(NFW - RF, (150 * 1024 - HS_BYTES) / 4096);
    static constexpr int KSB = GH / 32;           // k-steps of the backward product (K = G*H)
    static constexpr int NFB = UTP * KSB;
    static constexpr int LDG = GH + 8;
    static constexpr int DP_BYTES = 2 * 16 * LDG * 2;
    static constexpr int RFB = cmin(NFB, RCAP);
    static constexpr int LFB = cmax(0, cmin(NFB - RFB, (150 * 1024 - DP_BYTES) / 4096));
    static constexpr int FWD_LDS = HS_BYTES + 4 * LF * 1024;      // for RT = 1; RT > 1 requires LF == 0 / LFB == 0
    static constexpr int BWD_LDS = DP_BYTES + 4 * LFB * 1024;
    static constexpr int GPM_F = 16 * UPM / 2;        // granules one member publishes per step, forward
    static constexpr int GPM_B = 16 * G * UPM / 2;    // ... backward
    static constexpr bool OK = (UT % P == 0) && (RF + LF == NFW) && (RFB + LFB == NFB);
};

template <int CELL, int UT, int P, int RT>
__global__ __launch_bounds__(256 * RT, 1) void rnn_seq_fwd_bf16_kernel(RnnArgs a) {
    using C = RnnCfg<CELL, UT, P>;
    constexpr int G = C::G, H = C::H, GH = C::GH, KS = C::KS, NFW = C::NFW, LDH = C::LDH, LF = C::LF, RF = C::RF;
    constexpr int UTP = C::UTP, UPM = C::UPM, GPM = C::GPM_F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // RT row tiles (16 batch rows each) share this workgroup as independent wave groups: 4 waves per row tile,
    // RT waves per SIMD -> the SIMD always has another chain's instructions to issue while one waits
    const int rt = threadIdx.x >> 8;
    unsigned short* hs = reinterpret_cast<unsigned short*>(smem + rt * C::HS_BYTES);   // [2][16][LDH] per row tile
    u16x8_t* wl = reinterpret_cast<u16x8_t*>(smem + RT * C::HS_BYTES);          // [4][LF][64] shared by the row tiles
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = a.T, B = a.B;
    const int cg = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;         // (tile group, direction), member
    if (cg >= a.ncl) return;
    const int dir = cg & 1, tile = (cg >> 1) * RT + rt, b0 = tile * 16;
    if (b0 >= B) return;                         // surplus row-tile group: terminated waves leave the barrier count
    const int cl = tile * 2 + dir;
    const int vw = pm * 4 + w;                                                   // virtual wave: owns units [vw*16*UTP, ..)
    const u16x8_t* __restrict__ Wp = reinterpret_cast<const u16x8_t*>(a.wpack) + ((size_t)dir * 4 * P + vw) * NFW * 64;
    unsigned long long* xb = a.xbuf + (size_t)cl * 2 * P * GPM;                 // [2 slots][P][GPM]
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(xb);
    int errflag = 0;
    const bool local = (P > 1 && RT == 1 && !a.force_agent) ? cluster_same_xcd(a.xcc + (size_t)cl * P, pm, P, tid, &errflag, a.spin) : false;

    u16x8_t wreg[RF > 0 ? RF : 1];
#pragma unroll
    for (int r = 0; r < RF; ++r) wreg[r] = Wp[r * 64 + lane];
#pragma unroll 4
    for (int fi = 0; fi < LF; ++fi) wl[(w * LF + fi) * 64 + lane] = Wp[(RF + fi) * 64 + lane];
    for (int i = tid; i < 16 * LDH; i += 256) hs[i] = 0;
    __syncthreads();

    // per-lane running pointers (rows g*4+r of the tile, this wave's first unit column); they advance by one
    // frame per step, so the loop body carries no 64-bit index arithmetic
    const int t0 = dir ? T - 1 : 0;
    const long long tstep = dir ? -1 : 1;
    const long long gstep = tstep * 2 * GH, cstep = tstep * 2 * H, ostep = tstep * a.ld_out;
    // rows past the end of a ragged batch tile read/write a scratch row (no exec-mask branches in the loop)
    BfPtr gptr[4];
    BfPtr cptr[4];
    BfPtr optr[4];
    long long gst[4], cst_[4], ost[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + g * 4 + r;
        const bool valid = b < B;
        const long long row = (long long)b;
        const int u0 = vw * (16 * UTP) + c;
        gptr[r] = GF(valid ? a.gates16 + ((row * T + t0) * 2 + dir) * GH + u0 : a.sink16 + u0);
        cptr[r] = GF((valid && a.cstate16) ? a.cstate16 + ((row * T + t0) * 2 + dir) * H + u0 : a.sink16 + u0);
        optr[r] = GF(valid ? a.out16 + row * a.obs + (long long)t0 * a.ld_out + dir * H + u0 : a.sink16 + u0);
        gst[r] = valid ? gstep : 0; cst_[r] = valid ? cstep : 0; ost[r] = valid ? ostep : 0;
    }
    float cst[UTP][4];
#pragma unroll
    for (int j = 0; j < UTP; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cst[j][r] = 0.f;

    // x-projection of the step about to run; it seeds the accumulators (acc = x.W_ih + b, then += h.W_hh)
    f32x4_t xn[G][UTP];
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int j = 0; j < UTP; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) xn[q][j][r] = gptr[r][q * H + j * 16];
    int cur = 0;
#ifdef LAS_PROF
    const bool prof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (prof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (prof && s >= 200 && s < 208) a.dbg[8 + (s - 200) * 8 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(k)
#endif
    for (int s = 0; s < T; ++s) {
        STAMP(0);
        f32x4_t acc[G][UTP];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < UTP; ++j) acc[q][j] = xn[q][j];
        if (s + 1 < T) {   // next step's x-projection flies under this step's MFMAs
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int j = 0; j < UTP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xn[q][j][r] = gptr[r][gst[r] + q * H + j * 16];
        }
        const unsigned short* hcur = hs + cur * 16 * LDH;
        u16x8_t av[KS];                  // all A fragments of h_{t-1} up front: one LDS round trip, then MFMAs back to back
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = *reinterpret_cast<const u16x8_t*>(&hcur[c * LDH + ks * 32 + g * 8]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int j = 0; j < UTP; ++j) {
                    const int fi = (q * UTP + j) * KS + ks;
                    const u16x8_t bv = fi < RF ? wreg[fi < RF ? fi : 0] : wl[(w * LF + (fi - RF)) * 64 + lane];
                    acc[q][j] = mfma_bf16_16x16x32(av[ks], bv, acc[q][j]);
                }
        }
        STAMP(1);
#ifdef LAS_PROF
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[G - 1][UTP - 1][3]));
#endif
        STAMP(2);
        unsigned short* hnext = hs + (cur ^ 1) * 16 * LDH;
        const unsigned slot_off = (unsigned)((s & 1) * P) * GPM * 8u;
        float sv_h[UTP][4], sv_g[G][UTP][4];     // results kept in registers; written to HBM after the exchange
#pragma unroll
        for (int j = 0; j < UTP; ++j) {
            const int unit = vw * (16 * UTP) + j * 16 + c;
            unsigned short hb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h;
                if (CELL == LAS_CELL_LSTM) {
                    const float gi = sigm<true>(acc[0][j][r]);
                    const float gj = tanhx<true>(acc[G > 1 ? 1 : 0][j][r]);
                    const float gf = sigm<true>(acc[G > 2 ? 2 : 0][j][r] + a.fb);
                    const float go = sigm<true>(acc[G > 3 ? 3 : 0][j][r]);
                    const float cc = cst[j][r] * gf + gi * gj;
                    cst[j][r] = cc;
                    h = tanhx<true>(cc) * go;
                    sv_g[0][j][r] = gi; sv_g[G > 1 ? 1 : 0][j][r] = gj; sv_g[G > 2 ? 2 : 0][j][r] = gf; sv_g[G > 3 ? 3 : 0][j][r] = go;
                } else {
                    h = tanhx<true>(acc[0][j][r]);
                }
                sv_h[j][r] = h;
                hb[r] = f2bf(h);
                hnext[(g * 4 + r) * LDH + unit] = hb[r];
            }
            if (P > 1 && s + 1 < T)     // publish this wave's slice: the lane's four rows of the tile as ONE 16-byte double granule
                granule16_store(xrs, slot_off + (unsigned)pm * GPM * 8u + ((unsigned)(w * UTP + j) * 64u + lane) * 16u, (unsigned)(s + 1),
                                (unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16), local);
        }
        STAMP(3);
        if (P > 1 && s + 1 < T) {       // gather the other members' slices of h_t into the LDS tile
            constexpr int NGT = (P > 1 ? (P - 1) * UTP : 1);              // double granules per thread: UTP per partner
            u32x4_t xv[NGT];
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                xv[n] = granule16_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 16u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NGT; ++n) ok &= xv[n].x == (unsigned)(s + 1) && xv[n].w == (unsigned)(s + 1);
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NGT; ++n) {
                    if (xv[n].x != (unsigned)(s + 1) || xv[n].w != (unsigned)(s + 1)) {
                        const int m = (pm + 1 + n / UTP) % P;
                        xv[n] = granule16_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 16u);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                const int di = (n % UTP) * 256 + tid;                      // double-granule index = (w'*UTP + j')*64 + lane'
                const int l2 = di & 63, wj = di >> 6;
                const int unit = m * UPM + wj * 16 + (l2 & 15);
                const int row = (l2 >> 4) * 4;
                hnext[row * LDH + unit] = (unsigned short)(xv[n].y & 0xffffu);
                hnext[(row + 1) * LDH + unit] = (unsigned short)(xv[n].y >> 16);
                hnext[(row + 2) * LDH + unit] = (unsigned short)(xv[n].z & 0xffffu);
                hnext[(row + 3) * LDH + unit] = (unsigned short)(xv[n].z >> 16);
            }
        }
        lds_barrier();
        STAMP(4);
        // bulk results of this step: nobody waits on these stores (the next vmcnt wait is a whole step away)
#pragma unroll
        for (int j = 0; j < UTP; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (CELL == LAS_CELL_LSTM) {
                    BfPtr gp = gptr[r] + j * 16;
                    gp[0] = sv_g[0][j][r]; gp[H] = sv_g[G > 1 ? 1 : 0][j][r]; gp[2 * H] = sv_g[G > 2 ? 2 : 0][j][r];
                    gp[3 * H] = sv_g[G > 3 ? 3 : 0][j][r];
                    cptr[r][j * 16] = cst[j][r];
                }
                optr[r][j * 16] = sv_h[j][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) { gptr[r] += gst[r]; optr[r] += ost[r]; if (CELL == LAS_CELL_LSTM) cptr[r] += cst_[r]; }
        cur ^= 1;
    }
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
#ifdef LAS_PROF
    if (prof) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#endif
}

// ------------------------------------------------------------------------------------------------
// forward sweep with helper waves.  In the kernel above the four compute waves also issue the step's HBM traffic
// (16 x-projection loads and 24 result stores per lane and step); removing those from the dependent chain is worth
// 0.4 us per step (measured by ablation: 1.77 -> 1.35 us).  Here waves 4-7 own ALL bulk HBM traffic of the workgroup, a
// quarter of the rows each: they keep the x-projections of the next steps in flight and hand them over through a
// 3-slot LDS ring in accumulator order, and they write the previous step's results (activated gates, c, h) from a second
// LDS ring with coalesced 16-byte stores.  The compute waves touch global memory only for the exchange granules.  One
// LDS barrier per step still orders everything.  (A single helper wave does not work: its ~80 memory instructions per
// step take longer than the compute chain, and the barrier then waits for it.)
// Rows past the end of a ragged batch tile alias the last valid row for their LOADS; they store nothing (round 6).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// L2 warmers.  Ablations (make abl ABL=32|64|128) showed that the bulk HBM traffic of a sweep costs the dependent chain
// 0.15 us per step NOT by its instruction count but by its MISSES: a load that goes to HBM (or walks the page table) sits in
// the CU's memory pipeline in front of the latency-critical granule traffic.  One extra workgroup per cluster -- on the same
// XCD, hence the same L2 -- follows the cluster's progress (the step tags in its exchange slots) and touches the operands of
// the next steps, so that the cluster's own loads are L2 hits.  Warmers are never waited for: a missing or late warmer only
// means cold loads; they leave when the cluster has published its last step or after a bounded number of polls.
//   seg[i]: base of row 0 / step 0 of operand i for this cluster, row stride rs[i] (elements), step stride ss[i] (elements,
//   signed), nb[i] bytes per row and step (multiple of 16).
// ------------------------------------------------------------------------------------------------
struct WarmSeg { const unsigned short* base; long long rs, ss; int nb; };
template <int NSEG, int LEAD, int AHEAD>      // keep steps [p + LEAD, p + AHEAD] warm (LEAD = how far ahead the cluster itself fetches, + 1)
__device__ __forceinline__ void l2_warmer(const WarmSeg (&seg)[NSEG], int rows, int T, const unsigned long long* tagp0, const unsigned long long* tagp1,
                                          int spin, unsigned short* sink, const int* xflag = nullptr, int xsc = 0) {
    typedef __attribute__((address_space(1))) const u32x4_t gcu4;
    unsigned acc = 0;
    int done = -1;                               // last step already touched
    int idle = 0;
    for (;;) {
        const unsigned t0 = (unsigned)(__hip_atomic_load(tagp0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const unsigned t1 = (unsigned)(__hip_atomic_load(tagp1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int p = (int)(t0 > t1 ? t0 : t1) - 1;                  // step whose result was published last (-1: none yet)
        int from = done + 1, to = p + AHEAD;
        if (from < p + LEAD) from = p + LEAD;
        if (to > T - 1) to = T - 1;
        if (xflag) {       // chunked x-projection: a line must never be touched before the chunk that writes it is complete (a stale
                           // copy in this XCD's L2 would be what the cluster reads later)
            int have = __hip_atomic_load(xflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (have < 1) have = 1;                // chunk 0 is complete by stream order (las_rnn_seq_fwd_chunked's precondition)
            const int th = (T + 1) / 2;
            if (have * xsc < th && to > have * xsc - 1) to = have * xsc - 1;      // (second half: every chunk is complete by then)
        }
        if (from <= to) {
            idle = 0;
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                const int per_row = seg[i].nb / 16, per_step = per_row * rows;
                const int total = per_step * (to - from + 1);
                for (int e = threadIdx.x; e < total; e += blockDim.x) {
                    const int st = from + e / per_step, r = (e % per_step) / per_row, q = e % per_row;
                    const u32x4_t v = *(gcu4*)(seg[i].base + (long long)r * seg[i].rs + (long long)st * seg[i].ss + q * 8);
                    acc ^= v.x ^ v.w;
                }
            }
            done = to;
        }
        if (p >= T - 2 || done >= T - 1) break;                      // the last step that is ever published is T - 2
        if (++idle > spin) break;
        __builtin_amdgcn_s_sleep(16);
    }
    if (acc == 0x9e3779b9u && sink) sink[threadIdx.x] = (unsigned short)acc;   // keeps the loads alive
}

#ifndef LAS_KS_SHARE_CU
#define LAS_KS_SHARE_CU 0        // 1: request only the LDS the kernel uses (timing experiments: lets other kernels share the sweep's CUs)
#endif
constexpr int ks_lds(int used) { return LAS_KS_SHARE_CU ? used : 159 * 1024; }   // all but 1 KB (the kernels have 256 B of static LDS): nothing that uses LDS fits next to the sweep

template <int CELL, int UT, int P, int RB = 16>
struct HwCfg {
    using C = RnnCfg<CELL, UT, P>;
    static constexpr int G = C::G, UPM = C::UPM, NFW = C::NFW;
    static constexpr int NHW = 4;                             // helper waves (one per SIMD, next to a compute wave)
    static constexpr int RFH = cmin(NFW, 32);                 // fragments pinned in registers (two waves per SIMD: 256 regs)
    static constexpr int LFH = NFW - RFH;                     // the rest: LDS
    static constexpr int XW = G * UPM, XP = XW + 4;           // x row: floats / padded pitch (4*XP = 16 mod 64 banks)
    static constexpr int OW = (CELL == LAS_CELL_LSTM ? (G + 2) : 1) * UPM, OP = OW + 4;   // result row [gates | c | h] or [h]
    static constexpr int NX = RB * XW / 8 / 64;               // 8-element (16-byte bf16) slots per step of the gate slice (64 lanes each)
    static constexpr int CE = RB == 16 ? 4 : 2;               // elements per access of the c / h slices (8- or 4-byte bf16)
    static constexpr int NC4 = RB * UPM / CE / 64;            // CE-element slots per step of the c (and h) slice
    static constexpr int XR_BYTES = 3 * RB * XP * 4, OR_BYTES = 2 * RB * OP * 4;          // x ring: 3 steps, result ring: 2 steps
    static constexpr int LDS = C::HS_BYTES + 4 * LFH * 1024 + XR_BYTES + OR_BYTES;
    static constexpr bool OK = C::OK && (XW % 64 == 0) && (NX % NHW == 0) && (NC4 % NHW == 0) && NC4 >= NHW && LDS <= 160 * 1024 &&
                               (RB == 16 || P > 1);
};

// RB = 8: see rnn_seq_bwd_ks_kernel -- the A operand repeats rows 0..7 in rows 8..15, every lane keeps the two accumulator rows
// r = 2*hsel + {0,1} of its (gl, unit) position: half the transcendentals, ring traffic and granule bytes per CU and step.
// RAGGED: rows of different lengths in one launch (beam search encodes utterances the reference feeds one at a time, unpadded): at
// frames t >= row_T[row] the row's h and c are forced to zero -- the backward direction, which starts in a short row's padding,
// reaches that row's last real frame with the zero state an unpadded run starts from; the forward direction's real frames come first;
// and the pad frames hold zeros, which is also the zero frame the pyramid appends to an odd-length utterance.
template <int CELL, int UT, int P, int RB, bool RAGGED = false>
__global__ __launch_bounds__(512, 1) void rnn_seq_fwd_hw_kernel(RnnArgs a) {
    static_assert(RB == 16 || RB == 8, "row tile");
    using C = RnnCfg<CELL, UT, P>;
    using HC = HwCfg<CELL, UT, P, RB>;
    constexpr int G = C::G, H = C::H, GH = C::GH, KS = C::KS, LDH = C::LDH, UTP = C::UTP, UPM = C::UPM, GPM = C::GPM_F;
    constexpr int RF = HC::RFH, LF = HC::LFH, XP = HC::XP, OP = HC::OP, XW = HC::XW, NHW = HC::NHW, CE = HC::CE;
    constexpr int NXH = HC::NX / NHW, NCH = HC::NC4 / NHW;
    constexpr int NV = RB == 16 ? 4 : 2;                      // accumulator rows a lane works on
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* hs = reinterpret_cast<unsigned short*>(smem);                                   // [2][16][LDH]
    u16x8_t* wl = reinterpret_cast<u16x8_t*>(smem + C::HS_BYTES);                                   // [4][LF][64]
    float* xring = reinterpret_cast<float*>(smem + C::HS_BYTES + 4 * LF * 1024);                    // [3][16][XP]
    float* oring = xring + 3 * RB * XP;                                                             // [2][RB][OP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = a.T, B = a.B;
    const int cg = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;
    if (cg >= a.ncl) return;
    const int dir = cg & 1, tile = cg >> 1, b0 = tile * RB;
    if (b0 >= B) return;
    const int cl = tile * 2 + dir;
    const int t0 = dir ? T - 1 : 0;
    const long long tstep = dir ? -1 : 1;
    if (pm >= P) {               // L2 warmer of cluster cl: the x-projection rows of the next steps (all members' columns)
        if (P == 1) return;
        const int rows = (B - b0) < RB ? (B - b0) : RB;
        // (touching the lines the results go to -- c, h -- as well was measured: 1.17 vs 1.15 us per step, left out)
        const unsigned long long* xbw = a.xbuf + (size_t)cl * 2 * P * GPM;
        const WarmSeg seg[1] = {{a.gates16 + ((long long)b0 * T + t0) * 2 * GH + dir * GH, (long long)T * 2 * GH, tstep * 2 * GH, GH * 2}};
        l2_warmer<1, 4, 12>(seg, rows, T, xbw, xbw + (size_t)P * GPM, a.spin, a.sink16, a.xflag, a.xsc);
        return;
    }
    int errflag = 0;
    const bool local = (P > 1 && !a.force_agent) ? cluster_same_xcd(a.xcc + (size_t)cl * P, pm, P, tid, &errflag, a.spin) : false;

    if (w >= 4) {
        // ============================ helper waves: all bulk HBM traffic ============================
        // HBM side: bf16, 16-byte (8 elements: gate slice) and 8-byte (4 elements: c / h slices) accesses; LDS rings: fp32 in
        // accumulator order, so the compute waves see exactly what they saw with fp32 storage.
        // Every access is (uniform base of the frame) + (32-bit per-lane element offset).
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) const u32x4_t gcu4;
        typedef __attribute__((address_space(1))) u32x4_t gu4;
        typedef __attribute__((address_space(1))) u32x2_t gu2;
        const int hw = w - 4;
        constexpr int GR8 = XW / 8;                    // 8-element pieces per row of the gate slice
        constexpr int UR4 = UPM / CE;                  // CE-element pieces per row of the c / h slice
        auto brow = [&](int row) { const int b = b0 + row; return (unsigned)(b < B ? b : B - 1); };
        unsigned xoff[NXH], coff[NCH], hoff[NCH];
        int xl[NXH], xo[NXH], co[NCH];
        // Round 6: rows past the end of the batch still LOAD the last valid row's operands (clamped addresses, no branch in the load
        // stream) but no longer STORE: up to RB - 1 lanes used to write "the same" results to that row's addresses, and in a train step the
        // copies were found to differ from the real row by one bf16 ulp in a few dozen saved gates (tools/probe_determinism.py: B = 5 / 8
        // of a 16-row tile, 64 units; the last store to land decided what BPTT read -- a run-to-run difference of 2e-6 in the gradients)
        unsigned xst = 0, cst_ = 0;                    // bit ii: the slot's row exists -> its results are stored
#pragma unroll
        for (int ii = 0; ii < NXH; ++ii) {             // gate slice: x-projection in, activated gates out (same addresses)
            const int idx = (ii * NHW + hw) * 64 + lane, row = idx / GR8, f = (idx % GR8) * 8;
            xoff[ii] = brow(row) * (unsigned)(T * 2 * GH) + dir * GH + (f / UPM) * H + pm * UPM + (f % UPM);
            xl[ii] = row * XP + f;
            xo[ii] = row * OP + f;
            xst |= (b0 + row < B ? 1u : 0u) << ii;
        }
#pragma unroll
        for (int ii = 0; ii < NCH; ++ii) {
            const int idx = (ii * NHW + hw) * 64 + lane, row = idx / UR4, u = (idx % UR4) * CE;
            coff[ii] = brow(row) * (unsigned)(T * 2 * H) + dir * H + pm * UPM + u;
            hoff[ii] = brow(row) * (unsigned)a.obs + dir * H + pm * UPM + u;
            co[ii] = row * OP + u;
            cst_ |= (b0 + row < B ? 1u : 0u) << ii;
        }
        auto gframe = [&](int s) { if (LAS_ABL & 128) s &= 1; if (LAS_ABL & 256) s &= 15; if (LAS_ABL & 1024) s = (s & 15) * 64; return a.gates16 + (long long)(t0 + s * tstep) * 2 * GH; };     // uniform frame bases
        auto cframe = [&](int s) { if (LAS_ABL & 128) s &= 1; if (LAS_ABL & 256) s &= 15; if (LAS_ABL & 1024) s = (s & 15) * 64; return a.cstate16 + (long long)(t0 + s * tstep) * 2 * H; };
        auto oframe = [&](int s) { if (LAS_ABL & 128) s &= 1; if (LAS_ABL & 256) s &= 15; if (LAS_ABL & 1024) s = (s & 15) * 64; return a.out16 + (long long)(t0 + s * tstep) * a.ld_out; };
        auto to_ring = [&](float* xr, const u32x4_t& v, int off) __attribute__((always_inline)) {   // 8 bf16 -> 2 x float4 in LDS
            f4v lo = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
            f4v hi = {__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
            *reinterpret_cast<f4v*>(xr + off) = lo;
            *reinterpret_cast<f4v*>(xr + off + 4) = hi;
        };
        int have_fr = a.xsc;                            // frames (from either end) of the x-projection known to be complete = chunks (a.xflag) x
                                                        // a.xsc; chunk 0 was produced in front of this launch in stream order (round 5: no set_word
                                                        // launch between the two).  Kept in FRAMES: `mm / a.xsc` was an integer division per step
        auto wait_chunk = [&](int st) __attribute__((always_inline)) {
            if (!a.xflag) return;
            const int mm = st < T - 1 - st ? st : T - 1 - st;
            if (mm < have_fr) return;
            int budget = a.spin < (1 << 16) ? a.spin : (1 << 16);     // ~0.1 s: a chunk is a sub-millisecond GEMM; if kernels of different
                                                                       // streams cannot overlap (a profiler that serialises them) fail fast
            for (;;) {
                const long long hf = (long long)__hip_atomic_load(a.xflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) * a.xsc;
                have_fr = hf > 0x7fffffffLL ? 0x7fffffff : (int)hf;
                if (mm < have_fr) break;
                if (--budget <= 0) {                   // the step is invalid from here on: report once, never wait again
                    if (a.status) a.status[0] = a.status_code;
                    have_fr = 0x7fffffff;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        };
        u32x4_t xq[NXH];                                // the loads in flight (this wave's share of one step)
        // ring: slot (s % 3) holds step s; steps 0 and 1 are staged here, step 2 is requested before the loop
        {
            wait_chunk(0);
            const unsigned short* gb = gframe(0);
#pragma unroll
            for (int ii = 0; ii < NXH; ++ii) xq[ii] = *(gcu4*)(gb + xoff[ii]);
#pragma unroll
            for (int ii = 0; ii < NXH; ++ii) to_ring(xring, xq[ii], xl[ii]);
        }
        if (T > 1) {
            wait_chunk(1);
            const unsigned short* gb = gframe(1);
#pragma unroll
            for (int ii = 0; ii < NXH; ++ii) xq[ii] = *(gcu4*)(gb + xoff[ii]);
#pragma unroll
            for (int ii = 0; ii < NXH; ++ii) to_ring(xring + RB * XP, xq[ii], xl[ii]);
        }
        if (T > 2) {
            wait_chunk(2);
            const unsigned short* gb = gframe(2);
#pragma unroll
            for (int ii = 0; ii < NXH; ++ii) xq[ii] = *(gcu4*)(gb + xoff[ii]);
        }
        __syncthreads();
        auto pack4 = [&](const float* src) __attribute__((always_inline)) {
            const f4v v = *reinterpret_cast<const f4v*>(src);
            u32x2_t r = {f2bf2(v.x, v.y), f2bf2(v.z, v.w)};
            return r;
        };
        typedef __attribute__((address_space(1))) unsigned int gu1;
        auto pack2 = [&](const float* src) __attribute__((always_inline)) {
            const float2 v = *reinterpret_cast<const float2*>(src);
            return f2bf2(v.x, v.y);
        };
        auto flush = [&](int slot, int s) __attribute__((always_inline)) {      // results of step s: LDS ring -> HBM (bf16)
            const float* orr = oring + slot * RB * OP;
            unsigned short* ob = oframe(s);
            if (CELL == LAS_CELL_LSTM) {
                unsigned short* gb = gframe(s);
                unsigned short* cb = cframe(s);
#pragma unroll
                for (int ii = 0; ii < NXH; ++ii) {
                    const u32x2_t lo = pack4(orr + xo[ii]), hi = pack4(orr + xo[ii] + 4);
                    const u32x4_t pk = {lo.x, lo.y, hi.x, hi.y};
                    if ((xst >> ii) & 1u) *(gu4*)(gb + xoff[ii]) = pk;
                }
#pragma unroll
                for (int ii = 0; ii < NCH; ++ii) {
                    if (!((cst_ >> ii) & 1u)) continue;
                    if (CE == 4) {
                        *(gu2*)(cb + coff[ii]) = pack4(orr + co[ii] + G * UPM);
                        *(gu2*)(ob + hoff[ii]) = pack4(orr + co[ii] + (G + 1) * UPM);
                    } else {
                        *(gu1*)(cb + coff[ii]) = pack2(orr + co[ii] + G * UPM);
                        *(gu1*)(ob + hoff[ii]) = pack2(orr + co[ii] + (G + 1) * UPM);
                    }
                }
            } else {
#pragma unroll
                for (int ii = 0; ii < NCH; ++ii) {
                    if (!((cst_ >> ii) & 1u)) continue;
                    if (CE == 4) *(gu2*)(ob + hoff[ii]) = pack4(orr + co[ii]);
                    else         *(gu1*)(ob + hoff[ii]) = pack2(orr + co[ii]);
                }
            }
        };
        for (int s = 0; s < T; ++s) {
            if (LAS_ABL & 16) { lds_barrier(); continue; }
            if (s + 2 < T) {        // step s+2 has had a whole step in flight: hand it to the ring, request step s+3
                float* xr = xring + ((s + 2) % 3) * RB * XP;
#pragma unroll
                for (int ii = 0; ii < NXH; ++ii) to_ring(xr, xq[ii], xl[ii]);
                if (s + 3 < T && !(LAS_ABL & 32)) {
                    wait_chunk(s + 3);
                    const unsigned short* gb = gframe(s + 3);
#pragma unroll
                    for (int ii = 0; ii < NXH; ++ii) xq[ii] = *(gcu4*)(gb + xoff[ii]);
                }
            }
            if (s >= 1 && !(LAS_ABL & 64)) flush((s - 1) & 1, s - 1);
            lds_barrier();
        }
        flush((T - 1) & 1, T - 1);
        return;
    }

    // ===================================== compute waves =====================================
    const int vw = pm * 4 + w;                                                   // virtual wave: owns units [vw*16*UTP, ..)
    const u16x8_t* __restrict__ Wp = reinterpret_cast<const u16x8_t*>(a.wpack) + ((size_t)dir * 4 * P + vw) * C::NFW * 64;
    unsigned long long* xb = a.xbuf + (size_t)cl * 2 * P * GPM;                 // [2 slots][P][GPM]
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(xb);
    u16x8_t wreg[RF > 0 ? RF : 1];
#pragma unroll
    for (int r = 0; r < RF; ++r) wreg[r] = Wp[r * 64 + lane];
#pragma unroll 4
    for (int fi = 0; fi < LF; ++fi) wl[(w * LF + fi) * 64 + lane] = Wp[(RF + fi) * 64 + lane];
    for (int i = tid; i < 16 * LDH; i += 256) hs[i] = 0;
    __syncthreads();

    const int gl = RB == 16 ? g : (g & 1), hsel = RB == 16 ? 0 : (g >> 1);
    const int row0 = RB == 16 ? g * 4 : gl * 4 + hsel * 2;     // the lane's rows are row0 + k, k < NV
    // LDS position of batch row r in the h tile.  RB = 8: rows r and r + 4 are written by the same 32-lane group (ds_write_b16),
    // and with the pitch the A-fragment reads want (8 dwords mod 16) four positions apart is the same bank -> interleave them
    auto hpos = [](int r) { return RB == 16 ? r : (((r & 3) << 1) | (r >> 2)); };
    float cst[UTP][NV];
#pragma unroll
    for (int j = 0; j < UTP; ++j)
#pragma unroll
        for (int r = 0; r < NV; ++r) cst[j][r] = 0.f;
    int rowT[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) rowT[r] = RAGGED ? a.row_T[(b0 + row0 + r) < B ? (b0 + row0 + r) : B - 1] : 0x7fffffff;
    int cur = 0;
#ifdef LAS_PROF
    const bool hprof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (hprof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#define HSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (hprof && s >= 200 && s < 208) a.dbg[8 + (s - 200) * 8 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HSTAMP(k)
#endif
    for (int s = 0; s < T; ++s) {
        HSTAMP(0);
        const float* xr = xring + (s % 3) * RB * XP;
        float* orr = oring + (s & 1) * RB * OP;
        const unsigned short* hcur = hs + cur * 16 * LDH;
        u16x8_t av[KS];                          // A fragments of h_{t-1} first: the MFMAs wait on nothing else
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = *reinterpret_cast<const u16x8_t*>(&hcur[hpos(c & (RB - 1)) * LDH + ((LAS_ABL & 2) ? 0 : ks) * 32 + g * 8]);
        __builtin_amdgcn_sched_barrier(0);       // all of them in flight before anything else (the compiler otherwise fetches them in pairs)
        float xv[G][UTP][NV];                    // x.W_ih + b from the ring: read under the MFMAs, added after them
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < UTP; ++j)
#pragma unroll
                for (int r = 0; r < NV; ++r) xv[q][j][r] = xr[(row0 + r) * XP + q * UPM + (w * UTP + j) * 16 + c];
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t acc[G][UTP];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < UTP; ++j) acc[q][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int j = 0; j < UTP; ++j) {
                    const int fi = (q * UTP + j) * KS + ks;
                    const u16x8_t bv = fi < RF ? wreg[fi < RF ? fi : 0] : wl[(w * LF + (fi - RF)) * 64 + lane];
                    if (!(LAS_ABL & 1) || ks == 0) acc[q][j] = mfma_bf16_16x16x32(av[ks], bv, acc[q][j]);
                }
            __builtin_amdgcn_sched_barrier(0);   // k-step-major issue order: G*UTP independent accumulators between dependent MFMAs
        }
        float pre[G][UTP][NV];                   // pre-activations of the lane's rows (RB = 8: its half of the duplicated tile)
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int j = 0; j < UTP; ++j)
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    float v;
                    if (RB == 16) v = acc[q][j][r];
                    else { const float lo = (r & 1) ? acc[q][j][1] : acc[q][j][0], hi = (r & 1) ? acc[q][j][3] : acc[q][j][2]; v = hsel ? hi : lo; }
                    pre[q][j][r] = v + xv[q][j][r];
                }
#ifdef LAS_PROF
        asm volatile("s_nop 0" :: "v"(pre[0][0][0]), "v"(pre[G - 1][UTP - 1][NV - 1]));     // MFMA results landed
#endif
        HSTAMP(1);
        unsigned short* hnext = hs + (cur ^ 1) * 16 * LDH;
        const unsigned slot_off = (unsigned)((s & 1) * P) * GPM * 8u;
#pragma unroll
        for (int j = 0; j < UTP; ++j) {
            const int ul = (w * UTP + j) * 16 + c;                   // unit inside this member's slice
            const int unit = pm * UPM + ul;
            unsigned short hb[NV];
            float res[NV][6];                    // gates, c, h of the lane's rows: h is published FIRST, the ring writes follow
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                float h;
                if (CELL == LAS_CELL_LSTM) {
                    constexpr bool TR = !(LAS_ABL & 4);
                    const float gi = TR ? sigm<true>(pre[0][j][r]) : pre[0][j][r] * 0.5f;
                    const float gj = TR ? tanhx<true>(pre[G > 1 ? 1 : 0][j][r]) : pre[G > 1 ? 1 : 0][j][r] * 0.25f;
                    const float gf = TR ? sigm<true>(pre[G > 2 ? 2 : 0][j][r] + a.fb) : pre[G > 2 ? 2 : 0][j][r] * 0.125f;
                    const float go = TR ? sigm<true>(pre[G > 3 ? 3 : 0][j][r]) : pre[G > 3 ? 3 : 0][j][r] * 0.75f;
                    float cc = cst[j][r] * gf + gi * gj;
                    h = (TR ? tanhx<true>(cc) : cc * 0.3f) * go;
                    if (RAGGED && t0 + s * (int)tstep >= rowT[r]) { cc = 0.f; h = 0.f; }
                    cst[j][r] = cc;
                    res[r][0] = gi; res[r][1] = gj; res[r][2] = gf; res[r][3] = go; res[r][4] = cc;
                } else {
                    h = tanhx<true>(pre[0][j][r]);
                    if (RAGGED && t0 + s * (int)tstep >= rowT[r]) h = 0.f;
                }
                res[r][5] = h;
                hb[r] = f2bf(h);
            }
            if (P > 1 && s + 1 < T) {   // publish this wave's slice: the lane's rows of the tile as ONE granule
                if constexpr (RB == 16)
                    granule16_store(xrs, slot_off + (unsigned)pm * GPM * 8u + ((unsigned)(w * UTP + j) * 64u + lane) * 16u, (unsigned)(s + 1),
                                    (unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[NV > 2 ? 2 : 0] | ((unsigned)hb[NV > 2 ? 3 : 0] << 16), local);
                else
                    granule8_store(xrs, slot_off + (unsigned)pm * GPM * 8u + ((unsigned)(w * UTP + j) * 64u + lane) * 8u, (unsigned)(s + 1),
                                   (unsigned)hb[0] | ((unsigned)hb[1] << 16), local);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                float* orow = orr + (row0 + r) * OP;
                hnext[hpos(row0 + r) * LDH + unit] = hb[r];
                if (CELL == LAS_CELL_LSTM) {
                    if (LAS_ABL & 8) { orow[ul] = res[r][5]; continue; }
                    orow[ul] = res[r][0]; orow[(G > 1 ? 1 : 0) * UPM + ul] = res[r][1]; orow[(G > 2 ? 2 : 0) * UPM + ul] = res[r][2];
                    orow[(G > 3 ? 3 : 0) * UPM + ul] = res[r][3];
                    orow[G * UPM + ul] = res[r][4];
                    orow[(G + 1) * UPM + ul] = res[r][5];
                } else {
                    orow[ul] = res[r][5];
                }
            }
        }
        HSTAMP(2);
        if (P > 1 && s + 1 < T) {       // gather the other members' slices of h_t into the LDS tile
            constexpr int NGT = (P > 1 ? (P - 1) * UTP : 1);
            if constexpr (RB == 16) {
            u32x4_t xv[NGT];
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                xv[n] = granule16_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 16u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NGT; ++n) ok &= xv[n].x == (unsigned)(s + 1) && xv[n].w == (unsigned)(s + 1);
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NGT; ++n) {
                    if (xv[n].x != (unsigned)(s + 1) || xv[n].w != (unsigned)(s + 1)) {
                        const int m = (pm + 1 + n / UTP) % P;
                        xv[n] = granule16_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 16u);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                const int di = (n % UTP) * 256 + tid;
                const int l2 = di & 63, wj = di >> 6;
                const int unit = m * UPM + wj * 16 + (l2 & 15);
                const int row = (l2 >> 4) * 4;
                hnext[row * LDH + unit] = (unsigned short)(xv[n].y & 0xffffu);
                hnext[(row + 1) * LDH + unit] = (unsigned short)(xv[n].y >> 16);
                hnext[(row + 2) * LDH + unit] = (unsigned short)(xv[n].z & 0xffffu);
                hnext[(row + 3) * LDH + unit] = (unsigned short)(xv[n].z >> 16);
            }
            } else {
            u32x2_t xv[NGT];
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                xv[n] = granule8_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 8u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NGT; ++n) ok &= xv[n].x == (unsigned)(s + 1);
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NGT; ++n) {
                    if (xv[n].x != (unsigned)(s + 1)) {
                        const int m = (pm + 1 + n / UTP) % P;
                        xv[n] = granule8_load(xrs, slot_off + (unsigned)m * GPM * 8u + ((unsigned)(n % UTP) * 256u + tid) * 8u);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / UTP) % P;
                const int di = (n % UTP) * 256 + tid;
                const int l2 = di & 63, wj = di >> 6, g2 = l2 >> 4;
                const int unit = m * UPM + wj * 16 + (l2 & 15);
                const int row = (g2 & 1) * 4 + (g2 >> 1) * 2;
                hnext[hpos(row) * LDH + unit] = (unsigned short)(xv[n].y & 0xffffu);
                hnext[hpos(row + 1) * LDH + unit] = (unsigned short)(xv[n].y >> 16);
            }
            }
        }
        HSTAMP(3);
        lds_barrier();
        HSTAMP(4);
        cur ^= 1;
    }
#ifdef LAS_PROF
    if (hprof) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#endif
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
}

template <int CELL, int UT, int P, int RT>
__global__ __launch_bounds__(256 * RT, 1) void rnn_seq_bwd_bf16_kernel(RnnArgs a) {
    using C = RnnCfg<CELL, UT, P>;
    constexpr int G = C::G, H = C::H, GH = C::GH, KSB = C::KSB, NFB = C::NFB, LDG = C::LDG, LFB = C::LFB, RFB = C::RFB;
    constexpr int UTP = C::UTP, UPM = C::UPM, GPM = C::GPM_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rt = threadIdx.x >> 8;
    unsigned short* dps = reinterpret_cast<unsigned short*>(smem + rt * C::DP_BYTES);  // [2][16][LDG] per row tile
    u16x8_t* wl = reinterpret_cast<u16x8_t*>(smem + RT * C::DP_BYTES);          // [4][LFB][64]
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = a.T, B = a.B;
    const int cg = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;
    if (cg >= a.ncl) return;
    const int dir = cg & 1, tile = (cg >> 1) * RT + rt, b0 = tile * 16;
    if (b0 >= B) return;
    const int cl = tile * 2 + dir;
    const int vw = pm * 4 + w;
    const u16x8_t* __restrict__ Wp = reinterpret_cast<const u16x8_t*>(a.wpack) + ((size_t)dir * 4 * P + vw) * NFB * 64;
    unsigned long long* xb = a.xbuf + (size_t)cl * 2 * P * GPM;
    int errflag = 0;
    const bool local = (P > 1 && RT == 1 && !a.force_agent) ? cluster_same_xcd(a.xcc + (size_t)cl * P, pm, P, tid, &errflag, a.spin) : false;

    u16x8_t wreg[RFB > 0 ? RFB : 1];
#pragma unroll
    for (int r = 0; r < RFB; ++r) wreg[r] = Wp[r * 64 + lane];
#pragma unroll 4
    for (int fi = 0; fi < LFB; ++fi) wl[(w * LFB + fi) * 64 + lane] = Wp[(RFB + fi) * 64 + lane];

    // the sweep visits t in the REVERSE of the forward order; the next visited frame t+tstep is also the
    // forward predecessor of t, so c_prev(t) is simply c(t+tstep) -- loaded one step further ahead
    const int t0 = dir ? 0 : T - 1;
    const long long tstep = dir ? 1 : -1;
    const long long gstep = tstep * 2 * GH, cstep = tstep * 2 * H, ostep = tstep * a.ld_out, dstep = tstep * a.ld_dout;
    BfPtr gptr[4];
    BfPtr cptr[4];
    BfPtr optr[4];
    BfPtr dptr[4];
    long long gst[4], cst_[4], ost[4], dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + g * 4 + r;
        const bool valid = b < B;
        const long long row = (long long)b;
        const int u0 = vw * (16 * UTP) + c;
        gptr[r] = GF(valid ? a.gates16 + ((row * T + t0) * 2 + dir) * GH + u0 : a.sink16 + u0);
        cptr[r] = GF((valid && a.cstate16) ? a.cstate16 + ((row * T + t0) * 2 + dir) * H + u0 : a.sink16 + u0);
        optr[r] = GF(valid ? a.out16 + row * a.obs + (long long)t0 * a.ld_out + dir * H + u0 : a.sink16 + u0);
        dptr[r] = GCF(valid ? a.dout16 + row * a.dobs + (long long)t0 * a.ld_dout + dir * H + u0 : a.sink16 + u0);
        gst[r] = valid ? gstep : 0; cst_[r] = valid ? cstep : 0; ost[r] = valid ? ostep : 0; dst[r] = valid ? dstep : 0;
    }
    f32x4_t dhr[UTP];
    float dcc[UTP][4];
#pragma unroll
    for (int j = 0; j < UTP; ++j) {
        dhr[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) dcc[j][r] = 0.f;
    }

    // per-step operands, ONE register set, refilled for step s+1 right after step s's gate math consumed
    // them, so the loads fly under the dG.W^T MFMAs:  dout, activated gates (lstm) | h_t (rnn), c_t, c_next
    constexpr int NG = CELL == LAS_CELL_LSTM ? 4 : 1;
    float n_do[UTP][4], n_g[NG][UTP][4], n_c[UTP][4], n_cn[UTP][4];
#pragma unroll
    for (int j = 0; j < UTP; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            n_do[j][r] = dptr[r][j * 16];
            if (CELL == LAS_CELL_LSTM) {
#pragma unroll
                for (int q = 0; q < NG; ++q) n_g[q][j][r] = gptr[r][q * H + j * 16];
                n_c[j][r] = cptr[r][j * 16];
                n_cn[j][r] = T > 1 ? cptr[r][cst_[r] + j * 16] : 0.f;
            } else {
                n_g[0][j][r] = optr[r][j * 16];
                n_c[j][r] = 0.f; n_cn[j][r] = 0.f;
            }
        }
    __syncthreads();  // LDS weight fragments visible

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        unsigned short* dpc = dps + cur * 16 * LDG;
        unsigned long long* xslot = xb + (size_t)((s & 1) * P) * GPM;
        float sv_z[G][UTP][4];
#pragma unroll
        for (int j = 0; j < UTP; ++j) {
            const int unit = vw * (16 * UTP) + j * 16 + c;
            unsigned short zb[G][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = n_do[j][r] + dhr[j][r];
                float dz[G];
                if (CELL == LAS_CELL_LSTM) {
                    const float gi = n_g[0][j][r], gj = n_g[NG > 1 ? 1 : 0][j][r], gf = n_g[NG > 2 ? 2 : 0][j][r],
                                go = n_g[NG > 3 ? 3 : 0][j][r];
                    const float cprev = (s + 1 < T) ? n_cn[j][r] : 0.f;
                    const float tc = tanhx<true>(n_c[j][r]);
                    const float dc = dcc[j][r] + dh * go * (1.f - tc * tc);
                    dcc[j][r] = dc * gf;
                    dz[0] = dc * gj * gi * (1.f - gi);
                    dz[G > 1 ? 1 : 0] = dc * gi * (1.f - gj * gj);
                    dz[G > 2 ? 2 : 0] = dc * cprev * gf * (1.f - gf);
                    dz[G > 3 ? 3 : 0] = dh * tc * go * (1.f - go);
                } else {
                    const float h = n_g[0][j][r];
                    dz[0] = dh * (1.f - h * h);
                }
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    zb[q][r] = f2bf(dz[q]);
                    dpc[(g * 4 + r) * LDG + q * H + unit] = zb[q][r];
                    sv_z[q][j][r] = dz[q];
                }
            }
            if (P > 1) {   // publish: per (gate q, tile j): 2 granules per lane
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        granule_store(xslot + (size_t)pm * GPM + (((w * UTP + j) * G + q) * 2 + k) * 64 + lane, (unsigned)(s + 1),
                                      (unsigned)zb[q][2 * k] | ((unsigned)zb[q][2 * k + 1] << 16), local);
            }
        }
        // advance to the next visited frame and refill the operand registers (dz of this step is written
        // to HBM after the exchange; the pointers keep the previous frame in gprev)
        BfPtr gprev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { gprev[r] = gptr[r]; gptr[r] += gst[r]; optr[r] += ost[r]; dptr[r] += dst[r]; if (CELL == LAS_CELL_LSTM) cptr[r] += cst_[r]; }
        if (s + 1 < T) {
#pragma unroll
            for (int j = 0; j < UTP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    n_do[j][r] = dptr[r][j * 16];
                    if (CELL == LAS_CELL_LSTM) {
#pragma unroll
                        for (int q = 0; q < NG; ++q) n_g[q][j][r] = gptr[r][q * H + j * 16];
                        n_c[j][r] = n_cn[j][r];
                        n_cn[j][r] = (s + 2 < T) ? cptr[r][cst_[r] + j * 16] : 0.f;
                    } else {
                        n_g[0][j][r] = optr[r][j * 16];
                    }
                }
        }
        if (P > 1) {
            constexpr int PER = GPM / 256, NGT = (P > 1 ? (P - 1) * PER : 1);
            unsigned long long xv[NGT];
            gather_granules<NGT, P, GPM>(xv, xslot, pm, tid, (unsigned)(s + 1), errflag, a.spin);
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int m = (pm + 1 + n / PER) % P;
                const int gi_ = tid + (n % PER) * 256;
                const unsigned v = (unsigned)xv[n];
                const int l2 = gi_ & 63, k = (gi_ >> 6) & 1, rest = gi_ >> 7;   // rest = (w'*UTP + j')*G + q
                const int q = rest % G, wj = rest / G;
                const int col = q * H + m * UPM + wj * 16 + (l2 & 15);
                const int row = (l2 >> 4) * 4 + 2 * k;
                dpc[row * LDG + col] = (unsigned short)(v & 0xffffu);
                dpc[(row + 1) * LDG + col] = (unsigned short)(v >> 16);
            }
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < UTP; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < G; ++q) gprev[r][q * H + j * 16] = sv_z[q][j][r];
        f32x4_t acc[UTP];
#pragma unroll
        for (int j = 0; j < UTP; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const u16x8_t av = *reinterpret_cast<const u16x8_t*>(&dpc[c * LDG + ks * 32 + g * 8]);
#pragma unroll
            for (int j = 0; j < UTP; ++j) {
                const int fi = j * KSB + ks;
                const u16x8_t bv = fi < RFB ? wreg[fi < RFB ? fi : 0] : wl[(w * LFB + (fi - RFB)) * 64 + lane];
                acc[j] = mfma_bf16_16x16x32(av, bv, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < UTP; ++j) dhr[j] = acc[j];
        cur ^= 1;
    }
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
}

// ====================================================================================================
// BPTT with a K-split exchange (P > 1).  dh_{t-1} = dG_t . W_hh^T.  Instead of all-gathering dG_t (16 x G*H
// bf16 -> every member reads (P-1)/P of it every step), each member multiplies ONLY its own gate columns
// (K slice = G*H/P) against the matching rows of W_hh^T for ALL hidden units, and the fp32 partial sums are
// reduce-scattered: a wave publishes, for every other member, the accumulator tile that member owns and adds
// the P-1 tiles it receives for its own tile.  Producer and consumer lanes hold the same (row, unit) position
// of the MFMA accumulator layout, so a received granule is added straight into a register: half the inbound
// granules of the all-gather, no LDS decode, exact fp32 partials.
//   wave w of member pm owns unit tiles (pm, w, j), j < UTP, and computes partial tiles (m, w, j) for all m.
// ====================================================================================================
// value of the neighbouring lane (lane ^ 1): one v_mov_b32 with DPP quad_perm [1,0,3,2], no LDS crossbar trip
__device__ __forceinline__ float swap_lane_pair(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

template <int CELL, int UT, int P, int RB = 16>
struct KsCfg {
    using C = RnnCfg<CELL, UT, P>;
    static constexpr int G = C::G, H = C::H, UTP = C::UTP, UPM = C::UPM;
    static constexpr int KP = G * UPM;               // own gate columns = K of the member's product
    static constexpr int KSP = KP / 32;
    static constexpr int NFR = P * UTP * KSP;        // B fragments per wave (all kept in registers)
    static constexpr int LDZ = KP + 16;              // bf16 row pitch of the own-dG tile (8 dwords mod 64 banks, see RnnCfg::LDH)
    static constexpr int DZ_BYTES = 2 * RB * LDZ * 2;
    static constexpr int GPD = 4 * UTP * 4 * 64;     // granules one member sends to ONE other member per step
    static constexpr bool OK = C::OK && NFR <= 64 && (KP % 32 == 0);
};

// RB = batch rows per tile.  RB = 8: the step is bound by per-lane work (gate backward, operand loads, granules), all of which
// scale with the rows a CU owns, while the MFMAs do not care -- so an 8-row tile feeds the MFMA an A operand whose rows 8..15
// repeat rows 0..7.  Lanes 32..63 then hold the same accumulators as lanes 0..31 and every lane keeps HALF of them (selected
// by hsel = lane >> 5): one row x one unit pair per lane instead of two rows, half the loads / stores / transcendentals /
// granule bytes per step, twice as many CUs per batch.
// CH: dout arrives in chunks (a.dflag): a separate instantiation -- the kernel sits at the register limit and the plain one must not change
// PG (round 5, with CH): the sweep PUBLISHES its progress -- dZ leaves with agent-scope (write-through) stores, and every a.pstep steps each
// member waits for its own stores and writes the step count into a.prog -- so that the layer's weight gradients can follow the sweep window
// by window on another stream (las_rnn_seq_bwd_db_progress) instead of starting when it ends.  A separate instantiation, like CH.
#ifndef LAS_PG_SC1_STORES
#define LAS_PG_SC1_STORES 0
#endif
template <int CELL, int UT, int P, int RB, bool CH = false, bool PG = false>
__global__ __launch_bounds__(256, 1) void rnn_seq_bwd_ks_kernel(RnnArgs a) {
    static_assert(RB == 16 || RB == 8, "row tile");
    using K = KsCfg<CELL, UT, P, RB>;
    constexpr int NR = RB == 16 ? 2 : 1;              // rows of the pair layout a lane owns
    constexpr int G = K::G, H = K::H, GH = G * H, UTP = K::UTP, UPM = K::UPM, KSP = K::KSP, NFR = K::NFR, LDZ = K::LDZ, GPD = K::GPD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* dzs = reinterpret_cast<unsigned short*>(smem);              // [2][RB][LDZ] own dG slice (bf16)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = a.T, B = a.B;
    const int cl = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;
    if (cl >= a.ncl) return;
    const int dir = cl & 1, b0 = (cl >> 1) * RB;
    if (b0 >= B) return;
    // (an L2 warmer workgroup like the forward sweep's was measured here too -- saved gates, cell states, dout of the next steps:
    //  1.27 vs 1.27 us per step; with EVERY access L2-resident (make abl ABL=512) the step is 1.20)
    const int vw = pm * 4 + w;
    const int gl = RB == 16 ? g : (g & 1), hsel = RB == 16 ? 0 : (g >> 1);
    // fragments (dir, vw, m, j, ks), all in registers
    const u16x8_t* __restrict__ Wp = reinterpret_cast<const u16x8_t*>(a.wpack) + ((size_t)dir * 4 * P + vw) * NFR * 64;
    u16x8_t wreg[NFR];
#pragma unroll
    for (int r = 0; r < NFR; ++r) wreg[r] = Wp[r * 64 + lane];
    // inbox of member d: [2 slots][P dst][P src][GPD]
    unsigned long long* xb = a.xbuf + (size_t)cl * 2 * P * P * GPD;
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(xb);
    int errflag = 0;
    const bool local = a.force_agent ? false : cluster_same_xcd(a.xcc + (size_t)cl * P, pm, P, tid, &errflag, a.spin);
    // "this sweep is on the machine": lets the host hold back work of other streams until then (las_wait_word)
    if (a.announce && a.status && cl == 0 && pm == 0 && tid == 0)
        __hip_atomic_store(a.status + 1, a.announce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const int t0 = dir ? 0 : T - 1;
    const long long tstep = dir ? 1 : -1;
    const long long gstep = tstep * 2 * GH, cstep = tstep * 2 * H, ostep = tstep * a.ld_out, dstep = tstep * a.ld_dout;
    // ---- element-wise ("pair") layout of the gate backward.  The MFMA accumulator layout gives a lane 4 rows x 1 unit; with
    // bf16 storage that would mean 2-byte HBM accesses (28 loads + 16 stores per lane and step, partial-line writes).
    // Instead a lane owns the unit PAIR (c & ~1, c | 1) of 2 rows -- even lanes rows g*4 + {0,1}, odd lanes rows g*4 + {2,3}
    // -- so every access is one aligned 32-bit word holding two adjacent units: 14 loads + 8 stores per step, and the own-dG
    // tile is written to LDS as packed pairs.  Only dh (4 values per tile) changes layout: one DPP swap with the
    // neighbouring lane.  dc (the carried cell gradient) lives in the pair layout throughout.
    typedef __attribute__((address_space(1))) unsigned int gu32;
    const int odd = c & 1;
    gu32* gptr[2];            // row base of the gate block (activated gates in, d(pre-activation) out), at this lane's unit pair
    const gu32* cptr[2];
    const gu32* optr[2];
    const gu32* dptr[2];
    long long gst[2], cst_[2], ost[2], dst[2];           // per-step advance in 32-bit words (0 for rows past the batch)
    float vrow[2];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int b = b0 + gl * 4 + odd * 2 + (RB == 16 ? rr : hsel);
        const bool valid = b < B;
        const long long row = (long long)b;
        const int u0 = vw * (16 * UTP) + (c & ~1);
        gptr[rr] = (gu32*)(valid ? a.gates16 + ((row * T + t0) * 2 + dir) * GH + u0 : a.sink16 + u0);
        cptr[rr] = (const gu32*)((valid && a.cstate16) ? a.cstate16 + ((row * T + t0) * 2 + dir) * H + u0 : a.sink16 + u0);
        optr[rr] = (const gu32*)(valid ? a.out16 + row * a.obs + (long long)t0 * a.ld_out + dir * H + u0 : a.sink16 + u0);
        dptr[rr] = (const gu32*)(valid ? a.dout16 + row * a.dobs + (long long)t0 * a.ld_dout + dir * H + u0 : a.sink16 + u0);
        gst[rr] = valid ? gstep / 2 : 0; cst_[rr] = valid ? cstep / 2 : 0; ost[rr] = valid ? ostep / 2 : 0; dst[rr] = valid ? dstep / 2 : 0;
        if (LAS_ABL & 512) { gst[rr] = 0; cst_[rr] = 0; ost[rr] = 0; dst[rr] = 0; }      // timing experiment: every step hits the same (L2-resident) frame
        vrow[rr] = valid ? 1.f : 0.f;
    }
    auto lo = [](unsigned v) { return __uint_as_float(v << 16); };
    auto hi = [](unsigned v) { return __uint_as_float(v & 0xffff0000u); };
    float bsum[G][UTP][2];                   // bias gradient: column sums of dz over this lane's rows and all steps, per unit of the pair
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int j = 0; j < UTP; ++j) { bsum[q][j][0] = 0.f; bsum[q][j][1] = 0.f; }
    f32x4_t dhr[UTP];                        // dh_{t-1} of this wave's tiles, accumulator layout (rows g*4+r, unit c);
                                             // RB = 8: only [0], [1] are kept = rows gl*4 + hsel and gl*4 + 2 + hsel
    float dcc[UTP][2][2];                    // carried dc, pair layout [row rr][unit k]
#pragma unroll
    for (int j = 0; j < UTP; ++j) {
        dhr[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) { dcc[j][rr][0] = 0.f; dcc[j][rr][1] = 0.f; }
    }
    constexpr int NG = CELL == LAS_CELL_LSTM ? 4 : 1;
    // chunked dout (las_rnn_seq_bwd_db_chunked): the frame of step st may be read once the chunk of its producer row is complete
    int dhave = 1;                           // chunk 0: produced in front of this launch in stream order (no flag launch on the chain)
    auto wait_dout = [&](int st) __attribute__((always_inline)) {
        if constexpr (!CH) return;
        const int f = dir ? st : T - 1 - st, pr = f >> a.dshift;
        const int mm = pr < a.dTq - 1 - pr ? pr : a.dTq - 1 - pr;
        const int need = (mm >> a.dcp) + 1;
        if (need <= dhave) return;
        int budget = a.spin < (1 << 16) ? a.spin : (1 << 16);
        for (;;) {
            dhave = __hip_atomic_load(a.dflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (dhave >= need) break;
            if (--budget <= 0) { errflag = 1; dhave = 0x7fffffff; break; }      // sticky: later waits return at once
            __builtin_amdgcn_s_sleep(16);
        }
    };
    wait_dout(0);
    // operands of one step as packed pairs, ONE register set refilled for step s+1 right after step s consumed it
    unsigned n_do[UTP][2], n_g[NG][UTP][2], n_c[UTP][2], n_cn[UTP][2];
#pragma unroll
    for (int j = 0; j < UTP; ++j)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            n_do[j][rr] = dptr[rr][j * 8];
            if (CELL == LAS_CELL_LSTM) {
#pragma unroll
                for (int q = 0; q < NG; ++q) n_g[q][j][rr] = gptr[rr][(q * H + j * 16) / 2];
                n_c[j][rr] = cptr[rr][j * 8];
                n_cn[j][rr] = T > 1 ? cptr[rr][cst_[rr] + j * 8] : 0u;
            } else {
                n_g[0][j][rr] = optr[rr][j * 8];
                n_c[j][rr] = 0u; n_cn[j][rr] = 0u;
            }
        }
    int cur = 0;
    int pleft = PG ? a.pstep : 0;                   // steps until the next publication of dZ progress (PG)
#ifdef LAS_PROF
    const bool kprof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (kprof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#define KSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (kprof && s >= 200 && s < 208) a.dbg[8 + (s - 200) * 8 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KSTAMP(k)
#endif
    for (int s = 0; s < T; ++s) {
        KSTAMP(0);
        unsigned short* dzc = dzs + cur * RB * LDZ;
        unsigned sv_z[G][UTP][2];            // d(pre-activation) of this step as packed pairs (written to HBM after the exchange)
        // ---- gate backward for the own units -> own dG slice in LDS (bf16 pairs) and in registers
#pragma unroll
        for (int j = 0; j < UTP; ++j) {
            // dh of this tile: accumulator layout -> pair layout (swap two values with the neighbouring lane)
            float dhp[2][2];                                          // [row rr][unit k]
            if (RB == 16) {
                const float s0 = odd ? dhr[j][0] : dhr[j][2], s1 = odd ? dhr[j][1] : dhr[j][3];
                const float x0 = swap_lane_pair(s0), x1 = swap_lane_pair(s1);
                const float o0 = odd ? dhr[j][2] : dhr[j][0], o1 = odd ? dhr[j][3] : dhr[j][1];
                dhp[0][0] = odd ? x0 : o0; dhp[0][1] = odd ? o0 : x0; dhp[1][0] = odd ? x1 : o1; dhp[1][1] = odd ? o1 : x1;
            } else {      // the lane's row is r = odd*2 + hsel: own value dhr[odd], the neighbour wants dhr[!odd]
                const float own = odd ? dhr[j][1] : dhr[j][0];
                const float x = swap_lane_pair(odd ? dhr[j][0] : dhr[j][1]);
                dhp[0][0] = odd ? x : own; dhp[0][1] = odd ? own : x; dhp[1][0] = 0.f; dhp[1][1] = 0.f;
            }
            const int ucol = (w * UTP + j) * 16 + (c & ~1);          // column of the pair inside one gate block of the slice
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                float dz[G][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float dh = (k ? hi(n_do[j][rr]) : lo(n_do[j][rr])) + dhp[rr][k];
                    if (CELL == LAS_CELL_LSTM) {
                        const float gi = k ? hi(n_g[0][j][rr]) : lo(n_g[0][j][rr]);
                        const float gj = k ? hi(n_g[NG > 1 ? 1 : 0][j][rr]) : lo(n_g[NG > 1 ? 1 : 0][j][rr]);
                        const float gf = k ? hi(n_g[NG > 2 ? 2 : 0][j][rr]) : lo(n_g[NG > 2 ? 2 : 0][j][rr]);
                        const float go = k ? hi(n_g[NG > 3 ? 3 : 0][j][rr]) : lo(n_g[NG > 3 ? 3 : 0][j][rr]);
                        const float cprev = (s + 1 < T) ? (k ? hi(n_cn[j][rr]) : lo(n_cn[j][rr])) : 0.f;
                        const float tc = tanhx<true>(k ? hi(n_c[j][rr]) : lo(n_c[j][rr]));
                        const float dc = dcc[j][rr][k] + dh * go * (1.f - tc * tc);
                        dcc[j][rr][k] = dc * gf;
                        dz[0][k] = dc * gj * gi * (1.f - gi);
                        dz[G > 1 ? 1 : 0][k] = dc * gi * (1.f - gj * gj);
                        dz[G > 2 ? 2 : 0][k] = dc * cprev * gf * (1.f - gf);
                        dz[G > 3 ? 3 : 0][k] = dh * tc * go * (1.f - go);
                    } else {
                        const float h = k ? hi(n_g[0][j][rr]) : lo(n_g[0][j][rr]);
                        dz[0][k] = dh * (1.f - h * h);
                    }
                }
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const unsigned pk = f2bf2(dz[q][0], dz[q][1]);
                    *reinterpret_cast<unsigned*>(&dzc[(gl * 4 + odd * 2 + (RB == 16 ? rr : hsel)) * LDZ + q * UPM + ucol]) = pk;
                    sv_z[q][j][rr] = pk;
                    bsum[q][j][0] = fmaf(dz[q][0], vrow[rr], bsum[q][j][0]);
                    bsum[q][j][1] = fmaf(dz[q][1], vrow[rr], bsum[q][j][1]);
                }
            }
        }
        KSTAMP(1);
        gu32* gprev[2];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) { gprev[rr] = gptr[rr]; gptr[rr] += gst[rr]; optr[rr] += ost[rr]; dptr[rr] += dst[rr]; if (CELL == LAS_CELL_LSTM) cptr[rr] += cst_[rr]; }
        if (s + 1 < T) {     // operands of the next step fly under this step's MFMAs and exchange
            wait_dout(s + 1);
#pragma unroll
            for (int j = 0; j < UTP; ++j)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    n_do[j][rr] = dptr[rr][j * 8];
                    if (CELL == LAS_CELL_LSTM) {
#pragma unroll
                        for (int q = 0; q < NG; ++q) n_g[q][j][rr] = gptr[rr][(q * H + j * 16) / 2];
                        if constexpr (CH) n_c[j][rr] = cptr[rr][j * 8];      // (this variant re-loads c_t -- an L1 / L2 hit -- instead of rotating
                                                                            //  registers: the pinned rotation below survives only in the plain loop shape)
                        else {
                        n_c[j][rr] = n_cn[j][rr];
                        asm volatile("" : "+v"(n_c[j][rr]));        // the copy happens HERE: the old register of n_cn is free for the load below
                        }
                        // unconditional (a conditional load makes the compiler drain vmcnt at the join, which puts the HBM latency of
                        // this whole prefetch on the dependent chain); past the last frame the address is clamped and the value unused
                        n_cn[j][rr] = cptr[rr][((s + 2 < T) ? cst_[rr] : 0) + j * 8];
                    } else {
                        n_g[0][j][rr] = optr[rr][j * 8];
                    }
                }
        }
        KSTAMP(2);
        lds_barrier();
        KSTAMP(3);
        // ---- partial dh tiles (m, w, j) for every member m:  own dG slice [16 x KP] . W_hh^T rows of those units
        f32x4_t acc[P][UTP];                     // acc[mo]: partial tile of member (pm + mo) % P; acc[0] = own
#pragma unroll
        for (int m = 0; m < P; ++m)
#pragma unroll
            for (int j = 0; j < UTP; ++j) acc[m][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        u16x8_t av[KSP];                         // all A fragments requested up front: the MFMAs then wait on LDS once, not per pair
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) av[ks] = *reinterpret_cast<const u16x8_t*>(&dzc[(c & (RB - 1)) * LDZ + ks * 32 + g * 8]);
        __builtin_amdgcn_sched_barrier(0);
        // (measured: contracting the partners' tiles first and the own tile under the sends / the poll is slower, 1.31-1.36 vs 1.29 us)
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
#pragma unroll
            for (int m = 0; m < P; ++m)
#pragma unroll
                for (int j = 0; j < UTP; ++j) acc[m][j] = mfma_bf16_16x16x32(av[ks], wreg[(m * UTP + j) * KSP + ks], acc[m][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef LAS_PROF
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[P - 1][UTP - 1][3]));
#endif
        KSTAMP(4);
        if (s + 1 < T) {
            // ---- reduce-scatter: send the tiles other members own, add the ones they computed for this wave
            // byte offsets into this cluster's exchange buffer (< 2 GB): slot, [dst][src] region, (wave tile, lane) x 16 B
            const unsigned slot_off = (unsigned)(s & 1) * P * P * GPD * 8u;
            if constexpr (RB == 16) {
            const unsigned lane_off = ((unsigned)(w * UTP) * 64u + lane) * 16u;
#pragma unroll
            for (int mo = 1; mo < P; ++mo) {
                const int m = (pm + mo) % P;
                const unsigned dst_off = slot_off + (unsigned)(m * P + pm) * GPD * 8u + lane_off;
#pragma unroll
                for (int j = 0; j < UTP; ++j)     // the four partial sums of a lane travel as two bf16 pairs in ONE 16-byte double granule
                    granule16_store(xrs, dst_off + (unsigned)j * 1024u, (unsigned)(s + 1), f2bf2(acc[mo][j][0], acc[mo][j][1]),
                                    f2bf2(acc[mo][j][2], acc[mo][j][3]), local);
                __builtin_amdgcn_sched_barrier(0);
            }
            KSTAMP(5);
            constexpr int NGT = (P - 1) * UTP;
            u32x4_t xv[NGT];
            const unsigned in_off = slot_off + (unsigned)(pm * P) * GPD * 8u + lane_off;
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int src = (pm + 1 + n / UTP) % P;
                xv[n] = granule16_load(xrs, in_off + (unsigned)src * GPD * 8u + (unsigned)(n % UTP) * 1024u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NGT; ++n) ok &= xv[n].x == (unsigned)(s + 1) && xv[n].w == (unsigned)(s + 1);
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NGT; ++n) {
                    if (xv[n].x != (unsigned)(s + 1) || xv[n].w != (unsigned)(s + 1)) {
                        const int src = (pm + 1 + n / UTP) % P;
                        xv[n] = granule16_load(xrs, in_off + (unsigned)src * GPD * 8u + (unsigned)(n % UTP) * 1024u);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UTP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[0][j][r];
#pragma unroll
                    for (int mo = 0; mo < P - 1; ++mo) {
                        const unsigned pk = (r >> 1) ? xv[mo * UTP + j].z : xv[mo * UTP + j].y;
                        v += __uint_as_float((r & 1) ? (pk & 0xffff0000u) : (pk << 16));
                    }
                    dhr[j][r] = v;
                }
            } else {
            // RB = 8: a lane keeps rows r = hsel and r = 2 + hsel of its (gl, unit c) position -- exactly the two values the
            // same lane of the owning member needs -- so the two partial sums travel as ONE bf16 pair in an 8-byte granule.
            const unsigned lane_off = ((unsigned)(w * UTP) * 64u + lane) * 8u;
#pragma unroll
            for (int mo = 1; mo < P; ++mo) {
                const int m = (pm + mo) % P;
                const unsigned dst_off = slot_off + (unsigned)(m * P + pm) * GPD * 8u + lane_off;
#pragma unroll
                for (int j = 0; j < UTP; ++j)
                    granule8_store(xrs, dst_off + (unsigned)j * 512u, (unsigned)(s + 1),
                                   f2bf2(hsel ? acc[mo][j][1] : acc[mo][j][0], hsel ? acc[mo][j][3] : acc[mo][j][2]), local);
                __builtin_amdgcn_sched_barrier(0);
            }
            KSTAMP(5);
            constexpr int NGT = (P - 1) * UTP;
            u32x2_t xv[NGT];
            const unsigned in_off = slot_off + (unsigned)(pm * P) * GPD * 8u + lane_off;
#pragma unroll
            for (int n = 0; n < NGT; ++n) {
                const int src = (pm + 1 + n / UTP) % P;
                xv[n] = granule8_load(xrs, in_off + (unsigned)src * GPD * 8u + (unsigned)(n % UTP) * 512u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NGT; ++n) ok &= xv[n].x == (unsigned)(s + 1);
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NGT; ++n) {
                    if (xv[n].x != (unsigned)(s + 1)) {
                        const int src = (pm + 1 + n / UTP) % P;
                        xv[n] = granule8_load(xrs, in_off + (unsigned)src * GPD * 8u + (unsigned)(n % UTP) * 512u);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UTP; ++j) {
                float v0 = hsel ? acc[0][j][1] : acc[0][j][0], v1 = hsel ? acc[0][j][3] : acc[0][j][2];
#pragma unroll
                for (int mo = 0; mo < P - 1; ++mo) {
                    const unsigned pk = xv[mo * UTP + j].y;
                    v0 += __uint_as_float(pk << 16);
                    v1 += __uint_as_float(pk & 0xffff0000u);
                }
                dhr[j][0] = v0; dhr[j][1] = v1;
            }
            }
        }
        KSTAMP(6);
        // ---- d(pre-activation) of this step to HBM as packed bf16 pairs (never waited on)
#pragma unroll
        for (int j = 0; j < UTP; ++j)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int q = 0; q < G; ++q) {
#if LAS_PG_SC1_STORES
                    if constexpr (PG) __hip_atomic_store((unsigned*)(gprev[rr] + (q * H + j * 16) / 2), sv_z[q][j][rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
#endif
                    gprev[rr][(q * H + j * 16) / 2] = sv_z[q][j][rr];
                }
        KSTAMP(7);
        cur ^= 1;
        if constexpr (PG) {
            if (--pleft == 0 || s + 1 == T) {                        // every a.pstep steps (a countdown: `(s + 1) % a.pstep` was an integer division
                pleft = a.pstep;                                      //  per step); uniform: every wave of the workgroup takes the branch
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's dZ stores of the steps so far have reached the XCD's L2 ...
                __builtin_amdgcn_s_barrier();                          // ... and every other wave's of this member
                if (tid == 0) {
#if !LAS_PG_SC1_STORES
                    // plain stores + ONE write-back of the XCD's dirty L2 lines per publication (4-byte agent-scope stores are a fabric write
                    // each: the sweep ran 1.77 -> 2.09 ms with them)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
                    __hip_atomic_store(a.prog + (size_t)cl * P + pm, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
#ifdef LAS_PROF
    if (kprof) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#endif
    // bias gradient partials of this (tile, direction): sum the rows held by the lane pair and by the four row groups
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int j = 0; j < UTP; ++j) {
            float v0 = bsum[q][j][0], v1 = bsum[q][j][1];
            v0 += swap_lane_pair(v0);     v1 += swap_lane_pair(v1);
            v0 += __shfl_xor(v0, 16, 64); v1 += __shfl_xor(v1, 16, 64);
            v0 += __shfl_xor(v0, 32, 64); v1 += __shfl_xor(v1, 32, 64);
            if (lane < 16) a.bpart[(size_t)cl * GH + q * H + vw * (16 * UTP) + j * 16 + c] = odd ? v1 : v0;
        }
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
}

// db[dir][col] += sum over batch tiles of bpart[(tile*2 + dir)][col]
__global__ __launch_bounds__(256) void bias_finish_kernel(const float* __restrict__ bpart, int ntiles, int GH, float* db_fw, float* db_bw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * GH) return;
    const int dir = i / GH, col = i % GH;
    float* db = dir ? db_bw : db_fw;
    if (!db) return;
    float acc = 0.f;
    for (int t = 0; t < ntiles; ++t) acc += bpart[((size_t)t * 2 + dir) * GH + col];
    db[col] += acc;
}

// fragments for the K-split BPTT: (dir, vw = pm*4+w, m, j, ks):
//   B[k][n] = W_hh[unit n = m*UPM + (w*UTP + j)*16 + (lane&15)][gate col of k],  k = ks*32 + 8*(lane>>4) + e in the
//   member's slice: q = k / UPM, gate col = q*H + pm*UPM + (k % UPM)
// the pack kernels also clear the launch's exchange state (error flag, scratch rows, placement handshake, bias partials, granule
// tags): one launch instead of a pack + a memset node in front of every sweep
__device__ __forceinline__ void zero_region(uint4* z, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ void pack_whh_ks_body(const float* W0, const float* W1, int ldw, int H, int G, int P,
                                                 unsigned short* out, uint4* zero, long long zero16) {
    zero_region(zero, zero16);
    const int UTP = H / 64 / P, UPM = H / P, KP = G * UPM, KSP = KP / 32, NFR = P * UTP * KSP, NVW = 4 * P;
    const long long total = 2LL * NVW * NFR * 512;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long rest = idx >> 9;
        const int fi = (int)(rest % NFR);
        const int vw = (int)((rest / NFR) % NVW);
        const int dir = (int)(rest / ((long long)NFR * NVW));
        const float* W = dir ? W1 : W0;
        const int pm = vw / 4, w = vw % 4;
        // fragment slot `mo` of member pm holds the tile of member (pm + mo) % P: the kernel indexes its accumulators by the
        // RELATIVE member offset, a compile-time constant (slot 0 = own tile), instead of selecting by a run-time member id
        const int ks = fi % KSP, mj = fi / KSP, j = mj % UTP, m = (pm + mj / UTP) % P;
        const int unit = m * UPM + (w * UTP + j) * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8 + e;
        const int q = k / UPM, col = q * H + pm * UPM + (k % UPM);
        out[idx] = f2bf(W[(long long)unit * ldw + col]);
    }
}
__global__ __launch_bounds__(256) void pack_whh_ks_kernel(const float* W0, const float* W1, int ldw, int H, int G, int P,
                                                          unsigned short* out, uint4* zero, long long zero16) {
    pack_whh_ks_body(W0, W1, ldw, H, G, P, out, zero, zero16);
}

// W_hh (fp32 [H, G*H]) -> bf16 MFMA B-fragment order for 4*P "virtual waves" of 16*UTP units each.
//  fwd: frag (dir,vw,q,j,ks): B[k][n] = W[ks*32 + 8*(lane>>4)+e][q*H + vw*16*UTP + j*16 + (lane&15)]
//  bwd: frag (dir,vw,j,ks):   B[k][n] = W[vw*16*UTP + j*16 + (lane&15)][ks*32 + 8*(lane>>4)+e]   (= W^T)
__device__ __forceinline__ void pack_whh_body(const float* W0, const float* W1, int ldw, int H, int G,
                                              int bwd, int P, unsigned short* out, uint4* zero, long long zero16) {
    zero_region(zero, zero16);
    const int UTP = H / 64 / P, GH = G * H, NVW = 4 * P;
    const int KS = bwd ? GH / 32 : H / 32;
    const int NF = bwd ? UTP * KS : G * UTP * KS;
    const long long total = 2LL * H * GH;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long rest = idx >> 9;
        const int fi = (int)(rest % NF);
        const int vw = (int)((rest / NF) % NVW);
        const int dir = (int)(rest / ((long long)NF * NVW));
        const float* W = dir ? W1 : W0;
        const int ks = fi % KS, qj = fi / KS;
        int row, col;
        if (!bwd) {
            const int j = qj % UTP, q = qj / UTP;
            row = ks * 32 + (lane >> 4) * 8 + e;
            col = q * H + vw * (16 * UTP) + j * 16 + (lane & 15);
        } else {
            const int j = qj;
            row = vw * (16 * UTP) + j * 16 + (lane & 15);
            col = ks * 32 + (lane >> 4) * 8 + e;
        }
        out[idx] = f2bf(W[(long long)row * ldw + col]);
    }
}
__global__ __launch_bounds__(256) void pack_whh_kernel(const float* W0, const float* W1, int ldw, int H, int G,
                                                       int bwd, int P, unsigned short* out, uint4* zero, long long zero16) {
    pack_whh_body(W0, W1, ldw, H, G, bwd, P, out, zero, zero16);
}

// Round 5 (las_rnn_seq_prepare): the packs and exchange-state clears of up to SEQ_PREP_MAX sweeps in ONE launch, once per optimiser step
// -- a pack depends only on the weights, and the r4 timeline had one pack launch (8-11 us + a launch boundary) on the dependency chain in
// front of each of a step's eight sweeps.  blockIdx.y = job.
constexpr int SEQ_PREP_MAX = 8;
struct SeqPrepJob { const float* w0; const float* w1; int ldw, H, G, P, kind; unsigned short* out; uint4* zero; long long zero16; };   // kind 0 / 1: pack_whh (fwd / bwd), 2: pack_whh_ks
struct SeqPrepJobs { SeqPrepJob j[SEQ_PREP_MAX]; };
__global__ __launch_bounds__(256) void seq_prepare_kernel(SeqPrepJobs jobs) {
    const SeqPrepJob& q = jobs.j[blockIdx.y];
    if (q.kind == 2) pack_whh_ks_body(q.w0, q.w1, q.ldw, q.H, q.G, q.P, q.out, q.zero, q.zero16);
    else pack_whh_body(q.w0, q.w1, q.ldw, q.H, q.G, q.kind, q.P, q.out, q.zero, q.zero16);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool mfma_shape_ok(int H) { return H == 64 || H == 128 || H == 256 || H == 512; }

// cluster width: enough members that each keeps its W_hh slice resident (LDS + <=48 register fragments);
// LAS_SEQ_P(p) in `flags` overrides it (development / tests)
static int pick_cluster(int cell, int H, int flags) {
    int P;
    if (cell == LAS_CELL_LSTM) P = H >= 512 ? 8 : (H >= 256 ? 4 : (H >= 128 ? 2 : 1));
    else                       P = H >= 512 ? 4 : (H >= 256 ? 2 : 1);     // (r4: the tanh cell at H = 256 on two CUs serves the 8-row helper-wave / K-split kernels: BPTT 1.77 -> 1.25 us per step;
                                                                          //  r6: H = 512 -- run.sh's listener -- on four: two members had 64 W_hh fragments per wave, too many for the helper-wave
                                                                          //  kernel (32 in registers + 32 in LDS do not fit beside its rings), and swept on the plain kernel at 2.7 us per
                                                                          //  step; run.sh's step 27.5 -> 24.8 ms)
    const int v = (flags >> 8) & 0xf;
    if (v == 1 || v == 2 || v == 4 || v == 8) P = v;
    return P;
}

struct SeqWs { size_t pack, err, sink, xcc, bpart, xbuf, xbuf_per, total; };
static SeqWs seq_ws_layout(int cell, int H, int B) {
    const size_t G = cell == LAS_CELL_LSTM ? 4 : 1;
    SeqWs w;
    w.pack = 0;
    size_t o = (2 * G * H * H * sizeof(float) + 255) & ~(size_t)255;   // f32: W^T copy; bf16: packed fragments (half of it)
    w.err = o; o += 256;
    w.sink = o; o += ((size_t)(G * H + 64) * sizeof(float) + 255) & ~(size_t)255;
    const size_t ncl = (size_t)((B + 7) / 8) * 2;             // 8-row tiles: the finest tiling any kernel uses
    w.xcc = o; o += (ncl * 8 * sizeof(unsigned long long) + 255) & ~(size_t)255;      // [cluster][<= 8 members] handshake granules
    w.bpart = o; o += (ncl * G * H * sizeof(float) + 255) & ~(size_t)255;
    w.xbuf = o;
    // per cluster: [2 slots][max(all-gather: 8*G*H, K-split reduce-scatter: P*16*H with P <= 8)] granule words
    w.xbuf_per = 2 * ((size_t)8 * 16 * H > (size_t)8 * G * H ? (size_t)8 * 16 * H : (size_t)8 * G * H);
    o += ncl * w.xbuf_per * sizeof(unsigned long long);
    w.total = o + 256;
    return w;
}

extern "C" size_t las_rnn_seq_workspace_bytes(int cell, int prec, int H, int B) {
    (void)prec;
    const size_t a = seq_ws_layout(cell, H, B > 0 ? B : 1).total, b = las_rnn_seq_mf32_ws_bytes(cell, H);
    return a > b ? a : b;
}

template <typename K>
static int set_lds(K kern, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// which kernels can sweep 8-row tiles (lane-compacted duplicate MFMA rows)
template <int CELL, int UT, int P>
static constexpr bool rb8_ok(bool bwd) {
    if constexpr (P > 1) return bwd ? KsCfg<CELL, UT, P, 8>::OK : HwCfg<CELL, UT, P, 8>::OK;
    else return false;
}

template <int CELL, int UT, int P, int RT>
static int launch_bf16_rt(bool bwd, const RnnArgs& a0, int ntiles, hipStream_t st) {
    using C = RnnCfg<CELL, UT, P>;
    constexpr int FL = RT * C::HS_BYTES + 4 * C::LF * 1024, BL = RT * C::DP_BYTES + 4 * C::LFB * 1024;
    if constexpr (!C::OK || FL > 160 * 1024 || BL > 160 * 1024) {
        las_set_error("rnn_seq: cluster width %d / %d row tiles cannot keep W_hh resident for H=%d", P, RT, C::H);
        return -2;
    } else {
        RnnArgs a = a0;
        if (a.rb == 8 && !rb8_ok<CELL, UT, P>(bwd)) { las_set_error("rnn_seq: no 8-row kernel for this configuration"); return -1; }
        a.ncl = cdiv(ntiles, RT) * 2;                      // (tile group, direction) pairs
        a.ncl_pad = (a.ncl + 7) / 8 * 8;                   // members of a cluster share blockIdx % 8 (same XCD: speed only)
        if ((long long)a.ncl_pad * P > las_device_cus()) {     // every member must be co-resident (1 workgroup per CU)
            las_set_error("rnn_seq: %d workgroups exceed the %d compute units of this device", a.ncl_pad * P, las_device_cus());
            return -2;
        }
        dim3 grid(a.ncl_pad * P), blk(256 * RT);
        // one more workgroup per cluster (same XCD) that keeps the operands of the next steps in that XCD's L2
        const bool warm = P > 1 && a.warm && (long long)a.ncl_pad * (P + 1) <= las_device_cus();
        const dim3 gridw(a.ncl_pad * (P + (warm ? 1 : 0)));
        a.warm = warm ? 1 : 0;
        if (!bwd && RT == 1 && a.rb == 8) {
            if constexpr (HwCfg<CELL, UT, P, 8>::OK) {
                constexpr int HL = ks_lds(HwCfg<CELL, UT, P, 8>::LDS);    // (the CU to itself, as for the K-split BPTT kernel: 67 KB used,
                                                                          //  198 registers x 8 waves -- a 64 KB / 110-register GEMM workgroup fits next to it)
                static int attr = set_lds(rnn_seq_fwd_hw_kernel<CELL, UT, P, 8>, HL);
                if (attr != 0) { las_set_error("hipFuncSetAttribute(fwd hw) failed: %d", attr); return attr; }
                if (a.row_T) {
                    static int attr2 = set_lds(rnn_seq_fwd_hw_kernel<CELL, UT, P, 8, true>, HL);
                    if (attr2 != 0) { las_set_error("hipFuncSetAttribute(fwd hw, ragged) failed: %d", attr2); return attr2; }
                    hipLaunchKernelGGL((rnn_seq_fwd_hw_kernel<CELL, UT, P, 8, true>), gridw, dim3(512), HL, st, a);
                } else
                hipLaunchKernelGGL((rnn_seq_fwd_hw_kernel<CELL, UT, P, 8>), gridw, dim3(512), HL, st, a);
            }
        } else if (!bwd && RT == 1 && HwCfg<CELL, UT, P>::OK && !a0.no_helpers) {
            if constexpr (HwCfg<CELL, UT, P>::OK) {
                constexpr int HL = ks_lds(HwCfg<CELL, UT, P>::LDS);
                static int attr = set_lds(rnn_seq_fwd_hw_kernel<CELL, UT, P, 16>, HL);
                if (attr != 0) { las_set_error("hipFuncSetAttribute(fwd hw) failed: %d", attr); return attr; }
                hipLaunchKernelGGL((rnn_seq_fwd_hw_kernel<CELL, UT, P, 16>), gridw, dim3(512), HL, st, a);
            }
        } else if (!bwd) {
            static int attr = set_lds(rnn_seq_fwd_bf16_kernel<CELL, UT, P, RT>, FL);
            if (attr != 0) { las_set_error("hipFuncSetAttribute(fwd) failed: %d", attr); return attr; }
            hipLaunchKernelGGL((rnn_seq_fwd_bf16_kernel<CELL, UT, P, RT>), grid, blk, FL, st, a);
        } else if (P > 1 && KsCfg<CELL, UT, P>::OK && a.ks_packed) {
            if constexpr (P > 1 && KsCfg<CELL, UT, P>::OK) {
                // The K-split kernel keeps its weights in registers and needs 9-17 KB of LDS and half of the register file: other
                // kernels' workgroups (the side stream's weight-gradient GEMMs) WOULD be scheduled onto the same CU and share its
                // SIMDs, LDS and L1 with the dependent chain.  Asking for (nearly) the whole LDS keeps the CU to the sweep (ks_lds).
                if (a.rb == 8 && a.dflag && a.prog) {
                    constexpr int KZ = ks_lds(KsCfg<CELL, UT, P, 8>::DZ_BYTES);
                    static int attr = set_lds(rnn_seq_bwd_ks_kernel<CELL, UT, P, 8, true, true>, KZ);
                    if (attr != 0) { las_set_error("hipFuncSetAttribute(bwd ks) failed: %d", attr); return attr; }
                    hipLaunchKernelGGL((rnn_seq_bwd_ks_kernel<CELL, UT, P, 8, true, true>), grid, dim3(256), KZ, st, a);
                } else if (a.rb == 8 && a.dflag) {
                    constexpr int KZ = ks_lds(KsCfg<CELL, UT, P, 8>::DZ_BYTES);
                    static int attr = set_lds(rnn_seq_bwd_ks_kernel<CELL, UT, P, 8, true>, KZ);
                    if (attr != 0) { las_set_error("hipFuncSetAttribute(bwd ks) failed: %d", attr); return attr; }
                    hipLaunchKernelGGL((rnn_seq_bwd_ks_kernel<CELL, UT, P, 8, true>), grid, dim3(256), KZ, st, a);
                } else if (a.rb == 8) {
                    constexpr int KZ = ks_lds(KsCfg<CELL, UT, P, 8>::DZ_BYTES);
                    static int attr = set_lds(rnn_seq_bwd_ks_kernel<CELL, UT, P, 8>, KZ);
                    if (attr != 0) { las_set_error("hipFuncSetAttribute(bwd ks) failed: %d", attr); return attr; }
                    hipLaunchKernelGGL((rnn_seq_bwd_ks_kernel<CELL, UT, P, 8>), grid, dim3(256), KZ, st, a);
                } else {
                    constexpr int KZ = ks_lds(KsCfg<CELL, UT, P, 16>::DZ_BYTES);
                    static int attr = set_lds(rnn_seq_bwd_ks_kernel<CELL, UT, P, 16>, KZ);
                    if (attr != 0) { las_set_error("hipFuncSetAttribute(bwd ks) failed: %d", attr); return attr; }
                    hipLaunchKernelGGL((rnn_seq_bwd_ks_kernel<CELL, UT, P, 16>), grid, dim3(256), KZ, st, a);
                }
            }
        } else {
            static int attr = set_lds(rnn_seq_bwd_bf16_kernel<CELL, UT, P, RT>, BL);
            if (attr != 0) { las_set_error("hipFuncSetAttribute(bwd) failed: %d", attr); return attr; }
            hipLaunchKernelGGL((rnn_seq_bwd_bf16_kernel<CELL, UT, P, RT>), grid, blk, BL, st, a);
        }
        return 0;
    }
}

// row tiles per workgroup: as many waves per SIMD as registers and LDS allow (measured per configuration)
static int pick_rt(int cell, int H, int P, bool bwd, int ntiles) {
    // measured on MI355X: extra row-tile waves on the same CU lose (the step is bound by the CU's MFMA +
    // transcendental issue rate, not by latency) -> one row tile per workgroup, tiles spread over CUs
    int rt = 1;
    (void)cell; (void)H; (void)P; (void)bwd;
    return rt < ntiles ? rt : (ntiles < 1 ? 1 : ntiles);
}

template <int CELL, int UT, int P>
static int launch_bf16(bool bwd, const RnnArgs& a, hipStream_t st, bool query) {
    if (query) {       // bit 0: an 8-row-tile kernel exists for the direction; bit 1: the forward helper-wave kernel exists for 16-row tiles
        int r = rb8_ok<CELL, UT, P>(bwd) ? 1 : 0;
        if constexpr (P > 1) r |= (!bwd && HwCfg<CELL, UT, P, 16>::OK) ? 2 : 0;
        return r;                                                   // (bwd: bit 0 also means the chunk-aware K-split variant exists)
    }
    const int ntiles = cdiv(a.B, a.rb);
    int rt = pick_rt(CELL, UT * 64, P, bwd, ntiles);
    for (; rt >= 1; --rt) {
        int rc;
        rc = launch_bf16_rt<CELL, UT, P, 1>(bwd, a, ntiles, st);
        if (rc != -2) return rc;
    }
    return -2;
}

// query = true: 1 if this configuration has an 8-row-tile kernel for the direction, else 0 (nothing is launched)
static int dispatch_bf16(int cell, int P, bool bwd, const RnnArgs& a, hipStream_t st, bool query = false) {
    const int key = (cell == LAS_CELL_LSTM ? 1000 : 0) + (a.H / 64) * 10 + P;
    switch (key) {
        case 1011: return launch_bf16<LAS_CELL_LSTM, 1, 1>(bwd, a, st, query);
        case 1021: return launch_bf16<LAS_CELL_LSTM, 2, 1>(bwd, a, st, query);
        case 1022: return launch_bf16<LAS_CELL_LSTM, 2, 2>(bwd, a, st, query);
        case 1042: return launch_bf16<LAS_CELL_LSTM, 4, 2>(bwd, a, st, query);
        case 1044: return launch_bf16<LAS_CELL_LSTM, 4, 4>(bwd, a, st, query);
        case 1088: return launch_bf16<LAS_CELL_LSTM, 8, 8>(bwd, a, st, query);
        case 11:   return launch_bf16<LAS_CELL_RNN, 1, 1>(bwd, a, st, query);
        case 21:   return launch_bf16<LAS_CELL_RNN, 2, 1>(bwd, a, st, query);
        case 41:   return launch_bf16<LAS_CELL_RNN, 4, 1>(bwd, a, st, query);
        case 42:   return launch_bf16<LAS_CELL_RNN, 4, 2>(bwd, a, st, query);
        case 82:   return launch_bf16<LAS_CELL_RNN, 8, 2>(bwd, a, st, query);
        case 84:   return launch_bf16<LAS_CELL_RNN, 8, 4>(bwd, a, st, query);
        default:
            if (query) return 0;
            las_set_error("rnn_seq: no bf16 kernel for cell=%d H=%d P=%d", cell, a.H, P);
            return -2;
    }
}

static int check_common(const char* who, int cell, int prec, int B, int T, int H, const void* gates, const void* w0,
                        const void* w1, int ldw, const void* out, int ld_out, const void* cstate) {
    LAS_ARG(cell == LAS_CELL_RNN || cell == LAS_CELL_LSTM, "%s: bad cell %d", who, cell);
    LAS_ARG(prec == LAS_PREC_F32 || prec == LAS_PREC_BF16, "%s: bad prec %d", who, prec);
    LAS_ARG(B > 0 && T > 0 && H > 0 && H <= 256 * F32_UPT, "%s: bad dims B=%d T=%d H=%d", who, B, T, H);
    const int G = cell == LAS_CELL_LSTM ? 4 : 1;
    LAS_ARG(gates && w0 && w1 && out, "%s: null pointer", who);
    LAS_ARG(ldw >= G * H && ld_out >= 2 * H, "%s: leading dimension too small", who);
    LAS_ARG(cell == LAS_CELL_RNN || cstate, "%s: lstm needs cstate", who);
    return 0;
}

// bf16 path: pack W_hh, zero the exchange granules, launch the (clustered) persistent sweep.  Every member of every
// cluster has to be resident at once (one workgroup per CU), so a batch with more row tiles than the device's CUs can
// hold is swept in row chunks, one launch after the other on the same stream, each with its own exchange region.
static int run_bf16(bool bwd, int cell, const RnnArgs& a_in, const float* w0, const float* w1, int ldw, void* ws, size_t ws_bytes,
                    int flags, hipStream_t st, float* db_fw = nullptr, float* db_bw = nullptr, int* db_done = nullptr) {
    const int G = cell == LAS_CELL_LSTM ? 4 : 1, H = a_in.H, B = a_in.B, T = a_in.T;
    const SeqWs L = seq_ws_layout(cell, H, B);
    LAS_ARG(ws && ws_bytes >= L.total, "las_rnn_seq: workspace too small (%zu < %zu)", ws_bytes, L.total);
    char* base = (char*)ws;
    int P = pick_cluster(cell, H, flags);
    int rc = -2;
    for (int attempt = 0; attempt < 5 && rc == -2; ++attempt) {
        if (attempt > 0) {                               // fall back to the next narrower cluster
            if (P == 1) break;
            P >>= 1;
        }
        RnnArgs a = a_in;
        a.wpack = base + L.pack;
        a.err = (int*)(base + L.err);
        a.sink = (float*)(base + L.sink);
        a.sink16 = (unsigned short*)(base + L.sink);
        a.force_agent = (flags & LAS_SEQ_AGENT_GRANULES) ? 1 : 0;
        a.no_helpers = (flags & LAS_SEQ_NO_HELPER_WAVES) ? 1 : 0;
        a.ks_packed = (bwd && P > 1 && !(flags & LAS_SEQ_NO_KSPLIT)) ? 1 : 0;
        // (err, sink, and for clusters the handshake / bias partials / granule tags: multiples of 256 bytes by construction)
        uint4* zr = (uint4*)(base + L.err);
        const long long z16 = (long long)(((P > 1 ? L.total : L.xbuf) - L.err) / 16);
        // LAS_SEQ_PREPARED: las_rnn_seq_prepare left this cluster width's pack and a clean exchange state in `ws` (first attempt only:
        // a narrower fall-back cluster packs for itself)
        if ((flags & LAS_SEQ_PREPARED) && attempt == 0) {}
        else if (a.ks_packed) hipLaunchKernelGGL(pack_whh_ks_kernel, dim3(cdiv(2LL * G * H * H, 256 * 4)), dim3(256), 0, st, w0, w1, ldw, H, G, P,
                                            (unsigned short*)a.wpack, zr, z16);
        else hipLaunchKernelGGL(pack_whh_kernel, dim3(cdiv(2LL * G * H * H, 256 * 4)), dim3(256), 0, st, w0, w1, ldw, H, G, bwd ? 1 : 0, P,
                                (unsigned short*)a.wpack, zr, z16);
        LAS_LAUNCHED();
        // row tiles per launch: clusters (tile, direction) are padded to a multiple of 8 workgroups per member
        int max_tiles = (las_device_cus() / P / 8) * 8 / 2;
        if (max_tiles < 1) { las_set_error("rnn_seq: cluster width %d does not fit %d compute units", P, las_device_cus()); rc = -2; continue; }
        // 8-row tiles (twice the CUs, half the per-lane work of a dependent step) while the whole batch still fits one launch
        const bool k8 = (bwd ? a.ks_packed != 0 : !a.no_helpers) && !(flags & LAS_SEQ_ROWS16) && cdiv(B, 8) <= max_tiles &&
                        (dispatch_bf16(cell, P, bwd, a, st, true) & 1);
        const int RBk = k8 ? 8 : 16;
        a.rb = RBk;
        const int ntiles = cdiv(B, RBk);
        const size_t per_cl = L.xbuf_per;                // granule words per cluster
        rc = 0;
        for (int tile0 = 0; tile0 < ntiles && rc == 0; tile0 += max_tiles) {
            const int b0 = tile0 * RBk, rows = (B - b0) < max_tiles * RBk ? (B - b0) : max_tiles * RBk;
            RnnArgs c = a;
            c.B = rows;
            c.gates16 = a.gates16 + (size_t)b0 * T * 2 * G * H;
            c.out16 = a.out16 + (size_t)b0 * a.obs;
            if (a.cstate16) c.cstate16 = a.cstate16 + (size_t)b0 * T * 2 * H;
            if (a.dout16) c.dout16 = a.dout16 + (size_t)b0 * a.dobs;
            c.xbuf = (unsigned long long*)(base + L.xbuf) + (size_t)tile0 * 2 * per_cl;
            c.xcc = (unsigned long long*)(base + L.xcc) + (size_t)tile0 * 2 * 8;
            c.bpart = (float*)(base + L.bpart) + (size_t)tile0 * 2 * G * H;
            c.ncl = c.ncl_pad = 0;                           // set per launch
            rc = dispatch_bf16(cell, P, bwd, c, st);
        }
        if (rc == 0 && a.ks_packed && (db_fw || db_bw)) {    // the K-split kernel left per-tile column sums of dz: finish the bias gradient
            hipLaunchKernelGGL(bias_finish_kernel, dim3(cdiv(2 * G * H, 256)), dim3(256), 0, st, (const float*)(base + L.bpart), ntiles, G * H,
                               db_fw, db_bw);
            LAS_LAUNCHED();
            if (db_done) *db_done = 1;
        }
    }
    return rc;
}

extern "C" int las_rnn_seq_prepare(const las_seq_prepare_desc* descs, int n, void* stream) {
    LAS_ARG(descs && n > 0, "las_rnn_seq_prepare: no descriptors");
    for (int i0 = 0; i0 < n; i0 += SEQ_PREP_MAX) {
        const int m = n - i0 < SEQ_PREP_MAX ? n - i0 : SEQ_PREP_MAX;
        SeqPrepJobs jobs;
        long long most = 0;
        for (int i = 0; i < m; ++i) {
            const las_seq_prepare_desc& d = descs[i0 + i];
            LAS_ARG(d.cell == LAS_CELL_RNN || d.cell == LAS_CELL_LSTM, "las_rnn_seq_prepare: bad cell %d", d.cell);
            LAS_ARG(mfma_shape_ok(d.H) && d.B > 0 && d.whh_fw && d.whh_bw && d.ws, "las_rnn_seq_prepare: bad descriptor %d (H = %d: only the speed mode's clustered sweeps take a prepared workspace)", i0 + i, d.H);
            const int G = d.cell == LAS_CELL_LSTM ? 4 : 1;
            LAS_ARG(d.ldw >= G * d.H, "las_rnn_seq_prepare: leading dimension too small");
            const SeqWs L = seq_ws_layout(d.cell, d.H, d.B);
            LAS_ARG(d.ws_bytes >= L.total, "las_rnn_seq_prepare: workspace too small (%zu < %zu)", d.ws_bytes, L.total);
            const int P = pick_cluster(d.cell, d.H, d.flags);
            const bool ks = d.bwd && P > 1 && !(d.flags & LAS_SEQ_NO_KSPLIT);
            char* base = (char*)d.ws;
            SeqPrepJob& q = jobs.j[i];
            q.w0 = d.whh_fw; q.w1 = d.whh_bw; q.ldw = d.ldw; q.H = d.H; q.G = G; q.P = P; q.kind = ks ? 2 : (d.bwd ? 1 : 0);
            q.out = (unsigned short*)(base + L.pack); q.zero = (uint4*)(base + L.err);
            q.zero16 = (long long)(((P > 1 ? L.total : L.xbuf) - L.err) / 16);
            const long long work = 2LL * G * d.H * d.H / 4 + q.zero16;
            if (work > most) most = work;
        }
        long long nb = cdiv(most, 256 * 8);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(seq_prepare_kernel, dim3((unsigned)nb, m), dim3(256), 0, (hipStream_t)stream, jobs);
        LAS_LAUNCHED();
    }
    return 0;
}

static void seq_common_args(RnnArgs& a, int flags, int* status, int code) {
    a.row_T = nullptr;
    a.dbg = nullptr; a.xbuf = nullptr; a.xcc = nullptr; a.bpart = nullptr; a.force_agent = 0; a.err = nullptr; a.sink = nullptr;
    a.ncl = a.ncl_pad = 0; a.ks_packed = 0; a.no_helpers = 0; a.rb = 16;
    a.warm = (flags & LAS_SEQ_NO_WARMERS) ? 0 : 1;
    a.xflag = nullptr; a.xsc = 0; a.dflag = nullptr; a.dcp = 0; a.dTq = 0; a.dshift = 0; a.prog = nullptr; a.pstep = 0;
    const int lg = (flags >> 16) & 0x1f;                    // LAS_SEQ_SPIN_LOG2(n): bound of the exchange spins = 2^n polls
    a.spin = lg ? (1 << lg) : LAS_SPIN_BUDGET_DEFAULT;
    a.status = status; a.status_code = code;
    a.announce = (flags >> 21) & 0x3ff;
}

// 1 if las_rnn_seq_fwd would serve (cell, prec, B, H, flags) with ONE launch of the helper-wave kernel -- the only kernel that
// understands x-projection chunks (las_rnn_seq_fwd_chunked)
static bool handover_room(int B, int rows_per_tile, int P) {
    const int ncl_pad = (cdiv(B, rows_per_tile) * 2 + 7) / 8 * 8;
    return (long long)ncl_pad * (P + 1) * 2 <= las_device_cus();
}

extern "C" int las_rnn_seq_fwd_chunks_ok(int cell, int prec, int B, int H, int flags) {
    if (prec != LAS_PREC_BF16 || !mfma_shape_ok(H) || B <= 0 || (flags & LAS_SEQ_NO_HELPER_WAVES)) return 0;
    const int P = pick_cluster(cell, H, flags);
    if (P <= 1) return 0;
    RnnArgs a; a.H = H; a.B = B;
    const int q = dispatch_bf16(cell, P, false, a, nullptr, true);
    const int max_tiles = (las_device_cus() / P / 8) * 8 / 2;
    if (max_tiles < 1) return 0;
    const bool k8 = !(flags & LAS_SEQ_ROWS16) && cdiv(B, 8) <= max_tiles && (q & 1);
    // The chunks' producers run WHILE the sweep holds its compute units (whole CUs: 159 KB of LDS each): the hand-over only makes sense
    // while the sweep -- clusters + warmers -- leaves at least half of the machine to them.  r4, B = 144 / 192 at H = 256 (180 / 240 of 256
    // CUs): the x-projection chunks crawl on the few free CUs, 52 ms per step instead of 26 / the sweep runs into its chunk-wait bound.
    if (!handover_room(B, k8 ? 8 : 16, P)) return 0;
    if (k8) return 1;
    return ((q & 2) && cdiv(B, 16) <= max_tiles) ? 1 : 0;
}

// rows of different lengths (las_rnn_seq_fwd_rows): the 8-row helper-wave kernel, whole batch in one launch
extern "C" int las_rnn_seq_fwd_rows_ok(int cell, int prec, int B, int H, int flags) {
    if (prec != LAS_PREC_BF16 || !mfma_shape_ok(H) || B <= 0 || (flags & (LAS_SEQ_NO_HELPER_WAVES | LAS_SEQ_ROWS16))) return 0;
    const int P = pick_cluster(cell, H, flags);
    if (P <= 1) return 0;
    RnnArgs a; a.H = H; a.B = B;
    const int q = dispatch_bf16(cell, P, false, a, nullptr, true);
    const int max_tiles = (las_device_cus() / P / 8) * 8 / 2;
    return (max_tiles >= 1 && cdiv(B, 8) <= max_tiles && (q & 1)) ? 1 : 0;
}

extern "C" int las_rnn_seq_io_dtype(int cell, int prec, int H) {
    (void)cell;
    return (prec == LAS_PREC_BF16 && mfma_shape_ok(H)) ? LAS_DT_BF16 : LAS_DT_F32;
}

static void bind_tensors(RnnArgs& a, void* gates, void* out, void* cstate, const void* dout) {
    a.gates = (float*)gates; a.out = (float*)out; a.cstate = (float*)cstate; a.dout = (const float*)dout;
    a.gates16 = (unsigned short*)gates; a.out16 = (unsigned short*)out; a.cstate16 = (unsigned short*)cstate;
    a.dout16 = (const unsigned short*)dout; a.sink16 = nullptr;
}

extern "C" int las_rnn_seq_fwd(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                               const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                               void* cstate, float forget_bias, int flags, int* status, void* ws, size_t ws_bytes, void* stream) {
    return las_rnn_seq_fwd_chunked(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, forget_bias, flags,
                                   status, nullptr, 0, ws, ws_bytes, stream);
}

static int rnn_seq_fwd_impl(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                            const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                            void* cstate, float forget_bias, int flags, int* status, const int* chunk_flag, int chunk_steps,
                            const int* row_T, void* ws, size_t ws_bytes, void* stream);
extern "C" int las_rnn_seq_fwd_chunked(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                       const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                                       void* cstate, float forget_bias, int flags, int* status, const int* chunk_flag, int chunk_steps,
                                       void* ws, size_t ws_bytes, void* stream) {
    return rnn_seq_fwd_impl(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, forget_bias, flags, status,
                            chunk_flag, chunk_steps, nullptr, ws, ws_bytes, stream);
}
extern "C" int las_rnn_seq_fwd_rows(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                    const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                                    void* cstate, float forget_bias, int flags, int* status, const int* row_T,
                                    void* ws, size_t ws_bytes, void* stream) {
    LAS_ARG(row_T, "las_rnn_seq_fwd_rows: null row_T");
    return rnn_seq_fwd_impl(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, forget_bias, flags, status,
                            nullptr, 0, row_T, ws, ws_bytes, stream);
}
static int rnn_seq_fwd_impl(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                            const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                            void* cstate, float forget_bias, int flags, int* status, const int* chunk_flag, int chunk_steps,
                            const int* row_T, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_common("las_rnn_seq_fwd", cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, cstate)) return rc;
    hipStream_t st = (hipStream_t)stream;
    RnnArgs a;
    a.B = B; a.T = T; a.H = H; a.whh[0] = whh_fw; a.whh[1] = whh_bw; a.ldw = ldw;
    a.ld_out = ld_out; a.obs = out_bstride;
    bind_tensors(a, gates, out, cstate, nullptr);
    a.ld_dout = 0; a.dobs = 0; a.fb = forget_bias; a.wpack = ws;
    seq_common_args(a, flags, status, LAS_SEQ_STATUS_FWD_TIMEOUT);
    LAS_ARG(!chunk_flag || (chunk_steps > 0 && las_rnn_seq_fwd_chunks_ok(cell, prec, B, H, flags)),
            "las_rnn_seq_fwd_chunked: this configuration is not served by the kernel that waits for x-projection chunks");
    a.xflag = chunk_flag; a.xsc = chunk_steps;
    a.row_T = row_T;
    LAS_ARG(!row_T || las_rnn_seq_fwd_rows_ok(cell, prec, B, H, flags), "las_rnn_seq_fwd_rows: rows of different lengths are served by the 8-row "
            "helper-wave kernel only (speed mode, clustered, the whole batch in one launch): ask las_rnn_seq_fwd_rows_ok");
#ifdef LAS_PROF
    if (const char* e = getenv("LAS_DBG_PTR")) a.dbg = (long long*)strtoull(e, nullptr, 0);   // development build only
#endif
    if (prec == LAS_PREC_BF16 && mfma_shape_ok(H)) {
        LAS_ARG(ld_out % 4 == 0 && out_bstride % 4 == 0 && (((uintptr_t)gates | (uintptr_t)out | (uintptr_t)cstate) & 15) == 0,
                "las_rnn_seq_fwd: bf16 tensors must be 16-byte aligned with pitches that are multiples of 4");
        if (int rc = run_bf16(false, cell, a, whh_fw, whh_bw, ldw, ws, ws_bytes, flags, st)) return rc;
    } else if (prec == LAS_PREC_F32 && !(flags & LAS_SEQ_F32_VALU) && las_rnn_seq_mf32_ok(cell, H)) {
        return las_rnn_seq_mf32_run(false, cell, a, ws, ws_bytes, flags, st);
    } else {
        const size_t lds = (size_t)H * F32_BT * sizeof(float);
        dim3 grid(cdiv(B, F32_BT), 2);
        if (cell == LAS_CELL_LSTM) hipLaunchKernelGGL(rnn_seq_fwd_f32_kernel<LAS_CELL_LSTM>, grid, dim3(256), lds, st, a);
        else                       hipLaunchKernelGGL(rnn_seq_fwd_f32_kernel<LAS_CELL_RNN>, grid, dim3(256), lds, st, a);
    }
    LAS_LAUNCHED();
    return 0;
}

extern "C" int las_rnn_seq_bwd_chunks_ok(int cell, int prec, int B, int H, int flags) {
    if (prec != LAS_PREC_BF16 || !mfma_shape_ok(H) || B <= 0 || (flags & (LAS_SEQ_NO_KSPLIT | LAS_SEQ_ROWS16))) return 0;
    const int P = pick_cluster(cell, H, flags);
    if (P <= 1) return 0;
    RnnArgs a; a.H = H; a.B = B;
    const int q = dispatch_bf16(cell, P, true, a, nullptr, true);
    const int max_tiles = (las_device_cus() / P / 8) * 8 / 2;
    return (max_tiles >= 1 && (q & 1) && cdiv(B, 8) <= max_tiles && handover_room(B, 8, P)) ? 1 : 0;      // served by ONE launch of the 8-row K-split kernel, with room beside it
}

extern "C" int las_rnn_seq_bwd(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                               const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                               const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                               float forget_bias, int flags, int* status, void* ws, size_t ws_bytes, void* stream) {
    return las_rnn_seq_bwd_db(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, dout, ld_dout,
                              dout_bstride, forget_bias, nullptr, nullptr, flags, status, ws, ws_bytes, stream);
}

extern "C" int las_rnn_seq_bwd_db(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                  const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                                  const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                                  float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                                  void* ws, size_t ws_bytes, void* stream) {
    return las_rnn_seq_bwd_db_chunked(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, dout, ld_dout,
                                      dout_bstride, forget_bias, dbias_fw, dbias_bw, flags, status, nullptr, 0, 0, ws, ws_bytes, stream);
}

static int rnn_seq_bwd_db_impl(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                               const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                               const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                               float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                               const int* chunk_flag, int chunk_rows, int n_rows, int* progress, int progress_steps,
                               void* ws, size_t ws_bytes, void* stream);
extern "C" int las_rnn_seq_bwd_db_chunked(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                          const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                                          const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                                          float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                                          const int* chunk_flag, int chunk_rows, int n_rows, void* ws, size_t ws_bytes, void* stream) {
    return rnn_seq_bwd_db_impl(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, dout, ld_dout, dout_bstride,
                               forget_bias, dbias_fw, dbias_bw, flags, status, chunk_flag, chunk_rows, n_rows, nullptr, 0, ws, ws_bytes, stream);
}
// words a las_rnn_seq_bwd_db_progress launch publishes (one per cluster member), 0 if the configuration has no progress-publishing kernel
extern "C" int las_rnn_seq_bwd_progress_words(int cell, int prec, int B, int H, int flags) {
    if (!las_rnn_seq_bwd_chunks_ok(cell, prec, B, H, flags)) return 0;
    return 2 * cdiv(B, 8) * pick_cluster(cell, H, flags);
}
extern "C" int las_rnn_seq_bwd_db_progress(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                           const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                                           const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                                           float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                                           const int* chunk_flag, int chunk_rows, int n_rows, int* progress, int progress_steps,
                                           void* ws, size_t ws_bytes, void* stream) {
    LAS_ARG(chunk_flag && progress && progress_steps > 0 && las_rnn_seq_bwd_progress_words(cell, prec, B, H, flags) > 0,
            "las_rnn_seq_bwd_db_progress: needs a chunked upstream gradient and a configuration the progress-publishing kernel serves");
    return rnn_seq_bwd_db_impl(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate, dout, ld_dout, dout_bstride,
                               forget_bias, dbias_fw, dbias_bw, flags, status, chunk_flag, chunk_rows, n_rows, progress, progress_steps, ws, ws_bytes, stream);
}
static int rnn_seq_bwd_db_impl(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                               const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                               const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                               float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                               const int* chunk_flag, int chunk_rows, int n_rows, int* progress, int progress_steps,
                               void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_common("las_rnn_seq_bwd", cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, cstate)) return rc;
    LAS_ARG(dout && ld_dout >= 2 * H, "las_rnn_seq_bwd: bad dout");
    LAS_ARG(prec != LAS_PREC_BF16 || !mfma_shape_ok(H) ||
            (ld_out % 4 == 0 && out_bstride % 4 == 0 && ld_dout % 4 == 0 && dout_bstride % 4 == 0 &&
             (((uintptr_t)gates | (uintptr_t)out | (uintptr_t)cstate | (uintptr_t)dout) & 15) == 0),
            "las_rnn_seq_bwd: bf16 tensors must be 16-byte aligned with pitches that are multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    const int G = cell == LAS_CELL_LSTM ? 4 : 1;
    RnnArgs a;
    a.B = B; a.T = T; a.H = H; a.whh[0] = whh_fw; a.whh[1] = whh_bw; a.ldw = ldw;
    a.ld_out = ld_out; a.obs = out_bstride;
    bind_tensors(a, gates, const_cast<void*>(out), const_cast<void*>(cstate), dout);
    a.ld_dout = ld_dout; a.dobs = dout_bstride; a.fb = forget_bias; a.wpack = ws;
    seq_common_args(a, flags, status, LAS_SEQ_STATUS_BWD_TIMEOUT);
    LAS_ARG(!chunk_flag || (chunk_rows > 0 && (chunk_rows & (chunk_rows - 1)) == 0 && (n_rows == T || n_rows == (T + 1) / 2) &&
                            las_rnn_seq_bwd_chunks_ok(cell, prec, B, H, flags)),
            "las_rnn_seq_bwd_db_chunked: bad chunk geometry, or a configuration the chunk-aware kernel does not serve");
    a.dflag = chunk_flag; a.dTq = n_rows; a.dshift = (chunk_flag && n_rows != T) ? 1 : 0;
    for (int c = chunk_rows; c > 1; c >>= 1) ++a.dcp;
    a.prog = progress; a.pstep = progress_steps;
#ifdef LAS_PROF
    if (const char* e = getenv("LAS_DBG_PTR")) a.dbg = (long long*)strtoull(e, nullptr, 0);   // development build only
#endif
    const bool bf = prec == LAS_PREC_BF16 && mfma_shape_ok(H);
    if (bf) {
        int db_done = 0;
        if (int rc = run_bf16(true, cell, a, whh_fw, whh_bw, ldw, ws, ws_bytes, flags, st, dbias_fw, dbias_bw, &db_done)) return rc;
        if (db_done) return 0;
    } else if (prec == LAS_PREC_F32 && !(flags & LAS_SEQ_F32_VALU) && las_rnn_seq_mf32_ok(cell, H)) {
        if (int rc = las_rnn_seq_mf32_run(true, cell, a, ws, ws_bytes, flags, st)) return rc;
    } else {
        LAS_ARG(ws && ws_bytes >= (size_t)2 * G * H * H * sizeof(float), "las_rnn_seq_bwd: workspace too small");
        hipLaunchKernelGGL(transpose_whh_kernel, dim3(cdiv(2LL * G * H * H, 256 * 4)), dim3(256), 0, st, whh_fw, whh_bw, ldw,
                           H, G * H, (float*)ws);
        LAS_LAUNCHED();
        const size_t lds = (size_t)G * H * F32_BT * sizeof(float);
        if (lds > 64 * 1024) {
            static int a1 = set_lds(rnn_seq_bwd_f32_kernel<LAS_CELL_LSTM>, 160 * 1024 - 1024);
            static int a2 = set_lds(rnn_seq_bwd_f32_kernel<LAS_CELL_RNN>, 160 * 1024 - 1024);
            LAS_ARG(a1 == 0 && a2 == 0 && lds <= 159 * 1024, "las_rnn_seq_bwd: H too large for the fp32 kernel");
        }
        dim3 grid(cdiv(B, F32_BT), 2);
        if (cell == LAS_CELL_LSTM) hipLaunchKernelGGL(rnn_seq_bwd_f32_kernel<LAS_CELL_LSTM>, grid, dim3(256), lds, st, a);
        else                       hipLaunchKernelGGL(rnn_seq_bwd_f32_kernel<LAS_CELL_RNN>, grid, dim3(256), lds, st, a);
    }
    LAS_LAUNCHED();
    // kernels that do not accumulate the bias gradient themselves: column sums of the finished d(pre-activation)
    for (int d = 0; d < 2; ++d) {
        float* db = d ? dbias_bw : dbias_fw;
        if (!db) continue;
        const void* gd = bf ? (const void*)((const unsigned short*)gates + (size_t)d * G * H) : (const void*)((const float*)gates + (size_t)d * G * H);
        if (int rc = las_colsum_dt(gd, bf ? LAS_DT_BF16 : LAS_DT_F32, B * T, G * H, 2 * G * H, 1.f, db, ws, ws_bytes, stream)) return rc;
    }
    return 0;
}
