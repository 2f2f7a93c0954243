// rnn_seq_f32.hip -- K2 / K2b in the PARITY mode (fp32 storage and arithmetic): the recurrent sweep of one bidirectional layer on the
// matrix cores, exact fp32 (round 4).
//
// Replaces, for H in {64, 128, 256, 512}, the round-1 kernels rnn_seq_{fwd,bwd}_f32_kernel (rnn_seq.hip: one workgroup per
// (direction, 8 batch rows), W_hh re-read from L2 at every step: 33 / 123 us per dependent step at H = 256 -- 547 of the parity
// mode's 662 ms per B = 48 / T = 1274 train step).  Same reference lines: tf.nn.bidirectional_dynamic_rnn's per-step while-loop
// (las/layers.py:49-53) over BasicRNNCell (las/layers.py:31) / BasicLSTMCell, every tensor fp32 (las/layers.py:25,52).
//
// v_mfma_f32_16x16x4_f32 is a k-ordered fp32 fma chain (one rounding per product, no wider accumulator), so the parity rows keep
// their meaning; it runs at the fp32 VALU's peak rate (64 FLOP / clk / SIMD), i.e. a dependent step of one (direction, 16-row tile)
// is 2 x 16 x H x G H flops = 32768 / P cycles of MFMA issue at H = 256 on P compute units.  Hence wide clusters: every member owns
// 64 gate columns (lstm: 16 hidden units x 4 gates, rnn: 64 units), P = G H / 64 members per (direction, tile) -- 16 at H = 256
// lstm, 96 CUs for B = 48 -- each keeping ITS W_hh slice in registers for the whole sweep (H / 4 VGPRs per lane).
//
//   forward   the four waves of a member split K = H: wave w contracts k in [w H/4, (w+1) H/4) for all 64 columns (4 accumulator
//             tiles), the partial tiles meet in LDS, thread (row, j) finishes the member's (row, unit) elements (gate math in
//             accurate transcendentals), h is published to the partners as tagged 16-byte granules {tag, h_u, h_u+4, tag} -- the
//             two values one MFMA lane needs for two consecutive k-steps: a wave polls its own K quarter straight into its A
//             operands (round 5: no LDS h tile, one barrier per step; 2.71 -> 2.37 us per step at H = 256).
//   BPTT      K-split like the speed mode's: a member contracts ITS 64 columns of d(pre-activation) against W_hh^T for ALL H
//             units (wave w: unit tiles [w H/64, (w+1) H/64)), keeps the partial tile(s) of its own units and sends the others
//             to their owners; a thread pair (rows 2r, 2r+1) shares the partners' granules, half each, fixed order.
// Exchange transport, placement handshake, bounded polls and the status word are the speed mode's (rnn_seq_args.h).
// No atomics; every sum has a fixed order: bit-reproducible.
#include "rnn_seq_args.h"
#ifndef LAS_KS_SHARE_CU
#define LAS_KS_SHARE_CU 0        // 1: timing experiments -- other kernels may share the sweep's CUs (see las_rnn_seq_mf32_run)
#endif

namespace {

template <int CELL, int H_>
struct F32Cfg {
    static constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1, H = H_, GH = G * H;
    static constexpr int UPC = 64 / G;             // hidden units a member owns
    static constexpr int P = H / UPC;              // members per (direction, 16-row tile)
    static constexpr int KS = H / 16;              // forward: k-steps (of 4) per wave
    static constexpr int GPM = 16 * UPC / 2;       // forward: 16-byte granules (2 fp32) a member publishes per step
    static constexpr int TPM = UPC / 16;           // 16-unit tiles a member owns (lstm 1, rnn 4)
    static constexpr int TW = H / 64;              // BPTT: unit tiles per wave
    static constexpr int PT = 80;                  // pitch of one accumulator register plane [lk][li] in LDS (16 mod 64 banks)
};

// column (0 .. G H) of position j of the member's n-tile nt:  lstm: gate nt of unit pm * 16 + j;  rnn: unit pm * 64 + nt * 16 + j
template <int CELL, int H>
__device__ __forceinline__ int col_of(int pm, int nt, int j) {
    return CELL == LAS_CELL_LSTM ? nt * H + pm * 16 + j : pm * 64 + nt * 16 + j;
}
template <int CELL, int H>
__global__ __launch_bounds__(256, 1) void rnn_seq_fwd_mf32_kernel(RnnArgs a) {
    using C = F32Cfg<CELL, H>;
    constexpr int G = C::G, GH = C::GH, P = C::P, KS = C::KS, GPM = C::GPM, PT = C::PT, UPC = C::UPC;
    __shared__ __attribute__((aligned(16))) float part2[2][4 * 4 * 4 * PT];     // [step parity][wave][n-tile][acc register i][lk][li]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int T = a.T, B = a.B;
    const int cg = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;
    if (cg >= a.ncl) return;
    const int dir = cg & 1, tile = cg >> 1, b0 = tile * 16;
    unsigned long long* xb = a.xbuf + (size_t)cg * 2 * P * GPM * 2;             // [2 slots][P][GPM] 16-byte granules
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(xb);
    int errflag = 0;
    const bool local = (P > 1 && !a.force_agent) ? cluster_same_xcd(a.xcc + (size_t)cg * 32, pm, P, tid, &errflag, a.spin) : false;

    // this wave's slice of W_hh: rows k = w H/4 + 4 ks + lk, the member's 64 columns -- B operands of the sweep, read once
    float wreg[4][KS];
    {
        const float* __restrict__ W = a.whh[dir];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wreg[nt][ks] = W[(long long)(w * (H / 4) + 4 * ks + lk) * a.ldw + col_of<CELL, H>(pm, nt, li)];
    }
    // A operands of a step straight from the exchange (round 5; before: every thread gathered 1/256 of the partners' granules into an
    // LDS tile [k][row], a second barrier, then the waves read their fragments back -- 865 + part of 2450 of the step's 6560 cycles).
    // A granule is {tag, h[row][u], h[row][u + 4], tag} with u = 8 q + lk: the two values ONE lane (row li, k residue lk) needs for the
    // k-steps 2 q and 2 q + 1, so a wave polls exactly its own K quarter (KS / 2 granules per lane, own member's units included) and
    // starts its MFMAs the moment that quarter is there.
    unsigned goff[KS / 2];
#pragma unroll
    for (int q = 0; q < KS / 2; ++q) {
        const int u = w * (H / 4) + 8 * q + lk, m = u / UPC, j = u % UPC;
        goff[q] = (unsigned)(m * GPM + (j >> 3) * 64 + lk * 16 + li) * 16u;        // [member][block of 8 units][k residue][row]: 1 KB per wave load
    }

    // element ownership of the gate math: thread (row er, position ej) finishes n-tile column ej of all four n-tiles
    const int er = tid >> 4, ej = tid & 15;
    const bool valid = b0 + er < B;
    const long long bl = valid ? b0 + er : B - 1;                                // padded rows of a ragged tile re-read the last row ...
    const int t0 = dir ? T - 1 : 0;
    const long long tstep = dir ? -1 : 1;
    const long long fr0 = (bl * T + t0) * 2 + dir;
    const float* gl = a.gates + fr0 * GH;                                       // loads
    float* gs = valid ? a.gates + fr0 * GH : a.sink;                            // ... and store into the scratch row
    float* cs = valid && a.cstate ? a.cstate + fr0 * H : a.sink;
    float* os = valid ? a.out + bl * a.obs + (long long)t0 * a.ld_out + dir * H : a.sink;
    const long long gstep = tstep * 2 * GH, cstep = valid ? tstep * 2 * H : 0, ostep = valid ? tstep * a.ld_out : 0;
    const long long gsstep = valid ? gstep : 0;
    int cols[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cols[nt] = col_of<CELL, H>(pm, nt, ej);
    // x-projection two steps ahead, bulk stores one step behind, both issued right AFTER the poll: vector-memory results return in
    // issue order, so anything still in flight in front of the poll's granule loads (a prefetch from HBM, the stores' write
    // acknowledgements) would be waited for before the first granule can be looked at
    float xn[4], xn2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { xn[nt] = gl[cols[nt]]; xn2[nt] = gl[(T > 1 ? gstep : 0) + cols[nt]]; }
    float cst = 0.f;
    float zl[4], hl[4], cl_ = 0.f;                                              // last step's results, stored one step late
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) zl[nt] = hl[nt] = 0.f;
#ifdef LAS_PROF
    const bool fprof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (fprof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#define FSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (fprof && s >= 200 && s < 208) a.dbg[8 + (s - 200) * 8 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FSTAMP(k)
#endif

    for (int s = 0; s < T; ++s) {
        FSTAMP(0);
        // ---- h_{s-1}: this wave's K quarter of all 16 rows, as published at the end of step s - 1 (slot (s - 1) & 1, tag s)
        float av[KS];
        if (s == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) av[ks] = 0.f;
        } else {
            const unsigned slot_off = (unsigned)(((s - 1) & 1) * P) * GPM * 16u, tag = (unsigned)s;
            u32x4_t xv[KS / 2];
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) xv[q] = granule16_load(xrs, slot_off + goff[q]);
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < KS / 2; ++q) ok &= xv[q].x == tag && xv[q].w == tag;
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int q = 0; q < KS / 2; ++q)
                    if (xv[q].x != tag || xv[q].w != tag) xv[q] = granule16_load(xrs, slot_off + goff[q]);
            }
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) { av[2 * q] = __uint_as_float(xv[q].y); av[2 * q + 1] = __uint_as_float(xv[q].z); }
        }
        FSTAMP(1);
        f32x4_t acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wreg[nt][ks], acc[nt], 0, 0, 0);
        float x[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { x[nt] = xn[nt]; xn[nt] = xn2[nt]; }
        {   // the x-projection of step s + 2 (unconditional, clamped: no branch join -- it interleaves with the MFMAs like the stores below)
            const long long adv = s + 2 < T ? 2 * gstep : 0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xn2[nt] = gl[adv + cols[nt]];
        }
        {   // bulk results of step s - 1 (nobody waits for these stores; step 0 writes zeros into the scratch row).  No branch: the
            // address arithmetic and the stores interleave with the MFMAs above
            const bool st = s > 0;
            float* gq = st ? gs - gsstep : a.sink;
            float* cq = st ? cs - cstep : a.sink;
            float* oq = st ? os - ostep : a.sink;
            if (CELL == LAS_CELL_LSTM) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) gq[cols[nt]] = zl[nt];            // activated gates, saved for BPTT (in place of the x-projection)
                cq[st ? pm * 16 + ej : 0] = cl_;
                oq[st ? pm * 16 + ej : 0] = hl[0];
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) oq[st ? pm * 64 + nt * 16 + ej : 0] = hl[nt];
            }
        }
        // (two copies of the partial tiles, by step parity: a wave whose quarter arrives early may be a step ahead of one still summing)
        float* part = part2[s & 1];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[((w * 4 + nt) * 4 + i) * PT + lk * 16 + li] = acc[nt][i];
        FSTAMP(2);
        lds_barrier();
        FSTAMP(3);
        // ---- finish the member's elements: pre-activation = x-projection + the four K-quarters, in this order
        float z[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int o = (nt * 4 + (er & 3)) * PT + (er >> 2) * 16 + ej;
            z[nt] = x[nt] + (((part[o] + part[16 * PT + o]) + part[32 * PT + o]) + part[48 * PT + o]);
        }
        float hv[4] = {0.f, 0.f, 0.f, 0.f};
        if (CELL == LAS_CELL_LSTM) {
            const float gi = sigmoid_acc(z[0]), gj = tanh_acc(z[G > 1 ? 1 : 0]), gf = sigmoid_acc(z[G > 2 ? 2 : 0] + a.fb), go = sigmoid_acc(z[G > 3 ? 3 : 0]);
            cst = cst * gf + gi * gj;
            hv[0] = tanh_acc(cst) * go;
            z[0] = gi; z[G > 1 ? 1 : 0] = gj; z[G > 2 ? 2 : 0] = gf; z[G > 3 ? 3 : 0] = go;
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) hv[nt] = tanh_acc(z[nt]);
        }
        FSTAMP(4);
        if (s + 1 < T) {
            // publish: the thread of unit position j with (j & 4) == 0 carries units j and j + 4 of its row (lane + 4 holds the other)
            const unsigned slot_off = (unsigned)((s & 1) * P) * GPM * 16u, tag = (unsigned)(s + 1);
            if (CELL == LAS_CELL_LSTM) {
                const float hn = dpp_f<0x104>(hv[0]);                       // row_shl:4 -- lane i reads lane i + 4 of its row (no LDS round trip)
                if (!(ej & 4))
                    granule16_store(xrs, slot_off + (unsigned)(pm * GPM + (ej >> 3) * 64 + (ej & 3) * 16 + er) * 16u, tag,
                                    __float_as_uint(hv[0]), __float_as_uint(hn), local);
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float hn = dpp_f<0x104>(hv[nt]);
                    if (!(ej & 4))
                        granule16_store(xrs, slot_off + (unsigned)(pm * GPM + (nt * 2 + (ej >> 3)) * 64 + (ej & 3) * 16 + er) * 16u, tag,
                                        __float_as_uint(hv[nt]), __float_as_uint(hn), local);
                }
            }
        }
        FSTAMP(5);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { zl[nt] = z[nt]; hl[nt] = hv[nt]; }
        cl_ = cst;
        gl += gstep; gs += gsstep; cs += cstep; os += ostep;
    }
    // the last step's results
    if (CELL == LAS_CELL_LSTM) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gs[cols[nt] - gsstep] = zl[nt];
        cs[pm * 16 + ej - cstep] = cl_;
        os[pm * 16 + ej - ostep] = hl[0];
    } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) os[pm * 64 + nt * 16 + ej - ostep] = hl[nt];
    }
#ifdef LAS_PROF
    if (fprof) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#endif
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
}

// BPTT.  gates: activated gates (lstm) in, d(pre-activation) out, in place; rnn: gates receives d(pre-activation), h comes from `out`.
template <int CELL, int H>
__global__ __launch_bounds__(256, 1) void rnn_seq_bwd_mf32_kernel(RnnArgs a) {
    using C = F32Cfg<CELL, H>;
    constexpr int G = C::G, GH = C::GH, P = C::P, TPM = C::TPM, TW = C::TW, PT = C::PT;
    __shared__ __attribute__((aligned(16))) float dzs[4 * 16 * 20];             // the member's dz tile [k residue 4][row 16][k-step 16 (+4)]
    __shared__ __attribute__((aligned(16))) float own[TPM * 4 * PT];            // partial tile(s) of the member's own units
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int T = a.T, B = a.B;
    const int cg = blockIdx.x % a.ncl_pad, pm = blockIdx.x / a.ncl_pad;
    if (cg >= a.ncl) return;
    const int dir = cg & 1, tile = cg >> 1, b0 = tile * 16;
    // receive slots: [2 slots][receiver P][sender P][own tile TPM][128 granules]
    unsigned long long* xb = a.xbuf + (size_t)cg * 2 * P * P * TPM * 128 * 2;
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(xb);
    int errflag = 0;
    const bool local = (P > 1 && !a.force_agent) ? cluster_same_xcd(a.xcc + (size_t)cg * 32, pm, P, tid, &errflag, a.spin) : false;

    // W_hh^T slice: B[k = the member's column c = 4 ks + lk][n = unit tt * 16 + li] = W_hh[unit][column c]
    float wreg[TW][16];
    {
        const float* __restrict__ W = a.whh[dir];
#pragma unroll
        for (int x = 0; x < TW; ++x)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int c = 4 * ks + lk;
                wreg[x][ks] = W[(long long)((w * TW + x) * 16 + li) * a.ldw + col_of<CELL, H>(pm, c >> 4, c & 15)];
            }
    }
    const int er = tid >> 4, ej = tid & 15;
    const bool valid = b0 + er < B;
    const long long bl = valid ? b0 + er : B - 1;
    const int t0 = dir ? 0 : T - 1;                                             // reverse of the forward order
    const long long tstep = dir ? 1 : -1;
    const long long fr0 = (bl * T + t0) * 2 + dir;
    const float* gl = a.gates + fr0 * GH;
    float* gs = valid ? a.gates + fr0 * GH : a.sink;
    const float* cl = a.cstate ? a.cstate + fr0 * H : nullptr;
    const float* ol = a.out + bl * a.obs + (long long)t0 * a.ld_out + dir * H;
    const float* dl = a.dout + bl * a.dobs + (long long)t0 * a.ld_dout + dir * H;
    const long long gstep = tstep * 2 * GH, cstep = tstep * 2 * H, ostep = tstep * a.ld_out, dstep = tstep * a.ld_dout;
    const long long gsstep = valid ? gstep : 0;
    int cols[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cols[nt] = col_of<CELL, H>(pm, nt, ej);
    float dhr[TPM], dcc = 0.f;
#pragma unroll
    for (int q = 0; q < TPM; ++q) dhr[q] = 0.f;

    // operands of a step: lstm -- four gates, c_t, c_{t-1}, dout of the thread's unit; rnn -- h and dout of its four units
    float ng[4], nd[TPM], nc = 0.f, ncp = 0.f;
    auto fetch = [&](long long go, long long co, long long oo, long long doff, bool hasp) {
        if (CELL == LAS_CELL_LSTM) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ng[nt] = gl[go + cols[nt]];
            nc = cl[co + pm * 16 + ej];
            // (no predecessor: any finite value, multiplied by zero below.  The offset is opaque to the compiler: knowing it may be
            // zero it reused nc's register on that path -- a branch join that needed nc's load COMPLETE, s_waitcnt vmcnt(0) in the
            // middle of the prefetch: one memory round trip, ~1200 of the step's 7700 cycles)
            long long po = hasp ? cstep : 0;
            asm volatile("" : "+s"(po));
            ncp = cl[co + po + pm * 16 + ej];
            nd[0] = dl[doff + pm * 16 + ej];
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { ng[nt] = ol[oo + pm * 64 + nt * 16 + ej]; nd[nt % TPM] = dl[doff + pm * 64 + nt * 16 + ej]; }
        }
    };
    fetch(0, 0, 0, 0, T > 1);
#ifdef LAS_PROF
    const bool fprof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (fprof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#endif
    for (int s = 0; s < T; ++s) {
        FSTAMP(0);
        float g_[4], d_[TPM];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) g_[nt] = ng[nt];
#pragma unroll
        for (int q = 0; q < TPM; ++q) d_[q] = nd[q];
        const float c = nc, cp = s + 1 < T ? ncp : 0.f;
        if (s + 1 < T) fetch(gstep, cstep, ostep, dstep, s + 2 < T);
        // ---- gate backward of the thread's element(s)
        float dz[4];
        if (CELL == LAS_CELL_LSTM) {
            const float dh = d_[0] + dhr[0];
            const float gi = g_[0], gj = g_[G > 1 ? 1 : 0], gf = g_[G > 2 ? 2 : 0], go = g_[G > 3 ? 3 : 0];
            const float tc = tanh_acc(c);
            const float dc = dcc + dh * go * (1.f - tc * tc);
            dcc = dc * gf;
            dz[0] = dc * gj * gi * (1.f - gi);
            dz[G > 1 ? 1 : 0] = dc * gi * (1.f - gj * gj);
            dz[G > 2 ? 2 : 0] = dc * cp * gf * (1.f - gf);
            dz[G > 3 ? 3 : 0] = dh * tc * go * (1.f - go);
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { const float dh = d_[nt % TPM] + dhr[nt % TPM]; dz[nt] = dh * (1.f - g_[nt] * g_[nt]); }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            gs[cols[nt]] = dz[nt];
            dzs[((ej & 3) * 16 + er) * 20 + nt * 4 + (ej >> 2)] = dz[nt];         // column c = nt 16 + ej: k-step c >> 2, residue c & 3
        }
        gl += gstep; gs += gsstep; if (CELL == LAS_CELL_LSTM) cl += cstep; ol += ostep; dl += dstep;
        if (s + 1 == T) break;                                                   // the last step's dh has no consumer
        FSTAMP(1);
        lds_barrier();
        FSTAMP(2);
        // ---- partial dh of ALL units from the member's 64 columns
        float av[16];
        {
            const float4* dp = reinterpret_cast<const float4*>(dzs + (lk * 16 + li) * 20);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = dp[q]; av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w; }
        }
        f32x4_t acc[TW];
#pragma unroll
        for (int x = 0; x < TW; ++x) acc[x] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int x = 0; x < TW; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wreg[x][ks], acc[x], 0, 0, 0);
        FSTAMP(3);
        const unsigned slot_off = (unsigned)((s & 1) * P) * (unsigned)(P * TPM * 128) * 16u, tag = (unsigned)(s + 1);
        // ---- reduce-scatter: every unit tile goes to the member that owns it (rows 4 lk .. 4 lk + 3, unit li of the tile per lane)
#pragma unroll
        for (int x = 0; x < TW; ++x) {
            const int tt = w * TW + x, owner = tt / TPM, lt = tt % TPM;          // (wave-uniform)
            if (owner == pm) {
#pragma unroll
                for (int i = 0; i < 4; ++i) own[(lt * 4 + i) * PT + lk * 16 + li] = acc[x][i];
            } else {
                const unsigned base = slot_off + (unsigned)(((owner * P + pm) * TPM + lt) * 128 + lk * 32 + li) * 16u;
                granule16_store(xrs, base, tag, __float_as_uint(acc[x][0]), __float_as_uint(acc[x][1]), local);
                granule16_store(xrs, base + 16u * 16u, tag, __float_as_uint(acc[x][2]), __float_as_uint(acc[x][3]), local);
            }
        }
        // ---- receive: the thread pair (rows er & ~1, er | 1; same unit) shares the partners' granules -- the even row polls the
        // partners at even list positions, the odd row the others; both sum BOTH rows' values in list order, then they swap halves
        float sa[TPM], sb[TPM];
#pragma unroll
        for (int q = 0; q < TPM; ++q) sa[q] = sb[q] = 0.f;
        FSTAMP(4);
        if constexpr (P > 1) {
            constexpr int NH = P / 2;                                            // list positions of one thread (the last may not exist)
            u32x4_t xv[NH * TPM];
            const int par = er & 1;
            const unsigned gidx = (unsigned)((er >> 2) * 32 + ((er >> 1) & 1) * 16 + ej);
#pragma unroll
            for (int n = 0; n < NH; ++n)
#pragma unroll
                for (int q = 0; q < TPM; ++q) {
                    const int pos = 2 * n + par;
                    xv[n * TPM + q] = (u32x4_t){tag, 0u, 0u, tag};
                    if (pos < P - 1)
                        xv[n * TPM + q] = granule16_load(xrs, slot_off + ((unsigned)((pm * P + (pm + 1 + pos) % P) * TPM + q) * 128u + gidx) * 16u);
                }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NH * TPM; ++n) ok &= xv[n].x == tag && xv[n].w == tag;
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NH; ++n)
#pragma unroll
                    for (int q = 0; q < TPM; ++q) {
                        const int pos = 2 * n + par;
                        if (pos < P - 1 && (xv[n * TPM + q].x != tag || xv[n * TPM + q].w != tag))
                            xv[n * TPM + q] = granule16_load(xrs, slot_off + ((unsigned)((pm * P + (pm + 1 + pos) % P) * TPM + q) * 128u + gidx) * 16u);
                    }
            }
#pragma unroll
            for (int n = 0; n < NH; ++n)
#pragma unroll
                for (int q = 0; q < TPM; ++q) { sa[q] += __uint_as_float(xv[n * TPM + q].y); sb[q] += __uint_as_float(xv[n * TPM + q].z); }
        }
        FSTAMP(5);
        lds_barrier();                                                           // own[] is complete; dzs may be rewritten
        FSTAMP(6);
#pragma unroll
        for (int q = 0; q < TPM; ++q) {
            // this thread's row: even rows are the granules' first value.  partner lane = the other row of the pair (tid ^ 16)
            const float mine = (er & 1) ? sb[q] : sa[q], give = (er & 1) ? sa[q] : sb[q];
            // the partner row's value without the LDS: of two copies of `give`, v_permlane16_swap leaves (even rows, even rows) and (odd rows, odd rows)
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(give), __float_as_uint(give), false, false);
            const float got = __uint_as_float((threadIdx.x & 16) ? sw[0] : sw[1]);
            const float se = (er & 1) ? got : mine, so = (er & 1) ? mine : got;     // even-position partners' sum, odd-position partners' sum
            dhr[q] = own[(q * 4 + (er & 3)) * PT + (er >> 2) * 16 + ej] + (se + so);
        }
    }
#ifdef LAS_PROF
    if (fprof) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#endif
    if (errflag) { if (a.err) a.err[0] = 1; if (a.status) a.status[0] = a.status_code; }
}

struct F32Ws { size_t err, sink, xcc, xbuf, per_cl, total; int max_cl; };
F32Ws f32_layout(int cell, int H) {
    const int G = cell == LAS_CELL_LSTM ? 4 : 1, P = G * H / 64, TPM = 64 / G / 16, GPM = 16 * (64 / G) / 2;
    F32Ws w;
    size_t o = 0;
    w.err = o; o += 256;
    w.sink = o; o += ((size_t)(G * H + 64) * sizeof(float) + 255) & ~(size_t)255;
    int cus = las_device_cus();
    w.max_cl = (cus / P / 8) * 8;                                               // clusters per launch: members of one cluster 8 blocks apart (same XCD)
    if (w.max_cl < 2) w.max_cl = 0;
    const size_t ncl = w.max_cl > 0 ? (size_t)w.max_cl : 8;
    w.xcc = o; o += ncl * 32 * sizeof(unsigned long long);
    const size_t fwd = (size_t)2 * P * GPM * 16, bwd = (size_t)2 * P * P * TPM * 128 * 16;
    w.per_cl = fwd > bwd ? fwd : bwd;
    w.xbuf = o; o += ncl * w.per_cl;
    w.total = o + 256;
    return w;
}

}  // namespace

bool las_rnn_seq_mf32_ok(int cell, int H) {
    if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
    return f32_layout(cell, H).max_cl >= 2;
}

size_t las_rnn_seq_mf32_ws_bytes(int cell, int H) { return las_rnn_seq_mf32_ok(cell, H) ? f32_layout(cell, H).total : 0; }

// One bidirectional layer, forward (bwd = false) or BPTT, through the clustered exact-fp32 kernels; the batch is swept in row chunks
// whose clusters are all co-resident (one workgroup per compute unit), one launch after the other on the same stream.
int las_rnn_seq_mf32_run(bool bwd, int cell, const RnnArgs& a_in, void* ws, size_t ws_bytes, int flags, hipStream_t st) {
    const int G = cell == LAS_CELL_LSTM ? 4 : 1, H = a_in.H, B = a_in.B, T = a_in.T;
    const F32Ws L = f32_layout(cell, H);
    LAS_ARG(ws && ws_bytes >= L.total, "las_rnn_seq (fp32 clusters): workspace too small (%zu < %zu)", ws_bytes, L.total);
    char* base = (char*)ws;
    const int P = G * H / 64;
    const int tiles = cdiv(B, 16), max_tiles = L.max_cl / 2;
    for (int tile0 = 0; tile0 < tiles; tile0 += max_tiles) {
        const int b0 = tile0 * 16, rows = (B - b0) < max_tiles * 16 ? (B - b0) : max_tiles * 16;
        RnnArgs c = a_in;
        c.B = rows;
        c.gates = a_in.gates + (size_t)b0 * T * 2 * G * H;
        c.out = a_in.out + (size_t)b0 * a_in.obs;
        if (a_in.cstate) c.cstate = a_in.cstate + (size_t)b0 * T * 2 * H;
        if (a_in.dout) c.dout = a_in.dout + (size_t)b0 * a_in.dobs;
        c.err = (int*)(base + L.err);
        c.sink = (float*)(base + L.sink);
        c.xcc = (unsigned long long*)(base + L.xcc);
        c.xbuf = (unsigned long long*)(base + L.xbuf);
        c.force_agent = (flags & LAS_SEQ_AGENT_GRANULES) ? 1 : 0;
        c.ncl = cdiv(rows, 16) * 2;
        c.ncl_pad = (c.ncl + 7) / 8 * 8;
        // err word, scratch row, handshake slots and the granule tags of the clusters this launch uses: one fill
        LAS_HIP(hipMemsetAsync(base, 0, L.xbuf + (size_t)c.ncl * L.per_cl, st));
        const dim3 grid(c.ncl_pad * P), block(256);
        // The CU to itself, like the speed mode's sweeps (rnn_seq.hip ks_lds): the kernels need 7-41 KB of LDS and one wave per SIMD, so
        // the side stream's weight-gradient GEMM workgroups WOULD be scheduled next to a member and share its SIMDs, LDS and L1 with
        // the dependent chain (BPTT sweeps of a train step: 3.7-4.4 us per step in place, 2.7 alone).  Dynamic LDS that nobody touches
        // fills the CU's allocation: nothing that uses LDS fits next to the sweep.
        constexpr int FILL = 159 * 1024 - 42 * 1024;
#define LAS_F32_LAUNCH(CELL, HH)                                                                                         \
        do { static int attr__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_seq_bwd_mf32_kernel<CELL, HH>),     \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, FILL) |                   \
                                 (int)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_seq_fwd_mf32_kernel<CELL, HH>),     \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, FILL);                    \
             LAS_ARG(attr__ == 0, "hipFuncSetAttribute(fp32 sweep) failed: %d", attr__);                                  \
             if (bwd) hipLaunchKernelGGL((rnn_seq_bwd_mf32_kernel<CELL, HH>), grid, block, fill, st, c);                 \
             else     hipLaunchKernelGGL((rnn_seq_fwd_mf32_kernel<CELL, HH>), grid, block, fill, st, c); } while (0)
        const size_t fill = LAS_KS_SHARE_CU ? 0 : (size_t)FILL;
        if (cell == LAS_CELL_LSTM) {
            switch (H) {
                case 64: LAS_F32_LAUNCH(LAS_CELL_LSTM, 64); break;
                case 128: LAS_F32_LAUNCH(LAS_CELL_LSTM, 128); break;
                case 256: LAS_F32_LAUNCH(LAS_CELL_LSTM, 256); break;
                default: LAS_F32_LAUNCH(LAS_CELL_LSTM, 512); break;
            }
        } else {
            switch (H) {
                case 64: LAS_F32_LAUNCH(LAS_CELL_RNN, 64); break;
                case 128: LAS_F32_LAUNCH(LAS_CELL_RNN, 128); break;
                case 256: LAS_F32_LAUNCH(LAS_CELL_RNN, 256); break;
                default: LAS_F32_LAUNCH(LAS_CELL_RNN, 512); break;
            }
        }
#undef LAS_F32_LAUNCH
        LAS_LAUNCHED();
    }
    return 0;
}
