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
//             accurate transcendentals), h is published to the partners as tagged 16-byte granules {tag, h_j, h_j+1, tag}.
//   BPTT      K-split like the speed mode's: a member contracts ITS 64 columns of d(pre-activation) against W_hh^T for ALL H
//             units (wave w: unit tiles [w H/64, (w+1) H/64)), keeps the partial tile(s) of its own units and sends the others
//             to their owners; a thread pair (rows 2r, 2r+1) shares the partners' granules, half each, fixed order.
// Exchange transport, placement handshake, bounded polls and the status word are the speed mode's (rnn_seq_args.h).
// No atomics; every sum has a fixed order: bit-reproducible.
#include "rnn_seq_args.h"

namespace {

template <int CELL, int H_>
struct F32Cfg {
    static constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1, H = H_, GH = G * H;
    static constexpr int UPC = 64 / G;             // hidden units a member owns
    static constexpr int P = H / UPC;              // members per (direction, 16-row tile)
    static constexpr int KS = H / 16;              // forward: k-steps (of 4) per wave
    static constexpr int KP = KS + 4;              // LDS pitch of one (k residue, row) run of the h tile: conflict-free 16-byte reads
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
// offset of hidden unit k, batch row r in the forward h tile [k-quarter 4][k residue 4][row 16][KP]
template <int H, int KP>
__device__ __forceinline__ int hs_off(int k, int r) {
    const int wq = k / (H / 4), in = k % (H / 4);
    return ((wq * 4 + (in & 3)) * 16 + r) * KP + (in >> 2);
}

template <int CELL, int H>
__global__ __launch_bounds__(256, 1) void rnn_seq_fwd_mf32_kernel(RnnArgs a) {
    using C = F32Cfg<CELL, H>;
    constexpr int G = C::G, GH = C::GH, P = C::P, KS = C::KS, KP = C::KP, GPM = C::GPM, PT = C::PT;
    __shared__ __attribute__((aligned(16))) float hs[16 * 16 * KP];
    __shared__ __attribute__((aligned(16))) float part[4 * 4 * 4 * PT];         // [wave][n-tile][acc register i][lk][li]
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
    for (int i = tid; i < 16 * 16 * KP; i += 256) hs[i] = 0.f;
    __syncthreads();

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
    float xn[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) xn[nt] = gl[cols[nt]];
    float cst = 0.f;
#ifdef LAS_PROF
    const bool fprof = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
    if (fprof) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); }
#define FSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (fprof && s >= 200 && s < 208) a.dbg[8 + (s - 200) * 8 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FSTAMP(k)
#endif

    for (int s = 0; s < T; ++s) {
        FSTAMP(0);
        float x[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) x[nt] = xn[nt];
        {   // next step's x-projection flies under this step (unconditional, clamped: no branch join in front of the MFMAs)
            const long long adv = s + 1 < T ? gstep : 0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xn[nt] = gl[adv + cols[nt]];
        }
        float av[KS];
        {
            const float4* hp = reinterpret_cast<const float4*>(hs + ((w * 4 + lk) * 16 + li) * KP);
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) { const float4 v = hp[q]; av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w; }
        }
        f32x4_t acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wreg[nt][ks], acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[((w * 4 + nt) * 4 + i) * PT + lk * 16 + li] = acc[nt][i];
        FSTAMP(1);
        lds_barrier();
        FSTAMP(2);
        // ---- finish the member's elements: pre-activation = x-projection + the four K-quarters, in this order
        float z[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int o = (nt * 4 + (er & 3)) * PT + (er >> 2) * 16 + ej;
            z[nt] = x[nt] + (((part[o] + part[16 * PT + o]) + part[32 * PT + o]) + part[48 * PT + o]);
        }
        float hv[4];
        if (CELL == LAS_CELL_LSTM) {
            const float gi = sigmoid_acc(z[0]), gj = tanh_acc(z[G > 1 ? 1 : 0]), gf = sigmoid_acc(z[G > 2 ? 2 : 0] + a.fb), go = sigmoid_acc(z[G > 3 ? 3 : 0]);
            cst = cst * gf + gi * gj;
            hv[0] = tanh_acc(cst) * go;
            z[0] = gi; z[G > 1 ? 1 : 0] = gj; z[G > 2 ? 2 : 0] = gf; z[G > 3 ? 3 : 0] = go;
            hs[hs_off<H, KP>(pm * 16 + ej, er)] = hv[0];
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                hv[nt] = tanh_acc(z[nt]);
                hs[hs_off<H, KP>(pm * 64 + nt * 16 + ej, er)] = hv[nt];
            }
        }
        FSTAMP(3);
        if constexpr (P > 1) if (s + 1 < T) {
            const unsigned slot_off = (unsigned)((s & 1) * P) * GPM * 16u, tag = (unsigned)(s + 1);
            // publish: lstm -- the even position of a unit pair carries both units' h; rnn -- the thread's four units as two granules
            if (CELL == LAS_CELL_LSTM) {
                const float hn = __shfl_xor(hv[0], 1, 64);
                if (!(ej & 1))
                    granule16_store(xrs, slot_off + (unsigned)(pm * GPM + er * 8 + (ej >> 1)) * 16u, tag, __float_as_uint(hv[0]), __float_as_uint(hn), local);
            } else {
                granule16_store(xrs, slot_off + (unsigned)(pm * GPM + (er * 16 + ej) * 2) * 16u, tag, __float_as_uint(hv[0]), __float_as_uint(hv[1]), local);
                granule16_store(xrs, slot_off + (unsigned)(pm * GPM + (er * 16 + ej) * 2 + 1) * 16u, tag, __float_as_uint(hv[2]), __float_as_uint(hv[3]), local);
            }
            // gather the partners' slices of h_t into the LDS tile: all loads in flight at once, then re-poll only the stale ones
            constexpr int NGR = (P - 1) * GPM, NG = (NGR + 255) / 256;
            u32x4_t xv[NG];
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int g = tid + 256 * n;
                xv[n] = (u32x4_t){tag, 0u, 0u, tag};
                if (g < NGR) xv[n] = granule16_load(xrs, slot_off + (unsigned)(((pm + 1 + g / GPM) % P) * GPM + g % GPM) * 16u);
            }
            int budget = errflag ? 1 : a.spin;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < NG; ++n) ok &= xv[n].x == tag && xv[n].w == tag;
                if (ok) break;
                if (--budget <= 0) { errflag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    const int g = tid + 256 * n;
                    if (g < NGR && (xv[n].x != tag || xv[n].w != tag))
                        xv[n] = granule16_load(xrs, slot_off + (unsigned)(((pm + 1 + g / GPM) % P) * GPM + g % GPM) * 16u);
                }
            }
            FSTAMP(4);
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int g = tid + 256 * n;
                if (g < NGR) {
                    const int m = (pm + 1 + g / GPM) % P, idx = g % GPM;
                    if (CELL == LAS_CELL_LSTM) {
                        const int r = idx >> 3, u = m * 16 + 2 * (idx & 7);
                        hs[hs_off<H, KP>(u, r)] = __uint_as_float(xv[n].y);
                        hs[hs_off<H, KP>(u + 1, r)] = __uint_as_float(xv[n].z);
                    } else {
                        const int r = idx >> 5, j = (idx >> 1) & 15, nt = 2 * (idx & 1);
                        hs[hs_off<H, KP>(m * 64 + nt * 16 + j, r)] = __uint_as_float(xv[n].y);
                        hs[hs_off<H, KP>(m * 64 + (nt + 1) * 16 + j, r)] = __uint_as_float(xv[n].z);
                    }
                }
            }
        }
        lds_barrier();
        FSTAMP(5);
        // bulk results of this step (nobody waits for these stores)
        if (CELL == LAS_CELL_LSTM) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) gs[cols[nt]] = z[nt];                 // activated gates, saved for BPTT (in place of the x-projection)
            cs[pm * 16 + ej] = cst;
            os[pm * 16 + ej] = hv[0];
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) os[pm * 64 + nt * 16 + ej] = hv[nt];
        }
        gl += gstep; gs += gsstep; cs += cstep; os += ostep;
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
            const float got = __shfl_xor(give, 16, 64);
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
#define LAS_F32_LAUNCH(CELL, HH)                                                                                         \
        do { if (bwd) hipLaunchKernelGGL((rnn_seq_bwd_mf32_kernel<CELL, HH>), grid, block, 0, st, c);                    \
             else     hipLaunchKernelGGL((rnn_seq_fwd_mf32_kernel<CELL, HH>), grid, block, 0, st, c); } while (0)
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
