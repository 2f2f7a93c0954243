// speller_wide.h -- the Speller decode loop (reference las/las.py:72-160) and its gradient for geometries OUTSIDE the one the one-launch loop
// kernels are built around (one decoder layer, D <= 512, T' <= 224): in particular the reference's own recipe, run.sh:59-76 -- a
// MultiRNNCell of TWO 1024-unit cells whose concatenated states are the attention query (S = 2048, las/las.py:185-199), T' = 319 frames
// behind the CNN listener, location-aware attention (las/layers.py:281-311).  Included by speller.hip (one translation unit).
//
// Round 5 served such calls with the generic row kernels: ONE workgroup per utterance pulls Ws (S x A fp32 = 1 MB), the keys and the
// encoder rows of its utterance through one CU at every step (61 us forward / 174 us backward per decode step at B = 48: 48 of 256 CUs, each
// bound by its ~50-80 GB/s of ingest), and the cells of layers >= 1 through the general GEMM (53 us for a 48-row product).  Here a step is a
// short chain of launches that each fill the machine:
//     forward   state rows (finish the top cell of step t-1 [+ in-loop logits])  ->  q = s . Ws for all rows (skinny MFMA product, Ws
//               fragments read once per step instead of once per row)  ->  energies (+ location conv) on (frame slice, utterance)
//               workgroups with the slices' softmax statistics  ->  alignment + context on (column slice, utterance) workgroups, which
//               also leave the cell input row  ->  one skinny product per layer with a pointwise launch between them
//     backward  d alpha on (frame slice, utterance)  ->  d energy, dq / du partials, d f  ->  dq, the conv's transpose  ->  d s = dq . Ws^T
//               (skinny)  ->  per layer: gate gradient, skinny product with the transposed fragments
// Keys / Wf / filter gradients are contracted over the steps after the loop (from dE, Q, f, d f of every step), as in the loop kernels.
// Second half of round 6 ("inside the launches"): unless LAS_SPELLER_NO_FUSED_STEP asks for one launch per phase, energies +
// alignment / context are ONE launch (wide_attend_kernel) and so are d alpha -> d energy -> dq (wide_attend_bwd_kernel): an utterance's
// workgroups hand their partial results to each other as tagged granules (bounded polls; the status word reports a partner that never ran);
// and the cells and their gate gradients are epilogues of the skinny products (gemm.hip: las_skinny_gemm_bf16_tanh / _tanh_bwd / _lstm_bwd,
// las_skinny_lstm_bf16).  8 dependent launches per decode step (either cell) instead of 13 (tanh) or 15 (LSTM) -- the same arithmetic.
// Arithmetic: speed mode = the loop kernels' ("bf rows": keys kept in bf16, q / context / cell products on bf16 operands with fp32
// accumulation, everything else fp32); parity mode = fp32 throughout (the same kernels with FAST = false, accurate transcendentals).
#pragma once

constexpr int WIDE_MAX_SPLIT = 8;
#define LAS_ACT_MAGIC_WIDE 0x4c415357u      // act_save header: the wide forward kept the conv outputs f of every step

struct WideDev {
    int nsplit, fper;          // frame slices per utterance, frames per slice
    int hsplit, h4per;         // context column slices per utterance, 4-column chunks per slice
    float* qbuf;               // [B, A]      q = s . Ws of the current step (forward)
    float* ebuf;               // [B, Tp]     energies (forward) / d alpha (backward)
    float* stat;               // [B, nsplit, 2] softmax statistics (forward); [B, nsplit] partial alpha . d alpha (backward)
    float *pdq, *pdu;          // [B, nsplit, A] partial dq / du of the frame slices
    unsigned short* sbf;       // [B, S]      state rows: bf16 (speed mode: A operand of the query product) -- or fp32 in the same buffer (parity mode)
    unsigned short* xu;        // [B, 2D]     [h_{l-1, t+1} ; h_{l, t}] bf16: A operand of an upper layer's cell product
    unsigned short* dqbf;      // [B, A]
    unsigned short* dgu;       // [B, G D]    gate gradient rows of an upper layer, bf16
    float* dS;                 // [B, S]      dq . Ws^T
    float* dWfW;               // [B, ceil(Tp / 8), C, A]  Wf-gradient partials of the after-loop keys kernel (written whole)
    int xcd_local;             // 1: the fused attention launches are 1-D grids that keep an utterance's slices on ONE XCD (workgroup j: XCD x = j % 8, k = j / 8,
                               //    utterance (k / SP) 8 + x, slice k % SP) and hand over at XCD scope (sc0 stores into that XCD's L2); 0: (slice, utterance) grids, device scope
    int sp;                    // slices per utterance of the fused launches (max(nsplit, hsplit) forward, nsplit reverse: set per launch)
    unsigned long long* bgran; // [B, wide_bgran_row(Tp)] {tag, value} granules of the fused REVERSE attention launch; tag = step + 1, zeroed per call
    unsigned long long* egran; // [B, Tp + 2 WIDE_MAX_SPLIT] {tag, value} granules of the fused attention launch (wide_attend_kernel): an utterance's
                               // energies, then its slices' (max, sum of exp); tag = step + 1, zeroed per call
};
__host__ __device__ __forceinline__ int wide_gran_row(int Tp) { return Tp + 2 * WIDE_MAX_SPLIT; }
// the fused reverse attention launch (wide_attend_bwd_kernel): per utterance WIDE_MAX_SPLIT granules of the slices' alpha . d alpha, then
// [slice][dq | du][A <= 256] partials, then the step's d f rows [Tp][16]
constexpr int WIDE_BG_PART = WIDE_MAX_SPLIT, WIDE_BG_DF = WIDE_MAX_SPLIT + WIDE_MAX_SPLIT * 2 * 256;
__host__ __device__ __forceinline__ int wide_bgran_row(int Tp) { return WIDE_BG_DF + Tp * 16; }

// ... for a granule whose first load is already in flight / in a register (several polls issued together instead of one round trip after the other)
__device__ __forceinline__ unsigned wide_poll_loaded(const DecDev& a, __amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned tag, int budget, u32x2_t g) {
    while (g.x != tag) {
        if (--budget <= 0) { if (a.lp.status) a.lp.status[0] = LAS_SPELLER_STATUS_TIMEOUT; break; }
        __builtin_amdgcn_s_sleep(2);
        g = granule8_load(rs, byte_off);
    }
    return g.y;
}
// (slice, utterance) of a fused attention workgroup: from the 2-D grid, or from the XCD-local 1-D grid (see WideDev.xcd_local); b < 0: nothing to do
__device__ __forceinline__ void wide_slice_of(const WideDev& w, const int B, int& s, int& b) {
    if (!w.xcd_local) { s = blockIdx.x; b = blockIdx.y; return; }
    const int j = blockIdx.x, x = j & 7, k = j >> 3;
    s = k % w.sp;
    b = (k / w.sp) * 8 + x;
    if (b >= B) b = -1;
}
// a granule another workgroup of this launch publishes: polled with a bound (las_speller_fwd_args.status reports a partner that never ran)
// The bound: a hand-over normally completes within microseconds, a partner kept off the machine by somebody else's kernel arrives when that
// kernel ends -- 2^15 rounds (~50 ms) cover the latter; and once the call's status word is set (an earlier launch of this call gave up: the
// step is lost and will be re-run) the following launches do not wait at all, so a lost step costs one bound, not one per decode step.
__device__ __forceinline__ int wide_poll_budget(const DecDev& a) {
    if (a.lp.status && __hip_atomic_load(a.lp.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 1;
    return a.lp.budget < (1 << 15) ? a.lp.budget : (1 << 15);
}
__device__ __forceinline__ unsigned wide_poll(const DecDev& a, __amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned tag, int budget) {
    u32x2_t g = granule8_load(rs, byte_off);
    while (g.x != tag) {
        if (--budget == 0) { if (a.lp.status) a.lp.status[0] = LAS_SPELLER_STATUS_TIMEOUT; break; }
        __builtin_amdgcn_s_sleep(2);
        g = granule8_load(rs, byte_off);
    }
    return g.y;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// (1) one workgroup per utterance: finish the TOP layer's cell of step t-1 [+ vocabulary projection, arg-max, Gumbel draw], resolve the
//     token entering step t, leave the concatenated state row s_t = [h_0 ; ... ; h_{NL-1}] for the query product.
template <int CELL, bool FAST>
__global__ __launch_bounds__(RNT) void wide_state_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hl = sm;                          // [D]
    float* red = sm + ((a.D + 3) & ~3);      // [32]
    int* redi = reinterpret_cast<int*>(red + 32);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, D = a.D, NL = a.NL, U = a.U, S = D * NL, TOP = NL - 1, GD = G * D;
    int greedy_tok = 1, sample_tok = 1;
    if (t > 0) {
        float* gp = a.gates + (((size_t)TOP * U + (t - 1)) * B + b) * GD;
        float* hnew = a.hs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
        for (int d = tid; d < D; d += RNT) {
            float h;
            if (CELL == LAS_CELL_LSTM) {
                const float* cprev = a.cs + (((size_t)TOP * (U + 1) + (t - 1)) * B + b) * D;
                float* cnew = a.cs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
                const float gi = sigm<FAST>(gp[d]), gj = tanhx<FAST>(gp[D + d]);
                const float gf = sigm<FAST>(gp[2 * D + d] + a.fb), go = sigm<FAST>(gp[3 * D + d]);
                const float c = cprev[d] * gf + gi * gj;
                h = tanhx<FAST>(c) * go;
                gp[d] = gi; gp[D + d] = gj; gp[2 * D + d] = gf; gp[3 * D + d] = go;
                cnew[d] = c;
            } else {
                h = tanhx<FAST>(gp[d]);
            }
            hnew[d] = h;
            hl[d] = h;
        }
        __syncthreads();
        if (logits_here(a, t, t < U ? a.tok_in[(size_t)t * B + b] : 0)) row_logits<FAST>(a, hl, t, b, tid, red, redi, greedy_tok, sample_tok);
    }
    if (t >= U) return;
    int tok = a.tok_in[(size_t)t * B + b];
    if (tok == -1) tok = greedy_tok;
    else if (tok == -2) tok = sample_tok;
    if (tid == 0) a.tok_in[(size_t)t * B + b] = tok;
    for (int i = tid; i < S; i += RNT) {
        const int l = i / D, d = i - l * D;
        const float v = (l == TOP && t > 0) ? hl[d] : a.hs[(((size_t)l * (U + 1) + t) * B + b) * D + d];
        if (FAST) w.sbf[(size_t)b * S + i] = f2bf(v);
        else reinterpret_cast<float*>(w.sbf)[(size_t)b * S + i] = v;
        // step 0 of the tanh-epilogue chain (speller_wide_host.h): the upper layers' recurrent halves [. ; h_{l,0}] of their first operand rows
        if (FAST && t == 0 && l >= 1 && l == 1) w.xu[(size_t)b * 2 * D + D + d] = f2bf(v);
    }
}

// f[t', c] = bias[c] + sum_k prev_align[t' + k - pad] w[k, c] for the frames [t0, t0 + nf) of one utterance (conv1d, SAME, cross-correlation:
// las/layers.py:295-296) -> fc [nf, C] in LDS.  `awin` is the slice's WINDOW of the previous alignment in LDS -- awin[j] = alpha_{t-1}[t0 - pad
// + j], zero where that frame does not exist (wide_stage_awin) -- so that frame fr meets tap k at awin[fr + k] and no tap needs a bound;
// locw [Kc + 1, C] in LDS: the filter, and the bias as row Kc (wide_stage_filter); part: nf C ksplit floats of scratch.
// One thread = (tap slice, block of 8 consecutive frames, channel): the 8 + 8 window values of a chunk of 8 taps stay in registers (one new
// value and one weight per 8 multiply-adds), every loop has a fixed trip count.  (Round 6, first version: one (tap slice, frame, channel)
// output per thread item with two LDS reads per multiply-add in a loop the compiler could not unroll -- 6.9 of the energy kernel's 12.3 us,
// profiles/r6_wide_phase_stamps.txt.)  Same taps per slice, same order within a slice, same order of the slices' sum as that version.
__host__ __device__ __forceinline__ int wide_conv_ks(int Kc) { return Kc >= 64 ? 8 : 1; }      // tap slices per (frame, channel) output
__host__ __device__ __forceinline__ int wide_conv_kper(int Kc) { const int ks = wide_conv_ks(Kc); return (Kc + ks - 1) / ks; }
// floats of the window: the slice's frames rounded up to 8, plus every tap a thread may touch (its slice's start + its taps rounded up to 8)
__host__ __device__ __forceinline__ int wide_awin_floats(int fper, int Kc) {
    const int ks = wide_conv_ks(Kc), kper = wide_conv_kper(Kc);
    return ((fper + 7) & ~7) + (ks - 1) * kper + ((kper + 7) & ~7) + 8;
}
__device__ __forceinline__ void wide_stage_awin(const DecDev& a, float* awin, int t, int b, int t0, int fper, int tid) {
    const int Tp = a.Tp, pad = (a.Kc - 1) / 2, n = wide_awin_floats(fper, a.Kc);
    const float* src = t > 0 ? a.alphas + ((size_t)(t - 1) * a.B + b) * Tp : (a.align0 ? a.align0 + (size_t)b * Tp : nullptr);
    for (int j = tid; j < n; j += RNT) {
        const int fr = t0 - pad + j;
        awin[j] = (src && fr >= 0 && fr < Tp) ? src[fr] : 0.f;
    }
}
__device__ __forceinline__ void wide_stage_filter(const DecDev& a, float* locw, int tid) {
    const int n = a.Kc * a.C;
    for (int i = tid; i < n + a.C; i += RNT) locw[i] = i < n ? a.loc_w[i] : a.loc_b[i - n];
}
__device__ __forceinline__ void wide_conv_slice(const DecDev& a, const float* awin, const float* locw, float* fc, float* part, int nf, int tid) {
    const int C = a.C, Kc = a.Kc, items = nf * C;
    const int ks = wide_conv_ks(Kc), kper = wide_conv_kper(Kc);
    const int nfb = (nf + 7) >> 3, items8 = nfb * C;
    for (int i = tid; i < items8 * ks; i += RNT) {
        const int kc = i / items8, r = i - kc * items8, fb = r / C, c = r - fb * C;
        const int k0 = kc * kper, kend = k0 + kper < Kc ? k0 + kper : Kc;
        const float init = kc == 0 ? locw[Kc * C + c] : 0.f;                    // (the bias rides behind the filter's rows: wide_stage_filter)
        float acc[8], x[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j] = init; x[j] = awin[fb * 8 + k0 + j]; }
        int k = k0;
        for (; k + 8 <= kend; k += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[8 + u] = awin[fb * 8 + k + 8 + u]; wv[u] = locw[(k + u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(x[j + u], wv[u], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[8 + j];
        }
        for (; k < kend; ++k) {                                // the slice's last taps (kper % 8 of them)
            const float wv = locw[k * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(awin[fb * 8 + k + j], wv, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (fb * 8 + j < nf) part[kc * items + (fb * 8 + j) * C + c] = acc[j];
    }
    __syncthreads();
    for (int it = tid; it < items; it += RNT) {
        float acc = part[it];
        for (int kc = 1; kc < ks; ++kc) acc += part[kc * items + it];
        fc[it] = acc;
    }
    __syncthreads();
}

struct WideLds { float *aprev, *locw, *wfl, *fc, *dfc, *qv, *ev, *red, *part; };
// part: the conv's tap-slice partials (8 fper C floats), later the 32 frame groups' dq and du partials (2 RNG A)
__host__ __device__ __forceinline__ size_t wide_part_floats(int A, int fper, int C) {
    size_t n = (size_t)2 * RNG * A;
    if (C > 0 && (size_t)8 * fper * C > n) n = (size_t)8 * fper * C;
    return (n + 3) & ~(size_t)3;
}
__device__ __forceinline__ WideLds wide_carve(float* sm, const DecDev& a, int fper) {
    auto u4 = [](int x) { return (x + 3) & ~3; };
    WideLds L; float* p = sm;
    const bool loc = a.mode == LAS_ATT_LOC;
    L.qv = p; p += u4(a.A);
    L.ev = p; p += u4(fper);
    L.red = p; p += 64;
    L.aprev = p; p += loc ? u4(wide_awin_floats(fper, a.Kc)) : 0;      // the slice's window of alpha_{t-1} (wide_stage_awin)
    L.locw = p; p += loc ? u4((a.Kc + 8) * a.C) : 0;
    L.wfl = p; p += loc ? u4(a.C * a.A) : 0;
    L.fc = p; p += loc ? u4(fper * a.C) : 0;
    L.dfc = p; p += loc ? u4(fper * a.C) : 0;
    L.part = p;
    return L;
}
static size_t wide_lds_bytes(const DecDev& a, int fper) {
    auto u4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    const bool loc = a.mode == LAS_ATT_LOC;
    size_t n = u4(a.A) + u4(fper) + 64;
    if (loc) n += u4((size_t)wide_awin_floats(fper, a.Kc)) + u4((size_t)(a.Kc + 8) * a.C) + u4((size_t)a.C * a.A) + 2 * u4((size_t)fper * a.C);
    return (n + wide_part_floats(a.A, fper, loc ? a.C : 0)) * sizeof(float) + 64;
}

// keys [b, tt, 4 a4 .. 4 a4 + 3] in the arithmetic of the mode
template <bool FAST>
__device__ __forceinline__ float4 wide_key4(const DecDev& a, int b, int tt, int a4) {
    if (FAST) {
        const uint2 v = reinterpret_cast<const uint2*>(a.keysbf + ((size_t)b * a.Tp + tt) * a.A)[a4];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
    return reinterpret_cast<const float4*>(a.keys + ((size_t)b * a.Tp + tt) * a.A)[a4];
}

// (2) energies of the frames [s fper, (s + 1) fper) of utterance b: e = u . tanh(keys + q [+ f . Wf]), -1e8 replace-mask, and the
//     slice's softmax statistics (max, sum of exp).  grid (nsplit, B).
// CT: the conv's channel count at compile time (10, the reference's default: the per-channel loops unroll and their LDS reads pipeline -- as
// runtime loops they were one LDS round trip per channel and frame) or 0 = any.
// FUSED: the slice's energies and statistics leave as granules for the utterance's other workgroups of the SAME launch (wide_attend_kernel).
template <bool FAST, bool LOC, int CT, bool FUSED>
__device__ __forceinline__ void wide_energy_body(const DecDev& a, const WideDev& w, const int t, const int s, const int b, float* sm) {
    const WideLds L = wide_carve(sm, a, w.fper);
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t grs = granule_rsrc(w.egran);
    const unsigned grow = (unsigned)b * (unsigned)wide_gran_row(a.Tp), gtag = (unsigned)t + 1u;
    const int B = a.B, Tp = a.Tp, A = a.A, C = CT > 0 ? CT : a.C;
    const int t0 = s * w.fper, nf = (Tp - t0 < w.fper ? Tp - t0 : w.fper);
    WSTAMP(0);
    for (int i = tid; i < A; i += RNT) L.qv[i] = w.qbuf[(size_t)b * A + i];
    if (LOC) {
        wide_stage_awin(a, L.aprev, t, b, t0, w.fper, tid);
        wide_stage_filter(a, L.locw, tid);
        for (int i = tid; i < C * A; i += RNT) L.wfl[i] = a.Wf[i];
    }
    __syncthreads();
    WSTAMP(1);
    if (nf <= 0) {
        if (tid == 0) {
            if (FUSED) {
                granule8_store(grs, (grow + Tp + 2 * s) * 8u, gtag, __float_as_uint(-INFINITY), w.xcd_local != 0);
                granule8_store(grs, (grow + Tp + 2 * s + 1) * 8u, gtag, __float_as_uint(0.f), w.xcd_local != 0);
            } else { w.stat[((size_t)b * w.nsplit + s) * 2] = -INFINITY; w.stat[((size_t)b * w.nsplit + s) * 2 + 1] = 0.f; }
        }
        return;
    }
    // 32 lanes per frame (lane sl: attention columns 4 sl .. 4 sl + 3 and 128 + 4 sl ..), 32 frames at a time.  The keys of a group's first two
    // frames are requested in front of the conv: they do not depend on it, and behind it their latency was on the chain once per frame
    const int sl = tid & 31, grp = tid >> 5, A4 = A / 4;
    float4 kq[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int slot = 0; slot < 2; ++slot)
            kq[p][slot] = (grp + p * RNG < nf && sl + 32 * slot < A4) ? wide_key4<FAST>(a, b, t0 + grp + p * RNG, sl + 32 * slot) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 uq[2];
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) uq[slot] = sl + 32 * slot < A4 ? reinterpret_cast<const float4*>(a.u)[sl + 32 * slot] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (LOC) {
        wide_conv_slice(a, L.aprev, L.locw, L.fc, L.part, nf, tid);
        WSTAMP(2);
        if (a.fcSave) {              // kept for the gradient loop and the after-loop filter / keys gradients
            float* fs = a.fcSave + (((size_t)t * B + b) * Tp + t0) * C;
            for (int i = tid; i < nf * C; i += RNT) fs[i] = L.fc[i];
            if (a.actS && t == 0 && b == 0 && s == 0 && tid == 0) a.actS[0] = LAS_ACT_MAGIC_WIDE;
        }
    }
    WSTAMP(3);
    const int len = a.enc_len[b];
    for (int fr0 = grp; fr0 < nf; fr0 += 2 * RNG) {
        if (fr0 != grp) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int slot = 0; slot < 2; ++slot)
                    if (fr0 + p * RNG < nf && sl + 32 * slot < A4) kq[p][slot] = wide_key4<FAST>(a, b, t0 + fr0 + p * RNG, sl + 32 * slot);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fr = fr0 + p * RNG, tt = t0 + fr;
            if (fr >= nf) break;
            float part = 0.f;
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int a4 = sl + 32 * slot;
                if (a4 >= A4) break;
                const float4 k4 = kq[p][slot];
                const float4 q4 = reinterpret_cast<const float4*>(L.qv)[a4];
                const float4 u4 = uq[slot];
                float4 p4 = make_float4(k4.x + q4.x, k4.y + q4.y, k4.z + q4.z, k4.w + q4.w);
                if (LOC) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float f = L.fc[fr * C + c];
                        const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                        p4.x = fmaf(f, w4.x, p4.x); p4.y = fmaf(f, w4.y, p4.y); p4.z = fmaf(f, w4.z, p4.z); p4.w = fmaf(f, w4.w, p4.w);
                    }
                }
                part += u4.x * tanhx<FAST>(p4.x) + u4.y * tanhx<FAST>(p4.y) + u4.z * tanhx<FAST>(p4.z) + u4.w * tanhx<FAST>(p4.w);
            }
            const float e = sub32_sum(part);
            if (sl == 0) {
                const float em = (tt < len) ? e : -1e8f;           // replace-mask, las/layers.py:205-207
                L.ev[fr] = em;
                if (FUSED) granule8_store(grs, (grow + tt) * 8u, gtag, __float_as_uint(em), w.xcd_local != 0);
                else w.ebuf[(size_t)b * Tp + tt] = em;
            }
        }
    }
    __syncthreads();
    WSTAMP(4);
    float m = -INFINITY;
    for (int i = tid; i < nf; i += RNT) m = fmaxf(m, L.ev[i]);
    m = block_max<RNT>(m, L.red);
    float ssum = 0.f;
    for (int i = tid; i < nf; i += RNT) ssum += expf(L.ev[i] - m);
    ssum = block_sum<RNT>(ssum, L.red);
    if (tid == 0) {
        if (FUSED) {
            granule8_store(grs, (grow + Tp + 2 * s) * 8u, gtag, __float_as_uint(m), w.xcd_local != 0);
            granule8_store(grs, (grow + Tp + 2 * s + 1) * 8u, gtag, __float_as_uint(ssum), w.xcd_local != 0);
        } else { w.stat[((size_t)b * w.nsplit + s) * 2] = m; w.stat[((size_t)b * w.nsplit + s) * 2 + 1] = ssum; }
    }
    WSTAMP(5);
}
template <bool FAST, bool LOC, int CT = 0>
__global__ __launch_bounds__(RNT) void wide_energy_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    wide_energy_body<FAST, LOC, CT, false>(a, w, t, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// (3) alignment (softmax over all frames from the slices' statistics), the context columns [4 c0, 4 c1) of utterance b, and the cell input
//     row [emb(token) ; context ; h_0] (fp32 for the weight gradients, bf16 as the product's A operand).  grid (hsplit, B).
template <bool FAST, bool FUSED>
__device__ __forceinline__ void wide_context_body(const DecDev& a, const WideDev& w, const int t, const int hs_, const int b, float* sm) {
    const int tid = threadIdx.x;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, D = a.D, E = a.E, V = a.V, U = a.U, I0D = E + Hd + D, H4 = Hd / 4;
    float* al = sm;                                  // [Tp] alignment (speed mode: rounded to bf16, the contraction's operand)
    float* part = sm + ((Tp + 3) & ~3);              // [ng][h4per] float4
    WSTAMP(10);
    // everything that does not depend on the alignment is requested first: the first four encoder rows of this thread's column chunk, and (slice
    // 0) the embedding row and h_0 -- behind the softmax each of them was a memory round trip of its own on the launch's critical path
    const int len = a.enc_len[b];
    const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;   // alpha is exactly 0 beyond len (exp underflow)
    const int c0 = hs_ * w.h4per, nch = (H4 - c0 < w.h4per ? H4 - c0 : w.h4per);
    const int ng = RNT / w.h4per;
    const int g = tid / w.h4per, ch = tid - g * w.h4per;
    const bool mine = nch > 0 && g < ng && ch < nch;
    auto enc4 = [&](int tu) -> float4 {
        if (FAST) {
            const uint2 v = reinterpret_cast<const uint2*>(a.encbf + ((size_t)b * Tp + tu) * Hd)[c0 + ch];
            return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        }
        return reinterpret_cast<const float4*>(a.enc + ((size_t)b * Tp + tu) * Hd)[c0 + ch];
    };
    const bool havepre = mine && g + 3 * ng < lim;
    float4 pre[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[u] = havepre ? enc4(g + u * ng) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* xrow = a.xin0 + ((size_t)t * B + b) * I0D;
    unsigned short* xb = FAST ? a.xbf + (size_t)b * I0D : nullptr;
    float ev = 0.f, hv = 0.f;
    int tok = 0;
    if (hs_ == 0) {
        tok = a.tok_in[(size_t)t * B + b];
        if (tid < E) ev = (a.emb[(size_t)tok * E + tid] + (a.emb_noise ? a.emb_noise[((size_t)t * V + tok) * E + tid] : 0.f)) *
                          (a.emb_mask ? a.emb_mask[((size_t)t * B + b) * E + tid] : 1.f);
        if (tid < D) hv = a.hs[(((size_t)0 * (U + 1) + t) * B + b) * D + tid];
    }
    if (FUSED) {      // the utterance's energies and slice statistics from this launch's other workgroups -> LDS (al: raw energies; part: statistics)
        const __amdgpu_buffer_rsrc_t grs = granule_rsrc(w.egran);
        const unsigned grow = (unsigned)b * (unsigned)wide_gran_row(Tp), gtag = (unsigned)t + 1u;
        const int budget = wide_poll_budget(a);
        for (int i = tid; i < Tp + 2 * w.nsplit; i += RNT) {
            const float v = __uint_as_float(wide_poll(a, grs, (grow + i) * 8u, gtag, budget));
            if (i < Tp) al[i] = v; else part[i - Tp] = v;
        }
        __syncthreads();
    }
    auto st = [&](int s, int j) -> float { return FUSED ? part[2 * s + j] : w.stat[((size_t)b * w.nsplit + s) * 2 + j]; };
    float m = -INFINITY;
    for (int s = 0; s < w.nsplit; ++s) m = fmaxf(m, st(s, 0));
    float l = 0.f;
    for (int s = 0; s < w.nsplit; ++s) {
        const float ms = st(s, 0), ls = st(s, 1);
        if (ls > 0.f) l += ls * expf(ms - m);
    }
    const float inv = 1.0f / l;
    float* arow = a.alphas + ((size_t)t * B + b) * Tp;
    for (int i = tid; i < Tp; i += RNT) {
        const float v = expf((FUSED ? al[i] : w.ebuf[(size_t)b * Tp + i]) - m) * inv;
        if (hs_ == 0) arow[i] = v;
        al[i] = FAST ? bf2f(f2bf(v)) : v;
    }
    __syncthreads();
    WSTAMP(11);
    if (mine) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int tt = g;
        if (havepre) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float av = al[tt + u * ng];
                acc.x = fmaf(av, pre[u].x, acc.x); acc.y = fmaf(av, pre[u].y, acc.y); acc.z = fmaf(av, pre[u].z, acc.z); acc.w = fmaf(av, pre[u].w, acc.w);
            }
            tt += 4 * ng;
        }
        for (; tt + 3 * ng < lim; tt += 4 * ng) {          // four independent loads in flight
            float4 e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e4[u] = enc4(tt + u * ng);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float av = al[tt + u * ng];
                acc.x = fmaf(av, e4[u].x, acc.x); acc.y = fmaf(av, e4[u].y, acc.y); acc.z = fmaf(av, e4[u].z, acc.z); acc.w = fmaf(av, e4[u].w, acc.w);
            }
        }
        for (; tt < lim; tt += ng) {
            const float4 e4 = enc4(tt);
            const float av = al[tt];
            acc.x = fmaf(av, e4.x, acc.x); acc.y = fmaf(av, e4.y, acc.y); acc.z = fmaf(av, e4.z, acc.z); acc.w = fmaf(av, e4.w, acc.w);
        }
        reinterpret_cast<float4*>(part)[g * w.h4per + ch] = acc;
    }
    __syncthreads();
    WSTAMP(12);
    for (int i = tid; i < nch * 4; i += RNT) {
        const int chq = i >> 2, e = i & 3;
        float cv = 0.f;
        for (int gg = 0; gg < ng; ++gg) cv += part[(gg * w.h4per + chq) * 4 + e];
        const int col = (c0 + chq) * 4 + e;
        xrow[E + col] = cv;
        if (FAST) xb[E + col] = f2bf(cv);
    }
    if (hs_ == 0) {
        for (int i = tid; i < E; i += RNT) {
            const float v = i == tid ? ev : (a.emb[(size_t)tok * E + i] + (a.emb_noise ? a.emb_noise[((size_t)t * V + tok) * E + i] : 0.f)) *
                                           (a.emb_mask ? a.emb_mask[((size_t)t * B + b) * E + i] : 1.f);
            xrow[i] = v;
            if (FAST) xb[i] = f2bf(v);
        }
        for (int i = tid; i < D; i += RNT) {
            const float v = i == tid ? hv : a.hs[(((size_t)0 * (U + 1) + t) * B + b) * D + i];
            xrow[E + Hd + i] = v;
            if (FAST) xb[E + Hd + i] = f2bf(v);
        }
    }
    WSTAMP(13);
}
template <bool FAST>
__global__ __launch_bounds__(RNT) void wide_context_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    wide_context_body<FAST, false>(a, w, t, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// (2) + (3) as ONE launch: workgroup (s, b) computes the energies of frame slice s, publishes them and the slice's statistics as tagged
// granules (write-through stores; the data is the flag), then -- as context slice s -- collects the utterance's energies from its peers and
// goes on with the alignment and its context columns.  Every workgroup of an utterance must be resident for that (the host launches this
// form only when max(nsplit, hsplit) B workgroups fit the device's compute units; a partner that never arrives ends the polls after
// lp.budget rounds and is reported through the status word -- las.layers.fallback_schedule then re-runs the step with the two launches).
// One kernel boundary (~4 us of launch ramp, drain and cold staging) per decode step becomes one hop through memory (~1.5 us).
template <bool FAST, bool LOC, int CT = 0>
__global__ __launch_bounds__(RNT) void wide_attend_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int s, b;
    wide_slice_of(w, a.B, s, b);
    if (b < 0) return;
    if (s < w.nsplit) wide_energy_body<FAST, LOC, CT, true>(a, w, t, s, b, sm);
    __syncthreads();                                    // (the context phase re-uses the energies' LDS)
    if (s < w.hsplit) wide_context_body<FAST, true>(a, w, t, s, b, sm);
}

// gate nonlinearity of layer `layer` (< TOP) at step t, and the bf16 input row of the layer above: [h_{layer, t+1} ; h_{layer+1, t}]
template <int CELL, bool FAST>
__global__ __launch_bounds__(256) void wide_pointwise_fwd_kernel(DecDev a, WideDev w, int layer, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int b = blockIdx.x, B = a.B, D = a.D, U = a.U, GD = G * D;
    float* gp = a.gates + (((size_t)layer * U + t) * B + b) * GD;
    float* hnew = a.hs + (((size_t)layer * (U + 1) + t + 1) * B + b) * D;
    const float* hup = a.hs + (((size_t)(layer + 1) * (U + 1) + t) * B + b) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float h;
        if (CELL == LAS_CELL_LSTM) {
            const float* cprev = a.cs + (((size_t)layer * (U + 1) + t) * B + b) * D;
            float* cnew = a.cs + (((size_t)layer * (U + 1) + t + 1) * B + b) * D;
            const float gi = sigm<FAST>(gp[d]), gj = tanhx<FAST>(gp[D + d]);
            const float gf = sigm<FAST>(gp[2 * D + d] + a.fb), go = sigm<FAST>(gp[3 * D + d]);
            const float c = cprev[d] * gf + gi * gj;
            h = tanhx<FAST>(c) * go;
            gp[d] = gi; gp[D + d] = gj; gp[2 * D + d] = gf; gp[3 * D + d] = go;
            cnew[d] = c;
        } else {
            h = tanhx<FAST>(gp[d]);
        }
        hnew[d] = h;
        if (FAST) {
            w.xu[(size_t)b * 2 * D + d] = f2bf(h);
            w.xu[(size_t)b * 2 * D + D + d] = f2bf(hup[d]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// (1) d alpha[t'] = dctx . enc[b, t', :] (+ what step t+1's location conv sent back) for the frames of slice s, and the slice's part of
//     alpha . d alpha.  grid (nsplit, B).
template <bool FAST, bool LOC, bool FUSED>
__device__ __forceinline__ void wide_dalpha_body(const DecDev& a, const WideDev& w, const int t, const int s, const int b, float* sm) {
    float* dctx = sm;                                  // [Hd]
    float* red = sm + ((a.Hd + 3) & ~3);               // [32]
    const int tid = threadIdx.x;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, E = a.E, U = a.U, I0D = E + Hd + a.D;
    const int t0 = s * w.fper, nf = (Tp - t0 < w.fper ? Tp - t0 : w.fper);
    WSTAMP(20);
    // 32 lanes per frame, 32 frames at a time; the encoder rows of a group's first two frames (up to 4 x 16 bytes per lane and frame: Hd <= 1024
    // in speed mode, 512 in parity mode) are requested in front of the staging barrier -- they do not depend on d context
    const int sl = tid & 31, grp = tid >> 5;
    constexpr int NP = 4;
    const int nvec = FAST ? Hd / 8 : Hd / 4;           // 16-byte pieces of a row
    uint4 pre[2][NP];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int fr = grp + p * RNG, h = sl + 32 * q;
            pre[p][q] = make_uint4(0u, 0u, 0u, 0u);
            if (fr < nf && h < nvec)
                pre[p][q] = FAST ? reinterpret_cast<const uint4*>(a.encbf + ((size_t)b * Tp + t0 + fr) * Hd)[h]
                                 : reinterpret_cast<const uint4*>(a.enc + ((size_t)b * Tp + t0 + fr) * Hd)[h];
        }
    const float* dxr = a.dXin0 + ((size_t)t * B + b) * I0D + E;
    for (int i = tid; i < Hd; i += RNT) dctx[i] = FAST ? bf2f(f2bf(dxr[i])) : dxr[i];
    __syncthreads();
    WSTAMP(21);
    float dot = 0.f;
    auto piece = [&](const uint4 v, const int h) -> float {          // one 16-byte piece of an encoder row against d context
        if (FAST) {
            float e[8];
            unpack8(v, e);
            const float4 d0 = reinterpret_cast<const float4*>(dctx)[2 * h], d1 = reinterpret_cast<const float4*>(dctx)[2 * h + 1];
            return d0.x * e[0] + d0.y * e[1] + d0.z * e[2] + d0.w * e[3] + d1.x * e[4] + d1.y * e[5] + d1.z * e[6] + d1.w * e[7];
        }
        const float4 d4 = reinterpret_cast<const float4*>(dctx)[h];
        return d4.x * __uint_as_float(v.x) + d4.y * __uint_as_float(v.y) + d4.z * __uint_as_float(v.z) + d4.w * __uint_as_float(v.w);
    };
    auto row16 = [&](int tt, int h) -> uint4 {
        return FAST ? reinterpret_cast<const uint4*>(a.encbf + ((size_t)b * Tp + tt) * Hd)[h] : reinterpret_cast<const uint4*>(a.enc + ((size_t)b * Tp + tt) * Hd)[h];
    };
    auto finish = [&](float acc, int tt) {
        float v = sub32_sum(acc);
        if (sl == 0) {
            if (LOC && t + 1 < U) v += a.dAext[(size_t)b * Tp + tt];
            w.ebuf[(size_t)b * Tp + tt] = v;
            dot = fmaf(a.alphas[((size_t)t * B + b) * Tp + tt], v, dot);
        }
    };
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int fr = grp + p * RNG, tt = t0 + fr;
        if (fr >= nf) break;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (sl + 32 * q < nvec) acc += piece(pre[p][q], sl + 32 * q);
        for (int h = sl + 32 * NP; h < nvec; h += 32) acc += piece(row16(tt, h), h);
        finish(acc, tt);
    }
    for (int fr = grp + 2 * RNG; fr < nf; fr += RNG) {
        const int tt = t0 + fr;
        float acc = 0.f;
        for (int h = sl; h < nvec; h += 32) acc += piece(row16(tt, h), h);
        finish(acc, tt);
    }
    WSTAMP(22);
    dot = block_sum<RNT>(dot, red);
    if (tid == 0) {
        if (FUSED) granule8_store(granule_rsrc(w.bgran), ((unsigned)b * (unsigned)wide_bgran_row(Tp) + s) * 8u, (unsigned)t + 1u, __float_as_uint(dot), w.xcd_local != 0);
        else w.stat[(size_t)b * w.nsplit + s] = dot;
    }
    WSTAMP(23);
}
template <bool FAST, bool LOC>
__global__ __launch_bounds__(RNT) void wide_dalpha_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    wide_dalpha_body<FAST, LOC, false>(a, w, t, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// (2) d energy of the slice's frames (kept for the after-loop keys gradient), the energies' backward: partial dq / du over the slice, and
//     d f[t', c] = sum_a dv[a] Wf[c, a] (kept for the conv's transpose and the filter gradient).  grid (nsplit, B).
template <bool FAST, bool LOC, int CT, bool FUSED>
__device__ __forceinline__ void wide_energy_bwd_body(const DecDev& a, const WideDev& w, const int t, const int s, const int b, float* sm) {
    const WideLds L = wide_carve(sm, a, w.fper);
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t grs = granule_rsrc(w.bgran);
    const unsigned grow = (unsigned)b * (unsigned)wide_bgran_row(a.Tp), gtag = (unsigned)t + 1u;
    const int B = a.B, Tp = a.Tp, A = a.A, C = CT > 0 ? CT : a.C;
    const int t0 = s * w.fper, nf = (Tp - t0 < w.fper ? Tp - t0 : w.fper);
    float* dfc = L.dfc;
    WSTAMP(30);
    // the keys of a group's first two frames are requested first (32 lanes per frame, as in the forward kernel)
    const int sl = tid & 31, grp = tid >> 5, A4 = A / 4;
    float4 kq[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int slot = 0; slot < 2; ++slot)
            kq[p][slot] = (grp + p * RNG < nf && sl + 32 * slot < A4) ? wide_key4<FAST>(a, b, t0 + grp + p * RNG, sl + 32 * slot) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 uq[2];
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) uq[slot] = sl + 32 * slot < A4 ? reinterpret_cast<const float4*>(a.u)[sl + 32 * slot] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < A; i += RNT) L.qv[i] = a.Q[((size_t)t * B + b) * A + i];
    const bool havef = LOC && a.actS && a.fcSave && a.actS[0] == LAS_ACT_MAGIC_WIDE;     // the forward kept f of every step
    if (LOC) {
        for (int i = tid; i < C * A; i += RNT) L.wfl[i] = a.Wf[i];
        if (!havef) {
            wide_stage_awin(a, L.aprev, t, b, t0, w.fper, tid);
            wide_stage_filter(a, L.locw, tid);
        }
    }
    __syncthreads();
    WSTAMP(31);
    float dot = 0.f;
    if (FUSED) {      // the slices' alpha . d alpha from this launch's other workgroups (every lane polls the same few granules)
        const int budget = wide_poll_budget(a);
        u32x2_t g[WIDE_MAX_SPLIT];
#pragma unroll
        for (int q = 0; q < WIDE_MAX_SPLIT; ++q) g[q] = granule8_load(grs, (grow + (q < w.nsplit ? q : 0)) * 8u);
#pragma unroll
        for (int q = 0; q < WIDE_MAX_SPLIT; ++q)
            if (q < w.nsplit) dot += __uint_as_float(wide_poll_loaded(a, grs, (grow + q) * 8u, gtag, budget, g[q]));
    } else {
        for (int q = 0; q < w.nsplit; ++q) dot += w.stat[(size_t)b * w.nsplit + q];
    }
    if (nf > 0) {
        if (LOC) {
            if (havef) {
                const float* fs = a.fcSave + (((size_t)t * B + b) * Tp + t0) * C;
                for (int i = tid; i < nf * C; i += RNT) L.fc[i] = fs[i];
            } else {
                wide_conv_slice(a, L.aprev, L.locw, L.fc, L.part, nf, tid);
                if (a.fcSave) {      // the after-loop keys / Wf gradient reads f of every step
                    float* fs = a.fcSave + (((size_t)t * B + b) * Tp + t0) * C;
                    for (int i = tid; i < nf * C; i += RNT) fs[i] = L.fc[i];
                }
            }
        }
        for (int i = tid; i < nf; i += RNT) {
            const int tt = t0 + i;
            const float de = a.alphas[((size_t)t * B + b) * Tp + tt] * (w.ebuf[(size_t)b * Tp + tt] - dot);    // 0 where masked: alpha = 0
            L.ev[i] = de;
            a.dE[((size_t)t * B + b) * Tp + tt] = de;
        }
    }
    __syncthreads();
    WSTAMP(32);
    float du_acc[8], dq_acc[8];                    // A <= 256: at most two float4 per lane
#pragma unroll
    for (int i = 0; i < 8; ++i) { du_acc[i] = 0.f; dq_acc[i] = 0.f; }
    for (int fr0 = grp; fr0 < nf; fr0 += 2 * RNG) {
        if (fr0 != grp) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int slot = 0; slot < 2; ++slot)
                    if (fr0 + p * RNG < nf && sl + 32 * slot < A4) kq[p][slot] = wide_key4<FAST>(a, b, t0 + fr0 + p * RNG, sl + 32 * slot);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fr = fr0 + p * RNG;
            if (fr >= nf) break;
            const float de = L.ev[fr];
            float4 dvs[2];
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int a4 = sl + 32 * slot;
                dvs[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a4 >= A4) continue;
                const float4 k4 = kq[p][slot];
                const float4 q4 = reinterpret_cast<const float4*>(L.qv)[a4];
                const float4 u4 = uq[slot];
                float4 p4 = make_float4(k4.x + q4.x, k4.y + q4.y, k4.z + q4.z, k4.w + q4.w);
                if (LOC) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float f = L.fc[fr * C + c];
                        const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                        p4.x = fmaf(f, w4.x, p4.x); p4.y = fmaf(f, w4.y, p4.y); p4.z = fmaf(f, w4.z, p4.z); p4.w = fmaf(f, w4.w, p4.w);
                    }
                }
                const float vx = tanhx<FAST>(p4.x), vy = tanhx<FAST>(p4.y), vz = tanhx<FAST>(p4.z), vw = tanhx<FAST>(p4.w);
                const float4 dv = make_float4(de * u4.x * (1.f - vx * vx), de * u4.y * (1.f - vy * vy), de * u4.z * (1.f - vz * vz), de * u4.w * (1.f - vw * vw));
                du_acc[slot * 4 + 0] += de * vx; du_acc[slot * 4 + 1] += de * vy; du_acc[slot * 4 + 2] += de * vz; du_acc[slot * 4 + 3] += de * vw;
                dq_acc[slot * 4 + 0] += dv.x; dq_acc[slot * 4 + 1] += dv.y; dq_acc[slot * 4 + 2] += dv.z; dq_acc[slot * 4 + 3] += dv.w;
                dvs[slot] = dv;
            }
            if (LOC) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float s1 = 0.f;
#pragma unroll
                    for (int slot = 0; slot < 2; ++slot) {
                        const int a4 = sl + 32 * slot;
                        if (a4 < A4) {
                            const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                            s1 += dvs[slot].x * w4.x + dvs[slot].y * w4.y + dvs[slot].z * w4.z + dvs[slot].w * w4.w;
                        }
                    }
                    s1 = sub32_sum(s1);
                    if (sl == 0) dfc[fr * C + c] = s1;
                }
            }
        }
    }
    WSTAMP(33);
    // the 32 frame groups' partials of dq and du through LDS, summed in fixed order (groups ascending): dq by threads 0 .. A - 1, du by the
    // next A (one trip; round 6, first version: two trips with four barriers).  (Nobody has touched `part` since the conv's own barriers.)
    float* pq = L.part;
    float* pu = L.part + (size_t)RNG * A;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int a4 = sl + 32 * slot;
        if (a4 < A4) {
            reinterpret_cast<float4*>(pq + (size_t)grp * A)[a4] = make_float4(dq_acc[slot * 4], dq_acc[slot * 4 + 1], dq_acc[slot * 4 + 2], dq_acc[slot * 4 + 3]);
            reinterpret_cast<float4*>(pu + (size_t)grp * A)[a4] = make_float4(du_acc[slot * 4], du_acc[slot * 4 + 1], du_acc[slot * 4 + 2], du_acc[slot * 4 + 3]);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * A; i += RNT) {
        const bool second = i >= A;
        const int col = second ? i - A : i;
        const float* src = second ? pu : pq;
        float v = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < RNG; ++g8) v += src[g8 * A + col];
        if (FUSED) granule8_store(grs, (grow + WIDE_BG_PART + (s * 2 + (second ? 1 : 0)) * 256 + col) * 8u, gtag, __float_as_uint(v), w.xcd_local != 0);
        else ((second ? w.pdu : w.pdq) + ((size_t)b * w.nsplit + s) * A)[col] = v;
    }
    WSTAMP(34);
    if (LOC && nf > 0 && a.dfcSave) {
        float* ds = a.dfcSave + (((size_t)t * B + b) * Tp + t0) * C;
        for (int i = tid; i < nf * C; i += RNT) {
            ds[i] = dfc[i];
            if (FUSED) { const int fr = i / C; granule8_store(grs, (grow + WIDE_BG_DF + (t0 + fr) * 16 + (i - fr * C)) * 8u, gtag, __float_as_uint(dfc[i]), w.xcd_local != 0); }
        }
    }
    WSTAMP(35);
}
template <bool FAST, bool LOC, int CT = 0>
__global__ __launch_bounds__(RNT) void wide_energy_bwd_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    wide_energy_bwd_body<FAST, LOC, CT, false>(a, w, t, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// the transposed conv's tap-slice partials: part[kc, j] = sum over the taps k of slice kc, c ascending inside a tap, of rows[j - k + Kc + 1, c]
// w[k, c] for the source frames j of the slice (rows: see wide_dq_kernel).  CT = C at compile time: one thread = (tap slice, 4 consecutive
// source frames) with the 4 + 1 rows two taps need in registers -- per pair of taps one new row pair and two filter rows from LDS for 8 CT
// multiply-adds (round 6, first version: one (tap slice, frame) per thread, 2 LDS reads per multiply-add: 6.5 of the kernel's 9 us); CT = 0:
// any C, one (tap slice, frame) per thread.  Same taps per slice and the same order inside a slice in both.
template <int CT>
__device__ __forceinline__ void wide_convT_items(const float* rows, const float* locw, float* part, int nf, int Kc, int C, int NKC, int kper, int tid) {
    if (CT > 0) {
        const int nsb = (nf + 3) >> 2;
        for (int i = tid; i < NKC * nsb; i += RNT) {
            const int kc = i / nsb, sb = i - kc * nsb;
            const int k0 = kc * kper, kend = k0 + kper < Kc ? k0 + kper : Kc;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            float W[5][CT > 0 ? CT : 1];
            const float* rb = rows + (size_t)(sb * 4 - k0 + Kc + 1) * CT;          // row of (source frame 4 sb, tap k0)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < CT; ++c) W[j + 1][c] = rb[j * CT + c];
            for (int k = k0; k < kend; k += 2) {
                float wa[CT > 0 ? CT : 1], wb[CT > 0 ? CT : 1];
                const bool two = k + 1 < kend;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    W[0][c] = rb[c - CT];                                          // (source frame 4 sb, tap k + 1)
                    wa[c] = locw[k * CT + c];
                    const float wn = locw[(k + 1) * CT + c];                       // (row Kc of the filter's LDS copy is never used: `two`)
                    wb[c] = two ? wn : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[j] = fmaf(W[j + 1][c], wa[c], acc[j]);
                if (two) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[j] = fmaf(W[j][c], wb[c], acc[j]);
                }
                rb -= 2 * CT;
#pragma unroll
                for (int c = 0; c < CT; ++c) { W[4][c] = W[2][c]; W[3][c] = W[1][c]; W[2][c] = W[0][c]; W[1][c] = rb[c]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sb * 4 + j < nf) part[kc * nf + sb * 4 + j] = acc[j];
        }
    } else {
        for (int i = tid; i < NKC * nf; i += RNT) {
            const int kc = i / nf, j = i - kc * nf;
            const int k0 = kc * kper, kend = k0 + kper < Kc ? k0 + kper : Kc;
            float acc = 0.f;
            for (int k = k0; k < kend; ++k) {
                const float* dr = rows + (size_t)(j - k + Kc + 1) * C;
                const float* wr = locw + k * C;
                for (int c = 0; c < C; ++c) acc = fmaf(dr[c], wr[c], acc);
            }
            part[i] = acc;
        }
    }
}

// (3) dq = sum of the slices' partials (-> dQ of the step, bf16 operand row of d s = dq . Ws^T) and du (slice 0 of an utterance), and the
//     conv's transpose d alpha_{t-1}[src] = sum_k sum_c d f[src - k + pad, c] w[k, c] for the source frames of slice s from the step's d f
//     rows.  grid (nsplit, B)  (first version: one workgroup per utterance -- 319 x 201 x 10 multiply-adds on 48 CUs, 31.7 us per step).
template <bool FAST, bool LOC, bool FUSED>
__device__ __forceinline__ void wide_dq_body(const DecDev& a, const WideDev& w, const int t, const int s, const int b, float* sm) {
    const int tid = threadIdx.x, B = a.B, Tp = a.Tp, A = a.A, C = a.C;
    const __amdgpu_buffer_rsrc_t grs = granule_rsrc(w.bgran);
    const unsigned grow = (unsigned)b * (unsigned)wide_bgran_row(Tp), gtag = (unsigned)t + 1u;
    const int budget = FUSED ? wide_poll_budget(a) : 0;
    WSTAMP(40);
    if (s == 0) {
        for (int i = tid; i < A; i += RNT) {
            float dq = 0.f, du = 0.f;
            if (FUSED) {
                u32x2_t gq[WIDE_MAX_SPLIT], gu[WIDE_MAX_SPLIT];
#pragma unroll
                for (int q = 0; q < WIDE_MAX_SPLIT; ++q) {
                    const int qc = q < w.nsplit ? q : 0;
                    gq[q] = granule8_load(grs, (grow + WIDE_BG_PART + (qc * 2) * 256 + i) * 8u);
                    gu[q] = granule8_load(grs, (grow + WIDE_BG_PART + (qc * 2 + 1) * 256 + i) * 8u);
                }
#pragma unroll
                for (int q = 0; q < WIDE_MAX_SPLIT; ++q) {
                    if (q < w.nsplit) {
                        dq += __uint_as_float(wide_poll_loaded(a, grs, (grow + WIDE_BG_PART + (q * 2) * 256 + i) * 8u, gtag, budget, gq[q]));
                        du += __uint_as_float(wide_poll_loaded(a, grs, (grow + WIDE_BG_PART + (q * 2 + 1) * 256 + i) * 8u, gtag, budget, gu[q]));
                    }
                }
            } else {
                for (int q = 0; q < w.nsplit; ++q) {
                    dq += w.pdq[((size_t)b * w.nsplit + q) * A + i];
                    du += w.pdu[((size_t)b * w.nsplit + q) * A + i];
                }
            }
            a.dQ[((size_t)t * B + b) * A + i] = dq;
            if (FAST) w.dqbf[(size_t)b * A + i] = f2bf(dq);
            a.duRows[(size_t)b * A + i] += du;
        }
    }
    WSTAMP(41);
    if (LOC && t > 0) {
        const int Kc = a.Kc, pad = (Kc - 1) / 2;
        const int t0 = s * w.fper, nf = (Tp - t0 < w.fper ? Tp - t0 : w.fper);
        if (nf <= 0) return;
        // d f rows that reach the slice's source frames, zero where the frame does not exist: local row rho <-> frame t0 + rho - (Kc + 1) + pad,
        // so that source frame t0 + j meets tap k at row j - k + Kc + 1 and no tap needs a bound
        const int nrows = ((nf + 3) & ~3) + Kc + 2;
        float* rows = sm;                                                          // [nrows, C]
        float* locw = rows + (((((w.fper + 3) & ~3) + Kc + 2) * C + 3) & ~3);      // [Kc + 2, C]
        float* part = locw + (((Kc + 2) * C + 3) & ~3);                            // [NKC, nf]
        const float* ds = a.dfcSave + ((size_t)t * B + b) * Tp * C;
        if (FUSED) {      // four granules per lane requested together, then checked
            for (int i0 = tid; i0 < nrows * C; i0 += 4 * RNT) {
                u32x2_t g[4];
                unsigned off[4];
                bool on[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * RNT, ic = i < nrows * C ? i : 0;
                    const int rho = ic / C, fr = t0 + rho - (Kc + 1) + pad;
                    on[k] = i < nrows * C && fr >= 0 && fr < Tp;
                    off[k] = (grow + WIDE_BG_DF + (on[k] ? fr : 0) * 16 + (ic - rho * C)) * 8u;
                    g[k] = granule8_load(grs, off[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * RNT;
                    if (i < nrows * C) rows[i] = on[k] ? __uint_as_float(wide_poll_loaded(a, grs, off[k], gtag, budget, g[k])) : 0.f;
                }
            }
        } else {
            for (int i = tid; i < nrows * C; i += RNT) {
                const int rho = i / C, fr = t0 + rho - (Kc + 1) + pad;
                rows[i] = (fr >= 0 && fr < Tp) ? ds[(size_t)fr * C + (i - rho * C)] : 0.f;
            }
        }
        for (int i = tid; i < Kc * C; i += RNT) locw[i] = a.loc_w[i];
        __syncthreads();
        WSTAMP(42);
        int NKC = RNT / nf;
        NKC = NKC < 1 ? 1 : (NKC > 16 ? 16 : NKC);
        const int kper = (Kc + NKC - 1) / NKC;
        if (C == 10) wide_convT_items<10>(rows, locw, part, nf, Kc, C, NKC, kper, tid);
        else         wide_convT_items<0>(rows, locw, part, nf, Kc, C, NKC, kper, tid);
        __syncthreads();
        WSTAMP(43);
        for (int j = tid; j < nf; j += RNT) {
            float acc = 0.f;
            for (int kc = 0; kc < NKC; ++kc) acc += part[kc * nf + j];
            a.dAext[(size_t)b * Tp + t0 + j] = acc;
        }
        WSTAMP(44);
    }
}
template <bool FAST, bool LOC>
__global__ __launch_bounds__(RNT) void wide_dq_kernel(DecDev a, WideDev w, int t) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    wide_dq_body<FAST, LOC, false>(a, w, t, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// (1) + (2) + (3) as ONE launch (the reverse counterpart of wide_attend_kernel): workgroup (s, b) computes d alpha of frame slice s and publishes the
// slice's alpha . d alpha; collects the utterance's sum from its peers; d energy, dq / du partials and d f of the slice -- published as granules
// (d f also stored plainly: the after-loop filter gradient reads it); then slice 0 sums the partials and every slice does its part of the conv's
// transpose from its neighbours' d f rows.  Two hand-overs through memory instead of two kernel boundaries; same operands, same order.
template <bool FAST, bool LOC, int CT = 0>
__global__ __launch_bounds__(RNT) void wide_attend_bwd_kernel(DecDev a, WideDev w, int t) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int s, b;
    wide_slice_of(w, a.B, s, b);
    if (b < 0) return;
    wide_dalpha_body<FAST, LOC, true>(a, w, t, s, b, sm);
    __syncthreads();                                    // (the d alpha values of the slice travel through w.ebuf: stored above, read below by other lanes)
    wide_energy_bwd_body<FAST, LOC, CT, true>(a, w, t, s, b, sm);
    __syncthreads();
    if (s == 0 || (LOC && t > 0)) wide_dq_body<FAST, LOC, true>(a, w, t, s, b, sm);
}

static size_t wide_dq_lds_bytes(const DecDev& a, int fper) {
    if (a.mode != LAS_ATT_LOC) return 64;
    auto u4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    return (u4((size_t)(((fper + 3) & ~3) + a.Kc + 2) * a.C) + u4((size_t)(a.Kc + 2) * a.C) + (size_t)16 * fper + 4) * sizeof(float) + 64;
}

// (4) gate backward of `layer` at step t: dh = [recurrent gradient from step t+1] + [d s of step t+1's attention] + extra (top layer:
//     dlogits . Wv^T of step t; below: the input gradient of the layer above, same step) -> d(pre-activation) over the saved gates, fp32
//     (weight gradients after the loop) and bf16 (the step's product).
template <int CELL, bool FAST>
__global__ __launch_bounds__(256) void wide_cell_bwd_kernel(DecDev a, WideDev w, int layer, int t, const float* rec, int rec_ld, int rec_off,
                                                            const float* dS, const float* extra, int extra_ld, unsigned short* gb_) {
    kernarg_warm<(int)(sizeof(DecDev) + sizeof(WideDev))>();
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int b = blockIdx.x, B = a.B, D = a.D, U = a.U, GD = G * D, S = D * a.NL;
    float* gp = a.gates + (((size_t)layer * U + t) * B + b) * GD;
    float* dCr = a.dC + ((size_t)layer * B + b) * D;
    unsigned short* gb = gb_ ? gb_ + (size_t)b * GD : nullptr;
    for (int d = threadIdx.x; d < D; d += 256) {
        float dh = extra[(size_t)b * extra_ld + d];
        if (rec) dh += rec[(size_t)b * rec_ld + rec_off + d];
        if (dS) dh += dS[(size_t)b * S + layer * D + d];
        if (CELL == LAS_CELL_LSTM) {
            const float gi = gp[d], gj = gp[D + d], gf = gp[2 * D + d], go = gp[3 * D + d];
            const float c = a.cs[(((size_t)layer * (U + 1) + t + 1) * B + b) * D + d];
            const float cp = a.cs[(((size_t)layer * (U + 1) + t) * B + b) * D + d];
            const float tc = tanhx<FAST>(c);
            const float dc = dCr[d] + dh * go * (1.f - tc * tc);
            dCr[d] = dc * gf;
            const float di = dc * gj * gi * (1.f - gi), dj = dc * gi * (1.f - gj * gj);
            const float df = dc * cp * gf * (1.f - gf), dO = dh * tc * go * (1.f - go);
            gp[d] = di; gp[D + d] = dj; gp[2 * D + d] = df; gp[3 * D + d] = dO;
            if (gb) { gb[d] = f2bf(di); gb[D + d] = f2bf(dj); gb[2 * D + d] = f2bf(df); gb[3 * D + d] = f2bf(dO); }
        } else {
            const float h = a.hs[(((size_t)layer * (U + 1) + t + 1) * B + b) * D + d];
            const float dp = dh * (1.f - h * h);
            gp[d] = dp;
            if (gb) gb[d] = f2bf(dp);
        }
    }
}

// After the loop: dKeys[b, t', :] += sum_t dE[t, b, t'] u (1 - tanh^2(keys + Q[t] [+ f[t] . Wf])), and in the same pass the Wf gradient
// partial of (utterance b, 8 frames): dWfW[b][slice][c][a] = sum_{t, t' in slice} f[t, b, t', c] dv[t, b, t', a]  (written whole; reduced
// over the slices by las_colsum).  workgroup = (8 frames, utterance), 32 lanes x float4 over the attention dim (A <= 256: two slots).
template <bool FAST, bool LOC, int CT = 0>
__global__ __launch_bounds__(256) void wide_dkeys_kernel(DecDev a, WideDev w, float* __restrict__ dKeys) {
    constexpr int LC = LOC ? (CT > 0 ? CT : 16) : 1;       // channel slots in registers: C itself when it is known at compile time (10), else the most the path accepts
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wf = sm;                                    // [C, A]
    float* part = sm + ((a.C * a.A + 3) & ~3);         // [8][C][A]
    const int b = blockIdx.y, fr = threadIdx.x >> 5, tt = blockIdx.x * 8 + fr, sl = threadIdx.x & 31;
    const int B = a.B, Tp = a.Tp, A = a.A, U = a.U, C = a.C;
    if (LOC) {
        for (int i = threadIdx.x; i < C * A; i += 256) wf[i] = a.Wf[i];
        __syncthreads();
    }
    const int ttc = tt < Tp ? tt : Tp - 1;
    for (int slot = 0; slot < 2; ++slot) {
        const int a4 = sl + 32 * slot;
        const bool on = tt < Tp && a4 < A / 4;
        const int a4c = a4 < A / 4 ? a4 : A / 4 - 1;
        if (slot == 1 && A / 4 <= 32) break;
        const float4 k4 = wide_key4<FAST>(a, b, ttc, a4c);
        const float4 u4 = reinterpret_cast<const float4*>(a.u)[a4c];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 dwf[LC];
#pragma unroll
        for (int c = 0; c < LC; ++c) dwf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
        for (int t = 0; t < U; ++t) {
            const float de = a.dE[((size_t)t * B + b) * Tp + ttc];
            const float4 q4 = reinterpret_cast<const float4*>(a.Q + ((size_t)t * B + b) * A)[a4c];
            float p0 = k4.x + q4.x, p1 = k4.y + q4.y, p2 = k4.z + q4.z, p3 = k4.w + q4.w;
            float f[LC];
            if (LOC) {
                const float* fr_ = a.fcSave + (((size_t)t * B + b) * Tp + ttc) * C;
#pragma unroll
                for (int c = 0; c < LC; ++c) f[c] = c < C ? fr_[c] : 0.f;
#pragma unroll
                for (int c = 0; c < LC; ++c) {
                    if (c < C) {
                        const float4 w4 = reinterpret_cast<const float4*>(wf + c * A)[a4c];
                        p0 = fmaf(f[c], w4.x, p0); p1 = fmaf(f[c], w4.y, p1); p2 = fmaf(f[c], w4.z, p2); p3 = fmaf(f[c], w4.w, p3);
                    }
                }
            }
            const float v0 = tanhx<FAST>(p0), v1 = tanhx<FAST>(p1), v2 = tanhx<FAST>(p2), v3 = tanhx<FAST>(p3);
            const float4 dv = make_float4(de * u4.x * (1.f - v0 * v0), de * u4.y * (1.f - v1 * v1), de * u4.z * (1.f - v2 * v2), de * u4.w * (1.f - v3 * v3));
            acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
            if (LOC) {
#pragma unroll
                for (int c = 0; c < LC; ++c) {
                    dwf[c].x = fmaf(f[c], dv.x, dwf[c].x); dwf[c].y = fmaf(f[c], dv.y, dwf[c].y);
                    dwf[c].z = fmaf(f[c], dv.z, dwf[c].z); dwf[c].w = fmaf(f[c], dv.w, dwf[c].w);
                }
            }
        }
        if (on) {
            float4* dk = reinterpret_cast<float4*>(dKeys + ((size_t)b * Tp + tt) * A) + a4;
            float4 o = *dk;
            o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
            *dk = o;
        }
        if (LOC) {
            if (a4 < A / 4) {
#pragma unroll
                for (int c = 0; c < LC; ++c) {
                    if (c < C) {
                        const float4 v = on ? dwf[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                        reinterpret_cast<float4*>(part + ((size_t)fr * C + c) * A)[a4] = v;
                    }
                }
            }
        }
    }
    if (LOC) {
        __syncthreads();
        float* out = w.dWfW + ((size_t)b * gridDim.x + blockIdx.x) * C * A;
        for (int i = threadIdx.x; i < C * A; i += 256) {
            float s_ = 0.f;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) s_ += part[(size_t)g8 * C * A + i];
            out[i] = s_;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct WideWs { size_t packWs, packWsT, packU[LAS_MAX_NL], packUB[LAS_MAX_NL], srow, xu, dqbf, dgu, qbuf, ebuf, stat, pdq, pdu, dS, dWfW, egran, bgran, total; };
static WideWs wide_layout(int B, int Tp, int A, int D, int NL, int G, int C) {
    WideWs w; size_t o = 0;
    const size_t S = (size_t)D * NL, GD = (size_t)G * D;
    w.packWs = o;  o += align256(las_skinny_pack_bytes((int)S, A));
    w.packWsT = o; o += align256(las_skinny_pack_bytes(A, (int)S));
    for (int l = 0; l < LAS_MAX_NL; ++l) {
        w.packU[l] = o;  o += (l >= 1 && l < NL) ? align256(las_skinny_pack_bytes(2 * D, (int)GD)) : 0;
        w.packUB[l] = o; o += (l >= 1 && l < NL) ? align256(las_skinny_pack_bytes((int)GD, 2 * D)) : 0;
    }
    w.srow = o;  o += align256((size_t)B * S * 4);
    w.xu = o;    o += align256((size_t)2 * B * 2 * D * 2);          // two rows per utterance: the tanh-epilogue products alternate (a product writes the NEXT step's half while it reads this step's)
    w.dqbf = o;  o += align256((size_t)B * A * 2);
    w.dgu = o;   o += align256((size_t)B * GD * 2);
    w.qbuf = o;  o += align256((size_t)B * A * 4);
    w.ebuf = o;  o += align256((size_t)B * Tp * 4);
    w.stat = o;  o += align256((size_t)B * WIDE_MAX_SPLIT * 2 * 4);
    w.pdq = o;   o += align256((size_t)B * WIDE_MAX_SPLIT * A * 4);
    w.pdu = o;   o += align256((size_t)B * WIDE_MAX_SPLIT * A * 4);
    w.dS = o;    o += align256((size_t)B * S * 4);
    w.dWfW = o;  o += align256(C > 0 ? (size_t)B * cdiv(Tp, 8) * C * A * 4 : 0);
    w.egran = o; o += align256((size_t)B * wide_gran_row(Tp) * 8);
    w.bgran = o; o += align256((size_t)B * wide_bgran_row(Tp) * 8);
    w.total = o;
    return w;
}

// the geometry the wide kernels serve (any number of layers / widths the Speller accepts)
static bool wide_geom_ok(const DecDev& d) {
    if (d.U < 2 || (d.A % 8) || (d.Hd % 8) || (d.D % 8) || (d.E % 8) || d.A > 256) return false;
    if (d.mode == LAS_ATT_LOC && (d.C < 1 || d.C > 16)) return false;
    return true;
}
static void wide_split(const DecDev& d, WideDev& w) {
    int ns = las_device_cus() / (d.B > 0 ? d.B : 1);
    ns = ns < 1 ? 1 : (ns > WIDE_MAX_SPLIT ? WIDE_MAX_SPLIT : ns);
    if (ns > cdiv(d.Tp, 8)) ns = cdiv(d.Tp, 8);
    w.fper = cdiv(d.Tp, ns);
    w.nsplit = cdiv(d.Tp, w.fper);
    int hs = las_device_cus() / (d.B > 0 ? d.B : 1);
    hs = hs < 1 ? 1 : (hs > 8 ? 8 : hs);
    const int H4 = d.Hd / 4;
    if (hs > H4) hs = H4;
    w.h4per = cdiv(H4, hs);
    if (w.h4per > RNT) w.h4per = RNT;
    w.hsplit = cdiv(H4, w.h4per);
}
static void wide_fill(const DecDev& d, WideDev& w, char* base, const WideWs& L) {
    wide_split(d, w);
    w.xcd_local = 0; w.sp = 0;
    w.qbuf = (float*)(base + L.qbuf); w.ebuf = (float*)(base + L.ebuf); w.stat = (float*)(base + L.stat);
    w.pdq = (float*)(base + L.pdq); w.pdu = (float*)(base + L.pdu);
    w.sbf = (unsigned short*)(base + L.srow);
    w.xu = (unsigned short*)(base + L.xu); w.dqbf = (unsigned short*)(base + L.dqbf); w.dgu = (unsigned short*)(base + L.dgu);
    w.dS = (float*)(base + L.dS); w.dWfW = (float*)(base + L.dWfW); w.egran = (unsigned long long*)(base + L.egran); w.bgran = (unsigned long long*)(base + L.bgran);
}
template <class K> static int wide_lds_attr(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
