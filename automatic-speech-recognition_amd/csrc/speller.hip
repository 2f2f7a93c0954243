// speller.hip -- K4-K7: the Speller decode loop (reference las/las.py:72-160) and its gradient.
//
// The reference runs U iterations of a tf.while_loop whose body re-projects the keys
// (las/layers.py:250), runs ~20 tiny TF ops and grows its outputs by concat (las/las.py:114-115).
// Here the key projection is hoisted (one K4 GEMM by the caller), the outputs are preallocated,
// and each step is TWO launches enqueued by one C call:
//   dec_step_fwd_kernel (one workgroup per utterance; everything row-local stays in LDS):
//       finish the previous step's cell (gate nonlinearity, h/c, [vocab logits, argmax, Gumbel
//       sample]) -> query projection s.Ws -> [location conv] -> energies u.tanh(K+q+f) ->
//       -1e8 replace-mask -> softmax -> context -> cell input row [emb ; ctx ; h_prev]
//   las_gemm: the cell contraction [B, E+Hd+D] x [E+Hd+D, G*D] (weight-streaming, skinny-M MFMA tile)
// The gradient mirrors it in reverse: dec_step_bwd_kernel (attention backward of step t+1 fused with
// the gate backward of step t) + one las_gemm (dG.W^T) per step; every weight gradient is a single
// tall contraction after the loop (split-K, deterministic).  No atomics anywhere.
#include "las_common.h"
#include "lstm_cell_rows.h"
#include <type_traits>
#include <math.h>

#define LAS_MAX_NL 4

// The decoder row kernels run ONE workgroup per utterance and are bound by L2/MALL latency (each step re-reads
// that utterance's keys / encoder rows / Ws): 16 waves per CU keep enough loads in flight to cover it.
constexpr int RNT = 1024;          // threads per row workgroup
constexpr int RNG = RNT / 32;      // 32-lane half-wave groups
constexpr int RNW = RNT / 64;      // waves
constexpr int RNH = RNT / 128;     // frame groups of the context reduction

// The per-step cell product inside the persistent loop kernels (dec_loop_*_kernel): C[M,N] = A[M,K] . Bp (+ bias) once per
// step.  The grid is 8 groups (workgroup id % 8 = the XCD the hardware dispatches it to) of pn product workgroups followed
// by R = ceil(M / 8) row workgroups; group x owns the utterances b = 8 r + x.  A arrives from the group's row workgroups as
// granules of 4 bf16 {tag, k..k+1, k+2..k+3, tag}; Bp is the pre-packed MFMA fragment array, product workgroup j keeps the
// fragments of its tpw column tiles [j tpw, (j+1) tpw) in registers for the whole loop.  The outputs go back to the rows as
// granules of 2 fp32 {tag, v[col], v[col+1], tag}; C (optional) additionally receives every column as plain fp32 for
// consumers after the loop.
struct LoopProd {
    const u16x8_t* Bp; int KS, K, N, nct, M, pn, R;
    const float* bias;
    float* C; long long c_step; int ldc;
    unsigned long long* gA; int gA_row;          // [M][gA_row] granules, gA_row = K / 4
    unsigned long long* gC; int gC_row;          // [M][gC_row] granules, gC_row = N / 2
    unsigned long long* xcc;                     // [8 (pn + R)] placement handshake slots (zeroed per launch)
    int nap;                                     // product workgroups sleep nap x 2048 clocks after a step before they poll again
    int budget;                                  // polls before a wave gives up (2^21 ~ seconds; LAS_SPELLER_SPIN_LOG2 for tests)
    int* status;                                 // optional device word: LAS_SPELLER_STATUS_TIMEOUT when a partner workgroup was not seen within the poll bound
};
// A poll that runs out of budget (a partner workgroup that was never dispatched: the grid needs every workgroup co-resident)
// reports through the status word and ENDS its wave (s_endpgm: like the trap it replaces it is a no-return path, so the
// register allocation of the loops -- which sit at their 128-VGPR budget -- is unchanged; terminated waves drop out of the
// workgroup's barriers).  The waves that are left run their remaining steps on whatever data they have, partners that wait for
// the ended waves' granules end the same way: the launch drains in bounded time and the host raises at its next status check.
#define LOOP_POLL_TIMEOUT(lp)                                                                                   \
    do { if ((lp).status) (lp).status[0] = LAS_SPELLER_STATUS_TIMEOUT; __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_endpgm(); } while (0)

struct DecDev {
    int B, Tp, Hd, A, D, NL, E, V, U, mode, Kc, C, step_logits, flags;
    int row_group;         // > 0: rows u rg .. u rg + rg - 1 share enc / keys (beam search: one utterance's hypotheses) -> XCD-local row workgroups
    int shared_ops;        // 1: enc / keys (and their bf16 copies) hold ONE block per group of row_group rows -- [B / row_group, Tp, .] -- not one per row
    LoopProd lp;
    float fb;
    unsigned long long seed;
    const float *enc, *keys; const int* enc_len;
    const float *Ws, *u, *emb, *Wv, *bv, *loc_w, *loc_b, *Wf;
    int *tok_in, *tok_out;
    const float* align0;
    const float* emb_mask;
    const float* emb_noise;   // [U,V,E] variational noise added to the embedding matrix at the look-up of step t (las/las.py:164-166), or null
    float *logits, *alphas, *hs, *cs, *gates, *xin0;
    unsigned* actS;        // optional (speed mode, prefetching row kernels): 32-byte header + [U,B,Tp,A] fp16 tanh(keys + q [+ f . Wf]) of every
                           // step, written by the forward rows and read by the gradient rows instead of recomputing it (las_speller_fwd_args.act_save)
    unsigned short* xbf;   // [B,I0D]  bf16 copy of the current step's cell input row (A operand of the skinny product)
    unsigned short* dgbf;  // [B,G*D]  bf16 copy of the current step's layer-0 gate gradient
    const unsigned short *Wsbf, *keysbf, *encbf;   // bf16 copies of Ws [S,A], keys [B,Tp,A], enc [B,Tp,Hd] (speed mode)
    const unsigned short *Wsbf2, *encbf2;          // row-pair interleaved copies: [S/2][A][2], [B][Tp/2][Hd][2]
    float* dE;             // [U,B,Tp] d energy of every step (keys gradient is contracted after the loop)
    // backward
    const float* dHl;      // [U,B,D]   dlogits . Wv^T
    float *dH, *dC;        // [NL,B,D]
    float* dXin0;          // [U,B,I0D]
    float *Q, *dQ;         // [U,B,A]
    float* duRows;         // [B,A]
    float* dAext;          // [B,Tp]   grad wrt alpha_t arriving from step t+1's location conv
    float* dVbuf;          // [B,Tp,A] location-aware attention: the step's d(pre-tanh) rows (dWf is contracted from them)
    float *fcSave, *dfcSave;   // [U,B,Tp,C] loop kernels, location-aware attention: conv output f of every step and its gradient
                               // (the keys / Wf / filter gradients are contracted over the steps after the loop)
    float* dKeys;          // [B,Tp,A]
    float *dlocwRows, *dlocbRows, *dWfRows;   // [B,Kc*C], [B,C], [B,C*A]
    const float* rec[LAS_MAX_NL]; int recLd[LAS_MAX_NL]; int recOff[LAS_MAX_NL];
};

// contraction operand in the arithmetic of the mode: speed mode rounds to bf16 (RNE) exactly like the MFMA paths do,
// so that the in-loop vocabulary projection (sampling / greedy steps) equals the after-loop GEMM of the same logits
template <bool FAST> __device__ __forceinline__ float opnd(float x) { return FAST ? bf2f(f2bf(x)) : x; }

__device__ __forceinline__ float gumbel_noise(unsigned long long seed, int t, int b, int v) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(((long long)t * 1000003 + b) * 65537 + v + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    const float u01 = ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    return -logf(-logf(u01));
}

// argmax with first-index tie-break across the block; all threads receive the index
__device__ __forceinline__ int block_argmax(float v, int i, float* redv, int* redi) {
    // the 64-lane butterfly 32, 16, 8, 4, 2, 1 without the LDS: lane swaps for the cross-row steps, row rotations for the rest (the combine
    // is symmetric -- both partners keep the same (value, index) -- so a rotation by d / 2 brings what the xor partner at distance d / 2 holds)
    auto take = [&](const float ov, const int oi) { if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } };
    { const float ov = __uint_as_float(xor32_get(__float_as_uint(v))); const int oi = (int)xor32_get((unsigned)i); take(ov, oi); }
    { const float ov = __uint_as_float(xor16_get(__float_as_uint(v))); const int oi = (int)xor16_get((unsigned)i); take(ov, oi); }
    { const float ov = dpp_f<0x128>(v); const int oi = (int)dpp_u<0x128>((unsigned)i); take(ov, oi); }
    { const float ov = dpp_f<0x124>(v); const int oi = (int)dpp_u<0x124>((unsigned)i); take(ov, oi); }
    { const float ov = dpp_f<0x122>(v); const int oi = (int)dpp_u<0x122>((unsigned)i); take(ov, oi); }
    { const float ov = dpp_f<0x121>(v); const int oi = (int)dpp_u<0x121>((unsigned)i); take(ov, oi); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = i; }
    __syncthreads();
    float bv = redv[0]; int bi = redi[0];
#pragma unroll
    for (int w = 1; w < RNW; ++w)
        if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
    return bi;
}

// In-loop vocabulary projection of one row (greedy / sampled / beam steps; the after-loop GEMM serves teacher-forced training):
// logits[v] = bv[v] + sum_d h[d] Wv[d, v] in the arithmetic of the mode, the greedy arg-max and the Gumbel-max sample.
// Small vocabularies (char units: V = 30) left 994 of the 1024 threads idle behind 30 threads x D serial multiply-adds (~10 us per
// decode step, every step of a beam search): now the largest power of two TPV <= min(RNT / V, 64) lanes share one logit (each sums
// D / TPV terms, then a TPV-lane butterfly).  Large vocabularies (subword units, V = 5000) keep one thread per logit.
// step_logits = 1: the logits of step t-1 are computed in the loop at every step (greedy inference; a search step).  2 (round 6, training with
// scheduled sampling): only where the token ENTERING step t is resolved on the device (tokens_in < 0) -- 9.7 us on the dependent chain of a step
// that a teacher-forced step does not need; the logits of ALL steps (the loss's) then come from the batched product behind the loop, as without
// sampling, and tokens_out holds the greedy token of the sampled steps only.
__device__ __forceinline__ bool logits_here(const DecDev& a, const int t, const int tok_raw) {
    return a.step_logits == 1 || (a.step_logits == 2 && t < a.U && tok_raw < 0);
}
template <bool FAST>
__device__ __forceinline__ void row_logits(const DecDev& a, const float* __restrict__ h, const int t, const int b, const int tid,
                                           float* red, int* redi, int& greedy_tok, int& sample_tok) {
    const int V = a.V, D = a.D, B = a.B;
    float bestv = -INFINITY, bests = -INFINITY;
    int besti = 0x7fffffff, bestsi = 0x7fffffff;
    float* lrow = a.logits + ((size_t)(t - 1) * B + b) * V;
    int tpv = 1;
    while (tpv < 64 && tpv * 2 * V <= RNT) tpv *= 2;
    if (tpv > 1) {
        const int v = tid / tpv, part = tid - v * tpv, vc = v < V ? v : V - 1;
        float acc = 0.f;
        for (int d = part; d < D; d += tpv) acc = fmaf(opnd<FAST>(h[d]), opnd<FAST>(a.Wv[(size_t)d * V + vc]), acc);
        if (tpv == 16) {       // (char vocabularies at 1024 threads: a 16-lane group = a DPP row, the butterfly 8, 4, 2, 1 as row rotations -- sub16_sum)
            acc += dpp_f<0x128>(acc); acc += dpp_f<0x124>(acc); acc += dpp_f<0x122>(acc); acc += dpp_f<0x121>(acc);
        } else {
            for (int o = tpv >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        }
        if (part == 0 && v < V) {
            acc += a.bv[v];
            lrow[v] = acc;
            bestv = acc; besti = v;
            bests = acc + gumbel_noise(a.seed, t, b, v); bestsi = v;
        }
    } else {
        for (int v = tid; v < V; v += RNT) {
            float acc = a.bv[v];
            for (int d = 0; d < D; ++d) acc = fmaf(opnd<FAST>(h[d]), opnd<FAST>(a.Wv[(size_t)d * V + v]), acc);
            lrow[v] = acc;
            if (acc > bestv) { bestv = acc; besti = v; }
            const float sc = acc + gumbel_noise(a.seed, t, b, v);
            if (sc > bests) { bests = sc; bestsi = v; }
        }
    }
    greedy_tok = block_argmax(bestv, besti, red, redi);
    sample_tok = block_argmax(bests, bestsi, red, redi);
    if (tid == 0) a.tok_out[(size_t)(t - 1) * B + b] = greedy_tok;
}

__device__ __forceinline__ float sub32_sum(float v) {  // sum over a 32-lane half-wave: the xor butterfly 16, 8, 4, 2, 1 without the LDS (las_common.h)
    v = xor16_sum(v);
    v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); v += dpp_f<0x121>(v);
    return v;
}

// LDS carve shared by the forward and backward row kernels
struct RowLds {
    float *s_state, *qv, *part, *ev, *hl, *aprev, *fc, *red, *x0, *x1, *x2, *ctxp;
    float *locw, *wfl;        // location-aware attention: the conv filter [Kc, C] and Wf [C, A], staged once per kernel
    int* redi;
};
__device__ __forceinline__ RowLds carve(float* sm, const DecDev& a, bool bwd) {
    RowLds r;
    const int S = a.D * a.NL;
    float* p = sm;
    r.s_state = p; p += S;
    r.qv = p;      p += a.A;
    r.part = p;    p += RNG * a.A;
    r.ev = p;      p += a.Tp;
    r.hl = p;      p += a.D;
    r.aprev = p;   p += a.Tp;
    r.fc = p;      p += (a.mode == LAS_ATT_LOC ? a.Tp * a.C : 0);
    if (a.mode == LAS_ATT_LOC) p += (4 - ((p - sm) & 3)) & 3;              // float4 reads of wfl
    r.locw = p;    p += (a.mode == LAS_ATT_LOC ? (a.Kc * a.C + 3) / 4 * 4 : 0);
    r.wfl = p;     p += (a.mode == LAS_ATT_LOC ? a.C * a.A : 0);
    r.red = p;     p += 32;
    r.redi = reinterpret_cast<int*>(p); p += 32;
    r.ctxp = p; p += RNH * a.Hd;
    r.x0 = p; r.x1 = p; r.x2 = p;
    if (bwd) {
        r.x0 = p; p += a.Hd;                 // dctx
        r.x1 = p; p += a.Tp;                 // dalpha / de
        r.x2 = p; p += (a.mode == LAS_ATT_LOC ? a.Tp * a.C : 0);   // dfc
    }
    return r;
}
static size_t row_lds_bytes(const DecDev& a, bool bwd) {
    size_t n = (size_t)a.D * a.NL + a.A + RNG * a.A + a.Tp + a.D + a.Tp + 64 + RNH * a.Hd;
    if (a.mode == LAS_ATT_LOC) n += (size_t)a.Tp * a.C + 4 + (a.Kc * a.C + 3) / 4 * 4 + (size_t)a.C * a.A;
    if (bwd) n += a.Hd + a.Tp + (a.mode == LAS_ATT_LOC ? (size_t)a.Tp * a.C : 0);
    return n * sizeof(float) + 64;
}

// location-aware attention: the filter and Wf are read Tp x C x Kc (resp. Tp x A x C) times per step -- from LDS, not through the L1
__device__ __forceinline__ void stage_loc_weights(const RowLds& L, const DecDev& a, int tid) {
    for (int i = tid; i < a.Kc * a.C; i += RNT) L.locw[i] = a.loc_w[i];
    for (int i = tid; i < a.C * a.A; i += RNT) L.wfl[i] = a.Wf[i];
}
// f[t', c] = bias[c] + sum_k prev_align[t' + k - pad] * w[k, c]   (conv1d, SAME, cross-correlation: las/layers.py:295-296)
__device__ __forceinline__ void loc_conv_fwd(const RowLds& L, const DecDev& a, int tid) {
    const int Tp = a.Tp, C = a.C, pad = (a.Kc - 1) / 2;
    for (int i = tid; i < Tp * C; i += RNT) {
        const int tt = i / C, c = i - tt * C;
        const int k0 = pad - tt > 0 ? pad - tt : 0, k1 = a.Kc < Tp + pad - tt ? a.Kc : Tp + pad - tt;     // taps that meet a frame
        float acc = a.loc_b[c];
        for (int k = k0; k < k1; ++k) acc = fmaf(L.aprev[tt + k - pad], L.locw[k * C + c], acc);
        L.fc[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// forward row kernel
// ------------------------------------------------------------------------------------------------
template <int CELL, bool FAST, bool LOC>
__global__ __launch_bounds__(RNT) void dec_step_fwd_kernel(DecDev a, int t) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const RowLds L = carve(sm, a, false);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, NL = a.NL, E = a.E, V = a.V, U = a.U;
    const int S = D * NL, TOP = NL - 1, GD = G * D, I0D = E + Hd + D;
    int greedy_tok = 1, sample_tok = 1;

    if (t > 0) {  // ---- finish the top layer's cell of step t-1
        float* gp = a.gates + (((size_t)TOP * U + (t - 1)) * B + b) * GD;
        float* hnew = a.hs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
        for (int d = tid; d < D; d += RNT) {
            float h;
            if (CELL == LAS_CELL_LSTM) {
                const float* cprev = a.cs + (((size_t)TOP * (U + 1) + (t - 1)) * B + b) * D;
                float* cnew = a.cs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
                const float gi = sigm<FAST>(gp[d]);
                const float gj = tanhx<FAST>(gp[D + d]);
                const float gf = sigm<FAST>(gp[2 * D + d] + a.fb);
                const float go = sigm<FAST>(gp[3 * D + d]);
                const float c = cprev[d] * gf + gi * gj;
                h = tanhx<FAST>(c) * go;
                gp[d] = gi; gp[D + d] = gj; gp[2 * D + d] = gf; gp[3 * D + d] = go;
                cnew[d] = c;
            } else {
                h = tanhx<FAST>(gp[d]);
            }
            hnew[d] = h;
            L.hl[d] = h;
        }
        __syncthreads();
        if (logits_here(a, t, t < U ? a.tok_in[(size_t)t * B + b] : 0)) {  // vocab projection + argmax (+ Gumbel sample) of step t-1
            row_logits<FAST>(a, L.hl, t, b, tid, L.red, L.redi, greedy_tok, sample_tok);
        }
    }
    if (t >= U) return;

    int tok = a.tok_in[(size_t)t * B + b];
    if (tok == -1) tok = greedy_tok;
    else if (tok == -2) tok = sample_tok;
    if (tid == 0) a.tok_in[(size_t)t * B + b] = tok;

    for (int i = tid; i < S; i += RNT) {
        const int l = i / D, d = i % D;
        L.s_state[i] = (l == TOP && t > 0) ? L.hl[d] : a.hs[(((size_t)l * (U + 1) + t) * B + b) * D + d];
    }
    if (LOC) {
        for (int i = tid; i < Tp; i += RNT)
            L.aprev[i] = t > 0 ? a.alphas[((size_t)(t - 1) * B + b) * Tp + i] : (a.align0 ? a.align0[(size_t)b * Tp + i] : 0.f);
        stage_loc_weights(L, a, tid);
    }
    __syncthreads();

    {   // query projection q = s . Ws : RNG k-groups x 32 float4 lanes over A, 8 independent 16-byte loads in flight per thread
        const int a4 = tid & 31, kg = tid >> 5;
        for (int a0 = a4; a0 < A / 4; a0 += 32) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* wp = reinterpret_cast<const float4*>(a.Ws) + a0;
            int k = kg;
            for (; k + 7 * RNG < S; k += 8 * RNG) {
                float4 wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = wp[(size_t)(k + RNG * u) * (A / 4)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float sk = L.s_state[k + RNG * u];
                    acc.x = fmaf(sk, wv[u].x, acc.x); acc.y = fmaf(sk, wv[u].y, acc.y);
                    acc.z = fmaf(sk, wv[u].z, acc.z); acc.w = fmaf(sk, wv[u].w, acc.w);
                }
            }
            for (; k < S; k += RNG) {
                const float4 wv = wp[(size_t)k * (A / 4)];
                const float sk = L.s_state[k];
                acc.x = fmaf(sk, wv.x, acc.x); acc.y = fmaf(sk, wv.y, acc.y); acc.z = fmaf(sk, wv.z, acc.z); acc.w = fmaf(sk, wv.w, acc.w);
            }
            reinterpret_cast<float4*>(L.part + kg * A)[a0] = acc;
        }
    }
    if (LOC) loc_conv_fwd(L, a, tid);   // f = conv1d(prev_align)
    __syncthreads();
    for (int i = tid; i < A; i += RNT) {
        float q = 0.f;
#pragma unroll
        for (int k8 = 0; k8 < RNG; ++k8) q += L.part[k8 * A + i];
        L.qv[i] = q;
    }
    __syncthreads();

    const int len = a.enc_len[b];
    {   // energies: a 32-lane half-wave per encoder frame, float4 over the attention dim, 4 frames in flight
        const int sl = tid & 31, grp = tid >> 5;
        const bool loc = LOC;
        for (int tb = grp; tb < Tp; tb += 4 * RNG) {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
            for (int a4 = sl; a4 < A / 4; a4 += 32) {
                float4 k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tt = tb + RNG * u;
                    k4[u] = tt < Tp ? reinterpret_cast<const float4*>(a.keys + ((size_t)b * Tp + tt) * A)[a4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float4 q4 = reinterpret_cast<const float4*>(L.qv)[a4];
                const float4 u4 = reinterpret_cast<const float4*>(a.u)[a4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tt = tb + RNG * u;
                    float4 p = make_float4(k4[u].x + q4.x, k4[u].y + q4.y, k4[u].z + q4.z, k4[u].w + q4.w);
                    if (loc && tt < Tp) {
                        for (int c = 0; c < a.C; ++c) {
                            const float f = L.fc[tt * a.C + c];
                            const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                            p.x = fmaf(f, w4.x, p.x); p.y = fmaf(f, w4.y, p.y); p.z = fmaf(f, w4.z, p.z); p.w = fmaf(f, w4.w, p.w);
                        }
                    }
                    part[u] += u4.x * tanhx<FAST>(p.x) + u4.y * tanhx<FAST>(p.y) + u4.z * tanhx<FAST>(p.z) + u4.w * tanhx<FAST>(p.w);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = tb + RNG * u;
                const float e = sub32_sum(part[u]);
                if (sl == 0 && tt < Tp) L.ev[tt] = (tt < len) ? e : -1e8f;   // replace-mask, las/layers.py:205-207
            }
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < Tp; i += RNT) m = fmaxf(m, L.ev[i]);
    m = block_max<RNT>(m, L.red);
    float ssum = 0.f;
    for (int i = tid; i < Tp; i += RNT) { const float e = expf(L.ev[i] - m); L.ev[i] = e; ssum += e; }
    ssum = block_sum<RNT>(ssum, L.red);
    const float inv = 1.0f / ssum;
    float* arow = a.alphas + ((size_t)t * B + b) * Tp;
    for (int i = tid; i < Tp; i += RNT) { const float al = L.ev[i] * inv; L.ev[i] = al; arow[i] = al; }
    __syncthreads();

    float* xrow = a.xin0 + ((size_t)t * B + b) * I0D;
    const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;   // alpha is exactly 0 beyond len (exp underflow)
    {   // context = sum_t alpha[t] * enc[b,t,:]  : 128 float4 lanes over Hd x RNH frame groups, 8 loads in flight
        const int h4 = tid & 127, half = tid >> 7;
        for (int h0 = h4; h0 < Hd / 4; h0 += 128) {
            const float4* ep = reinterpret_cast<const float4*>(a.enc + (size_t)b * Tp * Hd) + h0;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int tt = half;
            for (; tt + 7 * RNH < lim; tt += 8 * RNH) {
                float4 ev4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ev4[u] = ep[(size_t)(tt + RNH * u) * (Hd / 4)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float al = L.ev[tt + RNH * u];
                    acc.x = fmaf(al, ev4[u].x, acc.x); acc.y = fmaf(al, ev4[u].y, acc.y);
                    acc.z = fmaf(al, ev4[u].z, acc.z); acc.w = fmaf(al, ev4[u].w, acc.w);
                }
            }
            for (; tt < lim; tt += RNH) {
                const float4 e4 = ep[(size_t)tt * (Hd / 4)];
                const float al = L.ev[tt];
                acc.x = fmaf(al, e4.x, acc.x); acc.y = fmaf(al, e4.y, acc.y); acc.z = fmaf(al, e4.z, acc.z); acc.w = fmaf(al, e4.w, acc.w);
            }
            reinterpret_cast<float4*>(L.ctxp + half * Hd)[h0] = acc;
        }
        __syncthreads();
        for (int hd = tid; hd < Hd; hd += RNT) {
            float cv = 0.f;
#pragma unroll
            for (int hh = 0; hh < RNH; ++hh) cv += L.ctxp[hh * Hd + hd];
            xrow[E + hd] = cv;
            if (a.xbf) a.xbf[(size_t)b * I0D + E + hd] = f2bf(cv);
        }
    }
    unsigned short* xb = a.xbf ? a.xbf + (size_t)b * I0D : nullptr;
    for (int i = tid; i < E; i += RNT) {
        const float v = (a.emb[(size_t)tok * E + i] + (a.emb_noise ? a.emb_noise[((size_t)t * V + tok) * E + i] : 0.f)) *
                        (a.emb_mask ? a.emb_mask[((size_t)t * B + b) * E + i] : 1.f);
        xrow[i] = v;
        if (xb) xb[i] = f2bf(v);
    }
    for (int i = tid; i < D; i += RNT) {
        const float v = L.s_state[i];
        xrow[E + Hd + i] = v;
        if (xb) xb[E + Hd + i] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------
// speed-mode (bf16) row kernels, additive attention.  Same steps as above, but every per-step re-read
// (Ws, keys, encoder rows) comes from bf16 copies made once per call with 16-byte loads, which halves the
// bytes each row workgroup pulls through its CU's L2 port -- the bound of these kernels.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
// sum over a 16-lane group (a DPP row), in every lane.  Row rotations by 8, 4, 2, 1 on the VALU's data-parallel path instead of four
// __shfl_xor = four ds_bpermute round trips through the LDS queue (a dependent chain of ~100-cycle operations at the end of every frame group
// of the energies): BIT-identical to the xor butterfly -- after the step with distance d every lane equals its partner at distance d
// (a + b == b + a), so the value a rotation by d / 2 brings is the value the xor partner holds.
#ifndef LAS_SUB16_DPP
#define LAS_SUB16_DPP 1
#endif
__device__ __forceinline__ float sub16_sum(float v) {
#if LAS_SUB16_DPP
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));     // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));     // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));     // row_ror:1
#else
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
#endif
    return v;
}
// the speed mode's operand copies in one launch: job y = plain bf16 copy of n elements (R == 0) or row-pair interleaved copy of
// `batch` matrices: src [batch][R][C] fp32 -> dst [batch][ceil(R/2)][C][2] bf16, rows 2r and 2r+1 interleaved per column (zero past R)
struct BfCopyJob { const float* src; unsigned short* dst; size_t n; int R, C, batch; };
struct BfCopyJobs { BfCopyJob j[5]; };
__global__ __launch_bounds__(256) void bf_copies_kernel(BfCopyJobs jobs) {
    const BfCopyJob jb = jobs.j[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (jb.R == 0) {
        for (size_t i = i0; i < jb.n; i += stride) jb.dst[i] = f2bf(jb.src[i]);
        return;
    }
    const int R2 = (jb.R + 1) / 2;
    const size_t per = (size_t)R2 * jb.C, total = per * jb.batch;
    for (size_t i = i0; i < total; i += stride) {
        const size_t bt = i / per, q = i - bt * per;
        const int r2 = (int)(q / jb.C), c = (int)(q % jb.C);
        const float* sp = jb.src + bt * (size_t)jb.R * jb.C;
        const float lo = sp[(size_t)(2 * r2) * jb.C + c], hi = (2 * r2 + 1 < jb.R) ? sp[(size_t)(2 * r2 + 1) * jb.C + c] : 0.f;
        reinterpret_cast<unsigned int*>(jb.dst)[i] = f2bf2(lo, hi);
    }
}

struct BfLds {
    float *s_state, *qv, *ev, *hl, *x0, *x1, *red, *scr; int* redi;
    // location-aware attention in the one-launch loop kernels: previous alignment, conv output f and its gradient, the gradient that
    // step t + 1's conv sends back to alpha_t, the staged filter [Kc, C] and Wf [C, A]
    float *aprev, *fc, *dfc, *daext, *wft, *wfl;
    unsigned short* wfb;      // [16][A] bf16 copy of Wf (rows >= C zero): B operand of the d f product
    unsigned short* wcf;      // [ceil(Kc/32)][2][64][8] the conv filter as MFMA B fragments, bf16 high and low parts (loc_conv_mfma)
    unsigned int* dvb;        // [Tp][A/2] the step's d(pre-tanh) rows as bf16 pairs: A operand of the d f product (gradient loop)
    uint4* encl;              // loop kernels, additive attention: the first ENC_RES slabs of the utterance's encoder rows, resident for the whole loop
};
// Encoder rows that stay in the row workgroup's LDS for the whole loop (round 4).  A row pulled its utterance's 164 KB of encoder rows
// through the CU's 64 B / clock vector-memory path at EVERY decode step (1.07 us of issue time, exposed: their destination registers are
// only free after the query projection); the rows do not change over the loop, a CU runs one loop workgroup, and the row state needs
// 41 KB of the 160 KB: ENC_RES slabs of 32 Hd bytes (16 frames; 7 x 16 KB at Hd = 512 = 112 of 160 frames) are copied in once, the
// rest is streamed per step -- few enough registers to be requested at the head of the step with the other bulk loads.
#ifndef LAS_ENC_RES_F
#define LAS_ENC_RES_F 7
#endif
#ifndef LAS_ENC_RES_B
#define LAS_ENC_RES_B 0
#endif
template <int NE, bool LOC, bool BWD = false> struct EncRes {
    static constexpr int M = BWD ? LAS_ENC_RES_B : LAS_ENC_RES_F;
    static constexpr int N = LOC ? 0 : (NE < M ? NE : M);
};
__host__ __device__ __forceinline__ size_t enc_res_bytes(const DecDev& a, bool loc, int ne, bool bwd = false) {
    const int m = bwd ? LAS_ENC_RES_B : LAS_ENC_RES_F;
    return loc ? 0 : (size_t)(ne < m ? ne : m) * 32 * a.Hd;
}
static int loop_ne(int Tp) { return Tp <= 128 ? 8 : Tp <= 160 ? 10 : Tp <= 192 ? 12 : 14; }
__device__ __forceinline__ int up4(int x) { return (x + 3) & ~3; }
// floats of the zero-padded previous-alignment array: the filter's reach on both sides, rounded up to what the MFMA conv's fragments touch
__host__ __device__ __forceinline__ int loc_apad(const DecDev& a) { return (a.Tp + 15) / 16 * 16 + (a.Kc + 31) / 32 * 32 + 16; }
// frames of the zero-padded d f array (C values per frame) and floats per channel of the flipped, zero-padded filter: what the
// fragments of the transposed conv (loc_convT_mfma: blocks of 16 source frames x 16 shifts) touch
__host__ __device__ __forceinline__ int loc_dpad(const DecDev& a) { return (a.Tp + 15) / 16 * 16 + (a.Kc + 15 + 31) / 32 * 32 + 16; }
__host__ __device__ __forceinline__ int loc_wft_ld(const DecDev& a) { return 16 + (a.Kc + 15 + 31) / 32 * 32; }
__device__ __forceinline__ BfLds carve_bf(float* sm, const DecDev& a) {
    BfLds r; float* p = sm;
    r.s_state = p; p += up4(a.D * a.NL);
    r.qv = p;      p += up4(a.A);
    r.ev = p;      p += up4(a.Tp);
    r.hl = p;      p += up4(a.D);
    r.x0 = p;      p += up4(a.Hd);      // bwd: dctx
    r.x1 = p;      p += up4(a.Tp);      // bwd: d alpha / d energy
    r.red = p;     p += 32;
    r.redi = reinterpret_cast<int*>(p); p += 32;
    r.aprev = r.fc = r.dfc = r.daext = r.wft = r.wfl = nullptr; r.wfb = nullptr; r.wcf = nullptr; r.dvb = nullptr;
    if (a.mode == LAS_ATT_LOC) {
        // the conv input (previous alignment) and the transposed conv's input (d f) are zero-padded by the filter's reach on both
        // sides and up to whole MFMA fragments (loc_apad / loc_dpad): the convs' fragment reads carry no bounds checks.  aprev / dfc point at
        // frame 0 inside their padded arrays; the pads are zeroed once per launch (loc_stage_lds) and never written again.
        const int padl = (a.Kc - 1) / 2, padr = a.Kc - 1 - padl;
        r.aprev = p + padl;            p += up4(loc_apad(a));             // (frame tiles of 16 x tap steps of 32 for loc_conv_mfma)
        r.daext = p;                   p += up4(a.Tp);
        r.fc = p;                      p += up4(a.Tp * a.C);
        r.dfc = p + padr;              p += up4(loc_dpad(a) * a.C);       // CHANNEL-major: row c = loc_dpad frames, frame 0 at r.dfc + c * loc_dpad
        r.wft = p;                     p += up4(a.C * loc_wft_ld(a));
        r.wfl = p;                     p += up4(a.C * a.A);
        r.wfb = reinterpret_cast<unsigned short*>(p); p += up4(8 * a.A);
        r.wcf = reinterpret_cast<unsigned short*>(p); p += ((a.Kc + 31) / 32) * 2 * 256;
    }
    r.scr = p;                           // RNW x max(Hd, 2A) partials
    if (a.mode == LAS_ATT_LOC) r.dvb = reinterpret_cast<unsigned int*>(p + RNW * 2 * a.A);   // behind the dq / du partials of the same phase
    r.encl = reinterpret_cast<uint4*>(p + (size_t)RNW * (a.Hd > 2 * a.A ? a.Hd : 2 * a.A));   // (loop kernels, additive attention only: behind the scratch)
    return r;
}
static size_t bf_lds_bytes(const DecDev& a) {
    auto u4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    size_t scr = (size_t)RNW * (a.Hd > 2 * a.A ? a.Hd : 2 * a.A);
    size_t loc = 0;
    if (a.mode == LAS_ATT_LOC) {
        loc = u4(loc_apad(a)) + u4(a.Tp) + u4((size_t)a.Tp * a.C) + u4((size_t)loc_dpad(a) * a.C) + u4((size_t)a.C * loc_wft_ld(a)) +
              u4((size_t)a.C * a.A) + u4((size_t)8 * a.A) + (size_t)((a.Kc + 31) / 32) * 2 * 256;
        size_t need = (size_t)RNW * 256;                             // loc_convT_mfma: one 16 x 16 partial tile per wave
        if (need > scr) scr = need;
        need = (size_t)RNW * 2 * a.A + (size_t)((a.Tp + 15) / 16 * 16) * (a.A / 2);      // dq / du partials + the step's dv rows (bf16 pairs)
        if (need > scr) scr = need;
    }
    return (u4((size_t)a.D * a.NL) + u4(a.A) + 2 * u4(a.Tp) + u4(a.D) + u4(a.Hd) + 64 + loc + scr) * sizeof(float) + 64;
}
// location-aware attention, loop kernels: the conv1d over the previous alignment (las/layers.py:295-296; SAME, cross-correlation):
// f[t', c] = bias[c] + sum_k aprev[t' + k - pad] w[k, c] and, in the gradient loop, its transpose
// d alpha_{t-1}[src] = sum_c sum_k d f[src - k + pad, c] w[k, c]; alignment, d f and the filter live in LDS.  History (r3 traces): one
// thread per output walking the 201 taps, ~10 us per conv; sliding register windows of 8 frames x 8 taps, 5.0-5.6 us (fp32 VALU issue:
// 321,600 multiply-adds per row and step); now both run on the matrix cores.
// The conv on the matrix cores: f = T(aprev) . w with T the Toeplitz matrix of the zero-padded alignment, T[t', k] = P[t' + k]
// (M = frames in tiles of 16, K = taps in steps of 32, N = channels <= 16).  Lane (g, r) of an A fragment holds 8 CONSECUTIVE
// alignments P[16 mt + r + 32 ks + 8 g ...] -- read straight from the padded fp32 array in LDS, no Toeplitz matrix is ever built.
// fp32 accuracy from bf16 MFMAs by the usual split x = hi + lo (both bf16): hi.hi + hi.lo + lo.hi, the dropped lo.lo term is 2^-16
// relative.  One wave per frame tile walks all tap steps (no cross-wave reduction, no partials, no finish pass): 21 MFMAs + 56 LDS
// reads per wave instead of 314 multiply-adds + 80 LDS reads per thread -- r3 stamps: 5.0 us -> see profiles/r3_speller_loc_phase_stamps.txt.
// Call with all threads, after a barrier behind the writes of L.aprev; the result is in L.fc after the caller's next barrier.
__device__ __forceinline__ void loc_conv_mfma(const BfLds& L, const DecDev& a, const int tid) {
    const int lane = tid & 63, mt = tid >> 6, g = lane >> 4, r = lane & 15;
    if (mt * 16 >= a.Tp) return;
    const int NKS = (a.Kc + 31) >> 5;
    const float* Pp = L.aprev - (a.Kc - 1) / 2 + mt * 16 + r + g * 8;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    for (int ks = 0; ks < NKS; ++ks) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Pp[ks * 32 + e];
        unsigned int h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short h0 = f2bf(x[2 * e]), h1 = f2bf(x[2 * e + 1]);
            h[e] = (unsigned int)h0 | ((unsigned int)h1 << 16);
            l[e] = f2bf2(x[2 * e] - bf2f(h0), x[2 * e + 1] - bf2f(h1));
        }
        const u16x8_t ah = __builtin_bit_cast(u16x8_t, (u32x4_t){h[0], h[1], h[2], h[3]});
        const u16x8_t al = __builtin_bit_cast(u16x8_t, (u32x4_t){l[0], l[1], l[2], l[3]});
        const u16x8_t bh = *reinterpret_cast<const u16x8_t*>(L.wcf + ((size_t)(ks * 2) * 64 + lane) * 8);
        const u16x8_t bl = *reinterpret_cast<const u16x8_t*>(L.wcf + ((size_t)(ks * 2 + 1) * 64 + lane) * 8);
        acc0 = mfma_bf16_16x16x32(ah, bh, acc0);
        acc1 = mfma_bf16_16x16x32(ah, bl, acc1);
        acc2 = mfma_bf16_16x16x32(al, bh, acc2);
    }
    if (r < a.C) {                                                    // C layout: lane (g, n = r) holds frames 16 mt + 4 g + i, channel n
        const float bias = a.loc_b[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = mt * 16 + g * 4 + i;
            if (t < a.Tp) L.fc[t * a.C + r] = (acc0[i] + (acc1[i] + acc2[i])) + bias;
        }
    }
}
// The transposed conv on the matrix cores.  With P = the zero-padded d f array (P[x, c], indexed so that source frame s at flipped tap
// k' = Kc - 1 - k reads P[s + k', c]) and wf_c[k'] = w[Kc - 1 - k', c]:  out[s] = sum_c sum_k' P[s + k', c] wf_c[k'].  A plain
// Toeplitz product would have ONE useful output column per channel.  Instead the 16 frames of a block are the N dimension:
// s = 16 a + n, u = n + k'  ->  out[16 a + n] = sum_c sum_u P[16 a + u, c] wf_c[u - n]:  per channel a [frame blocks, u] x [u, 16 shifts]
// product -- A fragments are 8 consecutive frames of one channel (d f is kept channel-major in LDS), B fragments 8 consecutive taps of the flipped,
// zero-padded filter row (L.wft, written once per launch) -- all 16 columns useful.  The (channel, u step) pairs are dealt to the 16
// waves; their partial tiles meet in the scratch.  Same bf16 hi/lo split as loc_conv_mfma.  Needs ceil(Tp / 16) <= 16.
// Call with all threads after a barrier behind the writes of L.dfc; ends with d alpha_{t-1} in L.daext (barrier inside; the caller's
// next barrier publishes it).
__device__ __forceinline__ void loc_convT_mfma(const BfLds& L, const DecDev& a, const int tid) {
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, r = lane & 15;
    const int C = a.C, nA = (a.Tp + 15) >> 4, NKS = (a.Kc + 15 + 31) >> 5, ld = loc_wft_ld(a);
    const int ldd = loc_dpad(a);
    const float* P = L.dfc - (a.Kc - 1 - (a.Kc - 1) / 2);                           // channel-major rows of ldd frames (16-byte aligned starts)
    const float* pa = P + 16 * (r < nA ? r : nA - 1) + g * 8;                      // rows >= nA: any valid address, their outputs are dropped
    const float* pb = L.wft + 15 + g * 8 - r;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    for (int it = wv; it < C * NKS; it += RNW) {
        const int c = it / NKS, ks = it - c * NKS;
        float x[8], y[8];
        {
            const float4 x0 = *reinterpret_cast<const float4*>(pa + c * ldd + ks * 32), x1 = *reinterpret_cast<const float4*>(pa + c * ldd + ks * 32 + 4);
            x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = pb[c * ld + ks * 32 + e];
        unsigned int xh[4], xl[4], yh[4], yl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short h0 = f2bf(x[2 * e]), h1 = f2bf(x[2 * e + 1]), k0 = f2bf(y[2 * e]), k1 = f2bf(y[2 * e + 1]);
            xh[e] = (unsigned int)h0 | ((unsigned int)h1 << 16);
            xl[e] = f2bf2(x[2 * e] - bf2f(h0), x[2 * e + 1] - bf2f(h1));
            yh[e] = (unsigned int)k0 | ((unsigned int)k1 << 16);
            yl[e] = f2bf2(y[2 * e] - bf2f(k0), y[2 * e + 1] - bf2f(k1));
        }
        const u16x8_t ah = __builtin_bit_cast(u16x8_t, (u32x4_t){xh[0], xh[1], xh[2], xh[3]});
        const u16x8_t al = __builtin_bit_cast(u16x8_t, (u32x4_t){xl[0], xl[1], xl[2], xl[3]});
        const u16x8_t bh = __builtin_bit_cast(u16x8_t, (u32x4_t){yh[0], yh[1], yh[2], yh[3]});
        const u16x8_t bl = __builtin_bit_cast(u16x8_t, (u32x4_t){yl[0], yl[1], yl[2], yl[3]});
        acc0 = mfma_bf16_16x16x32(ah, bh, acc0);
        acc1 = mfma_bf16_16x16x32(ah, bl, acc1);
        acc2 = mfma_bf16_16x16x32(al, bh, acc2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) L.scr[(wv * 64 + lane) * 4 + i] = acc0[i] + (acc1[i] + acc2[i]);
    lds_barrier();
    if (tid < a.Tp) {                                   // C layout: (frame block a, shift n) sits in lane (a / 4) * 16 + n, register a % 4
        const int ab = tid >> 4, n = tid & 15;
        const float* sp = L.scr + (((ab >> 2) * 16 + n) * 4 + (ab & 3));
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < RNW; ++w) acc += sp[w * 256];
        L.daext[tid] = acc;
    }
}
__device__ __forceinline__ void loc_stage_lds(const BfLds& L, const DecDev& a, const int tid) {
    const int padl = (a.Kc - 1) / 2, padr = a.Kc - 1 - padl;
    {   // flipped filter rows for the transposed conv: wft[c][15 + x] = w[Kc - 1 - x, c] for 0 <= x < Kc, zero around
        const int ld = loc_wft_ld(a);
        for (int i = tid; i < a.C * ld; i += RNT) {
            const int c = i / ld, x = i - c * ld - 15;
            L.wft[i] = (x >= 0 && x < a.Kc) ? a.loc_w[(a.Kc - 1 - x) * a.C + c] : 0.f;
        }
    }
    for (int i = tid; i < a.C * a.A; i += RNT) L.wfl[i] = a.Wf[i];
    for (int i = tid; i < loc_apad(a); i += RNT) (L.aprev - padl)[i] = 0.f;
    for (int i = tid; i < (a.Kc + 31) / 32 * 1024; i += RNT) {          // filter as B fragments: lane l holds w[ks*32 + 8*(l>>4) + e][l&15]
        const int e = i & 7, l = (i >> 3) & 63, part = (i >> 9) & 1, ks = i >> 10;
        const int k = ks * 32 + (l >> 4) * 8 + e, n = l & 15;
        const float v = (k < a.Kc && n < a.C) ? a.loc_w[k * a.C + n] : 0.f;
        const unsigned short hi = f2bf(v);
        L.wcf[i] = part ? f2bf(v - bf2f(hi)) : hi;
    }
    for (int i = tid; i < loc_dpad(a) * a.C; i += RNT) (L.dfc - padr)[i] = 0.f;
    for (int i = tid; i < 16 * a.A; i += RNT) L.wfb[i] = (i / a.A) < a.C ? f2bf(a.Wf[i]) : (unsigned short)0;
}

template <int CELL, int NJ>
__global__ __launch_bounds__(RNT) void dec_step_fwd_bf_kernel(DecDev a, int t) {
    constexpr bool FAST = true;
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const BfLds L = carve_bf(sm, a);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, NL = a.NL, E = a.E, V = a.V, U = a.U;
    const int S = D * NL, TOP = NL - 1, GD = G * D, I0D = E + Hd + D;
    int greedy_tok = 1, sample_tok = 1;

    if (t > 0) {  // ---- finish the top layer's cell of step t-1
        float* gp = a.gates + (((size_t)TOP * U + (t - 1)) * B + b) * GD;
        float* hnew = a.hs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
        for (int d = tid; d < D; d += RNT) {
            float h;
            if (CELL == LAS_CELL_LSTM) {
                const float* cprev = a.cs + (((size_t)TOP * (U + 1) + (t - 1)) * B + b) * D;
                float* cnew = a.cs + (((size_t)TOP * (U + 1) + t) * B + b) * D;
                const float gi = sigm<FAST>(gp[d]);
                const float gj = tanhx<FAST>(gp[D + d]);
                const float gf = sigm<FAST>(gp[2 * D + d] + a.fb);
                const float go = sigm<FAST>(gp[3 * D + d]);
                const float c = cprev[d] * gf + gi * gj;
                h = tanhx<FAST>(c) * go;
                gp[d] = gi; gp[D + d] = gj; gp[2 * D + d] = gf; gp[3 * D + d] = go;
                cnew[d] = c;
            } else {
                h = tanhx<FAST>(gp[d]);
            }
            hnew[d] = h;
            L.hl[d] = h;
        }
        __syncthreads();
        if (logits_here(a, t, t < U ? a.tok_in[(size_t)t * B + b] : 0)) {  // vocab projection + argmax (+ Gumbel sample) of step t-1
            row_logits<FAST>(a, L.hl, t, b, tid, L.red, L.redi, greedy_tok, sample_tok);
        }
    }
    if (t >= U) return;

    int tok = a.tok_in[(size_t)t * B + b];
    if (tok == -1) tok = greedy_tok;
    else if (tok == -2) tok = sample_tok;
    if (tid == 0) a.tok_in[(size_t)t * B + b] = tok;

    for (int i = tid; i < S; i += RNT) {
        const int l = i / D, d = i % D;
        L.s_state[i] = (l == TOP && t > 0) ? L.hl[d] : a.hs[(((size_t)l * (U + 1) + t) * B + b) * D + d];
    }
    __syncthreads();

    const int a8 = tid & 15, grp = tid >> 4;       // 16 lanes x 16 bytes = 128 attention columns; 64 groups
    const int A8 = A >> 3;
    {   // query projection q = s . Ws : 64 k-groups, up to 8 independent 16-byte loads in flight per thread
        for (int a0 = a8; a0 < A8; a0 += 16) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            const uint4* wp = reinterpret_cast<const uint4*>(a.Wsbf) + a0;
            for (int k = grp; k < S; k += 8 * 64) {
                uint4 w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = k + 64 * u;
                    w8[u] = kk < S ? wp[(size_t)kk * A8] : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = k + 64 * u;
                    const float sk = kk < S ? bf2f(f2bf(L.s_state[kk])) : 0.f;   // both operands bf16, as in the pf kernel
                    float w[8];
                    unpack8(w8[u], w);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(sk, w[e], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {          // the 4 k-groups of this wave
                acc[e] = xor16_sum(acc[e]);
                acc[e] = xor32_sum(acc[e]);
            }
            if (lane < 16) {
                float4* o = reinterpret_cast<float4*>(L.scr + wv * A + a0 * 8);
                o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < A; i += RNT) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < RNW; ++w) q += L.scr[w * A + i];
        L.qv[i] = q;
    }
    __syncthreads();

    const int len = a.enc_len[b];
    const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;   // alpha is exactly 0 beyond len (exp underflow)
    {   // energies: a 16-lane group per encoder frame, 16-byte key loads, 4 frames in flight
        float q8[NJ][8], u8[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int a0 = a8 + 16 * j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q8[j][e] = a0 < A8 ? L.qv[a0 * 8 + e] : 0.f;
                u8[j][e] = a0 < A8 ? a.u[a0 * 8 + e] : 0.f;
            }
        }
        const uint4* kp = reinterpret_cast<const uint4*>(a.keysbf) + (size_t)b * Tp * A8;
        for (int tb = grp; tb < Tp; tb += 4 * 64) {
            uint4 k8[4][NJ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = tb + 64 * u;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int a0 = a8 + 16 * j;
                    k8[u][j] = (tt < len && a0 < A8) ? kp[(size_t)tt * A8 + a0] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = tb + 64 * u;
                float part = 0.f;
                if (tt < len) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (a8 + 16 * j < A8) {
                            float k[8];
                            unpack8(k8[u][j], k);
#pragma unroll
                            for (int e = 0; e < 8; ++e) part = fmaf(u8[j][e], tanhx<FAST>(k[e] + q8[j][e]), part);
                        }
                    }
                }
                part = sub16_sum(part);
                if (a8 == 0 && tt < Tp) L.ev[tt] = (tt < len) ? part : -1e8f;   // replace-mask, las/layers.py:205-207
            }
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < Tp; i += RNT) m = fmaxf(m, L.ev[i]);
    m = block_max<RNT>(m, L.red);
    float ssum = 0.f;
    for (int i = tid; i < Tp; i += RNT) { const float e = expf(L.ev[i] - m); L.ev[i] = e; ssum += e; }
    ssum = block_sum<RNT>(ssum, L.red);
    const float inv = 1.0f / ssum;
    float* arow = a.alphas + ((size_t)t * B + b) * Tp;
    for (int i = tid; i < Tp; i += RNT) { const float al = L.ev[i] * inv; L.ev[i] = al; arow[i] = al; }
    __syncthreads();

    float* xrow = a.xin0 + ((size_t)t * B + b) * I0D;
    unsigned short* xb = a.xbf + (size_t)b * I0D;
    {   // context = sum_t alpha[t] * enc[b,t,:] : one wave per frame (1 KiB row at Hd = 512), 8 frames in flight
        const int H8 = Hd >> 3;
        for (int h0 = lane; h0 < H8; h0 += 64) {
            const uint4* ep = reinterpret_cast<const uint4*>(a.encbf) + (size_t)b * Tp * H8 + h0;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int tt = wv; tt < lim; tt += 8 * RNW) {
                uint4 e8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t2 = tt + RNW * u;
                    e8[u] = t2 < lim ? ep[(size_t)t2 * H8] : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t2 = tt + RNW * u;
                    const float al = t2 < lim ? bf2f(f2bf(L.ev[t2])) : 0.f;      // both operands bf16, as in the pf kernel
                    float x[8];
                    unpack8(e8[u], x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(al, x[e], acc[e]);
                }
            }
            float4* o = reinterpret_cast<float4*>(L.scr + wv * Hd + h0 * 8);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        for (int hd = tid; hd < Hd; hd += RNT) {
            float cv = 0.f;
#pragma unroll
            for (int w = 0; w < RNW; ++w) cv += L.scr[w * Hd + hd];
            xrow[E + hd] = cv;
            xb[E + hd] = f2bf(cv);
        }
    }
    for (int i = tid; i < E; i += RNT) {
        const float v = (a.emb[(size_t)tok * E + i] + (a.emb_noise ? a.emb_noise[((size_t)t * V + tok) * E + i] : 0.f)) *
                        (a.emb_mask ? a.emb_mask[((size_t)t * B + b) * E + i] : 1.f);
        xrow[i] = v;
        xb[i] = f2bf(v);
    }
    for (int i = tid; i < D; i += RNT) {
        const float v = L.s_state[i];
        xrow[E + Hd + i] = v;
        xb[E + Hd + i] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------
// prefetching row kernels (speed mode, additive attention, single-layer state, S <= 512, A <= 128, Hd <= 512,
// T' <= 16*NE).  None of the bulk operands of a step (Ws, keys, encoder rows, saved gates) depends on the
// recurrent state, so every load is issued when the kernel starts and only arithmetic sits on the dependent
// chain; LDS-only barriers keep the loads in flight (a __syncthreads() would drain vmcnt at every phase).
// ------------------------------------------------------------------------------------------------
#ifndef LAS_LOC_WS_EARLY
#define LAS_LOC_WS_EARLY 1        // location-aware forward rows: 1 = the Ws fragments, 2 = also the keys, requested in FRONT of the conv (0: behind it, with the rest)
#endif
#ifdef LAS_ROW_STAMPS   // development aid (tools/micro/bench_rows.hip): phase timestamps of workgroup 0
__device__ unsigned long long g_stamps[32];
#define STAMPX(i) do { if (tid == 0 && b == 0) g_stamps[i] = wall_clock64(); } while (0)
#define STAMPQ(i) g_stamps[i] = wall_clock64()
#define STAMPL(v) g_stamps[31] = (v)
#define STAMPB(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int las_dev_row_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
__device__ unsigned long long g_wstamps[64];      // the wide path's kernels (speller_wide.h), workgroup (0, 0): tools/probe_wide_stamps.py
#define WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_wstamps[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int las_dev_wide_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(g_wstamps)); }
#else
#define STAMPX(i)
#define STAMPQ(i)
#define STAMPB(i)
#define STAMPL(v)
#define WSTAMP(i)
#endif
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// acc += a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2bf(unsigned int a, unsigned int b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// fp16 pairs (round to nearest): the saved attention activations lie in (-1, 1), where fp16 resolves 2^-11
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2h2(float a, float b) { const h16x2_t v = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float2 h22f(unsigned u) { const h16x2_t v = __builtin_bit_cast(h16x2_t, u); return make_float2((float)v.x, (float)v.y); }
#define LAS_ACT_MAGIC 0x4c415354u          // header word 0 of act_save once a forward kernel has filled it
#define LAS_ACT_HDR 8                       // header, in 32-bit words

// The row kernels run one workgroup per utterance and are bound by instruction issue on that one CU, so the
// contractions use packed-pair dot products: q = s.Ws over k-pairs (Wsbf2 [S/2][A][2]), context over frame pairs
// (encbf2 [B][T'/2][Hd][2]); the softmax statistics are computed per wave (no block reductions).
// LOOP (dec_loop_fwd_kernel): the function is one iteration of a persistent row workgroup.  The pre-activation gates of
// step t-1 are not in a.gates but arrive as granules from the product workgroups, the bf16 cell input row leaves as
// granules, the cell state is carried in a register; everything that does not depend on the gates is issued before the poll.
// four adjacent lanes (columns col .. col+3, col % 4 == 0 in the first) -> one granule of 4 bf16
__device__ __forceinline__ void put4_bf16(const __amdgpu_buffer_rsrc_t rs, const size_t row_gran, const int col, const float v, const unsigned tag,
                                          const bool local) {
    // (every caller's col is tid + a multiple of 4: the four lanes are one DPP quad; quad_perm [1,2,3,3] / [2,3,3,3] / [3,3,3,3])
    const float v1 = dpp_f<0xF9>(v), v2 = dpp_f<0xFE>(v), v3 = dpp_f<0xFF>(v);
    if (!(col & 3)) granule16_store(rs, (unsigned)((row_gran + (col >> 2)) * 16), tag, f2bf2(v, v1), f2bf2(v2, v3), local);
}
// Where the row kernels issue the bulk loads that are consumed two phases later (measured with tools/micro/bench_fused.hip,
// B = 48, T' = 160, profiles/r3_speller_phase_stamps.txt):
//   LAS_E8_LATE (forward, the context's encoder rows, 164 KB per row and step): inside the energies phase instead of in front of the
//       q reduction -- the q-reduction phase shrinks 1.32 -> 0.56 us, the energies phase grows 2.28 -> 3.12 us: 12.48 vs 12.51 us
//       per step, no gain (a wave that waits to issue vector memory is not covered by the other waves' transcendentals) -> off;
//   LAS_W8_LATE (backward, the Ws rows of the state gradient, 128 KB): one load per frame inside the energies-gradient loop instead
//       of all at once in front of a barrier: 10.75 -> 10.50 us per step -> on.
#ifndef LAS_E8_LATE
#define LAS_E8_LATE 0
#endif
#ifndef LAS_E8_BEHIND_LOOP
#define LAS_E8_BEHIND_LOOP 1      // the loop kernels' streamed encoder slabs are requested behind the query reduction too (bench_fused: 10.2-10.4 -> 10.0-10.1 us per forward step)
#endif
#ifndef LAS_E8_EARLY
#define LAS_E8_EARLY 0      // (the streamed slabs requested at the head of the step instead of behind the query projection: 10.2 vs 10.0 us, 6 spilled VGPRs)
#endif
#ifndef LAS_W8_LATE
#define LAS_W8_LATE 1
#endif
#ifndef LAS_ABL_SP
#define LAS_ABL_SP 0   // development: bit mask of parts of the forward row to leave out (timing experiments only; make abl_sp ABL=<mask>)
#endif
// the operand block (keys / encoder rows) of row b: its own, or -- LAS_SPELLER_SHARED_OPERANDS -- its group's (the utterance's)
__device__ __forceinline__ int op_row(const DecDev& a, const int b) { return a.shared_ops ? b / a.row_group : b; }

template <int CELL, int NE, bool LOOP, bool LOC = false>
__device__ __forceinline__ void pf_fwd_row(const DecDev& a, const int t, const int b, const int tid, float* sm, float& ccar, const bool local) {
    static_assert(LOOP || !LOC, "location-aware attention is served by the loop kernels only (the per-step path is dec_step_fwd_kernel<.,.,true>)");
    constexpr bool FAST = true;
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    constexpr int NK = (16 * NE + 63) / 64;            // frames per 16-lane group (64 groups): T' <= 16*NE
    STAMPX(0);
    const BfLds L = carve_bf(sm, a);
    unsigned int* sp = reinterpret_cast<unsigned int*>(L.hl);      // packed state pairs  [S/2]
    unsigned int* ap = reinterpret_cast<unsigned int*>(L.x1);      // packed alpha pairs  [T'/2]
    const int lane = tid & 63, wv = tid >> 6;
    const int a8 = tid & 15, grp = tid >> 4;           // energies: 16 lanes x 8 columns per frame
    const int a4 = tid & 31, kg = tid >> 5;            // query:    32 lanes x 4 columns per k-pair
    const int h4 = tid & 127, fg = tid >> 7;           // context:  128 lanes x 4 columns per frame pair
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, E = a.E, V = a.V, U = a.U;
    const int S = D, GD = G * D, I0D = E + Hd + D, A8 = A >> 3, A4 = A >> 2, H4 = Hd >> 2, S2 = (S + 1) >> 1, Tp2 = (Tp + 1) >> 1;

    uint4 w8[8];
    if (LOC && LAS_LOC_WS_EARLY) {
        // (experiment) the query projection's Ws fragments in front of the conv: they are the first bulk operand the step consumes
        const int a4c_ = a4 < A4 ? a4 : A4 - 1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kp = kg + 32 * u, kpc = kp < S2 ? kp : S2 - 1;
            w8[u] = reinterpret_cast<const uint4*>(a.Wsbf2)[(size_t)kpc * A4 + a4c_];
        }
    }
    uint4 k8[NK];
    if (LOC && LAS_LOC_WS_EARLY > 1) {
        const int a8c_ = a8 < A8 ? a8 : A8 - 1;
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
            k8[u] = reinterpret_cast<const uint4*>(a.keysbf)[((size_t)(LOOP ? b : op_row(a, b)) * Tp + ttc) * A8 + a8c_];
        }
    }
    if (LOC) {
        // f = conv1d(alpha_{t-1}): nothing in it depends on the gates of step t-1, so it runs while they are on their way -- and
        // BEFORE the step's bulk loads are issued: behind them their 60 destination registers are live and the conv's 40 spill.  The
        // alignment of the previous step is in LDS already (written by this row's softmax; the loop's barrier orders it); step 0
        // starts from align0 / zeros.
        if (t == 0) {
            if (tid < Tp) L.aprev[tid] = a.align0 ? a.align0[(size_t)b * Tp + tid] : 0.f;
            lds_barrier();
        }
        if (t < U) {
            STAMPX(25);
            loc_conv_mfma(L, a, tid);
            STAMPX(26);
            lds_barrier();
            STAMPX(27);
            if (a.actS && a.fcSave) {          // kept for the gradient loop and the after-loop filter / keys gradients (else: recomputed there)
                float* fs = a.fcSave + ((size_t)t * B + b) * Tp * a.C;
                for (int i = tid; i < Tp * a.C; i += RNT) fs[i] = L.fc[i];
            }
        }
    }
    // ---- every load of the step whose address does not depend on the recurrence, in consumption order.
    // All of them are UNCONDITIONAL with clamped addresses: a predicated load becomes an exec-masked branch whose
    // join makes the compiler drain vmcnt, which serialises the prefetch.  Out-of-range lanes are neutralised where
    // the value is used (zero multiplier / guarded store), never by a select on the loaded register.
    const int tm1 = t > 0 ? t - 1 : 0, tc = t < U ? t : U - 1, dd = tid < D ? tid : D - 1;
    float gr[4] = {0.f, 0.f, 0.f, 0.f}, cpv = 0.f;
    float* gp = a.gates + (((size_t)0 * U + tm1) * B + b) * GD;
    if (!LOOP) {
        gr[0] = gp[dd];
        if (CELL == LAS_CELL_LSTM) { gr[1] = gp[D + dd]; gr[2] = gp[2 * D + dd]; gr[3] = gp[3 * D + dd]; }
    }
    if (CELL == LAS_CELL_LSTM) {
        if (!LOOP) cpv = a.cs[(((size_t)0 * (U + 1) + tm1) * B + b) * D + dd];
        else { if (t == 0) ccar = a.cs[(size_t)b * D + dd]; cpv = ccar; }
    }
    const float s0 = a.hs[(size_t)b * D + dd];                       // initial state (used at t = 0)
    int tok = a.tok_in[(size_t)tc * B + b];
    const int len = a.enc_len[b];
    const int bo = LOOP ? b : op_row(a, b);          // (a search's hypothesis rows may share their utterance's operand block)
    const int a4c = a4 < A4 ? a4 : A4 - 1, a8c = a8 < A8 ? a8 : A8 - 1, h4c = h4 < H4 ? h4 : H4 - 1;
    if (!(LOC && LAS_LOC_WS_EARLY)) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kp = kg + 32 * u, kpc = kp < S2 ? kp : S2 - 1;
            w8[u] = reinterpret_cast<const uint4*>(a.Wsbf2)[(LAS_ABL_SP & 1) ? (size_t)(tid & 63) : (size_t)kpc * A4 + a4c];
        }
    }
    const float4 u40 = reinterpret_cast<const float4*>(a.u)[a8c * 2], u41 = reinterpret_cast<const float4*>(a.u)[a8c * 2 + 1];
    if (!(LOC && LAS_LOC_WS_EARLY > 1)) {
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
            k8[u] = reinterpret_cast<const uint4*>(a.keysbf)[(LAS_ABL_SP & 1024) ? (size_t)(tid & 63) : ((size_t)bo * Tp + ttc) * A8 + a8c];
        }
    }
    constexpr int NEL = LOOP ? EncRes<NE, LOC>::N : 0;      // encoder slabs resident in LDS (loop kernels): see EncRes
    constexpr int NER = NE - NEL > 0 ? NE - NEL : 1;
    uint4 e8[NER];
    auto e8_load = [&](const int u) __attribute__((always_inline)) {
        const int tp = fg + 8 * u;
        const int tpc = tp < Tp2 ? tp : Tp2 - 1;
        e8[u - NEL] = reinterpret_cast<const uint4*>(a.encbf2)[(LAS_ABL_SP & 2048) ? (size_t)(tid & 63) : ((size_t)bo * Tp2 + tpc) * H4 + h4c];
    };
    if (NEL > 0 && LAS_E8_EARLY) {                            // the streamed slabs ride with the other bulk loads
#pragma unroll
        for (int u = NEL; u < NE; ++u) e8_load(u);
    }
    STAMPX(1);
    if (LOOP && t > 0 && wv * 64 < D) {   // gates of step t-1 from the product workgroups: the data is the flag
        const __amdgpu_buffer_rsrc_t rs = granule_rsrc(a.lp.gC);
        u32x4_t gq[G];
        unsigned goff[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            goff[i] = (unsigned)(((size_t)b * (GD >> 1) + ((i * D + dd) >> 1)) * 16);
            gq[i] = granule16_load(rs, goff[i]);
        }
        int budget = a.lp.budget;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < G; ++i) ok &= gq[i].x == (unsigned)t && gq[i].w == (unsigned)t;
            if (ok) break;
            if (--budget == 0) LOOP_POLL_TIMEOUT(a.lp);    // a product workgroup never ran: report, never hang
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (gq[i].x != (unsigned)t || gq[i].w != (unsigned)t) gq[i] = granule16_load(rs, goff[i]);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) gr[i] = __uint_as_float((dd & 1) ? gq[i].z : gq[i].y);
    }
    STAMPX(9);
    int greedy_tok = 1, sample_tok = 1;
    {   // ---- finish the cell of step t-1 (or pick up the initial state)
        // computed by every lane (clamped operands) so that the gate loads stay at the head of the load stream;
        // only the stores are predicated
        float h, cnew = 0.f, gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f;
        if (CELL == LAS_CELL_LSTM) {
            gi = sigm<FAST>(gr[0]); gj = tanhx<FAST>(gr[1]);
            gf = sigm<FAST>(gr[2] + a.fb); go = sigm<FAST>(gr[3]);
            cnew = cpv * gf + gi * gj;
            h = tanhx<FAST>(cnew) * go;
            if (LOOP && t > 0) ccar = cnew;
        } else {
            h = tanhx<FAST>(gr[0]);
        }
        if (t == 0) h = s0;
        if (tid >= D) h = 0.f;
        if (t > 0 && tid < D && !(LAS_ABL_SP & 8)) {
            if (CELL == LAS_CELL_LSTM) {
                gp[tid] = gi; gp[D + tid] = gj; gp[2 * D + tid] = gf; gp[3 * D + tid] = go;
                a.cs[(((size_t)0 * (U + 1) + t) * B + b) * D + tid] = cnew;
            }
            a.hs[(((size_t)0 * (U + 1) + t) * B + b) * D + tid] = h;
        }
        const float hn = lane_xor1(h);
        if (tid < D) {
            L.s_state[tid] = h;
            if (!(tid & 1)) sp[tid >> 1] = f2bf2(h, (tid + 1 < D) ? hn : 0.f);
        }
        lds_barrier();
        if (t > 0 && logits_here(a, t, tok)) {  // vocab projection + argmax (+ Gumbel sample) of step t-1
            row_logits<FAST>(a, L.s_state, t, b, tid, L.red, L.redi, greedy_tok, sample_tok);
        }
    }
    if (t >= U) return;
    if (a.actS && t == 0 && b == 0 && tid == 0) a.actS[0] = LAS_ACT_MAGIC;     // (the gradient rows check it: a forward that ran other kernels clears it)

    STAMPX(2);
    if (tok < 0) {
        tok = tok == -1 ? greedy_tok : sample_tok;
        if (tid == 0) a.tok_in[(size_t)t * B + b] = tok;
    }
    // consumed at the very end
    const int ec = tid < E ? tid : E - 1;
    const float embv = a.emb[(size_t)tok * E + ec] + (a.emb_noise ? a.emb_noise[((size_t)t * V + tok) * E + ec] : 0.f);
    const float maskv = a.emb_mask ? a.emb_mask[((size_t)t * B + b) * E + ec] : 1.f;

    {   // query projection q = s . Ws : 4 columns x 2 state rows per dot2, 8 prefetched fragments per thread
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kp = kg + 32 * u;
            const unsigned int s2 = kp < S2 ? sp[kp] : 0u;
            acc[0] = dot2bf(w8[u].x, s2, acc[0]); acc[1] = dot2bf(w8[u].y, s2, acc[1]);
            acc[2] = dot2bf(w8[u].z, s2, acc[2]); acc[3] = dot2bf(w8[u].w, s2, acc[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = xor32_sum(acc[e]);
        if (lane < 32 && a4 < A4) reinterpret_cast<float4*>(L.scr + wv * A)[a4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    lds_barrier();
    STAMPX(3);
    // encoder rows for the context (the Ws registers are free now), consumed after the softmax.  164 KB per row and step: at the
    // L1's 64 B / clock that is 1.07 us of vector-memory issue.  Issued here -- by all 16 waves at once, in front of the q
    // reduction that only waves 0-1 execute -- those two waves sat in the issue queue for ~1 us while the other 14 waited at the
    // barrier (phase stamps: 1.32 us for a 16-term sum).  LAS_E8_LATE: the loads are issued in NK portions between the frames of
    // the energies phase instead, where the other waves of a SIMD have transcendental work to cover a wave that waits to issue.
    const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;   // alpha is exactly 0 beyond len (exp underflow)
    // The per-step kernel (!LOOP: a beam search's 256 rows on 256 CUs at once, not a training batch's 48) requests them BEHIND the
    // reduction: with every CU of the chip asking for its 164 KB at the same moment the reducing waves waited 4.6 us behind the requests
    // (r5 stamps, tools/probe_pf_stamps.py); the rows then land under the energies.
    constexpr bool E8_BEHIND = ((!LOOP && NEL == 0) || (LOOP && LAS_E8_BEHIND_LOOP)) && !LAS_E8_LATE;
    if (!E8_BEHIND && !LAS_E8_LATE && (NEL == 0 || !LAS_E8_EARLY)) {
#pragma unroll
        for (int u = NEL; u < NE; ++u) e8_load(u);
    }
    for (int i = tid; i < A; i += RNT) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < RNW; ++w) q += L.scr[w * A + i];
        L.qv[i] = q;
    }
    lds_barrier();
    if (E8_BEHIND) {
#pragma unroll
        for (int u = NEL; u < NE; ++u) e8_load(u);
    }
    STAMPX(4);
    {   // energies from the prefetched keys
        const float um = a8 < A8 ? 1.f : 0.f;        // lanes past the attention width carry clamped operands
        const float u8[8] = {u40.x * um, u40.y * um, u40.z * um, u40.w * um, u41.x * um, u41.y * um, u41.z * um, u41.w * um};
        float q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q8[e] = a8 < A8 ? L.qv[a8 * 8 + e] : 0.f;
        float kf[LOC ? NK : 1][8];
        if (LOC) {   // pre-activation = keys + q + f . Wf: channel-outer, so that a channel's Wf columns are read from LDS once
#pragma unroll
            for (int u = 0; u < NK; ++u) {
                unpack8(k8[u], kf[LOC ? u : 0]);
#pragma unroll
                for (int e = 0; e < 8; ++e) kf[LOC ? u : 0][e] += q8[e];
            }
            for (int c = 0; c < a.C; ++c) {
                const float4 w0 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a8c * 2], w1 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a8c * 2 + 1];
                const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int u = 0; u < NK; ++u) {
                    const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
                    const float f = L.fc[ttc * a.C + c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) kf[LOC ? u : 0][e] = fmaf(f, wf[e], kf[LOC ? u : 0][e]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            if (LAS_E8_LATE && NEL == 0) {
#pragma unroll
                for (int v = u * NE / NK; v < (u + 1) * NE / NK; ++v) e8_load(v);
                __builtin_amdgcn_sched_barrier(0);       // keep the portion in front of THIS frame's arithmetic
            }
            const int tt = grp + 64 * u;
            float part = 0.f;
            if (tt < len && tt < Tp) {
                float th[8];
                if (LOC) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) th[e] = tanhx<FAST>(kf[LOC ? u : 0][e]);
                } else {
                    float k[8];
                    unpack8(k8[u], k);
#pragma unroll
                    for (int e = 0; e < 8; ++e) th[e] = (LAS_ABL_SP & 2) ? (k[e] + q8[e]) * 0.1f : tanhx<FAST>(k[e] + q8[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf(u8[e], th[e], part);
                if (a.actS && a8 < A8)      // kept for the gradient rows (fp16: |error| <= 2^-12), which would otherwise recompute all of them
                    reinterpret_cast<uint4*>(a.actS + LAS_ACT_HDR)[(((size_t)t * B + b) * Tp + tt) * A8 + a8] =
                        make_uint4(f2h2(th[0], th[1]), f2h2(th[2], th[3]), f2h2(th[4], th[5]), f2h2(th[6], th[7]));
            }
            part = sub16_sum(part);
            if (a8 == 0 && tt < Tp) L.ev[tt] = (tt < len) ? part : -1e8f;   // replace-mask, las/layers.py:205-207
        }
    }
    lds_barrier();
    STAMPX(5);
    if (wv < 2) {   // softmax statistics per wave (NK frames per lane); waves 0-1 own the alpha pairs (T'/2 <= 128)
        float ev_[NK];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NK; ++j) { ev_[j] = lane + 64 * j < Tp ? L.ev[lane + 64 * j] : -INFINITY; m = fmaxf(m, ev_[j]); }
        m = wave_max(m);
        float ssum = 0.f;
#pragma unroll
        for (int j = 0; j < NK; ++j) ssum += expf(ev_[j] - m);
        ssum = wave_sum(ssum);
        const float inv = 1.0f / ssum;
        if (tid < Tp2) {
            const int i0 = 2 * tid, i1 = 2 * tid + 1;
            const float al0 = expf(L.ev[i0] - m) * inv, al1 = i1 < Tp ? expf(L.ev[i1] - m) * inv : 0.f;
            ap[tid] = f2bf2(al0, al1);
            if (LOC) { L.aprev[i0] = al0; if (i1 < Tp) L.aprev[i1] = al1; }      // next step's conv input (fp32)
            float* arow = a.alphas + ((size_t)t * B + b) * Tp;
            if (!(LAS_ABL_SP & 8)) {
            arow[i0] = al0;
            if (i1 < Tp) arow[i1] = al1;
            }
        }
    }
    lds_barrier();

    STAMPX(6);
    float* xrow = a.xin0 + ((size_t)t * B + b) * I0D;
    unsigned short* xb = a.xbf + (size_t)b * I0D;
    const __amdgpu_buffer_rsrc_t xrs = granule_rsrc(a.lp.gA);          // LOOP: the row leaves as granules (tag t + 1)
    const size_t xg0 = (size_t)b * a.lp.gA_row;
    {   // context = sum_t alpha[t] * enc[b,t,:] : 4 columns x 2 frames per dot2
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int tp = fg + 8 * u;
            const unsigned int al2 = 2 * tp < lim ? ap[tp] : 0u;
            const uint4 ev = u < NEL ? L.encl[(size_t)tp * H4 + h4c] : e8[u < NEL ? 0 : u - NEL];
            acc[0] = dot2bf(ev.x, al2, acc[0]); acc[1] = dot2bf(ev.y, al2, acc[1]);
            acc[2] = dot2bf(ev.z, al2, acc[2]); acc[3] = dot2bf(ev.w, al2, acc[3]);
        }
        if (h4 < H4) reinterpret_cast<float4*>(L.scr + fg * Hd)[h4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        lds_barrier();
    STAMPX(7);
        for (int hd = tid; hd < Hd; hd += RNT) {                        // Hd <= 512: one trip, whole 4-lane groups
            float cv = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) cv += L.scr[w * Hd + hd];
            if (!(LAS_ABL_SP & 8)) xrow[E + hd] = cv;
            if (LOOP) put4_bf16(xrs, xg0, E + hd, cv, (unsigned)t + 1u, local);
            else xb[E + hd] = f2bf(cv);
        }
    }
    if (tid < E) {
        const float v = embv * maskv;
        if (!(LAS_ABL_SP & 8)) xrow[tid] = v;
        if (LOOP) put4_bf16(xrs, xg0, tid, v, (unsigned)t + 1u, local);
        else xb[tid] = f2bf(v);
    }
    if (tid < D) {
        const float v = L.s_state[tid];
        if (!(LAS_ABL_SP & 8)) xrow[E + Hd + tid] = v;
        if (LOOP) put4_bf16(xrs, xg0, E + Hd + tid, v, (unsigned)t + 1u, local);
        else xb[E + Hd + tid] = f2bf(v);
    }
    STAMPX(8);
}

// Round 6: a beam search's hypothesis rows u rg .. u rg + rg - 1 read the SAME keys and encoder rows (205 KB per
// utterance at the bench geometry).  Workgroup i is dispatched to XCD i % 8, each XCD has its own L2, and with row = workgroup id an
// utterance's 16 rows landed on all 8 XCDs: every L2 fetched every utterance's operands at every step (54 MB per launch at 256 rows against
// 3.3 MB of distinct operands, r5_decode_pmc.json).  Here utterance u lives on XCD u % 8: row-workgroup j (with `off` workgroups in front
// of it in the grid) serves row (x + 8 k) rg + r with x = (j + off) % 8, k = (j / 8) / rg, r = (j / 8) % rg -- a bijection onto the rows
// when the grid holds 8 rg ceil(nutt / 8) row workgroups (the ones whose utterance does not exist leave at once).  -1: no row.
__device__ __forceinline__ int xcd_local_row(int j, int off, int rg, int nrows) {
    if (rg <= 0) return j < nrows ? j : -1;
    const int x = (j + off) & 7, slot = j >> 3, k = slot / rg, r = slot - k * rg;
    const int b = (x + 8 * k) * rg + r;
    return b < nrows ? b : -1;
}
static int xcd_local_grid(int nrows, int rg) { return rg > 0 ? 8 * rg * cdiv(nrows / rg, 8) : nrows; }

template <int CELL, int NE>
__global__ __launch_bounds__(RNT) void dec_step_fwd_pf_kernel(DecDev a, int t) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float ccar = 0.f;
    const int b = xcd_local_row(blockIdx.x, 0, a.row_group, a.B);
    if (b < 0) return;
    pf_fwd_row<CELL, NE, false>(a, t, b, threadIdx.x, sm, ccar, false);
}

// ------------------------------------------------------------------------------------------------
// Round 5, beam search at decode.py's batch (64 utterances x beam 16 = 1024 hypothesis rows): the attention rows of a search step with
// FOUR hypotheses of one utterance per workgroup (LAS_SPELLER_ROWS_SHARE4: rows 4g .. 4g+3 have identical enc / keys / enc_len -- the beam
// search tiles an utterance's encoder output over its hypotheses, las/beam_search.py:216).  One row per workgroup pulls Ws (128 KB), the
// keys (41 KB) and the encoder rows (164 KB) through its CU for every hypothesis: 333 KB x 1024 rows = 341 MB of L2 traffic per step, 47 us
// at 7 TB/s (r5 trace: the kernel is bound by exactly that).  Here the three operands are read ONCE per workgroup and applied to four
// states: a quarter of the traffic, the same arithmetic per row IN THE SAME ORDER as pf_fwd_row (query partials per k-pair group, 16-lane
// energy sums, per-wave softmax statistics, frame-pair dot products) -- results are bit-identical to one row per workgroup.
// Decode only: t = 0 of a U = 1 call with keep_state0 (the state comes from hs slot 0), tokens given, additive attention.
// ------------------------------------------------------------------------------------------------
constexpr int BR4 = 4;
static size_t beam_rows4_lds(const DecDev& a) {
    auto u4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    const size_t scr = (size_t)BR4 * ((size_t)RNW * a.A > (size_t)8 * a.Hd ? (size_t)RNW * a.A : (size_t)8 * a.Hd);
    return (BR4 * (u4(a.D) + u4((a.D + 1) / 2) + u4(a.A) + u4(a.Tp) + u4((a.Tp + 1) / 2)) + scr) * sizeof(float) + 64;
}
template <int NE>
__global__ __launch_bounds__(RNT) void dec_beam_rows4_kernel(DecDev a) {
    STAMPB(0);
    constexpr bool FAST = true;
    constexpr int R = BR4, NK = (16 * NE + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int a8 = tid & 15, grp = tid >> 4;           // energies: 16 lanes x 8 columns per frame
    const int a4 = tid & 31, kg = tid >> 5;            // query:    32 lanes x 4 columns per k-pair
    const int h4 = tid & 127, fg = tid >> 7;           // context:  128 lanes x 4 columns per frame pair
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, E = a.E, V = a.V;
    const int I0D = E + Hd + D, A8 = A >> 3, A4 = A >> 2, H4 = Hd >> 2, S2 = (D + 1) >> 1, Tp2 = (Tp + 1) >> 1;
    const int g0 = xcd_local_row(blockIdx.x, 0, a.row_group / R, B / R);       // (a group of four rows = one "row" of the mapping)
    if (g0 < 0) return;
    const int b0 = g0 * R;
    auto u4 = [](int x) { return (x + 3) & ~3; };
    float* s_state = sm;                                                          // [R][D]
    unsigned* sp = reinterpret_cast<unsigned*>(s_state + R * u4(D));              // [R][S2] packed state pairs
    float* qv = reinterpret_cast<float*>(sp + R * u4(S2));                        // [R][A]
    float* ev = qv + R * u4(A);                                                   // [R][Tp]
    unsigned* ap = reinterpret_cast<unsigned*>(ev + R * u4(Tp));                  // [R][Tp2] packed alpha pairs
    float* scr = reinterpret_cast<float*>(ap + R * u4(Tp2));                      // [R][16][A] query partials, then [R][8][Hd] context partials
    const int SD = u4(D), SS = u4(S2), SA = u4(A), ST = u4(Tp), ST2 = u4(Tp2);

    // ---- the operands every row of the workgroup shares, requested up front (clamped addresses, see pf_fwd_row)
    const int dd = tid < D ? tid : D - 1;
    const int len = a.enc_len[b0];
    const int a4c = a4 < A4 ? a4 : A4 - 1, a8c = a8 < A8 ? a8 : A8 - 1, h4c = h4 < H4 ? h4 : H4 - 1;
    // (the four states first: loads return in issue order, and the states are what the first phase waits for -- behind the 170 KB of Ws and
    //  keys they arrived 2 us later; r5 stamps)
    float s0[R];
    int tok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = b0 + r < B ? b0 + r : B - 1;
        s0[r] = a.hs[(size_t)b * D + dd];
        tok[r] = a.tok_in[b];
    }
    uint4 w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int kp = kg + 32 * u, kpc = kp < S2 ? kp : S2 - 1;
        w8[u] = reinterpret_cast<const uint4*>(a.Wsbf2)[(size_t)kpc * A4 + a4c];
    }
    const float4 u40 = reinterpret_cast<const float4*>(a.u)[a8c * 2], u41 = reinterpret_cast<const float4*>(a.u)[a8c * 2 + 1];
    uint4 k8[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) {
        const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
        k8[u] = reinterpret_cast<const uint4*>(a.keysbf)[((size_t)op_row(a, b0) * Tp + ttc) * A8 + a8c];
    }
    STAMPB(1);
#pragma unroll
    for (int r = 0; r < R; ++r) {        // the states entering the step (hs slot 0: keep_state0)
        const float h = tid < D ? s0[r] : 0.f;
        const float hn = lane_xor1(h);
        if (tid < D) {
            s_state[r * SD + tid] = h;
            if (!(tid & 1)) sp[r * SS + (tid >> 1)] = f2bf2(h, (tid + 1 < D) ? hn : 0.f);
        }
    }
    lds_barrier();
    STAMPB(2);
    {   // query projections q_r = s_r . Ws: the Ws fragments are read once for the four states
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; acc[r][2] = 0.f; acc[r][3] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kp = kg + 32 * u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned int s2 = kp < S2 ? sp[r * SS + kp] : 0u;
                acc[r][0] = dot2bf(w8[u].x, s2, acc[r][0]); acc[r][1] = dot2bf(w8[u].y, s2, acc[r][1]);
                acc[r][2] = dot2bf(w8[u].z, s2, acc[r][2]); acc[r][3] = dot2bf(w8[u].w, s2, acc[r][3]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] = xor32_sum(acc[r][e]);
            if (lane < 32 && a4 < A4) reinterpret_cast<float4*>(scr + ((size_t)r * RNW + wv) * A)[a4] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        }
    }
    lds_barrier();
    STAMPB(3);
    // the encoder rows for the context (the Ws registers are free now), consumed after the softmax -- once for the four rows
    const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;
    for (int i = tid; i < R * A; i += RNT) {
        const int r = i / A, col = i - r * A;
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < RNW; ++w) q += scr[((size_t)r * RNW + w) * A + col];
        qv[r * SA + col] = q;
    }
    lds_barrier();
    // ... requested HERE, behind the reduction that half the waves execute: in front of it the 160 KB of requests of all 16 waves sat in the
    // CU's issue queue and the reducing waves behind them (r5 stamps: 4.3 us for a 16-term sum); they land under the energies
    uint4 e8[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int tp = fg + 8 * u, tpc = tp < Tp2 ? tp : Tp2 - 1;
        e8[u] = reinterpret_cast<const uint4*>(a.encbf2)[((size_t)op_row(a, b0) * Tp2 + tpc) * H4 + h4c];
    }
    STAMPB(4);
    {   // energies: a frame's keys are unpacked once and meet the four queries
        const float um = a8 < A8 ? 1.f : 0.f;
        const float u8[8] = {u40.x * um, u40.y * um, u40.z * um, u40.w * um, u41.x * um, u41.y * um, u41.z * um, u41.w * um};
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int tt = grp + 64 * u;
            float k[8];
            unpack8(k8[u], k);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float part = 0.f;
                if (tt < len && tt < Tp) {
                    const float4 q0 = *reinterpret_cast<const float4*>(&qv[r * SA + a8c * 8]), q1 = *reinterpret_cast<const float4*>(&qv[r * SA + a8c * 8 + 4]);
                    const float q8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};      // (lanes past the attention width: u8 = 0)
#pragma unroll
                    for (int e = 0; e < 8; ++e) part = fmaf(u8[e], tanhx<FAST>(k[e] + q8[e]), part);
                }
                part = sub16_sum(part);
                if (a8 == 0 && tt < Tp) ev[r * ST + tt] = (tt < len) ? part : -1e8f;   // replace-mask, las/layers.py:205-207
            }
        }
    }
    lds_barrier();
    STAMPB(5);
    if (wv < 2 * R) {   // softmax: waves 2r, 2r + 1 own row r (as waves 0-1 own the single row of pf_fwd_row)
        const int r = wv >> 1, t2 = tid & 127;
        float ev_[NK];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NK; ++j) { ev_[j] = lane + 64 * j < Tp ? ev[r * ST + lane + 64 * j] : -INFINITY; m = fmaxf(m, ev_[j]); }
        m = wave_max(m);
        float ssum = 0.f;
#pragma unroll
        for (int j = 0; j < NK; ++j) ssum += expf(ev_[j] - m);
        ssum = wave_sum(ssum);
        const float inv = 1.0f / ssum;
        if (t2 < Tp2 && b0 + r < B) {
            const int i0 = 2 * t2, i1 = 2 * t2 + 1;
            const float al0 = expf(ev[r * ST + i0] - m) * inv, al1 = i1 < Tp ? expf(ev[r * ST + i1] - m) * inv : 0.f;
            ap[r * ST2 + t2] = f2bf2(al0, al1);
            float* arow = a.alphas + (size_t)(b0 + r) * Tp;
            arow[i0] = al0;
            if (i1 < Tp) arow[i1] = al1;
        }
    }
    lds_barrier();
    STAMPB(6);
    {   // contexts: the encoder rows in registers meet the four alignments
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int tp = fg + 8 * u;
                const unsigned int al2 = 2 * tp < lim ? ap[r * ST2 + tp] : 0u;
                acc[0] = dot2bf(e8[u].x, al2, acc[0]); acc[1] = dot2bf(e8[u].y, al2, acc[1]);
                acc[2] = dot2bf(e8[u].z, al2, acc[2]); acc[3] = dot2bf(e8[u].w, al2, acc[3]);
            }
            if (h4 < H4) reinterpret_cast<float4*>(scr + ((size_t)r * 8 + fg) * Hd)[h4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    lds_barrier();
    STAMPB(7);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (b0 + r >= B) break;
        float* xrow = a.xin0 + (size_t)(b0 + r) * I0D;
        unsigned short* xb = a.xbf + (size_t)(b0 + r) * I0D;
        for (int hd = tid; hd < Hd; hd += RNT) {
            float cv = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) cv += scr[((size_t)r * 8 + w) * Hd + hd];
            xrow[E + hd] = cv;
            xb[E + hd] = f2bf(cv);
        }
        if (tid < E) {
            const float v = (a.emb[(size_t)tok[r] * E + tid] + (a.emb_noise ? a.emb_noise[(size_t)tok[r] * E + tid] : 0.f)) *
                            (a.emb_mask ? a.emb_mask[(size_t)(b0 + r) * E + tid] : 1.f);
            xrow[tid] = v;
            xb[tid] = f2bf(v);
        }
        if (tid < D) {
            const float v = s_state[r * SD + tid];
            xrow[E + Hd + tid] = v;
            xb[E + Hd + tid] = f2bf(v);
        }
    }
    STAMPB(8);
}

// Round 5, beam search (las/beam_search.py:94-158 is ONE loop): the attention rows of a search step and ANOTHER cell step that depends on
// the step's tokens only -- the LM's first layer (las/beam_search.py:109-116) -- as one grid.  The first `nlm` workgroups run the LM cell
// (32 rows x 16 units each, 512 of the 1024 threads; the others leave at once), the rest one attention row each.  The LM's second layer
// then rides with the Speller's cell (las_speller_fwd_args.companion): a search step is three dependent launches instead of five.
constexpr int PF_LM_LDS = 32 * LC_LD * 2 + 2 * 4 * 64 * 4 * 4;      // the cell workgroups' row tile + gate exchange
template <int CELL, int NE>
__global__ __launch_bounds__(RNT) void dec_step_fwd_pf_lm_kernel(DecDev a, int t, LstmCellLaunch lm, int nlm) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int)blockIdx.x < nlm) {
        if (threadIdx.x >= 512) return;
        const int nu = lm.H >> 4;
        float (&gates)[2][4][64][4] = *reinterpret_cast<float (*)[2][4][64][4]>(sm + 32 * LC_LD / 2);
        lstm_cell_rows_body<false, false, false>(lm, reinterpret_cast<unsigned short*>(sm), gates, (int)blockIdx.x % nu, ((int)blockIdx.x / nu) * 32);
        return;
    }
    float ccar = 0.f;
    const int b = xcd_local_row((int)blockIdx.x - nlm, nlm, a.row_group, a.B);
    if (b < 0) return;
    pf_fwd_row<CELL, NE, false>(a, t, b, threadIdx.x, sm, ccar, false);
}

// ------------------------------------------------------------------------------------------------
// The whole decode loop in ONE launch (speed mode, single-layer additive-attention geometry).  Eight independent groups,
// one per XCD: workgroup id = 8 j + x is dispatched to XCD x; j < pn are "product" workgroups that keep their fragments of
// the cell weights in REGISTERS for all U steps, j >= pn are the row workgroups of the utterances b = 8 (j - pn) + x (the
// same arithmetic as the per-step row kernel).  Per step the rows publish the bf16 cell input row as granules, the group's
// products contract it (MFMA tile rows = the group's <= 16 utterances) and publish the pre-activation gates as granules:
// the data is the flag (tag = step + 1), no fences, no kernel boundaries, no per-step weight re-fetch, and -- because a
// group lives on one XCD -- every exchange is served by that XCD's L2 (workgroup-scope stores, sc1 loads; the placement is
// verified by an XCC_ID handshake at kernel start, otherwise agent-scope write-through stores: correct anywhere, slower).
// Why: with two launches per step the product was bound by per-CU INGEST (every column owner re-reads its 40 KB of weights
// and all 48 input rows each step, ~5 us), and a chip-wide exchange through memory costs ~1.5 us per hop plus the polling
// traffic of 200 workgroups.  Progress: products wait only for rows, rows only for products, the host launches the grid only
// if 8 (pn + R) workgroups fit the device's compute units at once; polls are bounded and trap (never hang).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool loop_same_xcd(unsigned long long* slots, const int x, const int members, const int tid, int* status) {
    const unsigned tag = 0x58434400u;                                            // "XCD\0"
    const unsigned mine = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;       // HW_REG_XCC_ID[3:0]
    if (tid == 0) __hip_atomic_store(slots + blockIdx.x, ((unsigned long long)tag << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int same = 1;
    if (tid < members) {
        const unsigned long long* p = slots + (size_t)tid * 8 + x;
        unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int budget = 1 << 20;
        while ((unsigned)(v >> 32) != tag) {
            if (--budget == 0) { if (status) status[0] = LAS_SPELLER_STATUS_TIMEOUT; break; }    // a member was never dispatched
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        same = (unsigned)(v >> 32) == tag && (unsigned)v == mine;
    }
    return __syncthreads_and(same) != 0;
}

template <int TPW, int KW, int RM>
__device__ __forceinline__ void loop_product(const LoopProd& p, const int U, const bool reverse, const int x, const int j, const bool local,
                                             float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int Rx = (p.M - x + 7) >> 3;                                // utterances of this group: b = 8 r + x < M
    if (Rx <= 0) return;                                              // (a batch of fewer than 8 utterances leaves groups empty)
    const int ks0 = w * KW;                                           // the 16 waves split K, KW k-steps each (16 * KW >= KS)
    u16x8_t bf[TPW][KW];
    unsigned aoff[KW];
    bool need[KW];
    {
        // tile rows >= Rx and k-steps >= KS do not exist: their loads get an offset beyond the buffer's range (the hardware returns
        // zeros without a memory access) -- 6 of 16 tile rows exist at B = 48, and the row's 36 k-steps leave 4 of the 16 waves empty:
        // 98 KB -> 28 KB through the workgroup's vector-memory path per poll round
#pragma unroll
        for (int u = 0; u < KW; ++u) {
            const int kk = min(ks0 + u, p.KS - 1);
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) bf[tl][u] = p.Bp[((size_t)min(j * TPW + tl, p.nct - 1) * p.KS + kk) * 64 + lane];
            const int k = min(kk * 32 + g * 8, p.K - 8);
            need[u] = c < Rx && ks0 + u < p.KS && (ks0 + u) * 32 + g * 8 + 8 <= p.K;
            aoff[u] = need[u] ? (unsigned)(((size_t)(c * 8 + x) * p.gA_row + (k >> 2)) * 16) : 0x80000000u;
        }
    }
    const int on_ = Rx * 16;                                          // outputs of one column tile that exist
    const int rd_tl = tid / on_, rd_o = tid - rd_tl * on_;            // the (tile, row, column) this thread reduces in the first trip
    const __amdgpu_buffer_rsrc_t ars = granule_rsrc(p.gA), crs = granule_rsrc(p.gC);
    for (int s = 0; s < U; ++s) {
        const int step = reverse ? U - 1 - s : s;
        const unsigned tag = (unsigned)step + 1u;
        if (tid == 0 && blockIdx.x == 0) { STAMPQ(20); }
        f32x4_t acc[TPW];
        {
            u32x4_t q0[KW], q1[KW];
#pragma unroll
            for (int u = 0; u < KW; ++u) { q0[u] = granule16_load(ars, aoff[u]); q1[u] = granule16_load(ars, aoff[u] + 16u); }
            int budget = p.budget;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < KW; ++u) ok &= !need[u] || (q0[u].x == tag && q0[u].w == tag && q1[u].x == tag && q1[u].w == tag);
                if (ok) break;
                if (--budget == 0) LOOP_POLL_TIMEOUT(p);
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int u = 0; u < KW; ++u) {
                    if (q0[u].x != tag || q0[u].w != tag) q0[u] = granule16_load(ars, aoff[u]);
                    if (q1[u].x != tag || q1[u].w != tag) q1[u] = granule16_load(ars, aoff[u] + 16u);
                }
            }
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) acc[tl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < KW; ++u) {
                const bool on = need[u];
                const u32x4_t av = {on ? q0[u].y : 0u, on ? q0[u].z : 0u, on ? q1[u].y : 0u, on ? q1[u].z : 0u};
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) acc[tl] = mfma_bf16_16x16x32(__builtin_bit_cast(u16x8_t, av), bf[tl][u], acc[tl]);
            }
        }
        // tile rows that do not exist are neither written nor reduced (80 -> 30 KB of partial tiles through LDS per step, one trip of
        // the reduction instead of two: 11.3 -> 10.0 us per forward step)
        if (g * 4 < Rx) {
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) *reinterpret_cast<f32x4_t*>(red + ((size_t)(w * TPW + tl) * 64 + lane) * 4) = acc[tl];
        }
        if (tid == 0 && blockIdx.x == 0) { STAMPQ(21); }
        __syncthreads();
        if (tid == 0 && blockIdx.x == 0) { STAMPQ(22); }
        for (int idx = tid; idx < TPW * on_; idx += RNT) {            // existing rows only (Rx * 16 is a multiple of 16: the pair shuffle stays inside the trip)
            int tl, o;
            if (RM == 1) { tl = idx / on_; o = idx - tl * on_; }          // (RM = 1, forward loop: two registers fewer across the step loop -- the kernel must stay at <= 120 VGPRs, see dec_loop_fwd_kernel)
            else { tl = rd_tl; o = rd_o; if (idx != tid) { tl = idx / on_; o = idx - tl * on_; } }
            const int r16 = o >> 4, c16 = o & 15;
            const int l2 = (r16 >> 2) * 16 + c16, reg = r16 & 3;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < ((LAS_ABL_SP & 32) ? 1 : RNW); ++ww) v += red[((size_t)(ww * TPW + tl) * 64 + l2) * 4 + reg];
            const int ct = j * TPW + tl, col = ct * 16 + c16, orow = r16 * 8 + x;
            const bool valid = r16 < Rx && ct < p.nct && col < p.N;
            if (!(LAS_ABL_SP & 4) && p.bias && valid) v += p.bias[col];
            const float nb = lane_xor1(v);
            if (valid) {
                if (p.C && !(LAS_ABL_SP & 64)) p.C[(long long)step * p.c_step + (long long)orow * p.ldc + col] = v;
                if (!(c16 & 1))
                    granule16_store(crs, (unsigned)(((size_t)orow * p.gC_row + (col >> 1)) * 16), tag, __float_as_uint(v), __float_as_uint(nb), local);
            }
        }
        if (tid == 0 && blockIdx.x == 0) { STAMPQ(23); }
        __syncthreads();                                              // the partial tiles are rewritten next step
        // the rows need >= 6 us for their step: stay off the L2 and the power budget for the first ~4 us of it
        for (int i = 0; i < p.nap; ++i) __builtin_amdgcn_s_sleep(32);
    }
}

// REGISTER BUDGET (round 4): a loop workgroup is 16 waves on one CU; at 121-128 VGPRs per lane it needs the CU's WHOLE register file, so that
// one small wave of anybody else on any CU would keep the grid from becoming resident.  The additive-attention kernels of T' <= 160 are kept
// at <= 120 (tests/test_cabi_and_host.py reads the compiler's report); T' > 160 and the location-aware instances use all 128.  (Found while
// chasing exchange time-outs of tools/host_time_ranks.py -- several PROCESSES on one device; the rule did not end them: they occur with every
// kernel variant tried, and in sweeps this round did not touch, on some of the pool's boxes and not on others.  Sharing the device between processes is not a
// supported way to run the one-launch kernels; a time-out is reported through the status word, nothing hangs.)
template <int CELL, int NE, bool LOC = false>
__global__ __launch_bounds__(RNT) void dec_loop_fwd_kernel(DecDev a) {
    constexpr int TPW = 5, KW = 3;                                    // 26 x 5 column tiles >= 128, 16 waves x 3 k-steps x 32 >= 1280 (host-checked)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const bool local = loop_same_xcd(a.lp.xcc, x, a.lp.pn + a.lp.R, threadIdx.x, a.lp.status);
    if (threadIdx.x == 0 && blockIdx.x == 0) { STAMPL(local); }
    if (j < a.lp.pn) { loop_product<TPW, KW, 1>(a.lp, a.U, false, x, j, local, sm); return; }
    const int b = (j - a.lp.pn) * 8 + x;
    if (b >= a.B) return;
    float ccar = 0.f;
    if (LOC) loc_stage_lds(carve_bf(sm, a), a, threadIdx.x);          // filter + Wf: once per launch (step 0's barrier publishes them)
    if (EncRes<NE, LOC>::N > 0) {                                     // resident encoder slabs (pair-interleaved copy): once per launch
        const int H4 = a.Hd >> 2, Tp2 = (a.Tp + 1) >> 1;
        uint4* dst = carve_bf(sm, a).encl;
        const uint4* src = reinterpret_cast<const uint4*>(a.encbf2) + (size_t)b * Tp2 * H4;
        for (int i = threadIdx.x; i < EncRes<NE, LOC>::N * 8 * H4; i += RNT) {
            const int tp = i / H4;
            dst[i] = src[(size_t)(tp < Tp2 ? tp : Tp2 - 1) * H4 + (i - tp * H4)];
        }
    }
    for (int t = 0; t <= a.U; ++t) {
        int bb = b, tid = threadIdx.x;
        asm volatile("" : "+s"(bb), "+v"(tid));                       // keep the row's address arithmetic inside the iteration:
        pf_fwd_row<CELL, NE, true, LOC>(a, t, bb, tid, sm, ccar, local);   // hoisted out of the loop it costs ~70 spilled VGPRs
        lds_barrier();                                                // the next step rewrites the row's LDS state
    }
}

// gate nonlinearity of a non-top layer (multi-layer Speller only)
template <int CELL, bool FAST>
__global__ __launch_bounds__(256) void dec_pointwise_fwd_kernel(DecDev a, int layer, int t) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int b = blockIdx.x, B = a.B, D = a.D, U = a.U, GD = G * D;
    float* gp = a.gates + (((size_t)layer * U + t) * B + b) * GD;
    float* hnew = a.hs + (((size_t)layer * (U + 1) + t + 1) * B + b) * D;
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
    }
}

// ------------------------------------------------------------------------------------------------
// backward row kernels
// ------------------------------------------------------------------------------------------------
// gate backward of `layer` at step t for one row: dh = dH[layer] + extra ; writes d(pre-act) over gates
template <int CELL, bool FAST>
__device__ __forceinline__ void cell_bwd_row(const DecDev& a, int layer, int t, int b, const float* extra, int extra_ld) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int B = a.B, D = a.D, U = a.U, GD = G * D;
    float* gp = a.gates + (((size_t)layer * U + t) * B + b) * GD;
    const float* dHr = a.dH + ((size_t)layer * B + b) * D;
    float* dCr = a.dC + ((size_t)layer * B + b) * D;
    const float* ex = extra + (size_t)b * extra_ld;
    unsigned short* gb = (layer == 0 && a.dgbf) ? a.dgbf + (size_t)b * GD : nullptr;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float dh = dHr[d] + ex[d];
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

template <int CELL, bool FAST>
__global__ __launch_bounds__(256) void dec_pointwise_bwd_kernel(DecDev a, int layer, int t, const float* extra, int extra_ld) {
    cell_bwd_row<CELL, FAST>(a, layer, t, blockIdx.x, extra, extra_ld);
}

// Part A: attention backward of step t_att (if >= 0); Part B: top-layer gate backward of step t_cell (if >= 0).
template <int CELL, bool FAST, bool LOC>
__global__ __launch_bounds__(RNT) void dec_step_bwd_kernel(DecDev a, int t_att, int t_cell) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const RowLds L = carve(sm, a, true);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, NL = a.NL, E = a.E, U = a.U;
    const int S = D * NL, I0D = E + Hd + D;
    constexpr bool loc = LOC;
    float* dctx = L.x0;
    float* dal = L.x1;
    float* dfc = L.x2;

    if (t_att >= 0) {
        const int t = t_att;
        const float* dxr = a.dXin0 + ((size_t)t * B + b) * I0D;
        if (Hd <= RNT && Tp <= RNT && A <= RNT) {          // one pass: the three rows' loads in flight together (three loops = three round trips)
            const float v0 = dxr[E + (tid < Hd ? tid : Hd - 1)];
            const float v1 = a.alphas[((size_t)t * B + b) * Tp + (tid < Tp ? tid : Tp - 1)];
            const float v2 = a.Q[((size_t)t * B + b) * A + (tid < A ? tid : A - 1)];
            if (tid < Hd) dctx[tid] = v0;
            if (tid < Tp) L.ev[tid] = v1;
            if (tid < A) L.qv[tid] = v2;
        } else {
        for (int i = tid; i < Hd; i += RNT) dctx[i] = dxr[E + i];
        for (int i = tid; i < Tp; i += RNT) L.ev[i] = a.alphas[((size_t)t * B + b) * Tp + i];
        for (int i = tid; i < A; i += RNT) L.qv[i] = a.Q[((size_t)t * B + b) * A + i];
        }
        if (loc) {
            for (int i = tid; i < Tp; i += RNT)
                L.aprev[i] = t > 0 ? a.alphas[((size_t)(t - 1) * B + b) * Tp + i] : (a.align0 ? a.align0[(size_t)b * Tp + i] : 0.f);
            stage_loc_weights(L, a, tid);
        }
        __syncthreads();
        if (loc) loc_conv_fwd(L, a, tid);          // recompute f = conv1d(prev_align) (dfc is written for every frame below)
        const int len = a.enc_len[b];
        const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;
        {   // dalpha[t'] = dctx . enc[b,t',:]   (+ what step t+1's location conv sent back)
            // half-wave per frame, float4 over Hd, two frames (8 x 16-byte loads) in flight per lane
            const int sl = tid & 31, grp = tid >> 5;
            for (int tb = grp; tb < Tp; tb += 2 * RNG) {
                float acc2[2] = {0.f, 0.f};
                // (round 5) all eight loads of the two frames first, from clamped addresses: inside `if (tt < lim)` every frame's four
                // loads ended in a wait at the branch join -- two memory round trips in a row per iteration instead of one
                float4 e4[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = tb + RNG * u, ttc = tt < lim ? tt : lim - 1;
                    const float4* ep = reinterpret_cast<const float4*>(a.enc + ((size_t)b * Tp + ttc) * Hd);
#pragma unroll
                    for (int n4 = 0; n4 < 4; ++n4) {
                        const int h0 = sl + 32 * n4;
                        e4[u][n4] = ep[h0 < Hd / 4 ? h0 : Hd / 4 - 1];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = tb + RNG * u;
                    if (tt < lim) {
                        const float4* ep = reinterpret_cast<const float4*>(a.enc + ((size_t)b * Tp + tt) * Hd);
#pragma unroll
                        for (int n4 = 0; n4 < 4; ++n4) {
                            const int h0 = sl + 32 * n4;
                            if (h0 < Hd / 4) {
                                const float4 d4 = reinterpret_cast<const float4*>(dctx)[h0];
                                acc2[u] += d4.x * e4[u][n4].x + d4.y * e4[u][n4].y + d4.z * e4[u][n4].z + d4.w * e4[u][n4].w;
                            }
                        }
                        for (int h0 = sl + 128; h0 < Hd / 4; h0 += 32) {       // Hd > 512
                            const float4 e = ep[h0];
                            const float4 d4 = reinterpret_cast<const float4*>(dctx)[h0];
                            acc2[u] += d4.x * e.x + d4.y * e.y + d4.z * e.z + d4.w * e.w;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = tb + RNG * u;
                    float v = sub32_sum(acc2[u]);
                    if (sl == 0 && tt < Tp) {
                        if (loc && t + 1 < U) v += a.dAext[(size_t)b * Tp + tt];
                        dal[tt] = v;
                    }
                }
            }
        }
        __syncthreads();
        float dot = 0.f;
        for (int i = tid; i < Tp; i += RNT) dot = fmaf(L.ev[i], dal[i], dot);
        dot = block_sum<RNT>(dot, L.red);
        for (int i = tid; i < Tp; i += RNT) dal[i] = L.ev[i] * (dal[i] - dot);   // d energy (0 where masked: alpha = 0)
        __syncthreads();

        // energies backward: half-wave per frame; per-lane partials of du, dq (and dWf) over its frames
        const int sl = tid & 31, grp = tid >> 5;
        float du_acc[8], dq_acc[8];    // A <= 256 -> at most 2 float4 per lane
#pragma unroll
        for (int i = 0; i < 8; ++i) { du_acc[i] = 0.f; dq_acc[i] = 0.f; }
        float4 u4s[2];                               // (loop-invariant: was re-loaded -- and waited for -- per frame and slot)
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) u4s[slot] = reinterpret_cast<const float4*>(a.u)[min(sl + 32 * slot, A / 4 - 1)];
        for (int tb = grp; tb < Tp; tb += 2 * RNG) {
          float4 kpre[2][2], dkpre[2][2];            // [frame][a4 slot]: keys and dKeys prefetched for both frames
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int tt = tb + RNG * u;
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int a4 = sl + 32 * slot;
                // (unconditional, clamped addresses, zero by select: no branch join in front of the loads that follow)
                const bool on = tt < Tp && a4 < A / 4;
                const size_t o4 = ((size_t)b * Tp + (tt < Tp ? tt : Tp - 1)) * (A / 4) + (a4 < A / 4 ? a4 : A / 4 - 1);
                const float4 kv = reinterpret_cast<const float4*>(a.keys)[o4], dkv = reinterpret_cast<const float4*>(a.dKeys)[o4];
                kpre[u][slot] = on ? kv : make_float4(0.f, 0.f, 0.f, 0.f);
                dkpre[u][slot] = on ? dkv : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int tt = tb + RNG * u;
            if (tt >= Tp) continue;
            const float de = dal[tt];
            float4* dkp = reinterpret_cast<float4*>(a.dKeys + ((size_t)b * Tp + tt) * A);
            float4 dvs[2];
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int a4 = sl + 32 * slot;
                dvs[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a4 >= A / 4) continue;
                const float4 k4 = kpre[u][slot];
                const float4 q4 = reinterpret_cast<const float4*>(L.qv)[a4];
                const float4 u4 = u4s[slot];
                float4 p = make_float4(k4.x + q4.x, k4.y + q4.y, k4.z + q4.z, k4.w + q4.w);
                if (loc) {
                    for (int c = 0; c < a.C; ++c) {
                        const float f = L.fc[tt * a.C + c];
                        const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                        p.x = fmaf(f, w4.x, p.x); p.y = fmaf(f, w4.y, p.y); p.z = fmaf(f, w4.z, p.z); p.w = fmaf(f, w4.w, p.w);
                    }
                }
                const float vx = tanhx<FAST>(p.x), vy = tanhx<FAST>(p.y), vz = tanhx<FAST>(p.z), vw = tanhx<FAST>(p.w);
                const float4 dv = make_float4(de * u4.x * (1.f - vx * vx), de * u4.y * (1.f - vy * vy),
                                              de * u4.z * (1.f - vz * vz), de * u4.w * (1.f - vw * vw));
                du_acc[slot * 4 + 0] += de * vx; du_acc[slot * 4 + 1] += de * vy;
                du_acc[slot * 4 + 2] += de * vz; du_acc[slot * 4 + 3] += de * vw;
                dq_acc[slot * 4 + 0] += dv.x; dq_acc[slot * 4 + 1] += dv.y;
                dq_acc[slot * 4 + 2] += dv.z; dq_acc[slot * 4 + 3] += dv.w;
                {
                    float4 dk = dkpre[u][slot];
                    dk.x += dv.x; dk.y += dv.y; dk.z += dv.z; dk.w += dv.w;
                    dkp[a4] = dk;
                }
                dvs[slot] = dv;
                // the filter-projection gradient dWf[c, a] = sum_t' f[t', c] dv[t', a] is contracted AFTER this loop from the
                // step's dv rows (a per-frame read-modify-write of the row-private [C, A] partials was a 50-deep dependent
                // chain of global accesses per lane and step: 285 us per decode step at the reference's K = 201, C = 10)
                if (loc) reinterpret_cast<float4*>(a.dVbuf + ((size_t)b * Tp + tt) * A)[a4] = dv;
            }
            if (loc) {   // d f[t', c] = sum_a dv[a] Wf[c, a]: one half-wave reduction per channel, nothing carried in registers
                for (int c = 0; c < a.C; ++c) {
                    float s1 = 0.f;
#pragma unroll
                    for (int slot = 0; slot < 2; ++slot) {
                        const int a4 = sl + 32 * slot;
                        if (a4 < A / 4) {
                            const float4 w4 = reinterpret_cast<const float4*>(L.wfl + (size_t)c * A)[a4];
                            s1 += dvs[slot].x * w4.x + dvs[slot].y * w4.y + dvs[slot].z * w4.z + dvs[slot].w * w4.w;
                        }
                    }
                    s1 = sub32_sum(s1);
                    if (sl == 0) dfc[tt * a.C + c] = s1;
                }
            }
          }
        }
        if (loc) {   // dWf partials: half-wave group g owns channel g % C and the frames t' = g / C (mod groups-per-channel)
            __syncthreads();                                           // dVbuf rows of this workgroup are complete (and visible)
            const int gpc = RNG / a.C;                                 // groups per channel (C <= 16 -> >= 2)
            const int c = grp % a.C, part = grp / a.C;
            if (part < gpc) {
                for (int a4 = sl; a4 < A / 4; a4 += 32) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4* dvp = reinterpret_cast<const float4*>(a.dVbuf + (size_t)b * Tp * A) + a4;
                    int tt = part;
                    for (; tt + 7 * gpc < Tp; tt += 8 * gpc) {                  // 8 independent 16-byte loads in flight
                        float4 dv8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) dv8[u] = dvp[(size_t)(tt + u * gpc) * (A / 4)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const float f = L.fc[(tt + u * gpc) * a.C + c];
                            acc.x = fmaf(f, dv8[u].x, acc.x); acc.y = fmaf(f, dv8[u].y, acc.y);
                            acc.z = fmaf(f, dv8[u].z, acc.z); acc.w = fmaf(f, dv8[u].w, acc.w);
                        }
                    }
                    for (; tt < Tp; tt += gpc) {
                        const float f = L.fc[tt * a.C + c];
                        const float4 dv = dvp[(size_t)tt * (A / 4)];
                        acc.x = fmaf(f, dv.x, acc.x); acc.y = fmaf(f, dv.y, acc.y); acc.z = fmaf(f, dv.z, acc.z); acc.w = fmaf(f, dv.w, acc.w);
                    }
                    float4* wr = reinterpret_cast<float4*>(a.dWfRows + (((size_t)b * RNG + grp) * a.C + c) * A) + a4;
                    float4 o = *wr;
                    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
                    *wr = o;
                }
            }
        }
        // reduce the 8 half-wave partials of du / dq through LDS
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int a4 = sl + 32 * slot;
            if (a4 < A / 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) L.part[grp * A + a4 * 4 + e] = dq_acc[slot * 4 + e];
            }
        }
        __syncthreads();
        for (int i = tid; i < A; i += RNT) {
            float s = 0.f;
#pragma unroll
            for (int g8 = 0; g8 < RNG; ++g8) s += L.part[g8 * A + i];
            L.qv[i] = s;                                   // qv now holds dq
            a.dQ[((size_t)t * B + b) * A + i] = s;
        }
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int a4 = sl + 32 * slot;
            if (a4 < A / 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) L.part[grp * A + a4 * 4 + e] = du_acc[slot * 4 + e];
            }
        }
        __syncthreads();
        for (int i = tid; i < A; i += RNT) {
            float s = 0.f;
#pragma unroll
            for (int g8 = 0; g8 < RNG; ++g8) s += L.part[g8 * A + i];
            a.duRows[(size_t)b * A + i] += s;
        }
        // d state = dq . Ws^T ; total gradient of the states consumed at step t
        // half-wave per state row (coalesced 512-B row segments), 8 rows in flight
        __syncthreads();
        {
            const int sl2 = tid & 31, grp2 = tid >> 5;
            for (int ib = grp2; ib < S; ib += 8 * RNG) {
                float acc8[8];
                if (A / 4 <= 32) {
                    // (round 5) one 16-byte piece of each of the eight rows per lane, all eight requested before the first is used: as
                    // `if (i < S) { for (a4 ...) load, use }` the "8 rows in flight" were eight round trips in a row (16 per step)
                    float4 w8[8];
                    const int a4c = sl2 < A / 4 ? sl2 : A / 4 - 1;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = ib + RNG * u;
                        w8[u] = reinterpret_cast<const float4*>(a.Ws + (size_t)(i < S ? i : S - 1) * A)[a4c];
                    }
                    const float4 d4 = reinterpret_cast<const float4*>(L.qv)[a4c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = ib + RNG * u;
                        const float4 w4 = w8[u];
                        acc8[u] = 0.f;
                        if (i < S && sl2 < A / 4) acc8[u] += w4.x * d4.x + w4.y * d4.y + w4.z * d4.z + w4.w * d4.w;
                    }
                } else
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = ib + RNG * u;
                    acc8[u] = 0.f;
                    if (i < S) {
                        const float4* wr = reinterpret_cast<const float4*>(a.Ws + (size_t)i * A);
                        for (int a4 = sl2; a4 < A / 4; a4 += 32) {
                            const float4 w4 = wr[a4];
                            const float4 d4 = reinterpret_cast<const float4*>(L.qv)[a4];
                            acc8[u] += w4.x * d4.x + w4.y * d4.y + w4.z * d4.z + w4.w * d4.w;
                        }
                    }
                }
                // (round 5) the recurrent gradients the eight sums are added to are requested together: one lane per half-wave loaded
                // pointer / pitch / offset of rec[l], waited, loaded the value, waited, stored -- sixteen round trips in a row
                float r8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = ib + RNG * u, ic = i < S ? i : S - 1, l = ic / D, d = ic % D;
                    r8[u] = a.rec[l][(size_t)b * a.recLd[l] + a.recOff[l] + d];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = ib + RNG * u;
                    const float v = sub32_sum(acc8[u]);
                    if (sl2 == 0 && i < S) {
                        const int l = i / D, d = i % D;
                        a.dH[((size_t)l * B + b) * D + d] = r8[u] + v;
                    }
                }
            }
        }
        if (loc) {   // conv1d backward: filter / bias partials per row, and d alpha_{t-1}
            const int pad = (a.Kc - 1) / 2, C = a.C;
            for (int i = tid; i < a.Kc * C; i += RNT) {
                const int k = i / C, c = i - k * C;
                const int t0 = pad - k > 0 ? pad - k : 0, t1 = Tp < Tp + pad - k ? Tp : Tp + pad - k;
                float acc = 0.f;
                for (int tt = t0; tt < t1; ++tt) acc = fmaf(dfc[tt * C + c], L.aprev[tt + k - pad], acc);
                a.dlocwRows[(size_t)b * a.Kc * C + i] += acc;
            }
            for (int c = tid; c < C; c += RNT) {
                float acc = 0.f;
                for (int tt = 0; tt < Tp; ++tt) acc += dfc[tt * C + c];
                a.dlocbRows[(size_t)b * C + c] += acc;
            }
            // d alpha_{t-1}[src] = sum_k sum_c dfc[src - k + pad, c] w[k, c]: the taps of a source frame are split over NKC
            // thread groups (160 threads x 2010 serial taps before), partial sums meet in LDS (the du / dq scratch is free now)
            const int NKC = RNT / Tp > 0 ? (RNT / Tp < 8 ? RNT / Tp : 8) : 1;
            __syncthreads();
            for (int i = tid; i < NKC * Tp; i += RNT) {
                const int kc = i / Tp, src = i - kc * Tp;
                const int kper = (a.Kc + NKC - 1) / NKC;
                int k0 = kc * kper, k1 = k0 + kper < a.Kc ? k0 + kper : a.Kc;
                if (k0 < src + pad - (Tp - 1)) k0 = src + pad - (Tp - 1);        // 0 <= src - k + pad < Tp
                if (k1 > src + pad + 1) k1 = src + pad + 1;
                float acc = 0.f;
                for (int k = k0; k < k1; ++k) {
                    const float* dr = dfc + (src - k + pad) * C;
                    const float* wr = L.locw + k * C;
                    for (int c = 0; c < C; ++c) acc = fmaf(dr[c], wr[c], acc);
                }
                L.part[i] = acc;
            }
            __syncthreads();
            for (int src = tid; src < Tp; src += RNT) {
                float acc = 0.f;
                for (int kc = 0; kc < NKC; ++kc) acc += L.part[kc * Tp + src];
                a.dAext[(size_t)b * Tp + src] = acc;
            }
        }
        __syncthreads();
    }
    if (t_cell >= 0) cell_bwd_row<CELL, FAST>(a, NL - 1, t_cell, b, a.dHl + (size_t)t_cell * B * D, D);
}

// speed-mode gradient row kernel (additive attention).  Part A: attention backward of step t_att; Part B: top-layer
// gate backward of step t_cell.  The keys gradient is NOT accumulated here: the step's d energy is stored and
// dkeys_kernel contracts over the steps after the loop (no per-step read-modify-write of [Tp, A] per row).
template <int CELL, int NJ>
__global__ __launch_bounds__(RNT) void dec_step_bwd_bf_kernel(DecDev a, int t_att, int t_cell) {
    constexpr bool FAST = true;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const BfLds L = carve_bf(sm, a);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, NL = a.NL, E = a.E;
    const int S = D * NL, I0D = E + Hd + D, A8 = A >> 3, H8 = Hd >> 3;
    float* dctx = L.x0;
    float* dal = L.x1;

    if (t_att >= 0) {
        const int t = t_att;
        const float* dxr = a.dXin0 + ((size_t)t * B + b) * I0D;
        for (int i = tid; i < Hd; i += RNT) dctx[i] = dxr[E + i];
        for (int i = tid; i < Tp; i += RNT) L.ev[i] = a.alphas[((size_t)t * B + b) * Tp + i];
        for (int i = tid; i < A; i += RNT) L.qv[i] = a.Q[((size_t)t * B + b) * A + i];
        __syncthreads();
        const int len = a.enc_len[b];
        const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;
        {   // dalpha[t'] = dctx . enc[b,t',:] : one wave per frame, 8 frames in flight
            for (int tt = wv; tt < Tp; tt += 8 * RNW) {
                float acc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = 0.f;
                for (int h0 = lane; h0 < H8; h0 += 64) {
                    const uint4* ep = reinterpret_cast<const uint4*>(a.encbf) + (size_t)b * Tp * H8 + h0;
                    uint4 e8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t2 = tt + RNW * u;
                        e8[u] = t2 < lim ? ep[(size_t)t2 * H8] : make_uint4(0u, 0u, 0u, 0u);
                    }
                    const float4 d0 = reinterpret_cast<const float4*>(dctx)[h0 * 2], d1 = reinterpret_cast<const float4*>(dctx)[h0 * 2 + 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        float x[8];
                        unpack8(e8[u], x);
                        acc[u] += d0.x * x[0] + d0.y * x[1] + d0.z * x[2] + d0.w * x[3] + d1.x * x[4] + d1.y * x[5] + d1.z * x[6] + d1.w * x[7];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t2 = tt + RNW * u;
                    const float v = wave_sum(acc[u]);
                    if (lane == 0 && t2 < Tp) dal[t2] = t2 < lim ? v : 0.f;
                }
            }
        }
        __syncthreads();
        float dot = 0.f;
        for (int i = tid; i < Tp; i += RNT) dot = fmaf(L.ev[i], dal[i], dot);
        dot = block_sum<RNT>(dot, L.red);
        float* der = a.dE + ((size_t)t * B + b) * Tp;
        for (int i = tid; i < Tp; i += RNT) {
            const float de = L.ev[i] * (dal[i] - dot);   // d energy (0 where masked: alpha = 0)
            dal[i] = de;
            der[i] = de;
        }
        __syncthreads();

        // energies backward: 16-lane group per frame; per-lane partials of du and dq over its frames
        const int a8 = tid & 15, grp = tid >> 4;
        float q8[NJ][8], u8[NJ][8], du_acc[NJ][8], dq_acc[NJ][8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int a0 = a8 + 16 * j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q8[j][e] = a0 < A8 ? L.qv[a0 * 8 + e] : 0.f;
                u8[j][e] = a0 < A8 ? a.u[a0 * 8 + e] : 0.f;
                du_acc[j][e] = 0.f; dq_acc[j][e] = 0.f;
            }
        }
        {
            const uint4* kp = reinterpret_cast<const uint4*>(a.keysbf) + (size_t)b * Tp * A8;
            for (int tb = grp; tb < lim; tb += 4 * 64) {
                uint4 k8[4][NJ];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tt = tb + 64 * u;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int a0 = a8 + 16 * j;
                        k8[u][j] = (tt < lim && a0 < A8) ? kp[(size_t)tt * A8 + a0] : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tt = tb + 64 * u;
                    if (tt >= lim) continue;
                    const float de = dal[tt];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (a8 + 16 * j < A8) {
                            float k[8];
                            unpack8(k8[u][j], k);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float v = tanhx<FAST>(k[e] + q8[j][e]);
                                du_acc[j][e] = fmaf(de, v, du_acc[j][e]);
                                dq_acc[j][e] = fmaf(de * u8[j][e], 1.f - v * v, dq_acc[j][e]);
                            }
                        }
                    }
                }
            }
        }
        // reduce the 64 group partials: 4 groups per wave by shuffles, 16 waves through LDS (dq | du side by side)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int a0 = a8 + 16 * j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = dq_acc[j][e], w = du_acc[j][e];
                v = xor16_sum(v); v = xor32_sum(v);
                w = xor16_sum(w); w = xor32_sum(w);
                if (lane < 16 && a0 < A8) {
                    L.scr[wv * 2 * A + a0 * 8 + e] = v;
                    L.scr[wv * 2 * A + A + a0 * 8 + e] = w;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * A; i += RNT) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < RNW; ++w) sacc += L.scr[w * 2 * A + i];
            if (i < A) { L.qv[i] = sacc; a.dQ[((size_t)t * B + b) * A + i] = sacc; }     // qv now holds dq
            else a.duRows[(size_t)b * A + (i - A)] += sacc;
        }
        __syncthreads();
        {   // d state = dq . Ws^T : 16-lane group per state row (256-byte bf16 row at A = 128), 8 rows in flight
            float dq8[NJ][8];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) dq8[j][e] = (a8 + 16 * j) < A8 ? L.qv[(a8 + 16 * j) * 8 + e] : 0.f;
            const uint4* wp = reinterpret_cast<const uint4*>(a.Wsbf);
            for (int ib = grp; ib < S; ib += 8 * 64) {
                uint4 w8[8][NJ];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = ib + 64 * u;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int a0 = a8 + 16 * j;
                        w8[u][j] = (i < S && a0 < A8) ? wp[(size_t)i * A8 + a0] : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = ib + 64 * u;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float w[8];
                        unpack8(w8[u][j], w);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc = fmaf(w[e], dq8[j][e], acc);
                    }
                    acc = sub16_sum(acc);
                    if (a8 == 0 && i < S) {
                        const int l = i / D, d = i % D;
                        a.dH[((size_t)l * B + b) * D + d] = a.rec[l][(size_t)b * a.recLd[l] + a.recOff[l] + d] + acc;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (t_cell >= 0) cell_bwd_row<CELL, FAST>(a, NL - 1, t_cell, b, a.dHl + (size_t)t_cell * B * D, D);
}

// prefetching gradient row kernel (same eligibility as dec_step_fwd_pf_kernel).  The recurrent inputs are the
// dXin0 row of step t_att (context and state gradient) and this row's dC; everything else is issued at once.
// dalpha = enc . dctx and dstate = Ws . dq contract over pairs that are adjacent in the natural layouts.
// LOOP (dec_loop_bwd_kernel): one iteration of a persistent row workgroup: the dXin0 row of step t_att (context and state
// gradient) arrives as granules from the product workgroups, the bf16 gate gradient of step t_cell leaves as granules, dC and
// the running du column are carried in registers.
#ifndef LAS_KEYS_HOISTED
#define LAS_KEYS_HOISTED 1
#endif
// the keys of the energies gradient: one column pair per lane, frame wv + 16 u.  They do not change over the loop: the loop kernel
// loads them ONCE (NE registers) -- requested per step they cost 1.2 us of the 10.5 (tools/micro/bench_fused.hip, round 4)
template <int NE>
__device__ __forceinline__ void bwd_keys_load(const DecDev& a, const int b, const int tid, unsigned (&k2)[NE]) {
    const int lane = tid & 63, wv = tid >> 6, A2 = a.A >> 1, c2c = lane < A2 ? lane : A2 - 1;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int t2 = wv + RNW * u, t2c = t2 < a.Tp ? t2 : a.Tp - 1;
        k2[u] = reinterpret_cast<const unsigned*>(a.keysbf)[(LAS_ABL_SP & 256) ? (size_t)(tid & 63) : ((size_t)b * a.Tp + t2c) * A2 + c2c];
    }
}
template <int CELL, int NE, bool LOOP, bool LOC = false>
__device__ __forceinline__ void pf_bwd_row(const DecDev& a, const int t_att, const int t_cell, const int b, const int tid, float* sm,
                                           float& dccar, float& ducar, const bool local, const unsigned (&k2h)[NE], const bool acts) {
    static_assert(LOOP || !LOC, "location-aware attention is served by the loop kernels only");
    constexpr int LC = 10;                            // channels the location-aware variant keeps in registers (a.C <= LC, host-checked)
    constexpr bool FAST = true;
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    constexpr int NK = (16 * NE + 63) / 64;            // frames per 16-lane group (64 groups): T' <= 16*NE
    STAMPX(10);
    const BfLds L = carve_bf(sm, a);
    const int lane = tid & 63, wv = tid >> 6;
    const int a8 = tid & 15, grp = tid >> 4;
    const int B = a.B, Tp = a.Tp, Hd = a.Hd, A = a.A, D = a.D, E = a.E, U = a.U;
    const int S = D, GD = G * D, I0D = E + Hd + D, A8 = A >> 3, H8 = Hd >> 3;
    unsigned int* dcp = reinterpret_cast<unsigned int*>(L.x0);   // packed dctx pairs [Hd/2]
    unsigned int* dqp = reinterpret_cast<unsigned int*>(L.hl);   // packed dq pairs   [A/2]
    float* dal = L.x1;
    float* dhs = L.s_state;                         // gradient of the state consumed at step t_att (attention path)
    const bool att = t_att >= 0, cel = t_cell >= 0;

    // ---- loads, in consumption order; unconditional with clamped addresses (see dec_step_fwd_pf_kernel)
    const int ta = att ? t_att : 0, tcl = cel ? t_cell : 0;
    const int dd = tid < D ? tid : D - 1, h2c = tid < (Hd >> 1) ? tid : (Hd >> 1) - 1, tpc = tid < Tp ? tid : Tp - 1;
    const int a2c = tid < 2 * A ? tid : 2 * A - 1, a8c = a8 < A8 ? a8 : A8 - 1, l8c = lane < H8 ? lane : H8 - 1;
    float apv = 0.f;                                  // LOC: the alignment that entered step t_att's conv (alpha_{t-1}, or align0 / zeros)
    if (LOC && !acts) {
        const float* aps = ta > 0 ? a.alphas + ((size_t)(ta - 1) * B + b) * Tp : (a.align0 ? a.align0 + (size_t)b * Tp : a.alphas + (size_t)b * Tp);
        apv = aps[tpc];
        if (ta == 0 && !a.align0) apv = 0.f;
    }
    if (LOC && att && !acts) {   // recompute f = conv1d(alpha_{t-1}) of step t_att while dXin0 is on its way -- before the bulk loads (register pressure); keep it for the after-loop keys / Wf gradient
        if (tid < Tp) L.aprev[tid] = apv;
        lds_barrier();
        loc_conv_mfma(L, a, tid);
        lds_barrier();
        float* fs = a.fcSave + ((size_t)ta * B + b) * Tp * a.C;
        for (int i = tid; i < Tp * a.C; i += RNT) fs[i] = L.fc[i];
    }
    const float* dxr = a.dXin0 + ((size_t)ta * B + b) * I0D;
    float2 dcv = make_float2(0.f, 0.f);
    float recv0 = 0.f;
    if (!LOOP) { dcv = reinterpret_cast<const float2*>(dxr + E)[h2c]; recv0 = dxr[E + Hd + dd]; }
    const float alv = a.alphas[((size_t)ta * B + b) * Tp + tpc];
    // threads [0, A) pick up the query column, threads [A, 2A) this row's running du column
    float qd = a2c < A ? a.Q[((size_t)ta * B + b) * A + a2c] : a.duRows[(size_t)b * A + (a2c - A)];
    if (LOOP && a2c >= A) qd = ducar;                 // (duRows starts at zero; the register copy is the live one)
    const int len = a.enc_len[b];
    // encoder rows for dalpha: 16-lane group per frame (3 frames per group), lane a8 covers column chunks a8 + 16 i.
    // Loop kernels, additive attention: the first TRES frames are resident in LDS (EncRes; copied in by dec_loop_bwd_kernel); their
    // lanes request an offset beyond the buffer's range instead (zeros, no memory access), rounds that are resident as a whole hold
    // no registers at all.
    constexpr int TRES = LOOP ? 16 * EncRes<NE, LOC, true>::N : 0;        // resident frames
    constexpr int UF = TRES / 64;                                  // rounds of 64 frames that are resident as a whole
    constexpr int NKR = NK - UF > 0 ? NK - UF : 1;
    uint4 e8[NKR][4];
    const int A2 = A >> 1, c2c = lane < A2 ? lane : A2 - 1;
    unsigned k2[NE];                                  // keys for the energies gradient (or the saved activations): one column pair per lane,
                                                      // one frame per wave and round (sums over frames stay inside the lane, no cross-lane reduction)
    float2 u2;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, cv = 0.f, cpv = 0.f, hv = 0.f, dcr = 0.f, dhl;     // cell part operands (saved by the forward pass)
    float* gp = a.gates + (((size_t)0 * U + tcl) * B + b) * GD;
    const bool polls = LOOP && att && wv * 64 < (D > (Hd >> 1) ? D : (Hd >> 1));   // dXin0[t_att] from the product workgroups
    if (!LOC) {
        {
            const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.encbf) + (size_t)b * Tp * Hd, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int u = UF; u < NK; ++u) {
                const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hc = a8 + 16 * i, hcc = hc < H8 ? hc : H8 - 1;
                    const unsigned off = (LAS_ABL_SP & 16) ? (unsigned)(tid & 63) * 16u : (unsigned)(((size_t)ttc * H8 + hcc) * 16);
                    e8[u - UF][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ers, tt < TRES ? 0x80000000u : off, 0, 0));
                }
            }
        }
        if (acts) {   // the forward rows kept tanh(keys + q) of this step (fp16 pairs): nothing to recompute
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int t2 = wv + RNW * u, t2c = t2 < Tp ? t2 : Tp - 1;
                k2[u] = (a.actS + LAS_ACT_HDR)[(LAS_ABL_SP & 256) ? (size_t)(tid & 63) : (((size_t)ta * B + b) * Tp + t2c) * A2 + c2c];
            }
        } else if (LOOP && LAS_KEYS_HOISTED) {
#pragma unroll
            for (int u = 0; u < NE; ++u) k2[u] = k2h[u];
        } else {
            bwd_keys_load<NE>(a, b, tid, k2);
        }
        u2 = reinterpret_cast<const float2*>(a.u)[c2c];
        dhl = a.dHl[((size_t)tcl * B + b) * D + dd];
        if (CELL == LAS_CELL_LSTM) {
            gs[0] = gp[dd]; gs[1] = gp[D + dd]; gs[2] = gp[2 * D + dd]; gs[3] = gp[3 * D + dd];
            cv = a.cs[(((size_t)0 * (U + 1) + tcl + 1) * B + b) * D + dd];
            cpv = a.cs[(((size_t)0 * (U + 1) + tcl) * B + b) * D + dd];
            dcr = LOOP ? dccar : a.dC[(size_t)b * D + dd];
        } else {
            hv = a.hs[(((size_t)0 * (U + 1) + tcl + 1) * B + b) * D + dd];
        }
        if (polls) {
            const __amdgpu_buffer_rsrc_t rs = granule_rsrc(a.lp.gC);
            const unsigned tag = (unsigned)t_att + 1u;
            const unsigned o0 = (unsigned)(((size_t)b * a.lp.gC_row + h2c) * 16), o1 = (unsigned)(((size_t)b * a.lp.gC_row + ((Hd + dd) >> 1)) * 16);
            u32x4_t q0 = granule16_load(rs, o0), q1 = granule16_load(rs, o1);
            int budget = a.lp.budget;
            for (;;) {
                const bool ok0 = q0.x == tag && q0.w == tag, ok1 = q1.x == tag && q1.w == tag;
                if (ok0 && ok1) break;
                if (--budget == 0) LOOP_POLL_TIMEOUT(a.lp);    // a product workgroup never ran: report, never hang
                __builtin_amdgcn_s_sleep(1);
                if (!ok0) q0 = granule16_load(rs, o0);
                if (!ok1) q1 = granule16_load(rs, o1);
            }
            dcv = make_float2(__uint_as_float(q0.y), __uint_as_float(q0.z));
            recv0 = __uint_as_float((dd & 1) ? q1.z : q1.y);
        }
    } else {
        // location-aware rows that were handed f and the activations (acts) run the transposed conv of the PREVIOUS iteration while dXin0 is on
        // its way: the encoder rows' 48 destination registers and the late operands are requested behind it (in front of it they spill, and
        // a reload from scratch drains every load in flight), the dXin0 granules in front of it
        const bool convT_first = LOC && acts && att && t_att + 1 < U;
        auto e8_issue = [&]() __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.encbf) + (size_t)b * Tp * Hd, 0, 0x7fffffff, 0x00020000);
    #pragma unroll
            for (int u = UF; u < NK; ++u) {
                const int tt = grp + 64 * u, ttc = tt < Tp ? tt : Tp - 1;
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hc = a8 + 16 * i, hcc = hc < H8 ? hc : H8 - 1;
                    const unsigned off = (LAS_ABL_SP & 16) ? (unsigned)(tid & 63) * 16u : (unsigned)(((size_t)ttc * H8 + hcc) * 16);
                    e8[u - UF][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ers, tt < TRES ? 0x80000000u : off, 0, 0));
                }
            }
        };
        auto late_operands = [&]() __attribute__((always_inline)) {      // consumed behind the attention part
            if (acts) {   // the forward rows kept tanh(keys + q [+ f . Wf]) of this step (fp16 pairs): nothing to recompute
    #pragma unroll
                for (int u = 0; u < NE; ++u) {
                    const int t2 = wv + RNW * u, t2c = t2 < Tp ? t2 : Tp - 1;
                    k2[u] = (a.actS + LAS_ACT_HDR)[(LAS_ABL_SP & 256) ? (size_t)(tid & 63) : (((size_t)ta * B + b) * Tp + t2c) * A2 + c2c];
                }
            } else if (LOOP && LAS_KEYS_HOISTED && !LOC) {
    #pragma unroll
                for (int u = 0; u < NE; ++u) k2[u] = k2h[u];
            } else {
                bwd_keys_load<NE>(a, b, tid, k2);
            }
            u2 = reinterpret_cast<const float2*>(a.u)[c2c];
            dhl = a.dHl[((size_t)tcl * B + b) * D + dd];
            if (CELL == LAS_CELL_LSTM) {
                gs[0] = gp[dd]; gs[1] = gp[D + dd]; gs[2] = gp[2 * D + dd]; gs[3] = gp[3 * D + dd];
                cv = a.cs[(((size_t)0 * (U + 1) + tcl + 1) * B + b) * D + dd];
                cpv = a.cs[(((size_t)0 * (U + 1) + tcl) * B + b) * D + dd];
                dcr = LOOP ? dccar : a.dC[(size_t)b * D + dd];
            } else {
                hv = a.hs[(((size_t)0 * (U + 1) + tcl + 1) * B + b) * D + dd];
            }
        };
        if (!convT_first) { e8_issue(); late_operands(); }
        const __amdgpu_buffer_rsrc_t rs = granule_rsrc(a.lp.gC);
        const unsigned o0 = (unsigned)(((size_t)b * a.lp.gC_row + h2c) * 16), o1 = (unsigned)(((size_t)b * a.lp.gC_row + ((Hd + dd) >> 1)) * 16);
        u32x4_t q0 = {0u, 0u, 0u, 0u}, q1 = q0;
        if (polls) { q0 = granule16_load(rs, o0); q1 = granule16_load(rs, o1); }
        if (convT_first) {
            // the forward rows kept f and the activations: nothing to recompute in this iteration.  The wait for dXin0 takes the transposed
            // conv of the PREVIOUS iteration instead (d f of step t_att + 1 is still in LDS; what it sends back to alpha_{t_att} is consumed
            // by this iteration's softmax gradient): 4 us per step off the chain
            STAMPX(28);
            loc_convT_mfma(L, a, tid);
            STAMPX(29);
            lds_barrier();
            e8_issue();
            late_operands();
        }
        if (polls) {
            const unsigned tag = (unsigned)t_att + 1u;
            int budget = a.lp.budget;
            for (;;) {
                const bool ok0 = q0.x == tag && q0.w == tag, ok1 = q1.x == tag && q1.w == tag;
                if (ok0 && ok1) break;
                if (--budget == 0) LOOP_POLL_TIMEOUT(a.lp);    // a product workgroup never ran: report, never hang
                __builtin_amdgcn_s_sleep(1);
                if (!ok0) q0 = granule16_load(rs, o0);
                if (!ok1) q1 = granule16_load(rs, o1);
            }
            dcv = make_float2(__uint_as_float(q0.y), __uint_as_float(q0.z));
            recv0 = __uint_as_float((dd & 1) ? q1.z : q1.y);
        }
    }
    STAMPX(24);
    const float recv = att ? recv0 : 0.f;

    STAMPX(11);
    if (tid < D) dhs[tid] = 0.f;
    if (att) {
        const int t = t_att;
        const int lim = len > 0 ? (len < Tp ? len : Tp) : Tp;
        if (tid < (Hd >> 1)) dcp[tid] = f2bf2(dcv.x, dcv.y);
        if (tid < Tp) L.ev[tid] = alv;
        if (tid < A) L.qv[tid] = qd;
        lds_barrier();
    STAMPX(12);
        {   // dalpha[t'] = dctx . enc[b,t',:] : 16-lane group per frame, packed pairs along the columns
            float acc[NK];
#pragma unroll
            for (int u = 0; u < NK; ++u) acc[u] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hc = a8 + 16 * i;
                const uint4 d4 = hc < H8 ? reinterpret_cast<const uint4*>(dcp)[hc] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int u = 0; u < NK; ++u) {
                    const int tt = grp + 64 * u;
                    uint4 ev = e8[u < UF ? 0 : u - UF][i];
                    if (u < UF || (u * 64 < TRES && tt < TRES)) ev = L.encl[(size_t)tt * H8 + (hc < H8 ? hc : H8 - 1)];   // (whole 16-lane groups: wave-uniform up to the group)
                    acc[u] = dot2bf(ev.x, d4.x, acc[u]); acc[u] = dot2bf(ev.y, d4.y, acc[u]);
                    acc[u] = dot2bf(ev.z, d4.z, acc[u]); acc[u] = dot2bf(ev.w, d4.w, acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < NK; ++u) {
                const int tt = grp + 64 * u;
                const float v = sub16_sum(acc[u]);
                if (a8 == 0 && tt < Tp) dal[tt] = tt < lim ? (LOC && t + 1 < U ? v + L.daext[tt] : v) : 0.f;   // (+ what step t+1's conv sent back)
            }
        }
        // the state-gradient operand (the encoder registers are free), consumed after the energies: 128 KB per row and step.
        // LAS_W8_LATE: issued one load per frame inside the energies-gradient loop (transcendental work of the other waves covers
        // a wave that waits to issue) instead of all at once in front of a barrier -- see pf_fwd_row
        uint4 w8[8];
        auto w8_load = [&](const int u) __attribute__((always_inline)) {
            const int kk = grp + 64 * u;
            const int kkc = kk < S ? kk : S - 1;
            w8[u] = reinterpret_cast<const uint4*>(a.Wsbf)[(LAS_ABL_SP & 512) ? (size_t)(tid & 63) : (size_t)kkc * A8 + a8c];
        };
        if (!LAS_W8_LATE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w8_load(u);
        }
        lds_barrier();
    STAMPX(14);
        float de_own = 0.f;
        if (wv < NK) {   // the first NK waves own one frame per lane; each sums alpha . dalpha for itself
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const int i = lane + 64 * j;
                if (i < Tp) dot = fmaf(L.ev[i], dal[i], dot);
            }
            dot = wave_sum(dot);
            if (tid < Tp) {
                de_own = L.ev[tid] * (dal[tid] - dot);          // d energy (0 where masked: alpha = 0)
                a.dE[((size_t)t * B + b) * Tp + tid] = de_own;
            }
        }
        lds_barrier();                                           // all three waves have read dal
        if (tid < Tp) dal[tid] = de_own;
        lds_barrier();
    STAMPX(15);
        float du0 = 0.f, du1 = 0.f, dq0 = 0.f, dq1 = 0.f;
        // energies backward: sums over this wave's frames stay in the lane (columns 2*lane, 2*lane+1).  Two instances of the same
        // loop: the saved activations (acts, wave-uniform) or their recomputation from keys + q [+ f . Wf] -- 20 tanh per lane and,
        // location-aware, 20 FMAs per frame: 1.2 of the 10.5 us of a gradient step (tools/micro/bench_fused.hip)
        auto energies_bwd = [&](auto ACT) __attribute__((always_inline)) {
            constexpr bool ACTS = decltype(ACT)::value;
            const float q0 = ACTS ? 0.f : L.qv[2 * c2c], q1 = ACTS ? 0.f : L.qv[2 * c2c + 1];
            float2 wf2[LOC && !ACTS ? LC : 1];        // LOC: this lane's two columns of every Wf row
            if (LOC && !ACTS) {
#pragma unroll
                for (int c = 0; c < LC; ++c) wf2[c] = c < a.C ? reinterpret_cast<const float2*>(L.wfl + (size_t)c * A)[c2c] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                if (LAS_W8_LATE && (!LOC || ACTS) && u < 8) { w8_load(u); __builtin_amdgcn_sched_barrier(0); }
                const int t2 = wv + RNW * u;
                const float de = t2 < lim ? dal[t2] : 0.f;
                float v0, v1;
                if (ACTS) {
                    const float2 h = h22f(k2[u]);
                    const bool on = t2 < len && t2 < Tp;          // (frames the forward row did not visit hold no value)
                    v0 = on ? h.x : 0.f; v1 = on ? h.y : 0.f;
                } else {
                    float p0 = __uint_as_float(k2[u] << 16) + q0, p1 = __uint_as_float(k2[u] & 0xffff0000u) + q1;
                    if (LOC) {   // + f[t2, :] . Wf
                        const float* fr = L.fc + (t2 < Tp ? t2 : Tp - 1) * a.C;
#pragma unroll
                        for (int c = 0; c < LC; ++c) {
                            const float f = c < a.C ? fr[c] : 0.f;
                            p0 = fmaf(f, wf2[LOC && !ACTS ? c : 0].x, p0); p1 = fmaf(f, wf2[LOC && !ACTS ? c : 0].y, p1);
                        }
                    }
                    v0 = tanhx<FAST>(p0); v1 = tanhx<FAST>(p1);
                }
                du0 = fmaf(de, v0, du0); du1 = fmaf(de, v1, du1);
                if (!LOC) { dq0 = fmaf(de * u2.x, 1.f - v0 * v0, dq0); dq1 = fmaf(de * u2.y, 1.f - v1 * v1, dq1); }
                if (LOC) {   // keep the row of d(pre-tanh) (bf16 pairs): d f = dv . Wf^T is one small MFMA product after the loop
                    const float dv0 = de * u2.x * (1.f - v0 * v0), dv1 = de * u2.y * (1.f - v1 * v1);
                    dq0 += dv0; dq1 += dv1;
                    if (lane < A2 && t2 < Tp) L.dvb[t2 * A2 + lane] = f2bf2(dv0, dv1);
                }
            }
        };
        {
            if (acts) energies_bwd(std::true_type{}); else energies_bwd(std::false_type{});
    STAMPX(16);
            if (lane < A2) {
                reinterpret_cast<float2*>(L.scr + wv * 2 * A)[lane] = make_float2(dq0, dq1);
                reinterpret_cast<float2*>(L.scr + wv * 2 * A + A)[lane] = make_float2(du0, du1);
            }
            if (LAS_W8_LATE && LOC && !acts) {   // location-aware, recomputing: the loop above is at the register limit (Wf columns, d v rows) -- the Ws rows are
#pragma unroll                  // requested here; the d f product and the dq / du reduction below cover their latency
                for (int u = 0; u < 8; ++u) w8_load(u);
            }
        }
        lds_barrier();
        if (LOC && wv * 16 < Tp) {
            // d f[t', c] = sum_a dv[t', a] Wf[c, a]: wave w owns frames [16 w, 16 w + 16): A fragments = the dv rows (bf16), B fragments =
            // Wf^T (bf16, channels padded to 16), fp32 accumulation.  (A first version reduced 10 per-lane partial sums per frame over
            // the wave with a 17-shuffle butterfly: 100 butterflies per step and 43 spilled VGPRs, +11 us per step.)
            const int g = lane >> 4, c16 = lane & 15;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < A / 32; ++ks) {
                const u16x8_t av = *reinterpret_cast<const u16x8_t*>(reinterpret_cast<const unsigned short*>(L.dvb) + (size_t)(wv * 16 + c16) * A + ks * 32 + g * 8);
                const u16x8_t bv = *reinterpret_cast<const u16x8_t*>(L.wfb + (size_t)c16 * A + ks * 32 + g * 8);
                acc = mfma_bf16_16x16x32(av, bv, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = wv * 16 + g * 4 + r;
                if (fr < Tp && c16 < a.C) L.dfc[c16 * loc_dpad(a) + fr] = acc[r];
            }
        }
        {
            float sacc = 0.f;
            if (tid < 2 * A) {
#pragma unroll
                for (int w = 0; w < RNW; ++w) sacc += L.scr[w * 2 * A + tid];
            }
            const float nb = lane_xor1(sacc);
            if (tid < A) {
                a.dQ[((size_t)t * B + b) * A + tid] = sacc;
                if (!(tid & 1)) dqp[tid >> 1] = f2bf2(sacc, nb);
            } else if (tid < 2 * A) {
                a.duRows[(size_t)b * A + (tid - A)] = qd + sacc;
                ducar = qd + sacc;
            }
        }
        lds_barrier();
    STAMPX(17);
        {   // d state = Ws . dq : 16-lane group per state row, 8 columns (4 pairs) per lane, 8 prefetched rows
            const uint4 q4 = a8 < A8 ? reinterpret_cast<const uint4*>(dqp)[a8] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = grp + 64 * u;
                float acc = dot2bf(w8[u].x, q4.x, 0.f);
                acc = dot2bf(w8[u].y, q4.y, acc); acc = dot2bf(w8[u].z, q4.z, acc); acc = dot2bf(w8[u].w, q4.w, acc);
                acc = sub16_sum(acc);
                if (a8 == 0 && i < S) dhs[i] = acc;
            }
        }
        if (LOC) {
            // d f of this step is complete (the barriers of the dq / du reduction): keep it for the after-loop filter gradient, and send
            // d alpha_{t-1}[src] = sum_k sum_c dfc[src - k + pad, c] w[k, c] back to the previous step (loc_convT_mfma; the scratch is free)
            const int C = a.C;
            float* ds = a.dfcSave + ((size_t)t * B + b) * Tp * C;
            for (int i = tid; i < Tp * C; i += RNT) ds[i] = L.dfc[(i % C) * loc_dpad(a) + i / C];      // (saved frame-major, as the after-loop kernels read it)
            if (!acts) {
                STAMPX(28);
                loc_convT_mfma(L, a, tid);
                STAMPX(29);
            }
        }
        lds_barrier();
    } else {
        lds_barrier();
    }
    STAMPX(18);
    if (cel && tid < D) {   // gate backward of step t_cell (D % 4 == 0: whole 4-lane groups)
        const float dh = dhs[tid] + recv + dhl;
        unsigned short* gb = a.dgbf + (size_t)b * GD;
        const __amdgpu_buffer_rsrc_t grs = granule_rsrc(a.lp.gA);     // LOOP: the bf16 copy leaves as granules (tag t_cell + 1)
        const size_t gg0 = (size_t)b * a.lp.gA_row;
        const unsigned gtag = (unsigned)t_cell + 1u;
        if (CELL == LAS_CELL_LSTM) {
            const float gi = gs[0], gj = gs[1], gf = gs[2], go = gs[3];
            const float tc = tanhx<FAST>(cv);
            const float dc = dcr + dh * go * (1.f - tc * tc);
            a.dC[(size_t)b * D + tid] = dc * gf;
            dccar = dc * gf;
            const float di = dc * gj * gi * (1.f - gi), dj = dc * gi * (1.f - gj * gj);
            const float df = dc * cpv * gf * (1.f - gf), dO = dh * tc * go * (1.f - go);
            gp[tid] = di; gp[D + tid] = dj; gp[2 * D + tid] = df; gp[3 * D + tid] = dO;
            if (LOOP) {
                put4_bf16(grs, gg0, tid, di, gtag, local); put4_bf16(grs, gg0, D + tid, dj, gtag, local);
                put4_bf16(grs, gg0, 2 * D + tid, df, gtag, local); put4_bf16(grs, gg0, 3 * D + tid, dO, gtag, local);
            } else {
                gb[tid] = f2bf(di); gb[D + tid] = f2bf(dj); gb[2 * D + tid] = f2bf(df); gb[3 * D + tid] = f2bf(dO);
            }
        } else {
            const float dp = dh * (1.f - hv * hv);
            gp[tid] = dp;
            if (LOOP) put4_bf16(grs, gg0, tid, dp, gtag, local);
            else gb[tid] = f2bf(dp);
        }
    }
    STAMPX(19);
}

template <int CELL, int NE>
__global__ __launch_bounds__(RNT) void dec_step_bwd_pf_kernel(DecDev a, int t_att, int t_cell) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float dccar = 0.f, ducar = 0.f;
    unsigned nok[NE] = {};
    const bool acts = a.actS && a.actS[0] == LAS_ACT_MAGIC;
    pf_bwd_row<CELL, NE, false>(a, t_att, t_cell, blockIdx.x, threadIdx.x, sm, dccar, ducar, false, nok, acts);
}

// the whole gradient loop in one launch (see dec_loop_fwd_kernel): the product workgroups compute
// dXin0[t] = dG(t) . W0^T from the gate-gradient granules of step t (plain fp32 copy of every column for the after-loop
// contractions + granules for the chain columns [E, I0D)), the row workgroups run attention backward of step t + 1 and
// the gate backward of step t
template <int CELL, int NE, bool LOC = false>
__global__ __launch_bounds__(RNT) void dec_loop_bwd_kernel(DecDev a) {
    constexpr int TPW = 3, KW = 4;                                    // 26 x 3 column tiles >= 64, 16 waves x 4 k-steps x 32 >= 2048 (host-checked)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const bool local = loop_same_xcd(a.lp.xcc, x, a.lp.pn + a.lp.R, threadIdx.x, a.lp.status);
    if (j < a.lp.pn) { loop_product<TPW, KW, 2>(a.lp, a.U, true, x, j, local, sm); return; }
    const int b = (j - a.lp.pn) * 8 + x;
    if (b >= a.B) return;
    float dccar = 0.f, ducar = 0.f;
    if (LOC) { loc_stage_lds(carve_bf(sm, a), a, threadIdx.x); lds_barrier(); }
    if (EncRes<NE, LOC, true>::N > 0) {                               // resident encoder frames (natural layout): once per launch
        const int H8 = a.Hd >> 3;
        uint4* dst = carve_bf(sm, a).encl;
        const uint4* src = reinterpret_cast<const uint4*>(a.encbf) + (size_t)b * a.Tp * H8;
        for (int i = threadIdx.x; i < EncRes<NE, LOC, true>::N * 16 * H8; i += RNT) {
            const int tt = i / H8;
            dst[i] = src[(size_t)(tt < a.Tp ? tt : a.Tp - 1) * H8 + (i - tt * H8)];
        }
    }
    unsigned k2[NE] = {};
    const bool acts = a.actS && a.actS[0] == LAS_ACT_MAGIC;           // the forward rows kept their attention activations
    if (LAS_KEYS_HOISTED && !LOC && !acts) bwd_keys_load<NE>(a, b, threadIdx.x, k2);      // (location-aware: the loop is at the register limit)
    for (int t = a.U - 1; t >= -1; --t) {
        int bb = b, tid = threadIdx.x;
        asm volatile("" : "+s"(bb), "+v"(tid));                       // see dec_loop_fwd_kernel
        pf_bwd_row<CELL, NE, true, LOC>(a, (t + 1 < a.U) ? t + 1 : -1, t, bb, tid, sm, dccar, ducar, local, k2, acts);
        lds_barrier();
    }
}

// Location-aware attention, after the gradient loop.  (1) keys gradient as dkeys_kernel, with the conv term f[t, b, t', :] . Wf in the
// pre-activation, and in the same pass the filter-projection gradient dWf[c, a] = sum_{t, t'} f[t, b, t', c] dv[t, b, t', a]: per
// workgroup (utterance b, 8 frames) one [C, A] partial, written to the row-private slice dWfRows[b][blockIdx.x] (reduced over the
// slices by las_colsum afterwards; C <= 10, A <= 128, ceil(T' / 8) <= RNG slices -- host-checked).
__global__ __launch_bounds__(256) void dkeys_loc_kernel(DecDev a, float* __restrict__ dKeys) {
    constexpr int LC = 10;
    __shared__ float wf[LC * 128];
    __shared__ float part[8][LC][128 + 4];
    const int b = blockIdx.y, fr = threadIdx.x >> 5, tt = blockIdx.x * 8 + fr, sl = threadIdx.x & 31;
    const int B = a.B, Tp = a.Tp, A = a.A, U = a.U, C = a.C;
    for (int i = threadIdx.x; i < C * A; i += 256) wf[i] = a.Wf[i];
    __syncthreads();
    const bool on = tt < Tp && sl < A / 4;
    const int ttc = tt < Tp ? tt : Tp - 1, a4 = sl < A / 4 ? sl : A / 4 - 1;
    const unsigned short* kr = a.keysbf + ((size_t)b * Tp + ttc) * A + a4 * 4;
    const float k0 = bf2f(kr[0]), k1 = bf2f(kr[1]), k2 = bf2f(kr[2]), k3 = bf2f(kr[3]);
    const float4 u4 = reinterpret_cast<const float4*>(a.u)[a4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dwf[LC];
#pragma unroll
    for (int c = 0; c < LC; ++c) dwf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int t = 0; t < U; ++t) {
        const float de = a.dE[((size_t)t * B + b) * Tp + ttc];
        const float4 q4 = reinterpret_cast<const float4*>(a.Q + ((size_t)t * B + b) * A)[a4];
        const float* fr_ = a.fcSave + (((size_t)t * B + b) * Tp + ttc) * C;
        float f[LC];
#pragma unroll
        for (int c = 0; c < LC; ++c) f[c] = c < C ? fr_[c] : 0.f;
        float p0 = k0 + q4.x, p1 = k1 + q4.y, p2 = k2 + q4.z, p3 = k3 + q4.w;
#pragma unroll
        for (int c = 0; c < LC; ++c) {
            const float4 w4 = reinterpret_cast<const float4*>(wf + (c < C ? c : 0) * A)[a4];
            p0 = fmaf(f[c], w4.x, p0); p1 = fmaf(f[c], w4.y, p1); p2 = fmaf(f[c], w4.z, p2); p3 = fmaf(f[c], w4.w, p3);
        }
        const float v0 = tanh_fast(p0), v1 = tanh_fast(p1), v2 = tanh_fast(p2), v3 = tanh_fast(p3);
        const float4 dv = make_float4(de * u4.x * (1.f - v0 * v0), de * u4.y * (1.f - v1 * v1), de * u4.z * (1.f - v2 * v2), de * u4.w * (1.f - v3 * v3));
        acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
#pragma unroll
        for (int c = 0; c < LC; ++c) {
            dwf[c].x = fmaf(f[c], dv.x, dwf[c].x); dwf[c].y = fmaf(f[c], dv.y, dwf[c].y);
            dwf[c].z = fmaf(f[c], dv.z, dwf[c].z); dwf[c].w = fmaf(f[c], dv.w, dwf[c].w);
        }
    }
    if (on) {
        float4* dk = reinterpret_cast<float4*>(dKeys + ((size_t)b * Tp + tt) * A) + a4;
        float4 o = *dk;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *dk = o;
    }
#pragma unroll
    for (int c = 0; c < LC; ++c) {
        float4 v = on ? dwf[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* pr = &part[fr][c][sl * 4];
        pr[0] = v.x; pr[1] = v.y; pr[2] = v.z; pr[3] = v.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * A; i += 256) {
        const int c = i / A, col = i - c * A;
        float s_ = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) s_ += part[g8][c][col];
        a.dWfRows[(((size_t)b * RNG + blockIdx.x) * C + c) * A + col] += s_;
    }
}

// (2) filter and bias gradient: dlocw[k, c] = sum_{t, t'} dfc[t, b, t', c] alpha_{t-1}[b, t' + k - pad], dlocb[c] = sum dfc -- per utterance
// (row-private slices dlocwRows[b] / dlocbRows[b], reduced over the utterances afterwards).  Workgroup = (256 of the Kc x C outputs,
// utterance); every step's dfc row and alignment are staged in LDS once and shared by the workgroup's outputs.
__global__ __launch_bounds__(256) void dlocw_kernel(DecDev a) {
    extern __shared__ __attribute__((aligned(16))) float sm2[];
    const int b = blockIdx.y, B = a.B, Tp = a.Tp, U = a.U, C = a.C, pad = (a.Kc - 1) / 2;
    float* dfc = sm2;                    // [Tp * C]
    float* apv = sm2 + Tp * C;           // [Tp]
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool on = i < a.Kc * C;
    const int k = on ? i / C : 0, c = on ? i - k * C : 0;
    const int t0 = pad - k > 0 ? pad - k : 0, t1 = Tp < Tp + pad - k ? Tp : Tp + pad - k;
    float acc = 0.f, accb = 0.f;
    for (int t = 0; t < U; ++t) {
        __syncthreads();
        const float* ds = a.dfcSave + ((size_t)t * B + b) * Tp * C;
        for (int j = threadIdx.x; j < Tp * C; j += 256) dfc[j] = ds[j];
        for (int j = threadIdx.x; j < Tp; j += 256)
            apv[j] = t > 0 ? a.alphas[((size_t)(t - 1) * B + b) * Tp + j] : (a.align0 ? a.align0[(size_t)b * Tp + j] : 0.f);
        __syncthreads();
        if (on) for (int tt = t0; tt < t1; ++tt) acc = fmaf(dfc[tt * C + c], apv[tt + k - pad], acc);
        if (blockIdx.x == 0 && threadIdx.x < C) for (int tt = 0; tt < Tp; ++tt) accb += dfc[tt * C + threadIdx.x];
    }
    if (on) a.dlocwRows[(size_t)b * a.Kc * C + i] += acc;
    if (blockIdx.x == 0 && threadIdx.x < C) a.dlocbRows[(size_t)b * C + threadIdx.x] += accb;
}

// (2b) The same contraction on the matrix cores (round 4; the VALU kernel above took 1.44 ms per step at K = 201, C = 10, B = 48 --
// two LDS reads per multiply-add).  Per (step, utterance) it is a Toeplitz product
//     dlocw[k, c] += sum_t' P[t' + k] . dfc[t', c],     P = the previous alignment, zero-padded by (Kc - 1) / 2 on both sides,
// i.e. [M = taps] x [K = frames] x [N = channels] with A[k][t'] = P[t' + k]: lane (g, r) of an A fragment is 8 CONSECUTIVE elements
// of the padded array in LDS (no Toeplitz matrix is built, as in loc_conv_mfma); the B fragments are 8 consecutive frames of one
// channel from a channel-major LDS copy of the step's d f.  fp32 accuracy from bf16 MFMAs by x = hi + lo (hi.hi + hi.lo + lo.hi).
// Workgroup = (slice s of the decode steps, utterance b), 4 waves x up to 4 tap tiles; the slices' partial sums are reduced in fixed
// order afterwards (las_colsum over [B * DLW_SPLIT] rows): deterministic.  The bias gradient (column sums of d f) rides along.
constexpr int DLW_SPLIT = 8;
__global__ __launch_bounds__(256) void dlocw_mfma_kernel(DecDev a, float* __restrict__ wpart, float* __restrict__ bpart) {
    extern __shared__ __attribute__((aligned(16))) float sm3[];
    const int b = blockIdx.y, sp = blockIdx.x, B = a.B, Tp = a.Tp, U = a.U, C = a.C, Kc = a.Kc, pad = (Kc - 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, r = lane & 15;
    const int NKS = (Tp + 31) >> 5, TK = NKS * 32;            // frames rounded up to whole k-steps (the tail holds zeros)
    const int NMT = (Kc + 15) >> 4;                           // tap tiles (<= 16: Kc <= 256)
    const int LP = TK + 16 * NMT + 8;                         // padded alignment: index j <-> frame j - pad
    const int DP = TK + 4;                                    // pitch of the channel-major d f rows
    float* P = sm3;                                           // [LP]
    float* dT = sm3 + ((LP + 3) & ~3);                        // [16][DP]
    float* red = dT + 16 * DP;                                // [16][16] bias-gradient partials
    const int per = (U + DLW_SPLIT - 1) / DLW_SPLIT, ta = sp * per, tb = min(U, ta + per);
    f32x4_t acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (int i = tid; i < LP; i += 256) P[i] = 0.f;
    for (int i = tid; i < 16 * DP; i += 256) dT[i] = 0.f;
    for (int t = ta; t < tb; ++t) {
        __syncthreads();
        const float* ds = a.dfcSave + ((size_t)t * B + b) * Tp * C;
        for (int j = tid; j < Tp * C; j += 256) { const int tt = j / C, c = j - tt * C; dT[c * DP + tt] = ds[j]; }
        for (int j = tid; j < Tp; j += 256)
            P[pad + j] = t > 0 ? a.alphas[((size_t)(t - 1) * B + b) * Tp + j] : (a.align0 ? a.align0[(size_t)b * Tp + j] : 0.f);
        __syncthreads();
        {   // bias gradient: thread (channel tid >> 4, residue tid & 15) sums its frames
            const float* row = dT + (tid >> 4) * DP;
            for (int tt = tid & 15; tt < Tp; tt += 16) accb += row[tt];
        }
        for (int ks = 0; ks < NKS; ++ks) {
            unsigned int bh[4], bl[4];
            {
                const float4 x0 = *reinterpret_cast<const float4*>(dT + r * DP + ks * 32 + g * 8);
                const float4 x1 = *reinterpret_cast<const float4*>(dT + r * DP + ks * 32 + g * 8 + 4);
                const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned short h0 = f2bf(x[2 * e]), h1 = f2bf(x[2 * e + 1]);
                    bh[e] = (unsigned int)h0 | ((unsigned int)h1 << 16);
                    bl[e] = f2bf2(x[2 * e] - bf2f(h0), x[2 * e + 1] - bf2f(h1));
                }
            }
            const u16x8_t Bh = __builtin_bit_cast(u16x8_t, (u32x4_t){bh[0], bh[1], bh[2], bh[3]});
            const u16x8_t Bl = __builtin_bit_cast(u16x8_t, (u32x4_t){bl[0], bl[1], bl[2], bl[3]});
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mt = wv + 4 * i;
                if (mt < NMT) {
                    const float* Pp = P + ks * 32 + g * 8 + mt * 16 + r;
                    unsigned int ah[4], al[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = Pp[2 * e], x1 = Pp[2 * e + 1];
                        const unsigned short h0 = f2bf(x0), h1 = f2bf(x1);
                        ah[e] = (unsigned int)h0 | ((unsigned int)h1 << 16);
                        al[e] = f2bf2(x0 - bf2f(h0), x1 - bf2f(h1));
                    }
                    const u16x8_t Ah = __builtin_bit_cast(u16x8_t, (u32x4_t){ah[0], ah[1], ah[2], ah[3]});
                    const u16x8_t Al = __builtin_bit_cast(u16x8_t, (u32x4_t){al[0], al[1], al[2], al[3]});
                    acc[i][0] = mfma_bf16_16x16x32(Ah, Bh, acc[i][0]);
                    acc[i][1] = mfma_bf16_16x16x32(Ah, Bl, acc[i][1]);
                    acc[i][2] = mfma_bf16_16x16x32(Al, Bh, acc[i][2]);
                }
            }
        }
    }
    // C layout: lane (g, n = r) holds taps 16 mt + 4 g + i, channel n
    float* wo = wpart + ((size_t)b * DLW_SPLIT + sp) * Kc * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mt = wv + 4 * i;
        if (mt < NMT && r < C)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = mt * 16 + g * 4 + q;
                if (k < Kc) wo[k * C + r] = acc[i][0][q] + (acc[i][1][q] + acc[i][2][q]);
            }
    }
    __syncthreads();
    red[tid] = accb;
    __syncthreads();
    if (tid < C) {
        float s_ = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s_ += red[tid * 16 + q];
        bpart[((size_t)b * DLW_SPLIT + sp) * C + tid] = s_;
    }
}

// dKeys[b,t',:] = sum over steps t of dE[t,b,t'] * u * (1 - tanh^2(keys[b,t',:] + Q[t,b,:]))  (speed mode, after the loop)
// workgroup = (utterance, 8 encoder frames); 32 lanes x float4 over the attention dim per frame
__global__ __launch_bounds__(256) void dkeys_kernel(DecDev a, float* __restrict__ dKeys) {
    const int b = blockIdx.y, tt = blockIdx.x * 8 + (threadIdx.x >> 5), sl = threadIdx.x & 31;
    const int B = a.B, Tp = a.Tp, A = a.A, U = a.U;
    if (tt >= Tp) return;
    for (int a4 = sl; a4 < A / 4; a4 += 32) {
        const unsigned short* kr = a.keysbf + ((size_t)b * Tp + tt) * A + a4 * 4;
        const float k0 = bf2f(kr[0]), k1 = bf2f(kr[1]), k2 = bf2f(kr[2]), k3 = bf2f(kr[3]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int t = 0; t < U; ++t) {                  // (unrolled: eight steps' loads in flight -- the loop was bound by one L2 round trip per step)
            const float de = a.dE[((size_t)t * B + b) * Tp + tt];
            const float4 q4 = reinterpret_cast<const float4*>(a.Q + ((size_t)t * B + b) * A)[a4];
            const float v0 = tanh_fast(k0 + q4.x), v1 = tanh_fast(k1 + q4.y), v2 = tanh_fast(k2 + q4.z), v3 = tanh_fast(k3 + q4.w);
            acc.x = fmaf(de, 1.f - v0 * v0, acc.x); acc.y = fmaf(de, 1.f - v1 * v1, acc.y);
            acc.z = fmaf(de, 1.f - v2 * v2, acc.z); acc.w = fmaf(de, 1.f - v3 * v3, acc.w);
        }
        const float4 u4 = reinterpret_cast<const float4*>(a.u)[a4];
        float4* o = reinterpret_cast<float4*>(dKeys + ((size_t)b * Tp + tt) * A) + a4;
        float4 cur = *o;
        cur.x = fmaf(acc.x, u4.x, cur.x); cur.y = fmaf(acc.y, u4.y, cur.y); cur.z = fmaf(acc.z, u4.z, cur.z); cur.w = fmaf(acc.w, u4.w, cur.w);
        *o = cur;
    }
}

// demb[v,:] += sum over the positions (t,b) with token v of dXin0[t,b,0:E] (* the dropout mask).
// Round 4: the positions are bucketed by token ONCE per step on the device and only the rows that occur are reduced -- the old
// kernel (grid V x 32, every block scanning its share of the positions for `tok[i] == v`: O(V n) compares) took 1.22 ms at V = 5000.
//   (1) emb_hist_kernel   one workgroup: token histogram (integer LDS atomics: counts do not depend on the order) + exclusive scan
//   (2) emb_place_kernel  position i goes to slot start[v] + #{j < i : tok[j] == v}: a STABLE counting sort, each thread counts its
//                         predecessors in an LDS copy of the token list -- the order inside a bucket is by position, always
//   (3) emb_reduce_kernel one workgroup per vocabulary row that occurs: sixteen 64-lane groups take its positions four at a time,
//                         their partial sums meet in LDS in fixed order -> deterministic, no atomics on floats
// Tokens outside [0, V) are ignored, as before.  Serves V <= EMB_MAX_V and n <= EMB_MAX_N (the token list as 16-bit values in LDS).
constexpr int EMB_MAX_V = 16384, EMB_MAX_N = 49152;
__global__ __launch_bounds__(1024) void emb_hist_kernel(const int* __restrict__ tok, int n, int V, int* __restrict__ start) {
    extern __shared__ int cnt[];                  // [V] counts, then [1024] scan partials behind them
    int* part = cnt + V;
    const int tid = threadIdx.x;
    for (int v = tid; v < V; v += 1024) cnt[v] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) { const int v = tok[i]; if (v >= 0 && v < V) atomicAdd(&cnt[v], 1); }
    __syncthreads();
    const int per = (V + 1023) / 1024, v0 = tid * per, v1 = min(V, v0 + per);
    int s_ = 0;
    for (int v = v0; v < v1; ++v) s_ += cnt[v];
    part[tid] = s_;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {    // inclusive scan of the partials
        const int x = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int v = v0; v < v1; ++v) { start[v] = run; run += cnt[v]; }
    if (tid == 1023) start[V] = part[1023];
}

__global__ __launch_bounds__(256) void emb_place_kernel(const int* __restrict__ tok, int n, int V, const int* __restrict__ start,
                                                        int* __restrict__ pos) {
    extern __shared__ __attribute__((aligned(16))) unsigned short tk[];       // tokens [0, end of this block) (0xFFFF = ignored)
    const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x, iend = min(n, i0 + 256);
    for (int j = threadIdx.x; j < ((iend + 7) & ~7); j += 256) {
        const int v = j < iend ? tok[j] : -1;
        tk[j] = (v >= 0 && v < V) ? (unsigned short)v : (unsigned short)0xFFFF;
    }
    __syncthreads();
    if (i >= n) return;
    const unsigned v = tk[i];
    if (v == 0xFFFFu) return;
    int rank = 0;
    const uint4* t8 = reinterpret_cast<const uint4*>(tk);
    const unsigned vv = v | (v << 16);
#pragma unroll 8
    for (int j8 = 0; j8 < i0 / 8; ++j8) {          // whole groups of 8 in front of this block: the same address for every thread
        const uint4 q = t8[j8];
        const unsigned w[4] = {q.x ^ vv, q.y ^ vv, q.z ^ vv, q.w ^ vv};
#pragma unroll
        for (int e = 0; e < 4; ++e) rank += ((w[e] & 0xFFFFu) == 0) + ((w[e] >> 16) == 0);
    }
    for (int j = i0; j < i; ++j) rank += tk[j] == v;
    pos[start[v] + rank] = i;
}

__global__ __launch_bounds__(1024) void emb_reduce_kernel(const int* __restrict__ start, const int* __restrict__ pos,
                                                          const float* __restrict__ dXin0, int ld, int E, const float* __restrict__ mask,
                                                          float* __restrict__ demb) {
    // sixteen 64-lane groups; group g takes the bucket's positions 4 g .. 4 g + 3, then 64 further on: four independent row loads in
    // flight per lane (a bucket of a frequent token is hundreds of positions long; HBM latency, not bytes, is what a serial walk pays)
    __shared__ float red[16][256];
    const int v = blockIdx.x, s0 = start[v], s1 = start[v + 1];
    if (s0 == s1) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const bool small = s1 - s0 <= 4;                 // (block-uniform) one group's work: no cross-group reduction
    if (small && grp) return;
    for (int e0 = 0; e0 < E; e0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = s0 + grp * 4; k < s1; k += 64) {
            size_t i[4];
            float x[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) i[j] = (size_t)pos[min(k + j, s1 - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = e0 + lane + 64 * q;
                    x[j][q] = (e < E && k + j < s1) ? dXin0[i[j] * ld + e] * (mask ? mask[i[j] * E + e] : 1.f) : 0.f;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += x[j][q];
        }
        if (small) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int e = e0 + lane + 64 * q; if (e < E) demb[(size_t)v * E + e] += acc[q]; }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[grp][lane + 64 * q] = acc[q];
        __syncthreads();
        if (threadIdx.x < 256) {
            const int e = e0 + threadIdx.x;
            float s_ = 0.f;
#pragma unroll
            for (int g16 = 0; g16 < 16; ++g16) s_ += red[g16][threadIdx.x];
            if (e < E) demb[(size_t)v * E + e] += s_;
        }
        __syncthreads();
    }
}

// the round-1 form (any V, any n): per (vocab row, row chunk) partials, then las_colsum over the chunk axis
constexpr int EMB_CHUNKS = 32;
__global__ __launch_bounds__(256) void emb_grad_kernel(const int* tok, const float* dXin0, int n, int ld, int E, int V,
                                                       const float* mask, float* part) {
    const int v = blockIdx.x, ch = blockIdx.y;
    const int per = (n + EMB_CHUNKS - 1) / EMB_CHUNKS;
    const int i0 = ch * per, i1 = min(n, i0 + per);
    for (int e = threadIdx.x; e < E; e += 256) {
        float acc = 0.f;
        for (int i = i0; i < i1; ++i)
            if (tok[i] == v) acc += dXin0[(size_t)i * ld + e] * (mask ? mask[(size_t)i * E + e] : 1.f);
        part[((size_t)ch * V + v) * E + e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
#include "speller_wide.h"

struct BwdWs {
    size_t packF, packB, xbf, dgbf, granX, granF, granG, granB, xccs, wsbf, wsbf2, keysbf, encbf, encbf2, dE, embp, dHl, dH, dC, dXin0, Q, dQ, duRows, dAext, tmp, dlocw, dlocb, dlocwP, dlocbP, dWf, dV, fcS, dfcS, wide, gemm, total;
};
static BwdWs bwd_layout(int B, int Tp, int Hd, int A, int D, int NL, int E, int V, int U, int G, int Kc, int C) {
    BwdWs w; size_t o = 0;
    const size_t f = sizeof(float);
    const size_t I0D = (size_t)E + Hd + D;
    w.packF = o;  o += align256(las_skinny_pack_bytes((int)I0D, G * D));      // W0 fragments (step product)
    w.packB = o;  o += align256(las_skinny_pack_bytes(G * D, (int)I0D));      // W0^T fragments (step gradient)
    w.xbf = o;    o += align256((size_t)B * I0D * 2);
    w.dgbf = o;   o += align256((size_t)B * G * D * 2);
    w.granX = o;  o += align256((size_t)B * (I0D / 4 + 1) * 16);              // loop kernels: cell input rows (4 bf16 per granule)
    w.granF = o;  o += align256((size_t)B * (G * D / 2) * 16);                //               pre-activation gates (2 fp32 per granule)
    w.granG = o;  o += align256((size_t)B * (G * D / 4) * 16);                //               gate gradients (4 bf16 per granule)
    w.granB = o;  o += align256((size_t)B * ((Hd + D) / 2) * 16);             //               dXin0 chain columns (2 fp32 per granule)
    w.xccs = o;   o += align256(256 * 8);                                         //               placement handshake slots
    w.wsbf = o;   o += align256((size_t)D * NL * A * 2);
    w.wsbf2 = o;  o += align256((size_t)(D * NL + 1) * A * 2);
    w.keysbf = o; o += align256((size_t)B * Tp * A * 2);
    w.encbf = o;  o += align256((size_t)B * Tp * Hd * 2);
    w.encbf2 = o; o += align256((size_t)B * (Tp + 1) * Hd * 2);
    w.dE = o;     o += align256((size_t)U * B * Tp * f);
    w.embp = o;   o += align256((V <= EMB_MAX_V && (size_t)U * B <= EMB_MAX_N) ? ((size_t)V + 1 + (size_t)U * B) * sizeof(int)      // start[V + 1] + pos[n]
                                                                                   : (size_t)EMB_CHUNKS * V * E * f);
    w.dHl = o;    o += align256((size_t)U * B * D * f);
    w.dH = o;     o += align256((size_t)NL * B * D * f);
    w.dC = o;     o += align256((size_t)NL * B * D * f);
    w.dXin0 = o;  o += align256((size_t)U * B * I0D * f);
    w.Q = o;      o += align256((size_t)U * B * A * f);
    w.dQ = o;     o += align256((size_t)U * B * A * f);
    w.duRows = o; o += align256((size_t)B * A * f);
    w.dAext = o;  o += align256((size_t)B * Tp * f);
    w.tmp = o;    o += align256((size_t)NL * B * 2 * D * f);
    w.dlocw = o;  o += align256((size_t)B * (Kc > 0 ? Kc : 1) * (C > 0 ? C : 1) * f);
    w.dlocb = o;  o += align256((size_t)B * (C > 0 ? C : 1) * f);
    w.dWf = o;    o += align256((size_t)B * RNG * (C > 0 ? C : 1) * A * f);   // one slice per half-wave group
    w.dV = o;     o += align256(C > 0 ? (size_t)B * Tp * A * f : 0);
    w.fcS = o;    o += align256(C > 0 ? (size_t)U * B * Tp * C * f : 0);
    w.dfcS = o;   o += align256(C > 0 ? (size_t)U * B * Tp * C * f : 0);
    w.dlocwP = o; o += align256(C > 0 ? (size_t)B * DLW_SPLIT * Kc * C * f : 0);      // dlocw_mfma_kernel: one partial per (utterance, step slice); written whole
    w.dlocbP = o; o += align256(C > 0 ? (size_t)B * DLW_SPLIT * C * f : 0);
    w.wide = o;   o += align256(wide_layout(B, Tp, A, D, NL, G, C).total);             // speller_wide.h: packs, operand rows, per-step scratch
    w.gemm = o;
    size_t big = (size_t)I0D * G * D;                 // largest split-K target (dcellW[0])
    if ((size_t)D * V > big) big = (size_t)D * V;
    if ((size_t)8 * V * E > big) big = (size_t)8 * V * E;         // las_colsum scratch for the embedding gradient
    o += align256(big * 8 * f);
    w.total = o;
    return w;
}

static size_t act_save_f_offset(int U, int B, int Tp, int A) { return align256((size_t)LAS_ACT_HDR * 4 + (size_t)U * B * Tp * A * 2); }
extern "C" size_t las_speller_act_save_bytes(int U, int B, int Tp, int A, int C) {
    return act_save_f_offset(U, B, Tp, A) + (size_t)U * B * Tp * (C > 0 ? C : 0) * sizeof(float);    // header, activations (fp16), conv outputs f (fp32, location-aware)
}
extern "C" size_t las_speller_workspace_bytes(int B, int Tp, int Hd, int A, int D, int NL, int E, int V, int U, int cell) {
    const int G = cell == LAS_CELL_LSTM ? 4 : 1;
    return bwd_layout(B, Tp, Hd, A, D, NL, E, V, U, G, 256, 16).total;
}

static int fill_dev(const las_speller_fwd_args* f, DecDev& d) {
    LAS_ARG(f, "speller: null args");
    LAS_ARG(f->B > 0 && f->Tp > 0 && f->Hd > 0 && f->A > 0 && f->D > 0 && f->E > 0 && f->V > 0 && f->U > 0,
            "speller: non-positive dimension");
    LAS_ARG(f->NL >= 1 && f->NL <= LAS_MAX_NL, "speller: NL=%d unsupported (1..%d)", f->NL, LAS_MAX_NL);
    LAS_ARG(f->D % 4 == 0 && f->Hd % 4 == 0, "speller: D and Hd must be multiples of 4");
    LAS_ARG(f->A % 4 == 0 && f->A <= 256, "speller: attention size must be a multiple of 4 and <= 256 (got %d)", f->A);
    LAS_ARG(f->cell == LAS_CELL_RNN || f->cell == LAS_CELL_LSTM, "speller: bad cell");
    LAS_ARG(f->mode == LAS_ATT_ADD || f->mode == LAS_ATT_LOC, "speller: bad attention mode");
    LAS_ARG(f->mode == LAS_ATT_ADD || (f->loc_w && f->loc_b && f->Wf && f->Kc > 0 && f->Kc <= 256 && f->C > 0 && f->C <= 16),
            "speller: location-aware attention needs loc_w/loc_b/Wf, Kc<=256, C<=16");
    LAS_ARG(f->enc && f->keys && f->enc_len && f->Ws && f->u && f->emb && f->Wv && f->bv && f->cellW && f->cellb,
            "speller: null parameter pointer");
    LAS_ARG(f->tokens_in && f->logits && f->alphas && f->hs && f->gates && f->xin0, "speller: null buffer pointer");
    LAS_ARG(f->cell == LAS_CELL_RNN || f->cs, "speller: lstm needs cs");
    LAS_ARG(!f->step_logits || f->tokens_out, "speller: step_logits needs tokens_out");
    LAS_ARG((((uintptr_t)f->keys | (uintptr_t)f->u | (uintptr_t)f->Ws) & 15) == 0, "speller: keys/u/Ws must be 16-byte aligned");
    d.B = f->B; d.Tp = f->Tp; d.Hd = f->Hd; d.A = f->A; d.D = f->D; d.NL = f->NL; d.E = f->E; d.V = f->V; d.U = f->U;
    d.mode = f->mode; d.Kc = f->mode == LAS_ATT_LOC ? f->Kc : 0; d.C = f->mode == LAS_ATT_LOC ? f->C : 0;
    d.step_logits = f->step_logits; d.flags = f->flags; d.fb = f->forget_bias; d.seed = f->seed;
    d.row_group = (f->row_group > 0 && f->B % f->row_group == 0) ? f->row_group : 0;
    d.shared_ops = ((f->flags & LAS_SPELLER_SHARED_OPERANDS) && d.row_group > 0) ? 1 : 0;
    LAS_ARG(!(f->flags & LAS_SPELLER_SHARED_OPERANDS) || d.row_group > 0, "speller: LAS_SPELLER_SHARED_OPERANDS needs row_group > 0 with B %% row_group == 0");
    d.enc = f->enc; d.keys = f->keys; d.enc_len = f->enc_len; d.Ws = f->Ws; d.u = f->u; d.emb = f->emb;
    d.Wv = f->Wv; d.bv = f->bv; d.loc_w = f->loc_w; d.loc_b = f->loc_b; d.Wf = f->Wf;
    d.tok_in = f->tokens_in; d.tok_out = f->tokens_out; d.align0 = f->align0; d.emb_mask = f->emb_mask; d.emb_noise = f->emb_noise; d.logits = f->logits; d.alphas = f->alphas;
    d.hs = f->hs; d.cs = f->cs; d.gates = f->gates; d.xin0 = f->xin0; d.actS = nullptr;
    d.lp = LoopProd{};
    d.lp.status = f->status;
    d.lp.budget = 1 << (((f->flags >> 8) & 31) ? ((f->flags >> 8) & 31) : 21);
    d.xbf = nullptr; d.dgbf = nullptr; d.Wsbf = d.keysbf = d.encbf = d.Wsbf2 = d.encbf2 = nullptr; d.dE = nullptr;
    d.dHl = nullptr; d.dH = d.dC = d.dXin0 = d.Q = d.dQ = d.duRows = d.dAext = d.dKeys = nullptr;
    d.dlocwRows = d.dlocbRows = d.dWfRows = nullptr; d.dVbuf = d.fcSave = d.dfcSave = nullptr;
    for (int l = 0; l < LAS_MAX_NL; ++l) { d.rec[l] = nullptr; d.recLd[l] = 0; d.recOff[l] = 0; }
    return 0;
}

#define GEMM_OK(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

static int g_last_variant[2] = {0, 0};     // (process-wide: a backward pass runs on the autograd engine's thread)
extern "C" int las_speller_last_variant(int which) { return g_last_variant[which ? 1 : 0]; }

// speed mode, additive attention: the row kernels read bf16 copies of Ws / keys / encoder rows (made once per call)
static bool bf_rows_ok(const DecDev& d) {
    return !(d.flags & LAS_SPELLER_NO_BF_ROWS) && d.mode == LAS_ATT_ADD && (d.A % 8) == 0 && (d.Hd % 8) == 0 && d.A <= 256;
}
// ... and, for the common single-layer geometry, the fully prefetching variants
static bool pf_geom_ok(const DecDev& d) {
    return d.NL == 1 && d.D <= 512 && d.A <= 128 && d.Hd <= 512 && d.Tp <= 224 && d.E <= 1024 && (d.E % 2) == 0 && (d.D % 2) == 0 &&
           (d.A % 8) == 0 && (d.Hd % 8) == 0;
}
static bool pf_rows_ok(const DecDev& d) { return !(d.flags & LAS_SPELLER_NO_PF_ROWS) && bf_rows_ok(d) && pf_geom_ok(d); }
// ... and the whole loop in one launch: 8 groups of pn product + R row workgroups, all co-resident (one per compute unit),
// tpw column tiles per product workgroup and kw k-steps per product wave as instantiated in the kernels
constexpr int LOOP_TPW_F = 5, LOOP_KW_F = 3, LOOP_TPW_B = 3, LOOP_KW_B = 4;
static bool loop_geom_ok(const DecDev& d, int ncols, int K, int tpw, int kw) {
    const int R = cdiv(d.B, 8), pn = las_device_cus() / 8 - R;
    // (U >= 4: a launch of the persistent grid costs ~100 us before its first step -- 256 workgroups, placement handshake -- which
    //  30 us saved per step only repays from the fourth step on; beam search calls the step with U = 1: 141 vs 43 us, r3 decode trace)
    return d.U >= 4 && (d.E % 4) == 0 && (d.D % 4) == 0 && (d.Hd % 4) == 0 && ((d.E + d.Hd + d.D) % 8) == 0 && (K % 8) == 0 && R <= 16 &&
           pn >= 1 && pn + R <= 32 && pn * tpw >= cdiv(ncols, 16) && 16 * kw >= cdiv(K, 32);
}
static bool loop_ok(const DecDev& d, int ncols, int K, int tpw, int kw) {
    return !(d.flags & LAS_SPELLER_NO_FUSED_STEP) && pf_rows_ok(d) && loop_geom_ok(d, ncols, K, tpw, kw);
}
// Location-aware attention (round 3): served by the SAME loop kernels (pf_fwd_row / pf_bwd_row with LOC = true: conv1d over the
// previous alignment from LDS, the f . Wf term in the energies, d f / d alpha_{t-1} in the gradient loop; keys / Wf / filter gradients
// contracted over the steps afterwards) when BOTH loops are eligible, so that forward and gradient stay in one arithmetic family
// (bf16 row operands); otherwise the per-step fp32-operand row kernels dec_step_{fwd,bwd}_kernel<.,.,true>.
static bool loc_loop_ok(const DecDev& d, int G) {
    const int GD = G * d.D, I0D = d.E + d.Hd + d.D;
    return d.mode == LAS_ATT_LOC && !(d.flags & (LAS_SPELLER_NO_PF_ROWS | LAS_SPELLER_NO_BF_ROWS | LAS_SPELLER_NO_FUSED_STEP)) &&
           pf_geom_ok(d) && (d.A % 32) == 0 && d.C >= 1 && d.C <= 10 && d.Kc * d.C <= 4096 && cdiv(d.Tp, 8) <= RNG &&
           cdiv(d.Tp, 16) <= RNW && bf_lds_bytes(d) <= 128 * 1024 &&     // the MFMA convs: one wave per 16-frame tile; the row state in LDS
           loop_geom_ok(d, GD, I0D, LOOP_TPW_F, LOOP_KW_F) && loop_geom_ok(d, d.Hd + d.D, GD, LOOP_TPW_B, LOOP_KW_B);
}
static void loop_prod_dims(LoopProd& p, int B, int ncols, int K) {
    p.KS = cdiv(K, 32); p.K = K; p.N = ncols; p.nct = cdiv(ncols, 16); p.M = B; p.R = cdiv(B, 8);
    p.pn = las_device_cus() / 8 - p.R;
}
static constexpr size_t LOOP_LDS_MAX = 159 * 1024;                 // a loop workgroup has its CU to itself (160 KB less the kernels' static LDS)
template <class K> static int loop_lds_attr(K kernel) {   // the product workgroups' partial tiles need > 64 KB of dynamic LDS
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LOOP_LDS_MAX);
}
#define LAS_LOOP_LAUNCH1(KERNEL, CELL, NE, LOC, grid, lds, st, d)                                          \
    do {                                                                                                   \
        static int attr__ = loop_lds_attr(KERNEL<CELL, NE, LOC>);                                          \
        if (attr__ != 0) { las_set_error("hipFuncSetAttribute(speller loop) failed: %d", attr__); return attr__; } \
        hipLaunchKernelGGL((KERNEL<CELL, NE, LOC>), grid, dim3(RNT), lds, st, d);                          \
    } while (0)
#define LAS_LOOP_LAUNCH2(KERNEL, CELL, Tp, LOC, grid, lds, st, d)                                          \
    do {                                                                                                   \
        if ((Tp) <= 128)      LAS_LOOP_LAUNCH1(KERNEL, CELL, 8, LOC, grid, lds, st, d);                    \
        else if ((Tp) <= 160) LAS_LOOP_LAUNCH1(KERNEL, CELL, 10, LOC, grid, lds, st, d);                   \
        else if ((Tp) <= 192) LAS_LOOP_LAUNCH1(KERNEL, CELL, 12, LOC, grid, lds, st, d);                   \
        else                  LAS_LOOP_LAUNCH1(KERNEL, CELL, 14, LOC, grid, lds, st, d);                   \
    } while (0)
#define LAS_LOOP_LAUNCH(KERNEL, CELL, Tp, loc, grid, lds, st, d)                                           \
    do { if (loc) LAS_LOOP_LAUNCH2(KERNEL, CELL, Tp, true, grid, lds, st, d); else LAS_LOOP_LAUNCH2(KERNEL, CELL, Tp, false, grid, lds, st, d); } while (0)
static int make_bf_copies(DecDev& d, char* base, const BwdWs& w, hipStream_t st) {
    unsigned short* wsb = (unsigned short*)(base + w.wsbf);
    unsigned short* kb = (unsigned short*)(base + w.keysbf);
    unsigned short* eb = (unsigned short*)(base + w.encbf);
    const int nblk = d.shared_ops ? d.B / d.row_group : d.B;       // operand blocks: one per row, or one per group of rows
    const size_t nW = (size_t)d.D * d.NL * d.A, nK = (size_t)nblk * d.Tp * d.A, nE = (size_t)nblk * d.Tp * d.Hd;
    if (d.flags & LAS_SPELLER_REUSE_PREP) {   // the caller vouches that an earlier call left the copies of the SAME tensors here
        d.Wsbf = wsb; d.keysbf = kb; d.encbf = eb;
        d.Wsbf2 = (unsigned short*)(base + w.wsbf2); d.encbf2 = (unsigned short*)(base + w.encbf2);
        return 0;
    }
    // ... and row-pair interleaved copies for the packed dot products of the prefetching forward kernel: all five in ONE launch
    // (they sit on the chain between the Listener's last sweep and the decode loop; five launches were 25 us)
    unsigned short* wsb2 = (unsigned short*)(base + w.wsbf2);
    unsigned short* eb2 = (unsigned short*)(base + w.encbf2);
    const int S = d.D * d.NL;
    BfCopyJobs jobs;
    jobs.j[0] = {d.Ws, wsb, nW, 0, 0, 1};
    jobs.j[1] = {d.keys, kb, nK, 0, 0, 1};
    jobs.j[2] = {d.enc, eb, nE, 0, 0, 1};
    jobs.j[3] = {d.Ws, wsb2, 0, S, d.A, 1};
    jobs.j[4] = {d.enc, eb2, 0, d.Tp, d.Hd, nblk};
    hipLaunchKernelGGL(bf_copies_kernel, dim3(512, 5), dim3(256), 0, st, jobs);
    LAS_LAUNCHED();
    d.Wsbf = wsb; d.keysbf = kb; d.encbf = eb; d.Wsbf2 = wsb2; d.encbf2 = eb2;
    return 0;
}

#include "speller_wide_host.h"

template <int CELL, bool FAST>
static int speller_fwd_impl(const las_speller_fwd_args* f, DecDev d, hipStream_t st) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int B = d.B, D = d.D, NL = d.NL, U = d.U, E = d.E, Hd = d.Hd, V = d.V;
    const int GD = G * D, I0D = E + Hd + D;
    const size_t lds = row_lds_bytes(d, false);
    LAS_ARG(lds <= 150 * 1024, "speller: row state does not fit LDS (%zu bytes)", lds);
    if (lds > 64 * 1024) {   // location-aware attention with the reference's K = 201, C = 10: the staged filter + Wf push the carve past 64 KB
        static int attr__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_step_fwd_kernel<CELL, FAST, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        LAS_ARG(attr__ == 0, "hipFuncSetAttribute(dec_step_fwd_kernel) failed: %d", attr__);
        LAS_ARG(d.mode == LAS_ATT_LOC, "speller: row state does not fit LDS (%zu bytes)", lds);
    }
    for (int l = 0; l < NL && !f->keep_state0; ++l) {
        LAS_HIP(hipMemsetAsync(d.hs + (size_t)l * (U + 1) * B * D, 0, (size_t)B * D * sizeof(float), st));
        if (CELL == LAS_CELL_LSTM) LAS_HIP(hipMemsetAsync(d.cs + (size_t)l * (U + 1) * B * D, 0, (size_t)B * D * sizeof(float), st));
    }
    // per-step cell product: with bf16 arithmetic the weights are packed once into MFMA fragments and every
    // step runs the skinny-M kernel (M = batch rows); fp32 mode keeps the generic exact path
    const BwdWs wl_ = bwd_layout(B, d.Tp, Hd, d.A, D, NL, E, V, U, G, d.Kc, d.C);
    const bool skinny = FAST && f->ws && f->ws_bytes >= wl_.embp && (I0D % 8) == 0 && las_skinny_ok(B, I0D, GD, I0D, d.xin0);
    void* packF = skinny ? (char*)f->ws + wl_.packF : nullptr;
    if (skinny) d.xbf = (unsigned short*)((char*)f->ws + wl_.xbf);
    bool bfrows = skinny && bf_rows_ok(d);
    bool pf = bfrows && pf_rows_ok(d);
    bool locloop = skinny && loc_loop_ok(d, G);
    bool loop = locloop || (pf && loop_ok(d, GD, I0D, LOOP_TPW_F, LOOP_KW_F));
    // round 6: geometries outside the loop kernels' (the reference's run.sh recipe: two 1024-unit layers, T' = 319, location-aware) take
    // the wide path (speller_wide.h) instead of the per-utterance fp32-operand rows
    const bool wide = wide_selected<FAST>(d, f->ws && f->ws_bytes >= wl_.gemm, skinny, loop, pf);
    if (wide) bfrows = pf = locloop = loop = false;
    const size_t lds_bf = bf_lds_bytes(d);
    if (bfrows || locloop || (wide && FAST)) {
        if (!wide) LAS_ARG(lds_bf <= (locloop ? 128 : 64) * 1024, "speller: row state does not fit LDS (%zu bytes)", lds_bf);   // (loop launches: 96 KB attribute)
        GEMM_OK(make_bf_copies(d, (char*)f->ws, wl_, st));
    }
    if (skinny && !(d.flags & LAS_SPELLER_REUSE_PREP)) GEMM_OK(las_skinny_pack(f->cellW[0], GD, I0D, GD, 0, packF, st));
    if (f->act_save) {   // the prefetching rows keep their attention activations for the gradient rows; any other kernel family leaves the header cleared
        if (wide && d.mode == LAS_ATT_LOC && U > 1) {   // the wide path keeps the conv outputs f only (its own header word)
            LAS_HIP(hipMemsetAsync(f->act_save, 0, 4, st));
            d.actS = (unsigned*)f->act_save;
            d.fcSave = (float*)((char*)f->act_save + act_save_f_offset(U, B, d.Tp, d.A));
        } else if (FAST && (loop || pf) && U > 1) {
            d.actS = (unsigned*)f->act_save;
            if (d.mode == LAS_ATT_LOC) d.fcSave = (float*)((char*)f->act_save + act_save_f_offset(U, B, d.Tp, d.A));
        } else LAS_HIP(hipMemsetAsync(f->act_save, 0, 4, st));
    }
    g_last_variant[0] = (wide ? LAS_SPELLER_RAN_WIDE : loop ? LAS_SPELLER_RAN_LOOP : pf ? LAS_SPELLER_RAN_PF_ROWS : bfrows ? LAS_SPELLER_RAN_BF_ROWS : LAS_SPELLER_RAN_F32_ROWS) |
                        (skinny ? LAS_SPELLER_RAN_SKINNY : 0) | (d.mode == LAS_ATT_LOC ? LAS_SPELLER_RAN_LOC : 0) |
                        (wide && FAST && NL > 1 ? LAS_SPELLER_RAN_UPPER_SKINNY : 0);
    LAS_ARG(!d.shared_ops || ((d.flags & LAS_SPELLER_NO_LOGITS) && pf && !loop && !wide && U == 1),
            "speller: LAS_SPELLER_SHARED_OPERANDS is served by the search step's prefetching row kernels only (LAS_SPELLER_NO_LOGITS, U = 1)");
    if (d.flags & LAS_SPELLER_NO_LOGITS)
        LAS_ARG(CELL == LAS_CELL_LSTM && NL == 1 && U == 1 && skinny && pf && !loop && (D % 32) == 0 && (I0D % 32) == 0 && d.step_logits,
                "speller: LAS_SPELLER_NO_LOGITS needs U = 1, one LSTM layer, speed mode with the prefetching row kernels, D and E + Hd + D multiples of 32");
    if (loop) {   // the whole loop in one launch
        const size_t lds_pr = (size_t)RNW * LOOP_TPW_F * 1024;       // the product workgroups' partial tiles (80 KB)
        const size_t lds_rw = lds_bf + enc_res_bytes(d, locloop, loop_ne(d.Tp));   // row state + resident encoder slabs
        const size_t lds_lp = lds_rw < lds_pr ? lds_pr : lds_rw;
        LAS_ARG(lds_lp <= LOOP_LDS_MAX, "speller: the loop's row state does not fit LDS (%zu bytes)", lds_lp);
        loop_prod_dims(d.lp, B, GD, I0D);
        d.lp.Bp = reinterpret_cast<const u16x8_t*>(packF); d.lp.bias = f->cellb[0]; d.lp.C = nullptr; d.lp.c_step = 0; d.lp.ldc = 0;
        d.lp.gA = (unsigned long long*)((char*)f->ws + wl_.granX); d.lp.gA_row = I0D / 4;
        d.lp.gC = (unsigned long long*)((char*)f->ws + wl_.granF); d.lp.gC_row = GD / 2;
        d.lp.xcc = (unsigned long long*)((char*)f->ws + wl_.xccs);
        // tags of an earlier call must not match: granX, granF, (granG, granB,) xccs are one contiguous stretch -> one fill
        LAS_HIP(hipMemsetAsync((char*)f->ws + wl_.granX, 0, wl_.xccs + 256 * 8 - wl_.granX, st));
        LAS_LOOP_LAUNCH(dec_loop_fwd_kernel, CELL, d.Tp, locloop, dim3(8 * (d.lp.pn + d.lp.R)), lds_lp, st, d);
        LAS_LAUNCHED();
    }
    if (wide) GEMM_OK((wide_fwd_steps<CELL, FAST>(f, d, wl_, packF, st)));
    for (int t = 0; t <= U && !loop && !wide; ++t) {
        // (t == U only finishes the last cell [+ logits]: the prefetching kernel would issue a whole step's bulk loads first --
        //  the generic bf16 row kernel loads on demand and returns after the cell; half of a beam-search step's Speller time)
        if (pf && t == 0 && f->companion_rows && (d.flags & LAS_SPELLER_NO_LOGITS)) {
            // beam-search step: the LM's first layer as extra workgroups of the attention-row launch
            const LstmCellLaunch& lm = *f->companion_rows;
            if (int rc = las_lstm_cell_check(lm)) return rc;
            LAS_ARG(!lm.fast && !lm.x_bf16, "las_speller_fwd: companion_rows must be an exact (fast = 0) cell with fp32 rows");
            const int nlm = (lm.H / 16) * cdiv(lm.M, 32);
            const size_t ldsc = lds_bf > (size_t)PF_LM_LDS ? lds_bf : (size_t)PF_LM_LDS;
            if (d.Tp <= 128)      hipLaunchKernelGGL((dec_step_fwd_pf_lm_kernel<CELL, 8>), dim3(nlm + xcd_local_grid(B, d.row_group)), dim3(RNT), ldsc, st, d, t, lm, nlm);
            else if (d.Tp <= 160) hipLaunchKernelGGL((dec_step_fwd_pf_lm_kernel<CELL, 10>), dim3(nlm + xcd_local_grid(B, d.row_group)), dim3(RNT), ldsc, st, d, t, lm, nlm);
            else if (d.Tp <= 192) hipLaunchKernelGGL((dec_step_fwd_pf_lm_kernel<CELL, 12>), dim3(nlm + xcd_local_grid(B, d.row_group)), dim3(RNT), ldsc, st, d, t, lm, nlm);
            else                  hipLaunchKernelGGL((dec_step_fwd_pf_lm_kernel<CELL, 14>), dim3(nlm + xcd_local_grid(B, d.row_group)), dim3(RNT), ldsc, st, d, t, lm, nlm);
        }
        else if (pf && t == 0 && U == 1 && f->keep_state0 && (d.flags & LAS_SPELLER_ROWS_SHARE4) && (d.flags & LAS_SPELLER_NO_LOGITS)) {
            // beam-search step over many rows: four hypotheses of an utterance per workgroup, shared operands read once
            const size_t l4 = beam_rows4_lds(d);
            static int attr4 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_beam_rows4_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) |
                               (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_beam_rows4_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) |
                               (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_beam_rows4_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) |
                               (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_beam_rows4_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            LAS_ARG(attr4 == 0 && l4 <= 128 * 1024, "speller: hipFuncSetAttribute(dec_beam_rows4_kernel) failed (%d) or the rows' state does not fit LDS (%zu)", attr4, l4);
            LAS_ARG((B % BR4) == 0 && d.tok_in, "speller: LAS_SPELLER_ROWS_SHARE4 needs a row count that is a multiple of 4");
            if (d.Tp <= 128)      hipLaunchKernelGGL((dec_beam_rows4_kernel<8>), dim3(xcd_local_grid(B / BR4, d.row_group / BR4)), dim3(RNT), l4, st, d);
            else if (d.Tp <= 160) hipLaunchKernelGGL((dec_beam_rows4_kernel<10>), dim3(xcd_local_grid(B / BR4, d.row_group / BR4)), dim3(RNT), l4, st, d);
            else if (d.Tp <= 192) hipLaunchKernelGGL((dec_beam_rows4_kernel<12>), dim3(xcd_local_grid(B / BR4, d.row_group / BR4)), dim3(RNT), l4, st, d);
            else                  hipLaunchKernelGGL((dec_beam_rows4_kernel<14>), dim3(xcd_local_grid(B / BR4, d.row_group / BR4)), dim3(RNT), l4, st, d);
        }
        else if (pf && t == U)          hipLaunchKernelGGL((dec_step_fwd_bf_kernel<CELL, 1>), dim3(B), dim3(RNT), lds_bf, st, d, t);
        else if (pf && d.Tp <= 128)     hipLaunchKernelGGL((dec_step_fwd_pf_kernel<CELL, 8>), dim3(xcd_local_grid(B, d.row_group)), dim3(RNT), lds_bf, st, d, t);
        else if (pf && d.Tp <= 160)     hipLaunchKernelGGL((dec_step_fwd_pf_kernel<CELL, 10>), dim3(xcd_local_grid(B, d.row_group)), dim3(RNT), lds_bf, st, d, t);
        else if (pf && d.Tp <= 192)     hipLaunchKernelGGL((dec_step_fwd_pf_kernel<CELL, 12>), dim3(xcd_local_grid(B, d.row_group)), dim3(RNT), lds_bf, st, d, t);
        else if (pf)                    hipLaunchKernelGGL((dec_step_fwd_pf_kernel<CELL, 14>), dim3(xcd_local_grid(B, d.row_group)), dim3(RNT), lds_bf, st, d, t);
        else if (bfrows && d.A <= 128)  hipLaunchKernelGGL((dec_step_fwd_bf_kernel<CELL, 1>), dim3(B), dim3(RNT), lds_bf, st, d, t);
        else if (bfrows)                hipLaunchKernelGGL((dec_step_fwd_bf_kernel<CELL, 2>), dim3(B), dim3(RNT), lds_bf, st, d, t);
        else if (d.mode == LAS_ATT_LOC) hipLaunchKernelGGL((dec_step_fwd_kernel<CELL, FAST, true>), dim3(B), dim3(RNT), lds, st, d, t);
        else                            hipLaunchKernelGGL((dec_step_fwd_kernel<CELL, FAST, false>), dim3(B), dim3(RNT), lds, st, d, t);
        LAS_LAUNCHED();
        if (t == U) break;
        if (d.flags & LAS_SPELLER_NO_LOGITS) {   // beam-search step: the cell (product + gate math) in one launch, no projection
            LstmCellLaunch c;
            c.x = d.xbf; c.x_bf16 = 1; c.ldx = I0D; c.I = I0D; c.ids = nullptr; c.id_shift = 0; c.xrows = nullptr; c.h = nullptr; c.ldh = 0;
            c.Wx = packF; c.Wh = nullptr; c.bias = f->cellb[0]; c.c_prev = d.cs; c.fb = d.fb;
            c.c_out = d.cs + (size_t)B * D; c.h_out = d.hs + (size_t)B * D; c.gates_out = d.gates; c.M = B; c.H = D; c.fast = 1; c.h_bf16 = 0; c.h_out_bf16 = nullptr;
            if (int rc = f->companion ? las_lstm_cell_rows_launch2(c, *f->companion, st) : las_lstm_cell_rows_launch(c, st)) return rc;
            break;
        }
        if (skinny) {
            GEMM_OK(las_skinny_gemm_bf16(d.xbf, I0D, B, I0D, packF, GD, d.gates + ((size_t)0 * U + t) * B * GD, GD, f->cellb[0], st));
        } else {
            // (parity mode: the skinny fp32 product takes K slices when it is given scratch -- the after-loop contractions' region)
            const bool hw = f->ws && f->ws_bytes > wl_.gemm;
            GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, I0D, 1.f, d.xin0 + (size_t)t * B * I0D, I0D, 0, f->cellW[0], GD, 0, 0.f,
                             d.gates + ((size_t)0 * U + t) * B * GD, GD, 0, f->cellb[0], LAS_ACT_NONE, 1, 0, 0,
                             hw ? (char*)f->ws + wl_.gemm : nullptr, hw ? f->ws_bytes - wl_.gemm : 0, st));
        }
        for (int l = 1; l < NL; ++l) {
            hipLaunchKernelGGL((dec_pointwise_fwd_kernel<CELL, FAST>), dim3(B), dim3(256), 0, st, d, l - 1, t);
            LAS_LAUNCHED();
            float* gl = d.gates + ((size_t)l * U + t) * B * GD;
            GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, D, 1.f, d.hs + ((size_t)(l - 1) * (U + 1) + t + 1) * B * D, D, 0,
                             f->cellW[l], GD, 0, 0.f, gl, GD, 0, f->cellb[l], LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
            GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, D, 1.f, d.hs + ((size_t)l * (U + 1) + t) * B * D, D, 0,
                             f->cellW[l] + (size_t)D * GD, GD, 0, 1.f, gl, GD, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
        }
    }
    if (d.step_logits != 1) {  // vocab projection of all steps at once (dense MFMA work): las/las.py:156-158
        GEMM_OK(las_gemm(f->prec, 0, 0, U * B, V, D, 1.f, d.hs + ((size_t)(NL - 1) * (U + 1) + 1) * B * D, D, 0, d.Wv, V, 0,
                         0.f, d.logits, V, 0, d.bv, LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
    }
    return 0;
}

extern "C" int las_speller_fwd(const las_speller_fwd_args* f, void* stream) {
    DecDev d;
    if (int rc = fill_dev(f, d)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = f->prec == LAS_PREC_BF16;
    if (f->cell == LAS_CELL_LSTM)
        return fast ? speller_fwd_impl<LAS_CELL_LSTM, true>(f, d, st) : speller_fwd_impl<LAS_CELL_LSTM, false>(f, d, st);
    return fast ? speller_fwd_impl<LAS_CELL_RNN, true>(f, d, st) : speller_fwd_impl<LAS_CELL_RNN, false>(f, d, st);
}

template <int CELL, bool FAST>
static int speller_bwd_impl(const las_speller_bwd_args* bk, DecDev d, int part, hipStream_t st) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const las_speller_fwd_args* f = &bk->f;
    const int B = d.B, D = d.D, NL = d.NL, U = d.U, E = d.E, Hd = d.Hd, V = d.V, A = d.A, Tp = d.Tp;
    const int GD = G * D, I0D = E + Hd + D, TOP = NL - 1, prec = f->prec;
    const bool loc = d.mode == LAS_ATT_LOC;
    const BwdWs w = bwd_layout(B, Tp, Hd, A, D, NL, E, V, U, G, d.Kc, d.C);
    LAS_ARG(f->ws && f->ws_bytes >= w.total, "las_speller_bwd: workspace too small (%zu < %zu)", f->ws_bytes, w.total);
    char* base = (char*)f->ws;
    float* dHl = (float*)(base + w.dHl);
    d.dHl = dHl; d.dH = (float*)(base + w.dH); d.dC = (float*)(base + w.dC); d.dXin0 = (float*)(base + w.dXin0);
    d.Q = (float*)(base + w.Q); d.dQ = (float*)(base + w.dQ); d.duRows = (float*)(base + w.duRows);
    d.dAext = (float*)(base + w.dAext); d.dKeys = bk->d_keys;
    d.dlocwRows = (float*)(base + w.dlocw); d.dlocbRows = (float*)(base + w.dlocb); d.dWfRows = (float*)(base + w.dWf);
    d.dVbuf = (float*)(base + w.dV); d.fcSave = (float*)(base + w.fcS); d.dfcSave = (float*)(base + w.dfcS);
    float* tmp = (float*)(base + w.tmp);          // [NL][B][2D] input/recurrent grads of layers >= 1
    void* gws = base + w.gemm;
    const size_t gws_bytes = f->ws_bytes - w.gemm;
    const size_t lds = row_lds_bytes(d, true);
    LAS_ARG(lds <= 150 * 1024, "speller bwd: row state does not fit LDS (%zu bytes)", lds);
    if (lds > 64 * 1024) {
        static int attr__ = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_step_bwd_kernel<CELL, FAST, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        LAS_ARG(attr__ == 0, "hipFuncSetAttribute(dec_step_bwd_kernel) failed: %d", attr__);
        LAS_ARG(d.mode == LAS_ATT_LOC, "speller bwd: row state does not fit LDS (%zu bytes)", lds);
    }

    const bool skinny = FAST && (GD % 8) == 0 && las_skinny_ok(B, GD, I0D, GD, d.gates);
    void* packB = base + w.packB;
    if (skinny) d.dgbf = (unsigned short*)(base + w.dgbf);
    bool bfrows = skinny && bf_rows_ok(d);
    bool pf = bfrows && pf_rows_ok(d);
    bool locloop = skinny && loc_loop_ok(d, G);
    const bool wide = wide_selected<FAST>(d, true, skinny, locloop || (pf && loop_ok(d, Hd + D, GD, LOOP_TPW_B, LOOP_KW_B)), pf);
    if (wide) bfrows = pf = locloop = false;
    const size_t lds_bf = bf_lds_bytes(d);
    if (bfrows || locloop || wide) {
        if (!wide) LAS_ARG(lds_bf <= (locloop ? 128 : 64) * 1024, "speller bwd: row state does not fit LDS (%zu bytes)", lds_bf);
        if ((part & 1) && (FAST || !wide)) GEMM_OK(make_bf_copies(d, base, w, st));
        d.dE = (float*)(base + w.dE);
    }
    // the whole loop in one launch: the in-loop product covers the chain columns [E, I0D) only (every product workgroup then
    // feeds granules to the rows, which is what makes the single-buffered exchange safe); the embedding columns of dXin0 are
    // one tall contraction after the loop (part 2)
    const bool loop = locloop || (pf && loop_ok(d, Hd + D, GD, LOOP_TPW_B, LOOP_KW_B));
    if (part & 1)
        g_last_variant[1] = (wide ? LAS_SPELLER_RAN_WIDE : loop ? LAS_SPELLER_RAN_LOOP : pf ? LAS_SPELLER_RAN_PF_ROWS : bfrows ? LAS_SPELLER_RAN_BF_ROWS : LAS_SPELLER_RAN_F32_ROWS) |
                            (skinny ? LAS_SPELLER_RAN_SKINNY : 0) | (loc ? LAS_SPELLER_RAN_LOC : 0) | (wide && FAST && NL > 1 ? LAS_SPELLER_RAN_UPPER_SKINNY : 0);
    if ((loop || pf || (wide && loc)) && bk->f.act_save) {   // (the rows check the header: only what a forward of the same family left is used)
        d.actS = (unsigned*)bk->f.act_save;
        if (d.mode == LAS_ATT_LOC) d.fcSave = (float*)((char*)bk->f.act_save + act_save_f_offset(U, B, Tp, A));   // f of every step: kept by the forward rows, or recomputed into the same place
    }
    if (skinny && (part & 1)) {   // B[k = gate col][n = input row] = W0[n][k]
        if (loop) GEMM_OK(las_skinny_pack(f->cellW[0] + (size_t)E * GD, GD, GD, Hd + D, 1, packB, st));
        else      GEMM_OK(las_skinny_pack(f->cellW[0], GD, GD, I0D, 1, packB, st));
    }
    if (part & 1) {
    d.rec[0] = d.dXin0; d.recLd[0] = I0D; d.recOff[0] = E + Hd;   // rebased per step below
    for (int l = 1; l < NL; ++l) { d.rec[l] = tmp + (size_t)l * B * 2 * D; d.recLd[l] = 2 * D; d.recOff[l] = D; }

    LAS_HIP(hipMemsetAsync(base + w.dH, 0, w.dXin0 - w.dH, st));                      // dH, dC
    LAS_HIP(hipMemsetAsync(base + w.duRows, 0, w.dV - w.duRows, st));                 // duRows .. dWfRows (incl. tmp)
    // dlogits . Wv^T for every step, and the queries Q = S . Ws, as dense GEMMs up front
    GEMM_OK(las_gemm(prec, 0, 1, U * B, D, V, 1.f, bk->dlogits, V, 0, d.Wv, V, 0, 0.f, dHl, D, 0, nullptr, LAS_ACT_NONE, 1, 0, 0,
                     nullptr, 0, st));
    for (int l = 0; l < NL; ++l)
        GEMM_OK(las_gemm(prec, 0, 0, U * B, A, D, 1.f, d.hs + (size_t)l * (U + 1) * B * D, D, 0, d.Ws + (size_t)l * D * A, A, 0,
                         l ? 1.f : 0.f, d.Q, A, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));

    if (loop) {
        const size_t lds_pr = (size_t)RNW * LOOP_TPW_B * 1024;
        const size_t lds_rw = lds_bf + enc_res_bytes(d, locloop, loop_ne(Tp), true);     // row state + resident encoder frames
        const size_t lds_lp = lds_rw < lds_pr ? lds_pr : lds_rw;
        LAS_ARG(lds_lp <= LOOP_LDS_MAX, "speller bwd: the loop's row state does not fit LDS (%zu bytes)", lds_lp);
        loop_prod_dims(d.lp, B, Hd + D, GD);
        d.lp.Bp = reinterpret_cast<const u16x8_t*>(packB); d.lp.bias = nullptr;
        d.lp.C = d.dXin0 + E; d.lp.c_step = (long long)B * I0D; d.lp.ldc = I0D;
        d.lp.gA = (unsigned long long*)(base + w.granG); d.lp.gA_row = GD / 4;
        d.lp.gC = (unsigned long long*)(base + w.granB); d.lp.gC_row = (Hd + D) / 2;
        d.lp.xcc = (unsigned long long*)(base + w.xccs);
        LAS_HIP(hipMemsetAsync(base + w.granG, 0, w.xccs + 256 * 8 - w.granG, st));       // granG, granB, xccs: contiguous, one fill
        LAS_LOOP_LAUNCH(dec_loop_bwd_kernel, CELL, Tp, locloop, dim3(8 * (d.lp.pn + d.lp.R)), lds_lp, st, d);
        LAS_LAUNCHED();
    }
    if (wide) GEMM_OK((wide_bwd_steps<CELL, FAST>(bk, d, w, base, packB, dHl, tmp, gws, gws_bytes, st)));
    for (int t = U - 1; t >= -1 && !loop && !wide; --t) {
        DecDev ds = d;
        if (t + 1 < U) ds.rec[0] = d.dXin0 + (size_t)(t + 1) * B * I0D;
        const int ta = (t + 1 < U) ? t + 1 : -1;
        if (pf && Tp <= 128)      hipLaunchKernelGGL((dec_step_bwd_pf_kernel<CELL, 8>), dim3(B), dim3(RNT), lds_bf, st, ds, ta, t);
        else if (pf && Tp <= 160) hipLaunchKernelGGL((dec_step_bwd_pf_kernel<CELL, 10>), dim3(B), dim3(RNT), lds_bf, st, ds, ta, t);
        else if (pf && Tp <= 192) hipLaunchKernelGGL((dec_step_bwd_pf_kernel<CELL, 12>), dim3(B), dim3(RNT), lds_bf, st, ds, ta, t);
        else if (pf)              hipLaunchKernelGGL((dec_step_bwd_pf_kernel<CELL, 14>), dim3(B), dim3(RNT), lds_bf, st, ds, ta, t);
        else if (bfrows && A <= 128) hipLaunchKernelGGL((dec_step_bwd_bf_kernel<CELL, 1>), dim3(B), dim3(RNT), lds_bf, st, ds, (t + 1 < U) ? t + 1 : -1, t);
        else if (bfrows) hipLaunchKernelGGL((dec_step_bwd_bf_kernel<CELL, 2>), dim3(B), dim3(RNT), lds_bf, st, ds, (t + 1 < U) ? t + 1 : -1, t);
        else if (loc) hipLaunchKernelGGL((dec_step_bwd_kernel<CELL, FAST, true>), dim3(B), dim3(RNT), lds, st, ds, (t + 1 < U) ? t + 1 : -1, t);
        else          hipLaunchKernelGGL((dec_step_bwd_kernel<CELL, FAST, false>), dim3(B), dim3(RNT), lds, st, ds, (t + 1 < U) ? t + 1 : -1, t);
        LAS_LAUNCHED();
        if (t < 0) break;
        for (int l = TOP; l >= 0; --l) {
            const float* dG = d.gates + ((size_t)l * U + t) * B * GD;
            if (l == 0 && skinny) {
                GEMM_OK(las_skinny_gemm_bf16(d.dgbf, GD, B, GD, packB, I0D, d.dXin0 + (size_t)t * B * I0D, I0D, nullptr, st));
            } else if (l == 0) {
                GEMM_OK(las_gemm(prec, 0, 1, B, I0D, GD, 1.f, dG, GD, 0, f->cellW[0], GD, 0, 0.f, d.dXin0 + (size_t)t * B * I0D,
                                 I0D, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
            } else {
                float* tl = tmp + (size_t)l * B * 2 * D;
                GEMM_OK(las_gemm(prec, 0, 1, B, 2 * D, GD, 1.f, dG, GD, 0, f->cellW[l], GD, 0, 0.f, tl, 2 * D, 0, nullptr,
                                 LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
                hipLaunchKernelGGL((dec_pointwise_bwd_kernel<CELL, FAST>), dim3(B), dim3(256), 0, st, d, l - 1, t, (const float*)tl, 2 * D);
                LAS_LAUNCHED();
            }
        }
    }

    if (wide) {}    // (wide_bwd_steps ran its own keys kernel)
    else if (locloop) {  // keys gradient with the conv term in the pre-activation; the same pass leaves the Wf-gradient partials
        hipLaunchKernelGGL(dkeys_loc_kernel, dim3(cdiv(Tp, 8), B), dim3(256), 0, st, d, bk->d_keys);
        LAS_LAUNCHED();
    } else if (bfrows) {   // keys gradient: contraction over the steps, every (utterance, frame) independent
        hipLaunchKernelGGL(dkeys_kernel, dim3(cdiv(Tp, 8), B), dim3(256), 0, st, d, bk->d_keys);
        LAS_LAUNCHED();
    }
    // d_enc[b] += alphas[:,b,:]^T . dctx[:,b,:]   (batched over utterances; contraction over the U steps)
    GEMM_OK(las_gemm(prec, 1, 0, Tp, Hd, U, 1.f, d.alphas, B * Tp, Tp, d.dXin0 + E, B * I0D, I0D, 1.f, bk->d_enc, Hd,
                     (long long)Tp * Hd, nullptr, LAS_ACT_NONE, B, 0, 0, nullptr, 0, st));
    }
    if (!(part & 2)) return 0;
    // ---- weight gradients: one tall contraction each (K = U*B), deterministic split-K
    const int UB = U * B;
    GEMM_OK(las_gemm(prec, 1, 0, I0D, GD, UB, 1.f, d.xin0, I0D, 0, d.gates, GD, 0, 1.f, bk->dcellW[0], GD, 0, nullptr,
                     LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
    for (int l = 0; l < NL; ++l) {
        const float* dG = d.gates + (size_t)l * U * B * GD;
        GEMM_OK(las_colsum(dG, UB, GD, GD, 1.f, bk->dcellb[l], gws, gws_bytes, st));
        if (l > 0) {
            GEMM_OK(las_gemm(prec, 1, 0, D, GD, UB, 1.f, d.hs + ((size_t)(l - 1) * (U + 1) + 1) * B * D, D, 0, dG, GD, 0, 1.f,
                             bk->dcellW[l], GD, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
            GEMM_OK(las_gemm(prec, 1, 0, D, GD, UB, 1.f, d.hs + (size_t)l * (U + 1) * B * D, D, 0, dG, GD, 0, 1.f,
                             bk->dcellW[l] + (size_t)D * GD, GD, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
        }
        GEMM_OK(las_gemm(prec, 1, 0, D, A, UB, 1.f, d.hs + (size_t)l * (U + 1) * B * D, D, 0, d.dQ, A, 0, 1.f,
                         bk->dWs + (size_t)l * D * A, A, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
    }
    GEMM_OK(las_gemm(prec, 1, 0, D, V, UB, 1.f, d.hs + ((size_t)TOP * (U + 1) + 1) * B * D, D, 0, bk->dlogits, V, 0, 1.f,
                     bk->dWv, V, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
    GEMM_OK(las_colsum(bk->dlogits, UB, V, V, 1.f, bk->dbv, gws, gws_bytes, st));
    GEMM_OK(las_colsum(d.duRows, B, A, A, 1.f, bk->du, gws, gws_bytes, st));
    if (loop)   // dXin0[:, :, 0:E] = dG . W0[0:E, :]^T for all steps at once (the loop kernel left these columns out)
        GEMM_OK(las_gemm(prec, 0, 1, UB, E, GD, 1.f, d.gates, GD, 0, f->cellW[0], GD, 0, 0.f, d.dXin0, I0D, 0, nullptr, LAS_ACT_NONE, 1,
                         0, 0, nullptr, 0, st));
    if (V <= EMB_MAX_V && UB <= EMB_MAX_N) {      // bucket the positions by token, reduce the rows that occur (accumulates into demb)
        int* estart = (int*)(base + w.embp);
        int* epos = estart + V + 1;
        // (the token list / the histogram in LDS: up to 96 / 68 KB at the limits -- above the 64 KB a launch gets without asking)
        static int attr_h = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(emb_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (EMB_MAX_V + 1024) * (int)sizeof(int));
        static int attr_p = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(emb_place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (EMB_MAX_N + 264) * (int)sizeof(unsigned short));
        LAS_ARG(attr_h == 0 && attr_p == 0, "hipFuncSetAttribute(emb_*_kernel) failed: %d %d", attr_h, attr_p);
        hipLaunchKernelGGL(emb_hist_kernel, dim3(1), dim3(1024), (size_t)(V + 1024) * sizeof(int), st, (const int*)d.tok_in, UB, V, estart);
        LAS_LAUNCHED();
        hipLaunchKernelGGL(emb_place_kernel, dim3(cdiv(UB, 256)), dim3(256), (size_t)((UB + 263) & ~7) * sizeof(unsigned short), st,
                           (const int*)d.tok_in, UB, V, (const int*)estart, epos);
        LAS_LAUNCHED();
        hipLaunchKernelGGL(emb_reduce_kernel, dim3(V), dim3(1024), 0, st, (const int*)estart, (const int*)epos, (const float*)d.dXin0, I0D, E,
                           (const float*)d.emb_mask, bk->demb);
        LAS_LAUNCHED();
    } else {
        float* epart = (float*)(base + w.embp);
        hipLaunchKernelGGL(emb_grad_kernel, dim3(V, EMB_CHUNKS), dim3(256), 0, st, (const int*)d.tok_in, (const float*)d.dXin0, UB, I0D,
                           E, V, d.emb_mask, epart);
        LAS_LAUNCHED();
        GEMM_OK(las_colsum(epart, EMB_CHUNKS, V * E, V * E, 1.f, bk->demb, gws, gws_bytes, st));
    }
    bool loc_rows = loc;                           // the per-utterance rows dlocwRows / dlocbRows hold the filter / bias gradient
    if (locloop || (wide && loc)) {  // filter / bias gradient from the saved d f rows of every step
        const int nks = (Tp + 31) / 32, nmt = (d.Kc + 15) / 16;
        const size_t lds = (size_t)(((nks * 32 + 16 * nmt + 8 + 3) & ~3) + 16 * (nks * 32 + 4) + 256) * sizeof(float);
        if (nmt <= 16 && lds <= 64 * 1024) {
            float* wp = (float*)(base + w.dlocwP);
            float* bp = (float*)(base + w.dlocbP);
            hipLaunchKernelGGL(dlocw_mfma_kernel, dim3(DLW_SPLIT, B), dim3(256), lds, st, d, wp, bp);
            LAS_LAUNCHED();
            GEMM_OK(las_colsum(wp, B * DLW_SPLIT, d.Kc * d.C, d.Kc * d.C, 1.f, bk->dloc_w, gws, gws_bytes, st));
            GEMM_OK(las_colsum(bp, B * DLW_SPLIT, d.C, d.C, 1.f, bk->dloc_b, gws, gws_bytes, st));
            loc_rows = false;
        } else {
            hipLaunchKernelGGL(dlocw_kernel, dim3(cdiv(d.Kc * d.C, 256), B), dim3(256), (size_t)(Tp * d.C + Tp) * sizeof(float), st, d);
            LAS_LAUNCHED();
        }
    }
    if (loc_rows) {
        GEMM_OK(las_colsum(d.dlocwRows, B, d.Kc * d.C, d.Kc * d.C, 1.f, bk->dloc_w, gws, gws_bytes, st));
        GEMM_OK(las_colsum(d.dlocbRows, B, d.C, d.C, 1.f, bk->dloc_b, gws, gws_bytes, st));
    }
    if (loc && wide) {
        GEMM_OK(las_colsum((const float*)(base + w.wide + wide_layout(B, Tp, A, D, NL, G, d.C).dWfW), B * cdiv(Tp, 8), d.C * A, d.C * A, 1.f, bk->dWf, gws,
                           gws_bytes, st));
    } else if (loc) {
        GEMM_OK(las_colsum(d.dWfRows, B * RNG, d.C * A, d.C * A, 1.f, bk->dWf, gws, gws_bytes, st));
    }
    return 0;
}

extern "C" int las_speller_bwd(const las_speller_bwd_args* bk, void* stream) { return las_speller_bwd_part(bk, 3, stream); }

extern "C" int las_speller_bwd_part(const las_speller_bwd_args* bk, int part, void* stream) {
    LAS_ARG(bk, "las_speller_bwd: null args");
    LAS_ARG(part >= 1 && part <= 3, "las_speller_bwd_part: part must be 1 (loop + input gradients), 2 (parameter gradients) or 3");
    DecDev d;
    if (int rc = fill_dev(&bk->f, d)) return rc;
    LAS_ARG(bk->dlogits && bk->d_enc && bk->d_keys && bk->dWs && bk->du && bk->demb && bk->dWv && bk->dbv && bk->dcellW && bk->dcellb,
            "las_speller_bwd: null gradient pointer");
    LAS_ARG(bk->f.mode == LAS_ATT_ADD || (bk->dloc_w && bk->dloc_b && bk->dWf), "las_speller_bwd: null location gradient pointer");
    LAS_ARG((((uintptr_t)bk->d_keys) & 15) == 0, "las_speller_bwd: d_keys must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool fast = bk->f.prec == LAS_PREC_BF16;
    if (bk->f.cell == LAS_CELL_LSTM)
        return fast ? speller_bwd_impl<LAS_CELL_LSTM, true>(bk, d, part, st) : speller_bwd_impl<LAS_CELL_LSTM, false>(bk, d, part, st);
    return fast ? speller_bwd_impl<LAS_CELL_RNN, true>(bk, d, part, st) : speller_bwd_impl<LAS_CELL_RNN, false>(bk, d, part, st);
}
