// beam.hip -- K10: one pruning step of BeamSearch.decode (reference las/beam_search.py:119-152 and
// _select_best_k :297-312) for many utterances at once, on device.
//
// Reference semantics restated (see include/las_hip.h): raw logits are the scores; a hypothesis'
// score is the float32 running sum (0 + np.float32 -> np.float32 in BeamState.update,
// las/beam_search.py:27); candidates are ranked by sum/len in float32; t=0 expands hypothesis 0 only;
// SOS is never re-emitted after t=0.  The per-hypothesis top-64 cut (las/beam_search.py:123) cannot
// bind while beam < 64 (the ranking key is monotone in the logit inside one hypothesis), so the kernel
// selects the global top-`beam` of the num_live x V candidate grid directly and requires beam < topn.
// Tie order = the order a stable ascending sort of the reference's candidate bank would give:
// (score/len, hypothesis index, logit, token id).
// One workgroup per utterance; `beam` rounds of a block-wide arg-max over the candidates that are
// strictly below the previous pick -- HBM-bound on the logits (read `beam` times from L2).
#include "las_common.h"

struct BKey { float norm; int i; float l; int v; };

__device__ __forceinline__ bool bless(const BKey& a, const BKey& b) {   // a < b in rank order
    if (a.norm != b.norm) return a.norm < b.norm;
    if (a.i != b.i) return a.i < b.i;
    if (a.l != b.l) return a.l < b.l;
    return a.v < b.v;
}

#ifdef LAS_BEAM_STAMPS   // development aid (make ablf F=beam D=-DLAS_BEAM_STAMPS; tools/probe_beam_stamps.py): 100 MHz phase stamps of utterance 0's workgroup
__device__ unsigned long long g_beam_stamps[16];
#define BSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 0) g_beam_stamps[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int las_dev_beam_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_beam_stamps), sizeof(g_beam_stamps)); }
#else
#define BSTAMP(i)
#endif
// The `beam` best candidates of one utterance, best first, into L.picks[] (LDS); returns how many exist.  Called by all 256 threads.
//
// Up to 512 candidates (beam x V of a char model): a candidate becomes two sortable integers -- KA = order(norm) (32 bits) and
// KB = hypothesis : order(logit) : token (58 bits), order() = the usual monotone map of float bits -- and a WAVE finds the value of the
// n-th largest KA among its candidates by bisection over the 32 bits: every probe is one compare per candidate slot whose ballot is
// counted on the scalar unit, no data crosses lanes.  Ties at the threshold (equal normalised scores) are cut the same way on KB.
// Two levels: each of the 4 waves selects the `beam` best of its quarter of the grid (2 candidates per lane), the <= 4 x beam survivors
// meet in LDS and wave 0 selects among them again, then ranks the <= 64 winners among themselves by counting.  r3 decode traces at
// beam 16, V = 30: 16 rounds of (re-scan + 8-level LDS tree, 9 barriers) 44 us -> 16 rounds of (wave butterfly + 1 barrier) 28 us ->
// one wave bisecting 8 slots of 64-bit keys 19 us -> this.
// Larger grids (subword vocabularies) take the round-based path: block-wide maximum of the candidates strictly below the previous pick,
// wave butterfly (64-lane xor shuffles) + one LDS slot per wave, one barrier per round (the slots are double-buffered).
struct BeamLds {
    BKey picks[64];
    BKey wbest[2][4];
    unsigned ka[320]; unsigned long long kb[320];        // [0, 256): the waves' survivors (wave w at 64 w), [256, 320): the winners
    int ci[320], cv[320]; float cl[320];
    int cnt[4];
    int count;
};
__device__ __forceinline__ unsigned f_order(float f) {                         // unsigned order == float order (-0 canonicalised by the caller)
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ BKey bkey_shfl_xor(const BKey& k, int off) {
    BKey y;
    y.norm = __shfl_xor(k.norm, off, 64); y.i = __shfl_xor(k.i, off, 64); y.l = __shfl_xor(k.l, off, 64); y.v = __shfl_xor(k.v, off, 64);
    return y;
}
__device__ __forceinline__ void bkey_max(BKey& best, const BKey& y) {          // i < 0: no candidate
    if (y.i >= 0 && (best.i < 0 || bless(best, y))) best = y;
}
__device__ __forceinline__ bool beam_make_key(BKey& k, const float* lg, const float* sc, const int* ln, int idx, int V, int t, int start_id) {
    const int i = idx / V;
    const int v = idx - i * V;
    if (t > 0 && v == start_id) return false;           // las/beam_search.py:127-128
    const float l = lg[idx];
    k.norm = (sc[i] + l) / (float)(ln[i] + 1);          // float32 sum, float32 divide (las/beam_search.py:27,306)
    k.i = i; k.l = l; k.v = idx;                        // v carries the flat candidate id until the very end
    return k.norm == k.norm;                            // NaN never ranks
}
// One wave: the `beam` largest (KA, KB) among its flagged candidates (bit s of `on`: this lane's slot s) are appended to the LDS lists at
// [obase, obase + n); returns n = min(beam, flagged).  Largest T with count(key >= T) >= n is the n-th largest key.
template <int NS>
__device__ __forceinline__ int beam_wave_select(const unsigned (&KA)[NS], const unsigned long long (&KB)[NS], const int (&ci)[NS],
                                                const float (&cl)[NS], const int (&cv)[NS], const unsigned on, const int beam, const int lane,
                                                BeamLds& L, const int obase) {
    // (ballots through the builtin: hip's __ballot materialises the predicate as an integer and compares it again -- v_cmp, s_and, s_nop,
    //  v_cndmask, v_cmp, s_bcnt1 per count, ~170 cycles per probe of two slots: r5 stamps, the 64 probes of the two levels were 5.4 us of a
    //  14.5 us launch.  The slots' validity as wave masks, one compare + s_and + s_bcnt1 per count.)
    unsigned long long onm[NS];
    int nvalid = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) { onm[s] = __builtin_amdgcn_ballot_w64(((on >> s) & 1u) != 0); nvalid += __builtin_popcountll(onm[s]); }
    const int need = nvalid < beam ? nvalid : beam;
    if (need == 0) return 0;
    unsigned TA = 0;
    for (int b = 31; b >= 0; --b) {
        const unsigned probe = TA | (1u << b);
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(KA[s] >= probe) & onm[s]);
        if (cnt >= need) TA = probe;                     // uniform (ballots)
    }
    int gt = 0, eq = 0;
    unsigned tie = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool o = (on >> s) & 1u;
        if (o && KA[s] == TA) tie |= 1u << s;
        gt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(o && KA[s] > TA)); eq += __builtin_popcountll(__builtin_amdgcn_ballot_w64(((tie >> s) & 1u) != 0));
    }
    unsigned long long TB = 0;
    if (eq > need - gt) {                                // equal normalised scores at the threshold: the (need - gt) largest KB of the ties
        for (int b = 57; b >= 0; --b) {
            const unsigned long long probe = TB | (1ull << b);
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(((tie >> s) & 1u) && KB[s] >= probe));
            if (cnt >= need - gt) TB = probe;
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = obase;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool sel = ((on >> s) & 1u) && (KA[s] > TA || (KA[s] == TA && KB[s] >= TB));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
        if (sel) {
            const int pos = base + __popcll(m & lt);
            L.ka[pos] = KA[s]; L.kb[pos] = KB[s]; L.ci[pos] = ci[s]; L.cl[pos] = cl[s]; L.cv[pos] = cv[s];
        }
        base += __popcll(m);
    }
    return need;
}
__device__ __forceinline__ int beam_rank(const float* lg, const float* sc, const int* ln, int nb, int V, int t, int start_id, int beam, BeamLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncand = nb * V;
    if (ncand <= 512) {
        {   // level 1: every wave, the two candidates per lane of its quarter of the grid
            unsigned KA[2]; unsigned long long KB[2]; int ci[2], cv[2]; float cl[2];
            unsigned on = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                KA[s] = 0; KB[s] = 0; ci[s] = 0; cv[s] = 0; cl[s] = 0.f;
                const int idx = s * 256 + tid;
                BKey k;
                if (idx < ncand && beam_make_key(k, lg, sc, ln, idx, V, t, start_id)) {
                    on |= 1u << s;
                    KA[s] = f_order(k.norm + 0.f);
                    KB[s] = ((unsigned long long)(unsigned)k.i << 52) | ((unsigned long long)f_order(k.l + 0.f) << 20) | (unsigned)(k.v - k.i * V);
                    ci[s] = k.i; cl[s] = k.l; cv[s] = k.v;
                }
            }
            BSTAMP(7);
            const int n = beam_wave_select<2>(KA, KB, ci, cl, cv, on, beam, lane, L, w * 64);
            if (lane == 0) L.cnt[w] = n;
            BSTAMP(8);
        }
        __syncthreads();
        BSTAMP(9);
        if (w == 0) {   // level 2: the survivors, then the winners' order
            int need;
            if (4 * beam <= 64) {
                // every wave kept <= beam survivors: all of them in ONE slot (lane = wave's list * beam + position), a quarter of the ballots
                unsigned KA[1]; unsigned long long KB[1]; int ci[1], cv[1]; float cl[1];
                const int sw = (lane >= beam) + (lane >= 2 * beam) + (lane >= 3 * beam), pos = lane - sw * beam;
                const bool o = lane < 4 * beam && pos < L.cnt[sw];
                const int e = sw * 64 + (o ? pos : 0);
                KA[0] = L.ka[e]; KB[0] = L.kb[e]; ci[0] = L.ci[e]; cl[0] = L.cl[e]; cv[0] = L.cv[e];
                need = beam_wave_select<1>(KA, KB, ci, cl, cv, o ? 1u : 0u, beam, lane, L, 256);
            } else {    // (wave s's list = slot s)
                unsigned KA[4]; unsigned long long KB[4]; int ci[4], cv[4]; float cl[4];
                unsigned on = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool o = lane < L.cnt[s];
                    const int e = s * 64 + (o ? lane : 0);
                    KA[s] = L.ka[e]; KB[s] = L.kb[e]; ci[s] = L.ci[e]; cl[s] = L.cl[e]; cv[s] = L.cv[e];
                    if (o) on |= 1u << s;
                }
                need = beam_wave_select<4>(KA, KB, ci, cl, cv, on, beam, lane, L, 256);
            }
            BSTAMP(10);
            // (one wave: its LDS writes are ordered before its later LDS reads; keep the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < need) {
                const unsigned ma = L.ka[256 + lane];
                const unsigned long long mb = L.kb[256 + lane];
                int r = 0;
                for (int q = 0; q < need; ++q) {         // winner q's key from lane q (a scalar broadcast: the LDS reads of this loop were 1.6 us)
                    const unsigned qa = __builtin_amdgcn_readlane(ma, q);
                    const unsigned long long qb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((unsigned)(mb >> 32), q) << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane((unsigned)mb, q);
                    r += (qa > ma || (qa == ma && qb > mb)) ? 1 : 0;
                }
                const BKey k = {0.f, L.ci[256 + lane], L.cl[256 + lane], L.cv[256 + lane]};
                L.picks[r] = k;
            }
            if (lane == 0) L.count = need;
        }
        __syncthreads();
        return L.count;
    }
    BKey last = {0.f, 0, 0.f, 0};
    int count = 0;
    for (int pick = 0; pick < beam; ++pick) {
        BKey best = {0.f, -1, 0.f, 0};
        for (int idx = tid; idx < ncand; idx += 256) {
            BKey k;
            if (beam_make_key(k, lg, sc, ln, idx, V, t, start_id) && (pick == 0 || bless(k, last))) bkey_max(best, k);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bkey_max(best, bkey_shfl_xor(best, off));
        if (lane == 0) L.wbest[pick & 1][w] = best;
        __syncthreads();
        best = L.wbest[pick & 1][0];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) bkey_max(best, L.wbest[pick & 1][ww]);
        if (best.i < 0) break;                           // uniform: every thread combined the same four slots
        last = best;
        if (tid == 0) L.picks[pick] = best;
        count = pick + 1;
    }
    __syncthreads();
    return count;
}

__global__ __launch_bounds__(256) void beam_step_kernel(const float* __restrict__ logits, const float* __restrict__ score,
                                                        const int* __restrict__ length, const int* __restrict__ nlive,
                                                        int beam, int V, int t, int start_id, int* __restrict__ out_parent,
                                                        int* __restrict__ out_token, float* __restrict__ out_score,
                                                        int* __restrict__ out_n) {
    __shared__ BeamLds L;
    const int u = blockIdx.x, tid = threadIdx.x;
    int nb = nlive[u];
    if (nb > beam) nb = beam;
    if (t == 0 && nb > 1) nb = 1;                       // las/beam_search.py:119
    const float* lg = logits + (size_t)u * beam * V;
    const float* sc = score + (size_t)u * beam;
    const int* ln = length + (size_t)u * beam;
    const int count = beam_rank(lg, sc, ln, nb, V, t, start_id, beam, L);
    if (tid == 0) out_n[u] = count;
    for (int j = tid; j < count; j += 256) {             // ascending, best last (las/beam_search.py:310-312)
        const BKey k = L.picks[count - 1 - j];
        out_parent[(size_t)u * beam + j] = k.i;
        out_token[(size_t)u * beam + j] = k.v - k.i * V;
        out_score[(size_t)u * beam + j] = sc[k.i] + k.l;
    }
}

extern "C" int las_beam_step(const float* logits, const float* score, const int* length, const int* nlive, int nutt, int beam,
                             int V, int topn, int t, int start_id, int* out_parent, int* out_token, float* out_score,
                             int* out_n, void* stream) {
    LAS_ARG(logits && score && length && nlive && out_parent && out_token && out_score && out_n, "las_beam_step: null pointer");
    LAS_ARG(nutt > 0 && beam > 0 && V > 0 && t >= 0, "las_beam_step: bad dims");
    LAS_ARG(beam < topn && beam <= 64, "las_beam_step: needs beam < topn (%d) and beam <= 64 (got %d)", topn, beam);
    hipLaunchKernelGGL(beam_step_kernel, dim3(nutt), dim3(256), 0, (hipStream_t)stream, logits, score, length, nlive, beam, V, t,
                       start_id, out_parent, out_token, out_score, out_n);
    LAS_LAUNCHED();
    return 0;
}

// ====================================================================================================
// Device-resident beam search (K10b).  The reference keeps its hypotheses as Python objects and crosses to the
// host after every sess.run (las/beam_search.py:94-158).  Here the whole loop state of `nutt` utterances lives in
// device buffers: per live hypothesis its running score / length / last token, per step the picks
// (parent, token, score) as back-pointer records, the retired (EOS) hypotheses as (step, pick) references, and the
// per-utterance termination of the reference's `while t < dec_step and len(selected) < beam_size` (:94).  One launch per
// step prunes every utterance (same ranking code as beam_step_kernel) AND does that bookkeeping; a second kernel
// gathers the recurrent state rows of the survivors.  The host only replays the launches and reads the records once,
// after the last step.
// ====================================================================================================
struct BeamLoopDev {
    float* logits;                  // [nutt, beam, V] raw logits of this step (row = utterance * beam + live slot); written when proj_w is set
    float* score; int* length;      // [nutt, beam] live hypotheses: running float32 sum / tokens after SOS
    int* nlive; int* nsel; int* done;      // [nutt]
    const int* dec_step;            // [nutt] step bound int(audiolen * convert_rate) (las/beam_search.py:78)
    int* step;                      // [1] device-resident step counter t (read by every workgroup, advanced by the gather kernel)
    int *hist_parent, *hist_token, *hist_slot; float* hist_score; int* hist_n;   // [Umax, nutt, beam] / [Umax, nutt]
    int *sel_t, *sel_j;             // [nutt, selcap] retired hypotheses in the reference's append order
    int* src_row;                   // [nutt, beam] out: live slot k of the next step continues row src_row (global row index)
    int* next_token;                // [nutt * beam] out: token entering the next step for each live slot
    int nutt, beam, V, Umax, selcap, start_id, end_id;
    const float* file_in; float* file_out; long long file_n;   // optional per-step record: file_out[t][0..file_n) = file_in[0..file_n) (workgroups >= nutt)
    // optional vocabulary projection inside the kernel: logits[r] = proj_b + bf16([h0[r] ; h1[r]]) . proj_w (fragments of a [k0 + k1, V] matrix)
    const float *proj_h0, *proj_h1; int proj_k0, proj_k1; const u16x8_t* proj_w; const float* proj_b;
    // fold_gather (round 5): the state gather of the surviving parents inside this launch
    int gather, ntens; const float* g_in[16]; float* g_out[16]; int g_width[16];
};
constexpr int BEAM_PROJ_TILES = 8;

// The last workgroup to finish advances the device step counter: every workgroup has read step[0] (at its start) by then, and nothing else
// of this launch reads it afterwards.  step[1] counts the arrivals and is reset for the next launch.
__device__ __forceinline__ void beam_finish(const BeamLoopDev& a) {
    if (!a.gather) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int n = atomicAdd(a.step + 1, 1);
        if (n == (int)gridDim.x - 1) { a.step[1] = 0; __threadfence(); a.step[0] += 1; }
    }
}
// state rows of utterance u follow their hypotheses: out[k][u * beam + j] = in[k][src_row[u * beam + j]] (all sources are rows of the
// SAME utterance, read from `in`, written to `out`: no hazard inside the launch)
__device__ __forceinline__ void beam_gather_rows(const BeamLoopDev& a, const int u, const int* srcs) {
    const int tid = threadIdx.x, beam = a.beam;
    for (int k = 0; k < a.ntens; ++k) {
        const int w = a.g_width[k];
        const float* ip = a.g_in[k];
        float* op = a.g_out[k];
        if (((w & 3) | ((size_t)ip & 15) | ((size_t)op & 15)) == 0) {
            const int w4 = w >> 2;
            for (int i = tid; i < beam * w4; i += 256) {
                const int j = i / w4, c = i - j * w4;
                const int src = srcs[j];
                reinterpret_cast<float4*>(op + ((size_t)u * beam + j) * w)[c] = reinterpret_cast<const float4*>(ip + (size_t)src * w)[c];
            }
        } else {
            for (int i = tid; i < beam * w; i += 256) {
                const int j = i / w, c = i - j * w;
                op[((size_t)u * beam + j) * w + c] = ip[(size_t)srcs[j] * w + c];
            }
        }
    }
}      // (row tiles of 16 hypotheses) x (column tiles of 16 tokens) the in-kernel projection handles

__global__ __launch_bounds__(256) void beam_loop_kernel(BeamLoopDev a) {
    __shared__ BeamLds L;
    __shared__ int srcs[64];                                // fold_gather: the rows the utterance's new live slots continue (beam <= 64)
    kernarg_warm<(int)sizeof(BeamLoopDev)>();
    const int u = blockIdx.x, tid = threadIdx.x, beam = a.beam, V = a.V;
#ifdef LAS_BEAM_STAMPS
    const unsigned long long bs_e0 = wall_clock64();      // (written with stamp 1, for launches inside the step bound only: the replayed graph runs past it)
#endif
    // Every word the launch branches on, and (usual geometry) every operand of the in-kernel projection, is requested HERE, before the
    // first branch: the step counter, the utterance's done / bound / live words and then the projection's loads were four dependent
    // round trips in a row at the head of a 15 us launch (round 5).  All addresses are valid whatever the words turn out to be.
    const int uc = u < a.nutt ? u : 0;
    // (the step counter through the VECTOR memory path: as a scalar load its miss -- the previous launch wrote it -- is waited for by the next
    //  lgkmcnt(0) of the argument loads, in front of the projection's operands; r5 stamps)
    const int t_v = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int done_u = a.done[uc], bound_u = a.dec_step[uc], live_u = a.nlive[uc], nsel_u = a.nsel[uc];
    // the live hypotheses' running sums and lengths: read by the ranking (per candidate) and again by the bookkeeping (per pick) -- one
    // round trip each, behind the projection; requested here, kept in LDS
    __shared__ float sc_s[64];
    __shared__ int ln_s[64];
    const float sc_pre = a.score[(size_t)uc * beam + (tid < beam ? tid : 0)];
    const int ln_pre = a.length[(size_t)uc * beam + (tid < beam ? tid : 0)];
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, r = lane & 15;
    const int RT = (beam + 15) >> 4, CT = (V + 15) >> 4, KS0 = a.proj_w ? a.proj_k0 >> 5 : 0, KS = KS0 + (a.proj_w ? a.proj_k1 >> 5 : 0);
    const bool usual = a.proj_w && RT == 1 && CT <= 2 && KS <= 32 && KS > 0;
    float4 xa[8][2]; u16x8_t bw[8][2];
    if (usual) {
        // (beam <= 16, a char vocabulary, K <= 1024): every operand of this wave's <= 8 k-steps is requested before the first product
        // (a loop with one round trip to L2 per k-step was 8 us of this kernel)
        const int hr = r < beam ? r : beam - 1, row = uc * beam + hr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ks = w + 4 * j, kc = ks < KS ? ks : KS - 1;
            const float* src = kc < KS0 ? a.proj_h0 + (size_t)row * a.proj_k0 + kc * 32 + g * 8
                                        : a.proj_h1 + (size_t)row * a.proj_k1 + (kc - KS0) * 32 + g * 8;
            xa[j][0] = *reinterpret_cast<const float4*>(src); xa[j][1] = *reinterpret_cast<const float4*>(src + 4);
            bw[j][0] = a.proj_w[(size_t)kc * 64 + lane];
            bw[j][1] = a.proj_w[((size_t)(CT > 1 ? KS : 0) + kc) * 64 + lane];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int t = __builtin_amdgcn_readfirstlane(t_v);
#ifdef LAS_BEAM_STAMPS
    const unsigned long long bs_e1 = wall_clock64();
    if (t < a.Umax && threadIdx.x == 0 && blockIdx.x == 0) { g_beam_stamps[0] = bs_e0; g_beam_stamps[1] = bs_e1; }
#endif
    if (t >= a.Umax) { beam_finish(a); return; }
    if (u >= a.nutt) {                                    // filing workgroups: this step's row tensor (the alignments) under the DEVICE step counter
        const int nf = (int)gridDim.x - a.nutt;
        float* dst = a.file_out + (size_t)t * a.file_n;
        if (((a.file_n & 3) | ((size_t)a.file_in & 15) | ((size_t)dst & 15)) == 0) {
            const long long n4 = a.file_n >> 2;
            for (long long i = (long long)(u - a.nutt) * 256 + tid; i < n4; i += (long long)nf * 256)
                reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(a.file_in)[i];
        } else {
            for (long long i = (long long)(u - a.nutt) * 256 + tid; i < a.file_n; i += (long long)nf * 256) dst[i] = a.file_in[i];
        }
        beam_finish(a);
        return;
    }
    int* hn = a.hist_n + (size_t)t * a.nutt + u;
    if (done_u || t >= bound_u) {                        // retired utterance: nothing to do (its rows compute ignored garbage)
        if (tid == 0) { *hn = 0; a.done[u] = 1; a.nlive[u] = 0; }
        for (int k = tid; k < beam; k += 256) { a.src_row[(size_t)u * beam + k] = u * beam; srcs[k] = u * beam; }
        if (a.gather) { __syncthreads(); beam_gather_rows(a, u, srcs); }      // (rows of a retired utterance: ignored garbage, but finite)
        beam_finish(a);
        return;
    }
    int nb = live_u;
    if (nb > beam) nb = beam;
    if (t == 0 && nb > 1) nb = 1;                       // las/beam_search.py:119
    const float* lg = a.logits + (size_t)u * beam * V;
    float* sc = a.score + (size_t)u * beam;
    int* ln = a.length + (size_t)u * beam;
    if (tid < 64) { sc_s[tid] = sc_pre; ln_s[tid] = ln_pre; }      // (beam <= 64; published by the barriers below)
    if (!a.proj_w) __syncthreads();
    if (a.proj_w) {
        // the step's logits are computed HERE: [beam rows] x [K = k0 + k1] x [V] on the matrix cores (the Speller's output layer and the
        // LM's, pre-scaled by lm_weight and shifted to its token columns, are one concatenated product) -- the 4 waves split the k-steps,
        // their partial tiles meet in LDS; the ranking then reads the logits from LDS.  Saves two launches per decode step.
        __shared__ float red[4][BEAM_PROJ_TILES][64][4];
        __shared__ float lgs[BEAM_PROJ_TILES * 256];
        // (the usual geometry's two accumulators are NOT elements of acc[]: the general path indexes that array with run-time tile numbers,
        //  which puts it in scratch -- every MFMA of the usual path was a scratch load and a scratch store, r5 ISA)
        f32x4_t acc[BEAM_PROJ_TILES];
        f32x4_t ua0 = {0.f, 0.f, 0.f, 0.f}, ua1 = ua0;
        if (usual) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (w + 4 * j < KS) {
                    const u32x4_t pk = {f2bf2(xa[j][0].x, xa[j][0].y), f2bf2(xa[j][0].z, xa[j][0].w), f2bf2(xa[j][1].x, xa[j][1].y), f2bf2(xa[j][1].z, xa[j][1].w)};
                    const u16x8_t av = __builtin_bit_cast(u16x8_t, pk);
                    ua0 = mfma_bf16_16x16x32(av, bw[j][0], ua0);
                    if (CT > 1) ua1 = mfma_bf16_16x16x32(av, bw[j][1], ua1);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < BEAM_PROJ_TILES; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int ks = w; ks < KS; ks += 4) {
#pragma unroll
            for (int rt = 0; rt < BEAM_PROJ_TILES; ++rt) {
                if (rt >= RT) break;
                const int hr = rt * 16 + r, row = u * beam + (hr < beam ? hr : beam - 1);
                const float* src = ks < KS0 ? a.proj_h0 + (size_t)row * a.proj_k0 + ks * 32 + g * 8
                                            : a.proj_h1 + (size_t)row * a.proj_k1 + (ks - KS0) * 32 + g * 8;
                const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                const u32x4_t pk = {f2bf2(x0.x, x0.y), f2bf2(x0.z, x0.w), f2bf2(x1.x, x1.y), f2bf2(x1.z, x1.w)};
                const u16x8_t av = __builtin_bit_cast(u16x8_t, pk);
#pragma unroll
                for (int ct = 0; ct < BEAM_PROJ_TILES; ++ct) {
                    if (rt * CT + ct >= BEAM_PROJ_TILES || ct >= CT) break;
                    acc[rt * CT + ct] = mfma_bf16_16x16x32(av, a.proj_w[((size_t)ct * KS + ks) * 64 + lane], acc[rt * CT + ct]);
                }
            }
        }
        }
        BSTAMP(2);
        if (usual) {
            *reinterpret_cast<f32x4_t*>(&red[w][0][lane][0]) = ua0;
            if (CT > 1) *reinterpret_cast<f32x4_t*>(&red[w][1][lane][0]) = ua1;
        } else {
#pragma unroll
            for (int i = 0; i < BEAM_PROJ_TILES; ++i)
                if (i < RT * CT) *reinterpret_cast<f32x4_t*>(&red[w][i][lane][0]) = acc[i];
        }
        __syncthreads();
        BSTAMP(3);
        for (int idx = tid; idx < RT * CT * 256; idx += 256) {        // C layout: (row r16, col c16) of a tile sits in lane (r16 / 4) * 16 + c16, register r16 % 4
            const int tile = idx >> 8, o = idx & 255, r16 = o >> 4, c16 = o & 15, l2 = (r16 >> 2) * 16 + c16, reg = r16 & 3;
            const int rt = tile / CT, ct = tile - rt * CT, hr = rt * 16 + r16, col = ct * 16 + c16;
            if (hr < beam && col < V) {
                const float v = ((red[0][tile][l2][reg] + red[1][tile][l2][reg]) + (red[2][tile][l2][reg] + red[3][tile][l2][reg])) + a.proj_b[col];
                lgs[hr * V + col] = v;
                if (a.logits) a.logits[((size_t)u * beam + hr) * V + col] = v;
            }
        }
        __syncthreads();
        lg = lgs;
    }
    BSTAMP(4);
    const int count = beam_rank(lg, sc_s, ln_s, nb, V, t, a.start_id, beam, L);
    BSTAMP(5);
    if (tid < 64) {
    // the reference's bookkeeping (las/beam_search.py:147-152) in its iteration order = ascending rank (best last): lane j of the
    // first wave is pick j (beam <= 64); the positions of the retired / surviving picks in their lists are prefix counts of ballots
    const int j = tid;
    const bool act = j < count;
    BKey k = {0.f, 0, 0.f, 0};
    if (act) k = L.picks[count - 1 - j];
    const int v = k.v - k.i * V;
    const float news = sc_s[k.i] + k.l;                  // (the old sums: the LDS copies, untouched by the stores below)
    const int newl = ln_s[k.i] + 1;
    const bool isend = act && v == a.end_id, live = act && !isend;
    const unsigned long long mend = __builtin_amdgcn_ballot_w64(isend), mlive = __builtin_amdgcn_ballot_w64(live), below = (1ull << j) - 1ull;
    const int ns0 = nsel_u;
    const int eidx = ns0 + __popcll(mend & below), lidx = __popcll(mlive & below);
    const int nl = __popcll(mlive);
    int ns = ns0 + __popcll(mend);
    const size_t hb = ((size_t)t * a.nutt + u) * beam;
    if (act) { a.hist_parent[hb + j] = k.i; a.hist_token[hb + j] = v; a.hist_score[hb + j] = news; }
    if (isend && eidx < a.selcap) { a.sel_t[(size_t)u * a.selcap + eidx] = t; a.sel_j[(size_t)u * a.selcap + eidx] = j; }
    if (live) {
        a.hist_slot[hb + lidx] = j;                               // live slot lidx of step t+1 is pick j of step t
        a.src_row[(size_t)u * beam + lidx] = u * beam + k.i;
        srcs[lidx] = u * beam + k.i;
        a.next_token[(size_t)u * beam + lidx] = v;
        sc[lidx] = news; ln[lidx] = newl;
    }
    if (j >= nl && j < beam) { a.src_row[(size_t)u * beam + j] = u * beam; srcs[j] = u * beam; a.next_token[(size_t)u * beam + j] = a.start_id; }
    const bool exhausted = (t + 1 == bound_u);
    if (exhausted) {                                               // `if t == dec_step: selected.extend(beam_set)` (:155-156)
        if (live && ns + lidx < a.selcap) { a.sel_t[(size_t)u * a.selcap + ns + lidx] = t; a.sel_j[(size_t)u * a.selcap + ns + lidx] = j; }
        ns += nl;
    }
    if (tid == 0) {
        *hn = count;
        const bool fin = exhausted || ns >= beam || nl == 0;        // loop condition of :94 (+ no live hypothesis left)
        a.nsel[u] = ns;
        a.nlive[u] = fin ? 0 : nl;
        if (fin) a.done[u] = 1;
    }
    }
    BSTAMP(6);
    if (a.gather) {
        __syncthreads();                                               // wave 0's src_row stores (workgroup scope) before the other waves read them
        beam_gather_rows(a, u, srcs);
        beam_finish(a);
    }
}

// rows of `ntens` state tensors follow their hypotheses: out[k][r] = in[k][src_row[r]] (r = global row), then t += 1.
struct GatherDev { const float* in[16]; float* out[16]; int width[16]; int ntens; const int* src_row; int* step; int nrows; };
// One workgroup per ROW, all tensors (round 5; before: one workgroup per (row, tensor) copying 4 bytes per lane): the row's source index is
// read once, every tensor's piece of the row is requested before the first is stored (clamped addresses, predicated stores: no branch
// around a load), 16 bytes per lane where the widths and addresses allow.
__global__ __launch_bounds__(256) void beam_gather_kernel(GatherDev g) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const int src = g.src_row[r];
    bool vec = true;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < g.ntens) vec = vec && ((g.width[k] & 3) | ((size_t)g.in[k] & 15) | ((size_t)g.out[k] & 15)) == 0 && g.width[k] <= 1024;
    if (vec) {
        f32x4_t v[16];                                      // (native vectors and constant indices only: HIP's float4 array, or in[k < ntens ? k : 0],
#pragma unroll                                          //  put the whole thing in scratch)
        for (int k = 0; k < 16; ++k) {
            const bool on = k < g.ntens;
            const float* ip = on ? g.in[k] : g.in[0];
            const int w = on ? g.width[k] : g.width[0], w4 = w >> 2;
            v[k] = reinterpret_cast<const f32x4_t*>(ip + (size_t)src * w)[tid < w4 ? tid : 0];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < g.ntens && tid < (g.width[k] >> 2)) reinterpret_cast<f32x4_t*>(g.out[k] + (size_t)r * g.width[k])[tid] = v[k];
    } else {
        for (int k = 0; k < g.ntens; ++k) {
            const int w = g.width[k];
            const float* ip = g.in[k] + (size_t)src * w;
            float* op = g.out[k] + (size_t)r * w;
            for (int i = tid; i < w; i += 256) op[i] = ip[i];
        }
    }
    if (r == 0 && tid == 0) g.step[0] += 1;       // no workgroup of this kernel reads the counter
}
// ... and one workgroup per (row, tensor): few rows (a search over 16 utterances: 256) are latency-, not bandwidth-bound -- 7x the workgroups
// spread over the chip finish sooner than 256 that each wait for their source index and then for seven pieces (56.6 against 55.0 us per step)
__global__ __launch_bounds__(256) void beam_gather_rt_kernel(GatherDev g) {
    const int r = blockIdx.x, k = blockIdx.y;
    const int src = g.src_row[r];
    const int w = g.width[k];
    const float* ip = g.in[k] + (size_t)src * w;
    float* op = g.out[k] + (size_t)r * w;
    for (int i = threadIdx.x; i < w; i += 256) op[i] = ip[i];
    if (r == 0 && k == 0 && threadIdx.x == 0) g.step[0] += 1;       // no workgroup of this kernel reads the counter
}
__global__ void beam_advance_kernel(int* step) { step[0] += 1; }

extern "C" int las_beam_loop_step(const las_beam_loop_args* p, void* stream) {
    LAS_ARG(p, "las_beam_loop_step: null args");
    LAS_ARG(p->score && p->length && p->nlive && p->nsel && p->done && p->dec_step && p->step, "las_beam_loop_step: null state pointer");
    LAS_ARG(p->hist_parent && p->hist_token && p->hist_slot && p->hist_score && p->hist_n && p->sel_t && p->sel_j && p->src_row && p->next_token,
            "las_beam_loop_step: null record pointer");
    LAS_ARG(p->nutt > 0 && p->beam > 0 && p->V > 0 && p->Umax > 0 && p->selcap >= 3 * p->beam, "las_beam_loop_step: bad dims");
    LAS_ARG(p->beam < p->topn && p->beam <= 64, "las_beam_loop_step: needs beam < topn (%d) and beam <= 64 (got %d)", p->topn, p->beam);
    LAS_ARG(p->ntens >= 0 && p->ntens <= 16, "las_beam_loop_step: at most 16 state tensors");
    hipStream_t st = (hipStream_t)stream;
    BeamLoopDev a;
    a.logits = p->logits; a.score = p->score; a.length = p->length; a.nlive = p->nlive; a.nsel = p->nsel; a.done = p->done;
    a.dec_step = p->dec_step; a.step = p->step;
    a.hist_parent = p->hist_parent; a.hist_token = p->hist_token; a.hist_slot = p->hist_slot; a.hist_score = p->hist_score; a.hist_n = p->hist_n;
    a.sel_t = p->sel_t; a.sel_j = p->sel_j; a.src_row = p->src_row; a.next_token = p->next_token;
    a.nutt = p->nutt; a.beam = p->beam; a.V = p->V; a.Umax = p->Umax; a.selcap = p->selcap; a.start_id = p->start_id; a.end_id = p->end_id;
    a.proj_w = reinterpret_cast<const u16x8_t*>(p->proj_w); a.proj_b = p->proj_b; a.proj_h0 = p->proj_h0; a.proj_h1 = p->proj_h1;
    a.proj_k0 = p->proj_k0; a.proj_k1 = p->proj_h1 ? p->proj_k1 : 0;
    if (p->proj_w) {
        LAS_ARG(p->proj_h0 && p->proj_b && p->proj_k0 > 0 && (p->proj_k0 % 32) == 0 && (a.proj_k1 % 32) == 0,
                "las_beam_loop_step: proj_w needs proj_h0, proj_b and row widths that are multiples of 32");
        LAS_ARG(((p->beam + 15) / 16) * ((p->V + 15) / 16) <= BEAM_PROJ_TILES && p->beam * p->V <= BEAM_PROJ_TILES * 256,
                "las_beam_loop_step: the in-kernel projection handles ceil(beam / 16) * ceil(V / 16) <= %d tiles (beam %d, V %d)", BEAM_PROJ_TILES, p->beam, p->V);
        LAS_ARG((((uintptr_t)p->proj_h0 | (uintptr_t)p->proj_h1) & 15) == 0, "las_beam_loop_step: proj_h0 / proj_h1 must be 16-byte aligned");
    } else {
        LAS_ARG(p->logits, "las_beam_loop_step: null logits (and no projection)");
    }
    a.file_in = p->file_in; a.file_out = p->file_out; a.file_n = (long long)p->nutt * p->beam * p->file_width;
    int nfile = 0;
    if (p->file_in) {
        LAS_ARG(p->file_out && p->file_width > 0, "las_beam_loop_step: file_in without file_out / file_width");
        nfile = (int)((a.file_n + 4095) / 4096);
        if (nfile > 64) nfile = 64;
    }
    a.gather = (p->fold_gather && p->ntens > 0) ? 1 : 0;
    a.ntens = a.gather ? p->ntens : 0;
    for (int k = 0; k < a.ntens; ++k) {
        LAS_ARG(p->state_in[k] && p->state_out[k] && p->state_width[k] > 0, "las_beam_loop_step: bad state tensor %d", k);
        a.g_in[k] = p->state_in[k]; a.g_out[k] = p->state_out[k]; a.g_width[k] = p->state_width[k];
    }
    hipLaunchKernelGGL(beam_loop_kernel, dim3(p->nutt + nfile), dim3(256), 0, st, a);
    LAS_LAUNCHED();
    if (a.gather) return 0;                                            // gathered and counted inside the launch
    if (p->ntens > 0) {
        GatherDev g;
        g.ntens = p->ntens; g.src_row = p->src_row; g.step = p->step; g.nrows = p->nutt * p->beam;
        for (int k = 0; k < p->ntens; ++k) {
            LAS_ARG(p->state_in[k] && p->state_out[k] && p->state_width[k] > 0, "las_beam_loop_step: bad state tensor %d", k);
            g.in[k] = p->state_in[k]; g.out[k] = p->state_out[k]; g.width[k] = p->state_width[k];
        }
        if (g.nrows >= 512) hipLaunchKernelGGL(beam_gather_kernel, dim3(g.nrows), dim3(256), 0, st, g);                // ... and *step += 1
        else                hipLaunchKernelGGL(beam_gather_rt_kernel, dim3(g.nrows, p->ntens), dim3(256), 0, st, g);
        LAS_LAUNCHED();
    } else {
        hipLaunchKernelGGL(beam_advance_kernel, dim3(1), dim3(1), 0, st, p->step);
        LAS_LAUNCHED();
    }
    return 0;
}

// ----------------------------------------------------------------------------------------------------
// After the last step: the token ids of every retired hypothesis by walking the back-pointer records on the device (one thread per
// (utterance, selection slot); the reference keeps whole token lists in its BeamState objects instead, las/beam_search.py:38-45).
//   ids / rows [nutt * selcap, Umax]: token and global state row (utterance * beam + live slot) of the hypothesis at step p < len;
//   len [nutt * selcap]: tokens after SOS (0 = unused slot); score [nutt * selcap]: the running float32 sum.
// ----------------------------------------------------------------------------------------------------
struct BacktrackDev {
    const int *hist_parent, *hist_token, *hist_slot; const float* hist_score; const int *sel_t, *sel_j, *nsel;
    int nutt, beam, Umax, selcap;
    int *ids, *rows, *len; float* score;
};
__global__ __launch_bounds__(64) void beam_backtrack_kernel(BacktrackDev a) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= a.nutt * a.selcap) return;
    const int u = w / a.selcap, s = w - u * a.selcap;
    int n = a.nsel[u];
    if (n > a.selcap) n = a.selcap;
    if (s >= n) { a.len[w] = 0; a.score[w] = 0.f; return; }
    const int ts = a.sel_t[w], js = a.sel_j[w];
    int* ids = a.ids + (size_t)w * a.Umax;
    int* rows = a.rows + (size_t)w * a.Umax;
    int j = js;
    for (int tt = ts; ; --tt) {
        const size_t hb = ((size_t)tt * a.nutt + u) * a.beam;
        const int slot = a.hist_parent[hb + j];
        ids[tt] = a.hist_token[hb + j];
        rows[tt] = u * a.beam + slot;
        if (tt == 0) break;
        j = a.hist_slot[hb - (size_t)a.nutt * a.beam + slot];       // pick of step tt-1 that became live slot `slot`
    }
    a.len[w] = ts + 1;
    a.score[w] = a.hist_score[((size_t)ts * a.nutt + u) * a.beam + js];
}

extern "C" int las_beam_backtrack(const las_beam_loop_args* p, int* ids, int* rows, int* len, float* score, void* stream) {
    LAS_ARG(p && ids && rows && len && score, "las_beam_backtrack: null pointer");
    LAS_ARG(p->hist_parent && p->hist_token && p->hist_slot && p->hist_score && p->sel_t && p->sel_j && p->nsel, "las_beam_backtrack: null record pointer");
    LAS_ARG(p->nutt > 0 && p->beam > 0 && p->Umax > 0 && p->selcap > 0, "las_beam_backtrack: bad dims");
    BacktrackDev a;
    a.hist_parent = p->hist_parent; a.hist_token = p->hist_token; a.hist_slot = p->hist_slot; a.hist_score = p->hist_score;
    a.sel_t = p->sel_t; a.sel_j = p->sel_j; a.nsel = p->nsel;
    a.nutt = p->nutt; a.beam = p->beam; a.Umax = p->Umax; a.selcap = p->selcap;
    a.ids = ids; a.rows = rows; a.len = len; a.score = score;
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3(cdiv(p->nutt * p->selcap, 64)), dim3(64), 0, (hipStream_t)stream, a);
    LAS_LAUNCHED();
    return 0;
}
