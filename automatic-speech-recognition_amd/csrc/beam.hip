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

__global__ __launch_bounds__(256) void beam_step_kernel(const float* __restrict__ logits, const float* __restrict__ score,
                                                        const int* __restrict__ length, const int* __restrict__ nlive,
                                                        int beam, int V, int t, int start_id, int* __restrict__ out_parent,
                                                        int* __restrict__ out_token, float* __restrict__ out_score,
                                                        int* __restrict__ out_n) {
    __shared__ float s_norm[256], s_l[256];
    __shared__ int s_i[256], s_v[256];
    __shared__ BKey picks[64];
    __shared__ int s_count;
    const int u = blockIdx.x, tid = threadIdx.x;
    int nb = nlive[u];
    if (nb > beam) nb = beam;
    if (t == 0 && nb > 1) nb = 1;                       // las/beam_search.py:119
    const float* lg = logits + (size_t)u * beam * V;
    const float* sc = score + (size_t)u * beam;
    const int* ln = length + (size_t)u * beam;
    BKey last = {0.f, 0, 0.f, 0};
    int count = 0;
    for (int pick = 0; pick < beam; ++pick) {
        BKey best = {0.f, -1, 0.f, 0};
        bool has = false;
        const int ncand = nb * V;
        for (int idx = tid; idx < ncand; idx += 256) {
            const int i = idx / V;
            const int v = idx - i * V;
            if (t > 0 && v == start_id) continue;      // las/beam_search.py:127-128
            const float l = lg[idx];
            BKey k;
            k.norm = (sc[i] + l) / (float)(ln[i] + 1);  // float32 sum, float32 divide (las/beam_search.py:27,306)
            k.i = i; k.l = l; k.v = idx;                // v carries the flat candidate id until the very end
            if (!(k.norm == k.norm)) continue;          // NaN never ranks
            if (pick > 0 && !bless(k, last)) continue;
            if (!has || bless(best, k)) { best = k; has = true; }
        }
        // block reduce through LDS (tree over 256 slots)
        __syncthreads();
        s_norm[tid] = best.norm; s_i[tid] = has ? best.i : -1; s_l[tid] = best.l; s_v[tid] = best.v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const BKey x = {s_norm[tid], s_i[tid], s_l[tid], s_v[tid]};
                const BKey y = {s_norm[tid + off], s_i[tid + off], s_l[tid + off], s_v[tid + off]};
                if (y.i >= 0 && (x.i < 0 || bless(x, y))) {
                    s_norm[tid] = y.norm; s_i[tid] = y.i; s_l[tid] = y.l; s_v[tid] = y.v;
                }
            }
            __syncthreads();
        }
        best.norm = s_norm[0]; best.i = s_i[0]; best.l = s_l[0]; best.v = s_v[0];
        has = best.i >= 0;
        if (!has) break;                                 // uniform: every thread reads slot 0
        last = best;
        if (tid == 0) picks[pick] = best;
        count = pick + 1;
    }
    __syncthreads();
    if (tid == 0) { s_count = count; out_n[u] = count; }
    __syncthreads();
    for (int j = tid; j < count; j += 256) {             // ascending, best last (las/beam_search.py:310-312)
        const BKey k = picks[count - 1 - j];
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
    const float* logits;            // [nutt, beam, V] raw logits of this step (row = utterance * beam + live slot)
    float* score; int* length;      // [nutt, beam] live hypotheses: running float32 sum / tokens after SOS
    int* nlive; int* nsel; int* done;      // [nutt]
    const int* dec_step;            // [nutt] step bound int(audiolen * convert_rate) (las/beam_search.py:78)
    int* step;                      // [1] device-resident step counter t (read by every workgroup, advanced by the gather kernel)
    int *hist_parent, *hist_token, *hist_slot; float* hist_score; int* hist_n;   // [Umax, nutt, beam] / [Umax, nutt]
    int *sel_t, *sel_j;             // [nutt, selcap] retired hypotheses in the reference's append order
    int* src_row;                   // [nutt, beam] out: live slot k of the next step continues row src_row (global row index)
    int* next_token;                // [nutt * beam] out: token entering the next step for each live slot
    int nutt, beam, V, Umax, selcap, start_id, end_id;
};

__global__ __launch_bounds__(256) void beam_loop_kernel(BeamLoopDev a) {
    __shared__ float s_norm[256], s_l[256];
    __shared__ int s_i[256], s_v[256];
    __shared__ BKey picks[64];
    const int u = blockIdx.x, tid = threadIdx.x, beam = a.beam, V = a.V;
    const int t = a.step[0];
    int* hn = a.hist_n + (size_t)t * a.nutt + u;
    if (t >= a.Umax) return;
    if (a.done[u] || t >= a.dec_step[u]) {               // retired utterance: nothing to do (its rows compute ignored garbage)
        if (tid == 0) { *hn = 0; a.done[u] = 1; a.nlive[u] = 0; }
        for (int k = tid; k < beam; k += 256) { a.src_row[(size_t)u * beam + k] = u * beam; }
        return;
    }
    int nb = a.nlive[u];
    if (nb > beam) nb = beam;
    if (t == 0 && nb > 1) nb = 1;                       // las/beam_search.py:119
    const float* lg = a.logits + (size_t)u * beam * V;
    float* sc = a.score + (size_t)u * beam;
    int* ln = a.length + (size_t)u * beam;
    BKey last = {0.f, 0, 0.f, 0};
    int count = 0;
    for (int pick = 0; pick < beam; ++pick) {
        BKey best = {0.f, -1, 0.f, 0};
        bool has = false;
        const int ncand = nb * V;
        for (int idx = tid; idx < ncand; idx += 256) {
            const int i = idx / V;
            const int v = idx - i * V;
            if (t > 0 && v == a.start_id) continue;    // las/beam_search.py:127-128
            const float l = lg[idx];
            BKey k;
            k.norm = (sc[i] + l) / (float)(ln[i] + 1);  // float32 sum, float32 divide (las/beam_search.py:27,306)
            k.i = i; k.l = l; k.v = idx;
            if (!(k.norm == k.norm)) continue;
            if (pick > 0 && !bless(k, last)) continue;
            if (!has || bless(best, k)) { best = k; has = true; }
        }
        __syncthreads();
        s_norm[tid] = best.norm; s_i[tid] = has ? best.i : -1; s_l[tid] = best.l; s_v[tid] = best.v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const BKey x = {s_norm[tid], s_i[tid], s_l[tid], s_v[tid]};
                const BKey y = {s_norm[tid + off], s_i[tid + off], s_l[tid + off], s_v[tid + off]};
                if (y.i >= 0 && (x.i < 0 || bless(x, y))) {
                    s_norm[tid] = y.norm; s_i[tid] = y.i; s_l[tid] = y.l; s_v[tid] = y.v;
                }
            }
            __syncthreads();
        }
        best.norm = s_norm[0]; best.i = s_i[0]; best.l = s_l[0]; best.v = s_v[0];
        has = best.i >= 0;
        if (!has) break;
        last = best;
        if (tid == 0) picks[pick] = best;
        count = pick + 1;
    }
    __syncthreads();
    if (tid == 0) {
        // the reference's bookkeeping (las/beam_search.py:147-152), in its iteration order = ascending rank (best last)
        float nsc[64]; int nln[64];
        int nl = 0, ns = a.nsel[u];
        const size_t hb = ((size_t)t * a.nutt + u) * beam;
        for (int j = 0; j < count; ++j) {
            const BKey k = picks[count - 1 - j];
            const int v = k.v - k.i * V;
            const float news = sc[k.i] + k.l;
            const int newl = ln[k.i] + 1;
            a.hist_parent[hb + j] = k.i; a.hist_token[hb + j] = v; a.hist_score[hb + j] = news;
            if (v == a.end_id) {
                if (ns < a.selcap) { a.sel_t[(size_t)u * a.selcap + ns] = t; a.sel_j[(size_t)u * a.selcap + ns] = j; }
                ++ns;
            } else {
                nsc[nl] = news; nln[nl] = newl;
                a.hist_slot[hb + nl] = j;                         // live slot nl of step t+1 is pick j of step t
                a.src_row[(size_t)u * beam + nl] = u * beam + k.i;
                a.next_token[(size_t)u * beam + nl] = v;
                ++nl;
            }
        }
        *hn = count;
        for (int k = nl; k < beam; ++k) { a.src_row[(size_t)u * beam + k] = u * beam; a.next_token[(size_t)u * beam + k] = a.start_id; }
        for (int k = 0; k < nl; ++k) { sc[k] = nsc[k]; ln[k] = nln[k]; }
        const bool exhausted = (t + 1 == a.dec_step[u]);
        if (exhausted) {                                           // `if t == dec_step: selected.extend(beam_set)` (:155-156)
            for (int k = 0; k < nl; ++k) {
                if (ns < a.selcap) { a.sel_t[(size_t)u * a.selcap + ns] = t; a.sel_j[(size_t)u * a.selcap + ns] = a.hist_slot[hb + k]; }
                ++ns;
            }
        }
        const bool fin = exhausted || ns >= beam || nl == 0;        // loop condition of :94 (+ no live hypothesis left)
        a.nsel[u] = ns;
        a.nlive[u] = fin ? 0 : nl;
        if (fin) a.done[u] = 1;
    }
}

// rows of `ntens` state tensors follow their hypotheses: out[k][r] = in[k][src_row[r]] (r = global row), then t += 1.
struct GatherDev { const float* in[16]; float* out[16]; int width[16]; int ntens; const int* src_row; int* step; int nrows; };
__global__ __launch_bounds__(256) void beam_gather_kernel(GatherDev g) {
    const int r = blockIdx.x, k = blockIdx.y;
    const int src = g.src_row[r];
    const int w = g.width[k];
    const float* ip = g.in[k] + (size_t)src * w;
    float* op = g.out[k] + (size_t)r * w;
    for (int i = threadIdx.x; i < w; i += 256) op[i] = ip[i];
}
__global__ void beam_advance_kernel(int* step) { step[0] += 1; }

extern "C" int las_beam_loop_step(const las_beam_loop_args* p, void* stream) {
    LAS_ARG(p, "las_beam_loop_step: null args");
    LAS_ARG(p->logits && p->score && p->length && p->nlive && p->nsel && p->done && p->dec_step && p->step, "las_beam_loop_step: null state pointer");
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
    hipLaunchKernelGGL(beam_loop_kernel, dim3(p->nutt), dim3(256), 0, st, a);
    LAS_LAUNCHED();
    if (p->ntens > 0) {
        GatherDev g;
        g.ntens = p->ntens; g.src_row = p->src_row; g.step = p->step; g.nrows = p->nutt * p->beam;
        for (int k = 0; k < p->ntens; ++k) {
            LAS_ARG(p->state_in[k] && p->state_out[k] && p->state_width[k] > 0, "las_beam_loop_step: bad state tensor %d", k);
            g.in[k] = p->state_in[k]; g.out[k] = p->state_out[k]; g.width[k] = p->state_width[k];
        }
        hipLaunchKernelGGL(beam_gather_kernel, dim3(g.nrows, p->ntens), dim3(256), 0, st, g);
        LAS_LAUNCHED();
    }
    hipLaunchKernelGGL(beam_advance_kernel, dim3(1), dim3(1), 0, st, p->step);
    LAS_LAUNCHED();
    return 0;
}
