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
