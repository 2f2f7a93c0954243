// loss_opt.hip -- K8 (label-smoothed masked cross-entropy, fwd + grad in one pass) and
// K9 (clip_by_global_norm + TF-style Adam on one flat bucket).  HBM-bound streaming kernels:
// vectorised, fixed-order reductions (no atomics -> bit-reproducible training steps).
#include "las_common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// K8: LAS._get_loss (reference las/las.py:320-333) + label_smoothing (las/utils.py:5-12)
//   soft = (1-eps)*onehot + eps/V ;  ce = -sum_k soft_k * log_softmax(l)_k ; mask = (y != 0)
//   d l_k = scale * mask * (softmax_k - soft_k)
// one wave per (b,t) row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, long long sb, long long st,
                                                      const int* __restrict__ y, int ldy, int B, int U, int V, float eps,
                                                      int smooth, const float* __restrict__ scale_ptr,
                                                      float* __restrict__ dlogits, float* __restrict__ row_ce,
                                                      float* __restrict__ row_mask) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * U) return;
    const int b = (int)(row / U), t = (int)(row % U);
    const float* lp = logits + b * sb + t * st;
    const int label = y[(long long)b * ldy + t];
    const float mask = (label != 0 || (smooth & 2)) ? 1.f : 0.f;     // bit 1 of `smooth`: every position counts (RNNLM loss)
    float m = -INFINITY, lsum = 0.f;
    for (int k = lane; k < V; k += 64) { const float l = lp[k]; m = fmaxf(m, l); lsum += l; }
    m = wave_max(m);
    lsum = wave_sum(lsum);
    float se = 0.f;
    for (int k = lane; k < V; k += 64) se += expf(lp[k] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    const float e = (smooth & 1) ? eps : 0.f;
    const float ly = (label >= 0 && label < V) ? lp[label] : 0.f;
    const float ce = lse - (1.f - e) * ly - (e / V) * lsum;
    if (lane == 0) { row_ce[row] = ce * mask; row_mask[row] = mask; }
    if (dlogits) {
        const float sc = scale_ptr[0] * mask;
        float* dp = dlogits + b * sb + t * st;
        for (int k = lane; k < V; k += 64) {
            const float p = expf(lp[k] - lse);
            const float soft = (k == label ? (1.f - e) : 0.f) + e / V;
            dp[k] = sc * (p - soft);
        }
    }
}

// fixed-order sum of n values (single workgroup), out[0] += sum(a), out[1] += sum(b);  scale != NULL (bit 2 of las_ce_loss's `smooth`):
// out[0] = sum(a), out[1] = sum(b), out[2] = sum(a) * scale[0] -- the caller neither zeroes `out` nor multiplies afterwards
__global__ __launch_bounds__(1024) void sum2_kernel(const float* __restrict__ a, const float* __restrict__ b2, long long n,
                                                    float* __restrict__ out, const float* __restrict__ scale) {
    __shared__ float red[2][16];
    float s0 = 0.f, s1 = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) { s0 += a[i]; if (b2) s1 += b2[i]; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int i = 0; i < 16; ++i) { t0 += red[0][i]; t1 += red[1][i]; }
        if (scale) { out[0] = t0; out[1] = b2 ? t1 : 0.f; out[2] = t0 * scale[0]; }
        else { out[0] += t0; if (b2) out[1] += t1; }
    }
}

extern "C" size_t las_ce_loss_workspace_bytes(int B, int U) { return (size_t)2 * B * U * sizeof(float) + 256; }

// The same row with the logits kept in REGISTERS (round 5): a subword vocabulary (V = 5000, BASELINE configs[3]) made the three passes of
// ce_rows_kernel four trips through memory -- 183 MB of logits read three times and written once, 170 us on the dependency chain between the
// two Speller loops; here a lane loads its NV = ceil(V / 64) values once (all loads in flight together), the statistics and the gradient come
// from registers.  Same arithmetic in the same order: bit-identical to ce_rows_kernel.
template <int NV>
__global__ __launch_bounds__(256) void ce_rows_reg_kernel(const float* __restrict__ logits, long long sb, long long st,
                                                          const int* __restrict__ y, int ldy, int B, int U, int V, float eps,
                                                          int smooth, const float* __restrict__ scale_ptr,
                                                          float* __restrict__ dlogits, float* __restrict__ row_ce,
                                                          float* __restrict__ row_mask) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * U) return;
    const int b = (int)(row / U), t = (int)(row % U);
    const float* lp = logits + b * sb + t * st;
    const int label = y[(long long)b * ldy + t];
    const float mask = (label != 0 || (smooth & 2)) ? 1.f : 0.f;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int k = lane + 64 * i; v[i] = lp[k < V ? k : V - 1]; }
    float m = -INFINITY, lsum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (lane + 64 * i < V) { m = fmaxf(m, v[i]); lsum += v[i]; }
    m = wave_max(m);
    lsum = wave_sum(lsum);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (lane + 64 * i < V) se += expf(v[i] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    const float e = (smooth & 1) ? eps : 0.f;
    const float ly = (label >= 0 && label < V) ? lp[label] : 0.f;
    const float ce = lse - (1.f - e) * ly - (e / V) * lsum;
    if (lane == 0) { row_ce[row] = ce * mask; row_mask[row] = mask; }
    if (dlogits) {
        const float sc = scale_ptr[0] * mask;
        float* dp = dlogits + b * sb + t * st;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = lane + 64 * i;
            if (k < V) {
                const float p = expf(v[i] - lse);
                const float soft = (k == label ? (1.f - e) : 0.f) + e / V;
                dp[k] = sc * (p - soft);
            }
        }
    }
}

extern "C" int las_ce_loss(const float* logits, long long sb, long long st, const int* y, int ldy, int B, int U, int V,
                           float epsilon, int smooth, float* sums, const float* scale_ptr, float* dlogits, void* ws,
                           size_t ws_bytes, void* stream) {
    LAS_ARG(logits && y && sums && B > 0 && U > 0 && V > 0 && ldy >= U, "las_ce_loss: bad arguments");
    LAS_ARG(!dlogits || scale_ptr, "las_ce_loss: dlogits needs scale_ptr");
    LAS_ARG(ws && ws_bytes >= las_ce_loss_workspace_bytes(B, U), "las_ce_loss: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* row_ce = (float*)ws;
    float* row_mask = row_ce + (size_t)B * U;
    const dim3 cg(cdiv((long long)B * U, 4));
    if (V > 1024 && V <= 64 * 40)
        hipLaunchKernelGGL(ce_rows_reg_kernel<40>, cg, dim3(256), 0, s, logits, sb, st, y, ldy, B, U, V, epsilon, smooth, scale_ptr, dlogits, row_ce, row_mask);
    else if (V > 1024 && V <= 64 * 80)
        hipLaunchKernelGGL(ce_rows_reg_kernel<80>, cg, dim3(256), 0, s, logits, sb, st, y, ldy, B, U, V, epsilon, smooth, scale_ptr, dlogits, row_ce, row_mask);
    else if (V > 1024 && V <= 64 * 128)
        hipLaunchKernelGGL(ce_rows_reg_kernel<128>, cg, dim3(256), 0, s, logits, sb, st, y, ldy, B, U, V, epsilon, smooth, scale_ptr, dlogits, row_ce, row_mask);
    else
    hipLaunchKernelGGL(ce_rows_kernel, cg, dim3(256), 0, s, logits, sb, st, y, ldy, B, U, V,
                       epsilon, smooth, scale_ptr, dlogits, row_ce, row_mask);
    LAS_LAUNCHED();
    LAS_ARG(!(smooth & 4) || scale_ptr, "las_ce_loss: bit 2 of smooth (write sums, sums[2] = scaled loss) needs scale_ptr");
    hipLaunchKernelGGL(sum2_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_ce, (const float*)row_mask, (long long)B * U, sums,
                       (smooth & 4) ? scale_ptr : (const float*)nullptr);
    LAS_LAUNCHED();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K9: tf.clip_by_global_norm + AdamOptimizer.apply_gradients (reference las/las.py:272-283)
// ------------------------------------------------------------------------------------------------
constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (long long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 16; ++i) t += red[i]; out[0] = t; }
}

extern "C" size_t las_sumsq_workspace_bytes(long long n) { (void)n; return SS_BLOCKS * sizeof(float); }

extern "C" int las_sumsq(const float* g, long long n, float* out, void* ws, size_t ws_bytes, void* stream) {
    LAS_ARG(g && out && n >= 0, "las_sumsq: bad arguments");
    LAS_ARG(ws && ws_bytes >= las_sumsq_workspace_bytes(n), "las_sumsq: workspace too small");
    LAS_ARG((((uintptr_t)g) & 15) == 0, "las_sumsq: g must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SS_BLOCKS), dim3(256), 0, s, g, n, (float*)ws);
    LAS_LAUNCHED();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, s, (const float*)ws, SS_BLOCKS, out);
    LAS_LAUNCHED();
    return 0;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        const float* __restrict__ sumsq, float clip, float lr_t, float b1,
                                                        float b2, float eps, const int* __restrict__ status,
                                                        const float* __restrict__ guard, int* __restrict__ applied) {
    // a recurrent sweep of this step reported a time-out (status) -- on this rank or, through the all-reduced guard slot, on
    // any rank: the gradients are garbage, leave theta / m / v alone (uniform branch: every thread reads the same words)
    if ((status && status[0] != 0) || (guard && guard[0] != 0.f)) return;
    if (applied && blockIdx.x == 0 && threadIdx.x == 0) applied[0] += 1;      // (one launch at a time per parameter bucket: no atomic needed)
    float gs = 1.f;
    if (clip > 0.f) {   // g * clip / max(norm, clip)   
        const float nrm = sqrtf(sumsq[0]);
        gs = clip / fmaxf(nrm, clip);
    }
    const long long n4 = n / 4;
    float4* t4 = reinterpret_cast<float4*>(theta);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 th = t4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#define ADAM1(c)                                                \
        { const float gc = gg.c * gs;                           \
          mm.c = b1 * mm.c + (1.f - b1) * gc;                   \
          vv.c = b2 * vv.c + (1.f - b2) * gc * gc;              \
          th.c -= lr_t * mm.c / (sqrtf(vv.c) + eps); }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        t4[i] = th; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gc = g[i] * gs;
            const float mm = b1 * m[i] + (1.f - b1) * gc;
            const float vv = b2 * v[i] + (1.f - b2) * gc * gc;
            m[i] = mm; v[i] = vv;
            theta[i] -= lr_t * mm / (sqrtf(vv) + eps);
        }
}

extern "C" int las_clip_adam(float* theta, const float* g, float* m, float* v, long long n, const float* sumsq, float clip,
                             float lr_t, float beta1, float beta2, float eps, const int* status, const float* guard, int* applied, void* stream) {
    LAS_ARG(theta && g && m && v && n >= 0, "las_clip_adam: bad arguments");
    LAS_ARG(clip <= 0.f || sumsq, "las_clip_adam: clipping needs sumsq");
    LAS_ARG(((((uintptr_t)theta) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
            "las_clip_adam: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    int nb = cdiv(n / 4 + 1, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, theta, g, m, v, n, sumsq, clip, lr_t,
                       beta1, beta2, eps, status, guard, applied);
    LAS_LAUNCHED();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BasicLSTMCell gate math for ONE step (char RNNLM shallow fusion, reference lang/char_rnn_model.py:57-66,
// las/beam_search.py:226-236): z = [x,h].kernel + bias comes from las_gemm; here
//   i,j,f,o = split(z,4); c' = c*sigmoid(f+fb) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_pointwise_kernel(const float* __restrict__ z, const float* __restrict__ xrows,
                                                             const int* __restrict__ ids, int id_shift, const float* __restrict__ c_prev, int N,
                                                             int H, float fb, float* __restrict__ c_out, float* __restrict__ h_out) {
    const long long total = (long long)N * H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long n = idx / H;
        const int u = (int)(idx % H);
        const float* zr = z + n * 4 * H;
        float zi = zr[u], zj = zr[H + u], zf = zr[2 * H + u], zo = zr[3 * H + u];
        if (xrows) {                                     // + the input half of a one-hot input: row max(ids[n] - id_shift, 0) of W_x
            int id = ids[n] - id_shift;
            if (id < 0) id = 0;
            const float* xr = xrows + (long long)id * 4 * H;
            zi += xr[u]; zj += xr[H + u]; zf += xr[2 * H + u]; zo += xr[3 * H + u];
        }
        const float gi = sigmoid_acc(zi), gj = tanh_acc(zj), gf = sigmoid_acc(zf + fb), go = sigmoid_acc(zo);
        const float c = c_prev[idx] * gf + gi * gj;
        c_out[idx] = c;
        h_out[idx] = tanh_acc(c) * go;
    }
}

extern "C" int las_lstm_pointwise(const float* z, const float* c_prev, int N, int H, float forget_bias, float* c_out, float* h_out,
                                  void* stream) {
    LAS_ARG(z && c_prev && c_out && h_out && N > 0 && H > 0, "las_lstm_pointwise: bad arguments");
    int nb = cdiv((long long)N * H, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, nullptr, nullptr, 0, c_prev, N, H, forget_bias, c_out,
                       h_out);
    LAS_LAUNCHED();
    return 0;
}

// the same with the input half of a ONE-HOT input added on the way in: z[n] + xrows[max(ids[n] - id_shift, 0)] (xrows [V, 4H] = the input
// rows of the TF cell kernel; the shift / clamp is the LAS-id -> LM-id map of the shallow fusion, las/beam_search.py:109-116)
extern "C" int las_lstm_pointwise_rows(const float* z, const float* xrows, const int* ids, int id_shift, const float* c_prev, int N, int H,
                                       float forget_bias, float* c_out, float* h_out, void* stream) {
    LAS_ARG(z && xrows && ids && c_prev && c_out && h_out && N > 0 && H > 0, "las_lstm_pointwise_rows: bad arguments");
    int nb = cdiv((long long)N * H, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, xrows, ids, id_shift, c_prev, N, H, forget_bias,
                       c_out, h_out);
    LAS_LAUNCHED();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// R1 for a whole block of rows in ONE launch: z = [x ; h] . kernel + bias and the gate math above (lang/char_rnn_model.py:57-66),
// for the beam search's LM step (M = utterances x beam rows, a few hundred).  r3 decode trace: per LM layer the step ran two
// skinny-M products (2048 workgroups of one 16 x 16 tile each -- a shape made for M <= 48) and a gate kernel, 14-23 us; here a
// workgroup owns 32 rows x 16 units: the four gates' column tiles of those units (weights as las_gemm_skinny_pack fragments,
// TF gate order i, j, f, o = column blocks of H), K = I + H contracted in chunks staged through LDS as bf16, 8 waves = 4 gates x
// 2 row tiles, gates exchanged through LDS, then c' / h' written directly.  A one-hot first layer passes ids / xrows instead of x.
// ------------------------------------------------------------------------------------------------
#include "lstm_cell_rows.h"

template <bool FAST, bool XBF>
__global__ __launch_bounds__(512) void lstm_cell_rows_kernel(LstmCellLaunch a) {
    __shared__ __attribute__((aligned(16))) unsigned short As[32 * LC_LD];
    __shared__ float gates[2][4][64][4];
    lstm_cell_rows_body<FAST, XBF, true>(a, As, gates, blockIdx.x, blockIdx.y * 32);
}
// two independent cells in one grid: blockIdx.z = 0 the Speller's (fast, bf16 x), 1 the LM's (exact, fp32 / one-hot input)
__global__ __launch_bounds__(512, 2) void lstm_cell_rows_pair_kernel(LstmCellLaunch a, LstmCellLaunch b) {
    __shared__ __attribute__((aligned(16))) unsigned short As[32 * LC_LD];
    __shared__ float gates[2][4][64][4];
    if (blockIdx.z == 0) lstm_cell_rows_body<true, true, false>(a, As, gates, blockIdx.x, blockIdx.y * 32);
    else                 lstm_cell_rows_body<false, false, false>(b, As, gates, blockIdx.x, blockIdx.y * 32);
}

// the pipelined body (lstm_cell_rows_big_body): 128, 64 or 32 rows per workgroup, M >= LB_MIN_ROWS
template <bool FAST, bool XBF, int ROWS>
__global__ __launch_bounds__(512) void lstm_cell_rows_big_kernel(LstmCellLaunch a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    lstm_cell_rows_big_body<FAST, XBF, ROWS>(a, lb_smem, blockIdx.x, blockIdx.y * ROWS);
}
// ... and two independent cells in one grid (blockIdx.z as in the pair kernel above).  At 128 rows and 240 VGPRs a CU holds one workgroup
// and the two problems run one behind the other -- but without the unpipelined 32-row bodies' four rounds per problem (r5, M = 1024: 29 us
// against 35); at 32 / 64 rows the body needs few registers and both problems share a CU.
template <bool BBF, int ROWS>       // the second problem's rows: bf16 (the LM's state copies, h_out_bf16 of the step before) or fp32
__global__ __launch_bounds__(512) void lstm_cell_rows_big_pair_kernel(LstmCellLaunch a, LstmCellLaunch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    if (blockIdx.z == 0) lstm_cell_rows_big_body<true, true, ROWS>(a, lb_smem, blockIdx.x, blockIdx.y * ROWS);
    else                 lstm_cell_rows_big_body<false, BBF, ROWS>(b, lb_smem, blockIdx.x, blockIdx.y * ROWS);
}
constexpr int LB_MIN_ROWS = 128;          // below: the unpipelined 32-row body (a training batch's per-step fallback: M <= 48)
static inline int lb_rows_for(int M) { return M > 512 ? 128 : (M > 256 ? 64 : 32); }   // the largest tile that still gives 256 CUs a workgroup each at H = 512
// the pipelined body takes ONE element type for all its rows: bf16 (x_bf16 / h_bf16) or fp32
static inline int lb_rows_type(const LstmCellLaunch& a) {          // 1 bf16, 0 fp32, -1 mixed
    const int xt = a.x ? (a.x_bf16 ? 1 : 0) : -1, ht = a.h ? (a.h_bf16 ? 1 : 0) : -1;
    if (xt >= 0 && ht >= 0 && xt != ht) return -1;
    return xt >= 0 ? xt : (ht >= 0 ? ht : 0);
}
static int g_lb_rows = 0, g_lb_pair_rows = 0;        // 0: the launcher's choice
#ifdef LAS_DEV   // development builds only (make prof): A/B switches, not part of the shipping library
extern "C" void las_dev_lstm_cell_rows(int rows) { g_lb_rows = (rows == 32 || rows == 64 || rows == 128) ? rows : 0; }
extern "C" void las_dev_lstm_cell_pair_rows(int rows) { g_lb_pair_rows = (rows == 32 || rows == 64 || rows == 128) ? rows : 0; }
#endif
static inline bool lb_serves(const LstmCellLaunch& a) { return a.M >= LB_MIN_ROWS && lb_rows_type(a) >= 0; }
template <typename K>
static int lb_attr(K kern, int rows) { return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lb_lds_bytes(rows)); }
template <int ROWS>
static int lb_launch(const LstmCellLaunch& a, hipStream_t st) {
    static int attr = lb_attr(lstm_cell_rows_big_kernel<true, true, ROWS>, ROWS) | lb_attr(lstm_cell_rows_big_kernel<true, false, ROWS>, ROWS) |
                      lb_attr(lstm_cell_rows_big_kernel<false, true, ROWS>, ROWS) | lb_attr(lstm_cell_rows_big_kernel<false, false, ROWS>, ROWS);
    if (attr != 0) { las_set_error("hipFuncSetAttribute(lstm_cell_rows_big) failed: %d", attr); return attr; }
    const dim3 gb(a.H / 16, cdiv(a.M, ROWS));
    constexpr int LDS = lb_lds_bytes(ROWS);
    const bool abf = lb_rows_type(a) == 1;
    if (a.fast && abf) hipLaunchKernelGGL((lstm_cell_rows_big_kernel<true, true, ROWS>), gb, dim3(512), LDS, st, a);
    else if (a.fast)   hipLaunchKernelGGL((lstm_cell_rows_big_kernel<true, false, ROWS>), gb, dim3(512), LDS, st, a);
    else if (abf)      hipLaunchKernelGGL((lstm_cell_rows_big_kernel<false, true, ROWS>), gb, dim3(512), LDS, st, a);
    else               hipLaunchKernelGGL((lstm_cell_rows_big_kernel<false, false, ROWS>), gb, dim3(512), LDS, st, a);
    LAS_LAUNCHED();
    return 0;
}
template <int ROWS>
static int lb_launch2(const LstmCellLaunch& a, const LstmCellLaunch& b, hipStream_t st) {
    static int attr = lb_attr(lstm_cell_rows_big_pair_kernel<false, ROWS>, ROWS) | lb_attr(lstm_cell_rows_big_pair_kernel<true, ROWS>, ROWS);
    if (attr != 0) { las_set_error("hipFuncSetAttribute(lstm_cell_rows_big_pair) failed: %d", attr); return attr; }
    const int bx = (a.H > b.H ? a.H : b.H) / 16, by = cdiv(a.M > b.M ? a.M : b.M, ROWS);
    constexpr int LDS = lb_lds_bytes(ROWS);
    if (lb_rows_type(b) == 1) hipLaunchKernelGGL((lstm_cell_rows_big_pair_kernel<true, ROWS>), dim3(bx, by, 2), dim3(512), LDS, st, a, b);
    else                      hipLaunchKernelGGL((lstm_cell_rows_big_pair_kernel<false, ROWS>), dim3(bx, by, 2), dim3(512), LDS, st, a, b);
    LAS_LAUNCHED();
    return 0;
}

int las_lstm_cell_check(const LstmCellLaunch& a) {
    LAS_ARG((a.x || a.h) && a.bias && a.c_prev && a.c_out && a.h_out && a.M > 0 && a.H > 0, "las_lstm_cell_rows: bad arguments");
    LAS_ARG(!(a.x && a.xrows), "las_lstm_cell_rows: x (dense input rows) and xrows (one-hot input) are exclusive");
    LAS_ARG(!a.x || (a.Wx && a.I > 0 && (a.I % 32) == 0 && (a.ldx % (a.x_bf16 ? 8 : 4)) == 0 && (((uintptr_t)a.x) & 15) == 0),
            "las_lstm_cell_rows: x needs I %% 32 == 0, ldx %% 4 == 0 (8 for bf16), 16-byte alignment and Wx_packed");
    LAS_ARG(!a.xrows || a.ids, "las_lstm_cell_rows: xrows without ids");
    LAS_ARG(!a.h || (a.Wh && (a.ldh % 4) == 0 && (((uintptr_t)a.h) & 15) == 0), "las_lstm_cell_rows: h needs ldh %% 4 == 0, 16-byte alignment and Wh_packed");
    LAS_ARG((a.H % 32) == 0, "las_lstm_cell_rows: needs H %% 32 == 0");
    LAS_ARG(!(a.h_bf16 || a.h_out_bf16) || lb_serves(a),
            "las_lstm_cell_rows: bf16 h / h_out_bf16 are served by the pipelined body only (M >= %d, one element type for x and h)", LB_MIN_ROWS);
    LAS_ARG(!a.h_bf16 || (a.ldh % 8) == 0, "las_lstm_cell_rows: bf16 h needs ldh %% 8 == 0");
    return 0;
}

int las_lstm_cell_rows_launch(const LstmCellLaunch& a, hipStream_t st) {
    if (int rc = las_lstm_cell_check(a)) return rc;
    // (fp32 rows at <= 256 rows of x AND h -- the LM's second layer in a 16-utterance search: the unpipelined body's two 512-column chunks beat
    //  eight pipelined 128-column chunks of 32 rows, 8.1 against 9.4 us)
    if (lb_serves(a) && !(a.M <= 256 && lb_rows_type(a) == 0 && a.x && a.h && !a.h_out_bf16)) {
        const int rows = g_lb_rows ? g_lb_rows : lb_rows_for(a.M);
        return rows == 128 ? lb_launch<128>(a, st) : (rows == 64 ? lb_launch<64>(a, st) : lb_launch<32>(a, st));
    }
    const dim3 grid(a.H / 16, cdiv(a.M, 32));
    if (a.fast && a.x_bf16) hipLaunchKernelGGL((lstm_cell_rows_kernel<true, true>), grid, dim3(512), 0, st, a);
    else if (a.fast)        hipLaunchKernelGGL((lstm_cell_rows_kernel<true, false>), grid, dim3(512), 0, st, a);
    else if (a.x_bf16)      hipLaunchKernelGGL((lstm_cell_rows_kernel<false, true>), grid, dim3(512), 0, st, a);
    else                    hipLaunchKernelGGL((lstm_cell_rows_kernel<false, false>), grid, dim3(512), 0, st, a);
    LAS_LAUNCHED();
    return 0;
}

int las_lstm_cell_rows_launch2(const LstmCellLaunch& a, const LstmCellLaunch& b, hipStream_t st) {
    if (int rc = las_lstm_cell_check(a)) return rc;
    if (int rc = las_lstm_cell_check(b)) return rc;
    if (!(a.fast && a.x_bf16 && !b.fast && !b.x_bf16 && !b.h_bf16) && !(lb_serves(a) && lb_rows_type(a) == 1 && a.fast && !b.fast && lb_serves(b))) {   // not a pair a kernel is compiled for: two launches
        if (int rc = las_lstm_cell_rows_launch(a, st)) return rc;
        return las_lstm_cell_rows_launch(b, st);
    }
    if (lb_serves(a) && lb_rows_type(a) == 1 && lb_serves(b)) {
        // (two problems: 64-row tiles also at 1024 rows -- at 100-120 VGPRs both problems' workgroups share a CU and cover each other's
        //  round trips, 18.1-18.4 against 18.6-18.7 ms per 211-step search; at 128 rows and 140-180 VGPRs they run one behind the other)
        const int mm = a.M > b.M ? a.M : b.M;
        const int rows = g_lb_pair_rows ? g_lb_pair_rows : (mm > 256 ? 64 : 32);
        return rows == 128 ? lb_launch2<128>(a, b, st) : (rows == 64 ? lb_launch2<64>(a, b, st) : lb_launch2<32>(a, b, st));
    }
    const int gx = (a.H > b.H ? a.H : b.H) / 16, gy = cdiv(a.M > b.M ? a.M : b.M, 32);
    hipLaunchKernelGGL(lstm_cell_rows_pair_kernel, dim3(gx, gy, 2), dim3(512), 0, st, a, b);
    LAS_LAUNCHED();
    return 0;
}

extern "C" int las_lstm_cell_rows_args(const las_lstm_cell_args* a, void* stream) {
    LAS_ARG(a, "las_lstm_cell_rows_args: null args");
    return las_lstm_cell_rows_launch(*a, (hipStream_t)stream);
}

extern "C" int las_lstm_cell_rows(const float* x, int ldx, int I, const int* ids, int id_shift, const float* xrows, const float* h, int ldh,
                                  const void* Wx_packed, const void* Wh_packed, const float* bias, const float* c_prev, int M, int H,
                                  float forget_bias, float* c_out, float* h_out, void* stream) {
    LAS_ARG(h && Wh_packed, "las_lstm_cell_rows: h and Wh_packed are required");
    LAS_ARG((x != nullptr) != (xrows != nullptr), "las_lstm_cell_rows: exactly one of x (dense input rows) and xrows (one-hot input) must be given");
    LstmCellLaunch a;
    a.x = x; a.x_bf16 = 0; a.ldx = ldx; a.I = x ? I : 0; a.ids = ids; a.id_shift = id_shift; a.xrows = xrows; a.h = h; a.ldh = ldh;
    a.Wx = Wx_packed; a.Wh = Wh_packed; a.bias = bias; a.c_prev = c_prev; a.fb = forget_bias; a.c_out = c_out; a.h_out = h_out;
    a.gates_out = nullptr; a.M = M; a.H = H; a.fast = 0; a.h_bf16 = 0; a.h_out_bf16 = nullptr;
    return las_lstm_cell_rows_launch(a, (hipStream_t)stream);
}


// gradient of the BasicLSTMCell gate math (R1 training, reference lang/char_rnn_model.py:54-66,177-190):
//   given z (pre-activations [N,4H], i,j,f,o), c_prev, dh (gradient w.r.t. h') and dc_in (gradient w.r.t. c' arriving from step t+1):
//   dz [N,4H] and dc_prev [N,H].   (dc_in may be NULL = zeros.)
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                                                 const float* __restrict__ dh, const float* __restrict__ dc_in, int N, int H,
                                                                 float fb, float* __restrict__ dz, float* __restrict__ dc_prev) {
    const long long total = (long long)N * H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long n = idx / H;
        const int u = (int)(idx % H);
        const float* zr = z + n * 4 * H;
        float* dzr = dz + n * 4 * H;
        const float gi = sigmoid_acc(zr[u]), gj = tanh_acc(zr[H + u]), gf = sigmoid_acc(zr[2 * H + u] + fb), go = sigmoid_acc(zr[3 * H + u]);
        const float cp = c_prev[idx];
        const float c = cp * gf + gi * gj;
        const float tc = tanh_acc(c);
        const float g = dh[idx];
        const float dc = (dc_in ? dc_in[idx] : 0.f) + g * go * (1.f - tc * tc);
        dzr[u] = dc * gj * gi * (1.f - gi);
        dzr[H + u] = dc * gi * (1.f - gj * gj);
        dzr[2 * H + u] = dc * cp * gf * (1.f - gf);
        dzr[3 * H + u] = g * tc * go * (1.f - go);
        dc_prev[idx] = dc * gf;
    }
}

extern "C" int las_lstm_pointwise_bwd(const float* z, const float* c_prev, const float* dh, const float* dc_in, int N, int H,
                                      float forget_bias, float* dz, float* dc_prev, void* stream) {
    LAS_ARG(z && c_prev && dh && dz && dc_prev && N > 0 && H > 0, "las_lstm_pointwise_bwd: bad arguments");
    int nb = cdiv((long long)N * H, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, c_prev, dh, dc_in, N, H, forget_bias, dz, dc_prev);
    LAS_LAUNCHED();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// bf16 weight shadows of the speed mode (las.layers._shadow): after every optimiser step ALL of them are rebuilt from the fp32
// masters by ONE launch.  A shadow is D = pad(op([S0 | S1])): up to two fp32 sources concatenated along the columns, optionally
// transposed, zero-padded to dst_rows x dst_cols, stored as bf16 (or fp32).  grid = (max tiles over the shadows, shadows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_shadows_kernel(const las_shadow_desc* __restrict__ descs) {
    const las_shadow_desc d = descs[blockIdx.y];
    const int tiles_c = (d.dst_cols + 31) / 32, tiles_r = (d.dst_rows + 31) / 32;
    if ((int)blockIdx.x >= tiles_c * tiles_r) return;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, four rows each
    __shared__ float tile[32][33];
    const int cols = d.cols0 + d.cols1;
    auto src = [&](int r, int c) -> float {
        if (r >= d.rows || c >= cols) return 0.f;
        return c < d.cols0 ? d.src0[(long long)r * d.ld0 + c] : d.src1[(long long)r * d.ld1 + (c - d.cols0)];
    };
    // destination element (i, j) = L(j, i) when transposed, else L(i, j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = ty * 4 + k;
        // read coalesced along the SOURCE columns
        tile[y][tx] = d.transpose ? src(tc * 32 + y, tr * 32 + tx) : src(tr * 32 + y, tc * 32 + tx);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = ty * 4 + k;
        const int i = tr * 32 + y, j = tc * 32 + tx;
        if (i >= d.dst_rows || j >= d.dst_cols) continue;
        const float v = d.transpose ? tile[tx][y] : tile[y][tx];
        if (d.dst_bf16) reinterpret_cast<unsigned short*>(d.dst)[(long long)i * d.dst_ld + j] = f2bf(v);
        else            reinterpret_cast<float*>(d.dst)[(long long)i * d.dst_ld + j] = v;
    }
}

extern "C" int las_build_shadows(const las_shadow_desc* descs_dev, int n, int max_tiles, void* stream) {
    LAS_ARG(descs_dev && n > 0 && max_tiles > 0, "las_build_shadows: bad arguments");
    hipLaunchKernelGGL(build_shadows_kernel, dim3(max_tiles, n), dim3(256), 0, (hipStream_t)stream, descs_dev);
    LAS_LAUNCHED();
    return 0;
}
