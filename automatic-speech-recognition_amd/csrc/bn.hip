// bn.hip -- tf.layers.batch_normalization over the last axis (+ the ReLU behind it) for the CNN listener's recurrent stack
// (reference las/layers.py:114-116,155-161: every BLSTM layer ends in dense -> [bn ->] relu(bn(.)); momentum 0.99, epsilon 1e-3).
//
// Until round 5 this ran through torch's batch-norm kernels (SURVEY section 2.1 allows the secondary encoder's conv / bn on the library
// path): at run.sh's sizes -- [48 x 319, 512] activations, four layers -- 1.8 ms of a 34 ms train step between the sweeps (statistics
// 126 us, backward reduce 249 us, two apply kernels and two ReLU kernels per layer: profiles/r6_runsh_rnn_kernel_stats.csv).  Here a layer's
// forward is statistics (every workgroup one pass over a [rows / S, 64] block held in registers: local mean, then the centred sum of
// squares; Chan's combination of the S partials in fixed order) + apply (normalise, scale, shift, ReLU in one pass), its backward one
// reduce (d beta, d gamma from dy masked by the ReLU) + one apply.  HBM-bound: 4 C bytes per row and pass.  No atomics, fixed orders.
#include "las_common.h"

constexpr int BN_RL = 16;            // row lanes of a statistics workgroup (256 threads = 16 row lanes x 16 float4 column lanes)
constexpr int BN_MAXR = 16;          // rows a thread keeps in registers: a workgroup covers up to 256 rows x 64 columns

// partials [S][3][C]: count, mean, M2 of the block's rows per column
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long rows, int C, int rows_per, float* __restrict__ part) {
    __shared__ float red[BN_RL][64 + 4];
    const int cb = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, rl = tid >> 4, c4 = tid & 15;
    const long long r0 = (long long)sp * rows_per;
    const long long nrow = rows - r0 < rows_per ? rows - r0 : rows_per;          // rows of this block (> 0 by construction)
    const int col = cb * 64 + c4 * 4;
    const bool con = col < C;                                                    // (C % 4 == 0)
    float4 v[BN_MAXR];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const long long r = (long long)i * BN_RL + rl;
        const bool on = con && r < nrow;
        v[i] = on ? *reinterpret_cast<const float4*>(x + (r0 + r) * C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w;
    }
    red[rl][c4 * 4] = s.x; red[rl][c4 * 4 + 1] = s.y; red[rl][c4 * 4 + 2] = s.z; red[rl][c4 * 4 + 3] = s.w;
    __syncthreads();
    float m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < BN_RL; ++q) t += red[q][c4 * 4 + e];
        m[e] = t / (float)nrow;
    }
    __syncthreads();
    float4 q2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < BN_MAXR; ++i) {
        const long long r = (long long)i * BN_RL + rl;
        if (con && r < nrow) {
            const float a = v[i].x - m[0], b = v[i].y - m[1], c = v[i].z - m[2], d = v[i].w - m[3];
            q2.x = fmaf(a, a, q2.x); q2.y = fmaf(b, b, q2.y); q2.z = fmaf(c, c, q2.z); q2.w = fmaf(d, d, q2.w);
        }
    }
    red[rl][c4 * 4] = q2.x; red[rl][c4 * 4 + 1] = q2.y; red[rl][c4 * 4 + 2] = q2.z; red[rl][c4 * 4 + 3] = q2.w;
    __syncthreads();
    if (tid < 64 && cb * 64 + tid < C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < BN_RL; ++q) t += red[q][tid];
        // (every thread of a column group computed the same mean: recompute this column's from the first pass is not needed -- lane c4 = tid / 4 holds it)
        float* p = part + (size_t)sp * 3 * C;
        p[cb * 64 + tid] = (float)nrow;
        p[2 * C + cb * 64 + tid] = t;
    }
    if (rl == 0 && con) {
        float* p = part + (size_t)sp * 3 * C + C;
        p[col] = m[0]; p[col + 1] = m[1]; p[col + 2] = m[2]; p[col + 3] = m[3];
    }
}

// Chan et al.: combine the S blocks' (n, mean, M2) per column in block order -> batch mean, rstd = 1 / sqrt(biased variance + eps); moving
// statistics (optional): moving = (1 - momentum) moving + momentum x {mean, UNBIASED variance} (what the fused TF / torch kernels feed them)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int S, int C, float eps, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mu = 0.f, m2 = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* p = part + (size_t)s * 3 * C;
        const float nb = p[c], mb = p[C + c], qb = p[2 * C + c];
        const float nt = n + nb, d = mb - mu;
        mu += d * (nb / nt);
        m2 += qb + d * d * (n * nb / nt);
        n = nt;
    }
    const float var = m2 / n;
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long long n4, int C4, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = reinterpret_cast<const float4*>(mean)[c], rs = reinterpret_cast<const float4*>(rstd)[c];
        const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
        float4 o = make_float4((v.x - mu.x) * rs.x * g.x + b.x, (v.y - mu.y) * rs.y * g.y + b.y, (v.z - mu.z) * rs.z * g.z + b.z,
                               (v.w - mu.w) * rs.w * g.w + b.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// backward reduce: partial d beta = sum dyr, d gamma = sum dyr xhat over the block's rows, dyr = dy (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                            long long rows, int C, int rows_per, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int relu, float* __restrict__ part) {
    __shared__ float red[2][BN_RL][64 + 4];
    const int cb = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, rl = tid >> 4, c4 = tid & 15;
    const long long r0 = (long long)sp * rows_per;
    const long long nrow = rows - r0 < rows_per ? rows - r0 : rows_per;
    const int col = cb * 64 + c4 * 4;
    const bool con = col < C;
    const int colc = con ? col : 0;
    const float4 mu = *reinterpret_cast<const float4*>(mean + colc), rs = *reinterpret_cast<const float4*>(rstd + colc);
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = rl; r < nrow; r += BN_RL) {
        if (!con) break;
        const size_t o = (size_t)(r0 + r) * C + col;
        const float4 xv = *reinterpret_cast<const float4*>(x + o);
        float4 g = *reinterpret_cast<const float4*>(dy + o);
        if (relu) {
            const float4 yv = *reinterpret_cast<const float4*>(y + o);
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        sb.x += g.x; sb.y += g.y; sb.z += g.z; sb.w += g.w;
        sg.x = fmaf(g.x, (xv.x - mu.x) * rs.x, sg.x); sg.y = fmaf(g.y, (xv.y - mu.y) * rs.y, sg.y);
        sg.z = fmaf(g.z, (xv.z - mu.z) * rs.z, sg.z); sg.w = fmaf(g.w, (xv.w - mu.w) * rs.w, sg.w);
    }
    red[0][rl][c4 * 4] = sb.x; red[0][rl][c4 * 4 + 1] = sb.y; red[0][rl][c4 * 4 + 2] = sb.z; red[0][rl][c4 * 4 + 3] = sb.w;
    red[1][rl][c4 * 4] = sg.x; red[1][rl][c4 * 4 + 1] = sg.y; red[1][rl][c4 * 4 + 2] = sg.z; red[1][rl][c4 * 4 + 3] = sg.w;
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        if (cb * 64 + c < C) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < BN_RL; ++q) t += red[which][q][c];
            part[((size_t)sp * 2 + which) * C + cb * 64 + c] = t;
        }
    }
}
// sums over the S blocks in block order -> dbeta / dgamma (written to scratch sums[2][C] for the apply pass, and ACCUMULATED (+=) into the
// caller's gradient buffers: they may be views of a flat gradient bucket)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int S, int C, float* __restrict__ sums,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float b = 0.f, g = 0.f;
    for (int s = 0; s < S; ++s) { b += part[((size_t)s * 2) * C + c]; g += part[((size_t)s * 2 + 1) * C + c]; }
    sums[c] = b; sums[C + c] = g;
    if (dbeta) dbeta[c] += b;
    if (dgamma) dgamma[c] += g;
}
// dx = gamma rstd (dyr - mean(dyr) - xhat mean(dyr xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                           long long n4, int C4, float inv_rows, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, int relu, float* __restrict__ dx) {
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        const float4 mu = reinterpret_cast<const float4*>(mean)[c], rs = reinterpret_cast<const float4*>(rstd)[c];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[c];
        const float4 sb = reinterpret_cast<const float4*>(sums)[c], sg = reinterpret_cast<const float4*>(sums + C)[c];
        float4 o;
        o.x = ga.x * rs.x * (g.x - sb.x * inv_rows - (xv.x - mu.x) * rs.x * sg.x * inv_rows);
        o.y = ga.y * rs.y * (g.y - sb.y * inv_rows - (xv.y - mu.y) * rs.y * sg.y * inv_rows);
        o.z = ga.z * rs.z * (g.z - sb.z * inv_rows - (xv.z - mu.z) * rs.z * sg.z * inv_rows);
        o.w = ga.w * rs.w * (g.w - sb.w * inv_rows - (xv.w - mu.w) * rs.w * sg.w * inv_rows);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

static int bn_split(long long rows, int* rows_per) {
    long long per = BN_RL * BN_MAXR;                       // 256 rows per block at most (held in registers)
    long long S = (rows + per - 1) / per;
    *rows_per = (int)per;
    return (int)S;
}
extern "C" size_t las_bn_workspace_bytes(long long rows, int C) {
    int rp;
    const int S = bn_split(rows > 0 ? rows : 1, &rp);
    return ((size_t)S * 3 * C + 2 * (size_t)C) * sizeof(float) + 256;
}
extern "C" int las_bn_relu_fwd(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                               float* moving_mean, float* moving_var, float momentum, int relu, float* y, void* ws, size_t ws_bytes, void* stream) {
    LAS_ARG(x && gamma && beta && mean && rstd && y && rows > 0 && C > 0 && (C % 4) == 0, "las_bn_relu_fwd: bad arguments (C must be a multiple of 4)");
    LAS_ARG((moving_mean == nullptr) == (moving_var == nullptr), "las_bn_relu_fwd: moving_mean and moving_var go together");
    LAS_ARG(ws && ws_bytes >= las_bn_workspace_bytes(rows, C), "las_bn_relu_fwd: workspace too small");
    LAS_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) == 0, "las_bn_relu_fwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    int rp;
    const int S = bn_split(rows, &rp);
    float* part = (float*)ws;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(cdiv(C, 64), S), dim3(256), 0, st, x, rows, C, rp, part);
    LAS_LAUNCHED();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)part, S, C, eps, mean, rstd, moving_mean, moving_var, momentum);
    LAS_LAUNCHED();
    const long long n4 = rows * (C / 4);
    int nb = cdiv(n4, 256 * 4);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nb), dim3(256), 0, st, x, n4, C / 4, (const float*)mean, (const float*)rstd, gamma, beta, relu, y);
    LAS_LAUNCHED();
    return 0;
}
extern "C" int las_bn_relu_bwd(const float* x, const float* y, const float* dy, long long rows, int C, const float* gamma, const float* mean,
                               const float* rstd, int relu, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    LAS_ARG(x && dy && gamma && mean && rstd && dx && rows > 0 && C > 0 && (C % 4) == 0, "las_bn_relu_bwd: bad arguments");
    LAS_ARG(!relu || y, "las_bn_relu_bwd: the ReLU's mask needs y");
    LAS_ARG(ws && ws_bytes >= las_bn_workspace_bytes(rows, C), "las_bn_relu_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int rp;
    const int S = bn_split(rows, &rp);
    float* part = (float*)ws;
    float* sums = part + (size_t)S * 3 * C;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C, 64), S), dim3(256), 0, st, x, y, dy, rows, C, rp, mean, rstd, relu, part);
    LAS_LAUNCHED();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)part, S, C, sums, dgamma, dbeta);
    LAS_LAUNCHED();
    const long long n4 = rows * (C / 4);
    int nb = cdiv(n4, 256 * 4);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb), dim3(256), 0, st, x, y, dy, n4, C / 4, 1.0f / (float)rows, mean, rstd, gamma, (const float*)sums, relu, dx);
    LAS_LAUNCHED();
    return 0;
}
