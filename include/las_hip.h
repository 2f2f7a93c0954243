/* las_hip.h -- C ABI of liblas_hip.so, the MI355X (gfx950) engine behind the LAS hot path.
 *
 * The reference (30stomercury/Automatic-Speech-Recognition) has no FFI: its hot path sits
 * behind the TensorFlow session boundary  sess.run(fetches, feed_dict)  (train.py:115-117,
 * test.py:107, las/beam_search.py:209,222).  This header is what replaces that boundary: each
 * entry point names the reference graph fragment (file:line) whose stock TF ops it stands in for.
 *
 * Conventions
 *   - extern "C"; raw DEVICE pointers owned by the caller; the library never allocates or frees
 *     caller tensors.  Scratch is passed in explicitly (ws, ws_bytes), sized by *_workspace_bytes.
 *   - every dimension / leading dimension is an explicit int; tensors are fp32 in HBM (speed mode: the Listener's
 *     activations are bf16, see LAS_DT_*), row-major,
 *     batch-major [B,T,C] exactly as the reference lays them out (time_major=False,
 *     las/layers.py:24,53).  Weights keep the TF layout [in,out] with the [x;h] row concat and
 *     gate order i,j,f,o.
 *   - `prec` selects the arithmetic of the contractions:  LAS_PREC_F32 = exact fp32 FMA chains
 *     (parity mode), LAS_PREC_BF16 = operands rounded to bf16 (RNE), fp32 MFMA accumulation
 *     (speed mode).  Recurrent state, accumulators, parameters, parameter gradients and optimiser math are always fp32.
 *   - `stream` is a hipStream_t (passed as void*); all work is enqueued asynchronously on it,
 *     no hidden synchronisation, no global mutable state besides an init-once attribute cache; the library
 *     reads no environment variables (development switches are explicit `flags` arguments).
 *   - return value: 0 ok; <0 invalid argument / unsupported shape (message via las_last_error());
 *     >0 a hipError_t.
 */
#ifndef LAS_HIP_H
#define LAS_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LAS_PREC_F32 = 0, LAS_PREC_BF16 = 1 };
enum { LAS_CELL_RNN = 0, LAS_CELL_LSTM = 1 };   /* BasicRNNCell (las/layers.py:31) / BasicLSTMCell */
enum { LAS_ACT_NONE = 0, LAS_ACT_TANH = 1 };
enum { LAS_ATT_ADD = 0, LAS_ATT_LOC = 1 };      /* las/las.py:44-49 */
enum { LAS_DT_F32 = 0, LAS_DT_BF16 = 1 };       /* element type of a tensor in HBM (see las_gemm_kk) */

#define LAS_HIP_ABI_VERSION 600      /* bumped whenever an argument struct or a signature changes: las_version() of a library
                                        built from another header differs, and the Python loader refuses it */
int         las_version(void);
const char* las_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1/K3/K4  dense contractions  (tf.layers.dense / Tensordot->MatMul+BiasAdd(+Tanh):
 * las/layers.py:71-74, :89-93, :250-251, :305; the input half of the cell MatMul las/layers.py:31;
 * and every matmul gradient of those ops).
 *   C[b] = act( alpha * op(A[b]) . op(B[b]) + beta * C[b] + bias )      b = 0..batch-1
 * op(A) is M x K, op(B) is K x N.  transX=0: X stored row-major as written; transX=1: stored
 * transposed (A as K x M, B as N x K).  bias (length N) may be NULL.
 * a_mask_period > 0: logical rows r of the CONTRACTION index of a transA=1 product with
 * (r % a_mask_period) == a_mask_skip contribute zero -- used for dW_hh = sum_t h_{t-1}^T dG_t where
 * the t=0 (fw) / t=T-1 (bw) frame has no predecessor inside its utterance.
 * ws (optional, may be NULL): scratch for a deterministic split-K of tall contractions
 * (partials [s][M][N] reduced in fixed order); without it the product runs unsplit.
 */
int las_gemm(int prec, int transA, int transB, int M, int N, int K,
             float alpha, const float* A, int lda, long long strideA,
             const float* B, int ldb, long long strideB,
             float beta, float* C, int ldc, long long strideC,
             const float* bias, int act, int batch,
             int a_mask_period, int a_mask_skip, void* ws, size_t ws_bytes, void* stream);

/* Scratch las_gemm's split-K wants for this product (0: it runs unsplit); at most LAS_GEMM_WS_CAP.  A caller that passes at least
 * this much gets the same split -- hence bit-identical sums -- whatever else its buffer could hold. */
#define LAS_GEMM_WS_CAP ((size_t)256 << 20)
size_t las_gemm_workspace_bytes(int prec, int M, int N, int K, int batch);

/* las_gemm with both operands in `in_dtype` (LAS_DT_F32 = las_gemm; LAS_DT_BF16: activations / gradients that already
 * live in HBM as bf16 -- the weight-gradient contractions X^T . dZ of the speed mode; branch-free path only: aligned
 * pitches, K resp. row counts multiples of 4, no contraction mask).  C, bias fp32. */
int las_gemm_dt(int prec, int transA, int transB, int M, int N, int K,
                float alpha, const void* A, int lda, long long strideA,
                const void* B, int ldb, long long strideB, int in_dtype,
                float beta, float* C, int ldc, long long strideC,
                const float* bias, int act, int batch,
                int a_mask_period, int a_mask_skip, void* ws, size_t ws_bytes, void* stream);

/* Both weight gradients of a recurrent layer's direction(s) (the matmul gradient of the cell's kernel, las/layers.py:31 under
 * bidirectional_dynamic_rnn, las/layers.py:49-53) in one pass over d(pre-activation):
 *   dW[0:I, :] += X^T . dZ_d        dW[I:I+H, :] += sum over utterances and frames of h_prev^T . dZ_d
 * X bf16 [B*T, ldx] (layer input; columns >= I up to the next multiple of 8 must be zero), out bf16 [B, Tp, ld_out >= 2 H] (batch stride
 * out_bstride elements) and dZ bf16 [B*T, lddz >= 2 GH]: the layer's output / d(pre-activation) tensors of BOTH directions (direction d's
 * column block starts at d H / d GH).  dir 0: h_prev(b, t) = out_fw[b, t-1] (zero at t = 0); dir 1: out_bw[b, t+1] (zero at t = T-1);
 * dir 2: both directions in ONE launch (dW = forward, dW2 = backward direction's gradient; X2 = the backward direction's own input copy
 * -- input dropout draws one mask per direction -- or NULL = X).  dW fp32 [I + H, GH], accumulated.  Needs H % 128 == 0, GH % 128 == 0,
 * 16-byte aligned operands, pitches multiples of 8.  Deterministic split-K over the frames (scratch from
 * las_wgrad_ih_hh_workspace_bytes(.., ndir), reduced in fixed order). */
size_t las_wgrad_ih_hh_workspace_bytes(int I, int H, int GH, int B, int T, int ndir);
/* The same contraction over a WINDOW of `nframes` frames per utterance: frames [t0_fw, t0_fw + nframes) for the forward direction's gradient,
 * [t0_bw, t0_bw + nframes) for the backward direction's (dir as below); accumulates into dW / dW2 like las_wgrad_ih_hh, so consecutive
 * windows that tile [0, T) give the whole gradient.  max_workgroups > 0 caps the launch (longer k-chunks, fewer split-K partials): a window
 * that runs beside the sweep that is producing dZ should stay small.  Workspace: las_wgrad_ih_hh_workspace_bytes of the whole sequence. */
int las_wgrad_ih_hh_window(const void* X, const void* X2, int ldx, int I, const void* out, int ld_out, long long out_bstride, const void* dZ, int lddz,
                           int B, int T, int H, int GH, int dir, int t0_fw, int t0_bw, int nframes, int max_workgroups,
                           float* dW, float* dW2, void* ws, size_t ws_bytes, void* stream);
int las_wgrad_ih_hh(const void* X, const void* X2, int ldx, int I, const void* out, int ld_out, long long out_bstride, const void* dZ, int lddz,
                    int B, int T, int H, int GH, int dir, float* dW, float* dW2, void* ws, size_t ws_bytes, void* stream);

/* Speed-mode storage type of activations in HBM.  LAS_PREC_BF16 keeps the Listener's activations (x-projections /
 * saved gates, cell states, h, dense outputs and their gradients) as bf16 -- SURVEY 8(d)'s algorithmic bytes -- while
 * accumulators, recurrent state inside the sweeps and all parameters / optimiser state stay fp32. */
/* The dependency-chain products of the Listener in speed mode, both operands bf16 with the contraction index contiguous:
 *   C[M,N] = act( A[M,K] . B[N,K]^T + bias ),   C bf16 (c_dtype = LAS_DT_BF16) or fp32.
 * x-projection and dense layers pass the bf16 shadow of W^T as B, their input gradients the shadow of W
 * (las/layers.py:31,49-53,71-74,89-93 and the matmul gradients of those ops).  K % 64 == 0 (pad with zero columns),
 * N % 4 == 0, row pitches lda / ldb multiples of 8 elements, 16-byte aligned operands.  LDS-DMA staged, 128x128x64 tiles. */
int las_gemm_kk(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                void* C, int c_dtype, long long ldc, const float* bias, int act, void* stream);
/* ... with the Tanh gradient of the producer of the result's consumer fused into the epilogue: C[m,n] *= 1 - y[m,n]^2
 * (y bf16, pitch ldy; NULL = las_gemm_kk).  Used for dX of a recurrent layer whose input IS the tanh output of the dense
 * layer below (las/layers.py:71-74 feeding :80): the separate las_tanh_bwd pass over dX disappears. */
int las_gemm_kk_tanhgrad(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                         void* C, int c_dtype, long long ldc, const float* bias, int act, const void* y, long long ldy, void* stream);

/* out[j] = beta*out[j] + sum_r X[r*ldx + j],  r < rows, j < cols   (BiasAdd gradient);
 * fixed-order two-stage reduction, ws >= las_colsum_workspace_bytes(cols). */
size_t las_colsum_workspace_bytes(int cols);
int las_colsum(const float* X, int rows, int cols, int ldx, float beta, float* out,
               void* ws, size_t ws_bytes, void* stream);

int las_colsum_dt(const void* X, int dtype, int rows, int cols, int ldx, float beta, float* out,
                  void* ws, size_t ws_bytes, void* stream);            /* X fp32 or bf16 (LAS_DT_*) */

/* dX = dY * (1 - Y*Y)   elementwise over rows x cols (Tanh gradient of dense(.., tanh)). */
int las_tanh_bwd(const float* Y, int ldy, const float* dY, int lddy, float* dX, int lddx,
                 int rows, int cols, void* stream);
int las_tanh_bwd_dt(const void* Y, int y_dt, int ldy, const void* dY, int dy_dt, int lddy, void* dX, int dx_dt, int lddx,
                    int rows, int cols, void* stream);               /* each tensor fp32 or bf16 */

/* ------------------------------------------------------------------------------------------
 * K2/K2b  recurrent sweep of one bidirectional layer
 * (tf.nn.bidirectional_dynamic_rnn without sequence_length, las/layers.py:49-53: every padded
 * frame is run; zero initial state; bw runs t=T-1..0).
 *
 * Element type of gates / out / cstate / dout: las_rnn_seq_io_dtype(cell, prec, H) -- fp32 in parity mode, bf16 in speed
 * mode (the recurrent state, accumulators and gate math stay fp32 inside the kernel).
 * gates : [B,T,2,G*H]  (G=1 rnn, 4 lstm; dir 0 = fw, 1 = bw).
 *         fwd in : x_t . W_ih + bias (the K1 product).   fwd out (lstm): activated i,j,f,o.
 *         bwd in : what fwd left.                        bwd out: d(pre-activation) for K1's bwd.
 * whh_* : [H, G*H] recurrent half of the TF kernel (row stride ldw).
 * out   : h for both directions, element (b,t,dir*H+u) at out[b*out_bstride + t*ld_out + dir*H + u]
 *         (ld_out = 2H; out_bstride lets the caller keep a zero pad frame per utterance so the
 *         pyramid concat of las/layers.py:83-88 is a pure view).
 * cstate: lstm only, [B,T,2,H] cell states (saved for bwd).
 * dout  : gradient w.r.t. out, same addressing scheme.
 */
/* `flags` (development / test switches, 0 in normal use; explicit per call -- the library reads no environment):
 *   LAS_SEQ_AGENT_GRANULES   cluster members always exchange through agent-scope (write-through) granules,
 *                            as if they ran on different XCDs
 *   LAS_SEQ_NO_KSPLIT        BPTT uses the all-gather cluster kernel instead of the K-split reduce-scatter one
 *   LAS_SEQ_NO_HELPER_WAVES  forward sweep without the helper waves that own the bulk HBM traffic
 *   LAS_SEQ_ROWS16           clustered sweeps always use 16-row batch tiles (default: 8-row tiles, i.e. twice the CUs and half
 *                            the per-lane work of a dependent step, whenever the whole batch fits one launch that way)
 *   LAS_SEQ_NO_WARMERS       clustered sweeps without the per-cluster "L2 warmer" workgroup (one extra workgroup on the cluster's
 *                            XCD that touches the operands of the next steps so that the cluster's own loads hit in L2)
 *   LAS_SEQ_F32_VALU         LAS_PREC_F32: the round-1 VALU kernels (one workgroup per (direction, 8 rows), W_hh streamed from L2)
 *                            instead of the clustered exact-fp32 MFMA kernels (csrc/rnn_seq_f32.hip) -- tests cross-check the two
 *   LAS_SEQ_P(p)             cluster width override (1, 2, 4, 8 workgroups per (direction, 16-row tile))
 *   LAS_SEQ_SPIN_LOG2(n)     bound of every exchange spin = 2^n polls (default 2^22)
 *   LAS_SEQ_PREPARED         NOT a development switch: `ws` was prepared by las_rnn_seq_prepare (below) for these weights, this cell / H /
 *                            direction of the pass / flags and a batch >= B, and no sweep has used it since -- the launch then skips its own
 *                            weight pack + exchange-state clear (one launch less on the dependency chain per sweep)
 * `status` (may be NULL): caller-owned, caller-zeroed int32 DEVICE word.  The clustered bf16 sweeps exchange h / dh
 *   between workgroups with bounded spins; if a partner does not publish within the bound (it is not co-resident:
 *   shared or partitioned device) the launch finishes with undefined results and stores LAS_SEQ_STATUS_* here.
 *   The word is sticky (never cleared by the library); the caller reads it at its next synchronisation point. */
enum { LAS_SEQ_AGENT_GRANULES = 1, LAS_SEQ_NO_KSPLIT = 2, LAS_SEQ_NO_HELPER_WAVES = 4, LAS_SEQ_ROWS16 = 8, LAS_SEQ_NO_WARMERS = 16,
       LAS_SEQ_F32_VALU = 32, LAS_SEQ_PREPARED = 64 };
#define LAS_SEQ_P(p) (((p) & 0xf) << 8)
#define LAS_SEQ_SPIN_LOG2(n) (((n) & 0x1f) << 16)
/* LAS_SEQ_ANNOUNCE(n), n in 1..1023 (las_rnn_seq_bwd*, clustered kernels): `status` then points to TWO ints and the launch
 * stores n into status[1] as soon as its first cluster is resident on the device -- see las_wait_word. */
#define LAS_SEQ_ANNOUNCE(n) (((n) & 0x3ff) << 21)
enum { LAS_SEQ_STATUS_OK = 0, LAS_SEQ_STATUS_FWD_TIMEOUT = 1, LAS_SEQ_STATUS_BWD_TIMEOUT = 2 };

/* Workspace of a sweep.  MANDATORY for every clustered kernel: LAS_PREC_BF16 with H in {64,128,256,512}, and -- since round 4 -- LAS_PREC_F32 with
 * those H as well (the exact-fp32 MFMA sweeps; a NULL / smaller `ws` is refused with "workspace too small", it does not fall back).  Only the
 * round-1 VALU kernels (other H, or LAS_SEQ_F32_VALU) run the forward sweep without one.  The clustered kernels also need all their workgroups
 * resident at once: on a shared / partitioned device they report LAS_SEQ_STATUS_* instead (bounded spins); pass LAS_SEQ_F32_VALU there. */
size_t las_rnn_seq_workspace_bytes(int cell, int prec, int H, int B);
/* Everything a speed-mode sweep does in front of its persistent kernel depends on the weights only: W_hh of both directions re-packed into
 * MFMA fragment order (forward, BPTT and K-split BPTT layouts differ) and the cleared exchange state of the workspace.  The reference
 * rebuilds nothing per layer call either (its weights are graph variables, las/layers.py:28-54).  las_rnn_seq_prepare does that work for n
 * (layer, pass) pairs in ONE launch -- once per optimiser step, off the dependency chain -- each into a workspace of its own
 * (las_rnn_seq_workspace_bytes(cell, LAS_PREC_BF16, H, B)); the sweep that then gets such a workspace together with LAS_SEQ_PREPARED in
 * `flags` launches nothing but its persistent kernel.  A prepared workspace serves ONE sweep (the exchange state is dirty afterwards) with
 * the same cell / H / flags / direction of the pass (bwd = 0: las_rnn_seq_fwd*, 1: las_rnn_seq_bwd*) and any batch B' <= B.
 * whh_fw / whh_bw: as for las_rnn_seq_fwd.  Only LAS_PREC_BF16 shapes (H in {64,128,256,512}).  `descs` is HOST memory. */
typedef struct las_seq_prepare_desc {
    const float* whh_fw; const float* whh_bw; int ldw;
    int cell, H, B, bwd, flags;
    void* ws; size_t ws_bytes;
} las_seq_prepare_desc;
int las_rnn_seq_prepare(const las_seq_prepare_desc* descs, int n, void* stream);
/* Element type of gates / out / cstate / dout for (cell, prec, H): LAS_DT_BF16 when the speed-mode MFMA sweeps serve the
 * shape (prec = LAS_PREC_BF16 and H in {64,128,256,512}), else LAS_DT_F32 (parity mode, or a speed-mode shape that falls
 * back to the fp32 VALU sweep).  The caller allocates those tensors -- and makes the K1 product write them -- accordingly. */
int las_rnn_seq_io_dtype(int cell, int prec, int H);
int las_rnn_seq_fwd(int cell, int prec, int B, int T, int H, void* gates,
                    const float* whh_fw, const float* whh_bw, int ldw,
                    void* out, int ld_out, long long out_bstride, void* cstate,
                    float forget_bias, int flags, int* status, void* ws, size_t ws_bytes, void* stream);
/* las_rnn_seq_fwd for an x-projection that is still being computed: `gates` is filled in time chunks of `chunk_steps` sweep steps,
 * chunk k = frames [k*cs, (k+1)*cs) and [T-(k+1)*cs, T-k*cs) of every utterance (both ends of the sequence first: the forward
 * direction consumes t = 0, 1, .., the backward direction t = T-1, T-2, ..), e.g. by las_gemm_kk_frames launches on ANOTHER
 * stream, each followed by las_set_word(chunk_flag, k+1).  The sweep reads a frame only after *chunk_flag has reached its chunk
 * (bounded wait -> LAS_SEQ_STATUS_FWD_TIMEOUT).  Chunk 0 must be complete IN STREAM ORDER in front of this call (produced on `stream`,
 * or on a stream `stream` has waited for): the sweep never waits for it and the value of *chunk_flag only matters from 2 on -- no
 * flag launch is needed between chunk 0's product and the sweep (round 5).  Only the helper-wave kernel
 * supports this: ask las_rnn_seq_fwd_chunks_ok first.  chunk_flag = NULL: las_rnn_seq_fwd. */
int las_rnn_seq_fwd_chunks_ok(int cell, int prec, int B, int H, int flags);
int las_rnn_seq_fwd_chunked(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                            const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                            void* cstate, float forget_bias, int flags, int* status, const int* chunk_flag, int chunk_steps,
                            void* ws, size_t ws_bytes, void* stream);
/* las_rnn_seq_fwd for a batch whose rows have DIFFERENT lengths (beam search encodes utterances the reference feeds one at a time,
 * unpadded -- its encoder has no length mask): row_T [B] device ints, row_T[b] <= T frames of row b.  At frames t >= row_T[b] the row's
 * state and outputs are forced to ZERO: the backward direction reaches the row's last real frame with the zero state of an unpadded run,
 * the forward direction's real frames come first, and the zero pad frame is the one the pyramid appends to an odd-length utterance --
 * every real frame equals the one-utterance-at-a-time result.  Served by the 8-row helper-wave kernel (las_rnn_seq_fwd_rows_ok). */
int las_rnn_seq_fwd_rows_ok(int cell, int prec, int B, int H, int flags);
int las_rnn_seq_fwd_rows(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                         const float* whh_bw, int ldw, void* out, int ld_out, long long out_bstride,
                         void* cstate, float forget_bias, int flags, int* status, const int* row_T,
                         void* ws, size_t ws_bytes, void* stream);
/* C = act(A . B^T + bias) like las_gemm_kk, restricted to the frames [lo0, lo0+nlo) and [hi0, hi0+nhi) of every utterance of
 * [nb, T, *] tensors A and C (row pitches lda / ldc per frame); las_set_word: stream-ordered store of a device word. */
int las_gemm_kk_frames(int nb, int T, int lo0, int nlo, int hi0, int nhi, int N, int K, const void* A, long long lda,
                       const void* B, long long ldb, void* C, int c_dtype, long long ldc, const float* bias, int act,
                       const void* y_tanh, long long ldy, void* stream);       /* y_tanh as in las_gemm_kk_tanhgrad (may be NULL) */
int las_set_word(int* word, int value, void* stream);

int las_rnn_seq_bwd(int cell, int prec, int B, int T, int H, void* gates,
                    const float* whh_fw, const float* whh_bw, int ldw,
                    const void* out, int ld_out, long long out_bstride, const void* cstate,
                    const void* dout, int ld_dout, long long dout_bstride,
                    float forget_bias, int flags, int* status, void* ws, size_t ws_bytes, void* stream);
/* Same, and additionally accumulates (+=) the bias gradients of the two directions, dbias_fw / dbias_bw [G*H] fp32 (either
 * may be NULL) = column sums of d(pre-activation) over all B*T frames (the bias of the TF cell kernel, las/layers.py:31).
 * The cluster BPTT kernel sums them in fp32 registers while it sweeps (before the bf16 rounding of the stored gradient). */
int las_rnn_seq_bwd_db(int cell, int prec, int B, int T, int H, void* gates,
                       const float* whh_fw, const float* whh_bw, int ldw,
                       const void* out, int ld_out, long long out_bstride, const void* cstate,
                       const void* dout, int ld_dout, long long dout_bstride,
                       float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                       void* ws, size_t ws_bytes, void* stream);
/* las_rnn_seq_bwd_db for an upstream gradient `dout` that is still being computed: dout [B, Tp, 2H] is the input gradient of the
 * dense layer above, whose rows are pyramid frame pairs (n_rows = ceil(T/2)) or single frames (n_rows = T); it is filled in chunks
 * of `chunk_rows` (a power of two) of those rows from both ends of the sequence (chunk k = rows [k*c, (k+1)*c) and
 * [n_rows-(k+1)*c, n_rows-k*c) of every utterance), e.g. by las_gemm_kk_frames launches on another stream each followed by
 * las_set_word(chunk_flag, k+1).  The sweep reads a frame of dout only after *chunk_flag has reached the chunk of its row
 * (bounded wait -> LAS_SEQ_STATUS_BWD_TIMEOUT); chunk 0 must be complete in stream order in front of this call and is never waited for.  Only the 8-row K-split cluster kernel supports it (las_rnn_seq_bwd_chunks_ok). */
int las_rnn_seq_bwd_chunks_ok(int cell, int prec, int B, int H, int flags);
/* ... and a sweep that PUBLISHES ITS PROGRESS (round 5): d(pre-activation) leaves the sweep with agent-scope (write-through) stores, and every
 * progress_steps sweep steps -- and at the end -- member m of cluster c stores the number of steps whose dZ has reached memory into
 * progress[c * P + m] (las_rnn_seq_bwd_progress_words(...) caller-zeroed ints; 0 = the configuration has no such kernel).  After s steps the
 * forward direction's dZ exists for frames [T - s, T), the backward direction's for [0, s): the layer's weight gradients can follow the sweep
 * window by window on another stream (las_wait_words_min, las_wgrad_ih_hh_window) instead of starting when it ends -- the bottom layer's are
 * the end-of-step tail of the reference's train step (las/las.py:272-283 can only run behind them). */
int las_rnn_seq_bwd_progress_words(int cell, int prec, int B, int H, int flags);
int las_rnn_seq_bwd_db_progress(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                                const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                                const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                                float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                                const int* chunk_flag, int chunk_rows, int n_rows, int* progress, int progress_steps,
                                void* ws, size_t ws_bytes, void* stream);
int las_rnn_seq_bwd_db_chunked(int cell, int prec, int B, int T, int H, void* gates, const float* whh_fw,
                               const float* whh_bw, int ldw, const void* out, int ld_out, long long out_bstride,
                               const void* cstate, const void* dout, int ld_dout, long long dout_bstride,
                               float forget_bias, float* dbias_fw, float* dbias_bw, int flags, int* status,
                               const int* chunk_flag, int chunk_rows, int n_rows, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4-K7  Speller: the whole decode loop of Speller.__call__ (las/las.py:72-143) with
 * Speller.decode (las/las.py:145-160), AdditiveAttention / LocationAwareAttention
 * (las/layers.py:234-257 / :281-311), BaseAttention.mask/attend (las/layers.py:172-213).
 * One call enqueues all U steps (per step: one fused row kernel = finish previous cell +
 * [vocab logits/argmax/sample] + query projection + energies + mask + softmax + context + cell
 * input assembly, then the cell contraction through las_gemm).  All pointers device.  Layouts:
 *   enc   [B,Tp,Hd]   encoder output h           keys [B,Tp,A]  hoisted dense(hidden) (K4, las_gemm)
 *   enc_len int32 [B] (already int-cast as las/layers.py:193 does)
 *   Ws [S,A] (S=D*NL)  u [A]   emb [V,E]   Wv [D,V]  bv [V]
 *   loc_w [Kc,C], loc_b [C], Wf [C,A] (mode LOC only, else NULL)
 *   cellW[l] [(I_l+D), G*D], cellb[l] [G*D]  with I_0 = E+Hd, I_l = D   (HOST arrays of device ptrs)
 *   tokens_in int32 [U,B]: the token whose embedding enters step t (t=0: SOS; teacher forcing:
 *   teacher[:,t-1]).  -1 = greedy argmax of step t-1 (inference, las/las.py:111); -2 = a sample
 *   from Categorical(logits of step t-1) (scheduled sampling, las/las.py:101-105,170-175; Gumbel
 *   arg-max driven by `seed`).  Resolved tokens are written back in place.  Any negative entry
 *   requires step_logits=1 or 2.
 * Time-major internal results (the Python boundary returns [B,U,.] views):
 *   logits [U,B,V]   alphas [U,B,Tp]   tokens_out int32 [U,B] (argmax per step; step_logits=1 only)
 *   step_logits=2 (training with scheduled sampling): in-loop logits + draws only at the steps whose entering token is device-resolved
 *   (tokens_in < 0), every step's logits from the batched product behind the loop; tokens_out is written at those steps only.
 * Saved for backward (caller-allocated):
 *   hs  [NL,U+1,B,D]  h states (slot 0 = initial state: zeroed by the call, or supplied by the
 *       caller when keep_state0=1 -- single-step use by beam search)   cs [NL,U+1,B,D] (lstm)
 *   gates [NL,U,B,G*D] activated gates (lstm) / pre-activation scratch (rnn)
 *   xin0 [U,B,E+Hd+D]  first-layer cell input rows  [emb(token) ; context ; h_prev]
 */
enum { LAS_SPELLER_STATUS_TIMEOUT = 3 };   /* (LAS_SEQ_STATUS_* are 1 and 2) */
enum { LAS_SPELLER_NO_PF_ROWS = 1,    /* speed mode without the fully prefetching row kernels (generic bf16 rows) */
       LAS_SPELLER_NO_BF_ROWS = 2,    /* speed mode with the fp32-operand row kernels */
       LAS_SPELLER_NO_FUSED_STEP = 4,   /* speed mode with two launches per step (row kernel, then the cell product)
                                           instead of the whole loop in one launch (product + row workgroups, granule hand-off) */
       LAS_SPELLER_REUSE_PREP = 8 };    /* (las_speller_bwd: operand copies only) the workspace still holds the bf16 copies of enc / keys / Ws and the
                                           packed cell weights that an earlier call made from the SAME tensors -- skip rebuilding
                                           them (beam search calls the step U = 1 at a time against a fixed encoder output) */
enum { LAS_SPELLER_NO_LOGITS = 16 };    /* (las_speller_fwd, U = 1, speed mode, one LSTM layer, D and E + Hd + D multiples of 32) the beam search's step:
                                           attention row kernel, then the cell in ONE launch (product + gate math: hs / cs slot 1 and the activated
                                           gates); NO vocabulary projection, logits / tokens_out are left alone -- the caller projects inside
                                           las_beam_loop_step (proj_*) */
enum { LAS_SPELLER_ROWS_SHARE4 = 32 };  /* (with LAS_SPELLER_NO_LOGITS; round 5) the caller vouches that rows 4g .. 4g+3 have IDENTICAL enc / keys / enc_len
                                           (beam search: the hypotheses of one utterance are consecutive rows, beam % 4 == 0): the attention rows of
                                           four hypotheses run in one workgroup that reads Ws, the keys and the encoder rows once -- a quarter of
                                           the step's L2 traffic; bit-identical to one row per workgroup.  B % 4 == 0, tokens >= 0. */
enum { LAS_SPELLER_WIDE = 64,           /* (round 6) take the wide per-step path (csrc/speller_wide.h: the query projection as one product over all rows, energies and
                                           context on (slice, utterance) workgroups, every layer's cell product on pre-packed MFMA fragments) wherever its
                                           geometry allows; by default it serves the multi-layer and location-aware calls of both modes
                                           outside the one-launch loop kernels' geometry (e.g. run.sh's 2 x 1024 decoder at T' = 319) */
       LAS_SPELLER_NO_WIDE = 128,       /* never take it (round 5's per-utterance row kernels) */
       LAS_SPELLER_SHARED_OPERANDS = 1 << 14 };  /* (with row_group > 0 and LAS_SPELLER_NO_LOGITS: a beam search's step) enc and keys hold ONE block per
                                           group of row_group rows -- enc [B / row_group, Tp, Hd], keys [B / row_group, Tp, A] -- instead of one copy per
                                           hypothesis row (the reference feeds np.tile(h), las/beam_search.py:216: 16 copies of every utterance's
                                           205 KB, each read through its own addresses -- 54 MB per step at 256 rows).  enc_len stays per row. */
#define LAS_SPELLER_SPIN_LOG2(n) (((n) & 31) << 8)   /* tests: the loop kernels' poll budget is 2^n instead of 2^21 */
typedef struct {
    int B, Tp, Hd, A, D, NL, E, V, U, cell, mode, prec, Kc, C, step_logits, keep_state0;
    int flags;                     /* LAS_SPELLER_* development / test switches, 0 in normal use */
    float forget_bias;
    unsigned long long seed;
    const float *enc, *keys; const int* enc_len;
    const float *Ws, *u, *emb, *Wv, *bv, *loc_w, *loc_b, *Wf;
    const float* const* cellW; const float* const* cellb;   /* host arrays [NL] of device ptrs */
    int* tokens_in; int* tokens_out;
    float *logits, *alphas;
    const float* align0;           /* optional [B,Tp]: previous alignment entering step 0 (else zeros) */
    const float* emb_mask;         /* optional [U,B,E]: inverted-dropout mask on the embedded input token
                                      (tf.layers.dropout, las/las.py:107-108), already scaled by 1/keep */
    const float* emb_noise;        /* optional [U,V,E]: variational noise (--add_vn): the reference adds a fresh N(0, 0.075)
                                      draw to the WHOLE embedding matrix at every look-up (las/las.py:164-166), i.e. one
                                      [V,E] noise matrix per decode step, shared by the rows that look up the same token */
    float *hs, *cs, *gates, *xin0;
    void* act_save;                /* optional, saved for backward like the four above (las_speller_act_save_bytes): speed mode's row kernels keep
                                      tanh(keys + q [+ f . Wf]) of every step here (fp16 [U,B,Tp,A] behind a 32-byte header; location-aware: also
                                      the conv outputs f, fp32 [U,B,Tp,C]) and the gradient rows read them instead of recomputing the conv, C FMAs
                                      and a tanh per (frame, column) of every step; NULL (or a forward
                                      that ran another kernel family: the header says so) = recompute */
    void* ws; size_t ws_bytes;
    int* status;                   /* optional device int (the sweeps' status word): the one-launch loop kernels need all their
                                      workgroups co-resident (one per compute unit); a workgroup that does not see a partner
                                      within the poll bound stores LAS_SPELLER_STATUS_TIMEOUT here and the launch DRAINS
                                      (every workgroup finishes its steps without waiting) instead of hanging or trapping --
                                      results of that call are invalid, the host checks the word at its next synchronisation */
    const struct las_lstm_cell_args* companion;   /* optional, LAS_SPELLER_NO_LOGITS calls only: ANOTHER cell step (the LM's first layer in a beam
                                      search -- it depends on the tokens only) that is launched together with the Speller's cell, as
                                      the second problem of one grid: two dependent-launch slots of a search step become one */
    const struct las_lstm_cell_args* companion_rows;   /* optional, LAS_SPELLER_NO_LOGITS calls only (round 5): a cell step that depends on the
                                      step's TOKENS only (the LM's first layer, las/beam_search.py:109-116; exact transcendentals, fp32 or
                                      one-hot input) launched as extra workgroups of the attention-row launch.  With the LM's SECOND layer as
                                      `companion` and the state gather inside las_beam_loop_step (fold_gather) a search step is three
                                      dependent launches: rows + LM 1, Speller cell + LM 2, beam. */
    int row_group;                 /* optional (0 = none; round 6): rows g row_group .. (g + 1) row_group - 1 have IDENTICAL enc / keys / enc_len -- a beam
                                      search's hypotheses of one utterance (las/beam_search.py:216 tiles the encoder output over them).  The
                                      per-step attention-row launches then place an utterance's rows on ONE XCD (workgroup id -> row mapping,
                                      csrc/speller.hip xcd_local_row), so that its keys and encoder rows enter one L2 instead of eight.
                                      B % row_group == 0, else ignored.  Results do not depend on it. */
} las_speller_fwd_args;
size_t las_speller_workspace_bytes(int B, int Tp, int Hd, int A, int D, int NL, int E, int V, int U, int cell);
size_t las_speller_act_save_bytes(int U, int B, int Tp, int A, int C);   /* C = location-aware channels (0: additive attention) */
int las_speller_fwd(const las_speller_fwd_args* a, void* stream);

/* Backward of the loop above for the tokens recorded in tokens_in (gradients do not flow through
 * sampled tokens, as with tf.distributions.Categorical.sample, las/las.py:170-175).
 *   dlogits [U,B,V] in.   `gates` is overwritten with d(pre-activation).
 *   Accumulates (+=) into caller-zeroed gradient buffers:
 *   d_enc [B,Tp,Hd], d_keys [B,Tp,A], dWs, du, demb [V,E], dWv, dbv, dcellW[l], dcellb[l],
 *   dloc_w, dloc_b, dWf.
 */
typedef struct {
    las_speller_fwd_args f;           /* the forward arguments, with saved buffers filled in */
    const float* dlogits;
    float *d_enc, *d_keys, *dWs, *du, *demb, *dWv, *dbv, *dloc_w, *dloc_b, *dWf;
    float* const* dcellW; float* const* dcellb;              /* host arrays [NL] of device ptrs */
} las_speller_bwd_args;
int las_speller_bwd(const las_speller_bwd_args* a, void* stream);
/* The same in two launches so that the caller can take the parameter gradients off the dependency chain:
 *   part 1 = the reverse loop and the input gradients (d_enc, d_keys);
 *   part 2 = every parameter gradient (reads what part 1 left in `ws`, `gates`, `xin0`, `hs`; may run on another
 *            stream ordered after part 1);   part 3 = both (= las_speller_bwd). */
int las_speller_bwd_part(const las_speller_bwd_args* a, int part, void* stream);
/* Which kernel family served the process's LAST las_speller_fwd (which = 0) / las_speller_bwd* part 1 (which = 1): a bit mask.  The
 * reference has ONE Speller graph (las/las.py:72-160); this library picks among several kernel families by geometry, and a caller (bench.py,
 * the tests) must be able to say which one a number or a parity statement belongs to. */
enum { LAS_SPELLER_RAN_LOOP = 1,       /* the whole decode / gradient loop in one launch (dec_loop_*_kernel) */
       LAS_SPELLER_RAN_PF_ROWS = 2,    /* per-step launches, prefetching bf16 row kernels */
       LAS_SPELLER_RAN_BF_ROWS = 4,    /* per-step launches, generic bf16 row kernels */
       LAS_SPELLER_RAN_F32_ROWS = 8,   /* per-step launches, fp32-operand row kernels (parity mode; speed mode outside the other families' geometry) */
       LAS_SPELLER_RAN_SKINNY = 16,    /* layer 0's per-step cell product: pre-packed bf16 MFMA fragments (las_skinny_gemm_bf16) */
       LAS_SPELLER_RAN_LOC = 32,       /* location-aware attention */
       LAS_SPELLER_RAN_UPPER_SKINNY = 64,    /* layers >= 1: per-step cell products on pre-packed bf16 fragments (round 6) */
       LAS_SPELLER_RAN_WIDE = 128 };         /* the wide per-step path (csrc/speller_wide.h, round 6) */
int las_speller_last_variant(int which);

/* ------------------------------------------------------------------------------------------
 * K8  LAS._get_loss (las/las.py:320-333) + label_smoothing (las/utils.py:5-12), forward and
 * gradient in one pass.  logits element (b,t,v) at logits[b*sb + t*st + v] (so the Speller's
 * time-major [U,B,V] buffer is consumed in place: sb=V, st=B*V); dlogits uses the same strides.
 * y int32 [B,ldy] (first U columns used).  `smooth`: bit 0 = label smoothing with `epsilon`; bit 1 = no PAD mask, every
 * position counts (the RNNLM's mean sparse cross entropy, lang/char_rnn_model.py:146-149).
 * sums[0] += sum(ce*mask), sums[1] += sum(mask)   (caller zeroes sums; the division
 * sum/(n+1e-9) is the caller's so that data-parallel ranks can all-reduce both terms first).
 * bit 2 of `smooth` (needs scale_ptr; sums then has THREE floats): sums[0] = sum(ce*mask), sums[1] = sum(mask),
 * sums[2] = sums[0] * scale_ptr[0] are WRITTEN -- the scaled loss without a fill in front and a multiply behind.
 * dlogits = scale_ptr[0] * mask * (softmax - smoothed_onehot)    (scale = 1/(n_total+1e-9), a
 * device scalar; NULL dlogits skips the gradient).
 */
size_t las_ce_loss_workspace_bytes(int B, int U);
int las_ce_loss(const float* logits, long long sb, long long st, const int* y, int ldy,
                int B, int U, int V, float epsilon, int smooth,
                float* sums, const float* scale_ptr, float* dlogits,
                void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9  tf.clip_by_global_norm + tf.train.AdamOptimizer.apply_gradients (las/las.py:272-283) on one
 * flat fp32 parameter bucket.  las_sumsq: out[0] = sum g^2 (deterministic two-stage reduction;
 * ws >= las_sumsq_workspace_bytes(n)).  las_clip_adam: theta,m,v updated in place;
 * g *= clip/max(sqrt(sumsq[0]),clip) when clip>0;  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the
 * caller;  theta -= lr_t * m / (sqrt(v) + eps)   (TF "epsilon-hat" placement).
 * status / guard (either may be NULL): device words checked by the kernel itself -- when status[0] != 0 (a recurrent sweep
 * of this step reported an exchange time-out, LAS_SEQ_STATUS_*) or guard[0] != 0 (the same, summed over the data-parallel
 * ranks with the gradient bucket) the update is SKIPPED: theta, m, v keep their values, no host synchronisation needed.
 * applied (may be NULL; round 6): a device int that the kernel increments when -- and only when -- it DOES apply the update.  The status
 * word is sticky until the host clears it, so after a time-out every later step is skipped too: when the host finally notices (its
 * polls are asynchronous), applied[0] tells it exactly which step was the first one lost, and LAS.train re-runs from there
 * (las/las.py LAS._recover).
 */
/* Stream-ordered, BOUNDED wait on a device word: work enqueued behind it on `stream` starts once *word == value or after
 * max_us microseconds.  A scheduling aid (never a correctness dependency): the weight-gradient GEMMs of a side stream are
 * held back until the next recurrent sweep has announced itself (LAS_SEQ_ANNOUNCE), so that they neither delay its start nor
 * compete with the chain GEMMs in front of it.  No reference counterpart (the reference has one stream). */
int las_wait_word(const int* word, int value, int max_us, void* stream);
/* ... for the announcement word itself (status[1] of LAS_SEQ_ANNOUNCE): passes once the word HAS REACHED n in the cyclic order of
 * 1..1023 -- also when later sweeps have announced themselves meanwhile (a hold enqueued late must not sit out its bound). */
int las_wait_announce(const int* word, int n, int max_us, void* stream);
/* Stream-ordered wait until EVERY one of n device words is >= need (las_rnn_seq_bwd_db_progress' progress words).  Unlike las_wait_announce this
 * is a correctness dependency: on a time-out (max_us) `code` is stored into status[0] (may be NULL) -- the step is invalid, las_clip_adam skips it. */
int las_wait_words_min(const int* words, int n, int need, int max_us, int* status, int code, void* stream);
/* Diagnostics (round 5): a "foreign" kernel that stays RESIDENT -- n workgroups of 256 threads (workgroup L on XCD L % 8), `lds` bytes of LDS
 * and 32 or 64 VGPRs per lane each: the footprint of a collective's channel -- until *stop != 0 (or max_ms).  resident[0] (caller-zeroed)
 * counts the workgroups that have started.  The recurrent sweeps and the one-launch Speller loops need all THEIR workgroups resident at
 * once (LAS_SEQ_STATUS_*, LAS_SPELLER_STATUS_TIMEOUT): tests run a train step beside this kernel to show what they tolerate. */
int las_occupy(const int* stop, int* resident, int n, int lds, int vgprs, int max_ms, void* stream);

/* bf16 (or fp32) weight shadows of the speed mode, rebuilt after every optimiser step by ONE launch over a device-resident
 * descriptor table: D = zero-pad(op([src0 | src1])), op = transpose or identity; src1 may be NULL (cols1 = 0).
 * max_tiles >= max over the descriptors of ceil(dst_rows/32) * ceil(dst_cols/32).  (No reference counterpart: the
 * reference's fp32 graph has no operand copies; this replaces ~50 small copy / cat / cast kernels per step.) */
typedef struct las_shadow_desc {
    const float* src0; const float* src1;
    int ld0, ld1, rows, cols0, cols1, transpose;
    void* dst;
    int dst_rows, dst_cols, dst_ld, dst_bf16;
} las_shadow_desc;
int las_build_shadows(const las_shadow_desc* descs_dev, int n, int max_tiles, void* stream);

/* Input dropout of a bidirectional recurrent layer (las/layers.py:37-47: fw_cell and bw_cell in separate DropoutWrappers, input_keep_prob =
 * 1 - dropout_rate): ONE launch leaves the two directions' operand blocks y_fw / y_bw [rows, ldy] = x * mask_d / keep with INDEPENDENT
 * Bernoulli(keep) masks, in fp32 or bf16 (x_dt / y_dt: LAS_DT_*), columns K .. ldy - 1 zero (the product's K padding; ldy % 4 == 0).  The
 * masks are a counter-based function of (seed, direction, row, column) -- 16 bits of a splitmix64 word per element -- and are never stored:
 * las_dropout_pair_bwd regenerates them,  dx = (mask_fw * g_fw + mask_bw * g_bw) / keep  (g_*: [rows, ldg], the first K columns).
 * (tf.nn.dropout draws from TF's Philox stream; the reference fixes no seed, so only the distribution is contractual.) */
int las_dropout_pair_fwd(const void* x, int x_dt, long long rows, int K, int ldx, void* y_fw, void* y_bw, int y_dt, int ldy,
                         float keep, unsigned long long seed, void* stream);
int las_dropout_pair_bwd(const void* g_fw, const void* g_bw, int g_dt, int ldg, long long rows, int K, void* dx, int dx_dt, int lddx,
                         float keep, unsigned long long seed, void* stream);

/* tf.layers.batch_normalization over the last axis of a [rows, C] block in TRAINING mode, with the ReLU the CNN listener puts behind it
 * (las/layers.py:114-116,155-161; momentum 0.99 -> `momentum` 0.01 here, epsilon 1e-3):
 *   fwd: mean / rstd [C] = batch statistics (biased variance; kept by the caller for backward), y = [relu](gamma (x - mean) rstd + beta);
 *        moving_mean / moving_var (both or neither): m = (1 - momentum) m + momentum {mean, unbiased variance}   (UPDATE_OPS, las/las.py:272)
 *   bwd: dx; dgamma / dbeta (either may be NULL) are ACCUMULATED (+=).  relu: y is the forward's output (its sign is the mask).
 * C % 4 == 0, 16-byte aligned pointers, ws >= las_bn_workspace_bytes(rows, C).  Deterministic (fixed-order reductions, no atomics). */
size_t las_bn_workspace_bytes(long long rows, int C);
int las_bn_relu_fwd(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                    float* moving_mean, float* moving_var, float momentum, int relu, float* y, void* ws, size_t ws_bytes, void* stream);
int las_bn_relu_bwd(const float* x, const float* y, const float* dy, long long rows, int C, const float* gamma, const float* mean,
                    const float* rstd, int relu, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

size_t las_sumsq_workspace_bytes(long long n);
int las_sumsq(const float* g, long long n, float* out, void* ws, size_t ws_bytes, void* stream);
int las_clip_adam(float* theta, const float* g, float* m, float* v, long long n,
                  const float* sumsq, float clip, float lr_t, float beta1, float beta2, float eps,
                  const int* status, const float* guard, int* applied, void* stream);

/* ------------------------------------------------------------------------------------------
 * R1  one BasicLSTMCell step of the char RNNLM used for shallow fusion (lang/char_rnn_model.py:57-66 with
 * forget_bias 0, driven from las/beam_search.py:226-236): gate math on z = [x,h].kernel + bias [N,4H]
 * (the contraction itself is las_gemm): c' = c*sigmoid(f+fb) + sigmoid(i)*tanh(j), h' = tanh(c')*sigmoid(o).
 */
int las_lstm_pointwise(const float* z, const float* c_prev, int N, int H, float forget_bias,
                       float* c_out, float* h_out, void* stream);
/* ... with the input half of a ONE-HOT input (the LM's first layer: the reference feeds tf.one_hot ids, lang/char_rnn_model.py:106-112)
 * added on the way in: z[n] + xrows[max(ids[n] - id_shift, 0)], xrows [V,4H] = the input rows of the cell kernel.  id_shift / the clamp
 * are the LAS-id -> LM-id map of the shallow fusion (las/beam_search.py:109-116: LM ids = LAS ids - 2, SOS fed as id 0), so that the
 * beam search's next_token buffer is read as it is. */
int las_lstm_pointwise_rows(const float* z, const float* xrows, const int* ids, int id_shift, const float* c_prev, int N, int H,
                            float forget_bias, float* c_out, float* h_out, void* stream);
/* The whole cell for a block of rows in ONE launch (the beam search's LM step, M = utterances x beam rows): z = [x ; h] . kernel + bias
 * with the kernel's input rows / recurrent rows given as las_gemm_skinny_pack fragments (Wx_packed: [I, 4H], Wh_packed: [H, 4H]; operands
 * rounded to bf16, fp32 accumulation), then the gate math of las_lstm_pointwise.  Dense input: x [M, I] (I % 32 == 0) and Wx_packed; one-hot
 * input (lang/char_rnn_model.py:106-112): x = NULL and ids / id_shift / xrows as in las_lstm_pointwise_rows (fp32 row look-up, no rounding
 * inside the kernel).  H % 32 == 0; c_out / h_out [M, H] may not alias h / c_prev (other workgroups still read them). */
int las_lstm_cell_rows(const float* x, int ldx, int I, const int* ids, int id_shift, const float* xrows, const float* h, int ldh,
                       const void* Wx_packed, const void* Wh_packed, const float* bias, const float* c_prev, int M, int H,
                       float forget_bias, float* c_out, float* h_out, void* stream);
/* The same call with its arguments in a struct (what las_speller_fwd_args.companion points to): x fp32 or, with x_bf16, bf16; either
 * of x / h may be NULL (with its packed weights); fast = the Speller's approximated transcendentals instead of expf / tanhf;
 * gates_out (optional [M, 4H]) receives the ACTIVATED gates i, j, f, o. */
typedef struct las_lstm_cell_args {
    const void* x; int x_bf16, ldx, I;
    const int* ids; int id_shift; const float* xrows;
    const void* h; int ldh;          /* fp32 rows, or bf16 rows when h_bf16 (below) */
    const void *Wx, *Wh;
    const float *bias, *c_prev;
    float fb;
    float *c_out, *h_out, *gates_out;
    int M, H, fast;
    int h_bf16;                      /* (round 5) h holds bf16 rows: the copy a previous launch left in h_out_bf16.  Served by the 128-row
                                        workgroups only (M >= 384, x absent or bf16 too): the beam search's LM cells at decode.py's batch */
    void* h_out_bf16;                /* optional [M, H] bf16: h' rounded as the next launch's operand staging would round it (RNE) -- the
                                        next layer's x and, gathered, the next step's h, at half the bytes */
} las_lstm_cell_args;
int las_lstm_cell_rows_args(const las_lstm_cell_args* a, void* stream);
/* ... and its gradient, for training the RNNLM (lang/char_rnn_model.py:177-190, truncated BPTT over num_unrollings steps):
 * dz [N,4H] and dc_prev [N,H] from z, c_prev, dh (gradient w.r.t. h') and dc_in (gradient w.r.t. c', may be NULL). */
int las_lstm_pointwise_bwd(const float* z, const float* c_prev, const float* dh, const float* dc_in, int N, int H,
                           float forget_bias, float* dz, float* dc_prev, void* stream);

/* ------------------------------------------------------------------------------------------
 * K10  one pruning step of BeamSearch.decode (las/beam_search.py:119-152, :297-312) for `nutt`
 * utterances at once.  Per utterance: nlive live hypotheses with raw logits [nlive,V] (raw logits
 * are the scores, las/beam_search.py:123-124), running float32 score (0 + np.float32 sums stay
 * float32 in BeamState.update, :27) and length (tokens after SOS so far).  Only hypothesis 0 is
 * expanded at t=0 (:119); SOS is skipped after t=0 (:127-128); candidates are ranked by
 * (score+logit)/(length+1) in float32 (:306) and the best `beam` are returned in ASCENDING order
 * (best last, :310-312).  Ties follow a stable ascending sort of the reference's candidate bank:
 * (key, hypothesis index, logit, token id).  The per-hypothesis top-`topn` cut (:123, 64) cannot
 * bind while beam < topn, which this entry point requires.
 *   logits [nutt,beam,V]   score f32 [nutt,beam]   length int32 [nutt,beam]   nlive int32 [nutt]
 * Outputs (count in out_n[nutt]): out_parent, out_token int32 [nutt,beam]; out_score f32 [nutt,beam]
 * = new running sums.
 */
int las_beam_step(const float* logits, const float* score, const int* length, const int* nlive,
                  int nutt, int beam, int V, int topn, int t, int start_id,
                  int* out_parent, int* out_token, float* out_score, int* out_n, void* stream);

/* ------------------------------------------------------------------------------------------
 * K10b  device-resident BeamSearch.decode loop (las/beam_search.py:94-158) for `nutt` utterances at once: ONE call per
 * step prunes every utterance exactly as las_beam_step does AND keeps the reference's bookkeeping on the device, so
 * the host never waits inside the loop:
 *   - live hypotheses: score / length [nutt,beam] and nlive [nutt] updated in place (compacted, ascending rank);
 *   - back-pointer records of step t = *step: hist_parent (live slot of step t), hist_token, hist_score [Umax,nutt,beam],
 *     hist_n [Umax,nutt] picks, hist_slot [Umax,nutt,beam] (live slot k of step t+1 = pick hist_slot[t][u][k]);
 *   - retired hypotheses (EOS, :148-152; the live ones on step exhaustion, :155-156) appended as (sel_t, sel_j)
 *     references into the records, nsel [nutt]; selcap >= 3*beam;
 *   - done[u] set when `t+1 == dec_step[u]`, `nsel >= beam` (:94) or no live hypothesis is left; a done utterance is
 *     skipped by later calls;
 *   - src_row [nutt,beam] (global row the new live slot continues) and next_token [nutt*beam] for the next step, and
 *     the `ntens` recurrent-state tensors gathered accordingly: state_out[k][row] = state_in[k][src_row[row]]
 *     (state_width[k] floats per row; decoder h/c, previous alignment, LM states);
 *   - optionally the logits themselves are computed by the call (proj_*, below) instead of being read;
 *   - optionally a per-step record of one row tensor (the step's alignments): file_out[t][r] = file_in[r] for the nutt*beam rows of
 *     file_width floats (file_in NULL = none);
 *   - finally *step += 1 (device-resident step counter: the launch sequence is identical every step, hipGraph friendly).
 * All state is caller-owned and caller-initialised (score 0, length 0, nlive = beam, nsel = done = 0, *step = 0,
 * next_token = start_id).  The host reads the records once after the last step and rebuilds token ids by back-tracking.
 */
typedef struct {
    float* logits; float* score; int* length; int* nlive; int* nsel; int* done; const int* dec_step; int* step;
    int *hist_parent, *hist_token, *hist_slot; float* hist_score; int* hist_n;
    int *sel_t, *sel_j; int* src_row; int* next_token;
    int nutt, beam, V, Umax, selcap, topn, start_id, end_id;
    int ntens; const float* state_in[16]; float* state_out[16]; int state_width[16];
    const float* file_in; float* file_out; int file_width;
    /* optional: the vocabulary projection inside the step (two launches less per decode step): with proj_w != NULL
     * logits[row] = proj_b + bf16([proj_h0[row] ; proj_h1[row]]) . proj_w, proj_w = las_gemm_skinny_pack fragments of a [proj_k0 + proj_k1, V]
     * matrix (the Speller's output kernel on top of the LM's softmax kernel, pre-scaled by lm_weight and shifted to its token columns;
     * proj_h1 NULL = no second part), proj_h0 / proj_h1 fp32 [nutt*beam, proj_k0 / proj_k1] (multiples of 32), fp32 accumulation.
     * ceil(beam/16) * ceil(V/16) <= 8.  `logits` (may be NULL then) additionally receives the values. */
    const float* proj_h0; int proj_k0; const float* proj_h1; int proj_k1; const void* proj_w; const float* proj_b;
    /* round 5: fold_gather != 0 -- the state gather runs INSIDE the pruning launch (an utterance's workgroup copies the rows of its own
     * surviving parents as soon as it has ranked them: rows of different utterances never mix) and the step counter is advanced by the
     * workgroup that finishes last; `step` then points to TWO ints (step[1]: arrival counter, caller-zeroed).  One launch per step. */
    int fold_gather;
} las_beam_loop_args;
int las_beam_loop_step(const las_beam_loop_args* a, void* stream);
/* After the last step: walk the back-pointer records of every retired hypothesis on the device (the reference carries whole token
 * lists in its BeamState objects, las/beam_search.py:38-45).  For selection slot w = u*selcap + s (s < min(nsel[u], selcap)):
 * len[w] = tokens after SOS, ids[w][p] / rows[w][p] (p < len[w]; arrays [nutt*selcap, Umax]) = token and global state row
 * (u*beam + live slot) at step p, score[w] = running float32 sum; unused slots get len 0.  Only the record pointers and
 * nutt / beam / Umax / selcap of `a` are read. */
int las_beam_backtrack(const las_beam_loop_args* a, int* ids, int* rows, int* len, float* score, void* stream);

/* ------------------------------------------------------------------------------------------
 * Skinny-M product for per-step recurrences driven from the host (the char RNNLM step inside beam search,
 * lang/char_rnn_model.py:57-66 / las/beam_search.py:226-236; the Speller's own cell product uses the same kernel inside
 * las_speller_fwd): the weight W [K, N] (row-major fp32, leading dimension ldw) is packed ONCE into bf16 MFMA fragments
 * (las_gemm_skinny_pack_bytes(K, N) bytes), then every call computes C[M, N] = bf16(A[M, K]) . bf16(W) + bias (accumulate = 0) or
 * C += ... (accumulate = 1) with fp32 accumulation, for 1 <= M <= 1024 rows: grid = (N / 16 column tiles) x (M / 16 row tiles),
 * eight waves split K with every load in flight at once.  Same arithmetic as las_gemm in LAS_PREC_BF16 (operands rounded to
 * bf16, fp32 sums) at a fifth of the time for M = 256.
 */
size_t las_gemm_skinny_pack_bytes(int K, int N);
int las_gemm_skinny_pack(const float* W, int ldw, int K, int N, void* packed, void* stream);
int las_gemm_skinny(const float* A, int lda, int M, int K, const void* packed, int N, float* C, int ldc, const float* bias,
                    int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * F1  input side of the step loop: TFRecord reader + bucket_by_sequence_length + pinned batch ring + H2D upload
 * (reference tfrecord_data_loader.py:54-109: list_files / parallel_interleave(cycle_length 16) / map(data_parser) /
 * bucket_by_sequence_length(pad_to_bucket_boundary) / shuffle(64) / repeat / prefetch -- TensorFlow's C++ input threads in the
 * reference, one producer thread per reader here; train.py:45-55,114-117 is the consumer).  Host code; the only device
 * interaction is the asynchronous copy of las_input_upload.  Batch order is a pure function of (files, seed) and equals the
 * Python restatement in tfrecord_data_loader.py (same splitmix64 stream).  rank / world: lock-step data parallelism -- a bucket
 * emits when it holds world x batch_limit utterances and this rank keeps rows rank, rank + world, ... of that global batch, so
 * every rank runs the same bucket shape at every step.
 */
typedef struct las_input_config {
    int feat_dim;                 /* MFCC coefficients per channel (13): a frame is feat_dim x 3 floats */
    int is_training;              /* 1: shuffle + repeat forever; 0: one pass, then las_input_next returns 1 */
    int n_bounds;                 /* number of bucket boundaries (<= 16) */
    int bounds[16];               /* tfrecord_data_loader.py:75,80 */
    int batch_limit[17];          /* tfrecord_data_loader.py:83, per rank */
    int max_tokenlen;             /* token rows are padded to this (219 train / 227 eval, :76,:81) */
    int shuffle_buffer;           /* batches (64 train, 0 eval) */
    int cycle_length;             /* files read round robin (16) */
    unsigned long long seed;
    int rank, world;
    int slots;                    /* pinned batches in flight (>= 2) */
} las_input_config;

typedef struct las_input_batch {
    int slot, B, T, bucket, global_B, max_tokenlen;
    const float* feat;            /* [B, T, feat_dim, 3] zero padded, pinned host memory (valid until the slot is released) */
    const int* token;             /* [B, max_tokenlen] zero padded */
    const int* featlen;           /* [B] */
    const int* tokenlen;          /* [B] */
} las_input_batch;

unsigned int las_crc32c(const void* data, size_t n);              /* CRC-32C (Castagnoli) of a host buffer: the TFRecord framing's checksum */
void* las_input_open(const char* const* files, int nfiles, const las_input_config* cfg);   /* NULL on error (las_last_error) */
int las_input_next(void* reader, las_input_batch* out);          /* blocks; 0 = a batch, 1 = end of data, < 0 = error */
/* asynchronous copy of the slot's features / tokens to caller-owned device buffers on `stream`; the reader does not refill the
 * slot before that copy has executed.  Follow with las_input_release. */
int las_input_upload(void* reader, int slot, float* d_feat, int* d_token, void* stream);
int las_input_release(void* reader, int slot);
long long las_input_records(void* reader);                       /* records parsed so far */
void las_input_close(void* reader);

#ifdef __cplusplus
}
#endif
#endif /* LAS_HIP_H */
