// rnn_seq_args.h -- launch arguments and the cluster-exchange primitives shared by the recurrent sweep kernels
// (rnn_seq.hip: speed mode, bf16 MFMA; rnn_seq_f32.hip: parity mode, exact-fp32 MFMA).
#pragma once
#include "las_common.h"

struct RnnArgs {
    int B, T, H;
    float* gates;
    const float* whh[2];
    int ldw;
    float* out; int ld_out; long long obs;
    float* cstate;
    const float* dout; int ld_dout; long long dobs;
    // the same tensors as seen by the speed-mode (bf16 storage) kernels; exactly one family is used per launch
    unsigned short *gates16, *out16, *cstate16; const unsigned short* dout16; unsigned short* sink16;
    float fb;
    const void* wpack;
    long long* dbg;   // LAS_PROF builds only: device buffer for s_memtime stamps (env LAS_DBG_PTR)
    unsigned long long* xbuf; int* err;     // cluster exchange granules / bounded-spin error flag
    unsigned long long* xcc;                 // [cluster][member] placement handshake granules (zeroed per launch)
    float* bpart;                            // BPTT: [cluster][G*H] column sums of d(pre-activation) over the tile's rows and all steps
    int force_agent;                         // env LAS_AGENT_GRANULES=1: never use the same-XCD transport
    float* sink;                             // scratch rows for the padded part of a ragged batch tile
    int ncl, ncl_pad;                        // clusters = batch tiles x 2 directions (padded to a multiple of 8)
    int ks_packed;                           // wpack holds the K-split BPTT fragment order
    int spin;                                // bound of every exchange spin (polls); a timeout is reported through `status`
    int* status;                             // caller-owned sticky device word (may be NULL): LAS_SEQ_STATUS_* on failure
    int status_code;
    int no_helpers;                          // LAS_SEQ_NO_HELPER_WAVES
    int announce;                            // LAS_SEQ_ANNOUNCE(n): cluster 0 stores n into status[1] once its members are resident
    const int* xflag; int xsc;               // forward: the x-projection arrives in time chunks of xsc steps from both ends of the sequence,
                                             // *xflag = number of chunks complete (another stream's kernels write it); NULL: all there
    const int* dflag; int dcp, dTq, dshift;  // BPTT: dout arrives in chunks of 2^dcp producer rows (dTq per utterance; row = frame >> dshift) from
                                             // both ends of the sequence; *dflag = chunks complete
    int* prog; int pstep;                    // BPTT (PG instances): member m of cluster c stores the number of sweep steps whose dZ has reached memory into
                                             // prog[c * P + m] every pstep steps and at the end (agent scope, behind write-through dZ stores)
    int warm;                                // extra "L2 warmer" workgroups (one per cluster) are part of the grid
    int rb;                                  // batch rows per tile (16; 8 for the kernels that compact duplicated MFMA rows)
    const int* row_T;                        // forward, optional: frames of every batch row (<= T); a row's state and outputs are ZERO at t >= row_T[row]
};

#define LAS_SPIN_BUDGET_DEFAULT (1 << 22)

// (the 16-byte granule form {tag, a, b, tag} -- granule_rsrc / granule16_store / granule16_load -- lives in las_common.h:
// the Speller's fused step kernels use the same transport)
__device__ __forceinline__ unsigned granule_wait(const unsigned long long* p, unsigned tag, int* err, int spin) {
    unsigned long long x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int budget = *err ? 1 : spin;                     // sticky: after one timeout never wait again (no hang)
    while ((unsigned)(x >> 32) != tag) {
        if (--budget == 0) { *err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
        x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (unsigned)x;
}

// Placement handshake: every member publishes the XCC_ID it runs on (agent-scope granule, valid under any placement)
// and reads its partners'; true only if all P agree.  A timeout or a mismatch selects the agent-scope transport.
__device__ __forceinline__ bool cluster_same_xcd(unsigned long long* slots, int pm, int P, int tid, int* err, int spin) {
    const unsigned tag = 0x58434400u;                                           // "XCD\0"
    const unsigned mine = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;      // HW_REG_XCC_ID[3:0]
    if (tid == 0) __hip_atomic_store(slots + pm, ((unsigned long long)tag << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int same = 1;
    if (tid < P) same = (granule_wait(slots + tid, tag, err, spin) == mine) && !*err;
    return __syncthreads_and(same) != 0;
}


// ---- rnn_seq_f32.hip: the parity mode's clustered exact-fp32 MFMA sweeps (H in {64, 128, 256, 512}) ----------------------------
bool las_rnn_seq_mf32_ok(int cell, int H);
size_t las_rnn_seq_mf32_ws_bytes(int cell, int H);
int las_rnn_seq_mf32_run(bool bwd, int cell, const RnnArgs& a, void* ws, size_t ws_bytes, int flags, hipStream_t st);
