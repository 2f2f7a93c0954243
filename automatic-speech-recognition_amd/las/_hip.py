"""ctypes binding of liblas_hip.so (the C ABI declared in include/las_hip.h).

This is the only place the Python host code touches native code.  There is NO fallback: if the
shared library is missing or a call fails, a RuntimeError is raised (the product path never
routes through the CPU oracle).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t,
                    c_void_p)

import torch

PREC_F32, PREC_BF16 = 0, 1
CELL_RNN, CELL_LSTM = 0, 1
ACT_NONE, ACT_TANH = 0, 1
ATT_ADD, ATT_LOC = 0, 1
DT_F32, DT_BF16 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "liblas_hip.so")   # (override: development builds)
_lib = None


class LstmCellArgs(Structure):
    """include/las_hip.h las_lstm_cell_args"""
    _fields_ = [("x", c_void_p), ("x_bf16", c_int), ("ldx", c_int), ("I", c_int), ("ids", c_void_p), ("id_shift", c_int), ("xrows", c_void_p),
                ("h", c_void_p), ("ldh", c_int), ("Wx", c_void_p), ("Wh", c_void_p), ("bias", c_void_p), ("c_prev", c_void_p), ("fb", c_float),
                ("c_out", c_void_p), ("h_out", c_void_p), ("gates_out", c_void_p), ("M", c_int), ("H", c_int), ("fast", c_int),
                ("h_bf16", c_int), ("h_out_bf16", c_void_p)]


class SpellerFwdArgs(Structure):
    _fields_ = [(n, c_int) for n in ("B", "Tp", "Hd", "A", "D", "NL", "E", "V", "U", "cell", "mode", "prec", "Kc", "C",
                                     "step_logits", "keep_state0", "flags")] + [
        ("forget_bias", c_float), ("seed", ctypes.c_ulonglong),
        ("enc", c_void_p), ("keys", c_void_p), ("enc_len", c_void_p),
        ("Ws", c_void_p), ("u", c_void_p), ("emb", c_void_p), ("Wv", c_void_p), ("bv", c_void_p),
        ("loc_w", c_void_p), ("loc_b", c_void_p), ("Wf", c_void_p),
        ("cellW", POINTER(c_void_p)), ("cellb", POINTER(c_void_p)),
        ("tokens_in", c_void_p), ("tokens_out", c_void_p),
        ("logits", c_void_p), ("alphas", c_void_p), ("align0", c_void_p), ("emb_mask", c_void_p), ("emb_noise", c_void_p),
        ("hs", c_void_p), ("cs", c_void_p), ("gates", c_void_p), ("xin0", c_void_p), ("act_save", c_void_p),
        ("ws", c_void_p), ("ws_bytes", c_size_t), ("status", c_void_p), ("companion", POINTER(LstmCellArgs)),
        ("companion_rows", POINTER(LstmCellArgs)), ("row_group", c_int)]


class BeamLoopArgs(Structure):
    _fields_ = [(n, c_void_p) for n in ("logits", "score", "length", "nlive", "nsel", "done", "dec_step", "step",
                                        "hist_parent", "hist_token", "hist_slot", "hist_score", "hist_n",
                                        "sel_t", "sel_j", "src_row", "next_token")] + \
               [(n, c_int) for n in ("nutt", "beam", "V", "Umax", "selcap", "topn", "start_id", "end_id", "ntens")] + \
               [("state_in", c_void_p * 16), ("state_out", c_void_p * 16), ("state_width", c_int * 16)] + \
               [("file_in", c_void_p), ("file_out", c_void_p), ("file_width", c_int)] + \
               [("proj_h0", c_void_p), ("proj_k0", c_int), ("proj_h1", c_void_p), ("proj_k1", c_int), ("proj_w", c_void_p), ("proj_b", c_void_p)] + \
               [("fold_gather", c_int)]


class SpellerBwdArgs(Structure):
    _fields_ = [("f", SpellerFwdArgs), ("dlogits", c_void_p),
                ("d_enc", c_void_p), ("d_keys", c_void_p), ("dWs", c_void_p), ("du", c_void_p),
                ("demb", c_void_p), ("dWv", c_void_p), ("dbv", c_void_p),
                ("dloc_w", c_void_p), ("dloc_b", c_void_p), ("dWf", c_void_p),
                ("dcellW", POINTER(c_void_p)), ("dcellb", POINTER(c_void_p))]


class InputConfig(Structure):
    """include/las_hip.h las_input_config"""
    _fields_ = [("feat_dim", c_int), ("is_training", c_int), ("n_bounds", c_int), ("bounds", c_int * 16), ("batch_limit", c_int * 17),
                ("max_tokenlen", c_int), ("shuffle_buffer", c_int), ("cycle_length", c_int), ("seed", ctypes.c_ulonglong),
                ("rank", c_int), ("world", c_int), ("slots", c_int)]


class InputBatch(Structure):
    """include/las_hip.h las_input_batch"""
    _fields_ = [(n, c_int) for n in ("slot", "B", "T", "bucket", "global_B", "max_tokenlen")] + \
               [("feat", c_void_p), ("token", c_void_p), ("featlen", c_void_p), ("tokenlen", c_void_p)]


class ShadowDesc(ctypes.Structure):
    """include/las_hip.h las_shadow_desc"""
    _fields_ = [("src0", c_void_p), ("src1", c_void_p), ("ld0", c_int), ("ld1", c_int), ("rows", c_int), ("cols0", c_int),
                ("cols1", c_int), ("transpose", c_int), ("dst", c_void_p), ("dst_rows", c_int), ("dst_cols", c_int),
                ("dst_ld", c_int), ("dst_bf16", c_int)]


class SeqPrepareDesc(ctypes.Structure):
    """include/las_hip.h las_seq_prepare_desc"""
    _fields_ = [("whh_fw", c_void_p), ("whh_bw", c_void_p), ("ldw", c_int), ("cell", c_int), ("H", c_int), ("B", c_int), ("bwd", c_int),
                ("flags", c_int), ("ws", c_void_p), ("ws_bytes", c_size_t)]


_SIGS = {
    "las_version": (c_int, []),
    "las_rnn_seq_prepare": (c_int, [POINTER(SeqPrepareDesc), c_int, c_void_p]),
    "las_last_error": (c_char_p, []),
    "las_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_longlong,
                         c_void_p, c_int, c_longlong, c_float, c_void_p, c_int, c_longlong, c_void_p, c_int,
                         c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "las_gemm_dt": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_longlong,
                            c_void_p, c_int, c_longlong, c_int, c_float, c_void_p, c_int, c_longlong, c_void_p, c_int,
                            c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "las_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "las_wgrad_ih_hh_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "las_wgrad_ih_hh": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_gemm_kk": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_longlong,
                            c_void_p, c_int, c_void_p]),
    "las_gemm_kk_tanhgrad": (c_int, [c_int, c_int, c_int, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_longlong,
                            c_void_p, c_int, c_void_p, c_longlong, c_void_p]),
    "las_colsum_workspace_bytes": (c_size_t, [c_int]),
    "las_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_tanh_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "las_rnn_seq_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "las_rnn_seq_io_dtype": (c_int, [c_int, c_int, c_int]),
    "las_colsum_dt": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_tanh_bwd_dt": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "las_rnn_seq_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_int, c_longlong, c_void_p, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_rnn_seq_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_longlong,
                                c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_rnn_seq_bwd_db": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_longlong,
                                   c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_speller_workspace_bytes": (c_size_t, [c_int] * 10),
    "las_speller_act_save_bytes": (c_size_t, [c_int] * 5),
    "las_speller_fwd": (c_int, [POINTER(SpellerFwdArgs), c_void_p]),
    "las_speller_bwd": (c_int, [POINTER(SpellerBwdArgs), c_void_p]),
    "las_speller_bwd_part": (c_int, [POINTER(SpellerBwdArgs), c_int, c_void_p]),
    "las_speller_last_variant": (c_int, [c_int]),
    "las_bn_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "las_bn_relu_fwd": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_bn_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "las_dropout_pair_fwd": (c_int, [c_void_p, c_int, c_longlong, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, ctypes.c_ulonglong, c_void_p]),
    "las_dropout_pair_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_void_p, c_int, c_int, c_float, ctypes.c_ulonglong, c_void_p]),
    "las_ce_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "las_ce_loss": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_sumsq_workspace_bytes": (c_size_t, [c_longlong]),
    "las_sumsq": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_clip_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_float, c_float,
                              c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "las_build_shadows": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "las_wait_word": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "las_wait_announce": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "las_set_word": (c_int, [c_void_p, c_int, c_void_p]),
    "las_occupy": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "las_gemm_kk_frames": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_longlong, c_void_p, c_longlong,
                                   c_void_p, c_int, c_longlong, c_void_p, c_int, c_void_p, c_longlong, c_void_p]),
    "las_rnn_seq_bwd_chunks_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "las_rnn_seq_bwd_db_chunked": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                           c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_longlong,
                                           c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "las_rnn_seq_bwd_progress_words": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "las_rnn_seq_bwd_db_progress": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                            c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_longlong,
                                            c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "las_wait_words_min": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "las_wgrad_ih_hh_window": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_rnn_seq_fwd_rows_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "las_rnn_seq_fwd_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_longlong,
                                     c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "las_rnn_seq_fwd_chunks_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "las_rnn_seq_fwd_chunked": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                        c_longlong, c_void_p, c_float, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "las_lstm_pointwise": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "las_lstm_pointwise_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "las_lstm_cell_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "las_lstm_cell_rows_args": (c_int, [POINTER(LstmCellArgs), c_void_p]),
    "las_lstm_pointwise_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "las_beam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "las_beam_loop_step": (c_int, [POINTER(BeamLoopArgs), c_void_p]),
    "las_beam_backtrack": (c_int, [POINTER(BeamLoopArgs), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "las_gemm_skinny_pack_bytes": (c_size_t, [c_int, c_int]),
    "las_gemm_skinny_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "las_gemm_skinny": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "las_crc32c": (ctypes.c_uint, [c_char_p, c_size_t]),
    "las_input_open": (c_void_p, [POINTER(c_char_p), c_int, POINTER(InputConfig)]),
    "las_input_next": (c_int, [c_void_p, POINTER(InputBatch)]),
    "las_input_upload": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "las_input_release": (c_int, [c_void_p, c_int]),
    "las_input_records": (c_longlong, [c_void_p]),
    "las_input_close": (None, [c_void_p]),
}


ABI_VERSION = 600      # include/las_hip.h LAS_HIP_ABI_VERSION


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load liblas_hip.so once; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liblas_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C automatic-speech-recognition_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name, None)
            if fn is None:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing and not os.environ.get("LAS_ALLOW_PARTIAL"):
            raise RuntimeError("liblas_hip.so does not export: %s" % ", ".join(missing))
        if "las_version" not in missing and l.las_version() != ABI_VERSION:       # argument structs are laid out by this file: a stale build
            raise RuntimeError("liblas_hip.so at %s was built from another include/las_hip.h (ABI %d, this package binds %d): rebuild it"
                               % (LIB_PATH, l.las_version(), ABI_VERSION))
        # development switches (A/B measurements, `make prof` builds only -- the shipping library does not export them): LAS_DEV_KK_BIG=0 -> 128 x 128 tiles only in las_gemm_kk;
        # LAS_DEV_ZGROUP=0 -> 3-D grid order for split-K / batched las_gemm; LAS_DEV_TN_TR=0 -> weight gradients through the
        # register-transposing kernel instead of the LDS-transposing one
        for env, sym in (("LAS_DEV_KK_BIG", "las_dev_gemm_kk_big"), ("LAS_DEV_ZGROUP", "las_dev_gemm_zgroup"),
                         ("LAS_DEV_TN_TR", "las_dev_gemm_tn_tr"), ("LAS_DEV_F32_VALU", "las_dev_gemm_f32_valu"),
                         ("LAS_DEV_F32_FAST_LD", "las_dev_gemm_f32_fast_ld"), ("LAS_DEV_LB_ROWS", "las_dev_lstm_cell_rows"),
                         ("LAS_DEV_LB_PAIR_ROWS", "las_dev_lstm_cell_pair_rows")):
            if os.environ.get(env) is not None and hasattr(l, sym):
                getattr(l, sym)(int(os.environ[env]))
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().las_last_error()
        raise RuntimeError("liblas_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("las HIP path needs tensors on a ROCm device (got %s); there is no CPU fallback"
                               % t.device)


def p(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------
# thin typed wrappers
# ------------------------------------------------------------------------------------------------
_ws_cache = {}


_ws_epoch = {}


def workspace(dev, nbytes, tag="default"):
    """Grow-only scratch buffer per (device, tag).  Every request bumps the tag's epoch: a caller that wants to find what its
    previous call left in the buffer (the Speller's backward reusing the forward's operand copies) compares epochs."""
    key = (_devkey(dev), tag)
    _ws_epoch[key] = _ws_epoch.get(key, 0) + 1
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("liblas_hip: scratch '%s' (%d bytes) would have to be allocated inside a HIP graph capture; run the step "
                               "once eagerly before capturing it (the buffer then exists and the graph only records launches)" % (tag, nbytes))
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def workspace_epoch(dev, tag):
    return _ws_epoch.get((_devkey(dev), tag), 0)


GEMM_WS_BYTES = 256 << 20        # include/las_hip.h LAS_GEMM_WS_CAP


# ---- auxiliary streams ------------------------------------------------------------------------------------------------------------
# The train step hands work between the launch stream and three auxiliary streams WHILE kernels of both run: side (weight gradients
# during the BPTT sweeps, x-projection chunks a running forward sweep waits for), chain (the backward hand-over's chunks), comm (the
# data-parallel exchange's early part).  That only works when every one of them has a hardware queue of its own: HIP multiplexes the
# streams of one priority onto GPU_MAX_HW_QUEUES (4) hardware queues in the order of their first use, and two streams on one queue
# serialise (tools/probe_streams.py, profiles/r4_probe_streams.txt: with nine other streams used first, a fresh torch.cuda.Stream()
# lands on the launch stream's queue one time in four -- round 3's silent loss of the hand-overs in the driver's pytest process).
# So the three are created TOGETHER, at the first use of the library on a device, at HIGH priority: HIP keeps a separate queue pool per
# priority, so no stream anybody else creates (torch's pool hands out normal-priority streams) can ever share a queue with them,
# whatever the process did before.  LAS_AUX_PRIORITY=0 puts them back into the normal pool (A/B measurements).
AUX_PRIORITY = int(os.environ.get("LAS_AUX_PRIORITY", "-1"))
_aux = {}
_side_used = False
_on_side = False
_chain_used = False


def aux_streams(dev=None):
    """{'side', 'chain', 'comm'} of a device, created (and given their hardware queues: a stream's first launch does that) on first use"""
    key = _devkey("cuda" if dev is None else dev)
    a = _aux.get(key)
    if a is None:
        d = torch.device(key)
        with torch.cuda.device(d):
            a = {n: torch.cuda.Stream(device=d, priority=AUX_PRIORITY) for n in ("side", "chain", "comm")}
            w = torch.zeros(4, dtype=torch.int32, device=d)
            for i, st in enumerate(a.values()):
                with torch.cuda.stream(st):
                    check(lib().las_set_word(p(w[i:]), 1, stream()), "las_set_word")
            torch.cuda.synchronize(d)
        _aux[key] = a
    return a


def side_stream():
    return aux_streams()["side"]


class on_side_stream:
    """Run the enclosed launches on the side stream, after everything already enqueued on the current stream.
    Tensors touched inside must be kept alive by the caller (record_stream)."""

    def __init__(self, after=None):
        self.after = after          # optional event: order the side work after it instead of after the whole stream

    def __enter__(self):
        global _side_used, _on_side
        self.ctx = torch.cuda.stream(side_stream())
        if self.after is not None:
            side_stream().wait_event(self.after)
        else:
            side_stream().wait_stream(torch.cuda.current_stream())
        self.ctx.__enter__()
        _side_used = True
        _on_side = True
        return self

    def __exit__(self, *exc):
        global _on_side
        _on_side = False
        return self.ctx.__exit__(*exc)


def chain_stream():
    """Second auxiliary stream: chunks of dependency-chain GEMMs that run WHILE the sweep that consumes them runs (the side
    stream cannot carry them: its weight-gradient GEMMs are held back / long)."""
    return aux_streams()["chain"]


class on_chain_stream:
    """Run the enclosed launches on the chain stream, after `after` (an event) or everything already enqueued on the current stream."""

    def __init__(self, after=None):
        self.after = after

    def __enter__(self):
        global _chain_used
        if self.after is not None:
            chain_stream().wait_event(self.after)
        else:
            chain_stream().wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(chain_stream())
        self.ctx.__enter__()
        _chain_used = True
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def join_chain_stream():
    global _chain_used
    if _chain_used:
        torch.cuda.current_stream().wait_stream(chain_stream())
        _chain_used = False


def comm_stream():
    """Stream the data-parallel exchange of the bucket's early part is issued from (orders it behind main / side / chain work
    without making any of those streams wait for the collective)."""
    return aux_streams()["comm"]


_deferred = []


def defer_side(fn):
    """Queue host work that only feeds the side stream (parameter gradients).  It is run by the NEXT backward node
    after that node has enqueued its own dependency-chain kernels (run_deferred), so the host time it costs does
    not sit between two chain kernels."""
    _deferred.append(fn)


def run_deferred():
    while _deferred:
        _deferred.pop(0)()


def join_side_stream():
    """Make the current stream wait for all side-stream work (call before the gradients are consumed)."""
    global _side_used
    run_deferred()
    join_chain_stream()
    if _side_used:
        torch.cuda.current_stream().wait_stream(side_stream())
        _side_used = False


def _tag(t):
    """scratch is per STREAM (split-K partials, column-sum partials): launches of two streams may overlap.  While a HIP graph is being
    captured the launch stream's scratch is used: the captured step replays on the launch stream, in order with its eager users (and a
    buffer allocated on torch's capture stream would come from the graph's private pool and outlive the graph in this cache)."""
    if torch.cuda.is_current_stream_capturing():
        cur = torch.cuda.current_stream()
        if any(cur == st for a in _aux.values() for st in a.values()):
            # a captured region that forks onto the side / chain streams would give concurrently replaying branches ONE set of
            # split-K / column-sum / weight-gradient partial buffers: refuse instead of racing (only single-stream regions are captured)
            raise RuntimeError("liblas_hip: library work on an auxiliary stream inside a HIP graph capture is not supported "
                               "(per-stream scratch cannot be told apart during capture); capture single-stream regions only")
        return t
    sid = torch.cuda.current_stream().stream_id
    return t if sid == 0 else "%s_s%d" % (t, sid)


def gemm(prec, A, B, C, transA=False, transB=False, M=None, N=None, K=None, lda=None, ldb=None, ldc=None,
         alpha=1.0, beta=0.0, bias=None, act=ACT_NONE, batch=1, strideA=0, strideB=0, strideC=0,
         mask_period=0, mask_skip=0, a_off=0, b_off=0, c_off=0):
    """C = act(alpha * op(A).op(B) + beta*C + bias).  A, B both fp32 or both bf16 (speed-mode activations), C fp32;
    *_off are element offsets."""
    require_gpu(A, B, C, bias)
    if A.dtype != B.dtype:
        raise RuntimeError("las_gemm: operands must share one element type (got %s, %s)" % (A.dtype, B.dtype))
    # split-K scratch sized by what THIS product's split wants (las_gemm_workspace_bytes; the buffer per (device, stream) only grows):
    # nine streams used to pin 256 MB each whatever they ran.  The byte count handed over is capped so a buffer grown by another
    # product cannot change this one's split degree (= its summation order).
    need = int(lib().las_gemm_workspace_bytes(prec, M, N, K, batch))
    ws = workspace(C.device, need, _tag("gemm")) if need else None
    es = A.element_size()
    rc = lib().las_gemm_dt(prec, int(transA), int(transB), M, N, K, alpha,
                           c_void_p(A.data_ptr() + es * a_off), lda, strideA,
                           c_void_p(B.data_ptr() + es * b_off), ldb, strideB, DT_BF16 if A.dtype == torch.bfloat16 else DT_F32, beta,
                           c_void_p(C.data_ptr() + 4 * c_off), ldc, strideC, p(bias), act, batch,
                           mask_period, mask_skip, p(ws), min(ws.numel(), GEMM_WS_BYTES) if need else 0, stream())
    check(rc, "las_gemm")


def gemm_kk(A, B, C, M, N, K, lda, ldb, ldc, bias=None, act=ACT_NONE, a_off=0, b_off=0, c_off=0, tanh_y=None, ldy=0):
    """C[M,N] = act(A[M,K] . B[N,K]^T + bias): A, B bf16 (contraction contiguous), C bf16 or fp32; *_off element offsets."""
    require_gpu(A, B, C, bias)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C.dtype in (torch.bfloat16, torch.float32)
    cdt = DT_BF16 if C.dtype == torch.bfloat16 else DT_F32
    if tanh_y is not None:           # fused Tanh gradient: C *= 1 - tanh_y^2 (bf16 [M, N], pitch ldy)
        assert tanh_y.dtype == torch.bfloat16
        check(lib().las_gemm_kk_tanhgrad(M, N, K, c_void_p(A.data_ptr() + 2 * a_off), lda, c_void_p(B.data_ptr() + 2 * b_off), ldb,
                                         c_void_p(C.data_ptr() + C.element_size() * c_off), cdt, ldc, p(bias), act, p(tanh_y), ldy,
                                         stream()), "las_gemm_kk_tanhgrad")
        return
    check(lib().las_gemm_kk(M, N, K, c_void_p(A.data_ptr() + 2 * a_off), lda, c_void_p(B.data_ptr() + 2 * b_off), ldb,
                            c_void_p(C.data_ptr() + C.element_size() * c_off), cdt, ldc, p(bias), act, stream()), "las_gemm_kk")


def wgrad_ih_hh(x, ldx, I, out, ld_out, out_bstride, gates, lddz, B, T, H, GH, d, dW, dW2=None, x2=None):
    """dW[0:I] += x^T . dZ_d, dW[I:I+H] += h_prev^T . dZ_d in one pass over dZ_d (las_wgrad_ih_hh); d = 2: both directions in one launch
    (dW forward, dW2 backward direction; x2: the backward direction's input copy).  out / gates: the tensors of both directions (bf16)."""
    require_gpu(x, out, gates, dW, dW2, x2)
    assert x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and gates.dtype == torch.bfloat16 and dW.dtype == torch.float32
    nb = int(lib().las_wgrad_ih_hh_workspace_bytes(I, H, GH, B, T, 2 if d == 2 else 1))
    ws = workspace(dW.device, nb, _tag("wgrad"))
    check(lib().las_wgrad_ih_hh(p(x), p(x2), ldx, I, p(out), ld_out, out_bstride, p(gates), lddz, B, T, H, GH, d, p(dW), p(dW2),
                                p(ws), ws.numel(), stream()), "las_wgrad_ih_hh")


def wgrad_ih_hh_window(x, ldx, I, out, ld_out, out_bstride, gates, lddz, B, T, H, GH, d, t0_fw, t0_bw, nframes, max_wgs, dW, dW2=None, x2=None):
    """las_wgrad_ih_hh over a window of `nframes` frames per utterance (forward direction from t0_fw, backward direction from t0_bw)"""
    require_gpu(x, out, gates, dW, dW2, x2)
    assert x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and gates.dtype == torch.bfloat16 and dW.dtype == torch.float32
    nb = int(lib().las_wgrad_ih_hh_workspace_bytes(I, H, GH, B, T, 2 if d == 2 else 1))
    ws = workspace(dW.device, nb, _tag("wgrad"))
    check(lib().las_wgrad_ih_hh_window(p(x), p(x2), ldx, I, p(out), ld_out, out_bstride, p(gates), lddz, B, T, H, GH, d, int(t0_fw), int(t0_bw),
                                       int(nframes), int(max_wgs), p(dW), p(dW2), p(ws), ws.numel(), stream()), "las_wgrad_ih_hh_window")


def wait_words_min(words, n, need, max_us=50000):
    """stream-ordered wait until every one of words[0:n] is >= need; a time-out marks the step invalid through the status word (code 2)"""
    check(lib().las_wait_words_min(p(words), int(n), int(need), int(max_us), p(status_word(words.device)), 2, stream()), "las_wait_words_min")


def rnn_seq_bwd_progress_words(cell, prec, B, H, flags=None):
    return int(lib().las_rnn_seq_bwd_progress_words(cell, prec, B, H, seq_flags if flags is None else flags))


def skinny_pack(W, K, N, ldw=None, row0=0):
    """bf16 MFMA-fragment copy of W[row0:row0+K, :N] (fp32, row-major) for skinny_gemm; made once per weight version"""
    require_gpu(W)
    ldw = W.stride(0) if ldw is None else ldw
    buf = torch.empty(int(lib().las_gemm_skinny_pack_bytes(K, N)), dtype=torch.uint8, device=W.device)
    check(lib().las_gemm_skinny_pack(c_void_p(W.data_ptr() + 4 * row0 * ldw), ldw, K, N, p(buf), stream()), "las_gemm_skinny_pack")
    return buf


def skinny_gemm(A, packed, C, M, K, N, lda, ldc, bias=None, accumulate=False, c_off=0):
    """C[M,N] (+)= bf16(A[M,K]) . packed + bias   (A, C fp32; 1 <= M <= 1024)"""
    require_gpu(A, packed, C, bias)
    check(lib().las_gemm_skinny(p(A), lda, M, K, p(packed), N, c_void_p(C.data_ptr() + 4 * c_off), ldc, p(bias), int(bool(accumulate)),
                                stream()), "las_gemm_skinny")


def _dt(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    assert t.dtype == torch.float32, t.dtype
    return DT_F32


def colsum(X, rows, cols, ldx, out, beta=0.0, x_off=0):
    """out[j] = beta*out[j] + sum_r X[r, j]; X fp32 or bf16, out fp32."""
    require_gpu(X, out)
    nb = lib().las_colsum_workspace_bytes(cols)
    ws = workspace(X.device, nb, _tag("colsum"))
    check(lib().las_colsum_dt(c_void_p(X.data_ptr() + X.element_size() * x_off), _dt(X), rows, cols, ldx, beta, p(out), p(ws),
                              ws.numel(), stream()), "las_colsum")


def tanh_bwd(Y, ldy, dY, lddy, dX, lddx, rows, cols):
    """dX = dY * (1 - Y*Y); every tensor fp32 or bf16."""
    require_gpu(Y, dY, dX)
    check(lib().las_tanh_bwd_dt(p(Y), _dt(Y), ldy, p(dY), _dt(dY), lddy, p(dX), _dt(dX), lddx, rows, cols, stream()), "las_tanh_bwd")


# ---- optional per-call device timing (bench.py roofline leg): HIP events on the launch stream -----
_PROF = None


def prof_begin():
    global _PROF
    _PROF = {}


def prof_end():
    """-> {name: [ms, ...]} ; call after torch.cuda.synchronize()."""
    global _PROF
    out = {k: [a.elapsed_time(b) for a, b in v] for k, v in (_PROF or {}).items()}
    _PROF = None
    return out


ROCTX = bool(os.environ.get("LAS_ROCTX"))      # roctx ranges (torch.cuda.nvtx maps to roctx on ROCm) around the phases of a step
                                               # and the K2 / K5-K7 launches: rocprofv3 --marker-trace shows them on the timeline


class roctx_range:
    """with roctx_range("listener fwd"): ...   -- a no-op unless LAS_ROCTX=1."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if ROCTX:
            torch.cuda.nvtx.range_push(self.name)
        if _PROF is not None:                      # prof_begin(): the phase's span on the launch stream, by HIP events
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROF is not None and hasattr(self, "e0"):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _PROF.setdefault("phase:" + self.name, []).append((self.e0, e1))
        if ROCTX:
            torch.cuda.nvtx.range_pop()
        return False


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if ROCTX:
            torch.cuda.nvtx.range_push(self.name)
        if _PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROF is not None:
            self.e1.record()
            _PROF.setdefault(self.name, []).append((self.e0, self.e1))
        if ROCTX:
            torch.cuda.nvtx.range_pop()
        return False


# ---- development switches of the sweeps / the Speller (explicit `flags` arguments of the C ABI).  The library itself
# reads no environment; this host layer maps the documented LAS_* variables to flags ONCE, at import, so the
# tools/ scripts keep working, and tests set `seq_flags` / `speller_flags` directly.
SEQ_AGENT_GRANULES, SEQ_NO_KSPLIT, SEQ_NO_HELPER_WAVES, SEQ_ROWS16, SEQ_NO_WARMERS, SEQ_F32_VALU, SEQ_PREPARED = 1, 2, 4, 8, 16, 32, 64
SPELLER_NO_PF_ROWS, SPELLER_NO_BF_ROWS, SPELLER_NO_FUSED_STEP, SPELLER_REUSE_PREP, SPELLER_NO_LOGITS, SPELLER_ROWS_SHARE4 = 1, 2, 4, 8, 16, 32
SPELLER_WIDE, SPELLER_NO_WIDE, SPELLER_SHARED_OPERANDS = 64, 128, 1 << 14      # csrc/speller_wide.h: force / forbid the wide per-step path (LAS_SPELLER_WIDE=1 / LAS_NO_WIDE=1)
SEQ_STATUS = {1: "forward sweep: a cluster partner did not publish h within the spin bound (or a chunk of the x-projection did not "
                 "complete while the sweep was waiting for it: kernels of different streams must be able to overlap -- under a "
                 "tool that serialises kernels, e.g. rocprofv3 --pmc, set LAS_ALLOW_SERIAL_STREAMS=1)",
              2: "BPTT sweep: a cluster partner did not publish its partial dh within the spin bound",
              3: "Speller loop kernel: a partner workgroup was not seen within the poll bound (the one-launch decode loop needs one "
                 "workgroup per compute unit co-resident; on a shared / partitioned device set LAS_NO_FUSED_STEP=1)"}


def seq_p(p):
    return (int(p) & 0xf) << 8


def speller_spin_log2(n):
    return (int(n) & 0x1f) << 8


def seq_spin_log2(n):
    return (int(n) & 0x1f) << 16


def _flags_from_env():
    e = os.environ.get
    f = (SEQ_AGENT_GRANULES if e("LAS_AGENT_GRANULES") == "1" else 0) | (SEQ_NO_KSPLIT if e("LAS_NO_KSPLIT") else 0) | \
        (SEQ_NO_HELPER_WAVES if e("LAS_NO_HELPER_WAVES") else 0) | (SEQ_ROWS16 if e("LAS_ROWS16") else 0) | \
        (SEQ_NO_WARMERS if e("LAS_NO_WARMERS") else 0) | (SEQ_F32_VALU if e("LAS_F32_VALU") else 0)
    if e("LAS_SEQ_P"):
        f |= seq_p(e("LAS_SEQ_P"))
    if e("LAS_SPIN_LOG2"):
        f |= seq_spin_log2(e("LAS_SPIN_LOG2"))
    g = (SPELLER_NO_PF_ROWS if e("LAS_NO_PF_ROWS") == "1" else 0) | (SPELLER_NO_BF_ROWS if e("LAS_NO_BF_ROWS") == "1" else 0) | \
        (SPELLER_NO_FUSED_STEP if e("LAS_NO_FUSED_STEP") == "1" else 0) | (SPELLER_WIDE if e("LAS_SPELLER_WIDE") == "1" else 0) | \
        (SPELLER_NO_WIDE if e("LAS_NO_WIDE") == "1" else 0)
    return f, g


seq_flags, speller_flags = _flags_from_env()

_status = {}


def _devkey(dev):
    """'cuda:<index>' for any spelling of a device ('cuda', torch.device('cuda'), a tensor's device): ONE status word /
    probe / overlap record per physical device"""
    d = torch.device(dev) if not isinstance(dev, torch.device) else dev
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def status_word(dev):
    """The sticky int32 device word the sweeps report exchange timeouts through (one per device)."""
    key = _devkey(dev)
    t = _status.get(key)
    if t is None:
        t = torch.zeros(2, dtype=torch.int32, device=dev)      # [0] sticky error code, [1] launch announcements (LAS_SEQ_ANNOUNCE)
        _status[key] = t
    return t


_announce = [0]


def next_announce():
    """The value the NEXT BPTT sweep will store into status_word[1] when it is resident (1..1023, cyclic)."""
    return _announce[0] % 1023 + 1


_overlap = {}
_overlap_failed = {}
# (the `make dbg` library synchronises the device after every launch: streams cannot overlap there by construction, so loading it
#  implies the serial schedule -- same kernel instances, every producer in front of its consumer)
ALLOW_SERIAL_STREAMS = bool(os.environ.get("LAS_ALLOW_SERIAL_STREAMS")) or os.path.basename(LIB_PATH).endswith("_dbg.so")
FORCE_SERIAL_STREAMS = os.environ.get("LAS_ALLOW_SERIAL_STREAMS") == "force"     # no probe: every hand-over with its producers first


def _probe_overlap(dev, other):
    """ms a bounded (4 ms) waiter on the current stream needs when the store it waits for is issued on `other` right behind it
    (best of two): ~0.03 when the two streams' kernels run side by side, the bound when they share a queue / are serialised"""
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    best = 1e9
    for _ in range(2):
        flag[0:1].zero_()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().las_wait_word(p(flag), 1, 4000, stream()), "las_wait_word")
        e1.record()
        with torch.cuda.stream(other):
            check(lib().las_set_word(p(flag), 1, stream()), "las_set_word")
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1))
    return best


def streams_overlap(dev):
    """True if kernels of the launch stream and of BOTH auxiliary streams (side: x-projection chunks, held weight gradients;
    chain: the backward hand-over's chunks) really run concurrently on this device: a bounded waiter on the current stream,
    then the store it waits for on the other stream.  Probed once per (device, launch stream); a True is cached.

    The auxiliary streams own their hardware queues by construction (aux_streams), so on a bare MI355X this cannot fail; it does
    under a tool that serialises kernels (rocprofv3 --pmc), or with another high-priority stream of the caller's on the same queue.
    The cross-stream hand-overs are what the timed step runs, so losing them is an ERROR, not a mode to fall into quietly: a failed
    probe is repeated once and then raises -- unless LAS_ALLOW_SERIAL_STREAMS=1 says that serialised streams are expected (counter
    passes), in which case the hand-overs are switched off and `last_variants` of the step says so."""
    a = aux_streams(dev)
    if FORCE_SERIAL_STREAMS:
        return False
    cur = torch.cuda.current_stream(dev)
    if any(cur == st for st in a.values()):
        # asked from inside an auxiliary stream's context (the Speller's side-stream part): the answer is the launch stream's
        known = [v for (d, _), v in _overlap.items() if d == _devkey(dev)]
        if known:
            return all(known)
        with torch.cuda.stream(torch.cuda.default_stream(dev)):
            return streams_overlap(dev)
    key = (_devkey(dev), cur.stream_id)
    ok = _overlap.get(key)
    if ok is None and key in _overlap_failed:
        raise RuntimeError(_overlap_failed[key])      # probed once per (device, launch stream): a failure is remembered, not re-measured
    if ok is None:
        worst = 0.0
        for attempt in range(2):
            worst = max(_probe_overlap(dev, a["side"]), _probe_overlap(dev, a["chain"]))
            if worst < 2.0:
                break
        ok = worst < 2.0
        if not ok and not ALLOW_SERIAL_STREAMS:
            _overlap_failed[key] = (
                "liblas_hip: kernels of the launch stream and of the auxiliary streams do not run concurrently on %s (a waiter needed "
                "%.1f ms for a store issued on another stream).  The train step's cross-stream hand-overs (x-projection chunks, "
                "backward hand-over, held weight gradients) need that.  Under a tool that serialises kernels (rocprofv3 --pmc, "
                "AMD_SERIALIZE_KERNEL, a debugger) set LAS_ALLOW_SERIAL_STREAMS=1 (the same kernels then run with every producer "
                "enqueued in front of its consumer); otherwise run the step on the default stream." % (key[0], worst))
            raise RuntimeError(_overlap_failed[key])
        _overlap[key] = ok                      # (a False is only ever cached when the environment asked for it)
    return ok


def hold_until_next_sweep(dev, max_us=1500):
    """Enqueue (on the current = side stream) a bounded wait for the next BPTT sweep's announcement."""
    check(lib().las_wait_announce(c_void_p(status_word(dev).data_ptr() + 4), next_announce(), max_us, stream()), "las_wait_announce")


def hold_until_last_sweep(dev, max_us=1500):
    """... for the announcement of the BPTT sweep that was launched LAST (for side work that the sweep's own node releases after its
    launch -- run_deferred -- but that could start on the device before the sweep does)."""
    if _announce[0]:
        check(lib().las_wait_announce(c_void_p(status_word(dev).data_ptr() + 4), (_announce[0] - 1) % 1023 + 1, max_us, stream()), "las_wait_announce")


def check_status(dev=None):
    """Synchronising check of the sweep status word(s): raises RuntimeError if any sweep reported a timeout.
    Call it wherever the host already waits for the device (loss read-out, end of a bench loop, checkpoint)."""
    for key, t in list(_status.items()):
        if dev is not None and _devkey(dev) != key:
            continue
        code = int(t[0].item())
        if code:
            t.zero_()
            _probe.pop(key, None)                  # (a probe of the same word may still be in flight: it reports nothing new)
            raise RuntimeError("liblas_hip recurrent sweep failed on %s (status %d): %s -- the cluster workgroups were not "
                               "all resident (shared / partitioned GPU?); results of that step are invalid"
                               % (key, code, SEQ_STATUS.get(code, "unknown")))


_probe = {}


def poll_status(dev, raise_on_error=True):
    """Non-blocking companion of check_status for callers that never wait for the device (LAS.train called in a loop
    without reading the loss): enqueue a copy of the status word to pinned host memory; when an EARLIER probe has
    completed with a non-zero code, raise (or, raise_on_error=False, return the code: LAS.train then re-runs the lost steps,
    LAS._recover).  A timeout is therefore reported at most a couple of steps late.  Returns 0 when nothing is known to be wrong."""
    key = _devkey(dev)
    if key not in _status:
        return 0
    pr = _probe.get(key)
    if pr is not None and pr[1].query():
        code = int(pr[0][0])
        if code:
            _probe.pop(key, None)
            if not raise_on_error:
                return code                      # (the word stays set: the device keeps skipping updates until the caller has recovered)
            _status[key].zero_()
            raise RuntimeError(status_message(key, code))
        pr = None
    if pr is None:
        pin = torch.zeros(2, dtype=torch.int32).pin_memory()
        ev = torch.cuda.Event()
        pin.copy_(_status[key], non_blocking=True)
        ev.record()
        _probe[key] = (pin, ev)
    return 0


def status_message(key, code):
    return "liblas_hip recurrent sweep failed on %s (status %d): %s" % (key, code, SEQ_STATUS.get(code, "unknown"))


def clear_status(dev):
    """Zero the device's status word (after the caller has dealt with what it reported) and forget any probe in flight."""
    key = _devkey(dev)
    if key in _status:
        _status[key].zero_()
    _probe.pop(key, None)


def rnn_seq_ws(cell, prec, H, B, dev):
    # per STREAM (_tag): the encoders of utterances of different lengths run side by side on several streams in beam search
    return workspace(dev, lib().las_rnn_seq_workspace_bytes(cell, prec, H, B), _tag("rnn_seq"))


def rnn_seq_io_dtype(cell, prec, H):
    """torch dtype of gates / out / cstate / dout for this sweep: bf16 when the speed-mode MFMA kernels serve it."""
    return torch.bfloat16 if lib().las_rnn_seq_io_dtype(cell, prec, H) == DT_BF16 else torch.float32


def _check_io(cell, prec, H, *tensors):
    want = rnn_seq_io_dtype(cell, prec, H)
    for t in tensors:
        if t is not None and t.dtype != want:
            raise RuntimeError("las_rnn_seq: tensors must be %s for (cell=%d, prec=%d, H=%d), got %s" % (want, cell, prec, H, t.dtype))


def gemm_kk_frames(A, B, C, nb, T, lo0, nlo, hi0, nhi, N, K, lda, ldb, ldc, bias=None, act=ACT_NONE, tanh_y=None, ldy=0):
    """las_gemm_kk on the frames [lo0, lo0+nlo) and [hi0, hi0+nhi) of every utterance of the [nb, T, *] tensors A (bf16) and C
    (tanh_y: as in gemm_kk)."""
    require_gpu(A, B, C, bias, tanh_y)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C.dtype in (torch.bfloat16, torch.float32)
    cdt = DT_BF16 if C.dtype == torch.bfloat16 else DT_F32
    check(lib().las_gemm_kk_frames(nb, T, lo0, nlo, hi0, nhi, N, K, p(A), lda, p(B), ldb, p(C), cdt, ldc, p(bias), act,
                                   p(tanh_y), ldy, stream()), "las_gemm_kk_frames")


def set_word(word, value):
    check(lib().las_set_word(p(word), int(value), stream()), "las_set_word")


def rnn_seq_fwd_chunks_ok(cell, prec, B, H, flags=None):
    return bool(lib().las_rnn_seq_fwd_chunks_ok(cell, prec, B, H, seq_flags if flags is None else flags))


def rnn_seq_fwd_rows_ok(cell, prec, B, H, flags=None):
    return bool(lib().las_rnn_seq_fwd_rows_ok(cell, prec, B, H, seq_flags if flags is None else flags))


def rnn_seq_prepare(jobs):
    """las_rnn_seq_prepare: jobs = [(cell, H, B, bwd, whh_fw, whh_bw, ldw, wf_off, wb_off, ws), ...] (fp32 kernels, element offsets of W_hh,
    one workspace tensor per job) -- packs + exchange-state clears of all of them in one launch on the current stream."""
    arr = (SeqPrepareDesc * len(jobs))()
    for d, (cell, H, B, bwd, wf, wb, ldw, wf_off, wb_off, ws) in zip(arr, jobs):
        require_gpu(wf, wb, ws)
        d.whh_fw, d.whh_bw, d.ldw = wf.data_ptr() + 4 * wf_off, wb.data_ptr() + 4 * wb_off, ldw
        d.cell, d.H, d.B, d.bwd, d.flags = cell, H, B, int(bool(bwd)), seq_flags
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    check(lib().las_rnn_seq_prepare(arr, len(jobs), stream()), "las_rnn_seq_prepare")


def rnn_seq_fwd(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate,
                forget_bias=1.0, wf_off=0, wb_off=0, flags=None, chunk_flag=None, chunk_steps=0, row_T=None, prepared_ws=None):
    """prepared_ws: a workspace rnn_seq_prepare has prepared for exactly this sweep (LAS_SEQ_PREPARED: no pack launch in front of it)."""
    require_gpu(gates, whh_fw, whh_bw, out, cstate)
    _check_io(cell, prec, H, gates, out, cstate)
    ws = rnn_seq_ws(cell, prec, H, B, gates.device) if prepared_ws is None else prepared_ws
    fl = (seq_flags if flags is None else flags) | (SEQ_PREPARED if prepared_ws is not None else 0)
    if row_T is not None:
        require_gpu(row_T)
        assert chunk_flag is None and row_T.dtype == torch.int32 and row_T.numel() == B
        with _timed("rnn_seq_fwd_rows[T=%d,H=%d]" % (T, H)):
            check(lib().las_rnn_seq_fwd_rows(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                             c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                             p(cstate), forget_bias, fl, p(status_word(gates.device)), p(row_T),
                                             p(ws), ws.numel(), stream()), "las_rnn_seq_fwd_rows")
        return
    if chunk_flag is not None:
        with _timed("rnn_seq_fwd[T=%d,H=%d]" % (T, H)):
            check(lib().las_rnn_seq_fwd_chunked(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                                c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                                p(cstate), forget_bias, fl, p(status_word(gates.device)), p(chunk_flag), chunk_steps,
                                                p(ws), ws.numel(), stream()), "las_rnn_seq_fwd_chunked")
        return
    with _timed("rnn_seq_fwd[T=%d,H=%d]" % (T, H)):
        check(lib().las_rnn_seq_fwd(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                    c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                    p(cstate), forget_bias, fl, p(status_word(gates.device)), p(ws), ws.numel(), stream()),
              "las_rnn_seq_fwd")


def rnn_seq_bwd_chunks_ok(cell, prec, B, H, flags=None):
    return bool(lib().las_rnn_seq_bwd_chunks_ok(cell, prec, B, H, seq_flags if flags is None else flags))


def rnn_seq_bwd(cell, prec, B, T, H, gates, whh_fw, whh_bw, ldw, out, ld_out, out_bstride, cstate,
                dout, ld_dout, dout_bstride, forget_bias=1.0, wf_off=0, wb_off=0, db_fw=None, db_bw=None, flags=None,
                chunk_flag=None, chunk_rows=0, n_rows=0, prepared_ws=None, progress=None, progress_steps=0):
    """db_fw / db_bw: optional [G*H] bias-gradient tensors, accumulated (+=) by the sweep itself.
    chunk_flag / chunk_rows / n_rows: dout is still being produced in chunks (las_rnn_seq_bwd_db_chunked)."""
    require_gpu(gates, whh_fw, whh_bw, out, cstate, dout)
    _check_io(cell, prec, H, gates, out, cstate, dout)
    ws = rnn_seq_ws(cell, prec, H, B, gates.device) if prepared_ws is None else prepared_ws
    fl = (seq_flags if flags is None else flags) | (SEQ_PREPARED if prepared_ws is not None else 0)
    fl |= next_announce() << 21                    # LAS_SEQ_ANNOUNCE: status_word[1] = this number once the sweep is resident
    _announce[0] += 1
    if chunk_flag is not None and progress is not None:
        with _timed("rnn_seq_bwd[T=%d,H=%d]" % (T, H)):
            check(lib().las_rnn_seq_bwd_db_progress(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                                    c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                                    p(cstate), p(dout), ld_dout, dout_bstride, forget_bias, p(db_fw), p(db_bw),
                                                    fl, p(status_word(gates.device)), p(chunk_flag), chunk_rows, n_rows,
                                                    p(progress), int(progress_steps), p(ws), ws.numel(), stream()), "las_rnn_seq_bwd_db_progress")
        return
    if chunk_flag is not None:
        with _timed("rnn_seq_bwd[T=%d,H=%d]" % (T, H)):
            check(lib().las_rnn_seq_bwd_db_chunked(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                                   c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                                   p(cstate), p(dout), ld_dout, dout_bstride, forget_bias, p(db_fw), p(db_bw),
                                                   fl, p(status_word(gates.device)), p(chunk_flag), chunk_rows, n_rows,
                                                   p(ws), ws.numel(), stream()), "las_rnn_seq_bwd_db_chunked")
        return
    with _timed("rnn_seq_bwd[T=%d,H=%d]" % (T, H)):
        check(lib().las_rnn_seq_bwd_db(cell, prec, B, T, H, p(gates), c_void_p(whh_fw.data_ptr() + 4 * wf_off),
                                       c_void_p(whh_bw.data_ptr() + 4 * wb_off), ldw, p(out), ld_out, out_bstride,
                                       p(cstate), p(dout), ld_dout, dout_bstride, forget_bias, p(db_fw), p(db_bw),
                                       fl, p(status_word(gates.device)), p(ws), ws.numel(), stream()), "las_rnn_seq_bwd_db")


SPELLER_FAMILIES = ((1, "loop"), (2, "pf_rows"), (4, "bf_rows"), (8, "f32_rows"), (16, "skinny_cell0"), (32, "loc"), (64, "skinny_upper_cells"), (128, "wide"))


def speller_last_variant():
    """{'fwd': [...], 'bwd': [...]}: the kernel families that served this thread's last las_speller_fwd / las_speller_bwd_part(1)
    (las_speller_last_variant, process-wide; bench.py and the tests name the family a number or a parity statement belongs to)."""
    l = lib()
    out = {}
    for name, which in (("fwd", 0), ("bwd", 1)):
        m = int(l.las_speller_last_variant(which))
        out[name] = [n for bit, n in SPELLER_FAMILIES if m & bit]
    return out
