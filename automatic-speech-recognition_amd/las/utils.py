"""las.utils -- host-side helpers with the reference's names (reference las/utils.py).

`label_smoothing` is folded into the K8 cross-entropy kernel on the hot path; the tensor form is kept
for API parity.  `edit_distance` / `wer` / `convert_idx_to_string` are pinned against the reference by
tests/golden/reference_host_golden.json (G3, G4)."""
import numpy as np
import torch


def label_smoothing(inputs, epsilon=0.01):
    """(1-eps)*onehot + eps/K over the last axis  (reference las/utils.py:5-12; eps=0.01)."""
    return (1.0 - epsilon) * inputs + epsilon / inputs.shape[-1]


def _ids_to_text(ids, id_to_token, unit):
    text = "".join(id_to_token[int(i)] for i in ids).split("<EOS>")[0].strip()
    if unit == "char":
        text = text.replace("<SPACE>", " ")
    elif unit == "subword":
        text = text.replace("</w>", " ")
    return " ".join(text.split())


def convert_idx_to_token_tensor(inputs, id_to_token, unit="char"):
    """reference las/utils.py:15-33 wrapped the conversion in tf.py_func; eager code just returns the str."""
    if torch.is_tensor(inputs):
        inputs = inputs.detach().cpu().numpy()
    return _ids_to_text(inputs, id_to_token, unit)


def convert_idx_to_string(inputs, id_to_token, unit="char"):
    """ids -> text: join tokens, cut at the first <EOS>, map <SPACE> (char) or </w> (subword) to ' ',
    normalise whitespace (reference las/utils.py:35-46)."""
    return _ids_to_text(inputs, id_to_token, unit)


def edit_distance(s1, s2):
    """Levenshtein distance between two token sequences; returns (distance, len(s1)) exactly as
    reference las/utils.py:54-67 (float64 DP table, denominator = reference length).  The DP runs one
    numpy row at a time instead of the reference's O(n*m) Python loop."""
    n, m = len(s1), len(s2)
    prev = np.arange(n + 1, dtype=np.float64)              # column j = 0
    for j in range(1, m + 1):
        cur = np.empty(n + 1, dtype=np.float64)
        cur[0] = j
        neq = np.asarray([x != s2[j - 1] for x in s1], dtype=np.float64) if n else np.zeros(0)
        # substitution / match and deletion candidates are vectorised; insertion needs the running minimum
        cand = np.minimum(prev[:-1] + neq, prev[1:] + 1.0)
        run = cur[0]
        for i in range(1, n + 1):
            run = min(cand[i - 1], run + 1.0)
            cur[i] = run
        prev = cur
    return float(prev[-1]), n


def wer(s1, s2):
    """reference las/utils.py:48-52."""
    e, length = edit_distance(s1, s2)
    return e / length


def get_save_vars():
    """Everything a checkpoint needs (reference las/utils.py:69-82 filtered TF collections: trainables,
    Adam slots, global_step); here: the default variable store's state_dict."""
    from las import variables as V
    return V.default_store().state_dict()
