"""las.checkpoint -- save / resume.

The reference keeps `tf.train.Saver(var_list=get_save_vars(), max_to_keep=30)` checkpoints named
`las_E{epoch}` (train.py:82-90,128-132), restores latest-or-init, and lets test.py / decode.py pick an
epoch (test.py:83-86, las/beam_search.py:272-281).  Same file naming and policy here; the payload is the
variable store's state_dict (parameters + Adam slots + global_step) written with torch.save."""
import glob
import os
import re

import torch

from las import variables as V

MAX_TO_KEEP = 30


def _epoch_of(path):
    m = re.search(r"las_E(\d+)$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(save_dir):
    """tf.train.latest_checkpoint analogue: highest las_E{e} in save_dir, or None."""
    cands = [p for p in glob.glob(os.path.join(save_dir, "las_E*")) if _epoch_of(p) >= 0]
    return max(cands, key=_epoch_of) if cands else None


def save(save_dir, epoch, store=None):
    store = store or V.default_store()
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "las_E%d" % epoch)
    torch.save(store.state_dict(), path)
    cands = sorted((p for p in glob.glob(os.path.join(save_dir, "las_E*")) if _epoch_of(p) >= 0), key=_epoch_of)
    for old in cands[:-MAX_TO_KEEP]:
        os.remove(old)
    return path


def restore(save_dir, restore_epoch=-1, store=None):
    """Load `las_E{restore_epoch}` (or the latest when -1).  Returns the path, or None if nothing exists."""
    store = store or V.default_store()
    path = latest_checkpoint(save_dir) if restore_epoch == -1 else os.path.join(save_dir, "las_E%d" % restore_epoch)
    if path is None or not os.path.exists(path):
        return None
    # weights_only: the payload is tensors, ints and dicts of them; a checkpoint path must never be able to run pickled code
    store.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return path
