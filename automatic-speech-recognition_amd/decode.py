"""decode.py -- beam-search evaluation entry point (counterpart of the reference's decode.py: utterances
sorted by token length, one BeamSearch.decode per utterance, per-utterance and corpus WER, same log lines)."""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from las import checkpoint, layers, variables                      # noqa: E402
from las.arguments import parse_args                               # noqa: E402
from las.beam_search import BeamSearch                             # noqa: E402
from las.las import LAS, Listener, Speller                         # noqa: E402
from las.utils import convert_idx_to_string, edit_distance        # noqa: E402
from utils.tokenizer import CharEncoder, SubwordEncoder            # noqa: E402


def main():
    import torch
    args = parse_args()
    logging.basicConfig(stream=sys.stdout, format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO, datefmt='%I:%M:%S')
    if args.apply_lm:
        raise SystemExit("RNNLM shallow fusion needs trained LM weights (reference README marks it NOT READY; SURVEY 8(f) F4)")
    tokenizer = CharEncoder() if args.unit.lower() == "char" else SubwordEncoder(args.subword_dir)
    args.vocab_size = tokenizer.get_vocab_size()
    id_to_token, token_to_id = tokenizer.id_to_token, tokenizer.token_to_id
    layers.set_cell(args.cell)
    layers.set_precision(args.dtype)
    variables.reset_default_store(device=torch.device("cuda", 0), seed=args.seed)
    las = LAS(args, Listener, Speller, token_to_id)
    las.build_variables()
    bs = BeamSearch(args, las, token_to_id, None)
    ckpt = bs.restore_las(None, args.save_dir, args.restore_epoch)
    logging.info("LAS restored: {}".format(ckpt))
    if not args.synthetic:
        raise SystemExit("feature files {split}-feats.pkl are produced by the reference's preprocess.py (SURVEY F4); run with --synthetic True")
    from data import SyntheticBatches
    (audio, audiolen), (y, tokenlen) = next(SyntheticBatches(args.feat_dim, args.vocab_size, seed=args.seed + 2, batch_scale=0.1, max_frames=700))
    order = np.argsort(tokenlen)                                   # decode.py:122-124
    error, N, count = 0, 0, 0
    logging.info("Decoding...")
    for i in order[: (8 if args.max_steps < 0 else args.max_steps)]:
        xs = (audio[i:i + 1, :audiolen[i]], audiolen[i:i + 1])
        beam_states = bs.decode(None, xs)
        hyp = convert_idx_to_string(beam_states[-1].token_ids[1:], id_to_token, args.unit)
        ref = convert_idx_to_string(y[i], id_to_token, args.unit)
        dist, n = edit_distance(ref.split(" "), hyp.split(" "))
        error += dist
        N += n
        logging.info("Utt {}/{}, WER: {}".format(count, len(order), dist / n))
        count += 1
        if args.verbose > 0:
            logging.info("REF | {}".format(ref))
            logging.info("HYP | {}\n".format(hyp))
    logging.info("Dev WER: {}".format(error / N))


if __name__ == "__main__":
    main()
