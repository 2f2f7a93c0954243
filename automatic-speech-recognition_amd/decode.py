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
    if args.synthetic:
        from data import SyntheticBatches
        (audio, audiolen), (y, tokenlen) = next(SyntheticBatches(args.feat_dim, args.vocab_size, seed=args.seed + 2, batch_scale=0.1, max_frames=700))
        dev_feats = [audio[i, :audiolen[i]] for i in range(len(audio))]
        dev_tokens = list(y)
    elif os.path.exists(os.path.join(args.feat_dir, "{}-feats.pkl".format(args.split))):
        import joblib                                              # decode.py:80-85: the preprocess.py dumps
        dev_feats = joblib.load(args.feat_dir + "/{}-feats.pkl".format(args.split))
        dev_tokens = np.load(args.feat_dir + "/{}-{}s.npy".format(args.split, args.unit), allow_pickle=True)
        tokenlen = np.load(args.feat_dir + "/{}-{}len.npy".format(args.split, args.unit), allow_pickle=True)
    else:
        # same utterances from the packed split (create_tfrecord.py writes {split}-1.tfrecord)
        from tfrecord_data_loader import data_parser, tf_record_iterator
        path = os.path.join(args.tfrecord_dir or "data/tfrecord_{}_bpe_5k".format(args.feat_type), "{}-1.tfrecord".format(args.split))
        if not os.path.exists(path):
            raise Exception("Run preprocess.py first")
        recs = [data_parser(r) for r in tf_record_iterator(path)]
        dev_feats = [r[0][0] for r in recs]
        dev_tokens = [r[1][0] for r in recs]
        tokenlen = np.asarray([r[1][1] for r in recs])
    order = np.argsort(tokenlen)                                   # decode.py:122-124
    error, N, count = 0, 0, 0
    logging.info("Decoding...")
    limit = args.max_steps if args.max_steps >= 0 else (8 if args.synthetic else len(order))
    for i in order[:limit]:
        audio = np.asarray(dev_feats[i], np.float32)
        xs = (audio[None], np.asarray([audio.shape[0]], np.int32))
        beam_states = bs.decode(None, xs)
        hyp = convert_idx_to_string(beam_states[-1].token_ids[1:], id_to_token, args.unit)
        ref = convert_idx_to_string(dev_tokens[i], id_to_token, args.unit)
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
