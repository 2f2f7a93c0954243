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


def load_lm(init_dir, store=None):
    """reference decode.py:27-39: build the inference CharRNN from `result.json` (hyper-parameters written by
    train_lm.py:270-272) and `vocab.json` of the LM output directory."""
    import json
    from lang.char_rnn_model import CharRNN
    with open(os.path.join(init_dir, 'result.json'), 'r') as f:
        result = json.load(f)
    params = dict(result['params'])
    with open(os.path.join(init_dir, 'vocab.json'), 'r') as f:
        vocab_index_dict = json.load(f)
    params["vocab_size"] = len(vocab_index_dict)
    params["num_unrollings"] = 1
    params["batch_size"] = 1
    logging.info('Creating rnnlm graph')
    lm = CharRNN(is_training=False, use_batch=True, store=store, **params)
    return lm, result


def restore_lm(lm, save_path):
    """reference decode.py:41-53: restore the LM's variables (this build: a torch-saved {name: array} dictionary written
    by train_lm.py; the names are the reference's TF variable names under the `lm` scope)."""
    import torch
    sd = torch.load(save_path, map_location="cpu", weights_only=True)
    lm.params()                                     # create the variables, then overwrite them
    lm.store.load({k: v for k, v in sd["params"].items() if k.startswith(lm.scope + "/")})
    logging.info("Rnnlm restored: {}".format(save_path))


def main():
    import torch
    args = parse_args()
    logging.basicConfig(stream=sys.stdout, format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO, datefmt='%I:%M:%S')
    tokenizer = CharEncoder() if args.unit.lower() == "char" else SubwordEncoder(args.subword_dir)
    args.vocab_size = tokenizer.get_vocab_size()
    id_to_token, token_to_id = tokenizer.id_to_token, tokenizer.token_to_id
    layers.set_cell(args.cell)
    layers.set_precision(args.dtype)
    # replicas only: under torch.distributed.run every rank decodes its round-robin share of the
    # length-sorted utterance list and the (errors, words) counts are all-reduced at the end
    from las import parallel
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = parallel.init_from_env(dev)
    rank, world = (dp.rank, dp.world) if dp is not None else (0, 1)
    variables.reset_default_store(device=dev, seed=args.seed)
    las = LAS(args, Listener, Speller, token_to_id)
    lm = None
    if args.apply_lm:                                              # decode.py:68-75
        logging.info("Apply RNNLM...")
        st = variables.default_store()
        if args.synthetic and not os.path.exists(os.path.join(args.lm_dir, "result.json")):
            from lang.char_rnn_model import CharRNN              # smoke / bench runs: the shipped LM shape, random weights
            lm = CharRNN(False, 1, 1, 28, 512, embedding_size=0, num_layers=2, store=st)
            lm.params()
        else:
            lm, result = load_lm(args.lm_dir, st)
            lm.params()
    las.build_variables()
    bs = BeamSearch(args, las, token_to_id, lm)
    ckpt = bs.restore_las(None, args.save_dir, args.restore_epoch)
    logging.info("LAS restored: {}".format(ckpt))
    if lm is not None and os.path.exists(os.path.join(args.lm_dir, "result.json")):
        restore_lm(lm, result['best_model'])
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
    todo = parallel.shard(list(order[:limit]), rank, world)
    nb = max(1, int(args.decode_batch))
    chunks = [todo[c0:c0 + nb] for c0 in range(0, len(todo), nb)]  # decode.py:131-149, `decode_batch` utterances per call

    def batches():
        for chunk in chunks:
            xs_list = []
            for i in chunk:
                audio = np.asarray(dev_feats[i], np.float32)
                xs_list.append((audio[None], np.asarray([audio.shape[0]], np.int32)))
            yield xs_list

    # the model's objects live as long as the process: out of the cyclic collector's way (a full collection walks them all, 30-60 ms --
    # a batch and a half -- every few batches otherwise)
    import gc
    gc.collect()
    gc.freeze()
    # decode_batches: the encoders of the next chunk run under the search of this one
    for chunk, results in zip(chunks, bs.decode_batches(None, batches())):
        for i, beam_states in zip(chunk, results):
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
    error, N = parallel.reduce_error_counts(dp, error, N, dev)
    if rank == 0:
        logging.info("Dev WER: {}".format(error / N))


if __name__ == "__main__":
    main()
