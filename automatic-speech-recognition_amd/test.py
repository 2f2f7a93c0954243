"""test.py -- greedy evaluation entry point (counterpart of the reference's test.py: same flags, same
artefacts log_dir/test_pred.txt + test_gt.txt, same corpus-WER arithmetic test.py:127-136)."""
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from las import checkpoint, layers, variables                      # noqa: E402
from las.arguments import parse_args                               # noqa: E402
from las.las import LAS, Listener, Speller                         # noqa: E402
from las.utils import convert_idx_to_string, edit_distance        # noqa: E402
from utils.tokenizer import CharEncoder, SubwordEncoder            # noqa: E402


def corpus_counts(texts_gt, texts_pred):
    error, N = 0, 0
    for ref, hyp in zip(texts_gt, texts_pred):
        e, n = edit_distance(ref.split(" "), hyp.split(" "))
        error += e
        N += n
    return error, N


def corpus_wer(texts_gt, texts_pred):
    error, N = corpus_counts(texts_gt, texts_pred)
    return error / N


def main():
    import torch
    args = parse_args()
    logging.basicConfig(stream=sys.stdout, format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO, datefmt='%I:%M:%S')
    tokenizer = CharEncoder() if args.unit.lower() == "char" else SubwordEncoder(args.subword_dir)
    args.vocab_size = tokenizer.get_vocab_size()
    id_to_token = tokenizer.id_to_token
    layers.set_cell(args.cell)
    layers.set_precision(args.dtype)
    # replicas only: ranks take the evaluation batches round-robin; (errors, words) are all-reduced
    from las import parallel
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = parallel.init_from_env(dev)
    rank, world = (dp.rank, dp.world) if dp is not None else (0, 1)
    variables.reset_default_store(device=dev, seed=args.seed)
    las = LAS(args, Listener, Speller, id_to_token)
    las.build_variables()
    ckpt = checkpoint.restore(args.save_dir, args.restore_epoch)
    logging.info("restored: {}".format(ckpt))
    if args.synthetic:
        from data import SyntheticBatches
        src = SyntheticBatches(args.feat_dim, args.vocab_size, seed=args.seed + 1, batch_scale=0.25, max_frames=700)
        batches = (next(src) for _ in range(4 if args.max_steps < 0 else args.max_steps))
    else:
        # test.py:47-61: one pass over data/tfrecord_{feat_type}_bpe_5k/dev-1.tfrecord
        from tfrecord_data_loader import data_parser, get_num_records, tfrecord_iterator
        eval_filenames = os.path.join(args.tfrecord_dir or "data/tfrecord_{}_bpe_5k".format(args.feat_type), "dev-1.tfrecord")
        if not os.path.exists(eval_filenames):
            raise Exception("Run preprocess.py, create_tfrecord.py first")
        batches, _, _ = tfrecord_iterator(eval_filenames, data_parser, args.feat_dim, is_training=False)
        logging.info("Total num eval records: {}".format(get_num_records([eval_filenames])))
        if args.max_steps >= 0:
            import itertools
            batches = itertools.islice(batches, args.max_steps)
    output_id, gt_id = [], []
    import itertools
    for xs, ys in itertools.islice(batches, rank, None, world):
        _, y_hat = las.inference(xs)
        output_id += y_hat.cpu().numpy().tolist()
        gt_id += ys[0].tolist()
    texts_pred = [convert_idx_to_string(o, id_to_token, args.unit) for o in output_id]
    texts_gt = [convert_idx_to_string(g, id_to_token, args.unit) for g in gt_id]
    os.makedirs(args.log_dir, exist_ok=True)
    sfx = "" if world == 1 else ".rank%d" % rank
    with open(os.path.join(args.log_dir, "test_pred.txt" + sfx), 'w') as fout:
        fout.write("\n".join(texts_pred))
    with open(os.path.join(args.log_dir, "test_gt.txt" + sfx), 'w') as fout:
        fout.write("\n".join(texts_gt))
    error, N = corpus_counts(texts_gt, texts_pred)
    total = parallel.reduce_error_counts(dp, len(texts_gt), 0, dev)[0]
    error, N = parallel.reduce_error_counts(dp, error, N, dev)
    if rank == 0:
        logging.info("total utterances: {}, WER: {}".format(int(total), error / N))


if __name__ == "__main__":
    main()
