"""train.py -- training entry point (counterpart of the reference's train.py, same flags and log lines).

    python train.py --enc_type pblstm --feat_dim 13 --unit char --dropout_rate 0 --synthetic True ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # data parallel

Replaces the `sess.run` step loop of train.py:114-133: every iteration is one eager `las.train(xs, ys)` on
this rank's batch; with WORLD_SIZE>1 the gradient bucket is all-reduced over RCCL (las/parallel.py).

The loop never waits for the device: batches arrive through `las.input_pipeline.DeviceFeeder` (background reader thread --
the C++ TFRecord reader of liblas_hip.so or the synthetic source -- pinned staging, host->device copies on a copy stream,
`--prefetch` batches ahead), and the per-step log line is printed from an asynchronous copy of the loss one or two steps later
(`LaggedLog`).  Data parallel runs are LOCK STEP: every rank draws the same bucket at every step and keeps its rows of that
bucket's global batch (tfrecord_data_loader: rank / world), so no rank waits for another rank's longer utterances."""
import json
import logging
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from las import checkpoint, layers, variables                      # noqa: E402
from las.arguments import parse_args                               # noqa: E402
from las.las import LAS, Listener, Speller                         # noqa: E402
from las.utils import convert_idx_to_string                        # noqa: E402
from utils.tokenizer import CharEncoder, SubwordEncoder            # noqa: E402


def main():
    args = parse_args()
    logging.basicConfig(stream=sys.stdout, format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO,
                        datefmt='%I:%M:%S')
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        print('=' * 60 + '\n')
        logging.info('Parameters are:\n%s\n', json.dumps(vars(args), sort_keys=False, indent=4))
        print('=' * 60 + '\n')
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(args.seed * 1000003 + rank)                 # dropout masks: per-rank streams
    dp = None
    if world > 1:
        import torch.distributed as dist
        from las.parallel import DataParallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dp = DataParallel()
        # the ranks share the host with each other and with their reader threads: no machine-wide intra-op pools per rank
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))

    # tokenizer (the reference always builds SubwordEncoder, train.py:60 -- SURVEY Q15; --unit is honoured here)
    tokenizer = CharEncoder() if args.unit.lower() == "char" else SubwordEncoder(args.subword_dir)
    args.vocab_size = tokenizer.get_vocab_size()
    id_to_token = tokenizer.id_to_token

    layers.set_cell(args.cell)
    layers.set_precision(args.dtype)
    variables.reset_default_store(device=dev, seed=args.seed)
    las = LAS(args, Listener, Speller, id_to_token)
    las.dp = dp
    las.build_variables()
    st = variables.default_store()
    os.makedirs(args.save_dir, exist_ok=True)
    ckpt = checkpoint.restore(args.save_dir, -1)                  # restore latest or init (train.py:84-90)
    if dp is not None:
        dp.broadcast_(st.flat)

    num_train_batches = 2619                                      # train.py:108
    from las.input_pipeline import LaggedLog, feeder_for
    if args.synthetic:
        from data import SyntheticBatches
        source = SyntheticBatches(args.feat_dim, args.vocab_size, seed=args.seed, rank=rank, batch_scale=max(args.stack, 1))
    else:
        # train.py:45-55: data/tfrecord_{feat_type}_bpe_5k/train-*.tfrecord.  Every rank walks the same record stream (same
        # seed) and keeps its rows of every global batch: one bucket shape per step on all ranks
        import glob
        from tfrecord_data_loader import data_parser, tfrecord_iterator
        pattern = os.path.join(args.tfrecord_dir or "data/tfrecord_{}_bpe_5k".format(args.feat_type), "train-*.tfrecord")
        files = sorted(glob.glob(pattern))
        if not files:
            raise Exception("Run preprocess.py, create_tfrecord.py first")
        source, _, _ = tfrecord_iterator(files, data_parser, args.feat_dim, seed=args.seed, rank=rank, world=world,
                                         native=os.environ.get("LAS_PY_READER") != "1", batch_scale=max(args.stack, 1))
    # --stack k: every bucket emits k times the reference's rows, i.e. k of its batches as ONE step (LAS.train_stacked's arithmetic:
    # the update of k data-parallel ranks) -- the sweeps are latency-bound on a fifth of the compute units, k = 4 is 1.96x the rate
    batches = feeder_for(source, dev, args.feat_dim, is_training=True, depth=max(2, int(os.environ.get("LAS_PREFETCH", "3"))),
                         batch_scale=max(args.stack, 1))

    if rank == 0:
        logging.info("Total weights: {}".format(st.num_params()))
    training_steps = num_train_batches * args.epoch if args.max_steps < 0 else args.max_steps
    if rank == 0:
        logging.info("Total num train batches: {}".format(num_train_batches))
        logging.info("Training...")
    loss_ = []

    def report(info, value):                                       # runs when the step's loss has reached the host (lagged)
        gs, tfrate = info
        loss_.append(value)
        if rank == 0:
            logging.info("Step: {}, Loss: {:.3f}, tf rate: {:.3f}".format(gs, value, tfrate))

    log = LaggedLog(report)
    import time
    t_loop, n_utt = time.perf_counter(), 0
    for step in range(training_steps):
        xs, ys = next(batches)
        batch_loss, _, gs, logits, alphas, _, tfrate = las.train(xs, ys)
        n_utt += int(xs[0].shape[0])
        log.push((gs, tfrate), batch_loss)
        if rank == 0 and args.verbose > 0:                         # (the text summaries synchronise: debugging aid, as in the reference)
            logging.info("HYP: {}".format(convert_idx_to_string(torch.argmax(logits[0], -1).cpu().numpy(), id_to_token, args.unit)))
            logging.info("REF: {}\n".format(convert_idx_to_string(ys[0][0].cpu().numpy(), id_to_token, args.unit)))
        if gs and gs % num_train_batches == 0:
            log.drain(True)
            las.check_status()
            e_ = gs // num_train_batches
            if rank == 0:
                logging.info('=' * 19 + ' Epoch %d, Step %d, Ave loss %f' + '=' * 19 + '\n', e_, gs, np.mean(loss_))
                checkpoint.save(args.save_dir, e_)
            loss_ = []
    log.drain(True)
    torch.cuda.synchronize()
    las.check_status()                                             # (LAS.train polls the status word without waiting in between)
    if rank == 0:
        dt = time.perf_counter() - t_loop
        logging.info("train loop: {} steps, {} utterances in {:.2f} s = {:.1f} utterances/s on this rank (input pipeline and "
                     "host->device copies included)".format(training_steps, n_utt, dt, n_utt / max(dt, 1e-9)))
    batches.close()
    if rank == 0 and args.max_steps >= 0:
        checkpoint.save(args.save_dir, max(1, st.global_step // num_train_batches))


if __name__ == "__main__":
    main()
