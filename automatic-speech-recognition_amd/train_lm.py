"""train_lm.py -- char RNNLM training entry point (counterpart of the reference's train_lm.py: same flags, same artefacts
under --output_dir: vocab.json, result.json {params, vocab_file, encoding, latest_model, best_model, best_valid_ppl,
test_ppl}, lang/save_model/model-<step>, lang/best_model/model-<step>; same log lines).

The graph of the reference (three CharRNN instances sharing variables: training, validation, evaluation with batch 1 /
unroll 1, train_lm.py:239-249) becomes three CharRNN objects over ONE VariableStore; `saver.save/restore` becomes a
torch-saved state_dict of that store (parameters + Adam slots), which decode.py's restore_lm reads for shallow fusion."""
import argparse
import codecs
import json
import logging
import os
import shutil
import string
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    p = argparse.ArgumentParser()
    # (flag, type, default) exactly as reference train_lm.py:21-118
    for flag, typ, default, hlp in [
            ('--data_file', str, 'data/tiny_shakespeare.txt', 'data file'),
            ('--encoding', str, 'utf-8', 'the encoding of the data file.'),
            ('--output_dir', str, 'lang/output', 'directory to store final and intermediate results and models.'),
            ('--n_save', int, 1, 'how many times to save the model during each epoch.'),
            ('--max_to_keep', int, 5, 'how many recent models to keep.'),
            ('--hidden_size', int, 128, 'size of RNN hidden state vector'),
            ('--embedding_size', int, 0, 'size of character embeddings'),
            ('--num_layers', int, 2, 'number of layers in the RNN'),
            ('--num_unrollings', int, 10, 'number of unrolling steps.'),
            ('--model', str, 'lstm', 'which model to use (rnn, lstm or gru).'),
            ('--num_epochs', int, 50, 'number of epochs'),
            ('--batch_size', int, 20, 'minibatch size'),
            ('--train_frac', float, 0.9, 'fraction of data used for training.'),
            ('--valid_frac', float, 0.05, 'fraction of data used for validation.'),
            ('--dropout', float, 0.0, 'dropout rate, default to 0 (no dropout).'),
            ('--input_dropout', float, 0.0, 'dropout rate on input layer, default to 0 (no dropout), and no dropout if using one-hot representation.'),
            ('--max_grad_norm', float, 5., 'clip global grad norm'),
            ('--learning_rate', float, 2e-3, 'initial learning rate'),
            ('--decay_rate', float, 0.95, 'decay rate'),
            ('--progress_freq', int, 100, 'frequency for progress report in training and evalution.'),
            ('--verbose', int, 0, 'whether to show progress report in training and evalution.'),
            ('--init_model', str, '', 'initial model'),
            ('--best_model', str, '', 'current best model'),
            ('--best_valid_ppl', float, np.inf, 'current valid perplexity'),
            ('--init_dir', str, '', 'continue from the outputs in the given directory'),
            ('--dtype', str, 'f32', 'contraction arithmetic: f32 or bf16 (MI355X build)'),
            ('--seed', int, 0, 'initialiser seed (MI355X build)')]:
        p.add_argument(flag, type=typ, default=default, help=hlp)
    p.add_argument('--log_to_file', dest='log_to_file', action='store_true')
    p.add_argument('--debug', dest='debug', action='store_true', help='show debug information')
    p.add_argument('--test', dest='test', action='store_true', help='use the first 1000 character to as data to test the implementation')
    p.set_defaults(log_to_file=False, debug=False, test=False)
    return p


def text_cleaning(text, save_path="data/libri_cleaned.txt"):
    """reference train_lm.py:359-376: drop empty lines, join with spaces, ?! -> ., strip other punctuation and digits, upper."""
    text = "\n".join(item for item in text.split('\n') if item)
    text = text.replace("\n", " ")
    text = text.replace("  ", " ")
    trans = str.maketrans("?!", "..", '"#$%&\'()*+,-/:;<=>@[\\]^_`{|}~' + "1234567890")
    text = text.translate(trans)
    text = text.upper()
    if save_path:
        try:
            os.makedirs(os.path.dirname(save_path), exist_ok=True)
            with open(save_path, "w+") as f:
                f.write(text)
        except OSError:
            pass
    return text


def create_vocab():
    """reference train_lm.py:378-386 (28 symbols: '.', ' ', A-Z)."""
    unique_chars = [".", " "] + list(string.ascii_uppercase[:26])
    vocab_index_dict = {c: i for i, c in enumerate(unique_chars)}
    index_vocab_dict = {i: c for i, c in enumerate(unique_chars)}
    return vocab_index_dict, index_vocab_dict, len(unique_chars)


def load_vocab(vocab_file, encoding):
    with codecs.open(vocab_file, 'r', encoding=encoding) as f:
        vocab_index_dict = json.load(f)
    index_vocab_dict = {index: char for char, index in vocab_index_dict.items()}
    return vocab_index_dict, index_vocab_dict, len(vocab_index_dict)


def save_vocab(vocab_index_dict, vocab_file, encoding):
    with codecs.open(vocab_file, 'w', encoding=encoding) as f:
        json.dump(vocab_index_dict, f, indent=2, sort_keys=True)


class Corpus:
    """The cleaned text cut into its train / validation / test parts (fractions --train_frac / --valid_frac, the rest is the test
    part) with one batch iterator per part."""

    def __init__(self, text, train_frac, valid_frac, vocab):
        from lang.char_rnn_model import BatchGenerator
        self.vocab = vocab
        n = len(text)
        cut = [0, int(train_frac * n), int(train_frac * n) + int(valid_frac * n), n]
        self.parts = {name: text[cut[k]:cut[k + 1]] for k, name in enumerate(("train", "valid", "test"))}
        self._make = lambda part, B, U: BatchGenerator(self.parts[part], B, U, vocab[2], vocab[0], vocab[1])

    def size(self, part):
        return len(self.parts[part])

    def batches(self, part, batch_size, unroll):
        return self._make(part, batch_size, unroll)


class RunDir:
    """What a run leaves under --output_dir (the artefacts decode.py and a later --init_dir run read): vocab.json, result.json
    and the rotating `lang/save_model/model-<step>` / best `lang/best_model/model-<step>` checkpoints."""

    def __init__(self, root, resume, keep):
        self.root, self.keep = root, keep
        self.latest_prefix = os.path.join(root, 'lang/save_model/model')
        self.best_prefix = os.path.join(root, 'lang/best_model/model')
        self.record = {}
        if resume:
            with open(self.path('result.json')) as f:
                self.record = json.load(f)
        else:
            if os.path.exists(root):
                shutil.rmtree(root)
            for prefix in (self.latest_prefix, self.best_prefix):
                os.makedirs(os.path.dirname(prefix))

    def path(self, name):
        return os.path.join(self.root, name)

    def checkpoint(self, store, step, best=False):
        import glob
        import torch
        prefix = self.best_prefix if best else self.latest_prefix
        target = "%s-%d" % (prefix, step)
        torch.save(store.state_dict(), target)
        if not best and self.keep:
            by_step = sorted(glob.glob(prefix + "-*"), key=lambda q: int(q.rsplit("-", 1)[1]))
            for stale in by_step[:-self.keep]:
                os.remove(stale)
        return target

    def write(self, **fields):
        self.record.update(fields)
        with open(self.path('result.json'), 'w') as f:
            json.dump(self.record, f, indent=2, sort_keys=True)


HYPER = ('batch_size', 'num_unrollings', 'hidden_size', 'max_grad_norm', 'embedding_size', 'num_layers', 'learning_rate', 'model',
         'dropout', 'input_dropout')


def main(argv=None):
    import torch
    from lang.char_rnn_model import CharRNN, batches2string
    from las import layers, variables
    args = build_parser().parse_args(argv)
    resume = bool(args.init_dir)
    run = RunDir(args.init_dir or args.output_dir, resume, args.max_to_keep)
    log_file = run.path('experiment_log.txt') if args.log_to_file else 'stdout'
    logging.basicConfig(format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO, datefmt='%I:%M:%S',
                        **({'stream': sys.stdout} if log_file == 'stdout' else {'filename': log_file}))
    print('=' * 60 + '\nAll final and intermediate outputs will be stored in %s/\nAll information will be logged to %s\n'
          % (run.root, log_file) + '=' * 60 + '\n')
    # ---- hyper-parameters, best-so-far and vocabulary: from the command line, or from the run that is being continued
    if resume:
        hyper = dict(run.record['params'])
        init_model, best_model, best_ppl = run.record['latest_model'], run.record['best_model'], run.record['best_valid_ppl']
        encoding = run.record.get('encoding', 'utf-8')
        vocab = load_vocab(run.path('vocab.json'), encoding)
    else:
        hyper = {k: getattr(args, k) for k in HYPER}
        init_model, best_model, best_ppl, encoding = args.init_model, args.best_model, args.best_valid_ppl, args.encoding
        vocab = create_vocab()
        save_vocab(vocab[0], run.path('vocab.json'), encoding)
        logging.info('Vocabulary is saved in %s', run.path('vocab.json'))
    hyper['vocab_size'] = vocab[2]
    logging.info('Parameters are:\n%s\n', json.dumps(hyper, sort_keys=True, indent=4))
    logging.info('Reading data from: %s', args.data_file)
    with codecs.open(args.data_file, 'r', encoding=encoding) as f:
        text = text_cleaning(f.read(), save_path=run.path("libri_cleaned.txt"))
    if args.test:
        text = text[:1000]
    logging.info('Number of characters: %s', len(text))
    corpus = Corpus(text, args.train_frac, args.valid_frac, vocab)
    logging.info('Vocab size: %d', vocab[2])
    B, U = hyper['batch_size'], hyper['num_unrollings']
    feeds = {"train": corpus.batches("train", B, U), "valid": corpus.batches("valid", B, U), "test": corpus.batches("test", 1, 1)}
    if args.debug:
        logging.info(batches2string(feeds["train"].next(), vocab[1]))
    # ---- three views of ONE parameter store (reference train_lm.py:239-249: training / validation / batch-1 evaluation graphs)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    layers.set_precision(args.dtype)
    store = variables.VariableStore(device=dev, seed=args.seed)
    models = {"train": CharRNN(is_training=True, use_batch=True, store=store, **hyper),
              "valid": CharRNN(is_training=False, use_batch=True, store=store, **hyper),
              "test": CharRNN(is_training=False, use_batch=False, store=store, **hyper)}
    trainer = models["train"]
    trainer.params()
    logging.info('Model size (number of parameters): %s\n', store.num_params())
    if init_model:
        sd = torch.load(init_model, map_location="cpu", weights_only=True)     # (tensors, ints and dicts of them: a checkpoint path must never run pickled code)
        store.load_state_dict(sd)
        trainer.global_step = int(sd.get("global_step", 0))

    def sweep(part, train=False, **kw):
        return models[part].run_epoch(None, corpus.size(part), feeds[part], is_training=train, verbose=args.verbose,
                                      freq=args.progress_freq, **kw)[0]

    run.write(params=hyper, vocab_file=run.path('vocab.json'), encoding=encoding)
    latest = ''
    # every epoch is cut into --n_save slices; after each slice: checkpoint, validate, keep the best
    for done in range(args.num_epochs * args.n_save):
        logging.info('=' * 19 + ' Epoch %d: %d/%d' + '=' * 19 + '\n', done // args.n_save + 1, done % args.n_save + 1, args.n_save)
        logging.info('Training on training set')
        sweep("train", train=True, divide_by_n=args.n_save)
        store.global_step = trainer.global_step
        latest = run.checkpoint(store, trainer.global_step)
        logging.info('Latest model saved in %s\n', latest)
        logging.info('Evaluate on validation set')
        valid_ppl = sweep("valid")
        if not best_model or valid_ppl < best_ppl:
            best_model, best_ppl = run.checkpoint(store, trainer.global_step, best=True), valid_ppl
        logging.info('Best model is saved in %s', best_model)
        logging.info('Best validation ppl is %f\n', best_ppl)
        run.write(latest_model=latest, best_model=best_model, best_valid_ppl=float(best_ppl))
    logging.info('Latest model is saved in %s', latest)
    logging.info('Evaluate the best model on test set')
    if best_model:
        store.load_state_dict(torch.load(best_model, map_location="cpu", weights_only=True))
    run.write(test_ppl=float(sweep("test")))


if __name__ == '__main__':
    main()
