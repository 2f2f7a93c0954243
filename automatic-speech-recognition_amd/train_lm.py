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


def _save(store, prefix, step, keep=None):
    import glob
    import torch
    path = "%s-%d" % (prefix, step)
    torch.save(store.state_dict(), path)
    if keep:
        old = sorted(glob.glob(prefix + "-*"), key=lambda q: int(q.rsplit("-", 1)[1]))
        for q in old[:-keep]:
            os.remove(q)
    return path


def main(argv=None):
    import torch
    from lang.char_rnn_model import BatchGenerator, CharRNN, batches2string
    from las import layers, variables
    args = build_parser().parse_args(argv)
    args.save_model = os.path.join(args.output_dir, 'lang/save_model/model')
    args.save_best_model = os.path.join(args.output_dir, 'lang/best_model/model')
    args.vocab_file = ''
    if args.init_dir:
        args.output_dir = args.init_dir
    else:
        if os.path.exists(args.output_dir):
            shutil.rmtree(args.output_dir)
        for paths in [args.save_model, args.save_best_model]:
            os.makedirs(os.path.dirname(paths))
    args.log_file = os.path.join(args.output_dir, 'experiment_log.txt') if args.log_to_file else 'stdout'
    kw = dict(format='%(asctime)s %(levelname)s:%(message)s', level=logging.INFO, datefmt='%I:%M:%S')
    logging.basicConfig(**(dict(stream=sys.stdout) if args.log_file == 'stdout' else dict(filename=args.log_file)), **kw)
    print('=' * 60)
    print('All final and intermediate outputs will be stored in %s/' % args.output_dir)
    print('All information will be logged to %s' % args.log_file)
    print('=' * 60 + '\n')
    if args.init_dir:
        with open(os.path.join(args.init_dir, 'result.json'), 'r') as f:
            result = json.load(f)
        params = result['params']
        args.init_model = result['latest_model']
        best_model = result['best_model']
        best_valid_ppl = result['best_valid_ppl']
        args.encoding = result.get('encoding', 'utf-8')
        args.vocab_file = os.path.join(args.init_dir, 'vocab.json')
    else:
        params = {'batch_size': args.batch_size, 'num_unrollings': args.num_unrollings, 'hidden_size': args.hidden_size,
                  'max_grad_norm': args.max_grad_norm, 'embedding_size': args.embedding_size, 'num_layers': args.num_layers,
                  'learning_rate': args.learning_rate, 'model': args.model, 'dropout': args.dropout,
                  'input_dropout': args.input_dropout}
        best_model = ''
        best_valid_ppl = args.best_valid_ppl
    logging.info('Parameters are:\n%s\n', json.dumps(params, sort_keys=True, indent=4))
    logging.info('Reading data from: %s', args.data_file)
    with codecs.open(args.data_file, 'r', encoding=args.encoding) as f:
        text_origin = f.read()
    text = text_cleaning(text_origin, save_path=os.path.join(args.output_dir, "libri_cleaned.txt"))
    if args.test:
        text = text[:1000]
    logging.info('Number of characters: %s', len(text))
    logging.info('Creating train, valid, test split')
    train_size = int(args.train_frac * len(text))
    valid_size = int(args.valid_frac * len(text))
    test_size = len(text) - train_size - valid_size
    train_text = text[:train_size]
    valid_text = text[train_size:train_size + valid_size]
    test_text = text[train_size + valid_size:]
    if args.vocab_file:
        vocab_index_dict, index_vocab_dict, vocab_size = load_vocab(args.vocab_file, args.encoding)
    else:
        logging.info('Creating vocabulary')
        vocab_index_dict, index_vocab_dict, vocab_size = create_vocab()
        vocab_file = os.path.join(args.output_dir, 'vocab.json')
        save_vocab(vocab_index_dict, vocab_file, args.encoding)
        logging.info('Vocabulary is saved in %s', vocab_file)
        args.vocab_file = vocab_file
    params['vocab_size'] = vocab_size
    logging.info('Vocab size: %d', vocab_size)
    batch_size, num_unrollings = params['batch_size'], params['num_unrollings']
    train_batches = BatchGenerator(train_text, batch_size, num_unrollings, vocab_size, vocab_index_dict, index_vocab_dict)
    valid_batches = BatchGenerator(valid_text, batch_size, num_unrollings, vocab_size, vocab_index_dict, index_vocab_dict)
    test_batches = BatchGenerator(test_text, 1, 1, vocab_size, vocab_index_dict, index_vocab_dict)
    if args.debug:
        logging.info(batches2string(train_batches.next(), index_vocab_dict))
    logging.info('Creating graph')
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    layers.set_precision(args.dtype)
    store = variables.VariableStore(device=dev, seed=args.seed)
    train_model = CharRNN(is_training=True, use_batch=True, store=store, **params)
    valid_model = CharRNN(is_training=False, use_batch=True, store=store, **params)
    test_model = CharRNN(is_training=False, use_batch=False, store=store, **params)
    train_model.params()
    logging.info('Model size (number of parameters): %s\n', store.num_params())
    logging.info('Start training\n')
    result = {'params': params, 'vocab_file': args.vocab_file, 'encoding': args.encoding}
    saved_path = ''
    try:
        if args.init_model:
            sd = torch.load(args.init_model, map_location="cpu", weights_only=False)
            store.load_state_dict(sd)
            train_model.global_step = int(sd.get("global_step", 0))
        for i in range(args.num_epochs):
            for j in range(args.n_save):
                logging.info('=' * 19 + ' Epoch %d: %d/%d' + '=' * 19 + '\n', i + 1, j + 1, args.n_save)
                logging.info('Training on training set')
                ppl, _, global_step = train_model.run_epoch(None, train_size, train_batches, is_training=True, verbose=args.verbose,
                                                            freq=args.progress_freq, divide_by_n=args.n_save)
                store.global_step = train_model.global_step
                saved_path = _save(store, args.save_model, train_model.global_step, keep=args.max_to_keep)
                logging.info('Latest model saved in %s\n', saved_path)
                logging.info('Evaluate on validation set')
                valid_ppl, _, _ = valid_model.run_epoch(None, valid_size, valid_batches, is_training=False, verbose=args.verbose,
                                                        freq=args.progress_freq)
                if (not best_model) or (valid_ppl < best_valid_ppl):
                    best_model = _save(store, args.save_best_model, train_model.global_step)
                    best_valid_ppl = valid_ppl
                logging.info('Best model is saved in %s', best_model)
                logging.info('Best validation ppl is %f\n', best_valid_ppl)
                result['latest_model'] = saved_path
                result['best_model'] = best_model
                result['best_valid_ppl'] = float(best_valid_ppl)
                with open(os.path.join(args.output_dir, 'result.json'), 'w') as f:
                    json.dump(result, f, indent=2, sort_keys=True)
        logging.info('Latest model is saved in %s', saved_path)
        logging.info('Best model is saved in %s', best_model)
        logging.info('Best validation ppl is %f\n', best_valid_ppl)
        logging.info('Evaluate the best model on test set')
        if best_model:
            store.load_state_dict(torch.load(best_model, map_location="cpu", weights_only=False))
        test_ppl, _, _ = test_model.run_epoch(None, test_size, test_batches, is_training=False, verbose=args.verbose,
                                              freq=args.progress_freq)
        result['test_ppl'] = float(test_ppl)
    finally:
        with open(os.path.join(args.output_dir, 'result.json'), 'w') as f:
            json.dump(result, f, indent=2, sort_keys=True)


if __name__ == '__main__':
    main()
