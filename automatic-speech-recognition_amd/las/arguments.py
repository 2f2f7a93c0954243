"""las.arguments -- the single flag namespace shared by train.py / test.py / decode.py.

Flag names, types and defaults are the drop-in contract (reference las/arguments.py:12-232; pinned by
golden G2).  They are declared as one table instead of fifty add_argument blocks; the MI355X build adds
a few flags of its own at the end (cell type, arithmetic mode, data-parallel bucket options)."""
import argparse


def str2bool(v):
    """'yes/true/t/y/1' -> True, 'no/false/f/n/0' -> False (case-insensitive), else ArgumentTypeError
    (reference las/arguments.py:4-10)."""
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


# (flags, type, default, help)
_FLAGS = [
    # features
    (("--dataset",), str, "LibriSpeech", "Dataset: LibriSpeech or TEDLIUM."),
    (("--unit",), str, "subword", "Encoding unit for texts processing."),
    (("--sample_rate",), int, 16000, "Sample rate."),
    (("--feat_dim",), int, 39, "The feature dimension."),
    (("--frame_length",), int, 25, "Frame length in ms."),
    (("--frame_step",), int, 10, "Frame step in ms."),
    (("--feat_type",), str, "mfcc", "mfcc"),
    (("--cmvn",), str2bool, True, "Apply cmvn or not."),
    (("--augmentation",), str2bool, False, "Apply data augmentation or not."),
    (("--split",), str, "dev", "Split used for evaluation."),
    # training
    (("--verbose", "-vb"), int, 0, "Verbosity."),
    (("--batch_size", "-bs"), int, 32, "The training batch size."),
    (("--lr",), float, 1e-3, "The training learning rate."),
    (("--grad_clip",), float, 5, "Apply gradient clipping."),
    (("--dropout_rate",), float, 0.5, "The probability of drop out."),
    (("--epoch",), int, 10, "The number of training epochs."),
    (("--restore_epoch",), int, -1, "The epoch you want to restore."),
    (("--label_smoothing",), str2bool, True, "Apply label smoothing."),
    (("--apply_bn",), str2bool, False, "Apply batch normalization."),
    (("--add_vn",), str2bool, False, "Apply variational noise to weights."),
    (("--ctc",), str2bool, False, "Apply ctc."),
    (("--ctc_weight",), float, 0.2, "Weighting of ctc."),
    # Listener
    (("--enc_type",), str, "cnn", "Listener type: cnn or pblstm."),
    (("--enc_units",), int, 64, "The hidden dimension of the BLSTMs in Listener."),
    (("--num_enc_channels",), int, 32, "The number of channels in CNN layers of Listener."),
    (("--num_enc_layers",), int, 2, "The number of layers of BLSTMs in Listener."),
    # Attention
    (("--attention_size",), int, 128, "Attention size."),
    (("--loc_kernel_size",), int, 201, "Kernel size in location-aware attention."),
    (("--loc_num_channels",), int, 10, "Number of channels in location-aware attention"),
    (("--mode",), str, "add", "Additive attention or loction-aware attention."),
    # Speller
    (("--dec_units",), int, 128, "The hidden dimension of the LSTM in Speller."),
    (("--num_dec_layers",), int, 2, "The number of layers of LSTM in Speller."),
    (("--embedding_size",), int, 128, "The dimension of the embedding matrix is: [vocab_size, embedding_size]."),
    (("--scheduled_sampling",), str2bool, True, "Apply schduled sampling."),
    (("--warmup_step",), int, 100000, "Steps of pure teacher forcing before scheduled sampling starts."),
    (("--max_step",), int, 500000, "Max step in scheduled sampling."),
    (("--min_rate",), float, 0.4, "Minimum teacher-forcing rate in scheduled sampling."),
    # beam search
    (("--convert_rate",), float, 0.166, "Convert the length of audio to estimate the required decoding steps."),
    (("--beam_size",), int, 10, "Size for beam search."),
    (("--apply_lm",), str2bool, False, "Apply language model."),
    (("--lm_weight",), float, 0.5, "Weighting of recoring with language model."),
    # directories
    (("--train_100hr_corpus_dir",), str, "data/LibriSpeech/LibriSpeech_train/train-clean-100", ""),
    (("--train_360hr_corpus_dir",), str, "data/LibriSpeech/LibriSpeech_train/train-clean-360", ""),
    (("--train_500hr_corpus_dir",), str, "data/LibriSpeech/LibriSpeech_train/train-other-500", ""),
    (("--dev_data_dir",), str, "data/LibriSpeech-100/LibriSpeech_dev/dev-clean", ""),
    (("--test_data_dir",), str, "data/LibriSpeech-100/LibriSpeech_test/test-clean", ""),
    (("--feat_dir",), str, "data/LibriSpeech/features", "Path to save features."),
    (("--subword_dir",), str, "subword/", "Path to vocab files of BPE subword unit."),
    (("--log_dir",), str, "log/", "Save log file.."),
    (("--save_dir",), str, "model/las/", "Save trained model."),
    (("--summary_dir",), str, "summary/", "Save summary."),
]

# additions of the MI355X build (not in the reference; defaults keep reference behaviour)
_EXTRA = [
    (("--cell",), str, "rnn", "Recurrent cell: 'rnn' = BasicRNNCell as the reference builds, 'lstm' = BasicLSTMCell."),
    (("--dtype",), str, "f32", "Contraction arithmetic: f32 (parity) or bf16 (MFMA speed mode)."),
    (("--seed",), int, 0, "Initialiser seed."),
    (("--synthetic",), str2bool, False, "Train on synthetic [B,T,feat_dim,3] batches (no TFRecords needed)."),
    (("--max_steps",), int, -1, "Stop after this many steps (-1: run all epochs)."),
    (("--tfrecord_dir",), str, "", "Directory holding train-*.tfrecord / dev-1.tfrecord (default data/tfrecord_<feat>_bpe_5k)."),
    (("--stack",), int, 1, "Bucket batches per train step on one GPU (the reference's 48 / 96 rows times this; 1 = the reference). k batches in one "
                           "step are the update of k data-parallel ranks; the latency-bound recurrent sweeps make k = 2 / 4 cost 1.4 / 2.0 steps."),
    (("--decode_batch",), int, 64, "Utterances per device-resident beam-search batch in decode.py (rows of a search step = this x beam_size; "
                                  "64 x 16 rows decode 1.8x the utterances/s of 16 x 16)."),
    (("--lm_dir",), str, "lang/output/", "Output directory of train_lm.py (result.json, vocab.json, models) for --apply_lm."),
]


def build_parser(extra=True):
    parser = argparse.ArgumentParser(
        description="Listen, Attend and Spell (LAS) end-to-end speech recognition on MI355X")
    for flags, typ, default, helptext in _FLAGS + (_EXTRA if extra else []):
        parser.add_argument(*flags, type=typ, default=default, help=helptext)
    return parser


def parse_args(argv=None):
    """reference las/arguments.py:12 (argv=None -> sys.argv[1:])."""
    return build_parser().parse_args(argv)


def reference_flag_names():
    return [f[0][0].lstrip("-") for f in _FLAGS]
