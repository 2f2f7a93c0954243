"""preprocess.py -- feature extraction entry point (counterpart of the reference's preprocess.py:50-108): MFCC / log-mel
filterbank + CMVN + delta / delta-delta cube [T, feat_dim, 3], tokenisation, dumps that create_tfrecord / decode.py read.

PARITY STATUS: unpinned.  The reference delegates the arithmetic to `speechpy` (speechpy.feature.mfcc / mfe,
speechpy.processing.cmvn, speechpy.feature.extract_derivative_feature; unpinned in requirements.txt) and reads audio with
`soundfile`; neither is installed in the build image and neither can be fetched.  The functions below RESTATE speechpy
2.4's published algorithm as called at reference preprocess.py:71-86 (frame stacking without zero padding and with a
rectangular window, 512-point power spectrum / fft_length, 40 triangular mel filters from 300 Hz -- speechpy's
`low_freq or 300` turns the default 0 into 300 -- log, orthonormal DCT-II, c0 replaced by log frame energy; CMVN with
variance normalisation and eps 2^-30; derivatives over a +-2 window taken ALONG THE FEATURE AXIS exactly as speechpy's
`derivative_extraction` pads and slides along axis 1).  Offline CPU work, not on the hot path."""
import math
import os
import string
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def frequency_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_to_frequency(mel):
    return 700.0 * (np.exp(mel / 1127.0) - 1.0)


def _triangle(x, left, middle, right):
    out = np.zeros(x.shape)
    first_half = np.logical_and(left < x, x <= middle)
    out[first_half] = (x[first_half] - left) / (middle - left)
    second_half = np.logical_and(middle <= x, x < right)
    out[second_half] = (right - x[second_half]) / (right - middle)
    return out


def filterbanks(num_filter, coefficients, sampling_freq, low_freq=None, high_freq=None):
    high_freq = high_freq or sampling_freq / 2
    low_freq = low_freq or 300                                   # speechpy: a low frequency of 0 becomes 300 Hz
    mels = np.linspace(frequency_to_mel(low_freq), frequency_to_mel(high_freq), num_filter + 2)
    hertz = mel_to_frequency(mels)
    freq_index = (np.floor((coefficients + 1) * hertz / sampling_freq)).astype(int)
    fb = np.zeros([num_filter, coefficients])
    for i in range(num_filter):
        left, middle, right = int(freq_index[i]), int(freq_index[i + 1]), int(freq_index[i + 2])
        z = np.linspace(left, right, num=right - left + 1)
        fb[i, left:right + 1] = _triangle(z, left=left, middle=middle, right=right)
    return fb


def stack_frames(sig, sampling_frequency, frame_length, frame_stride):
    """speechpy.processing.stack_frames(..., filter=ones, zero_padding=False)"""
    n = sig.shape[0]
    fl = int(np.round(sampling_frequency * frame_length))
    fs = float(np.round(sampling_frequency * frame_stride))
    numframes = int(math.floor((n - fl) / fs))
    if numframes < 1:
        return np.zeros((0, fl))
    idx = np.tile(np.arange(0, fl), (numframes, 1)) + np.tile(np.arange(0, numframes * fs, fs), (fl, 1)).T
    return sig[np.array(idx, dtype=np.int32)]


def _zero_handling(x):
    return np.where(x == 0, np.finfo(float).eps, x)


def mfe(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_filters=40, fft_length=512, low_frequency=0,
        high_frequency=None):
    signal = signal.astype(float)
    frames = stack_frames(signal, sampling_frequency, frame_length, frame_stride)
    high_frequency = high_frequency or sampling_frequency / 2
    power_spectrum = 1.0 / fft_length * np.square(np.absolute(np.fft.rfft(frames, fft_length)))
    coefficients = power_spectrum.shape[1]
    frame_energies = _zero_handling(np.sum(power_spectrum, 1))
    fb = filterbanks(num_filters, coefficients, sampling_frequency, low_frequency, high_frequency)
    features = _zero_handling(np.dot(power_spectrum, fb.T))
    return features, frame_energies


def mfcc(signal, sampling_frequency, frame_length=0.020, frame_stride=0.01, num_cepstral=13, num_filters=40, fft_length=512,
         low_frequency=0, high_frequency=None, dc_elimination=True):
    from scipy.fftpack import dct
    feature, energy = mfe(signal, sampling_frequency, frame_length, frame_stride, num_filters, fft_length, low_frequency, high_frequency)
    if len(feature) == 0:
        return np.empty((0, num_cepstral))
    feature = np.log(feature)
    feature = dct(feature, type=2, axis=-1, norm='ortho')[:, :num_cepstral]
    if dc_elimination:
        feature[:, 0] = np.log(energy)
    return feature


def cmvn(vec, variance_normalization=False):
    eps = 2 ** -30
    mean_subtracted = vec - np.mean(vec, axis=0)
    if variance_normalization:
        return mean_subtracted / (np.std(mean_subtracted, axis=0) + eps)
    return mean_subtracted


def derivative_extraction(feat, DeltaWindows):
    rows, cols = feat.shape
    DIF = np.zeros(feat.shape, dtype=feat.dtype)
    Scale = 0
    FEAT = np.pad(feat, ((0, 0), (DeltaWindows, DeltaWindows)), 'edge')       # padded (and differenced) along the feature axis
    for i in range(DeltaWindows):
        offset = DeltaWindows
        Range = i + 1
        dif = Range * FEAT[:, offset + Range:offset + Range + cols] - FEAT[:, offset - Range:offset - Range + cols]
        Scale += 2 * np.power(Range, 2)
        DIF += dif
    return DIF / Scale


def extract_derivative_feature(feature):
    first = derivative_extraction(feature, DeltaWindows=2)
    second = derivative_extraction(first, DeltaWindows=2)
    return np.concatenate((feature[:, :, None], first[:, :, None], second[:, :, None]), axis=2)


def read_audio(path):
    """(samples float, sampling rate).  .wav through scipy; .npy as raw 16 kHz samples; .flac needs `soundfile` (the
    reference's reader, preprocess.py:68), which this image does not have."""
    if path.endswith(".npy"):
        return np.load(path).astype(float), 16000
    if path.endswith(".wav"):
        from scipy.io import wavfile
        fs, a = wavfile.read(path)
        if a.dtype.kind == "i":
            a = a.astype(float) / np.iinfo(a.dtype).max
        return a.astype(float), fs
    try:
        import soundfile as sf
    except ImportError:
        raise RuntimeError("reading %s needs the `soundfile` package (not available offline); convert to .wav" % path)
    return sf.read(path)


def process_audios(audio_path, args):
    """reference preprocess.py:50-91.  Returns (feats: list of float32 [L, feat_dim, 3] (or [L, feat_dim] without --cmvn), featlen)."""
    feats, featlen = [], []
    for p in audio_path:
        audio, fs = read_audio(p)
        if args.feat_type == 'mfcc':
            feat = mfcc(audio, fs, frame_length=args.frame_length / 1000, frame_stride=args.frame_step / 1000, num_cepstral=args.feat_dim)
        elif args.feat_type == 'fbank':
            feat, _ = mfe(audio, fs, frame_length=args.frame_length / 1000, frame_stride=args.frame_step / 1000, num_filters=args.feat_dim)
        else:
            raise ValueError(args.feat_type)
        if args.cmvn:
            feat = cmvn(feat, True)
            feat = extract_derivative_feature(feat)
        feats.append(feat.astype(np.float32))
        featlen.append(len(feats[-1]))
    return feats, featlen


def process_texts(texts, tokenizer):
    """reference preprocess.py:93-108: strip punctuation, encode with EOS."""
    tokens, tokenlen = [], []
    for sentence in texts:
        sentence = sentence.translate(str.maketrans('', '', string.punctuation))
        tokens.append(tokenizer.encode(sentence, with_eos=True))
        tokenlen.append(len(tokens[-1]))
    return tokens, np.array(tokenlen).astype(np.int32)


def data_preparation(libri_path):
    """reference preprocess.py:26-48: walk speaker/chapter folders, pair every transcript line with its audio file."""
    from glob import glob
    texts, audio_path = [], []
    for path in sorted(glob(libri_path + "/*/*")):
        tp = glob(path + "/*txt")
        if not tp:
            continue
        for line in open(tp[0]).readlines():
            line_ = line.split(" ")
            base = path + "/" + line_[0]
            audio_path.append(base + (".wav" if os.path.exists(base + ".wav") else ".flac"))
            texts.append(line[len(line_[0]) + 1:-1].replace("'", ""))
    return texts, audio_path


def main():
    import joblib
    from las.arguments import parse_args
    from utils.tokenizer import CharEncoder, SubwordEncoder
    args = parse_args()
    tokenizer = CharEncoder() if args.unit.lower() == "char" else SubwordEncoder(args.subword_dir)
    os.makedirs(args.feat_dir, exist_ok=True)
    for split, path in (("train-100", args.train_100hr_corpus_dir), ("dev", args.dev_data_dir), ("test", args.test_data_dir)):
        if not os.path.isdir(path):
            continue
        texts, audio_path = data_preparation(path)
        feats, featlen = process_audios(audio_path, args)
        tokens, tokenlen = process_texts(texts, tokenizer)
        joblib.dump(feats, os.path.join(args.feat_dir, "%s-feats.pkl" % split))
        np.save(os.path.join(args.feat_dir, "%s-featlen.npy" % split), np.asarray(featlen))
        np.save(os.path.join(args.feat_dir, "%s-%ss.npy" % (split, args.unit)), np.asarray(tokens, dtype=object), allow_pickle=True)
        np.save(os.path.join(args.feat_dir, "%s-%slen.npy" % (split, args.unit)), tokenlen)
        print("%s: %d utterances -> %s" % (split, len(feats), args.feat_dir))


if __name__ == "__main__":
    main()
