"""preprocess.py (F4): the MFCC / CMVN / delta cube restated from speechpy's published algorithm (parity unpinned: speechpy is
not installable here).  Structural checks on a synthetic signal: shapes of the [T, feat_dim, 3] cube the data pipeline
expects (reference preprocess.py:72-86, tfrecord_data_loader.py:44), CMVN statistics, the filterbank and the orthonormal DCT."""
import numpy as np

import helpers  # noqa: F401  (sys.path)


def test_mfcc_cube_shapes_and_cmvn():
    import preprocess as P
    fs = 16000
    t = np.arange(int(1.3 * fs)) / fs
    rng = np.random.RandomState(0)
    sig = 0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1800 * t) + 0.05 * rng.randn(len(t))
    f = P.mfcc(sig, fs, frame_length=0.025, frame_stride=0.010, num_cepstral=13)
    nfr = int(np.floor((len(sig) - 400) / 160))
    assert f.shape == (nfr, 13) and np.isfinite(f).all()
    fb, en = P.mfe(sig, fs, frame_length=0.025, frame_stride=0.010, num_filters=40)
    assert fb.shape == (nfr, 40) and en.shape == (nfr,) and (fb > 0).all()
    assert np.allclose(f[:, 0], np.log(en))                       # dc_elimination: c0 = log frame energy
    n = P.cmvn(f, True)
    assert np.abs(n.mean(0)).max() < 1e-9 and np.abs(n.std(0) - 1).max() < 1e-6
    cube = P.extract_derivative_feature(n)
    assert cube.shape == (nfr, 13, 3) and np.array_equal(cube[:, :, 0], n)
    # the two sinusoids land in the right mel bands
    banks = P.filterbanks(40, 257, fs, 0, None)
    assert banks.shape == (40, 257) and banks.min() >= 0 and banks.max() <= 1.0 + 1e-12
    centre = lambda hz: int(np.argmax(banks[:, int(round(hz * 512 / fs))]))        # the filter that covers that rfft bin
    top4 = set(np.argsort(fb.mean(0))[-4:])
    assert centre(440) in top4 and centre(1800) in top4


def test_process_audios_and_texts(tmp_path):
    import preprocess as P
    from utils.tokenizer import CharEncoder
    from las.arguments import parse_args
    a = parse_args([])
    a.feat_type, a.feat_dim, a.cmvn = "mfcc", 13, True
    p = str(tmp_path / "utt.npy")
    np.save(p, np.random.RandomState(1).randn(16000).astype(np.float32) * 0.1)
    feats, featlen = P.process_audios([p], a)
    assert feats[0].dtype == np.float32 and feats[0].shape[1:] == (13, 3) and featlen[0] == feats[0].shape[0]
    toks, tl = P.process_texts(["HELLO, WORLD!"], CharEncoder())
    assert toks[0] == CharEncoder().encode("HELLO WORLD", with_eos=True) and tl[0] == len(toks[0])
