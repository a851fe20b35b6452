"""Pin oracle/jegal_oracle.py against outputs of the REAL reference (tests/golden/*, made by
oracle/make_golden.py in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

TOL = 2e-5


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def gsd():
    return O.tensors(synth.gestsync_state_dict(include_unused=False))


@pytest.fixture(scope="module")
def jsd():
    return O.tensors(synth.jegal_state_dict())


def test_gestsync_clip_matches_reference(golden_dir, gsd):
    g = np.load(os.path.join(golden_dir, "gestsync_clip.npz"))
    T = int(g["T"])
    frames = synth.synth_frames(int(g["seed"]), 1, T)[0]
    f01 = torch.from_numpy(frames.astype(np.float32) / np.float32(255.0))
    with torch.no_grad():
        dedup = O.gestsync_clip_feats(gsd, f01, naive=False)
        naive = O.gestsync_clip_feats(gsd, f01, naive=True)
    assert rel(naive, g["feats"]) < TOL
    assert rel(dedup, g["feats"]) < TOL
    # window de-duplication is exact up to conv chunking noise
    assert rel(dedup, naive) < 1e-6


def test_gestsync_forward_vid_parts(golden_dir, gsd):
    g = np.load(os.path.join(golden_dir, "gestsync_clip.npz"))
    with torch.no_grad():
        out = O.gestsync_head(gsd, torch.from_numpy(g["out_conv"][:2]))
    assert rel(out, g["out_full"]) < TOL
    frames = synth.synth_frames(int(g["seed"]), 1, int(g["T"]))[0]
    f01 = O.pad_clip(torch.from_numpy(frames.astype(np.float32) / np.float32(255.0)))
    x = f01[:25].permute(3, 0, 1, 2).unsqueeze(0)
    with torch.no_grad():
        c1 = O.vgg_vid(gsd, x[:, :, :5], upto="conv1")
    assert rel(c1[0, :, 0, ::6, ::6], g["conv1_pool_t0"]) < TOL


def test_jegal_gesture(golden_dir, jsd):
    g = np.load(os.path.join(golden_dir, "jegal_gesture.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    vf = rng.standard_normal((2, 40, 1024)).astype(np.float32)
    vf[1, 30:] = 0
    vm = np.ones((2, 40), np.float32)
    vm[1, 30:] = 0
    with torch.no_grad():
        out = O.jegal_forward_inference(jsd, visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm))
        fg = O.jegal_forward_gestures(jsd, torch.from_numpy(vf), torch.from_numpy(vm).unsqueeze(1))
    assert rel(out, g["gesture"]) < TOL
    assert rel(fg, g["fwd_gestures"]) < TOL


AUDIO_WB = [[["a", 3, 9], ["b", 10, 10], ["c", 12, 30]], [["d", 0, 5], ["e", 6, 20]]]


def test_jegal_audio(golden_dir, jsd):
    g = np.load(os.path.join(golden_dir, "jegal_audio.npz"))
    mel = torch.from_numpy(synth.synth_mel(int(g["seed"]), 2, 160))
    with torch.no_grad():
        fa = O.jegal_forward_audio(jsd, mel)
        c = O.jegal_forward_inference(jsd, audio=mel, audio_mask=torch.ones(2, 40), word_boundaries=AUDIO_WB)
    assert rel(fa, g["fwd_audio"]) < TOL
    assert c.shape == g["content"].shape
    assert rel(c, g["content"]) < TOL


def _text_pack(g):
    tbatch = [["w0", "w1", "w2", "w3"], ["x0", "x1", "x2"]]
    return (g["states"], g["mask"], tbatch, g["ids"], g["offsets"])


def test_jegal_text(golden_dir, jsd):
    g = np.load(os.path.join(golden_dir, "jegal_text.npz"))
    with torch.no_grad():
        ft = O.jegal_forward_text(jsd, torch.from_numpy(g["states"]), torch.from_numpy(g["mask"]).unsqueeze(1))
        c = O.jegal_forward_inference(jsd, text=_text_pack(g))
    assert rel(ft, g["fwd_text"]) < TOL
    assert c.shape == g["content"].shape
    assert rel(c, g["content"]) < TOL


def test_jegal_vta(golden_dir, jsd):
    g = np.load(os.path.join(golden_dir, "jegal_vta.npz"))
    gt = np.load(os.path.join(golden_dir, "jegal_text.npz"))
    rng = np.random.default_rng(9002)
    vf = rng.standard_normal((2, 40, 1024)).astype(np.float32)
    vf[1, 30:] = 0
    vm = np.ones((2, 40), np.float32)
    vm[1, 30:] = 0
    wb2 = [[["w0", 2, 6], ["w1", 7, 12], ["w2", 13, 13], ["w3", 15, 30]], [["x0", 1, 4], ["x1", 5, 9], ["x2", 10, 22]]]
    mel = torch.from_numpy(synth.synth_mel(9003, 2, 160))
    with torch.no_grad():
        ge, ce = O.jegal_forward_inference(jsd, visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm),
                                           text=_text_pack(gt), audio=mel, audio_mask=torch.ones(2, 40), word_boundaries=wb2)
    assert rel(ge, g["gesture"]) < TOL
    assert rel(ce, g["content"]) < TOL


def test_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    ge, ce = synth.planted_retrieval(9005, 64)
    ge[7] = ge[3]
    sim = O.similarity_matrix(ce, ge).numpy()
    np.testing.assert_allclose(sim, g["sim"], atol=1e-6)
    m = O.compute_metrics(g["sim"])
    for k in ("R5", "R10", "R25", "R50", "MR"):
        assert m[k] == float(g[k]), k
    gest, cont, bounds, targets = synth.planted_spotting(9006, 20, n_frames=60, n_words=10, noise=2.0)
    assert O.spotting_accuracy(gest, cont, bounds, targets) == pytest.approx(float(g["spot_acc"]))
    np.testing.assert_allclose(O.attn_matrix(gest[0], cont[0]), g["attn0"], atol=1e-6)
    for P in (2, 4, 6):
        np.testing.assert_allclose(O.similarity_cos(ce[:1], ge[:P]), g[f"asd{P}"], atol=1e-6)


def test_load_text(golden_dir, tmp_path):
    ref = json.load(open(os.path.join(golden_dir, "load_text.json")))
    for name, r in ref.items():
        p = tmp_path / (name + ".txt")
        p.write_text(r["file"], encoding="utf-8")
        text, wbs = O.load_text(str(p))
        assert text == r["text"]
        assert wbs == r["word_boundaries"]


def test_cv_resize_restatement_properties():
    """The cv2.resize(INTER_LINEAR, uint8) restatement (parity unpinned: cv2 absent) at least has the properties the
    published algorithm guarantees: identity at equal size, constants preserved, exact 2x box average within rounding."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    assert np.array_equal(O.cv_resize_linear_u8(img), img)
    assert np.all(O.cv_resize_linear_u8(np.full((720, 1280, 3), 137, np.uint8)) == 137)
    big = rng.integers(0, 256, (540, 960, 3), dtype=np.uint8)
    box = big.astype(np.float32).reshape(270, 2, 480, 2, 3).mean((1, 3))
    assert np.abs(O.cv_resize_linear_u8(big).astype(np.float32) - box).max() <= 0.5
    out = O.mask_resize_frames(rng.integers(0, 256, (2, 300, 500, 3), dtype=np.uint8), [-1, 100])
    assert out.shape == (2, 270, 480, 3) and out[0, :111].max() == 0 and out[0, 111].max() > 0 and out[1, :80].max() == 0


def test_xlmr_oracle_matches_transformers(golden_dir):
    """the XLM-RoBERTa restatement against transformers.XLMRobertaModel itself (seeded weights of
    synth.xlmr_state_dict strict-loaded into the third-party model by oracle/make_golden.py xlmr)."""
    g = np.load(os.path.join(golden_dir, "xlmr.npz"))
    sd = synth.xlmr_state_dict()
    with torch.no_grad():
        out = O.xlmr_forward(sd, g["input_ids"], g["attention_mask"]).numpy()
        out1 = O.xlmr_forward(sd, g["input_ids"][:1]).numpy()
    assert np.abs(out - g["last_hidden_state"]).max() < 2e-5
    assert np.abs(out1 - g["last_hidden_state_nomask"]).max() < 2e-5


def test_xlmr_oracle_matches_transformers_12_layers(golden_dir):
    """The full depth of xlm-roberta-base (12 layers, 514 positions; reduced vocabulary): tests/golden/xlmr12.npz, round 3."""
    g = np.load(os.path.join(golden_dir, "xlmr12.npz"))
    assert int(g["layers"]) == 12
    sd = synth.xlmr_state_dict(layers=12)
    with torch.no_grad():
        out = O.xlmr_forward(sd, g["input_ids"], g["attention_mask"]).numpy()
    m = g["attention_mask"].astype(bool)
    assert np.abs(out[m] - g["last_hidden_state"][m]).max() < 5e-5


def test_logmel_oracle_and_load_wav_match_the_reference(golden_dir):
    """tests/golden/logmel_sample1.npz holds what the reference's OWN utils/audio_utils.py (load_wav :20-25,
    wav2filterbanks :28-66, imported with an empty librosa module; mel_basis passed in) returns for its samples/sample1.wav.  The
    oracle's torch.stft restatement and the package's load_wav are held to it; the mel basis itself (librosa.filters.mel) stays
    unpinned -- both sides take jegal_amd.audio.mel_filterbank()."""
    import hashlib
    from jegal_amd import audio
    g = np.load(os.path.join(golden_dir, "logmel_sample1.npz"))
    wav = audio.load_wav(os.path.join(golden_dir, "sample1.wav"))
    assert str(wav.dtype) == str(g["wav_dtype"]) and wav.shape == (int(g["n_samples"]),)
    assert hashlib.sha256(np.ascontiguousarray(wav).tobytes()).hexdigest() == str(g["wav_sha256"])
    feats = O.wav2filterbanks(wav.astype("float32")[None], torch.from_numpy(audio.mel_filterbank()))[0].numpy()
    assert feats.shape == g["features"].shape == (216, 80)
    assert np.abs(feats - g["features"]).max() <= 1e-6
