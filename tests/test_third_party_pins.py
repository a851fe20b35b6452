"""Known-answer pins for the two third-party algorithms the path restates from their published definitions:
OpenCV's 8-bit INTER_LINEAR resize (oracle/jegal_oracle.py:cv_resize_linear_u8, the checker of jg_mask_resize) and librosa's
Slaney mel filter bank (jegal_amd/audio.py:mel_filterbank, the basis jg_logmel is fed).  cv2 and librosa are not in the build image
and not in /root/reference, so the expected values are HAND-DERIVED from the published formulas; the derivations sit next to the
numbers in tests/golden/third_party_known_answers.json.  What this does not pin: the compiled cv2 wheel bit for bit (IPP
dispatch) and librosa's float32 rounding of every entry -- those stay "parity unpinned" (DESIGN.md section 2)."""
import json
import os

import numpy as np

import jegal_oracle as O
from jegal_amd.audio import _hz_to_mel, _mel_to_hz, mel_filterbank


def _ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "third_party_known_answers.json")))


def test_slaney_mel_basis_known_answers(golden_dir):
    k = _ka(golden_dir)["mel_slaney"]
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), 82))
    for i, hz in k["edges_hz"].items():
        assert abs(edges[int(i)] - hz) <= 2e-6 * max(hz, 1.0) + 1e-9, (i, edges[int(i)], hz)
    mb = mel_filterbank(16000, 512, 80, 0.0, 8000.0)
    assert mb.shape == (80, 257) and mb.dtype == np.float32
    for b, v in k["filter0"].items():
        assert abs(float(mb[0, int(b)]) - v) < 2e-8, (b, mb[0, int(b)], v)
    assert abs(2.0 / (edges[2] - edges[0]) - k["enorm_0"]) < 1e-8
    # structure: every filter is one triangle (non-negative, single run of support), peak inside its band, ~unit area
    area = mb.astype(np.float64).sum(1) * 31.25
    lo, hi = k["row_sum_times_bin_hz_range"]
    assert (area > lo).all() and (area < hi).all(), (area.min(), area.max())
    freqs = np.arange(257) * 31.25
    for m in range(80):
        nz = np.nonzero(mb[m])[0]
        assert len(nz) and (np.diff(nz) == 1).all() and (mb[m] >= 0).all()
        assert edges[m] < freqs[nz[0]] and freqs[nz[-1]] < edges[m + 2]
        assert edges[m] <= freqs[np.argmax(mb[m])] <= edges[m + 2]
    d = k["librosa_doc_example"]
    ex = mel_filterbank(**{kk: (int(v) if kk in ("sr", "n_fft", "n_mels") else v) for kk, v in d["args"].items()})
    assert np.allclose(np.round(ex[0, :2].astype(np.float64), 3), d["row0_first2_rounded3"], atol=1e-9)
    assert np.allclose(np.round(ex[1, :2].astype(np.float64), 3), d["row1_first2_rounded3"], atol=1e-9)


def test_cv2_inter_linear_u8_known_answers(golden_dir):
    k = _ka(golden_dir)["cv2_resize_inter_linear_u8"]
    c = k["constant"]
    img = np.full(tuple(c["src_hw"]) + (3,), c["value"], np.uint8)
    out = O.cv_resize_linear_u8(img, c["dst_hw"][1], c["dst_hw"][0])
    assert out.shape == tuple(c["dst_hw"]) + (3,) and (out == c["value"]).all()
    for v in (0, 1, 254, 255):                                       # and at the ends of the range, both directions
        assert (O.cv_resize_linear_u8(np.full((300, 500, 1), v, np.uint8), 480, 270) == v).all()
        assert (O.cv_resize_linear_u8(np.full((228, 314, 1), v, np.uint8), 480, 270) == v).all()
    d = k["down2"]
    src = np.asarray(d["src"], np.uint8)[:, :, None]
    assert O.cv_resize_linear_u8(src, 2, 2)[:, :, 0].tolist() == d["dst"]
    rng = np.random.default_rng(0)                                    # the 2:1 rule on random data: (sum of the 2x2 block + 2) >> 2
    r = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    blk = r.astype(np.int64).reshape(20, 2, 32, 2, 3).sum((1, 3))
    assert np.array_equal(O.cv_resize_linear_u8(r, 32, 20), ((blk + 2) >> 2).astype(np.uint8))
    u = k["up2_row"]
    assert O.cv_resize_linear_u8(np.asarray(u["src"], np.uint8)[:, :, None], u["dst_w"], 1)[:, :, 0].tolist() == u["dst"]
    t = k["to3x3"]
    got = O.cv_resize_linear_u8(np.asarray(t["src"], np.uint8)[:, :, None], 3, 3)[:, :, 0]
    for pos, v in t["dst_known"].items():
        y, x = (int(a) for a in pos.split(","))
        assert int(got[y, x]) == v, (pos, got[y, x], v)
    # identity size is the identity (f = 0 everywhere: a0 = b0 = 2048)
    assert np.array_equal(O.cv_resize_linear_u8(r, 64, 40), r)
