"""Frames -> embedding parity against the oracle on inputs that are NOT uniform noise.

Every frames -> embedding oracle comparison of rounds 1-3 drew its clips from synth.synth_frames (noise under a 110-row mask) -- the
distribution the default precision mode's bias corrections are calibrated on (api.hip, calibrate_impl).  The reference's inputs
are natural crops (inference_embs.py:235-286): smooth, low contrast, saturated regions, a mask that follows the chin per frame.
Three seeded families (synth.synth_frames_structured), two full-length clips each, in JG_PREC_FP16_BC (calibrated on the built-in
noise clips), the calibration-free JG_PREC_FP16_W2 and the run-time corrected default JG_PREC_FP16_RC, measured errors printed."""
import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3
T = 150


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def oracle_sd():
    return O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())


@pytest.fixture(scope="module")
def engines():
    from jegal_amd._lib import Engine, PREC_FP16_BC, PREC_FP16_W2, PREC_FP16_RC
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    out = {}
    for name, mode in (("fp16_bc", PREC_FP16_BC), ("fp16_w2", PREC_FP16_W2), ("fp16_rc", PREC_FP16_RC)):
        e = Engine(0, precision=mode)
        GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
        JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
        out[name] = e
    yield out
    for e in out.values():
        e.close()


@pytest.mark.parametrize("kind", ["smooth", "saturated", "jitter"])
def test_structured_clips_vs_oracle(engines, oracle_sd, kind):
    gsd, jsd = oracle_sd
    frames = synth.synth_frames_structured(4100, 2, T, kind)
    refs = []
    with torch.no_grad():
        for b in range(2):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            refs.append((f.numpy(), O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy()))
    dev = torch.from_numpy(frames).cuda()
    for name, e in engines.items():
        emb = e.extract_gesture(dev).cpu().numpy()
        feats = e.gestsync_clip(dev).cpu().numpy()
        for b in range(2):
            r, mx, rf = rel(emb[b], refs[b][1]), float(np.abs(emb[b] - refs[b][1]).max()), rel(feats[b], refs[b][0])
            print(f"\n[{kind}] {name} clip {b}: embedding rel-L2 {r:.3e} max-abs {mx:.3e} | GestSync feats rel-L2 {rf:.3e}", end="")
            assert np.isfinite(emb[b]).all()
            assert r < TOL and mx < TOL, (kind, name, b, r, mx)
            assert rf < TOL, (kind, name, b, rf)


def test_calibrating_on_structured_clips_keeps_noise_clips_in_tolerance(oracle_sd):
    """The other direction of the calibration question: bias corrections recorded on SMOOTH clips (jg_calibrate_gesture), tested
    on the noise clips of BASELINE configs[1] and on the saturated family."""
    from jegal_amd._lib import Engine, PREC_FP16_BC
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    gsd, jsd = oracle_sd
    e = Engine(0, precision=PREC_FP16_BC)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    e.calibrate(torch.from_numpy(synth.synth_frames_structured(77, 2, 24, "smooth")).cuda())
    for kind, fr in (("noise", synth.synth_frames(1234, 1, T)), ("saturated", synth.synth_frames_structured(4100, 1, T, "saturated"))):
        with torch.no_grad():
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(fr[0].astype(np.float32) / np.float32(255.0)))
            ref = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy()
        emb = e.extract_gesture(torch.from_numpy(fr).cuda())[0].cpu().numpy()
        r = rel(emb, ref)
        print(f"\ncalibrated on smooth clips, tested on {kind}: rel-L2 {r:.3e}", end="")
        assert r < TOL, (kind, r)
    e.close()


def test_rc_row_sample_on_temporally_periodic_clips(engines, oracle_sd):
    """VERDICT r5 item 6.  JG_PREC_FP16_RC takes E[x] of a 150-frame clip from 394 of its 3 150 token rows (16-row runs every 128 rows).  A
    clip whose content oscillates at the sample's own periods (128 / 21 = 6.095 frames between sampled runs, 21 frames, 128 frames;
    synth 'periodic') is where a sampled mean could alias.  Measured against hi+lo weights (no E[x] at all): the sample must not cost
    more than 10 % of W2's error."""
    gsd, jsd = oracle_sd
    frames = synth.synth_frames_structured(4107, 2, T, "periodic")
    dev = torch.from_numpy(frames).cuda()
    errs = {}
    with torch.no_grad():
        refs = []
        for b in range(2):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            refs.append((f.numpy(), O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy()))
    for name in ("fp16_w2", "fp16_rc"):
        emb = engines[name].extract_gesture(dev).cpu().numpy()
        feats = engines[name].gestsync_clip(dev).cpu().numpy()
        errs[name] = (max(rel(emb[b], refs[b][1]) for b in range(2)), max(rel(feats[b], refs[b][0]) for b in range(2)))
        print(f"\n[periodic] {name}: embedding rel-L2 {errs[name][0]:.3e} | GestSync feats {errs[name][1]:.3e}", end="")
    assert errs["fp16_rc"][0] < TOL and errs["fp16_rc"][1] < TOL
    assert errs["fp16_rc"][0] <= 1.1 * errs["fp16_w2"][0] and errs["fp16_rc"][1] <= 1.1 * errs["fp16_w2"][1], errs
