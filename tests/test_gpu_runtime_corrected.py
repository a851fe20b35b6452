"""JG_PREC_FP16_RC (round 5): single-fp16 Linear weights on the GestSync transformer with the systematic part of
the weight-rounding error, (w - fp16(w)) . E[x], rebuilt per GEMM call and PER CLIP from a fixed sample of the clip's own rows --
no calibration pass, nothing depends on calibration data or on the rest of the batch.  Through the C ABI."""
import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def rc():
    from jegal_amd._lib import Engine, PREC_FP16_RC
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    e = Engine(0, precision=PREC_FP16_RC)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    yield e
    e.close()


def test_rc_full_length_clips_vs_oracle(rc):
    gsd, jsd = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())
    T = 150
    frames = synth.synth_frames(4321, 3, T)
    emb = rc.extract_gesture(torch.from_numpy(frames).cuda()).cpu().numpy()
    feats = rc.gestsync_clip(torch.from_numpy(frames).cuda()).cpu().numpy()
    for b in (0, 2):
        with torch.no_grad():
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            g = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy()
        r, mx, rf = rel(emb[b], g), float(np.abs(emb[b] - g).max()), rel(feats[b], f.numpy())
        print(f"\nRC clip {b}: embedding rel-L2 {r:.3e} max-abs {mx:.3e} | GestSync feats {rf:.3e}", end="")
        assert r < TOL and mx < TOL and rf < TOL


def test_rc_result_of_a_clip_does_not_depend_on_the_batch(rc):
    """The correction of a clip comes from ITS rows (sampled relative to the clip's first row): the same clip alone, first, last or in
    the middle of a batch, in either lane of a two-lane batch, gives the same bits (every GEMM instance accumulates k in the same
    order, tests/test_gpu_precision_uploads.py::test_gemm_tile_choice_never_changes_a_bit)."""
    T = 60
    clips = synth.synth_frames(777, 9, T)
    dev = torch.from_numpy(clips).cuda()
    alone_f = rc.gestsync_clip(dev[4:5]).clone()
    alone_e = rc.extract_gesture(dev[4:5]).clone()
    for lo, hi in ((4, 6), (2, 5), (0, 9), (3, 5)):
        assert torch.equal(rc.gestsync_clip(dev[lo:hi])[4 - lo], alone_f[0]), (lo, hi)
        # (the JEGAL branch of a 60-frame clip ALONE takes the register-staged GEMM, M < 128, whose fp32 summation order differs)
        assert rel(rc.extract_gesture(dev[lo:hi])[4 - lo], alone_e[0]) < 1e-5, (lo, hi)
    # two different neighbours: nothing leaks from them
    other = torch.from_numpy(synth.synth_frames_structured(5, 2, T, "saturated")).cuda()
    mixed = rc.gestsync_clip(torch.cat([other[:1], dev[4:5], other[1:]]))
    assert torch.equal(mixed[1], alone_f[0])
    rc.set_option("dual_stream", 0)
    try:
        assert torch.equal(rc.gestsync_clip(dev)[4], alone_f[0])
    finally:
        rc.set_option("dual_stream", 1)


@pytest.mark.parametrize("opt", ["fuse_ln", "gemm_glds", "qkv0_linear", "stream_fp16", "gemm_big_tile", "dual_stream"])
def test_rc_options_fall_back_to_hi_lo_or_stay_within_tolerance(rc, opt):
    """Where the per-clip epilogue is not available (unfused LayerNorm, register-staged GEMM) the run-time corrected layers run hi+lo:
    never single fp16 without its correction."""
    gsd, jsd = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())
    T = 60
    frames = synth.synth_frames(5150, 2, T)
    with torch.no_grad():
        f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[0].astype(np.float32) / np.float32(255.0)))
        g = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
    dev = torch.from_numpy(frames).cuda()
    base = rc.extract_gesture(dev).cpu()
    rc.set_option(opt, 0)
    try:
        alt = rc.extract_gesture(dev).cpu()
    finally:
        rc.set_option(opt, 1)
    print(f"\nRC {opt}=0: rel {rel(alt[0], g):.3e} (default {rel(base[0], g):.3e})", end="")
    assert rel(base[0], g) < TOL and rel(alt[0], g) < TOL
    assert torch.equal(rc.extract_gesture(dev).cpu(), base)


def test_rc_short_and_windowed_inputs(rc):
    """Clips too short for the fused plan (M < 1024 tokens) and the forward_vid drop-in (windows have no clip structure) run hi+lo."""
    from jegal_amd.gestsync import GestSync
    gsd = O.tensors(synth.gestsync_state_dict(include_unused=False))
    frames = synth.synth_frames(31, 1, 5)
    feats = rc.gestsync_clip(torch.from_numpy(frames).cuda()).cpu()
    with torch.no_grad():
        ref = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[0].astype(np.float32) / np.float32(255.0)))
    assert rel(feats[0], ref) < TOL
    x = torch.rand(2, 3, 25, 270, 480)
    out = GestSync(engine=rc).load_state_dict(synth.gestsync_state_dict(include_unused=False)).forward_vid(x.cuda()).cpu()
    with torch.no_grad():
        refw = O.gestsync_forward_vid(gsd, x)
    assert rel(out, refw) < TOL
    with pytest.raises(Exception):
        rc.calibrate()                                   # there is nothing to calibrate in this mode


def test_rc_forward_vid_with_many_windows(rc):
    """forward_vid on >= 49 windows (M >= 1024 tokens, no clip structure): the unfused hi+lo plan, not the LayerNorm-fused kernels that
    only exist with the per-clip bias (before round 6 this call failed with hipErrorInvalidValue)."""
    from jegal_amd.gestsync import GestSync
    gsd = O.tensors(synth.gestsync_state_dict(include_unused=False))
    g = torch.Generator().manual_seed(77)
    x = torch.rand(52, 3, 25, 270, 480, generator=g)
    out = GestSync(engine=rc).load_state_dict(synth.gestsync_state_dict(include_unused=False)).forward_vid(x.cuda()).cpu()
    assert torch.isfinite(out).all()
    pick = [0, 25, 51]
    with torch.no_grad():
        refw = O.gestsync_forward_vid(gsd, x[pick])
    assert rel(out[pick], refw) < TOL


@pytest.mark.parametrize("B,T,chunk", [(5, 13, 32), (40, 25, 64), (3, 37, 32), (2, 24, 32), (5, 12, 32), (8, 10, 32)])
def test_rc_odd_shapes_vs_oracle(rc, B, T, chunk):
    """Rows per clip below the sampling threshold (every row is used), not a multiple of 16 or of the tile height, more than 32 clips in
    one pass (two clip groups in the correction product), a batch too small for the fused plan (M < 1024: hi+lo), and clips too short for
    the per-clip bias (T <= 12, i.e. fewer than 256 rows per clip, in a batch of >= 1024 rows: ADVICE r5 -- the unfused hi+lo plan)."""
    gsd = O.tensors(synth.gestsync_state_dict(include_unused=False))
    frames = synth.synth_frames(900 + T, B, T)
    rc.set_chunk(chunk)
    try:
        feats = rc.gestsync_clip(torch.from_numpy(frames).cuda()).cpu()
    finally:
        rc.set_chunk(32)
    assert torch.isfinite(feats).all()
    worst = 0.0
    for b in sorted({0, B // 2, B - 1}):
        with torch.no_grad():
            ref = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
        worst = max(worst, rel(feats[b], ref))
    print(f"\nRC B={B} T={T} chunk={chunk}: GestSync feats rel-L2 (worst of 3 clips) {worst:.3e}", end="")
    assert worst < TOL


def test_rc_padded_batch_with_lengths_equals_the_clips_alone(rc):
    """jg_gestsync_clip_ragged: clips of 60 and 75 frames padded to 90 with copies of their last frames (how the feature-extraction
    driver batches).  With the lengths each clip's correction comes from its OWN rows: the kept rows are bit-identical to the clip
    run alone and to another padding length; WITHOUT the lengths the padding rows enter the statistics (still within tolerance here,
    but no longer the clip's own result)."""
    lens = [60, 75]
    clips = [synth.synth_frames(70 + i, 1, t)[0] for i, t in enumerate(lens)]

    def padded(Tpad, with_lengths=True):
        batch = np.empty((2, Tpad, 270, 480, 3), np.uint8)
        for i, c in enumerate(clips):
            batch[i, :lens[i]] = c
            batch[i, lens[i]:] = c[-1]
        return rc.gestsync_clip(torch.from_numpy(batch).cuda(), lengths=lens if with_lengths else None)
    a, b, loose = padded(90), padded(101), padded(90, with_lengths=False)
    for i, c in enumerate(clips):
        alone = rc.gestsync_clip(torch.from_numpy(c[None]).cuda())[0]
        assert torch.equal(a[i, :lens[i]], alone), i
        assert torch.equal(b[i, :lens[i]], alone), i
        d = rel(loose[i, :lens[i]], alone)
        print(f"\nclip {i}: padded batch without lengths vs the clip alone: rel-L2 {d:.2e}", end="")
        assert d < TOL
    with pytest.raises(Exception):
        rc.gestsync_clip(torch.from_numpy(clips[0][None]).cuda(), lengths=[61])          # beyond T


def test_rc_short_clip_in_a_padded_batch_is_within_tolerance_of_itself_alone(rc):
    """the bit-identity of jg_gestsync_clip_ragged's kept rows with the clip run alone holds for clips of >= 49 frames (alone
    they take the same fused plan).  A 30-frame clip alone has 630 token rows < 1 024 and runs the unfused hi+lo plan; padded to 64 frames
    next to a longer clip it runs the fused run-time corrected plan: equal within the contract, not bit for bit -- and still independent
    of the padding length and of its neighbour."""
    gsd = O.tensors(synth.gestsync_state_dict(include_unused=False))
    lens = [30, 64]
    clips = [synth.synth_frames(170 + i, 1, t)[0] for i, t in enumerate(lens)]

    def padded(Tpad, order=(0, 1)):
        batch = np.empty((2, Tpad, 270, 480, 3), np.uint8)
        for slot, i in enumerate(order):
            batch[slot, :lens[i]] = clips[i]
            batch[slot, lens[i]:] = clips[i][-1]
        out = rc.gestsync_clip(torch.from_numpy(batch).cuda(), lengths=[lens[i] for i in order])
        return {i: out[slot, :lens[i]].clone() for slot, i in enumerate(order)}
    a, b, c = padded(64), padded(80), padded(64, order=(1, 0))
    alone = rc.gestsync_clip(torch.from_numpy(clips[0][None]).cuda())[0]
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0])          # independent of padding length and batch position
    d = rel(a[0], alone)
    with torch.no_grad():
        ref = O.gestsync_clip_feats(gsd, torch.from_numpy(clips[0].astype(np.float32) / np.float32(255.0)))
    print(f"\n30-frame clip: padded batch (fused plan) vs alone (unfused hi+lo plan) rel-L2 {d:.2e}; vs the oracle {rel(a[0], ref):.2e} / {rel(alone, ref):.2e}", end="")
    assert d < TOL and rel(a[0], ref) < TOL and rel(alone, ref) < TOL
    assert torch.equal(a[1], rc.gestsync_clip(torch.from_numpy(clips[1][None]).cuda())[0])      # the 64-frame clip: bit-identical to itself alone


def test_rc_long_clip_whole_path(rc):
    """One 260-frame clip (longer than any AVS clip: 220) through the whole gesture path in the default mode: 5 460 token rows per
    clip (the every-eighth-run sample: 43 runs), the JEGAL branch on the online-softmax attention (S = 260 > 160)."""
    gsd, jsd = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())
    T = 260
    frames = synth.synth_frames(2600, 1, T)
    emb = rc.extract_gesture(torch.from_numpy(frames).cuda()).cpu().numpy()
    with torch.no_grad():
        f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[0].astype(np.float32) / np.float32(255.0)))
        g = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy()
    r, mx = rel(emb[0], g), float(np.abs(emb[0] - g).max())
    print(f"\nRC T = {T}: embedding rel-L2 {r:.3e} max-abs {mx:.3e}", end="")
    assert r < TOL and mx < TOL


def test_two_lane_gesture_path_is_reproducible_under_a_poisoned_workspace(rc):
    """Companion of test_xlmr_is_reproducible_under_a_poisoned_workspace (round 6): 32 clips run as two lanes on two streams; with the
    workspace filled with NaN bytes before every chunk, a read of anything the step has not written itself cannot hide behind the
    previous run's values.  40 runs, every GEMM tile choice: bit-identical."""
    frames = torch.from_numpy(synth.synth_frames(5, 32, 30)).cuda()
    base = rc.extract_gesture(frames).clone()
    assert torch.isfinite(base).all()
    rc.set_option("ws_poison", 1)
    try:
        for it in range(40):
            rc.set_option("gemm_tile", it & 3)
            assert torch.equal(rc.extract_gesture(frames), base), it
    finally:
        rc.set_option("ws_poison", 0)
        rc.set_option("gemm_tile", 0)
