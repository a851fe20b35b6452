"""Engine options, long sequences, scale and multi-rank (run with -m gpu): the kernel paths and switches the golden / end-to-end
file does not reach -- long sequences (VALU attention, T in (160,500], text L > 32), every engine option, the W2_ALL precision mode,
bias-correction calibrated on mismatched data, a non-degenerate ASD fixture from the reference's own evaluate_asd,
the lifted spotting limits, >= 8 clips of configs 2 and 3 against the oracle, two engines in one process, and the
2-rank sharded retrieval on the GPU.  Everything goes through the C ABI; tolerance as in test_gpu_golden_endtoend.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def engine():
    from jegal_amd._lib import Engine
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return Engine.get("cuda:0")


@pytest.fixture(scope="module")
def models(engine):
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    gs = GestSync(engine=engine).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    jg = JEGAL(engine=engine).load_state_dict(synth.jegal_state_dict())
    return gs, jg


@pytest.fixture(scope="module")
def oracle_sd():
    return O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())


# ------------------------------------------------------------------ (a) long sequences: VALU attention + MFMA NB = 5
@pytest.mark.parametrize("T", [150, 161, 220, 500])
def test_jegal_gesture_long_clips(models, oracle_sd, T):
    """JEGAL clips are 25-220 frames (dataset/avs_*.csv), the PE table allows 500 (modules.py:136): T = 161, 220 and 500
    run attn_kernel<64> (VALU), T = 150 the 5-block MFMA kernel; ragged batch with a key mask (modules.py:61-75)."""
    _, jg = models
    _, jsd = oracle_sd
    rng = np.random.default_rng(700 + T)
    vf = rng.standard_normal((2, T, 1024)).astype(np.float32)
    valid = T - 37
    vf[1, valid:] = 0
    vm = np.ones((2, T), np.float32)
    vm[1, valid:] = 0
    out = jg.forward_inference(visual_feats=torch.from_numpy(vf).cuda(), visual_mask=torch.from_numpy(vm).cuda())
    with torch.no_grad():
        ref = O.jegal_forward_inference(jsd, visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm))
    r0, r1 = rel(out[0], ref[0]), rel(out[1, :valid], ref[1, :valid])
    print(f"T={T}: rel {r0:.3e} / {r1:.3e} (padded clip, valid rows)")
    assert r0 < TOL and r1 < TOL
    # The reference also computes the PADDED query rows of clip 1 (zero feature rows; the caller strips them): held to the
    # same 1e-3.  (Round 2 allowed 2e-3: the bias correction (w - fp16(w)).E[x] of the default mode is pure error on x = 0;
    # the input projection proj_ip_rgb now keeps hi+lo weights in that mode, api.hip finalize_jegal.)
    assert rel(out[1], ref[1]) < TOL and rel(out[1, valid:], ref[1, valid:]) < TOL


@pytest.mark.parametrize("L", [33, 70, 200])
def test_jegal_text_long_sequences(models, oracle_sd, L):
    """Text encoder (d = 768, dk = 96: attn_kernel<96>) beyond one 32-key block, with a padded clip."""
    _, jg = models
    _, jsd = oracle_sd
    rng = np.random.default_rng(800 + L)
    st = rng.standard_normal((2, L, 768)).astype(np.float32)
    mk = np.ones((2, L), np.float32)
    mk[1, L - 9:] = 0
    out = jg.forward_text(torch.from_numpy(st).cuda(), torch.from_numpy(mk).cuda().unsqueeze(1))
    with torch.no_grad():
        ref = O.jegal_forward_text(jsd, torch.from_numpy(st), torch.from_numpy(mk).unsqueeze(1))
    assert rel(out, ref) < TOL


# ------------------------------------------------------------------ (b) every A/B switch and W2_ALL
OPTIONS = ["attn_mfma", "fuse_ln", "gemm_glds", "gemm_persistent", "gemm_big_tile", "gemm_small_tile", "gemm_tall_tile",
           "gemm_counted", "conv1_zero_skip", "conv2_row_skip", "qkv0_linear", "conv1_direct", "edge_dedup", "dual_stream", "conv1_mfma16", "stream_fp16"]


@pytest.fixture(scope="module")
def option_case(oracle_sd):
    gsd, jsd = oracle_sd
    B, T = 3, 60                               # M = 3*60*21 = 3780 tokens: fused LN path, LDS-DMA GEMMs, zero tiles in conv1
    frames = synth.synth_frames(5150, B, T)
    with torch.no_grad():
        ref = []
        for b in range(B):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            ref.append(O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]))
    return torch.from_numpy(frames).cuda(), torch.stack(ref)


@pytest.mark.parametrize("opt", OPTIONS)
def test_every_option_stays_within_tolerance(engine, models, option_case, opt):
    """include/jegal_hip.h promises "results stay within the parity tolerance either way" for every jg_set_option
    switch: flip each one off (the fallback kernel path) and compare the whole gesture path with the oracle."""
    frames, ref = option_case
    base = engine.extract_gesture(frames).cpu()
    engine.set_option(opt, 0)
    try:
        alt = engine.extract_gesture(frames).cpu()
    finally:
        engine.set_option(opt, 1)
    again = engine.extract_gesture(frames).cpu()
    assert torch.equal(again, base), "switching an option off and on again must restore the default path bit for bit"
    e_base, e_alt = rel(base, ref), rel(alt, ref)
    print(f"{opt}=0: rel {e_alt:.3e} (default {e_base:.3e}); |alt - default| rel {rel(alt, base):.3e}")
    assert e_base < TOL and e_alt < TOL
    if opt in ("conv1_zero_skip", "conv2_row_skip", "edge_dedup", "gemm_persistent", "gemm_counted", "dual_stream"):
        assert torch.equal(alt, base), f"{opt} only changes scheduling / skips exact zeros: must be bit-identical"


def test_gemm_stagger_option_is_bit_identical(engine, models, option_case):
    frames, _ = option_case
    base = engine.extract_gesture(frames)
    for v in (-1, 300):
        engine.set_option("gemm_stagger", v)
        try:
            assert torch.equal(engine.extract_gesture(frames), base)
        finally:
            engine.set_option("gemm_stagger", 0)


def test_precision_w2_all(option_case):
    from jegal_amd._lib import Engine, PREC_FP16_W2_ALL
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    frames, ref = option_case
    e = Engine(0, precision=PREC_FP16_W2_ALL)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    err = rel(e.extract_gesture(frames).cpu(), ref)
    e.close()
    print("W2_ALL rel", err)
    assert err < TOL


def test_conv1_zero_band_skip_is_bit_identical(engine, models):
    """conv1 skips input tiles that conv1_zero_scan_kernel finds all-zero (bands of 16 rows, all 5 frames of a position).
    Skipping must never change a bit.  Patterns: the reference's face mask (zero prefix), no zero row at all, zero bands in
    the middle and at the bottom, whole black frames (fully skipped strips), black clips next to normal ones, and a single
    non-zero byte hidden in an otherwise black band (defeats the 16-byte probe, caught by the full row check)."""
    rng = np.random.default_rng(77)
    T = 9
    clips = rng.integers(0, 256, (6, T, 270, 480, 3), dtype=np.uint8)
    clips[0, :, :110] = 0                                   # face mask
    clips[2, :, 60:150] = 0                                 # band in the middle
    clips[2, :, 230:] = 0                                   # ... and the bottom
    clips[3] = 0                                            # black clip
    clips[3, 4, 135, 7, 1] = 3                              # one byte in one frame
    clips[4, :, :200] = 0
    clips[4, 2:5] = 0                                       # black frames inside a masked clip
    clips[5, :, :110] = 0
    clips[5, 6, 50, 479, 2] = 255                           # last byte of a row inside the mask
    frames = torch.from_numpy(clips).cuda()
    engine.set_option("conv1_zero_skip", 0)
    try:
        ref = engine.debug_conv1_pool(frames, 4).clone()
    finally:
        engine.set_option("conv1_zero_skip", 1)
    out = engine.debug_conv1_pool(frames, 4)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, ref)
    # and against the stack + implicit-GEMM formulation (independent kernels): same fp16 operands, fp32 sums in another order
    engine.set_option("conv1_direct", 0)
    try:
        alt = engine.debug_conv1_pool(frames, 4)
    finally:
        engine.set_option("conv1_direct", 1)
    assert rel(out.float(), alt.float()) < 3e-4


def _expected_conv_rows(zero_rows, pad=4):
    """numpy restatement of the zero-band scan: zero_rows (B,T) = number of leading all-zero rows of every frame -> per position
    s2 (conv1.hip, conv1_s2_of_mask) -> the output pixels conv2 .. conv5 compute (common.h, conv_skip_decode)."""
    B, T = zero_rows.shape
    P = T + 2 * pad - 4
    bands = np.zeros((B, T), np.int64)                      # leading all-zero 16-row bands (band rt = rows 12 rt .. 12 rt + 15)
    for rt in range(22):
        ok = (np.minimum(12 * rt + 16, 270) <= zero_rows) & (bands == rt)
        bands = bands + ok
    s2 = np.zeros((B, P), np.int64)
    for p in range(P):
        fr = np.clip(p + np.arange(5) - pad, 0, T - 1)
        nb = bands[:, fr].min(axis=1)                       # bands zero in all five frames
        # tile rt is skipped when bands rt and rt-1 are zero -> L = nb (for a prefix of zero bands), s2 = clamp(L - 2, 0, 19)
        s2[:, p] = np.clip(nb - 2, 0, 19)
    s3 = s2 // 2
    rows = [((20 - s2) * 37).sum(), ((10 - s3) * 19).sum(), ((10 - np.maximum(s3 - 1, 0)) * 10).sum(), ((10 - np.maximum(s3 - 2, 0)) * 10).sum()]
    full = [B * P * 740, B * P * 190, B * P * 100, B * P * 100]
    return [int(r) for r in rows], full, int(s2.min())


def _run_row_skip_case(engine, clips, zero_rows):
    frames = torch.from_numpy(clips).cuda()
    engine.set_option("ws_poison", 1)           # rows that the skips leave unwritten are NaN: reading one would show
    engine.set_option("dual_stream", 0)         # one conv stack per call: jg_debug_conv_rows reports the last one
    try:
        out = engine.extract_gesture(frames).clone()
        got_min = engine.debug_conv2_rowskip()
        got_rows, got_full = engine.debug_conv_rows()
    finally:
        engine.set_option("ws_poison", 0)
        engine.set_option("dual_stream", 1)
    assert torch.isfinite(out).all()
    engine.set_option("conv2_row_skip", 0)
    try:
        ref = engine.extract_gesture(frames)
        assert engine.debug_conv2_rowskip() == 0 and engine.debug_conv_rows()[0] == [0, 0, 0, 0]
    finally:
        engine.set_option("conv2_row_skip", 1)
    want_rows, want_full, want_min = _expected_conv_rows(zero_rows)
    assert (got_rows, got_full, got_min) == (want_rows, want_full, want_min)
    assert torch.equal(out, ref)
    return out


def test_conv2_row_skip_follows_the_zero_bands(engine, models):
    """Behind conv1's zero-band skip the leading rows of conv2 .. conv5 of a position do not depend on the position: the layers
    leave them out and their consumers read them from the const chain built at weight load.  The count is PER POSITION (round 3;
    it used to be the minimum over the launch): rows 0..109 masked -> bands 0..7 zero -> tiles 0..7 skipped (L = 8) -> conv2
    skips L - 2 = 6 rows (conv3: 3, conv4: 2, conv5: 1) of that position.  A clip with a smaller mask, or ONE unmasked byte in
    one frame, only costs the positions that see it; unmasked input skips nothing, black clips all but the last row; and no
    case changes a bit of the embeddings.  The workspace is poisoned with NaN patterns: reading a left-out row would show."""
    rng = np.random.default_rng(99)
    B, T = 2, 30
    a = rng.integers(1, 256, (B, T, 270, 480, 3), dtype=np.uint8); a[:, :, :110] = 0
    za = np.full((B, T), 110)
    _run_row_skip_case(engine, a, za)
    b = a.copy(); b[1, :, :110] = rng.integers(1, 256, (T, 110, 480, 3), dtype=np.uint8); b[1, :, :64] = 0      # bands 0..4 -> L = 5
    zb = za.copy(); zb[1] = 64
    _run_row_skip_case(engine, b, zb)
    c = a.copy(); c[0, 17, 3, 100, 1] = 9              # one byte in one frame of one clip: the five positions that read it have L = 0
    zc = za.copy(); zc[0, 17] = 3
    _run_row_skip_case(engine, c, zc)
    d = rng.integers(1, 256, (B, T, 270, 480, 3), dtype=np.uint8)
    _run_row_skip_case(engine, d, np.zeros((B, T), np.int64))
    e = np.zeros((B, T, 270, 480, 3), dtype=np.uint8)   # black clips: every row is the constant row
    _run_row_skip_case(engine, e, np.full((B, T), 270))


def test_row_skip_with_per_frame_mask_heights(engine, models):
    """The reference blanks rows 0..y2+15 PER FRAME (inference_embs.py:264-270): y2 follows the chin.  Mask heights drawn per
    frame from 80..140: every position skips what ITS five frames allow (bit-identical to computing everything), and the
    computed-row counts match the numpy restatement of the scan."""
    rng = np.random.default_rng(123)
    B, T = 3, 40
    clips = rng.integers(1, 256, (B, T, 270, 480, 3), dtype=np.uint8)
    zr = rng.integers(80, 141, (B, T))
    for b in range(B):
        for t in range(T):
            clips[b, t, :zr[b, t]] = 0
    _run_row_skip_case(engine, clips, zr)
    rows, full, _ = _expected_conv_rows(zr)
    print("computed / full rows conv2..conv5:", [f"{r / f:.3f}" for r, f in zip(rows, full)])
    assert rows[0] < 0.85 * full[0]                     # the jittered masks still skip a good part of conv2


def test_dual_stream_lanes_are_bit_identical_and_stream_ordered(engine, models):
    """jg_extract_gesture runs batches of >= 8 clips as two parts on two internal streams.  Same bits as one stream; the
    caller's stream is joined on both sides: frames produced on a side stream just before the call and embeddings consumed
    right after it, without any host synchronisation, must behave as with a single stream."""
    B, T = 16, 50                              # parts of 6 and 10 clips
    host = synth.synth_frames(777, B, T)
    frames = torch.from_numpy(host).cuda()
    engine.set_option("dual_stream", 0)
    try:
        ref = engine.extract_gesture(frames).clone()
    finally:
        engine.set_option("dual_stream", 1)
    engine.set_option("ws_poison", 1)
    try:
        out = engine.extract_gesture(frames).clone()
    finally:
        engine.set_option("ws_poison", 0)
    assert torch.equal(out, ref)
    feats = engine.gestsync_clip(frames).clone()                # jg_gestsync_clip runs in lanes as well
    engine.set_option("dual_stream", 0)
    try:
        assert torch.equal(engine.gestsync_clip(frames), feats)
    finally:
        engine.set_option("dual_stream", 1)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    pinned = torch.from_numpy(host).pin_memory()
    with torch.cuda.stream(side):
        for _ in range(3):
            dev = torch.empty_like(frames)
            dev.copy_(pinned, non_blocking=True)                 # the call must wait for this copy ...
            emb = engine.extract_gesture(dev)
            s = emb.double().sum()                                # ... and this reduction for the two lanes
        total = float(s.item())
    assert total == float(ref.double().sum().item())


def test_lane_priority_option(models):
    """The lane streams are created with the device's highest priority by default (their own hardware queues, whatever streams the
    application owns); the option moves one or both back to normal priority.  Same bits in every setting, with a crowd of used
    application streams around, also when the option changes between two calls."""
    from jegal_amd._lib import Engine, JegalError
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    gs, jg = models
    crowd = []
    for _ in range(6):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda").add_(1)
        crowd.append(st)
    torch.cuda.synchronize()
    frames = torch.from_numpy(synth.synth_frames(4242, 16, 40)).cuda()
    outs = []
    for prio in (3, 0, 1, 2):
        e = Engine(0)
        try:
            e.set_option("lane_priority", prio)
            GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
            JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
            e.set_option("ws_poison", 1)
            outs.append(e.extract_gesture(frames).clone())
            e.set_option("lane_priority", 3 - prio)             # a change after the first two-lane call drains and re-creates the lane streams
            outs.append(e.extract_gesture(frames).clone())
            with pytest.raises(JegalError, match="0..3"):
                e.set_option("lane_priority", 4)
        finally:
            e.close()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_conv1_call_sequence_stress(engine, models):
    """Back-to-back conv1 launches of changing geometry (clip lengths 6..60, both paddings, zero-skip on and off) without any
    host synchronisation in between, each checked against the first result for its geometry.  Guards the hand-counted waits of the
    loader waves (inline-asm frame loads: a VALU-written base register once reached the load 5 wait states too early and
    faulted only in sequences like this one) and workgroups that own no strip at all (tiny launches)."""
    seq = [(6, 12, 1), (6, 12, 0), (6, 4, 1), (6, 4, 0), (9, 4, 1), (9, 4, 0), (30, 4, 1), (7, 4, 1), (60, 4, 1), (25, 0, 1)] * 3
    frames = {T: torch.from_numpy(synth.synth_frames(9001 + T, 1, T)).cuda() for T in {q[0] for q in seq}}
    first = {}
    try:
        for T, pad, zs in seq:
            engine.set_option("conv1_zero_skip", zs)
            out = engine.debug_conv1_pool(frames[T], pad)
            key = (T, pad)
            if key in first:
                assert torch.equal(out, first[key]), (T, pad, zs)
            else:
                first[key] = out.clone()
    finally:
        engine.set_option("conv1_zero_skip", 1)


# ------------------------------------------------------------------ (d) bias correction calibrated on the wrong data
def test_bias_correction_with_mismatched_calibration(option_case):
    """JG_PREC_FP16_BC folds (w - fp16(w)).E[x] into the bias with E[x] from calibration clips.  Calibrate on data that
    looks nothing like the test clips -- all-zero frames, and a smooth bright gradient without the face mask -- and
    on the test distribution; the test clips must stay within 1e-3 of the oracle in every case."""
    from jegal_amd._lib import Engine, PREC_FP16_BC
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    frames, ref = option_case
    e = Engine(0, precision=PREC_FP16_BC)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    errs = {"builtin": rel(e.extract_gesture(frames).cpu(), ref)}
    zeros = torch.zeros((2, 12, 270, 480, 3), dtype=torch.uint8, device="cuda")
    e.calibrate(zeros)
    errs["zeros"] = rel(e.extract_gesture(frames).cpu(), ref)
    yy, xx = torch.meshgrid(torch.arange(270), torch.arange(480), indexing="ij")
    grad = ((yy * 0.6 + xx * 0.3) % 256).to(torch.uint8)[None, None, :, :, None].expand(2, 12, 270, 480, 3).contiguous().cuda()
    e.calibrate(grad)
    errs["gradient"] = rel(e.extract_gesture(frames).cpu(), ref)
    big = torch.from_numpy(synth.synth_frames(99, 5, 12)).cuda()
    e.set_chunk(2)                                   # calibration batch larger than a chunk: means accumulate over chunks
    e.calibrate(big)
    errs["chunked"] = rel(e.extract_gesture(frames).cpu(), ref)
    e.set_chunk(32)
    e.calibrate(big)
    errs["unchunked"] = rel(e.extract_gesture(frames).cpu(), ref)
    e.close()
    print("bias-correction calibration robustness:", errs)
    assert all(v < TOL for v in errs.values()), errs
    assert abs(errs["chunked"] - errs["unchunked"]) < 2e-5


def test_two_handles_are_bit_identical_and_independent(engine, models, option_case):
    """Calibration is deterministic (no atomics) and options are per handle: a second engine in the same process gives
    the same bits, and its options do not leak into the first."""
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    frames, _ = option_case
    base = engine.extract_gesture(frames)
    e2 = Engine(0)
    GestSync(engine=e2).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e2).load_state_dict(synth.jegal_state_dict())
    e2.set_option("fuse_ln", 0)
    e2.set_option("attn_mfma", 0)
    other = e2.extract_gesture(frames)
    assert torch.equal(engine.extract_gesture(frames), base)          # engine 1 untouched by engine 2's options
    assert not torch.equal(other, base)
    e2.set_option("fuse_ln", 1)
    e2.set_option("attn_mfma", 1)
    assert torch.equal(e2.extract_gesture(frames), base)
    # re-loading a state_dict on a live engine (the drivers do) replaces the weights and leaves results unchanged
    GestSync(engine=e2).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e2).load_state_dict(synth.jegal_state_dict())
    assert torch.equal(e2.extract_gesture(frames), base)
    e2.close()


def test_argument_validation(engine, models):
    with pytest.raises(ValueError):
        engine.extract_gesture(torch.zeros((1, 4, 100, 100, 3), dtype=torch.uint8))          # wrong crop size
    with pytest.raises(ValueError):
        engine.extract_gesture(torch.zeros((4, 270, 480, 3), dtype=torch.uint8))              # not 5-D
    with pytest.raises(ValueError):
        engine.calibrate(torch.zeros((1, 4, 270, 480, 4), dtype=torch.uint8))
    with pytest.raises(ValueError):
        engine.extract_gesture(torch.zeros((1, 4, 270, 480, 3), dtype=torch.uint8), out=torch.empty(1, 4, 512))   # CPU out
    out = engine.extract_gesture(torch.zeros((1, 4, 270, 480, 3), dtype=torch.uint8))                          # CPU frames are moved
    assert out.shape == (1, 4, 512) and out.is_cuda


# ------------------------------------------------------------------ (c) ASD
def test_asd_golden_from_reference_evaluate_asd(engine, golden_dir):
    """tests/golden/asd.npz: 48 queries pushed through the reference's own evaluate_asd(df) (pkl files, csv rows):
    candidate lists of 1/3/5/6 clips, positives that lose -- per-query argmax for 2/4/6 speakers and the counts."""
    from jegal_amd import metrics as M
    g = np.load(os.path.join(golden_dir, "asd.npz"))
    n = int(g["n"])
    contents, positives, negatives = synth.planted_asd(int(g["seed"]), n)
    assert sorted({len(x) for x in negatives}) == [0, 2, 4, 5]
    q = M.video_level(engine, contents)                                         # temporal means (load_feats :26-39)
    cands = [M.video_level(engine, [positives[i]] + list(negatives[i])).cpu().numpy() for i in range(n)]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([c.shape[0] for c in cands])
    pred = engine.asd(q, torch.from_numpy(np.concatenate(cands, 0)), off).cpu().numpy()
    assert np.array_equal(pred, g["preds"])
    assert (pred != 0).any(axis=0).all() and (pred[:, 2] >= 4).any()            # the positive loses; late candidates win
    acc = M.asd_accuracy(q, cands, engine=engine)
    assert [round(a * n) for a in acc] == [int(x) for x in g["correct"]]
    ref_acc, ref_pred = O.asd_accuracy(contents, positives, negatives)
    assert np.array_equal(ref_pred, g["preds"]) and tuple(ref_acc) == tuple(acc)


# ------------------------------------------------------------------ (f) spotting limits
def test_spot_large_clips_and_limits(engine):
    """W > 64 words and T > 2048 frames used to overrun LDS silently; now W <= 1024 / T <= 8192 are computed and anything
    beyond is rejected (Python) or flagged pred = -1 / NaN (C ABI, device offsets)."""
    from jegal_amd import metrics as M
    gest, cont, bounds, targets = synth.planted_spotting(31, 3, n_frames=2500, n_words=100, noise=1.0)
    acc = M.spotting_accuracy(gest, cont, bounds, targets, engine=engine)
    assert acc == pytest.approx(O.spotting_accuracy(gest, cont, bounds, targets))
    g, c = np.concatenate(gest), np.concatenate(cont)
    pred, score = engine.spot(torch.from_numpy(g), torch.from_numpy(c), [0, 2500, 5000, 7500], [0, 100, 200, 300], targets)
    for i in range(3):
        a = O.attn_matrix(gest[i], cont[i])                                       # (W,T)
        assert int(pred[i]) == int(np.argmax(a[targets[i]]))
        assert float(score[i]) == pytest.approx(float(a[targets[i]].max()), rel=1e-4)
    with pytest.raises(ValueError):
        engine.spot(torch.zeros(9000, 512), torch.zeros(4, 512), [0, 9000], [0, 4], [0])
    with pytest.raises(ValueError):
        engine.spot(torch.zeros(10, 512), torch.zeros(4, 512), [0, 10], [0, 4], [4])
    # straight through the C ABI with an out-of-range clip: flagged, not computed
    import ctypes
    gd, cd = torch.randn(9000, 512, device="cuda"), torch.randn(4, 512, device="cuda")
    go, co, tg = (torch.tensor(v, dtype=torch.int32, device="cuda") for v in ([0, 9000], [0, 4], [1]))
    pr, sc = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert engine.lib.jg_spot(engine.h, P(gd), P(cd), P(go), P(co), P(tg), 1, 512, 0.07, P(pr), P(sc)) == 0
    torch.cuda.synchronize()
    assert int(pr[0]) == -1 and bool(torch.isnan(sc[0]))


# ------------------------------------------------------------------ (e) >= 8 clips of configs 2 and 3
def test_config2_eight_clips_vs_oracle(engine, models, oracle_sd):
    """BASELINE configs[1]: 8 full-length clips of the seed-1234 batch, computed inside the 32-clip batch, against
    the CPU oracle (window-de-duplicated form, exact)."""
    gsd, jsd = oracle_sd
    T = 150
    frames = synth.synth_frames(1234, 32, T)
    emb = engine.extract_gesture(torch.from_numpy(frames).cuda()).cpu()
    worst = 0.0
    with torch.no_grad():
        for b in (0, 3, 7, 12, 18, 23, 28, 31):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            ref = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
            r, mx = rel(emb[b], ref), float((emb[b] - ref).abs().max())
            worst = max(worst, r)
            assert r < TOL and mx < TOL, (b, r, mx)
    print(f"config 2, 8 clips: worst rel-L2 {worst:.3e}")


def test_config3_eight_clips_vs_oracle(engine, models, oracle_sd):
    """BASELINE configs[2] (batch 64, tri-modal): 8 clips (gesture + content embeddings) against the oracle."""
    gs, jg = models
    gsd, jsd = oracle_sd
    B, T, W = 64, 150, 10
    frames = torch.from_numpy(synth.synth_frames(1234, B, T)).cuda()
    feats = torch.cat([gs.extract_clip_feats(frames[i:i + 32]) for i in range(0, B, 32)])
    mel = synth.synth_mel(1235, B, 4 * T)
    states, tmask, ids, offs = synth.synth_text(1236, B, W)
    wbs = synth.synth_boundaries(B, W)
    tbatch = [[w[0] for w in wb] for wb in wbs]
    pack = (torch.from_numpy(states), torch.from_numpy(tmask), tbatch, ids, offs)
    g, c = jg.forward_inference(visual_feats=feats, visual_mask=torch.ones(B, T), text=pack, audio=torch.from_numpy(mel),
                                audio_mask=torch.ones(B, T), word_boundaries=wbs)
    gn, cn = engine.l2norm(g).cpu(), engine.l2norm(c).cpu()
    fr = frames.cpu().numpy()
    with torch.no_grad():
        for b in (1, 9, 17, 25, 33, 41, 52, 62):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(fr[b].astype(np.float32) / np.float32(255.0)))
            p1 = (torch.from_numpy(states[b:b + 1]), torch.from_numpy(tmask[b:b + 1]), tbatch[b:b + 1], ids[b:b + 1], offs[b:b + 1])
            rg, rc = O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T), text=p1,
                                               audio=torch.from_numpy(mel[b:b + 1]), audio_mask=None, word_boundaries=wbs[b:b + 1])
            eg, ec = rel(gn[b], O.l2_normalize(rg[0])), rel(cn[b], O.l2_normalize(rc[0]))
            assert eg < TOL and ec < TOL, (b, eg, ec)


# ------------------------------------------------------------------ multi-rank retrieval on the GPU
_WORKER = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
from jegal_amd import dist as jdist, synth, metrics as M
from jegal_amd._lib import Engine
backend = {backend!r}
ndev = torch.cuda.device_count()
local = int(os.environ["LOCAL_RANK"]) % ndev
torch.cuda.set_device(local)
jdist.init_from_env(backend)
r, w = jdist.rank(), jdist.world_size()
assert w == 2
eng = Engine(local)
N = 10000
g, c = synth.planted_retrieval(1237, N)
lo, hi = jdist.shard_range(N)
m = M.retrieval_metrics(torch.from_numpy(c[lo:hi]).cuda(), torch.from_numpy(g[lo:hi]).cuda(), engine=eng)
rk, ti = eng.sim_rank(eng.l2norm(torch.from_numpy(c)), eng.l2norm(torch.from_numpy(g)))
ref = M.metrics_from_ranks(rk.cpu().numpy(), ti.cpu().numpy())
assert m == ref, (m, ref)
assert 0.0 < m["R1"] < 1.0
# config 5: spotting sharded by clips (4000 clips, W = 30), two counters all-reduced -- equals the single-rank call
import torch.distributed as td
gest, cont, bounds, targets = synth.planted_spotting(1238, 4000)
lo, hi = jdist.shard_range(4000)
acc = M.spotting_accuracy(gest[lo:hi], cont[lo:hi], bounds[lo:hi], targets[lo:hi], engine=eng)
pred, score = eng.spot(torch.from_numpy(np.concatenate(gest)), torch.from_numpy(np.concatenate(cont)), M._offsets(gest), M._offsets(cont), targets)
c1, n1 = M.spotting_counts(pred.cpu().numpy(), score.cpu().numpy(), bounds, targets)
assert n1 == 4000 and acc == 100.0 * c1 / n1 and 10.0 < acc < 90.0, (acc, c1)
# ASD sharded by queries, four counters all-reduced
contents, positives, negatives = synth.planted_asd(11, 301)
q = np.stack([np.asarray(c, np.float32).mean(0) for c in contents])
cands = [np.stack([np.asarray(p_, np.float32).mean(0)] + [np.asarray(g_, np.float32).mean(0) for g_ in negs]) for p_, negs in zip(positives, negatives)]
lo, hi = jdist.shard_range(301)
a = M.asd_accuracy(q[lo:hi], cands[lo:hi], engine=eng)
pr = eng.asd(torch.from_numpy(q), torch.cat([torch.from_numpy(c) for c in cands]).cuda(), M._offsets(cands)).cpu().numpy()
c2, c4, c6, nq = M.asd_counts(pr)
assert nq == 301 and a == (c2 / nq, c4 / nq, c6 / nq), (a, c2, c4, c6)
jdist.barrier()
print("rank", r, "ok", m["R1"], m["MR"], acc, a)
"""


def _run_two(script):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append((p.returncode, out.decode()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return outs


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_sharded_retrieval_on_gpu(tmp_path, backend):
    """Config 4's exchange with jg_sim_rank on the device: 2 ranks, queries sharded ceil(N/2), gallery all-gathered,
    R@K / MR identical to the single-rank result; config 5's sharded spotting (4000 clips) and the sharded ASD evaluation with
    their counter all-reduce, identical to the single-rank calls.  nccl (RCCL over xGMI) needs 2 GPUs; with one GPU the same test
    runs over gloo with both ranks on device 0 (host-staged gather)."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, backend=backend))
    outs = _run_two(script)
    for rc, out in outs:
        assert rc == 0 and " ok" in out, out


def test_bench_launcher_starts_ranks(tmp_path):
    """`python bench.py --gpus 2` must itself start two fresh ranks and print ONE line with n_gpus == 2.  On a 1-GPU box
    the launcher is exercised with --oversubscribe (ranks share the GPU, gloo)."""
    import json
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "4",
            "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        args.append("--oversubscribe")
    r = subprocess.run(args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0
    assert res["retrieval_config4"]["equals_single_rank"] is True
    # asking for more GPUs than visible without the testing flag fails loudly instead of silently running one rank
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_c_abi_rccl_exchange_world_of_one():
    """jg_comm_* / jg_allgather / jg_allreduce_sum_i64 (round 6; SURVEY 8b lists jg_allgather): the exchange primitive of the retrieval
    metric for a consumer of the C ABI without PyTorch, on RCCL bound with dlopen.  One GPU here: a communicator of ONE rank (the collective
    code path, ids, streams and the gallery assembly are the same as with eight); with two visible devices a second handle joins."""
    from jegal_amd._lib import Engine
    eng = Engine(0)
    try:
        uid = Engine.comm_unique_id()
        assert isinstance(uid, bytes) and len(uid) == 128
        eng.comm_init(uid, 0, 1)
        ge, ce = synth.planted_retrieval(1237, 512)
        e1, e2 = eng.l2norm(torch.from_numpy(ce)), eng.l2norm(torch.from_numpy(ge))
        gallery = eng.allgather(e2)
        assert gallery.shape == e2.shape and torch.equal(gallery, e2)
        rank, ties = eng.sim_rank(e1, gallery)
        ref_rank, ref_ties = eng.sim_rank(e1, e2)
        assert torch.equal(rank, ref_rank) and torch.equal(ties, ref_ties)
        counts = eng.allreduce_sum_i64([int((rank < k).sum()) for k in (1, 5, 10, 25, 50)])
        assert counts.cpu().tolist() == [int((ref_rank < k).sum()) for k in (1, 5, 10, 25, 50)]
        eng.comm_destroy()
        with pytest.raises(Exception):
            eng.allgather(e2)                               # no communicator any more
    finally:
        eng.close()
