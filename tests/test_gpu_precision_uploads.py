"""Precision modes and the upload kernels (run with -m gpu): the bf16 reported mode next to the fp16 modes, the padded query rows of
ragged JEGAL batches per precision mode, the masked upload's unpack kernel (incl. bad metadata), the source-resolution upload
(jg_mask_resize_packed + GestureStreamer(source_hw=...)), conv1 launches with fewer strips than CUs, GEMM tile choice.
Everything goes through the C ABI."""
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
def oracle_sd():
    return O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())


def _engine(mode):
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    e = Engine(0, precision=mode)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    jg = JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    return e, jg


def test_precision_modes_report(oracle_sd):
    """north_star and BASELINE configs[2] name bf16.  JG_PREC_BF16 runs every GEMM / conv / attention MFMA of
    the path as a bf16 MFMA on bf16 weights and activations; its measured error on a full-length clip is REPORTED next to the
    fp16 modes.  The fp16 modes with a weight remedy hold the 1e-3 contract, bf16 does not (8 significant bits) -- that, with
    a GPU number, is why fp16 is the default.  The bf16 figure is asserted to be finite, of the expected order and worse than
    every fp16 mode, not to pass."""
    from jegal_amd._lib import PREC_FP16, PREC_FP16_W2, PREC_FP16_BC, PREC_BF16, PREC_FP16_RC
    gsd, jsd = oracle_sd
    T = 150
    frames = synth.synth_frames(1234, 1, T)
    with torch.no_grad():
        f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[0].astype(np.float32) / np.float32(255.0)))
        ref = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
    err, mx = {}, {}
    for name, mode in (("fp16", PREC_FP16), ("fp16_w2", PREC_FP16_W2), ("fp16_bc", PREC_FP16_BC), ("fp16_rc", PREC_FP16_RC), ("bf16", PREC_BF16)):
        e, _ = _engine(mode)
        out = e.extract_gesture(torch.from_numpy(frames).cuda())[0]
        assert torch.isfinite(out).all()
        err[name] = rel(out, ref)
        mx[name] = float((out.cpu() - ref).abs().max())
        e.close()
    print("precision modes, rel-L2 of the unit-norm gesture embedding vs the fp32 oracle (T = 150):",
          {k: f"{v:.2e}" for k, v in err.items()}, "max-abs:", {k: f"{v:.2e}" for k, v in mx.items()})
    assert err["fp16_w2"] < TOL and err["fp16_bc"] < TOL and err["fp16_rc"] < TOL
    assert err["fp16_rc"] < err["fp16"]                      # the run-time correction does what the calibrated one does
    assert max(err["fp16"], err["fp16_w2"], err["fp16_bc"], err["fp16_rc"]) < err["bf16"] < 5e-2


def test_bf16_content_path_is_reported(oracle_sd):
    """The tri-modal content path (audio CNN, text encoder, word pooling, fusion) in JG_PREC_BF16: finite, of bf16 order."""
    from jegal_amd._lib import PREC_BF16
    _, jsd = oracle_sd
    B, T, W = 2, 40, 6
    mel = synth.synth_mel(31, B, 4 * T)
    states, tmask, ids, offs = synth.synth_text(32, B, W)
    wbs = synth.synth_boundaries(B, W, stride=6, length=4)
    tbatch = [[w[0] for w in wb] for wb in wbs]
    pack = (torch.from_numpy(states), torch.from_numpy(tmask), tbatch, ids, offs)
    with torch.no_grad():
        ref = O.jegal_forward_inference(jsd, text=pack, audio=torch.from_numpy(mel), audio_mask=None, word_boundaries=wbs)
    e, jg = _engine(PREC_BF16)
    out = jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wbs)
    r = rel(e.l2norm(out), O.l2_normalize(ref))
    print(f"bf16 content embedding rel-L2 vs oracle: {r:.2e}")
    e.close()
    assert torch.isfinite(out).all() and r < 5e-2


def test_padded_query_rows_per_precision_mode(oracle_sd):
    """Ragged JEGAL batches (dataset.py:336-340): the reference also computes the zero-padded QUERY rows (callers strip them).
    Their first Linear sees x = 0 exactly, where a bias correction (w - fp16(w)).E[x] would be pure error (measured in round 2:
    1.8e-3 on those rows), so proj_ip_rgb keeps hi+lo weights in the default mode.  Valid AND padded rows hold 1e-3 in both modes."""
    from jegal_amd._lib import PREC_FP16_W2, PREC_FP16_BC
    _, jsd = oracle_sd
    T, valid = 150, 113
    rng = np.random.default_rng(850)
    vf = rng.standard_normal((2, T, 1024)).astype(np.float32)
    vf[1, valid:] = 0
    vm = np.ones((2, T), np.float32)
    vm[1, valid:] = 0
    with torch.no_grad():
        ref = O.jegal_forward_inference(jsd, visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm))
    res = {}
    for name, mode in (("fp16_w2", PREC_FP16_W2), ("fp16_bc", PREC_FP16_BC)):
        e, jg = _engine(mode)
        out = jg.forward_inference(visual_feats=torch.from_numpy(vf).cuda(), visual_mask=torch.from_numpy(vm).cuda())
        res[name] = (rel(out[1, :valid], ref[1, :valid]), rel(out[1, valid:], ref[1, valid:]))
        e.close()
    print("padded clip (valid rows, padded rows) rel-L2:", {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in res.items()})
    for a, _ in res.values():
        assert a < TOL
    assert res["fp16_w2"][1] < TOL and res["fp16_bc"][1] < TOL


def test_unpack_masked_rebuilds_the_dense_batch():
    from jegal_amd._lib import Engine
    eng = Engine.get("cuda:0")
    rng = np.random.default_rng(3)
    F = 7
    rows = np.array([0, 110, 270, 1, 269, 135, 64], np.int32)
    dense = rng.integers(1, 256, (F, 270, 480, 3), dtype=np.uint8)
    packed, offs, off = [], [], 0
    for f in range(F):
        dense[f, :rows[f]] = 0
        offs.append(off)
        packed.append(dense[f, rows[f]:].reshape(-1))
        off += packed[-1].size
    packed = np.concatenate(packed + [np.zeros(16, np.uint8)])
    dst = torch.full((F, 270, 480, 3), 7, dtype=torch.uint8, device="cuda")
    eng.unpack_masked(torch.from_numpy(packed).cuda(), torch.from_numpy(rows).cuda(), torch.from_numpy(np.array(offs, np.int64)).cuda(), dst)
    assert np.array_equal(dst.cpu().numpy(), dense)
    # bad metadata: row0 outside 0..270, a misaligned / negative offset, rows running past the end of `packed` -- such a
    # frame comes out all zero, nothing is read out of bounds; the good frames are untouched
    bad_rows, bad_offs = rows.copy(), np.array(offs, np.int64)
    bad_rows[1] = 300
    bad_rows[2] = -5
    bad_offs[3] = offs[3] + 4                     # not a multiple of 16
    bad_offs[5] = packed.size - 1440 * 10 - (packed.size - 1440 * 10) % 16          # 135 kept rows do not fit behind it
    bad_offs[6] = -16
    dst.fill_(7)
    eng.unpack_masked(torch.from_numpy(packed).cuda(), torch.from_numpy(bad_rows).cuda(), torch.from_numpy(bad_offs).cuda(), dst)
    got = dst.cpu().numpy()
    for f in (1, 2, 3, 5, 6):
        assert got[f].max() == 0, f
    for f in (0, 4):
        assert np.array_equal(got[f], dense[f]), f


@pytest.mark.parametrize("H,W", [(228, 314), (294, 294), (360, 640)])
def test_source_resolution_streamer_is_bit_identical(H, W):
    """GestureStreamer(source_hw=(H, W)) ships the decoder's frames (inference_embs.py:255-276 resizes 228x314 /
    294x294 crops up to 270x480 on the host) -- only the source rows below each frame's mask -- and jg_mask_resize_packed builds the
    crops on the device.  Embeddings must equal load_rgb_masked_frames' crops (jg_mask_resize of the full frames, itself
    bit-exact against the oracle's cv2 restatement) through the resident path, bit for bit; the packed kernel alone equals the
    unpacked one, incl. no-face frames (-1), a fully masked frame and mask_y beyond the frame."""
    from jegal_amd._lib import Engine
    from jegal_amd.extract import GestureStreamer, _SourcePacker
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    eng = Engine.get("cuda:0")
    GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
    rng = np.random.default_rng(H + W)
    T, n = 8, 5
    src = rng.integers(0, 256, (n, T, H, W, 3), dtype=np.uint8)
    my = rng.integers(H // 4, H // 2, (n, T)).astype(np.int64)
    my[0, 1], my[1, 2], my[2, 3], my[3, 0] = -1, H - 1, H + 40, 0
    # kernel alone
    pk = _SourcePacker(n, T, H, W, pinned=False)
    for b in range(n):
        pk.add(src[b], my[b])
    dst = torch.empty((n * T, 270, 480, 3), dtype=torch.uint8, device="cuda")
    eng.mask_resize_packed(pk.buf[:pk.used].cuda(), pk.offs.cuda(), pk.mask_y.cuda(), H, W, dst)
    ref_crops = torch.stack([eng.mask_resize(torch.from_numpy(src[b]), my[b].astype(np.int32)) for b in range(n)])
    assert torch.equal(dst.reshape(n, T, 270, 480, 3), ref_crops)
    np.testing.assert_array_equal(ref_crops[1].cpu().numpy(), O.mask_resize_frames(src[1], list(my[1])))
    assert pk.used < src.size * 0.8                                     # the masked rows did not cross the link
    # through the streamer (batches of 2, ragged last batch)
    ref = np.concatenate([eng.extract_gesture(ref_crops[i:i + 2]).cpu().numpy() for i in range(0, n, 2)])
    st = GestureStreamer(eng, batch=2, frames=T, source_hw=(H, W))
    got = list(st.run(iter(src), mask_rows=list(my)))
    assert [f for f, _ in got] == [0, 2, 4]
    np.testing.assert_array_equal(np.concatenate([e for _, e in got]), ref)
    # a frame whose rows would run past the packed buffer is written as zeros, not read
    bad = pk.offs.clone()
    bad[7] = pk.used - 10
    eng.mask_resize_packed(pk.buf[:pk.used].cuda(), bad.cuda(), pk.mask_y.cuda(), H, W, dst)
    assert int(dst[7].max()) == 0 and torch.equal(dst[6], ref_crops.reshape(-1, 270, 480, 3)[6])


@pytest.mark.parametrize("nclip,T", [(1, 8), (3, 8), (3, 10), (2, 8), (1, 5)])
def test_conv1_launches_with_fewer_strips_than_cus(nclip, T):
    """conv1_direct_kernel is a persistent kernel over nclip * (T + 4) * 5 column strips.  With fewer strips than CUs the strip
    ranges of the eight XCDs leave some workgroups without a strip (60, 180, 210 strips: the last XCD's range is shorter than
    its workgroup count).  Round 2 let those workgroups run through both roles with "harmless" frame loads; on unmasked clips
    that faulted (a wild frame address, found in round 3 through the masked-upload test).  They now exit at once; this test
    runs such launches on unmasked noise repeatedly and checks the result against the independent stack + implicit-GEMM path."""
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    eng = Engine.get("cuda:0")
    GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    rng = np.random.default_rng(nclip * 100 + T)
    frames = torch.from_numpy(rng.integers(1, 256, (nclip, T, 270, 480, 3), dtype=np.uint8)).cuda()
    outs = [eng.debug_conv1_pool(frames, 4).clone() for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    eng.set_option("conv1_direct", 0)
    try:
        alt = eng.debug_conv1_pool(frames, 4)
    finally:
        eng.set_option("conv1_direct", 1)
    assert rel(outs[0].float(), alt.float()) < 3e-4


def test_gemm_tile_choice_never_changes_a_bit():
    """launch_glds picks the plain GEMMs' tile (128x128 / 256x128 / 256x256) by a cost estimate; every instance accumulates k in
    the same order, so forcing any of them (option gemm_tile) must reproduce the default bit for bit - on the gesture path (M-partial
    last tiles: 3 x 60 x 21 tokens, 180 JEGAL tokens) and on the XLM-R front end."""
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    from jegal_amd.xlmr import XLMRoberta
    eng = Engine(0)
    GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
    xl = XLMRoberta(engine=eng).load_state_dict(synth.xlmr_state_dict(layers=2))
    frames = torch.from_numpy(synth.synth_frames(31, 3, 60)).cuda()
    ids, mask = synth.xlmr_inputs(3, 24, 40)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    base = eng.extract_gesture(frames).clone()
    xbase = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
    for tile in (1, 2, 3):
        eng.set_option("gemm_tile", tile)
        try:
            assert torch.equal(eng.extract_gesture(frames), base), f"gemm_tile={tile} changed the gesture embeddings"
            assert torch.equal(xl(ids_d, attention_mask=mask_d).last_hidden_state, xbase), f"gemm_tile={tile} changed the XLM-R states"
        finally:
            eng.set_option("gemm_tile", 0)
    eng.close()
