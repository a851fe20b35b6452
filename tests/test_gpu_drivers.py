"""End-to-end through the command-line drivers (the counterparts of extract_gestsync_feats.py,
extract_jegal_embs.py and evaluate_*.py): crops -> .npy feats -> .pkl -> metrics, checked against the
oracle on the same files."""
import os
import pickle

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import drivers, synth

pytestmark = pytest.mark.gpu


def test_drivers_roundtrip(tmp_path):
    import pandas as pd
    frames_dir, feat_dir, video_dir, res_dir = (str(tmp_path / d) for d in ("frames", "feats", "videos", "res"))
    n, T, W = 5, 30, 4
    rows = []
    rng = np.random.default_rng(0)
    for i in range(n):
        vid, track = f"vid{i:02d}_1.0-2.0", "00000"
        os.makedirs(os.path.join(frames_dir, vid), exist_ok=True)
        os.makedirs(os.path.join(video_dir, vid), exist_ok=True)
        Ti = T + 2 * i
        np.save(os.path.join(frames_dir, vid, track + ".npy"), synth.synth_frames(100 + i, 1, Ti)[0])
        np.save(os.path.join(video_dir, vid, track + ".mel.npy"), synth.synth_mel(200 + i, 1, 4 * Ti)[0])
        st, mk, ids, offs = synth.synth_text(300 + i, 1, W)
        np.savez(os.path.join(video_dir, f"{vid}__{track}.npz"), states=st[0], mask=mk[0], ids=ids[0], offsets=offs[0])
        wb = [[f"w{j}", 6 * j, 6 * j + 4] for j in range(W)]
        rows.append({"video_id": vid, "filename": f"{vid}/{track}", "phrase": " ".join(w[0] for w in wb),
                     "word_boundaries": str(wb), "target_word_boundary": str(wb[i % W])})
    csv = str(tmp_path / "avs.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)

    assert drivers.main(["extract_gestsync_feats", "--checkpoint_path_gestsync", "synthetic", "--frames_dir", frames_dir,
                         "--result_dir", feat_dir]) == 0
    f0 = np.load(os.path.join(feat_dir, rows[0]["video_id"], "00000.npy"))
    assert f0.shape == (T, 1024)
    # resume: second run finds every output and recomputes nothing
    mt = os.path.getmtime(os.path.join(feat_dir, rows[0]["video_id"], "00000.npy"))
    drivers.main(["extract_gestsync_feats", "--checkpoint_path_gestsync", "synthetic", "--frames_dir", frames_dir, "--result_dir", feat_dir])
    assert os.path.getmtime(os.path.join(feat_dir, rows[0]["video_id"], "00000.npy")) == mt

    assert drivers.main(["extract_jegal_embs", "--file_path", csv, "--checkpoint_path", "synthetic", "--res_dir", res_dir,
                         "--video_dir", video_dir, "--feature_dir", feat_dir, "--text_states_dir", video_dir,
                         "--modalities", "vta", "--batch_size", "3"]) == 0
    pk = os.path.join(res_dir, "vta")
    files = sorted(os.listdir(pk))
    assert files == sorted(r["video_id"] + "__00000.pkl" for r in rows)
    feats = [pickle.load(open(os.path.join(pk, f), "rb")) for f in files]
    for i, f in enumerate(feats):
        assert f["gesture_emb"].shape == (T + 2 * i, 512) and f["content_emb"].shape == (W, 512)
        assert np.allclose(np.linalg.norm(f["gesture_emb"], axis=1), 1, atol=1e-5)
        assert f["info"]["phrase"] == rows[i]["phrase"]                 # pandas Series row, as the reference stores

    # pkl contents vs the oracle for one clip (padded batch of 3 vs the clip alone on the CPU)
    jsd = O.tensors(synth.jegal_state_dict())
    i = 1
    vis = torch.from_numpy(np.load(os.path.join(feat_dir, rows[i]["video_id"], "00000.npy")))[None]
    mel = torch.from_numpy(np.load(os.path.join(video_dir, rows[i]["video_id"], "00000.mel.npy")))[None]
    z = np.load(os.path.join(video_dir, rows[i]["video_id"] + "__00000.npz"))
    pack = (z["states"][None], z["mask"][None], [rows[i]["phrase"].split(" ")], z["ids"][None], z["offsets"][None])
    with torch.no_grad():
        g, c = O.jegal_forward_inference(jsd, visual_feats=vis, visual_mask=torch.ones(1, vis.shape[1]), text=pack, audio=mel,
                                         audio_mask=None, word_boundaries=[eval(rows[i]["word_boundaries"])])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(feats[i]["gesture_emb"], O.l2_normalize(g[0]).numpy()) < 1e-3
    assert rel(feats[i]["content_emb"], O.l2_normalize(c[0]).numpy()) < 1e-3

    # evaluators on the written pkls == the reference's metric definitions on the same arrays
    res = drivers.main(["evaluate_retrieval", "--path", pk])
    res = drivers.cmd_evaluate_retrieval(["--path", pk])
    gv = [f["gesture_emb"].mean(axis=0) for f in feats]
    cv = [f["content_emb"].mean(axis=0) for f in feats]
    assert res["Content to Gesture"] == O.compute_metrics(O.similarity_matrix(cv, gv).numpy())
    assert res["Gesture to Content"] == O.compute_metrics(O.similarity_matrix(gv, cv).numpy())
    acc = drivers.cmd_evaluate_spotting(["--path", pk])
    wbs = [eval(r["word_boundaries"]) for r in rows]
    tg = [wb.index(eval(r["target_word_boundary"])) for wb, r in zip(wbs, rows)]
    assert acc == pytest.approx(O.spotting_accuracy([f["gesture_emb"] for f in feats], [f["content_emb"] for f in feats], wbs, tg))


def test_gesture_streamer_matches_resident_path():
    """Pinned double-buffered streaming (ragged last batch, both producer interfaces) returns exactly what
    the HBM-resident call returns, in order."""
    from jegal_amd._lib import Engine
    from jegal_amd.extract import GestureStreamer
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    eng = Engine(0)
    GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
    T, n = 8, 7
    clips = synth.synth_frames(77, n, T)
    # same batching as the streamer: the GEMM tile configuration (and with it the fp32 summation order) depends on M
    ref = np.concatenate([eng.extract_gesture(torch.from_numpy(clips[i:i + 3]).cuda()).cpu().numpy() for i in range(0, n, 3)])
    st = GestureStreamer(eng, batch=3, frames=T)
    got = list(st.run(iter(clips)))
    assert [f for f, _ in got] == [0, 3, 6] and [e.shape[0] for _, e in got] == [3, 3, 1]
    np.testing.assert_array_equal(np.concatenate([e for _, e in got]), ref)

    def fill(buf, k):                       # a producer writing straight into the pinned buffer
        lo, hi = 3 * k, min(n, 3 * k + 3)
        if lo >= n:
            return 0
        buf[:hi - lo] = clips[lo:hi]
        return hi - lo
    got2 = list(st.run_filled(fill))
    np.testing.assert_array_equal(np.concatenate([e for _, e in got2]), ref)
    with pytest.raises(ValueError):
        list(st.run([clips[0][:4]]))
    # masked upload: only the rows below each frame's mask cross the link, jg_unpack_masked rebuilds the batch on the device
    stm = GestureStreamer(eng, batch=3, frames=T, masked=True)
    got3 = list(stm.run(iter(clips), mask_rows=[110] * n))
    assert [f for f, _ in got3] == [0, 3, 6]
    np.testing.assert_array_equal(np.concatenate([e for _, e in got3]), ref)
    from jegal_amd.extract import _MaskedPacker
    pk = _MaskedPacker(3, T)
    for b in range(3):
        pk.add(clips[b], 110)
    assert pk.n == 3 and pk.used == 3 * T * 160 * 480 * 3            # 160 of 270 rows cross the link
    # per-frame mask heights (and a frame shipped whole, one not at all)
    rng = np.random.default_rng(5)
    rows = rng.integers(60, 150, (n, T))
    rows[0, 0], rows[1, 2] = 0, 270
    jit = rng.integers(1, 256, (n, T, 270, 480, 3), dtype=np.uint8)
    for b in range(n):
        for t in range(T):
            jit[b, t, :rows[b, t]] = 0
    refj = np.concatenate([eng.extract_gesture(torch.from_numpy(jit[i:i + 3]).cuda()).cpu().numpy() for i in range(0, n, 3)])
    got4 = list(stm.run(iter(jit), mask_rows=list(rows)))
    np.testing.assert_array_equal(np.concatenate([e for _, e in got4]), refj)


@pytest.mark.parametrize("H,W", [(270, 480), (360, 640), (720, 1280), (301, 533), (135, 240), (228, 314), (294, 294)])
def test_mask_resize_matches_oracle(H, W):
    """face-mask + cv2-style bilinear resize kernel == the numpy restatement, bit for bit
    (up- and down-scaling, identity size, odd sizes; face / no-face / mask-beyond-frame rows)."""
    from jegal_amd._lib import Engine
    from jegal_amd.extract import load_rgb_masked_frames, FACE_OVAL_IDX
    eng = Engine(0)
    rng = np.random.default_rng(H * 1000 + W)
    T = 4
    frames = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    mask_y = [-1, H // 3, 0, H + 50]
    got = eng.mask_resize(torch.from_numpy(frames), mask_y).cpu().numpy()
    ref = O.mask_resize_frames(frames, mask_y)
    np.testing.assert_array_equal(got, ref)
    assert got[0, :111].max() == 0 and got[3].max() == 0
    # through the reference-shaped entry point: landmarks -> y2 + 15
    face = [{"x": 0.5, "y": 0.0} for _ in range(468)]
    face[FACE_OVAL_IDX[3]] = {"x": 0.5, "y": 0.25}
    kp = {"kps": [{"face": None}, {"face": face}, {"face": face}, {"face": None}], "resolution": (H, W)}
    got2 = load_rgb_masked_frames(eng, frames, kp).cpu().numpy()
    my = int(0.25 * H) + 15
    np.testing.assert_array_equal(got2, O.mask_resize_frames(frames, [-1, my, my, -1]))


def test_ragged_batches_by_last_frame_padding_are_exact():
    """The feature-extraction driver batches clips of different lengths by padding each with copies of its last frame
    (inference_embs.py:283 edge-pads with the last frame, and window t only reaches frame t+12): the first T_i feature rows of a
    padded clip are those of the clip itself.  Bit for bit between two paddings of different length (same kernel path, nothing of
    the padding may leak into the kept rows); against the clip run alone within the path-to-path noise (a single short clip takes
    the unfused fp32-stream transformer, the batch the fused one: both are within 1e-3 of the fp32 reference, 4.5e-4 of each other)."""
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    eng = Engine.get("cuda:0")
    gs = GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    lens = [17, 30, 26]
    clips = [synth.synth_frames(40 + i, 1, t)[0] for i, t in enumerate(lens)]

    def padded(Tpad):
        batch = np.empty((3, Tpad, 270, 480, 3), np.uint8)
        for i, c in enumerate(clips):
            batch[i, :lens[i]] = c
            batch[i, lens[i]:] = c[-1]
        # (lengths: the run-time corrected default mode takes each clip's statistics from its own frames, as the driver does)
        return gs.extract_clip_feats(torch.from_numpy(batch).cuda(), lengths=lens).cpu()

    out, out_longer = padded(max(lens)), padded(max(lens) + 7)
    for i, c in enumerate(clips):
        assert torch.equal(out[i, :lens[i]], out_longer[i, :lens[i]]), f"clip {i}: the padding leaks into the kept rows"
        solo = gs.extract_clip_feats(torch.from_numpy(c).cuda())[0].cpu()
        r = float((out[i, :lens[i]] - solo).norm() / solo.norm())
        assert r < 1e-3, (i, r)


def _ragged_dataset(tmp_path, lens, words, seed0=500):
    """A ragged AVS-like dataset on disk in the drivers' formats; returns (csv, dirs, clips)."""
    import pandas as pd
    feat_dir, video_dir, res_dir = (str(tmp_path / d) for d in ("feats", "videos", "res"))
    rows, clips = [], []
    for i, (T, W) in enumerate(zip(lens, words)):
        vid, track = f"vid{i:02d}_1.0-2.0", "00000"
        c = synth.synth_ragged_clip(seed0 + i, T, W)
        clips.append(c)
        os.makedirs(os.path.join(feat_dir, vid), exist_ok=True)
        os.makedirs(os.path.join(video_dir, vid), exist_ok=True)
        np.save(os.path.join(feat_dir, vid, track + ".npy"), c["feats"])
        np.save(os.path.join(video_dir, vid, track + ".mel.npy"), c["mel"])
        np.savez(os.path.join(video_dir, f"{vid}__{track}.npz"), states=c["states"], mask=c["mask"], ids=c["ids"], offsets=c["offsets"])
        wb = c["word_boundaries"]
        rows.append({"video_id": vid, "filename": f"{vid}/{track}", "phrase": c["phrase"], "word_boundaries": str(wb),
                     "target_word_boundary": str(wb[(3 * i + 1) % W])})
    csv = str(tmp_path / "avs.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    return csv, (feat_dir, video_dir, res_dir), clips, rows


def _oracle_alone(jsd, c, mod):
    """The reference pipeline on ONE clip alone in its batch (extract_jegal_embs.py:141 runs batch_size=1)."""
    vis = torch.from_numpy(c["feats"])[None] if "v" in mod else None
    pack = (c["states"][None], c["mask"][None], [c["phrase"].split(" ")], c["ids"][None], c["offsets"][None]) if "t" in mod else None
    mel = torch.from_numpy(c["mel"])[None] if "a" in mod else None
    with torch.no_grad():
        out = O.jegal_forward_inference(jsd, visual_feats=vis, visual_mask=None if vis is None else torch.ones(1, vis.shape[1]), text=pack,
                                        audio=mel, audio_mask=None, word_boundaries=[c["word_boundaries"]])
    g = cemb = None
    if vis is not None and (pack is not None or mel is not None):
        g, cemb = out
    elif vis is not None:
        g = out
    else:
        cemb = out
    return (None if g is None else O.l2_normalize(g[0]).numpy(), None if cemb is None else O.l2_normalize(cemb[0]).numpy())


@pytest.mark.parametrize("mod", ["vta", "ta", "a", "t", "vt", "va", "v"])
def test_extract_jegal_embs_is_batch_invariant_on_ragged_clips(tmp_path, mod):
    """the reference's dataset driver runs ONE clip per step (evaluation/extract_jegal_embs.py:141); this
    driver batches 16.  Sixteen ragged clips (T 25..220, W 3..12, every last word multi-sub-word and ending on the clip's last
    frame) in one padded batch: every .pkl row must be what the oracle gives for that clip ALONE (< 1e-3), the last content row
    included -- it is the one that the padded text length (jegal.py:168-171) and the zero-padded audio conv stack (jegal.py:41-63)
    used to change by 0.37 / 3.5e-2 -- and the retrieval / spotting numbers of the written .pkl files equal those of the
    per-clip oracle pipeline."""
    rng = np.random.default_rng(11)
    lens = [25, 220, 150, 31, 77, 26, 199, 64, 120, 48, 33, 181, 90, 55, 140, 29]
    words = [3, 12, 10, 4, 7, 3, 11, 6, 9, 5, 4, 12, 8, 5, 10, 3]
    csv, (feat_dir, video_dir, res_dir), clips, rows = _ragged_dataset(tmp_path, lens, words)
    assert drivers.main(["extract_jegal_embs", "--file_path", csv, "--checkpoint_path", "synthetic", "--res_dir", res_dir,
                         "--video_dir", video_dir, "--feature_dir", feat_dir, "--text_states_dir", video_dir,
                         "--modalities", mod, "--batch_size", "16"]) == 0
    pk = os.path.join(res_dir, mod)
    files = sorted(os.listdir(pk))
    assert len(files) == 16
    feats = [pickle.load(open(os.path.join(pk, f), "rb")) for f in files]
    jsd = O.tensors(synth.jegal_state_dict())
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    worst_g = worst_c = worst_last = 0.0
    ref = []
    for c, f, T, W in zip(clips, feats, lens, words):
        g, ce = _oracle_alone(jsd, c, mod)
        ref.append((g, ce))
        if g is not None:
            assert f["gesture_emb"].shape == (T, 512)
            worst_g = max(worst_g, rel(f["gesture_emb"], g))
        else:
            assert f["gesture_emb"] is None
        if ce is None:                                   # --modalities v: gesture only
            assert f["content_emb"] is None
            continue
        assert f["content_emb"].shape == (W, 512)
        worst_c = max(worst_c, rel(f["content_emb"], ce))
        worst_last = max(worst_last, float(np.abs(f["content_emb"][-1] - ce[-1]).max()))
    print(f"\n[{mod}] 16 ragged clips in one batch vs the oracle on each clip alone: gesture {worst_g:.2e}, content {worst_c:.2e}, "
          f"last word max-abs {worst_last:.2e}")
    assert worst_g < 1e-3 and worst_c < 1e-3 and worst_last < 1e-3
    if mod != "vta":
        return
    # metrics of the written pkls == the per-clip oracle pipeline's
    res = drivers.cmd_evaluate_retrieval(["--path", pk])
    gv = [g.mean(axis=0) for g, _ in ref]
    cv = [c.mean(axis=0) for _, c in ref]
    # Sixteen clips of random features have no retrieval structure: the rank of the diagonal is decided by similarity gaps of the
    # order of the embedding tolerance itself.  The driver's numbers must be those of the metric on the WRITTEN embeddings (exact),
    # and any rank that differs from the per-clip oracle pipeline's must sit on a near-tie of the oracle's similarities (< 2e-3).
    gw = [f["gesture_emb"].mean(axis=0) for f in feats]
    cw = [f["content_emb"].mean(axis=0) for f in feats]
    for key, (a_ref, b_ref), (a_got, b_got) in (("Content to Gesture", (cv, gv), (cw, gw)), ("Gesture to Content", (gv, cv), (gw, cw))):
        s_ref, s_got = O.similarity_matrix(a_ref, b_ref).numpy(), O.similarity_matrix(a_got, b_got).numpy()
        assert res[key] == O.compute_metrics(s_got), key
        d_ref, d_got = s_ref - np.diag(s_ref)[:, None], s_got - np.diag(s_got)[:, None]
        flips = (d_ref > 0) != (d_got > 0)
        np.fill_diagonal(flips, False)
        assert np.abs(d_ref[flips]).max(initial=0.0) < 2e-3, (key, int(flips.sum()), np.abs(d_ref[flips]).max(initial=0.0))
        if not flips.any():
            assert res[key] == O.compute_metrics(s_ref), key
    acc = drivers.cmd_evaluate_spotting(["--path", pk])
    wbs = [c["word_boundaries"] for c in clips]
    tg = [wb.index(eval(r["target_word_boundary"])) for wb, r in zip(wbs, rows)]
    assert acc == pytest.approx(O.spotting_accuracy([g for g, _ in ref], [c for _, c in ref], wbs, tg))
    # and the batch size does not matter: the same files from batches of 1 (the reference's own setting) and of 5
    for bs in ("1", "5"):
        rd = res_dir + "_bs" + bs
        assert drivers.main(["extract_jegal_embs", "--file_path", csv, "--checkpoint_path", "synthetic", "--res_dir", rd,
                             "--video_dir", video_dir, "--feature_dir", feat_dir, "--text_states_dir", video_dir,
                             "--modalities", mod, "--batch_size", bs]) == 0
        for fn, f16, (g, ce) in zip(files, feats, ref):
            f1 = pickle.load(open(os.path.join(rd, mod, fn), "rb"))
            # (the content path takes the same kernels at every batch size; a 25-frame clip ALONE takes the register-staged GEMM for
            # its gesture rows, M < 128, whose fp32 summation order differs: path-to-path noise, both within 1e-3 of the oracle)
            assert rel(f1["content_emb"], f16["content_emb"]) < 2e-4, (bs, fn)
            assert rel(f1["content_emb"], ce) < 1e-3 and rel(f1["gesture_emb"], g) < 1e-3, (bs, fn)


def test_forward_inference_default_keeps_the_reference_semantics_at_b_gt_1():
    """The facade (not the driver) mirrors models/jegal.py at B > 1, padded-length dependence included: the default call on a
    padded batch equals the oracle on the SAME padded batch; per_clip=True equals the oracle on each clip alone."""
    from jegal_amd._lib import Engine
    from jegal_amd.jegal import JEGAL
    eng = Engine.get("cuda:0")
    jg = JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
    jsd = O.tensors(synth.jegal_state_dict())
    a, b = synth.synth_ragged_clip(900, 30, 3), synth.synth_ragged_clip(901, 34, 6)
    L = max(len(a["ids"]), len(b["ids"]))
    st = np.zeros((2, L, 768), np.float32); tm = np.zeros((2, L), np.int64); ids = np.ones((2, L), np.int64); offs = np.zeros((2, L, 2), np.int64)
    for i, c in enumerate((a, b)):
        l = len(c["ids"])
        st[i, :l], tm[i, :l], ids[i, :l], offs[i, :l] = c["states"], c["mask"], c["ids"], c["offsets"]
    mel = np.zeros((2, 4 * 34, 80), np.float32)
    mel[0, :120], mel[1] = a["mel"], b["mel"]
    tb = [a["phrase"].split(" "), b["phrase"].split(" ")]
    wbs = [a["word_boundaries"], b["word_boundaries"]]
    pack = (torch.from_numpy(st), torch.from_numpy(tm), tb, ids, offs)
    with torch.no_grad():
        ref_batch = O.jegal_forward_inference(jsd, text=(st, tm, tb, ids, offs), audio=torch.from_numpy(mel), word_boundaries=wbs)
    got = jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wbs).cpu()
    rel = lambda x, y: float((x - y).norm() / y.norm())
    assert rel(got[0, :3], ref_batch[0, :3]) < 1e-3 and rel(got[1], ref_batch[1]) < 1e-3
    got_pc = jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wbs, audio_lens=[120, 136], per_clip=True).cpu()
    _, alone = _oracle_alone(jsd, a, "ta")
    with torch.no_grad():
        alone_raw = O.jegal_forward_inference(jsd, text=(a["states"][None], a["mask"][None], [tb[0]], a["ids"][None], a["offsets"][None]),
                                              audio=torch.from_numpy(a["mel"])[None], word_boundaries=[wbs[0]])[0]
    assert rel(got_pc[0, :3], alone_raw) < 1e-3
    # the two semantics really differ on the last word of the shorter clip (else this test proves nothing)
    assert float((ref_batch[0, 2] - alone_raw[2]).abs().max()) > 1e-2
    with pytest.raises(ValueError):
        jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wbs, per_clip=True)       # audio_lens missing


def test_extract_jegal_embs_reads_wav_like_the_reference(tmp_path):
    """The reference's dataset reads <video_dir>/<file>.wav and computes the log-mel itself (dataset.py:229-235,279-298).  The driver
    does the same through the GPU front end (jg_logmel) when no precomputed .mel.npy exists: the reference's own samples/sample1.wav
    (data fixture, 34691 samples -> 216 mel frames -> 54 audio steps) in an `a`-only run equals the oracle chain wav -> mel (torch.stft)
    -> forward_audio -> word pooling -> fusion; an unreadable wav drops the sample as the reference does."""
    import shutil
    import pandas as pd
    from jegal_amd import audio
    video_dir, res_dir = str(tmp_path / "videos"), str(tmp_path / "res")
    os.makedirs(os.path.join(video_dir, "vidA_0-2"))
    os.makedirs(os.path.join(video_dir, "vidB_0-2"))
    wav_src = os.path.join(os.path.dirname(__file__), "golden", "sample1.wav")
    shutil.copy(wav_src, os.path.join(video_dir, "vidA_0-2", "00000.wav"))
    with open(os.path.join(video_dir, "vidB_0-2", "00000.wav"), "wb") as f:
        f.write(b"not a wav file")
    wb = [["amount", 1, 6], ["of", 7, 9], ["water", 10, 20], ["in", 21, 23], ["the", 40, 53]]
    rows = [{"video_id": v, "filename": f"{v}/00000", "phrase": " ".join(w[0] for w in wb), "word_boundaries": str(wb), "target_word_boundary": str(wb[0])}
            for v in ("vidA_0-2", "vidB_0-2")]
    csv = str(tmp_path / "avs.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    assert drivers.main(["extract_jegal_embs", "--file_path", csv, "--checkpoint_path", "synthetic", "--res_dir", res_dir, "--video_dir", video_dir,
                         "--feature_dir", video_dir, "--modalities", "a"]) == 0
    files = sorted(os.listdir(os.path.join(res_dir, "a")))
    assert files == ["vidA_0-2__00000.pkl"]                              # the broken wav was dropped
    got = pickle.load(open(os.path.join(res_dir, "a", files[0]), "rb"))
    assert got["gesture_emb"] is None and got["content_emb"].shape == (5, 512)
    wav = audio.load_wav(wav_src).astype(np.float32)
    mel = O.wav2filterbanks(wav[None], torch.from_numpy(audio.mel_filterbank()))
    assert mel.shape == (1, 216, 80)
    jsd = O.tensors(synth.jegal_state_dict())
    with torch.no_grad():
        ref = O.l2_normalize(O.jegal_forward_inference(jsd, audio=mel, word_boundaries=[wb])[0]).numpy()
    r = float(np.linalg.norm(got["content_emb"] - ref) / np.linalg.norm(ref))
    print(f"\nwav -> content embedding through the driver vs the oracle chain: rel-L2 {r:.2e}")
    assert r < 1e-3


def test_inference_embs_command_config0_one_call(tmp_path, monkeypatch):
    """BASELINE configs[0] in ONE call: `inference_embs --modalities vta` on the reference's own
    samples/sample1.wav + samples/sample1.txt (data fixtures: tests/golden/sample1.wav, load_text.json) and a seeded 56-frame
    228 x 314 source clip (the sample .avi cannot be decoded here; frames + per-frame chin rows stand in for decord + mediapipe).
    The .pkl must be what the oracle chain gives: mask + cv2-style resize -> /255 -> GestSync windows -> JEGAL gesture encoder;
    wav -> log-mel -> audio CNN; text -> XLM-RoBERTa -> text encoder; word pooling, fusion; L2-normalised; info = {fname,
    word_boundaries, text} (inference_embs.py:526-646).  Then the six other --modalities, which crash in the reference."""
    import json
    from jegal_amd import audio
    from test_gpu_xlmr import StubTokenizer
    monkeypatch.setattr(drivers, "_load_tokenizer", lambda name: StubTokenizer())
    gold = os.path.join(os.path.dirname(__file__), "golden")
    txt = tmp_path / "sample1.txt"
    txt.write_text(json.load(open(os.path.join(gold, "load_text.json")))["sample1"]["file"], encoding="utf-8")
    rng = np.random.default_rng(4242)
    T, H, W = 56, 228, 314
    src = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    mask_y = rng.integers(70, 110, T).astype(np.int32)
    mask_y[7] = -1                                                        # a frame without a detected face (inference_embs.py:272-276)
    np.save(tmp_path / "sample1.npy", src)
    np.save(tmp_path / "mask_y.npy", mask_y)
    res = str(tmp_path / "res")
    common = ["--checkpoint_path_gestsync", "synthetic", "--checkpoint_path_jegal", "synthetic", "--res_dir", res,
              "--video_path", str(tmp_path / "sample1.npy"), "--mask_y", str(tmp_path / "mask_y.npy"),
              "--audio_path", os.path.join(gold, "sample1.wav"), "--text_path", str(txt), "--xlmr_checkpoint", "synthetic", "--tokenizer", "stub"]
    assert drivers.main(["inference_embs", "--modalities", "vta"] + common) == 0
    got = pickle.load(open(os.path.join(res, "sample1.pkl"), "rb"))
    ref_text = json.load(open(os.path.join(gold, "load_text.json")))["sample1"]
    assert got["info"] == {"fname": "sample1", "word_boundaries": ref_text["word_boundaries"][0], "text": ref_text["text"][0]}

    # the oracle chain on the same inputs
    gsd, jsd, xsd = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict()), synth.xlmr_state_dict()
    wbs = ref_text["word_boundaries"]
    with torch.no_grad():
        crops = O.mask_resize_frames(src, mask_y)
        feats = O.gestsync_clip_feats(gsd, torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)))
        wav = audio.load_wav(os.path.join(gold, "sample1.wav")).astype("float32")
        mel = O.wav2filterbanks(wav[None], torch.from_numpy(audio.mel_filterbank()))
        enc = StubTokenizer()([ref_text["text"][0].split(" ")])
        states = O.xlmr_forward(xsd, enc["input_ids"].numpy(), enc["attention_mask"].numpy())
        pack = (states, enc["attention_mask"], [ref_text["text"][0].split(" ")], enc["input_ids"], enc["offset_mapping"])
        g, c = O.jegal_forward_inference(jsd, visual_feats=feats[None], visual_mask=torch.ones(1, T), text=pack, audio=mel,
                                         audio_mask=torch.ones(1, mel.shape[1] // 4), word_boundaries=wbs)
        g, c = O.l2_normalize(g[0]).numpy(), O.l2_normalize(c[0]).numpy()
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert got["gesture_emb"].shape == (T, 512) and got["content_emb"].shape == (len(wbs[0]), 512)
    rg, rc = rel(got["gesture_emb"], g), rel(got["content_emb"], c)
    print(f"\ninference_embs vta on sample1.wav / sample1.txt + 56 source frames vs the oracle chain: gesture {rg:.2e}, content {rc:.2e}")
    assert rg < 1e-3 and rc < 1e-3 and np.abs(got["gesture_emb"] - g).max() < 1e-3 and np.abs(got["content_emb"] - c).max() < 1e-3
    # the other six modality sets: outputs per forward_inference's own convention (jegal.py:377-415)
    for mod in ("vt", "va", "ta", "v", "t", "a"):
        res_m = str(tmp_path / ("res_" + mod))
        args = ["inference_embs", "--modalities", mod] + common
        args[args.index("--res_dir") + 1] = res_m
        assert drivers.main(args) == 0
        d = pickle.load(open(os.path.join(res_m, "sample1.pkl"), "rb"))
        assert (d["gesture_emb"] is not None) == ("v" in mod) and (d["content_emb"] is not None) == (mod != "v")
        if "v" in mod:
            assert rel(d["gesture_emb"], g) < 1e-3
        if mod in ("vt", "va", "ta", "t", "a"):
            with torch.no_grad():
                cm = O.jegal_forward_inference(jsd, text=pack if "t" in mod else None, audio=mel if "a" in mod else None,
                                               audio_mask=torch.ones(1, mel.shape[1] // 4) if "a" in mod else None, word_boundaries=wbs)
            assert rel(d["content_emb"], O.l2_normalize(cm[0]).numpy()) < 1e-3, mod
    # argument checks of inference_embs.py:650-664
    with pytest.raises(ValueError):
        drivers.main(["inference_embs", "--modalities", "va", "--checkpoint_path_gestsync", "synthetic", "--checkpoint_path_jegal", "synthetic",
                      "--res_dir", res, "--audio_path", os.path.join(gold, "sample1.wav"), "--text_path", str(txt)])


def test_inference_embs_audit_flag_reports_the_error_of_the_default_mode(tmp_path, monkeypatch, capsys):
    """`inference_embs ... --audit`: the clip runs a second time on a JG_PREC_FP32 engine with the same checkpoints and the
    driver prints rel-L2 / max-abs of the embeddings it saved against that run -- the 1e-3 contract checked on the caller's own clip
    where no CPU reference is at hand.  Here the reference IS at hand (the oracle chain): the audit engine must agree with it to
    2e-5, so the printed figures are the default mode's true error; and `--precision 6` writes the audit embeddings themselves."""
    import json
    from jegal_amd import audio
    from test_gpu_xlmr import StubTokenizer
    monkeypatch.setattr(drivers, "_load_tokenizer", lambda name: StubTokenizer())
    gold = os.path.join(os.path.dirname(__file__), "golden")
    txt = tmp_path / "sample1.txt"
    txt.write_text(json.load(open(os.path.join(gold, "load_text.json")))["sample1"]["file"], encoding="utf-8")
    rng = np.random.default_rng(4243)
    T = 40
    crops = rng.integers(0, 256, (T, 270, 480, 3), dtype=np.uint8)
    crops[:, :110] = 0
    np.save(tmp_path / "sample1.npy", crops)
    res = str(tmp_path / "res")
    common = ["--checkpoint_path_gestsync", "synthetic", "--checkpoint_path_jegal", "synthetic", "--video_path", str(tmp_path / "sample1.npy"),
              "--audio_path", os.path.join(gold, "sample1.wav"), "--text_path", str(txt), "--xlmr_checkpoint", "synthetic", "--tokenizer", "stub"]
    capsys.readouterr()
    assert drivers.main(["inference_embs", "--modalities", "vta", "--audit", "--res_dir", res] + common) == 0
    text = capsys.readouterr().out
    line = [ln for ln in text.splitlines() if ln.startswith("Audit (fp32 engine")]
    assert len(line) == 1 and "gesture rel-L2" in line[0] and "content rel-L2" in line[0] and "WARNING" not in text, text
    got = pickle.load(open(os.path.join(res, "sample1.pkl"), "rb"))
    res32 = str(tmp_path / "res32")
    from jegal_amd._lib import Engine
    Engine.get("cuda:0").close()                       # the process-wide driver engine is finalized in the default mode: a new one for mode 6
    try:
        assert drivers.main(["inference_embs", "--modalities", "vta", "--precision", "6", "--res_dir", res32] + common) == 0
    finally:
        Engine.get("cuda:0").close()                   # ... and the later tests get a default-mode engine again
    got32 = pickle.load(open(os.path.join(res32, "sample1.pkl"), "rb"))
    ref_text = json.load(open(os.path.join(gold, "load_text.json")))["sample1"]
    gsd, jsd, xsd = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict()), synth.xlmr_state_dict()
    wbs = ref_text["word_boundaries"]
    with torch.no_grad():
        feats = O.gestsync_clip_feats(gsd, torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)))
        wav = audio.load_wav(os.path.join(gold, "sample1.wav")).astype("float32")
        mel = O.wav2filterbanks(wav[None], torch.from_numpy(audio.mel_filterbank()))
        enc = StubTokenizer()([ref_text["text"][0].split(" ")])
        states = O.xlmr_forward(xsd, enc["input_ids"].numpy(), enc["attention_mask"].numpy())
        pack = (states, enc["attention_mask"], [ref_text["text"][0].split(" ")], enc["input_ids"], enc["offset_mapping"])
        g, c = O.jegal_forward_inference(jsd, visual_feats=feats[None], visual_mask=torch.ones(1, T), text=pack, audio=mel,
                                         audio_mask=torch.ones(1, mel.shape[1] // 4), word_boundaries=wbs)
        g, c = O.l2_normalize(g[0]).numpy(), O.l2_normalize(c[0]).numpy()
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"\n--precision 6 vs the oracle chain: gesture {rel(got32['gesture_emb'], g):.2e} content {rel(got32['content_emb'], c):.2e}; "
          f"default vs oracle: gesture {rel(got['gesture_emb'], g):.2e} content {rel(got['content_emb'], c):.2e}\n{line[0]}")
    assert rel(got32["gesture_emb"], g) < 2e-5 and rel(got32["content_emb"], c) < 2e-5
    # the printed audit figures are the default mode's distance to the oracle, to the audit mode's own 2e-5
    import re
    nums = [float(x) for x in re.findall(r"rel-L2 ([0-9.e+-]+)", line[0])]
    assert abs(nums[0] - rel(got["gesture_emb"], g)) < 5e-5 and abs(nums[1] - rel(got["content_emb"], c)) < 5e-5
