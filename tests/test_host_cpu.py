"""CPU-side tests: the C-ABI library loads and exports every declared symbol, host logic
(segments, sharding, metric bookkeeping), and the world_size-2 gloo path."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import dist as jdist
from jegal_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    hdr = open(os.path.join(ROOT, "include", "jegal_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(jg_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "jegal_amd", "libjegal_hip.so"))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/jegal_hip.h but not exported"
    from jegal_amd import _lib
    assert set(_lib.EXPORTS) == set(declared)


def test_hot_kernels_do_not_spill():
    """the LN-fused GEMM carried 6 VGPR spills (28 B scratch), the 512x128 conv instance 8.  Read
    .vgpr_spill_count of every kernel from the gfx950 code objects inside libjegal_hip.so so that cannot come back."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernel_resources
    res = kernel_resources()
    hot = {k: v for k, v in res.items() if any(t in k for t in ("gemm_glds_kernel", "gemm_mlp_kernel", "conv1_direct_kernel", "attn_mfma"))}
    assert len(hot) >= 10
    bad = {k: v for k, v in hot.items() if v["spill"] != 0 or v["vgpr"] > 256}
    assert not bad, bad
    # and no kernel of the library at all may spill VGPRs: round 4 found audio_conv0_kernel at 512 VGPRs + 316 spills after a
    # three-line change (3.8 ms instead of 0.1 ms per call) -- invisible to every parity test
    spilled = {k: v["spill"] for k, v in res.items() if v["spill"] != 0}
    assert not spilled, spilled


def test_no_packed_fp32_instructions_in_the_library():
    """Round 6: on MI355X a v_pk_{fma,mul,add}_f32 whose low half selects the HIGH register of its second operand (op_sel:[0,1,..], what hipcc
    emits for `f32x4 * pair.y`) can read that operand as 0 in lanes 48-63 while waves of ANOTHER kernel issue MFMAs on the same SIMD
    (tools/experiments/pk_opsel_mfma/repro.hip) -- that is how two concurrent XLM-RoBERTa passes corrupted each other.  The library is built
    with the packed-fp32 feature off (csrc/Makefile, NOPK): no kernel may contain such an instruction, with or without selects."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import packed_opsel_scan as P
    with_selects = {k: v for k, v in P.scan().items() if v}
    assert not with_selects, {k: len(v) for k, v in with_selects.items()}
    old = P.PAT
    try:
        P.PAT = re.compile(r"\bv_pk_((fma|mul|add)_f32|mov_b32)\b")           # v_pk_mov_b32 reads register pairs with selects too (same feature)
        res = P.scan()
    finally:
        P.PAT = old
    assert len(res) >= 150                       # every kernel of every translation unit was looked at
    packed = {k: len(v) for k, v in res.items() if v}
    assert not packed, packed


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from jegal_amd._lib import Engine
    with pytest.raises(RuntimeError):
        Engine.get()


def test_text_word_segments_match_oracle(golden_dir):
    from jegal_amd.jegal import text_word_segments
    g = np.load(os.path.join(golden_dir, "jegal_text.npz"))
    tbatch = [["w0", "w1", "w2", "w3"], ["x0", "x1", "x2"]]
    segs, invalid = text_word_segments(g["ids"], g["offsets"], tbatch)
    assert invalid == []
    # clip 0: starts 1,2,4,5 ; last word runs to L=9 (swallows </s> and <pad>), jegal.py:168-171
    assert segs[0] == [(1, 2), (2, 4), (4, 5), (5, 9)]
    assert segs[1] == [(1, 2), (2, 3), (3, 9)]
    emb = torch.randn(2, 9, 16)
    words, inv = O.word_level_text(emb, tbatch, g["ids"], g["offsets"])
    for b in range(2):
        mine = torch.stack([emb[b, lo:hi].mean(0) for lo, hi in segs[b]])
        assert torch.allclose(mine, words[b], atol=1e-6)
    # more words than detected starts -> sample dropped (jegal.py:161-165)
    segs2, invalid2 = text_word_segments(g["ids"], g["offsets"], [["a"] * 7, ["x0"]])
    assert invalid2 == [0] and segs2[0] is None and segs2[1] == [(1, 2)]


def test_audio_word_segments():
    from jegal_amd.jegal import audio_word_segments
    wb = [[["a", 3, 9], ["b", 10, 10], ["c", 12, 30]], [["d", 0, 5], ["e", 6, 20]]]
    segs = audio_word_segments(wb, 40)
    assert segs[0] == [(0, 7), (7, 8), (9, 28)]
    assert segs[1] == [(0, 6), (6, 21)]
    assert audio_word_segments([[["a", 0, 5], ["b", 30, 60]]], 40)[0] == [(0, 6), (30, 40)]   # slice clamps
    with pytest.raises(IndexError):
        audio_word_segments([[["a", 0, 5], ["b", 50, 60]]], 40)                              # empty slice


def test_shard_range_matches_reference_rule():
    # extract_gestsync_feats.py:366-370
    n = 10
    parts = [jdist.shard_range(n, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sum(hi - lo for lo, hi in parts) == n


def test_metrics_from_ranks_equals_reference_compute_metrics(golden_dir):
    from jegal_amd.metrics import metrics_from_ranks
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    sim = g["sim"]
    d = np.diag(sim)[:, None]
    rank = (sim > d).sum(1)
    ties = (sim == d).sum(1)
    assert ties.max() == 2            # the planted duplicate gallery row
    m = metrics_from_ranks(rank, ties)
    ref = O.compute_metrics(sim)
    for k in ("R1", "R5", "R10", "R25", "R50", "MR"):
        assert m[k] == ref[k], k
    for k in ("R5", "R10", "R25", "R50", "MR"):
        assert m[k] == float(g[k])


def test_load_text_driver(golden_dir, tmp_path):
    import json
    from jegal_amd.extract import load_text
    ref = json.load(open(os.path.join(golden_dir, "load_text.json")))
    for name, r in ref.items():
        p = tmp_path / (name + ".txt")
        p.write_text(r["file"], encoding="utf-8")
        text, wbs = load_text(str(p))
        assert text == r["text"] and wbs == r["word_boundaries"]


_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as td
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
from jegal_amd import dist as jdist, synth
from jegal_amd.metrics import metrics_from_ranks
import jegal_oracle as O
jdist.init_from_env("gloo")
r, w = jdist.rank(), jdist.world_size()
assert w == 2
N = 101
g, c = synth.planted_retrieval(5, N)
g[7] = g[3]
lo, hi = jdist.shard_range(N)
e1 = torch.from_numpy(c[lo:hi]); e2 = torch.from_numpy(g[lo:hi])
gallery, off = jdist.all_gather_rows(e2)          # ragged: 51 + 50 rows
assert gallery.shape == (N, 512) and off == lo
assert torch.equal(gallery, torch.from_numpy(g))
sim = (e1 @ gallery.T).numpy()                    # stand-in for jg_sim_rank on the CPU test box
d = sim[np.arange(hi - lo), np.arange(lo, hi)][:, None]
rank = torch.from_numpy((sim > d).sum(1).astype(np.int32)).reshape(-1, 1)
ties = torch.from_numpy((sim == d).sum(1).astype(np.int32)).reshape(-1, 1)
rank_all, _ = jdist.all_gather_rows(rank); ties_all, _ = jdist.all_gather_rows(ties)
m = metrics_from_ranks(rank_all.flatten().numpy(), ties_all.flatten().numpy())
full = (torch.from_numpy(c) @ torch.from_numpy(g).T).numpy()
ref = O.compute_metrics(full)
assert m == ref, (m, ref)
tot = jdist.all_reduce_sum(torch.tensor([hi - lo]))
assert int(tot) == N
jdist.barrier()
print("rank", r, "ok")
"""


def _run_two_ranks(script):
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
            out, _ = p.communicate(timeout=240)
            outs.append((p.returncode, out.decode()))
    finally:
        for p in procs:            # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.kill()
    return outs


def test_two_rank_gloo_sharded_retrieval(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    # the rendezvous port is picked by bind(0)+close and can be taken by another process before rank 0
    # listens on it: a failed rendezvous is retried on a fresh port, a failed ASSERTION in a worker is not
    for attempt in range(3):
        outs = _run_two_ranks(script)
        if all(rc == 0 for rc, _ in outs):
            break
        text = "\n".join(o for _, o in outs)
        if "AssertionError" in text or attempt == 2:
            raise AssertionError(text)
    for rc, out in outs:
        assert " ok" in out


def test_bench_launcher_fails_fast_when_a_rank_dies():
    """bench.py --gpus 2 without a torchrun environment starts the ranks itself; a rank that dies before the rendezvous (here:
    no HIP device in this container) must take the job down promptly with a non-zero exit, not leave the launcher waiting for
    the process-group timeout."""
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run")
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, timeout=170)
    assert r.returncode != 0
    assert b"rank" in r.stderr
    assert time.monotonic() - t0 < 150


def test_cli_drivers_pick_the_calibration_free_mode_for_real_checkpoints():
    """the default bias-corrected mode depends on a calibration that was validated on the synthetic weights
    only; a real checkpoint runs a calibration-free mode -- since round 5 the run-time corrected one (mode 5: per-clip E[x]),
    before that hi+lo (mode 1) -- unless calibration clips or an explicit --precision are given."""
    from types import SimpleNamespace as NS
    from jegal_amd.drivers import pick_precision, REAL_CHECKPOINT_PRECISION
    from jegal_amd._lib import PREC_FP16_RC
    assert REAL_CHECKPOINT_PRECISION == PREC_FP16_RC == 5
    assert pick_precision(NS(precision=None, calibrate_frames=None), ["synthetic", None]) == 5        # the library default, whatever the checkpoint
    assert pick_precision(NS(precision=None, calibrate_frames=None), ["/ckpt/gestsync.pth"]) == 5
    assert pick_precision(NS(precision=None, calibrate_frames="clips.npy"), ["/ckpt/gestsync.pth"]) == 3
    assert pick_precision(NS(precision=0, calibrate_frames=None), ["/ckpt/jegal.pth"]) == 0
    # calibration clips only count when the command can run the calibration (needs GestSync: extract_jegal_embs cannot)
    assert pick_precision(NS(precision=None, calibrate_frames="clips.npy"), ["/ckpt/jegal.pth"], can_calibrate=False) == 5


def test_masked_packer_layout():
    """Host side of the masked upload (jegal_amd.extract._MaskedPacker): the rows >= row0 of every frame back to back, row0 and byte
    offsets per frame -- exactly what jg_unpack_masked consumes (include/jegal_hip.h); uniform and per-frame mask heights."""
    from jegal_amd.extract import _MaskedPacker, FRAME_ROW_BYTES
    T = 3
    rng = np.random.default_rng(0)
    clips = rng.integers(1, 256, (2, T, 270, 480, 3), dtype=np.uint8)
    pk = _MaskedPacker(2, T, pinned=False)
    pk.add(clips[0], 110)
    pk.add(clips[1], np.array([0, 270, 135]))
    assert pk.n == 2
    row0, offs, buf = pk.row0.numpy(), pk.offs.numpy(), pk.buf.numpy()
    assert row0.tolist() == [110, 110, 110, 0, 270, 135]
    kept = (270 - row0.astype(np.int64)) * FRAME_ROW_BYTES
    assert offs.tolist() == np.concatenate(([0], np.cumsum(kept)[:-1])).tolist() and pk.used == int(kept.sum())
    assert all(o % 16 == 0 for o in offs)
    for f in range(2 * T):
        b, t = divmod(f, T)
        assert np.array_equal(buf[offs[f]:offs[f] + kept[f]], clips[b, t, row0[f]:].reshape(-1))
    with pytest.raises(ValueError):
        pk.add(clips[0], 110)                       # batch is full
    pk.reset()
    with pytest.raises(ValueError):
        pk.add(clips[0], 271)


def test_word_segments_per_clip_equal_the_oracle_on_each_clip_alone():
    """Host side of the batch-invariant content path: with per-sample lengths the row ranges of a padded
    batch are those the reference's word pooling (jegal.py:131-252, restated in the oracle) takes on each clip ALONE; without
    them the last word runs to the padded length (the reference's B > 1 behaviour, jegal.py:168-171)."""
    import jegal_oracle as O
    from jegal_amd import synth
    from jegal_amd.jegal import audio_word_segments, text_word_segments
    lens, words = [25, 40, 31], [3, 7, 4]
    clips = [synth.synth_ragged_clip(40 + i, T, W) for i, (T, W) in enumerate(zip(lens, words))]
    L = max(len(c["ids"]) for c in clips)
    ids = np.ones((3, L), np.int64); offs = np.zeros((3, L, 2), np.int64); emb = np.zeros((3, L, 8), np.float32)
    rng = np.random.default_rng(0)
    for i, c in enumerate(clips):
        l = len(c["ids"])
        ids[i, :l], offs[i, :l] = c["ids"], c["offsets"]
        emb[i] = rng.standard_normal((L, 8))
    tb = [c["phrase"].split(" ") for c in clips]
    segs, invalid = text_word_segments(ids, offs, tb, lengths=[len(c["ids"]) for c in clips])
    assert invalid == []
    for i, c in enumerate(clips):
        l = len(c["ids"])
        alone, _ = O.word_level_text(torch.from_numpy(emb[i:i + 1, :l]), [tb[i]], ids[i:i + 1, :l], offs[i:i + 1, :l])
        mine = np.stack([emb[i, lo:hi].mean(0) for lo, hi in segs[i]])
        np.testing.assert_allclose(mine, alone[0].numpy(), rtol=1e-6, atol=1e-6)
        assert segs[i][-1][1] == l                                   # the last word swallows </s>, not the pads
    padded, _ = text_word_segments(ids, offs, tb)
    assert padded[0][-1][1] == L and padded[0][:-1] == segs[0][:-1]  # only the last word differs
    # audio: slices are clipped to the clip's own audio steps
    wbs = [c["word_boundaries"] for c in clips]
    a = audio_word_segments(wbs, lens)
    for i in range(3):
        assert a[i][-1] == (wbs[i][-1][1], lens[i]) and all(hi - lo == w[2] - w[1] + 1 for (lo, hi), w in zip(a[i], wbs[i]))
    long_wb = [[["w0", 0, 10], ["w1", 11, 60]]]
    assert audio_word_segments(long_wb, [30])[0][-1] == (11, 30) and audio_word_segments(long_wb, 40)[0][-1] == (11, 40)
    with pytest.raises(ValueError):
        text_word_segments(ids, offs, tb, lengths=[0, 5, 5])


_WORKER_SPOT_ASD = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
from jegal_amd import dist as jdist, synth, metrics as M
import jegal_oracle as O
jdist.init_from_env("gloo")
r, w = jdist.rank(), jdist.world_size()
assert w == 2
# ---- spotting (config 5 shape, small): every rank scores ITS contiguous block, two counters are all-reduced
n = 41
gest, cont, bounds, targets = synth.planted_spotting(1238, n, n_frames=60, n_words=8, noise=3.0)
lo, hi = jdist.shard_range(n)
pred, score = [], []
for i in range(lo, hi):                           # stand-in for jg_spot on the CPU test box: the oracle's argmax / probability
    _, p, s = O.spotting_correct(gest[i], cont[i], bounds[i], targets[i])
    pred.append(p); score.append(s)
correct, total = M.reduce_counts(M.spotting_counts(pred, score, bounds[lo:hi], targets[lo:hi]))
assert total == n
acc = 100.0 * correct / total
ref = O.spotting_accuracy(gest, cont, bounds, targets)
assert abs(acc - ref) < 1e-9 and 5.0 < acc < 95.0, (acc, ref)
# ---- ASD: queries sharded, four counters all-reduced
contents, positives, negatives = synth.planted_asd(7, 37)
(ref2, ref4, ref6), ref_pred = O.asd_accuracy(contents, positives, negatives)
lo, hi = jdist.shard_range(len(contents))
c2, c4, c6, nq = M.reduce_counts(M.asd_counts(ref_pred[lo:hi]))
assert nq == len(contents)
assert (c2 / nq, c4 / nq, c6 / nq) == (ref2, ref4, ref6)
# a rank without a single item still takes part in the reduction
z = M.reduce_counts([0, 0] if r == 1 else [3, 5])
assert z == [3, 5]
jdist.barrier()
print("rank", r, "ok", acc, ref2)
"""


def test_two_rank_gloo_sharded_spotting_and_asd(tmp_path):
    """SURVEY 8e / VERDICT r3 item 5: spotting and ASD shard by clips / queries with a counter all-reduce only.  World-size-2 gloo
    run of jegal_amd.metrics' host logic (spotting_counts / asd_counts / reduce_counts) with the oracle standing in for the device
    kernels: the sharded accuracies equal the single-rank oracle's exactly."""
    script = tmp_path / "worker_spot.py"
    script.write_text(_WORKER_SPOT_ASD.format(root=ROOT))
    for attempt in range(3):
        outs = _run_two_ranks(script)
        if all(rc == 0 for rc, _ in outs):
            break
        text = "\n".join(o for _, o in outs)
        if "AssertionError" in text or attempt == 2:
            raise AssertionError(text)
    for rc, out in outs:
        assert " ok" in out


def test_source_packer_layout():
    """Host side of the source-resolution upload (jegal_amd.extract._SourcePacker): per frame the source rows mask_y+1 .. H-1 back to
    back, mask_y and byte offsets per frame -- what jg_mask_resize_packed consumes (include/jegal_hip.h)."""
    from jegal_amd.extract import _SourcePacker
    T, H, W = 3, 228, 314
    rng = np.random.default_rng(1)
    clips = rng.integers(0, 256, (2, T, H, W, 3), dtype=np.uint8)
    pk = _SourcePacker(2, T, H, W, pinned=False)
    pk.add(clips[0], 90)
    pk.add(clips[1], np.array([-1, H - 1, H + 30]))                      # whole frame / nothing / nothing (mask beyond the frame)
    my, offs, buf = pk.mask_y.numpy(), pk.offs.numpy(), pk.buf.numpy()
    assert my.tolist() == [90, 90, 90, -1, H - 1, H + 30]
    row0 = np.clip(my.astype(np.int64) + 1, 0, H)
    kept = (H - row0) * W * 3
    assert offs.tolist() == np.concatenate(([0], np.cumsum(kept)[:-1])).tolist() and pk.used == int(kept.sum())
    for f in range(2 * T):
        b, t = divmod(f, T)
        assert np.array_equal(buf[offs[f]:offs[f] + kept[f]], clips[b, t, row0[f]:].reshape(-1))
    with pytest.raises(ValueError):
        pk.add(clips[0], 90)
    pk.reset()
    with pytest.raises(ValueError):
        pk.add(clips[0], -2)


def test_importing_the_package_does_not_touch_the_environment():
    """`import jegal_amd` used to set GPU_MAX_HW_QUEUES for the whole process (and its children).  Now an application calls
    jegal_amd.want_hw_queues() before its first HIP call; the call never overrides the caller's own setting."""
    import subprocess
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import jegal_amd; assert 'GPU_MAX_HW_QUEUES' not in os.environ; "
            "assert jegal_amd.want_hw_queues() == '8' and os.environ['GPU_MAX_HW_QUEUES'] == '8'; "
            "os.environ['GPU_MAX_HW_QUEUES'] = '4'; assert jegal_amd.want_hw_queues() == '4'; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
