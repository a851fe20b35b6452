"""JG_PREC_FP32, the audit mode (round 6): every GEMM / convolution / attention in fp32 on the device (exact-fp32 MFMA,
fp32 activations end to end) -- the on-device stand-in for the reference's fp32 CPU path (inference_embs.py:497: autocast does nothing
without CUDA).  Held to 2e-5 of the reference goldens and of the oracle, i.e. ~50 x tighter than the fp16 modes' 1e-3 contract; the
stage mask (option audit_stages) decomposes the fp16 modes' error by stage.  Through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
AUD_TOL = 2e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def maxabs(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def fp32():
    from jegal_amd._lib import Engine, PREC_FP32
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    e = Engine(0, precision=PREC_FP32)
    gs = GestSync(engine=e).load_state_dict(synth.gestsync_state_dict())
    jg = JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    yield e, gs, jg
    e.close()


@pytest.fixture(scope="module")
def oracle_sd():
    return O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())


def test_audit_short_clips_vs_oracle(fp32, oracle_sd):
    e, _, _ = fp32
    gsd, jsd = oracle_sd
    B, T = 3, 7
    frames = synth.synth_frames(611, B, T)
    dev = torch.from_numpy(frames).cuda()
    feats = e.gestsync_clip(dev).cpu()
    emb = e.extract_gesture(dev).cpu()
    f01 = e.gestsync_clip(torch.from_numpy(frames.astype(np.float32) / np.float32(255.0)).cuda()).cpu()       # fp32 frames in [0, 1]
    for b in range(B):
        with torch.no_grad():
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            g = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
        print(f"\naudit clip {b}: feats rel {rel(feats[b], f):.2e} | embedding rel {rel(emb[b], g):.2e} max-abs {maxabs(emb[b], g):.2e}", end="")
        assert rel(feats[b], f) < AUD_TOL and rel(f01[b], f) < AUD_TOL
        assert rel(emb[b], g) < AUD_TOL and maxabs(emb[b], g) < AUD_TOL


def test_audit_reference_goldens(fp32, golden_dir):
    """The goldens are outputs of the REAL reference (oracle/make_golden.py): the audit mode reproduces them to fp32 summation order."""
    _, gs, jg = fp32
    g = np.load(os.path.join(golden_dir, "gestsync_clip.npz"))
    frames = synth.synth_frames(int(g["seed"]), 1, int(g["T"]))[0]
    f01 = O.pad_clip(torch.from_numpy(frames.astype(np.float32) / np.float32(255.0)))
    vol = f01.permute(3, 0, 1, 2)
    x = torch.stack([vol[:, i:i + 25] for i in range(f01.shape[0] - 24)])
    out, out_conv = gs.forward_vid(x[:2].cuda(), return_feats=True)
    feats = gs.extract_clip_feats(torch.from_numpy(frames).cuda())[0]
    errs = {"out_conv": rel(out_conv, g["out_conv"][:2]), "out_full": rel(out, g["out_full"]), "feats": rel(feats, g["feats"])}
    # JEGAL branches
    rng = np.random.default_rng(9002)
    vf = rng.standard_normal((2, 40, 1024)).astype(np.float32)
    vf[1, 30:] = 0
    vm = np.ones((2, 40), np.float32)
    vm[1, 30:] = 0
    vf, vm = torch.from_numpy(vf), torch.from_numpy(vm)
    gg = np.load(os.path.join(golden_dir, "jegal_gesture.npz"))
    errs["fwd_gestures"] = rel(jg.forward_gestures(vf.cuda(), vm.cuda().unsqueeze(1)), gg["fwd_gestures"])
    errs["gesture"] = rel(jg.forward_inference(visual_feats=vf.cuda(), visual_mask=vm.cuda()), gg["gesture"])
    ga = np.load(os.path.join(golden_dir, "jegal_audio.npz"))
    mel = torch.from_numpy(synth.synth_mel(int(ga["seed"]), 2, 160)).cuda()
    wb = [[["a", 3, 9], ["b", 10, 10], ["c", 12, 30]], [["d", 0, 5], ["e", 6, 20]]]
    errs["fwd_audio"] = rel(jg.forward_audio(mel), ga["fwd_audio"])
    errs["audio_content"] = rel(jg.forward_inference(audio=mel, audio_mask=torch.ones(2, 40), word_boundaries=wb), ga["content"])
    gt = np.load(os.path.join(golden_dir, "jegal_text.npz"))
    pack = (torch.from_numpy(gt["states"]), torch.from_numpy(gt["mask"]), [["w0", "w1", "w2", "w3"], ["x0", "x1", "x2"]],
            torch.from_numpy(gt["ids"]), torch.from_numpy(gt["offsets"]))
    errs["fwd_text"] = rel(jg.forward_text(torch.from_numpy(gt["states"]).cuda(), torch.from_numpy(gt["mask"]).cuda().unsqueeze(1)), gt["fwd_text"])
    errs["text_content"] = rel(jg.forward_inference(text=pack), gt["content"])
    gv = np.load(os.path.join(golden_dir, "jegal_vta.npz"))
    wb2 = [[["w0", 2, 6], ["w1", 7, 12], ["w2", 13, 13], ["w3", 15, 30]], [["x0", 1, 4], ["x1", 5, 9], ["x2", 10, 22]]]
    mel2 = torch.from_numpy(synth.synth_mel(9003, 2, 160)).cuda()
    ge, ce = jg.forward_inference(visual_feats=vf.cuda(), visual_mask=vm.cuda(), text=pack, audio=mel2, audio_mask=torch.ones(2, 40),
                                  word_boundaries=wb2)
    errs["vta_gesture"], errs["vta_content"] = rel(ge, gv["gesture"]), rel(ce, gv["content"])
    print("\naudit vs reference goldens: " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()), end="")
    assert max(errs.values()) < AUD_TOL, errs


def test_audit_ragged_audio_and_long_sequences(fp32, oracle_sd):
    """Per-clip mel lengths in a zero-padded batch (the tail zeroing of every conv layer), T = 220 gesture tokens with a key mask and
    L = 70 text tokens (key chunks of the fp32 attention kernel beyond one chunk)."""
    e, _, jg = fp32
    _, jsd = oracle_sd
    mel = synth.synth_mel(77, 3, 240)
    valid = [240, 163, 96]
    padded = mel.copy()
    for b, v in enumerate(valid):
        padded[b, v:] = 0
    out = e.jegal_audio(torch.from_numpy(padded).cuda(), valid_len=valid).cpu()
    for b, v in enumerate(valid):
        with torch.no_grad():
            ref = O.jegal_forward_audio(jsd, torch.from_numpy(mel[b:b + 1, :v]))[0]
        assert rel(out[b, :ref.shape[0]], ref) < AUD_TOL, (b, rel(out[b, :ref.shape[0]], ref))
    rng = np.random.default_rng(5)
    T = 220
    vf = torch.from_numpy(rng.standard_normal((2, T, 1024)).astype(np.float32))
    vm = torch.ones(2, T)
    vm[1, 170:] = 0
    vf[1, 170:] = 0
    got = jg.forward_inference(visual_feats=vf.cuda(), visual_mask=vm.cuda()).cpu()
    with torch.no_grad():
        ref = O.jegal_forward_inference(jsd, visual_feats=vf, visual_mask=vm)
    assert rel(got, ref) < AUD_TOL, rel(got, ref)
    L = 70
    st = torch.from_numpy(rng.standard_normal((2, L, 768)).astype(np.float32))
    tm = torch.ones(2, L)
    tm[0, 50:] = 0
    ft = jg.forward_text(st.cuda(), tm.cuda().unsqueeze(1)).cpu()
    with torch.no_grad():
        rt = O.jegal_forward_text(jsd, st, tm.unsqueeze(1))
    assert rel(ft, rt) < AUD_TOL, rel(ft, rt)


def test_audit_xlmr_against_transformers_golden(golden_dir):
    """XLM-RoBERTa in fp32 against the outputs of transformers.XLMRobertaModel itself (tests/golden/xlmr*.npz, oracle/make_golden.py)."""
    from jegal_amd._lib import Engine, PREC_FP32
    from jegal_amd.xlmr import XLMRoberta
    e = Engine(0, precision=PREC_FP32)
    try:
        for name, layers in (("xlmr.npz", 4), ("xlmr12.npz", 12)):
            path = os.path.join(golden_dir, name)
            if not os.path.exists(path):
                continue
            g = np.load(path)
            sd = synth.xlmr_state_dict(layers=layers)
            x = XLMRoberta(engine=e).load_state_dict(sd)
            ids, mask = torch.from_numpy(g["input_ids"]).cuda(), torch.from_numpy(g["attention_mask"]).cuda()
            out = x(ids, attention_mask=mask).last_hidden_state.cpu()
            m = torch.from_numpy(g["attention_mask"]).bool()
            r = rel(out[m], torch.from_numpy(g["last_hidden_state"])[m])
            print(f"\naudit xlmr {layers} layers vs transformers golden: rel-L2 {r:.2e}", end="")
            assert r < AUD_TOL
    finally:
        e.close()


def test_audit_stage_mask_decomposes_the_fp16_error(oracle_sd):
    """Option audit_stages on a default-mode (JG_PREC_FP16_RC) handle that kept the fp32 matrices: mask 0 is the default mode bit for bit,
    mask 7 the audit mode bit for bit, and the masks in between say which stage the fp16 error comes from (DESIGN.md section 3 has the
    full-length table from tools/precision_floor.py)."""
    from jegal_amd._lib import Engine, PREC_FP16_RC, PREC_FP32
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    gsd, jsd = oracle_sd
    B, T = 2, 64
    frames = synth.synth_frames(4242, B, T)
    dev = torch.from_numpy(frames).cuda()
    with torch.no_grad():
        refs = []
        for b in range(B):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            refs.append(O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy())
    outs = {}
    for tag, prec, aw in (("rc", PREC_FP16_RC, 0), ("rc+aw", PREC_FP16_RC, 1), ("fp32", PREC_FP32, 0)):
        e = Engine(0, precision=prec)
        try:
            if aw:
                e.set_option("audit_weights", 1)
            GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
            JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
            if aw:
                for mask in (0, 1, 2, 4, 3, 6, 7):
                    e.set_option("audit_stages", mask)
                    outs[f"mask{mask}"] = e.extract_gesture(dev).cpu().numpy()
                e.set_option("audit_stages", 0)
            else:
                outs[tag] = e.extract_gesture(dev).cpu().numpy()
                if prec == PREC_FP16_RC:
                    with pytest.raises(Exception):
                        e.set_option("audit_stages", 1)             # the fp32 matrices were not kept
        finally:
            e.close()
    assert np.array_equal(outs["mask0"], outs["rc"])
    assert np.array_equal(outs["mask7"], outs["fp32"])
    errs = {k: max(rel(v[b], refs[b]) for b in range(B)) for k, v in outs.items()}
    print("\nstage decomposition (T = 64; 1 conv, 2 GestSync transformer, 4 JEGAL in fp32): " + "  ".join(f"{k} {v:.2e}" for k, v in sorted(errs.items())), end="")
    assert errs["fp32"] < AUD_TOL and errs["rc"] < 1e-3
    assert errs["mask7"] < errs["mask3"] <= errs["rc"] * 1.05          # more fp32 stages, less error
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(errs, open(os.path.join(ROOT, "gpurun_out", "audit_stage_table_T64.json"), "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def test_audit_full_length_clip_vs_oracle(fp32, oracle_sd):
    """BASELINE configs[1] shapes (T = 150) in the audit mode."""
    e, _, _ = fp32
    gsd, jsd = oracle_sd
    T = 150
    frames = synth.synth_frames(1234, 1, T)
    emb = e.extract_gesture(torch.from_numpy(frames).cuda()).cpu()
    with torch.no_grad():
        f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[0].astype(np.float32) / np.float32(255.0)))
        g = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
    print(f"\naudit T = 150: embedding rel-L2 {rel(emb[0], g):.2e} max-abs {maxabs(emb[0], g):.2e}", end="")
    assert rel(emb[0], g) < AUD_TOL and maxabs(emb[0], g) < AUD_TOL


def test_split_operand_options_reduce_the_error(oracle_sd):
    """Options jegal_fp32_ends (default on) and jegal_ffn_x3 (default off: +3 % step time) move GEMMs of the JEGAL branch to fp32 activations on
    the split-operand kernel; conv_round_diffuse (default on) rounds conv weights with error diffusion across the taps.  Each must stay
    inside the contract and the more exact arrangement must not be worse (T = 64, two clips; DESIGN.md section 3 has the T = 150 table)."""
    from jegal_amd._lib import Engine, PREC_FP16_RC
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    gsd, jsd = oracle_sd
    B, T = 2, 64
    frames = synth.synth_frames(4242, B, T)
    dev = torch.from_numpy(frames).cuda()
    with torch.no_grad():
        refs = []
        for b in range(B):
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
            refs.append(O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0]).numpy())
    errs = {}
    for tag, opts in (("round5", {"jegal_fp32_ends": 0, "conv_round_diffuse": 0}), ("default", {}), ("ffn_x3", {"jegal_ffn_x3": 1})):
        e = Engine(0, precision=PREC_FP16_RC)
        try:
            for k, v in opts.items():
                e.set_option(k, v)
            GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
            JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
            emb = e.extract_gesture(dev).cpu().numpy()
        finally:
            e.close()
        errs[tag] = max(rel(emb[b], refs[b]) for b in range(B))
    print("\ngesture rel-L2 (T = 64): " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()), end="")
    assert errs["round5"] < 1e-3 and errs["default"] < 0.8 * errs["round5"] and errs["ffn_x3"] < 0.9 * errs["default"], errs
