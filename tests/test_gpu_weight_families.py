"""Parity across WEIGHT FAMILIES (run with -m gpu, through the C ABI).

The reference runs released checkpoints (README.md:52-59, inference_embs.py:92-119); none is available offline, and every parity
number of rounds 1-4 came from one draw of He-initialised Gaussian weights.  The oracle is a function of the state_dict, so the same
comparison runs on other draws (synth._Gen): two more Gaussian seeds, heavy-tailed weights with BatchNorm / LayerNorm scales that
span decades, and sharp attention (q / k projections x 2 and x 4).  Per family: the gesture path on two full-length clips (T = 150), the
tri-modal content path at config-3 shapes (two clips) and 12 layers of XLM-RoBERTa, in every precision treatment a driver could
select.  Every rel-L2 / max-abs is printed and collected in gpurun_out/family_table.json (DESIGN.md section 3 quotes it).

Conditioning.  The x 4 family (attention logits x 16) is ill-conditioned as a NETWORK: the test measures, in float64, how much a
relative perturbation (1e-6) of the conv stack's output moves the final embedding through GestSync's transformer + ff_vid + the JEGAL
branch.  Gaussian draw: 0.10, heavy: 0.08, x 2: 0.28, x 4: 1.7 (T = 40 figures) -- the x 4 net amplifies every upstream rounding 17 times
more than the net the 1e-3 contract was written for, in ANY implementation (with random features at the JEGAL branch's input that
branch alone amplifies x 24: a 1e-4 input perturbation moves its fp64 output by 2.3e-3).  No 16-bit operand format (fp16: 2^-11 per
operand; the reference's own CUDA autocast path is the same arithmetic) can hold 1e-3 there.

Round 6 (VERDICT r5 item 2 / ADVICE r5): the widened bound (1e-3 x factor ratio / 3, which let a 33 % error pass) is gone.  Every family also runs
in the fp32 AUDIT mode (JG_PREC_FP32: exact-fp32 MFMAs, fp32 activations), which must reproduce the fp32 oracle to 2e-5 on EVERY family,
`sharp` included (gesture / content path: measured 1.1e-6 .. 5.9e-6; for XLM-RoBERTa, whose `sharp` draw amplifies perturbations 7 400 x,
the bound scales with the measured conditioning beyond the regime -- fp32 summation order is a perturbation like any other).  Rule for the fp16 modes: a family whose conditioning is
within 3 x the Gaussian draw's must meet the plain 1e-3 in the mode the drivers select.  Beyond that no 16-bit operand format can
(reported), and the test asserts what a user WITHOUT an oracle would see: the distance between the default mode and the audit mode --
what `python -m jegal_amd.drivers inference_embs ... --audit` prints -- equals the true error to the audit mode's own accuracy, so the
driver's WARNING fires exactly when the contract is broken.
"""
import json
import os

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3
AUD_TOL = 2e-5
T = 150
FAMILIES = [("gauss", 0), ("gauss", 1), ("gauss", 2), ("heavy", 0), ("sharp2", 0), ("sharp", 0)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = {}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def record(key, **kw):
    TABLE[key] = kw
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "family_table.json"), "w") as f:
            json.dump(TABLE, f, indent=1, sort_keys=True)
    except OSError:
        pass


def fam_id(f):
    return f"{f[0]}+{f[1]}"


def amplification(gt, jt, conv):
    """Relative change of the fp64 gesture embedding per relative perturbation (1e-6, seeded Gaussian) of the conv stack's output
    (512, P-4): GestSync transformer + ff_vid + mean + JEGAL gesture branch + normalisation."""
    gd = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in gt.items()}
    jd = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in jt.items()}
    c = conv.double()
    n = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(c.shape)))

    def f(v):
        ft = O.gestsync_feats_from_conv(gd, v)
        return O.l2_normalize(O.jegal_forward_inference(jd, visual_feats=ft[None], visual_mask=torch.ones(1, ft.shape[0], dtype=torch.float64))[0])
    with torch.no_grad():
        a, b = f(c), f(c * (1 + 1e-6 * n))
    return float((b - a).norm() / a.norm()) / 1e-6


_GAUSS_AMP = []
_GAUSS_XLMR_AMP = []


def gauss_amplification(frames0):
    """The same factor for the reference family (the seeded Gaussian draw the 1e-3 contract was written on), once per session."""
    if not _GAUSS_AMP:
        gt, jt = O.tensors(synth.gestsync_state_dict(include_unused=False)), O.tensors(synth.jegal_state_dict())
        with torch.no_grad():
            _, conv = O.gestsync_clip_feats(gt, torch.from_numpy(frames0.astype(np.float32) / np.float32(255.0)), return_conv=True)
        _GAUSS_AMP.append(amplification(gt, jt, conv))
    return _GAUSS_AMP[0]


def gesture_modes():
    import jegal_amd._lib as L
    return [("bc_builtin", L.PREC_FP16_BC, None), ("bc_own_clips", L.PREC_FP16_BC, "own"), ("w2", L.PREC_FP16_W2, None), ("rc", L.PREC_FP16_RC, None),
            ("fp32_audit", L.PREC_FP32, None)]


@pytest.mark.parametrize("family", FAMILIES, ids=fam_id)
def test_gesture_and_content_across_weight_families(family):
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    import jegal_amd._lib as L
    name, off = family
    gsd = synth.gestsync_state_dict(seed=synth.GESTSYNC_SEED + off, include_unused=False, family=name)
    jsd = synth.jegal_state_dict(seed=synth.JEGAL_SEED + off, family=name)
    gt, jt = O.tensors(gsd), O.tensors(jsd)
    B, W = 2, 10
    frames = synth.synth_frames(1234, B, T)
    mel = synth.synth_mel(1235, B, 4 * T)
    states, tmask, ids, offs = synth.synth_text(1236, B, W)
    wbs = synth.synth_boundaries(B, W)
    tbatch = [[w[0] for w in wb] for wb in wbs]
    pack = (torch.from_numpy(states), torch.from_numpy(tmask), tbatch, ids, offs)
    refs = []
    conv0 = None
    with torch.no_grad():
        for b in range(B):
            f, conv = O.gestsync_clip_feats(gt, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)), return_conv=True)
            conv0 = conv if conv0 is None else conv0
            g = O.l2_normalize(O.jegal_forward_inference(jt, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
            refs.append((f.numpy(), g.numpy()))
        cref = O.l2_normalize(O.jegal_forward_inference(jt, text=pack, audio=torch.from_numpy(mel), audio_mask=None, word_boundaries=wbs)).numpy()
    assert np.isfinite(cref).all() and all(np.isfinite(r[1]).all() for r in refs)
    amp, amp_ref = amplification(gt, jt, conv0), gauss_amplification(frames[0])
    ratio = amp / amp_ref
    in_regime = ratio <= 3.0
    aud_bound = AUD_TOL          # the gesture / content path: the plain 2e-5 on every family, `sharp` included (measured 1.1e-6 .. 5.9e-6)
    print(f"\n[{fam_id(family)}] conditioning (fp64, d embedding / d conv features): {amp:.3f} = {ratio:.1f} x the Gaussian draw's -> "
          f"{'in the regime of the 1e-3 contract' if in_regime else 'OUTSIDE the regime of any 16-bit operand format: fp16 modes reported, audit mode asserted'}", end="")
    record(f"{fam_id(family)}/conditioning", amplification=amp, ratio_to_gauss=ratio, in_regime=in_regime, audit_bound=aud_bound)
    dev = torch.from_numpy(frames).cuda()
    worst, outs = {}, {}
    for mname, mode, cal in gesture_modes():
        e = Engine(0, precision=mode)
        try:
            GestSync(engine=e).load_state_dict(gsd)
            jg = JEGAL(engine=e).load_state_dict(jsd)
            if cal == "own":
                e.calibrate(dev)
            emb = e.extract_gesture(dev).cpu().numpy()
            feats = e.gestsync_clip(dev).cpu().numpy()
            cont = e.l2norm(jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wbs)).cpu().numpy()
        finally:
            e.close()
        assert np.isfinite(emb).all() and np.isfinite(cont).all()
        rg = max(rel(emb[b], refs[b][1]) for b in range(B))
        mg = max(float(np.abs(emb[b] - refs[b][1]).max()) for b in range(B))
        rf = max(rel(feats[b], refs[b][0]) for b in range(B))
        rc = max(rel(cont[b], cref[b]) for b in range(B))
        mc = float(np.abs(cont - cref).max())
        print(f"\n[{fam_id(family)}] {mname:13s} gesture rel-L2 {rg:.3e} max-abs {mg:.3e} | GestSync feats {rf:.3e} | content rel-L2 {rc:.3e} max-abs {mc:.3e}", end="")
        record(f"{fam_id(family)}/{mname}", gesture_rel=rg, gesture_maxabs=mg, feats_rel=rf, content_rel=rc, content_maxabs=mc)
        worst[mname] = max(rg, mg, rc, mc)
        outs[mname] = (emb, cont)
    # (1) the audit mode reproduces the fp32 oracle on every family
    assert worst["fp32_audit"] < aud_bound, (family, worst["fp32_audit"], aud_bound)
    # (2) the contract: the mode a driver selects for a checkpoint it has never seen (drivers.pick_precision) holds the plain 1e-3 on every
    #     family whose conditioning is in the regime
    from jegal_amd.drivers import REAL_CHECKPOINT_PRECISION
    sel = {L.PREC_FP16_W2: "w2", L.PREC_FP16_RC: "rc"}[REAL_CHECKPOINT_PRECISION]
    if in_regime:
        assert worst[sel] < TOL, (family, sel, worst)
    # (3) what `--audit` reports (default mode vs audit mode, no oracle) is the true error: the driver warns exactly when the contract breaks
    seen = max(max(rel(outs[sel][0][b], outs["fp32_audit"][0][b]) for b in range(B)), rel(outs[sel][1], outs["fp32_audit"][1]))
    true = max(max(rel(outs[sel][0][b], refs[b][1]) for b in range(B)), rel(outs[sel][1], cref))
    print(f"\n[{fam_id(family)}] --audit would report {seen:.3e} for the {sel} mode; true error vs the oracle {true:.3e}", end="")
    record(f"{fam_id(family)}/audit_report", reported=seen, true=true)
    assert abs(seen - true) < max(5e-5, 2 * aud_bound), (family, seen, true)
    if not in_regime:
        assert (seen >= TOL) == (true >= TOL), (family, seen, true)


@pytest.mark.parametrize("family", FAMILIES, ids=fam_id)
def test_xlmr_12_layers_across_weight_families(family):
    from jegal_amd._lib import Engine
    from jegal_amd.xlmr import XLMRoberta
    name, off = family
    sd = synth.xlmr_state_dict(seed=synth.XLMR_SEED + off, layers=12, family=name)
    ids, mask = synth.xlmr_inputs(55, 8, 48)
    m = torch.from_numpy(mask).bool()

    def xlmr_amp(w_sd, base=None):
        """relative change of the output per relative perturbation (1e-4, fp32 oracle) of the word embeddings"""
        sd2 = dict(w_sd)
        w = w_sd["embeddings.word_embeddings.weight"]
        sd2["embeddings.word_embeddings.weight"] = (w * (1 + 1e-4 * np.random.default_rng(7).standard_normal(w.shape))).astype(np.float32)
        with torch.no_grad():
            base = O.xlmr_forward(w_sd, ids, mask) if base is None else base
            return rel(O.xlmr_forward(sd2, ids, mask)[m].numpy(), base[m].numpy()) / 1e-4
    with torch.no_grad():
        ref = O.xlmr_forward(sd, ids, mask)
    if not _GAUSS_XLMR_AMP:
        _GAUSS_XLMR_AMP.append(xlmr_amp(synth.xlmr_state_dict(layers=12)))
    amp = xlmr_amp(sd, ref)
    ratio = amp / _GAUSS_XLMR_AMP[0]
    in_regime = ratio <= 3.0
    aud_bound = AUD_TOL * max(1.0, ratio / 3.0)
    print(f"\n[{fam_id(family)}] xlmr conditioning (d out / d embeddings): {amp:.2f} = {ratio:.1f} x the Gaussian draw's -> "
          f"{'in regime' if in_regime else 'OUTSIDE the regime: fp16 reported, audit asserted'}", end="")
    record(f"{fam_id(family)}/xlmr_conditioning", amplification=amp, ratio_to_gauss=ratio, in_regime=in_regime, audit_bound=aud_bound)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    errs, outs = {}, {}
    import jegal_amd._lib as L
    for mname in ("hi_lo", "bc_builtin_ids", "bc_own_ids", "fp32_audit"):
        e = Engine(0, precision=L.PREC_FP32 if mname == "fp32_audit" else None)
        try:
            x = XLMRoberta(engine=e).load_state_dict(sd)
            if mname == "bc_builtin_ids":
                x.calibrate()
            elif mname == "bc_own_ids":
                x.calibrate(ids_d, mask_d)
            out = x(ids_d, attention_mask=mask_d).last_hidden_state.cpu()
        finally:
            e.close()
        assert torch.isfinite(out).all()
        errs[mname] = rel(out[m].numpy(), ref[m].numpy())
        outs[mname] = out[m].numpy()
        mx = float((out[m] - ref[m]).abs().max())
        print(f"\n[{fam_id(family)}] xlmr {mname:15s} rel-L2 {errs[mname]:.3e} max-abs {mx:.3e}", end="")
        record(f"{fam_id(family)}/xlmr_{mname}", rel=errs[mname], maxabs=mx)
    assert errs["fp32_audit"] < aud_bound, (family, errs["fp32_audit"], aud_bound)
    if in_regime:
        assert errs["hi_lo"] < TOL, (family, errs)          # the calibration-free default
    seen = rel(outs["hi_lo"], outs["fp32_audit"])
    print(f"\n[{fam_id(family)}] xlmr: an audit of the default mode would report {seen:.3e}; true error {errs['hi_lo']:.3e}", end="")
    record(f"{fam_id(family)}/xlmr_audit_report", reported=seen, true=errs["hi_lo"])
    assert abs(seen - errs["hi_lo"]) < max(5e-5, 2 * aud_bound), (family, seen, errs)
