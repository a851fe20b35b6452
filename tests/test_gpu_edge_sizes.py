"""Degenerate sizes through the C ABI against the fp32 restatement (run with -m gpu): one- and two-frame clips (every window is
edge padding), clips just below / at / above the 25-frame window, single-token / single-word content inputs, the shortest audio the
CNN accepts, one-clip metric calls.  The reference handles all of these (its loops are per clip); a batched engine has to as well."""
import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def stack():
    from jegal_amd._lib import Engine
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    eng = Engine(0)
    gsd, jsd = synth.gestsync_state_dict(include_unused=False), synth.jegal_state_dict()
    GestSync(engine=eng).load_state_dict(gsd)
    jg = JEGAL(engine=eng).load_state_dict(jsd)
    yield eng, jg, O.tensors(gsd), O.tensors(jsd)
    eng.close()


@pytest.mark.parametrize("B,T", [(1, 1), (2, 2), (1, 24), (3, 25), (1, 26), (2, 37)])
def test_short_clips_vs_oracle(stack, B, T):
    eng, jg, gsd, jsd = stack
    frames = synth.synth_frames(900 + T, B, T)
    emb = eng.extract_gesture(torch.from_numpy(frames).cuda()).cpu()
    worst = 0.0
    for b in range(B):
        with torch.no_grad():
            f = O.gestsync_clip_feats(gsd, torch.from_numpy(frames[b]).float() / 255.0)
            ref = O.l2_normalize(O.jegal_forward_inference(jsd, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
        worst = max(worst, rel(emb[b], ref))
    print(f"B={B} T={T}: gesture embedding rel-L2 vs oracle {worst:.3e}")
    assert torch.isfinite(emb).all() and worst < TOL


def test_minimal_content_inputs_vs_oracle(stack):
    eng, jg, gsd, jsd = stack
    # one word, one sub-word token between <s> and </s>; the shortest mel the audio CNN takes for that word (4 frames -> 1 step)
    states, tmask, ids, offs = synth.synth_text(77, 1, 1)
    wb = [[["w0", 0, 0]]]
    mel = synth.synth_mel(78, 1, 4)
    pack = (torch.from_numpy(states), torch.from_numpy(tmask), [["w0"]], ids, offs)
    with torch.no_grad():
        ref_t = O.jegal_forward_inference(jsd, text=pack)
        ref_a = O.jegal_forward_inference(jsd, audio=torch.from_numpy(mel), word_boundaries=wb)
        ref_ta = O.jegal_forward_inference(jsd, text=pack, audio=torch.from_numpy(mel), word_boundaries=wb)
    got_t = jg.forward_inference(text=pack)
    got_a = jg.forward_inference(audio=torch.from_numpy(mel), word_boundaries=wb)
    got_ta = jg.forward_inference(text=pack, audio=torch.from_numpy(mel), word_boundaries=wb)
    for name, g, r in (("t", got_t, ref_t), ("a", got_a, ref_a), ("ta", got_ta, ref_ta)):
        e = rel(eng.l2norm(g.reshape(-1, 512)), O.l2_normalize(r.reshape(-1, 512)))
        print(f"one-word clip, modalities {name}: content embedding rel-L2 vs oracle {e:.3e}")
        assert tuple(g.shape) == tuple(r.shape) == (1, 1, 512) and e < TOL


def test_single_item_metric_calls(stack):
    eng = stack[0]
    from jegal_amd import metrics as M
    rng = np.random.default_rng(5)
    g = rng.standard_normal((1, 512)).astype(np.float32)
    c = rng.standard_normal((1, 512)).astype(np.float32)
    m = M.retrieval_metrics(torch.from_numpy(c).cuda(), torch.from_numpy(g).cuda(), engine=eng)      # a gallery of one: rank 0 whatever the vectors
    assert m["R1"] == 1.0 and m["MR"] == 1.0
    # one clip, one word, one frame: the softmax over a single frame is 1 -> correct iff the frame lies in the window
    ge, ce = [g / np.linalg.norm(g)], [c / np.linalg.norm(c)]
    acc = M.spotting_accuracy(ge, ce, [[["w", 0, 0]]], [0], engine=eng)
    ref = O.spotting_accuracy([torch.from_numpy(ge[0])], [torch.from_numpy(ce[0])], [[["w", 0, 0]]], [0])
    assert acc == ref == 100.0
