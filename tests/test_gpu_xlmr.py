"""XLM-RoBERTa text front end on the HIP engine, through the C ABI (jg_xlmr_encode).

Reference call site: models/jegal.py:116-129 (``mroberta(input_ids, attention_mask=text_mask).last_hidden_state``).  The model is
third-party (transformers); parity is pinned against transformers.XLMRobertaModel with seeded weights (tests/golden/xlmr.npz)."""
import os

import numpy as np
import pytest
import torch

import jegal_oracle as O
from jegal_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3        # relative L2 over the valid token rows (rows have unit-order LayerNorm scale)


def rel(a, b):
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def xlmr():
    from jegal_amd._lib import Engine
    from jegal_amd.xlmr import XLMRoberta
    eng = Engine(0)
    m = XLMRoberta(engine=eng).load_state_dict({"roberta." + k: v for k, v in synth.xlmr_state_dict().items()})
    yield m
    eng.close()


def test_xlmr_matches_transformers_golden(xlmr, golden_dir):
    g = np.load(os.path.join(golden_dir, "xlmr.npz"))
    out = xlmr(torch.from_numpy(g["input_ids"]), attention_mask=torch.from_numpy(g["attention_mask"])).last_hidden_state.cpu()
    ref = torch.from_numpy(g["last_hidden_state"])
    m = torch.from_numpy(g["attention_mask"]).bool()
    e = rel(out[m], ref[m])
    print("xlmr vs transformers golden: rel-L2 %.3e, max-abs %.3e" % (e, float((out[m] - ref[m]).abs().max())))
    assert e < TOL
    out1 = xlmr(torch.from_numpy(g["input_ids"][:1])).last_hidden_state.cpu()
    assert rel(out1, torch.from_numpy(g["last_hidden_state_nomask"])) < TOL


@pytest.mark.parametrize("B,L", [(1, 3), (2, 33), (5, 70), (2, 200), (1, 512)])
def test_xlmr_lengths_vs_oracle(xlmr, B, L):
    """Short, odd and long sequences (MFMA attention up to 160 tokens, the VALU kernel beyond; 512 = the position table's limit),
    ragged padding, against the fp32 restatement."""
    ids, mask = synth.xlmr_inputs(100 + L, B, L)
    sd = synth.xlmr_state_dict()
    with torch.no_grad():
        ref = O.xlmr_forward(sd, ids, mask)
    out = xlmr(torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda()).last_hidden_state.cpu()
    m = torch.from_numpy(mask).bool()
    e = rel(out[m], ref[m])
    print(f"B={B} L={L}: rel-L2 {e:.3e}")
    assert torch.isfinite(out).all() and e < TOL


def test_xlmr_properties_and_errors(xlmr):
    ids, mask = synth.xlmr_inputs(7, 4, 40)
    a = xlmr(ids, attention_mask=mask).last_hidden_state
    b = xlmr(ids, attention_mask=mask).last_hidden_state
    assert torch.equal(a, b)                                         # deterministic
    # a sample's valid rows do not depend on what else is in the batch, nor on how much padding follows
    one = xlmr(ids[2:3], attention_mask=mask[2:3]).last_hidden_state
    n = int(mask[2].sum())
    assert rel(one[0, :n].cpu(), a[2, :n].cpu()) < 1e-5
    with pytest.raises(ValueError):
        xlmr(ids[0])                                                  # not (B, L)
    with pytest.raises(ValueError):
        xlmr(ids, attention_mask=mask[:, :10])
    with pytest.raises(RuntimeError):
        xlmr(np.ones((1, 600), np.int32))                             # beyond the position table
