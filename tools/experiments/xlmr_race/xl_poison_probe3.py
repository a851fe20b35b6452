"""Debug probe 3: group the outputs of N poisoned runs by equality (is the un-poisoned first run the odd one, or do the runs vary?)."""
import sys, os, hashlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import jegal_oracle as O
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
B, L = 64, 32
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ids, mask = synth.xlmr_inputs(3, B, L)
sd = synth.xlmr_state_dict(layers=2)
eng = Engine(0)
eng.set_option("xlmr_lanes", lanes)
xl = XLMRoberta(engine=eng).load_state_dict(sd)
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
with torch.no_grad():
    ref = O.xlmr_forward(sd, ids, mask)
m = torch.from_numpy(mask).bool()
def rel(a, b):
    return float((a - b).norm() / b.norm())
first = xl(ids_d, attention_mask=mask_d).last_hidden_state.cpu()
print("first (un-poisoned) run vs oracle on valid rows: rel-L2 %.3e" % rel(first[m], ref[m]))
eng.set_option("ws_poison", 1)
groups = {}
for it in range(150):
    out = xl(ids_d, attention_mask=mask_d).last_hidden_state.cpu()
    key = hashlib.md5(out.numpy().tobytes()).hexdigest()
    g = groups.setdefault(key, [0, out])
    g[0] += 1
order = sorted(groups.values(), key=lambda g: -g[0])
print("distinct outputs over 150 poisoned runs:", len(order), "sizes", [g[0] for g in order][:10])
maj = order[0][1]
print("majority == first run:", bool(torch.equal(maj, first)), "| majority vs oracle %.3e" % rel(maj[m], ref[m]))
for g in order[1:4]:
    d = (g[1] - maj).abs().amax(-1)
    bad = torch.nonzero(d > 0)
    seqs = sorted(set(int(b) for b in bad[:, 0]))
    print("  minority (x%d): sequences %s, valid-row error vs oracle %.3e" % (g[0], seqs[:12], rel(g[1][m], ref[m])))
import ctypes
eng.lib.jg_debug_counter.restype = ctypes.c_int64
eng.lib.jg_debug_counter.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("debug counters:", [eng.lib.jg_debug_counter(eng.h, i) for i in range(8)])
