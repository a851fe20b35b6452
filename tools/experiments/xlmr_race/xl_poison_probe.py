"""Debug probe: run-to-run differences of XLM-RoBERTa outputs (lanes, folding, poisoned workspace).  Runs on the library as committed; the
variants that need debug_scaffolding.patch are probe2-4.
python tools/xl_poison_probe.py lanes fold [calibrate B L]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
lanes, fold = int(sys.argv[1]), int(sys.argv[2])
cal = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 24
L = int(sys.argv[5]) if len(sys.argv) > 5 else 40
ids, mask = synth.xlmr_inputs(3, B, L)
eng = Engine(0)
eng.set_option("xlmr_lanes", lanes)
eng.set_option("xlmr_fold", fold)
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
xl = XLMRoberta(engine=eng).load_state_dict(synth.xlmr_state_dict(layers=2))
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
if os.environ.get("USER_STREAM"):           # not the null stream: the handle follows torch's current stream
    _s = torch.cuda.Stream(); torch.cuda.synchronize(); torch.cuda.set_stream(_s)
if cal:
    xl.calibrate(ids_d, mask_d)
base = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
n = 0
seqs = set()
eng.set_option("ws_poison", 1)
for it in range(150):
    eng.set_option("gemm_tile", it & 3)
    out = xl(ids_d, attention_mask=mask_d).last_hidden_state
    d = (out - base).abs().amax(-1).cpu().numpy()
    bad = np.argwhere(d > 0)
    if len(bad):
        n += 1
        seqs |= set(int(b) for b, _ in bad)
print(f"opts {os.environ.get('OPTS')} lanes {lanes} fold {fold} calibrated {cal} B {B} L {L}: runs with differences: {n} of 150; sequences {sorted(seqs)}")
# with the dual-accumulation scaffolding (debug_scaffolding_dual_acc.patch): two accumulator sets fed by the same fragments, compared behind the k loop
try:
    import ctypes
    fn = eng.lib.jg_canary_read
    buf = (ctypes.c_uint * 4096)()
    fn(buf, 0)
    print(f"dual accumulation: accumulator quads that differ between the two sets {buf[0]}, tiles checked {buf[1]}")
    for k in range(min(buf[0], 40)):
        r = buf[8 + 8 * k: 16 + 8 * k]
        f = lambda u: np.array([u], np.uint32).view(np.float32)[0]
        print(f"  wg {r[0]} wave {r[1] >> 16} lane {r[1] >> 8 & 255} i {r[1] >> 4 & 15} j {r[1] & 15} tile m0 {r[7] >> 12} nblk {r[7] & 4095} "
              f"differing elements mask {r[6]:04b}: acc.x {f(r[2]):.5f} acc2.x {f(r[4]):.5f} acc.z {f(r[3]):.5f} acc2.z {f(r[5]):.5f}")
except AttributeError:
    pass
