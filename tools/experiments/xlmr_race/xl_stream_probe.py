"""Debug probe 7: does the two-lane difference depend on the stream the handle is bound to?  For the null stream and for the n-th user stream
(n = 1..4: HIP deals streams onto its hardware queues in creation order, so n shifts which queues the lane streams share): first whether the
lanes overlap at all (time of 1 lane vs 2 lanes at B = 256, L = 64), then the poisoned-workspace repeat test at B = 64, L = 32.
python tools/experiments/xlmr_race/xl_stream_probe.py"""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
sd = synth.xlmr_state_dict(layers=2)
big = [torch.from_numpy(a).cuda() for a in synth.xlmr_inputs(5, 256, 64)]
small = [torch.from_numpy(a).cuda() for a in synth.xlmr_inputs(3, 64, 32)]
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
keep = []
for n in (0, 1, 2, 3, 4, 0):
    torch.cuda.synchronize()
    if n:
        for _ in range(n):
            keep.append(torch.cuda.Stream())
        torch.cuda.set_stream(keep[-1])
    else:
        torch.cuda.set_stream(torch.cuda.default_stream())
    eng = Engine(0)            # a fresh handle: its lane streams are created at the first two-lane call, AFTER the user streams above
    xl = XLMRoberta(engine=eng).load_state_dict(sd)
    t = {}
    for lanes in (1, 2):
        eng.set_option("xlmr_lanes", lanes)
        for _ in range(5):
            xl(big[0], attention_mask=big[1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            xl(big[0], attention_mask=big[1])
        torch.cuda.synchronize(); t[lanes] = (time.perf_counter() - t0) / 30 * 1e3
    eng.set_option("xlmr_lanes", 1)
    one = xl(small[0], attention_mask=small[1]).last_hidden_state.clone()
    eng.set_option("xlmr_lanes", 2)
    eng.set_option("ws_poison", 1)
    bad = 0
    for it in range(RUNS):
        eng.set_option("gemm_tile", it & 3)
        if not torch.equal(xl(small[0], attention_mask=small[1]).last_hidden_state, one):
            bad += 1
    print(f"{'null stream' if not n else f'user stream #{len(keep)}'}: 1 lane {t[1]:.3f} ms, 2 lanes {t[2]:.3f} ms ({'overlap' if t[2] < 0.97 * t[1] else 'NO overlap'}); "
          f"runs that differ from the one-lane result: {bad} of {RUNS}", flush=True)
    eng.close()
