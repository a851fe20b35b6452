"""Do the two lanes overlap?  Time per encode with 1 and 2 lanes, the handle bound to the null stream or to a user stream.
python tools/experiments/xlmr_race/xl_lane_overlap.py [B L]"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ids, mask = synth.xlmr_inputs(3, B, L)
eng = Engine(0)
xl = XLMRoberta(engine=eng).load_state_dict(synth.xlmr_state_dict(layers=4))
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
for user in (0, 1, 0, 1):
    if user:
        s = torch.cuda.Stream(); torch.cuda.synchronize(); torch.cuda.set_stream(s)
    else:
        torch.cuda.synchronize(); torch.cuda.set_stream(torch.cuda.default_stream())
    for lanes in (1, 2):
        eng.set_option("xlmr_lanes", lanes)
        for _ in range(5):
            xl(ids_d, attention_mask=mask_d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            xl(ids_d, attention_mask=mask_d)
        torch.cuda.synchronize()
        print(f"{'user stream' if user else 'null stream'} lanes {lanes}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per encode (B {B} L {L}, 4 layers)")
