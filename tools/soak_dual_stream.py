"""Soak test for the two-lane path: many back-to-back 32-clip steps on the caller's stream, outputs compared bit for bit with the
one-stream result at intervals (no host synchronisation in between), different inputs alternating so that a stale read would show."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
eng.set_chunk(32)
fa = torch.from_numpy(synth.synth_frames(1, 32, 150)).cuda()
fb = torch.from_numpy(synth.synth_frames(2, 32, 150)).cuda()
eng.set_option("dual_stream", 0)
ra, rb = eng.extract_gesture(fa).clone(), eng.extract_gesture(fb).clone()
eng.set_option("dual_stream", 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
outs = [torch.empty_like(ra) for _ in range(4)]
bad = 0
t0 = time.perf_counter()
for i in range(n):
    src, ref = (fa, ra) if i % 3 else (fb, rb)
    out = outs[i % 4]
    eng.extract_gesture(src, out)
    if i % 97 == 0:
        bad += int(not torch.equal(out, ref))          # torch.equal runs on the same stream, right behind the call
torch.cuda.synchronize()
print("%d steps in %.1f s, %d mismatches" % (n, time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
