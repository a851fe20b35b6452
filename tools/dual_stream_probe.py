"""Probe: does running two independent half-batches on two HIP streams (two engine handles) beat one full batch per step?
The persistent GEMM / conv kernels run in rounds of one tile per CU; the last round of every launch is partly empty
(788 LN tiles on 256 CUs = 3.08 rounds).  Two streams let one half's next kernel fill the other half's tail."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL

def make():
    e = Engine(0)
    GestSync(engine=e).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=e).load_state_dict(synth.jegal_state_dict())
    e.set_chunk(32)
    return e

frames = torch.from_numpy(synth.synth_frames(1234, 32, 150)).cuda()
out = torch.empty(32, 150, 512, device="cuda")
e1 = make()
def run_single(n):
    for _ in range(n):
        e1.extract_gesture(frames, out)
for _ in range(5): run_single(1)
torch.cuda.synchronize(); t0 = time.perf_counter(); run_single(100); torch.cuda.synchronize()
t_single = (time.perf_counter() - t0) / 100
print("one stream, 32 clips per call: %.3f ms per 32 clips" % (t_single * 1e3))
ref = out.clone()

for split in ((16, 16), (16, 16), (14, 18), (12, 20), (20, 12), (12, 20)):
    e2 = make()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a, b = split
    def run_dual(n):
        for _ in range(n):
            with torch.cuda.stream(s1):
                e1.extract_gesture(frames[:a], out[:a])
            with torch.cuda.stream(s2):
                e2.extract_gesture(frames[a:], out[a:])
    out.zero_()
    for _ in range(5): run_dual(1)
    torch.cuda.synchronize(); t0 = time.perf_counter(); run_dual(100); torch.cuda.synchronize()
    t_dual = (time.perf_counter() - t0) / 100
    print("two streams, %d + %d clips: %.3f ms per 32 clips (%.1f %%), identical: %s" % (a, b, t_dual * 1e3, 100 * (t_dual / t_single - 1), torch.equal(out, ref)))
    e2.close()
