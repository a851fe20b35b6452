"""Where a streamed batch's time goes (GestureStreamer(source_hw=...), packers pre-filled): per batch, on one clock, the H2D copies
and the compute call (mask + resize kernel included), from torch events -- is the steady state bound by the link, the compute, or a gap?
Usage: python tools/stream_timeline.py [batches]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
from jegal_amd.extract import GestureStreamer

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
MARKS = "--no-marks" not in sys.argv
_crowd = []
for _ in range(int(os.environ.get("EXTRA_STREAMS", "0"))):          # other (used) streams of the application, created before the engine
    _s = torch.cuda.Stream()
    with torch.cuda.stream(_s):
        torch.zeros(1, device="cuda").add_(1)
    _crowd.append(_s)
torch.cuda.synchronize()
eng = Engine(0); eng.set_chunk(32)
for o in sys.argv[2:]:
    if "=" in o:
        k, v = o.split("="); eng.set_option(k, int(v))
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
SH, SW, B, T = 228, 314, 32, 150
src = np.random.default_rng(4321).integers(0, 256, (B, T, SH, SW, 3), dtype=np.uint8)
my = int(round(109 * SH / 270.0))


class Probe(GestureStreamer):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.marks = []

    def _upload(self, slot, n):
        if not MARKS:
            return super()._upload(slot, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.copy)
        super()._upload(slot, n)
        e1.record(self.copy)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.marks.append((e0, e1, c0, c1))
        self._c = (c0, c1)


st = Probe(eng, B, T, source_hw=(SH, SW))
_extract = eng.extract_gesture


def extract(*a, **k):                     # the compute call, bracketed on the compute stream
    if not MARKS:
        return _extract(*a, **k)
    c0, c1 = st._c
    c0.record(torch.cuda.current_stream())
    r = _extract(*a, **k)
    c1.record(torch.cuda.current_stream())
    return r


eng.extract_gesture = extract
for s_ in range(2):
    for b in range(B):
        st.packer[s_].add(src[b], my)
for _ in st.run_filled(lambda pk, k: B if k < 3 else 0):
    pass
torch.cuda.synchronize()
st.marks.clear()
t0 = time.perf_counter()
got = 0
for _, emb in st.run_filled(lambda pk, k: B if k < nb else 0):
    got += emb.shape[0]
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{got} clips in {dt * 1e3:.1f} ms: {got / dt:.1f} clips/s, {dt / nb * 1e3:.2f} ms per batch; bytes per batch {st.packer[0].used / 1e6:.0f} MB")
if not MARKS:
    sys.exit(0)
base = st.marks[0][0]
for i, (e0, e1, c0, c1) in enumerate(st.marks):
    print(f"batch {i}: upload {base.elapsed_time(e0):7.2f} .. {base.elapsed_time(e1):7.2f} ms ({e0.elapsed_time(e1):5.2f}),"
          f" compute {base.elapsed_time(c0):7.2f} .. {base.elapsed_time(c1):7.2f} ms ({c0.elapsed_time(c1):5.2f})")
