"""PCIe-inclusive throughput: clips resident in (pageable) host memory -> GestureStreamer -> embeddings on the host.
Reported next to bench.py's HBM-resident figure in DESIGN.md section 7; never the bench `value`."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
from jegal_amd.extract import GestureStreamer

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = Engine(0); eng.set_chunk(32)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
clips = synth.synth_frames(1234, 32, 150)                  # 32 distinct clips, cycled
st = GestureStreamer(eng, 32, 150)
def gen(n):
    for i in range(n):
        yield clips[i % 32]
for _ in st.run(gen(64)):                                  # warm-up
    pass
torch.cuda.synchronize()
t0 = time.perf_counter(); got = 0
for first, emb in st.run(gen(32 * nb)):
    got += emb.shape[0]
dt = time.perf_counter() - t0
print(f"streamed {got} clips in {dt:.3f} s: {got / dt:.1f} clips/s incl. host packing (memcpy into pinned memory) + H2D + D2H")
# producer writes straight into the pinned buffers (here: pre-filled) -> PCIe + compute overlap only
for s_ in range(2):
    st.h_in[s_].numpy()[:] = clips
torch.cuda.synchronize()
t0 = time.perf_counter(); got = 0
for first, emb in st.run_filled(lambda buf, k: 32 if k < nb else 0):
    got += emb.shape[0]
dt = time.perf_counter() - t0
ref = eng.extract_gesture(torch.from_numpy(clips).cuda()).cpu().numpy()
assert np.array_equal(emb, ref), "streamed embeddings differ from the resident path"
print(f"pinned-producer stream: {got} clips in {dt:.3f} s: {got / dt:.1f} clips/s incl. H2D + D2H "
      f"({got * 150 * 270 * 480 * 3 / dt / 1e9:.1f} GB/s of frames)")
t0 = time.perf_counter()
for i in range(3):
    st.d_in[0].copy_(st.h_in[0], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"pinned H2D alone: {st.h_in[0].numel() / dt / 1e9:.1f} GB/s -> {32 / dt:.0f} clips/s ceiling")
