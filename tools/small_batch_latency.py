"""Latency of jg_extract_gesture for small batches (host-launch-bound regime): wall time per call back to back, and with a host
synchronisation after every call (what a one-clip-at-a-time caller such as inference_embs.py sees).  Usage: python tools/small_batch_latency.py [opt=val ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
eng = Engine(0)
for o in sys.argv[1:]:
    k, v = o.split("="); eng.set_option(k, int(v))
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
for B, T in ((1, 25), (1, 150), (2, 150), (4, 150), (8, 150), (32, 150)):
    fr = torch.from_numpy(synth.synth_frames(1234, B, T)).cuda()
    out = torch.empty((B, T, 512), dtype=torch.float32, device="cuda")
    for _ in range(5):
        eng.extract_gesture(fr, out)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        eng.extract_gesture(fr, out)
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_b2b = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        eng.extract_gesture(fr, out)
        torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / n
    print(f"B={B:2d} T={T:3d}: host issue {t_issue * 1e3:6.3f} ms/call, back to back {t_b2b * 1e3:6.3f} ms/call ({B / t_b2b:7.1f} clips/s), synchronised {t_sync * 1e3:6.3f} ms/call", flush=True)
