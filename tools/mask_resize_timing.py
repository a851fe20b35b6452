"""Time jg_mask_resize / jg_mask_resize_packed on one 32-clip batch of decoder-resolution frames (the kernel runs on the upload
stream of GestureStreamer(source_hw=...), next to the extraction: its duration is CU time taken from the compute).
  python tools/mask_resize_timing.py [H W]      (JG_MASK_RESIZE_GENERIC=1: the per-pixel kernel)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd._lib import Engine
from jegal_amd.extract import _SourcePacker

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (228, 314)
eng = Engine(0)
F = 32 * 150
rng = np.random.default_rng(0)
src = torch.from_numpy(rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)).cuda()
my = torch.full((F,), int(round(109 * H / 270.0)), dtype=torch.int32, device="cuda")
dst = torch.empty((F, 270, 480, 3), dtype=torch.uint8, device="cuda")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_full = timed(lambda: eng.mask_resize(src, my))          # (includes the wrapper's output allocation)
row0 = int(my[0]) + 1
packed = src[:, row0:].contiguous().reshape(-1)
offs = (torch.arange(F, dtype=torch.int64) * (H - row0) * W * 3).cuda()
ref = dst.clone()
t_packed = timed(lambda: eng.mask_resize_packed(packed, offs, my, H, W, dst))
assert torch.equal(ref, dst)
print(f"{H}x{W} -> 270x480, {F} frames: full frames {t_full:.3f} ms, packed (rows below the mask) {t_packed:.3f} ms; "
      f"bytes read {packed.numel() / 1e6:.0f} MB + written {dst.numel() / 1e6:.0f} MB = {(packed.numel() + dst.numel()) / t_packed / 1e6:.0f} GB/s")
