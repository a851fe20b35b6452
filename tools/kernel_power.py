"""Board power and shader clock while ONE kernel of the path runs back to back for a few seconds (rocm-smi, read-only):
is a kernel that leaves its matrix pipe idle a third of the time (SQ counters) stall-bound or power-bound?
  python tools/kernel_power.py          -> prints one line per kernel; copy into profiles/
Kernels: conv1_direct (32 masked clips; 32 dense clips), the 256x256 GEMM (qkv shape, random operands), the LN-fused GEMM (out_proj
shape), the S = 21 attention is inside the full step only.  Power cap: rocm-smi --showmaxpower."""
import json, os, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL

eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
eng.set_option("dual_stream", 0)


def sample(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            c = json.loads(out).get("card0", {})
            p = [float(v) for k, v in c.items() if "ower" in k and "W" in k]
            s = [v for k, v in c.items() if "sclk" in k]
            if p and s:
                rows.append((p[0], int("".join(ch for ch in s[0] if ch.isdigit()))))
        except Exception:
            pass
        time.sleep(0.15)


def probe(name, fn, seconds=4.0, work=None):
    fn(); torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows), daemon=True)
    t0 = time.perf_counter(); n = 0
    th.start()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    rows = rows[len(rows) // 4:]                   # drop the ramp
    pw = np.array([r[0] for r in rows]); ck = np.array([r[1] for r in rows])
    extra = f", {work / (dt / n) / 1e12:.0f} TFLOP/s" if work else ""
    print(f"{name:58s} {dt / n * 1e3:8.3f} ms per call{extra}; power {pw.mean():6.0f} W (max {pw.max():.0f}), sclk {ck.mean():5.0f} MHz (min {ck.min()}) over {len(rows)} samples", flush=True)


masked = torch.from_numpy(synth.synth_frames(1234, 32, 150)).cuda()
dense = torch.randint(1, 256, masked.shape, dtype=torch.uint8, device="cuda")
conv1_flop = 32 * 154 * 13904 * 64 * 735 * 2
probe("conv1_direct + scan + edge fix, 32 masked clips", lambda: eng.debug_conv1_pool(masked, 4), work=conv1_flop * 14 / 22)
probe("conv1_direct + scan + edge fix, 32 dense clips", lambda: eng.debug_conv1_pool(dense, 4), work=conv1_flop)
del dense
M = 100800
a = (torch.rand((M, 512), device="cuda") - 0.5).half(); w = (torch.rand((1536, 512), device="cuda") - 0.5).half()
probe("gemm 256x256 tile, 100800 x 512 -> 1536 (qkv), random data", lambda: eng.debug_gemm(M, 1536, 512, 0, 10, a, w), work=10 * 2 * M * 512 * 1536)
w2 = (torch.rand((512, 512), device="cuda") - 0.5).half()
probe("gemm LN-fused 128x512 tile, 100800 x 512 -> 512 (out_proj)", lambda: eng.debug_gemm(M, 512, 512, 8, 10, a, w2), work=10 * 2 * M * 512 * 512)
out = torch.empty((32, 150, 512), dtype=torch.float32, device="cuda")
probe("full step, one stream (32 clips)", lambda: eng.extract_gesture(masked, out))
eng.set_option("dual_stream", 1)
probe("full step, two lanes (32 clips)", lambda: eng.extract_gesture(masked, out))
idle = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
print([ln for ln in idle.splitlines() if "Max" in ln])
