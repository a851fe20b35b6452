"""In-kernel shader clock of the MFMA kernels under sustained load (MI355X_MICROARCH.md, "DVFS give-back" item 6: the guide's
instrument for a power-bound claim -- board power and pp_dpm_sclk are not).

A DIAGNOSTIC build of the library (`make -C jegal_amd/csrc diag` -> tools/bin/libjegal_hip_diag.so, -DJG_CLOCK_STAMPS) stamps
s_memtime (shader cycles) and s_memrealtime (100 MHz) around the main loop of conv1_direct_kernel's MFMA waves and around the
persistent loop of gemm_glds_kernel; the stamps go to a buffer of their own.  Each kernel runs back to back for >= 2.5 s on the bench
inputs (random operands for the GEMMs), then:   clock = median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz.
The product library carries no stamp.

    python tools/inkernel_clock.py [--opt name=int ...]      -> one line per kernel; copy into profiles/
"""
import ctypes, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tools", "bin", "libjegal_hip_diag.so")
if not os.path.exists(DIAG):
    subprocess.run(["make", "-C", os.path.join(ROOT, "jegal_amd", "csrc"), "diag", "-j8"], check=True)
import jegal_amd._lib as L
L.LIB_PATH = DIAG
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL

eng = Engine(0)
lib = L.load_library()
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.lstrip("-").replace("opt ", "").split("=")
        eng.set_option(k, int(v))
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
eng.set_option("dual_stream", 0)


def clock(reader):
    buf = (ctypes.c_ulonglong * 2048)()
    assert getattr(lib, reader)(buf, 2048) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
    a = a[(a[:, 1] > 0) & (a[:, 0] > 0)]
    ghz = a[:, 0] / a[:, 1] * 0.1
    return float(np.median(ghz)), float(ghz.min()), float(ghz.max()), len(ghz), float(np.median(a[:, 1]) * 0.01)


def probe(name, fn, reader, seconds=2.5, work=None):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = (time.perf_counter() - t0) / n
    med, lo, hi, wgs, loop_us = clock(reader)
    extra = f", {work / dt / 1e12:.0f} TFLOP/s" if work else ""
    print(f"{name:62s} {dt * 1e3:8.3f} ms per call{extra}; in-kernel clock {med:.3f} GHz (min {lo:.3f}, max {hi:.3f} over {wgs} workgroups; "
          f"stamped loop {loop_us:.0f} us)", flush=True)


masked = torch.from_numpy(synth.synth_frames(1234, 32, 150)).cuda()
dense = torch.randint(1, 256, masked.shape, dtype=torch.uint8, device="cuda")
conv1_flop = 32 * 154 * 13904 * 64 * 735 * 2
probe("conv1_direct (+ scan + edge fix), 32 masked clips", lambda: eng.debug_conv1_pool(masked, 4), "jg_clock_read_conv1", work=conv1_flop * 14 / 22)
probe("conv1_direct (+ scan + edge fix), 32 dense clips", lambda: eng.debug_conv1_pool(dense, 4), "jg_clock_read_conv1", work=conv1_flop)
smooth = torch.from_numpy(synth.synth_frames_structured(4100, 32, 150, "smooth")).cuda()
probe("conv1_direct, 32 SMOOTH masked clips (low-contrast blobs + gradients, +-2 of pixel noise)", lambda: eng.debug_conv1_pool(smooth, 4), "jg_clock_read_conv1",
      work=conv1_flop * 14 / 22)
del smooth
zeros = torch.zeros_like(masked)
probe("conv1_direct, all-zero frames (every tile skipped: idle reference)", lambda: eng.debug_conv1_pool(zeros, 4), "jg_clock_read_conv1")
del dense, zeros
M = 100800
a = (torch.rand((M, 512), device="cuda") - 0.5).half(); w = (torch.rand((1536, 512), device="cuda") - 0.5).half()
probe("gemm 256x256 tile, 100800 x 512 -> 1536 (qkv), random data", lambda: eng.debug_gemm(M, 1536, 512, 0, 10, a, w), "jg_clock_read_gemm", work=10 * 2 * M * 512 * 1536)
az, wz = torch.zeros_like(a), torch.zeros_like(w)
probe("gemm 256x256 tile, same shape, all-zero operands", lambda: eng.debug_gemm(M, 1536, 512, 0, 10, az, wz), "jg_clock_read_gemm", work=10 * 2 * M * 512 * 1536)
w2 = (torch.rand((512, 512), device="cuda") - 0.5).half()
probe("gemm LN-fused 128x512 tile, 100800 x 512 -> 512 (out_proj), random", lambda: eng.debug_gemm(M, 512, 512, 8, 10, a, w2), "jg_clock_read_gemm", work=10 * 2 * M * 512 * 512)
