"""Same-box yardstick for the path's Linear GEMM shapes (TOOLS ONLY, never the product path).

For every shape of the GestSync transformer / JEGAL branch: the vendor fp16 GEMM behind torch.matmul (hipBLASLt / rocBLAS)
next to the library's own kernel (jg_debug_gemm_ex), on the SAME random operands, interleaved rounds in one process
(cdna_hip_programming.md rules 24 / 25).  Prints TFLOP/s and the ratio.  The vendor number is a yardstick for "is 1.07 PFLOP/s
the chip or the kernel", nothing in jegal_amd calls it.

Usage: python tools/gemm_yardstick.py [--rounds 5] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd._lib import Engine  # noqa: E402

SHAPES = [  # (M, K, N, what, debug_gemm mode)
    (100800, 512, 1536, "GestSync qkv", 0),
    (100800, 512, 512, "GestSync out_proj (+res+LN fused in the library)", 8),
    (100800, 512, 2048, "GestSync linear1 (+ReLU)", 4),
    (100800, 2048, 512, "GestSync linear2 (+res+LN fused in the library)", 8),
    (100800, 512, 1024, "M=100800 512->1024", 0),
    (4800, 512, 512, "JEGAL out / align", 0),
    (4800, 512, 1536, "JEGAL qkv", 0),
    (4800, 512, 2048, "JEGAL ff1", 4),
    (4800, 2048, 512, "JEGAL ff2", 2),
]


def time_torch(a, w, iters, relu):
    out = torch.empty((a.shape[0], w.shape[0]), dtype=torch.float16, device=a.device)
    torch.matmul(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, w.t(), out=out)
        if relu:
            out.relu_()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    eng = Engine(0)
    dev = eng.device
    torch.manual_seed(0)
    rows = []
    for M, K, N, what, mode in SHAPES:
        a = (torch.randn((M, K), device=dev) * 0.5).half()
        w = (torch.randn((N, K), device=dev) * 0.05).half()
        tv, tl, tlp = [], [], []
        for _ in range(args.rounds):
            tv.append(time_torch(a, w, args.iters, False))
            tl.append(eng.debug_gemm(M, N, K, mode=mode, iters=args.iters, a16=a, w16=w))
            tlp.append(eng.debug_gemm(M, N, K, mode=mode & ~8, iters=args.iters, a16=a, w16=w) if mode & 8 else tl[-1])
        gf = 2.0 * M * N * K / 1e9
        med = lambda v: sorted(v)[len(v) // 2]
        r = {"shape": f"{M}x{K}->{N}", "what": what, "gflop": gf, "vendor_ms": med(tv), "lib_ms": med(tl), "lib_plain_ms": med(tlp),
             "vendor_tflops": gf / med(tv), "lib_tflops": gf / med(tl), "lib_plain_tflops": gf / med(tlp)}
        rows.append(r)
        print(f"{r['shape']:>20s}  vendor {r['vendor_ms']*1e3:8.1f} us {r['vendor_tflops']:7.0f} TF | library {r['lib_ms']*1e3:8.1f} us {r['lib_tflops']:7.0f} TF"
              f" (plain epilogue {r['lib_plain_ms']*1e3:8.1f} us {r['lib_plain_tflops']:7.0f} TF) | lib/vendor time {r['lib_ms']/r['vendor_ms']:.2f}  {what}", flush=True)
        del a, w
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
