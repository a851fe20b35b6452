"""Packed-fp32 instructions with operand selects in libjegal_hip.so, per kernel.

Round 6 found (tools/experiments/pk_opsel_mfma/repro.hip, stand-alone) that on MI355X a `v_pk_fma_f32 ... op_sel:[0,1,0]` can return its
low half WITHOUT the product (result = src2.lo) in lanes 48-63 while waves of another kernel issue MFMAs on the same SIMD; the plain form
(no op_sel / op_sel_hi) never did.  hipcc forms the select variants whenever one half of a 64-bit register pair is broadcast into a
packed multiply (`f32x4 * pair.y`).  This scan disassembles every gfx950 code object embedded in the library and lists, per kernel, the
v_pk_{fma,mul,add}_f32 instructions that carry op_sel / op_sel_hi.  tests/test_host_cpu.py keeps the count at 0.

    python tools/packed_opsel_scan.py [substring] [-v]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PAT = re.compile(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel(_hi)?:")


def scan(lib=None):
    """{demangled kernel name: [disassembly lines]} of the select-carrying packed fp32 instructions (kernels without any are listed with [])."""
    lib = lib or os.path.join(ROOT, "jegal_amd", "libjegal_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(blob)
            part, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"b{i}.co")
            open(part, "wb").write(blob[s:e])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={part}", f"--output={co}", "--unbundle"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            cur = None
            names = {}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    names.setdefault(cur, [])
                elif cur is not None and PAT.search(line):
                    names[cur].append(line.strip())
            if names:
                dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
                for raw, d in zip(names, dem):
                    out[d.strip() or raw] = names[raw]
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-v"]
    pat = args[0] if args else ""
    res = scan()
    total = 0
    for k, v in sorted(res.items()):
        if pat in k and v:
            total += len(v)
            print(f"{len(v):4d}  {k[:170]}")
            if "-v" in sys.argv:
                for l in v:
                    print("        " + l)
    print(f"{total} packed fp32 instructions with operand selects in {sum(1 for k, v in res.items() if v and pat in k)} of {len(res)} kernels")
