"""Per-kernel register / spill / LDS figures of libjegal_hip.so, read from the gfx950 code objects embedded in it
(the .hip_fatbin section holds one clang offload bundle per translation unit).  Used by tests/test_host_cpu.py to keep
`.vgpr_spill_count` of the hot kernels at 0, and by hand:  python tools/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(lib=None):
    lib = lib or os.path.join(ROOT, "jegal_amd", "libjegal_hip.so")
    res = {}
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
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                res[dem] = {"vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "spill": get("vgpr_spill_count"),
                            "scratch": get("private_segment_fixed_size"), "lds": get("group_segment_fixed_size")}
    return res


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for k, v in sorted(kernel_resources().items()):
        if pat in k:
            print(f"{v['vgpr']:4d} vgpr {v['sgpr']:4d} sgpr  spill {v['spill']:3d}  scratch {v['scratch']:4d}  lds {v['lds']:6d}  {k[:150]}")
