"""Condense rocprofv3 outputs of `bench.py` into the per-kernel tables committed under profiles/.

  python tools/pmc_summary.py <tag> <kernel_trace.csv> [<pmc FETCH_SIZE counter_collection.csv> <pmc WRITE_SIZE counter_collection.csv>]

Every process launches more than the timed steps (the calibration pass at weight load: 4 clips of 16 frames; warm-up steps), so
averages over all dispatches mix launch sizes.  This tool keeps ONE step: the dispatches from the last conv1 scan / conv1 launch
to the end of the process (a full 32-clip step), and writes
  profiles/<tag>_kernel_summary.csv    kernel, launches in the step, total and average microseconds
  profiles/<tag>_pmc_summary.json      per kernel and launch: FETCH_SIZE / WRITE_SIZE as reported (KB), HBM bytes with the gfx950
                                       correction of MI355X_MICROARCH.md (FETCH_SIZE x 2: 128-B requests tallied as 64 B), per launch
Counter passes are separate runs (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; TCC has 4 slots: both do not fit one pass)."""
import csv
import json
import os
import sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "attn_mfma_flash_kernel" in name:
        return "attn_mfma_flash_kernel<%s>" % ("96" if "ILi96E" in name or "<96>" in name else "64")
    for key in ("conv1_direct_kernel", "conv1_zero_scan_kernel", "conv1_skip_mask_kernel", "conv1_edge_fix_kernel", "attn_mfma_s32_kernel",
                "rc_col_mean_kernel", "rc_gemv_kernel", "attn_mfma_kernel", "attn_kernel", "layernorm_kernel", "window_gather_tiled_kernel", "window_gather_kernel", "group_mean_kernel",
                "maxpool_kernel", "l2norm_kernel", "cast_kernel", "pe_project_kernel", "conv_rowmap_scan_kernel", "conv_rowmap_fill_kernel",
                "unpack_masked_kernel", "segment_mean_kernel", "audio_conv0_kernel", "gemm_x3_kernel", "gemm32_kernel", "attention32_kernel", "gemm_kernel"):
        if key in name:
            if key == "attn_mfma_s32_kernel":
                return key + ("<gather>" if "ILb1E" in name else "")
            return key
    if "gemm_glds_kernel" in name:
        import re
        v = re.findall(r"L[bi](\d+)E", name[name.index("gemm_glds_kernelI"):name.index("Ev", name.index("gemm_glds_kernelI"))] + "E")
        cfg = dict(zip(["W2", "CONV", "MI", "WM", "WN", "LNF", "SPR"], v))
        tile = f"{16 * int(cfg['MI']) * int(cfg['WM'])}x{64 * int(cfg['WN'])}"
        return f"gemm_glds_kernel<{tile}{',conv' if cfg['CONV'] == '1' else ''}{',hi+lo' if cfg['W2'] == '1' else ''}{',LN-fused' if cfg.get('LNF') == '1' else ''}>"
    return name.split("(")[0][:60]


def last_step(rows, key_start="Start_Timestamp"):
    rows = sorted(rows, key=lambda r: int(r[key_start]))
    idx = [i for i, r in enumerate(rows) if "conv1_zero_scan_kernel" in r["Kernel_Name"] or "conv1_direct_kernel" in r["Kernel_Name"]]
    if not idx:
        return rows
    # start of the last step = the last scan launch (or the last conv1 launch when zero-skip is off)
    scans = [i for i in idx if "conv1_zero_scan_kernel" in rows[i]["Kernel_Name"]]
    return rows[(scans or idx)[-1]:]


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    rows = last_step(list(csv.DictReader(open(trace))))
    agg = OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] = max(a[2], us)
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_summary.csv")
    with open(out, "w") as f:
        f.write("kernel,launches_in_last_step,total_us,avg_us,max_us\n")
        for k, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{n},{tot:.1f},{tot / n:.1f},{mx:.1f}\n")
        f.write(f"\"(sum of kernel durations of the step)\",{sum(a[0] for a in agg.values())},{sum(a[1] for a in agg.values()):.1f},,\n")
    print("wrote", out)
    if len(sys.argv) >= 5:
        res = {}
        for path, cname in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == cname]
            for r in last_step(rows):
                k = short(r["Kernel_Name"])
                d = res.setdefault(k, {"launches": defaultdict(int), "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
                d["launches"][cname] += 1
                d[cname + "_KB"] += float(r["Counter_Value"])
        summary = {}
        for k, d in res.items():
            nf, nw = max(d["launches"]["FETCH_SIZE"], 1), max(d["launches"]["WRITE_SIZE"], 1)
            fetch, write = d["FETCH_SIZE_KB"] / nf * 1024.0, d["WRITE_SIZE_KB"] / nw * 1024.0
            summary[k] = {"launches_in_step": nf, "FETCH_SIZE_bytes_per_launch_as_reported": fetch, "WRITE_SIZE_bytes_per_launch": write,
                          "hbm_bytes_per_launch": 2.0 * fetch + write,
                          "note": "FETCH_SIZE x 2 + WRITE_SIZE (gfx950: 128-B read requests are tallied as 64 B; Infinity-Cache hits are counted too: upper bound)"}
        out = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
        json.dump(summary, open(out, "w"), indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
