"""Condense the SQ / GRBM counter passes of tools/profile_sq.sh into profiles/<tag>_sq_summary.json.

  python tools/sq_summary.py <tag> <mfma counter_collection.csv> <lds counter_collection.csv> <issue counter_collection.csv>

rocprofv3 reports every counter as the SUM over its instances: SQ_* over the shader engines / CUs that hold one, GRBM_GUI_ACTIVE
over the 8 XCDs.  Per kernel (the dispatches of the LAST step of the process, as tools/pmc_summary.py) this tool derives

  mfma_busy_frac    SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  /  (GRBM_GUI_ACTIVE / 8)      cycles a SIMD's matrix pipe was busy / kernel cycles
  clock_ghz         GRBM_GUI_ACTIVE / 8 / dispatch duration (the profiled pass's own duration: MI355X_MICROARCH.md, DVFS give-back;
                    reads high on dispatches under ~0.3 ms)
  cycles_per_mfma   SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA  (16 for 16x16x32, 32 for 32x32x16 f16: a sanity check of the units)
  lds_conflict_frac SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait / issue      SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (quad-cycles, disjoint)
"""
import csv
import json
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import ROOT, last_step, short  # noqa: E402

N_SIMD, N_XCD = 1024, 8


def collect(path):
    """kernel -> {counter: sum over the step's launches, 'launches': n, 'us': total duration}"""
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return {}
    # one dispatch = several rows (one per counter): cut the last step on dispatch level
    first = {}
    for r in rows:
        first.setdefault(r["Dispatch_Id"], r)
    keep = {r["Dispatch_Id"] for r in last_step(list(first.values()))}
    out = OrderedDict()
    seen = set()
    for r in rows:
        if r["Dispatch_Id"] not in keep:
            continue
        k = short(r["Kernel_Name"])
        d = out.setdefault(k, {"launches": 0, "us": 0.0})
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            d["launches"] += 1
            d["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def main():
    tag = sys.argv[1]
    mfma, lds, issue = (collect(p) if p and os.path.exists(p) else {} for p in (sys.argv[2:5] + [None] * 3)[:3])
    res = OrderedDict()
    for k in sorted(set(mfma) | set(lds) | set(issue), key=lambda k: -(mfma.get(k) or lds.get(k) or issue.get(k))["us"]):
        e = OrderedDict()
        m = mfma.get(k)
        if m and m.get("GRBM_GUI_ACTIVE"):
            cyc = m["GRBM_GUI_ACTIVE"] / N_XCD
            e["launches_in_step"] = m["launches"]
            e["us_per_launch_under_pmc"] = round(m["us"] / m["launches"], 1)
            e["kernel_cycles_per_launch"] = round(cyc / m["launches"])
            # GRBM_GUI_ACTIVE covers the dispatch's front and back porch as well as the kernel: below ~50 us per launch the quotient is
            # not a clock (round 5 printed 3-12 GHz for the 4-9 us kernels) and the busy fraction inherits the same inflated denominator
            if m["us"] / m["launches"] >= 50.0:
                e["clock_ghz_under_pmc"] = round(cyc / (m["us"] * 1e3), 3)
            else:
                e["short_launch"] = "GRBM_GUI_ACTIVE / duration is meaningless below 50 us per launch: no clock; fractions of kernel_cycles are lower bounds"
            e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / N_SIMD / cyc, 4)
            if m.get("SQ_INSTS_MFMA"):
                e["cycles_per_mfma"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / m["SQ_INSTS_MFMA"], 2)
                e["mfma_insts_per_launch"] = round(m["SQ_INSTS_MFMA"] / m["launches"])
            if m.get("SQ_BUSY_CYCLES"):
                e["sq_busy_cycles_per_se"] = round(m["SQ_BUSY_CYCLES"] / 32 / m["launches"])
        l = lds.get(k)
        if l and l.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4)
            e["lds_addr_conflict_frac"] = round(l.get("SQ_LDS_ADDR_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4)
            e["lds_unaligned_stall_frac"] = round(l.get("SQ_LDS_UNALIGNED_STALL", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4)
            if l.get("GRBM_GUI_ACTIVE"):
                # LDS-array active cycles summed over the CUs / (256 CUs x kernel cycles)
                e["lds_active_frac_of_kernel"] = round(l["SQ_LDS_IDX_ACTIVE"] / 256.0 / (l["GRBM_GUI_ACTIVE"] / N_XCD), 4)
            if l.get("SQ_INSTS_LDS"):
                e["lds_cycles_per_inst"] = round(l["SQ_LDS_IDX_ACTIVE"] / l["SQ_INSTS_LDS"], 2)
        s = issue.get(k)
        if s and s.get("SQ_WAVE_CYCLES"):
            w = s["SQ_WAVE_CYCLES"]
            e["wave_wait_any_frac"] = round(s.get("SQ_WAIT_ANY", 0.0) / w, 4)
            e["wave_wait_inst_any_frac"] = round(s.get("SQ_WAIT_INST_ANY", 0.0) / w, 4)
            e["wave_active_inst_any_frac"] = round(s.get("SQ_ACTIVE_INST_ANY", 0.0) / w, 4)
            e["wave_active_inst_valu_frac"] = round(s.get("SQ_ACTIVE_INST_VALU", 0.0) / w, 4)
            e["valu_insts_per_launch"] = round(s.get("SQ_INSTS_VALU", 0.0) / s["launches"])
        if e:
            res[k] = e
    res["_note"] = ("counters summed over instances by rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs, GRBM_GUI_ACTIVE / 8 XCDs; one "
                    "one-stream 32-clip step of `bench.py --no-extras --steps 1 --warmup 1 --opt dual_stream=0`; durations and clocks are those of "
                    "the profiled passes (slower than un-profiled runs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles")
    out = os.path.join(ROOT, "profiles", f"{tag}_sq_summary.json")
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)
    for k, e in list(res.items())[:8]:
        if isinstance(e, dict):
            print(k, {x: e[x] for x in ("us_per_launch_under_pmc", "clock_ghz_under_pmc", "mfma_busy_frac", "lds_conflict_frac") if x in e})


if __name__ == "__main__":
    main()
