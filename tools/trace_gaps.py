"""Busy time vs gaps of the LAST `n` kernel dispatches of a rocprofv3 kernel trace (kernel_trace.csv), per kernel name:
where a launch-bound sequence (XLM-R: ~90 short launches per pass) loses time between its kernels.
Usage: python tools/trace_gaps.py <kernel_trace.csv> <n dispatches per pass> [passes]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n * passes:]
busy = collections.defaultdict(float); cnt = collections.Counter(); gap_after = collections.defaultdict(float)
tot_gap = 0.0
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    tot_gap += g
    gap_after[a["Kernel_Name"][:60]] += g
for r in rows:
    k = r["Kernel_Name"][:60]
    busy[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[k] += 1
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} dispatches over {span / passes:.1f} us per pass: busy {sum(busy.values()) / passes:.1f} us, gaps {tot_gap / passes:.1f} us")
for k in sorted(busy, key=busy.get, reverse=True):
    print(f"  {cnt[k] / passes:6.1f} x {busy[k] / cnt[k]:8.1f} us = {busy[k] / passes:8.1f} us  gap after (avg) {gap_after[k] / cnt[k]:6.2f} us   {k}")
