"""Summarise a rocprofv3 kernel-trace CSV: per kernel start/end (ms, relative) for the last frame."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows)
# last frame = after the last gap > 20 ms... use the last clear kernel as separator if present; else take last N
t_end = ev[-1][1]
# find frame start: walk back until a gap > 2 ms between consecutive kernel activity
i = len(ev) - 1
lo = ev[i][0]
while i > 0:
    prev_end = max(e[1] for e in ev[:i])
    if lo - prev_end > 1_500_000: break
    i -= 1; lo = min(lo, ev[i][0])
t0 = ev[i][0]
for s, e, n in ev[i:]:
    print(f"{(s-t0)/1e6:8.3f} -> {(e-t0)/1e6:8.3f}  ({(e-s)/1e6:7.3f} ms)  {n}")
