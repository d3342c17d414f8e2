#!/usr/bin/env python
"""Timeline of ONE C3 training step from a rocprofv3 kernel trace: every dispatch in start order with its duration and the
gap to the previous dispatch's end (per-step sums at the end).
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 <repo>/tools/train_layer_table.py 3
  python tools/step_timeline.py <dir>/**/*_kernel_trace.csv [step index from the end, default 1] [name of a step's last kernel]"""
import csv, sys, glob, re
path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
delim = sys.argv[3] if len(sys.argv) > 3 else "adam_kernel"   # last kernel of a step ("warp3d_kernel<0" for bench.py's inference)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the batched weight pack's successor ... simpler: split at adam_kernel (last kernel of a step but the pack)
ends = [i for i, r in enumerate(rows) if delim in r["Kernel_Name"]]
if len(ends) < back + 1:
    raise SystemExit("not enough steps in the trace")
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("mmr::", ""))[:60]
print(f"{'t us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel   (grid x block)")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {short(r['Kernel_Name'])}  ({r.get('Grid_Size_X', '?')}/{r.get('Workgroup_Size_X', '?')}) q{r.get('Queue_Id', '?')}")
    busy += (e - s) / 1e3
    if gap > 0:
        gaps += gap
    prev_end = max(prev_end, e)
print(f"step: {len(step)} dispatches, span {(prev_end - t0) / 1e3:.1f} us, sum of durations {busy:.1f} us, sum of positive gaps {gaps:.1f} us")
