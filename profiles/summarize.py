#!/usr/bin/env python
"""Turn rocprofv3 CSVs (copied from gpurun_out/prof) into the small summaries committed here.

  python profiles/summarize.py stats  <kernel_stats.csv>              -> top kernels table (stdout)
  python profiles/summarize.py pmc    <fetch_counter.csv> <write_counter.csv> <out.json> [kernel-substring]

PMC units and corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read -> doubled here;
WRITE_SIZE is exact for 16-B-per-lane stores (narrower stores are uncalibrated and flagged)."""
import csv
import json
import sys
from collections import defaultdict


def stats(path, n=14):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
    for r in rows[:n]:
        name = r["Name"].split("(")[0][-70:]
        print(f"{name:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} "
              f"{100*float(r['TotalDurationNs'])/tot:6.2f}")


def pmc(fetch_csv, write_csv, out_json, sub="conv3d_k3_kernel<1, 2, 4, 4, 2>"):
    def per_kernel(path, counter):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        return acc
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in f:
        fk, n = f[k]
        wk = w.get(k, [0.0, n])[0]
        out[k.split("(")[0]] = {"launches": n, "fetch_KiB_raw_per_launch": fk / n, "write_KiB_per_launch": wk / max(n, 1),
                                "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024 / n,
                                "note": "fetch doubled per the gfx950 FETCH_SIZE correction (16 B/lane loads)"}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out.items():
        if sub in k:
            print(k, json.dumps(v))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(*sys.argv[2:])
