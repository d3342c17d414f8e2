#!/usr/bin/env python
"""Turn rocprofv3 CSVs (copied from gpurun_out/prof) into the small summaries committed here.

  python profiles/summarize.py stats  <kernel_stats.csv>              -> top kernels table (stdout)
  python profiles/summarize.py pmc    <fetch_counter.csv> <write_counter.csv> <out.json> [kernel-substring]
  python profiles/summarize.py sq     <counter_collection.csv> <out.json> [kernel-substring]
  python profiles/summarize.py round  <tag> [git sha of the profiled tree]   (everything tools/profile_round.sh wrote under
                                                                               gpurun_out/<tag>_* -> profiles/)

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


def sq(counter_csv, out_json, sub=None):
    """Per kernel: mean of every collected counter per launch, mean launch duration, and the derived figures
       clock_GHz        = GRBM_GUI_ACTIVE / 8 / duration          (rocprofv3 sums GUI_ACTIVE over the 8 XCDs)
       mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
       lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
       wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES   (waves parked at s_waitcnt / barriers)
       valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (both in quad-cycles: share of a wave's life spent executing VALU)
       valu_insts_per_wave_cycle, lds_insts_per_valu_inst: instruction mix of the VALU / LDS kernels (NCC, bending)"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(counter_csv)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            d = dur[k]
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            d[1] += 1
    out = {}
    for k, cs in acc.items():
        if sub and sub not in k:
            continue
        m = {c: v[0] / v[1] for c, v in cs.items()}
        e = {"launches": dur[k][1], "avg_launch_us": dur[k][0] / dur[k][1] / 1e3, "counters_per_launch": m}
        if "GRBM_GUI_ACTIVE" in m and dur[k][0] > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            e["clock_GHz"] = cyc / (dur[k][0] / dur[k][1])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQ_WAVE_CYCLES"):
            e["wait_frac"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_INST_ANY" in m:
                e["issue_stall_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
            if "SQ_ACTIVE_INST_VALU" in m:
                e["valu_active_frac"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_INSTS_VALU"):
            e["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
            if "SQ_INSTS_LDS" in m:
                e["lds_insts_per_valu_inst"] = m["SQ_INSTS_LDS"] / m["SQ_INSTS_VALU"]
        out[k] = e
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_launch_us"] * kv[1]["launches"])[:6]:
        print(k[-70:], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters_per_launch"})


def calib(counter_csv, out_json):
    """tools/ubench/fetch_calib: FETCH_SIZE (KiB, raw) of kernels whose traffic is known -> raw bytes / known bytes per kernel.
    0.5 for stream_b128 means the gfx950 'x 2' correction applies to raw_buffer_load_b128 (the NCC kernel's instruction)."""
    known = {"stream_b128": 256 << 20, "stream_dword": 256 << 20, "reread_b128": 256 << 20}
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(counter_csv)):
        if r["Counter_Name"] == "FETCH_SIZE":
            a = acc[r["Kernel_Name"].split("(")[0]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    out = {}
    for k, (v, n) in acc.items():
        for name, b in known.items():
            if name in k:
                out[name] = {"launches": n, "fetch_KiB_raw_per_launch": v / n, "requested_bytes": b,
                             "raw_over_requested": v / n * 1024 / b,
                             "unique_bytes": (32 << 20) if name == "reread_b128" else b}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


def round_(tag, head=None):
    import glob
    import os
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    g = lambda pat: sorted(glob.glob(os.path.join(root, "gpurun_out", pat)))
    # the commit that was PROFILED (the GPU box has no .git: pass it; default = HEAD now, right only if nothing was committed since)
    head = head or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
    cal = g(f"{tag}_pmc_calib/*/*counter_collection.csv")
    if cal:
        calib(cal[0], os.path.join(here, f"{tag}_fetch_calibration.json"))
    for wl in ("infer", "train", "ncc"):
        st = g(f"{tag}_stats_{wl}/*/*kernel_stats.csv")
        if st:
            shutil.copy(st[0], os.path.join(here, f"{tag}_{wl}_kernel_stats.csv"))
        f, w = g(f"{tag}_pmc_{wl}_FETCH_SIZE/*/*counter_collection.csv"), g(f"{tag}_pmc_{wl}_WRITE_SIZE/*/*counter_collection.csv")
        if f and w:
            pmc(f[0], w[0], os.path.join(here, f"{tag}_{wl}_pmc_traffic.json"), "conv3d_k3_kernel<1, 2, 4, 4, 2" if wl == "infer" else "")
        q = g(f"{tag}_pmc_{wl}_SQ/*/*counter_collection.csv")
        if q:
            sq(q[0], os.path.join(here, f"{tag}_{wl}_pmc_sq.json"))
    for name in ("default", "train", "ncc"):
        b = os.path.join(root, "gpurun_out", f"{tag}_bench_{name}.json")
        if os.path.exists(b):
            shutil.copy(b, os.path.join(here, f"{tag}_bench_{name}.json"))
    # the files bench.py reads `roofline.traffic` from: kernel family (bench.py / ops.py names) -> measured HBM bytes per launch
    for wl in ("infer", "train", "ncc"):
        pj = os.path.join(here, f"{tag}_{wl}_pmc_traffic.json")
        if not os.path.exists(pj):
            continue
        d = json.load(open(pj))
        out = {}
        for k, v in d.items():
            fam = family_of(k)
            if fam and (fam not in out or v["launches"] > out[fam]["launches"]):
                out[fam] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "kernel": k, "launches": v["launches"], "git": head,
                            "source": f"profiles/{tag}_{wl}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in "
                                      "separate passes via tools/profile_round.sh; (2*FETCH + WRITE) KiB per launch, fetch "
                                      "doubled per the gfx950 correction)"}
        json.dump(out, open(os.path.join(here, f"traffic_{wl}.json"), "w"), indent=1)


def family_of(kernel_name):
    """rocprof kernel name -> the family name ops.py times it under (None for kernels bench.py has no roofline for)."""
    import re
    k = kernel_name.replace("mmr::", "")
    m = re.search(r"conv3d_k3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        dt, wm, wn, mt, nt, var = map(int, m.groups())
        fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[dt]}_bn{wn * nt * 32}"
        return fam + ("_upfold" if var & (1 << 16) else "_cinit" if var & (1 << 17) else "_dgfold" if var & (1 << 19) else "")
    for name in ("ncc_fused4_kernel", "ncc_fused_kernel", "bending_fused_kernel"):
        if name in k:
            return name
    return None


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "sq":
        sq(*sys.argv[2:])
    elif sys.argv[1] == "round":
        round_(*sys.argv[2:4])
    else:
        pmc(*sys.argv[2:])
