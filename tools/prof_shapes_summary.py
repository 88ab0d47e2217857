#!/usr/bin/env python3
"""Folds the rocprofv3 CSVs written by tools/prof_shapes.sh into one JSON (stdout): per case and lq:: kernel the average
duration, algorithmic GB/s, HBM traffic per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide
coalesced reads; WRITE_SIZE as is; both in KiB), the SQ wait split, LDS conflict cycles, registers and LDS per block."""
import collections
import csv
import glob
import json
import os
import sys

if sys.argv[1] == "--table":
    # markdown table of a summary.json: one row per descriptor, K1 / K2 / K4 traversal kernel, rocprofv3 average duration ->
    # algorithmic GB/s and fraction of the 8 TB/s HBM peak, HBM traffic over algorithmic bytes from the PMC passes
    d = json.load(open(sys.argv[2]))
    print("| descriptor (outer, G, inner) | K1 kernel | K1 GB/s (frac) | K2 kernel | K2 GB/s (frac) | K4 kernel | K4 GB/s (frac) | HBM traffic / algorithmic (K1, K2, K4) |")
    print("|---|---|---|---|---|---|---|---|")
    for name, e in d.items():
        cells, traf = [], []
        for op in ("0", "1", "2"):
            ks = [(k, v) for k, v in e["kernels"].items() if "finalize" not in k and "<" in k and k.split("<")[1].split(",")[0].split(">")[0].strip() == op]
            if not ks:
                cells += ["-", "-"]
                traf.append("-")
                continue
            k, v = max(ks, key=lambda kv: kv[1].get("avg_us", 0))
            cells += [f"`{k.replace('lq::', '')}`", f"{v.get('algorithmic_GBs', 0):.0f} ({v.get('frac_of_8TBs', 0):.2f})"]
            traf.append(f"{v['traffic_over_algorithmic']:.3f}" if "traffic_over_algorithmic" in v else "-")
        print(f"| {name} {tuple(e['descriptor'])} | " + " | ".join(cells) + " | " + ", ".join(traf) + " |")
    sys.exit(0)

out = sys.argv[1]
cases = sys.argv[2:]
ALGO = {"0": 8, "1": 8, "2": 12}      # bytes per element by OP template argument: K1 fwd, K2 bwd, K4 fused


def short(name):
    return name.split("(")[0].replace("void ", "")


def op_of(name):
    a = name.split("<")[1].split(",")[0].split(">")[0].strip() if "<" in name else ""
    return a


def one(path_glob):
    f = sorted(glob.glob(path_glob), key=os.path.getsize)
    return f[-1] if f else None


res = {}
for c in cases:
    name, d = c.split(":")
    o, g, i = (int(v) for v in d.split(","))
    n = o * g * i
    entry = {"descriptor": [o, g, i], "elements": n, "kernels": {}}
    f = one(f"{out}/{name}/stats/*/*kernel_stats.csv")
    if f:
        for r in csv.DictReader(open(f)):
            if "lq::" not in r["Name"] or "selftest" in r["Name"]:
                continue
            k = short(r["Name"])
            e = entry["kernels"].setdefault(k, {})
            e.update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3)
            b = ALGO.get(op_of(k))
            if b and "finalize" not in k:
                e["algorithmic_bytes"] = b * n
                e["algorithmic_GBs"] = b * n / (float(r["AverageNs"]) * 1e-9) / 1e9
                e["frac_of_8TBs"] = e["algorithmic_GBs"] / 8000.0
    for pas in ("fetch", "write", "sq"):
        f = one(f"{out}/{name}/{pas}/*/*counter_collection.csv")
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            if "lq::" not in r["Kernel_Name"] or "selftest" in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_block_bytes": int(r["LDS_Block_Size"]),
                       "grid_threads": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
        for k, v in agg.items():
            e = entry["kernels"].setdefault(k, {})
            e.update(meta[k])
            m = {cn: sum(x) / len(x) for cn, x in v.items()}
            if pas == "fetch":
                e["hbm_read_bytes"] = m["FETCH_SIZE"] * 1024 * 2
            elif pas == "write":
                e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
            else:
                wc = m.get("SQ_WAVE_CYCLES", 0) or 1
                e["sq"] = {"waves": m.get("SQ_WAVES"), "wave_cycles": wc,
                           "frac_wait_any": m.get("SQ_WAIT_ANY", 0) / wc, "frac_wait_inst_any": m.get("SQ_WAIT_INST_ANY", 0) / wc,
                           "frac_active_inst_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc, "frac_active_valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                           "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": m.get("SQ_LDS_IDX_ACTIVE")}
    for k, e in entry["kernels"].items():
        if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
            e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
            if e.get("algorithmic_bytes"):
                e["traffic_over_algorithmic"] = e["hbm_traffic_bytes"] / e["algorithmic_bytes"]
    res[name] = entry
json.dump(res, sys.stdout, indent=1)
