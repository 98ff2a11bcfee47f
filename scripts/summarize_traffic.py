"""reduce the rocprofv3 --pmc CSVs of scripts/collect_traffic.sh to per-launch
HBM bytes of gpuscan_qual_column.  gfx950 corrections (MI355X_MICROARCH.md
section HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream -> doubled; WRITE_SIZE is exact for streaming stores; both are in KB."""
import csv
import glob
import json
import os
import sys

out_dir, dest = sys.argv[1], sys.argv[2]
KERNEL = "gpuscan_qual_column"


def per_launch(counter):
    vals = []
    for path in glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Kernel_Name", "").startswith(KERNEL) and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


def full_size(vals):
    """the bench also launches some kernels on a small chunk (the ROW-format variant of C2 scans
    1e7 rows with the same kernel): keep the launches of the dominant size"""
    if not vals:
        return vals
    top = max(vals)
    return [v for v in vals if v >= 0.5 * top]


fetch, write = full_size(per_launch("FETCH_SIZE")), full_size(per_launch("WRITE_SIZE"))
rec = {"kernel": KERNEL, "launches": len(fetch)}
if fetch and write:
    f = sum(fetch) / len(fetch) * 1024.0
    w = sum(write) / len(write) * 1024.0
    rec.update({
        "FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes_raw": w,
        "fetch_bytes_corrected": 2.0 * f,
        "hbm_bytes_per_launch": 2.0 * f + w,
        "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request of a 16 B/lane stream); "
                      "WRITE_SIZE as is; KB -> bytes",
    })
for path in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Name"].startswith(KERNEL):
            rec["rocprof_avg_ns"] = float(row["AverageNs"])
            rec["rocprof_calls"] = int(row["Calls"])
try:
    line = [l for l in open(os.path.join(out_dir, "stats.log")) if l.startswith("{")][-1]
    b = json.loads(line)
    rec["chunk_rows"] = b["config"]["chunk_rows"]
    rec["selectivity"] = b["config"]["selectivity"]
    rec["bench_launch_us"] = b["roofline"]["launch_us"]
    rec["algorithmic_bytes_per_launch"] = b["roofline"]["bytes_per_launch"]
except Exception as e:
    rec["bench_line_error"] = str(e)
# every other kernel of the bench command: HBM bytes per launch with the same corrections
others = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "").split("(")[0]
            if row.get("Counter_Name") != counter or not name.startswith(("gpu", "hashjoin", "ingest", "membw")):
                continue
            d = others.setdefault(name, {"FETCH_SIZE": [], "WRITE_SIZE": []})
            d[counter].append(float(row["Counter_Value"]))
rec["kernels"] = {}
for name, d in sorted(others.items()):
    if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
        # (launches of the dominant size only; the two passes see the same launches in the same order)
        keep = [i for i, v in enumerate(d["FETCH_SIZE"]) if v >= 0.5 * max(d["FETCH_SIZE"])]
        fs = [d["FETCH_SIZE"][i] for i in keep]
        ws = [d["WRITE_SIZE"][i] for i in keep if i < len(d["WRITE_SIZE"])] or d["WRITE_SIZE"]
        d = {"FETCH_SIZE": fs, "WRITE_SIZE": ws}
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024.0
        w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024.0
        rec["kernels"][name] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_corrected": 2.0 * f,
                                "write_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w}
json.dump(rec, open(dest, "w"), indent=1)
print(json.dumps(rec, indent=1))
