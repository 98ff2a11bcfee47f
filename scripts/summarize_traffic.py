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


fetch, write = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
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
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024.0
        w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024.0
        rec["kernels"][name] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_corrected": 2.0 * f,
                                "write_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w}
json.dump(rec, open(dest, "w"), indent=1)
print(json.dumps(rec, indent=1))
