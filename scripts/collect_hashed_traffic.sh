#!/bin/bash
# HBM traffic of the hashed GROUP BY's partition-plan kernels from the PMC counters (FETCH_SIZE and
# WRITE_SIZE in their own runs, --kernel-trace only; MI355X_MICROARCH.md: FETCH_SIZE of a wide
# coalesced stream counts 64 B per 128-B request -> doubled; both in KB).  1e8 rows, 1e6 groups.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_hashed
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -- \
      python3 $ROOT/scripts/gpu_preagg_parts_probe.py 1e8 1000000 parts > $OUT/$CTR.log 2>&1 || exit 1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = collections.defaultdict(list)
    for path in glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == ctr and "gpupreagg_hash" in row.get("Kernel_Name", ""):
                vals[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        v = v[len(v) // 2:]                       # steady-state launches
        res[k][ctr] = sum(v) / len(v) * 1024.0
for k, d in sorted(res.items()):
    f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    print("%-30s FETCH_SIZE %.3f GB raw (x2 for coalesced streams: %.3f GB)  WRITE_SIZE %.3f GB" % (k, f / 1e9, 2 * f / 1e9, w / 1e9))
PY
