#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (HBM traffic, separate passes, kernel trace only) of the kernels whose
# name contains FILTER, for "python3 SCRIPT ARGS...".  Raw counter values are in KB of 64-B... see
# MI355X_MICROARCH.md; summarize_traffic.py applies the gfx950 correction for the headline kernel,
# here the raw per-launch averages are printed next to each other.
# usage (GPU box, repo root): bash scripts/collect_fetch.sh TAG FILTER SCRIPT [ARGS...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; FILTER=$2; shift 2
OUT=$ROOT/gpurun_out/fetch_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -- \
      python3 $ROOT/"$@" > $OUT/$CTR.log 2>&1
  python3 - "$OUT/$CTR" "$FILTER" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sys.argv[2] in k:
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for (k, c), v in acc.items():
    print("%s %s launches=%d raw_per_launch=%.5g" % (k[:44], c, cnt[(k, c)], v / cnt[(k, c)]))
PY
done
