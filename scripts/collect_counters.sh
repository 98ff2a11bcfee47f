#!/bin/bash
# Instruction-mix / cache counters of the kernels whose name contains FILTER,
# for "python3 SCRIPT ARGS..."; one PMC pass per counter group, kernel trace
# only (never combined with other trace domains).
# usage (GPU box, repo root): bash scripts/collect_counters.sh TAG FILTER SCRIPT [ARGS...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; FILTER=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  NAME=$(echo $CTRS | tr ' ' '_')
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/$NAME -- \
      python3 $ROOT/"$@" > $OUT/$NAME.log 2>&1
  echo "== $CTRS rc=$?"
  python3 - "$OUT/$NAME" "$FILTER" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sys.argv[2] not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print("%s %s launches=%d per_launch=%.4g" % (k[:40], c, cnt[(k, c)], v / cnt[(k, c)]))
PY
done
