#!/bin/bash
# Instruction-mix counters of the Q1-shaped GpuPreAgg kernel (C5): is it
# VALU-, LDS- or memory-bound?  One PMC pass per counter group, kernel trace only.
# Run on the GPU box from the repo root:  bash scripts/collect_q1_counters.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_q1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  TAG=$(echo $CTRS | tr ' ' '_')
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/$TAG -- \
      python3 $ROOT/scripts/bench_configs.py 2e7 c5 > $OUT/$TAG.log 2>&1
  echo "== $CTRS rc=$?"
  python3 - "$OUT/$TAG" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "gpupreagg" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print("%s %s total=%.4g launches=%d per_launch=%.4g" % (k[:40], c, v, cnt[(k, c)], v / cnt[(k, c)]))
PY
done
