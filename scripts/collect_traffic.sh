#!/bin/bash
# HBM traffic of the GpuScan kernel from the PMC counters, collected the way
# MI355X_MICROARCH.md prescribes: counters in their own runs (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), with --kernel-trace only.
# Run on the GPU box from the repo root:  bash scripts/collect_traffic.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -- \
      python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$CTR.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
      python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
python3 $ROOT/scripts/summarize_traffic.py $OUT $ROOT/gpurun_out/gpuscan_traffic.json
# the kernel statistics of the same command (headline + C4 region + other operators), for profiles/
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/bench_kernel_stats.csv \;
grep "^{" $OUT/stats.log | tail -1 > $ROOT/gpurun_out/bench_under_rocprof.json || true
