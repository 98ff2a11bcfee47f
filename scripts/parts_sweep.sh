#!/bin/bash
# per-kernel time of the partition plan for several partition counts (rocprofv3 kernel stats)
# usage: scripts/parts_sweep.sh "64 256 1024 2048" [ngroups]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/parts_sweep
rm -rf $OUT; mkdir -p $OUT
for P in $1; do
  export STROM_GPUPREAGG_HASH_PARTS=$P
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$P -o run -- \
      python3 $GRAFT_REPO_ROOT/scripts/gpu_preagg_parts_probe.py 1e8 ${2:-100000} parts > $OUT/p$P.log 2>&1 || exit 1
  echo "== nparts=$P"; grep ngroups $OUT/p$P.log
  f=$(find $OUT/p$P -name "*kernel_stats.csv" | head -1)
  grep gpupreagg_hash "$f" | cut -d, -f1-4
done
