#!/bin/bash
# per-kernel time of the partition plan under ablation flags (results are wrong then: timing only)
# usage: scripts/parts_ablate.sh "0 16 32 64 128" [ngroups]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/parts_ablate
rm -rf $OUT; mkdir -p $OUT
for A in $1; do
  export STROM_GPUPREAGG_ABLATE=$A
  export STROM_DIAGNOSTIC_BUILD=1   # the headers refuse *_ABLATE builds otherwise
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a$A -o run -- \
      python3 $GRAFT_REPO_ROOT/scripts/gpu_preagg_parts_probe.py 1e8 ${2:-1000000} parts > $OUT/a$A.log 2>&1 || { tail -5 $OUT/a$A.log; exit 1; }
  echo "== ablate=$A"; grep ngroups $OUT/a$A.log
  python3 - $OUT/a$A <<'PY'
import csv, glob, sys, statistics, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gpupreagg_hash" in r["Kernel_Name"]:
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    tail = v[len(v) // 2:]
    print("   %-28s n=%2d  median of the later half %.1f us" % (k, len(v), statistics.median(tail)))
PY
done
