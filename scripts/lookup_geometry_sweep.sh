# scan+join+group-by (gpupreagg_packed_lookup): rows in flight per thread and work-group size
for Q in 1 2 4; do for B in 512 1024; do
  echo "== GPUPREAGG_QUADS=$Q GPUPREAGG_BLOCK=$B"
  STROM_GPUPREAGG_QUADS=$Q STROM_GPUPREAGG_BLOCK=$B python scripts/gpu_lookup_probe.py 1e8 2>&1 | grep -i "lookup\|error" | tail -2
done; done
