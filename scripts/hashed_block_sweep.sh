# the hashed partition plan: work-group sizes of the fold (one work-group per CU: its threads are
# the CU's whole occupancy) and of the LDS scatter
for FB in 256 512 1024; do for SB in 256 1024; do
  echo "== fold block $FB, scatter block $SB"
  STROM_GPUPREAGG_HASH_FOLD_BLOCK=$FB STROM_GPUPREAGG_HASH_SCATTER_BLOCK=$SB python scripts/gpu_preagg_parts_probe.py 1e8 100000,1000000 parts 2>&1 | grep ngroups
done; done
