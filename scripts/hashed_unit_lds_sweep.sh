# the hashed partition plan: the units' LDS table at 159 KB (one work-group of 1024 threads per CU)
# against 79 KB (two per CU, half the slots, twice the partitions beyond 3e5 groups)
for KB in 159 79; do
  echo "== unit LDS table budget $KB KB"
  STROM_GPUPREAGG_HASH_UNIT_LDS_KB=$KB python scripts/gpu_preagg_parts_probe.py 1e8 10000,100000,1000000,10000000 parts 2>&1 | grep ngroups
done
