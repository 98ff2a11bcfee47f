for U in 2 3; do echo "== GPUPREAGG_HASH_UNROLL=$U"; STROM_GPUPREAGG_HASH_UNROLL=$U python scripts/gpu_preagg_parts_probe.py 1e8 100000,1000000 parts 2>&1 | grep -i "ngroups\|error" ; done
