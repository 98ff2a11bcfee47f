for M in 0 1 2; do echo "== PROBE=$M"; STROM_GPUPREAGG_LOOKUP_PROBE=$M python scripts/gpu_lookup_ablate.py 2>&1 | grep -v amdgpu.ids | grep "real\|no accumulate\|WHERE"; done
