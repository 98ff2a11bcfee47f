# Q1 fold kernels with the LDS section offsets kept in vector registers (default) and as scalars
for S in 0 1; do echo "== GPUPREAGG_LDS_OFFSETS_SCALAR=$S"; STROM_GPUPREAGG_LDS_OFFSETS_SCALAR=$S python scripts/gpu_q1_probe.py 1e8 both 2>&1 | grep fold
 echo "   (no lane-private accumulators)"; STROM_GPUPREAGG_NO_REG=1 STROM_GPUPREAGG_LDS_OFFSETS_SCALAR=$S python scripts/gpu_q1_probe.py 1e8 decimal 2>&1 | grep fold; done
