echo "== default"; python scripts/gpu_q1_probe.py 1e8 both 2>&1 | grep fold
for B in 100000 140000; do echo "== PRIV_LDS=$B"; STROM_GPUPREAGG_PRIV_LDS=$B python scripts/gpu_q1_probe.py 1e8 both 2>&1 | grep fold; done
echo "== NREP=16"; STROM_GPUPREAGG_NREP=16 python scripts/gpu_q1_probe.py 1e8 decimal 2>&1 | grep fold
echo "== NREP=64"; STROM_GPUPREAGG_NREP=64 python scripts/gpu_q1_probe.py 1e8 decimal 2>&1 | grep fold
