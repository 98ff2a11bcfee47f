# C3 one-pass join kernel: geometry sweep (tile = BLOCK x 4 x QUADS rows, LDS stage of STAGE records)
echo "== default"; python scripts/gpu_c3_probe.py 2>&1 | grep "slots"
for Q in 4 8; do for S in 4096 8192; do echo "== QUADS=$Q STAGE=$S"; STROM_HASHJOIN_QUADS=$Q STROM_HASHJOIN_STAGE=$S python scripts/gpu_c3_probe.py 2>&1 | grep "3-byte"; done; done
echo "== BLOCK=512"; STROM_HASHJOIN_BLOCK=512 python scripts/gpu_c3_probe.py 2>&1 | tail -3
