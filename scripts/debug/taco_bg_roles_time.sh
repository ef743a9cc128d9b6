for d in 0 64 128 256 192 320 448; do
  echo "== CTTS_TACO_BG_DEBUG=$d"
  CTTS_TACO_BG_DEBUG=$d timeout 300 python scripts/debug/taco_batch_time.py 16 64 --steps 128 2>&1 | grep "B="
done
