# us/step of the batched decoder under CTTS_TACO_BG_DEBUG timing bits (wrong results): bash scripts/debug/taco_bg_roles_time.sh [bits ...]
bits=${@:-0 64 128 256 192 320 448}
for d in $bits; do
  echo "== CTTS_TACO_BG_DEBUG=$d"
  CTTS_TACO_BG_DEBUG=$d timeout 300 python scripts/debug/taco_batch_time.py 16 64 --steps 128 2>&1 | grep "B="
done
