bash scripts/gpu.sh tests r6u tests/test_tacotron_batched.py tests/test_tacotron.py tests/test_tacotron_long.py tests/test_tacotron_stop.py | tail -4
python scripts/debug/taco_batch_time.py 9 16 32 64 128 2>&1 | grep "B="
CTTS_TACO_BG_NO_FUSE=1 python scripts/debug/taco_batch_time.py 16 64 2>&1 | grep "B=" | sed "s/^/no-fuse /"
bash scripts/gpu.sh stats r6u_b16 scripts/debug/taco_batch_time.py 16 --steps 128 | head -8
