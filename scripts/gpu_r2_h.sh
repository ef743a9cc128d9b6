#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 4 5 6; do python scripts/w4_stamps.py $v 2>&1 | tail -1; done
CTTS_BF16_NO_W4=1 python bench.py --dtype bf16 --batch 8 --steps 3 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('pp', round(d['ms_per_step'],2), r['mean_launch_ms'], r['res_hbm']['mean_launch_ms'], r['skip_hbm']['mean_launch_ms'])"
