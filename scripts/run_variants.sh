#!/bin/bash
# Runs the trace micro-benchmark once per library variant (GPU box, repo root).
# Usage: bash scripts/run_variants.sh [bench args...]
for d in variants/*/; do
  name=$(basename $d)
  echo -n "$name: "
  PT_HIP_LIB=$PWD/$d/libpt_hip.so timeout -k 10 120 python3 scripts/raycast_bench.py --reps 3 --width 3840 --height 2160 --sphere-rays 8388608 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(' '.join('%s=%s' % (k, v['Mrays_s']) for k, v in r.items() if isinstance(v, dict) and 'Mrays_s' in v))
except Exception as e:
    print('failed', e)
" || break
done
