#!/bin/bash
# Runs the full bench once per library variant (GPU box, repo root); prints value and kernel times.
for d in variants/*/; do
  name=$(basename $d)
  echo -n "$name: "
  PT_HIP_LIB=$PWD/$d/libpt_hip.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], 'Mrays/s', ' '.join('%s=%.1f' % (k, v['ms']) for k, v in r['kernel_ms'].items()))
except Exception as e:
    print('failed', e)
" || break
done
