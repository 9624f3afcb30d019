#!/bin/bash
# On the GPU box: step time with the sparse dW2 kernel and with the dense product, alternating.
set -e
cd /root/repo
for i in 1 2; do
  for v in 0 1; do
    GTG_DW2_DENSE=$v timeout -k 10 200 python bench.py --steps 300 --warmup 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense' if $v else 'sparse', d['ms_per_step'])"
  done
done
