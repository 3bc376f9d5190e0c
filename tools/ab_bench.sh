#!/bin/bash
# Development: in-forward A/B of kernel variants with the development library. Usage: tools/ab_bench.sh "4=0" "4=3" ...
# Alternates the settings twice on one box; prints ms/step and the per-class kernel times.
export OCM_VIT_LIB=$PWD/exp_libs/libocm_vit_dev.so
for round in 1 2; do
  for k in "$@"; do
    OCM_KNOBS="$k" python bench.py --steps 20 --warmup 5 --no-slab --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb=d['kernel_breakdown']
print('knobs %-12s' % '$k', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (n[:5], v['avg_us']) for n,v in kb.items()), 'peaked %.2e' % d['attn_linf_by_weight_set']['peaked']['linf'])
"
  done
done
