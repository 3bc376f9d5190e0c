#!/bin/bash
# Development: in-forward A/B of whole library builds (e.g. `make -C vit-ocm-wmsegmentation_amd/csrc abl15` against the
# product library). Usage: tools/ab_libs.sh vit-ocm-wmsegmentation_amd/libocm_vit.so exp_libs/abl15.so [--slab]
# Alternates the libraries twice on one box; prints ms/step and the per-class kernel times (or the slab sweep time).
SLAB=0; LIBS=()
for a in "$@"; do if [ "$a" = "--slab" ]; then SLAB=1; else LIBS+=("$a"); fi; done
for round in 1 2; do
  for lib in "${LIBS[@]}"; do
    export OCM_VIT_LIB=$PWD/$lib
    if [ $SLAB = 1 ]; then
      echo "$lib: $(python tools/sweep_slab.py 2>/dev/null | tail -1 | cut -c1-120)"
    else
      python bench.py --steps 20 --warmup 5 --no-slab --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb=d['kernel_breakdown']
print('%-44s' % '$lib', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (n[:5], v['avg_us']) for n,v in kb.items()), 'peaked %.2e' % d['attn_linf_by_weight_set']['peaked']['linf'])
"
    fi
  done
done
