#!/bin/bash
# Development: A/B of library builds or knob settings in the Swin-T forward (tools/bench_swin.py, batch 256, split-bf16).
#   tools/ab_swin.sh exp_libs/base.so vit-ocm-wmsegmentation_amd/libocm_vit.so      two builds, alternating three times on one box
#   tools/ab_swin.sh "0=0" "0=13"                                                    knob settings of exp_libs/libocm_vit_dev.so
for round in 1 2 3; do
  for a in "$@"; do
    if [[ "$a" == *.so ]]; then
      echo "$a: $(OCM_VIT_LIB=$PWD/$a python tools/bench_swin.py --precision bf16x3 2>/dev/null | head -1 | cut -c1-90)"
    else
      echo "knobs $a: $(OCM_VIT_LIB=$PWD/exp_libs/libocm_vit_dev.so OCM_KNOBS=$a python tools/bench_swin.py --precision bf16x3 2>/dev/null | head -1 | cut -c1-90)"
    fi
  done
done
