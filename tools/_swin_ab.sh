for round in 1 2 3; do
 for lib in exp_libs/swin_base.so vit-ocm-wmsegmentation_amd/libocm_vit.so; do
   echo "$lib: $(OCM_VIT_LIB=$PWD/$lib python tools/bench_swin.py --precision bf16x3 2>/dev/null | head -1 | cut -c1-90)"
 done
done
