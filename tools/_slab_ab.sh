export OCM_VIT_LIB=$PWD/exp_libs/libocm_vit_dev.so
for round in 1 2; do
 for k in "0=0" "7=1"; do
   echo "knobs $k: $(OCM_KNOBS=$k python tools/sweep_slab.py 2>/dev/null | tail -1 | cut -c1-110)"
 done
done
