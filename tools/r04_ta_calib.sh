#!/bin/bash
# GPU box: what does TA_TA_BUSY_sum read on kernels whose texture-addresser load is known?  (the per-level
# tool divided by 256 x GRBM_GUI_ACTIVE, the ablation note by 256 x duration x clock -- 8x apart.)
#   tools/ubench/gather_rate   mode 0 = every lane its own random line: the addresser's known saturation (0.43 lines/clk/CU)
#   tools/ubench/copy_rate     coalesced streaming copy
# usage: tools/r04_ta_calib.sh <tag>
T=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in gather_rate copy_rate; do
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$b/p --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max -- $GRAFT_REPO_ROOT/tools/ubench/$b > $OUT/$b.p.out 2> $OUT/$b.p.err || \
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$b/p --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/tools/ubench/$b > $OUT/$b.p.out 2> $OUT/$b.p.err
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$b/t -- $GRAFT_REPO_ROOT/tools/ubench/$b > $OUT/$b.t.out 2> $OUT/$b.t.err
done
python3 $GRAFT_REPO_ROOT/tools/ta_calib.py $OUT | tee $OUT/ta_calibration.txt
