#!/bin/bash
# Diagnostic PMC passes (one rocprofv3 run per counter group) over a short bench run; summary to stdout.
#   tools/pmc_passes.sh "GRBM_GUI_ACTIVE TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum" ...
export TMPDIR=/tmp
i=0
dirs=""
for grp in "$@"; do
  d=/tmp/pmc_pass_$i
  rm -rf $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 bench.py --cpu-iters 0 --skip-general --skip-blas1 --steps 20 --warmup 2 --spinup-seconds 0 --min-seconds 0 --traffic off --skip-permuted --skip-unstructured --skip-unstructured3d --skip-configs > /dev/null 2>&1
  dirs="$dirs $d"
  i=$((i+1))
done
python3 tools/pmc_summary.py $dirs
