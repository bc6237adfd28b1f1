#!/bin/bash
# Builds the library with several (waves per SIMD, sweep cap) settings of roots_kernel_t on the GPU box and times the solver kernels.
#   gpurun -- 'bash tools/roots_variants.sh'
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/roots_variants
mkdir -p $OUT
rm -f $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
for v in "2 400" "2 64" "3 64" "4 64"; do
    set -- $v
    (cd $R/matchinglib_poselib_amd/csrc && touch ransac_5pt.hip && make EXTRA_CXXFLAGS="-DMLPL_ROOTS_WAVES=$1 -DMLPL_SWEEP_CAP=$2" > $OUT/build_$1_$2.log 2>&1)
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/w$1_c$2 -o t -- python $R/tools/solver_timing.py 41472 > $OUT/w$1_c$2.log 2>&1
    echo "== waves $1 cap $2" >> $OUT/summary.txt
    grep solver_wave3 $OUT/w$1_c$2.log >> $OUT/summary.txt
    python $R/tools/rocpd_kernels.py $OUT/w$1_c$2/t_results.db roots_kernel | grep -E "grid +442368" >> $OUT/summary.txt || true
    python $R/tools/rocpd_kernels.py $OUT/w$1_c$2/t_results.db solve5pt | grep -E "grid +(884736|2654208)" >> $OUT/summary.txt || true
    rm -rf $OUT/w$1_c$2
done
cat $OUT/summary.txt
