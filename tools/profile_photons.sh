#!/bin/bash
# rocprofv3 kernel statistics of the photon emission pass (tools/bench_photons.py).
# usage (on the GPU box): tools/profile_photons.sh
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_photons
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o ph -- python3 $R/tools/bench_photons.py > $OUT/bench_photons.txt 2> $OUT/ktrace.log || { tail -5 $OUT/ktrace.log; exit 1; }
cat $OUT/bench_photons.txt
for f in $(find $OUT/ktrace -name "*kernel_stats.csv"); do cp $f $OUT/photons_kernel_stats.csv; cut -c1-160 $f | head -6; done
