#!/bin/bash
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of one 8-lane batch of config 4 at r = 10 cm (S = 2025): the Gram kernel's bytes and time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c4g}; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
CMD="python3 $R/tools/experiments/config4_prof.py 9.8 10.0"
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $R; python tools/pmc_simple.py $O/c4.md $O/c4.json $O/kt $O/f $O/w > /dev/null 2>&1; rm -rf $O/kt $O/f $O/w
grep -i "gram_lds\|gemm_tn\|kernel |" $O/c4.md | cut -c1-200; tail -1 $O/kt.log | cut -c1-300
