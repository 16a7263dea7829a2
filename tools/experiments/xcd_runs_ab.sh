#!/bin/bash
# A/B of the XCD-aware tile order (xcd_run_index: gram_lds, gemm_tn_f64, synth_ls) (EMAGLS_XCD_RUNS=0: dispatch order): FETCH_SIZE per launch and kernel time, 16-lane launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gx}; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
CMD="python3 $R/bench.py --steps 64 --warmup 0 --slots 1 --batch 32 --no-cpu-baseline --no-sh-roofline --no-secondary"
for v in 1 0; do
  export EMAGLS_XCD_RUNS=$v
  timeout 300 rocprofv3 --kernel-trace -d $O/kt$v -o p -- $CMD > $O/kt$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$v -o p -- $CMD > $O/f$v.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w$v -o p -- $CMD > $O/w$v.log 2>&1
  (cd $R; python tools/pmc_simple.py $O/xcd$v.md $O/xcd$v.json $O/kt$v $O/f$v $O/w$v > /dev/null 2>&1; rm -rf $O/kt$v $O/f$v $O/w$v; echo "EMAGLS_XCD_RUNS=$v"; grep -i "gram_lds\|gemm_tn\|synth_ls\|kernel |" $O/xcd$v.md | cut -c1-200)
done
