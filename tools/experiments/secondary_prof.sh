#!/bin/bash
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the secondary workloads (verdict r3 item 2): binauralDecode (100 s),
# the SH-basis launch at D = 2^20, one lane batch of config 4 (r = 5 cm and r = 10 cm), config 5's batch of 8 subjects.
#   bash tools/experiments/secondary_prof.sh <tag>        -> gpurun_out/<tag>_<workload>.{md,json}
tag=${1:-r04}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export TMPDIR=/tmp; cd /tmp
run() {   # name, python script and arguments
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_${name}_kt -o p -- python3 "$@" > $R/gpurun_out/${tag}_${name}.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${tag}_${name}_$c -o p -- python3 "$@" > /dev/null 2>&1
  done
  (cd $R; python tools/pmc_simple.py gpurun_out/${tag}_${name}.md gpurun_out/${tag}_${name}.json gpurun_out/${tag}_${name}_kt gpurun_out/${tag}_${name}_FETCH_SIZE gpurun_out/${tag}_${name}_WRITE_SIZE;
   rm -rf gpurun_out/${tag}_${name}_kt gpurun_out/${tag}_${name}_FETCH_SIZE gpurun_out/${tag}_${name}_WRITE_SIZE; head -8 gpurun_out/${tag}_${name}.md | cut -c1-170)
}
run decode_real $R/tools/experiments/decode_prof.py real
run decode_complex $R/tools/experiments/decode_prof.py complex
run shbasis $R/tools/experiments/shbasis_prof.py
run config4_r5cm $R/tools/experiments/config4_prof.py 4.8 5.0
run config4_r10cm $R/tools/experiments/config4_prof.py 9.8 10.0
run config4_r2cm $R/tools/experiments/config4_prof.py 2.0 2.2 10
run config5_batch $R/tools/experiments/config5_prof.py batch
run config5_single $R/tools/experiments/config5_prof.py single
run config3_batch $R/tools/experiments/config3_prof.py batch
(cd $R; python tools/secondary_traffic.py $tag)
