# usage: bash tools/exp_timeline.sh <tag> [slots] [batch]  -> gpurun_out/<tag>_timeline.md (kernel-trace of the bench, steady-state window)
tag=${1:-tl}; sl=${2:-4}; bs=${3:-8}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_trace -o bench -- python3 $R/bench.py --steps 128 --warmup 0 --slots $sl --batch $bs --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_trace.log 2>&1
cd $R
python tools/timeline.py gpurun_out/${tag}_trace 10 > gpurun_out/${tag}_timeline.md 2>&1
rm -rf gpurun_out/${tag}_trace
head -34 gpurun_out/${tag}_timeline.md
