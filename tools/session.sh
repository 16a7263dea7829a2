#!/bin/bash
# One GPU-box session, assembled from steps (round 6: replaces the per-session scripts tools/r5_*.sh).  Through gpurun:
#     gpurun --timeout 2400 -- 'bash tools/session.sh <tag> <step> [<step> ...]'
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.  Steps (':' separates a step from its arguments,
# ',' the items of a list; ENV=VALUE items are exported for that step only):
#   tests:<file-or--k-expr>[,...]      pytest -m gpu -x -q -s on the given test files (a name without '/' is a -k expression)
#   suite                              the whole gpu suite the way the round-end driver runs it, then smoke()
#   bench:<steps>:<warmup>[:ENV=V,...] bench.py without the secondary figures; one summary line (value, ms/step, sweep launch, frac)
#   full                               bench.py as the driver runs it (--steps 20 --warmup 5, everything)
#   timing:<n>[,<n>...]                in-kernel stamps of the resident sweep for launches of n designs (tools/sweep_timing.py)
#   prof20                             rocprofv3 kernel trace of the driver's command: fill timeline, kernel averages, sweep launches
#   pmc                                the PMC passes of tools/gpu_round.sh (32-design chunks) -> pmc.md, pmc_traffic.json
#   cold                               first-call traces: config 4 rank share on new radii, config 3 with 256 host-array jobs
#   secondary                          rocprofv3 + PMC passes of the secondary workloads (tools/experiments/secondary_prof.sh)
#   fuzz:<n>:<seed>                    random campaign (tools/fuzz_random.py)
#   py:<script>[:args,...]             any script under tools/
tag=${1:-s}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/$tag; mkdir -p $O
export EMAGLS_BUILD_TAG=$tag
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
summ() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f, "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "sweep_us", round(r.get("avg_launch_us", 0)), "designs/launch", r.get("designs_per_launch"),
          "frac", round(r.get("frac", 0), 3), "parity", (d.get("parity") or {}).get("rel_complex_error"))
except Exception as e:
    print(f, "FAILED", e); print(open(f.replace(".json", ".err")).read()[-1200:])
PY
}
for step in "$@"; do
  IFS=':' read -r name a1 a2 a3 <<< "$step"
  case $name in
    tests)
      files=(); kexpr=""
      IFS=',' read -ra items <<< "$a1"
      for it in "${items[@]}"; do if [[ $it == */* || $it == *.py ]]; then files+=("$it"); else kexpr="$it"; fi; done
      [ ${#files[@]} -eq 0 ] && files=(tests)
      timeout 2400 python -m pytest "${files[@]}" -m gpu -x -q -s ${kexpr:+-k "$kexpr"} > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
      grep -h "rel = \|worst rel\|microphones at\|40-digit" $O/tests.log | cut -c1-220 | tail -30; tail -4 $O/tests.log | cut -c1-300 ;;
    suite)
      timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/suite.log 2>&1; echo "pytest rc=$?" | tee -a $O/suite.log; tail -30 $O/suite.log | cut -c1-200
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log ;;
    bench)
      envs=(); IFS=',' read -ra items <<< "$a3"; nm=b${a1}; for it in "${items[@]}"; do [ -n "$it" ] && envs+=("$it") && nm=${nm}_${it//[^A-Za-z0-9]/}; done
      n=1; while [ -e $O/${nm}_$n.json ]; do n=$((n+1)); done
      env "${envs[@]}" timeout 900 python bench.py --steps $a1 --warmup ${a2:-5} $B > $O/${nm}_$n.json 2> $O/${nm}_$n.err; summ $O/${nm}_$n.json ;;
    full)
      timeout 1200 python bench.py --steps 20 --warmup 5 > $O/full.json 2> $O/full.err; echo "bench rc=$?"; summ $O/full.json; tail -2 $O/full.err ;;
    timing)
      IFS=',' read -ra items <<< "$a1"
      for n in "${items[@]}"; do timeout 300 python tools/sweep_timing.py $n > $O/sweep_timing_$n.log 2>&1; cut -c1-200 $O/sweep_timing_$n.log | tail -25; done ;;
    prof20)
      ( export TMPDIR=/tmp; cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 $B > $R/$O/prof_default20.log 2>&1 )
      python tools/fill_timeline.py $O/prof_default20 2 > $O/fill_timeline20.md 2>&1
      python tools/kernel_avgs.py $O/prof_default20 > $O/default20_kernels.md 2>&1
      python tools/sweep_launches.py $O/prof_default20 $O/prof_default20.log > $O/default20_sweep_launches.md 2>&1
      rm -rf $O/prof_default20; tail -3 $O/default20_sweep_launches.md ;;
    pmc)
      ( export TMPDIR=/tmp; cd /tmp
        PMCCMD="python3 $R/bench.py --steps 128 --warmup 0 --slots 1 --batch 32 $B"
        for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES; do
          timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$c -o pmc -- $PMCCMD > $R/$O/pmc_$c.log 2>&1
        done )
      python tools/pmc_summary.py $O/pmc_traffic.json $O/pmc.md $O/pmc_* > $O/pmc_summary.log 2>&1
      rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*; head -40 $O/pmc.md ;;
    cold)
      EMAGLS_JOBS_TRACE=1 timeout 600 python -c "
from tools import bench_secondary as B
import json
print(json.dumps(B.config4_rank_share_runner(reps=2)))" > $O/cold_config4.log 2> $O/cold_config4.trace
      tail -1 $O/cold_config4.log | cut -c1-500
      EMAGLS_JOBS_TRACE=1 timeout 600 python tools/experiments/jobs_host_arrays.py > $O/cold_config3.log 2> $O/cold_config3.trace; head -10 $O/cold_config3.log ;;
    secondary)
      bash tools/experiments/secondary_prof.sh $tag > $O/secondary_prof.log 2>&1; tail -20 $O/secondary_prof.log ;;
    fuzz)
      timeout 3000 python tools/fuzz_random.py ${a1:-60} ${a2:-1} > $O/fuzz_${a2:-1}.log 2>&1; grep -v " ok rel" $O/fuzz_${a2:-1}.log | cut -c1-400 | tail -30 ;;
    py)
      IFS=',' read -ra items <<< "$a2"
      timeout 2400 python tools/$a1 "${items[@]}" > $O/$(basename $a1 .py).log 2>&1; tail -30 $O/$(basename $a1 .py).log | cut -c1-300 ;;
    *) echo "unknown step $step" ;;
  esac
done
