cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sweep_variants or batch_of or lane_batch or config3_full or emagls2_filters_thin" 2>&1 | tail -3
for nb in 1 2; do
  EMAGLS_SWEEP_NBUF=$nb timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/exp_nbuf${nb}.json 2> gpurun_out/exp_nbuf${nb}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_nbuf${nb}.json"))
print("NBUF=${nb}", round(d["value"],1), "sets/s; single", d["single_design_latency_ms"], "ms; sweep us/bin", round(d["roofline"]["us_per_bin"],3), "single-design sweep", round(d["roofline"]["avg_launch_us_single_design"],1))
PY
done
