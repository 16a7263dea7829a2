# throughput of bench.py over (resident batches, designs per batch); usage: bash tools/exp_slots.sh "4x8 2x16 3x16 4x16"
cd $GRAFT_REPO_ROOT
for cfg in ${1:-4x8 2x16 3x16}; do
  sl=${cfg%x*}; bs=${cfg#*x}
  timeout 600 python bench.py --steps 192 --warmup 48 --slots $sl --batch $bs --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/exp_${cfg}.json 2> gpurun_out/exp_${cfg}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/exp_${cfg}.json"))
    print("${cfg}", round(d["value"],1), "sets/s; sweep us/bin", round(d["roofline"]["us_per_bin"],3), "avg launch us", round(d["roofline"]["avg_launch_us"],1))
except Exception as e:
    print("${cfg} failed", e); print(open("gpurun_out/exp_${cfg}.err").read()[-800:])
PY
done
