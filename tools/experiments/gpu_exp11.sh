#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_l}
for jr in 1 2 4; do for st in "20 5" "128 32"; do set -- $st
  EMAGLS_JACOBI_RUN=$jr timeout 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_jr${jr}_k$1.json 2> gpurun_out/${tag}_jr${jr}_k$1.err
  echo "jacobi run $jr steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_jr${jr}_k$1.json | cut -c1-12)"
done; done
for sl in 2 3 5 6; do for st in "20 5" "128 32"; do set -- $st
  timeout 200 python bench.py --steps $1 --warmup $2 --slots $sl --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_sl${sl}_k$1.json 2> gpurun_out/${tag}_sl${sl}_k$1.err
  echo "slots $sl steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_sl${sl}_k$1.json | cut -c1-12)"
done; done
