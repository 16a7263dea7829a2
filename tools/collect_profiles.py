"""Copies what a GPU session (tools/session.sh <tag> suite full secondary pmc prof20 timing:20,32 bench:128:32) left under gpurun_out/ into
the tracked files of profiles/ (round 6's names).

    python tools/collect_profiles.py <tag>
"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
S = os.path.join(G, tag)


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("profiles/%s <- %s" % (dst, os.path.relpath(src, ROOT)))


if os.path.exists(os.path.join(S, "suite.log")):
    with open(os.path.join(P, "r06_tests.log"), "w") as f:
        f.write("# tools/session.sh %s suite: python -m pytest tests/ -x -q -m gpu --durations=25, then __graft_entry__.smoke()\n" % tag)
        f.write(open(os.path.join(S, "suite.log")).read())
        if os.path.exists(os.path.join(S, "smoke.log")):
            f.write(open(os.path.join(S, "smoke.log")).read())
    print("profiles/r06_tests.log")
cp(os.path.join(S, "full.json"), "r06_bench_steps20.json")
b128 = sorted(glob.glob(os.path.join(S, "b128_*.json")))
if b128:
    cp(b128[-1], "r06_bench_steps128.json")
for name in ("default20_kernels.md", "default20_sweep_launches.md", "fill_timeline20.md", "pmc.md"):
    cp(os.path.join(S, name), "r06_" + name)
cp(os.path.join(S, "pmc_traffic.json"), "pmc_traffic.json")
if os.path.exists(os.path.join(S, "sweep_timing_20.log")):
    with open(os.path.join(P, "r06_sweep_timing.md"), "w") as f:
        f.write("# Round 6: in-kernel stamps of the register-resident sweep (tools/sweep_timing.py 20 / 32; EMAGLS_SWEEP_TIMING=1; session %s)\n\n```\n" % tag)
        for n in ("20", "32"):
            q = os.path.join(S, "sweep_timing_%s.log" % n)
            if os.path.exists(q):
                f.write("".join(l for l in open(q) if "amdgpu.ids" not in l))
        f.write("```\n")
    print("profiles/r06_sweep_timing.md")
for src in sorted(glob.glob(os.path.join(G, tag + "_*.md"))):
    cp(src, "r06_secondary_" + os.path.basename(src)[len(tag) + 1:])
cp(os.path.join(G, tag + "_secondary_traffic.json"), "secondary_traffic.json")
