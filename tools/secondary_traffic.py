"""profiles/secondary_traffic.json from the per-workload summaries tools/experiments/secondary_prof.sh leaves under gpurun_out/:
for each workload the HBM bytes one execute moves -- sum over kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 x dispatches, divided by
the executes of the profiled command -- and its dominant kernel.  tools/bench_secondary.py puts these next to its own timings
(`roofline` blocks of configs 4 / 5 and binauralDecode).      python tools/secondary_traffic.py <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.csrc_hash import csrc_sha16  # noqa: E402
# workload -> (executes of the profiled command, designs per execute)
WORKLOADS = {"decode_real": (6, 1), "decode_complex": (6, 1), "shbasis": (6, 1), "config4_r5cm": (6, 8), "config4_r10cm": (6, 8),
             "config4_r2cm": (6, 8), "config5_batch": (6, 8), "config5_single": (6, 1), "config3_batch": (6, 8)}


def main():
    tag = sys.argv[1]
    out = {"note": "HBM bytes per execute = sum over kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 x dispatches / executes of the profiled "
                   "command (tools/experiments/secondary_prof.sh; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)",
           "source": tag, "csrc_sha16": csrc_sha16(),
           "stamp_note": "csrc_sha16: sha256 over emagls_amd/csrc at the time of the profile (tools/csrc_hash.py); tools/bench_secondary.py marks the "
                         "blocks stale when the tree's differs"}
    for w, (nexec, designs) in WORKLOADS.items():
        path = os.path.join(ROOT, "gpurun_out", f"{tag}_{w}.json")
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        skip = ("rocclr", "vectorized_elementwise", "at::native")   # (buffer fills / copies of the harness, not of the path)
        ks = {k: v for k, v in d.items() if not any(x in k for x in skip)}
        total = sum(v.get("bytes", 0) * v["dispatches"] for v in ks.values())
        busy = sum(v["avg_us"] * v["dispatches"] for v in ks.values())
        dom = max(ks.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["dispatches"])
        out[w] = {"executes": nexec, "designs_per_execute": designs, "bytes_per_execute": int(total / nexec),
                  "kernel_time_us_per_execute": round(busy / nexec, 1),
                  "dominant_kernel": {"name": dom[0][:90], "launches_per_execute": round(dom[1]["dispatches"] / nexec, 2), "avg_us": round(dom[1]["avg_us"], 1),
                                      "bytes_per_launch": int(dom[1].get("bytes", 0))}}
    for path in (os.path.join(ROOT, "profiles", "secondary_traffic.json"), os.path.join(ROOT, "gpurun_out", f"{tag}_secondary_traffic.json")):
        with open(path, "w") as f:     # (gpurun merges only gpurun_out/ back: copy that file over profiles/secondary_traffic.json)
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
