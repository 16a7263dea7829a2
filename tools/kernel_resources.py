"""Per-kernel resources of the built code objects (VGPRs, LDS, workgroup size) and whether a workgroup can share a CU with a
resident sweep workgroup (4 waves x 224 VGPRs, 77 KB LDS): what decides if a kernel of another batch runs next to a sweep or
waits for it to end.    python tools/kernel_resources.py
"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SWEEP_VGPR, SWEEP_LDS = 224, 77 * 1024


def main():
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for obj in sorted(glob.glob(os.path.join(ROOT, "emagls_amd", "build", "*.o"))):
            out = os.path.join(td, os.path.basename(obj) + ".co")
            fat = os.path.join(td, os.path.basename(obj) + ".fat")
            r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
            if r.returncode or not os.path.exists(fat):
                continue
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={out}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode or not os.path.exists(out) or os.path.getsize(out) == 0:
                continue
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\." + key + r":\s+(\S+)", blk)
                    return m.group(1) if m else "0"
                name = f("name").strip("'")
                rows.append((os.path.basename(obj)[:-2], name, int(f("vgpr_count")), int(blk.split()[0]), int(f("group_segment_fixed_size")),
                             int(f("max_flat_workgroup_size")), int(f("private_segment_fixed_size")), f("uses_dynamic_stack") == "true" or "dynamic" in blk))
    print("| file | kernel | VGPR | AGPR | static LDS | max WG | scratch | next to a sweep WG (static LDS only) |")
    print("|---|---|---|---|---|---|---|---|")
    for file, name, v, a, lds, wg, scr, _ in rows:
        waves_per_simd = max(1, wg // 64 // 4)
        alloc = (v + a + 7) // 8 * 8
        fits = waves_per_simd * alloc + SWEEP_VGPR <= 512 and lds + SWEEP_LDS <= 160 * 1024
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("emagls::", "").replace("(anonymous namespace)::", "")
        print(f"| {file} | `{dem[:60]}` | {v} | {a} | {lds} | {wg} | {scr} | {'yes' if fits else 'NO'} |")


if __name__ == "__main__":
    main()
