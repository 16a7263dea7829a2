"""Per-kernel launch footprint from a rocprofv3 --kernel-trace run (rocpd sqlite): workgroup size, number of workgroups,
LDS per workgroup (static + dynamic, as dispatched), registers, scratch -- and how many workgroups of the kernel fit a CU
that already hosts 0 / 1 / 2 workgroups of the resident sweep (LDS 160 KB, 512 VGPRs per SIMD lane, 8 waves per SIMD...).

    python tools/kernel_footprint.py <dir>
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

LDS_CU = 160 * 1024


def main():
    d = sys.argv[1]
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    want = [c for c in ("arch_vgpr_count", "accum_vgpr_count", "sgpr_count") if c in scols]
    sel = ", ".join("s." + c for c in want)
    q = (f"select s.kernel_name, d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z, "
         f"d.grid_size_x*d.grid_size_y*d.grid_size_z, d.group_segment_size, d.private_segment_size, d.end-d.start"
         f"{', ' + sel if sel else ''} from {kd} d join {ks} s on d.kernel_id=s.id")
    per = defaultdict(list)
    for r in cur.execute(q):
        per[r[0]].append(r[1:])
    sweep_lds, sweep_waves, sweep_vgpr = 0, 0, 0
    for n, rows in per.items():
        if "sweep_persist" in n or "sweep_synth" in n:
            sweep_lds = max(r[2] for r in rows)
            sweep_waves = rows[0][0] // 64
            sweep_vgpr = (rows[0][5] + rows[0][6]) if len(rows[0]) > 6 else 0
    print(f"resident sweep: LDS {sweep_lds} B / workgroup, {sweep_waves} waves, {sweep_vgpr} VGPRs (arch + acc)")
    print("| kernel | launches | avg us | wg size | workgroups | LDS B/wg | scratch B | VGPR (arch+acc) | fit per CU next to 0 / 1 / 2 sweep wgs |")
    print("|---|---|---|---|---|---|---|---|---|")

    def fit(lds, waves, vgpr, nsweep):
        lds_free = LDS_CU - nsweep * sweep_lds
        by_lds = lds_free // lds if lds else 99
        # 4 SIMDs, 8 wave slots each (CDNA3/4: 8 with <= 64 VGPRs ... 512 / vgprs), waves of a workgroup spread over the SIMDs
        slots = 0
        per_simd_used = nsweep * sweep_waves / 4.0
        if vgpr:
            gran = -(-vgpr // 8) * 8
            per_simd = min(8, 512 // gran)
            sweep_gran = -(-max(sweep_vgpr, 1) // 8) * 8
            vg_free = 512 - per_simd_used * sweep_gran
            per_simd = min(per_simd, int(vg_free // gran), int(8 - per_simd_used))
            slots = int(per_simd * 4 // max(waves, 1))
        else:
            slots = int((8 - per_simd_used) * 4 // max(waves, 1))
        return max(0, min(by_lds, slots))

    for n, rows in sorted(per.items(), key=lambda kv: -sum(r[4] for r in kv[1])):
        wg = rows[0][0]
        nwg = sorted(set(r[1] // r[0] for r in rows))
        lds = max(r[2] for r in rows)
        scr = max(r[3] for r in rows)
        vg = (rows[0][5] + rows[0][6]) if len(rows[0]) > 6 else 0
        avg = sum(r[4] for r in rows) / len(rows) / 1e3
        fits = " / ".join(str(fit(lds, wg // 64 or 1, vg, k)) for k in (0, 1, 2))
        nw = ",".join(str(x) for x in nwg[:4]) + ("..." if len(nwg) > 4 else "")
        print(f"| `{n[:60]}` | {len(rows)} | {avg:.1f} | {wg} | {nw} | {lds} | {scr} | {vg} | {fits} |")


if __name__ == "__main__":
    main()
