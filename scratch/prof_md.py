"""rocprofv3 sqlite -> markdown kernel table: python scratch/prof_md.py <db> "<title>" "<command>" > profiles/x.md"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
ncalls = [r for r in rows if 'sh_coeff' in r[0]]
nexec = ncalls[0][1] if ncalls else 1
print(f"# {sys.argv[2]}\n\n`{sys.argv[3]}`\n\n{nexec} executes of the design are in the trace. Durations in microseconds.\n")
print("| kernel | calls | calls / design | total us | avg us | us / design | % |\n|---|---|---|---|---|---|---|")
tot = 0.0
for n, c, t, a, p in rows[:40]:
    tot += t / nexec
    print(f"| `{n[:110]}` | {c} | {c / nexec:.1f} | {t:.1f} | {a:.3f} | {t / nexec:.1f} | {p:.2f} |")
print(f"\nSum of the rows above: {tot:.1f} us per design.")
