import sqlite3, sys
con=sqlite3.connect(sys.argv[1]); cur=con.cursor()
rows=list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
ncalls=[r for r in rows if 'sh_coeff' in r[0]][0][1]
tot=0
for n,c,t,a,p in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    per=t/ncalls; tot+=per
    print(f"{n[:64]:64s} x{c/ncalls:6.1f} avg {a:8.2f} us  per-design {per:8.1f} us")
print('sum', round(tot,1))
