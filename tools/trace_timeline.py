import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,queue_id,stream_id,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
t0=rows[0][3]
for r in rows:
    d=(r[4]-r[3])/1e6
    if d>100: print(f"{(r[3]-t0)/1e6:9.1f} {(r[4]-t0)/1e6:9.1f} {d:8.1f} q{r[1]} s{r[2]} grid {r[5]//r[6]} {r[0][:60]}")
