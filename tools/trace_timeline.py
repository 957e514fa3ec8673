import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,queue_id,stream_id,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
t0=rows[0][3]
# group finder kernels between encode launches
cur=None; acc={}
for r in rows:
    s=(r[3]-t0)/1e6; e=(r[4]-t0)/1e6; nm=r[0]
    if 'encode_kernel' in nm: print(f"E   {s:9.1f} {e:9.1f} {e-s:8.1f} grid {r[5]//r[6]}"); continue
    if 'mf_scan' in nm: print(f"  scan {s:9.1f} {e:9.1f} {e-s:8.1f}"); continue
    if e-s>30: print(f"    k  {s:9.1f} {e:9.1f} {e-s:8.1f} {nm[:70]}")
