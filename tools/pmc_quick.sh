cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters spec env
  rm -rf /tmp/pq_$1; env $4 rocprofv3 --pmc $2 -d /tmp/pq_$1 -o pmc -- python $R/bench.py --pmc-child $3 > /dev/null 2>&1
  python - <<PY
import glob,sqlite3
db=glob.glob('/tmp/pq_$1/**/*.db',recursive=True)[0]
con=sqlite3.connect(db)
rows=con.execute("select name,counter_name,sum(counter_value),max(duration),dispatch_id from pmc_events group by name,counter_name,dispatch_id order by dispatch_id").fetchall()
for r in rows:
    if 'integrate_kernel' in r[0] and r[3]>2e7: print('$1', r[0][:60], r[1], '%.3e'%r[2], 'ms %.1f'%(r[3]/1e6))
PY
}
run anim_sah_w WRITE_SIZE 1920,1080,256,50,1,0,3,0,1,1,120 X=1
run anim_morton_w WRITE_SIZE 1920,1080,256,50,1,0,3,0,1,1,120 TOR_ACCEL_ORDER=morton
run anim_brute_w WRITE_SIZE 1920,1080,256,50,1,0,0,0,1,1,120 X=1
run rand_sah_w WRITE_SIZE 1920,1080,256,50,1,0,3,0,1,1,-1 X=1
run anim_nostage_w WRITE_SIZE 1920,1080,256,50,1,0,3,0,1,1,120 TOR_STAGE_LDS=0
run anim_sah_tcc "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" 1920,1080,256,50,1,0,3,0,1,1,120 X=1
run anim_sah_atom "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" 1920,1080,256,50,1,0,3,0,1,1,120 X=1
run anim_sah_wb "TCC_WRITEBACK_sum TCC_EA0_WRREQ_WRITE_DRAM_sum" 1920,1080,256,50,1,0,3,0,1,1,120 X=1
