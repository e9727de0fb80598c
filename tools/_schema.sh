cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/sch -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-train --no-exclusive > /tmp/sch.log 2>&1
python3 - <<'PY'
import sqlite3, glob
f=glob.glob('/tmp/sch/**/*_results.db', recursive=True)[0]
c=sqlite3.connect(f)
for (n,t,sql) in c.execute("select name,type,sql from sqlite_master"):
    if n=='counters_collection' or n.startswith('rocpd_info_kernel_symbol'):
        print(t, n); print((sql or '')[:2500]); print()
print(c.execute("select kernel_name, display_name from rocpd_info_kernel_symbol where kernel_name like '%pc_kernel%' limit 2").fetchall())
PY
