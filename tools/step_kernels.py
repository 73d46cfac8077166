"""Every kernel of one train step in launch order (name, start us, duration us) from a rocprofv3 results .db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
step = rows[adam[-3] + 1:adam[-2] + 1]
t0 = step[0][1]
for nm, st, en in step:
    nm = nm.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void at::native::', '')
    print("%9.1f %8.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, nm[:150]))
