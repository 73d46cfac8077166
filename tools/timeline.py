"""Print one train step's kernel timeline from a rocprofv3 results .db: big kernels individually, torch glue grouped,
with the idle gap before each entry."""
import sqlite3, sys, re
from collections import Counter
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # consecutive steps to print (default: one)
step = rows[adam[-2 - nsteps] + 1:adam[-2] + 1]
big = ('lstm_persist', 'gemm_f32', 'gemm_bf3', 'gemm_bf6', 'dec_persist', 'att_m', 'pyramid', 'colsum', 'adam', 'sumsq')
t0 = step[0][1]; prev_end = t0; groups = []; cur = []
for nm, st, en in step:
    nm = nm.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void at::native::', '')
    gap = max(0, st - prev_end) / 1e3; prev_end = max(prev_end, en)
    ent = ((st - t0) / 1e3, (en - st) / 1e3, nm, gap)
    if any(b in nm for b in big):
        if cur: groups.append(cur); cur = []
        groups.append([ent + ('BIG',)])
    else:
        cur.append(ent + ('s',))
if cur: groups.append(cur)
idle = 0
for g in groups:
    gp = sum(x[3] for x in g); idle += gp
    if g[0][4] == 'BIG':
        print("%9.0f  gap %6.1f  %-50s %8.1f" % (g[0][0], gp, g[0][2][:50], g[0][1]))
    else:
        c = Counter(x[2][:34] for x in g)
        print("%9.0f  gap %6.1f    [%d small, %.0f us] " % (g[0][0], gp, len(g), sum(x[1] for x in g)) + "; ".join("%s x%d" % kv for kv in c.most_common(5)))
print("step span %.2f ms, kernels %d, idle %.2f ms" % ((step[-1][2] - t0) / 1e6, len(step), idle / 1e3))
