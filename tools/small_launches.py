"""Kernels shorter than 60 us of ONE train step from a rocprofv3 results .db (the step between the last two Adam kernels... of
the timed region): count, total, grouped by name.   python3 tools/small_launches.py prof/x_results.db"""
import sqlite3, sys, re
from collections import Counter, defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
step = rows[adam[-3] + 1:adam[-2] + 1]
small = [(re.sub(r'\(.*$', '', n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void at::native::', ''))[:70], (e - s) / 1e3)
         for n, s, e in step if (e - s) < 60000]
tot = defaultdict(float); cnt = Counter()
for n, d in small:
    tot[n] += d; cnt[n] += 1
for n in sorted(tot, key=lambda k: -tot[k]):
    print('%3d x %-70s %7.1f us' % (cnt[n], n, tot[n]))
print('step: %d kernels, %.2f ms; under 60 us: %d launches, %.1f us' % (len(step), (step[-1][2] - step[0][1]) / 1e6, len(small), sum(d for _, d in small)))
