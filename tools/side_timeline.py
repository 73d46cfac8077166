"""One cfg-5 step from a rocprofv3 results .db: every kernel of the step's BACKWARD with start / end / stream, the side-stream
products marked, so that what runs beside which chain can be read off."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "select name, start, end, %s from kernels order by start" % ("stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0"))
rows = list(db.execute(q))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
step = rows[adam[-3] + 1:adam[-2] + 1]
t0 = step[0][1]
main_stream = max(set(r[3] for r in step), key=lambda s: sum(1 for r in step if r[3] == s))
for nm, st, en, sid in step:
    nm = nm.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void at::native::', '')
    dur = (en - st) / 1e3
    if sid != main_stream or dur > 40 or 'lstm_persist' in nm or 'dec_persist' in nm:
        print("%9.1f -> %9.1f  %8.1f us  %s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, dur, "SIDE" if sid != main_stream else "    ", nm[:90]))
print("step span %.2f ms, %d kernels, %d on the side stream" % ((step[-1][2] - t0) / 1e6, len(step), sum(1 for r in step if r[3] != main_stream)))
