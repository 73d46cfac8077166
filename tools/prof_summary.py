"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv (top kernels by total time)."""
import csv, glob, sys
path = sys.argv[1]
f = glob.glob(path + '/*/*kernel_stats.csv')[0] if not path.endswith('.csv') else path
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    print("%-60s calls %6s tot %8.2f ms avg %8.2f us  %5.1f%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('total kernel ms %.2f' % (tot / 1e6))
