"""Per-kernel sums of every counter in a rocprofv3 counter_collection csv (last dispatch of each kernel name):
python3 tools/pmc_table.py OUT/.../x_counter_collection.csv [name filter]"""
import csv, re, sys, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ''
disp = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    d = disp.setdefault(int(row['Dispatch_Id']), dict(name=row['Kernel_Name'], c={}))
    d['c'][row['Counter_Name']] = d['c'].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
last = collections.OrderedDict()
for d in disp.values():
    if flt in d['name']:
        last[d['name']] = d
for n, d in last.items():
    print(re.sub(r'\(.*$', '', n.replace('void (anonymous namespace)::', '')))
    for k, v in sorted(d['c'].items()):
        print('   %-32s %16.0f' % (k, v))
