"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc pass of tools/pmc_mfma_probe.py:
    python3 tools/pmc_mfma_summary.py OUT/.../m_counter_collection.csv ARITH profiles/r03_pmc_mfma.json
mfma_busy_frac(kernel) = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
reports the sum over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back).  SQ_VALU_MFMA_BUSY_CYCLES counts, per SIMD, the cycles
its matrix pipe is occupied (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32), summed over the chip: the fraction is the
share of the chip's matrix-pipe cycles the kernel fills - with six (bf16x6) or three (bf16x3) MFMA products per algorithmic
product, all of them counted as busy.  Results are merged into the json under the arithmetic's name."""
import csv, json, os, re, sys, collections
path, arith, out = sys.argv[1:4]
disp = collections.OrderedDict()
for row in csv.DictReader(open(path)):
    d = disp.setdefault(int(row['Dispatch_Id']), dict(name=row['Kernel_Name'], grid=row.get('Grid_Size'), c={}))
    d['c'][row['Counter_Name']] = d['c'].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', n)
def frac(ds):
    busy = sum(d['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in ds)
    cyc = sum(d['c'].get('GRBM_GUI_ACTIVE', 0.0) for d in ds) / 8.0
    return (busy / (1024.0 * cyc) if cyc > 0 else None), busy, cyc
ds = list(disp.values())
gemm_idx = [i for i, d in enumerate(ds) if 'gemm_' in d['name'] and 'kernel' in d['name']]
gate = gemm_idx[-8:]                               # the probe's trailing gate GEMMs, in its order
step = [d for i, d in enumerate(ds) if i < gate[0]]
by_name = collections.OrderedDict()
for d in step:
    by_name.setdefault(short(d['name']), []).append(d)
kern = {}
for n, lst in by_name.items():
    f, busy, cyc = frac(lst)
    if busy > 0:
        kern[n] = dict(mfma_busy_frac=round(f, 4), dispatches=len(lst), kernel_cycles_per_dispatch=round(cyc / len(lst)),
                       mfma_busy_cycles_per_simd_per_dispatch=round(busy / 1024.0 / len(lst)))
labels = ['gemm/0/in-proj', 'gemm/0/dW_ih', 'gemm/1/in-proj', 'gemm/1/dX', 'gemm/1/dW_ih', 'gemm/2/in-proj', 'gemm/2/dX', 'gemm/2/dW_ih']
keys, gate_rows = {}, {}
for lab, i in zip(labels, gate):
    f, busy, cyc = frac([ds[i]])
    keys[lab] = round(f, 4)
    gate_rows[lab] = dict(kernel=short(ds[i]['name']), mfma_busy_frac=round(f, 4), kernel_cycles=round(cyc))
def fam(pred):
    sel = [d for d in step if pred(short(d['name']))]
    return round(frac(sel)[0], 4) if sel and frac(sel)[0] is not None else None
keys['lstm_fwd'] = fam(lambda n: n.startswith('lstm_persist_fwd'))
keys['lstm_bwd'] = fam(lambda n: n.startswith('lstm_persist_bwd'))
keys['gemm_tn'] = fam(lambda n: n.startswith('gemm_') and '<false, false' in n)
keys['dec_fwd'] = fam(lambda n: n.startswith('dec_persist_fwd'))
keys['dec_bwd'] = fam(lambda n: n.startswith('dec_persist_bwd'))
res = json.load(open(out)) if os.path.exists(out) else {}
res['_how'] = ('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/pmc_mfma_probe.py ARITH; '
               'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); two cfg-2 train steps + the encoder gate GEMMs alone')
res[arith] = dict(kernels_of_the_train_step=kern, encoder_gate_gemm=gate_rows, bench_keys=keys)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res[arith], indent=1))
