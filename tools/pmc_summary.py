"""HBM-side traffic of the persistent LSTM kernels from two rocprofv3 --pmc passes of tools/pmc_probe.py (FETCH_SIZE and
WRITE_SIZE, each its own run: the two do not fit one pass), as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes:
counter values are KB, FETCH_SIZE is DOUBLED on gfx950 (it tallies 128-byte requests at 64 bytes).
    python3 tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [T]
Copies the two lstm_persist rows of each csv next to the json (profiles/r0N_pmc_{FETCH,WRITE}_SIZE_lstm_persist.csv)."""
import csv, json, os, sys
fcsv, wcsv, out = sys.argv[1:4]
T = int(sys.argv[4]) if len(sys.argv) > 4 else 800
B, H, ndir = 32, 512, 2
def rows(path, counter):
    r = {}
    keep = []
    for row in csv.DictReader(open(path)):
        if 'lstm_persist' in row['Kernel_Name'] and row['Counter_Name'] == counter:
            kind = 'bwd' if 'bwd' in row['Kernel_Name'] else 'fwd'
            name = row['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0]
            r[kind] = (name, float(row['Counter_Value']))
            keep.append(row)
    return r, keep
f, fk = rows(fcsv, 'FETCH_SIZE'); w, wk = rows(wcsv, 'WRITE_SIZE')
# algorithmic bytes per time step (both directions, 32 rows): forward = pre-activations read + activated gates written
# (2 x 16 B x 4H/4 ... = 2 x B x ndir x 4H x 4) + c, y written; backward = gates read + dG written + dy, c read + the h rows of dW_hh
# (bf16x6, the default: dW_hh is a GEMM after the kernel, so the backward does not read the h rows: PMC_FUSED_DW=0)
fused = os.environ.get('PMC_FUSED_DW', '0') == '1'
alg = {'fwd': B * ndir * (2 * 4 * H + 2 * H) * 4, 'bwd': B * ndir * (2 * 4 * H + 2 * H + (H if fused else 0)) * 4}
# packed rows (tools/pmc_probe.py's default since round 6: the instantiation the step runs): the algorithmic bytes are those of
# the batch's VALID (utterance, frame) pairs, not B x T - PMC_VALID_PAIRS, printed by the probe
pairs = int(os.environ.get('PMC_VALID_PAIRS', '0'))
if pairs:
    alg = {k: v / float(B) * pairs / T for k, v in alg.items()}          # per "time step" of the launch's T = max length
res = {'shape': 'T=%d B=%d H=%d ndir=%d (cfg-2 encoder layer 0%s); separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                'passes of tools/pmc_probe.py, FETCH_SIZE doubled as the gfx950 guide prescribes'
                % (T, B, H, ndir, '; PACKED rows with the bench batch\'s lengths, %d valid (utterance, frame) pairs' % pairs if pairs else '')}
for kind, key in (('bwd', 'lstm_persist_bwd_kernel<512>'), ('fwd', 'lstm_persist_fwd_kernel<512>')):
    hbm = (2.0 * f[kind][1] + w[kind][1]) * 1024.0
    res[key] = {'kernel': f[kind][0], 'FETCH_SIZE_KB': f[kind][1], 'WRITE_SIZE_KB': w[kind][1], 'hbm_side_bytes_per_launch': hbm,
                'hbm_side_bytes_per_time_step': round(hbm / T, 2), 'algorithmic_bytes_per_time_step': alg[kind]}
json.dump(res, open(out, 'w'), indent=1)
d = os.path.dirname(out) or '.'
pre = os.path.basename(out).split('_pmc_')[0]
for nm, keep in (('FETCH_SIZE', fk), ('WRITE_SIZE', wk)):
    with open(os.path.join(d, '%s_pmc_%s_lstm_persist.csv' % (pre, nm)), 'w', newline='') as fh:
        wr = csv.DictWriter(fh, fieldnames=list(keep[0].keys())); wr.writeheader(); wr.writerows(keep)
print(json.dumps(res, indent=1))
