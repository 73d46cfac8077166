"""HBM-side traffic and rate of the persistent decoder kernels from the two --pmc passes of tools/pmc_probe_dec.py
(FETCH_SIZE doubled on gfx950, counter values are KB; kernel durations from the same csv's timestamps):
    python3 tools/pmc_summary_dec.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json B Tp L"""
import csv, json, sys
fcsv, wcsv, out = sys.argv[1:4]
B, Tp, L = (int(v) for v in sys.argv[4:7])
A = D = O = 512; C = 10
def rows(path, counter):
    r = {}
    for row in csv.DictReader(open(path)):
        n = row['Kernel_Name']
        if row['Counter_Name'] != counter: continue
        kind = 'fwd' if 'dec_persist_fwd' in n else 'bwd' if 'dec_persist_bwd' in n else 'att_m' if 'att_m_kernel' in n else None
        if kind: r[kind] = (n.replace('void (anonymous namespace)::', '').split('(')[0], float(row['Counter_Value']),
                            (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3)
    return r
f, w = rows(fcsv, 'FETCH_SIZE'), rows(wcsv, 'WRITE_SIZE')
# algorithmic bytes per decoder step: the saved scores S [B][Tp][A] dominate (written by the forward, read by att_m and by
# the backward), then fconv / ws / energy, gates, X rows
alg = {'fwd': B * Tp * A * 4 + B * (C * Tp + 2 * Tp + 4 * D + D + 2 * (D + O + 128)) * 4,
       'att_m': B * Tp * A * 4 + B * C * Tp * 4,
       'bwd': B * Tp * A * 4 + B * (2 * C * Tp + 2 * Tp + 8 * D + 2 * D + 2 * (D + O + 128) + A) * 4}
res = {'shape': 'B=%d Tp=%d L=%d D=A=O=512 C=10 (cfg-5 decoder when B=8, Tp=200): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                'tools/pmc_probe_dec.py, FETCH_SIZE doubled as the gfx950 guide prescribes; us = kernel duration in the FETCH pass' % (B, Tp, L)}
for kind in ('fwd', 'att_m', 'bwd'):
    hbm = (2.0 * f[kind][1] + w[kind][1]) * 1024.0
    us = f[kind][2]
    res[kind] = {'kernel': f[kind][0], 'FETCH_SIZE_KB': f[kind][1], 'WRITE_SIZE_KB': w[kind][1], 'us': round(us, 1),
                 'hbm_side_bytes_per_decoder_step': round(hbm / L, 1), 'algorithmic_bytes_per_decoder_step': alg[kind],
                 'achieved_GBps_hbm_side': round(hbm / us / 1e3, 1), 'achieved_GBps_algorithmic': round(alg[kind] * L / us / 1e3, 1)}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
