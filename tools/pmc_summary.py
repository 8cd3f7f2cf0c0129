"""Summarise rocprofv3 --pmc passes of bench.py for one kernel into profiles/rNN/pmc_bench_<tag>.json.
   python tools/pmc_summary.py <kernel substring> <out.json> <pass dir> [<pass dir> ...]
Every pass directory is searched for *counter_collection.csv; a counter's value for a dispatch is the sum of its
rows (rocprofv3 emits one row per instance), the summary is the mean over the kernel's dispatches.
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are emitted in kilobytes (x 1024);
on gfx950 FETCH_SIZE counts a 128-B request as 64 B and is doubled."""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    kernel_sub, out_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
    kname = None
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path, newline='') as f:
                for row in csv.DictReader(f):
                    if kernel_sub not in row['Kernel_Name']:
                        continue
                    kname = row['Kernel_Name']
                    per[row['Counter_Name']][(path, row['Dispatch_Id'])] += float(row['Counter_Value'])
    if not per:
        raise SystemExit('no rows for kernel %r' % kernel_sub)
    mean = {c: sum(v.values()) / len(v) for c, v in per.items()}
    n_disp = {c: len(v) for c, v in per.items()}
    der = {}
    if 'FETCH_SIZE' in mean:
        der['hbm_read_bytes_gfx950_corrected'] = mean['FETCH_SIZE'] * 1024.0 * 2.0
    if 'WRITE_SIZE' in mean:
        der['hbm_write_bytes'] = mean['WRITE_SIZE'] * 1024.0
    if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
        der['hbm_bytes_per_launch'] = der['hbm_read_bytes_gfx950_corrected'] + der['hbm_write_bytes']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean and 'GRBM_GUI_ACTIVE' in mean:
        # busy cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs -> x 128
        der['mfma_util'] = mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * mean['GRBM_GUI_ACTIVE'])
    if 'SQ_WAVE_CYCLES' in mean and 'SQ_WAIT_INST_ANY' in mean:
        der['wave_wait_inst_frac'] = mean['SQ_WAIT_INST_ANY'] / mean['SQ_WAVE_CYCLES']
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import srchash
    json.dump({'kernel': kname, 'csrc_sha16': srchash.source_hash(), 'csrc_files': list(srchash.HAND_EVAL_SOURCES),
               'dispatches': n_disp, 'counters_mean_per_dispatch': mean, 'derived': der,
               'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate --pmc passes'},
              open(out_path, 'w'), indent=1)
    print(json.dumps(der, indent=1))

if __name__ == '__main__':
    main()
