"""Copies the judged pieces of a tools/profile.sh run (gpurun_out/prof_<tag>/ + gpurun_out/bench_<tag>.json) into profiles/:
the rocprofv3 summary, the kernel stats, the bench line and the HBM traffic of the default fused kernel (FETCH_SIZE doubled
as MI355X_MICROARCH.md prescribes for gfx950 16-byte-per-lane reads, WRITE_SIZE as is)."""
import collections, csv, json, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
d = f'gpurun_out/prof_{tag}/'


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


f, w = pmc(d + 'pmc_fetch/pmc_counter_collection.csv'), pmc(d + 'pmc_write/pmc_counter_collection.csv')
k = [n for n in f if 'k_fused' in n and 'true' in n][0]
avg = lambda a: sum(a) / len(a)
fetch, write = avg(f[k]['FETCH_SIZE']), avg(w[k]['WRITE_SIZE'])
import os
out = json.load(open(f'profiles/{tag}_traffic.json')) if os.path.exists(f'profiles/{tag}_traffic.json') else {}
out.update({'kernel': k, 'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write, 'TCC_HIT_per_launch': avg(w[k]['TCC_HIT_sum']),
            'TCC_MISS_per_launch': avg(w[k]['TCC_MISS_sum']), 'hbm_bytes_per_launch': 2 * fetch * 1024 + write * 1024,
            'launches_averaged': len(f[k]['FETCH_SIZE'])})
json.dump(out, open(f'profiles/{tag}_traffic.json', 'w'), indent=1)
# counter-derived figures bench.py copies into its `roofline` object
sq, inst = pmc(d + 'pmc_sq/pmc_counter_collection.csv'), pmc(d + 'pmc_inst/pmc_counter_collection.csv')
ks, ki = [n for n in sq if 'k_fused' in n and 'true' in n][0], [n for n in inst if 'k_fused' in n and 'true' in n][0]
cycles = avg(f[k]['GRBM_GUI_ACTIVE']) / 8.0            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
rows_stats = [r for r in csv.DictReader(open(d + 'stats/stats_kernel_stats.csv')) if r['Name'] == k]
prof_ms_all = float(rows_stats[0]['AverageNs']) / 1e6 if rows_stats else None
# what bench.py's staleness guard compares with its live HIP-event mean: the launches of the two TIMED Loops of the traced run (launches 30 .. 89: Loop 0
# is the warm-up), not the mean over all launches (which also holds the warm-up, the cold-aggregates and the PCIe-inclusive Loops: 4 - 5 % higher)
_tr = [r for r in csv.DictReader(open(d + 'stats/stats_kernel_trace.csv')) if r['Kernel_Name'] == k]
_dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in _tr]
prof_ms = sum(_dur[30:90]) / 60 if len(_dur) >= 90 else prof_ms_all
n_tiles = 31250
fig = {'kernel': k, 'hbm_bytes_per_launch': out['hbm_bytes_per_launch'],
       'mfma_busy_pct': 100.0 * avg(sq[ks]['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024 * cycles),
       'valu_busy_pct': 100.0 * 4.0 * avg(sq[ks]['SQ_ACTIVE_INST_VALU']) / (1024 * cycles),
       'lds_bank_conflict_share': avg(inst[ki]['SQ_LDS_BANK_CONFLICT']) / avg(inst[ki]['SQ_ACTIVE_INST_LDS']),
       'valu_insts_per_tile': avg(inst[ki]['SQ_INSTS_VALU']) / n_tiles, 'mfma_insts_per_tile': avg(inst[ki]['SQ_INSTS_MFMA']) / n_tiles,
       'effective_clock_ghz': cycles / (prof_ms_all * 1e6) if prof_ms_all else None, 'profiled_avg_launch_ms': prof_ms, 'profiled_avg_launch_ms_all_launches': prof_ms_all,
       'definitions': 'mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); valu_busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / '
                      'the same; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; '
                      'per launch averages over all launches of the kernel in `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`'}
json.dump(fig, open(f'profiles/{tag}_pmc.json', 'w'), indent=1)
shutil.copy(d + 'summary.txt', f'profiles/{tag}_rocprofv3_summary.txt')
shutil.copy(d + 'stats/stats_kernel_stats.csv', f'profiles/{tag}_kernel_stats.csv')
shutil.copy(f'gpurun_out/bench_{tag}.json', f'profiles/{tag}_bench_1gpu.json')
# per-Loop means of the default kernel from the kernel trace: bench.py times only its `--steps` Loops with HIP events, while the
# rocprofv3 mean over all launches also contains the warm-up Loop, the PCIe-inclusive Loop and the one-body launch
rows = [r for r in csv.DictReader(open(d + 'stats/stats_kernel_trace.csv')) if r['Kernel_Name'] == k]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
loops = [dur[i:i + 30] for i in range(0, len(dur) - len(dur) % 30, 30)]
with open(f'profiles/{tag}_fused_kernel_per_loop.txt', 'w') as fo:
    fo.write(f'{k}\n{len(dur)} launches in the rocprofv3 kernel trace of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`, mean {sum(dur) / len(dur):.4f} ms\n')
    names = ['warm-up Loop', 'timed step 1', 'timed step 2', 'exact-path comparison / cold-aggregates Loop', 'cold-aggregates Loop', 'PCIe-inclusive Loop']
    for i, l in enumerate(loops):
        fo.write(f'Loop {i} ({names[i] if i < len(names) else "extra"}): mean of its 30 launches {sum(l) / len(l):.4f} ms, min {min(l):.4f}, max {max(l):.4f}\n')
    fo.write(f'remaining launches (one-body comparison run): {[round(x, 4) for x in dur[len(loops) * 30:]]}\n')
    fo.write('bench.py reports the HIP-event mean over the timed steps only (roofline.avg_launch_ms).\n')
print(open(f'profiles/{tag}_fused_kernel_per_loop.txt').read())
for r in list(csv.DictReader(open(d + 'stats/stats_kernel_stats.csv')))[:3]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'])
print('hbm bytes per launch', out['hbm_bytes_per_launch'])
