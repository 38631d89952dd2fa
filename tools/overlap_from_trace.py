"""Copy / kernel overlap from a rocprofv3 --kernel-trace of tools/bench_slice.py (loopback group on one GPU: the transfers between ranks
are device-to-device copies, which the runtime executes as __amd_rocclr_copyBuffer kernels).
Usage: python tools/overlap_from_trace.py <dir with *_kernel_trace.csv>
Prints the per-kernel average durations, the share of the copy time during which a compute kernel was running, and the device
time per rank and iteration from the first pack kernel to the last operation."""
import csv, glob, os, sys
from collections import defaultdict
import numpy as np
d = sys.argv[1]
kf = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
assert kf, 'no kernel trace under ' + d
with open(kf[-1]) as fh:
    K = list(csv.DictReader(fh))
ks = np.array([[int(r['Start_Timestamp']), int(r['End_Timestamp'])] for r in K])
names = np.array([r['Kernel_Name'] for r in K])
packs = np.array(['k_slice_pack' in n for n in names])
t_begin = ks[packs][:, 0].min()
keep = ks[:, 0] >= t_begin
ks, names = ks[keep], names[keep]
is_copy = np.array(['copyBuffer' in n for n in names])
agg = defaultdict(list)
for (s, e), n in zip(ks, names): agg[n[:72]].append(e - s)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:7]:
    print(f'  {n:72s} {len(v):6d} calls  {np.mean(v) / 1e3:8.1f} us avg')
iv = sorted((s, e) for (s, e), c in zip(ks, is_copy) if not c)
merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
ms = np.array([m[0] for m in merged]); me = np.array([m[1] for m in merged])
def covered(s, e):
    tot = 0
    for i in range(np.searchsorted(me, s, 'right'), len(ms)):
        if ms[i] >= e: break
        tot += min(e, me[i]) - max(s, ms[i])
    return tot
cs = ks[is_copy]
ct = int(sum(e - s for s, e in cs)); co = int(sum(covered(s, e) for s, e in cs))
print(f'copies between ranks: {len(cs)} copies, {ct / 1e6:.2f} ms in all, {co / 1e6:.2f} ms ({100 * co / max(ct, 1):.0f} %) of it while a compute kernel was running')
n_ri = int(packs.sum())
span = ks[:, 1].max() - t_begin
print(f'device time from the first pack kernel to the last operation: {span / 1e6:.2f} ms = {span / 1e3 / n_ri:.1f} us per rank and iteration '
      f'({n_ri} rank-iterations, one after the other on one GPU); compute kernels busy {sum(e - s for s, e in merged) / 1e6:.2f} ms')
