"""Per-kernel totals from a rocprofv3 rocpd database (t_results.db), split at the largest idle gaps into phases.
usage: python tools/summarize_db.py DB [n_phases]"""
import sqlite3, re, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
nph = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gaps = sorted(((rows[i + 1][1] - rows[i][2], i + 1) for i in range(len(rows) - 1)), reverse=True)[:nph - 1]
cuts = [0] + sorted(i for _, i in gaps) + [len(rows)]
for a, b in zip(cuts, cuts[1:]):
    rs = rows[a:b]
    d = collections.defaultdict(lambda: [0, 0])
    for n, s, e in rs:
        n = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:48]
        d[n][0] += 1; d[n][1] += e - s
    busy = sum(v[1] for v in d.values())
    print(f'phase: {len(rs)} launches, busy {busy / 1e6:.3f} ms, span {(rs[-1][2] - rs[0][1]) / 1e6:.3f} ms')
    for n, v in sorted(d.items(), key=lambda x: -x[1][1])[:22]:
        print('  %-48s %6d %9.3f ms  %6.1f us avg' % (n, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
