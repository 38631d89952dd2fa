"""Scan hipcc's ISA listings for the pattern that bit k_small16: an MFMA as the LAST matrix instruction of a basic block,
the block ending in a branch, and a successor block that reads one of the MFMA's destination registers within its first few instructions
- across that edge hipcc (ROCm 7.2) did not insert the wait states an MFMA result needs before a non-MFMA read.
Usage: python tools/scan_mfma_hazard.py [--fail] file.s [...]   (listings from `hipcc -S --cuda-device-only`)
Prints the suspicious edges (kernel, line, MFMA, reading instruction, distance).  A distance of >= 12 issued instructions is taken as safe
for the 8-pass instructions used here (16x16x4 f32, 32x32x16 bf16); the 16-pass 32x32x2 f32 needs 19."""
import re, sys

REG = re.compile(r'\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b')

def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out

def scan(path):
    lines = open(path, errors='replace').read().split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    hits, kernel = [], '?'
    def instrs(start, limit):
        out, i = [], start
        while i < len(lines) and len(out) < limit:
            t = lines[i].strip()
            if t and not t.startswith((';', '.', '#')) and not t.endswith(':'):
                out.append((i, t))
                if t.startswith(('s_branch', 's_endpgm')): break
            i += 1
        return out
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m: kernel = m.group(1)[:90]
        t = l.strip()
        if not t.startswith('v_mfma'): continue
        need = 19 if ('32x32x2_f32' in t or '32x32x1_' in t or '32x32x4_' in t) else 12
        dst = regs(t.split(None, 1)[1].split(',')[0])
        # walk forward to the end of the block: at most `need` instructions, stop at another MFMA writing/reading (the hardware interlocks those)
        ins = instrs(i + 1, need)
        for pos, (j, u) in enumerate(ins):
            if u.startswith('v_mfma'): break
            ops = u.split(None, 1)[1] if ' ' in u else ''
            if u.startswith('s_cbranch') or u.startswith('s_branch'):
                succ = []
                tgt = ops.strip()
                if tgt in labels: succ.append(labels[tgt])
                if u.startswith('s_cbranch'): succ.append(j + 1)
                for s0 in succ:
                    for pos2, (j2, u2) in enumerate(instrs(s0, need - pos)):
                        if u2.startswith('v_mfma'): break
                        ops2 = u2.split(None, 1)[1] if ' ' in u2 else ''
                        srcs = regs(','.join(ops2.split(',')[1:])) if not u2.startswith(('ds_write', 'global_store', 'buffer_store', 'v_accvgpr_write')) else regs(ops2)
                        if u2.startswith('v_accvgpr_write'): srcs = set()
                        if srcs & dst and not u2.startswith('s_nop'):
                            waits = sum(int(x.split()[1]) + 1 for _, x in ins[:pos] if x.startswith('s_nop')) + sum(int(x.split()[1]) + 1 for _, x in instrs(s0, pos2 + 1)[:pos2] if x.startswith('s_nop'))
                            if pos + pos2 + waits < need:
                                hits.append((kernel, i + 1, t, j2 + 1, u2, pos + pos2 + waits))
                            break
                break
    return hits

tot = 0
fail = '--fail' in sys.argv[1:]          # `make hazard-scan`: a hit fails the build
for p in [a for a in sys.argv[1:] if a != '--fail']:
    h = scan(p)
    tot += len(h)
    print(f'{p}: {len(h)} suspicious edges')
    for k, i, t, j, u, dist in h[:40]:
        print(f'  {k}\n    line {i}: {t}\n    line {j}: {u}   ({dist} issue slots apart)')
print('total', tot)
if fail and tot: sys.exit(1)
