#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel (reads a `hipcc -S --cuda-device-only` file).
usage: tools/isa_segments.py <file.s> <mangled-name-substring> [min-instructions]"""
import collections
import re
import sys


def main(path, pat, min_tot=60):
    s = open(path).read()
    funcs = re.split(r'\n(_Z\w+):', s)
    body = None
    for i in range(1, len(funcs), 2):
        if pat in funcs[i]:
            body = funcs[i + 1].split('.Lfunc_end')[0]
            print('kernel', funcs[i][:100])
            break
    if body is None:
        raise SystemExit('kernel not found')
    lines = body.split('\n')
    seg, cur = [], [0, collections.Counter(), 'start']
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith('.LBB') or t.startswith('; %bb'):
            seg.append(cur)
            cur = [i + 1, collections.Counter(), t.split(':')[0][:20]]
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        cur[1][t.split()[0]] += 1
    seg.append(cur)
    for st, c, name in seg:
        tot = sum(c.values())
        if tot < min_tot:
            continue
        v = sum(n for o, n in c.items() if o.startswith('v_'))
        sc = sum(n for o, n in c.items() if o.startswith('scratch'))
        print(f"{st:6d} {name:14s} tot {tot:5d} valu {v:5d} scratch {sc:3d}", sorted(c.items(), key=lambda x: -x[1])[:9])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
