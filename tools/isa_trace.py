#!/usr/bin/env python3
"""Compressed trace of one kernel's ISA: VALU counts between memory ops / waits (reads a hipcc -S file)."""
import re
import sys


def main(path, pat, limit=300):
    s = open(path).read()
    funcs = re.split(r'\n(_Z\w+):', s)
    body = None
    for i in range(1, len(funcs), 2):
        if re.search(pat, funcs[i]):
            body = funcs[i + 1].split('.Lfunc_end')[0].split('\n')
            print('kernel', funcs[i][:80])
            break
    if body is None:
        raise SystemExit('kernel not found')
    out, valu = [], 0
    for l in body:
        t = l.strip()
        if not t or t.startswith(';'):
            continue
        if t.startswith('.LBB'):
            out.append((valu, t)); valu = 0
            continue
        if t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('v_'):
            valu += 1
            continue
        if op.startswith(('global_', 'ds_', 's_waitcnt', 's_barrier', 's_cbranch', 's_branch', 's_load', 'buffer_', 'scratch_')):
            out.append((valu, t[:70])); valu = 0
    res = []
    for v, t in out:
        key = t if t.startswith(('s_waitcnt', '.LBB', 's_cbranch', 's_branch')) else t.split()[0]
        if res and res[-1][1] == key and v < 3:
            res[-1][2] += 1; res[-1][0] += v
        else:
            res.append([v, key, 1])
    for v, k, c in res[:limit]:
        print(f"{v:5d} valu | {k} x{c}")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 300)
