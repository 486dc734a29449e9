"""Compare the per-kernel gfx950 ISA of two `hipcc -S --cuda-device-only` outputs.

Used to show that adding an experimental kernel / template parameter leaves the already validated
kernels byte-for-byte the same instructions:  python scripts/isa_diff.py before.s after.s
"""
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        body = re.sub(r';.*', '', m.group(2))
        body = re.sub(r'\.LBB\d+_\d+', '.LBB', body)
        out[m.group(1)] = '\n'.join(l.strip() for l in body.splitlines() if l.strip())
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    bad = 0
    for k in sorted(a):
        same = a[k] == b.get(k)
        bad += 0 if same else 1
        print(('SAME ' if same else 'DIFF ') + k)
    for k in sorted(b):
        if k not in a:
            print('NEW  ' + k)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
