#!/usr/bin/env python3
"""Instruction histogram of the kernels in a hipcc -S listing whose mangled name contains a pattern.

usage: isa_hist.py listing.s pattern [top]
"""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    text = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pat not in name:
            continue
        ops = collections.Counter()
        for line in body.split('\n'):
            line = line.strip()
            if not line or line[0] in '.;/' or line.endswith(':'):
                continue
            ops[line.split()[0]] += 1
        print(name, sum(ops.values()))
        print('   ', ' '.join(f'{k}:{v}' for k, v in ops.most_common(top)))


if __name__ == '__main__':
    main()
