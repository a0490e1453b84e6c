#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing:  isa_count.py file.s <symbol-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = re.compile(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):.*?\n(.*?)s_endpgm', re.S | re.M)
m = pat.search(s)
print(m.group(1))
lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
c = Counter(l.split()[0] for l in lines if not l.endswith(':'))
print(len(lines), "instructions")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{v:6d} {k}")
