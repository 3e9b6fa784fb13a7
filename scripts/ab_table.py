#!/usr/bin/env python3
"""Tabulate a scripts/ab_gemm.sh log: per case, the medians each arm printed (one per repetition) and the sum of minima."""
import collections
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
arms, arm = [], None
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+) ", line)
    if m:
        arm = m.group(1)
        if arm not in arms:
            arms.append(arm)
        continue
    m = re.match(r"(SPLIT .*?)\s+([\d.]+) ms", line)
    if m and arm:
        d[m.group(1).strip()][arm].append(float(m.group(2)))
print("%-36s" % "case" + "".join("%16s" % a for a in arms))
tot = collections.defaultdict(float)
for c, v in d.items():
    print("%-36s" % c + "".join("%16s" % ("/".join("%.3f" % x for x in v[a])) for a in arms))
    for a in arms:
        tot[a] += min(v[a]) if v[a] else 0
print("%-36s" % "sum(min)" + "".join("%16.3f" % tot[a] for a in arms))
