#!/usr/bin/env python3
"""Aggregate the TM lines an AMVS_TIMERS build prints (per-wave phase cycle totals)."""
import collections
import re
import sys
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0] * 6)
    n = collections.Counter()
    for l in open(path, errors="replace"):
        m = re.match(r"TM mode (\d+) draw (\d+) rows (\d+) ph (.*)", l)
        if not m:
            continue
        mode, rows = int(m.group(1)), int(m.group(3))
        for i, x in enumerate(m.group(4).split()[:6]):
            acc[mode][i] += int(x) / rows
        n[mode] += 1
    for mode in sorted(acc):
        a = [x / n[mode] for x in acc[mode]]
        tot = sum(a)
        print(path, "mode", mode, "waves", n[mode], "ticks/row", [round(x) for x in a], "total", round(tot),
              "frac", [round(x / tot, 3) for x in a])
