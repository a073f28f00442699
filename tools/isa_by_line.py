#!/usr/bin/env python3
"""Static instruction mix of one kernel by SOURCE LINE, from an assembly listing built with -gline-tables-only.

usage: isa_by_line.py listing.s <substring of the kernel's symbol> [top N]

For every (file, line) the number of vector and scalar instructions attributed to it and their issue cost in
cycles (classes and costs of tools/isa_mix.py).  Inlined code is charged to the line the instruction came from,
so a helper called from many places shows once, with all its copies added up.
"""
import re, sys
from collections import Counter, defaultdict
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_mix import classify, COST

def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    files = {}
    lines = open(path).read().split("\n")
    for l in lines:
        m = re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]*)"', l)
        if m: files[int(m.group(1))] = m.group(2)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    cur = ("?", 0)
    agg = defaultdict(Counter)
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"): break
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)(;.*)?$", l)
        if not m or l.lstrip().startswith((";", ".")): continue
        agg[cur][classify(m.group(1), m.group(2))] += 1
    rows = []
    for k, c in agg.items():
        valu = sum(v for kk, v in c.items() if kk in COST)
        cyc = sum(COST[kk] * v for kk, v in c.items() if kk in COST)
        scal = c["salu"] + c["sctl"]
        rows.append((cyc + 2.2 * scal, k, valu, cyc, scal, c))
    rows.sort(reverse=True)
    print("%-28s %6s %8s %6s  classes" % ("file:line", "VALU", "cycles", "SALU"))
    for tot, k, valu, cyc, scal, c in rows[:top]:
        print("%-28s %6d %8.0f %6d  %s" % ("%s:%d" % k, valu, cyc, scal, " ".join("%s=%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))
    tv = sum(r[2] for r in rows); tc = sum(r[3] for r in rows); ts = sum(r[4] for r in rows)
    print("TOTAL VALU %d cost %.0f SALU %d" % (tv, tc, ts))

if __name__ == "__main__":
    main()
