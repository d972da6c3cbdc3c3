"""Tuning aid (CPU): per-shape comparison of two tests/conv_table.py outputs (same box, two library builds)."""
import sys


def rd(f):
    d = {}
    for l in open(f):
        if l.startswith("(") and l[1].isdigit():
            k = l[:l.index(")") + 1]
            r = l[l.index(")") + 1:].split()
            d[k] = (int(r[0]), float(r[1]))
    return d


b, n = rd(sys.argv[1]), rd(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
for k, (c, ms) in sorted(b.items(), key=lambda kv: -kv[1][1])[:top]:
    if k in n:
        print("%-42s %2d %7.3f -> %7.3f  %+5.1f%%" % (k, c, ms, n[k][1], (ms / n[k][1] - 1) * 100))
print("total %.2f -> %.2f ms" % (sum(v[1] for v in b.values()), sum(v[1] for v in n.values())))
