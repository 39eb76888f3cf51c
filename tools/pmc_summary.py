"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter values per dispatch (millions)."""
import collections
import csv
import sys


def main(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:70]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    rows = []
    for k, v in tot.items():
        n = max(cnt[(k, c)] for c in v)
        rows.append((sum(v.get("GRBM_GUI_ACTIVE", 0.0) for _ in [0]), k, n, {c: x / cnt[(k, c)] / 1e6 for c, x in v.items()}))
    rows.sort(reverse=True)
    for total, k, n, m in rows[:30]:
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        line = f"{k:70s} n={n:4d} gui={gui:8.3f}"
        for c, x in sorted(m.items()):
            if c != "GRBM_GUI_ACTIVE":
                line += f" {c.replace('SQ_', '')}={x:.2f}"
        print(line)


if __name__ == "__main__":
    main(sys.argv[1])
