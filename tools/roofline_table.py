"""Per-kernel roofline table of one training step from bench.py's event table (`bench.py --kernel-table`, stderr):
every timed launch shape with its time per step, achieved TFLOP/s and GB/s (algorithmic work: bench.kernel_work), the
roofline that binds it -- the larger of flops/peak and bytes/peak -- and the fraction of that roofline.

    python tools/roofline_table.py profiles/r02_bench_event_tables.txt [bf16|fp32]

Where bench_traffic.json (HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE counters, tools/pmc_traffic.sh) holds
the launch shape, two more columns give the counter bytes and their ratio to the algorithmic bytes the GB/s column uses.
"""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK = {"bf16": (2500.0, 8000.0), "fp32": (157.3, 8000.0)}      # TFLOP/s (dense MFMA), GB/s (HBM3E)


def main():
    path, which = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
    tf_peak, bw_peak = PEAK[which]
    rows, on = [], False
    for line in open(path):
        if line.startswith(("bf16:", "fp32:")):
            on = line.startswith(which + ":")
            continue
        m = re.match(r"\s+([\d.]+) ms x(\d+)\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s\s+(\(.*\))\s*$", line)
        if on and m:
            ms, n, tf, gb, key = float(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), ast.literal_eval(m.group(5))
            rows.append((ms * n, ms, n, tf, gb, key))
    total = sum(r[0] for r in rows)
    try:
        counters = json.load(open(os.path.join(ROOT, "bench_traffic.json")))
    except (OSError, ValueError):
        counters = {}
    print("| kernel (entry point, shape) | launches x ms | % of step | TFLOP/s | GB/s | bound | fraction | counter MB | counter / algorithmic |")
    print("|---|---|---|---|---|---|---|---|---|")
    for tot, ms, n, tf, gb, key in sorted(rows, key=lambda r: -r[0]):
        f_m, f_h = tf / tf_peak, gb / bw_peak
        bound, frac = ("mfma", f_m) if f_m >= f_h else ("hbm", f_h)
        shape = " ".join(str(k) for k in key[2:])
        ent = counters.get("|".join(map(str, key)))
        cb = "%.0f" % (ent["traffic_bytes"] / 1e6) if ent else ""
        alg = gb * ms * 1e6                                  # GB/s x ms = algorithmic bytes per launch
        cr = "%.2f" % (ent["traffic_bytes"] / alg) if ent and alg > 0 else ""
        print("| %s %s | %d x %.3f | %.1f | %.0f | %.0f | %s | %.2f | %s | %s |" % (key[0].replace("stgcn_", ""), shape, n, ms, 100 * tot / total, tf, gb, bound, frac, cb, cr))
    print("\ntotal of the timed launches %.2f ms" % total)
    # families: time per step, share, time-weighted fraction of the binding roofline and its range over launch shapes
    fam_of = lambda name: ("temporal conv forward / data gradient" if name == "stgcn_conv_rows" else
                           "temporal conv weight gradient" if name == "stgcn_conv_wgrad" else
                           "graph conv forward, dX, dW, dA" if name.startswith("stgcn_gcn") else "elementwise")
    fams = {}
    for tot, ms, n, tf, gb, key in rows:
        fams.setdefault(fam_of(key[0]), []).append((tot, max(tf / tf_peak, gb / bw_peak)))
    print("\n| family | ms per step | % of step | time-weighted fraction (range) |\n|---|---|---|---|")
    for name, v in sorted(fams.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
        t = sum(x for x, _ in v)
        print("| %s | %.2f | %.0f | %.2f (%.2f-%.2f) |" % (name, t, 100 * t / total, sum(x * f for x, f in v) / t, min(f for _, f in v), max(f for _, f in v)))
    print("| all timed launches | %.2f | 100 | %.2f |" % (total, sum(x * f for v in fams.values() for x, f in v) / total))


if __name__ == "__main__":
    main()
