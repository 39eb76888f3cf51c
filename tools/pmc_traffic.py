"""Summarise the runs of tools/pmc_traffic.sh: per kernel case, the mean FETCH_SIZE and WRITE_SIZE per launch of the
entry point (summed over the kernels the entry point launches, e.g. a weight gradient and its slab reduction), and
traffic_bytes = 2 * FETCH_SIZE + WRITE_SIZE in bytes (both counters are in KB; on gfx950 FETCH_SIZE counts a 128-byte
request of a 16-byte-per-lane streaming read as 64 bytes), keyed like bench.py's timing keys.  Prints JSON.

    python tools/pmc_traffic.py <directory of the runs>            # on the GPU box, by tools/pmc_traffic.sh
    python tools/pmc_traffic.py --merge gpurun_out/prof/traffic.json   # afterwards: updates bench_traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_launch(csv_path):
    """-> (sum over the entry point's kernels of the mean counter value per dispatch, kernel names, dispatches)."""
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(csv_path)):
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    most = max(cnt.values())
    names = [k for k, n in cnt.items() if n >= 0.9 * most]     # the timed entry point's kernels; set-up kernels run once
    return sum(tot[k] / cnt[k] for k in names), names, most


def main(d):
    import ast
    from bench import kernel_work
    out = {}
    for log in sorted(glob.glob(os.path.join(d, "c*_FETCH_SIZE.log"))):
        tag = os.path.basename(log).split("_")[0]
        keys = [l.split(" ", 2) for l in open(log) if l.startswith("KEY ")]
        if not keys:
            continue
        case, key = keys[0][1], keys[0][2].strip()
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(d, "%s_%s" % (tag, ctr), "*", "*counter_collection.csv"))
            vals[ctr] = per_launch(f[0])
        parts = key.split("|")
        kk = tuple(ast.literal_eval(p) if p in ("True", "False") or p.lstrip("-").replace(".", "", 1).isdigit() else p for p in parts)
        fl, by, _ = kernel_work(kk)
        traffic = (2 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024
        out[key] = {"traffic_bytes": int(traffic), "fetch_size_kb": int(vals["FETCH_SIZE"][0]), "write_size_kb": int(vals["WRITE_SIZE"][0]),
                    "algorithmic_bytes": int(by), "counter_over_algorithmic": round(traffic / by, 3) if by else None,
                    "kernels": [n[:80] for n in vals["FETCH_SIZE"][1]], "dispatches": vals["FETCH_SIZE"][2],
                    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, over tools/kbench.py %s "
                           "(tools/pmc_traffic.sh); traffic = 2*FETCH_SIZE + WRITE_SIZE" % case}
    print(json.dumps(out, indent=1))


def merge(path):
    dst = os.path.join(ROOT, "bench_traffic.json")
    try:
        rec = json.load(open(dst))
    except (OSError, ValueError):
        rec = {}
    new = json.load(open(path))
    rec.update(new)
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print("bench_traffic.json: %d entries (%d updated)" % (len(rec), len(new)))


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2])
    else:
        main(sys.argv[1])
