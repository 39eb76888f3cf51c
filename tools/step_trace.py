"""Per-step launch statistics from a rocprofv3 --kernel-trace CSV of bench.py: launches, busy time, gaps, and the
kernels shorter than 20 us.  usage: step_trace.py kernel_trace.csv [skip [sequence]]   (sequence: also every launch in order)

A step ends with its sgd_kernel launch.  bench.py runs, after its timed steps, min(steps, 5) launch-by-launch steps
that price the dominant kernel; `skip` (default 5) steps are dropped from the end of the trace so that the step
analysed is one of the timed ones (a HIP-graph replay unless --no-graph); skip=0 analyses the last step of the trace."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
step = rows[idx[-2 - skip] + 1:idx[-1 - skip] + 1]
print("step %d from the end of the trace (%s)" % (skip + 1, "a timed step" if skip else "the last step"))
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
print("step span %.3f ms, busy %.3f ms, launches %d" % ((t1 - t0) / 1e6, sum(map(dur, step)) / 1e6, len(step)))
gaps = [int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]) for i in range(len(step) - 1)]
print("gaps: %.3f ms total, max %.1f us" % (sum(g for g in gaps if g > 0) / 1e6, max(gaps) / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    agg[r["Kernel_Name"][:80]][0] += 1
    agg[r["Kernel_Name"][:80]][1] += dur(r)
small = [(k, v) for k, v in agg.items() if v[1] / v[0] < 20000]
print("kernels under 20 us: %d launches, %.3f ms" % (sum(v[0] for _, v in small), sum(v[1] for _, v in small) / 1e6))
for k, v in sorted(small, key=lambda kv: -kv[1][1])[:16]:
    print("  %3d x %6.1f us  %s" % (v[0], v[1] / v[0] / 1e3, k))
print("largest gaps:")
for g, i in sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:10]:
    print("  %6.1f us  after %-40s before %s" % (g / 1e3, step[i]["Kernel_Name"][:40], step[i + 1]["Kernel_Name"][:40]))
if len(sys.argv) > 3 and sys.argv[3] == "sequence":
    print("launch sequence of the step (us, kernel):")
    for r in step:
        print("  %7.1f  %s" % (dur(r) / 1e3, r["Kernel_Name"][:110]))
hist = collections.Counter(min(int(g / 1000), 10) for g in gaps if g > 0)
print("gap histogram (us: count):", sorted(hist.items()))
