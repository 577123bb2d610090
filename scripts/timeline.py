"""Print the tail (or a window) of the kernel timeline of the last engine call in a rocprofv3 --kernel-trace
sqlite file: start, duration, queue, kernel, grid size -- to see what the queues wait for.
  python scripts/timeline.py kt_results.db [n_last_kernels] [skip_from_end]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wg = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end, queue_id" + (", %s" % grid if grid else ", 0") + (", %s" % wg if wg else ", 1")
rows = [r for r in db.execute("select %s from kernels order by start" % sel)
        if re.sub(r"^void ", "", r[0]).startswith("k_")]
# engine calls are separated by milliseconds of host work
calls = [[rows[0]]]
for r in rows[1:]:
    if r[1] - max(x[2] for x in calls[-1][-50:]) > 2e6:
        calls.append([])
    calls[-1].append(r)
big = [c for c in calls if len(c) >= 0.9 * max(len(x) for x in calls)]
call = big[-2] if len(big) > 1 else big[-1]
t0 = call[0][1]
print("call: %d kernels, span %.2f ms" % (len(call), (max(x[2] for x in call) - t0) / 1e6))
part = call[len(call) - n - skip: len(call) - skip]
for name, s, e, q, g, w in part:
    nm = re.sub(r"^void ", "", name).split("(")[0]
    print("%10.1f us  %7.1f us  q%-2d %-22s wgs %d" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm, (g // max(w, 1)) if g else 0))
