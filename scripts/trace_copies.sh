#!/bin/bash
# Kernel + memory-copy trace of the 2D 1024^2 workload: what the queues do around the R hand-off copies.
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/trace_copies
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tcp
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tcp -- python3 $GRAFT_REPO_ROOT/bench.py --workload sparse2d:1024 --steps 2 --warmup 1 --extras none --no-cpu-baseline > "$OUT/line.json" 2> "$OUT/err.txt"
DB=$(find /tmp/tcp -name '*_results.db' | head -1)
python3 - "$DB" > "$OUT/merged_timeline.txt" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("tables:", [t for t in tabs if 'copy' in t.lower() or t in ('kernels',)])
ev = []
for name, s, e, q in db.execute("select name, start, end, queue_id from kernels"):
    nm = re.sub(r"^void ", "", name).split("(")[0]
    ev.append((s, e, "q%d" % q, nm, 0))
ct = [t for t in tabs if t.lower() in ("memory_copies", "memory_copy")]
for t in ct:
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    print(t, cols)
    sz = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
    nm = "name" if "name" in cols else "''"
    for name, s, e, b in db.execute("select %s, start, end, %s from %s" % (nm, sz, t)):
        ev.append((s, e, "copy", str(name), b))
ev.sort()
# last engine call: events after the last gap of more than 2 ms without kernels
ks = [x for x in ev if x[2] != "copy"]
cut = ks[0][0]
last_end = ks[0][1]
starts = [ks[0][0]]
for x in ks[1:]:
    if x[0] - last_end > 2e6: starts.append(x[0])
    last_end = max(last_end, x[1])
t0 = starts[-1]
for s, e, q, nm, b in ev:
    if s < t0 - 5e6: continue
    print("%10.1f us %9.1f us %-5s %-24s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm[:24], ("%.1f MB" % (b / 1e6)) if b else ""))
PY
grep -c copy "$OUT/merged_timeline.txt" || true
