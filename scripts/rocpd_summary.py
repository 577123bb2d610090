"""Summaries of rocprofv3's sqlite output (rocpd): per-kernel statistics of a --kernel-trace run as CSV,
and per-kernel HBM traffic from two --pmc passes (FETCH_SIZE, WRITE_SIZE) as JSON + markdown table.

  python scripts/rocpd_summary.py stats  kt_results.db  out.csv
  python scripts/rocpd_summary.py pmc    f_results.db w_results.db  out.json out.md
  python scripts/rocpd_summary.py gaps   kt_results.db
  python scripts/rocpd_summary.py counters  pmc_results.db  out.md      (every counter of the pass, per kernel)
"""
import csv
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([short(name)[:120], n, int(tot), "%.1f" % avg, "%.3f" % (100.0 * tot / total), mn, mx])


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, n, tot in db.execute("select name, count(*), sum(counter_value) from pmc_events "
                                   "where counter_name = ? group by name", (counter,)):
        k = short(name)
        if k.startswith("k_"):
            out[k] = (n, tot)
    return out


def counters(db_path, out_md):
    """Sum of every counter of a --pmc pass per kernel, and the ratios that say how busy the matrix pipe was."""
    db = sqlite3.connect(db_path)
    agg, cnt = {}, {}
    for name, counter, n, tot in db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                                            "group by name, counter_name"):
        k = short(name)
        if not k.startswith("k_"):
            continue
        agg.setdefault(k, {})[counter] = tot
        cnt[k] = n
    names = sorted({c for v in agg.values() for c in v})
    # The SQ counters come as one value per shader engine (8 XCDs x 4 = 32 per launch on an MI355X).
    # SQ_BUSY_CYCLES / 32 is then the launch's duration in shader cycles, and the matrix pipes of the
    # chip's 1024 SIMDs could have been busy for 1024 times that many cycles.  One MOPS_F64 unit is
    # 512 flops (a v_mfma_f64_16x16x4_f64 = 2048 flops = 4 units = 64 busy cycles).
    inst, simds = 32, 1024
    with open(out_md, "w") as f:
        f.write("| kernel | counter rows | " + " | ".join(names) + " | MFMA busy share of SIMD cycles | issued F64 MFMA TFLOP per launch-second |\n")
        f.write("|---|---|" + "---|" * (len(names) + 2) + "\n")
        for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0)):
            v = agg[k]
            ratio = rate = ""
            if v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                ratio = "%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["SQ_BUSY_CYCLES"] / inst * simds))
            if v.get("SQ_BUSY_CYCLES") and "SQ_INSTS_VALU_MFMA_MOPS_F64" in v:
                # flops per shader cycle of launch time; times the clock gives the rate
                rate = "%.1f flop/cycle" % (v["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / (v["SQ_BUSY_CYCLES"] / inst))
            f.write("| %s | %d | " % (k, cnt[k]) + " | ".join("%.4g" % v.get(c, 0) for c in names) + " | %s | %s |\n" % (ratio, rate))
        f.write("\nCounter rows = launches x %d shader engines.  MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / %d x %d SIMDs); "
                "the last column is SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops per shader cycle of launch time (chip peak: 32768 flop/cycle).\n" % (inst, inst, simds))


def pmc(fetch_db, write_db, out_json, out_md, tag="dense", command=None):
    fe, wr = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        if k not in wr or wr[k][0] != fe[k][0]:
            continue
        n = fe[k][0]
        fetch = 2.0 * 1024.0 * fe[k][1] / n         # KB -> bytes, doubled on gfx950
        write = 1024.0 * wr[k][1] / n
        kernels[k] = {"dispatches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "traffic_bytes_per_launch": fetch + write}
    json.dump({"workload_tag": tag,
               "command": command or "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py "
                          "--no-cpu-baseline --steps 1 --warmup 0",
               "corrections": "counter unit KB; FETCH_SIZE doubled on gfx950 (128-B requests counted as 64 B); "
                              "WRITE_SIZE as is",
               "kernels": kernels}, open(out_json, "w"), indent=1)
    with open(out_md, "w") as f:
        f.write("| kernel | launches | FETCH_SIZE per launch | x2 | WRITE_SIZE per launch | traffic per launch |\n")
        f.write("|---|---|---|---|---|---|\n")
        for k, v in kernels.items():
            f.write("| %s | %d | %.1f MB | %.1f MB | %.1f MB | **%.0f MB** |\n" % (
                k, v["dispatches"], v["fetch_bytes_per_launch"] / 2e6, v["fetch_bytes_per_launch"] / 1e6,
                v["write_bytes_per_launch"] / 1e6, v["traffic_bytes_per_launch"] / 1e6))


def gaps(db_path):
    """Per hardware queue, for every stage of ONE factorization in a --kernel-trace run of bench.py (by
    default the timed step) and for the whole factorization: kernels, busy time, idle time between kernels, and how long after the
    latest kernel end anywhere on the device each kernel started (the start latency of a launch that
    waited for another queue's kernel)."""
    import bisect
    import collections
    import statistics
    db = sqlite3.connect(db_path)
    rows = [(short(n), s, e, q) for n, s, e, q in
            db.execute("select name, start, end, queue_id from kernels order by start") if short(n).startswith("k_")]
    # bench.py under the profiler runs NFACT factorizations with the same launches each (the first call, the
    # timed steps, the profiled step): the last of them is the last 1/NFACT of the kernels; inside it every
    # stage (group of fronts) starts with the launch that builds its S entries
    nfact = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    which = int(sys.argv[4]) if len(sys.argv) > 4 else min(2, nfact)       # (bench.py: 1 the first call, 2 the timed step, 3 the profiled one)
    per = len(rows) // nfact
    lastf = rows[(which - 1) * per:which * per]
    last = []
    for r in lastf:
        if r[0] == "k_build_s" or not last:
            if not (last and all(x[0] == "k_build_s" for x in last[-1])):
                last.append([])
        last[-1].append(r)
    print("window: factorization %d of the %d in the trace (%d of %d kernels; bench.py's timed step is the second, its "
          "first call the first, its step with events around every launch the third), stage by stage (a stage starts "
          "with the launch that builds its S entries)" % (which, nfact, len(lastf), len(rows)))

    def report(title, ks):
        print("%s: kernels %d, span %.1f ms" % (title, len(ks), (max(x[2] for x in ks) - ks[0][1]) / 1e6))
        ends = sorted(x[2] for x in ks)
        byq = collections.defaultdict(list)
        for x in ks:
            byq[x[3]].append(x)
        for q, l in sorted(byq.items()):
            l.sort(key=lambda x: x[1])
            busy = sum(x[2] - x[1] for x in l)
            idle = sum(max(0, l[i + 1][1] - l[i][2]) for i in range(len(l) - 1))
            lat = []
            for x in l:
                i = bisect.bisect_right(ends, x[1]) - 1
                if i >= 0:
                    lat.append(x[1] - ends[i])
            names = collections.Counter(x[0] for x in l).most_common(2)
            print("  queue %d: %5d kernels, busy %.1f ms, idle between kernels %.1f ms, start latency median %.1f us, "
                  "sum %.1f ms  %s" % (q, len(l), busy / 1e6, idle / 1e6, statistics.median(lat) / 1e3 if lat else 0.0,
                                       sum(lat) / 1e6, names))

    for k, c in enumerate(last):
        report("stage %d" % (k + 1), c)
    if len(last) > 1:
        report("the whole factorization (idle includes the host work between the stages)", [x for c in last for x in c])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:])
