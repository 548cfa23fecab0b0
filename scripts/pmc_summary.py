"""Summarise rocprofv3 output (rocpd sqlite `*_results.db`, the ROCm 7.2 default) for the ndp:: kernels.

  python scripts/pmc_summary.py hbm   <FETCH_SIZE pass dir> <WRITE_SIZE pass dir>   > profiles/rNN_pmc_hbm.csv
  python scripts/pmc_summary.py stats <--kernel-trace --stats pass dir>              > profiles/rNN_kernel_stats.csv

gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE
counts half of the wide coalesced reads, so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

k_wgrad and k_reduce_adam are each launched twice per step (the D instance first, then the G instance); they are
split by dispatch order.
"""
import csv
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


def _rows(root, sql):
    for path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(path)
        try:
            yield from con.execute(sql)
        finally:
            con.close()


def _label(name, seen):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip()
    base = re.sub(r"<.*>$", "", name)
    if base in ("ndp::k_wgrad", "ndp::k_reduce_adam"):
        name = base
        seen[name] += 1
        name += "[D]" if seen[name] % 2 == 1 else "[G]"
    return name


def collect_counter(root, counter):
    tot, cnt, seen = defaultdict(float), defaultdict(int), defaultdict(int)
    sql = ("select kernel_name, sum(value) from counters_collection where counter_name = '%s' "
           "and kernel_name like '%%ndp::%%' group by dispatch_id order by dispatch_id" % counter)
    for name, value in _rows(root, sql):
        name = _label(name, seen)
        tot[name] += float(value)
        cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def hbm(fetch_dir, write_dir):
    fetch = collect_counter(fetch_dir, "FETCH_SIZE")
    write = collect_counter(write_dir, "WRITE_SIZE")
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg",
                  "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024 [gfx950: FETCH_SIZE counts half of wide coalesced reads]"])
    for name in sorted(fetch):
        f, n = fetch[name]
        w = write.get(name, (0.0, 0))[0]
        out.writerow([name, n, "%.1f" % f, "%.1f" % w, int((2 * f + w) * 1024)])


def stats(root):
    dur, seen = defaultdict(list), defaultdict(int)
    sql = "select name, duration from kernels where name like '%ndp::%' order by dispatch_id"
    for name, d in _rows(root, sql):
        dur[_label(name, seen)].append(float(d))
    total = sum(sum(v) for v in dur.values())
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_ndp_kernel_time"])
    for name in sorted(dur, key=lambda k: -sum(dur[k])):
        v = dur[name]
        out.writerow([name, len(v), "%.1f" % (sum(v) / 1e3), "%.3f" % (sum(v) / len(v) / 1e3), "%.3f" % (min(v) / 1e3),
                      "%.3f" % (max(v) / 1e3), "%.2f" % (100.0 * sum(v) / total)])


if __name__ == "__main__":
    if sys.argv[1] == "hbm":
        hbm(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        sys.exit(__doc__)
