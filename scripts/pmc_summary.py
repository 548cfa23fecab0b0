"""Summarise rocprofv3 output (rocpd sqlite `*_results.db`, the ROCm 7.2 default) for the ndp:: kernels.

  python scripts/pmc_summary.py hbm   <FETCH_SIZE pass dir> <WRITE_SIZE pass dir>   > profiles/rNN_pmc_hbm.csv
  python scripts/pmc_summary.py stats <--kernel-trace --stats pass dir>              > profiles/rNN_kernel_stats.csv
  python scripts/pmc_summary.py sq    <--pmc pass dir> [<--pmc pass dir> ...]        > profiles/rNN_pmc_sq.csv
      (passes written with --output-format csv; one row per kernel and grid size: every counter's average per launch
       plus the ratios the design notes quote -- MFMA busy share, wait shares, LDS bank-conflict share, L2 hit rate)

gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE
counts half of the wide coalesced reads, so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

k_wgrad (k_wgrad_wide at large M) and k_reduce_adam are each launched twice per step (the D instance first, then the G
instance); they are split by dispatch order.
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
    if base in ("ndp::k_wgrad", "ndp::k_wgrad_wide", "ndp::k_reduce_adam"):
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


def _csv_rows(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as fh:
            yield from csv.DictReader(fh)


def sq(roots, match=("ndp::",)):
    """Per (kernel, grid): average counter values per launch over all launches of all passes, and derived ratios.
    Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
    SQ_BUSY_CYCLES is summed over the SQs (one per XCD shader engine); SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
    over SIMDs."""
    agg, launches = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    meta, dur_ns, dur_n = {}, defaultdict(float), defaultdict(int)
    for root in roots:
        seen = defaultdict(int)
        last = None
        for r in _csv_rows(root):
            name = r["Kernel_Name"]
            if not any(m in name for m in match):
                continue
            did = (root, r["Dispatch_Id"])
            if did != last:                       # rows of one dispatch are adjacent: one label per dispatch
                label = _label(name, seen)
                last = did
                k_ = (label, r["Grid_Size"])
                dur_ns[k_] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur_n[k_] += 1
            key = (label, r["Grid_Size"])
            meta[key] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[key][r["Counter_Name"]] += 1
    counters = sorted({c for v in agg.values() for c in v})
    derived = ["avg_us_under_pmc", "mfma_busy_of_peak_2p4GHz", "mfma_busy_of_elapsed_clock", "wait_inst_any_share", "wait_any_share", "active_inst_any_share", "vmem_share",
               "lds_share", "lds_bank_conflict_share", "l2_hit_rate"]
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid", "workgroup", "lds_bytes", "vgpr", "agpr", "launches"] + derived + counters)

    def ratio(c, num, den, scale=1.0):
        return "%.4f" % (scale * c[num] / c[den]) if c.get(den) and num in c else ""
    for key in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", agg[k].get("SQ_WAVE_CYCLES", 0.0))):
        c = {n: agg[key][n] / launches[key][n] for n in agg[key]}
        hits = c.get("TCC_HIT_sum")
        miss = c.get("TCC_MISS_sum")
        row = [key[0], key[1]] + list(meta[key]) + [max(launches[key].values())]
        us = dur_ns[key] / max(dur_n[key], 1) / 1e3
        row.append("%.3f" % us)
        # SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's matrix pipe was executing, summed over the 1,024 SIMDs (32 per
        # v_mfma_f32_16x16x4_f32).  / (1,024 x elapsed x 2.4 GHz) = the fraction of the 157.3 TFLOP/s fp32 peak the
        # launch achieved (the roofline's clock); / (1,024 x GRBM_GUI_ACTIVE / 8) = share of the cycles the chip
        # really clocked during the launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs).
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        row.append("%.4f" % (mf / (1024.0 * us * 1e-6 * 2.4e9)) if mf is not None and us > 0 else "")
        row.append("%.4f" % (mf / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)) if mf is not None and c.get("GRBM_GUI_ACTIVE") else "")
        row += [ratio(c, n, "SQ_WAVE_CYCLES") for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                                                         "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS")]
        row.append(ratio(c, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))
        row.append("%.4f" % (hits / (hits + miss)) if hits is not None and miss is not None and hits + miss > 0 else "")
        row += ["%.1f" % c[n] if n in c else "" for n in counters]
        out.writerow(row)


if __name__ == "__main__":
    if sys.argv[1] == "sq":
        sq(sys.argv[2:])
    elif sys.argv[1] == "hbm":
        hbm(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        sys.exit(__doc__)
