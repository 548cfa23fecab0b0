set -x
R=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $R
export TMPDIR=/tmp
cd /tmp
NDP_BENCH_EXTRAS=h2d_per_launch,config4,forward_model rocprofv3 --kernel-trace --output-format csv -d $R/slow -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/slow.json 2> $R/slow.err
NDP_BENCH_EXTRAS=forward_model rocprofv3 --kernel-trace --output-format csv -d $R/fast -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/fast.json 2> $R/fast.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for tag in ("slow", "fast"):
    f = glob.glob("gpurun_out/r3i/%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_fm_" in r["Kernel_Name"] or "k_adam" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print(tag, len(rows), "fm kernels; columns:", list(rows[0].keys())[:14])
    # one step in the middle of the batch-8 timing: find k_adam launches, take the window between the 20th and 21st
    adams = [i for i, r in enumerate(rows) if "k_adam(" in r["Kernel_Name"] or r["Kernel_Name"].endswith("k_adam")]
    adams = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"] and "advance" not in r["Kernel_Name"]]
    a, b = adams[20], adams[21]
    t0 = int(rows[a]["End_Timestamp"])
    qs = collections.Counter(r["Queue_Id"] for r in rows[a:b])
    print(" step span us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3, "queues", dict(qs))
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a + 1:b + 1]) / 1e3
    print(" sum of kernel durations us", busy)
    prev_end = t0
    gaps = []
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gaps.append(((s - prev_end) / 1e3, r["Kernel_Name"][:40], r["Queue_Id"], (e - s) / 1e3))
        prev_end = max(prev_end, e)
    gaps.sort(reverse=True)
    for g in gaps[:12]:
        print("   gap %.1f us before %s (queue %s, dur %.1f)" % g)
PY
