#!/bin/bash
# HBM traffic of k_scan from PMC counters, as MI355X_MICROARCH.md prescribes: separate passes for FETCH_SIZE and
# WRITE_SIZE (they do not fit one pass), --pmc never combined with trace options.  Run on the GPU box:
#   bash tools/collect_pmc.sh [extra bench.py args]   -> gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}/..., summary printed as JSON
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_scan" --output-format csv -d $OUT/$C -- \
    python3 bench.py --no-cpu-baseline --no-dedup-line --no-pretok-line --merges 2500 --roofline-merges 2500 "$@" > $OUT/bench_$C.json 2> $OUT/$C.err
done
python3 - <<'PY'
import csv, glob, json
def load(c):
    f = glob.glob(f"gpurun_out/pmc/{c}/*/*counter_collection.csv")[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("yb::k_scan(") and r["Counter_Name"] == c]
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
n_all = len(f)
f = [x for x in f if x > 64.0]   # launches after the stop flag do nothing (< 1 us, no traffic): not part of the average,
w = w[:len(f)] if len(w) >= len(f) else w   # as they are not part of bench.py's timed average either
rf = json.load(open("gpurun_out/pmc/bench_FETCH_SIZE.json"))["roofline"]  # k_scan of the auxiliary pass
mf, mw = sum(f) / len(f), sum(w) / len(w)
traffic = 2 * mf * 1024 + mw * 1024
out = {"kernel": "yb::k_scan", "dispatches": len(f), "dispatches_incl_noop": n_all, "mean_FETCH_SIZE_KiB": mf, "mean_WRITE_SIZE_KiB": mw,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> x2; unit KiB -> x1024",
       "traffic_bytes_per_launch": traffic, "algo_bytes_per_launch_same_run": rf["algo_bytes_per_launch"],
       "actual_stream_bytes_per_launch_same_run": rf["actual_stream_bytes_per_launch"],
       "traffic_over_actual": traffic / rf["actual_stream_bytes_per_launch"], "traffic_over_algorithmic": traffic / rf["algo_bytes_per_launch"]}
json.dump(out, open("gpurun_out/pmc/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -f gpurun_out/pmc/*/*/*counter_collection.csv   # tens of MB; the summary is what gets committed
