#!/bin/bash
# Dev tool: rocprofv3's derived metrics (VALUBusy, LdsUtil, latencies, stalls, occupancy ...) for the per-merge kernels, a few
# per pass.   tools/pmc_derived.sh OUT_DIR MERGES   -> OUT_DIR/pmc_derived.txt   (mean per kernel name over working dispatches)
# (every launch is collected: 20,000 merges take minutes per pass and the latency groups longer still -- keep MERGES small, and do
# not pipe the output into tail: a pass that prints nothing for 7 minutes is taken for hung on the gpurun boxes)
OUT=${1:-gpurun_out/pmc_derived}
MERGES=${2:-3000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "VALUBusy SALUBusy GRBM_GUI_ACTIVE" "LdsUtil LDSBankConflict LdsLatency" "VmemLatency SmemLatency InstrFetchLatency" "MemUnitStalled OccupancyPercent MeanOccupancyPerCU" "SIMD_UTILIZATION VALUUtilization SerializedAtomicRatio"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "k_apply|k_scan_skip" --output-format csv -d "$ROOT/$OUT/raw$i" -- python3 "$ROOT/tools/quick_job.py" --merges $MERGES --runs 1 --sample 0 > "$ROOT/$OUT/run$i.log" 2>&1 || echo "group $i ($G) failed"
  echo "pass $i ($G) done" | tee -a "$ROOT/$OUT/progress.txt"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/pmc_derived.txt", "w") as g:
    for f in sorted(glob.glob(out + "/raw*/**/*counter_collection.csv", recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0][-44:]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = list(per.values())
        n = len(ds)
        for lo, hi, tag in ((0, min(n, 70), "first dispatches"), (max(0, n - 200), n, "last 200 dispatches")):
            sel = ds[lo:hi]
            by = collections.OrderedDict()
            for d in sel: by.setdefault(d["name"], []).append(d)
            for name, rows in by.items():
                names = sorted({k for d in rows for k in d if k != "name"})
                line = f"{tag}: {name} x{len(rows)}: " + "  ".join(f"{k} {sum(d.get(k, 0) for d in rows) / len(rows):.4g}" for k in names)
                print(line); g.write(line + "\n")
PY
rm -rf "$ROOT/$OUT"/raw*
