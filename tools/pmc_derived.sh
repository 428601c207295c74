#!/bin/bash
# Dev tool: rocprofv3's derived metrics for the per-batch kernels, ONE or TWO counters per pass (a group that cannot be
# scheduled together makes rocprofv3 abort at start: "Could not create PMC packets", which is what happened to the latency /
# occupancy groups of round 2).   tools/pmc_derived.sh OUT_DIR MERGES   -> OUT_DIR/pmc_derived.txt
# Every launch is collected; a pass over 21,000 merges (about 5,000 launches since launches apply batches) takes a minute or two.
# The program after "--" is python3 itself (no env / bash -c hop: the profiler has initialised the GPU by then).
OUT=${1:-gpurun_out/pmc_derived}
MERGES=${2:-21000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "VALUBusy SALUBusy" "LdsUtil LDSBankConflict" "VmemLatency" "MemUnitStalled" "OccupancyPercent" "SerializedAtomicRatio" "MeanOccupancyPerCU"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "k_apply|k_scan_skip" --output-format csv -d "$ROOT/$OUT/raw$i" -- python3 "$ROOT/tools/quick_job.py" --merges $MERGES --runs 1 --sample 0 > "$ROOT/$OUT/run$i.log" 2>&1 || echo "pass $i ($G) failed: $(tail -2 "$ROOT/$OUT/run$i.log" | tr '\n' ' ')" | tee -a "$ROOT/$OUT/progress.txt"
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
            d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0][-40:]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = list(per.values())
        n = len(ds)
        # windows by dispatch order: the streaming launches, the dense start of the sparse phase, its middle, its end
        for lo, hi, tag in ((0, min(n, 70), "dispatches 0-70 (streaming)"), (150, min(n, 400), "dispatches 150-400 (about merges 300-1,000)"),
                            (n // 3, min(n, n // 3 + 300), "middle 300 dispatches"), (max(0, n - 300), n, "last 300 dispatches (about merges 20,000+)")):
            by = collections.OrderedDict()
            for d in ds[lo:hi]: by.setdefault(d["name"], []).append(d)
            for name, rows in by.items():
                names = sorted({k for d in rows for k in d if k != "name"})
                line = f"{tag}: {name} x{len(rows)}: " + "  ".join(f"{k} {sum(d.get(k, 0) for d in rows) / len(rows):.4g}" for k in names)
                print(line); g.write(line + "\n")
PY
rm -rf "$ROOT/$OUT"/raw*
