#!/bin/bash
# Dev tool: per-dispatch instruction counters of the per-merge kernels (first 6000 merges of the config-3 job).
#   tools/pmc_scan.sh OUT_DIR   -> OUT_DIR/pmc_scan_summary.txt
set -e
OUT=${1:-gpurun_out/pmc_scan}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/raw" -- python3 "$ROOT/tools/quick_job.py" --merges 6000 --runs 1 --sample 0 > "$ROOT/$OUT/run.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
per = collections.OrderedDict()
for r in rows:
    d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ds = [d for d in per.values() if "k_scan_skip" in d["name"] or "k_apply<" in d["name"]]
print("dispatches", len(ds))
with open(out + "/pmc_scan_summary.txt", "w") as g:
    names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]
    for lo, hi in [(0, 50), (50, 150), (150, 300), (300, 1000), (1000, 3000), (3000, 6000)]:
        sel = ds[lo:hi]
        if not sel: continue
        line = f"dispatches {lo}-{hi} ({sel[0]['name'][:30]}): " + "  ".join(f"{n[3:]} {sum(d.get(n, 0) for d in sel) / len(sel):.0f}" for n in names)
        print(line); g.write(line + "\n")
PY
rm -rf "$ROOT/$OUT/raw"
