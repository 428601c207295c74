#!/bin/bash
# Dev tool: what the streaming-phase kernel (k_apply) spends its cycles on -- instruction counts, VALU/LDS busy, waits, LDS
# bank conflicts -- per dispatch, first 24 merges of the config-3 job.   tools/pmc_apply.sh OUT_DIR -> OUT_DIR/pmc_apply.txt
OUT=${1:-gpurun_out/pmc_apply}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "k_apply" --output-format csv -d "$ROOT/$OUT/raw$i" -- python3 "$ROOT/tools/quick_job.py" --merges 24 --runs 1 --sample 0 > "$ROOT/$OUT/run$i.log" 2>&1 || echo "group $i failed"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/pmc_apply.txt", "w") as g:
    for f in sorted(glob.glob(out + "/raw*/**/*counter_collection.csv", recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = [d for d in per.values() if "k_apply<" in d["name"] and d.get("SQ_WAVES", 1) > 0]
        names = sorted({k for d in ds for k in d if k != "name"})
        big = [d for d in ds if max(d.get(n, 0) for n in names) > 1e6]
        line = f"{len(big)} working dispatches: " + "  ".join(f"{n} {sum(d.get(n, 0) for d in big) / max(1, len(big)):.4g}" for n in names)
        print(line); g.write(line + "\n")
PY
rm -rf "$ROOT/$OUT"/raw*
