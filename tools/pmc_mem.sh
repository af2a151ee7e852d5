#!/bin/bash
# Memory-pipeline PMC passes (TA / TCP) for the render kernel.  Usage: tools/pmc_mem.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc_mem}; shift || true
ARGS=${@:---spp 64 --steps 1 --warmup 0 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for SET in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TD_TC_STALL_sum" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -- python bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel<0>" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = "%-40s per-dispatch %.6g (dispatches %d)" % (k, tot[k] / max(n[k], 1), n[k])
        print(line); fh.write(line + "\n")
PY
