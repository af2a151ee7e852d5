#!/bin/bash
# Memory-path counters (TA / TCP / TCC) of the render kernel, one --pmc set per run (kernel-trace only).  For scenes whose BVH is
# served by L1 / L2 (cfg 5's 100k-triangle mesh).  Usage (through gpurun): tools/pmc_mem.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc_mem}; shift || true
ARGS=${@:---scene 101 --width 3840 --height 2160 --spp 32 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
# (at most 4 counters of one TCP / TCC instance and 2 of a TA per pass: more fails with "exceeds the capabilities of the hardware")
for SET in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum" \
           "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUFFER_TOTAL_CYCLES_sum TA_TA_BUSY_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 ${PASS_TIMEOUT:-100} rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -- python bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel<0" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = "%-40s per-dispatch %.6g (dispatches %d)" % (k, tot[k] / max(n[k], 1), n[k])
        print(line); fh.write(line + "\n")
PY
rm -rf "$OUT"/pass*/
