#!/bin/bash
# PMC passes for the render kernel (run on the GPU box through gpurun).  KERNEL_FILTER (default "render_kernel<0") selects the
# dispatches that are summed, e.g. KERNEL_FILTER=init_rng_kernel.  Each --pmc set is its own run, with
# --kernel-trace only (no other trace domains), as MI355X_MICROARCH.md prescribes.  Usage: tools/pmc_passes.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0 --no-builders-tree}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
# PMC_ONLY=traffic: just the two fabric-traffic passes (the second spp point of tools/pmc_to_lane_ops.py --traffic)
if [ "${PMC_ONLY:-}" = "traffic" ]; then
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -- python bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
else
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pass$i" -- python bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
fi
python - "$OUT" "${KERNEL_FILTER:-render_kernel<0}" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; flt = sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if flt in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = "%-28s per-dispatch %.6g (dispatches %d)" % (k, tot[k] / max(n[k], 1), n[k])
        print(line); fh.write(line + "\n")
PY
