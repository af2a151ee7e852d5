#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats of the bench command's headline part (--no-cpu-baseline
# --no-other-configs --cfg5-spp 0: the parity leg and the other configurations launch the same kernel on other workloads and would
# enter the average), then the two HBM
# traffic counters in their own --pmc passes (kernel-trace only, no other trace domains).  Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-vX}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
[ "${2:-}" = "pmc-only" ] || timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --no-cpu-baseline --no-other-configs --cfg5-spp 0 --no-builders-tree > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "trace run failed"; tail -5 "$OUT/bench.err"; exit 1; }
[ "${2:-}" = "pmc-only" ] || cat "$OUT/bench.json"
[ "${2:-}" = "pmc-only" ] || find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
[ "${2:-}" = "pmc-only" ] || cat "$OUT/kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-calibration --no-other-configs --cfg5-spp 0 --no-builders-tree > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || { echo "pmc $C failed"; exit 1; }
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel<0" in row["Kernel_Name"]:
            res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
json.dump({k: v for k, v in res.items()}, open(out + "/pmc_render_kernel0.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf "$OUT/trace" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
