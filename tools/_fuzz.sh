#!/bin/bash
# round 5's fuzz campaign through every kernel variant (one gpurun call).  usage: tools/_fuzz.sh <outdir>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r05_fuzz}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 900 python tools/fuzz_campaign.py $FIRST $COUNT > $O/$name.txt 2>&1; echo "$name rc=$?"; tail -2 $O/$name.txt; }
FIRST=3000 COUNT=800; run default_variants FUZZ_X=0
FIRST=4000 COUNT=500; run wide_cap3 FUZZ_WIDE=1 FUZZ_CACHE_MAX=3
FIRST=5000 COUNT=300; run wide_cap0 FUZZ_WIDE=1 FUZZ_CACHE_MAX=0
FIRST=6000 COUNT=300; run narrow_cap3 FUZZ_CACHE_MAX=3
FIRST=7000 COUNT=300; run wide_all_cached FUZZ_WIDE=1
FIRST=8000 COUNT=300; run tuned_trees FUZZ_TUNE=1
