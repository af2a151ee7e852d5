set -e
for W in 8 4 2 1; do
  for S in 0 1; do
    echo "== world $W spread $S"; SRT_SPREAD=$S timeout -k 10 120 python tools/diag.py --spp 1024 --world $W --rank 0 | grep -E '"ms"|mray_s|trav_simd'
  done
done
