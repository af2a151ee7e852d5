#!/bin/bash
# other BASELINE scenes through bench.py at reduced spp (quick regression numbers)
for cfg in "--scene 1 --bvh 0 --width 1920 --height 1080 --spp 512" "--scene 101 --width 1920 --height 1080 --spp 256" "--scene 0 --bvh 0 --width 1024 --height 1024 --spp 256"; do
  echo -n "$cfg: "; timeout -k 10 200 python bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mray/s, kernel %.1f ms, checksum %d' % (d['value'], d['kernel_ms_per_step'], d['fb_checksum']))"
done
