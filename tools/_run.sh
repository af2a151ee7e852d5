set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_t2; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 > $O/tests.log 2>&1; tail -9 $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.log; tail -2 $O/bench.log
python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['frac_arch'], d['cpu_baseline']['value'])"
