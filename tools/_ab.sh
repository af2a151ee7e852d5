cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/asm
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/asm/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/asm/pytest.log
bash tools/exp_variants.sh 2>&1 | tee gpurun_out/asm/variants.txt
