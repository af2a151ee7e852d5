"""bench.py's GPU-free pieces: it imports without a GPU, knows the box's usable cores, finds the committed PMC entries of every scene
it prices, and refuses to run without a GPU with a message (the render path has no CPU fallback)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_helpers_without_gpu():
    b = _bench()
    cores, note = b.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1) and "sched_getaffinity" in note
    assert abs(b.ARCH_PEAK_GLANEOPS - 256 * 4 * 32 * 2.4) < 1e-9
    for scene in (100, 101, 1):                      # cfg 2 / 3, cfg 5, cfg 4: every workload of the bench line has a PMC entry
        entry, path = b.lane_ops_entry(scene)
        assert entry is not None and path.startswith("profiles/"), scene
        assert entry["lane_ops_per_ray"] > 500 and 16 < entry["lanes_per_valu_instruction"] < 64
        assert len(entry["kernel_code_sha256"]) == 64
    assert b.lane_ops_entry(12345) == (None, None)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in ([], ["--gpus", "2"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]        # and prints no line
