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


def test_hsa_ipc_mode_is_set_before_torch_is_imported():
    """HSA_ENABLE_IPC_MODE_LEGACY is read when the HSA runtime starts, so bench.py must put it into the environment at
    module level, before the first `import torch` of the file (which lives inside main())."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    set_line, first_torch = None, None
    for node in tree.body:      # module level only
        if isinstance(node, ast.Expr) and "HSA_ENABLE_IPC_MODE_LEGACY" in ast.get_source_segment(src, node):
            set_line = node.lineno
    for node in ast.walk(tree):
        names = []
        if isinstance(node, ast.Import):
            names = [a.name for a in node.names]
        elif isinstance(node, ast.ImportFrom):
            names = [node.module or ""]
        if any(n == "torch" or n.startswith("torch.") for n in names):
            first_torch = node.lineno if first_torch is None else min(first_torch, node.lineno)
    assert set_line is not None and first_torch is not None and set_line < first_torch
    # and importing the module (no GPU needed) leaves the variable set
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    code = "import importlib.util, os; s = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "0", out.stderr[-1000:]


def test_traffic_is_attached_to_every_record_with_a_two_point_entry():
    b = _bench()
    entry = {"fabric_bytes_per_pixel": 100.0, "fabric_bytes_per_ray": 2.0, "fabric_note": "test"}
    t = b.traffic_of(entry, "profiles/x.json", pixels=1000, rays=50000.0, kms=2.0)
    assert t["traffic"] == 100.0 * 1000 + 2.0 * 50000 and abs(t["traffic_GBs"] - t["traffic"] / 2e-3 / 1e9) < 1e-9
    assert b.traffic_of({"lane_ops_per_ray": 1.0}, "x", 1, 1.0, 1.0) == {"traffic": None}
