"""The C-ABI library loads and exports every symbol include/srt_c_api.h declares (no compute calls, no GPU)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "srt_c_api.h")).read()
    return sorted(set(re.findall(r"SRT_API\s+[\w\s\*]+?\b(srt_\w+)\s*\(", src)))


def test_header_and_binding_agree(srt):
    names = declared_symbols()
    assert len(names) >= 40
    assert sorted(srt.binding.PROTOTYPES) == names


def test_library_exports_everything(srt):
    L = C.CDLL(srt.binding.LIB_PATH)
    for name in declared_symbols():
        assert getattr(L, name) is not None
    assert b"gfx950" in srt.binding.lib().srt_version()


def test_no_gpu_fails_loudly(srt):
    """Without a usable GPU the device context cannot be created and nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = srt.binding.lib().srt_create(0, C.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in srt.binding.lib().srt_last_error(None)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("the cpu oracle", "").replace("cpu oracle", ""), os.path.join(dirpath, f)


def test_missing_rccl_is_reported_not_fatal(srt):
    """when RCCL cannot be loaded every srt_comm_* entry point returns SRT_ERR_UNSUPPORTED with a message (it used to
    dereference a null dlerror() and crash).  SRT_RCCL_LIB points the loader at a library that does not exist; the loader caches
    its result per process, so this runs in a child process.  No GPU is touched."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, importlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "srt = importlib.import_module('cuda-spectral-ray-tracer_amd')\n"
        "L = srt.binding.lib()\n"
        "assert L.srt_comm_available() == -5, L.srt_comm_available()\n"
        "buf = (C.c_ubyte * 128)()\n"
        "assert L.srt_comm_unique_id(buf) == -5\n"
        "h = C.c_void_p()\n"
        "dev = (C.c_int * 1)(0)\n"
        "assert L.srt_comm_init_all(dev, 1, C.byref(h)) == -5 and not h.value\n"
        "msg = L.srt_comm_last_error(None)\n"
        "assert b'cannot load RCCL' in msg and b'no-such-dir' in msg, msg\n"
        "print('ok')\n") % ROOT
    env = dict(os.environ, SRT_RCCL_LIB="/no-such-dir/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_kernel_identity_from_the_code_object(srt):
    """tools/kernel_id.py finds the production render kernels inside the gfx950 code object of the built library (offload bundle in
    .hip_fatbin -> ELF symbol table -> the kernel's bytes) -- what bench.py ties an imported PMC figure to.  No GPU needed."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("kernel_id", os.path.join(ROOT, "tools", "kernel_id.py"))
    kid = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kid)
    hs = kid.code_hashes(srt.binding.LIB_PATH)
    assert set(hs) == {(0, 0), (0, 1), (1, 0), (1, 1), (1, 1, 1), (0, 0, 1)}      # render_kernel<0, NARROW, ALL_CACHED> + the PAIRED variants of <0,1,1> and <0,0,0>
    assert all(len(v) == 64 and int(v, 16) >= 0 for v in hs.values()) and len(set(hs.values())) == 6
    assert kid.code_hash(srt.binding.LIB_PATH, 1, 1, 1)[0] == hs[(1, 1, 1)] and kid.variant_name(1, 1, 1) == "render_kernel<0,1,1,paired>"
    h, note = kid.code_hash(srt.binding.LIB_PATH, 1, 1)
    assert h == hs[(1, 1)] and "machine code" in note
    assert kid.code_hash("/no/such/library.so", 1, 1)[0] is None
    # the committed PMC entries name the kernel they were taken on; say (do not require) whether this build is that kernel
    f = os.path.join(ROOT, "profiles", "r04", "lane_ops_per_ray.json")
    if os.path.exists(f):
        doc = json.load(open(f))
        for scene, variant in (("scene_100", (1, 1)), ("scene_101", (0, 0)), ("scene_1", (1, 1))):
            assert len(doc[scene]["kernel_code_sha256"]) == 64
            print("%s: PMC pass %s this build's render_kernel<0,%d,%d>" % (scene, "==" if doc[scene]["kernel_code_sha256"] == hs[variant] else "!=", *variant))


def test_test_knobs_are_fenced():
    """no launch-time getenv decides the kernel variant.  srt_kernels.hip (the launch plan and the launchers) reads no
    environment at all; srt_capi.cpp reads SRT_WIDE_REFS / SRT_LDS_CACHE_MAX / SRT_DEBUG_LANE_LIMIT only inside the SRT_TEST_KNOBS=1
    block of srt_create; srt_comm.cpp honours SRT_COMM_TEST_SAME_DEVICE only next to SRT_TEST_KNOBS.  (The GPU suite checks the
    behaviour: tests/test_gpu_parity.py::test_stray_knob_variables_do_not_change_the_plan.)"""
    csrc = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "csrc")
    kern = open(os.path.join(csrc, "srt_kernels.hip")).read()
    assert "getenv" not in kern
    capi = open(os.path.join(csrc, "srt_capi.cpp")).read()
    a = capi.index('getenv("SRT_TEST_KNOBS")')
    b = capi.index("}", capi.index('getenv("SRT_LDS_CACHE_MAX")'))
    for name in ("SRT_WIDE_REFS", "SRT_LDS_CACHE_MAX", "SRT_DEBUG_LANE_LIMIT"):
        hits = [m.start() for m in re.finditer(r'getenv\("%s"\)' % name, capi)]
        assert len(hits) == 1 and a < hits[0] < b, name
    comm = open(os.path.join(csrc, "srt_comm.cpp")).read()
    i = comm.index('getenv("SRT_COMM_TEST_SAME_DEVICE")')
    assert 'getenv("SRT_TEST_KNOBS")' in comm[i:i + 200]
