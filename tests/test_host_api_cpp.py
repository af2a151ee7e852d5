"""The C++ host mirror (csrc/host_api.hpp: scene_manager / camera_builder / frame_buffer / render_manager)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import oracle_scene_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "host_api_demo")


def demo_is_stale():
    """The demo embeds the struct layouts of include/srt_c_api.h: rebuild whenever a header or the library is newer."""
    if not os.path.exists(EXE):
        return True
    deps = [os.path.join(ROOT, "include", "srt_c_api.h"), os.path.join(PKG, "csrc", "host_api.hpp"), os.path.join(PKG, "libsrt_hip.so"),
            os.path.join(ROOT, "tests", "cpp", "host_api_demo.cpp")]
    return any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps if os.path.exists(d))


def build_demo():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "host_api_demo.cpp"),
                           "-L" + PKG, "-lsrt_hip", "-Wl,-rpath," + PKG, "-lpthread"])


def test_cpp_mirror_compiles_and_links(srt):
    build_demo()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,spp", [((0, 0), 6), ((40, 40), 6), ((40, 40), 12)])   # 12 spp: the cost probe + ordered queue run per chunk
def test_render_manager_matches_oracle(srt, orc, tmp_path, chunk, spp):
    if demo_is_stale():
        build_demo()
    W, H, depth = 72, 56, 8
    out = str(tmp_path / "img.bin")
    subprocess.check_call([EXE, "1", str(W), str(H), str(spp), str(depth), str(chunk[0]), str(chunk[1]), out], timeout=120)
    got = np.fromfile(out, np.float32).reshape(3, H, W)
    scene = srt.Scene.builtin(srt.SCENE_PRISM).build_bvh(srt.BVH_REFERENCE, 1984)
    osc = oracle_scene_for(orc, scene, 0)
    cam = scene.default_camera(W, H)
    cw, ch = chunk if chunk[0] else (W, H)
    bx, by = cw // 28 + 1, ch // 16 + 1
    # the reference's chunk walk (render_manager.cu:3-66): RNG states persist across chunks (Q13)
    import ctypes as C
    states = np.zeros(28 * 16 * bx * by * 6, np.uint32)
    seeded = False
    want = np.zeros((3, H * W), np.float32)
    for oy in range(0, H, ch):
        for ox in range(0, W, cw):
            w, h = min(cw, W - ox), min(ch, H - oy)
            if not seeded:
                # init_random_states: XORWOW(1984 + idx) for every lane of the chunk grid
                for idx in range(28 * 16 * bx * by):
                    s = orc.Rng()
                    orc.lib().orc_rng_init(1984 + idx, C.byref(s))
                    states[6 * idx: 6 * idx + 6] = [s.d] + list(s.v)
                seeded = True
            r = osc.render(cam, w, h, spp, depth, bx=bx, by=by, offx=ox, offy=oy, states=states)
            for c in range(3):
                img = orc.unswizzle(r["fb"][c], 28, 16, bx, by, w, h, ox, oy, W, H)
                mask = np.zeros((H, W), bool)
                mask[oy:oy + h, ox:ox + w] = True
                want[c][mask.ravel()] = img[mask.ravel()]
    assert np.array_equal(got.reshape(3, -1), want)


@pytest.mark.gpu
def test_cli_driver_writes_bmp_and_log(srt, orc, tmp_path):
    """srt_render: the reference's flags (io/params.h:236-304), renders/<title>.bmp and logs/<ts>_<title>_log.txt."""
    exe = os.path.join(PKG, "srt_render")
    assert os.path.exists(exe)
    subprocess.check_call([exe, "-s", "1", "-xr", "64", "-ar", "4/3", "-ns", "4", "-bl", "8", "-t", "Prism Test", "--save", "--do-log",
                           "--no-show", "-lsub", "unit"], cwd=str(tmp_path), timeout=120)
    bmp = tmp_path / "renders" / "prism_test.bmp"                        # string_to_filename: lower case, spaces -> _
    raw = bmp.read_bytes()
    assert raw[:2] == b"BM" and int.from_bytes(raw[18:22], "little") == 64 and int.from_bytes(raw[22:26], "little") == 48
    logs = list((tmp_path / "logs" / "unit").glob("*_prism_test_log.txt"))
    assert len(logs) == 1
    text = logs[0].read_text()
    for key in ("image width: 64", "image height: 48", "# primitives: 20", "# materials: 3", "samples per pixel: 4", "bounce limit: 8",
                "chunk width: 64", "threads x: 28", "blocks y: 4", "total rendering time (seconds):", "Mray/s:"):
        assert key in text, key
    # pixels = the oracle's quantised framebuffer (BMP rows are bottom-up, BGR)
    W, H = 64, 48
    scene = srt.Scene.builtin(srt.SCENE_PRISM).build_bvh(srt.BVH_REFERENCE, 1984)
    ref = oracle_scene_for(orc, scene, 0).render(scene.default_camera(W, H), W, H, 4, 8)
    img = np.frombuffer(raw[54:], np.uint8).reshape(H, W * 3)[::-1].reshape(H, W, 3)
    for c, plane in enumerate(ref["fb"]):
        want = orc.unswizzle(plane, 28, 16, W // 28 + 1, H // 16 + 1, W, H, 0, 0, W, H).reshape(H, W).astype(np.uint8)
        assert np.array_equal(img[:, :, 2 - c], want)


@pytest.mark.gpu
def test_cli_driver_two_and_three_ranks_mock_transport(srt, tmp_path):
    """srt_render --gpus N: host_api.hpp's render_manager::init_renderer(bounce, spp, devices) -> srt_comm_init_all, chunks rendered
    with srt_render_frame_multi.  On ONE GPU over the test transport (tests/cpp/mock_rccl.cpp; every rank on device 0): the BMP of a
    2-rank and of a 3-rank run, whole-image and in 40x24 chunks, equals the single-GPU run's byte for byte."""
    exe = os.path.join(PKG, "srt_render")
    mock = os.path.join(ROOT, "tests", "cpp", "_build", "libmock_rccl.so")
    if not os.path.exists(mock):
        pytest.skip("tests/cpp/_build/libmock_rccl.so not built (__graft_entry__.build())")
    base = ["-s", "0", "-xr", "96", "-ar", "4/3", "-ns", "12", "-bl", "8", "--save", "--no-show"]
    def run(title, extra, env=None):
        subprocess.check_call([exe] + base + ["-t", title] + extra, cwd=str(tmp_path), timeout=180, env=env)
        return (tmp_path / "renders" / (title + ".bmp")).read_bytes()
    env = dict(os.environ, SRT_RCCL_LIB=mock, SRT_COMM_TEST_SAME_DEVICE="1", SRT_TEST_KNOBS="1")
    for chunks in ([], ["-xc", "40", "-yc", "24"]):
        one = run("one" + str(len(chunks)), chunks)
        assert one[:2] == b"BM" and len(one) == 54 + 96 * 72 * 3 and any(one[54:])
        for n in (2, 3):
            assert run("ranks%d_%d" % (n, len(chunks)), chunks + ["--gpus", str(n)], env) == one, (n, chunks)


@pytest.mark.gpu
def test_cli_driver_tunes_its_own_tree_for_throughput_bound_renders(srt, tmp_path):
    """srt_render --sah (this build's own tree): scene_manager::tune_tree_for_throughput (host_api.hpp) -- reinsertion + child order
    from a probe frame when the render has at least 6 pixels per lane, the tree as built otherwise.  The tree is an input of the
    traversal: the image of the tuned run equals the untuned run's except where two triangles tie exactly in t (a handful of
    bytes at most), and a chain-bound render is byte-identical."""
    exe = os.path.join(PKG, "srt_render")
    def run(title, xres, env_extra):
        p = subprocess.run([exe, "-s", "100", "--sah", "-xr", str(xres), "-ar", "16/9", "-ns", "2", "-bl", "8", "-t", title, "--save", "--no-show"],
                           cwd=str(tmp_path), capture_output=True, timeout=300, env=dict(os.environ, **env_extra))
        assert p.returncode == 0, p.stderr[-500:]
        return (tmp_path / "renders" / (title + ".bmp")).read_bytes(), p.stderr.decode(errors="replace")
    big_tuned, log_tuned = run("big_tuned", 1920, {})
    big_plain, log_plain = run("big_plain", 1920, {"SRT_NO_TREE_TUNING": "1"})
    assert "BVH: 3 reinsertion passes; child order profiled (480x270 x 8 spp probe):" in log_tuned and "nodes swapped" in log_tuned
    # ONE probe recipe in every front end: the Python helper bench.py uses arrives at the same tree for the same workload
    import re
    r = srt.Renderer(0)
    scene = srt.Scene.builtin(100, 0).build_bvh(srt.BVH_SAH, 1984)
    r.upload_scene(scene)
    note = srt.tune_tree_for_throughput(r, scene, 1920, 1080, 8)
    r.close()
    assert re.search(r"(\d+) nodes swapped", note).group(1) == re.search(r"(\d+) nodes swapped", log_tuned).group(1), (note, log_tuned)
    assert "BVH: tree as built (SRT_NO_TREE_TUNING)" in log_plain
    a, b = np.frombuffer(big_tuned, np.uint8), np.frombuffer(big_plain, np.uint8)
    assert a.size == b.size == 54 + 1920 * 1080 * 3 and int(np.count_nonzero(a != b)) <= 64
    small_tuned, log_small = run("small_tuned", 320, {})
    small_plain, _ = run("small_plain", 320, {"SRT_NO_TREE_TUNING": "1"})
    assert "chain-bound launch" in log_small and small_tuned == small_plain


@pytest.mark.gpu
def test_cli_driver_reports_failure(srt, tmp_path):
    """A render that cannot run (device index that does not exist; more GPUs than the box has) must not exit 0 with a black
    image: the reference dies in checkCudaErrors -> exit(99) (utils/cuda_utility.cu:8-18), this driver returns non-zero."""
    exe = os.path.join(PKG, "srt_render")
    for extra in (["--gpu", "63"], ["--gpus", "64"]):
        p = subprocess.run([exe, "-s", "1", "-xr", "32", "-ar", "4/3", "-ns", "2", "-bl", "4", "-t", "fail", "--no-show"] + extra,
                           cwd=str(tmp_path), capture_output=True, timeout=120)
        assert p.returncode != 0, extra
        assert b"renderer:" in p.stderr or b"not yet initialized" in p.stderr, p.stderr[-300:]
