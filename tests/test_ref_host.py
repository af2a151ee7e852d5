"""Pinning of the host-side rows f1 / f2 / f3 against the REFERENCE ITSELF, as far as it compiles in this image without stand-ins:
primitives/transform.cu, io/params.cpp (+ io/params.h), _log_/log_context.cpp (+ utils/utility.h), image/image.cpp and
io/save_image.cpp (+ the vendored CImg.h) build unmodified with hipcc -x hip from where they lie under /root/reference (recipe:
oracle/Makefile target `ref`; harnesses of ours: oracle/ref_host_driver.cpp, oracle/ref_image_driver.cpp; outputs
oracle/_ref/libref_host.so, libref_image.so, git-ignored, built by __graft_entry__.build() where the reference is mounted).

  * transform::assign_rot_matrix (transform.cu:4-34)  == srt_rotation_matrix (the matrix Builder::rotate_about_origin uses), bit for bit
  * param_manager::parseArgs (io/params.h:236-304)    == srt_cli::parseArgs (csrc/srt_cli.hpp, what srt_render parses with) on a table of argv's
  * log_context (log_context.cpp:5-125)               == srt_cli::log_context: directory, file name (up to the time stamp), content
  * get_image + save_img (image.cpp:3-18, save_image.cpp:8-20: CImg's BMP) == srt_cli::save_img: the same file, byte for byte

The product side of the last three is reached through tests/cpp/cli_driver.cpp (the header srt_render is built from) and, for the
parser, through the srt_render binary itself (--dump-params)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HOST = os.path.join(ROOT, "oracle", "_ref", "libref_host.so")
REF_IMAGE = os.path.join(ROOT, "oracle", "_ref", "libref_image.so")
CLI_SO = os.path.join(ROOT, "tests", "cpp", "_build", "libcli_driver.so")
SRT_RENDER = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "srt_render")


class Params(C.Structure):
    _fields_ = [("title", C.c_char * 256), ("log_subdir", C.c_char * 256), ("scene", C.c_uint), ("xres", C.c_uint), ("yres", C.c_uint),
                ("ar", C.c_float), ("xcsize", C.c_uint), ("ycsize", C.c_uint), ("n_samples", C.c_uint), ("bounce_limit", C.c_uint),
                ("do_log", C.c_int), ("show_render", C.c_int), ("do_save", C.c_int)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["ar"] = np.float32(d["ar"]).view(np.uint32).item()      # compared as bits
        return d


def _build_ref():
    if os.path.isdir("/root/reference/io") and not (os.path.exists(REF_HOST) and os.path.exists(REF_IMAGE)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])


def _ref_host():
    _build_ref()
    if not os.path.exists(REF_HOST):
        pytest.skip("oracle/_ref/libref_host.so not built (the reference is not mounted here)")
    return C.CDLL(REF_HOST)


def _cli():
    if not os.path.exists(CLI_SO):
        os.makedirs(os.path.dirname(CLI_SO), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-fvisibility=hidden", "-o", CLI_SO, os.path.join(ROOT, "tests", "cpp", "cli_driver.cpp")])
    return C.CDLL(CLI_SO)


def _deg2rad(d):      # degrees_to_radians, utils/cuda_utility.cuh:40-43 with PI of utils/utility.h:10, in fp32
    return np.float32(np.float32(np.float32(d) * np.float32(3.1415926535897932385)) / np.float32(180.0))


def test_rotation_matrix_equals_reference_function(srt):
    """transform::assign_rot_matrix compiled from the reference == srt_rotation_matrix for X / Y / Z, the three angles the built-in
    scenes use (+25, -18, +10 degrees: scene/scene.cu:116,121,127,166) and a sweep, bit for bit -- layout and sense of the matrix
    (f1) rest on reference code.  An unknown axis leaves the matrix untouched on both sides."""
    R = _ref_host()
    L = srt.binding.lib()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(3)
    angles = [_deg2rad(25.0), _deg2rad(-18.0), _deg2rad(10.0), np.float32(0.0), np.float32(-0.0), np.float32(np.pi), np.float32(1e-8), np.float32(1e8)]
    angles += list(rng.uniform(-10, 10, 500).astype(np.float32))
    for axis in (0, 1, 2, 3, 7):
        for th in angles:
            start = rng.uniform(-1, 1, 9).astype(np.float32) if axis in (0, 7) else np.eye(3, dtype=np.float32).reshape(9)
            a, b = start.copy(), start.copy()
            assert R.ref_rot_matrix(C.c_float(float(th)), axis, a.ctypes.data_as(fp)) == 0
            assert L.srt_rotation_matrix(C.c_float(float(th)), axis, b.ctypes.data_as(fp)) == 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (axis, float(th), a, b)
    # the sense of the Y rotation, spelled out: vec3::matrix_mul (math/vec3.cuh:80-91) maps (1, 0, 0) to (cos, 0, -sin)
    m = np.eye(3, dtype=np.float32).reshape(9)
    R.ref_rot_matrix(C.c_float(float(_deg2rad(25.0))), 2, m.ctypes.data_as(fp))
    v = m.reshape(3, 3) @ np.array([1.0, 0.0, 0.0], np.float32)
    assert v[0] > 0.9 and v[2] < -0.4 and v[1] == 0.0


def _rotated_square(centre, half, theta):
    """corners (+-half, +-half) about `centre` in the x-z plane, turned by the reference's Y matrix: x' = c x + s z, z' = -s x + c z
    (transform.cu:18-23 applied by vec3::matrix_mul)"""
    c, s = np.cos(theta), np.sin(theta)
    out = []
    for x, z in ((-half, -half), (half, -half), (half, half), (-half, half)):
        out.append((centre[0] + c * x + s * z, centre[1] - s * x + c * z))
    return np.array(out)


def test_cornell_boxes_and_pyramid_are_turned_in_the_reference_sense(srt):
    """SIGN-SENSITIVE geometry check of CORNELL (scene/scene.cu:115-128): box1 is turned by +25 degrees, box2 and the pyramid by
    -18 degrees about the vertical axis through their own centres, in the sense of the reference's rotation matrix (pinned above).
    A transposed matrix or a flipped angle moves every base corner by tens of units and fails."""
    tris = srt.Scene.builtin(srt.SCENE_CORNELL, 0).triangles()

    def base_corners(ts, y):
        pts = []
        for t in ts:
            for v in (t.v0, t.v1, t.v2):
                p = np.array(list(v), np.float64)
                if abs(p[1] - y) < 1e-3 and not any(np.linalg.norm(p - q) < 1e-3 for q in pts):
                    pts.append(p)
        return np.array(pts)[:, [0, 2]]

    def same_point_set(a, b, tol=2e-3):
        return a.shape == b.shape and all(np.min(np.linalg.norm(b - p, axis=1)) < tol for p in a)

    for first, count, y, angle, shift in ((12, 12, 0.0, 25.0, (265.0, 295.0)), (24, 12, 0.0, -18.0, (130.0, 65.0)), (36, 6, 166.0, -18.0, (130.0, 65.0))):
        got = base_corners(tris[first:first + count], y)
        centre = (shift[0] + 82.5, shift[1] + 82.5)
        want = _rotated_square(centre, 82.5, float(_deg2rad(angle)))
        wrong = _rotated_square(centre, 82.5, float(_deg2rad(-angle)))
        assert got.shape == (4, 2)
        assert same_point_set(got, want), (first, got, want)
        assert not same_point_set(got, wrong)                      # the test can tell the two senses apart


ARGV_TABLE = [
    [],
    ["-s", "1"], ["--scene", "2", "-t", "My Title"], ["-xr", "1920", "-ar", "16/9"], ["-ar", "16/9", "-xr", "1920"], ["--xres", "333", "--aspect-ratio", "1.5"],
    ["-ar", "4/3/2", "-xr", "100"], ["-xr", "1", "-ar", "1000"], ["-ar", "0.5"], ["-xc", "40"], ["-yc", "40"], ["-xc", "40", "-yc", "25"],
    ["--xcsize", "0", "--ycsize", "0"], ["-ns", "1024", "-bl", "16"], ["--nsamples", "7", "--bounce-limit", "0"],
    ["--do-log", "--no-show", "--save"], ["-lsub", "runs/a b", "-t", "Prism 1"], ["-t"], ["-s"], ["--save", "-xr"],
    ["-xr", "abc"], ["-xr", "700abc"], ["-ar", "x"], ["-ar", "16/x", "-xr", "200"], ["-ns", ""], ["-s", "1x"], ["-bl", "-1"], ["-ns", "99999999999999999999"],
    ["--bogus", "-xr", "50"], ["-xr", "50", "--bogus"], ["-t", "-s", "-s", "2"], ["-xr", "640", "-xr", "320", "-ar", "2", "-ar", "4/1"],
    ["-s", "5", "-t", "Synthetic five"], ["-s", "100"], ["-s", "2", "-xr", "1280", "-ar", "1280/720", "-ns", "256", "-bl", "16", "-xc", "640", "--do-log", "-lsub", "x"],
]


@pytest.mark.parametrize("argv", ARGV_TABLE, ids=lambda a: " ".join(a) or "defaults")
def test_parse_args_equals_reference_param_manager(argv):
    """param_manager::parseArgs (compiled from io/params.h) vs srt_cli::parseArgs on the same argv: every parsed value incl. the
    defaults (xres 600, AR 1, spp 500, bounce 10), yres = uint(xres / ar), the xcsize / ycsize defaulting rules, flags that miss
    their value, values that do not parse (previous value kept), repeated flags, unknown flags."""
    R, P = _ref_host(), _cli()
    full = [b"prog"] + [a.encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    a, b = Params(), Params()
    for fn in (R.ref_parse_args, P.cli_parse_args):
        fn.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Params)]
    assert R.ref_parse_args(len(full), arr, C.byref(a)) == 0
    assert P.cli_parse_args(len(full), arr, C.byref(b)) == 0
    assert a.as_dict() == b.as_dict()
    # ... and the binary itself (reference flags only: its own flags are not in this table)
    out = subprocess.run([SRT_RENDER, "--dump-params"] + argv, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-500:]
    kv = dict(l.split("=", 1) for l in out.stdout.splitlines() if re.match(r"^[a-z_]+=", l))
    d = a.as_dict()
    if d["scene"] < 3 or d["title"]:
        assert kv["title"] == d["title"].decode()
    assert kv["log_subdir"] == d["log_subdir"].decode()
    for k in ("scene", "xres", "yres", "xcsize", "ycsize", "n_samples", "bounce_limit", "do_log", "show_render", "do_save"):
        assert int(kv[k]) == d[k], (k, kv[k], d[k])
    assert np.float32(float(kv["ar"])).view(np.uint32).item() == d["ar"]


def _log_both(tmp_path, title, subdir, entries):
    R, P = _ref_host(), _cli()
    n = len(entries)
    names = (C.c_char_p * n)(*[e[0].encode() for e in entries])
    kinds = (C.c_int * n)(*[e[1] for e in entries])
    svals = (C.c_char_p * n)(*[(e[2] if isinstance(e[2], str) else "").encode() for e in entries])
    dvals = (C.c_double * n)(*[(float(e[2]) if not isinstance(e[2], str) else 0.0) for e in entries])
    res = {}
    for tag, fn in (("ref", R.ref_log_to_file), ("own", P.cli_log_to_file)):
        fn.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
        d = tmp_path / tag
        d.mkdir()
        cwd = os.getcwd()
        os.chdir(d)
        try:
            buf = C.create_string_buffer(1 << 16)
            assert fn(title.encode(), subdir.encode(), n, names, kinds, svals, dvals, buf, C.sizeof(buf)) == 0
        finally:
            os.chdir(cwd)
        files = sorted(str(p.relative_to(d)) for p in d.rglob("*") if p.is_file())
        assert len(files) == 1, files
        res[tag] = (files[0], (d / files[0]).read_text(), buf.value.decode())
    return res


@pytest.mark.parametrize("title,subdir", [("Cornell Box", ""), ("My Render 01", "runs/Batch A"), ("UPPER lower", "x"), ("t", "a/b/c")])
def test_run_log_equals_reference_log_context(tmp_path, title, subdir):
    """The run log as main.cpp:158-160 sets it up and to_file() writes it (log_context.cpp:5-65): directory logs[/<subdir>], file name
    <epoch ms>_<title>_log.txt lower-cased with blanks as underscores, one `key: value` line per entry in insertion order; integers via
    to_string, float with 7 and double with 16 significant digits, a key added twice printed twice with its last value, sum_value."""
    entries = [("image width", 1, 1920), ("image height", 1, 1080), ("scene type", 0, "Cornell Box"), ("# primitives", 2, 42), ("# materials", 2, 7),
               ("samples per pixel", 1, 1024), ("bounce limit", 1, 16), ("chunk width", 1, 1920), ("chunk total byte size", 2, 1920 * 1080 * 12),
               ("signed", 3, -5), ("a float", 4, 0.1), ("another float", 4, 123456.789), ("tiny float", 4, 1.5e-12), ("huge float", 4, 3.0e30),
               ("total rendering time (seconds)", 5, 0.36143217), ("a double", 5, 1.0 / 3.0), ("big double", 5, 1.0e22), ("whole double", 5, 12.0),
               ("chunk width", 1, 640), ("acc", 6, 1.25), ("acc", 6, 2.5), ("acc", 6, 1e-3), ("scene type", 6, 1.0), ("empty", 0, "")]
    res = _log_both(tmp_path, title, subdir, entries)
    (rf, rtext, rbuf), (of, otext, obuf) = res["ref"], res["own"]
    strip = lambda f: re.sub(r"(^|/)\d{10,}_", r"\1<ms>_", f)
    assert re.search(r"(^|/)\d{13}_", rf) and re.search(r"(^|/)\d{13}_", of)          # epoch milliseconds lead the name
    assert strip(rf) == strip(of), (rf, of)
    assert rtext == otext and rbuf == obuf and rtext == rbuf
    assert strip(rf).endswith("<ms>_" + title.lower().replace(" ", "_") + "_log.txt") and strip(rf).startswith("logs/")


def test_bmp_writer_equals_reference_cimg_output(tmp_path):
    """get_image + save_img of the reference (image/image.cpp:3-18, io/save_image.cpp:8-20; CImg's BMP writer) vs srt_cli::save_img
    for the same uchar planes: renders/<name>.bmp, the same bytes -- odd widths (row padding), 1 x 1, a gradient, noise."""
    _build_ref()
    if not os.path.exists(REF_IMAGE):
        pytest.skip("oracle/_ref/libref_image.so not built (the reference is not mounted here)")
    try:
        R = C.CDLL(REF_IMAGE)
    except OSError as e:      # (libX11 is a link-time dependency of the vendored CImg.h)
        pytest.skip("libref_image.so does not load here: %s" % e)
    P = _cli()
    up = C.POINTER(C.c_ubyte)
    rng = np.random.default_rng(11)
    for k, (w, h) in enumerate(((1, 1), (2, 3), (5, 4), (7, 1), (64, 48), (251, 13))):
        planes = [rng.integers(0, 256, w * h, dtype=np.uint8) for _ in range(3)]
        if k == 4:
            planes = [(np.arange(w * h) * (c + 1) % 256).astype(np.uint8) for c in range(3)]
        name = "img %d.bmp" % k
        out = {}
        for tag, fn in (("ref", R.ref_save_image), ("own", P.cli_save_image)):
            fn.argtypes = [up, up, up, C.c_uint, C.c_uint, C.c_char_p]
            d = tmp_path / ("%s%d" % (tag, k))
            d.mkdir()
            cwd = os.getcwd()
            os.chdir(d)
            try:
                assert fn(planes[0].ctypes.data_as(up), planes[1].ctypes.data_as(up), planes[2].ctypes.data_as(up), w, h, name.encode()) == 0
            finally:
                os.chdir(cwd)
            out[tag] = (d / "renders" / name).read_bytes()
        assert out["ref"] == out["own"], (w, h, len(out["ref"]), len(out["own"]), out["ref"][:54].hex(), out["own"][:54].hex())
