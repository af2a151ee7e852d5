"""The XORWOW recurrence of the oracle (orc_rng_next) and of the product (rng_next, srt_device.h) against an INDEPENDENT implementation:
rocRAND's rocrand_device::xorwow_engine::next() (ROCm's own header, compiled as it is by oracle/rocrand_xorwow_pin.hip, on the host and in a kernel).

The reference draws everything from cuRAND's XORWOW (utils/cuda_utility.cu:19-26, rendering/rendering.cu:137); cuRAND is not in this image.
rocRAND implements the same published generator, so from identical state words the two must produce identical outputs and identical next states
-- that is what these tests hold.  rocRAND's SEEDING and its uniform float MAPPING are deliberately its own (different scramble constants; 2^-32 + v 2^-32
instead of v 2^-32 + 2^-33), so curand_init's scramble and curand_uniform's mapping stay restated from the published definition and are not pinned
here: test_rocrand_seeding_and_mapping_are_not_curands states exactly that, so that nobody reads more into this file than it proves."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_SO = os.path.join(ROOT, "oracle", "_build", "librocrand_pin.so")
u32 = np.uint32


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def pin():
    if not os.path.exists(PIN_SO):
        import subprocess
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(PIN_SO):
        pytest.skip("oracle/_build/librocrand_pin.so not built (needs hipcc + /opt/rocm/include/rocrand)")
    L = C.CDLL(PIN_SO)
    for f in ("rr_host_steps", "rr_host_digest", "rr_device_digest", "rr_rocrand_seed_words"):
        getattr(L, f).restype = C.c_int
    L.rr_rocrand_uniform.restype = C.c_float
    L.rr_rocrand_uniform.argtypes = [C.c_uint32]
    L.rr_rocrand_seed_words.argtypes = [C.c_uint64, C.c_void_p]
    return L


def _curand_seed_words(seeds):
    """curand_init(seed, 0, 0) for 32-bit seeds, numpy restatement (the same as tests/test_gpu_parity.py::_xorwow_host): rows of (d, x0..x4)."""
    with np.errstate(over="ignore"):
        s0 = seeds.astype(u32) ^ u32(0xaad26b49); s1 = np.zeros_like(s0) ^ u32(0xf7dcefdd)
        t0 = u32(1099087573) * s0; t1 = u32(2591861531) * s1
        return np.stack([u32(6615241) + t1 + t0, u32(123456789) + t0, u32(362436069) ^ t0, u32(521288629) + t1, u32(88675123) ^ t1, u32(5783321) + t0], axis=1).astype(u32)


def _numpy_digest(words, steps):
    """xor / weighted wrapping sum / last of `steps[i]` outputs of stream i and the words afterwards -- vectorised restatement of the recurrence."""
    w = words.copy(); n = len(w)
    xo = np.zeros(n, u32); sm = np.zeros(n, u32); last = np.zeros(n, u32)
    with np.errstate(over="ignore"):
        for k in range(int(steps.max())):
            m = steps > k
            t = w[:, 1] ^ (w[:, 1] >> u32(2))
            n4 = (w[:, 5] ^ (w[:, 5] << u32(4))) ^ (t ^ (t << u32(1)))
            new = np.stack([w[:, 0] + u32(362437), w[:, 2], w[:, 3], w[:, 4], w[:, 5], n4], axis=1)
            w = np.where(m[:, None], new, w)
            v = w[:, 5] + w[:, 0]
            xo = np.where(m, xo ^ v, xo); sm = np.where(m, sm + v * u32(2 * k + 1), sm); last = np.where(m, v, last)
    return xo, sm, last, w


def test_oracle_recurrence_equals_rocrand_host(pin, orc):
    """orc_rng_init + orc_rng_next, output by output and word by word, against rocRAND's engine started from the oracle's words: the streams the
    reference's lanes own (seed 1984 + idx, rendering.cu:137), 64-bit seeds, and arbitrary state words (incl. all-zero xorshift words and d = 0)."""
    L = orc.lib()
    rng = np.random.default_rng(5)
    seeds = [1984, 1985, 1984 + 2073599, 0, 1, 0xffffffff, 0x100000000, 0xdeadbeefcafef00d] + [int(x) for x in rng.integers(0, 1 << 63, 24)]
    n = 4096
    for seed in seeds:
        s = orc.Rng(); L.orc_rng_init(C.c_uint64(seed), C.byref(s))
        words = np.array([s.d] + list(s.v), dtype=u32)
        out = np.zeros(n, u32); after = np.zeros(6, u32)
        assert pin.rr_host_steps(_ptr(words), n, _ptr(out), _ptr(after)) == 0
        mine = np.array([L.orc_rng_next(C.byref(s)) for _ in range(n)], dtype=u32)
        assert np.array_equal(mine, out), "seed %#x: first differing draw %d" % (seed, int(np.argmax(mine != out)))
        assert np.array_equal(np.array([s.d] + list(s.v), dtype=u32), after)
    for words in [np.zeros(6, u32), np.array([0, 1, 0, 0, 0, 0], u32), np.full(6, 0xffffffff, u32)] + [rng.integers(0, 1 << 32, 6, dtype=np.uint64).astype(u32) for _ in range(16)]:
        s = orc.Rng(); s.d = int(words[0])
        for k in range(5): s.v[k] = int(words[1 + k])
        out = np.zeros(512, u32); after = np.zeros(6, u32)
        assert pin.rr_host_steps(_ptr(words), 512, _ptr(out), _ptr(after)) == 0
        mine = np.array([L.orc_rng_next(C.byref(s)) for _ in range(512)], dtype=u32)
        assert np.array_equal(mine, out) and np.array_equal(np.array([s.d] + list(s.v), dtype=u32), after)


def test_numpy_restatement_equals_rocrand_host(pin):
    """The vectorised numpy restatement the GPU tests use (seed scramble as curand_init's, then the recurrence) against rocRAND on 2^14 streams with
    ragged step counts -- so that the GPU test below compares three independent parties, not two copies of one."""
    rng = np.random.default_rng(6)
    n = 1 << 14
    seeds = np.concatenate([np.arange(1984, 1984 + n // 2, dtype=u32), rng.integers(0, 1 << 32, n // 2, dtype=np.uint64).astype(u32)])
    words = _curand_seed_words(seeds)
    steps = rng.integers(0, 300, n).astype(u32)
    xo, sm, last, after = _numpy_digest(words, steps)
    # the harness takes one step count per call: group the streams by count
    for c in np.unique(steps)[::37]:
        idx = np.nonzero(steps == c)[0]
        w = np.ascontiguousarray(words[idx]); dig = np.zeros((len(idx), 3), u32); wo = np.zeros((len(idx), 6), u32)
        assert pin.rr_host_digest(_ptr(w), len(idx), int(c), _ptr(dig), _ptr(wo)) == 0
        assert np.array_equal(dig[:, 0], xo[idx]) and np.array_equal(dig[:, 1], sm[idx]) and np.array_equal(wo, after[idx])
        if c: assert np.array_equal(dig[:, 2], last[idx])


def test_rocrand_seeding_and_mapping_are_not_curands(pin, orc):
    """What this file does NOT pin, stated as a test: rocRAND scrambles seeds with its own constants and maps to floats its own way, so the streams of
    rocrand_init(seed) are not curand_init(seed)'s and only the recurrence above is common ground."""
    L = orc.lib()
    w = np.zeros(6, u32)
    assert pin.rr_rocrand_seed_words(C.c_uint64(1984), _ptr(w)) == 0
    s = orc.Rng(); L.orc_rng_init(C.c_uint64(1984), C.byref(s))
    assert not np.array_equal(w, np.array([s.d] + list(s.v), dtype=u32))
    # the unscrambled constants under both scrambles are Marsaglia's (seed-independent part): recover them from rocRAND's own scramble
    with np.errstate(over="ignore"):
        t0 = u32(1228688033) * (u32(1984) ^ u32(0x2c7f967f)); t1 = u32(2073658381) * u32(0xa03697cb)
        base = np.array([w[0] - t1 - t0, w[1] - t0, w[2] ^ t0, w[3] - t1, w[4] ^ t1, w[5] - t0], dtype=u32)
    assert list(base) == [6615241, 123456789, 362436069, 521288629, 88675123, 5783321]       # the same start words the oracle scrambles (srt_oracle.c orc_rng_init)
    assert pin.rr_rocrand_uniform(0) == np.float32(2.0 ** -32)                                 # rocRAND: 2^-32 + v 2^-32
    z = orc.Rng(); z.d = 0
    for k in range(5): z.v[k] = 0
    f = np.float32                                                                            # the oracle's: v 2^-32 + 2^-33 (curand_uniform as published), from the zero state: v = 362437
    assert f(L.orc_random_float(C.byref(z))) == f(362437) * f(2.3283064e-10) + f(2.3283064e-10) / f(2.0) != f(pin.rr_rocrand_uniform(362437))


@pytest.mark.gpu
def test_product_rng_next_equals_rocrand_on_device(pin, gpu):
    """rng_seed + rng_next of the product (op-sweep kinds 31-33: xor of the outputs, weighted sum, fold of the words afterwards; 2^16 streams, ragged step
    counts up to 2 000) against rocRAND's xorwow_engine::next() run ON THE GPU from the words curand_init's scramble gives those seeds, and both against
    the numpy restatement."""
    rng = np.random.default_rng(7)
    n = 1 << 16
    seeds = np.concatenate([np.arange(1984, 1984 + n // 2, dtype=u32), rng.integers(0, 1 << 32, n // 2, dtype=np.uint64).astype(u32)])
    words = _curand_seed_words(seeds)
    for c in (0, 1, 5, 64, 2000):
        steps = np.full(n, c, u32)
        dig = np.zeros((n, 3), u32); wo = np.zeros((n, 6), u32)
        rc = pin.rr_device_digest(_ptr(np.ascontiguousarray(words)), n, c, _ptr(dig), _ptr(wo))
        assert rc == 0, "hip error %d in the rocRAND harness" % rc
        with np.errstate(over="ignore"):
            fold = wo[:, 0] ^ (wo[:, 1] * u32(3)) ^ (wo[:, 2] * u32(5)) ^ (wo[:, 3] * u32(7)) ^ (wo[:, 4] * u32(11)) ^ (wo[:, 5] * u32(13))
        a = seeds.view(np.float32); b = steps.view(np.float32)
        for kind, want in ((31, dig[:, 0]), (32, dig[:, 1]), (33, fold)):
            got = gpu.op_sweep(kind, a, b).view(u32)
            assert np.array_equal(got, want), "kind %d, %d steps: %d of %d streams differ from rocRAND" % (kind, c, int(np.sum(got != want)), n)
        if c <= 64:
            xo, sm, _, after = _numpy_digest(words, steps)
            assert np.array_equal(xo, dig[:, 0]) and np.array_equal(sm, dig[:, 1]) and np.array_equal(after, wo)
    # ragged: every lane of a wave runs another number of steps (the loop of kinds 31-33 diverges)
    steps = rng.integers(0, 400, n).astype(u32)
    xo, sm, _, after = _numpy_digest(words, steps)
    with np.errstate(over="ignore"):
        fold = after[:, 0] ^ (after[:, 1] * u32(3)) ^ (after[:, 2] * u32(5)) ^ (after[:, 3] * u32(7)) ^ (after[:, 4] * u32(11)) ^ (after[:, 5] * u32(13))
    a = seeds.view(np.float32); b = steps.view(np.float32)
    for kind, want in ((31, xo), (32, sm), (33, fold)):
        assert np.array_equal(gpu.op_sweep(kind, a, b).view(u32), want)
