"""CPU tests: the C-ABI library loads here (no GPU) and exports every symbol
include/gpusort.h declares; host-side sizing logic; no compute calls."""
import ctypes as C
import os

import pytest
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpusort.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(gs):
    from gpu_sort_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 14
    raw = C.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"libgpusort.so does not export {s}"
        assert s in _lib.SIGNATURES, f"python binding lacks {s}"
    assert set(_lib.SIGNATURES) == set(syms)


def test_library_is_in_tree(gs):
    assert os.path.realpath(gs.LIB_PATH).startswith(os.path.realpath(ROOT))
    assert gs.lib.gs_version() == 100


def test_temp_bytes_and_geometry(gs):
    lib = gs.lib
    g, t, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.gs_lsb_geometry(1 << 30, 0, C.byref(g), C.byref(t), C.byref(c))
    assert t.value == 8192 and g.value * c.value * t.value >= (1 << 30) > (g.value - 1) * c.value * t.value
    need = lib.gs_lsb_temp_bytes(1 << 30, 0)
    assert need >= 256 * g.value * 4 + 1024 + ((1 << 30) // t.value) * 512 and need % 256 == 0
    assert need < (1 << 30) * 4 // 50          # workspace stays far below the key bytes
    lib.gs_lsb_geometry(100, 0, C.byref(g), C.byref(t), C.byref(c))
    assert g.value == 1 and c.value == 8
    assert lib.gs_lsb_temp_bytes(0, 0) > 0


def test_argument_validation_without_gpu(gs):
    # these return hipErrorInvalidValue (1) before touching the device
    lib = gs.lib
    sel = C.c_int(0)
    keys = (C.c_void_p * 2)(16, 32)
    assert lib.gs_lsb_sort_u32(None, 0, keys, None, C.byref(sel), 10, 0, 32, 0, 0, None) == 1   # no workspace
    assert lib.gs_lsb_sort_u32(None, 0, keys, None, C.byref(sel), 10, 5, 4, 0, 0, None) == 1    # bad bits
    assert lib.gs_lsb_sort_u32(None, 0, keys, None, C.byref(sel), 1 << 32, 0, 32, 0, 0, None) == 1
    assert lib.gs_lsb_sort_u32(None, 0, keys, None, C.byref(sel), 0, 0, 32, 0, 0, None) == 0    # n == 0: no-op
    assert lib.gs_lsb_sort_u32(None, 0, keys, None, C.byref(sel), 10, 7, 7, 0, 0, None) == 0    # 0 bits: no-op
    assert sel.value == 0
    assert b"invalid" in lib.gs_error_string(1).lower()


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib.util, sys
    src = os.path.join(ROOT, "gpu-sort_amd", "_lib.py")
    fake = tmp_path / "pkg"
    fake.mkdir()
    (fake / "_lib.py").write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("fake_lib", str(fake / "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("import must fail when libgpusort.so is absent")


def test_package_import_brings_torch_in_before_the_library():
    """The HIP runtime the library binds to is decided when it is loaded: torch's bundled runtime must already be in
    the global symbol scope (see gpu-sort_amd/_lib.py).  A fresh interpreter that imports only the package must
    therefore find torch loaded before libgpusort.so."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "orig = ctypes.CDLL.__init__\n"
        "def spy(self, name, *a, **k):\n"
        "    if name and 'libgpusort' in str(name):\n"
        "        t = sys.modules.get('torch')\n"
        "        print('torch fully imported at load:', t is not None and hasattr(t, 'cuda') and hasattr(t, 'Tensor'))\n"
        "    orig(self, name, *a, **k)\n"
        "ctypes.CDLL.__init__ = spy\n"
        "import gpu_sort_amd\n"
        "print('order ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "order ok" in out.stdout, out.stderr[-2000:]
    assert "torch fully imported at load: True" in out.stdout, out.stdout


def test_shim_reports_failures_instead_of_returning_unsorted_data():
    """rdxsrt_unstable_sort's shim must not return the input pointers when the scratch allocation (or the
    sort) failed.  No GPU here, so every allocation fails: the result must be {nullptr, nullptr} and stderr must say why."""
    import subprocess
    exe = os.path.join(ROOT, "gpu-sort_amd", "drivers", "shim_errors")
    if not os.path.exists(exe):
        pytest.skip("drivers not built (run __graft_entry__.build())")
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-device path needs a box without a GPU")
    out = subprocess.run([exe, "nodevice"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    assert out.stderr.count("gpusort: rdxsrt_unstable_sort") == 5          # three shim calls + the two host-pointer wrappers
    assert "rdxsrt_unstable_sort_keys: device allocation / copy in" in out.stderr
    assert "rdxsrt_unstable_sort_pairs: device allocation / copy in" in out.stderr


def test_rccl_host_library_exports_and_split_rule(gs):
    """libgpusort_rccl.so (the C++ host of the bucket-sharded sort, include/gpusort_rccl.h) loads here, exports what its
    header declares, and its bucket -> rank rule agrees with the Python host's (gpu-sort_amd/sharded.py::compute_splits)
    on random and on lopsided bucket sizes -- both sides of an exchange must cut at the same places."""
    import numpy as np
    from gpu_sort_amd import sharded
    path = os.path.join(ROOT, "gpu-sort_amd", "lib", "libgpusort_rccl.so")
    if not os.path.exists(path):
        pytest.skip("libgpusort_rccl.so not built (run __graft_entry__.build())")
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gpusort_rccl.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))
    assert syms == ["gs_msb_sharded_temp_bytes", "gs_msb_sort_u32_sharded", "gs_sharded_compute_splits", "gs_sharded_exchange_plan",
                    "gs_sharded_selftest"]
    lib = C.CDLL(path)
    for name in syms:
        assert hasattr(lib, name)
    lib.gs_msb_sharded_temp_bytes.restype = C.c_size_t
    lib.gs_msb_sharded_temp_bytes.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    assert lib.gs_msb_sharded_temp_bytes(1 << 30, (1 << 30) + (1 << 28), 0, 8) >= gs.lib.gs_msb_finish_temp_bytes((1 << 30) + (1 << 28), 0, 8)
    lib.gs_sharded_compute_splits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        for kind in ("uniform", "lopsided", "empty", "one_bucket"):
            h = rng.integers(0, 1 << 22, size=(world, 256)).astype(np.uint64)
            if kind == "lopsided":
                h[:, 17] *= 300
            if kind == "empty":
                h[:] = 0
            if kind == "one_bucket":
                h[:] = 0
                h[:, 200] = 12345
            dest = np.zeros(256, np.uint8)
            per = np.zeros(world, np.uint64)
            lib.gs_sharded_compute_splits(h.ctypes.data_as(C.c_void_p), world, dest.ctypes.data_as(C.c_void_p), per.ctypes.data_as(C.c_void_p))
            ed, ep = sharded.compute_splits(h, world)
            assert np.array_equal(dest, ed) and np.array_equal(per.astype(np.int64), ep), (world, kind)
    # adversarial sizes: keys_before * world / n lands exactly ON an integer, or one key short of it, with totals beyond
    # 2^53 / world (where a float64 quotient rounds the wrong way): both hosts use exact integers and must agree
    def both(h, world):
        dest = np.zeros(256, np.uint8)
        per = np.zeros(world, np.uint64)
        lib.gs_sharded_compute_splits(h.ctypes.data_as(C.c_void_p), world, dest.ctypes.data_as(C.c_void_p), per.ctypes.data_as(C.c_void_p))
        ed, ep = sharded.compute_splits(h, world)
        assert np.array_equal(dest, ed) and np.array_equal(per.astype(np.int64), ep), world
        return dest
    for world in (2, 3, 5, 7, 8):
        for scale in (1, (1 << 40) + 1, (1 << 54) // (256 * world)):
            for delta in (0, 1, -1):
                h = np.zeros((world, 256), np.uint64)
                h[0, :] = world * scale                 # every bucket boundary sits on an exact multiple of n / (256 * world) ...
                if delta == 1:
                    h[0, 0] += 1                        # ... or one key past it
                if delta == -1:
                    h[0, 0] -= 1                        # ... or one key short of it
                    h[0, 255] += 1
                d = both(h, world)
                assert np.all(np.diff(d.astype(np.int64)) >= 0) and d[0] == 0 and d[-1] <= world - 1
    # exact integer quotient (the float64 form this replaced gives 1 here): before * world / n = 1 - 2^-55
    h = np.zeros((2, 256), np.uint64)
    h[0, 0] = (1 << 54) - 1
    h[0, 1] = 1
    h[1, 2] = (1 << 54)
    d = both(h, 2)
    assert d[0] == 0 and d[1] == 0 and d[2] == 1                  # bucket 1 starts at (2^54 - 1) * 2 / 2^55 < 1 -> rank 0
    # the exchange plan, a pure function: for every pair, what the sender cuts for the receiver is what the receiver
    # expects from the sender, the pieces tile the receive buffer in source order, every rank derives the same rounds
    lib.gs_sharded_exchange_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for world in range(2, 9):
        h = rng.integers(0, 1 << 21, size=(world, 256)).astype(np.uint64)
        h[world // 2, 40:60] *= 50
        dest = both(h, world)
        plans = []
        for r in range(world):
            so, ro = np.zeros(world + 1, np.uint64), np.zeros(world + 1, np.uint64)
            pieces, rounds = np.zeros((world, 256), np.uint64), C.c_uint64(0)
            lib.gs_sharded_exchange_plan(h.ctypes.data_as(C.c_void_p), dest.ctypes.data_as(C.c_void_p), r, world, so.ctypes.data_as(C.c_void_p),
                                         ro.ctypes.data_as(C.c_void_p), pieces.ctypes.data_as(C.c_void_p), C.byref(rounds))
            plans.append((so, ro, pieces, rounds.value))
        assert len({pl[3] for pl in plans}) == 1 and plans[0][3] >= 1
        for a in range(world):
            so, ro, pieces, _ = plans[a]
            assert so[0] == 0 and so[world] == h[a].sum() and np.all(np.diff(so.astype(np.int64)) >= 0)
            es, er = sharded.exchange_plan(h, dest, a, world)             # the Python host's plan: same cuts
            assert np.array_equal(np.diff(so.astype(np.int64)), es) and np.array_equal(np.diff(ro.astype(np.int64)), er)
            for b in range(world):
                sent = int(so[b + 1] - so[b])                                  # a -> b
                assert sent == int(plans[b][1][a + 1] - plans[b][1][a])        # = what b expects from a
                assert sent == int(plans[b][2][a].sum()) == int(h[a][dest == b].sum())
            # my pieces: source-major, buckets in order inside a source, only buckets I own
            assert np.array_equal(pieces, np.where(dest[None, :] == a, h, 0))
            assert int(ro[world]) == int(pieces.sum())
    # argument validation happens before any GPU or RCCL call
    lib.gs_msb_sort_u32_sharded.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 2 + [C.c_uint64] + [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    n_out = C.c_uint64(0)
    assert lib.gs_msb_sort_u32_sharded(None, 0, None, None, 10, None, None, None, None, None, None, 10, C.byref(n_out), None, 0, 1, 0, None) == 1
