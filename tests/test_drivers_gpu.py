"""The drop-in C++ drivers built on include/gpusort.hpp run and print what the
reference's drivers print (lsb/sort.cu:68-72,148-151; msb/src/test.cu:53-56)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "gpu-sort_amd", "drivers")


def _run(args):
    exe = os.path.join(DRV, args[0])
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (run __graft_entry__.build())")
    return subprocess.run([exe] + args[1:], capture_output=True, text=True, timeout=120, check=True).stdout.splitlines()


def test_lsb_driver_output():
    out = _run(["lsb_sort", "--n=300000", "--t=2"])
    assert len(out) == 4
    for first32, js in ((out[0], out[1]), (out[2], out[3])):
        vals = [float(x) for x in first32.split()]
        assert len(vals) == 32 and all(a >= b for a, b in zip(vals, vals[1:]))      # keys-only sort is DESCENDING
        assert 0.0 < vals[-1] <= vals[0] <= 1.0                                       # uniform (0,1] float keys
        rec = json.loads(js)
        assert set(rec) == {"time_sort_kv_gpu", "time_sort_k_gpu"} and all(v > 0 for v in rec.values())


def test_msb_driver_output():
    out = _run(["msb_test", "20"])
    assert out[0].startswith("Time Sort K: ") and out[1].startswith("Time Sort KV: ")
    assert out[2].strip() == "Adjacent inversions in result: 0"


def test_msb_harness_entropy_sweep():
    """msb/tests counterpart: 12 entropy levels x repeats, gtest-style verdicts, profile table."""
    out = _run(["msb_harness", "-r", "1", "-k", "150000", "-p", "80000", "--gtest_filter=Entropy_UINT"])
    text = "\n".join(out)
    assert "[       OK ] Sort_Keys.Entropy_UINT" in text and "FAILED" not in text
    assert sum("SORTKEYS.ENTROPIES" in l for l in out) >= 12
    rows = [l for l in out if l.startswith("sort_keys_UINT\t")]
    assert len(rows) == 12 and all(len(r.split("\t")) == 8 for r in rows)
    out = _run(["msb_harness", "-r", "1", "-p", "80000", "--gtest_filter=Sort_Pairs.UINT_UINT"])
    text = "\n".join(out)
    assert "[       OK ] Sort_Pairs.UINT_UINT" in text and "FAILED" not in text


def test_msb_harness_64bit_cases():
    """The UINT64 / DOUBLE keys and the 64-bit value combinations of msb/tests (test_sort_keys.cu:154-195,
    test_sort_pairs.cu:223-281) run through rdxsrt_unstable_sort's wide path."""
    out = _run(["msb_harness", "-r", "1", "-k", "120000", "-p", "70000", "--gtest_filter=64"])
    text = "\n".join(out)
    for name in ("Sort_Keys.Entropy_UINT64", "Sort_Pairs.UINT_UINT64", "Sort_Pairs.UINT64_UINT", "Sort_Pairs.UINT64_UINT64"):
        assert f"[       OK ] {name}" in text
    assert "FAILED" not in text and "SKIPPED" not in text
    out = _run(["msb_harness", "-r", "1", "-k", "120000", "--gtest_filter=DOUBLE"])
    assert "[       OK ] Sort_Keys.Entropy_DOUBLE" in "\n".join(out)
    rows = [l for l in out if l.startswith("sort_keys_DOUBLE\t")]
    assert len(rows) == 12


def test_lsb_types_driver_all_type_pairs():
    """Typed DeviceRadixSort driver (32- and 64-bit keys and values) against std::stable_sort."""
    out = _run(["lsb_types", "200003"])
    assert out[-1] == "ALL CORRECT" and not any("FAIL" in line for line in out)
    assert sum(line.endswith(": CORRECT") for line in out) > 7 * 9


def test_shim_too_small_data_manager_yields_null():
    """A pre-allocated RDXSRT_GPUDataManager sized for fewer keys: {nullptr, nullptr}, not the unsorted input."""
    out = _run(["shim_errors", "gpu"])
    assert out[-1] == "OK"


def test_cpp_sharded_host_on_a_one_rank_rccl_communicator():
    """gs_msb_sort_u32_sharded (C++ host, RCCL called directly: first pass -> ncclAllGather of the bucket sizes ->
    one grouped ncclSend/ncclRecv exchange -> finish) on the one GPU of the box: a one-rank communicator, keys and pairs,
    every rank's slice verified (sorted, global multiset unchanged)."""
    for extra in ([], ["--pairs"]):
        out = _run(["msb_sharded", "--log2n", "24", "--reps", "2"] + extra)
        rec = json.loads(out[-1])
        assert rec["verified"] is True and rec["n_gpus"] == 1 and rec["rank0_received"] == 1 << 24
