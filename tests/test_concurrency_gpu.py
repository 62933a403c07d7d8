"""GPU test of the ABI's re-entrancy promise (no global mutable state: SURVEY.md 8b)."""
import pytest

pytestmark = pytest.mark.gpu


def test_entry_points_are_reentrant_across_threads_and_streams(gs, cuda):
    """"every entry re-entrant given distinct streams/workspaces".  Four host threads (ctypes drops
    the GIL inside the library) each run LSB, MSB and segmented sorts on their own stream, buffers and workspace,
    concurrently and repeatedly; every result is checked."""
    import threading
    import numpy as np
    import torch
    dev = cuda
    n, errors = 300007, []

    def worker(tid):
        progress = [None]
        try:
            torch.cuda.set_device(dev)               # a new host thread has no current device yet
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for it in range(12):
                    progress[0] = (it, 'generate')
                    keys = gs.generate_uniform_keys(n, seed=100 * tid + it, device=dev) if it % 2 else \
                        gs.generate_zipf_keys(n, seed=100 * tid + it, device=dev)
                    ref = torch.sort(keys.to(torch.int64) & 0xFFFFFFFF)[0]
                    a, b = keys.clone(), torch.empty_like(keys)
                    kind = (tid + it) % 3
                    progress[0] = (it, kind)
                    if kind == 0:
                        dk = gs.DoubleBuffer(a, b)
                        nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
                        temp = torch.empty(nb, dtype=torch.uint8, device=dev)
                        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
                        out = dk.Current()
                    elif kind == 1:
                        out = gs.rdxsrt_unstable_sort(a, None, n, b, None, synchronize=False).sorted_keys
                    else:
                        dk = gs.DoubleBuffer(a, b)
                        ob = torch.tensor([0], dtype=torch.int32, device=dev)
                        oe = torch.tensor([n], dtype=torch.int32, device=dev)
                        nb = gs.DeviceSegmentedRadixSort.SortKeys(None, 0, dk, n, 1, ob, oe)
                        temp = torch.empty(nb, dtype=torch.uint8, device=dev)
                        gs.DeviceSegmentedRadixSort.SortKeys(temp, nb, dk, n, 1, ob, oe, 0, 32, key_type=gs.GS_KEY_U32)
                        out = dk.Current()
                    if not torch.equal(out.to(torch.int64) & 0xFFFFFFFF, ref):
                        errors.append((tid, it, kind))
                stream.synchronize()
        except Exception as e:       # noqa: BLE001 -- reported below
            errors.append((tid, progress[0], repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
