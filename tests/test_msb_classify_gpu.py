"""M4 pinned: the classification of every MSB level -- which sub-buckets go on to the next level,
which ranges become local-sort tasks, how tiny neighbours are merged, which size class a task lands in, and which
buckets the heavy-hitter path takes -- read back from the device and compared with the CPU restatement in oracle/
(oracle.msb_level_lists; reference rules: msb/src/sort/cuda_radix_sort.h:1084-1087,1241-1247,
cuda_radix_sort_config.h:9).  A wrong threshold or a wrong class still sorts correctly; only this test sees it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(O, n):
    rng = np.random.default_rng(7)
    few = rng.integers(0, 2**32, size=40, dtype=np.uint64).astype(np.uint32)
    return {
        "uniform": O.gen_uniform(n, seed=1),
        "zipf": O.gen_zipf(n, seed=2),
        "entropy_and_3": O.gen_entropy_and(n, 3, seed=3),
        "hot_top_byte": (O.gen_uniform(n, seed=4) & np.uint32(0x00ffffff)) | np.uint32(0x5a000000),
        "few_values_plus_noise": np.where(rng.random(n) < 0.9, few[rng.integers(0, 40, size=n)], O.gen_uniform(n, seed=5)).astype(np.uint32),
    }


@pytest.mark.parametrize("pivot", [False, True])
@pytest.mark.parametrize("name", ["uniform", "zipf", "entropy_and_3", "hot_top_byte", "few_values_plus_noise"])
def test_classification_matches_the_oracle(gs, oracle, cuda, name, pivot):
    from gpu_sort_amd.msb import msb_classify_upto
    n = (1 << 23) + 12345
    keys = _inputs(oracle, n)[name]
    for stop in (0, 1, 2):
        exp_b, exp_t = oracle.msb_level_lists(keys, stop, pivot=pivot)
        a = torch.from_numpy(keys.view(np.int32).copy()).to(cuda)
        b = torch.empty_like(a)
        got_b, got_t, _ = msb_classify_upto(a, b, n, stop, pivot=pivot)
        assert got_b == exp_b, (name, stop, len(got_b), len(exp_b))
        for c in range(4):
            assert got_t[c] == exp_t[c], (name, stop, c, len(got_t[c]), len(exp_t[c]), sorted(got_t[c] ^ exp_t[c])[:4])
        if not exp_b:
            break


def test_census_accounts_for_every_key(gs, oracle, cuda):
    """gs_msb_census: keys per level, keys in heavy-hitter buckets and keys handed to local sorts add up (every key is
    finished by exactly one local sort, inside a heavy-hitter bucket, or by the last level's scatter)."""
    from gpu_sort_amd.msb import msb_census, msb_algorithmic_bytes
    n = (1 << 24) + 777
    for gen in (oracle.gen_uniform, oracle.gen_zipf):
        keys = gen(n, seed=11)
        a = torch.from_numpy(keys.view(np.int32).copy()).to(cuda)
        b = torch.empty_like(a)
        nb = gs.lib.gs_msb_temp_bytes(n, 0)
        dm = torch.empty(nb, dtype=torch.uint8, device=cuda)
        seq = gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=dm)
        assert np.array_equal(seq.sorted_keys.cpu().numpy().view(np.uint32), np.sort(keys))
        cen = msb_census(dm, n)
        assert cen[0]["keys"] == n and cen[0]["buckets"] == 1
        finished_by_tasks = sum(c["task_keys"] for c in cen)
        in_pivot_middle = 0        # heavy-hitter buckets: their strangers are in task_keys, the rest stays in place
        last_level = cen[3]["keys"]
        # keys that go on from level L = keys of level L + 1
        for L in range(3):
            passed_on = cen[L + 1]["keys"]
            assert passed_on <= cen[L]["keys"]
        exp_b, exp_t = oracle.msb_level_lists(keys, 0)
        assert cen[1]["keys"] == sum(s for _, s in exp_b) and cen[0]["task_keys"] == sum(s for c in exp_t.values() for _, s, _ in c)
        by = msb_algorithmic_bytes(cen, n)
        assert by["lsb_downsweep"] == 8 * n and by["msb_histogram"] == 4 * sum(c["keys"] for c in cen[1:])
        assert finished_by_tasks + last_level <= n + 0 and finished_by_tasks + last_level + sum(c["pivot_keys"] for c in cen) >= n
