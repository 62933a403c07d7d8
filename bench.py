#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s sorting 2^30 uint32 keys on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N = 1 : BASELINE.json configs[1] -- LSB radix sort, 2^30 uniform-random u32 keys, keys-only.
  N > 1 : configs[4] shape -- MSB bucket-sharded sort, 2^30 keys per GPU: first digit pass, ONE RCCL
          all-to-all of the top-level buckets, rest of the MSB sort on the receiver
          (launched by torch.distributed.run, one rank per GPU).
A "step" is one complete sort of one batch of synthetic keys already resident in HBM.  Every
step sorts its own pre-generated input buffer (seed = step index), so no restore copy sits in
the timed region (the reference restores inputs outside its timed call, lsb/sort.cu:141-146).

Also reported on the same line:
  verified     -- the LAST timed step's output checked on the device outside the timed region (sortedness + the
                  multiset checksum of its input, taken before the sort)
  also         -- N = 1, default workload only: BASELINE.json configs[2] (LSB pairs), configs[3] (MSB Zipf keys), MSB
                  uniform keys and the one-rank rehearsal of configs[4]'s pipeline, 5 device-timed sorts each with a
                  device-side check, after the headline's buffers are freed (the reference's drivers print both of their
                  sorts in one run too, lsb/sort.cu:148-151, msb/src/test.cu:53-56).  `--no-also` skips it.
  roofline     -- the dominant kernel (lsb_downsweep): algorithmic bytes per launch
                  (8 B/key x keys per launch) / its average launch duration,
                  measured live with hipEvents on the launch stream during the timed steps.
  cpu_baseline -- the reference's CPU check sort (std::sort) timed on this host on a bounded
                  sample of the same workload (oracle/cpu_baseline.cpp; checker only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
LSB_BYTES_PER_KEY = {False: 48, True: 80}       # (4 + 8) x 4 passes; pairs (4 + 16) x 4
DOWNSWEEP_BYTES_PER_KEY = {False: 8, True: 16}  # read + write keys (+ values) per launch


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=30, help="keys per GPU = 2^log2n (headline: 30)")
    ap.add_argument("--pairs", action="store_true", help="configs[2]: key + value pairs")
    ap.add_argument("--algo", choices=["lsb", "msb"], default=None)
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations reported under `also`")
    ap.add_argument("--also-steps", type=int, default=5)
    ap.add_argument("--also-late-alloc", action="store_true", help="allocate the `also` arrays after the headline run instead of first (placement experiments)")
    ap.add_argument("--cpu-sample-log2", type=int, default=27)
    ap.add_argument("--verify", action="store_true", help="device-side sortedness + checksum on every step")
    ap.add_argument("--exchange-groups", type=int, default=0,
                    help="N>1: collectives the all-to-all is cut into (group g is finished while g+1.. are in flight); "
                         "1 = exchange, then finish; 0 (default) = probe 1, 2 and 4 during the warm-up (two untimed sorts "
                         "each, max over ranks) and run the timed steps with the faster one")
    ap.add_argument("--one-rank-rccl", action="store_true",
                    help="with --force-sharded: a ONE-rank RCCL group, and the N>1 code path (size all_gather, grouped "
                         "asynchronous all_to_all, one finish per group) run on it -- what a one-GPU box can show of it")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the bucket-sharded pipeline even on one rank (exercises the N>1 code path)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal where every rank uses cuda:0 and the exchange goes over gloo (not a measurement)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launch plumbing only (CPU, gloo): spawn the ranks, rendezvous, one all_reduce, rank 0 prints a "
                         "JSON line with n_gpus -- what tests/test_bench_launch.py runs where there is no GPU")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves as CHILD processes
    (python -m torch.distributed.run, one rank per GPU), before anything in this process has touched the GPU, pass
    their output through and exit with their code.  Rank 0 prints the single JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:          # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def load_pmc_traffic(pairs=False):
    """HBM bytes per downsweep launch from the committed rocprofv3 --pmc summary, or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_pairs.json" if pairs else "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def load_pmc_traffic_msb(dist):
    """{kernel group: HBM bytes per sort} of the MSB path from the committed --pmc summary (profiles/pmc_traffic_msb.json), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_msb.json")) as f:
            d = json.load(f)
        return d.get(dist, {}).get("hbm_bytes_per_sort", {}) if d.get("log2n") == 30 else {}
    except Exception:
        return {}


def cpu_baseline(n_log2, pairs):
    from oracle import oracle as O
    n = 1 << n_log2
    keys = O.gen_uniform(n, seed=0)
    if pairs:
        dt, _, _ = O.time_std_stable_sort_pairs(keys, O.gen_enumerated(n))
        what = "std::stable_sort of (key,value) structs"
    else:
        dt, _ = O.time_std_sort(keys)
        what = "std::sort"
    out = {"value": n / dt / 1e9, "unit": "Gkeys/s", "cores": 1, "kind": "port",
           "sample": f"first 2^{n_log2} keys of the step-0 input (uniform splitmix64, seed 0), {what}, 1 thread, {dt:.2f} s",
           "host_threads_available": O.hardware_threads()}
    # BASELINE.json configs[0]: the reference's CPU path at 2^20 keys, median of 5
    k20 = O.gen_uniform(1 << 20, seed=0)
    t20 = sorted(O.time_std_sort(k20)[0] for _ in range(5))
    out["config1_2p20"] = {"value": (1 << 20) / t20[2] / 1e9, "unit": "Gkeys/s", "cores": 1, "ms_median_of_5": round(t20[2] * 1e3, 3),
                           "ms_min": round(t20[0] * 1e3, 3), "sample": "2^20 uniform u32 keys, std::sort, 1 thread"}
    if not pairs:
        dt_mt, used, _ = O.time_std_sort_mt(keys, 0)
        out["all_cores"] = {"value": n / dt_mt / 1e9, "unit": "Gkeys/s", "cores": used,
                            "sample": f"same 2^{n_log2} keys, {used} threads std::sort + inplace_merge tree, {dt_mt:.2f} s"}
    return out


def box_facts(torch, dev):
    """What can differ between two boxes: clocks, partition modes, memory, allocator."""
    out = {}
    try:
        p = torch.cuda.get_device_properties(dev)
        out.update(name=p.name, gcn_arch=getattr(p, "gcnArchName", None), cus=p.multi_processor_count,
                   total_mem_GiB=round(p.total_memory / 2**30, 1), clock_mhz=getattr(p, "clock_rate", 0) // 1000 or None,
                   memory_clock_mhz=getattr(p, "memory_clock_rate", 0) // 1000 or None, l2_MiB=round(getattr(p, "L2_cache_size", 0) / 2**20, 1))
        free, total = torch.cuda.mem_get_info(dev)
        out["free_mem_GiB"] = round(free / 2**30, 1)
    except Exception as e:          # never let a diagnostic break the bench line
        out["error"] = repr(e)
    for key, path in (("compute_partition", "current_compute_partition"), ("memory_partition", "current_memory_partition")):
        try:
            import glob
            vals = sorted({open(f).read().strip() for f in glob.glob(f"/sys/class/drm/card*/device/{path}")})
            out[key] = vals[0] if len(vals) == 1 else vals
        except Exception:
            out[key] = None
    out["allocator"] = "torch caching allocator (hipMalloc-backed, 2 MiB-aligned blocks for these sizes)"
    return out


def also_configs(gs, torch, dev, n, steps, log2n, bufs=None):
    """BASELINE.json configs[2..4] next to the headline (N = 1): each workload sorted `steps` times from a restored
    input, every sort bracketed by an event pair on its stream (median reported), the last result checked on the
    device.  Returns {workload: {...}}."""
    from gpu_sort_amd.msb import msb_census, msb_algorithmic_bytes
    out = {}

    def median_ms(fn, restore):
        evs = []
        for _ in range(steps + 1):                       # first one untimed (warm-up)
            restore()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); res = fn(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in evs[1:])
        return ms[len(ms) // 2], ms[0], res

    if bufs is None:
        bufs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(5)]
    src, work, alt, vals, vals_alt = bufs

    # ---- configs[2]: LSB, (u32 key, u32 value) pairs
    gs.generate_uniform_keys(n, seed=101, device=dev, out=src)
    nbytes = gs.lib.gs_lsb_temp_bytes(n, 1)
    temp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    state = {}

    def restore_pairs():
        work.copy_(src)
        gs.generate_enumerated_values(n, device=dev, out=vals)

    def lsb_pairs():
        dk, dv = gs.DoubleBuffer(work, alt), gs.DoubleBuffer(vals, vals_alt)
        gs.DeviceRadixSort.SortPairs(temp, nbytes, dk, dv, n, key_type=gs.GS_KEY_U32)
        return dk.Current(), dv.Current()

    prof = gs.KernelProfile()
    with prof:
        med, mn, (rk, rv) = median_ms(lsb_pairs, restore_pairs)
    k = prof.read()
    inv = gs.check_sorted(rk)[0]
    bad = gs.check_pairs_enumerated(src, rk, rv)[0]
    ds_ms = k["lsb_downsweep"][0] / k["lsb_downsweep"][1]
    gbs = 16 * n / (ds_ms * 1e-3) / 1e9
    out[f"lsb_radix_sort_2^{log2n}_u32_uniform_pairs"] = {
        "baseline_config": 2, "ms_device_median": round(med, 4), "ms_device_min": round(mn, 4), "rate": round(n / med / 1e6, 3), "unit": "Gpairs/s",
        "whole_sort_frac_of_peak": round(LSB_BYTES_PER_KEY[True] * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": {"kernel": "lsb_downsweep", "algorithmic_bytes": 16 * n, "avg_launch_ms": round(ds_ms, 4), "achieved": round(gbs, 1),
                     "frac": round(gbs / HBM_PEAK_GBS, 4)},
        "verified": bool(inv == 0 and bad == 0), "check": "sorted on the device + every value still points at its key (gs_check_pairs_enumerated_u32)"}
    del vals, vals_alt, temp, bufs

    # ---- MSB: configs[3] (Zipf keys) and uniform keys
    nbytes = gs.lib.gs_msb_temp_bytes(n, 0)
    temp = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    for dist_name, gen, cfg in (("zipf", gs.generate_zipf_keys, 3), ("uniform", gs.generate_uniform_keys, None)):
        gen(n, seed=102, device=dev, out=src)
        pre = gs.check_sorted(src)[1:]

        def restore():
            work.copy_(src)

        def msb():
            return gs.rdxsrt_unstable_sort(work, None, n, alt, None, pre_allocated_dm=temp, synchronize=False).sorted_keys

        prof = gs.KernelProfile()
        with prof:
            med, mn, res = median_ms(msb, restore)
        k = prof.read()
        invs, sm, xr = gs.check_sorted(res)
        census = msb_census(temp, n, False)
        by = msb_algorithmic_bytes(census, n, False)
        per_sort = {g: k[g][0] / (steps + 1) for g in by if g in k}
        dom = max(per_sort, key=per_sort.get)
        gbs = by[dom] / (per_sort[dom] * 1e-3) / 1e9
        tot = sum(by.values())
        out[f"msb_radix_sort_2^{log2n}_u32_{dist_name}_keys_only"] = {
            **({"baseline_config": cfg} if cfg else {}),
            "ms_device_median": round(med, 4), "ms_device_min": round(mn, 4), "rate": round(n / med / 1e6, 3), "unit": "Gkeys/s",
            "algorithmic_bytes_per_key": round(tot / n, 2), "whole_sort_frac_of_peak": round(tot / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {"kernel": dom, "algorithmic_bytes": by[dom], "ms_per_sort": round(per_sort[dom], 4), "achieved": round(gbs, 1),
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": (load_pmc_traffic_msb(dist_name).get(dom) if log2n == 30 else None),
                         "all_kernel_groups": {g: {"algorithmic_bytes": by[g], "ms_per_sort": round(per_sort[g], 4),
                                                   "frac": round(by[g] / (per_sort[g] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for g in per_sort}},
            "verified": bool(invs == 0 and (sm, xr) == pre), "check": "sorted on the device + multiset checksum of the input"}
    del temp, alt

    # ---- configs[4]'s pipeline on ONE rank (first digit pass -> [exchange: nothing to move] -> finish on the pieces):
    # the like-for-like N = 1 point of the `--gpus N` curve
    from gpu_sort_amd import sharded
    gs.generate_uniform_keys(n, seed=103, device=dev, out=src)
    runner = sharded.ShardedSorter(n, False, dev, pipeline="msb")
    pre = runner.input_checksum(src)
    state["res"] = None

    def restore():
        work.copy_(src)

    def shard():
        state["res"] = runner.sort(work, None)
        return state["res"]

    med, mn, (sk, _, cnt) = median_ms(shard, restore)
    ok = runner.verify(sk, cnt, pre)[0]
    runner.stage_times = {}
    work.copy_(src)
    runner.sort(work, None)
    st = runner.stage_times
    runner.stage_times = None
    out[f"msb_sharded_one_rank_2^{log2n}_u32_uniform_keys_only"] = {
        "baseline_config": 4, "note": "configs[4]'s pipeline (first pass + finish on received pieces, host-side split included) on one rank, no exchange",
        "ms_device_median": round(med, 4), "ms_device_min": round(mn, 4), "rate": round(n / med / 1e6, 3), "unit": "Gkeys/s",
        "stages_ms_serialised": {kk: round(float(v), 3) for kk, v in st.items()} if st else None,
        "verified": bool(ok), "check": "slice sorted + multiset checksum of the input"}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run_launch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "launch plumbing only", "dry_run": True, "n_gpus": world,
                              "rank_sum": float(t.item()), "steps": args.steps, "warmup": args.warmup}), flush=True)
        dist.destroy_process_group()
        return
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world == 1 and args.one_rank_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import gpu_sort_amd as gs

    n = 1 << args.log2n
    algo = args.algo or ("lsb" if world == 1 and not args.force_sharded else "msb")
    steps, warmup = args.steps, args.warmup
    total = steps + warmup

    gen = gs.generate_uniform_keys if args.dist == "uniform" else gs.generate_zipf_keys
    # the five arrays of the `also` workloads are allocated FIRST, the way a long-lived caller holds its sort buffers from start-up
    # on: what the pairs sort takes depends on where its arrays lie (profiles/README.md, round 3), and arrays allocated after the
    # headline's 23 x 4 GiB of inputs have been freed landed on the slow side in every run (4.0-4.2 ms per downsweep against 3.3)
    also_wanted = (world == 1 and not args.force_sharded and not args.one_rank_rccl and algo == "lsb" and not args.pairs
                   and args.dist == "uniform" and not args.no_also and args.also_steps > 0)
    also_bufs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(5)] if also_wanted and not args.also_late_alloc else None
    # one input buffer per step (4 GiB each at 2^30): nothing but the sort runs in the timed region
    inputs = [gen(n, seed=i, start=rank * n, device=dev) for i in range(total)]
    alt = torch.empty(n, dtype=torch.int32, device=dev)
    vals = vals_alt = None
    if args.pairs:
        vals = [gs.generate_enumerated_values(n, device=dev) for _ in range(total)]
        vals_alt = torch.empty(n, dtype=torch.int32, device=dev)

    sharded_path = world > 1 or args.force_sharded
    comm_info = None
    if dist.is_initialized():
        # what the COMMUNICATOR says (not argv), and a start-up check that it delivers a 1 GiB + 4 byte message whole
        from gpu_sort_amd import sharded as _sh
        comm_info = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank0_device": str(dev)}
        if not args.rehearse_on_one_gpu:
            comm_info["selftest_words_per_rank"] = _sh.communicator_selftest(dev)      # MAX_MSG words; raises on a truncated message
            try:        # informational: a message above the cap (the sorter never sends one)
                _sh.communicator_selftest(dev, elements=(1 << 28) + 1)
                comm_info["message_of_1GiB_plus_4B_whole"] = True
            except RuntimeError:
                comm_info["message_of_1GiB_plus_4B_whole"] = False
            torch.cuda.empty_cache()
    if sharded_path:
        from gpu_sort_amd import sharded
        # default: exchange after the first MSB digit pass; --algo lsb|msb: group-by-destination + full local sort
        runner = sharded.ShardedSorter(n, args.pairs, dev, local_algo=args.algo or "lsb",
                                       pipeline="partition" if args.algo else "msb", groups=max(1, args.exchange_groups),
                                       force_exchange=args.one_rank_rccl)
        nbytes = 0
        temp = None
    elif algo == "lsb":
        nbytes = gs.lib.gs_lsb_temp_bytes(n, int(args.pairs))
        temp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    else:
        nbytes = gs.lib.gs_msb_temp_bytes(n, int(args.pairs))
        temp = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)

    checks = []

    last = {}

    def one_step(i):
        if sharded_path:
            sk, sv, cnt = runner.sort(inputs[i], vals[i] if args.pairs else None)
            if args.verify:
                checks.append(runner.verify(sk, cnt, pre[i])[0])
            return
        if algo == "lsb":
            dk = gs.DoubleBuffer(inputs[i], alt)
            if args.pairs:
                dv = gs.DoubleBuffer(vals[i], vals_alt)
                gs.DeviceRadixSort.SortPairs(temp, nbytes, dk, dv, n, key_type=gs.GS_KEY_U32)
            else:
                gs.DeviceRadixSort.SortKeys(temp, nbytes, dk, n, key_type=gs.GS_KEY_U32)
            res = dk.Current()
        else:
            seq = gs.rdxsrt_unstable_sort(inputs[i], vals[i] if args.pairs else None, n, alt, vals_alt,
                                          pre_allocated_dm=temp, synchronize=False)
            res = seq.sorted_keys
        last["res"] = res
        if args.verify:
            checks.append(res)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last_pre = None
    if not sharded_path:
        last_pre = gs.check_sorted(inputs[total - 1])[1:]     # multiset checksum of the LAST timed step's input
    pre = None
    if args.verify and not sharded_path:
        pre = [gs.check_sorted(inputs[i])[1:] for i in range(total)]
    elif args.verify:
        pre = [runner.input_checksum(inputs[i]) for i in range(total)]

    for i in range(warmup):
        one_step(i)
    barrier()
    # N > 1 (or the one-rank RCCL group): whether finishing group g while groups g+1.. are in flight pays depends on
    # how the collective's kernels and ours share the GPU, which only the machine at hand can tell: probe both forms
    # in the warm-up (identical decision on every rank: max over ranks of the probe times) and keep the faster one
    probe = None
    if sharded_path and not args.algo and args.exchange_groups == 0 and (world > 1 or args.one_rank_rccl):
        probe = {}
        for g in (1, 2, 4):
            runner.groups = g
            runner.sort(inputs[0], vals[0] if args.pairs else None)       # settle (buffers, connections)
            barrier()
            tp = time.perf_counter()
            for _ in range(2):
                runner.sort(inputs[0], vals[0] if args.pairs else None)
            barrier()
            t = torch.tensor([(time.perf_counter() - tp) / 2 * 1e3], dtype=torch.float64,
                             device="cpu" if args.rehearse_on_one_gpu else dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            probe[g] = round(float(t.item()), 3)
        runner.groups = min(probe, key=probe.get)
        barrier()
    prof = gs.KernelProfile()
    # BASELINE.md sections 3-4 / SURVEY.md 8d: besides the wall clock over the K steps (the contract's `value`), every
    # step is bracketed by an event pair on the stream the sort is enqueued on -> median and min of the per-step device time
    step_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    with prof:
        for i in range(warmup, total):
            ea, eb = step_events[i - warmup]
            ea.record()
            one_step(i)
            eb.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = prof.read()
    step_ms = sorted(a.elapsed_time(b) for a, b in step_events)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: one more sharded sort with a host synchronisation after every stage and the
    # exchange as ONE collective, so that the exchange time is broken out (max over ranks)
    stages = None
    groups_used = runner.last.get("groups") if sharded_path and runner.last else None
    if sharded_path and not args.algo:
        runner.stage_times = {}
        runner.sort(inputs[-1], vals[-1] if args.pairs else None)
        st, runner.stage_times = runner.stage_times, None
        if st:
            names = ["first_pass_ms", "exchange_ms", "finish_ms"]
            t = torch.tensor([st[k] for k in names], dtype=torch.float64, device="cpu" if (args.rehearse_on_one_gpu or world == 1) else dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            stages = {k: round(float(v), 3) for k, v in zip(names, t.tolist())}

    verified = None
    if not sharded_path and not args.verify:
        # the last timed step's output, checked outside the timed region: sorted + same multiset as its input
        inv, sm, xr = gs.check_sorted(last["res"])
        verified = inv == 0 and (sm, xr) == last_pre
    if args.verify and not sharded_path:
        verified = True
        for j, res in enumerate(checks):
            inv, s, x = gs.check_sorted(res)
            verified = verified and inv == 0 and (s, x) == pre[j]
    elif args.verify:
        verified = all(checks)

    if comm_info is not None and sharded_path and runner.last:
        got = [None] * world
        if world > 1:
            dist.all_gather_object(got, int(runner.last["count"]))
        else:
            got = [int(runner.last["count"])]
        comm_info["received_keys_per_rank_last_sort"] = got
    if rank == 0:
        keys_total = n * world * steps
        value = keys_total / elapsed / 1e9
        ms_per_step = elapsed / steps * 1e3
        dom = "lsb_downsweep" if "lsb_downsweep" in kernels else ("msb_partition" if "msb_partition" in kernels else None)
        roofline = None
        whole = None
        census = None
        if algo == "msb" and not sharded_path:
            # the MSB path's bytes are data-dependent -> from the census of the last timed sort (what every
            # level partitioned / handed to local sorts); the roofline entry is the kernel group that took the most time
            from gpu_sort_amd.msb import msb_census, msb_algorithmic_bytes
            census = msb_census(temp, n, args.pairs)
            by = msb_algorithmic_bytes(census, n, args.pairs)
            per_sort = {k: kernels[k][0] / steps for k in by if k in kernels}
            dom = max(per_sort, key=per_sort.get)
            achieved = by[dom] / (per_sort[dom] * 1e-3) / 1e9
            msb_traffic = load_pmc_traffic_msb(args.dist) if (args.log2n == 30 and not args.pairs) else {}
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": msb_traffic.get(dom),
                        "algorithmic_bytes_per_sort": by[dom], "ms_per_sort": round(per_sort[dom], 4),
                        "launches_per_sort": kernels[dom][1] // steps,
                        "all_kernel_groups": {k: {"algorithmic_bytes": by[k], "ms_per_sort": round(per_sort[k], 4),
                                                  "GBps": round(by[k] / (per_sort[k] * 1e-3) / 1e9, 1)} for k in per_sort}}
            tot = sum(by.values())
            gbs = tot / (ms_per_step * 1e-3) / 1e9
            whole = {"algorithmic_bytes_per_key": round(tot / n, 2), "achieved_GBps": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                     "census": [{k: c[k] for k in ("buckets", "keys", "pivot_buckets", "pivot_keys", "task_keys", "tasks")} for c in census]}
        elif dom and dom in kernels:
            ms, cnt = kernels[dom]
            avg_ms = ms / cnt
            # keys per launch: the whole array on one GPU; a rank's received slice (~n) when sharded
            alg_bytes = DOWNSWEEP_BYTES_PER_KEY[args.pairs] * n
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            pmc = load_pmc_traffic(args.pairs)
            traffic = None
            if pmc and pmc.get("kernel") == dom and pmc.get("log2n") == args.log2n and bool(pmc.get("pairs")) == args.pairs:
                traffic = pmc.get("hbm_bytes_per_launch")
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4),
                        "launches": cnt}
        if algo == "lsb" and world == 1 and not sharded_path:
            gbs = LSB_BYTES_PER_KEY[args.pairs] * n / (ms_per_step * 1e-3) / 1e9
            whole = {"algorithmic_bytes_per_key": LSB_BYTES_PER_KEY[args.pairs], "achieved_GBps": round(gbs, 1),
                     "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                     "logical_GBps_cub_style": round(8 * n / (ms_per_step * 1e-3) / 1e9, 1)}
        also = None
        if (world == 1 and not sharded_path and algo == "lsb" and not args.pairs and args.dist == "uniform"
                and not args.no_also and args.also_steps > 0):
            inputs.clear()
            last.clear()
            checks.clear()
            del alt, temp
            torch.cuda.empty_cache()
            also = also_configs(gs, torch, dev, n, args.also_steps, args.log2n, also_bufs)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(min(args.cpu_sample_log2, args.log2n), args.pairs)
        line = {
            "metric": "Gkeys/s sorting 2^30 uint32 keys; achieved HBM GB/s vs roofline",
            "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 4),
            "step_ms_device": {"median": round(step_ms[len(step_ms) // 2], 4), "min": round(step_ms[0], 4),
                               "max": round(step_ms[-1], 4), "how": "hipEvent pair around every timed step on the sort's stream (rank 0)"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo)" if args.rehearse_on_one_gpu else "") + (" (one-rank RCCL group: the exchange is a device-local copy)" if args.one_rank_rccl else ""),
            "config": {"workload": (f"{algo}_radix_sort_2^{args.log2n}_u32_{args.dist}_"
                                    f"{'pairs' if args.pairs else 'keys_only'}" + ("_per_gpu_sharded" if sharded_path else "")),
                       "keys_per_gpu": n, "has_values": args.pairs,
                       **({"exchange_groups": groups_used} if sharded_path else {}),
                       **({"exchange_groups_probe_ms": probe} if probe else {}),
                       "algorithm": ((f"shard_partition+local_{args.algo}" if args.algo else "msb_first_pass+all_to_all+msb_finish")
                                     if sharded_path else algo),
                       "distribution": args.dist, "parallelism": "single" if not sharded_path else f"msb_bucket_shard{world}"},
            "stages_ms_serialised": stages, "rccl": comm_info,
            "roofline": roofline, "whole_sort": whole, "cpu_baseline": cpu, "also": also, "box": box_facts(torch, dev),
            "kernels_ms_total": {k: [round(v[0], 3), v[1]] for k, v in kernels.items()},
        }
        if verified is not None:
            line["verified"] = bool(verified)
        print(json.dumps(line), flush=True)
    if world > 1 or args.one_rank_rccl:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
