"""Probe of the collective library on this box: a ONE-rank RCCL group exchanging with itself, by message size.
Found with it (ROCm 7.2 RCCL under torch 2.10, MI355X): a self send/recv of 2 GiB or more silently delivers only the
first half of the message; 1 GiB is whole.  gpu-sort_amd/sharded.py therefore caps every message at MAX_MSG."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
for logn in (26, 27, 28, 29, 30):
    n = 1 << logn
    src = torch.arange(n, dtype=torch.int32, device=dev)
    for api in ("single", "list"):
        dst = torch.full((n,), -1, dtype=torch.int32, device=dev)
        if api == "single":
            dist.all_to_all_single(dst, src, [n], [n])
        else:
            dist.all_to_all([dst], [src])
        torch.cuda.synchronize()
        bad = (dst != src)
        nb = int(bad.sum())
        msg = ""
        if nb:
            idx = torch.nonzero(bad).flatten()
            msg = f" first {int(idx[0])} last {int(idx[-1])} values {dst[idx[:3]].tolist()}"
        print(f"2^{logn} int32 ({4 * n >> 20} MiB) {api}: mismatches {nb}{msg}", flush=True)
    del src, dst
dist.destroy_process_group()
