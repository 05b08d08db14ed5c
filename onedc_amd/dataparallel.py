"""Image-batch data parallelism over the GPUs of one node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Pictures are independent units, so the data path has NO collective: rank r decodes the batches
`shard(n_batches, r, world)`.  The only collectives are one flat broadcast of the weights from rank 0 at
start-up (reference: every process would read the checkpoint itself, inference.py:87-93) and one 8-byte
MAX reduction of the elapsed time when benchmarking.
"""
import numpy as np
import torch


def shard(n_items, rank, world):
    """Static round-robin assignment of work items (batches, pictures) to ranks: item i -> rank i % world."""
    return list(range(rank, n_items, world))


def broadcast_state_dicts(sds, specs, device, rank, world, dist=None):
    """Rank 0 owns the weights (`sds`: list of state dicts, ignored elsewhere); `specs` (list of
    {key: shape}) is known to every rank.  ONE flat fp32 broadcast, then views into the flat buffer.
    The collective is issued whenever a process group exists, also for a group of one (a single-GPU run of
    `bench.py --broadcast-weights` exercises RCCL initialisation and the broadcast on the device buffer)."""
    if dist is None:
        import torch.distributed as dist
    total = sum(int(np.prod(s)) for spec in specs for s in spec.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == 0:
        off = 0
        for sd, spec in zip(sds, specs):
            for k, shape in spec.items():
                n = int(np.prod(shape))
                flat[off:off + n] = sd[k].reshape(-1).to(device=device, dtype=torch.float32)
                off += n
    if world > 1 or dist.is_initialized():
        dist.broadcast(flat, src=0)
    out, off = [], 0
    for spec in specs:
        sd = {}
        for k, shape in spec.items():
            n = int(np.prod(shape))
            sd[k] = flat[off:off + n].view(*shape)
            off += n
        out.append(sd)
    return out


def max_over_ranks(value, device, world, dist=None):
    """The job's step time is the slowest rank's."""
    if dist is None:
        import torch.distributed as dist
    if world <= 1 and not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
