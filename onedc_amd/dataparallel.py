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


# ---- host cores per rank -------------------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-63,128-191' (sysfs cpulist format) -> sorted list of CPU numbers."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def plan_core_sets(world, allowed, gpu_nodes=None, node_cpus=None, min_cores=2):
    """Which host cores each rank's threads (rANS pool, pipeline workers, torch) may run on: a pure function of the topology.

    allowed    CPUs this process may use (os.sched_getaffinity)
    gpu_nodes  NUMA node of rank r's GPU (from /sys/bus/pci/devices/<BDF>/numa_node), or None / -1 where unknown
    node_cpus  {node: [cpus]} (from /sys/devices/system/node/node*/cpulist)
    The ranks whose GPUs sit on one NUMA node share that node's allowed CPUs in equal contiguous slices, in rank order
    (so a rank's threads stay on the socket its GPU hangs off); a rank with an unknown node, or whose node offers it fewer
    than `min_cores`, gets its slice of the CPUs no node group claimed -- or, with no topology at all, the r-th contiguous
    slice of `allowed`.  Returns a list of `world` CPU lists, pairwise disjoint and never empty, or None when `allowed` cannot
    give every rank `min_cores`."""
    allowed = sorted(set(allowed))
    if world < 1 or len(allowed) // world < min_cores:
        return None
    plan = [None] * world
    if gpu_nodes and node_cpus:
        groups = {}
        for r in range(world):
            node = gpu_nodes[r] if r < len(gpu_nodes) else None
            if node is not None and node >= 0 and node in node_cpus:
                groups.setdefault(node, []).append(r)
        allowed_set = set(allowed)
        for node, ranks in groups.items():
            cpus = [c for c in sorted(node_cpus[node]) if c in allowed_set]
            per = len(cpus) // len(ranks)
            if per < min_cores:
                continue
            for i, r in enumerate(ranks):
                plan[r] = cpus[i * per:(i + 1) * per]
    rest_ranks = [r for r in range(world) if plan[r] is None]
    if rest_ranks:
        taken = {c for p in plan if p for c in p}
        free = [c for c in allowed if c not in taken]
        per = len(free) // len(rest_ranks)
        if per < min_cores:                          # the topology-aware split starves somebody: plain contiguous slices
            per = len(allowed) // world
            return [allowed[r * per:(r + 1) * per] for r in range(world)]
        for i, r in enumerate(rest_ranks):
            plan[r] = free[i * per:(i + 1) * per]
    return plan


def read_topology(world):
    """(gpu_nodes, node_cpus) of this machine from sysfs; (None, None) where it does not say (containers, CPU-only boxes).
    The PCI address of HIP device r comes from torch's device properties (no kernel launch, no context needed beyond the
    runtime's own initialisation)."""
    import glob
    import os
    node_cpus = {}
    for path in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"):
        try:
            node_cpus[int(os.path.basename(os.path.dirname(path))[4:])] = parse_cpulist(open(path).read())
        except (OSError, ValueError):
            pass
    gpu_nodes = []
    for r in range(world):
        node = None
        try:
            p = torch.cuda.get_device_properties(r)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        except Exception:                              # noqa: BLE001 -- unknown stays unknown
            node = None
        gpu_nodes.append(node)
    if not node_cpus or all(n is None or n < 0 for n in gpu_nodes):
        return None, None
    return gpu_nodes, node_cpus
