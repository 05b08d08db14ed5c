"""The N > 1 path on CPU: two processes, gloo, 127.0.0.1.  Checks the weight broadcast (rank 1 ends up
with bit-identical tensors of the right shapes), the static sharding of batches (a partition), the
MAX reduction of step times, and that two ranks entropy-decoding their own shards with the host coder
reproduce what a single process decodes (the data path needs no collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from onedc_amd import arch, synth
from onedc_amd.dataparallel import broadcast_state_dicts, max_over_ranks, parse_cpulist, plan_core_sets, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = arch.codec_decode_spec()
        sds = [synth.codec_state_dict(0)] if rank == 0 else [None]
        got = broadcast_state_dicts(sds, [spec], torch.device("cpu"), rank, world)[0]
        ref = synth.codec_state_dict(0)
        assert set(got) == set(spec)
        for k, shape in spec.items():
            assert tuple(got[k].shape) == tuple(shape), k
            assert torch.equal(got[k], ref[k].float()), k
        # sharding: every batch exactly once
        mine = shard(7, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        assert sorted(sum(gathered, [])) == list(range(7))
        # slowest rank decides
        t = max_over_ranks(1.0 + rank, torch.device("cpu"), world)
        assert t == float(world)
        # data path without collectives: each rank decodes its own streams with the host coder
        from onedc_amd.entropy import rans
        from onedc_amd.entropy.entropy_models import EntropyCoder, GaussianEncoder
        ge = GaussianEncoder()
        ge.update(force=True, entropy_coder=EntropyCoder(False, 1))
        tables = ge.get_cdf_info()
        rng = np.random.default_rng(123)
        streams, syms, idxs = [], [], []
        for b in range(6):
            idx = rng.integers(0, 256, 4096).astype(np.int16)
            sym = np.clip(np.rint(rng.normal(0, 3, 4096)), -60, 60).astype(np.int16)
            e = rans.RansEncoder(False, 1)
            e.add_cdf(*tables)
            e.encode_with_indexes(sym, idx, 0)
            e.flush()
            streams.append(e.get_encoded_stream().tobytes())
            syms.append(sym)
            idxs.append(idx)
        for b in shard(6, rank, world):
            d = rans.RansDecoder(1)
            d.add_cdf(*tables)
            d.set_stream(np.frombuffer(streams[b], dtype=np.uint8))
            assert np.array_equal(d.decode_stream(idxs[b], 0), syms[b])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_shard_and_decode(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_is_a_partition():
    for n in (0, 1, 5, 16, 129):
        for world in (1, 2, 4, 8):
            parts = [shard(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def _check_plan(plan, world, allowed):
    assert plan is not None and len(plan) == world
    seen = set()
    for cores in plan:
        assert len(cores) >= 2 and set(cores) <= set(allowed)
        assert not (seen & set(cores))                      # pairwise disjoint
        seen |= set(cores)


def test_host_core_plan_follows_the_numa_topology():
    """WORLD_SIZE = 8 on a faked 2-socket node (128 cores, SMT siblings numbered +128; GPUs 0-3 on node 0, 4-7 on node 1):
    every rank gets a non-empty core set on ITS GPU's node, all sets disjoint; degraded inputs fall back sanely."""
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    node_cpus = {0: parse_cpulist("0-63,128-191"), 1: parse_cpulist("64-127,192-255")}
    allowed = list(range(256))
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    plan = plan_core_sets(8, allowed, gpu_nodes, node_cpus)
    _check_plan(plan, 8, allowed)
    for r in range(8):
        assert set(plan[r]) <= set(node_cpus[gpu_nodes[r]]) and len(plan[r]) == 32
    # GPUs enumerated in an interleaved order: contiguous slices would put ranks on the wrong socket, the plan does not
    inter = [0, 1, 0, 1, 0, 1, 0, 1]
    plan = plan_core_sets(8, allowed, inter, node_cpus)
    _check_plan(plan, 8, allowed)
    assert all(set(plan[r]) <= set(node_cpus[inter[r]]) for r in range(8))
    # a cgroup that only allows part of the machine
    sub = list(range(16, 48)) + list(range(64, 96))
    plan = plan_core_sets(8, sub, gpu_nodes, node_cpus)
    _check_plan(plan, 8, sub)
    assert all(set(plan[r]) <= set(node_cpus[gpu_nodes[r]]) for r in range(8))
    # no topology (containers): contiguous slices
    plan = plan_core_sets(8, allowed, None, None)
    _check_plan(plan, 8, allowed)
    assert plan[3] == list(range(96, 128))
    # one GPU with an unknown node (numa_node = -1) shares what the node groups left over
    plan = plan_core_sets(4, list(range(64)), [0, 0, -1, 1], {0: list(range(32)), 1: list(range(32, 64))})
    _check_plan(plan, 4, list(range(64)))
    # all allowed cores on one node although the GPUs sit on two: nobody may end up empty
    plan = plan_core_sets(8, list(range(64)), gpu_nodes, node_cpus)
    _check_plan(plan, 8, list(range(64)))
    # too few cores for the ranks: no pinning at all
    assert plan_core_sets(8, list(range(8)), gpu_nodes, node_cpus) is None


def _bench(*argv, env=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], cwd=root, env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_bare_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher environment (the shape of the driver's single-GPU command with N = 2): the
    parent starts two rank processes itself, the ranks meet in a process group (gloo here: --dry-run-ranks is the CPU rehearsal of
    the protocol, no GPU and no model), run the contract's bracket, reduce MAX, and stdout carries exactly ONE JSON line."""
    import json
    r = _bench("--gpus", "2", "--dry-run-ranks", "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_counted_by_all_reduce"] == 2 and line["launcher"] == "self" and line["dry_run"]
    assert line["steps"] == 5 and line["scaling"] == "weak"
    # the stand-in step of rank r takes 2 (r + 1) ms: the job's step time is the SLOWEST rank's (rank 1: >= 4 ms)
    assert line["ms_per_step"] >= 4.0
    assert line["value"] == pytest.approx(16 * 2 * 5 / (line["ms_per_step"] * 5e-3), rel=1e-3)
    assert len(line["repeatability"]["values"]) == 3 and line["value"] == line["repeatability"]["values"][0]


def test_bare_bench_launch_fails_when_a_rank_fails():
    """A rank that dies costs the launch its exit code (the other rank is left waiting in the rendezvous: the parent ends it), and
    no JSON line is printed."""
    r = _bench("--gpus", "2", "--dry-run-ranks", "--steps", "2", "--warmup", "1", env={"ONEDC_DRY_RUN_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_dry_run_under_an_external_launcher_environment():
    """The driver's N > 1 launch line sets RANK / WORLD_SIZE itself: bench.py must then NOT start ranks of its own."""
    import json
    port = _free_port()
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-ranks", "--steps", "3",
                                       "--warmup", "1"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    json_lines = [[l for l in o[0].splitlines() if l.startswith("{")] for o in outs]
    assert len(json_lines[0]) == 1 and len(json_lines[1]) == 0                 # rank 0 prints, rank 1 does not
    line = json.loads(json_lines[0][0])
    assert line["n_gpus"] == 2 and line["launcher"] == "external" and line["ranks_counted_by_all_reduce"] == 2


@pytest.mark.gpu
def test_bare_bench_gpus_1_on_the_gpu():
    """The driver's BENCH command shape (`python bench.py --gpus 1 ...`, no launcher): one process, no process group, one line with
    the round-6 objects -- the repeatability of the headline, the median-of-passes roofline and its consistency flag, the CPU
    model.  Runs as a child process before this process has touched the GPU."""
    import json
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: a child launch from here is not allowed on the pool")
    r = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-side-lines", "--cpu-sample", "256",
               env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["config"]["launcher"] == "none"
    rep = line["repeatability"]
    assert rep["regions"] == 3 and rep["values"][0] == line["value"] and rep["min"] <= line["value_median_of_3"] <= rep["max"]
    rf = line["roofline"]
    assert rf["achieved_min"] <= rf["achieved"] <= rf["achieved_max"] and len(rf["achieved_passes"]) == 3
    assert rf["isolated_inconsistent"] is False, rf["isolated_check"]
    assert line["cpu_baseline"]["cpu_model"] and line["cpu_baseline"]["cores"] >= 1
    assert set(line["device"]["stages"]) >= {"codec_synthesis", "unet_step", "vae_decoder"}
    assert not torch.cuda.is_initialized()


@pytest.mark.gpu
def test_bench_under_torchrun_with_rccl_broadcast_on_one_gpu(tmp_path):
    """The launch line the driver uses for N > 1, with N = 1 (the only size a 1-GPU box allows):
    `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 --broadcast-weights` -- process-group
    initialisation on the nccl (= RCCL) backend, the flat weight broadcast on the DEVICE buffer, the barrier-bracketed
    timed region and the MAX reduction all run on hardware.  Started as a fresh child process, and only while this
    process has not touched the GPU (this file sorts in front of the tests/test_gpu_* files for that reason)."""
    import json
    import subprocess
    import sys
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: a child launch from here is not allowed on the pool")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--broadcast-weights", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["weights"].startswith("rank 0 builds") and line["config"]["images_per_gpu_per_step"] == 4
    assert line["step_ms"]["n"] >= 1 and line["handoff"]["value"] > 0 and line["device"]["stage_s_ms"] > 0
    assert line["host_rans"]["decode_ns_per_symbol"] > 0
    assert line["config"]["rccl_ranks"] == 1 and line["config"]["launcher"].startswith("external")
    assert not torch.cuda.is_initialized()
