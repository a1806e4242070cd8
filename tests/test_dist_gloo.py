"""The N > 1 path on CPU: two processes, gloo backend.  Clips are sharded with zen_amd.dist.shard_units,
each rank separates its own clips (the CPU oracle stands in for the GPU engine here -- this is a test of
the sharding / aggregation plumbing, not of the kernels), and the reduced counters and checksums must
equal a single-process run over all clips."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zen_amd import dist as zdist  # noqa: E402


def clip(i, n):
    return np.random.default_rng(100 + i).uniform(-1, 1, n).astype(np.float32)


def separate(ids, n):
    from oracle import oracle as o
    eng = o.HPRIOffline(44100.0, 1024, 256, 2.0, 2.0)
    hops = len(ids) * (o.chunk_padder(n, 1024, 3)[0] + o.chunk_padder(n, 256, 11)[0])
    chk = 0.0
    for i in ids:
        h, p, _ = eng.process(clip(i, n))
        chk += float(np.abs(p).astype(np.float64).sum() + np.abs(h).astype(np.float64).sum())
    return hops, chk


def worker(rank, world, port, n_clips, n, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    g = zdist.Group("gloo")
    ids = zdist.shard_units(n_clips, g.world, g.rank)
    g.barrier()
    hops, chk = separate(ids, n)
    t = g.max(1.0 + rank)                      # max over ranks
    tot = g.sum([hops, chk, len(ids)])
    if g.rank == 0:
        out.put((t, tot))
    g.close()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_offline_matches_single_process():
    n_clips, n, world = 5, 6000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, free_port_holder[0], n_clips, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, tot = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hops, chk = separate(list(range(n_clips)), n)
    assert t == 2.0                              # max(1.0, 2.0)
    assert tot[0] == hops and tot[2] == n_clips
    assert abs(tot[1] - chk) <= 1e-9 * abs(chk)


free_port_holder = [free_port()]


def test_shard_units_partitions():
    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in zdist.shard_units(512, world, r))
        assert got == list(range(512))
        sizes = [len(zdist.shard_units(512, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert zdist.shard_units(512, 8, 3)[:3] == [3, 11, 19] and len(zdist.shard_units(512, 8, 3)) == 64
    lengths = [100, 1, 1, 1, 50, 50, 3, 3]
    parts = [zdist.shard_units(8, 2, r, lengths) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == list(range(8))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 10
    with pytest.raises(ValueError):
        zdist.shard_units(4, 2, 2)


def test_time_shards_cover_the_clip():
    for n, world, align in ((26460000, 8, 4096), (161571, 3, 1024), (1000, 4, 256), (5, 2, 1)):
        sh = zdist.time_shards(n, world, align)
        assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == n
        for (b0, e0), (b1, e1) in zip(sh, sh[1:]):
            assert e0 == b1 and b0 <= e0
        assert all(b % align == 0 for b, _ in sh)


def _fake_sysfs(root, gpus):
    """gpus: [(kfd node id, render minor, numa node, cpulist)]; node 0 is a CPU-only KFD node, as on a real box."""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    put(os.path.join(root, "class/kfd/kfd/topology/nodes/0/properties"), "cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n")
    for node, minor, numa, cpus in gpus:
        put(os.path.join(root, "class/kfd/kfd/topology/nodes/%d/properties" % node),
            "cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % minor)
        put(os.path.join(root, "class/drm/renderD%d/device/numa_node" % minor), "%d\n" % numa)
        put(os.path.join(root, "class/drm/renderD%d/device/local_cpulist" % minor), cpus + "\n")


def test_gpu_numa_cpus_from_a_sysfs_tree(tmp_path, monkeypatch):
    """The ranks of `bench.py --gpus N` (and the children of `zen batch --gpus N`) pin themselves to the CPUs next to their
    GPU before their first GPU call; the topology comes from sysfs alone (KFD nodes -> render minor -> local_cpulist)."""
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "ZEN_NO_NUMA_BIND"):
        monkeypatch.delenv(v, raising=False)
    root = str(tmp_path)
    _fake_sysfs(root, [(2, 128, 0, "0-3,64-67"), (3, 129, 1, "4-7"), (10, 136, -1, "0-127")])
    assert zdist.gpu_numa_cpus(0, root) == [0, 1, 2, 3, 64, 65, 66, 67]
    assert zdist.gpu_numa_cpus(1, root) == [4, 5, 6, 7]
    assert zdist.gpu_numa_cpus(2, root) is None          # numa_node -1: the box does not say
    assert zdist.gpu_numa_cpus(3, root) is None          # no such GPU
    assert zdist.gpu_numa_cpus(0, str(tmp_path / "nothing")) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")     # remapped: device 0 is the second GPU
    assert zdist.gpu_numa_cpus(0, root) == [4, 5, 6, 7]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # binding: only CPUs this process may use anyway; restored afterwards
    before = os.sched_getaffinity(0)
    try:
        first = sorted(before)[0]
        _fake_sysfs(root, [(2, 128, 0, "%d" % first)])
        assert zdist.bind_to_gpu_numa(0, root) == [first] and os.sched_getaffinity(0) == {first}
        monkeypatch.setenv("ZEN_NO_NUMA_BIND", "1")
        assert zdist.bind_to_gpu_numa(0, root) is None
    finally:
        os.sched_setaffinity(0, before)
