"""Multi-GPU plumbing for the paths that shard.

Only independent work units shard: whole clips of the batched offline HPR-I, or independent realtime
streams (replicas).  There is no data-path collective -- each rank owns its units end to end; the
process group (RCCL on GPUs, gloo in the CPU tests) carries a barrier and a few scalars: the max-over-ranks
wall time, unit counts and an output checksum so that rank 0 can print the whole-job figure.
"""
import os
import socket
import subprocess
import sys


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_torchrun():
    """True when RANK / WORLD_SIZE are already in the environment (torch.distributed.run, or spawn_ranks)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _parse_cpulist(text):
    """'0-31,64-95' -> [0..31, 64..95] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(index, sysfs="/sys"):
    """CPUs local to GPU `index` (the order HIP enumerates: the GPU nodes of the KFD topology, filtered by ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES when those hold plain indices), from sysfs alone -- no HIP call, so it can run before anything touches
    a GPU: node N of /sys/class/kfd/kfd/topology/nodes is a GPU when its simd_count > 0; its drm_render_minor M names
    /sys/class/drm/renderD<M>/device, whose local_cpulist is the answer.  None when the box does not say (no KFD, numa_node -1,
    an unreadable file): the caller then leaves the affinity alone."""
    try:
        root = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
        gpus = []
        for name in sorted(os.listdir(root), key=lambda v: int(v) if v.isdigit() else 1 << 30):
            props = {}
            with open(os.path.join(root, name, "properties")) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props.get("drm_render_minor", "-1")))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis:
                ids = [v.strip() for v in vis.split(",") if v.strip()]
                if not all(v.isdigit() for v in ids):
                    return None                        # UUIDs: not resolved here
                gpus = [gpus[int(v)] for v in ids if int(v) < len(gpus)]
        if not (0 <= index < len(gpus)) or gpus[index] < 0:
            return None
        dev = os.path.join(sysfs, "class", "drm", "renderD%d" % gpus[index], "device")
        with open(os.path.join(dev, "numa_node")) as f:
            if int(f.read().strip()) < 0:
                return None
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError, IndexError):
        return None


def bind_to_gpu_numa(index, sysfs="/sys"):
    """Pin this process (and the threads it will start: the runtime's, the host-vector pipeline's) to the CPUs next to GPU
    `index`, BEFORE its first GPU call.  The offline path moves 12 bytes per sample over the host link from / to pageable host
    memory (zen/offline.h:141-147): a rank whose buffers sit on the other socket pays the inter-socket link for every one of
    them.  Returns the CPU list, or None (nothing changed) when the topology is unknown or ZEN_NO_NUMA_BIND is set."""
    if os.environ.get("ZEN_NO_NUMA_BIND"):
        return None
    cpus = gpu_numa_cpus(index, sysfs)
    if not cpus:
        return None
    try:
        allowed = os.sched_getaffinity(0)
        want = set(cpus) & allowed
        if not want:
            return None
        os.sched_setaffinity(0, want)
        return sorted(want)
    except (AttributeError, OSError):
        return None


def spawn_ranks(argv, n, timeout=None, env_extra=None):
    """Run `argv` as n fresh child processes, one per GPU of this node (RANK = LOCAL_RANK = 0..n-1,
    WORLD_SIZE = n, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT), the same environment torch.distributed.run
    would give them.  The caller must not have touched the GPU: nothing here imports torch or loads the HIP
    library, and the children are new interpreters (no fork of a GPU context, no exec from one).

    Rank 0's stdout is relayed to ours (the one JSON line of bench.py); every rank's stderr goes to ours.
    Returns 0 when every rank exited 0, else the first non-zero exit code (the remaining ranks are
    terminated so that a dead rank cannot leave the others waiting in a collective forever)."""
    if n < 1:
        raise ValueError("spawn_ranks: n must be >= 1")
    import tempfile
    import time
    port = _free_port()
    procs = []
    rc = 0
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
            if env_extra:
                env.update(env_extra)
            # (the child binds itself to its GPU's NUMA node first thing -- bench.py main(), zen_amd.dist.bind_to_gpu_numa -- as a
            # rank started by torch.distributed.run has to)
            procs.append(subprocess.Popen(argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = None if timeout is None else time.monotonic() + timeout
        try:
            while True:
                codes = [p.poll() for p in procs]
                bad = [c for c in codes if c not in (None, 0)]
                if bad:
                    rc = bad[0]
                    break
                if all(c == 0 for c in codes):
                    break
                if deadline is not None and time.monotonic() > deadline:
                    rc = 124
                    break
                time.sleep(0.05)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        out0.seek(0)
        data = out0.read()
    if data:
        sys.stdout.write(data.decode(errors="replace"))
        sys.stdout.flush()
    return rc


def shard_units(n_units, world, rank, lengths=None):
    """Unit ids owned by `rank`.

    Equal-length units: round robin (rank r gets ids r, r+world, ...), so every rank gets
    floor/ceil(n/world) units.  With `lengths`, longest-first greedy assignment to the least-loaded rank
    (deterministic: ties go to the lower rank), which keeps the per-rank sample counts balanced."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank %d/%d" % (world, rank))
    if lengths is None:
        return list(range(rank, n_units, world))
    if len(lengths) != n_units:
        raise ValueError("lengths must have n_units entries")
    order = sorted(range(n_units), key=lambda i: (-lengths[i], i))
    load = [0] * world
    owner = [0] * n_units
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += lengths[i]
    return [i for i in range(n_units) if owner[i] == rank]


def time_shards(n_samples, world, align=1):
    """Split [0, n_samples) into `world` contiguous output ranges (one long clip over several
    GPUs).  Boundaries are multiples of `align` (e.g. the larger hop); empty tails are possible for tiny
    clips.  Each rank then calls zen_hip_hpri_process_range on its range: no exchange between ranks, the
    halo of input each shard needs is reported by zen_hip_hpri_range_halo."""
    if world < 1:
        raise ValueError("world must be >= 1")
    per = -(-n_samples // world)
    per = -(-per // align) * align
    out = []
    for r in range(world):
        b = min(n_samples, r * per)
        e = min(n_samples, (r + 1) * per)
        out.append((b, e))
    return out


class Group:
    """Thin wrapper over torch.distributed (or nothing, for world == 1)."""

    def __init__(self, backend=None, device=None, force=False):
        """force: initialise the process group for a world of one too (RCCL communicator, all-reduce, barrier, destroy on
        ONE GPU: the plumbing of the 8-GPU run, exercised where only one GPU exists -- tests/test_gpu_round4.py and
        ZEN_FORCE_PROCESS_GROUP=1 python bench.py)."""
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.device = device
        if self.world > 1 or force:
            import torch
            import torch.distributed as dist
            self.torch = torch
            if not launched_by_torchrun():            # a plain process: give torch.distributed the rendezvous of a world of one
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def _tensor(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64)
        return t.to(self.device) if self.device is not None else t

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        t = self._tensor([float(value)])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, values):
        values = [float(v) for v in values]
        if self.dist is None:
            return values
        t = self._tensor(values)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
