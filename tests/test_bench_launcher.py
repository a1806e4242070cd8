"""bench.py --gpus N must really start N ranks.

CPU tier: the launch / rendezvous / sharding / aggregation plumbing with the gloo backend and empty steps
(`--dry`); the GPU path differs only in the backend name ("nccl" = RCCL) and in what a step does."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(cmd, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    return r.returncode, lines, r.stderr.decode()


def test_plain_launch_spawns_one_rank_per_gpu():
    rc, lines, err = run([sys.executable, BENCH, "--gpus", "2", "--dry", "--backend", "gloo",
                          "--workload", "offline_batch", "--steps", "2"])
    assert rc == 0, err
    assert len(lines) == 1                       # only rank 0 prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_reported"] == 2 and j["dry"] is True
    assert j["config"]["parallelism"].startswith("clips sharded x2")
    assert j["config"]["units_all_ranks"] == 128 and j["config"]["units_rank0"] == 64


def test_plain_launch_time_shards_and_replicas():
    rc, lines, err = run([sys.executable, BENCH, "--gpus", "2", "--dry", "--backend", "gloo",
                          "--workload", "offline_long", "--steps", "1"])
    assert rc == 0, err
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "time-sharded x2" in j["config"]["parallelism"]
    rc, lines, err = run([sys.executable, BENCH, "--gpus", "3", "--dry", "--backend", "gloo", "--hops", "100",
                          "--steps", "1"])
    assert rc == 0, err
    j = json.loads(lines[0])
    assert j["n_gpus"] == 3 and j["ranks_reported"] == 3 and j["config"]["units_all_ranks"] == 300
    assert j["config"]["parallelism"] == "replicas x3"
    # the default N > 1 line also carries the path that shards: the offline batch, clips dealt to the ranks
    sh = j["offline_batch_sharded"]
    assert sh["ranks_reported"] == 3 and sh["clips_total"] == 3 * 64 and sh["clips_rank0"] == 64
    # ... and the host-clip leg (every rank HPRIOffline::process on its own host vectors at once): per-rank times gathered
    hs = j["offline_host_sharded"]
    assert hs["ranks_reported"] == 3 and hs["per_rank_wall_ms"] == [1.0, 2.0, 3.0] and hs["slowest_rank_wall_ms"] == 3.0


def test_torchrun_launch_is_not_respawned():
    """The driver's own launch line: the environment already names the ranks, bench.py must not spawn again."""
    rc, lines, err = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", BENCH, "--gpus", "2", "--dry",
                          "--backend", "gloo", "--steps", "1"])
    assert rc == 0, err
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_reported"] == 2


def test_single_gpu_line_has_no_launcher():
    rc, lines, err = run([sys.executable, BENCH, "--dry", "--steps", "1"])
    assert rc == 0, err
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks_reported"] == 1


def test_spawn_ranks_reports_a_dead_rank_and_stops_the_others():
    sys.path.insert(0, ROOT)
    from zen_amd import dist as zdist
    code = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
            "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
            "if r == 1: sys.exit(7)\n"
            "time.sleep(60)\n")
    import time
    t0 = time.monotonic()
    rc = zdist.spawn_ranks([sys.executable, "-c", code], 3, timeout=50)
    assert rc == 7
    assert time.monotonic() - t0 < 30            # ranks 0 and 2 were terminated, not waited for


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_shard_clips_on_the_gpu_path():
    """The real N-rank code path (engine per rank, clips sharded, counters reduced) on a 1-GPU box: two ranks
    share the device (ZEN_ALLOW_GPU_SHARING) and talk over gloo; rank 0 reports the aggregate line."""
    env = dict(os.environ, ZEN_ALLOW_GPU_SHARING="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--workload", "offline_batch", "--clip-seconds", "2", "--steps", "2", "--warmup", "1", "--settle-ms", "0",
              "--no-cpu-baseline"]

    def line(extra):
        r = subprocess.run([sys.executable, BENCH] + common + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])

    two = line(["--gpus", "2", "--backend", "gloo", "--clips", "2"])
    one = line(["--clips", "4"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"].startswith("clips sharded x2")
    assert two["config"]["clips_per_gpu"] == 2 and one["n_gpus"] == 1
    # rank r owns clips r, r+2 (round robin); the checksum is over the first 4096 samples of each rank's FIRST clip
    assert two["checksum"] > 0 and one["checksum"] > 0


@pytest.mark.gpu
def test_default_line_at_two_ranks_carries_the_sharded_offline_batch():
    """`bench.py --gpus N` (what the driver runs for the scaling curve): `value` is N replicas of the realtime stream,
    and the same line carries the sharded offline batch -- the path north_star scales -- with ranks_reported."""
    env = dict(os.environ, ZEN_ALLOW_GPU_SHARING="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--hops", "300", "--steps", "2", "--warmup", "1",
                        "--settle-ms", "0", "--leg-clips", "2", "--clip-seconds", "2", "--leg-steps", "2", "--host-shard-seconds", "20"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["ranks_reported"] == 2 and j["config"]["parallelism"] == "replicas x2"
    assert j["value"] > 0 and j["roofline"]["frac"] > 0
    # the sharded figure sits where the driver's record keeps it (inside `config`), and in the compact `legs` summary
    cfg, sh = j["config"], j["legs"]["offline_batch_sharded"]
    assert cfg["offline_batch_sharded_ranks_reported"] == 2 and cfg["offline_batch_sharded_clips_total"] == 4
    assert cfg["offline_batch_sharded_x_realtime"] > 0 and sh["hops_per_s"] > 0 and sh["x_realtime"] == cfg["offline_batch_sharded_x_realtime"]
    # the path a user runs: both ranks separate their own HOST clip at the same time (zen/offline.h:141-147's timed region)
    hs = j["legs"]["offline_host_sharded"]
    assert hs["ranks_reported"] == 2 and len(hs["per_rank_wall_ms"]) == 2 and min(hs["per_rank_wall_ms"]) > 0
    assert hs["slowest_rank_wall_ms"] >= max(hs["per_rank_wall_ms"]) - 1e-6
    assert cfg["offline_host_sharded_x_realtime"] == hs["value"] > 0 and cfg["offline_host_sharded_link_GBps"] > 0
    assert "cpu_baseline" not in j and "per_hop_api_us" not in cfg   # rank 0 at N = 1 only
