"""N > 1 path on CPU: two gloo ranks shard a frame list with no data-path collective; the only
communication is the timing barrier and the max-over-ranks reduce that bench.py uses."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from arap_flow_amd import shard
rank, world, local = shard.dist_env()
dist.init_process_group("gloo", rank=rank, world_size=world)
lines = ["rgb%%03d.png msk%%03d.png c%%03d.txt f%%03d.flo o%%03d.png m%%03d.png" %% ((i,) * 6) for i in range(13)]
mine = shard.shard_lines(lines, rank, world)
dist.barrier()
t = shard.max_over_ranks(0.25 * (rank + 1), dist)
gathered = [None] * world
dist.all_gather_object(gathered, mine)          # test-only: proves the shards partition the list
if rank == 0:
    print(json.dumps({"t": t, "shards": gathered}))
dist.barrier()
dist.destroy_process_group()
''' % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_partition_the_frame_list(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["t"] == 0.5                                           # max over ranks
    a, b = res["shards"]
    assert len(a) == 7 and len(b) == 6
    assert sorted(a + b) == sorted("rgb%03d.png msk%03d.png c%03d.txt f%03d.flo o%03d.png m%03d.png" % ((i,) * 6)
                                   for i in range(13))
    assert not set(a) & set(b)


def test_shard_indices_edge_cases():
    from arap_flow_amd import shard
    assert shard.shard_indices(0, 0, 4) == []
    assert shard.shard_indices(3, 3, 8) == []
    assert shard.shard_indices(512, 5, 8) == list(range(5, 512, 8)) and len(shard.shard_indices(512, 5, 8)) == 64
    import pytest
    with pytest.raises(ValueError):
        shard.shard_indices(4, 4, 4)


def test_bench_gpus_flag_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment starts two fresh ranks itself (before it touches torch or
    HIP) and relays rank 0's line.  ARAP_BENCH_STUB=1 replaces the solve by a sleep, so this runs without a GPU; what is
    under test is the launcher, the rendezvous on 127.0.0.1, the barrier-bracketed timing and the max over ranks."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ARAP_BENCH_STUB="1", ARAP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["gpus_flag"] == 2 and d["steps"] == 3
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the job time is the slowest rank's
    assert d["ms_per_step"] >= 19.0
    assert abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_refuses_a_world_size_that_differs_from_gpus(tmp_path):
    """a launcher environment with another world size than --gpus would print a line whose n_gpus is not what was asked
    for: refused with exit code 2 before anything is imported"""
    env = dict(os.environ, ARAP_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=60)          # default --gpus 1 under a 2-rank launcher
    assert r.returncode == 2
