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
