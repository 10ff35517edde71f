"""The N > 1 path on CPUs: world_size-2 gloo runs of the rank plumbing bench.py uses,
and the block-distribution rules.  (The GPU data path has no collective to test.)"""
import os
import subprocess
import sys
import textwrap

from gcn10_amd import shard
from tests.conftest import ROOT


def test_round_robin_share_matches_reference_loop():
    ids = [2234, 2261, 2256, 2257, 2290, 2262, 2264]          # src/test/blocks.txt head
    # for (i = rank; i < n_blocks; i += size), src/main.c:171
    assert shard.blocks_for_rank(ids, 0, 3) == [2234, 2257, 2264]
    assert shard.blocks_for_rank(ids, 1, 3) == [2261, 2290]
    assert shard.blocks_for_rank(ids, 2, 3) == [2256, 2262]
    assert shard.blocks_for_rank(ids, 0, 1) == ids
    assert shard.blocks_for_rank(ids, 7, 8) == []
    got = sorted(b for r in range(4) for b in shard.blocks_for_rank(ids, r, 4))
    assert got == sorted(ids)


def test_world_size_one_needs_no_torch_distributed(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    g = shard.Group()
    assert (g.rank, g.world) == (0, 1)
    g.barrier()
    assert g.max(3.5) == 3.5 and g.sum(2.0) == 2.0
    g.close()


def test_gloo_world_size_two(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from gcn10_amd import shard
        g = shard.Group(backend="gloo")
        g.barrier()
        elapsed = 1.0 + g.rank            # rank 1 is the slow one
        worst = g.max(elapsed)
        total = g.sum(10.0 * (g.rank + 1))
        mine = shard.blocks_for_rank(list(range(10, 21)), g.rank, g.world)
        g.barrier()
        with open(os.path.join(%r, "rank%%d.json" %% g.rank), "w") as f:     # one file per rank:
            json.dump({"rank": g.rank, "world": g.world, "worst": worst,      # stdout of two ranks interleaves
                       "total": total, "mine": mine}, f)
        g.close()
    """ % (ROOT, str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:           # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert all(r["worst"] == 2.0 and r["total"] == 30.0 for r in recs)       # max / sum over ranks
    assert recs[0]["mine"] == [10, 12, 14, 16, 18, 20] and recs[1]["mine"] == [11, 13, 15, 17, 19]


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
