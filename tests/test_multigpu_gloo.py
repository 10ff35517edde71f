"""The N > 1 path on CPUs: world-size-2 runs of the rank plumbing bench.py uses (files by default,
torch.distributed/gloo on request), bench.py's own launcher and aggregation with a stand-in engine,
and the block-distribution rules.  (The GPU data path has no collective to test.)"""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

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


def test_world_size_one_needs_nothing(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    g = shard.Group()
    assert (g.rank, g.world) == (0, 1)
    g.barrier()
    assert g.max(3.5) == 3.5 and g.sum(2.0) == 2.0 and g.all_gather({"a": 1}) == [{"a": 1}]
    g.close()
    assert "torch" not in sys.modules or True      # importing shard never imports torch (checked below)


def test_shard_and_bench_do_not_import_torch():
    code = "import sys; sys.path.insert(0, %r); import bench; from gcn10_amd import shard; " \
           "shard.Group(); print('torch' in sys.modules)" % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() == "False"


WORKER = """
    import json, os, sys
    sys.path.insert(0, %r)
    from gcn10_amd import shard
    g = shard.Group(%s)
    g.barrier()
    elapsed = 1.0 + g.rank            # rank 1 is the slow one
    worst = g.max(elapsed)
    total = g.sum(10.0 * (g.rank + 1))
    everyone = g.all_gather({"rank": g.rank, "pid": os.getpid()})
    mine = shard.blocks_for_rank(list(range(10, 21)), g.rank, g.world)
    g.barrier()
    with open(os.path.join(%r, "rank%%d.json" %% g.rank), "w") as f:     # one file per rank:
        json.dump({"rank": g.rank, "world": g.world, "worst": worst,      # stdout of two ranks interleaves
                   "total": total, "mine": mine, "backend": g.backend,
                   "ranks_seen": [e["rank"] for e in everyone], "torch": "torch" in sys.modules}, f)
    g.close()
"""


def _free_port():
    with socket.socket() as sk:           # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("backend", ["file", "gloo"])
def test_world_size_two_under_torchrun(tmp_path, backend):
    """The driver's launch line: torch.distributed.run starts the ranks; the group itself is files
    (default, no torch in the ranks) or gloo when asked for."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(WORKER % (ROOT, "" if backend == "file" else "backend='gloo'", str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("GCN10_RDV_DIR", None)
    env.pop("GCN10_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert all(r["worst"] == 2.0 and r["total"] == 30.0 for r in recs)       # max / sum over ranks
    assert all(r["ranks_seen"] == [0, 1] and r["backend"] == backend for r in recs)
    assert recs[0]["mine"] == [10, 12, 14, 16, 18, 20] and recs[1]["mine"] == [11, 13, 15, 17, 19]
    if backend == "file":
        assert not any(r["torch"] for r in recs)


def test_file_group_times_out_when_a_rank_is_missing(tmp_path):
    g = shard.FileGroup(0, 2, directory=str(tmp_path / "rdv"), timeout_s=0.3)
    with pytest.raises(TimeoutError):
        g.barrier()


def test_file_group_refuses_a_job_that_spans_nodes(monkeypatch):
    """the default (node-local) file rendezvous fails at once, instead of timing out after
    600 s, when torchrun says the job has more ranks than this node holds."""
    for k, v in (("RANK", "1"), ("LOCAL_RANK", "1"), ("WORLD_SIZE", "4"), ("LOCAL_WORLD_SIZE", "2")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("GCN10_RDV_DIR", raising=False)
    monkeypatch.delenv("GCN10_DIST_BACKEND", raising=False)
    with pytest.raises(RuntimeError, match="one node"):
        shard.Group()


def test_file_group_ignores_an_earlier_attempts_files(tmp_path, monkeypatch):
    """Workers restarted by the same elastic agent: the directory name carries the restart count, and a rank
    clears its own stale files when it joins a directory that already exists."""
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    monkeypatch.delenv("GCN10_RDV_DIR", raising=False)
    first = shard.default_rendezvous_dir()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert shard.default_rendezvous_dir() != first and "/" not in os.path.basename(first)
    d = tmp_path / "rdv"
    d.mkdir()
    (d / "s000000_r0.json").write_text("null")          # what a dead attempt of rank 0 left behind
    (d / "s000000_r1.json").write_text("null")
    g = shard.FileGroup(0, 2, directory=str(d), timeout_s=0.3)
    assert not (d / "s000000_r0.json").exists() and (d / "s000000_r1.json").exists()
    g._closed = True


FAKE = dict(GCN10_BENCH_ENGINE="tests.fake_engine:FakeEngine", GCN10_FAKE_DEVICES="2", GCN10_FAKE_LAUNCH_S="0.004")


def _bench(args, extra_env=None, launcher=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GCN10_RDV_DIR")}
    env.update(FAKE)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)


def test_bench_starts_its_own_ranks_and_aggregates():
    """`python bench.py --gpus 2` with no launcher: two rank processes, one JSON line, per-rank records,
    whole-job time = first start to last end (the stand-in engine makes rank 1 the slower one)."""
    # (10 ms per stand-in launch: a scheduling hiccup of a few milliseconds on a busy host must not decide
    # which rank looks slower)
    out = _bench(["--gpus", "2", "--steps", "5", "--warmup", "1", "--size", "2048", "--no-cpu-baseline"],
                 extra_env={"GCN10_FAKE_LAUNCH_S": "0.01"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["scaling"] == "weak"
    assert rec["data"].startswith("FAKE ENGINE") and rec["config"]["rank_sync"] == "file"
    pr = rec["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["device"] for r in pr] == [0, 1]
    assert pr[0]["pci_bus_id"] != pr[1]["pci_bus_id"]
    assert pr[1]["elapsed_s"] > pr[0]["elapsed_s"] * 1.15                     # device 1 sleeps 1.5x
    span = max(r["end_offset_ms"] for r in pr)
    assert abs(rec["ms_per_step"] * 5 - span) < 0.5
    assert rec["ms_per_step"] * 5e-3 >= max(r["elapsed_s"] for r in pr) - 1e-6
    want = 2048 * 2048 * 1 * 2 * 5 / (rec["ms_per_step"] * 5e-3) / 1e9
    assert abs(rec["value"] - want) / want < 1e-3
    assert rec["cpu_baseline"] is None and "also" not in rec                  # N = 1 only


def test_bench_same_line_under_the_torchrun_launcher():
    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "2048", "--no-cpu-baseline"],
                 launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())])
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and len(rec["per_rank"]) == 2 and rec["config"]["rank_sync"] == "file"


def test_bench_rehearsal_switch_and_refusals():
    # more ranks than devices: refused without --oversubscribe, shared round-robin with it
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--size", "2048", "--no-cpu-baseline"],
                 extra_env={"GCN10_FAKE_DEVICES": "1"})
    assert out.returncode != 0
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--size", "2048", "--no-cpu-baseline",
                  "--oversubscribe"], extra_env={"GCN10_FAKE_DEVICES": "1"})
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert [r["device"] for r in rec["per_rank"]] == [0, 0]
    # a launcher's world size that does not match --gpus is refused
    out = _bench(["--gpus", "2", "--no-cpu-baseline"], extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (out.stderr + out.stdout)


def test_bench_single_rank_with_the_stand_in_engine():
    out = _bench(["--steps", "3", "--warmup", "1", "--size", "2048", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and len(rec["per_rank"]) == 1
    assert rec["roofline"]["copy_ceiling"]["kernel"] == "stream_copy_kernel"
    assert rec["also"]["workload"].startswith("config4") and "round1_style_ms_per_launch" in rec["also"]


def test_bench_calibration_code_path_with_the_stand_in_engine():
    """The placement calibration of bench.py (candidates held together, spacers, the winner tuned once more,
    landcover placements, next-best candidates handed to the config-4 leg) runs through on the stand-in engine."""
    out = _bench(["--steps", "3", "--warmup", "1", "--size", "2048", "--no-cpu-baseline", "--tune-arenas", "7"])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "calibration skipped" not in out.stderr
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    pl = rec["roofline"]["placement"]
    assert len(pl["allocations_tried_best_ms"]) == 7 and len(pl["landcover_placements_tried_best_ms"]) == 3
    assert pl["best_ms"] == min(pl["allocations_tried_best_ms"])            # the 'fast' arena of the stand-in won
    assert rec["also"]["workload"].startswith("config4")
    # and switched off
    out = _bench(["--steps", "2", "--warmup", "0", "--size", "2048", "--no-cpu-baseline", "--no-tune"])
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["roofline"]["placement"] is None


def test_bench_strong_scaling_splits_one_block_into_row_bands():
    """--scaling strong: SURVEY 8(e) "within a single huge tile, split by row ranges across GPUs"."""
    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "2050", "--no-cpu-baseline",
                  "--scaling", "strong"])
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2
    bands = [r["rows"] for r in rec["per_rank"]]
    assert bands[0][0] == 0 and bands[0][1] == bands[1][0] and bands[1][1] == 2050 and bands[0][1] % 16 == 0
    want = 2050 * 2050 * 1 * 3 / (rec["ms_per_step"] * 3e-3) / 1e9          # ONE block over both GPUs
    assert abs(rec["value"] - want) / want < 1e-3


def test_bench_line_has_the_contract_fields_and_cpu_baseline_schema():
    """The one JSON line: every field of the driver's contract, roofline / cpu_baseline objects, and the
    cpu_baseline legs (P = 1, the box share, all physical cores, the fused best-CPU pass) on a small width."""
    out = _bench(["--steps", "2", "--warmup", "1", "--size", "2048"], extra_env={"GCN10_CPU_BASELINE_PROCS": "2"})
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "per_rank"):
        assert k in rec, k
    assert rec["metric"] == "CN Gpixels/sec" and rec["unit"] == "Gpx/s" and rec["dtype"] == "u8"
    assert rec["higher_is_better"] is True and rec["vs_baseline"] is None and "workload" in rec["config"]
    ro = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch",
              "avg_launch_ms", "copy_ceiling", "frac_of_copy", "placement"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and ro["unit"] == "GB/s"
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    cb = rec["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "runs", "best_cpu", "host_physical_cores"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == "CN Gpx/s" and cb["value"] > 0
    labels = [r["label"] for r in cb["runs"]]
    assert labels[0] == "P=1" and any("share" in l for l in labels)
    assert cb["value"] == max(r["gpx_per_s"] for r in cb["runs"]) and cb["cores"] in [r["procs"] for r in cb["runs"]]
    assert cb["best_cpu"]["gpx_per_s"] > 0
