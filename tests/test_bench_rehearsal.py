"""The driver's multi-GPU command, rehearsed.  The scaling bench runs on a node this pool does not lend out, so its first
execution must not die on launcher / rendezvous / argument plumbing:

* CPU tier (`PA_BENCH_REHEARSE=dry`): `bench.py --gpus 8` exactly as the driver starts it -- `python -m torch.distributed.run
  --nproc-per-node 8 ... bench.py --gpus 8 ...` -- and bench.py's own launcher (no WORLD_SIZE in the environment), down to the
  hierarchy, the rank-aware internal tiling and the shard every rank would build, without touching a GPU.
* GPU tier: `PA_BENCH_REHEARSE=1 python bench.py --gpus N` with 2 and 4 ranks sharing the box's one card (gloo transport: the
  whole N > 1 product path -- sharded levels, pack / exchange / unpack, reductions, max over ranks, ONE JSON line), and
  `PA_BENCH_REHEARSE=rccl` with `PA_RCCL_TIMEOUT`: the RCCL bring-up is attempted, cannot succeed with two ranks on one card,
  and the run must degrade to the gloo transport and still print its line.  Eight ranks on one card are not possible here: the
  pool allows six processes on a card and pytest itself holds one; the 8-rank shard is covered by the dry run and `--sim-of 8`.
No scaling number is derived from any of this."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_json_line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[0])


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("how", ["driver_command", "self_launch"])
def test_dry_run_of_the_eight_gpu_command(how):
    args = ["--gpus", "8", "--steps", "2", "--warmup", "1", "--base", "64", "--box", "16"]
    if how == "driver_command":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
               os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env(PA_BENCH_REHEARSE="dry"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _one_json_line(out)
    assert r["dry_run"] is True and r["n_gpus"] == 8 and r["steps"] == 2 and r["warmup"] == 1
    assert len(r["boxes_per_level_by_rank"]) == 8 and all(all(n >= 1 for n in b) for b in r["boxes_per_level_by_rank"]), r
    assert sum(r["cells_by_rank"]) == r["cells"] == 3 * 64 ** 3 and max(r["cells_by_rank"]) == min(r["cells_by_rank"])


def test_dry_run_headline_shard_over_eight_ranks():
    """the driver's real N = 8 case: 64 boxes of 128^3 per level, 8 per rank and level, equal volumes"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env(PA_BENCH_REHEARSE="dry"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _one_json_line(out)
    assert r["n_gpus"] == 8 and r["swept_boxes_per_level"] == [64, 64, 64] and r["boxes_per_level_by_rank"] == [[8, 8, 8]] * 8
    assert r["cells"] == 402653184 and r["cells_by_rank"] == [402653184 // 8] * 8


def _check_line(r, n):
    assert r["n_gpus"] == n and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "strong" and r["unit"] == "Mcells/s"
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["higher_is_better"] is True and r["dtype"] == "f64"
    assert r["config"]["exchange"]["mode"] not in ("", "none") and f"sharded over {n} ranks" in r["config"]["parallelism"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["achieved"] > 0 and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["launches"] >= 2
    assert "secondary" not in r and "cpu_baseline" not in r  # rank 0 at N = 1 only


@pytest.mark.gpu
@pytest.mark.parametrize("n,base,box", [(2, 128, 32), (4, 64, 16)])
def test_rehearsal_of_the_multi_gpu_bench_line(n, base, box):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--base", str(base), "--box", str(box), "--no-secondary"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env(PA_BENCH_REHEARSE="1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = _one_json_line(out)
    _check_line(r, n)
    assert "gloo" in r["config"]["exchange"]["mode"]


@pytest.mark.gpu
@pytest.mark.parametrize("limit", ["1", "60"])
def test_rccl_bring_up_failure_degrades_to_gloo(limit):
    """two ranks on ONE card: ncclCommInitRank cannot succeed -- it fails (limit 60: by its own error) or is abandoned after
    PA_RCCL_TIMEOUT seconds (limit 1); either way every rank must agree on the gloo transport and the line must come out, rc 0"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--base", "64", "--box", "16", "--no-secondary"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env(PA_BENCH_REHEARSE="rccl", PA_RCCL_TIMEOUT=limit))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    r = _one_json_line(out)
    _check_line(r, 2)
    assert "RCCL transport failed" in r["config"]["exchange"]["mode"] and "gloo" in r["config"]["exchange"]["mode"]
