"""bench.py's launcher plumbing (no GPU needed): `python bench.py --gpus N` without a launcher must start its N ranks as a CHILD
`python -m torch.distributed.run` (never an exec), refuse quickly when the node has fewer GPUs, and `--form abi` is one process.
The GPU half (a two-rank gloo rehearsal through the very same self-spawn path reproduces the one-rank marker) is marked gpu."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_dry_run_prints_the_child_command():
    r = _run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["child_command"]
    assert d["launcher"] == "self-spawn" and d["n_gpus"] == 4
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]  # the ranks get the same arguments, minus --dry-run


def test_one_gpu_and_external_launcher_do_not_spawn():
    d = json.loads(_run(["--dry-run"]).stdout)
    assert "child_command" not in d and d["n_gpus"] == 1
    d = json.loads(_run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}).stdout)
    assert "child_command" not in d and d["launcher"] == "external"
    d = json.loads(_run(["--gpus", "2", "--form", "abi", "--dry-run"]).stdout)
    assert d["form"] == "abi" and d["devices"] == [0, 1] and "single process" in d["launcher"]


def test_more_gpus_than_the_node_has_fails_fast():
    import torch
    have = torch.cuda.device_count()
    t = time.time()
    r = _run(["--gpus", str(have + 1) if have else "2"])
    assert r.returncode == 3 and "nothing was launched" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t < 60
    r = _run(["--gpus", str(have + 1) if have else "2", "--form", "abi"])
    assert r.returncode == 3 and "GPU(s)" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,n", [(2, 3200), (4, 3000)])
def test_rank_rehearsal_through_the_self_spawn_path(tmp_path, ranks, n):
    """Two / four ranks sharing the one card (gloo: collectives through the host), started by bench.py itself, select the marker one
    rank selects, with the same tsq; the JSON says how many ranks RCCL saw (0 here: no RCCL transfer has run on this box).  n = 3,000
    pads to 24 row tiles of 128, so four ranks SHARE the rows of W (one all-gather) when that is the faster form; 3,200 pads to 26."""
    common = ["--individuals", str(n), "--markers", "150001", "--steps", "2", "--warmup", "1", "--no-secondary", "--cpu-sample", "0"]
    one = json.loads(_run(common, timeout=900).stdout.strip().splitlines()[-1])
    many_p = _run(["--gpus", str(ranks)] + common, env={"EAGLE_BENCH_BACKEND": "gloo"}, timeout=900)
    assert many_p.returncode == 0, many_p.stderr[-2000:]
    many = json.loads(many_p.stdout.strip().splitlines()[-1])
    assert one["n_gpus"] == 1 and many["n_gpus"] == ranks
    assert many["launcher"].startswith("self-spawned") and many["rccl_ranks"] == 0 and "gloo" in many["collective_backend"]
    assert sum(many["markers_per_rank"]) == 150001 and len(many["markers_per_rank"]) == ranks
    assert max(many["markers_per_rank"]) - min(many["markers_per_rank"]) <= 1
    assert many["selected_marker"] == one["selected_marker"]
    # the rows of W come from the row-block product when they are shared between ranks: same sums in another order
    assert abs(many["tsqmax"] - one["tsqmax"]) <= 1e-9 * abs(one["tsqmax"])
    if "replicated" in many["w_sharing"]:
        assert many["tsqmax"] == one["tsqmax"]
    assert one["rccl_ranks"] == 0 and one["launcher"] == "single process"
