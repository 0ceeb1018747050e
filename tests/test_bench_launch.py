"""bench.py --gpus N must run N ranks whether or not a launcher started it: without
WORLD_SIZE in the environment it starts the ranks itself.  Checked here without a GPU through
its --selftest-ranks mode (gloo rendezvous + one all-reduce per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = os.environ.copy()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [1, 2, 3])
def test_bench_starts_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--selftest-ranks"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0's line only
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["rank_sum"] == n * (n + 1) / 2
    # the self-checks an N > 1 line carries (here over gloo, on stand-in data)
    m = j["multi_gpu"]
    assert m["backend"] == "gloo" and m["ranks_seen"] == n
    assert m["ms_per_step_per_rank"] == [1.0 + r for r in range(n)]
    assert m["ms_per_step_min"] == 1.0 and m["ms_per_step_max"] == float(n)
    assert m["frames_identical_across_ranks"] is True and len(set(m["frame_hashes"])) == 1 and len(m["frame_hashes"]) == n


def test_a_rank_with_another_picture_is_noticed():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--selftest-ranks", "--selftest-differ-rank", "2"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["multi_gpu"]
    assert m["frames_identical_across_ranks"] is False and len(set(m["frame_hashes"])) == 2


def test_a_failing_rank_fails_the_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-ranks", "--selftest-fail-rank", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr


def test_under_a_launcher_the_world_must_match():
    env = _env()
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--selftest-ranks"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_torchrun_still_works():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", BENCH, "--gpus", "2", "--selftest-ranks"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2


def test_turntable_gives_every_rank_every_frame_once():
    """Whole-frame mode: in K steps every rank renders each of the K turntable frames exactly once
    (equal work per GPU at every N), and the ranks start evenly spread round the table."""
    import bench
    for frames in (1, 5, 20, 37):
        for world in (1, 2, 4, 8):
            starts = []
            for rank in range(world):
                seen = sorted(bench.turntable_frame(i, rank, world, frames) for i in range(frames))
                assert seen == list(range(frames))
                starts.append(bench.turntable_frame(0, rank, world, frames))
                # the warm-up steps (negative) stay on the table
                assert 0 <= bench.turntable_frame(-3, rank, world, frames) < frames
            assert starts == sorted(starts) and starts[0] == 0
