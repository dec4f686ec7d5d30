"""`python bench.py --gpus N` must start its N ranks itself (VERDICT r01 item 5 / ADVICE): the parent spawns fresh child
processes with the torch.distributed.run environment before any GPU call and forwards rank 0's JSON line.  Here the ranks
run with --stub-device (gloo rendezvous, barrier, max-over-ranks, stream partition; the decode step is a sleep), so the
launcher and the aggregation are exercised on CPU; the real path differs only in the backend (nccl) and the step body."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    out = run("--gpus", "2", "--steps", "20", "--warmup", "0", "--batch", "3", "--stub-device")
    assert out["n_gpus"] == 2 and out["data"] == "stub" and out["scaling"] == "weak"
    assert out["config"]["streams"] == 6                       # 2 ranks x 3 streams, partitioned without overlap
    assert out["metric"].startswith("STUB ") and "@2 GPU" in out["metric"]
    # max over ranks: rank 1 sleeps 10 % longer than rank 0
    assert out["ms_per_step"] >= 1.05
    assert abs(out["value"] - 6 * 20 / (out["ms_per_step"] * 20 / 1e3)) / out["value"] < 1e-3


def test_single_rank_needs_no_launcher():
    out = run("--gpus", "1", "--steps", "5", "--warmup", "0", "--stub-device")
    assert out["n_gpus"] == 1 and out["config"]["streams"] == 1


def test_under_an_external_launcher_env_it_is_one_rank():
    # torch.distributed.run sets RANK/WORLD_SIZE: bench.py must not spawn again; a --gpus / WORLD_SIZE mismatch is an error
    e = {k: v for k, v in os.environ.items()}
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-device"], env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "launcher started 1 rank" in p.stderr


def test_metric_names():
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert bench.metric_name("1.5B", 1, 1) == base              # the headline line carries BASELINE.json's metric string verbatim
    assert "@8 GPU" in bench.metric_name("v6-7B", 16, 8) and "RWKV-6 7B Q5_K_M" in bench.metric_name("v6-7B", 16, 8)


def test_a_dying_rank_stops_the_launcher_within_seconds():
    # rank 1 exits before the rendezvous; the parent must notice, stop rank 0 (which is waiting in init_process_group)
    # and return non-zero quickly instead of sitting out the backend's timeout
    import time
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "0", "--stub-device",
                        "--stub-fail-rank", "1"], env=e, capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert p.returncode != 0
    assert "rank 1 exited with code 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]       # no result line from a failed run
    assert took < 60, took


def test_stream_partition_of_cfg4():
    # cfg 4: 128 independent streams over 8 GPUs = 16 per rank, no overlap, no gap; every rank's first tokens follow its GLOBAL stream ids
    sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd", "wrk"))
    import replicas
    seen = []
    parts = replicas.partition_streams(128, 8)
    for r in range(8):
        mine = parts[r]
        assert len(mine) == 16
        assert mine == [g for g in range(128) if g % 8 == r]        # batch index -> GPU b mod 8
        first = [(17 + 101 * g) % 65535 for g in mine]          # bench.py's first-token rule
        assert len(set(first)) == 16
        seen += mine
    assert sorted(seen) == list(range(128))
