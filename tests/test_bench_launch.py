"""bench.py's launching paths on CPU (`--dry-run`: gloo, CPU tensors, numpy stand-ins
of the device steps — tests/numpy_ops.py). What is under test: that
`python bench.py --gpus N` with no launcher around it starts N fresh ranks by
itself, that the driver's `python -m torch.distributed.run ... bench.py --gpus N`
form works too, and that the one JSON line carries the weak / strong / uint64
legs with the rank count really seen. Nothing here is a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % proc.stdout[-500:]
    return json.loads(lines[0])


def _check(d, world):
    assert d["n_gpus"] == world and d["correct"] is True and d["scaling"] == "weak"
    assert d["ranks_seen"] == {"world_size": world, "allreduce_of_ones": world}
    assert d["data"].startswith("dry-run")
    assert set(d["phases_ms"]) == {"partition", "count_exchange", "key_exchange", "local_sort"}
    for leg in ("strong", "config5_u64"):
        assert d[leg]["correct"] is True and d[leg]["ranks_seen"]["allreduce_of_ones"] == world
    # every leg says what it sent over the links and how the local sort stands against the HBM peak
    for leg in (d, d["strong"], d["config5_u64"]):
        x = leg["exchange_stats"]
        assert x["bytes_out_per_gpu"] > 0 and x["xgmi_peak_GBps"] == round((world - 1) * 153.0, 1)
        assert x["key_exchange_GBps"] > 0 and x["xgmi_frac"] > 0 and x["slices"] >= 1
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["bytes"] > 0 and d["roofline"]["frac"] > 0
    assert d["strong"]["local_sort_roofline"]["bytes"] > 0
    assert d["strong"]["elements_total"] == d["config"]["elements_per_gpu"]           # the same array, split N ways
    assert d["config5_u64"]["dtype"] == "u64" and d["config5_u64"]["elements_per_gpu"] == d["config"]["elements_per_gpu"]


@pytest.mark.parametrize("world", [2, 8])
def test_plain_invocation_launches_its_own_ranks(world):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps", "2", "--warmup", "1", "--dry-run",
                        "--log2n", "11"], capture_output=True, text=True, env=env, timeout=600)
    _check(_line(p), world)


def test_drivers_torchrun_form():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--dry-run", "--log2n", "11"], capture_output=True, text=True, timeout=600)
    _check(_line(p), 2)


def test_reference_round_trip_yardstick():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reference_bitonic_round_trips(26) == 58           # G_ref at 2^26, lws 1024
    assert bench.reference_bitonic_round_trips(12) == 12
    assert bench.contract_bytes_per_elem("satradix_u32", 16, 28) == 96
    assert bench.contract_bytes_per_elem("sbitonic", 16, 16) == 8 * 136


def test_watchdog_ends_a_stuck_rank_with_a_nonzero_exit():
    """A leg that takes longer than --watchdog seconds (here: any leg, the limit is 50 ms) ends the rank with
    exit code 3 — a plain exit of a fresh child process — and the launcher reports failure instead of hanging."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "200", "--warmup", "1", "--dry-run", "--log2n", "15",
                        "--watchdog", "0.05"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode != 0
    assert "watchdog" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
