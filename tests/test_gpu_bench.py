"""GPU tier: bench.py's contract on the box -- the one-line JSON of a 1-rank run, and the driver's N > 1 launch
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with two ranks that share the box's one device
(XRIT_BENCH_SHARE_DEVICE=1: gloo for the barrier and the reductions, RCCL refuses two ranks on one device; every rank
still demodulates its own capture segment on the GPU, the timing is the max over ranks, the value the sum)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "2", "--burst-log2", "24", "--no-cpu", "--no-exact", "--no-serial-floor"]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _check_line(d, n):
    assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 2
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = the samples all ranks processed / the slowest rank's time
    assert abs(d["value"] - n * (1 << 24) * 3 / (d["ms_per_step"] * 3e-3) / 1e6) <= 0.01 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["loop_passes"]["costas_unconverged"] == 0


@pytest.mark.gpu
def test_bench_line_of_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_line(_last_json(r.stdout), 1)


@pytest.mark.gpu
def test_bench_two_ranks_launched_like_the_driver_does():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, XRIT_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines            # rank 0 prints, nobody else
    _check_line(json.loads(lines[0]), 2)


@pytest.mark.gpu
def test_circuit_rate_burst_runs_the_two_pass_plan_with_four_walkers_per_cu():
    """C1 at burst size (2^28 samples, no decimator: 63 M symbols): the relay plans two passes from the timing guess over
    segments that still hold 49 152 symbols -- up to four walkers per CU (1023 segments on a 256-CU part; 512 before round 4's
    second half) -- and every step closes its Costas loop in one pass over the samples."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--decimation", "1", "--steps", "3", "--warmup", "2",
                        "--no-cpu", "--no-exact", "--no-serial-floor"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    lp = d["loop_passes"]
    assert lp["clock"] == 0 and lp["clock_relay"] == 2 and lp["costas"] == 1 and lp["costas_unconverged"] == 0, lp
    assert 3 * 256 <= lp["clock_relay_segments"] <= 4 * 256, lp
    assert d["value"] > 0 and d["config"]["decimation"] == 1
