"""bench.py command-line contract that needs no GPU: `--gpus N` either runs N ranks or fails loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MRS_BENCH_SHARE_GPU")}
    env.update(kw)
    return env


def test_bare_gpus_flag_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert p.returncode != 0 and "needs 2 visible GPUs" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]       # no JSON line that could be mistaken for a measurement


def test_world_size_must_match_gpus_flag():
    """a launcher that starts fewer ranks than --gpus says is an error, not a silent one-rank measurement"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_free_port_is_bindable():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    import socket
    port = bench.free_port()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", port))
