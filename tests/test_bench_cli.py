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


def test_compact_line_is_small_strict_json():
    """The driver keeps a bounded tail of stdout: the contract line must stay under 4 KB of strict JSON whatever the detail blocks hold
    (round 5's 21 KB line left BENCH_r05.parsed null).  Input: a full result of an earlier round kept under profiles/, plus NaN / inf / numpy
    scalars planted in it."""
    import json
    import numpy as np
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    d["roofline"]["polar_frac"] = float("nan")
    d["roofline"]["bev_scatter_frac"] = np.float32(0.82)
    d["sweeps"]["ring_q1"]["hbm_frac"] = float("inf")
    d["exchange"] = {"process_group": {"backend": "nccl", "world_size": 8, "gpus_flag": 8, "devices_visible": 8}, "impl": "cabi", "verify": {"ok": True}}
    text = bench.compact_line(bench._strict(d), "bench_detail.json")
    assert len(text.encode()) < bench.COMPACT_LIMIT and "\n" not in text

    def bad(c):
        raise ValueError(c)
    c = json.loads(text, parse_constant=bad)
    assert c["value"] == d["value"] and c["ms_per_step"] == d["ms_per_step"] and c["roofline"]["frac"] == d["roofline"]["frac"]
    assert c["roofline"]["traffic"] == d["roofline"]["traffic"] and "polar_frac" not in c["roofline"] and "ring_q1_frac" not in c["roofline"]
    assert abs(c["roofline"]["bev_scatter_frac"] - 0.82) < 1e-6 and c["cpu_baseline"]["kind"] == "port" and c["exchange"]["process_group"]["world_size"] == 8
    assert c["verify"] == {"ok": True, "checked": 8} and c["detail"] == "bench_detail.json"
    # strict detail: NaN / inf become null
    s = bench._strict({"a": float("nan"), "b": [np.int64(3), float("-inf")], "c": np.bool_(True)})
    assert s == {"a": None, "b": [3, None], "c": True}
