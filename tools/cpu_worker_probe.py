"""What one worker of the all-core CPU baseline spends where, alone and next to 31 others (development aid for bench.py's cpu_baseline leg)."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mr_slam_amd import synth
scans = [synth.uniform_scan(s, 120000) for s in range(16)]
td = tempfile.mkdtemp()
p = os.path.join(td, "w.npz"); np.savez(p, **{f"s{i:03d}": synth.to_soa(scans[i]) for i in range(16)})
for env in ({}, {"OMP_PROC_BIND": "false"}, {"OMP_WAIT_POLICY": "active"}, {"OMP_PROC_BIND": "false", "OMP_WAIT_POLICY": "active"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_worker", "ring", p, "3", "8", repr(time.time() + 6)], capture_output=True, text=True, env=e,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("alone", env, r.stdout.strip()[-220:], r.stderr[-200:], flush=True)
for nw, th in ((32, 8), (16, 8), (64, 4), (128, 2)):
    res = bench._cpu_workers("ring", [p] * nw, th)
    print(nw, "workers x", th, "threads:", round(sum(x["units"] for x in res) / max(x["seconds"] for x in res)), "scans/s; slowest", {k: round(v, 3) for k, v in max(res, key=lambda x: x["seconds"]).items()}, flush=True)
