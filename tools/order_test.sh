cd $GRAFT_REPO_ROOT
for order in A B C D; do
python - $order <<'PY' 2>&1 | grep -v amdgpu.ids | tail -2
import sys, numpy as np
sys.path.insert(0, "bindings/pybind/_built")
o = sys.argv[1]
pts = np.random.default_rng(0).uniform(-5, 5, (1000, 3))
def t():
    import torch; torch.cuda.init(); return torch.cuda.device_count()
def g():
    import pygicp; return pygicp.downsample(pts, 0.5).shape
try:
    if o == "A": r = (t(), g())
    if o == "B": r = (g(), t())
    if o == "C":
        import pygicp; r = (t(), g())
    if o == "D":
        import torch, pygicp; r = (g(), t())
    print(o, "ok", r)
except Exception as e:
    print(o, "FAIL", type(e).__name__, str(e)[:100])
PY
done
