"""Host-side pieces of the node twin (mr_slam_amd/node.py, compat/util.py) that need no GPU: the `util`-named mirrors registered by
compat.install(node=True) and the small helpers, against the reference's own util.py imported in place."""
import ast
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402


def test_install_node_registers_util_with_the_names_the_nodes_use():
    saved = dict(sys.modules)
    try:
        from mr_slam_amd import compat
        compat.install(node=True)
        import util
        assert util.__name__ == "mr_slam_amd.compat.util"
        for name in ("load_pc_infer", "generate_RING", "generate_RINGplusplus", "fast_corr", "fast_corr_RINGplusplus", "calculate_row_shift",
                     "solve_translation", "solve_translation_bev", "rotate_bev", "getSE3", "robotid_to_key", "device"):
            assert hasattr(util, name), name
    finally:
        for k in ("util", "voxelocc", "voxelfeat", "gputransform", "torch_radon", "pygicp"):
            sys.modules.pop(k, None)
            if k in saved:
                sys.modules[k] = saved[k]


@pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")
def test_every_util_name_the_ring_nodes_call_has_a_mirror():
    """names called in main_RING.py / main_RINGplusplus.py that come from `from util import *` (defined in util.py, not in the node file)"""
    from mr_slam_amd.compat import util as mirror
    util_src = ast.parse(open(os.path.join(ref_import.RING_ROS, "util.py")).read())
    util_defs = {n.name for n in util_src.body if isinstance(n, ast.FunctionDef)}
    for fn in ("main_RING.py", "main_RINGplusplus.py"):
        path = os.path.join(ref_import.RING_ROS, fn)
        if not os.path.exists(path):
            pytest.skip(fn + " not staged")
        tree = ast.parse(open(path).read())
        own = {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
        called = {n.func.id for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)}
        needed = (called & util_defs) - own
        assert needed and all(hasattr(mirror, n) for n in needed), sorted(n for n in needed if not hasattr(mirror, n))


@pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")
def test_helpers_equal_the_reference_util():
    from mr_slam_amd import node
    from mr_slam_amd.compat import util as mirror
    with ref_import.reference_modules("oracle") as ref:
        u = ref.util
        for s in range(-60, 120):
            assert mirror.calculate_row_shift(s) == u.calculate_row_shift(s)
        for x, y, yaw in ((0.0, 0.0, 0.0), (1.5, -2.25, 0.7), (-3.0, 4.0, -2.9)):
            assert np.array_equal(mirror.getSE3(x, y, yaw), u.getSE3(x, y, yaw))
        rng = np.random.default_rng(0)
        pc = rng.uniform(-90, 90, (5000, 3))
        assert np.array_equal(mirror.load_pc_infer(pc), u.load_pc_infer(pc))
        assert mirror.robotid_to_key(2) == u.robotid_to_key(2)


def test_bind_detect_loop_icp_runs_the_nodes_own_function_with_its_signature():
    """bind_detect_loop_icp re-binds the NODE'S OWN detect_loop_icp (its code object, its live globals): same signature, names the node defines
    after the binding line resolve, keyword overrides win, and a call that does not follow the loop's pattern falls back to the node's own
    fast_corr.  (Empty candidate lists: no device work, so this runs without a GPU.)"""
    import inspect
    from mr_slam_amd import node
    want = {"ring": ["robotid_current", "idx_current", "pc_current", "RING_current", "TIRING_current", "robotid_candidate", "pc_candidates",
                     "RING_candidates", "TIRING_candidates"],
            "ringpp": ["robotid_current", "idx_current", "pc_current", "bev_current", "TIRING_current", "robotid_candidate", "pc_candidates",
                       "bev_candidates", "TIRING_candidates"],
            "disco": ["robotid_current", "idx_current", "pc_current", "DiSCO_current", "fft_current", "robotid_candidate", "pc_candidates",
                      "DiSCO_candidates", "FFT_candidates"]}
    for kind, names in want.items():
        ns = {}
        exec("def detect_loop_icp(" + ", ".join(names) + "):\n"
             "    seen.append((len(pc_candidates), later, helper(3)))\n"
             "    return 'node'\n", ns)
        ns["seen"] = []
        ns["helper"] = lambda x: x + 1
        fn = node.bind_detect_loop_icp(ns, kind, helper=lambda x: x * 10)
        assert list(inspect.signature(fn).parameters) == names and fn.__wrapped__ is ns["detect_loop_icp"]
        ns["later"] = "defined after the binding line"          # like `f` / `pub` in the node's __main__ block
        assert fn(0, 1, None, None, None, 2, [], [], []) == "node"
        assert ns["seen"] == [(0, "defined after the binding line", 30)]
    if ref_import.available():
        for kind, rel in (("ring", os.path.join(ref_import.RING_ROS, "main_RING.py")), ("ringpp", os.path.join(ref_import.RING_ROS, "main_RINGplusplus.py")),
                          ("disco", os.path.join(ref_import.DISCO_ROS, "main.py"))):
            if not os.path.exists(rel):
                continue
            tree = ast.parse(open(rel).read())
            ref_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "detect_loop_icp"][0]
            assert [a.arg for a in ref_fn.args.args] == want[kind]
