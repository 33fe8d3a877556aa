"""Host logic on the far side of the boundary (row N2): the loop-message ids and the loopinfo.txt line, pinned to the reference's own
robotid_to_key (RING_ros/util.py:253-260, imported from /root/reference where that tree is present) and to the literal statement of
main_RING.py:221-222,229-231 (the NCLT record decoder is pinned in tests/test_oracle_bev.py)."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402


def test_loop_ids_and_loopinfo_line():
    from mr_slam_amd import preprocess as P
    id0, id1 = P.loop_ids(0, 41, 2, 7)
    assert id0 == (97 << 56) + 42 and id1 == (99 << 56) + 8                  # main_RING.py:221-222: key + index + 1
    assert chr(id0 >> 56) == "a" and chr(id1 >> 56) == "c" and (id0 & ((1 << 56) - 1)) == 42
    assert id0 < 2 ** 63                                                      # fits dislam_msgs/Loop.msg's int64 up to robot 'z' + ...
    assert P.loop_ids(25, 0, 0, 0)[0] == (122 << 56) + 1
    # main_RING.py:229-231: ' '.join(str(i) for i in [ids..., position xyz, orientation xyzw])
    assert P.loopinfo_line(0, 41, 2, 7, (1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0)) == "0 41 2 7 1.0 2.0 3.0 0.0 0.0 0.0 1.0"
    line = P.loopinfo_line(1, 5, 0, 9, (np.float64(0.25), -1.5, 2.0), (0.0, 0.0, 0.5, 0.5))
    assert line.split(" ") == ["1", "5", "0", "9", "0.25", "-1.5", "2.0", "0.0", "0.0", "0.5", "0.5"]


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")
def test_robotid_to_key_equals_the_reference_function():
    from mr_slam_amd import preprocess as P
    with ref_import.reference_modules("dropin") as ref:
        for robot in range(26):
            with contextlib.redirect_stdout(io.StringIO()):                   # the reference prints every call
                want = ref.util.robotid_to_key(robot)
            assert P.robotid_to_key(robot) == want


def test_bench_valu_roofline_helper():
    """bench.py's VALU floor: always the bounds insts x 2 and insts x 4 cycles over 1024 SIMDs at 2.4 GHz; the class-weighted floor when the
    profile carries the instruction-class counters (profiles/r03_ubench.md: the cost depends on the instruction)"""
    import bench
    n = 196.46e6
    e = {"counters": {"SQ_INSTS_VALU": n}}
    v = bench.valu_roofline(e, 0.405)
    assert v and abs(v["floor_ms_bounds"][0] - n * 2 / (1024 * 2.4e9) * 1e3) < 1e-12 and abs(v["floor_ms_bounds"][1] - 2 * v["floor_ms_bounds"][0]) < 1e-12
    assert "floor_ms" not in v and 0.35 < v["frac_bounds"][0] < 0.45 and 0.7 < v["frac_bounds"][1] < 0.9
    e["valu_pipe_cycles_est"] = 3.0 * n
    v = bench.valu_roofline(e, 0.405)
    assert abs(v["floor_ms"] - 3.0 * n / (1024 * 2.4e9) * 1e3) < 1e-12 and abs(v["mean_cycles_per_inst"] - 3.0) < 1e-12 and 0.5 < v["frac"] < 0.7
    assert bench.valu_roofline({}, 0.4) is None and bench.valu_roofline(None, 0.4) is None and bench.valu_roofline(e, 0.0) is None


def test_reference_python_staging_for_the_gpu_box():
    """tools/stage_reference_py.py: the five reference Python files the GPU-box test executes through the drop-in are copied into the git-ignored
    scratch directory tests/_refpy/ (same relative layout), and tests/golden/ref_import.py falls back to it where /root/reference is absent."""
    import os
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stage_reference_py as S
    assert "tests/_refpy/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    assert not any(l.strip().startswith("tests/_refpy") for l in open(os.path.join(ROOT, ".gpurunignore")).read().splitlines()) \
        if os.path.exists(os.path.join(ROOT, ".gpurunignore")) else True
    if not os.path.isdir(os.path.join(S.REF, "LoopDetection")):
        pytest.skip("reference tree not present")
    assert S.stage(quiet=True)
    for f in S.FILES:
        staged = os.path.join(ROOT, "tests", "_refpy", f)
        assert os.path.isfile(staged) and open(staged, "rb").read() == open(os.path.join(S.REF, f), "rb").read()


def test_committed_pmc_summary_prices_instructions_by_class():
    """profiles/r03_pmc.json: the VALU-pipe estimate of every profiled kernel lies between the all-2-cycle and the all-4-cycle bounds and the fused
    descriptor kernel carries the launch size it was profiled at (bench.py scales its counters to 1024 scans)."""
    import json
    import os
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))
    assert pmc["k_bev_radon3"]["launch_scans"] == 16384
    seen = 0
    for k, e in pmc.items():
        n = e["counters"].get("SQ_INSTS_VALU")
        if not n or not e.get("valu_pipe_cycles_est"):
            continue
        seen += 1
        assert 2.0 * n <= e["valu_pipe_cycles_est"] <= 4.0 * n + 1e-6, k
        lo, hi = e["valu_busy_frac_bounds"]
        assert lo <= e["valu_issue_frac"] <= hi + 1e-9
    assert seen >= 8


def test_bench_database_slot_plan():
    """bench.py's database slots (N = 1: two sets written alternately; N > 1: one set + copies): a launch reads the entry built DEPTH launches
    earlier, across the step boundary; a group's kernel never reads a slot it writes; with two sets no slot read during a step is written in
    that step after the read's launch, and the step's LAST group of launches reads only the set the next step does not write (the lagged join)."""
    import bench
    for CH, FUSE in ((48, 16), (6, 2), (3, 3), (5, 0), (48, 1)):
        DEPTH = min(FUSE + 2, CH) if FUSE else 2
        built = {}                                     # slot -> (step, launch) that wrote it
        for two in (True, False):
            built.clear()
            for c in range(CH):                        # step 0 (setup): both sets / the copies hold step "-1"
                built[bench.write_slot(c, 0, CH, two)] = (-1, c)
                if two:
                    built[bench.write_slot(c, 1, CH, two)] = (-1, c)
            for step in range(1, 5):
                par = step % 2 if two else 0
                if not two:                            # the copies made when the step starts
                    for c in range(DEPTH):
                        built[CH + c] = built[CH - DEPTH + c]
                G = FUSE if FUSE else 1
                for g0 in range(0, CH, G):
                    grp = range(g0, min(g0 + G, CH))
                    writes = {bench.write_slot(c, par, CH, two) for c in grp}
                    for c in grp:
                        r = bench.read_slot(c, par, CH, DEPTH, two)
                        assert r not in writes, (CH, FUSE, two, c)
                        st, lc = built[r]
                        want = (step, c - DEPTH) if c >= DEPTH else (step - 1 if step > 1 else -1, CH - DEPTH + c)
                        assert (st, lc) == want, (CH, FUSE, two, step, c, (st, lc), want)
                        if two and c >= DEPTH:
                            assert r // CH == par          # only this step's set
                        if two and c < DEPTH:
                            assert r // CH == 1 - par      # the other set: not written during this step at all
                    for c in grp:
                        built[bench.write_slot(c, par, CH, two)] = (step, c)
                if two and G < CH and (CH - 1) // G * G >= DEPTH:
                    last = range((CH - 1) // G * G, CH)    # the batch of sweeps the lagged join lets run into the next step
                    assert all(bench.read_slot(c, par, CH, DEPTH, two) // CH == par for c in last)
