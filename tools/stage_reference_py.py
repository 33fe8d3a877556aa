#!/usr/bin/env python3
"""Stages the handful of REFERENCE Python files that tests/test_reference_dropin_gpu.py executes through the drop-in on a GPU box
(where /root/reference does not exist) into tests/_refpy/ -- a git-ignored scratch directory that travels with the gpurun snapshot like
the built .so files, is never committed and is never imported by the product.  Run by __graft_entry__.build() wherever the reference
tree is present.  Files (read where they lie, copied byte for byte, same relative layout so that tests/golden/ref_import.py finds them):
  LoopDetection/src/RING_ros/{util.py, config.py}            generate_RING, generate_RINGplusplus, fast_corr, ...
  LoopDetection/src/disco_ros/{config.py, main.py, models/DiSCO.py}   DiSCO.forward, phase_corr, detect_loop_icp
  LoopDetection/src/RING_ros/{main_RING.py, main_RINGplusplus.py}     detect_loop_icp (tests/test_node_gpu.py: the node's loop against its twin)"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MRSLAM_REFERENCE", "/root/reference")
FILES = ["LoopDetection/src/RING_ros/util.py", "LoopDetection/src/RING_ros/config.py", "LoopDetection/src/disco_ros/config.py",
         "LoopDetection/src/disco_ros/main.py", "LoopDetection/src/disco_ros/models/DiSCO.py",
         "LoopDetection/src/RING_ros/main_RING.py", "LoopDetection/src/RING_ros/main_RINGplusplus.py"]


def stage(quiet=False):
    if not os.path.isdir(os.path.join(REF, "LoopDetection")):
        return False
    dst_root = os.path.join(ROOT, "tests", "_refpy")
    for f in FILES:
        dst = os.path.join(dst_root, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, f), dst)
    with open(os.path.join(dst_root, "README"), "w") as o:
        o.write("scratch copies of reference files for the GPU-box test (tools/stage_reference_py.py); git-ignored, never committed\n")
    if not quiet:
        print(f"staged {len(FILES)} reference Python files under tests/_refpy/")
    return True


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
