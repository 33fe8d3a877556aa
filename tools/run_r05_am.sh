#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 200 python tools/quick_knn.py 2>&1 | tail -n 1 | cut -c1-420
cp mr_slam_amd/libmrslam_hip.so /tmp/q4.so; cp mr_slam_amd/libmrslam_hip_q8.so mr_slam_amd/libmrslam_hip.so
timeout 200 python tools/quick_knn.py 2>&1 | tail -n 1 | cut -c1-420
timeout 300 python -m pytest tests/test_gicp_gpu.py -m gpu -q -x -k "knn" 2>&1 | grep -E "passed|failed" | tail -n 1
cp /tmp/q4.so mr_slam_amd/libmrslam_hip.so
