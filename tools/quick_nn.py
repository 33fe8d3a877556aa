"""Round-4 A/B of the two nearest-neighbour search cores on the bench's GICP shape (development aid).
usage: quick_nn.py [n_pairs] [--feat]   (with --feat also the RING++ kNN front end on 64 scans; the core of that call is chosen by
MRS_DEV=1 MRS_NN_CORE=0|1 in the environment)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mr_slam_amd import gicp

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
srcs, tgts = bench._gicp_pairs(n_pairs, 0)
out = {"pairs": n_pairs}


def sync():
    torch.cuda.synchronize()


w = gicp.GicpBatch(1); w.set_sources([srcs[0][:4000]]); w.set_targets([tgts[0][:4000]]); w.align(); del w
ref = {}
for core in (1, 0, 2, 3):
    b = gicp.GicpBatch(n_pairs)
    b.set_search(core)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    sync(); t = time.perf_counter()
    b.set_sources(srcs); b.set_targets(tgts); sync()
    t_set = time.perf_counter() - t
    t = time.perf_counter(); b.compute_covariances(0); sync(); t_c0 = time.perf_counter() - t
    t = time.perf_counter(); b.compute_covariances(1); sync(); t_c1 = time.perf_counter() - t

    def timed(**prm):
        b.set_params(**prm); sync(); t = time.perf_counter(); T, conv, its = b.align(); sync()
        return time.perf_counter() - t, T, conv, its, b.nn_passes
    t_cold, T5, _, _, _ = timed(force_iterations=5)
    t_f, T20, _, _, nn_f = timed(force_iterations=20)
    b.set_sources(srcs); b.compute_covariances(0)
    t_nat, Tn, conv, its, nn_n = timed(force_iterations=0)
    r = {"set_clouds_s": t_set, "cov_src_ms": 1e3 * t_c0, "cov_tgt_ms": 1e3 * t_c1, "cold5_ms": 1e3 * t_cold, "forced20_ms": 1e3 * t_f,
         "forced20_iters_per_s": n_pairs * 20 / t_f, "natural_ms": 1e3 * t_nat, "natural_pairs_per_s": n_pairs / t_nat,
         "natural_incl_cov_pairs_per_s": n_pairs / (t_nat + t_c0 + t_c1), "mean_its": float(np.mean(its)), "nn_passes_nat": nn_n, "converged": int(conv.sum()), "searched_nat": b.searched_fraction}
    if core in ref:
        r["max_abs_dT_vs_first_run"] = float(np.abs(Tn - ref[core]).max())
    ref.setdefault(core, Tn)
    if core != 1:
        r["max_abs_dT_vs_core1"] = float(np.abs(Tn - ref[1]).max())
        r["max_abs_dT20_vs_core1"] = float(np.abs(T20 - ref["T20"]).max())
    else:
        ref["T20"] = T20
    out.setdefault(f"core{core}", []).append(r)
    print(core, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    del b
if "--feat" in sys.argv:
    from mr_slam_amd import pointfeat
    shard = bench.make_shard(64, 1, 0, "cuda:0")
    pts = bench.make_shard.whole[0, :64].permute(0, 2, 1).reshape(64 * bench.N_POINTS, 3).contiguous()
    offs = np.arange(65, dtype=np.int64) * bench.N_POINTS
    pointfeat.point_features(pts[:8 * bench.N_POINTS], offs[:9], 30, want=("planes",)); sync()
    t = time.perf_counter(); pl = pointfeat.point_features(pts, offs, 30, want=("planes",)); sync()
    out["feat_ms_64_scans"] = 1e3 * (time.perf_counter() - t)
    out["feat_checksum"] = float(pl["planes"].double().sum())
    print("feat", out["feat_ms_64_scans"], out["feat_checksum"], os.environ.get("MRS_NN_CORE"))
print(json.dumps(out))
