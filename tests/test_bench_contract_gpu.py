"""bench.py prints ONE compact JSON line (< 4 KB, strict JSON: the driver keeps a bounded tail of stdout) with the fields the driver reads, and
writes every other block to the detail file (small configuration, one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strict_loads(text):
    def bad(c):
        raise ValueError(f"not strict JSON: {c}")
    return json.loads(text, parse_constant=bad)          # NaN / Infinity are refused


def _check_compact(last_line, detail):
    """the LAST stdout line: strict JSON under 4 KB that carries the contract fields, and whose scalars are the detail file's"""
    assert len(last_line.encode()) < 4096, len(last_line)
    c = _strict_loads(last_line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "timed_region_s", "roofline"):
        assert k in c, k
    assert c["value"] == detail["value"] and c["ms_per_step"] == detail["ms_per_step"] and c["n_gpus"] == detail["n_gpus"]
    assert "workload" in c["config"] and "model" not in c["config"] and c["config"]["launches_per_step"] == detail["config"]["launches_per_step"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["frac"] == detail["roofline"]["frac"] and r["algorithmic_bytes_per_launch"] == detail["roofline"]["algorithmic_bytes_per_launch"]
    if "cpu_baseline" in detail:
        b = c["cpu_baseline"]
        assert b["kind"] in ("port", "reference") and b["value"] > 0 and b["cores"] >= 1 and b["sample"] and b["unit"] == "pairs/s"
    return c


def _run(extra_env=None, extra_args=(), tmp=None):
    import tempfile
    env = dict(os.environ)
    env.update(extra_env or {})
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "detail.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "32",
                            "--chunks", "3", "--cpu-sample", "8", "--gicp-pairs", "2", "--gicp-iters", "6", "--detail-file", path, *extra_args],
                           capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.strip()]
        d = _strict_loads(open(path).read())          # every block of the result
    d["_compact"] = _check_compact(lines[-1], d)      # the contract line is the LAST line of stdout
    return d


def test_bench_line_contract():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "pairs/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 3 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]       # batch x chunks x steps / time
    assert d["config"]["pairs_per_rank_per_step"] == 96 and d["config"]["launches_per_step"] == 3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["gicp"]["iterations"] == 6 and d["gicp"]["iters_per_s"] > 0 and d["gicp"]["nn_passes"] == 6
    assert d["gicp"]["cold"]["iters_per_s"] > 0 and d["gicp"]["natural"]["converged"] == 2
    # north_star's own targets ride in the `roofline` block as scalars (the driver's record keeps those): both rasterisers' HBM fractions,
    # the GICP rates and a numeric roofline per GICP kernel, each measured with HIP events on a launch of that kernel alone
    for k in ("bev_scatter_frac", "polar_frac", "gicp_iters_per_s", "gicp_natural_pairs_per_s", "gicp_pairs_per_s_incl_covariances", "gicp_linearize_frac",
              "gicp_linearize_gbs", "gicp_linearize_ms", "gicp_nn_round3_all_ms", "gicp_nn_certify_ms", "gicp_nn_certified_pass_1mm_ms", "gicp_knn_select_ms",
              "gicp_cov_from_knn_ms"):
        assert isinstance(r[k], float) and r[k] > 0, k
    assert r["gicp_iters_per_s"] == d["gicp"]["iters_per_s"] and r["bev_scatter_frac"] == d["roofline_bev_scatter"]["frac"]
    for name, blk in d["gicp"]["roofline"].items():
        if "frac" in blk:
            assert 0 < blk["frac"] <= 1.0 and abs(blk["achieved"] - blk["bytes"] / (blk["ms"] * 1e-3) / 1e9) < 1e-6 * blk["achieved"], name
    assert 0 < d["gicp"]["natural"]["searched_fraction"] <= 1.0 and d["gicp"]["kernel_counts"]["correspondences"] > 0
    for leg in ("roofline_polar", "roofline_radon", "sweeps", "pipeline_shard", "dropin_latency"):
        assert leg in d, leg
    # the compact line carries the targets' scalars out of those blocks
    cr = d["_compact"]["roofline"]
    for k in ("bev_scatter_frac", "polar_frac", "frac_of_max_hbm_only_march_only", "gicp_iters_per_s", "gicp_iters_per_s_natural", "gicp_searched_fraction",
              "gicp_natural_pairs_per_s", "gicp_pairs_per_s_incl_covariances", "gicp_linearize_frac", "gicp_nn_certify_frac", "gicp_cov_from_knn_frac",
              "ring_q1_frac", "ringpp_q1_frac", "ring_q4_pairs_per_s", "node_twin_pairs_per_s"):
        assert isinstance(cr[k], float) and cr[k] > 0, k
    assert d["_compact"]["verify"] == {"ok": True, "checked": 8} and d["_compact"]["detail"] == "detail.json"
    assert abs(cr["gicp_iters_per_s"] - d["gicp"]["iters_per_s"]) < 1e-4 * cr["gicp_iters_per_s"]
    assert d["_compact"]["cpu_baseline"]["gicp_iters_per_s"] > 0
    assert d["roofline_polar"]["bound"] == "hbm" and d["sweeps"]["ring_q1"]["pairs_per_s"] > 0 and d["sweeps"]["disco_q4"]["queries_per_s"] > 0
    # round 5: the one-query sweeps on the database's resident (DMA-tiled) format next to the row layout; the node's shape; GICP protocols
    for k in ("ring_q1", "ring_q1_row_layout", "ringpp_q1", "ringpp_q1_row_layout", "disco_q1"):
        assert d["sweeps"][k]["pairs_per_s"] > 0, k
    ns = d["node_shape"]
    assert ns["matches_pairwise"] is True and ns["twin_pairs_per_s"] > 20 * ns["reference_loop_through_dropin"]["pairs_per_s"] and ns["append"]["entries_per_s"] > 0
    assert d["gicp"]["warm"]["iters_per_s"] > 0 and "cold start" in d["gicp"]["protocol"] and d["gicp"]["shared_submaps"]["pairs_per_s_incl_covariances"] > 0
    assert r["gicp_iters_per_s_warm"] == d["gicp"]["warm"]["iters_per_s"] and r["gicp_pairs_per_s_incl_covariances_shared_submaps"] > 0
    assert 0 < r["frac_of_max_hbm_only_march_only"] and set(r["fused_phase_floors_ms_per_launch"]) == {"full", "hbm_only_no_march", "march_only_no_rasteriser"}
    # default step: BEV + Radon + normalisation of a group of launches in one kernel; the rasteriser's own roofline rides along
    assert d["config"]["fused_launches"] == 3 and "k_bev_radon3" in r["kernel"] and d["kernel_ms"]["bev_radon"] > 0
    assert set(r["fused_grid_ms_per_launch"]) == {"persistent", "per_pair"} and r["valu_roofline"] is None or r["valu_roofline"]["floor_ms_bounds"][0] > 0
    b3 = d["builds"]
    assert b3["disco_build"]["scans_per_s"] > 0 and b3["ringpp_build"]["scans_per_s"] > 0 and b3["ingest"]["scans_per_s"] > b3["ingest"]["per_scan_calls"]["scans_per_s"] * 0.5
    hf = b3["ingest_from_host"]          # round 6: pinned host clouds -> copy stream -> voxel grid + crop + descriptors, double-buffered
    assert hf["last_batch_bit_identical_to_resident_input"] is True and hf["scans_per_s"] > 0 and 0 < hf["frac_of_pinned_copy"] <= 1.05
    assert hf["pinned_copy_only_gbs"] > 1.0 and cr["host_fed_scans_per_s"] > 0
    # --verify (default 8): outputs of the timed loop's last fused launch against the oracle, after the timed region
    v = d["verify"]
    assert v["ok"] and v["checked"] == 8 and v["bev_mismatches"] == 0 and v["sinogram_mismatches"] == 0 and v["angle_mismatches"] == 0, v
    assert v["timed_vs_fresh_launch_mismatches"] == 0 and v["max_err_dist"] < 1e-5 and v["max_err"] < 2e-5, v
    # default schedule: the group's correlation in one launch, the per-launch sweeps on the side stream -- checked against fresh sweeps
    assert d["config"]["corr_launches_grouped"] == 3 and d["config"]["sweep_stream"] == "side" and v["sweep_mismatches"] == 0
    assert d["kernel_ms"]["corr"] > 0 and d["kernel_ms"]["sweep_standalone"] > 0 and "sweep" not in d["kernel_ms"]
    assert d["side_stream"]["sweeps_stream_time_ms_per_launch"] > 0
    # the kernel times of a step's launches (the sweeps overlap them) fit inside the step
    assert 3 * (d["kernel_ms"]["bev_radon"] + d["kernel_ms"]["corr"]) <= d["ms_per_step"] * 1.02
    b = d["roofline_bev_scatter"]
    assert b["bound"] == "hbm" and "k_cart_lds" in b["kernel"] and abs(b["frac"] - b["achieved"] / b["peak"]) < 1e-12


def test_bench_line_per_launch_schedule():
    """--corr-group 0 --sweep-stream main: one correlation launch per 1024 pairs and the sweeps on the compute stream (the round-2 schedule);
    the same verification block must hold"""
    d = _run(extra_args=("--corr-group", "0", "--sweep-stream", "main", "--no-extra-legs", "--no-cpu-baseline", "--gicp-pairs", "0"))
    assert d["config"]["corr_launches_grouped"] == 1 and d["config"]["sweep_stream"] == "main"
    v = d["verify"]
    assert v["ok"] and v["sweep_mismatches"] == 0 and v["angle_mismatches"] == 0 and v["max_err_dist"] < 1e-5, v
    assert d["kernel_ms"]["corr"] > 0 and d["kernel_ms"]["sweep"] > 0


def test_bench_database_slots_across_steps():
    """6 launches in groups of 2: the grouped correlation launch and the side-stream sweeps read entries written DEPTH = 4 launches earlier,
    across the step boundary (two sets of slots written alternately, nothing copied)"""
    d = _run(extra_args=("--chunks", "6", "--fuse", "2", "--no-extra-legs", "--no-cpu-baseline", "--gicp-pairs", "0"))
    assert d["config"]["database_slots"].startswith("two sets") and d["config"]["corr_launches_grouped"] == 2 and d["config"]["launches_per_step"] == 6
    assert d["config"]["sweep_join"] == "lag"          # three batches of sweeps per step, the last one runs into the next step
    v = d["verify"]
    assert v["ok"] and v["sweep_mismatches"] == 0 and v["angle_mismatches"] == 0 and v["max_err_dist"] < 1e-5, v


def test_bench_line_two_kernel_step():
    """--fuse 0: the rasteriser and the Radon kernel as separate launches (the roofline block is then the rasteriser's)"""
    d = _run(extra_args=("--fuse", "0", "--no-extra-legs", "--no-cpu-baseline"))
    assert "fused_launches" not in d["config"] and "k_cart_lds" in d["roofline"]["kernel"]
    assert d["kernel_ms"]["bev"] > 0 and d["kernel_ms"]["radon"] > 0 and "bev_radon" not in d["kernel_ms"]
    assert abs(d["value"] - 32 * 3 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_collective_path_on_one_gpu():
    """MRS_BENCH_FORCE_DIST=1: the N > 1 code paths (process group, asynchronous RCCL collectives, max over ranks) with world size 1, both
    exchange designs: the default (database kept sharded, pre-planned all-to-all of the candidate rows, sharded top-1 sweep on a side stream)
    and --exchange allgather (fp16 replicas to every rank + owner re-scoring)."""
    d = _run({"MRS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["exchange"] == "fetch" and "all-to-all" in d["config"]["parallelism"]
    assert d["config"]["fused_grid"] == "persistent" and set(d["roofline"]["fused_grid_ms_per_launch"]) == {"per_pair", "persistent"}
    x = d["exchange"]
    assert x["design"] == "fetch" and x["fetch"]["rows_bytes_in_per_rank_per_launch"] == 0 and x["compute_stream_wait_ms_per_launch"] >= 0
    assert x["rescore"] is None and x["designs"]["sharded_topk_ms"] > 0 and x["fetch"]["launches_ahead"] >= 1
    d = _run({"MRS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534"}, ("--exchange", "allgather", "--no-extra-legs"))
    assert d["config"]["exchange"] == "allgather" and "all-gather" in d["config"]["parallelism"] and d["config"]["fused_grid"] == "per_pair"
    x = d["exchange"]
    assert x["allgather"]["bytes_in_per_rank_per_launch"] == 0 and x["compute_stream_wait_ms_per_launch"] >= 0      # world size 1: nothing inbound
    assert x["rescore"]["calls"] == 4 and x["rescore"]["rounds"] >= 4 and x["designs"]["sharded_topk_ms"] > 0


def test_bench_exchange_through_the_c_abi_on_one_gpu():
    """--exchange-impl cabi: the step functions' data-path exchanges (descriptor all-gather, planned candidate-row fetch, per-launch query
    gather, packed top-1 results) go through the product's own mrs_exchange_* (RCCL resolved at run time, the library's own communicator)
    on a communication stream instead of torch.distributed; world size 1 here (one GPU), both designs, checked by --verify-exchange."""
    for port, extra, design in ((29551, (), "fetch"), (29552, ("--exchange", "allgather"), "allgather"), (29553, ("--exchange", "allgather", "--replica", "f32"), "allgather")):
        d = _run({"MRS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)},
                 ("--exchange-impl", "cabi", "--verify-exchange", "--no-extra-legs", "--no-cpu-baseline", "--gicp-pairs", "0", *extra))
        x = d["exchange"]
        assert x["impl"] == "cabi" and x["design"] == design and "mrs_exchange" in x["impl_note"] and d["config"]["exchange_impl"] == "cabi"
        assert x["verify"]["ok"] and x["verify"]["checked"] == 32, x["verify"]
        c = d["_compact"]
        assert c["exchange"] == {"process_group": x["process_group"], "impl": "cabi", "verify_ok": True} and c["config"]["exchange_impl"] == "cabi"
    # the same flags on torch.distributed give the same verified answers (and say so)
    d = _run({"MRS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29554"},
             ("--verify-exchange", "--no-extra-legs", "--no-cpu-baseline", "--gicp-pairs", "0"))
    assert d["exchange"]["impl"] == "torch" and d["exchange"]["verify"]["ok"] and d["_compact"]["exchange"]["impl"] == "torch"


def _two_ranks(port, extra):
    env = dict(os.environ)
    env.update({"MRS_BENCH_BACKEND": "gloo", "MRS_BENCH_SHARE_GPU": "1"})
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "detail.json")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32",
               "--chunks", "3", "--gicp-pairs", "2", "--gicp-iters", "4", "--no-extra-legs", "--verify-exchange", "--detail-file", path, *extra]
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
        d = _strict_loads(open(path).read())
    d["_compact"] = _check_compact(lines[-1], d)
    assert d["_compact"]["exchange"]["process_group"]["world_size"] == 2
    return d


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N = 2 control flow on real kernels: two ranks on ONE GPU (RCCL refuses that, so the collectives run over gloo, staged through the
    host), max-over-ranks timing, one JSON line from rank 0.  Correctness of each exchange design is checked inside (--verify-exchange):
      fetch     : the fetched candidate rows are the owners' fp32 entries bit for bit, the scores equal those against the gathered exact
                  database, the sharded top-1 sweep equals a single-rank sweep over the gathered database (value and row);
      allgather : replica scores within 2e-3 of the exact ones, owner re-scoring, the top-k design."""
    d = _two_ranks(29541, ())
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["pairs_per_rank_per_step"] == 96 and d["config"]["exchange"] == "fetch"
    x = d["exchange"]
    assert 0 < x["fetch"]["rows_bytes_in_per_rank_per_launch"] <= 32 * 58560 and x["fetch"]["bytes_in_per_rank_per_launch"] < x["allgather"]["bytes_in_per_rank_per_launch"] * 2
    v = x["verify"]
    assert v["checked"] == 32 and v["remote_candidates"] > 0 and v["fetched_rows_bit_identical"] and v["sweep_value_equal"] and v["sweep_row_equal"]
    assert v["max_abs_dist_error"] < 1e-6 and v["angle_mismatches"] == 0
    assert d["gicp"]["pairs"] == 4
    d = _two_ranks(29542, ("--exchange", "allgather"))
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "allgather"
    x = d["exchange"]
    assert x["allgather"]["bytes_in_per_rank_per_launch"] == 32 * 29280 and x["rescore"]["calls"] == 3
    v = x["verify"]
    assert v["checked"] == 32 and v["remote_candidates"] > 0 and v["max_abs_dist_error"] < 2e-3
    assert v["angle_mismatches"] <= 3          # fp16 replicas may move the peak of a flat correlation (unrelated scans)


def test_bench_bare_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts the two ranks itself (torch.distributed.run, a free port),
    rank 0 prints the one JSON line with n_gpus == 2.  Test mode: both ranks on the one GPU over gloo.  Without that mode the same command on a
    one-GPU box must fail loudly instead of measuring one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    import tempfile
    td = tempfile.mkdtemp()
    path = os.path.join(td, "detail.json")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--chunks", "3",
            "--gicp-pairs", "0", "--no-extra-legs", "--exchange", "allgather", "--replica", "f32", "--verify-exchange", "--detail-file", path]
    p = subprocess.run(args, capture_output=True, text=True, env=dict(env, MRS_BENCH_BACKEND="gloo", MRS_BENCH_SHARE_GPU="1"), timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    d = _strict_loads(open(path).read())
    _check_compact([l for l in p.stdout.splitlines() if l.strip().startswith("{")][-1], d)
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "allgather" and d["config"]["pairs_per_rank_per_step"] == 96
    x = d["exchange"]
    assert x["process_group"] == {"backend": "gloo", "world_size": 2, "gpus_flag": 2, "devices_visible": x["process_group"]["devices_visible"]}
    # exact fp32 replicas: scores against the gathered database are the exact ones, nothing to re-score
    assert x["replica"] == "f32" and x["rescore"] is None and x["allgather"]["bytes_in_per_rank_per_launch"] == 32 * 58560
    v = x["verify"]
    assert v["checked"] == 32 and v["remote_candidates"] > 0 and v["max_abs_dist_error"] < 1e-6 and v["angle_mismatches"] == 0
    import torch
    if torch.cuda.device_count() < 2:
        p = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode != 0 and "needs 2 visible GPUs" in p.stderr and not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]

