"""Generates tests/golden/ref_corr.npz by RUNNING THE REFERENCE's own Python in this container:

  RING_ros/util.py         generate_RING, generate_RINGplusplus, build_neighbors_NN, forward_row_fft, fast_corr,
                           fast_corr_RINGplusplus, calculate_row_shift, solve_translation (+ its literal
                           solve_overdetermined_linear_system, CPU LAPACK), solve_translation_bev
  disco_ros/models/DiSCO.py  DiSCO.forward / forward_fft / fftshift2d
  disco_ros/main.py          phase_corr (function extracted by ast: the module imports rospy)
  generate_bev_pointfeat_cython/test.py  calculate_features (the numpy twin of the CUDA feature kernel)

imported through tests/golden/ref_import.py (stand-ins for the native boundary modules run the CPU checkers of
oracle/, i.e. the reference-built rasterisers + the pinned Radon restatement; torchvision's normalize is
(t - mean) / std).  The loop-closure sequence of main_RING.py:133-178 (module needs rospy) is replayed statement by
statement on the reference's functions.

Inputs are deterministic and rebuilt by the tests from `inputs()` below: cloud A = the NCLT scan of
tests/golden/nclt_scan.npz, B = A moved by a known yaw / translation + seeded noise, C = a synthetic scan.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_ref_corr.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import  # noqa: E402

YAW_B = 2 * np.pi * 17 / 119          # 17 rows of the RING angle grid (linspace(0, 2 pi, 120), endpoint included)
SHIFT_B = (0.05, -0.03)               # in normalised units (x / 70 m)


def inputs():
    """Clouds A, B, C as float32 [n,3] in the reference's normalised coordinates (after load_pc_infer)."""
    from mr_slam_amd import synth
    A = np.load(os.path.join(HERE, "nclt_scan.npz"))["hits"].astype(np.float32)
    rng = np.random.default_rng(2024)
    c, s = np.cos(YAW_B), np.sin(YAW_B)
    xy = A[:, :2].astype(np.float64) @ np.array([[c, s], [-s, c]]) + np.asarray(SHIFT_B)
    B = np.concatenate([xy, A[:, 2:3].astype(np.float64)], 1) + rng.normal(0, 0.002, A.shape)
    B = B[(np.abs(B[:, 0]) < 0.999) & (np.abs(B[:, 1]) < 0.999) & (B[:, 2] > 0.001) & (B[:, 2] < 0.999)].astype(np.float32)
    C = synth.lidar_scan(3, 20000)
    return A, B, C


def rolled_noisy(t, rows, seed, sigma=0.05):
    """A second descriptor for pairwise tests without storing it: t rolled along the angle axis + seeded noise."""
    rng = np.random.default_rng(seed)
    return (np.roll(np.asarray(t, np.float32), rows, axis=-2) + sigma * rng.standard_normal(t.shape).astype(np.float32)).astype(np.float32)


def shifted_bev(bev, dy, dx):
    """bev [C,H,W] moved by whole cells with zero fill (a pure translation of the scene)."""
    out = np.zeros_like(bev)
    H, W = bev.shape[-2:]
    ys, yd = (slice(0, H - dy), slice(dy, H)) if dy >= 0 else (slice(-dy, H), slice(0, H + dy))
    xs, xd = (slice(0, W - dx), slice(dx, W)) if dx >= 0 else (slice(-dx, W), slice(0, W + dx))
    out[..., yd, xd] = bev[..., ys, xs]
    return out


def main():
    A, B, C = inputs()
    rec = {"n_A": np.array([A.shape[0]]), "n_B": np.array([B.shape[0]]), "sum_B": np.array([B.astype(np.float64).sum()])}
    with ref_import.reference_modules("oracle") as ref:
        u = ref.util
        dev = u.device
        # ---------------------------------------------------------------- RING (R2, C1, C3): generate_RING + fast_corr
        ring = {}
        for name, pc in (("A", A), ("B", B), ("C", C)):
            bev, RING, TIRING = u.generate_RING(pc)                        # util.py:174-200
            ring[name] = (bev, RING, TIRING)
        for name in ("A", "B"):
            rec[f"ring_bev_occ_{name}"] = np.flatnonzero(ring[name][0].reshape(-1)).astype(np.int32)
            rec[f"ring_bev_val_{name}"] = ring[name][0].reshape(-1)[rec[f"ring_bev_occ_{name}"]]
            rec[f"ring_RING_{name}"] = ring[name][1].numpy()
            rec[f"ring_TIRING_{name}"] = ring[name][2].numpy()
        for a, b in (("A", "B"), ("A", "C"), ("B", "C"), ("A", "A"), ("B", "A")):
            dist, angle = u.fast_corr(ring[a][2], ring[b][2])              # util.py:362-374
            rec[f"fast_corr_{a}{b}"] = np.array([float(dist), float(angle)])
            print("fast_corr", a, b, float(dist), int(angle))
        # multi-channel fast_corr (no channel factor in its denominator, util.py:369)
        T2 = torch.cat([ring["A"][2], ring["C"][2]], 0); U2 = torch.cat([ring["B"][2], ring["C"][2]], 0)
        dist, angle = u.fast_corr(T2, U2)
        rec["fast_corr_2ch"] = np.array([float(dist), float(angle)])
        # main_RING.py:147-178 on (current = A, matched = B), statement by statement
        captured = {}
        orig = u.solve_overdetermined_linear_system

        def spy(Amat, b, method="pinv"):
            captured.setdefault("A", []).append(Amat.clone().numpy()); captured.setdefault("b", []).append(b.clone().numpy())
            return orig(Amat, b, method=method)
        u.solve_overdetermined_linear_system = spy
        cfg = u.cfg
        dist, angle_matched = u.fast_corr(ring["A"][2], ring["B"][2])
        angle_matched = int(angle_matched)
        angle_matched_extra = angle_matched - cfg.num_ring // 2
        angle_matched_rad = angle_matched * 2 * np.pi / cfg.num_ring
        angle_matched_extra_rad = angle_matched_extra * 2 * np.pi / cfg.num_ring
        row_shift = u.calculate_row_shift(angle_matched)
        row_shift_extra = u.calculate_row_shift(angle_matched_extra)
        RING_matched = ring["B"][1]
        RING_matched_shifted = torch.roll(RING_matched, row_shift, dims=1)
        RING_matched_shifted_extra = torch.roll(RING_matched, row_shift_extra, dims=1)
        x, y, error = u.solve_translation(ring["A"][1], RING_matched_shifted, angle_matched_rad, dev)
        x_e, y_e, error_e = u.solve_translation(ring["A"][1], RING_matched_shifted_extra, angle_matched_extra_rad, dev)
        u.solve_overdetermined_linear_system = orig
        rec["loop_AB"] = np.array([angle_matched, row_shift, row_shift_extra, angle_matched_rad, angle_matched_extra_rad])
        rec["solve_translation_AB"] = np.array([np.ravel(x)[0], np.ravel(y)[0], np.ravel(error)[0]], np.float32)
        rec["solve_translation_AB_extra"] = np.array([np.ravel(x_e)[0], np.ravel(y_e)[0], np.ravel(error_e)[0]], np.float32)
        rec["solve_translation_A_mat"] = np.stack(captured["A"]); rec["solve_translation_b"] = np.stack(captured["b"]).reshape(2, -1)
        # what the literal v.t() product should have been (pinv branch of the same reference function)
        for i, tag in enumerate(("", "_extra")):
            sol = orig(torch.from_numpy(captured["A"][i]), torch.from_numpy(captured["b"][i]), method="pinv")
            rec[f"solve_translation_AB{tag}_pinv"] = sol.numpy().reshape(-1)
        print("solve_translation literal", x, y, error, "| extra", x_e, y_e, error_e, "| pinv", rec["solve_translation_AB_pinv"])
        # ------------------------------------------------- RING++ (N1, A5, R2, C2, C4): generate_RINGplusplus + friends
        nbr = {}
        orig_nn = u.build_neighbors_NN

        def spy_nn(pc, k):
            out = orig_nn(pc, k); nbr["last"] = out; return out
        u.build_neighbors_NN = spy_nn
        bevA, RINGppA, TIRINGppA = u.generate_RINGplusplus(A)               # util.py:204-250
        k_indices, _, _, k_eigens, k_vectors = nbr["last"]
        u.build_neighbors_NN = orig_nn
        rec["pp_bev_nz_A"] = np.flatnonzero(bevA.numpy().reshape(-1)).astype(np.int32)
        rec["pp_bev_val_A"] = bevA.numpy().reshape(-1)[rec["pp_bev_nz_A"]]
        rec["pp_RING_A"] = RINGppA.numpy()
        rec["pp_TIRING_A"] = TIRINGppA.numpy()
        out, _ = u.forward_row_fft(RINGppA)                                 # util.py:295-300
        assert torch.equal(out, TIRINGppA)
        Tb = torch.from_numpy(rolled_noisy(TIRINGppA.numpy(), 23, seed=5))
        for tag, (a, b) in (("AA", (TIRINGppA, TIRINGppA)), ("Arolled", (TIRINGppA, Tb)), ("rolledA", (Tb, TIRINGppA))):
            dist, angle = u.fast_corr_RINGplusplus(a, b)                    # util.py:337-358
            rec[f"fast_corr_pp_{tag}"] = np.array([float(dist), float(angle)])
            print("fast_corr_RINGplusplus", tag, float(dist), int(angle))
        bevA_np = bevA.numpy()
        for tag, (dy, dx) in (("m7p11", (-7, 11)), ("p3m20", (3, -20)), ("zero", (0, 0))):
            yy, xx, neg = u.solve_translation_bev(torch.from_numpy(shifted_bev(bevA_np, dy, dx)), bevA)   # util.py:427-450
            rec[f"solve_translation_bev_{tag}"] = np.array([float(yy), float(xx), float(neg)])
            print("solve_translation_bev", tag, (dy, dx), "->", int(yy), int(xx), float(neg))
        # N1: util.build_neighbors_NN (sklearn kd-tree + numpy eig) and the feature kernel on every 4th point
        sel = np.arange(0, A.shape[0], 4)
        rec["pf_sel"] = sel.astype(np.int32)
        rec["pf_knn"] = np.asarray(k_indices)[sel].astype(np.int16)
        rec["pf_eigens"] = np.asarray(k_eigens, np.float32)[sel]
        from oracle import pyoracle as O
        feats = O.ref_point_features(A, np.asarray(k_indices, np.int32), np.asarray(k_eigens, np.float32))
        rec["pf_features_kernel"] = feats[sel]                               # reference kernel.cu:16-104 on the host
        import math
        from sklearn.neighbors import NearestNeighbors
        # test.py's own chain (kd-tree kNN -> np.linalg.eig eigenvalues + vectors -> calculate_features), functions only:
        # the module imports voxelfeat / knn_cuda / skimage at the top
        tw = ref.functions_of(os.path.join(ref_import.REF, "LoopDetection", "generate_bev_pointfeat_cython", "test.py"),
                              ["calculate_features", "calculate_entropy_array", "covariation_eigenvalue", "build_neighbors_NN"],
                              {"np": np, "math": math, "NearestNeighbors": NearestNeighbors, "print": lambda *a, **k: None})
        t_idx, _, _, t_eig, t_vec = tw["build_neighbors_NN"](A.astype(np.float32), 30)
        twin = np.stack([tw["calculate_features"](A.astype(np.float32), t_idx[i], t_eig[i], t_vec[i])[0] for i in sel])
        rec["pf_twin_eigens"] = np.asarray(t_eig, np.float32)[sel]
        rec["pf_features_numpy_twin"] = twin.astype(np.float32)             # test.py:68-98
        # ------------------------------------------------------------------------- DiSCO (D1, D2)
        d = ref.disco
        from oracle import pyoracle as O2
        dcfg = d.cfg
        spec = {}
        for name, pc in (("A", A), ("B", B)):
            soa = np.ascontiguousarray(pc.T).reshape(-1)
            bev = O2.ref_bev_polar(soa, 1, 1, dcfg.num_ring, dcfg.num_sector, 20, 1).reshape(-1, 3)[:, 2]
            x = torch.from_numpy(bev.reshape(1, 20, dcfg.num_ring, dcfg.num_sector).copy())
            net = d.DiSCO(output_dim=1024) if name == "A" else net
            sig, out, fft_result, unet_out = net.forward(x)                 # DiSCO.py:315-334
            spec[name] = fft_result
            rec[f"disco_sig_{name}"] = sig.detach().numpy()[0]
            rec[f"disco_fft_{name}"] = fft_result.detach().numpy()
        pc_ns = ref.functions_of(os.path.join(ref_import.DISCO_ROS, "main.py"), ["phase_corr", "fftshift2d", "roll_n"],
                                 {"torch": torch, "np": np, "cfg": dcfg})
        for a, b in (("A", "B"), ("B", "A"), ("A", "A")):
            yaw, corr = pc_ns["phase_corr"](spec[a], spec[b], torch.device("cpu"), None)   # main.py:260-272
            rec[f"phase_corr_{a}{b}"] = np.array([int(yaw), float(corr.max())])
            print("phase_corr", a, b, int(yaw))
    path = os.path.join(HERE, "ref_corr.npz")
    np.savez_compressed(path, **rec)
    print(os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
