"""The node-shaped twin (mr_slam_amd/node.py + C ABI mrs_loopdb_*): the device-resident descriptor lists and `detect_loop_icp` with the
reference's signature, checked against (a) the pairwise / sweep kernels bit for bit and (b) the reference's OWN `detect_loop_icp`
(RING_ros/main_RING.py:126-238, main_RINGplusplus.py:126-236, disco_ros/main.py:276-321), extracted from the reference file and executed on
the drop-in modules, on the same 1 000-entry candidate lists."""
import io
import math
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402

pytestmark = pytest.mark.gpu


def _norm_sinograms(n, seed):
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    return ring.normalize(torch.randn((n, 1, 120, 120), device="cuda:0", generator=g))[:, 0].contiguous()


def test_loopdb_ring_equals_the_sweep_bit_for_bit_through_growth_and_all_forms():
    import torch
    from mr_slam_amd import node, ring
    N = 700
    norm = _norm_sinograms(N + 1, 5)
    full = ring.fft_angle(norm)                               # what generate_RING hands the node: complex64 [.,120,120] (util.py:198)
    half = full[:, :61].contiguous()                          # the rows the database keeps
    db = node.LoopDatabase("ring", capacity=4)                # 4 -> 256 -> 512 -> 1024: three reallocations on the way
    for i in range(N):
        if i % 3 == 0:
            db.append(full[i:i + 1].cpu())                    # host tensor, the node's own object
        elif i % 3 == 1:
            db.append(full[i:i + 1])                          # the same on the device
        else:
            db.append(half[i:i + 1])                          # product form
        if i in (3, 4, 255, 256, 600):
            assert len(db) == i + 1
    q_full = full[N:N + 1]
    want_d, want_a = ring.corr_sweep_fft(half[N:N + 1], half[:N])
    for q in (q_full.cpu(), q_full, half[N:N + 1], q_full.cpu().numpy()):
        idx, d, a, alld, alla = db.query(q, 0.9, want_all=True)
        assert np.array_equal(alld, want_d.cpu().numpy()[0]) and np.array_equal(alla, want_a.cpu().numpy()[0])
        sel = np.nonzero(alld < np.float32(0.9))[0]
        assert np.array_equal(idx, sel) and np.array_equal(d, alld[sel]) and np.array_equal(a, alla[sel])
    thr = float(np.sort(want_d.cpu().numpy()[0])[10])         # a threshold that keeps 10 entries
    idx, d, a = db.query(q_full.cpu(), thr)
    assert len(idx) == 10 and np.all(np.diff(idx) > 0)
    # extend with a batch of device half spectra (a batch producer) and an empty database
    db2 = node.LoopDatabase("ring", capacity=8)
    assert db2.query(q_full, 0.5)[0].size == 0
    db2.extend_spectra(half[:300])
    db2.extend_spectra(half[300:N])
    assert len(db2) == N
    _, _, _, alld2, alla2 = db2.query(q_full, 0.5, want_all=True)
    assert np.array_equal(alld2, want_d.cpu().numpy()[0]) and np.array_equal(alla2, want_a.cpu().numpy()[0])


def test_tiled_sweep_and_dma_kernels_equal_the_register_staged_kernel():
    import torch
    from mr_slam_amd import ring
    for n in (1, 2, 7, 255, 1031):
        norm = _norm_sinograms(n + 1, 100 + n)
        spec = ring.half_spectrum(norm)
        q, dbs = spec[n:n + 1].contiguous(), spec[:n].contiguous()
        want_d, want_a = ring.corr_sweep_fft(q.repeat(2, 1, 1), dbs)        # two queries: the register-staged kernel (k_ring_corr_fft)
        d1, a1 = ring.corr_sweep_fft(q, dbs)                               # one query: the LDS-DMA kernel on the row layout
        d2, a2 = ring.corr_sweep_fft_tiled(q, ring.spec_to_tiled(dbs))     # and on DMA-tiled entries
        for d, a in ((d1[0], a1[0]), (d2, a2)):
            assert torch.equal(d, want_d[0]) and torch.equal(a, want_a[0]), n
        assert torch.equal(want_d[0], want_d[1])


def test_loopdb_ringpp_equals_fast_corr_ringplusplus():
    import torch
    from mr_slam_amd import node, ring
    N, C = 40, 6
    g = torch.Generator(device="cuda:0").manual_seed(9)
    tiring = torch.rand((N + 1, C, 120, 120), device="cuda:0", generator=g) * 3.0      # |row FFT| magnitudes: non-negative
    db = node.LoopDatabase("ringpp", channels=C, capacity=8)
    for i in range(N):
        db.append(tiring[i].cpu() if i % 2 else tiring[i])
    idx, d, a, alld, alla = db.query(tiring[N].cpu(), 2.0, want_all=True)
    assert len(idx) == N
    spec = ring.half_spectrum(ring.normalize(tiring))                                   # joint normalisation per entry + half spectrum
    want_d, want_a = ring.corr_sweep_fft(spec[N:N + 1], spec[:N].contiguous())
    assert np.array_equal(alld, want_d.cpu().numpy()[0]) and np.array_equal(alla, want_a.cpu().numpy()[0])
    for i in (0, 7, N - 1):                                                             # and the reference-named pairwise mirror
        dd, aa = ring.fast_corr_RINGplusplus(tiring[N], tiring[i])
        assert int(aa) == int(alla[i]) and abs(float(dd) - float(alld[i])) < 1e-5


def test_disco_database_two_launch_query_equals_knn_and_phase_corr():
    import torch
    from mr_slam_amd import disco, node
    N = 300
    g = torch.Generator(device="cuda:0").manual_seed(3)
    occ = (torch.rand((N + 1, 20, 40, 120), device="cuda:0", generator=g) < 0.05).float()
    sig, spec = disco.disco_from_bev(occ)                     # [N+1,1024], complex64 [N+1,1,40,120]
    db = node.DiscoDatabase(capacity=16)
    assert db.query(sig[N].cpu().numpy(), spec[N].cpu())[0] == -1
    for i in range(N):
        if i % 2:
            db.append(sig[i].cpu().numpy(), spec[i].cpu())    # the node's own objects (numpy signature, host spectrum)
        else:
            db.append(sig[i], spec[i])
    for qi in (N, 17):                                        # an unseen query and a stored one (distance 0, yaw bin of the self-correlation)
        idx, d2, yaw = db.query(sig[qi].cpu().numpy(), spec[qi].cpu())
        widx, wd2 = disco.signature_knn(sig[qi:qi + 1], sig[:N].contiguous(), 1)
        assert idx == int(widx[0, 0]) and abs(d2 - float(wd2[0, 0])) <= 1e-5 * max(1.0, float(wd2[0, 0]))
        wyaw = disco.phase_corr(spec[idx:idx + 1], spec[qi:qi + 1])     # phase_corr(FFT_candidate, fft_current): rocFFT path
        assert yaw == int(wyaw[0])
        idx_dev, _, yaw_dev = db.query(sig[qi], spec[qi])
        assert (idx_dev, yaw_dev) == (idx, yaw)
    assert db.query(sig[17], spec[17])[0] == 17


# --------------------------------------------------------------------------------------------------------------------------------------
# the reference's own detect_loop_icp on the same lists
class _Obj:
    pass


def _pose_msg():
    p = _Obj()
    p.position, p.orientation = _Obj(), _Obj()
    return p


def _translation_from_matrix(m):
    return np.array(m, copy=False)[:3, 3].copy()


def _quaternion_from_matrix(matrix):
    """x, y, z, w of a homogeneous rotation (stand-in for tf.transformations.quaternion_from_matrix, which the image does not have; the
    SAME function serves the reference function and the twin, so it cannot hide a difference between them)"""
    M = np.array(matrix, dtype=np.float64)[:4, :4]
    q = np.empty(4)
    t = np.trace(M)
    if t > M[3, 3]:
        q[3], q[2], q[1], q[0] = t, M[1, 0] - M[0, 1], M[0, 2] - M[2, 0], M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i], q[j], q[k], q[3] = t, M[i, j] + M[j, i], M[k, i] + M[i, k], M[k, j] - M[j, k]
    return q * (0.5 / math.sqrt(t * M[3, 3]))


class _Pub:
    def __init__(self):
        self.msgs = []

    def publish(self, m):
        self.msgs.append(m)


class _Loops:
    def __init__(self):
        self.Loops = []


def _node_namespace(util_module, cfg, extra):
    ns = {k: v for k, v in vars(util_module).items() if not k.startswith("__")}        # `from util import *`
    from mr_slam_amd.compat import pygicp
    ns.update(dict(np=np, cfg=cfg, pygicp=pygicp, time=__import__("time"), torch=__import__("torch"), Pose=_pose_msg, Loop=_Obj, Loops=_Loops,
                   translation_from_matrix=_translation_from_matrix, quaternion_from_matrix=_quaternion_from_matrix))
    ns.update(extra)
    return ns


def _scene_variants(n_base, n_total, n_points, seed):
    """n_total metric clouds: n_base ray-cast scenes, each candidate a yawed / shifted / jittered copy of one of them"""
    from mr_slam_amd import synth
    rng = np.random.default_rng(seed)
    base = [synth.lidar_scan(900 + b, n_points, metric=True).astype(np.float64) for b in range(n_base)]
    out = []
    for i in range(n_total):
        p = base[i % n_base]
        yaw = rng.uniform(0, 2 * np.pi)
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        q = p @ R.T + np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), 0.0]) + rng.normal(0, 0.02, p.shape)
        out.append(q)
    return base, out


@pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")
def test_detect_loop_icp_ring_twin_equals_the_reference_function_on_1000_candidates():
    import torch
    from mr_slam_amd import node
    path = os.path.join(ref_import.RING_ROS, "main_RING.py")
    if not os.path.exists(path):
        pytest.skip("main_RING.py not staged")
    N = 1000
    base, clouds = _scene_variants(6, N + 2, 5000, 1)
    with ref_import.reference_modules("dropin") as ref:
        u = ref.util
        cfg = sys.modules["config"]
        desc = [u.generate_RING(u.load_pc_infer(pc)) for pc in clouds]                 # util.py:174-200 on the drop-in modules
        PC, RING, TIRING = clouds[:N], [d[1] for d in desc[:N]], [d[2] for d in desc[:N]]
        ref_d = np.array([float(u.fast_corr(desc[N][2].to(u.device), t.to(u.device))[0]) for t in TIRING[:200]], np.float32)   # the loop as the node runs it
        ref_a = np.array([int(u.fast_corr(desc[N][2].to(u.device), t.to(u.device))[1]) for t in TIRING[:200]])
        u.device = torch.device("cpu")      # util's own global too: solve_overdetermined_linear_system builds s_new on it (util.py:500)
        results = {}
        for which in ("reference", "twin"):
            f, pub = io.StringIO(), _Pub()
            # `device` = cpu for the reference function: its solve_translation (util.py:388-423) then runs torch.svd through LAPACK, the
            # convention ring.solve_translation reproduces (its `v.t() @ ...` product depends on the SVD backend's sign / order choices:
            # mr_slam_amd/ring.py::solve_translation); the candidate loop itself is then the reference's own torch arithmetic on host tensors
            ns = _node_namespace(u, cfg, dict(f=f, pub=pub, device=torch.device("cpu")))
            ns = ref_import.reference_modules.functions_of(path, ["get_pose_msg_from_homo_matrix", "fast_gicp", "detect_loop_icp"], ns)
            fn = ns["detect_loop_icp"] if which == "reference" else node.bind_detect_loop_icp(ns, "ring")
            log = io.StringIO()
            old = sys.stdout
            sys.stdout = log
            try:
                for qi, robot in ((N, 0), (N + 1, 1)):                                   # two new scans of robot 0 / 1 against robot 2's list
                    fn(robot, 7 + qi, clouds[qi], desc[qi][1], desc[qi][2], 2, PC, RING, TIRING)
            finally:
                sys.stdout = old
            results[which] = (f.getvalue(), [(m.Loops[0].id0, m.Loops[0].id1) for m in pub.msgs], log.getvalue())
        # the candidate lists themselves: the reference's loop (util.fast_corr per entry, above) against one query of the device twin
        q = desc[N][2]
    _, _, _, alld, alla = node.twin_of(TIRING, "ring").query(q, cfg.dist_threshold, want_all=True)
    assert np.array_equal(alla[:200], ref_a) and np.abs(alld[:200] - ref_d).max() < 1e-5
    rf, tf = results["reference"], results["twin"]
    assert rf[0] == tf[0] and rf[0].count("\n") >= 1, "loopinfo.txt lines differ or no loop was accepted"
    assert rf[1] == tf[1]

    def stable(text):                                                                    # drop the timing line and the (1e-6-level) printed distances
        keep = [ln for ln in text.splitlines() if not ln.startswith(("ICP processed time", "Top 1 RING distance", "robotid:"))]
        return keep
    assert stable(rf[2]) == stable(tf[2])
    assert any(ln.startswith("Loop detected between id") for ln in stable(tf[2]))


@pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")
def test_detect_loop_icp_disco_twin_equals_the_reference_function():
    import torch
    from mr_slam_amd import bev, disco, node
    from sklearn.neighbors import KDTree
    path = os.path.join(ref_import.DISCO_ROS, "main.py")
    N = 400
    base, clouds = _scene_variants(5, N + 1, 5000, 2)
    sigs, ffts = [], []
    for pc in clouds:                                                                    # generate_DiSCO's two outputs per scan (main.py:84-90, 94-125)
        p = pc.astype(np.float32)
        keep = (np.abs(p[:, 0]) < 70.) & (np.abs(p[:, 1]) < 70.) & (p[:, 2] < 30.) & (p[:, 2] > 0.)    # load_pc_infer
        p = p[keep] / np.array([70., 70., 30.], np.float32)
        xyz, offs = bev.pack_scans([p], "cuda:0")
        s, fft = disco.disco_descriptors(xyz, offs)
        sigs.append(s[0].cpu().numpy())
        ffts.append(fft.cpu())                                                          # complex64 [1,1,40,120]
    with ref_import.reference_modules("dropin"):
        import importlib
        sys.path.insert(0, ref_import.DISCO_ROS)
        sys.modules.pop("config", None)
        cfg = importlib.import_module("config")
        sys.path.remove(ref_import.DISCO_ROS)
        cfg.num_sector = 120                                                             # main.py:484: the node's arguments overwrite the config
        results = {}
        for which in ("reference", "twin"):
            f, pub = io.StringIO(), _Pub()
            from mr_slam_amd.compat import pygicp
            ns = dict(np=np, torch=torch, cfg=cfg, pygicp=pygicp, time=__import__("time"), KDTree=KDTree, Pose=_pose_msg, Loop=_Obj, Loops=_Loops,
                      translation_from_matrix=_translation_from_matrix, quaternion_from_matrix=_quaternion_from_matrix, f=f, pub=pub,
                      device=torch.device("cuda:0"), corr2soft=None, yaw_diff_pc=[], robotid_to_key=__import__("mr_slam_amd.preprocess", fromlist=["x"]).robotid_to_key)
            ns = ref_import.reference_modules.functions_of(path, ["get_pose_msg_from_homo_matrix", "fast_gicp", "euler2rot", "getSE3", "roll_n", "fftshift2d",
                                                                  "phase_corr", "detect_loop_icp"], ns)
            fn = ns["detect_loop_icp"] if which == "reference" else node.bind_detect_loop_icp(ns, "disco")
            log = io.StringIO()
            old = sys.stdout
            sys.stdout = log
            try:
                fn(0, 11, clouds[N], sigs[N], ffts[N].to("cuda:0") if which == "reference" else ffts[N], 1, clouds[:N], sigs[:N],
                   [t.to("cuda:0") for t in ffts[:N]] if which == "reference" else ffts[:N])
            finally:
                sys.stdout = old
            results[which] = (f.getvalue(), [(m.Loops[0].id0, m.Loops[0].id1) for m in pub.msgs],
                              [ln for ln in log.getvalue().splitlines() if not ln.startswith("robotid:")])
    assert results["reference"][0] == results["twin"][0] and results["reference"][1] == results["twin"][1]
    assert results["reference"][2] == results["twin"][2]


@pytest.mark.skipif(not ref_import.available(), reason="reference Python files not present")
def test_detect_loop_icp_ringpp_twin_equals_the_reference_function():
    import torch
    from mr_slam_amd import node, ring
    path = os.path.join(ref_import.RING_ROS, "main_RINGplusplus.py")
    if not os.path.exists(path):
        pytest.skip("main_RINGplusplus.py not staged")
    N = 60
    base, clouds = _scene_variants(4, N + 1, 4000, 3)
    with ref_import.reference_modules("dropin") as ref:
        u = ref.util
        cfg = sys.modules["config"]
        cfg.dist_threshold = 0.41                                                       # main_RINGplusplus.py:397 (the node's argument default)
        desc = [ring.generate_RINGplusplus(u.load_pc_infer(pc)) for pc in clouds]      # (bev [6,R,S] device, RING cpu, TIRING cpu): util.py:204-250's outputs
        PC, BEV, TIRING = clouds[:N], [d[0] for d in desc[:N]], [d[2] for d in desc[:N]]
        results = {}
        for which in ("reference", "twin"):
            f, pub = io.StringIO(), _Pub()
            # rotate_bev is torchvision's rotate (util.py:67-70), absent from this image: both sides get the product's restatement
            ns = _node_namespace(u, cfg, dict(f=f, pub=pub, device=torch.device("cuda:0"), rotate_bev=ring.rotate_bev))
            ns = ref_import.reference_modules.functions_of(path, ["get_pose_msg_from_homo_matrix", "fast_gicp", "detect_loop_icp"], ns)
            fn = ns["detect_loop_icp"] if which == "reference" else node.bind_detect_loop_icp(ns, "ringpp")
            log = io.StringIO()
            old = sys.stdout
            sys.stdout = log
            try:
                if which == "reference":
                    fn(0, 5, clouds[N], desc[N][0], desc[N][2].to(u.device), 1, PC, BEV, [t.to(u.device) for t in TIRING])
                else:
                    fn(0, 5, clouds[N], desc[N][0], desc[N][2], 1, PC, BEV, TIRING)
            finally:
                sys.stdout = old
            results[which] = (f.getvalue(), [(m.Loops[0].id0, m.Loops[0].id1) for m in pub.msgs],
                              [ln for ln in log.getvalue().splitlines() if not ln.startswith(("ICP processed time", "Top 1 TIRING distance", "robotid:"))])
    assert results["reference"][0] == results["twin"][0] and results["reference"][1] == results["twin"][1]
    assert results["reference"][2] == results["twin"][2]
    assert any(ln.startswith(("Loop detected", "No loop detected")) for ln in results["twin"][2])


def test_exchange_through_the_c_abi_on_one_rank():
    """mrs_exchange_* (RCCL behind the C ABI; include/mrslam_hip.h) at world size 1: the all-gather of a descriptor shard and the request-based
    row fetch return what a local copy / gather returns.  (Several ranks need several GPUs: the driver's SCALE run.)"""
    import torch
    from mr_slam_amd import ring, shard
    x = shard.Exchange(device=0)
    assert (x.world, x.rank) == (1, 0)
    spec = ring.half_spectrum(_norm_sinograms(48, 21))                         # [48,61,120] complex64 = 58 560 B entries
    assert torch.equal(torch.view_as_real(x.allgather(spec)), torch.view_as_real(spec))
    spec16 = ring.half_spectrum_f16(_norm_sinograms(48, 21), want_f32=False)[1]   # the fp16 replica format
    assert torch.equal(x.allgather(spec16), spec16)
    rows = torch.tensor([5, 0, 47, 5, 13, 46], device="cuda:0")
    got = x.fetch_rows(spec, rows)
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(spec[rows]))
    with pytest.raises(Exception):
        x.fetch_rows(spec, torch.tensor([48], device="cuda:0"))                # outside the database
    # the planned form (round 6): the request phase once, every fetch stream-ordered on a communication stream; two fetches of one plan
    # against different databases, back to back
    plan = x.fetch_plan(rows, 48)
    assert plan.n == 6 and plan.rows_from_peers == 0 and plan.bytes_in(58560) == 0
    comm = torch.cuda.Stream()
    spec2 = spec.flip(0).contiguous()
    w1, f1 = plan.fetch(spec, async_op=True, stream=comm)
    w1.wait(); out1 = f1().clone()
    w2, f2 = plan.fetch(spec2, async_op=True, stream=comm)
    w2.wait(); out2 = f2()
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(out1), torch.view_as_real(spec[rows])) and torch.equal(torch.view_as_real(out2), torch.view_as_real(spec2[rows]))
    assert torch.equal(torch.view_as_real(plan.fetch(spec)), torch.view_as_real(spec[rows]))
    buf = torch.empty((1, 48, 61, 120, 2), dtype=torch.float32, device="cuda:0")
    x.allgather_into(buf, torch.view_as_real(spec).contiguous(), stream=comm)
    torch.cuda.synchronize()
    assert torch.equal(buf[0], torch.view_as_real(spec))


def test_one_twin_queried_from_two_threads_while_a_third_appends():
    """callback1 and callback3 both score against TIRING2 on separate rospy threads (main_RING.py:284-288, 378-382), and ctypes
    drops the GIL during mrs_loopdb_query.  Every call owns its output arrays and takes n from the C side: two threads hammering ONE twin with
    different queries, while a third appends, must each get exactly their own serial answers over the entries their call scored."""
    import threading
    import torch
    from mr_slam_amd import node, ring
    N, EXTRA = 400, 200
    norm = _norm_sinograms(N + EXTRA + 2, 77)
    half = ring.half_spectrum(norm).contiguous()
    db = node.LoopDatabase("ring", capacity=8)
    db.extend_spectra(half[:N])
    qs = [half[N + EXTRA:N + EXTRA + 1], half[N + EXTRA + 1:N + EXTRA + 2]]
    full = [ring.corr_sweep_fft(q, half[:N + EXTRA]) for q in qs]           # scores over everything that will ever be in the list
    want = [(d.cpu().numpy()[0], a.cpu().numpy()[0]) for d, a in full]
    errors, counts = [], [0, 0]
    stop = threading.Event()

    def reader(t):
        try:
            wd, wa = want[t]
            while not stop.is_set():
                idx, d, a, alld, alla = db.query(qs[t], 0.7, want_all=True)
                n = alld.size
                assert N <= n <= N + EXTRA and alla.size == n
                assert np.array_equal(alld, wd[:n]) and np.array_equal(alla, wa[:n]), "another thread's scores"
                sel = np.nonzero(wd[:n] < np.float32(0.7))[0]
                assert np.array_equal(idx, sel) and np.array_equal(d, wd[sel]) and np.array_equal(a, wa[sel])
                counts[t] += 1
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e))
            stop.set()

    def writer():
        try:
            import time
            t_end = time.time() + 30.0
            while min(counts) < 1 and time.time() < t_end and not stop.is_set():      # both readers are under way before the list starts to grow
                time.sleep(0.001)
            with torch.cuda.stream(torch.cuda.Stream()):
                for i in range(N, N + EXTRA):
                    db.append(half[i:i + 1])
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e))
        finally:
            stop.set()

    th = [threading.Thread(target=reader, args=(0,)), threading.Thread(target=reader, args=(1,)), threading.Thread(target=writer)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors
    assert len(db) == N + EXTRA and min(counts) >= 1, counts


def test_query_multi_equals_single_queries_bit_for_bit():
    """mrs_loopdb_query_multi: Q new descriptors against the list in ONE sweep (one robot's scan against several lists / the three callbacks at
    once / BASELINE configs[3]) -- row q = the scores of query(descriptor q), for host TIRING tensors, device tensors, device half spectra, RING
    and RING++."""
    import torch
    from mr_slam_amd import node, ring
    N = 700
    norm = _norm_sinograms(N + 5, 31)
    half = ring.half_spectrum(norm).contiguous()
    full = torch.cat([half, half[:, 1:60].flip(1).conj()], 1).contiguous()
    db = node.LoopDatabase("ring", capacity=64)
    db.extend_spectra(half[:N])
    for nq in (1, 2, 3, 5):
        singles = [db.query(full[N + i:N + i + 1], 2.0, want_all=True) for i in range(nq)]
        for q in (full[N:N + nq].cpu(), full[N:N + nq], half[N:N + nq], [full[N + i].cpu() for i in range(nq)]):
            D, A = db.query_multi(q)
            assert D.shape == (nq, N)
            for i in range(nq):
                assert np.array_equal(D[i], singles[i][3]) and np.array_equal(A[i], singles[i][4]), (nq, i)
    C = 6
    g = torch.Generator(device="cuda:0").manual_seed(19)
    tiring = torch.rand((60 + 3, C, 120, 120), device="cuda:0", generator=g) * 3.0
    dbp = node.LoopDatabase("ringpp", channels=C, capacity=8)
    for i in range(60):
        dbp.append(tiring[i])
    D, A = dbp.query_multi(tiring[60:63].cpu())
    for i in range(3):
        _, _, _, alld, alla = dbp.query(tiring[60 + i], 9.0, want_all=True)
        assert np.array_equal(D[i], alld) and np.array_equal(A[i], alla)
    assert node.LoopDatabase("ring").query_multi(full[:2])[0].shape == (2, 0)


def test_unchanged_candidate_loop_uses_device_twins_of_the_host_tensors():
    """the node's loop AS WRITTEN (main_RING.py:133-134, fast_corr per stored CPU tensor) through the drop-in keeps a device
    twin per host tensor (ring.DeviceMirror): same numbers as before, no upload after a tensor's first visit, an in-place change to a cached
    tensor is seen, and a dropped tensor frees its slot."""
    import gc
    import torch
    from mr_slam_amd import ring, synth
    m = ring.device_mirror()
    m.clear()
    scans = [synth.lidar_scan(300 + i, 20000) for i in range(4)]
    descs = [ring.generate_RING(s) for s in scans]               # (bev, RING cpu, TIRING cpu): the twin is seeded by generate_RING itself
    assert len(m) == 4 and m.misses == 0
    TIRING = [d[2] for d in descs]
    cur = TIRING[3]
    h0 = m.hits
    got = [ring.fast_corr(cur, TIRING[i]) for i in range(3)]
    got0 = got[0]
    assert m.misses == 0 and m.hits == h0 + 6                     # nothing uploaded
    for i in range(3):                                            # the same numbers as the uncached path (numpy arguments are never cached)
        wd, wa = ring.fast_corr(cur.numpy(), TIRING[i].numpy())
        assert got[i][0] == wd and got[i][1] == wa and isinstance(got[i][0], np.float32)
    # a host tensor the mirror has never seen (e.g. loaded from disk) is uploaded once
    fresh = TIRING[0].clone()
    ring.fast_corr(cur, fresh); ring.fast_corr(cur, fresh)
    assert m.misses == 1 and len(m) == 5
    # in-place change of a cached tensor: torch bumps its version, the twin is rebuilt
    d_before = ring.fast_corr(cur, fresh)[0]
    fresh.mul_(0.5)
    d_after, _ = ring.fast_corr(cur, fresh)
    assert m.misses == 2 and d_after == ring.fast_corr(cur.numpy(), fresh.numpy())[0] and d_after != d_before
    # a dropped tensor frees its slot (and its device memory)
    b0 = m.bytes
    del fresh
    gc.collect()
    assert len(m) == 4 and m.bytes < b0
    # RING++: the normalised half spectrum is the twin
    C = 6
    g = torch.Generator().manual_seed(5)
    A, B = torch.rand((C, 120, 120), generator=g) * 3, torch.rand((C, 120, 120), generator=g) * 3
    want = ring.fast_corr_RINGplusplus(A.numpy(), B.numpy())
    mm = m.misses
    for _ in range(3):
        got = ring.fast_corr_RINGplusplus(A, B)
        assert got[0] == want[0] and got[1] == want[1]
    assert m.misses == mm + 2
    # the memory bound evicts the oldest entries
    small = ring.DeviceMirror(max_bytes=3 * 120 * 120 * 8)
    ts = [torch.zeros((1, 120, 120), dtype=torch.complex64) for _ in range(5)]
    for t in ts:
        small.get(t, lambda x: x.cuda())
    assert len(small) == 3 and small.bytes == 3 * 120 * 120 * 8
    # a spectrum that is NOT Hermitian along the angle axis takes the general kernel, cached or not, and want_corr returns the curve
    nh = torch.randn((1, 120, 120), dtype=torch.complex64, generator=g)
    assert ring._tiring_twin(nh, "cuda:0")[0] == "full" and ring._tiring_twin(TIRING[0], "cuda:0")[0] == "half"
    r1, r2 = ring.fast_corr(nh, TIRING[0]), ring.fast_corr(nh.numpy(), TIRING[0].numpy())
    assert r1[0] == r2[0] and r1[1] == r2[1]
    d3, a3, curve = ring.fast_corr(cur, TIRING[0], want_corr=True)
    assert curve.shape == (120,) and int(a3) == int(got0[1]) and abs(float(d3) - float(got0[0])) < 1e-5
    m.clear()
