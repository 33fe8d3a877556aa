"""oracle/corr_oracle.py -- TEST INFRASTRUCTURE ONLY (not product code).

CPU restatement (torch CPU float32, the library the reference itself calls) of the
descriptor / correlation maths of the LoopDetection nodes.  Each function cites the
reference lines it follows under /root/reference/LoopDetection/src/.

PARITY UNPINNED inside the reference: no test, golden vector or fixture in the reference
tree pins any of these functions; RING_ros/util.py cannot be
imported here (top-level imports of voxelocc, voxelfeat, torch_radon, torchvision, skimage).
The restatement is cross-checked against an independent numpy.fft / direct-sum formulation in
tests/test_oracle_corr.py.
"""
import math

import numpy as np
import torch

NUM_RING = 120      # RING_ros/config.py:7-8
NUM_SECTOR = 120


def _normalize(t):
    """torchvision fn.normalize(t, mean=t.mean(), std=t.std()) == (t - mean) / std with the
    unbiased std (util.py:197, 339-340, 429-430)."""
    return (t - t.mean()) / t.std()


# ------------------------------------------------------------------------------ RING (R2, C1)
def ring_normalize(sino):
    """util.py:197.  sino [C,A,D] float32."""
    return _normalize(torch.as_tensor(sino, dtype=torch.float32))


def tiring_from_sinogram(sino):
    """util.py:197-198: normalise, then FFT along the angle axis (dim=-2), ortho."""
    return torch.fft.fft2(ring_normalize(sino), dim=-2, norm="ortho")


def fast_corr(a, b, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """util.py:362-374.  a, b complex64 [C,A,D] (TIRING).  Returns (dist float32, angle int)."""
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    corr = torch.fft.ifft2(a * b.conj(), dim=-2, norm="ortho")
    corr = torch.sqrt(corr.real ** 2 + corr.imag ** 2)
    corr = torch.sum(corr, dim=0)
    corr = torch.sum(corr, dim=-1).view(-1)
    corr = torch.fft.fftshift(corr)
    dist = 1 - torch.max(corr) / (0.15 * num_ring * num_sector)
    angle = -torch.argmax(corr) + num_ring // 2
    return dist.numpy(), int(angle), corr.numpy()


# ---------------------------------------------------------------------------- RING++ (R2, C2)
def forward_row_fft(x):
    """util.py:295-300: |FFT along the detector axis| (ortho)."""
    m = torch.fft.fft2(torch.as_tensor(x, dtype=torch.float32), dim=-1, norm="ortho")
    return torch.sqrt(m.real ** 2 + m.imag ** 2), m


def fast_corr_ringplusplus(a, b, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """util.py:337-358.  a, b float32 [C,A,D]."""
    a = _normalize(torch.as_tensor(a, dtype=torch.float32))
    b = _normalize(torch.as_tensor(b, dtype=torch.float32))
    a_fft = torch.fft.fft2(a, dim=-2, norm="ortho")
    b_fft = torch.fft.fft2(b, dim=-2, norm="ortho")
    corr = torch.fft.ifft2(a_fft * b_fft.conj(), dim=-2, norm="ortho")
    corr = torch.sqrt(corr.real ** 2 + corr.imag ** 2)
    corr = torch.sum(corr, dim=0)
    corr = torch.sum(corr, dim=-1).view(-1)
    corr = torch.fft.fftshift(corr)
    angle = num_ring // 2 - torch.argmax(corr)
    dist = 1 - torch.max(corr) / (0.15 * a.shape[0] * num_ring * num_sector)
    return dist.numpy(), int(angle), corr.numpy()


# ------------------------------------------------------------------------ translation (C3, C4)
def solve_overdetermined_svd(A, b):
    """util.py:488-506, method='svd', LITERAL.  Quirk: torch.svd returns V with A = U S V^T, so
    the pseudo-inverse is V S^+ U^T, but the reference multiplies by v.t().  Its result is
    therefore (V^2)^T w for the true least-squares solution w: correct only when LAPACK happens
    to return a symmetric V, otherwise rotated by about -2*rot_angle -- i.e. it depends on the
    SVD sign convention of the platform (CPU LAPACK vs GPU solver).  Kept for documentation;
    the product implements the intended pseudo-inverse (solve_overdetermined_pinv)."""
    Bn = A.size(0)
    A = A.view(Bn, -1)
    b = b.view(Bn, -1)
    u, s, v = torch.svd(A, some=False)
    s_new = torch.zeros(A.shape)
    for i in range(len(s)):
        s_new[i, i] = 1 / s[i]
    return v.t() @ s_new.t() @ u.t() @ b


def solve_overdetermined_pinv(A, b):
    """What util.py:488-506 intends (and what its method='pinv' branch computes)."""
    Bn = A.size(0)
    return torch.linalg.pinv(A.view(Bn, -1).double()).float() @ b.view(Bn, -1)


def row_shifts(query, positive):
    """The per-angle integer shifts of util.py:404-413.  query, positive float32 [C,H,W]."""
    query = torch.as_tensor(query, dtype=torch.float32)
    positive = torch.as_tensor(positive, dtype=torch.float32)
    _, H, W = query.shape
    b = torch.zeros(H)
    for i in range(H):
        qf = torch.fft.fft2(query[:, i, :], dim=-1, norm="ortho")
        pf = torch.fft.fft2(positive[:, i, :], dim=-1, norm="ortho")
        corr = torch.fft.ifft2(qf * pf.conj(), dim=-1, norm="ortho")
        corr = torch.sqrt(corr.imag ** 2 + corr.real ** 2)
        corr = torch.fft.fftshift(corr)
        corr = torch.sum(corr, dim=0)
        b[i] = W // 2 - torch.argmax(corr)
    return b


def solve_translation(query, positive, rot_angle, literal=True):
    """util.py:388-423.  Returns (x, y, error, shifts).  literal=True (default) is the reference's call, method='svd'
    (util.py:415) with its v.t() product as written (platform dependent); literal=False its 'pinv' branch."""
    query = torch.as_tensor(query, dtype=torch.float32)
    _, H, W = query.shape
    angles = torch.FloatTensor(np.linspace(0, 2 * np.pi, H).astype(np.float32)) + rot_angle
    A = torch.stack([torch.cos(angles), torch.sin(angles)], dim=1)
    b = row_shifts(query, positive)
    sol = solve_overdetermined_svd(A, b) if literal else solve_overdetermined_pinv(A, b)
    x, y = sol[0], sol[1]
    error = torch.norm(torch.matmul(A, torch.cat([x, y], dim=0)) - b)
    return x.numpy(), y.numpy(), error.numpy(), b.numpy()


def rotate_nearest(img, angle_deg):
    """torchvision.transforms.functional.rotate(img, angle) with its defaults (NEAREST,
    expand=False, centre = image centre, fill 0), as util.py:67-70 calls it.  Restated from
    torchvision's tensor path: inverse affine matrix for a rotation by -angle about the centre,
    affine_grid on pixel centres (align_corners=False), grid_sample(mode='nearest',
    padding_mode='zeros').  img float32 [C,H,W]."""
    img = torch.as_tensor(img, dtype=torch.float32)
    C, H, W = img.shape
    rot = math.radians(-angle_deg)   # torchvision: angle = -angle, then inverse matrix of RSS
    # inverse of R(rot) with unit scale and no shear: [[cos, sin],[ -sin, cos]] (about the centre)
    a, b = math.cos(rot), math.sin(rot)
    theta = torch.tensor([[a, b, 0.0], [-b, a, 0.0]], dtype=torch.float32)
    # base grid in pixel units relative to the centre (torchvision _gen_affine_grid)
    xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 - 0.5, W)
    ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 - 0.5, H)
    base = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W), torch.ones(H, W)], dim=-1)
    rescaled = theta.t() / torch.tensor([0.5 * W, 0.5 * H])
    grid = base.view(1, H * W, 3).bmm(rescaled.unsqueeze(0)).view(1, H, W, 2)
    out = torch.nn.functional.grid_sample(img[None], grid, mode="nearest", padding_mode="zeros",
                                          align_corners=False)
    return out[0]


def solve_translation_bev(a, b, num_ring=NUM_RING, num_sector=NUM_SECTOR):
    """util.py:427-450.  a, b float32 [C,H,W].  Returns (y, x, -max) like the reference."""
    a = _normalize(torch.as_tensor(a, dtype=torch.float32))
    b = _normalize(torch.as_tensor(b, dtype=torch.float32))
    a_fft = torch.fft.fft2(a, dim=(-2, -1), norm="ortho")
    b_fft = torch.fft.fft2(b, dim=(-2, -1), norm="ortho")
    corr = torch.fft.ifft2(a_fft * b_fft.conj(), dim=(-2, -1), norm="ortho")
    corr = torch.sqrt(corr.real ** 2 + corr.imag ** 2)
    corr = torch.sum(corr, dim=0)
    corr = torch.fft.fftshift(corr)
    nz = (corr == torch.max(corr)).nonzero()
    idx_x, idx_y = nz[0][0], nz[0][1]
    x = idx_x - num_sector // 2
    y = num_ring // 2 - idx_y
    return int(y), int(x), float(-torch.max(corr)), corr.numpy()


# ------------------------------------------------------------------------------ DiSCO (D1, D2)
def fftshift2d(x):
    """disco_ros/models/DiSCO.py:280-294 (roll every dim >= 1 by ceil(n/2))."""
    for dim in range(1, x.dim()):
        n = x.size(dim) // 2
        if x.size(dim) % 2 != 0:
            n += 1
        x = torch.cat([x.narrow(dim, n, x.size(dim) - n), x.narrow(dim, 0, n)], dim)
    return x


def disco_forward(bev, col=16):
    """DiSCO.forward with the UNet bypassed (disco_ros/models/DiSCO.py:315-334).
    bev float32 [B,H,R,S] -> (signature [B,4*col*col], spectrum complex [B,1,R,S])."""
    x = torch.as_tensor(bev, dtype=torch.float32)
    B, _, R, S = x.shape
    u = torch.sum(x, 1).unsqueeze(1)
    spec = torch.fft.fft2(u, norm="ortho")
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-15)
    mag = fftshift2d(mag).squeeze(1)
    sig = mag[:, (R // 2 - col):(R // 2 + col), (S // 2 - col):(S // 2 + col)].reshape(B, -1)
    return sig.numpy(), spec


def phase_corr(a, b, num_sector=NUM_SECTOR):
    """disco_ros/main.py:260-272.  a, b complex64 [1,1,R,S].  Returns (yaw bin, corr)."""
    corr = torch.fft.ifft2(torch.as_tensor(a) * torch.as_tensor(b).conj(), norm="ortho")
    corr = torch.sqrt(corr.imag ** 2 + corr.real ** 2 + 1e-15)
    corr = fftshift2d(corr).squeeze(1)
    angle = torch.argmax(corr) % num_sector
    return int(angle), corr.numpy()
