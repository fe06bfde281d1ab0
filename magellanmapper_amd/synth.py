"""Seeded synthetic nuclei volumes: N(500, 50) background plus
isotropic Gaussian blobs (amplitude 40 000, sigma 3 px) combined by ``max``, clipped to
uint16.  Used by tests (small, NumPy, real-valued centres) and by ``bench.py`` (full size,
generated on the device slab by slab with integer centres so that every blob is the same
template and composition is one ``scatter_reduce(amax)`` per z-offset).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

BLOBS_PER_MVOX = 70.0


def make_volume(seed: int, shape: Sequence[int], n_blobs: Optional[int] = None, *, amp=40000.0,
                blob_sigma=3.0, bg_mean=500.0, bg_sd=50.0, margin=6, dtype=np.uint16,
                centres=None) -> np.ndarray:
    """Small host volume (full-grid evaluation); same generator as the golden fixtures."""
    rng = np.random.default_rng(seed)
    vol = rng.normal(bg_mean, bg_sd, shape)
    if centres is None:
        if n_blobs is None:
            n_blobs = int(round(BLOBS_PER_MVOX * np.prod(shape) / 1e6))
        lo = np.full(3, margin, dtype=float)
        hi = np.asarray(shape, dtype=float) - margin
        centres = rng.uniform(lo, hi, (n_blobs, 3))
    grids = np.meshgrid(*[np.arange(s, dtype=float) for s in shape], indexing="ij")
    for c in centres:
        d2 = sum((g - ci) ** 2 for g, ci in zip(grids, c))
        np.maximum(vol, amp * np.exp(-d2 / (2 * blob_sigma ** 2)), out=vol)
    vol = np.clip(vol, 0, 65535)
    if dtype == np.uint16:
        return vol.astype(np.uint16)
    if dtype == np.uint8:
        return (vol / 257.0).astype(np.uint8)
    return (vol / 65535.0).astype(dtype)


def make_volume_device(shape: Sequence[int], seed: int, device, *, density=BLOBS_PER_MVOX,
                       amp=40000.0, blob_sigma=3.0, bg_mean=500.0, bg_sd=50.0,
                       z_range: Optional[Tuple[int, int]] = None, slab=64):
    """uint16 ``(z, y, x)`` tensor on ``device``; with ``z_range`` only that z-slab of the
    same global volume (each rank of a multi-GPU run generates just its share: centres come
    from one seeded host draw, background noise from a per-slab seeded device generator)."""
    import torch
    nz, ny, nx = (int(v) for v in shape)
    z0, z1 = z_range if z_range is not None else (0, nz)
    rng = np.random.default_rng(seed)
    n_blobs = int(round(density * nz * ny * nx / 1e6))
    margin = 6
    cz = rng.integers(margin, max(margin + 1, nz - margin), n_blobs)
    cy = rng.integers(margin, max(margin + 1, ny - margin), n_blobs)
    cx = rng.integers(margin, max(margin + 1, nx - margin), n_blobs)
    rad = int(4 * blob_sigma + 0.5)
    off = torch.arange(-rad, rad + 1, device=device)
    g1 = torch.exp(-(off.double() ** 2) / (2 * blob_sigma ** 2))
    tmpl_yx = (amp * g1[:, None] * g1[None, :]).float()          # (2r+1, 2r+1)
    g1_host = g1.cpu().numpy()                                    # (one copy before the loops)
    out = torch.empty((z1 - z0, ny, nx), dtype=torch.uint16, device=device)
    # (no host <-> device synchronisation inside the loops: the blobs of a slab are selected on the host, where the
    #  centres live anyway, and a z-offset without blobs is an empty scatter.  Several ranks sharing one GPU -- the
    #  functional multi-rank tests -- otherwise pay a scheduling quantum per `bool(tensor)`: 478 s instead of 0.5 s for
    #  four ranks' slabs)
    for s0 in range((z0 // slab) * slab, z1, slab):   # global slab grid: ranks agree on the noise
        s1 = min(s0 + slab, nz)
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) * 1000003 + s0)
        vol = torch.empty((s1 - s0, ny, nx), dtype=torch.float32, device=device)
        vol.normal_(bg_mean, bg_sd, generator=gen)
        pick = np.flatnonzero((cz >= s0 - rad) & (cz < s1 + rad))
        if len(pick):
            bz = torch.from_numpy(cz[pick]).to(device)
            by = torch.from_numpy(cy[pick]).to(device)
            bx = torch.from_numpy(cx[pick]).to(device)
            flat = vol.view(-1)
            yy = by[:, None] + off[None, :]                        # (n, 2r+1)
            xx = bx[:, None] + off[None, :]
            oky = (yy >= 0) & (yy < ny)
            okx = (xx >= 0) & (xx < nx)
            for dz in range(-rad, rad + 1):
                if not np.any((cz[pick] + dz >= s0) & (cz[pick] + dz < s1)):
                    continue
                zz = bz + dz
                okz = (zz >= s0) & (zz < s1)
                wz = float(g1_host[dz + rad])
                idx = ((zz[:, None, None] - s0) * ny + yy[:, :, None]) * nx + xx[:, None, :]
                ok = okz[:, None, None] & oky[:, :, None] & okx[:, None, :]
                vals = (tmpl_yx * wz)[None].expand(idx.shape)
                flat.scatter_reduce_(0, idx[ok], vals[ok], reduce="amax")
        vol.clamp_(0, 65535)
        a, b = max(s0, z0), min(s1, z1)
        out[a - z0:b - z0] = vol[a - s0:b - s0].to(torch.int32).to(torch.uint16)
    return out
