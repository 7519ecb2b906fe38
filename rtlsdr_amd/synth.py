"""Synthetic u8 IQ streams shaped like an RTL2832U capture.

Each stream is an FM-modulated carrier placed at +fs/4 + delta_s (rtl_fm tunes
a quarter of the capture rate below the wanted channel and rotates by -fs/4,
reference src/rtl_fm.c:1425-1429 and :1336), quantised to unsigned bytes around
127.5 with a few LSB of uniform noise:

    phi[n] = phi[n-1] + 2*pi*(fs/4 + delta_s)/fs + (2*pi*dev/fs) * sin(2*pi*f_m*n/fs)
    I = clamp(floor(127.5 + A*cos(phi) + u)),  Q = clamp(floor(127.5 + A*sin(phi) + u'))

The numpy form is exact and seeded (tests, fixtures); the torch form generates
gigabytes directly in HBM for bench.py (same shape, its own RNG stream).
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5D2000


def fm_iq_u8(nstreams: int, nsamples: int, fs: float = 2.4e6, dev_hz: float = 75e3,
             amplitude: float = 60.0, noise_lsb: int = 3, seed: int = SEED_BASE,
             first_stream: int = 0, quiet: tuple | None = None) -> np.ndarray:
    """uint8 [nstreams, 2*nsamples] interleaved I,Q.

    quiet = (period, on) in samples: the carrier is keyed - samples n with n % period >= on are 127, 127 (no
    signal, no noise), so that callback buffers come out loud, silent and half of each: what a power squelch
    (src/rtl_fm.c:1204-1215) has to tell apart."""
    out = np.empty((nstreams, 2 * nsamples), dtype=np.uint8)
    n = np.arange(nsamples, dtype=np.float64)
    for s in range(nstreams):
        sid = first_stream + s
        rng = np.random.default_rng(seed + sid)
        delta = rng.uniform(-2e3, 2e3)
        f_m = 1e3 + sid
        phi0 = rng.uniform(0, 2 * np.pi)
        # integral of the instantaneous frequency, closed form
        phi = (phi0 + 2 * np.pi * (fs / 4 + delta) / fs * n
               - (dev_hz / f_m) * (np.cos(2 * np.pi * f_m * n / fs) - 1.0))
        u = rng.integers(-noise_lsb, noise_lsb + 1, size=(2, nsamples)) if noise_lsb > 0 \
            else np.zeros((2, nsamples), dtype=np.int64)
        i = np.floor(127.5 + amplitude * np.cos(phi) + u[0])
        q = np.floor(127.5 + amplitude * np.sin(phi) + u[1])
        out[s, 0::2] = np.clip(i, 0, 255).astype(np.uint8)
        out[s, 1::2] = np.clip(q, 0, 255).astype(np.uint8)
        if quiet is not None:
            off = (np.arange(nsamples) % int(quiet[0])) >= int(quiet[1])
            out[s, 0::2][off] = 127
            out[s, 1::2][off] = 127
    return out


def random_u8(nstreams: int, nbytes: int, seed: int = 1234) -> np.ndarray:
    """Full-scale adversarial bytes (integer stages only)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(nstreams, nbytes), dtype=np.uint8)


def fm_iq_u8_torch(nstreams: int, nsamples: int, device, fs: float = 2.4e6,
                   dev_hz: float = 75e3, amplitude: float = 60.0, noise_lsb: int = 3,
                   seed: int = SEED_BASE, first_stream: int = 0, chunk_streams: int = 16):
    """The same signal model generated on `device` (torch uint8 [nstreams, 2*nsamples])."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed + first_stream)
    out = torch.empty((nstreams, 2 * nsamples), dtype=torch.uint8, device=device)
    n = torch.arange(nsamples, dtype=torch.float64, device=device)
    for s0 in range(0, nstreams, chunk_streams):
        s1 = min(nstreams, s0 + chunk_streams)
        k = s1 - s0
        sid = torch.arange(first_stream + s0, first_stream + s1, dtype=torch.float64, device=device)
        delta = (torch.rand(k, generator=g, device=device, dtype=torch.float64) - 0.5) * 4e3
        phi0 = torch.rand(k, generator=g, device=device, dtype=torch.float64) * (2 * np.pi)
        f_m = 1e3 + sid
        w = (2 * np.pi * (fs / 4 + delta) / fs)[:, None]
        phi = phi0[:, None] + w * n[None, :] - (dev_hz / f_m)[:, None] * (
            torch.cos((2 * np.pi * f_m / fs)[:, None] * n[None, :]) - 1.0)
        phi = phi.to(torch.float32) if False else phi
        u = torch.randint(-noise_lsb, noise_lsb + 1, (2, k, nsamples), generator=g, device=device,
                          dtype=torch.int16).to(torch.float32) if noise_lsb > 0 else \
            torch.zeros((2, k, nsamples), device=device)
        ph32 = torch.remainder(phi, 2 * np.pi).to(torch.float32)
        i = torch.floor(127.5 + amplitude * torch.cos(ph32) + u[0]).clamp_(0, 255)
        q = torch.floor(127.5 + amplitude * torch.sin(ph32) + u[1]).clamp_(0, 255)
        blk = out[s0:s1].view(k, nsamples, 2)
        blk[:, :, 0] = i.to(torch.uint8)
        blk[:, :, 1] = q.to(torch.uint8)
        del phi, ph32, u, i, q
    return out
