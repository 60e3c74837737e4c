"""Deterministic synthetic image pairs (no RNG, no files) -- SURVEY.md §8(d).

P0: smooth texture moved by a slowly varying flow (converges in a few iterations).
P1: textured background + a foreground rectangle with its own motion (a motion discontinuity,
    more inner iterations).  `k` selects the batch variant used for BASELINE config 5.
All values are floor()-ed to integers in [0, 255] like an 8-bit image, returned as float64.
"""
import numpy as np


def _tex(x, y, phase=0.3):
    return (127.5 + 40 * np.sin(0.11 * x + phase) * np.cos(0.07 * y) + 30 * np.sin(0.031 * x + 0.043 * y)
            + 25 * np.cos(0.19 * y - 0.05 * x) + 20 * np.sin(0.37 * x) * np.sin(0.29 * y))


def _grid(nx, ny):
    i, j = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
    return i, j


def _flow(i, j):
    return 1.5 + 0.5 * np.sin(0.01 * i), -0.75 + 0.25 * np.cos(0.013 * j)


def pair_p0(nx, ny):
    i, j = _grid(nx, ny)
    fx, fy = _flow(i, j)
    return np.floor(_tex(j, i)), np.floor(_tex(j - fx, i - fy))


def pair_p1(nx, ny, k=0):
    i, j = _grid(nx, ny)
    fx, fy = _flow(i, j)
    phase = 0.3 + 0.1 * k
    mx, my = 4 + (k % 3), -3

    def tex_b(x, y):
        return _tex(x, y, phase) + 15 * np.sin(1.3 * x) * np.cos(1.1 * y)

    def tex_f(x, y):
        return 127.5 + 60 * np.sin(0.9 * x + 0.4 * y) + 50 * np.cos(0.7 * y - 0.3 * x)

    def fg(x, y):
        return (x >= nx // 4) & (x < nx // 2) & (y >= ny // 4) & (y < ny // 2)

    I0 = np.floor(np.where(fg(j, i), tex_f(j, i), tex_b(j, i)))
    I1 = np.floor(np.where(fg(j - mx, i - my), tex_f(j - mx, i - my), tex_b(j - fx, i - fy)))
    return I0, I1


def pair(name, nx, ny, k=0):
    if name == "P0":
        return pair_p0(nx, ny)
    if name == "P1":
        return pair_p1(nx, ny, k)
    raise ValueError(name)


def sequence(nx, ny, frames, k=0):
    """`frames` images of the P1 scene in uniform motion (frame t = background moved by t * flow, foreground
    rectangle moved by t * (mx, my)): input of the temporal Brox method.  Shape (frames, ny, nx)."""
    i, j = _grid(nx, ny)
    fx, fy = _flow(i, j)
    phase = 0.3 + 0.1 * k
    mx, my = 2 + (k % 3), -1.5

    def tex_b(x, y):
        return _tex(x, y, phase) + 15 * np.sin(1.3 * x) * np.cos(1.1 * y)

    def tex_f(x, y):
        return 127.5 + 60 * np.sin(0.9 * x + 0.4 * y) + 50 * np.cos(0.7 * y - 0.3 * x)

    def fg(x, y):
        return (x >= nx // 4) & (x < nx // 2) & (y >= ny // 4) & (y < ny // 2)

    out = np.empty((frames, ny, nx))
    for t in range(frames):
        out[t] = np.floor(np.where(fg(j - t * mx, i - t * my), tex_f(j - t * mx, i - t * my),
                                   tex_b(j - 0.6 * t * fx, i - 0.6 * t * fy)))
    return out


def pair_device(name, nx, ny, k, device, dtype=None):
    """pair(name, nx, ny, k) evaluated with torch on `device` (float64 arithmetic, same formulas): the 64 distinct
    3840x2160 pairs of BASELINE config 5 take minutes in numpy and a moment on the GPU.  The device's sin / cos may
    differ from libm's in the last bit, which can move floor() at isolated pixels -- fine for benchmark inputs;
    parity tests use pair()."""
    import torch
    f64 = torch.float64
    i, j = torch.meshgrid(torch.arange(ny, dtype=f64, device=device), torch.arange(nx, dtype=f64, device=device), indexing="ij")
    fx, fy = 1.5 + 0.5 * torch.sin(0.01 * i), -0.75 + 0.25 * torch.cos(0.013 * j)

    def tex(x, y, phase=0.3):
        return (127.5 + 40 * torch.sin(0.11 * x + phase) * torch.cos(0.07 * y) + 30 * torch.sin(0.031 * x + 0.043 * y)
                + 25 * torch.cos(0.19 * y - 0.05 * x) + 20 * torch.sin(0.37 * x) * torch.sin(0.29 * y))

    if name == "P0":
        I0, I1 = torch.floor(tex(j, i)), torch.floor(tex(j - fx, i - fy))
    elif name == "P1":
        phase = 0.3 + 0.1 * k
        mx, my = 4 + (k % 3), -3

        def tex_b(x, y):
            return tex(x, y, phase) + 15 * torch.sin(1.3 * x) * torch.cos(1.1 * y)

        def tex_f(x, y):
            return 127.5 + 60 * torch.sin(0.9 * x + 0.4 * y) + 50 * torch.cos(0.7 * y - 0.3 * x)

        def fg(x, y):
            return (x >= nx // 4) & (x < nx // 2) & (y >= ny // 4) & (y < ny // 2)

        I0 = torch.floor(torch.where(fg(j, i), tex_f(j, i), tex_b(j, i)))
        I1 = torch.floor(torch.where(fg(j - mx, i - my), tex_f(j - mx, i - my), tex_b(j - fx, i - fy)))
    else:
        raise ValueError(name)
    dtype = dtype or f64
    return I0.to(dtype).contiguous(), I1.to(dtype).contiguous()
