"""
How much does a float32 Winograd F(4x4,3x3) Conv1 cost in image accuracy? (ship only if the
image stays within 1e-4 of the fp64 oracle.) CPU experiment, no GPU: the oracle's own forward pass in float64
with the stride-1 3x3 convs of the chosen resolutions replaced by an emulation of the kernel's arithmetic
(float32 input transform, float32 products summed in float32, float32 output transform), for two point sets.

    python tools/experiments/winograd_f43_error.py [resolution] [first_res_with_winograd] [last] [stress]

`stress` (any fourth argument): the trained-statistics network of `spec.make_stress_variables` instead of the random init.
"""
import sys
from fractions import Fraction
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402
from oracle import stylegan2_ref as ref  # noqa: E402


def cook_toom(points, m=4, r=3):
    """AT [m, n], G [n, r], BT [n, n] for F(m, r) on the finite `points` (+ infinity), exact rationals -> float64."""
    import sympy
    n = m + r - 1
    a = [sympy.Rational(p) for p in points]
    assert len(a) == n - 1
    x = sympy.symbols("x")
    AT = sympy.zeros(m, n)
    G = sympy.zeros(n, r)
    BT = sympy.zeros(n, n)
    M = sympy.prod([x - ai for ai in a])
    for j, aj in enumerate(a):
        Nj = sympy.prod([aj - al for l, al in enumerate(a) if l != j])
        for i in range(m):
            AT[i, j] = aj**i
        for k in range(r):
            G[j, k] = aj**k / Nj
        Mj = sympy.Poly(sympy.expand(M / (x - aj)), x).all_coeffs()[::-1]
        for c, v in enumerate(Mj):
            BT[j, c] = v
    AT[m - 1, n - 1] = 1
    G[n - 1, r - 1] = 1
    Mc = sympy.Poly(sympy.expand(M), x).all_coeffs()[::-1]
    for c, v in enumerate(Mc):
        BT[n - 1, c] = v
    to = lambda mat: np.array(mat.tolist(), dtype=np.float64)
    return to(AT), to(G), to(BT)


def check(AT, G, BT, m=4, r=3):
    rng = np.random.RandomState(0)
    d = rng.randn(m + r - 1)
    g = rng.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    want = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, want), (y, want)


def winograd_conv_f32(x64, w64, AT, G, BT, m):
    """x [B, C, H, W], w [B, O, C, 3, 3] (per-sample modulated weights), padding 1, float32 Winograd arithmetic."""
    B, C, H, W = x64.shape
    O = w64.shape[1]
    n = m + 2
    x = F.pad(x64.to(torch.float32), (1, 1, 1, 1))
    at, g, bt = (torch.from_numpy(v).to(torch.float32) for v in (AT, G, BT))
    U = torch.einsum("ij,bocjk,lk->bocil", torch.from_numpy(G), w64, torch.from_numpy(G)).to(torch.float32)  # weights: transformed in fp64 offline
    th, tw = H // m, W // m
    tiles = x.unfold(2, n, m).unfold(3, n, m)  # [B, C, th, tw, n, n]
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", bt, tiles, bt)
    Mm = torch.einsum("bocil,bcyxil->boyxil", U, V)
    Y = torch.einsum("ij,boyxjk,lk->boyxil", at, Mm, at)  # [B, O, th, tw, m, m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, O, H, W).to(torch.float64)


def main():
    resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    stress = len(sys.argv) > 4
    variables = sg2_spec.make_stress_variables(resolution, seed=0) if stress else sg2_spec.make_random_variables(resolution, seed=0, perturb=True)
    z = np.random.RandomState(1).randn(2, 512).astype(np.float32)
    want = ref.synthesize_z(z, variables, resolution, truncation_psi=1.2)
    sets = {
        "F(2,3) 0,+-1": (2, [0, 1, -1]),
        "F(4,3) 0,+-1,+-2": (4, [0, 1, -1, 2, -2]),
        "F(4,3) 0,+-1,+-1/2": (4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)]),
    }
    original = F.conv2d
    for label, (m, points) in sets.items():
        AT, G, BT = cook_toom(points, m=m)
        check(AT, G, BT, m=m)

        def patched(x, w, *args, **kwargs):
            groups = kwargs.get("groups", 1)
            if w.shape[-1] == 3 and kwargs.get("padding", 0) == 1 and first <= x.shape[-1] <= last and x.shape[-1] % m == 0 and groups > 1:
                B = groups
                C = x.shape[1] // B
                xb = x.reshape(B, C, x.shape[2], x.shape[3])
                wb = w.reshape(B, w.shape[0] // B, C, 3, 3)
                out = winograd_conv_f32(xb, wb, AT, G, BT, m)
                return out.reshape(1, -1, out.shape[2], out.shape[3])
            return original(x, w, *args, **kwargs)

        F.conv2d = patched
        try:
            got = ref.synthesize_z(z, variables, resolution, truncation_psi=1.2)
        finally:
            F.conv2d = original
        err = (got - want).abs()
        u8a, u8b = ref.convert_images_to_uint8(got), ref.convert_images_to_uint8(want)
        print(f"{label:22s} layers {first}..{last} of {resolution}: max|img - fp64| = {err.max().item():.3e}  (image range {want.abs().max().item():.1f}), "
              f"u8 differing {100.0 * float((u8a != u8b).mean()):.4f} %, max {int(np.abs(u8a.astype(int) - u8b.astype(int)).max())} LSB", flush=True)


if __name__ == "__main__":
    main()
