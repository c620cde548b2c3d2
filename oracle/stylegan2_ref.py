"""
ORACLE (test infrastructure, NOT product code): CPU restatement of the StyleGAN2 config-f
generator that GANce drives through `network.run` / `network.components.synthesis.run`
(reference call sites: gance/network_interface/network_functions.py:121-125,152-157,168).

PARITY UNPINNED. The arithmetic lives in the git submodule `gance/stylegan2`
(esologic/stylegan2_gance, a fork of NVlabs/stylegan2; `.gitmodules:1-3`, no pinned SHA) which
is EMPTY in /root/reference, TensorFlow 1.14 is not installable here, and the reference's tests at
this boundary (test/test_network_functions.py:100-118) only pin an output shape and `sum > 0` and
need pickles that are absent. So this file restates the PUBLISHED algorithm (Karras et al.,
"Analyzing and Improving the Image Quality of StyleGAN", CVPR 2020, sec. 2 + app. B; NVlabs
stylegan2 `training/networks_stylegan2.py`, `dnnlib/tflib/ops/{upfirdn_2d,fused_bias_act}.py`,
`dnnlib/tflib/tfutil.py::convert_images_to_uint8`) and is anchored by the analytic known-answer
checks of SURVEY.md §8(c) (tests/test_oracle_stylegan2.py), not by reference outputs.

It is written the LITERAL way the published code is (per-sample modulated weights, grouped conv,
conv_transpose followed by upfirdn with zero insertion and explicit padding) on purpose: the HIP
path uses a different decomposition (shared weights + input scaling + demod epilogue, parity-class
transposed conv), so agreement between the two is evidence, not tautology.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Variables = Dict[str, np.ndarray]

_RESAMPLE_KERNEL = (1.0, 3.0, 3.0, 1.0)


def _nf(stage: int, fmap_base: int = 16 << 10, fmap_max: int = 512) -> int:
    """networks_stylegan2.py `nf(stage)` with fmap_decay=1, fmap_min=1."""
    return int(np.clip(int(fmap_base / (2.0 ** stage)), 1, fmap_max))


def _t(variables: Variables, name: str, dtype: torch.dtype) -> torch.Tensor:
    return torch.from_numpy(np.asarray(variables[name])).to(dtype)


def _get_weight(raw: torch.Tensor, gain: float = 1.0, lrmul: float = 1.0) -> torch.Tensor:
    """`get_weight` with use_wscale=True: runtime_coef = gain / sqrt(fan_in) * lrmul."""
    fan_in = int(np.prod(raw.shape[:-1]))
    he_std = gain / np.sqrt(fan_in)
    return raw * (he_std * lrmul)


def _setup_kernel(k: tuple, dtype: torch.dtype) -> torch.Tensor:
    """upfirdn_2d.py `_setup_kernel`: separable outer product, normalised to sum 1."""
    k1 = torch.tensor(k, dtype=dtype)
    k2 = torch.outer(k1, k1)
    return k2 / k2.sum()


def upfirdn_2d(x: torch.Tensor, k: torch.Tensor, up: int, pad0: int, pad1: int) -> torch.Tensor:
    """
    upfirdn_2d.py `_upfirdn_2d_ref` for NCHW, down=1: zero-insert upsample, pad (negative pad not
    needed here), correlate with the FLIPPED kernel.
    """
    n, c, h, w = x.shape
    if up > 1:
        z = x.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(n, c, h * up, w * up)
    x = F.pad(x, (pad0, pad1, pad0, pad1))
    kf = torch.flip(k, (0, 1))[None, None].repeat(c, 1, 1, 1)
    return F.conv2d(x, kf, groups=c)


def upsample_2d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """upfirdn_2d.py `upsample_2d(x, k=[1,3,3,1])`: gain factor**2, pad0=(p+1)//2+factor-1, pad1=p//2."""
    k = _setup_kernel(_RESAMPLE_KERNEL, x.dtype) * (factor ** 2)
    p = k.shape[0] - factor
    return upfirdn_2d(x, k, up=factor, pad0=(p + 1) // 2 + factor - 1, pad1=p // 2)


def upsample_conv_2d(x: torch.Tensor, w_oihw_grouped: torch.Tensor, groups: int) -> torch.Tensor:
    """
    upfirdn_2d.py `upsample_conv_2d`: stride-2 VALID conv2d_transpose with the spatially flipped
    filter, then FIR [1,3,3,1] (x) [1,3,3,1] * 4 with pad0 = pad1 = 1 (p = (4-2)-(3-1) = 0).
    `w_oihw_grouped` is [groups*O, I, 3, 3] (un-flipped, cross-correlation orientation).
    """
    go, i, kh, kw = w_oihw_grouped.shape
    o = go // groups
    # TF: w[::-1, ::-1] then conv2d_transpose (the gradient of a cross-correlation). torch's
    # conv_transpose2d is that same gradient, with weight laid out [in, out/groups, kh, kw].
    wt = torch.flip(w_oihw_grouped, (2, 3)).reshape(groups, o, i, kh, kw)
    wt = wt.permute(0, 2, 1, 3, 4).reshape(groups * i, o, kh, kw)
    x = F.conv_transpose2d(x, wt, stride=2, padding=0, groups=groups)
    factor = 2
    k = _setup_kernel(_RESAMPLE_KERNEL, x.dtype) * (factor ** 2)
    p = (k.shape[0] - factor) - (kw - 1)
    return upfirdn_2d(x, k, up=1, pad0=(p + 1) // 2 + factor - 1, pad1=p // 2 + 1)


def apply_bias_act(x: torch.Tensor, b: torch.Tensor, act: str, lrmul: float = 1.0) -> torch.Tensor:
    """`apply_bias_act` -> `fused_bias_act`: lrelu has alpha 0.2 and gain sqrt(2); linear gain 1."""
    b = b * lrmul
    x = x + (b.reshape(1, -1, 1, 1) if x.ndim == 4 else b.reshape(1, -1))
    if act == "lrelu":
        return F.leaky_relu(x, 0.2) * np.sqrt(2.0)
    if act == "linear":
        return x
    raise ValueError(act)


def modulated_conv2d_layer(
    x: torch.Tensor,
    y: torch.Tensor,
    variables: Variables,
    scope: str,
    kernel: int,
    up: bool = False,
    demodulate: bool = True,
) -> torch.Tensor:
    """networks_stylegan2.py `modulated_conv2d_layer`, fused_modconv=True branch."""
    dtype = x.dtype
    batch, cin, h, w_ = x.shape
    w = _get_weight(_t(variables, f"{scope}/weight", dtype))  # [k,k,I,O]
    cout = w.shape[-1]
    ww = w[None]  # [B,k,k,I,O]
    mod_w = _get_weight(_t(variables, f"{scope}/mod_weight", dtype))  # [512, I]
    s = y @ mod_w
    s = s + _t(variables, f"{scope}/mod_bias", dtype)[None] + 1.0  # [B,I]
    ww = ww * s[:, None, None, :, None]
    if demodulate:
        d = torch.rsqrt((ww ** 2).sum(dim=(1, 2, 3)) + 1e-8)  # [B,O]
        ww = ww * d[:, None, None, None, :]
    # grouped conv: x -> [1, B*I, H, W], weight -> [B*O, I, k, k]
    xg = x.reshape(1, batch * cin, h, w_)
    wg = ww.permute(0, 4, 3, 1, 2).reshape(batch * cout, cin, kernel, kernel)
    if up:
        out = upsample_conv_2d(xg, wg, groups=batch)
    else:
        out = F.conv2d(xg, wg, padding=kernel // 2, groups=batch)
    return out.reshape(batch, cout, out.shape[2], out.shape[3])


def g_mapping(latents: torch.Tensor, variables: Variables, num_layers: int) -> torch.Tensor:
    """`G_mapping`: normalise z, 8 dense layers (lrmul 0.01, lrelu*sqrt2), broadcast to W rows."""
    dtype = latents.dtype
    x = latents * torch.rsqrt((latents ** 2).mean(dim=1, keepdim=True) + 1e-8)
    for i in range(8):
        w = _get_weight(_t(variables, f"G_mapping/Dense{i}/weight", dtype), lrmul=0.01)
        x = apply_bias_act(x @ w, _t(variables, f"G_mapping/Dense{i}/bias", dtype), "lrelu", lrmul=0.01)
    return x[:, None, :].repeat(1, num_layers, 1)


def truncate(dlatents: torch.Tensor, variables: Variables, psi: Optional[float]) -> torch.Tensor:
    """`G_main` truncation trick with truncation_cutoff=None: lerp(avg, w, psi) on every row."""
    if psi is None:
        return dlatents
    avg = _t(variables, "dlatent_avg", dlatents.dtype)[None, None]
    return avg + (dlatents - avg) * psi


def g_synthesis(
    dlatents: torch.Tensor,
    variables: Variables,
    resolution: int,
    noise_override: Optional[Dict[int, torch.Tensor]] = None,
    stop_after: Optional[int] = None,
    collect: Optional[list] = None,
) -> torch.Tensor:
    """
    `G_synthesis_stylegan2`, architecture 'skip', randomize_noise=False (stored noise buffers).
    dlatents [B, W, 512] -> images [B, 3, R, R] (float).
    `stop_after=n` (debug) returns the activation x after the n-th conv layer instead; `collect` (a list) receives the
    activation after every conv layer, in order (the layer-wise parity tests: one pass instead of one per layer).
    """
    dtype = dlatents.dtype
    res_log2 = int(np.log2(resolution))
    batch = dlatents.shape[0]

    def layer(x: torch.Tensor, layer_idx: int, scope: str, up: bool) -> torch.Tensor:
        x = modulated_conv2d_layer(x, dlatents[:, layer_idx], variables, scope, 3, up=up)
        if noise_override is not None and layer_idx in noise_override:
            noise = noise_override[layer_idx].to(dtype)
        else:
            noise = _t(variables, f"G_synthesis/noise{layer_idx}", dtype)
        x = x + noise * _t(variables, f"{scope}/noise_strength", dtype)
        return apply_bias_act(x, _t(variables, f"{scope}/bias", dtype), "lrelu")

    def torgb(x: torch.Tensor, y: Optional[torch.Tensor], res: int) -> torch.Tensor:
        scope = f"G_synthesis/{2**res}x{2**res}/ToRGB"
        t = modulated_conv2d_layer(x, dlatents[:, res * 2 - 3], variables, scope, 1, demodulate=False)
        t = apply_bias_act(t, _t(variables, f"{scope}/bias", dtype), "linear")
        return t if y is None else y + t

    x = _t(variables, "G_synthesis/4x4/Const/const", dtype).repeat(batch, 1, 1, 1)
    x = layer(x, 0, "G_synthesis/4x4/Conv", up=False)
    if collect is not None:
        collect.append(x)
    if stop_after == 1:
        return x
    y = torgb(x, None, 2)
    for res in range(3, res_log2 + 1):
        side = 2 ** res
        x = layer(x, res * 2 - 5, f"G_synthesis/{side}x{side}/Conv0_up", up=True)
        if collect is not None:
            collect.append(x)
        if stop_after == res * 2 - 4:
            return x
        x = layer(x, res * 2 - 4, f"G_synthesis/{side}x{side}/Conv1", up=False)
        if collect is not None:
            collect.append(x)
        if stop_after == res * 2 - 3:
            return x
        y = upsample_2d(y)
        y = torgb(x, y, res)
    return y


def convert_images_to_uint8(images_nchw: torch.Tensor) -> np.ndarray:
    """
    tfutil.py `convert_images_to_uint8(images, drange=[-1,1], nchw_to_nhwc=True)`:
    cast to float32, transpose, x*127.5 + (0.5 + 127.5), tf.saturate_cast(uint8) = clamp then
    truncate toward zero.
    """
    x = images_nchw.to(torch.float32).permute(0, 2, 3, 1)
    x = x * np.float32(127.5) + np.float32(128.0)
    return torch.clamp(x, 0.0, 255.0).to(torch.uint8).numpy()


def synthesize_w(
    dlatents: np.ndarray, variables: Variables, resolution: int, dtype: torch.dtype = torch.float64
) -> torch.Tensor:
    """Matrix path (network_functions.py:160-169): dlatents [B,W,512] -> float images NCHW."""
    with torch.no_grad():
        return g_synthesis(torch.from_numpy(np.asarray(dlatents)).to(dtype), variables, resolution)


def synthesize_z(
    z: np.ndarray,
    variables: Variables,
    resolution: int,
    truncation_psi: Optional[float] = 1.2,
    dtype: torch.dtype = torch.float64,
) -> torch.Tensor:
    """
    Vector path (network_functions.py:144-158): z [B,512] -> mapping -> truncation (psi=1.2 on
    all rows) -> synthesis. The published default randomize_noise=True draws fresh noise per
    call; with noise_strength == 0 (random init) it is irrelevant, otherwise the stored buffers are
    used here so the path stays deterministic (DESIGN.md records the deviation).
    """
    num_layers = int(np.log2(resolution)) * 2 - 2
    with torch.no_grad():
        w = g_mapping(torch.from_numpy(np.asarray(z)).to(dtype), variables, num_layers)
        w = truncate(w, variables, truncation_psi)
        return g_synthesis(w, variables, resolution)
