"""
CPU tests pinning the synthesis ORACLE with the analytic known-answer checks of SURVEY.md §8(c)
(the reference itself holds no golden image: test/test_network_functions.py:100-118 checks only
shape and sum > 0, and needs pickles that are absent).
"""

import numpy as np
import pytest
import torch

from gance_amd.stylegan2 import spec as sg2_spec
from oracle import stylegan2_ref as ref


@pytest.fixture(scope="module")
def net32():
    """Small generator with every optional term switched on."""
    return sg2_spec.make_random_variables(32, seed=0, perturb=True)


def test_layer_table_matches_published_architecture() -> None:
    """(vi) W = 18 rows at 1024, 14 at 256; channel ladder 512..32; dlatent row bookkeeping."""
    spec = sg2_spec.make_spec(1024)
    assert spec.num_layers == 18
    assert [c.cout for c in spec.convs] == [512] * 9 + [256, 256, 128, 128, 64, 64, 32, 32]
    assert [c.layer_idx for c in spec.convs] == list(range(17))
    assert [r.dlatent_row for r in spec.torgbs] == [1, 3, 5, 7, 9, 11, 13, 15, 17]
    assert sg2_spec.make_spec(256).num_layers == 14
    assert [c.cout for c in sg2_spec.make_spec(256).convs][-4:] == [256, 256, 128, 128]
    # parameter counts quoted in SURVEY.md §8 a18: 23.59 M conv weights + 4.65 M affine weights
    conv_params = sum(9 * c.cin * c.cout for c in spec.convs) + sum(r.cin * 3 for r in spec.torgbs)
    affine_params = sum(512 * c.cin for c in spec.convs) + sum(512 * r.cin for r in spec.torgbs)
    assert abs(conv_params / 1e6 - 23.59) < 0.02
    assert abs(affine_params / 1e6 - 4.65) < 0.01


def test_algorithmic_mac_count() -> None:
    """74 063 MMAC of 3x3 conv per 1024^2 frame (up layers on the input grid)."""
    spec = sg2_spec.make_spec(1024)
    macs = 0
    for conv in spec.convs:
        side = 2 ** conv.res_log2 // (2 if conv.up else 1)
        macs += 9 * conv.cin * conv.cout * side * side
    assert abs(macs / 1e6 - 74063) < 1.0


def test_uint8_conversion_known_answers() -> None:
    """(iv) u8(-1)=0, u8(0)=128, u8(1)=255 (clip), truncation toward zero, saturation."""
    values = torch.tensor([-1.0, 0.0, 1.0, 0.999, -2.0, 3.0, 0.0039, -0.0040]).reshape(1, 1, 1, 8)
    values = values.repeat(1, 3, 1, 1)
    out = ref.convert_images_to_uint8(values)[0, 0, :, 0]
    assert out.tolist() == [0, 128, 255, 255, 0, 255, 128, 127]


def test_fir_upsample_of_constant_is_constant_inside() -> None:
    """(iii) [1,3,3,1]x[1,3,3,1]/64*4 upsampling keeps a constant image, away from the border."""
    x = torch.full((1, 2, 6, 6), 2.5, dtype=torch.float64)
    y = ref.upsample_2d(x)
    assert y.shape == (1, 2, 12, 12)
    assert torch.allclose(y[:, :, 2:-2, 2:-2], torch.full_like(y[:, :, 2:-2, 2:-2], 2.5))
    assert float(y[0, 0, 0, 0]) == pytest.approx(2.5 * 0.75 * 0.75)  # zero padding, not clamp


def test_demodulated_weights_have_unit_norm(net32) -> None:
    """(ii) after demodulation every (sample, out-channel) filter has L2 norm 1 (eps 1e-8)."""
    scope = "G_synthesis/8x8/Conv1"
    w = ref._get_weight(torch.from_numpy(net32[f"{scope}/weight"]).double())  # pylint: disable=protected-access
    s = torch.randn(3, 512, dtype=torch.float64) + 1.0
    ww = w[None] * s[:, None, None, :, None]
    d = torch.rsqrt((ww ** 2).sum(dim=(1, 2, 3)) + 1e-8)
    norms = ((ww * d[:, None, None, None, :]) ** 2).sum(dim=(1, 2, 3)).sqrt()
    assert torch.allclose(norms, torch.ones_like(norms), atol=1e-6)


def test_truncation_psi_one_is_identity(net32) -> None:
    """(v) psi = 1 leaves dlatents unchanged; psi = 1.2 extrapolates away from dlatent_avg."""
    dl = torch.randn(2, 8, 512, dtype=torch.float64)
    assert torch.allclose(ref.truncate(dl, net32, 1.0), dl, atol=1e-14, rtol=0)
    avg = torch.from_numpy(net32["dlatent_avg"]).double()
    out = ref.truncate(dl, net32, 1.2)
    assert torch.allclose(out - avg, 1.2 * (dl - avg))


def test_random_init_vector_path_is_deterministic_and_noise_free() -> None:
    """(i) random init => noise_strength = 0 and biases 0: noise buffers cannot matter."""
    variables = sg2_spec.make_random_variables(16, seed=1)
    z = np.random.RandomState(0).randn(1, 512).astype(np.float32)
    a = ref.synthesize_z(z, variables, 16)
    changed = dict(variables)
    for name in variables:
        if "/noise" in name and "strength" not in name:
            changed[name] = variables[name] * 0.0 + 7.0
    b = ref.synthesize_z(z, changed, 16)
    assert torch.equal(a, b)


def test_upsample_conv_equals_dense_zero_insert_formulation() -> None:
    """
    The transposed-conv + FIR path equals: zero-insert x2, full correlation with the flipped 3x3
    filter, then the 4x4 FIR with pad 1/1 -- an independent restatement of upsample_conv_2d.
    """
    torch.manual_seed(0)
    x = torch.randn(1, 3, 5, 5, dtype=torch.float64)
    w = torch.randn(4, 3, 3, 3, dtype=torch.float64)  # [O, I, kh, kw]
    got = ref.upsample_conv_2d(x, w, groups=1)
    z = torch.zeros(1, 3, 9, 9, dtype=torch.float64)
    z[:, :, ::2, ::2] = x
    # conv_transpose(stride 2) of x with flipped w == full correlation of zero-inserted x with w
    t = torch.nn.functional.conv2d(torch.nn.functional.pad(z, (2, 2, 2, 2)), w)
    k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    k2 = torch.outer(k, k) / 16.0
    want = torch.nn.functional.conv2d(
        torch.nn.functional.pad(t, (1, 1, 1, 1)), k2[None, None].repeat(4, 1, 1, 1), groups=4
    )
    assert got.shape == (1, 4, 10, 10)
    assert torch.allclose(got, want, atol=1e-12)


def test_fp32_oracle_tracks_fp64_oracle(net32) -> None:
    """The honest tolerance floor: fp32 CPU vs fp64 CPU of the same restatement."""
    dl = np.random.RandomState(3).randn(2, 8, 512).astype(np.float32)
    a = ref.synthesize_w(dl, net32, 32, dtype=torch.float64)
    b = ref.synthesize_w(dl, net32, 32, dtype=torch.float32)
    assert float((a - b.double()).abs().max()) < 1e-4
