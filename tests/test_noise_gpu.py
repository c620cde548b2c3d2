"""
GPU parity tests of the `noise-blend` source: gance_gaussian_noise + gance_blend_* of
libgance_hip.so (through the C ABI) against the golden vectors captured from the reference's own
`gaussian_data` / `alpha_blend_vectors_max_rms_power_audio`, and against the CPU oracle.

Bars: network indices bit-exact. The noise field is float32 on both sides; the kernel keeps
scipy's float64 accumulation order per filter pass, but builds its Gaussian taps with libm's exp
and reduces the RMS in float64 where the reference has numpy's SIMD exp and a float32 pairwise
sum, so values agree to a few float32 ulps: 4e-6 absolute on the [-4, 4] field (observed <= 1e-6),
2e-6 on the unit-RMS field. The float64 blend inherits that through noise * (1 - alpha).
"""

import numpy as np
import pytest
import torch

from gance_amd import hip_lib, synthetic
from gance_amd.data_into_network_visualization import visualization_inputs
from gance_amd.vector_sources import primatives
from oracle import audio_ref

pytestmark = pytest.mark.gpu

NOISE_ATOL = 4e-6
UNIT_ATOL = 2e-6
NOISE_CASES = ["noise_n60_seed0_roll_k3", "noise_n600_seed5_noroll_k2"]


@pytest.mark.parametrize("name", NOISE_CASES)
def test_noise_blend_matches_reference_goldens(golden_dir, name: str) -> None:
    golden = np.load(golden_dir / f"{name}.npz")
    num_frames, L, seed, roll, num_networks, stride = (int(v) for v in golden["meta"])
    alpha, amp_lo, amp_hi = (float(v) for v in golden["alpha_amp"])
    audio = synthetic.synthetic_audio(num_frames, L, seed=seed)
    out = visualization_inputs.alpha_blend_vectors_max_rms_power_audio(
        alpha, bool(roll), (amp_lo, amp_hi), audio, L, list(range(num_networks))
    )
    assert out.b_vectors.data.dtype == np.float32 and out.b_vectors.data.shape == (num_frames * L,)
    assert out.combined.data.dtype == np.float64 and out.a_vectors.data.dtype == np.float64
    assert np.array_equal(out.network_indices.result.data, golden["network_indices"])
    np.testing.assert_allclose(out.b_vectors.data[::stride], golden["noise_sample"], rtol=0, atol=NOISE_ATOL)
    assert out.b_vectors.data.min() == -4.0 and out.b_vectors.data.max() == 4.0  # as the reference's
    np.testing.assert_allclose(out.a_vectors.data[::stride], golden["spectrogram_sample"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(out.combined.data[::stride], golden["combined_sample"], rtol=0, atol=NOISE_ATOL)
    # the bare source, default sigmas (across 20) and a two-axis filter
    got = primatives.gaussian_data(vector_length=L, num_vectors=num_frames)
    assert got.dtype == np.float32 and got.shape == (num_frames * L,)
    np.testing.assert_allclose(got[::stride], golden["gaussian_default_sample"], rtol=0, atol=UNIT_ATOL)
    got = primatives.gaussian_data(vector_length=L, num_vectors=num_frames, sigmas=primatives.Sigmas(3, 2))
    np.testing.assert_allclose(got[::stride], golden["gaussian_both_sample"], rtol=0, atol=UNIT_ATOL)


@pytest.mark.parametrize("num_vectors,length,sigmas,seed", [
    (40, 512, (50, 0), 3),     # radius 200 > N: the wrap goes round several times
    (257, 96, (0, 5), 4),      # within-vector only
    (33, 64, (0, 0), 5),       # no filter: only the RMS normalisation
    (1800, 512, (50, 0), 1234),  # the benchmark's frame count
])
def test_gaussian_data_matches_oracle(num_vectors, length, sigmas, seed) -> None:
    want = audio_ref.gaussian_data(length, num_vectors, sigmas[0], sigmas[1], seed=seed)
    got = primatives.gaussian_data(length, num_vectors, primatives.Sigmas(*sigmas), np.random.RandomState(seed))
    np.testing.assert_allclose(got, want, rtol=0, atol=UNIT_ATOL)
    np.testing.assert_allclose(np.sqrt(np.mean(np.square(got.astype(np.float64)))), 1.0, rtol=0, atol=1e-6)


def test_device_noise_blend_feeds_the_vector_path() -> None:
    """The float32 z vectors left in HBM are float32(combined), i.e. what network.run receives."""
    num_frames, L = 64, 512
    audio = synthetic.synthetic_audio(num_frames, L, seed=9)
    want = audio_ref.alpha_blend_vectors_max_rms_power_audio(0.4, True, (-3, 3), audio, L, [0, 1])
    result = visualization_inputs.alpha_blend_vectors_max_rms_power_audio_device(0.4, True, (-3, 3), audio, L, 2, keep_stages=True)
    try:
        assert result.vectors.shape == (num_frames, L) and result.vectors.dtype == torch.float32
        combined = result.blend.read_stage("blend_row")
        assert np.array_equal(result.vectors.cpu().numpy(), combined.astype(np.float32))
        np.testing.assert_allclose(combined.reshape(-1), want.combined, rtol=0, atol=NOISE_ATOL)
        assert np.array_equal(result.network_indices.cpu().numpy(), want.network_indices)
    finally:
        result.blend.close()


def test_gaussian_noise_rejects_bad_arguments() -> None:
    draws = torch.zeros((8, 16), dtype=torch.float32, device="cuda")
    out = torch.empty_like(draws)
    with pytest.raises(ValueError, match="feature range"):
        hip_lib.gaussian_noise_device(draws.data_ptr(), 8, 16, 1.0, 0.0, (4, -4), out.data_ptr())
    with pytest.raises(hip_lib.GanceHipError):
        hip_lib.gaussian_noise_device(draws.data_ptr(), 8, 16, -1.0, 0.0, None, out.data_ptr())
    with pytest.raises(hip_lib.GanceHipError):
        hip_lib.gaussian_noise_device(draws.data_ptr(), 8, 16, 1.0, 0.0, None, draws.data_ptr())  # in place
