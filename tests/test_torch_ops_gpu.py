"""
The C ABI as PyTorch-ROCm custom ops (gance_amd/torch_ops.py): results equal the host-buffer entry points
bit for bit (same kernels), the ops run on torch's current stream, compose with torch code, pass
`torch.library.opcheck`, and turn status codes / bad arguments into Python exceptions.
"""

import numpy as np
import pytest
import torch

from gance_amd import hip_lib, synthetic, torch_ops
from gance_amd.stylegan2 import spec as sg2_spec
from oracle import audio_ref, resize_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; the product path has no CPU fallback")
    variables = sg2_spec.make_random_variables(64, seed=5, perturb=True)
    eng = hip_lib.Engine(variables, 64, max_batch=6)
    yield eng
    eng.close()


def test_synthesis_ops_equal_the_host_entry_points(engine) -> None:
    rng = np.random.RandomState(3)
    dlatents = rng.randn(5, engine.num_layers, 512).astype(np.float32)
    z = rng.randn(4, 512).astype(np.float32)
    want_w, want_image = engine.synthesize_w(dlatents, want_float=True)
    want_z = engine.synthesize_z(z, truncation_psi=0.7)
    side_stream = torch.cuda.Stream()
    with torch.cuda.stream(side_stream):  # the ops must follow torch's CURRENT stream
        d_w = torch.from_numpy(dlatents).cuda()
        frames = torch.ops.gance.synthesize_w(d_w, engine.op_handle)
        frames2, image = torch.ops.gance.synthesize_w_image(d_w, engine.op_handle)
        frames_z = torch.ops.gance.synthesize_z(torch.from_numpy(z).cuda(), engine.op_handle, 0.7)
        total = frames.sum(dtype=torch.int64)  # a torch kernel queued behind the op on the same stream
    side_stream.synchronize()
    assert np.array_equal(frames.cpu().numpy(), want_w) and np.array_equal(frames2.cpu().numpy(), want_w)
    assert np.array_equal(image.cpu().numpy(), want_image)
    assert np.array_equal(frames_z.cpu().numpy(), want_z)
    assert int(total.item()) == int(want_w.astype(np.int64).sum())


def test_out_variants_write_into_a_slice(engine) -> None:
    rng = np.random.RandomState(4)
    dlatents = torch.from_numpy(rng.randn(3, engine.num_layers, 512).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.randn(2, 512).astype(np.float32)).cuda()
    buffer = torch.zeros((7, 64, 64, 3), dtype=torch.uint8, device="cuda")
    torch.ops.gance.synthesize_w_out(dlatents, engine.op_handle, buffer[1:4])
    torch.ops.gance.synthesize_z_out(z, engine.op_handle, 1.2, buffer[5:7])
    torch.cuda.synchronize()
    assert torch.equal(buffer[1:4], torch.ops.gance.synthesize_w(dlatents, engine.op_handle))
    assert torch.equal(buffer[5:7], torch.ops.gance.synthesize_z(z, engine.op_handle, 1.2))
    assert int(buffer[0].max()) == 0 and int(buffer[4].max()) == 0
    with pytest.raises(ValueError):
        torch.ops.gance.synthesize_w_out(dlatents, engine.op_handle, buffer[:2])


def test_resize_op_matches_the_oracle() -> None:
    rng = np.random.RandomState(8)
    frames = rng.randint(0, 256, size=(2, 40, 40, 3)).astype(np.uint8)
    got = torch.ops.gance.resize_bicubic(torch.from_numpy(frames).cuda(), 90).cpu().numpy()
    want = resize_ref.resize_bicubic_u8(frames, 90)
    assert int(np.abs(got.astype(int) - want.astype(int)).max()) <= 1
    out = torch.empty((2, 90, 90, 3), dtype=torch.uint8, device="cuda")
    torch.ops.gance.resize_bicubic_out(torch.from_numpy(frames).cuda(), out)
    assert np.array_equal(out.cpu().numpy(), got)


def test_blend_op_matches_the_oracle() -> None:
    num_frames, L = 60, 512
    audio = synthetic.synthetic_audio(num_frames, L, seed=0)
    latents = synthetic.synthetic_final_latents(num_frames // 2, L, seed=4)
    owner = hip_lib.Blend(num_frames, num_frames // 2, 0.25, True, (-5, 5), 12, 3)
    handle = torch_ops.register_blend(owner)
    try:
        dlatents, indices = torch.ops.gance.blend(torch.from_numpy(audio).cuda(), torch.from_numpy(np.ascontiguousarray(latents[0])).cuda(), handle)
        torch.cuda.synchronize()
    finally:
        torch_ops.unregister(handle)
        owner.close()
    want = audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, audio, L, [0, 1, 2])
    combined = np.asarray(want.combined)  # (18, N*L) float64
    got = dlatents.cpu().numpy()
    np.testing.assert_allclose(got[:, 0, :].reshape(-1), combined[0], rtol=0, atol=2e-6)  # float32 of the float64 blend
    np.testing.assert_allclose(got[:, 17, :].reshape(-1), combined[17], rtol=0, atol=0)
    assert np.array_equal(indices.cpu().numpy(), np.asarray(want.network_indices))


def test_ops_reject_cpu_tensors_wrong_dtypes_and_unknown_handles(engine) -> None:
    d_w = torch.zeros((1, engine.num_layers, 512), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception):
        torch.ops.gance.synthesize_w(d_w.cpu(), engine.op_handle)  # no CPU implementation
    with pytest.raises(TypeError):
        torch.ops.gance.synthesize_w(d_w.double(), engine.op_handle)
    with pytest.raises(ValueError):
        torch.ops.gance.synthesize_w(d_w[:, :3], engine.op_handle)
    with pytest.raises(ValueError):
        torch.ops.gance.synthesize_w(d_w, 987654)
    with pytest.raises(hip_lib.GanceHipError):  # status code of the C ABI (batch > max_batch) -> exception
        torch.ops.gance.synthesize_w(d_w.expand(7, -1, -1), engine.op_handle)


def test_opcheck(engine) -> None:
    d_w = torch.randn((2, engine.num_layers, 512), dtype=torch.float32, device="cuda")
    torch.library.opcheck(torch.ops.gance.synthesize_w.default, (d_w, engine.op_handle), test_utils=("test_schema", "test_faketensor"))
    frames = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    torch.library.opcheck(torch.ops.gance.resize_bicubic.default, (frames, 24), test_utils=("test_schema", "test_faketensor"))
