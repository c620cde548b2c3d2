"""
ORACLE tooling (build container only): capture golden vectors from the REFERENCE's own code.

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py

Imports gance.data_into_network_visualization.visualization_inputs / gance.apply_spectrogram /
gance.vector_sources.* from /root/reference with the leaf stubs of oracle/ref_stubs.py and writes
small .npz fixtures to tests/golden/. Inputs are regenerated from seeds by
gance_amd.synthetic (so they need not be stored); float stages are stored as a strided sample
plus (min, max, sum) over the full array, integer stages in full.
"""

import sys
from pathlib import Path

import numpy as np

REPO_ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO_ROOT))

from gance_amd import synthetic  # noqa: E402
from oracle import ref_stubs  # noqa: E402

GOLDEN_DIR = REPO_ROOT / "tests" / "golden"


def sampled(array: np.ndarray, stride: int) -> dict:
    """Strided sample + full-array statistics of a float stage."""
    flat = np.asarray(array).reshape(-1)
    return {
        "sample": flat[::stride].copy(),
        "stats": np.array([flat.min(), flat.max(), flat.sum(dtype=np.float64), float(flat.size)], dtype=np.float64),
    }


def blend_case(name: str, num_frames: int, seed: int, roll: bool, num_networks: int, stride: int) -> None:
    """One end-to-end `alpha_blend_projection_file` case with every reference intermediate."""
    from gance import apply_spectrogram  # pylint: disable=import-outside-toplevel,import-error
    from gance.data_into_network_visualization import visualization_inputs as vi  # pylint: disable=import-outside-toplevel,import-error
    from gance.vector_sources import vector_reduction, vector_sources_common as vsc  # pylint: disable=import-outside-toplevel,import-error
    from gance.vector_sources.vector_types import MatricesLabel  # pylint: disable=import-outside-toplevel,import-error

    L, alpha, amp, depth = 512, 0.25, (-5, 5), 12
    audio = synthetic.synthetic_audio(num_frames, L, seed=seed)
    latents = synthetic.synthetic_final_latents(num_frames // 2, L, seed=seed + 4)

    db = apply_spectrogram.compute_spectrogram(audio, L)
    scaled = apply_spectrogram.reshape_spectrogram_to_vectors(db, amplitude_range=amp, vector_length=L)
    smoothed_time = vsc.smooth_across_vectors(scaled, L, window_length=7, polyorder=3)
    smoothed = vsc.smooth_each_vector(data=smoothed_time, vector_length=L, window_length=5, polyorder=3)
    assert np.array_equal(smoothed, apply_spectrogram.compute_spectrogram_smooth_scale(audio, L, amp))
    raw_rms = vector_reduction._compute_raw_rms(audio, L)  # pylint: disable=protected-access
    rolling_layers = vector_reduction.reduce_vector_rms_rolling_average(time_series_audio_vectors=audio, vector_length=L)
    roll_values = vector_reduction.quantize_results_layers(rolling_layers, list(np.arange(0, 3))).result.data
    final = vi._create_spectrogram(audio, L, amp, roll)  # pylint: disable=protected-access
    out = vi.alpha_blend_projection_file(
        final_latents_matrices_label=MatricesLabel(latents, L, "golden"),
        alpha=alpha,
        fft_roll_enabled=roll,
        fft_amplitude_range=amp,
        blend_depth=depth,
        time_series_audio_vectors=audio,
        vector_length=L,
        network_indices=list(range(num_networks)),
    )
    assert np.array_equal(out.a_vectors.data, final)
    combined = out.combined.data
    assert combined.shape == (18, num_frames * L) and combined.dtype == np.float64
    assert all(np.array_equal(combined[0], combined[r]) for r in range(1, depth))
    assert all(np.array_equal(combined[depth], combined[r]) for r in range(depth, 18))

    arrays = {
        "meta": np.array([num_frames, L, num_frames // 2, seed, int(roll), num_networks, stride, depth], dtype=np.int64),
        "alpha_amp": np.array([alpha, amp[0], amp[1]], dtype=np.float64),
        "raw_rms": np.asarray(raw_rms),
        "rolling_average": np.asarray(rolling_layers.layers[0].data),
        "rolling_smoothed": np.asarray(rolling_layers.result.data),
        "roll_values": np.asarray(roll_values, dtype=np.int64),
        "network_indices": np.asarray(out.network_indices.result.data, dtype=np.int64),
        "network_index_smoothed": np.asarray(out.network_indices.layers[0].data),
    }
    for key, value in {
        "db": db,
        "scaled": scaled,
        "smoothed_time": smoothed_time,
        "smoothed": smoothed,
        "final": final,
        "combined_row0": combined[0],
        "combined_row_depth": combined[depth],
        "projected_row0": out.b_vectors.data[0],
    }.items():
        for suffix, array in sampled(value, stride).items():
            arrays[f"{key}_{suffix}"] = array
    np.savez_compressed(GOLDEN_DIR / f"{name}.npz", **arrays)
    print(f"wrote {name}.npz  ({(GOLDEN_DIR / (name + '.npz')).stat().st_size / 1024:.0f} KiB)")


def noise_case(name: str, num_frames: int, seed: int, roll: bool, num_networks: int, stride: int) -> None:
    """One end-to-end `alpha_blend_vectors_max_rms_power_audio` (noise-blend) case + the bare noise source."""
    from gance.data_into_network_visualization import visualization_inputs as vi  # pylint: disable=import-outside-toplevel,import-error
    from gance.vector_sources.primatives import Sigmas, gaussian_data  # pylint: disable=import-outside-toplevel,import-error

    L, alpha, amp = 512, 0.25, (-5, 5)
    audio = synthetic.synthetic_audio(num_frames, L, seed=seed)
    out = vi.alpha_blend_vectors_max_rms_power_audio(
        alpha=alpha,
        fft_roll_enabled=roll,
        fft_amplitude_range=amp,
        time_series_audio_vectors=audio,
        vector_length=L,
        network_indices=list(range(num_networks)),
    )
    assert out.b_vectors.data.dtype == np.float32 and out.combined.data.dtype == np.float64
    arrays = {
        "meta": np.array([num_frames, L, seed, int(roll), num_networks, stride], dtype=np.int64),
        "alpha_amp": np.array([alpha, amp[0], amp[1]], dtype=np.float64),
        "network_indices": np.asarray(out.network_indices.result.data, dtype=np.int64),
    }
    for key, value in {
        "spectrogram": out.a_vectors.data,
        "noise": out.b_vectors.data,
        "combined": out.combined.data,
        "gaussian_default": gaussian_data(vector_length=L, num_vectors=num_frames),  # Sigmas(20, 0)
        "gaussian_both": gaussian_data(vector_length=L, num_vectors=num_frames, sigmas=Sigmas(3, 2)),
    }.items():
        for suffix, array in sampled(value, stride).items():
            arrays[f"{key}_{suffix}"] = array
    np.savez_compressed(GOLDEN_DIR / f"{name}.npz", **arrays)
    print(f"wrote {name}.npz  ({(GOLDEN_DIR / (name + '.npz')).stat().st_size / 1024:.0f} KiB)")


def overlay_cases() -> None:
    """Known answers of the overlay gate's helpers, from the reference's own functions."""
    import pandas as pd  # pylint: disable=import-outside-toplevel
    from gance.overlay import overlay_common as oc  # pylint: disable=import-outside-toplevel,import-error
    from gance.vector_sources import vector_reduction  # pylint: disable=import-outside-toplevel,import-error

    rng = np.random.RandomState(77)
    arrays = {}
    # mask rectangles: sides whose pads have fractional parts on both sides of .5, boxes that
    # touch every border; stored as the inclusive bounds of the painted region (-1 = nothing painted)
    mask_rows = []
    for side in (64, 100, 150, 256, 1000):
        for _ in range(12):
            w, h = int(rng.randint(1, side // 2)), int(rng.randint(1, side // 3))
            x, y = int(rng.randint(0, side - w + 1)), int(rng.randint(0, side - h + 1))
            mask = np.asarray(oc._draw_mask(oc.ImageResolution(side, side), [oc.BoundingBox(x, y, w, h)]))  # pylint: disable=protected-access
            ys, xs = np.nonzero(mask)
            assert set(np.unique(mask)) <= {0, 255}
            bounds = [xs.min(), ys.min(), xs.max(), ys.max()] if len(ys) else [-1, -1, -1, -1]
            if len(ys):
                assert mask[bounds[1] : bounds[3] + 1, bounds[0] : bounds[2] + 1].all()  # a full rectangle
            mask_rows.append([side, x, y, w, h] + [int(v) for v in bounds])
    arrays["mask_cases"] = np.array(mask_rows, dtype=np.int64)
    # one full composite
    fg = rng.randint(0, 256, (96, 96, 3)).astype(np.uint8)
    bg = rng.randint(0, 256, (96, 96, 3)).astype(np.uint8)
    boxes = [oc.BoundingBox(10, 20, 18, 9), oc.BoundingBox(60, 70, 30, 20), oc.BoundingBox(0, 0, 4, 3)]
    arrays["composite_seed"] = np.array([77], dtype=np.int64)
    arrays["composite_boxes"] = np.array(boxes, dtype=np.int64)
    arrays["composite_out"] = oc.write_boxes_onto_image(fg, bg, boxes)
    arrays["composite_fg"] = fg
    arrays["composite_bg"] = bg
    # bounding box distances
    dist_rows = []
    for _ in range(20):
        a = [oc.BoundingBox(*[int(v) for v in rng.randint(0, 200, 4)]) for _ in range(int(rng.randint(1, 4)))]
        b = [oc.BoundingBox(*[int(v) for v in rng.randint(0, 200, 4)]) for _ in range(int(rng.randint(1, 4)))]
        result = oc.bounding_box_distance(a, b)
        dist_rows.append((np.array(a), np.array(b), result.distance, np.array(result.a_box), np.array(result.b_box)))
    arrays["distance_count"] = np.array([len(dist_rows)], dtype=np.int64)
    for index, (a, b, dist, a_box, b_box) in enumerate(dist_rows):
        arrays[f"distance_{index}_a"] = a.astype(np.int64)
        arrays[f"distance_{index}_b"] = b.astype(np.int64)
        arrays[f"distance_{index}_result"] = np.concatenate(([dist], a_box, b_box)).astype(np.float64)
    assert oc.bounding_box_distance([], [oc.BoundingBox(0, 0, 1, 1)]) is None
    # track length filter
    tracks = rng.rand(8, 60) > 0.4
    arrays["tracks_in"] = tracks
    arrays["tracks_lengths"] = np.array([1, 2, 3, 4, 5, 6, 10, 60], dtype=np.int64)
    arrays["tracks_out"] = np.array(
        [vector_reduction.track_length_filter(pd.Series(row), int(length)) for row, length in zip(tracks, arrays["tracks_lengths"])]
    )
    np.savez_compressed(GOLDEN_DIR / "overlay.npz", **arrays)
    print(f"wrote overlay.npz  ({(GOLDEN_DIR / 'overlay.npz').stat().st_size / 1024:.0f} KiB)")


def unit_cases() -> None:
    """Known answers for the array helpers of vector_sources_common / vector_reduction."""
    from gance.vector_sources import vector_reduction, vector_sources_common as vsc  # pylint: disable=import-outside-toplevel,import-error

    rng = np.random.RandomState(42)
    data = rng.randn(6 * 64)
    rolls = rng.randint(0, 3, size=6)
    matrices = rng.randn(18, 5 * 32).astype(np.float32)
    ramp = rng.rand(40) * 3.0 + 1.0
    arrays = {
        "data": data,
        "rolls": rolls.astype(np.int64),
        "rotated": vsc.rotate_vectors_over_time(data, 64, rolls),
        "smooth_across_7_3": vsc.smooth_across_vectors(data, 8, 7, 3),  # 48 vectors of 8
        "smooth_each_5_3": vsc.smooth_each_vector(data, 64, 5, 3),
        "smooth_each_default": vsc.smooth_each_vector(data, 64),
        "resample_64_to_100": vsc.scale_vectors_to_length_resample(data, 64, 100),
        "resample_255_to_512": vsc.scale_vectors_to_length_resample(rng.randn(3 * 255), 255, 512),
        "resample_255_input": None,
        "duplicated_x3": vsc.duplicate_to_vector_count(data, 64, 18),
        "promoted": vsc.promote_to_matrix_duplicate(data[:64], 4),
        "matrices": matrices,
        "sub_vectors_matrix": vsc.sub_vectors(matrices, 32),
        "sub_vectors_vector": vsc.sub_vectors(data, 64),
        "demoted": vsc.demote_to_vector_select(matrices, 0),
        "ramp": ramp,
        "quantized_3": vector_reduction.quantize_results_layers(
            vector_reduction.ResultLayers(result=vector_reduction.DataLabel(ramp, "ramp")), [0, 1, 2]
        ).result.data.astype(np.int64),
        "quantized_5": vector_reduction.quantize_results_layers(
            vector_reduction.ResultLayers(result=vector_reduction.DataLabel(ramp, "ramp")), [0, 1, 2, 3, 4]
        ).result.data.astype(np.int64),
    }
    rng2 = np.random.RandomState(42)
    rng2.randn(6 * 64), rng2.randint(0, 3, size=6), rng2.randn(18, 5 * 32), rng2.rand(40)  # replay the stream
    arrays["resample_255_input"] = rng2.randn(3 * 255)
    assert np.array_equal(vsc.scale_vectors_to_length_resample(arrays["resample_255_input"], 255, 512), arrays["resample_255_to_512"])
    np.savez_compressed(GOLDEN_DIR / "vector_helpers.npz", **arrays)
    print("wrote vector_helpers.npz")


def standalone_cases() -> None:
    """
    The stand-alone stage functions, called one by one on the REFERENCE with small inputs: the cases of its
    own unit test (test/test_vector_sources_common.py:16-63) and non-default parameters of each function.
    """
    from gance import apply_spectrogram  # pylint: disable=import-outside-toplevel,import-error
    from gance.vector_sources import vector_reduction, vector_sources_common as vsc  # pylint: disable=import-outside-toplevel,import-error

    arrays = {}
    for original, count, output in ((10, 2, 50), (10, 1, 1000)):  # the reference's own parametrisation
        total = original * count
        unscaled = np.sin(np.linspace(start=0, stop=total - 1, num=total))
        arrays[f"scale_{original}x{count}_to_{output}"] = vsc.scale_vectors_to_length_resample(
            data=unscaled, original_vector_length=original, output_vector_length=output
        )
    rng = np.random.RandomState(77)
    vectors = rng.randn(11 * 24)
    arrays["vectors_11x24"] = vectors
    arrays["scale_24_to_9"] = vsc.scale_vectors_to_length_resample(vectors, 24, 9)  # down-sampling, even -> odd
    arrays["scale_24_to_16"] = vsc.scale_vectors_to_length_resample(vectors, 24, 16)  # even -> even (Nyquist doubled)
    arrays["across_5_2"] = vsc.smooth_across_vectors(vectors, 24, window_length=5, polyorder=2)
    arrays["across_default"] = vsc.smooth_across_vectors(vectors, 24)
    arrays["each_9_3"] = vsc.smooth_each_vector(vectors, 24, window_length=9, polyorder=3)
    arrays["each_3_1"] = vsc.smooth_each_vector(vectors, 24, window_length=3, polyorder=1)
    # polynomial orders above 3 (the reference hands any order to scipy.signal.savgol_filter)
    arrays["each_9_5"] = vsc.smooth_each_vector(vectors, 24, window_length=9, polyorder=5)
    arrays["each_11_8"] = vsc.smooth_each_vector(vectors, 24, window_length=11, polyorder=8)
    arrays["across_7_4"] = vsc.smooth_across_vectors(vectors, 24, window_length=7, polyorder=4)
    arrays["across_7_6"] = vsc.smooth_across_vectors(vectors, 24, window_length=7, polyorder=6)  # interpolating: the filter is the identity
    remap_in = rng.rand(37) * 7.0 - 2.0
    arrays["remap_in"] = remap_in
    arrays["remap_out"] = np.array(vsc.remap_values_into_range(remap_in, (-2.0, 5.0), (10.0, -3.0)))
    num_frames, L = 40, 512
    audio = synthetic.synthetic_audio(num_frames, L, seed=9)
    arrays["audio_frames_seed"] = np.array([num_frames, 9], dtype=np.int64)
    db = apply_spectrogram.compute_spectrogram(audio, L)
    arrays["db"] = db
    arrays["vectors_no_range"] = apply_spectrogram.reshape_spectrogram_to_vectors(db, L, None)
    arrays["vectors_range_0_3"] = apply_spectrogram.reshape_spectrogram_to_vectors(db, L, (0, 3))
    arrays["smooth_scale_m2_2"] = apply_spectrogram.compute_spectrogram_smooth_scale(audio, L, (-2, 2))
    stereo = np.stack([audio, audio[::-1]], axis=1)
    arrays["db_stereo"] = apply_spectrogram.compute_spectrogram(stereo, L)
    arrays["db_full"] = apply_spectrogram.compute_spectrogram(audio, L, truncate=False)  # all 510 bins of the two-sided spectrum
    layers_order_5 = vector_reduction.reduce_vector_rms_rolling_average(audio, L, rolling_average_window=3, savgol_window_length=9, savgol_polyorder=5)
    arrays["rms_smoothed_9_5"] = np.asarray(layers_order_5.result.data)
    layers = vector_reduction.reduce_vector_rms_rolling_average(audio, L, rolling_average_window=5, savgol_window_length=9, savgol_polyorder=2)
    arrays["rms_raw"] = np.asarray(layers.layers[1].data)
    arrays["rms_rolling_5"] = np.asarray(layers.layers[0].data)
    arrays["rms_smoothed_9_2"] = np.asarray(layers.result.data)
    arrays["rms_labels"] = np.array([layers.result.label, layers.layers[0].label, layers.layers[1].label])
    quantized = vector_reduction.quantize_results_layers(layers, [0, 1, 2, 3])
    arrays["rms_quantized_4"] = np.asarray(quantized.result.data).astype(np.int64)
    arrays["rms_quantized_label"] = np.array([quantized.result.label])
    np.savez_compressed(GOLDEN_DIR / "standalone_api.npz", **arrays)
    print("wrote standalone_api.npz")


def read_wav_cases() -> None:
    """`read_wav_file` of the reference (music.py:172-209) on generated PCM files: int16 / int32 mono, int16 stereo, float32."""
    import tempfile  # pylint: disable=import-outside-toplevel

    from scipy.io import wavfile  # pylint: disable=import-outside-toplevel

    from gance.vector_sources import music  # pylint: disable=import-outside-toplevel,import-error

    rng = np.random.RandomState(21)
    pcm16 = np.concatenate([np.array([-32768, 32767, 0, -1, 1], dtype=np.int16), rng.randint(-32768, 32768, size=1200).astype(np.int16)])
    pcm32 = np.concatenate(
        [np.array([-2147483648, 2147483647, 0, -1, 1], dtype=np.int32), rng.randint(-2147483648, 2147483647, size=900, dtype=np.int64).astype(np.int32)]
    )
    stereo16 = rng.randint(-32768, 32768, size=(700, 2)).astype(np.int16)
    float32 = (rng.rand(500).astype(np.float32) * 2 - 1)
    arrays = {"pcm16": pcm16, "pcm32": pcm32, "stereo16": stereo16, "float32": float32}
    with tempfile.TemporaryDirectory() as directory:
        for name, rate in (("pcm16", 44100), ("pcm32", 48000), ("stereo16", 22050), ("float32", 30720)):
            path = Path(directory) / f"{name}_clip.wav"
            wavfile.write(str(path), rate, arrays[name])
            result = music.read_wav_file(path)
            arrays[f"{name}_out"] = np.asarray(result.wav_data)
            arrays[f"{name}_meta"] = np.array([str(result.sample_rate), result.name, str(np.asarray(result.wav_data).dtype)])
    np.savez_compressed(GOLDEN_DIR / "read_wav.npz", **arrays)
    print("wrote read_wav.npz")


def main() -> None:
    ref_stubs.install()
    GOLDEN_DIR.mkdir(parents=True, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "standalone":
        standalone_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "read_wav":
        read_wav_cases()
        return
    unit_cases()
    standalone_cases()
    read_wav_cases()
    blend_case("blend_n60_seed0_roll_k3", 60, 0, True, 3, 13)
    blend_case("blend_n60_seed1_noroll_k1", 60, 1, False, 1, 13)
    blend_case("blend_n60_seed2_roll_k1", 60, 2, True, 1, 13)
    blend_case("blend_n240_seed3_roll_k3", 240, 3, True, 3, 53)
    blend_case("blend_n1800_seed7_roll_k3", 1800, 7, True, 3, 257)
    noise_case("noise_n60_seed0_roll_k3", 60, 0, True, 3, 7)
    noise_case("noise_n600_seed5_noroll_k2", 600, 5, False, 2, 97)
    overlay_cases()


if __name__ == "__main__":
    main()
