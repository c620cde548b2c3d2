"""
StyleGAN2 config-f generator: static description (layer table, variable names, shapes), the
random-init weight generator used by the benchmark, and the flat weight-blob packing that the
C-ABI `gance_engine_create` consumes (`include/gance_hip.h`).

The reference loads a TF1 pickle (`gance/network_interface/network_functions.py:108-110`,
`pickle.load(f)[2]`) of the un-vendored NVlabs StyleGAN2 `Network` class. Variable names and
shapes here follow that class's published variable naming (`G_synthesis/64x64/Conv0_up/weight`,
`.../mod_weight`, `G_mapping/Dense0/weight`, `dlatent_avg`, `G_synthesis/noise3` ...), so a legacy
importer only has to copy arrays by name. Nothing here touches a GPU.
"""

from typing import Dict, List, NamedTuple, Tuple

import numpy as np

DLATENT_SIZE = 512
LATENT_SIZE = 512
MAPPING_LAYERS = 8
MAPPING_LRMUL = 0.01
NUM_CHANNELS = 3
FMAP_BASE = 16 << 10
FMAP_MAX = 512

Variables = Dict[str, np.ndarray]


def nf(stage: int, fmap_base: int = FMAP_BASE, fmap_max: int = FMAP_MAX) -> int:
    """Feature maps at a stage (stage = res_log2 - 1). Published formula, fmap_decay=1, fmap_min=1."""
    return int(np.clip(int(fmap_base / (2.0 ** stage)), 1, fmap_max))


class ConvLayer(NamedTuple):
    """One modulated 3x3 conv of G_synthesis."""

    layer_idx: int  # also the dlatent row it reads, and the index of its noise buffer
    scope: str  # TF variable scope under G_synthesis/
    res_log2: int  # OUTPUT resolution log2
    cin: int
    cout: int
    up: bool


class ToRGBLayer(NamedTuple):
    """One modulated 1x1 conv without demodulation."""

    scope: str
    res_log2: int
    cin: int
    dlatent_row: int  # res_log2 * 2 - 3


class SynthesisSpec(NamedTuple):
    """Static shape of a generator."""

    resolution: int
    res_log2: int
    num_layers: int  # dlatent rows W: 18 for 1024, 14 for 256
    convs: Tuple[ConvLayer, ...]
    torgbs: Tuple[ToRGBLayer, ...]


def make_spec(resolution: int) -> SynthesisSpec:
    """
    Layer table of G_synthesis_stylegan2 (skip architecture) for a power-of-two resolution.
    dlatent rows: 4x4 conv -> 0, ToRGB(r) -> 2r-3, Conv0_up(r) -> 2r-5, Conv1(r) -> 2r-4.
    """
    res_log2 = int(np.log2(resolution))
    if resolution != 2 ** res_log2 or resolution < 8:
        raise ValueError(f"resolution must be a power of two >= 8, got {resolution}")
    convs: List[ConvLayer] = [ConvLayer(0, "4x4/Conv", 2, nf(1), nf(1), False)]
    torgbs: List[ToRGBLayer] = [ToRGBLayer("4x4/ToRGB", 2, nf(1), 1)]
    for res in range(3, res_log2 + 1):
        side = 2 ** res
        convs.append(
            ConvLayer(res * 2 - 5, f"{side}x{side}/Conv0_up", res, nf(res - 2), nf(res - 1), True)
        )
        convs.append(
            ConvLayer(res * 2 - 4, f"{side}x{side}/Conv1", res, nf(res - 1), nf(res - 1), False)
        )
        torgbs.append(ToRGBLayer(f"{side}x{side}/ToRGB", res, nf(res - 1), res * 2 - 3))
    return SynthesisSpec(
        resolution=resolution,
        res_log2=res_log2,
        num_layers=res_log2 * 2 - 2,
        convs=tuple(convs),
        torgbs=tuple(torgbs),
    )


def variable_shapes(spec: SynthesisSpec) -> Dict[str, Tuple[int, ...]]:
    """Every variable of the generator, by TF name, in blob order (see `pack_variables`)."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for i in range(MAPPING_LAYERS):
        shapes[f"G_mapping/Dense{i}/weight"] = (LATENT_SIZE if i == 0 else DLATENT_SIZE, DLATENT_SIZE)
        shapes[f"G_mapping/Dense{i}/bias"] = (DLATENT_SIZE,)
    shapes["dlatent_avg"] = (DLATENT_SIZE,)
    shapes["G_synthesis/4x4/Const/const"] = (1, nf(1), 4, 4)
    for conv in spec.convs:
        scope = f"G_synthesis/{conv.scope}"
        shapes[f"{scope}/weight"] = (3, 3, conv.cin, conv.cout)
        shapes[f"{scope}/mod_weight"] = (DLATENT_SIZE, conv.cin)
        shapes[f"{scope}/mod_bias"] = (conv.cin,)
        shapes[f"{scope}/noise_strength"] = ()
        shapes[f"{scope}/bias"] = (conv.cout,)
    for rgb in spec.torgbs:
        scope = f"G_synthesis/{rgb.scope}"
        shapes[f"{scope}/weight"] = (1, 1, rgb.cin, NUM_CHANNELS)
        shapes[f"{scope}/mod_weight"] = (DLATENT_SIZE, rgb.cin)
        shapes[f"{scope}/mod_bias"] = (rgb.cin,)
        shapes[f"{scope}/bias"] = (NUM_CHANNELS,)
    for conv in spec.convs:
        side = 2 ** conv.res_log2
        shapes[f"G_synthesis/noise{conv.layer_idx}"] = (1, 1, side, side)
    return shapes


def make_random_variables(resolution: int, seed: int = 0, perturb: bool = False) -> Variables:
    """
    Random-init generator, the way the published TF code initialises it (equalised learning
    rate): conv / dense / mod weights N(0,1) (mapping weights N(0, 1/lrmul)), const N(0,1), noise
    buffers N(0,1), all biases 0, mod_bias 0, noise_strength 0, dlatent_avg 0.

    `perturb=True` is for parity tests only: it fills biases, noise strengths and dlatent_avg with
    small random values so that every term of every kernel is exercised (with the plain random
    init the noise and bias terms are identically zero).
    """
    spec = make_spec(resolution)
    rng = np.random.RandomState(seed)
    variables: Variables = {}
    for name, shape in variable_shapes(spec).items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf in ("bias", "mod_bias", "noise_strength") or name == "dlatent_avg":
            value = np.zeros(shape, dtype=np.float32)
            if perturb:
                scale = {"bias": 0.1, "mod_bias": 0.1, "noise_strength": 0.05}.get(leaf, 0.2)
                if name.startswith("G_mapping"):
                    scale = 5.0  # runtime-multiplied by lrmul=0.01
                value = np.asarray(rng.randn(*shape) * scale, dtype=np.float32)
        elif name.startswith("G_mapping") and leaf == "weight":
            value = np.asarray(rng.randn(*shape) / MAPPING_LRMUL, dtype=np.float32)
        else:
            value = np.asarray(rng.randn(*shape), dtype=np.float32)
        variables[name] = value
    return variables


def make_stress_variables(resolution: int, seed: int = 0) -> Variables:
    """
    A generator with the STATISTICS of a trained network, for parity tests only (no trained pickle exists in the
    reference tree; the legacy importer loads real ones): what a random init never shows a kernel. Conv / ToRGB weights
    are heavy-tailed (Student-t, 3 degrees of freedom, unit variance) with log-normal per-input-channel and
    per-output-channel scales spanning 10^-1 ... 10^+1; `mod_bias` ~ N(0, 3) so that |style| reaches ~10; noise strengths
    uniform in +-[0.1, 1]; biases uniform in [-2, 2]; `dlatent_avg` ~ N(0, 0.5); mapping biases as `perturb=True`.
    Winograd forms amplify rounding by the norms of their transforms (F(4x4,3x3), points 0, +-1, +-2: ~10 x 20 per
    layer), which is why the kernels' accuracy is stated on this network beside the random-init one.
    """
    spec = make_spec(resolution)
    rng = np.random.RandomState(seed)
    variables: Variables = {}

    def log_scale(count: int) -> np.ndarray:
        return np.exp(np.clip(rng.randn(count) * 1.15, -np.log(10.0), np.log(10.0)))

    for name, shape in variable_shapes(spec).items():
        leaf = name.rsplit("/", 1)[-1]
        if name.startswith("G_mapping"):
            value = rng.randn(*shape) / MAPPING_LRMUL if leaf == "weight" else rng.randn(*shape) * 5.0
        elif name == "dlatent_avg":
            value = rng.randn(*shape) * 0.5
        elif leaf == "weight":  # [k, k, cin, cout]
            value = rng.standard_t(3, size=shape) / np.sqrt(3.0)
            value = value * log_scale(shape[2])[None, None, :, None] * log_scale(shape[3])[None, None, None, :]
            if "ToRGB" in name:
                value = value * 0.01  # (keeps the image inside a few units, as a trained generator's: most uint8 values unsaturated)
        elif leaf == "mod_bias":
            value = rng.randn(*shape) * 3.0
        elif leaf == "noise_strength":
            value = np.asarray(rng.uniform(0.1, 1.0) * rng.choice([-1.0, 1.0]))
        elif leaf == "bias":
            value = rng.uniform(-2.0, 2.0, size=shape) * (0.1 if "ToRGB" in name else 1.0)
        else:  # mod_weight, const, noise buffers
            value = rng.randn(*shape)
        variables[name] = np.asarray(value, dtype=np.float32).reshape(shape)
    return variables


def pack_variables(variables: Variables, spec: SynthesisSpec) -> np.ndarray:
    """
    Flatten the raw (un-scaled) variables into the float32 blob `gance_engine_create` takes.
    Order = `variable_shapes(spec)` order, each array C-contiguous in its TF shape. The engine
    applies the equalised-LR runtime coefficients itself.
    """
    parts = []
    for name, shape in variable_shapes(spec).items():
        array = np.asarray(variables[name], dtype=np.float32)
        if tuple(array.shape) != tuple(shape):
            raise ValueError(f"variable {name}: expected shape {shape}, got {array.shape}")
        parts.append(np.ascontiguousarray(array).reshape(-1))
    return np.concatenate(parts)


def blob_size(spec: SynthesisSpec) -> int:
    """Number of floats in the packed blob."""
    return int(sum(int(np.prod(shape)) for shape in variable_shapes(spec).values()))
