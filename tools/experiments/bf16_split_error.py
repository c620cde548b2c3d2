"""
CPU emulation (not product code): error of an fp32 dot product computed on bf16 matrix cores from SPLIT operands.

x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): three bf16 numbers hold an fp32 number's 24
mantissa bits exactly (8 bits each). A product x * w is then the sum of nine bf16 x bf16 products (each exact in fp32),
accumulated in fp32 like the fp32 MFMA's. Dropping the smallest terms trades accuracy for matrix-core passes:
  3 terms: x0 w0 + x0 w1 + x1 w0                       (relative error per product ~ 2^-16)
  6 terms: + x1 w1 + x0 w2 + x2 w0                     (~ 2^-24: the terms left out are <= 2^-25 |x w|)
  9 terms: all                                         (exact products, fp32 accumulation only)
The bf16 MFMAs of gfx950 run at 16x the fp32 MFMA rate (2.5 PFLOP/s against 157 TFLOP/s), so k terms cost k / 16 of the
fp32 time: 0.19 / 0.375 / 0.56. This script measures what the K = 4608 sums of a 512-channel 3x3 convolution lose, against fp64:

    python tools/experiments/bf16_split_error.py
"""
import numpy as np


def bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bfloat16, returned as float32."""
    bits = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    rounded = (bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split(x: np.ndarray):
    x0 = bf16(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16(r1)
    x2 = bf16((r1 - x1).astype(np.float32))
    return x0, x1, x2


def split_f16(x: np.ndarray):
    """Two fp16 parts (22 mantissa bits in four bytes; fp16's exponent range: residuals below 6e-5 are subnormals)."""
    h0 = x.astype(np.float16).astype(np.float32)
    h1 = (x - h0).astype(np.float32).astype(np.float16).astype(np.float32)
    return h0, h1


def dot_fp32(a: np.ndarray, b: np.ndarray, chunk: int = 4) -> np.ndarray:
    """[M][K] x [K][N] with fp32 accumulation in MFMA order: k-steps of `chunk` products added to the running sum."""
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for k in range(0, a.shape[1], chunk):
        acc = (acc + (a[:, k : k + chunk].astype(np.float32) @ b[k : k + chunk].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return acc


def main() -> None:
    rng = np.random.RandomState(0)
    K, M, N = 4608, 64, 256
    for label, w_scale in (("weights N(0,1), activations N(0,1)", 1.0), ("weights log-normal x10^+-1 per row", None)):
        w = rng.randn(M, K).astype(np.float32)
        if w_scale is None:
            w *= (10.0 ** rng.uniform(-1, 1, size=(M, 1))).astype(np.float32)
        x = rng.randn(K, N).astype(np.float32)
        want = w.astype(np.float64) @ x.astype(np.float64)
        scale = np.abs(want).max()
        fp32 = dot_fp32(w, x)
        ws, xs = split(w), split(x)
        terms = {3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)], 9: [(i, j) for i in range(3) for j in range(3)]}
        print(label)
        print(f"  fp32 MFMA order           max |err| / max |y| = {np.abs(fp32 - want).max() / scale:.2e}")
        for count, pairs in terms.items():
            acc = np.zeros_like(fp32)
            # smallest terms first, as a kernel would issue them (the large term last keeps the small ones' bits)
            for i, j in sorted(pairs, key=lambda ij: -(ij[0] + ij[1])):
                acc = (acc + dot_fp32(ws[i], xs[j], chunk=32)).astype(np.float32)
            print(f"  bf16 split, {count} terms       max |err| / max |y| = {np.abs(acc - want).max() / scale:.2e}   (matrix time x{count / 16:.3f} of fp32)")
        # two fp16 parts, three terms; the second line with the weights as small as runtime-scaled Winograd weights are (x 1e-3): the
        # residuals become subnormals -- what a power-of-two scale of the weight image repairs (gemm_forms.hip split_weight_scale)
        for note, factor in (("", 1.0), (" weights x 1e-3", 1e-3), (" weights x 1e-3, stored x 4096", 1e-3 * 4096)):
            (w0, w1), (x0, x1) = split_f16((w * factor).astype(np.float32)), split_f16(x)
            acc = (dot_fp32(w1, x0, chunk=32) + dot_fp32(w0, x1, chunk=32)).astype(np.float32)
            acc = (acc + dot_fp32(w0, x0, chunk=32)).astype(np.float32)
            print(f"  fp16 split, 3 terms{note:32s} max |err| / max |y| = {np.abs(acc - want * factor).max() / (scale * factor):.2e}   (matrix time x0.188)")


if __name__ == "__main__":
    main()
