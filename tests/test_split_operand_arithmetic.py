"""
The arithmetic of the split-operand up kernels (gance_amd/csrc/upfir_split.hip, upfir_split_roles.hip), restated in numpy and pinned on
the CPU: why six bf16 x bf16 part products per fp32 product are "fp32 arithmetic" in the contract's sense.

* x = x0 + x1 + x2 EXACTLY, each part the round-to-nearest-even bfloat16 of what is left -- for every finite float32 whose parts stay
  normal (fp32's own exponent range: no overflow, the parts of a value are normal wherever 2^-17 of the value is).
* a bf16 x bf16 product has 16 significant bits: exact in the MFMA's fp32 accumulator; the three part products left out
  (x1 w2, x2 w1, x2 w2) are below 2^-24 |x w|.
* the six-term sum of a K = 4608 dot product (a 512-channel 3 x 3 layer), accumulated in float32, is as close to float64 as the float32
  MFMA's own summation order; three terms are not.
* the ring-column swizzle of the staging (upfir_split.hip: ring_column) is a bijection whose write groups (8 lanes, positions 4 apart)
  and read groups (16 neighbouring positions) each cover all 16-byte bank quads -- the LDS bank rules of MI355X_MICROARCH.md.
No GPU, no oracle/: the GPU-side parity of the same arithmetic is tests/test_synthesis_gpu.py::test_split_operand_*.
"""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bfloat16, returned as float32 (the host function of upfir_split.hip and v_cvt_pk_bf16_f32)."""
    bits = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rounded = (bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    x0 = bf16_rne(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16_rne(r1)
    r2 = (r1 - x1).astype(np.float32)
    return x0, x1, bf16_rne(r2), r1, r2


def test_three_bf16_parts_hold_a_float32_exactly() -> None:
    rng = np.random.RandomState(0)
    mantissas = rng.randint(0, 1 << 23, size=400_000).astype(np.uint32)
    exponents = rng.randint(20, 235, size=mantissas.size).astype(np.uint32)  # 2^-107 ... 2^108: the third part stays a normal number
    signs = rng.randint(0, 2, size=mantissas.size).astype(np.uint32)
    x = ((signs << 31) | (exponents << 23) | mantissas).view(np.float32)
    edge = np.array([1.0, -1.0, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 3.0e38, 1.1754944e-38 * 2.0**20, 0.0, np.float32(0.1), 65504.0, 2.0**-100], dtype=np.float32)
    x = np.concatenate([x, edge])
    x0, x1, x2, r1, r2 = split3(x)
    # the residuals are exact float32 subtractions (Sterbenz), the third residual is itself a bfloat16
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    assert np.array_equal(x2, r2)
    assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), x.astype(np.float64))
    # each part is at most half a unit of the part before it: 2^-8, 2^-16 of the value (round to nearest)
    nonzero = x != 0
    assert np.all(np.abs(x1[nonzero]) <= np.abs(x[nonzero]) * 2.0**-8 * (1 + 2.0**-7))
    assert np.all(np.abs(x2[nonzero]) <= np.abs(x[nonzero]) * 2.0**-16 * (1 + 2.0**-6))


def test_six_part_products_are_exact_and_the_three_left_out_are_below_fp32_resolution() -> None:
    rng = np.random.RandomState(1)
    x = (rng.randn(200_000) * 10.0 ** rng.uniform(-3, 3, size=200_000)).astype(np.float32)
    w = (rng.randn(200_000) * 10.0 ** rng.uniform(-3, 3, size=200_000)).astype(np.float32)
    xs, ws = split3(x)[:3], split3(w)[:3]
    terms = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]  # (x part, w part), smallest first: kTerms of the kernels
    total = np.zeros(x.size, dtype=np.float64)
    for i, j in terms:
        product64 = xs[i].astype(np.float64) * ws[j].astype(np.float64)
        assert np.array_equal((xs[i] * ws[j]).astype(np.float64), product64)  # 8 x 8 significant bits: exact in float32
        total += product64
    exact = x.astype(np.float64) * w.astype(np.float64)
    left_out = np.abs(exact - total)
    assert np.all(left_out <= np.abs(exact) * 2.0**-23)  # worst case of the three dropped terms; float32's own rounding of x w is 2^-24
    assert np.median(left_out / np.abs(exact)) < 2.0**-26


def _dot_fp32(a: np.ndarray, b: np.ndarray, chunk: int) -> np.ndarray:
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for k in range(0, a.shape[1], chunk):
        acc = (acc + (a[:, k : k + chunk].astype(np.float32) @ b[k : k + chunk].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return acc


def test_six_terms_match_float64_as_closely_as_the_fp32_matrix_cores_do() -> None:
    rng = np.random.RandomState(2)
    K, M, N = 4608, 16, 64  # a 512-channel 3 x 3 layer's sum per output
    w = (rng.randn(M, K) * 10.0 ** rng.uniform(-1, 1, size=(M, 1))).astype(np.float32)
    x = rng.randn(K, N).astype(np.float32)
    want = w.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(want).max()
    fp32_error = np.abs(_dot_fp32(w, x, 4) - want).max() / scale  # v_mfma_f32_16x16x4_f32: k-steps of four
    ws, xs = split3(w)[:3], split3(x)[:3]

    def split_sum(terms) -> float:
        acc = np.zeros((M, N), dtype=np.float32)
        for k in range(0, K, 32):  # v_mfma_f32_16x16x32_bf16: k-steps of 32, a term at a time
            for i, j in terms:
                acc = (acc + (ws[j][:, k : k + 32] @ xs[i][k : k + 32]).astype(np.float32)).astype(np.float32)
        return float(np.abs(acc - want).max() / scale)

    six = split_sum([(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])
    three = split_sum([(1, 0), (0, 1), (0, 0)])
    assert six < 2e-6 and six <= 2.0 * fp32_error, (six, fp32_error)  # (measured: 4e-7 against 1.2e-6)
    assert three > 4.0 * six  # three terms lose the x1 w1 / x0 w2 / x2 w0 products: not float32 any more


def ring_column(p: int) -> int:
    """upfir_split.hip::ring_column: c0 = P2, c1 = P3, c2 = P4 ^ P0, c3 = P1, c4 = P0, c5 = P5."""
    return ((p >> 2) & 3) | ((((p >> 4) ^ p) & 1) << 2) | (((p >> 1) & 1) << 3) | ((p & 1) << 4) | (p & 32)


def test_ring_column_swizzle_is_a_bijection_without_bank_conflicts() -> None:
    columns = [ring_column(p) for p in range(64)]
    assert sorted(columns) == list(range(64))
    # ds_write_b128 groups: 8 neighbouring lanes of one lane row r = positions 4 g + r, g = g0 .. g0 + 7 (g0 in {0, 8}): 8 distinct bank quads of 32 banks
    for r in range(4):
        for g0 in (0, 8):
            assert len({ring_column(4 * g + r) % 8 for g in range(g0, g0 + 8)}) == 8
    # ds_read_b128 groups: 16 neighbouring positions (a wave's tile column): 16 distinct bank quads of 64 banks
    for first in range(0, 64, 16):
        assert len({ring_column(first + n) % 16 for n in range(16)}) == 16
    # ... and shifted by one position (the dx = -1 fragments): one pair of lanes at most shares a quad
    for first in range(16, 64, 16):
        quads = [ring_column(first - 1 + n) % 16 for n in range(16)]
        assert len(set(quads)) >= 15
