"""COCO RLE: the oracle restatement against hand-worked strings and its own independent decoder, and the library's HOST
compressor (nopesac_rle_compress_host - no device work) against the oracle."""
import numpy as np
import pytest

from oracle import rle_oracle as R


def test_hand_worked_strings():
    assert R.encode(np.zeros((2, 2), bool))["counts"] == b"4"          # one run of 4 zeros
    assert R.encode(np.ones((2, 2), bool))["counts"] == b"04"          # empty zero run, then 4 ones
    m = np.zeros((3, 3), bool); m[1, 1] = True
    assert R.encode(m)["counts"] == b"414"                             # column-major index 4
    assert R.to_string([40]) == b"X1"                                  # 40 = 8 + (1 << 5): 'X' = 48+8+32, '1' = 48+1
    assert R.to_string([5, 3, 2, 1]) == b"532N"                        # 4th count stored as 1-3 = -2 -> 0b11110 -> 'N'
    assert R.from_string(b"532N") == [5, 3, 2, 1]
    assert R.from_string(b"X1") == [40]


def test_column_major_order_and_bbox():
    m = np.zeros((4, 6), bool)
    m[1:3, 2:5] = True                                                  # rows 1-2, cols 2-4
    rle = R.encode(m)
    assert R.run_lengths(m) == [9, 2, 2, 2, 2, 2, 5]
    assert R.to_bbox(rle).tolist() == [2.0, 1.0, 3.0, 2.0]
    assert R.to_bbox(R.encode(np.zeros((4, 6), bool))).tolist() == [0, 0, 0, 0]
    full = np.ones((4, 6), bool)
    assert R.to_bbox(R.encode(full)).tolist() == [0, 0, 6, 4]
    col = np.zeros((4, 6), bool); col[2:, 1] = True; col[:2, 2] = True  # one run wrapping over a column boundary
    assert R.to_bbox(R.encode(col)).tolist() == [1, 0, 2, 4]


@pytest.mark.parametrize("seed,h,w,p", [(0, 7, 5, 0.5), (1, 48, 64, 0.1), (2, 48, 64, 0.9), (3, 1, 9, 0.5), (4, 9, 1, 0.5),
                                        (5, 120, 160, 0.02)])
def test_round_trip(seed, h, w, p):
    rng = np.random.default_rng(seed)
    m = rng.random((h, w)) < p
    if seed == 5:                                                       # long runs -> multi-char counts, negative deltas
        m = np.zeros((h, w), bool); m[10:90, 20:100] = True; m[30:40, 50:60] = False
    rle = R.encode(m)
    assert np.array_equal(R.decode(rle), m)
    ys, xs = np.nonzero(m)
    if len(xs):
        assert R.to_bbox(rle).tolist() == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1]


@pytest.mark.parametrize("seed,h,w", [(0, 7, 5), (1, 48, 64), (2, 480, 640), (3, 1, 9), (4, 3, 3)])
def test_host_compressor_matches_oracle(seed, h, w):
    from nopesac_amd import rle
    rng = np.random.default_rng(seed)
    if seed == 2:
        m = np.zeros((h, w), bool); m[100:400, 50:600] = True; m[200:210] = False; m[0, 0] = True; m[-1, -1] = True
    elif seed == 4:
        m = np.zeros((h, w), bool)
    else:
        m = rng.random((h, w)) < 0.4
    flat = m.reshape(-1, order="F").astype(np.int8)
    pos = np.flatnonzero(np.diff(np.concatenate([[0], flat])) != 0).astype(np.uint32)
    s, bbox = rle.compress(pos, h, w)
    ref = R.encode(m)
    assert s == ref["counts"]
    assert bbox == R.to_bbox(ref).tolist()
