"""oracle/dropout_ref.py (the numpy mirror of the kernels' counter-based dropout mask): known answers of the 32-bit mixer
(lowbias32: 0x7feb352d / 0x846ca68b, shifts 16-15-16), frozen mask bits, and the statistics a dropout mask must have.
The GPU tests (tests/test_gpu_kernels.py) check that the HIP kernels apply exactly this mask in forward and backward."""
import numpy as np

from oracle import dropout_ref as D


def test_mixer_known_answers():
    assert D.mix32([0, 1, 2, 0xFFFFFFFF]).tolist() == [0, 1753845952, 3507691905, 1734902346]
    assert D.rowkey(1234, [0, 1]).tolist() == [1887047106, 1744608857]
    assert D.rowkey((7 << 32) | 99, [5]).tolist() == [4207172906]          # the high seed word enters the key
    assert D.thresh16(0.1) == 6554 and D.thresh16(0.5) == 32768 and D.thresh16(0.0) == 0


def test_frozen_mask_bits():
    bits = np.packbits(D.keep_mask(1234, 0.1, 2, 64)).tolist()
    assert bits == [255, 247, 126, 223, 255, 255, 255, 255, 255, 255, 255, 63, 191, 247, 255, 255]
    # row0 offsets address the same infinite mask
    assert np.array_equal(D.keep_mask(1234, 0.1, 3, 50, row0=7), D.keep_mask(1234, 0.1, 10, 50)[7:])
    assert np.array_equal(D.attention_keep_mask(9, 0.3, 2, 3, 16).reshape(-1, 16), D.keep_mask(9, 0.3, 2 * 3 * 16, 16))


def test_mask_statistics():
    for p in (0.1, 0.25, 0.5):
        m = D.keep_mask(77, p, 2048, 768)
        assert abs(m.mean() - (1 - p)) < 2e-3
        assert np.abs(m.mean(0) - (1 - p)).max() < 0.05 and np.abs(m.mean(1) - (1 - p)).max() < 0.07
        pairs = np.corrcoef(m[:, 0::2].ravel(), m[:, 1::2].ravel())[0, 1]          # the two halves of one hash word
        rows = np.corrcoef(m[0::2].ravel(), m[1::2].ravel())[0, 1]
        seeds = np.corrcoef(m.ravel(), D.keep_mask(78, p, 2048, 768).ravel())[0, 1]
        assert max(abs(pairs), abs(rows), abs(seeds)) < 5e-3
