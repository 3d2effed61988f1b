"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy mirror of the counter-based dropout mask of the HIP kernels
(cl-drd_amd/csrc/common.h: mix32 / drop_rowkey / drop_pair / dropout_keep).

The reference draws its masks from torch's Philox stream inside nn.Dropout (HF DistilBERT Embeddings / MultiHeadSelfAttention /
FFN, reached from models/nway_dual_encoder.py:52-64); a stream cannot be reproduced across implementations, so parity for
the dropout path is: (a) with p = 0 everything matches the reference bit-for-tolerance (tests/golden), and (b) with p > 0
the kernels apply exactly THIS mask, identically in forward and backward, with the reference's 1/(1-p) scaling - which
the GPU tests check against a torch computation that uses the mask from here.

    rowkey(seed, row) = mix32(row + (mix32(lo32(seed)) ^ hi32(seed) * 0x9E3779B9))
    h(row, col)       = mix32(rowkey ^ (col >> 1))
    keep(row, col)    = (h >> 16 if col odd else h & 0xFFFF) >= round(p * 65536)
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M
    x ^= x >> np.uint64(16)
    return x


def thresh16(p: float) -> int:
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def rowkey(seed: int, rows):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    sk = int(mix32(seed & 0xFFFFFFFF)) ^ (((seed >> 32) * 0x9E3779B9) & 0xFFFFFFFF)
    return mix32((np.asarray(rows, dtype=np.uint64) + np.uint64(sk)) & _M)


def keep_mask(seed: int, p: float, n_rows: int, n_cols: int, row0: int = 0) -> np.ndarray:
    """bool [n_rows, n_cols]: True where the element survives dropout(p) for mask rows row0 .. row0 + n_rows - 1."""
    rk = rowkey(seed, np.arange(row0, row0 + n_rows, dtype=np.uint64))[:, None]
    col = np.arange(n_cols, dtype=np.uint64)[None, :]
    h = mix32(rk ^ (col >> np.uint64(1)))
    bits = np.where((col & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return bits >= np.uint64(thresh16(p))


def attention_keep_mask(seed: int, p: float, nseq: int, H: int, L: int) -> np.ndarray:
    """bool [nseq, H, L(query), L(key)]: mask row ((seq*H + head)*L + query), mask column key."""
    return keep_mask(seed, p, nseq * H * L, L).reshape(nseq, H, L, L)
