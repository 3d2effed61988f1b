"""Run-file writer (reference retriever/retrieve_top_passages.py:90-107): the native bulk formatter must produce the bytes of the
reference's f-string loop (oracle/retrieval_ref.py: run_file_lines), including Python's float repr of the fp32 score."""
import ctypes as C
import os

import numpy as np
import pytest

from cldrd_amd import _lib
from cldrd_amd.retriever.retrieve_top_passages import write_run_file
from oracle import retrieval_ref as R


def _lib_or_skip():
    try:
        return _lib.load()
    except Exception as e:      # the library is built by __graft_entry__.build(); without it there is nothing to test here
        pytest.skip(f"libcldrd_hip.so not built: {e}")


def test_float_repr_matches_python_on_many_values():
    lib = _lib_or_skip()
    rng = np.random.default_rng(7)
    vals = np.concatenate([
        (rng.standard_normal(100000) * 30).astype(np.float32), (rng.standard_normal(20000) * 1e-5).astype(np.float32),
        rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32).view(np.float32),
        np.float32([0.0, -0.0, 1.0, -1.0, 100.0, 1e16, 1e15, 9.999999e15, 1e-4, 9.9e-5, 1e-5, 123456789.0, np.inf, -np.inf, 3.4e38,
                    1e-45, 12.5, 1e22, 0.1, 16777216.0, 2.5e-4, 99999.99])])
    buf = C.create_string_buffer(40)
    for v in vals:
        if np.isnan(v):
            continue
        n = lib.cldrd_py_float_repr(float(v), buf)
        assert buf.raw[:n].decode() == repr(float(v))


def test_bulk_writer_equals_the_reference_loop_on_a_million_lines(tmp_path):
    _lib_or_skip()
    rng = np.random.default_rng(11)
    nq, k = 1100, 1000                                         # 1.1 M lines
    qids = rng.permutation(10 ** 7)[:nq].astype(np.int64) + 3
    I = rng.integers(-1, 8841823, size=(nq, k)).astype(np.int64)
    D = -np.sort(-(rng.standard_normal((nq, k)) * 2 + 17).astype(np.float32), axis=1)
    D[5, -3:] = -np.inf                                        # missing results: id -1, score -inf (faiss contract)
    I[5, -3:] = -1
    path = tmp_path / "run" / "dev.run"
    n = write_run_file(str(path), qids.tolist(), I, D)
    assert n == nq * k
    want = "".join(R.run_file_lines(qids.tolist(), I.tolist(), D.tolist()))
    assert path.read_bytes() == want.encode()
    # one thread / many threads: same bytes
    p2 = tmp_path / "dev1.run"
    write_run_file(str(p2), qids.tolist(), I, D, nthreads=1)
    assert p2.read_bytes() == want.encode()


def test_list_inputs_and_duplicate_query_ids_follow_the_reference_dict(tmp_path):
    """The reference collects hits in a dict keyed by qid (:90-96): a repeated qid extends the first entry and the ranks run on."""
    qids = [7, 3, 7]
    I = np.array([[10, 11], [20, 21], [30, 31]], dtype=np.int64)
    D = np.array([[2.5, 1.5], [9.0, 8.0], [0.5, 0.25]], dtype=np.float32)
    path = tmp_path / "dup.run"
    n = write_run_file(str(path), qids, I, D)
    assert n == 6
    assert path.read_text() == "7\t10\t1\t2.5\n7\t11\t2\t1.5\n7\t30\t3\t0.5\n7\t31\t4\t0.25\n3\t20\t1\t9.0\n3\t21\t2\t8.0\n"
    # nested lists, unique ids: the plain loop, same text as the oracle
    p2 = tmp_path / "lists.run"
    write_run_file(str(p2), [1, 2], I[:2].tolist(), D[:2].tolist())
    assert p2.read_text() == "".join(R.run_file_lines([1, 2], I[:2].tolist(), D[:2].tolist()))
