"""CPU oracle for the CL-DRD hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference algorithm
(``models/nway_dual_encoder.py``, ``losses/*.py``, ``retriever/retrieval_utils.py`` and the
trainer step of ``trainer/multistep-curriculum/nway_listwise_1.py``).  It exists to *check* the HIP
path; it is never the thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package (``cl-drd_amd/``) does not
import anything from here and fails loudly when its HIP library is missing.

Parity status:
  * losses, encoder forward/backward, N-way scoring: PINNED against golden vectors produced by
    importing the reference (and the HuggingFace encoder it calls) in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), including the known-answer values
    printed by the reference's own ``__main__`` demos.
  * linear-warmup schedule: PINNED against ``transformers.get_linear_schedule_with_warmup``.
  * legacy ``transformers.AdamW`` step: PARITY UNPINNED (class removed from transformers 5.x; the
    restatement follows its published update rule).
  * faiss ``IndexFlatIP.search``: PARITY UNPINNED (faiss is not installed and not vendored; the
    restatement follows the documented exact inner-product contract and the call sites
    ``retriever/retrieval_utils.py:131-153``).
"""
