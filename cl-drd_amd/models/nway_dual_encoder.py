"""``models.nway_dual_encoder.NwayDualEncoder`` of the reference (``models/nway_dual_encoder.py:6-67``), MI355X-native.

Same constructor, attributes (``query_encoder`` / ``passage_encoder``, aliased when ``share_weights``), methods
(``forward`` / ``query_embs`` / ``passage_embs`` / ``nway_passage_embs``) and state-dict keys
(``query_encoder.<HF names>`` / ``passage_encoder.<HF names>``), so reference checkpoints load after the usual
``module.`` stripping.  The towers are :class:`cldrd_amd.encoder.HipEncoder` instances; scoring is a HIP kernel
(no [B, B*N, D] gather is materialised for in-batch negatives).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip_ops as ops  # noqa: F401
from .. import torch_ops  # noqa: F401  (registers torch.ops.cldrd.*)
from ..encoder import HipEncoder, encode_autograd


def score_mode(in_batch_loss: bool, all_in_batch_neg: bool) -> int:
    return 0 if not in_batch_loss else (1 if all_in_batch_neg else 2)


def _lengths(mapping):
    """Host-side token counts of a batch mapping ("lengths": list / CPU tensor, any shape, one entry per sequence), or None."""
    ln = mapping.get("lengths") if hasattr(mapping, "get") else None
    if ln is None:
        return None
    if isinstance(ln, torch.Tensor):
        if ln.is_cuda:
            return None                      # a device tensor would cost the host sync packing is meant to avoid
        return ln.reshape(-1).tolist()
    return [int(v) for v in ln]


class NwayDualEncoder(nn.Module):
    def __init__(self, model_name_or_path, share_weights, in_batch_loss=False, all_in_batch_neg=True):
        super().__init__()
        self.model_name_or_path = model_name_or_path
        self.share_weights = share_weights
        self.in_batch_loss = in_batch_loss
        self.all_in_batch_neg = all_in_batch_neg

        self.query_encoder = HipEncoder.from_pretrained(model_name_or_path)
        if self.share_weights:
            self.passage_encoder = self.query_encoder
        else:
            self.passage_encoder = HipEncoder.from_pretrained(model_name_or_path)
        # fp16 mode (the default, encoder.HipEncoder: CLDRD_AMP): the query side runs its whole FORWARD on fp16 MFMA operands, evaluation
        # included (the query tower is ~1 % of the FLOPs, and a query's rounding error is shared by every logit of its row).  bf16 mode:
        # bf16 like everything else.  With shared weights the one tower keeps an fp16 shadow too and only query calls use it.
        self.query_fp16 = self.query_encoder.amp_mode == "fp16"
        self.query_encoder.hp_forward = self.query_fp16

    # -- reference call surface -------------------------------------------------------------------------------
    def forward(self, queries, nway_passages):
        """queries: each value [bz, seq_len]; nway_passages: each value [bz, nway, seq_len] -> logits [bz, N']."""
        query_reps = self.query_embs(queries)                          # [bz, D]
        nway_passage_reps = self.nway_passage_embs(nway_passages)      # [bz, nway, D]
        assert query_reps.dim() == 2 and nway_passage_reps.dim() == 3
        bz, nway, D = nway_passage_reps.shape
        mode = score_mode(self.in_batch_loss, self.all_in_batch_neg)
        return torch.ops.cldrd.nway_score(query_reps, nway_passage_reps.reshape(bz * nway, D), bz, nway, mode)

    def query_embs(self, queries):
        return encode_autograd(self.query_encoder, queries["input_ids"], queries.get("attention_mask"), fp16=self.query_fp16)

    def passage_embs(self, passages):
        # "lengths" (ours, optional: host-side token counts, one per sequence): the batch is packed (encoder.HipEncoder.encode)
        return encode_autograd(self.passage_encoder, passages["input_ids"], passages.get("attention_mask"), fp16=False,
                               lengths=_lengths(passages))

    def nway_passage_embs(self, nway_passages):
        input_ids, attention_mask = nway_passages["input_ids"], nway_passages["attention_mask"]
        bz, nway, seq_len = input_ids.shape
        input_ids, attention_mask = input_ids.reshape(bz * nway, seq_len), attention_mask.reshape(bz * nway, seq_len)
        passage_reps = encode_autograd(self.passage_encoder, input_ids, attention_mask, fp16=False, lengths=_lengths(nway_passages))
        return passage_reps.view(bz, nway, -1)

    # -- MI355X-specific ----------------------------------------------------------------------------------------
    def towers(self):
        return [self.query_encoder] if self.share_weights else [self.query_encoder, self.passage_encoder]

    def fuse_flat(self):
        """Put the parameters (and gradients) of all towers into one contiguous fp32 buffer each; returns
        (flat_p, flat_g).  Used by the fused trainer: one clip-norm pass, one AdamW launch, contiguous RCCL buckets."""
        tw = self.towers()
        dev = tw[0].flat_p.device
        total = sum(t.layout.total for t in tw)
        if getattr(self, "_flat_p", None) is not None and self._flat_p.device == dev and \
                tw[0].flat_p.data_ptr() == self._flat_p.data_ptr():
            return self._flat_p, self._flat_g
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._tower_offsets = []
        for t in tw:
            n = t.layout.total
            t.adopt_flat(flat_p[off:off + n], flat_g[off:off + n])
            self._tower_offsets.append(off)
            off += n
        self._flat_p, self._flat_g = flat_p, flat_g
        return flat_p, flat_g
