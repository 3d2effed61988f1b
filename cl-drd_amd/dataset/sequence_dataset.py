"""Encode-side input contract of the index path (reference ``dataset/sequence_dataset.py:31-55``): batches
``{"seq": {"input_ids", "attention_mask"} int64 [b, L], "id": list[int]}``.

``SequenceDataset`` reads the reference's ``id\\ttext`` TSV and tokenises in ``collate_fn`` with any HuggingFace-style
tokenizer (needs a local vocab: there is no network here).  ``SyntheticSequenceDataset`` produces MSMARCO-shaped token ids
from the portable generator, which is what the benchmarks and tests use."""
from __future__ import annotations

import torch

from .. import synthetic as syn


class SequenceDataset(torch.utils.data.Dataset):
    def __init__(self, ids, seqs, tokenizer, max_length, is_query=False):
        self.ids, self.seqs, self.tokenizer, self.max_length, self.is_query = ids, seqs, tokenizer, int(max_length), is_query

    @classmethod
    def create_from_seqs_file(cls, path, tokenizer, max_length, is_query=False):
        ids, seqs = [], []
        with open(path) as fh:
            for line in fh:
                a = line.rstrip("\n").split("\t")
                if len(a) < 2:
                    continue
                ids.append(int(a[0]))
                seqs.append(a[1])
        return cls(ids, seqs, tokenizer, max_length, is_query)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self.ids[i], self.seqs[i]

    def collate_fn(self, batch):
        ids, seqs = zip(*batch)
        enc = self.tokenizer(list(seqs), padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
        return {"seq": {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}, "id": list(ids)}


class SyntheticSequenceDataset(torch.utils.data.Dataset):
    """n MSMARCO-shaped sequences of up to ``max_length`` tokens; deterministic in (seed, index range)."""

    def __init__(self, n, max_length, seed=99, vocab=syn.VOCAB, ragged=True, first_id=0, batch_size=512):
        self.n, self.max_length, self.seed, self.vocab, self.ragged = int(n), int(max_length), seed, vocab, ragged
        self.first_id, self.batch_size = first_id, batch_size

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def __getitem__(self, b):
        lo = b * self.batch_size
        rows = min(self.batch_size, self.n - lo)
        batch = syn.seq_batch(self.seed + 7919 * b, rows, self.max_length, vocab=self.vocab, ragged=self.ragged,
                              first_id=self.first_id + lo)
        if self.ragged:      # pad to the longest sequence of the batch, as the HF tokenizer does (padding=True)
            longest = int(batch["seq"]["attention_mask"].sum(1).max())
            batch["seq"] = {k: v[:, :longest].contiguous() for k, v in batch["seq"].items()}
        return batch

    def loader(self):
        """Iterate whole batches (already collated)."""
        return torch.utils.data.DataLoader(self, batch_size=None, shuffle=False, num_workers=0)
