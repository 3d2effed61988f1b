from .sequence_dataset import SequenceDataset, SyntheticSequenceDataset  # noqa: F401
