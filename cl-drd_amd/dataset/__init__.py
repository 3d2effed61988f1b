from .sequence_dataset import CachedSequenceDataset, SequenceDataset, SequenceTokenCache, SyntheticSequenceDataset  # noqa: F401
from .nway_dataset import LABEL_MODES, NwayDataset, TokenCache, labels_for_mode  # noqa: F401
