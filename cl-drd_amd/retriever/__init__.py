from .retrieval_utils import (FlatIPIndex, ShardedFlatIPIndex, construct_flatindex_from_embeddings, convert_index_to_gpu,  # noqa: F401
                              get_embeddings_from_scratch, index_retrieve, merge_shard_results, read_index, write_index)
