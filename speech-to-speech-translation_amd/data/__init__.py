from .synthetic import (  # noqa: F401
    SyntheticFisherCorpus,
    batch_by_size,
    collate,
    BOS, PAD, EOS, UNK,
)
from .dictionary import Dictionary  # noqa: F401,E402
from .data_cfg import S2STDataConfig  # noqa: F401,E402
from .s2st_dataset import S2STDataset, S2STDatasetCreator  # noqa: F401,E402
from .s2st_dataset_mtl import S2STMTLDataset, S2STMTLDatasetCreator  # noqa: F401,E402
from .iterators import EpochBatchIterator, numpy_seed  # noqa: F401,E402
