from .synthetic import (  # noqa: F401
    SyntheticFisherCorpus,
    batch_by_size,
    collate,
    BOS, PAD, EOS, UNK,
)
