"""Stand-in (see ../README.md): only `torchaudio.transforms.MFCC`, answered by this repository's restatement."""
from . import transforms  # noqa: F401
