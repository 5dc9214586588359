"""Stand-in (see ../README.md): only `librosa.filters.mel`, answered by this repository's restatement."""
from . import filters  # noqa: F401
