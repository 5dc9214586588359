"""MI355X-native s2st_transformer training path (drop-in for the reference's
``examples/s2s_trans`` plugin surface: task ``s2s_translation``, arch
``s2st_transformer``, criterion ``s2st_loss``).

The directory name contains hyphens, so import it with
``importlib.import_module("speech-to-speech-translation_amd")`` or through the
``s2st_amd`` alias module at the repo root.
"""
__version__ = "0.1.0"

# ``--user-dir <this directory>``: fairseq imports the package and expects the plugin modules to register themselves
# (examples/s2s_trans/__init__.py does ``from . import tasks, criterions, models``)
from . import tasks, criterions, models  # noqa: E402,F401
