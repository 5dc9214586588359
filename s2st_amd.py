"""Import alias: ``import s2st_amd`` == the ``speech-to-speech-translation_amd`` package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("speech-to-speech-translation_amd")
sys.modules[__name__] = _pkg
