#!/usr/bin/env python3
"""``python train.py DATA --task s2s_translation --arch s2st_transformer --criterion s2st_loss ...``: the counterpart of
``python -m fairseq_cli.train`` for the MI355X path (speech-to-speech-translation_amd/train.py)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s2st_amd  # noqa: E402,F401

if __name__ == "__main__":
    importlib.import_module("speech-to-speech-translation_amd.train").cli_main()
