"""Wrapper of the data config YAML: same properties and lookup rules as
``examples/s2s_trans/data/data_cfg.py:10-194`` (``S2STDataConfig``)."""
from __future__ import annotations

from copy import deepcopy
from pathlib import Path
from typing import Dict, Optional

import yaml


class S2STDataConfig:
    def __init__(self, yaml_path: Path):
        yaml_path = Path(yaml_path)
        if not yaml_path.is_file():
            raise FileNotFoundError(f"{yaml_path.as_posix()} not found")
        try:
            with open(yaml_path) as f:
                self.config = yaml.load(f, Loader=yaml.FullLoader) or {}
        except Exception as e:
            raise Exception(f"Failed to load config from {yaml_path.as_posix()}: {e}")
        self.root = yaml_path.parent
        self.use_hubert = False
        self.kd_encoder = False

    def _auto_convert_to_abs_path(self, x):
        if isinstance(x, str):
            if not Path(x).exists() and (self.root / x).exists():
                return (self.root / x).as_posix()
        elif isinstance(x, dict):
            return {k: self._auto_convert_to_abs_path(v) for k, v in x.items()}
        return x

    def _get(self, key, default):
        return self.config.get(key, default)

    src_vocab_filename = property(lambda s: s._get("src_vocab_filename", "dict.txt"))
    tgt_vocab_filename = property(lambda s: s._get("tgt_vocab_filename", "dict.txt"))
    speaker_set_filename = property(lambda s: s._get("speaker_set_filename", None))
    shuffle = property(lambda s: s._get("shuffle", False))
    prepend_tgt_lang_tag = property(lambda s: s._get("prepend_tgt_lang_tag", False))
    input_feat_per_channel = property(lambda s: s._get("input_feat_per_channel", 80))
    input_channels = property(lambda s: s._get("input_channels", 1))
    sample_rate = property(lambda s: s._get("sample_rate", 16_000))
    sampling_alpha = property(lambda s: s._get("sampling_alpha", 1.0))
    use_audio_input = property(lambda s: s._get("use_audio_input", False))
    use_sample_rate = property(lambda s: s._get("use_sample_rate", 16000))
    audio_root = property(lambda s: s._get("audio_root", ""))
    vocoder = property(lambda s: s._get("vocoder", None))

    @property
    def pre_tokenizer(self) -> Dict:
        return self._auto_convert_to_abs_path(self._get("pre_tokenizer", {"tokenizer": None}))

    @property
    def bpe_tokenizer(self) -> Dict:
        return self._auto_convert_to_abs_path(self._get("bpe_tokenizer", {"bpe": None}))

    def set_use_hubert(self, use_hubert):
        self.use_hubert = use_hubert

    def set_kd_encoder(self, kd_encoder):
        self.kd_encoder = kd_encoder

    def _split_transforms(self, key: str, split: str, is_train: bool):
        """Split-specific transform list: exact split name, then `_train` / `_eval`, then `*`."""
        cfg = deepcopy(self.config)
        table = cfg.get(key, {})
        cur = table.get(split)
        cur = table.get("_train") if cur is None and is_train else cur
        cur = table.get("_eval") if cur is None and not is_train else cur
        cur = table.get("*") if cur is None else cur
        cfg[key] = cur
        return cfg

    def get_feature_transforms(self, split, is_train):
        return self._split_transforms("transforms", split, is_train)

    def get_feature_transforms_for_src(self, split, is_train):
        return self._split_transforms("src_transforms", split, is_train)

    def get_feature_transforms_for_tgt(self, split, is_train):
        return self._split_transforms("tgt_transforms", split, is_train)

    @property
    def src_global_cmvn_stats_npz(self) -> Optional[str]:
        return self._auto_convert_to_abs_path(self.config.get("src_global_cmvn", {}).get("stats_npz_path", None))

    @property
    def tgt_global_cmvn_stats_npz(self) -> Optional[str]:
        return self._auto_convert_to_abs_path(self.config.get("tgt_global_cmvn", {}).get("stats_npz_path", None))
