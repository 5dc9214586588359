"""On-disk dataset of the s2s_translation task: TSV manifest -> items -> collated batches.

Counterpart of ``examples/s2s_trans/data/s2st_dataset.py:46-525`` (``S2STDataset``, ``S2STDatasetCreator``) over
``fairseq/data/audio/speech_to_text_dataset.py`` (``pack_frames`` :232-237, ``ordered_indices`` :352-360,
``num_tokens`` / ``size`` :339-343, ``_load_samples_from_tsv`` :443-461) -- host-side byte / integer work, restated
so that the produced batch dict is identical (same keys, dtypes, ordering, padding) to the reference collater's.
Manifest columns: id, src_audio, tgt_audio, src_n_frames, tgt_n_frames, tgt_text and optionally src_text,
src_orig (HuBERT mode waveform), speaker, src_lang, tgt_lang, tgt_text_orig.  Audio columns hold ``.npy`` paths or
``<zip>:<offset>:<length>`` byte ranges of an uncompressed zip (relative to ``audio_root`` of the config).
"""
from __future__ import annotations

import csv
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .audio_utils import get_features_or_waveform, read_waveform
from .data_cfg import S2STDataConfig
from .dictionary import Dictionary
from .feature_transforms import CompositeAudioFeatureTransform
from .synthetic import batch_by_size


@dataclass
class S2STDatasetItem:
    index: int
    src_speech: torch.Tensor
    tgt_speech: torch.Tensor
    src_text: Optional[torch.Tensor] = None
    tgt_text: Optional[torch.Tensor] = None
    src_orig: Optional[torch.Tensor] = None
    tgt_text_orig: Optional[str] = None
    speaker_id: Optional[int] = None


def _collate_frames(frames: List[torch.Tensor], is_audio_input: bool = False) -> torch.Tensor:
    """speech_to_text_dataset.py:97-116: zero-padded [n, max_len(, feat)]."""
    max_len = max(f.size(0) for f in frames)
    out = frames[0].new_zeros((len(frames), max_len) if is_audio_input else (len(frames), max_len, frames[0].size(1)))
    for i, v in enumerate(frames):
        out[i, : v.size(0)] = v
    return out


def collate_tokens(values: List[torch.Tensor], pad_idx: int, eos_idx: int, move_eos_to_beginning: bool = False):
    """fairseq/data/data_utils.py:35-70 with left_pad=False."""
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        dst = res[i][: len(v)]
        if move_eos_to_beginning:
            dst[0] = eos_idx  # (eos_idx given: the last token is not inspected)
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res


class S2STDataset:
    LANG_TAG_TEMPLATE = "<lang:{}>"

    def __init__(self, split: str, is_train_split: bool, cfg: S2STDataConfig, src_audio_paths: List[str],
                 src_orig_paths: Optional[List[str]], tgt_audio_paths: List[str], src_n_frames: List[int],
                 tgt_n_frames: List[int], src_texts=None, tgt_texts=None, tgt_text_orig=None, speakers=None,
                 src_langs=None, tgt_langs=None, ids=None, src_dict: Optional[Dictionary] = None,
                 tgt_dict: Optional[Dictionary] = None, pre_tokenizer=None, bpe_tokenizer=None, n_frames_per_step=1,
                 speaker_to_id=None, max_sample_size=9600000, random_crop=False, pad_audio=True, normalize=False):
        self.split, self.is_train_split, self.cfg = split, is_train_split, cfg
        self.n_samples = len(src_audio_paths)
        assert len(src_n_frames) == self.n_samples > 0
        assert tgt_texts is None or len(tgt_texts) == self.n_samples
        assert (tgt_dict is None and tgt_texts is None) or (tgt_dict is not None and tgt_texts is not None)
        self.src_audio_paths, self.src_orig_paths, self.tgt_audio_paths = src_audio_paths, src_orig_paths, tgt_audio_paths
        self.n_frames = self.src_n_frames = src_n_frames  # batching cost = source frames (:339-340)
        self.tgt_n_frames = tgt_n_frames
        self.src_texts, self.tgt_texts, self.tgt_text_orig = src_texts, tgt_texts, tgt_text_orig
        self.speakers, self.src_langs, self.tgt_langs, self.ids = speakers, src_langs, tgt_langs, ids
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        self.pre_tokenizer, self.bpe_tokenizer = pre_tokenizer, bpe_tokenizer
        self.n_frames_per_step, self.speaker_to_id = n_frames_per_step, speaker_to_id
        self.shuffle = cfg.shuffle if is_train_split else False
        self.max_sample_size, self.random_crop, self.pad_audio, self.normalize = max_sample_size, random_crop, pad_audio, normalize
        self.feature_transforms_src = self._transforms("src_transforms", cfg.get_feature_transforms_for_src(split, is_train_split))
        self.feature_transforms_tgt = self._transforms("tgt_transforms", cfg.get_feature_transforms_for_tgt(split, is_train_split))
        if cfg.prepend_tgt_lang_tag:
            tags = [self.LANG_TAG_TEMPLATE.format(t) for t in set(self.tgt_langs)]
            assert all(t in self.tgt_dict for t in tags)
        self.tgt_lens = self._text_lens(self.tgt_texts)
        self.src_lens = self._text_lens(self.src_texts)

    def _transforms(self, key, config):
        config = {k: (self.cfg._auto_convert_to_abs_path(v) if isinstance(v, dict) else v) for k, v in config.items()}
        return CompositeAudioFeatureTransform.from_config_dict(config, key)

    def _text_lens(self, texts):
        if texts is None:
            return [0] * self.n_samples
        return [len(self._tokenized(t).split(" ")) for t in texts]

    @staticmethod
    def tokenize(tokenizer, text: str):
        return text if tokenizer is None else tokenizer.encode(text)

    def _tokenized(self, text: str) -> str:
        return self.tokenize(self.bpe_tokenizer, self.tokenize(self.pre_tokenizer, text))

    def pack_frames(self, feature: torch.Tensor) -> torch.Tensor:
        if self.n_frames_per_step == 1:
            return feature
        n = feature.shape[0] // self.n_frames_per_step
        return feature[: self.n_frames_per_step * n].reshape(n, -1)

    def _encode(self, text: str, dictionary: Dictionary, lang: Optional[str]) -> torch.Tensor:
        t = dictionary.encode_line(self._tokenized(text), add_if_not_exist=False, append_eos=True).long()
        if self.cfg.prepend_tgt_lang_tag:
            idx = dictionary.index(self.LANG_TAG_TEMPLATE.format(lang))
            assert idx != dictionary.unk()
            t = torch.cat((torch.LongTensor([idx]), t), 0)
        return t

    def get_audio(self, wav_path: str) -> torch.Tensor:
        wav, _ = read_waveform(wav_path)
        wav = torch.from_numpy(np.asarray(wav)).float()
        if wav.dim() == 2:
            wav = wav.mean(-1)
        assert wav.dim() == 1
        if self.normalize:
            with torch.no_grad():
                wav = F.layer_norm(wav, wav.shape)
        return wav

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index: int) -> S2STDatasetItem:
        src_orig = self.get_audio(self.src_orig_paths[index]) if self.cfg.use_hubert else None
        feats = []
        for path, tf in ((self.src_audio_paths[index], self.feature_transforms_src),
                         (self.tgt_audio_paths[index], self.feature_transforms_tgt)):
            x = get_features_or_waveform(path, need_waveform=self.cfg.use_audio_input,
                                         use_sample_rate=self.cfg.use_sample_rate)
            if tf is not None:
                assert not self.cfg.use_audio_input
                x = tf(x)
            feats.append(torch.from_numpy(x).float())
        src_speech, tgt_speech = feats[0], self.pack_frames(feats[1])  # only the target is frame-stacked (:189)
        tgt_text = None if self.tgt_texts is None else self._encode(self.tgt_texts[index], self.tgt_dict,
                                                                   self.tgt_langs[index] if self.tgt_langs else None)
        src_text = None if self.src_texts is None else self._encode(self.src_texts[index], self.src_dict,
                                                                   self.src_langs[index] if self.src_langs else None)
        spk = None if self.speaker_to_id is None else self.speaker_to_id[self.speakers[index]]
        return S2STDatasetItem(index=index, src_speech=src_speech, tgt_speech=tgt_speech, src_text=src_text,
                               tgt_text=tgt_text, src_orig=src_orig,
                               tgt_text_orig=None if self.tgt_text_orig is None else self.tgt_text_orig[index],
                               speaker_id=spk)

    # -- batching -------------------------------------------------------------------------------------------
    def num_tokens(self, index):
        return self.n_frames[index]

    def size(self, index):
        return self.n_frames[index], self.tgt_lens[index]

    @property
    def sizes(self):
        return np.array(self.n_frames)

    def ordered_indices(self):
        order = [np.random.permutation(len(self))] if self.shuffle else [np.arange(len(self))]
        order.append([-n for n in self.n_frames])  # longest first, ties in original / random order
        return np.lexsort(order)

    def filter_indices_by_size(self, indices, max_positions):
        """fairseq_dataset.py:143-180: drop items whose (src frames, target TEXT length) exceed the limits."""
        ms, mt = max_positions
        keep = [i for i in indices if self.n_frames[i] <= ms and self.tgt_lens[i] <= mt]
        ignored = [i for i in indices if not (self.n_frames[i] <= ms and self.tgt_lens[i] <= mt)]
        return np.array(keep, dtype=np.int64), ignored

    def batches(self, max_tokens: int = 20000, max_sentences: int = 0, bsz_mult: int = 8, max_positions=None):
        idx = self.ordered_indices()
        if max_positions is not None:
            idx, _ = self.filter_indices_by_size(idx, max_positions)
        ntok = np.array([self.n_frames[i] for i in idx], dtype=np.int64)
        return batch_by_size(idx, ntok, max_tokens, max_sentences, bsz_mult)

    def collate_batch(self, indices) -> Dict[str, Any]:
        return self.collater([self[int(i)] for i in indices])

    # -- collater (s2st_dataset.py:326-455) -----------------------------------------------------------------
    def _collater_audio(self, audios, audio_size):
        out = audios[0].new_zeros(len(audios), audio_size)
        mask = torch.zeros(out.shape, dtype=torch.bool)
        for i, a in enumerate(audios):
            diff = len(a) - audio_size
            if diff == 0:
                out[i] = a
            elif diff < 0:
                assert self.pad_audio
                out[i] = torch.cat([a, a.new_full((-diff,), 0.0)])
                mask[i, diff:] = True
            else:
                start = np.random.randint(0, diff + 1) if self.random_crop else 0
                out[i] = a[start:start + audio_size]
        return out, mask

    def collater(self, samples: List[S2STDatasetItem]) -> Dict[str, Any]:
        if len(samples) == 0:
            return {}
        src_lens, order = torch.tensor([s.src_speech.shape[0] for s in samples], dtype=torch.long).sort(descending=True)
        sel = lambda t: t.index_select(0, order)  # noqa: E731
        id_ = sel(torch.tensor([s.index for s in samples], dtype=torch.long))
        src_feat = None if self.cfg.use_hubert else sel(_collate_frames([s.src_speech for s in samples], self.cfg.use_audio_input))
        audios, pad_mask = None, None
        if self.cfg.use_hubert:
            wavs = [s.src_orig for s in samples]
            sizes = [w.size(0) for w in wavs]
            size = min(max(sizes) if self.pad_audio else min(sizes), self.max_sample_size)
            audios, pad_mask = self._collater_audio(wavs, size)
            audios, pad_mask = sel(audios), sel(pad_mask)
        sd, td = self.src_dict, self.tgt_dict
        src_text = sel(collate_tokens([s.src_text for s in samples], sd.pad(), sd.eos()))
        src_text_len = sel(torch.tensor([s.src_text.size(0) for s in samples], dtype=torch.long))
        tgt_lens = sel(torch.tensor([s.tgt_speech.shape[0] for s in samples], dtype=torch.long))
        tgt_feat = sel(_collate_frames([s.tgt_speech for s in samples], self.cfg.use_audio_input))
        tgt_text = sel(collate_tokens([s.tgt_text for s in samples], td.pad(), td.eos()))
        tgt_text_len = sel(torch.tensor([s.tgt_text.size(0) for s in samples], dtype=torch.long))
        speaker = None
        if self.speaker_to_id is not None:
            speaker = sel(torch.tensor([s.speaker_id for s in samples], dtype=torch.long)).view(-1, 1)
        bsz, _, d = tgt_feat.size()
        prev = torch.cat((tgt_feat.new_zeros((bsz, 1, d)), tgt_feat[:, :-1, :]), dim=1)
        prev_src = sel(collate_tokens([s.src_text for s in samples], sd.pad(), sd.eos(), move_eos_to_beginning=True))
        prev_tgt = sel(collate_tokens([s.tgt_text for s in samples], td.pad(), td.eos(), move_eos_to_beginning=True))
        ol = order.tolist()
        return {
            "id": id_,
            "net_input": {"src_speech": src_feat, "src_speech_lens": src_lens, "prev_output_tokens": prev,
                          "prev_src_text_tokens": prev_src, "prev_tgt_text_tokens": prev_tgt,
                          "collated_audios_orig": audios, "padding_mask": pad_mask, "speaker": speaker},
            "speaker": speaker, "src_text": src_text, "src_text_len": src_text_len, "tgt_text": tgt_text,
            "tgt_text_len": tgt_text_len, "tgt_speech": tgt_feat, "target_lengths": tgt_lens,
            "durations": None, "pitches": None, "energies": None,
            "ntokens": int(tgt_lens.sum().item()), "src_txt_ntokens": int(src_text_len.sum().item()),
            "tgt_txt_ntokens": int(tgt_text_len.sum().item()), "nsentences": len(samples),
            "target_texts": [td.string(samples[i].tgt_text) for i in ol],
            "tgt_text_orig": [samples[i].tgt_text_orig for i in ol],
        }


class S2STDatasetCreator:
    KEY_ID, KEY_SRC_AUDIO, KEY_SRC_ORIG, KEY_TGT_AUDIO = "id", "src_audio", "src_orig", "tgt_audio"
    KEY_SRC_N_FRAMES, KEY_TGT_N_FRAMES = "src_n_frames", "tgt_n_frames"
    TGT_ORIG_TXT, KEY_SRC_TEXT, KEY_TGT_TEXT = "tgt_text_orig", "src_text", "tgt_text"
    KEY_SPEAKER, KEY_SRC_LANG, KEY_TGT_LANG = "speaker", "src_lang", "tgt_lang"
    DEFAULT = ""

    @classmethod
    def _load_samples_from_tsv(cls, root: str, split: str) -> List[Dict]:
        tsv_path = Path(root) / f"{split}.tsv"
        if not tsv_path.is_file():
            raise FileNotFoundError(f"Dataset not found: {tsv_path}")
        with open(tsv_path) as f:
            reader = csv.DictReader(f, delimiter="\t", quotechar=None, doublequote=False, lineterminator="\n",
                                    quoting=csv.QUOTE_NONE)
            samples = [dict(e) for e in reader]
        if len(samples) == 0:
            raise ValueError(f"Empty manifest: {tsv_path}")
        return samples

    @classmethod
    def _from_list(cls, split, is_train_split, samples, cfg, src_dict, tgt_dict, pre_tokenizer, bpe_tokenizer,
                   n_frames_per_step, speaker_to_id) -> S2STDataset:
        root = Path(cfg.audio_root)
        col = lambda k: [(root / s[k]).as_posix() for s in samples]  # noqa: E731
        opt = lambda k: [s.get(k, cls.DEFAULT) for s in samples]  # noqa: E731
        return S2STDataset(
            split, is_train_split, cfg, col(cls.KEY_SRC_AUDIO), col(cls.KEY_SRC_ORIG) if cfg.use_hubert else None,
            col(cls.KEY_TGT_AUDIO), [int(s[cls.KEY_SRC_N_FRAMES]) for s in samples],
            [int(s[cls.KEY_TGT_N_FRAMES]) for s in samples], src_texts=opt(cls.KEY_SRC_TEXT),
            tgt_texts=[s[cls.KEY_TGT_TEXT] for s in samples],
            tgt_text_orig=[s[cls.TGT_ORIG_TXT] for s in samples] if cfg.kd_encoder else None,
            speakers=opt(cls.KEY_SPEAKER), src_langs=opt(cls.KEY_SRC_LANG), tgt_langs=opt(cls.KEY_TGT_LANG),
            ids=[s[cls.KEY_ID] for s in samples], src_dict=src_dict, tgt_dict=tgt_dict, pre_tokenizer=pre_tokenizer,
            bpe_tokenizer=bpe_tokenizer, n_frames_per_step=n_frames_per_step, speaker_to_id=speaker_to_id)

    @classmethod
    def from_tsv(cls, root: str, cfg: S2STDataConfig, splits: str, src_dict, tgt_dict, pre_tokenizer, bpe_tokenizer,
                 is_train_split: bool, epoch: int, seed: int, n_frames_per_step: int = 1, speaker_to_id=None):
        names = splits.split(",")
        if len(names) > 1:
            # the reference's creator fails here as well: it uses ResamplingDataset / ConcatDataset without
            # importing them (examples/s2s_trans/data/s2st_dataset.py:577-589 -> NameError)
            raise NotImplementedError("multi-split (comma-separated) training sets are not supported by the s2s_translation "
                                      "dataset creator")
        return cls._from_list(names[0], is_train_split, cls._load_samples_from_tsv(root, names[0]), cfg, src_dict,
                              tgt_dict, pre_tokenizer, bpe_tokenizer, n_frames_per_step, speaker_to_id)
