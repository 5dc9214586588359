"""On-disk dataset of the ``s2s_translation_mtl`` task.

Counterpart of ``examples/s2s_trans/data/s2st_dataset_mtl.py:42-347`` -- the reference keeps a SEPARATE dataset class for
the mtl variant, and its batches differ from ``s2st_dataset.py``'s in integer content, not just in key names:

* ``src_text`` has its EOS removed (:192-195: ``src_text[:len(src_text) - 1]`` -- the source-text CTC targets of
  ``s2st_loss_mtl.py:159-167`` therefore end on the last word, and ``src_text_len`` is one less than the base task's);
  with ``prepend_tgt_lang_tag`` the SOURCE language tag goes in front (:196-200);
* no ``prev_src_text_tokens`` / ``src_txt_ntokens`` / HuBERT waveform keys (:322-344), ``source_texts`` is added (:318);
* ``prev_tgt_text_tokens`` is collated in SAMPLE order: the reference forgets the ``index_select(0, order)`` every other
  tensor gets (:290-296).  Reproduced as is -- the mtl model has no text decoder that would read it.

Manifest columns: the base set minus ``src_orig`` / ``tgt_text_orig``; the optional FastSpeech-style columns
``duration`` / ``pitch`` / ``energy`` (:389-405) are collated like the reference does (:298-311).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .audio_utils import get_features_or_waveform
from .s2st_dataset import S2STDataset, S2STDatasetCreator, _collate_frames, collate_tokens


@dataclass
class S2STMTLDatasetItem:  # TextToSpeechDatasetItem of s2st_dataset_mtl.py:28-39
    index: int
    src_speech: torch.Tensor
    tgt_speech: torch.Tensor
    src_text: Optional[torch.Tensor] = None
    tgt_text: Optional[torch.Tensor] = None
    speaker_id: Optional[int] = None
    duration: Optional[torch.Tensor] = None
    pitch: Optional[torch.Tensor] = None
    energy: Optional[torch.Tensor] = None


class S2STMTLDataset(S2STDataset):
    def __init__(self, *args, durations=None, pitches=None, energies=None, **kw):
        super().__init__(*args, **kw)
        self.durations, self.pitches, self.energies = durations, pitches, energies

    def __getitem__(self, index: int) -> S2STMTLDatasetItem:
        it = super().__getitem__(index)
        src_text = it.src_text
        if self.src_texts is not None:
            # :192-200 -- encode with EOS, drop it, THEN prepend the language tag (the base class' _encode prepends
            # before we get to drop, so the item is rebuilt here in the reference's order)
            t = self.src_dict.encode_line(self._tokenized(self.src_texts[index]), add_if_not_exist=False, append_eos=True).long()
            t = t[: len(t) - 1]
            if self.cfg.prepend_tgt_lang_tag:
                idx = self.src_dict.index(self.LANG_TAG_TEMPLATE.format(self.src_langs[index]))
                assert idx != self.src_dict.unk()
                t = torch.cat((torch.LongTensor([idx]), t), 0)
            src_text = t
        duration = pitch = energy = None
        if self.durations is not None:
            duration = torch.tensor(self.durations[index] + [0], dtype=torch.long)  # pad 0 for EOS
        if self.pitches is not None:
            pitch = torch.from_numpy(np.concatenate((get_features_or_waveform(self.pitches[index]), [0]))).float()
        if self.energies is not None:
            energy = torch.from_numpy(np.concatenate((get_features_or_waveform(self.energies[index]), [0]))).float()
        return S2STMTLDatasetItem(index=index, src_speech=it.src_speech, tgt_speech=it.tgt_speech, src_text=src_text,
                                  tgt_text=it.tgt_text, speaker_id=it.speaker_id, duration=duration, pitch=pitch,
                                  energy=energy)

    def collater(self, samples: List[S2STMTLDatasetItem]) -> Dict[str, Any]:
        """s2st_dataset_mtl.py:242-347."""
        if len(samples) == 0:
            return {}
        src_lens, order = torch.tensor([s.src_speech.shape[0] for s in samples], dtype=torch.long).sort(descending=True)
        sel = lambda t: t.index_select(0, order)  # noqa: E731
        sd, td = self.src_dict, self.tgt_dict
        id_ = sel(torch.tensor([s.index for s in samples], dtype=torch.long))
        src_feat = sel(_collate_frames([s.src_speech for s in samples], self.cfg.use_audio_input))
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
        # (sample order, NOT sorted: :290-296)
        prev_tgt = collate_tokens([s.tgt_text for s in samples], td.pad(), td.eos(), move_eos_to_beginning=True)
        durations = pitches = energies = None
        if self.durations is not None:
            durations = sel(collate_tokens([s.duration for s in samples], 0, None))
            assert src_text.shape[1] == durations.shape[1]
        if self.pitches is not None:
            pitches = sel(_collate_frames([s.pitch for s in samples], True))
            assert src_text.shape[1] == pitches.shape[1]
        if self.energies is not None:
            energies = sel(_collate_frames([s.energy for s in samples], True))
            assert src_text.shape[1] == energies.shape[1]
        ol = order.tolist()
        return {
            "id": id_,
            "net_input": {"src_speech": src_feat, "src_speech_lens": src_lens, "prev_output_tokens": prev,
                          "prev_tgt_text_tokens": prev_tgt},
            "speaker": speaker, "src_text": src_text, "src_text_len": src_text_len, "tgt_text": tgt_text,
            "tgt_text_len": tgt_text_len, "tgt_speech": tgt_feat, "target_lengths": tgt_lens,
            "durations": durations, "pitches": pitches, "energies": energies,
            "ntokens": int(tgt_lens.sum().item()), "tgt_txt_ntokens": int(tgt_text_len.sum().item()),
            "nsentences": len(samples),
            "source_texts": [sd.string(samples[i].src_text) for i in ol],
            "target_texts": [td.string(samples[i].tgt_text) for i in ol],
        }


class S2STMTLDatasetCreator(S2STDatasetCreator):
    """s2st_dataset_mtl.py:350-468."""
    KEY_DURATION, KEY_PITCH, KEY_ENERGY = "duration", "pitch", "energy"

    @classmethod
    def _from_list(cls, split, is_train_split, samples, cfg, src_dict, tgt_dict, pre_tokenizer, bpe_tokenizer,
                   n_frames_per_step, speaker_to_id) -> S2STMTLDataset:
        root = Path(cfg.audio_root)
        col = lambda k: [(root / s[k]).as_posix() for s in samples]  # noqa: E731
        opt = lambda k: [s.get(k, cls.DEFAULT) for s in samples]  # noqa: E731
        durations = [s.get(cls.KEY_DURATION, None) for s in samples]
        durations = [None if dd is None else [int(d) for d in dd.split(" ")] for dd in durations]
        durations = None if any(dd is None for dd in durations) else durations

        def paths(key):
            v = [s.get(key, None) for s in samples]
            v = [None if p is None else (root / p).as_posix() for p in v]
            return None if any(p is None for p in v) else v

        return S2STMTLDataset(
            split, is_train_split, cfg, col(cls.KEY_SRC_AUDIO), None, col(cls.KEY_TGT_AUDIO),
            [int(s[cls.KEY_SRC_N_FRAMES]) for s in samples], [int(s[cls.KEY_TGT_N_FRAMES]) for s in samples],
            src_texts=opt(cls.KEY_SRC_TEXT), tgt_texts=[s[cls.KEY_TGT_TEXT] for s in samples], speakers=opt(cls.KEY_SPEAKER),
            src_langs=opt(cls.KEY_SRC_LANG), tgt_langs=opt(cls.KEY_TGT_LANG), ids=[s[cls.KEY_ID] for s in samples],
            src_dict=src_dict, tgt_dict=tgt_dict, pre_tokenizer=pre_tokenizer, bpe_tokenizer=bpe_tokenizer,
            n_frames_per_step=n_frames_per_step, speaker_to_id=speaker_to_id, durations=durations,
            pitches=paths(cls.KEY_PITCH), energies=paths(cls.KEY_ENERGY))
