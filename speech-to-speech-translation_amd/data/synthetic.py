"""Synthetic "Fisher-shaped" fbank80 -> mel80 corpus and the batch (sample) schema.

This is the host-side data boundary of the hot path.  It mirrors

* the sample dict produced by ``S2STDataset.collater``
  (reference: examples/s2s_trans/data/s2st_dataset.py:329-455),
* n-frames-per-step packing (fairseq/data/audio/speech_to_text_dataset.py:234-239),
* ``collate_tokens(..., move_eos_to_beginning=True)`` for the text "prev" inputs
  (fairseq/data/data_utils.py:35-80),
* the ``batch_by_size_vec`` packing rule (fairseq/data/data_utils_fast.pyx:20-100),

but reads no files: utterances are generated from a seed, the
pattern of fairseq/benchmark/dummy_mt.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# fairseq Dictionary special symbols (fairseq/data/dictionary.py:27-40)
BOS, PAD, EOS, UNK = 0, 1, 2, 3


@dataclass
class Utterance:
    index: int
    src_speech: np.ndarray  # [S, 80] float32 (post-CMVN fbank)
    tgt_speech: np.ndarray  # [D, 80 * n_frames_per_step] float32 (packed log-mel)
    src_text: np.ndarray  # [Ls] int64, ends with EOS
    tgt_text: np.ndarray  # [Lt] int64, ends with EOS
    src_audio: Optional[np.ndarray] = None  # [N] float32 16 kHz (HuBERT mode)


class SyntheticFisherCorpus:
    """Seeded corpus with Fisher-like length statistics.

    ``src_n_frames = clip(round(lognormal(ln 300, 0.6)), 40, 3000)``,
    ``tgt_n_frames = max(16, round(0.6 * src * U(0.85, 1.15)))``, packed to
    ``D = tgt // n_frames_per_step`` decoder steps, text lengths ~0.12*src / 0.15*tgt (+EOS).
    Features are generated lazily per utterance from ``seed`` and the utterance index.
    """

    def __init__(
        self,
        n_utts: int = 4096,
        seed: int = 1234,
        n_frames_per_step: int = 4,
        feat_dim: int = 80,
        src_vocab: int = 44,
        tgt_vocab: int = 74,
        min_src: int = 40,
        max_src: int = 3000,
        median_src: float = 300.0,
        sigma: float = 0.6,
        with_audio: bool = False,
        mtl: bool = False,
        src_dict=None,
        tgt_dict=None,
    ):
        # mtl: batches in the format of the s2s_translation_mtl task's dataset (collate_mtl below)
        self.mtl, self.src_dict, self.tgt_dict = mtl, src_dict, tgt_dict
        rs = np.random.RandomState(seed)
        self.seed = seed
        self.nfps = n_frames_per_step
        self.feat_dim = feat_dim
        self.src_vocab = src_vocab
        self.tgt_vocab = tgt_vocab
        self.with_audio = with_audio
        src = np.clip(
            np.round(rs.lognormal(np.log(median_src), sigma, size=n_utts)), min_src, max_src
        ).astype(np.int64)
        tgt = np.maximum(
            4 * n_frames_per_step, np.round(0.6 * src * rs.uniform(0.85, 1.15, size=n_utts))
        ).astype(np.int64)
        self.src_n_frames = src
        self.tgt_n_frames = tgt
        self.tgt_steps = tgt // n_frames_per_step
        self.src_text_len = np.maximum(2, np.round(0.12 * src)).astype(np.int64) + 1
        self.tgt_text_len = np.maximum(2, np.round(0.15 * tgt)).astype(np.int64) + 1

    def __len__(self):
        return len(self.src_n_frames)

    def num_tokens(self, i: int) -> int:
        # batching cost = source frames (speech_to_text_dataset.py:343-344)
        return int(self.src_n_frames[i])

    def ordered_indices(self) -> np.ndarray:
        # longest first (speech_to_text_dataset.py:349-351 sorts by -n_frames)
        return np.argsort(-self.src_n_frames, kind="stable")

    def __getitem__(self, i: int) -> Utterance:
        rs = np.random.RandomState((self.seed * 1000003 + int(i) * 7919) % (2**31 - 1))
        s, d = int(self.src_n_frames[i]), int(self.tgt_steps[i])
        src = rs.standard_normal((s, self.feat_dim)).astype(np.float32)
        tgt = rs.standard_normal((d, self.feat_dim * self.nfps)).astype(np.float32)
        st = np.concatenate(
            [rs.randint(4, self.src_vocab, size=int(self.src_text_len[i]) - 1), [EOS]]
        ).astype(np.int64)
        tt = np.concatenate(
            [rs.randint(4, self.tgt_vocab, size=int(self.tgt_text_len[i]) - 1), [EOS]]
        ).astype(np.int64)
        audio = None
        if self.with_audio:
            audio = (0.1 * rs.standard_normal(160 * s)).astype(np.float32)
        return Utterance(int(i), src, tgt, st, tt, audio)

    def batches(self, max_tokens: int = 20000, max_sentences: int = 0, bsz_mult: int = 8):
        idx = self.ordered_indices()
        ntok = self.src_n_frames[idx]
        return batch_by_size(idx, ntok, max_tokens, max_sentences, bsz_mult)

    def collate_batch(self, indices: Sequence[int]) -> Dict:
        return self.collater([self[i] for i in indices])

    def collater(self, items) -> Dict:
        if self.mtl:
            return collate_mtl(list(items), self.src_dict, self.tgt_dict)
        return collate(list(items))


def batch_by_size(
    indices: np.ndarray,
    num_tokens_vec: np.ndarray,
    max_tokens: int,
    max_sentences: int = 0,
    bsz_mult: int = 1,
) -> List[np.ndarray]:
    """Greedy length-bucketed packing: cost(batch) = len(batch) * max(tokens) -- ``batch_by_size_vec``
    (fairseq/data/data_utils_fast.pyx:20-100, a Cython extension in the reference): native here too,
    ``s2st_batch_by_size`` of the C library (csrc/c_api.cpp); this wrapper only splits the index vector."""
    import ctypes as C
    from ..runtime import binding as bd
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    n = len(indices)
    if n == 0:
        return []
    ntok = np.ascontiguousarray(num_tokens_vec, dtype=np.int64)
    assert len(ntok) == n
    ends = np.zeros(n + 1, dtype=np.int32)
    fn = bd.lib().s2st_batch_by_size
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    count = fn(ntok.ctypes.data, n, int(max_tokens), int(max_sentences), int(bsz_mult), ends.ctypes.data)
    if count == -2 or (max_tokens > 0 and int(ntok.max()) > max_tokens):
        raise AssertionError(f"Sentences lengths should not exceed max_tokens={max_tokens}")
    if count < 0:
        raise bd.S2STHipError(f"s2st_batch_by_size failed with code {count}")
    return [b for b in np.split(indices, ends[:count]) if len(b) > 0]


def _collate_frames(frames: List[np.ndarray]) -> torch.Tensor:
    mx = max(f.shape[0] for f in frames)
    out = torch.zeros((len(frames), mx) + tuple(frames[0].shape[1:]), dtype=torch.float32)
    for i, f in enumerate(frames):
        out[i, : f.shape[0]] = torch.from_numpy(f)
    return out


def _collate_tokens(seqs: List[np.ndarray], move_eos_to_beginning: bool) -> torch.Tensor:
    mx = max(len(s) for s in seqs)
    out = torch.full((len(seqs), mx), PAD, dtype=torch.long)
    for i, s in enumerate(seqs):
        t = torch.from_numpy(np.asarray(s, dtype=np.int64))
        if move_eos_to_beginning:
            out[i, 0] = EOS
            out[i, 1 : len(s)] = t[:-1]
        else:
            out[i, : len(s)] = t
    return out


def collate(items: List[Utterance]) -> Dict:
    """Batch dict with the schema of S2STDataset.collater (s2st_dataset.py:427-455)."""
    lens = torch.tensor([u.src_speech.shape[0] for u in items], dtype=torch.long)
    src_lens, order = lens.sort(descending=True, stable=True)
    order_l = order.tolist()
    items = [items[i] for i in order_l]
    src = _collate_frames([u.src_speech for u in items])
    tgt = _collate_frames([u.tgt_speech for u in items])
    tgt_lens = torch.tensor([u.tgt_speech.shape[0] for u in items], dtype=torch.long)
    src_text = _collate_tokens([u.src_text for u in items], False)
    tgt_text = _collate_tokens([u.tgt_text for u in items], False)
    src_text_len = torch.tensor([len(u.src_text) for u in items], dtype=torch.long)
    tgt_text_len = torch.tensor([len(u.tgt_text) for u in items], dtype=torch.long)
    bsz, _, d = tgt.shape
    prev = torch.cat([tgt.new_zeros((bsz, 1, d)), tgt[:, :-1, :]], dim=1)
    audios, pad_mask = None, None
    if items[0].src_audio is not None:
        n = max(len(u.src_audio) for u in items)
        audios = torch.zeros((bsz, n), dtype=torch.float32)
        pad_mask = torch.zeros((bsz, n), dtype=torch.bool)
        for i, u in enumerate(items):
            audios[i, : len(u.src_audio)] = torch.from_numpy(u.src_audio)
            pad_mask[i, len(u.src_audio):] = True
    return {
        "id": torch.tensor([u.index for u in items], dtype=torch.long),
        "net_input": {
            "src_speech": src,
            "src_speech_lens": src_lens,
            "prev_output_tokens": prev,
            "prev_src_text_tokens": _collate_tokens([u.src_text for u in items], True),
            "prev_tgt_text_tokens": _collate_tokens([u.tgt_text for u in items], True),
            "collated_audios_orig": audios,
            "padding_mask": pad_mask,
            "speaker": None,
        },
        "speaker": None,
        "src_text": src_text,
        "src_text_len": src_text_len,
        "tgt_text": tgt_text,
        "tgt_text_len": tgt_text_len,
        "tgt_speech": tgt,
        "target_lengths": tgt_lens,
        "ntokens": int(tgt_lens.sum().item()),
        "src_txt_ntokens": int(src_text_len.sum().item()),
        "tgt_txt_ntokens": int(tgt_text_len.sum().item()),
        "nsentences": bsz,
    }


def collate_mtl(items: List[Utterance], src_dict=None, tgt_dict=None) -> Dict:
    """Batch dict with the schema of the mtl task's collater (examples/s2s_trans/data/s2st_dataset_mtl.py:242-347): the
    source text WITHOUT its EOS (:192-195), ``source_texts``, no ``prev_src_text_tokens`` / ``src_txt_ntokens`` / HuBERT
    keys, ``prev_tgt_text_tokens`` in sample (unsorted) order (:290-296)."""
    unsorted = list(items)
    b = collate(items)
    order = torch.tensor([u.src_speech.shape[0] for u in unsorted], dtype=torch.long).sort(descending=True, stable=True)[1].tolist()
    srt = [unsorted[i] for i in order]
    src_text = _collate_tokens([u.src_text[:-1] for u in srt], False)
    ni = b["net_input"]
    out = {
        "id": b["id"],
        "net_input": {"src_speech": ni["src_speech"], "src_speech_lens": ni["src_speech_lens"],
                      "prev_output_tokens": ni["prev_output_tokens"],
                      "prev_tgt_text_tokens": _collate_tokens([u.tgt_text for u in unsorted], True)},
        "speaker": None, "src_text": src_text,
        "src_text_len": torch.tensor([len(u.src_text) - 1 for u in srt], dtype=torch.long),
        "tgt_text": b["tgt_text"], "tgt_text_len": b["tgt_text_len"], "tgt_speech": b["tgt_speech"],
        "target_lengths": b["target_lengths"], "durations": None, "pitches": None, "energies": None,
        "ntokens": b["ntokens"], "tgt_txt_ntokens": b["tgt_txt_ntokens"], "nsentences": b["nsentences"],
    }
    if src_dict is not None:
        out["source_texts"] = [src_dict.string(u.src_text[:-1]) for u in srt]
    if tgt_dict is not None:
        out["target_texts"] = [tgt_dict.string(u.tgt_text) for u in srt]
    return out
