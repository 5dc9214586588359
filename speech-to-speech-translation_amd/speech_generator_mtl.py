"""Generator of the ``s2s_translation_mtl`` task: greedy CTC decoding of the SOURCE transcript + AR mel generation.

Counterpart of ``fairseq/speech_generator_for_s2st_mtl.py:37-158`` (AutoRegressiveSpeechGenerator of the mtl variant):
``generate(model, sample, has_targ, decode_source_text=..., decode_target_mel=...)``.

* ``decode_source_text`` (:63-95): ``log_softmax(ctc_proj(out_middle_layers[0]))`` -> argmax per encoder frame over the
  first ``src_lengths[b]`` frames (the SUB-SAMPLED lengths the encoder returns, s2st_transformer_mtl.py:160-168) ->
  collapse repeats (``itertools.groupby``) -> drop blanks (id 0) -> ``src_dict.string`` -> WER against
  ``sample["source_texts"]``; every hypothesis gets ``src_texts`` / ``hyps_src_texts``.  The projection, log-softmax
  and argmax run in libs2st_hip.so; the collapse is the reference's host-side list work.
* ``decode_target_mel`` (:97-148): the AR loop of the base generator (shared code).

The reference builds its WER scorer on every call and drops it (:63, the ``print`` of the score is commented out); the
script ``generate_waveform_mtl.py`` keeps its own.  Here ``self.scorer`` accumulates over calls and can be read or
``reset()``.
"""
from __future__ import annotations

from itertools import groupby
from typing import Dict, List

import torch

from .runtime import binding as bd
from .scoring import build_scorer
from .speech_generator import AutoRegressiveSpeechGenerator as _BaseARGenerator, PendingHypos


class AutoRegressiveSpeechGenerator(_BaseARGenerator):
    def __init__(self, model, vocoder, data_cfg=None, max_iter: int = 6000, eos_prob_threshold: float = 0.5, seed: int = 1):
        super().__init__(model, vocoder, data_cfg, max_iter=max_iter, eos_prob_threshold=eos_prob_threshold,
                         input_text=False, seed=seed)
        self.scorer = build_scorer("wer", None)

    def greedy_ctc_paths(self, model, tap: torch.Tensor) -> torch.Tensor:
        """tap [B, E, C] (raw output of encoder layer --middle-layers[0]) -> best label per frame [B, E] (int64)."""
        net_output = (None, None, {"out_middle_layers": [tap.transpose(0, 1)]})
        lprobs = model.get_normalized_probs(net_output, log_probs=True)  # [B, E, V] via the precise GEMM + row kernel
        B, E, V = lprobs.shape
        best = torch.empty(B * E, 1, dtype=torch.long, device=lprobs.device)
        # argmax over the vocabulary = over dim 1 of [B * E, V, 1]
        bd.call("s2st_argmax_dim1_f32", lprobs.contiguous().view(B * E, V, 1), best, B * E, V, 1)
        return best.view(B, E)

    @torch.no_grad()
    def generate(self, model, sample, has_targ: bool = False, **kwargs) -> List[Dict]:
        model.eval()
        eng = model.engine
        ni = sample["net_input"]
        src, src_lens = ni["src_speech"], ni["src_speech_lens"]
        bsz = src.shape[0]
        enc = eng.decode_begin(src, src_lens, self.max_iter, speaker=sample.get("speaker"))
        finalized: List[Dict] = PendingHypos(dict() for _ in range(bsz))
        if kwargs.get("decode_source_text"):
            if not eng.cfg.has_ctc:
                raise ValueError("decode_source_text needs the model's source-text CTC head (--ctc-weight > 0)")
            src_texts = sample["source_texts"]
            self._last_tap = enc["tap0"]
            best = self.greedy_ctc_paths(model, enc["tap0"]).cpu()
            lens = enc["encoder_lens"].cpu().tolist()
            hyps = []
            for b in range(bsz):
                indices = best[b, : lens[b]].tolist()
                collapsed = [k for k, _ in groupby(indices)]          # 1. collapse repeated labels
                hyps.append([x for x in collapsed if x != 0])          # 2. remove blanks
            hyp_texts = [model.src_dict.string(h) for h in hyps]
            for hyp, ref in zip(hyp_texts, src_texts):
                self.scorer.add_string(ref, hyp)
            for b in range(bsz):
                finalized[b]["src_texts"] = src_texts[b]
                finalized[b]["hyps_src_texts"] = hyp_texts[b]
                finalized[b]["hyps_src_tokens"] = hyps[b]
        self.defer_vocoder = bool(kwargs.get("defer_vocoder", False))
        try:
            if kwargs.get("decode_target_mel"):
                self._decode_mel(model, sample, bsz, finalized)
            if has_targ:
                self._add_targets(model, sample, bsz, finalized)
        finally:
            self.defer_vocoder = False
        return finalized
