"""Beam search over an aux ASR / ST text decoder on the HIP path.

Counterpart of what ``fairseq_cli/generate_for_s2st.py:107-111, 178-219`` runs: ``model.decoder`` swapped for
``model.aux_asr_decoder`` / ``model.aux_st_decoder`` and fairseq's ``SequenceGenerator`` (fairseq/sequence_generator.py:
189-571, ``finalize_hypos`` :607-716, ``search.BeamSearch.step`` fairseq/search.py:103-144) with the recipe's
``--beam 5`` (run_baseline.sh:185).  The decoder forward (embedding, positions, layers with causal self- and encoder
attention over the head's encoder tap, output projection) and the log-softmax run in libs2st_hip.so
(``s2st_engine_aux_decode``, ``s2st_log_softmax_rows_f32``); the search itself -- top-2k candidates, EOS bookkeeping,
finalisation, length normalisation -- is host-side integer / index work on small arrays, as in the reference, here in
numpy.  Round 6: the decoder runs INCREMENTALLY like the reference's (``incremental_state`` + ``reorder_incremental_state``,
fairseq/sequence_generator.py:330-400): ``s2st_engine_aux_inc_begin`` projects the encoder tap to every layer's static keys /
values once, ``s2st_engine_aux_inc_step`` embeds the hypotheses' last tokens, appends their key / value rows to the caches and
attends over them, and the surviving beams' indices reorder the caches on the device -- a hypothesis costs O(L) per step.
``incremental=False`` keeps rounds 2 - 5's form (the decoder re-run on the whole prefix each step): the reference result either way.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List

import numpy as np
import torch

from .runtime import binding as bd
from .runtime.engine import PAD


class AuxSequenceGenerator:
    def __init__(self, model, tgt_dict, which: str = "st", beam_size: int = 5, max_len_a: float = 0.0, max_len_b: int = 200,
                 max_len: int = 0, min_len: int = 1, normalize_scores: bool = True, len_penalty: float = 1.0,
                 unk_penalty: float = 0.0, temperature: float = 1.0, incremental: bool = True, **unused):
        if which not in ("asr", "st"):
            raise ValueError("which must be 'asr' or 'st'")
        if temperature != 1.0:
            raise NotImplementedError("temperature != 1")
        self.model, self.which, self.tgt_dict = model, which, tgt_dict
        self.incremental = bool(incremental)
        self.pad, self.unk, self.eos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.max_len = max_len or model.args.max_target_positions  # model.max_decoder_positions()
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        eng = model.engine
        if not (eng.cfg.has_asr if which == "asr" else eng.cfg.has_st):
            raise ValueError(f"the model has no aux {which} decoder")
        lib = eng.lib
        lib.s2st_engine_aux_decode_workspace.argtypes = [C.c_void_p] + [C.c_int32] * 4
        lib.s2st_engine_aux_decode_workspace.restype = C.c_int64
        lib.s2st_engine_aux_decode.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32] * 3 + \
            [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]

    # -- decoder forward on the device: tokens [Bb, L] (host int64) -> lprobs of the last position [Bb, V] (host fp32)
    def _step_lprobs(self, tap, enc_lens_dev, tokens: np.ndarray, E: int) -> np.ndarray:
        eng, dev = self.model.engine, self.model.engine.device
        Bb, L = tokens.shape
        w = 0 if self.which == "asr" else 1
        V = self.vocab_size
        need = int(eng.lib.s2st_engine_aux_decode_workspace(eng.h, w, Bb, L, E))
        if need < 0:
            raise bd.S2STHipError(f"s2st_engine_aux_decode_workspace failed ({need})")
        if eng.workspace is None or eng.workspace.numel() < need:
            eng.workspace = torch.empty(int(need * 1.25) + 4096, dtype=torch.float32, device=dev)
        tok = torch.from_numpy(np.ascontiguousarray(tokens)).to(dev)
        m = (tokens != PAD).astype(np.int64)
        pos = torch.from_numpy((np.cumsum(m, axis=1) * m + PAD).astype(np.int32)).to(dev)  # make_positions (utils.py:254-264)
        lens = torch.from_numpy(m.sum(1).astype(np.int32)).to(dev)
        d = eng.cfg.asr_dim if self.which == "asr" else eng.cfg.st_dim
        pe = eng.pe(d, L + 2)
        logits = torch.empty(Bb, L, V, dtype=torch.float32, device=dev)
        bd.check(eng.lib.s2st_engine_aux_decode(eng.h, w, tap.data_ptr(), enc_lens_dev.data_ptr(), tok.data_ptr(),
                                                pos.data_ptr(), lens.data_ptr(), pe.data_ptr(), Bb, L, E, logits.data_ptr(),
                                                eng.workspace.data_ptr(), eng.workspace.numel(), bd.stream_ptr()),
                 "s2st_engine_aux_decode")
        last = logits[:, L - 1, :].contiguous()
        lp = torch.empty(Bb, V, dtype=torch.float32, device=dev)
        bd.call("s2st_log_softmax_rows_f32", last, V, lp, V, Bb, V, 1)  # get_normalized_probs(log_probs=True)
        return lp.cpu().numpy()

    # -- the incremental form: caches live in a device buffer owned here for the length of one generate() ----------------
    def _inc_begin(self, tap, enc_lens_dev, E: int, max_len: int):
        eng, dev = self.model.engine, self.model.engine.device
        lib = eng.lib
        w = 0 if self.which == "asr" else 1
        Bb = tap.shape[0]
        lib.s2st_engine_aux_inc_state_floats.argtypes = [C.c_void_p] + [C.c_int32] * 4
        lib.s2st_engine_aux_inc_state_floats.restype = C.c_int64
        lib.s2st_engine_aux_inc_workspace.argtypes = [C.c_void_p] + [C.c_int32] * 3
        lib.s2st_engine_aux_inc_workspace.restype = C.c_int64
        lib.s2st_engine_aux_inc_begin.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 3 + \
            [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.s2st_engine_aux_inc_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p]
        n = int(lib.s2st_engine_aux_inc_state_floats(eng.h, w, Bb, E, max_len + 2))
        need = int(lib.s2st_engine_aux_inc_workspace(eng.h, w, Bb, E))
        if n < 0 or need < 0:
            raise bd.S2STHipError(f"s2st_engine_aux_inc_state_floats / _workspace failed ({n}, {need})")
        if eng.workspace is None or eng.workspace.numel() < need:
            eng.workspace = torch.empty(int(need * 1.25) + 4096, dtype=torch.float32, device=dev)
        state = torch.zeros(n, dtype=torch.float32, device=dev)
        bd.check(lib.s2st_engine_aux_inc_begin(eng.h, w, tap.data_ptr(), enc_lens_dev.data_ptr(), Bb, E, max_len + 2,
                                               state.data_ptr(), eng.workspace.data_ptr(), eng.workspace.numel(), bd.stream_ptr()),
                 "s2st_engine_aux_inc_begin")
        d = eng.cfg.asr_dim if self.which == "asr" else eng.cfg.st_dim
        return {"state": state, "Bb": Bb, "pe": eng.pe(d, max_len + 4), "w": w,
                "logits": torch.empty(Bb, self.vocab_size, dtype=torch.float32, device=dev),
                "lp": torch.empty(Bb, self.vocab_size, dtype=torch.float32, device=dev)}

    def _inc_step_lprobs(self, inc, step: int, last_tokens: np.ndarray, reorder) -> np.ndarray:
        eng, dev = self.model.engine, self.model.engine.device
        Bb, V = inc["Bb"], self.vocab_size
        tok = torch.from_numpy(np.ascontiguousarray(last_tokens, dtype=np.int64)).to(dev)
        pos = torch.full((Bb,), step + PAD + 1, dtype=torch.int32, device=dev)  # make_positions: no padding inside a prefix
        ro = None if reorder is None else torch.from_numpy(np.ascontiguousarray(reorder, dtype=np.int32)).to(dev)
        bd.check(eng.lib.s2st_engine_aux_inc_step(eng.h, inc["w"], step, tok.data_ptr(), bd.ptr(ro), pos.data_ptr(),
                                                  inc["pe"].data_ptr(), inc["logits"].data_ptr(), eng.workspace.data_ptr(),
                                                  eng.workspace.numel(), bd.stream_ptr()), "s2st_engine_aux_inc_step")
        bd.call("s2st_log_softmax_rows_f32", inc["logits"], V, inc["lp"], V, Bb, V, 1)  # get_normalized_probs(log_probs=True)
        return inc["lp"].cpu().numpy()

    @torch.no_grad()
    def generate(self, models, sample: Dict, **kwargs) -> List[List[Dict]]:
        model = models[0] if isinstance(models, (list, tuple)) else models
        assert model is self.model
        model.eval()
        ni = sample["net_input"]
        enc = model.forward_encoder(ni.get("src_speech"), ni.get("src_speech_lens"), ni.get("collated_audios_orig"),
                                    ni.get("padding_mask"))
        taps = enc["out_middle_layers"]
        tap = taps[0] if (self.which == "asr" or len(taps) == 1) else taps[1]  # [E, B, C] view of [B, E, C]
        tap = tap.transpose(0, 1).contiguous()
        bsz, E, _ = tap.shape
        beam = self.beam_size
        eng = model.engine
        enc_lens = enc["encoder_lens"] if "encoder_lens" in enc else None
        if enc_lens is None:
            pm = enc["encoder_padding_mask"]
            enc_lens = (E - pm[0].long().sum(1)) if pm else torch.full((bsz,), E, dtype=torch.long, device=tap.device)
        enc_lens = enc_lens.to(torch.int32)
        order = torch.arange(bsz, device=tap.device).repeat_interleave(beam)
        tap_b = tap.index_select(0, order).contiguous()
        lens_b = enc_lens.index_select(0, order).contiguous()
        src = ni.get("src_speech")
        src_len = int(src.shape[1]) if src is not None else int(ni["collated_audios_orig"].shape[1])
        max_len = min(int(self.max_len_a * src_len + self.max_len_b), self.max_len - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"

        ninf = np.float32(-math.inf)
        scores = np.zeros((bsz * beam, max_len + 1), dtype=np.float32)
        tokens = np.full((bsz * beam, max_len + 2), self.pad, dtype=np.int64)
        tokens[:, 0] = self.eos
        cands_to_ignore = np.zeros((bsz, beam), dtype=bool)
        finalized: List[List[Dict]] = [[] for _ in range(bsz)]
        finished = [False] * bsz
        cand_size = 2 * beam
        bbsz_offsets = (np.arange(bsz) * beam)[:, None]
        cand_offsets = np.arange(cand_size)
        V = self.vocab_size
        inc = self._inc_begin(tap_b, lens_b, E, max_len) if self.incremental else None
        reorder = None  # (the beams step s + 1 continues: fairseq's reorder_state, sequence_generator.py:393-400)
        for step in range(max_len + 1):
            if inc is not None:
                lprobs = self._inc_step_lprobs(inc, step, tokens[:, step], reorder)
            else:
                lprobs = self._step_lprobs(tap_b, lens_b, tokens[:, :step + 1], E)
            if step < self.min_len:
                lprobs[:, self.eos] = ninf
            lprobs[lprobs != lprobs] = ninf
            lprobs[:, self.pad] = ninf
            lprobs[:, self.unk] -= np.float32(self.unk_penalty)
            if step >= max_len:
                lprobs[:, :self.eos] = ninf
                lprobs[:, self.eos + 1:] = ninf
            # search.BeamSearch.step
            lp3 = lprobs.reshape(bsz, beam, V)
            if step == 0:
                lp3 = lp3[:, ::beam, :]
            else:
                lp3 = lp3 + scores.reshape(bsz, beam, -1)[:, :, step - 1][:, :, None]
            flat = lp3.reshape(bsz, -1)
            k = min(cand_size, flat.shape[1] - 1)
            idx = np.argsort(-flat, axis=1, kind="stable")[:, :k]
            cand_scores = np.take_along_axis(flat, idx, axis=1)
            cand_beams = idx // V
            cand_indices = idx % V
            if k < cand_size:  # (vocabularies smaller than 2 x beam: pad the candidate list with dead entries)
                padn = cand_size - k
                cand_scores = np.concatenate([cand_scores, np.full((bsz, padn), ninf, np.float32)], 1)
                cand_beams = np.concatenate([cand_beams, np.zeros((bsz, padn), np.int64)], 1)
                cand_indices = np.concatenate([cand_indices, np.full((bsz, padn), self.pad, np.int64)], 1)
            cand_bbsz_idx = cand_beams + bbsz_offsets
            eos_mask = (cand_indices == self.eos) & (cand_scores != ninf)
            eos_mask[:, :beam][cands_to_ignore] = False
            for s in range(bsz):  # finished sentences are not pruned from the batch here, only ignored
                if finished[s]:
                    eos_mask[s, :] = False
            sel = eos_mask[:, :beam]
            if sel.any():
                eos_bbsz_idx = cand_bbsz_idx[:, :beam][sel]
                eos_scores = cand_scores[:, :beam][sel].copy()
                self._finalize(step, eos_bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam, max_len)
            if all(finished):
                break
            assert step < max_len, f"{step} < {max_len}"
            eos_mask[:, :beam] = ~((~cands_to_ignore) & (~eos_mask[:, :beam]))
            active_mask = eos_mask.astype(np.int64) * cand_size + cand_offsets[None, :eos_mask.shape[1]]
            active_hypos = np.argsort(active_mask, axis=1, kind="stable")[:, :beam]
            new_ignore = np.take_along_axis(active_mask, active_hypos, axis=1)
            cands_to_ignore = new_ignore >= cand_size
            active_bbsz_idx = np.take_along_axis(cand_bbsz_idx, active_hypos, axis=1).reshape(-1)
            reorder = active_bbsz_idx
            tokens[:, :step + 1] = tokens[active_bbsz_idx, :step + 1]
            tokens.reshape(bsz, beam, -1)[:, :, step + 1] = np.take_along_axis(cand_indices, active_hypos, axis=1)
            if step > 0:
                scores[:, :step] = scores[active_bbsz_idx, :step]
            scores.reshape(bsz, beam, -1)[:, :, step] = np.take_along_axis(cand_scores, active_hypos, axis=1)
        for s in range(bsz):
            sc = np.array([h["score"] for h in finalized[s]], dtype=np.float32)
            finalized[s] = [finalized[s][i] for i in np.argsort(-sc, kind="stable")]
        return [[{"tokens": torch.from_numpy(h["tokens"]), "score": torch.tensor(h["score"]),
                  "positional_scores": torch.from_numpy(h["positional_scores"]), "attention": torch.empty(0),
                  "alignment": torch.empty(0)} for h in hs] for hs in finalized]

    def _finalize(self, step, bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam, max_len):
        """fairseq/sequence_generator.py:607-716 (no sentence pruning here, so batch index == sentence index)."""
        tokens_clone = tokens[bbsz_idx][:, 1:step + 2].copy()
        tokens_clone[:, step] = self.eos
        pos_scores = scores[bbsz_idx][:, :step + 1].copy()
        pos_scores[:, step] = eos_scores
        pos_scores[:, 1:] = pos_scores[:, 1:] - pos_scores[:, :-1]
        if self.normalize_scores:
            eos_scores = eos_scores / np.float32((step + 1) ** self.len_penalty)
        sents = bbsz_idx // beam
        for i, s in enumerate(sents.tolist()):
            if len(finalized[s]) < beam:
                finalized[s].append({"tokens": tokens_clone[i], "score": float(eos_scores[i]),
                                     "positional_scores": pos_scores[i]})
        for s in sorted(set(sents.tolist())):
            if not finished[s] and (len(finalized[s]) == beam or step == max_len):
                finished[s] = True
