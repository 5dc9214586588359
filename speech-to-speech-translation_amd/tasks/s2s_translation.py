"""``--task s2s_translation`` host side.

Mirror of examples/s2s_trans/tasks/s2s_translation.py:47-336 for the training path: flags,
dictionaries, ``build_model`` / ``build_criterion`` / ``train_step`` / ``valid_step`` /
``max_positions``, ``load_dataset``.  ``data`` is either the literal "synthetic" (seeded Fisher-shaped corpus,
data/synthetic.py: what bench.py uses) or a directory in the recipe's layout -- ``config.yaml``, dictionaries,
``<split>.tsv`` manifests over .npy / zip features (data/s2st_dataset.py).
"""
from __future__ import annotations

import argparse
from typing import Dict

import numpy as np
import torch

import os
from pathlib import Path

from ..data.data_cfg import S2STDataConfig
from ..data.dictionary import Dictionary
from ..data.s2st_dataset import S2STDatasetCreator
from ..data.synthetic import SyntheticFisherCorpus
from ..registry import HAVE_FAIRSEQ, TaskBase, register_task, CRITERIA, MODELS


def _flag_is_true(v) -> bool:
    """``--input-text`` / ``--use-hubert`` are STRING flags in the reference (s2s_translation.py:66-68); its task converts
    its own namespace copy with ``convertobool`` (:39-45, :79-80).  Here the task, model and criterion share one
    namespace and the model compares the string with 'true' (s2st_transformer.py:688), so the string stays as parsed
    and every use site converts."""
    return v is True or str(v).lower() in ("true", "1", "yes")


@register_task("s2s_translation")
class S2ST_TranslationTask(TaskBase):  # fairseq's LegacyFairseqTask when fairseq is importable
    @staticmethod
    def add_args(parser):
        """Flags of s2s_translation.py:49-71 that the training path reads."""
        a = parser.add_argument
        a("data", nargs="?", default="synthetic")
        a("--config-yaml", type=str, default="config.yaml")
        a("--max-source-positions", default=3000, type=int)
        a("--max-target-positions", default=2400, type=int)
        a("--n-frames-per-step", type=int, default=1)
        a("--eos-prob-threshold", type=float, default=0.5)
        a("--eval-inference", action="store_true")
        a("--use-hubert", type=str, default="false")
        a("--input-text", type=str, default="false", help="text-to-speech mode (t2s_transformer): the encoder reads src_text")
        a("--speaker-to-id", type=str, default=None, help="use speaker feature: JSON map speaker name -> id "
                                                       "(s2s_translation.py:71)")
        a("--src-vocab-size", type=int, default=44)
        a("--tgt-vocab-size", type=int, default=74)

    def __init__(self, args, src_dict: Dictionary, tgt_dict: Dictionary, device=None, data_cfg=None):
        if HAVE_FAIRSEQ:
            super().__init__(args)
        self.args = args
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        self.device = device
        self.data_cfg = data_cfg
        # s2s_translation.py:82-83: the task parses the JSON; ``args.speaker_to_id`` itself stays the STRING, and the
        # reference sizes the embedding tables by ``len(args.speaker_to_id)`` -- the string's length (:156-160)
        self.speaker_to_id = None
        if getattr(args, "speaker_to_id", None) is not None:
            import json
            self.speaker_to_id = json.loads(args.speaker_to_id) if isinstance(args.speaker_to_id, str) else dict(args.speaker_to_id)
        self.datasets: Dict[str, object] = {}

    @classmethod
    def setup_task(cls, args, device=None, **kw):
        """s2s_translation.py:93-119: dictionaries from the data directory's config; synthetic tables otherwise."""
        data = getattr(args, "data", "synthetic")
        if data and data != "synthetic" and os.path.isdir(data):
            data_cfg = S2STDataConfig(Path(data) / getattr(args, "config_yaml", "config.yaml"))
            dicts = []
            for name in (data_cfg.src_vocab_filename, data_cfg.tgt_vocab_filename):
                path = Path(data) / name
                if not path.is_file():
                    raise FileNotFoundError(f"Dict not found: {path.as_posix()}")
                dicts.append(Dictionary.load(path.as_posix()))
            if getattr(args, "train_subset", None) is not None:
                if not all(s.startswith("train") for s in args.train_subset.split(",")):
                    raise ValueError('Train splits should be named like "train*".')
            return cls(args, dicts[0], dicts[1], device=device, data_cfg=data_cfg)
        return cls(args, Dictionary.with_size(getattr(args, "src_vocab_size", 44)),
                   Dictionary.with_size(getattr(args, "tgt_vocab_size", 74)), device=device)

    @property
    def source_dictionary(self):
        return self.src_dict

    @property
    def target_dictionary(self):
        return self.tgt_dict

    def max_positions(self):
        return self.args.max_source_positions, self.args.max_target_positions

    def load_dataset(self, split, n_utts=4096, seed=1234, epoch=1, **kw):
        if self.data_cfg is not None:  # s2s_translation.py:121-135
            use_hubert = _flag_is_true(getattr(self.args, "use_hubert", "false"))
            self.data_cfg.set_use_hubert(use_hubert)
            self.data_cfg.set_kd_encoder(bool(getattr(self.args, "kd_encoder", False)))
            self.datasets[split] = S2STDatasetCreator.from_tsv(
                self.args.data, self.data_cfg, split, self.src_dict, self.tgt_dict, None, None,
                is_train_split=split.startswith("train"), epoch=epoch, seed=getattr(self.args, "seed", 1),
                n_frames_per_step=self.args.n_frames_per_step, speaker_to_id=self.speaker_to_id)
            return self.datasets[split]
        self.datasets[split] = SyntheticFisherCorpus(
            n_utts=n_utts, seed=seed, n_frames_per_step=self.args.n_frames_per_step,
            src_vocab=len(self.src_dict), tgt_vocab=len(self.tgt_dict), **kw)
        return self.datasets[split]

    def dataset(self, split):
        return self.datasets[split]

    def get_batch_iterator(self, dataset, max_tokens=None, max_sentences=None, max_positions=None,
                           required_batch_size_multiple=8, seed=1, num_shards=1, shard_id=0, epoch=1, num_workers=0, **kw):
        """fairseq/tasks/fairseq_task.py:269-305: length-ordered indices (under numpy_seed(seed)), size filter,
        max-tokens batches, then the sharded, per-epoch shuffled iterator."""
        from ..data.iterators import EpochBatchIterator, numpy_seed
        from ..data.synthetic import batch_by_size
        with numpy_seed(seed):
            indices = dataset.ordered_indices()
        if max_positions is not None and hasattr(dataset, "filter_indices_by_size"):
            indices, _ = dataset.filter_indices_by_size(indices, max_positions)
        ntok = np.array([dataset.num_tokens(int(i)) for i in indices], dtype=np.int64)
        batches = batch_by_size(np.asarray(indices, dtype=np.int64), ntok, max_tokens or 0, max_sentences or -1,
                                required_batch_size_multiple)
        collate = getattr(dataset, "collater", None) or (lambda items: dataset.collate_items(items))
        return EpochBatchIterator(dataset, collate, batches, seed=seed, num_shards=num_shards, shard_id=shard_id,
                                  epoch=epoch, num_workers=num_workers)

    def get_speaker_embeddings_path(self):
        """s2s_translation.py:145-151: ``speaker_emb_filename`` of the data config, under the data directory."""
        if self.data_cfg is not None and self.data_cfg.config.get("speaker_emb_filename") is not None:
            return os.path.join(self.args.data, self.data_cfg.config.get("speaker_emb_filename"))
        return None

    def build_model(self, args):
        """s2s_translation.py:174-184: the model, and with --eval-inference the generator validation uses."""
        from .. import models  # noqa: F401  (registers the architecture)
        args.n_frames_per_step = self.args.n_frames_per_step
        args.speaker_emb_path = self.get_speaker_embeddings_path()  # :179
        from ..registry import ARCHS
        arch = getattr(args, "arch", None) or "s2st_transformer"  # --arch picks the registered model, as in fairseq
        model = MODELS[ARCHS[arch][0] if arch in ARCHS else "s2st_transformer"].build_model(args, self)
        self.generator = None
        if getattr(args, "eval_inference", False):
            self.generator = self.build_generator_tts([model], args)
        return model

    @property
    def sr(self):
        """Target sample rate (s2s_translation.py:78: data config ``features.sample_rate``)."""
        if self.data_cfg is not None:
            feats = self.data_cfg.config.get("features") or {}
            if feats.get("sample_rate"):
                return int(feats["sample_rate"])
        return int(getattr(self.args, "sample_rate", 24000))

    def build_default_vocoder(self):
        """s2s_translation.py:208-215 -> get_vocoder (fairseq/models/text_to_speech/vocoder.py:147-158): Griffin-Lim
        from the data config's feature settings."""
        from ..vocoder import GriffinLimVocoder
        if self.data_cfg is not None:
            return GriffinLimVocoder.from_data_cfg(self.args, self.data_cfg, device=self.device)
        # synthetic data (no config.yaml): 80-bin log-mel at 24 kHz, hop 300 / window 1200 / n_fft 2048 -- the feature
        # geometry SURVEY 8(d) assumes for the Fisher targets
        a = self.args
        return GriffinLimVocoder(
            sample_rate=self.sr, win_size=int(getattr(a, "win_size", 1200)), hop_size=int(getattr(a, "hop_size", 300)),
            n_fft=int(getattr(a, "n_fft", 2048)), n_mels=int(getattr(a, "output_frame_dim", 80)),
            f_min=float(getattr(a, "f_min", 20.0)), f_max=float(getattr(a, "f_max", 8000.0)),
            spec_bwd_max_iter=int(getattr(a, "spec_bwd_max_iter", 32)), device=self.device,
            phase_rng=getattr(a, "gl_phase_rng", "numpy") or "numpy", seed=int(getattr(a, "seed", 1) or 1))

    def build_generator_tts(self, models, cfg, vocoder=None, **unused):
        """s2s_translation.py:186-204."""
        from ..speech_generator import AutoRegressiveSpeechGenerator
        if vocoder is None:
            vocoder = self.build_default_vocoder()
        elif vocoder is False:  # (tests: features only)
            vocoder = None
        return AutoRegressiveSpeechGenerator(
            models[0], vocoder, self.data_cfg, max_iter=self.args.max_target_positions,
            eos_prob_threshold=getattr(self.args, "eos_prob_threshold", 0.5),
            input_text=_flag_is_true(getattr(self.args, "input_text", False)))

    def build_generator(self, models, args, seq_gen_cls=None, extra_gen_cls_kwargs=None):
        """Text generator over an aux decoder (s2s_translation.py:312-336 -> FairseqTask.build_generator ->
        SequenceGenerator; fairseq_cli/generate_for_s2st.py:107-111 swaps ``model.decoder`` for the aux ASR / ST
        decoder first): here ``args.aux_decoder`` ("asr" | "st") picks the head."""
        from ..sequence_generator import AuxSequenceGenerator
        which = getattr(args, "aux_decoder", "st")
        return AuxSequenceGenerator(
            models[0], self.src_dict if which == "asr" else self.tgt_dict, which=which,
            beam_size=getattr(args, "beam", 5), max_len_a=getattr(args, "max_len_a", 0),
            max_len_b=getattr(args, "max_len_b", 200), min_len=getattr(args, "min_len", 1),
            len_penalty=getattr(args, "lenpen", 1.0), unk_penalty=getattr(args, "unkpen", 0.0),
            **(extra_gen_cls_kwargs or {}))

    def reduce_metrics(self, logging_outputs, criterion):
        """fairseq/tasks/fairseq_task.py:586-617: the criterion aggregates (s2st_loss.py:350-407)."""
        return criterion.__class__.reduce_metrics(logging_outputs)

    def build_criterion(self, args):
        from .. import criterions  # noqa: F401
        name = getattr(args, "criterion", None) or "s2st_loss"
        return CRITERIA[name if name in CRITERIA else "s2st_loss"].build_criterion(args, self)

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """fairseq/tasks/fairseq_task.py:465-497."""
        model.train()
        model.set_num_updates(update_num)
        loss, sample_size, logging_output = criterion(model, sample)
        if ignore_grad:
            loss = loss * 0
        if optimizer is not None:
            optimizer.backward(loss)
        else:
            loss.backward()
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        """fairseq_task.py:499-503 + s2s_translation.py:217-238: the loss, and with --eval-inference the MCD statistics
        of an AR decode of the batch against its targets."""
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        if getattr(self.args, "eval_inference", False):
            if getattr(self, "generator", None) is None:
                self.generator = self.build_generator_tts([model], self.args)
            hypos, inference_losses = self.valid_step_with_inference(sample, model, self.generator)
            for k, v in inference_losses.items():
                assert k not in logging_output
                logging_output[k] = v
        return loss, sample_size, logging_output

    def valid_step_with_inference(self, sample, model, generator):
        """s2s_translation.py:240-264: AR-generate, vocode prediction and target, MFCC -> DTW -> mel-cepstral
        distortion (HIP kernels: metrics.py), summed statistics."""
        from ..metrics import batch_mel_cepstral_distortion
        hypos = generator.generate(model, sample, has_targ=True)
        losses = {"mcd_loss": 0.0, "targ_frames": 0.0, "pred_frames": 0.0, "nins": 0.0, "ndel": 0.0}
        rets = batch_mel_cepstral_distortion([h["targ_waveform"] for h in hypos], [h["waveform"] for h in hypos],
                                             self.sr, normalize_type=None)
        for d, extra in rets:
            pathmap = extra[-1]
            losses["mcd_loss"] += float(d)
            losses["targ_frames"] += pathmap.size(0)
            losses["pred_frames"] += pathmap.size(1)
            losses["nins"] += float((pathmap.sum(dim=1) - 1).sum())
            losses["ndel"] += float((pathmap.sum(dim=0) - 1).sum())
        return hypos, losses
