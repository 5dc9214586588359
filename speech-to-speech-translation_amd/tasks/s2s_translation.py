"""``--task s2s_translation`` host side.

Mirror of examples/s2s_trans/tasks/s2s_translation.py:47-336 for the training path: flags,
dictionaries, ``build_model`` / ``build_criterion`` / ``train_step`` / ``valid_step`` /
``max_positions``.  Data comes from the seeded synthetic Fisher-shaped corpus
(data/synthetic.py) -- the on-disk TSV/ZIP reader is a later row of the scope table.
"""
from __future__ import annotations

import argparse
from typing import Dict

import torch

from ..data.synthetic import BOS, EOS, PAD, UNK, SyntheticFisherCorpus
from ..registry import register_task, CRITERIA, MODELS


class Dictionary:
    """Minimal symbol table with fairseq's special-symbol layout
    (fairseq/data/dictionary.py:27-40): <s>=0, <pad>=1, </s>=2, <unk>=3."""

    def __init__(self, n_symbols: int):
        self.symbols = ["<s>", "<pad>", "</s>", "<unk>"] + [f"s{i}" for i in range(n_symbols - 4)]

    def __len__(self):
        return len(self.symbols)

    def bos(self):
        return BOS

    def pad(self):
        return PAD

    def eos(self):
        return EOS

    def unk(self):
        return UNK


@register_task("s2s_translation")
class S2ST_TranslationTask:
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
        a("--src-vocab-size", type=int, default=44)
        a("--tgt-vocab-size", type=int, default=74)

    def __init__(self, args, src_dict: Dictionary, tgt_dict: Dictionary, device=None):
        self.args = args
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        self.device = device
        self.datasets: Dict[str, SyntheticFisherCorpus] = {}

    @classmethod
    def setup_task(cls, args, device=None, **kw):
        return cls(args, Dictionary(getattr(args, "src_vocab_size", 44)),
                   Dictionary(getattr(args, "tgt_vocab_size", 74)), device=device)

    @property
    def source_dictionary(self):
        return self.src_dict

    @property
    def target_dictionary(self):
        return self.tgt_dict

    def max_positions(self):
        return self.args.max_source_positions, self.args.max_target_positions

    def load_dataset(self, split, n_utts=4096, seed=1234, **kw):
        self.datasets[split] = SyntheticFisherCorpus(
            n_utts=n_utts, seed=seed, n_frames_per_step=self.args.n_frames_per_step,
            src_vocab=len(self.src_dict), tgt_vocab=len(self.tgt_dict), **kw)
        return self.datasets[split]

    def dataset(self, split):
        return self.datasets[split]

    def build_model(self, args):
        from .. import models  # noqa: F401  (registers the architecture)
        args.n_frames_per_step = self.args.n_frames_per_step
        return MODELS["s2st_transformer"].build_model(args, self)

    def build_criterion(self, args):
        from .. import criterions  # noqa: F401
        return CRITERIA["s2st_loss"].build_criterion(args, self)

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
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output
