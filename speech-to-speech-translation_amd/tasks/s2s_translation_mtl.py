"""``--task s2s_translation_mtl`` (examples/s2s_trans/tasks/s2s_translation_mtl.py:39-330): the speech-to-speech task
for the ``s2st_transformer_mtl`` model / ``s2st_loss_mtl`` criterion -- same data directory layout, dictionaries,
batching and validation as ``s2s_translation``; no HuBERT / text-input switches, ``--max-source-positions`` 6000."""
from __future__ import annotations

from ..registry import CRITERIA, MODELS, register_task
from .s2s_translation import S2ST_TranslationTask


@register_task("s2s_translation_mtl")
class S2ST_TranslationMTLTask(S2ST_TranslationTask):
    @staticmethod
    def add_args(parser):
        S2ST_TranslationTask.add_args(parser)
        parser.set_defaults(max_source_positions=6000)

    def build_model(self, args):
        from .. import models  # noqa: F401
        args.n_frames_per_step = self.args.n_frames_per_step
        model = MODELS["s2st_transformer_mtl"].build_model(args, self)
        self.generator = None
        if getattr(args, "eval_inference", False):
            self.generator = self.build_generator_tts([model], args)
        return model

    def build_criterion(self, args):
        from .. import criterions  # noqa: F401
        return CRITERIA["s2st_loss_mtl"].build_criterion(args, self)
