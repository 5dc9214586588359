"""``--task s2s_translation_mtl`` (examples/s2s_trans/tasks/s2s_translation_mtl.py:39-330): the speech-to-speech task
of the ``s2st_transformer_mtl`` model / ``s2st_loss_mtl`` criterion.

What differs from ``s2s_translation`` in the reference, all reproduced here:

* its OWN dataset (``data/s2st_dataset_mtl.py``): ``src_text`` without EOS (the source-text CTC targets), ``source_texts``
  in the batch, no ``prev_src_text_tokens`` / ``src_txt_ntokens``, no HuBERT waveform keys (:15, 103-114);
* its OWN generator (``speech_generator_mtl.py``): greedy CTC decoding of the source transcript + WER next to the AR
  mel decoder; ``build_generator`` IS the speech generator here (:165-183 -- the base task calls it
  ``build_generator_tts`` and keeps ``build_generator`` for the aux text decoders, which this variant does not have);
* flags (:41-59): ``--max-source-positions`` 6000, no ``--use-hubert`` / ``--kd-encoder`` / ``--input-text`` /
  ``--speaker-to-id`` (``self.speaker_to_id`` is never set by the reference's task: reading it in ``load_dataset``
  (:113) raises AttributeError unless something else put it there -- here it is simply ``None``);
* ``get_speaker_embeddings(args)`` takes the width from ``args.speaker_embed_dim`` (:133-150).
"""
from __future__ import annotations

from ..data.s2st_dataset_mtl import S2STMTLDatasetCreator
from ..data.synthetic import SyntheticFisherCorpus
from ..registry import CRITERIA, MODELS, register_task
from .s2s_translation import S2ST_TranslationTask


@register_task("s2s_translation_mtl")
class S2ST_TranslationMTLTask(S2ST_TranslationTask):
    @staticmethod
    def add_args(parser):
        S2ST_TranslationTask.add_args(parser)
        parser.set_defaults(max_source_positions=6000)

    def load_dataset(self, split, n_utts=4096, seed=1234, epoch=1, **kw):
        """s2s_translation_mtl.py:103-114."""
        if self.data_cfg is not None:
            self.datasets[split] = S2STMTLDatasetCreator.from_tsv(
                self.args.data, self.data_cfg, split, self.src_dict, self.tgt_dict, None, None,
                is_train_split=split.startswith("train"), epoch=epoch, seed=getattr(self.args, "seed", 1),
                n_frames_per_step=self.args.n_frames_per_step, speaker_to_id=self.speaker_to_id)
            return self.datasets[split]
        # synthetic corpus in the mtl dataset's batch format (EOS-less source text, source_texts, no shifted source text)
        self.datasets[split] = SyntheticFisherCorpus(
            n_utts=n_utts, seed=seed, n_frames_per_step=self.args.n_frames_per_step, src_vocab=len(self.src_dict),
            tgt_vocab=len(self.tgt_dict), mtl=True, src_dict=self.src_dict, tgt_dict=self.tgt_dict, **kw)
        return self.datasets[split]

    def build_model(self, args):
        from .. import models  # noqa: F401
        args.n_frames_per_step = self.args.n_frames_per_step
        model = MODELS["s2st_transformer_mtl"].build_model(args, self)
        self.generator = None
        if getattr(args, "eval_inference", False):
            self.generator = self.build_generator([model], args)
        return model

    def build_generator(self, models, cfg, vocoder=None, **unused):
        """s2s_translation_mtl.py:165-183: the mtl speech generator (CTC transcript + AR mel)."""
        from ..speech_generator_mtl import AutoRegressiveSpeechGenerator
        if vocoder is None:
            vocoder = self.build_default_vocoder()
        elif vocoder is False:  # (tests: features only)
            vocoder = None
        return AutoRegressiveSpeechGenerator(
            models[0], vocoder, self.data_cfg, max_iter=self.args.max_target_positions,
            eos_prob_threshold=getattr(self.args, "eos_prob_threshold", 0.5))

    def build_generator_tts(self, models, cfg, vocoder=None, **unused):
        return self.build_generator(models, cfg, vocoder=vocoder)

    def valid_step_with_inference(self, sample, model, generator):
        """s2s_translation_mtl.py:219-243: as the base task's, over the mel decoding of the mtl generator."""
        inner = generator.generate

        class _MelOnly:  # the reference calls generate(model, sample, has_targ=True) -- which decodes NOTHING with the mtl
            # generator (both switches default off) and then fails on hypo["waveform"]; the evident intent is the mel
            def generate(self_, m, s, has_targ=False):
                return inner(m, s, has_targ=has_targ, decode_target_mel=True)

        return super().valid_step_with_inference(sample, model, _MelOnly())

    def build_criterion(self, args):
        from .. import criterions  # noqa: F401
        return CRITERIA["s2st_loss_mtl"].build_criterion(args, self)
