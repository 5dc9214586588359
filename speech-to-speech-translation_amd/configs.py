"""Named flag sets of the s2s_translation / s2st_transformer / s2st_loss path.

``recipe_args(name)`` returns the argparse namespace the task / model / criterion constructors read: the named
flags below on top of ``base_architecture`` (examples/s2s_trans/models/s2st_transformer.py:792-830) and the
criterion / task defaults of the reference recipe (examples/s2s_trans/run_baseline.sh:22-47, 96-124:
``--n-frames-per-step 4 --bce-pos-weight 5.0 --label-smoothing 0.1 --encoder/decoder-normalize-before
--asr/st-ce-weight 0.3 --middle-layers 4,9``, 1-layer d=64 aux decoders, dropout .1 / .1 / .01).
bench.py, train.py and the README snippet take their configurations from here.
"""
from __future__ import annotations

import argparse

# BASELINE.json configs[1]: base 12 enc / 6 dec, d=512, n-frames-per-step 4, recipe flags; CTC head on (the recipe
# leaves --ctc-weight at 0.0; the benchmark keeps it on so that every head of the north star is in the step)
BASE_RECIPE = dict(
    encoder_transformer_layers=12, decoder_transformer_layers=6, encoder_embed_dim=512, decoder_embed_dim=512,
    encoder_ffn_embed_dim=2048, decoder_ffn_embed_dim=2048, encoder_attention_heads=4, decoder_attention_heads=4,
    encoder_normalize_before=True, decoder_normalize_before=True, prenet_dim=256, postnet_conv_dim=512,
    middle_layers="4,9", asr_decoder_layers=1, st_decoder_layers=1, asr_decoder_embed_dim=64,
    st_decoder_embed_dim=64, ctc_weight=0.3, asr_ce_weight=0.3, st_ce_weight=0.3,
    dropout=0.1, attention_dropout=0.1, activation_dropout=0.01, prenet_dropout=0.5, postnet_dropout=0.5)

# BASELINE.json configs[3]: + frozen hubert_base front end (--use-hubert true; 768-wide features at 50 fps).  CTC off
# as in run_baseline.sh: with --use-hubert and --ctc-weight > 0 the reference's own criterion fails (CTC input lengths
# are derived from the fbank lengths, s2st_loss.py:231-232), and so does this path, with the same message.
BASE_RECIPE_HUBERT = dict(BASE_RECIPE, use_hubert="true", hubert_hidden=768, ctc_weight=0.0)

# BASELINE.json configs[0]: tiny 2+2 layers, d=128 (CPU-runnable in the reference)
TINY = dict(
    encoder_transformer_layers=2, decoder_transformer_layers=2, encoder_embed_dim=128, decoder_embed_dim=128,
    encoder_ffn_embed_dim=256, decoder_ffn_embed_dim=256, encoder_attention_heads=4, decoder_attention_heads=4,
    encoder_normalize_before=True, decoder_normalize_before=True, prenet_dim=32, postnet_conv_dim=128,
    middle_layers="0,1", asr_decoder_layers=1, st_decoder_layers=1, asr_decoder_embed_dim=64,
    st_decoder_embed_dim=64, ctc_weight=0.3, asr_ce_weight=0.3, st_ce_weight=0.3)

RECIPES = {"base_recipe": BASE_RECIPE, "base_recipe_hubert": BASE_RECIPE_HUBERT, "tiny": TINY}

# task / criterion / optimizer flags of run_baseline.sh that base_architecture does not default
RECIPE_DEFAULTS = dict(
    n_frames_per_step=4, bce_pos_weight=5.0, label_smoothing=0.1, report_accuracy=True,
    src_vocab_size=44, tgt_vocab_size=74, hubert_hidden=768, use_hubert="false",
    lr=1.5e-3, warmup_updates=4000, clip_norm=1.0, seed=1, max_tokens=20000, update_freq=1,
    adam_betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=0.0)


def recipe_args(name: str, **overrides) -> argparse.Namespace:
    from .models.s2st_transformer import base_architecture
    if name not in RECIPES:
        raise KeyError(f"unknown configuration {name!r}; known: {sorted(RECIPES)}")
    a = argparse.Namespace(**{**RECIPE_DEFAULTS, **RECIPES[name], **overrides})
    return base_architecture(a)
