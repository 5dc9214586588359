"""Geometry and batches of the checkpoint golden (oracle/gen_golden_ckpt.py, tests/test_checkpoint.py).
The reference always builds its subsampler with 1024 channels (SURVEY Appendix B / oracle make_args), so the
committed checkpoint (parameters + two Adam moments, fp32) is kept near 1 MB by narrowing everything else:
8-dim features, 8-wide encoder / decoder, no ST decoder."""
import importlib

from test_engine import NANO, DATA

CKPT_CFG = dict(NANO, input_feat_per_channel=8, output_frame_dim=8, encoder_embed_dim=8, decoder_embed_dim=8,
                encoder_ffn_embed_dim=16, decoder_ffn_embed_dim=16, encoder_attention_heads=2,
                decoder_attention_heads=2, prenet_dim=8, postnet_conv_dim=8, asr_decoder_embed_dim=8,
                st_ce_weight=0.0)


def ckpt_batches():
    D = importlib.import_module(DATA)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=44, median_src=40, min_src=30, feat_dim=8)
    return [c.collate_batch(range(2)), c.collate_batch(range(2, 4))]
