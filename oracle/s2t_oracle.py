"""CPU restatement (torch fp32) of the ST / ASR pre-training stage: ``s2t_transformer_hubert`` + ``s2t_loss``.

TEST INFRASTRUCTURE -- imported only by tests/ (and the golden generator); never by the product path.
Follows fairseq/models/speech_to_text/s2t_transformer_me.py:82-330 (model: the speech encoder of s2st_transformer,
:333-420, + a fairseq TransformerDecoder over the target dictionary, :266-283, 473-492) and
examples/s2s_trans/criterions/s2t_loss.py:36-160 (label-smoothed NLL summed over non-pad tokens, accuracy counts).
Pinned by tests/golden/s2t_tiny.npz, generated from the reference's own classes by oracle/gen_golden_s2t.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import s2st_oracle as O


def make_args(**kw):
    """base_architecture of s2t_transformer_me.py:493-533 on the oracle's flag names."""
    d = dict(encoder_attention_heads=8, decoder_attention_heads=8, encoder_normalize_before=True,
             decoder_normalize_before=True, encoder_ffn_embed_dim=2048, decoder_ffn_embed_dim=2048)
    d.update(kw)
    if "encoder_layers" in d:
        d["encoder_transformer_layers"] = d.pop("encoder_layers")
    if "decoder_layers" in d:
        d["decoder_transformer_layers"] = d.pop("decoder_layers")
    a = O.make_args(**d)
    a.asr_ce_weight = a.st_ce_weight = a.ctc_weight = 0.0
    a.middle_layers = "0"
    return a


class S2TModel(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.a = a
        self.encoder = O.S2STEncoder(a)
        self.decoder = O.AuxTextDecoder(a, a.tgt_vocab_size, a.decoder_embed_dim, a.decoder_embed_dim,
                                        a.decoder_transformer_layers, tap=None, out_dim=a.decoder_embed_dim)

    def forward(self, src_tokens, src_lengths, prev_output_tokens):
        enc = self.encoder(src_tokens, src_lengths)
        return self.decoder(prev_output_tokens, enc), enc


def criterion_forward(model: S2TModel, sample, test_type="asr", eps=0.1):
    """s2t_loss.py:80-126: returns (loss, sample_size, log, outs)."""
    ni = sample["net_input"]
    key = "src" if test_type == "asr" else "tgt"
    logits, enc = model(ni["src_speech"], ni["src_speech_lens"], ni[f"prev_{key}_text_tokens"])
    lprobs = F.log_softmax(logits.float(), dim=-1)
    target = sample[f"{key}_text"]
    lp, tg = lprobs.view(-1, lprobs.size(-1)), target.view(-1)
    loss, nll = O.label_smoothed_nll_loss(lp, tg, eps)
    mask = tg.ne(O.PAD)
    n_correct = int((lp.argmax(1)[mask] == tg[mask]).sum())
    log = {"loss": float(loss), "nll_loss": float(nll), "ntokens": sample[f"{key}_txt_ntokens"],
           "nsentences": int(target.size(0)), "sample_size": sample[f"{key}_txt_ntokens"], "n_correct": n_correct,
           "total": int(mask.sum())}
    return loss, log["sample_size"], log, {"logits": logits, "encoder_out": enc["encoder_out"], "encoder_lens": enc["encoder_lens"]}
