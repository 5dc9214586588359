"""``--arch s2st_transformer`` on the MI355X engine.

Host-side mirror of examples/s2s_trans/models/s2st_transformer.py:580-830 (names, flags,
state_dict keys, forward signature and return structure).  The module owns no arithmetic:
its parameters are views into the engine's flat fp32 arena, and ``forward`` runs the HIP
schedule behind include/s2st_hip.h.  Training goes through the fused criterion
(criterions/s2st_loss.py), which hands loss + gradients back through one autograd node.
"""
from __future__ import annotations

import argparse
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ..registry import ModelBase, register_model, register_model_architecture
from ..runtime.engine import Engine


class _Holder(nn.Module):
    """Anonymous container so parameters appear under the reference's dotted names."""


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, is_buffer: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], tensor)


@register_model("s2st_transformer")
class S2STTransformerModel(ModelBase):  # fairseq's BaseFairseqModel when fairseq is importable, else nn.Module
    @staticmethod
    def add_args(parser):
        """Same flags as the reference (s2st_transformer.py:586-664)."""
        a = parser.add_argument
        a("--dropout", type=float)
        a("--output-frame-dim", type=int)
        a("--speaker-embed-dim", type=int)
        a("--speaker-embed-dim-dec", type=int)
        a("--middle-layers", default="6", type=str)
        a("--hubert-hidden", type=int, default=768)
        a("--conv_kernel-sizes", default="5,5", type=str)
        a("--conv-channels", default=1024, type=int)
        a("--input-feat-per-channel", default=80, type=int)
        a("--input_channels", default=1, type=int)
        a("--encoder-transformer-layers", type=int)
        a("--encoder-embed-dim", type=int)
        a("--encoder-ffn-embed-dim", type=int)
        a("--encoder-normalize-before", action="store_true")
        a("--encoder-attention-heads", type=int)
        a("--attention-dropout", type=float)
        a("--activation-dropout", "--relu-dropout", type=float)
        a("--activation-fn", type=str, default="relu")
        a("--no_scale_embedding", default=False)
        a("--prenet-dropout", type=float)
        a("--prenet-layers", type=int)
        a("--prenet-dim", type=int)
        a("--postnet-dropout", type=float)
        a("--postnet-layers", type=int)
        a("--postnet-conv-dim", type=int)
        a("--postnet-conv-kernel-size", type=int)
        a("--decoder-transformer-layers", type=int)
        a("--asr-decoder-layers", type=int)
        a("--st-decoder-layers", type=int)
        a("--decoder-embed-dim", type=int)
        a("--asr-decoder-embed-dim", type=int)
        a("--st-decoder-embed-dim", type=int)
        a("--decoder-ffn-embed-dim", type=int)
        a("--decoder-normalize-before", action="store_true")
        a("--decoder-attention-heads", type=int)
        a("--load-pretrained-encoder-from", type=str)
        a("--load-pretrained-hubert-from", type=str)
        a("--load-pretrained-decoder-from", type=str)

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        if getattr(args, "activation_fn", "relu") != "relu":
            raise NotImplementedError("the HIP path implements the reference default activation (relu)")
        # speaker conditioning (s2st_transformer.py:203-206, 441-444; tables tasks/s2s_translation.py:153-172): on when the
        # task was given --speaker-to-id; the engine config checks the widths
        if getattr(args, "speaker_to_id", None) is None and getattr(task, "speaker_to_id", None) is not None:
            import json
            args.speaker_to_id = json.dumps(task.speaker_to_id)
        if not hasattr(args, "src_vocab_size"):
            args.src_vocab_size = len(task.source_dictionary)
            args.tgt_vocab_size = len(task.target_dictionary)
        device = getattr(task, "device", None) or torch.device("cuda", torch.cuda.current_device())
        model = cls(args, device=device, precise=bool(getattr(args, "precise_gemm", False)))
        # (the reference's decoders keep the dictionaries: s2st_transformer_mtl.py:226-227 -- the mtl generator reads
        # model.decoder.src_dict for its CTC hypotheses)
        model.src_dict = getattr(task, "source_dictionary", None)
        model.tgt_dict = getattr(task, "target_dictionary", None)
        cls.load_pretrained_components(model, args)
        return model

    @staticmethod
    def load_pretrained_components(model, args):
        """``--load-pretrained-encoder-from`` / ``--load-pretrained-decoder-from`` (s2st_transformer.py:704-733, used by
        run_mix_tuning.sh:143-144): a path that does not exist is skipped with a warning, as the reference does."""
        import logging
        import os
        from ..checkpoint_utils import load_pretrained_component_from_model
        log = logging.getLogger(__name__)
        for comp in ("encoder", "decoder"):
            path = getattr(args, f"load_pretrained_{comp}_from", None)
            if path is None:
                continue
            if not os.path.exists(path):
                log.warning(f"skipped pretraining because {path} does not exist")
                continue
            loaded = load_pretrained_component_from_model(model, comp, path)
            log.info(f"loaded pretrained {comp} from: {path} ({len(loaded)} tensors)")

    def __init__(self, args, device: torch.device, precise: bool = False):
        super().__init__()
        self.args = args
        # frozen HuBERT front end (s2st_transformer.py:685-703): hubert_base geometry unless the
        # caller passes ``hubert_geometry`` (tests); weights from --load-pretrained-hubert-from
        self.hubert = None
        self._fe_stream = None  # front_end_ahead's stream
        if str(getattr(args, "use_hubert", "false")) == "true":
            from .hubert import HubertFrontend
            geo = dict(getattr(args, "hubert_geometry", None) or {})
            if not geo:
                geo = dict(embed=getattr(args, "hubert_hidden", 768))
            self.hubert = HubertFrontend(device, precise=precise, **geo)
            if self.hubert.embed != getattr(args, "hubert_hidden", 768):
                raise ValueError("--hubert-hidden must equal the HuBERT embedding width")
            path = getattr(args, "load_pretrained_hubert_from", None)
            if path:
                # a fairseq HuBERT checkpoint pickles cfg / task_state objects next to the tensors: full unpickle,
                # like the reference's checkpoint_utils (fairseq/checkpoint_utils.py:281-345)
                from ..checkpoint_utils import load_checkpoint_to_cpu
                ck = load_checkpoint_to_cpu(path)
                self.hubert.load_state_dict(ck["model"] if "model" in ck else ck, strict=True)
        self.engine = Engine(args, device, precise=precise)
        self._views = {}
        for name, pv, gv, is_buf in self.engine.named_views():
            if is_buf:
                _register(self, name, pv, True)
            else:
                p = nn.Parameter(pv)
                p.grad = gv  # gradients are written in place by the engine
                _register(self, name, p, False)
            self._views[name] = pv
        # bookkeeping buffers the reference's state_dict carries
        for holder in ["encoder", "decoder"] + [k for k in ("aux_asr_decoder", "aux_st_decoder") if k in self._modules]:
            _register(self, holder + ".embed_positions._float_tensor", torch.zeros(1, device=device), True)
        for k in ("aux_asr_decoder", "aux_st_decoder"):
            if k in self._modules:
                _register(self, k + ".version", torch.tensor([3.0], device=device), True)
        for i in range(args.postnet_layers):
            # (BatchNorm's update counter is bookkeeping the checkpoint carries, never read on the device: a host tensor,
            # so that bumping it is not a kernel launch on the step's stream)
            _register(self, f"decoder.postnet.convolutions.{i}.1.num_batches_tracked",
                      torch.zeros((), dtype=torch.long), True)
        if getattr(args, "text_encoder", False):  # t2s_transformer: BatchNorm counters of the encoder prenet
            for i in range(args.encoder_conv_layers):
                _register(self, f"encoder.prenet.{i}.1.num_batches_tracked",
                          torch.zeros((), dtype=torch.long), True)
        self._num_updates = 0
        self.reset_parameters()
        # speaker tables from the data directory's speaker_emb_filename: Embedding.from_pretrained(freeze=True)
        # (tasks/s2s_translation.py:161-171) -- loaded, excluded from training
        if self.engine.cfg.spk_frozen:
            # the engine keeps frozen tables in its BUFFER arena (outside the optimizer's sweep, weight decay included:
            # the reference's optimizer never sees a requires_grad=False parameter); they are module buffers here and
            # keep their state_dict key
            import numpy as np
            mat = torch.from_numpy(np.load(args.speaker_emb_path)).float()
            for n_, v_ in self._views.items():
                if n_.endswith("embed_speaker.weight"):
                    if tuple(mat.shape) != tuple(v_.shape):
                        raise ValueError(f"{args.speaker_emb_path}: {tuple(mat.shape)} does not fit {n_} {tuple(v_.shape)}")
                    with torch.no_grad():
                        v_.copy_(mat)

    # -- initialisation: the reference's schemes ("Init") -----------------
    @torch.no_grad()
    def reset_parameters(self):
        for name, p in self.named_parameters():
            if name.endswith("pos_emb_alpha"):
                p.fill_(1.0)
            elif p.dim() == 1:
                if name.endswith(".weight"):
                    p.fill_(1.0)  # LayerNorm / BatchNorm gain
                elif "norm" in name or name.endswith(".1.bias") or name.endswith("out_proj.bias"):
                    p.zero_()  # norm shifts; MHA out_proj bias (multihead_attention.py:107-108)
                else:
                    fan_in = self._fan_in(name)
                    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                    p.uniform_(-bound, bound)  # nn.Linear / nn.Conv1d default
            elif name.endswith("embed_speaker.weight"):  # torch.nn.Embedding default (tasks/s2s_translation.py:158-160)
                nn.init.normal_(p, mean=0, std=1.0)
            elif name == "encoder.embed_tokens.weight":  # t2s: plain nn.Embedding(padding_idx) (t2s_transformer.py:52-53)
                nn.init.normal_(p, mean=0, std=1.0)
                p[1].zero_()
            elif "embed_tokens" in name:
                nn.init.normal_(p, mean=0, std=p.shape[1] ** -0.5)
                p[1].zero_()  # padding_idx
            elif "output_projection" in name:
                nn.init.normal_(p, mean=0, std=p.shape[1] ** -0.5)
            elif any(k in name for k in (".k_proj.", ".v_proj.", ".q_proj.")):
                nn.init.xavier_uniform_(p, gain=1 / math.sqrt(2))  # multihead_attention.py:94-112
            elif ".out_proj." in name or "project_in_dim" in name or "project_out_dim" in name:
                nn.init.xavier_uniform_(p)
            elif p.dim() == 3 and name.startswith("decoder."):
                nn.init.xavier_uniform_(p, nn.init.calculate_gain("tanh"))  # decoder_init (:314-316)
            elif p.dim() == 3 and name.startswith("encoder.prenet."):
                nn.init.xavier_uniform_(p, nn.init.calculate_gain("relu"))  # encoder_init (t2s_transformer.py:32-34)
            else:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))  # nn.Linear / nn.Conv1d default
        for name, b in self.named_buffers():
            if name.endswith("running_var"):
                b.fill_(1.0)
            elif name.endswith("running_mean"):
                b.zero_()

    def _fan_in(self, bias_name: str) -> int:
        w = self._views.get(bias_name[: -len("bias")] + "weight")
        return int(w[0].numel()) if w is not None else 0

    # -- reference API --------------------------------------------------------------------------
    def set_num_updates(self, num_updates):
        self._num_updates = num_updates

    def max_positions(self):
        return (self.args.max_source_positions, self.args.max_target_positions)

    def get_targets(self, sample, test_type, net_output):
        return sample["src_text"] if test_type == "asr" else sample["tgt_text"]

    def _run(self, src_tokens, src_lengths, prev_output_tokens, target_lengths,
             prev_src_text_tokens=None, prev_tgt_text_tokens=None, want_attn=True, speaker=None):
        B, D, _ = prev_output_tokens.shape
        sample = {
            "speaker": speaker,
            "net_input": {"src_speech": src_tokens, "src_speech_lens": src_lengths,
                          "prev_output_tokens": prev_output_tokens,
                          "prev_src_text_tokens": prev_src_text_tokens,
                          "prev_tgt_text_tokens": prev_tgt_text_tokens},
            "target_lengths": target_lengths, "ntokens": int(target_lengths.sum()),
        }
        # text targets are only needed by the loss; the aux decoders need the shifted inputs
        if prev_src_text_tokens is not None:
            sample["src_text"] = prev_src_text_tokens
            sample["src_text_len"] = prev_src_text_tokens.ne(1).sum(1)
        if prev_tgt_text_tokens is not None:
            sample["tgt_text"] = prev_tgt_text_tokens
            sample["tgt_text_len"] = prev_tgt_text_tokens.ne(1).sum(1)
        return self.engine.forward(sample, training=self.training, want_attn=want_attn, with_loss=False)

    def _front_end(self, src_tokens, src_lengths, collated_audios, padding_mask):
        """s2st_transformer.py:245-252: with --use-hubert the encoder input is the frozen HuBERT's
        features and the lengths are its un-padded frame counts."""
        if self.hubert is None:
            return src_tokens, src_lengths
        self.hubert.eval()
        feats, pad = self.hubert.extract_features(collated_audios, padding_mask)
        return feats, (~pad).long().sum(-1)

    def prepare_sample(self, sample, training: bool = True):
        """Device-resident form of a collated sample (what ``DevicePrefetcher`` and bench.py hold): features, length
        / position vectors uploaded once (``Engine.prepare``).  With --use-hubert the prepared batch owns the staged
        waveform and a feature buffer [B, T', hubert_hidden] that ``front_end_sample`` refills on every step -- the
        frozen front end is part of the training step (s2st_transformer.py:245-252), its host-side staging is not."""
        from ..runtime.prefetch import PreparedBatch
        if isinstance(sample, tuple) or sample is None or len(sample) == 0:
            return sample
        if self.hubert is None:
            return PreparedBatch(self.engine.prepare(sample, training=training), sample)
        ni = sample["net_input"]
        if ni.get("collated_audios_orig") is None:
            raise ValueError("--use-hubert needs net_input['collated_audios_orig'] / ['padding_mask'] "
                             "(data config use_hubert: s2st_dataset.py:339-358)")
        wave, lens_dev, _, T = self.hubert.stage(ni["collated_audios_orig"], ni["padding_mask"])
        feats = torch.empty(wave.shape[0], T, self.hubert.embed, dtype=torch.float32, device=self.engine.device)
        proto = dict(sample)
        # the fbank lengths stay available as ctc_src_speech_lens: the criterion derives the CTC input lengths
        # from them even in this mode (s2st_loss.py:231-232)
        proto["net_input"] = dict(ni, src_speech=feats, src_speech_lens=self.hubert.last_frame_lens.clone(),
                                  ctc_src_speech_lens=ni["src_speech_lens"])
        pb = PreparedBatch(self.engine.prepare(proto, training=training), sample)
        pb.hubert_io = (wave, lens_dev, feats)
        return pb

    def front_end_sample(self, sample):
        """Training / validation entry of the --use-hubert branch (s2st_transformer.py:245-252 as reached from
        s2st_loss.py:207-218): a collated sample carries the raw audio and NO fbank tensor; the frozen HuBERT's
        features take the place of ``src_speech`` / ``src_speech_lens``.  Non-HuBERT models pass through."""
        if self.hubert is None or sample is None or len(sample) == 0:
            return sample
        if not isinstance(sample, tuple):
            sample = self.prepare_sample(sample, training=self.training)
        io = getattr(sample, "hubert_io", None)
        if io is None:
            raise ValueError("a batch prepared without the HuBERT front end was given to a --use-hubert model")
        ev = getattr(sample, "fe_ready", None)
        if ev is not None:  # computed ahead (front_end_ahead): the step only waits for it
            torch.cuda.current_stream().wait_event(ev)
            sample.fe_ready = None
            return sample
        if self._fe_stream is not None:  # (one front-end call at a time: they share the front end's workspace)
            torch.cuda.current_stream().wait_stream(self._fe_stream)
        self.hubert.eval()
        self.hubert.forward_into(*io)
        return sample

    def front_end_ahead(self, sample, after=None):
        """The frozen front end of an UPCOMING prepared batch, launched now on a second stream: HuBERT does not depend on the
        update in flight, so its forward for batch i + 1 runs beside the training step of batch i (whose small dependent
        kernels leave most of the chip idle) instead of in front of step i + 1.  Same kernels, same results; the step then
        only waits for the event (``front_end_sample``).  Call it right before ``train_step`` of the CURRENT batch: the
        second stream first waits for everything enqueued so far (earlier readers of this batch's feature buffer) and for
        ``after`` (the upload event of a prefetched batch).  No-op without --use-hubert, on CPU, or with
        S2ST_HUBERT_AHEAD=0 (A/B switch)."""
        import os
        io = getattr(sample, "hubert_io", None) if sample is not None else None
        if self.hubert is None or io is None or not io[0].is_cuda or getattr(sample, "fe_ready", None) is not None \
                or os.environ.get("S2ST_HUBERT_AHEAD", "1") == "0":
            return sample
        if self._fe_stream is None:
            from ..runtime import streams
            self._fe_stream = streams.get("front-end", self.engine.device)
        fe = self._fe_stream
        fe.wait_stream(torch.cuda.current_stream())
        if after is not None:
            fe.wait_event(after)
        self.hubert.eval()
        with torch.cuda.stream(fe):
            self.hubert.forward_into(*io)
            ev = torch.cuda.Event()
            ev.record(fe)
        for t in io:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(fe)
        sample.fe_ready = ev
        return sample

    def forward(self, src_tokens, src_lengths, collated_audios, padding_mask, prev_output_tokens,
                **kwargs):
        """Returns ``[(post_feat_out, eos_out, extra), (asr_logits, None) | None,
        (st_logits, None) | None]`` as s2st_transformer.py:752-786."""
        src_tokens, src_lengths = self._front_end(src_tokens, src_lengths, collated_audios, padding_mask)
        o = self._run(src_tokens, src_lengths, prev_output_tokens, kwargs["target_lengths"],
                      kwargs.get("prev_src_text_tokens"), kwargs.get("prev_tgt_text_tokens"), speaker=kwargs.get("speaker"))
        taps = [o[k].transpose(0, 1) for k in ("tap0", "tap1") if k in o]
        extra = {"attn": o.get("attn"), "feature_out": o["feature_out"], "out_middle_layers": taps}
        asr = (o["asr_logits"], None) if "asr_logits" in o else None
        st = (o["st_logits"], None) if "st_logits" in o else None
        return [(o["post_feat_out"], o["eos_out"], extra), asr, st]

    def forward_encoder(self, src_tokens, src_lengths, collated_audios=None, padding_mask=None,
                        speaker=None, **kwargs):
        src_tokens, src_lengths = self._front_end(src_tokens, src_lengths, collated_audios, padding_mask)
        B = src_tokens.shape[0]
        dummy = torch.zeros(B, 1, self.engine.cfg.out_dim)
        o = self._run(src_tokens, src_lengths, dummy, torch.ones(B, dtype=torch.long), want_attn=False, speaker=speaker)
        lens = o["encoder_lens"].long()
        E = o["encoder_out"].shape[1]
        pad = torch.arange(E, device=lens.device).unsqueeze(0) >= lens.unsqueeze(1)
        return {"encoder_out": [o["encoder_out"].transpose(0, 1)],
                "encoder_padding_mask": [pad] if bool(pad.any()) else [],
                "encoder_embedding": [], "encoder_states": [],
                "out_middle_layers": [o[k].transpose(0, 1) for k in ("tap0", "tap1") if k in o],
                "src_tokens": [], "src_lengths": []}

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        """CTC head over tap 0 (the reference keeps it on the decoder, s2st_transformer.py:458-463):
        ``(log_)softmax(ctc_proj(out_middle_layers[0]))`` -> [B, E, V].  The criterion does not come through here (its
        fused log-softmax + CTC kernel reads the logits directly and also returns ``ctc_lprobs``); this entry serves
        callers that hold a ``net_output`` -- e.g. greedy CTC decoding of the source transcript."""
        if not self.engine.cfg.has_ctc:
            raise ValueError("the model was built without a CTC head (--ctc-weight 0)")
        tap = net_output[2]["out_middle_layers"][0].transpose(0, 1).contiguous()  # [B, E, C]
        B, E, Cd = tap.shape
        w = self._views["decoder.ctc_proj.weight"]
        b = self._views["decoder.ctc_proj.bias"]
        V = w.shape[0]
        from ..runtime import binding as bd
        ld = (V + 3) // 4 * 4
        logits = torch.empty(B * E, ld, dtype=torch.float32, device=tap.device)
        bd.gemm(tap.view(B * E, Cd), w, logits, B * E, V, Cd, c_ld=ld, bias=b, precise=True)
        out = torch.empty(B * E, V, dtype=torch.float32, device=tap.device)
        bd.call("s2st_log_softmax_rows_f32", logits, ld, out, V, B * E, V, 1 if log_probs else 0)
        return out.view(B, E, V)


@register_model_architecture("s2st_transformer", "s2st_transformer")
def base_architecture(args):
    """Defaults of s2st_transformer.py:792-830 (including the `conv_chaFnnels` typo that pins
    conv_channels to 1024)."""
    def g(k, v):
        if getattr(args, k, None) is None:
            setattr(args, k, v)

    g("dropout", 0.1)
    g("output_frame_dim", 80)
    g("middle_layers", "6")
    g("conv_kernel_sizes", "5,5")
    args.conv_channels = 1024
    g("encoder_transformer_layers", 12)
    g("encoder_embed_dim", 512)
    g("encoder_ffn_embed_dim", 4 * args.encoder_embed_dim)
    if not hasattr(args, "encoder_normalize_before"):
        args.encoder_normalize_before = True
    g("encoder_attention_heads", 4)
    g("attention_dropout", args.dropout)
    g("activation_dropout", args.dropout)
    g("activation_fn", "relu")
    g("prenet_dropout", 0.5)
    g("prenet_layers", 2)
    g("prenet_dim", 256)
    g("postnet_dropout", 0.5)
    g("postnet_layers", 5)
    g("postnet_conv_dim", 512)
    g("postnet_conv_kernel_size", 5)
    g("asr_decoder_layers", 6)
    g("st_decoder_layers", 6)
    g("asr_decoder_embed_dim", 256)
    g("st_decoder_embed_dim", 256)
    g("decoder_transformer_layers", 6)
    g("decoder_embed_dim", 512)
    g("decoder_ffn_embed_dim", 4 * args.decoder_embed_dim)
    if not hasattr(args, "decoder_normalize_before"):
        args.decoder_normalize_before = False
    g("decoder_attention_heads", 4)
    # task / criterion flags the constructors read
    for k, v in dict(n_frames_per_step=4, input_feat_per_channel=80, input_channels=1,
                     max_source_positions=3000, max_target_positions=2400, no_scale_embedding=False,
                     ctc_weight=0.0, asr_ce_weight=0.0, st_ce_weight=0.0, bce_pos_weight=1.0,
                     label_smoothing=0.0, l1_loss_weight=1.0, mse_loss_weight=1.0, eos_loss_weight=1.0,
                     use_hubert="false").items():
        g(k, v)
    return args
