"""Host side of the HIP training engine (C ABI: include/s2st_hip.h, `s2st_engine_*`).

PyTorch is used for device memory (flat parameter / gradient / buffer arenas, the
activation workspace, caller-owned outputs) and streams; all arithmetic runs in the HIP
kernels.  The sample dict consumed here has the schema of ``S2STDataset.collater``
(reference: examples/s2s_trans/data/s2st_dataset.py:427-455).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import binding as bd

PAD = 1


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "enc_layers", "dec_layers", "enc_dim", "dec_dim", "enc_ffn", "dec_ffn", "enc_heads",
        "dec_heads", "enc_pre_ln", "dec_pre_ln", "in_dim", "conv_channels", "conv_k", "out_dim",
        "prenet_layers", "prenet_dim", "postnet_layers", "postnet_dim", "postnet_k", "tap_asr",
        "tap_st", "has_asr", "has_st", "has_ctc", "asr_layers", "asr_dim", "st_layers", "st_dim",
        "src_vocab", "tgt_vocab", "no_scale_embedding", "precise", "tap_dec", "has_ctc_tgt", "text_input", "enc_conv_layers", "enc_conv_k",
        "n_speakers", "spk_frozen", "spk_dim")] + [(n, C.c_float) for n in (
        "dropout", "attn_dropout", "act_dropout", "prenet_dropout", "postnet_dropout", "ctc_weight",
        "asr_weight", "st_weight", "w_l1", "w_mse", "w_eos", "bce_pos_weight", "label_smoothing", "ctc_tgt_weight", "enc_dropout")] + \
        [("s2t_mode", C.c_int32)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 120), ("offset", C.c_int64), ("numel", C.c_int64),
                ("ndim", C.c_int32), ("shape", C.c_int32 * 4), ("is_buffer", C.c_int32)]


_P = C.c_void_p


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("D", C.c_int32), ("Ls", C.c_int32),
                ("Lt", C.c_int32), ("E", C.c_int32),
                ("src", _P), ("enc_lens", _P), ("enc_pos", _P), ("ctc_in_lens", _P),
                ("prev", _P), ("tgt", _P), ("tgt_lens", _P), ("dec_pos", _P),
                ("prev_src_txt", _P), ("src_txt", _P), ("src_txt_lens", _P), ("src_txt_pos", _P),
                ("prev_tgt_txt", _P), ("tgt_txt", _P), ("tgt_txt_lens", _P), ("tgt_txt_pos", _P),
                ("pe_enc", _P), ("pe_dec", _P), ("pe_asr", _P), ("pe_st", _P),
                ("ntokens", C.c_int32), ("src_txt_ntokens", C.c_int32),
                ("tgt_txt_ntokens", C.c_int32), ("training", C.c_int32), ("want_attn", C.c_int32),
                ("seed", C.c_uint64), ("speaker", _P)]


class Outputs(C.Structure):
    _fields_ = [(n, _P) for n in ("post_feat", "feat", "eos", "attn", "enc_out", "tap0", "tap1",
                                  "asr_logits", "st_logits", "ctc_lprobs", "stats")]


class DropoutSite(C.Structure):  # s2st_dropout_site (include/s2st_hip.h)
    _fields_ = [("seed", C.c_uint64), ("kind", C.c_int32), ("ordinal", C.c_int32), ("p", C.c_float), ("reserved", C.c_int32),
                ("dims", C.c_int64 * 5), ("ctx", C.c_char * 24)]


SITE_KIND = {1: "lin", 2: "attn", 3: "rows", 4: "norm"}


class DecodeReplay(C.Structure):  # s2st_decode_replay (include/s2st_hip.h)
    _fields_ = [(n, C.c_void_p) for n in ("step", "seeds", "cur_feat", "cur_eos", "cur_attn", "pe_cur")]



STAT = dict(L1_SUM=0, MSE_SUM=1, BCE_SUM=2, ASR_NLL=3, ASR_SMOOTH=4, ASR_CORRECT=5, ASR_TOTAL=6,
            ST_NLL=7, ST_SMOOTH=8, ST_CORRECT=9, ST_TOTAL=10, LOSS=16, L1=17, MSE=18, EOS=19,
            CTC=20, ASR=21, ST=22, CTC_TGT=23, GNORM=24)


def config_from_args(a, precise: bool = False) -> ModelConfig:
    """Map the reference's flags (s2st_transformer.py:586-664, 792-830; s2st_loss.py:52-103)
    onto the engine config."""
    ks = [int(k) for k in str(a.conv_kernel_sizes).split(",")]
    if len(ks) != 2 or ks[0] != ks[1]:
        raise ValueError("engine supports two equal conv kernel sizes (reference default '5,5')")
    mids = [int(k) for k in str(a.middle_layers).split(",")]
    has_asr, has_st, has_ctc = a.asr_ce_weight > 0, a.st_ce_weight > 0, a.ctc_weight > 0
    c = ModelConfig()
    c.enc_layers, c.dec_layers = a.encoder_transformer_layers, a.decoder_transformer_layers
    c.enc_dim, c.dec_dim = a.encoder_embed_dim, a.decoder_embed_dim
    c.enc_ffn, c.dec_ffn = a.encoder_ffn_embed_dim, a.decoder_ffn_embed_dim
    c.enc_heads, c.dec_heads = a.encoder_attention_heads, a.decoder_attention_heads
    c.enc_pre_ln, c.dec_pre_ln = int(a.encoder_normalize_before), int(a.decoder_normalize_before)
    # --use-hubert: the subsampler's first conv reads HuBERT features (s2st_transformer.py:162-163)
    c.in_dim = (a.hubert_hidden if str(getattr(a, "use_hubert", "false")) == "true"
                else a.input_feat_per_channel * a.input_channels)
    c.conv_channels, c.conv_k = 1024, ks[0]  # --conv-channels is ignored by the reference (:802)
    c.out_dim = a.output_frame_dim * a.n_frames_per_step
    c.prenet_layers, c.prenet_dim = a.prenet_layers, a.prenet_dim
    c.postnet_layers, c.postnet_dim, c.postnet_k = a.postnet_layers, a.postnet_conv_dim, a.postnet_conv_kernel_size
    c.tap_asr = mids[0] if (has_asr or (has_ctc and not getattr(a, "text_encoder", False))) else -1
    c.tap_st = mids[1] if has_st and len(mids) > 1 else -1
    c.has_asr, c.has_st, c.has_ctc = int(has_asr), int(has_st), int(has_ctc)
    c.asr_layers, c.asr_dim = a.asr_decoder_layers, a.asr_decoder_embed_dim
    c.st_layers, c.st_dim = a.st_decoder_layers, a.st_decoder_embed_dim
    c.src_vocab, c.tgt_vocab = a.src_vocab_size, a.tgt_vocab_size
    c.no_scale_embedding = int(bool(a.no_scale_embedding))
    c.precise = int(precise)
    c.dropout, c.attn_dropout, c.act_dropout = a.dropout, a.attention_dropout, a.activation_dropout
    c.prenet_dropout, c.postnet_dropout = a.prenet_dropout, a.postnet_dropout
    c.ctc_weight, c.asr_weight, c.st_weight = a.ctc_weight, a.asr_ce_weight, a.st_ce_weight
    c.w_l1, c.w_mse, c.w_eos = a.l1_loss_weight, a.mse_loss_weight, a.eos_loss_weight
    c.bce_pos_weight, c.label_smoothing = a.bce_pos_weight, a.label_smoothing
    # s2st_transformer_mtl: target-text CTC head on the output of decoder layer --middle-layers-decoder
    c.ctc_tgt_weight = float(getattr(a, "ctc_weight_tgt", 0.0) or 0.0)
    c.has_ctc_tgt = int(c.ctc_tgt_weight > 0)
    c.tap_dec = int(str(getattr(a, "middle_layers_decoder", "6")).split(",")[0]) if c.has_ctc_tgt else -1
    # t2s_transformer: text encoder front (t2s_transformer.py:37-126)
    c.text_input = int(bool(getattr(a, "text_encoder", False)))
    c.enc_conv_layers = int(getattr(a, "encoder_conv_layers", 3) or 0) if c.text_input else 0
    c.enc_conv_k = int(getattr(a, "encoder_conv_kernel_size", 5) or 5)
    c.enc_dropout = float(getattr(a, "encoder_dropout", 0.5) or 0.0) if c.text_input else 0.0
    # speaker conditioning: table rows = len(args.speaker_to_id) -- of the STRING the flag carries, as in the reference
    # (tasks/s2s_translation.py:156-160) -- and widths that the additions themselves require
    spk = getattr(a, "speaker_to_id", None)
    c.n_speakers = len(spk) if spk is not None else 0
    c.spk_frozen = int(c.n_speakers > 0 and getattr(a, "speaker_emb_path", None) is not None)
    if c.n_speakers > 0 and c.text_input:
        # t2s_transformer.py:43-46: Embedding(len(speaker_to_id), speaker_embed_dim) + spk_emb_proj(C + dim -> C)
        c.spk_dim = int(getattr(a, "speaker_embed_dim", 64) or 64)
        if c.spk_dim % 4:
            raise ValueError("--speaker-embed-dim must be a multiple of 4 on this path")
    elif c.n_speakers > 0:
        if getattr(a, "speaker_embed_dim", 64) != c.enc_dim:
            raise ValueError(f"--speaker-embed-dim {getattr(a, 'speaker_embed_dim', 64)} must equal --encoder-embed-dim "
                             f"{c.enc_dim}: the row is added to the encoder states (s2st_transformer.py:203-206)")
        if getattr(a, "speaker_embed_dim_dec", 64) != c.out_dim:
            raise ValueError(f"--speaker-embed-dim-dec {getattr(a, 'speaker_embed_dim_dec', 64)} must equal output_frame_dim * "
                             f"n_frames_per_step = {c.out_dim}: the row replaces the first input frame (:441-444)")
    # s2t_transformer_hubert: speech encoder + ONE full-width text decoder, no mel decoder (models/s2t_transformer.py)
    c.s2t_mode = int(bool(getattr(a, "s2t_mode", False)))
    if c.s2t_mode and (c.has_asr or c.has_st or c.has_ctc or c.has_ctc_tgt or c.text_input or c.n_speakers):
        raise ValueError("s2t_transformer_hubert has no aux heads, CTC heads, text encoder or speaker tables")
    if c.text_input and (c.has_asr or c.has_st or c.has_ctc_tgt or c.enc_conv_k % 2 != 1):
        raise ValueError("t2s_transformer: no aux heads / target-text CTC head; --encoder-conv-kernel-size must be odd")
    if c.has_ctc_tgt and not (0 <= c.tap_dec < c.dec_layers):
        raise ValueError("--middle-layers-decoder must name a decoder layer (the reference would index an empty list)")
    return c


def sinusoidal_table(num: int, dim: int, padding_idx: int = PAD) -> torch.Tensor:
    """[sin | cos] table, exponent log(1e4)/(half-1), zero padding row
    (fairseq/modules/sinusoidal_positional_embedding.py:35-58)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    t = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num, -1)
    if dim % 2 == 1:
        t = torch.cat([t, torch.zeros(num, 1)], dim=1)
    t[padding_idx, :] = 0
    return t


def conv_out_len(n: int, k: int, stride: int = 2) -> int:
    return (n + 2 * (k // 2) - k) // stride + 1


class Engine:
    """Owns the arenas and the native engine handle."""

    def __init__(self, args, device: torch.device, precise: bool = False, share_with: Optional["Engine"] = None):
        """``share_with``: an INFERENCE twin of another engine -- the same fp32 parameter / gradient / buffer arenas (read-only
        there), its own bf16 copies, workspace, caches and streams: a second, independent decode chain over the same
        weights (``Engine.inference_twin``)."""
        self.args = args
        self.device = device
        self.lib = bd.lib()
        self.cfg = config_from_args(args, precise)
        self.lib.s2st_engine_create.argtypes = [C.POINTER(ModelConfig), C.POINTER(C.c_void_p)]
        self.lib.s2st_engine_destroy.argtypes = [C.c_void_p]
        self.lib.s2st_engine_destroy.restype = None
        self.lib.s2st_engine_num_params.argtypes = [C.c_void_p]
        self.lib.s2st_engine_param_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ParamInfo)]
        self.lib.s2st_engine_param_floats.argtypes = [C.c_void_p]
        self.lib.s2st_engine_param_floats.restype = C.c_int64
        self.lib.s2st_engine_buffer_floats.argtypes = [C.c_void_p]
        self.lib.s2st_engine_buffer_floats.restype = C.c_int64
        self.lib.s2st_engine_bind.argtypes = [C.c_void_p] * 4
        self.lib.s2st_engine_workspace_floats.argtypes = [C.c_void_p, C.POINTER(Batch)]
        self.lib.s2st_engine_workspace_floats.restype = C.c_int64
        self.lib.s2st_engine_forward.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Outputs),
                                                 C.c_void_p, C.c_int64, C.c_void_p]
        self.lib.s2st_engine_backward.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_void_p]
        self.lib.s2st_engine_num_segments.argtypes = [C.c_void_p]
        self.lib.s2st_engine_segment_range.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64),
                                                       C.POINTER(C.c_int64)]
        h = C.c_void_p()
        bd.check(self.lib.s2st_engine_create(C.byref(self.cfg), C.byref(h)), "s2st_engine_create")
        self.h = h
        self.n_params = int(self.lib.s2st_engine_param_floats(h))
        self.n_buffers = int(self.lib.s2st_engine_buffer_floats(h))
        self.infos: List[Tuple[str, int, int, Tuple[int, ...], bool]] = []
        for i in range(self.lib.s2st_engine_num_params(h)):
            pi = ParamInfo()
            bd.check(self.lib.s2st_engine_param_info(h, i, C.byref(pi)), "param_info")
            self.infos.append((pi.name.decode(), int(pi.offset), int(pi.numel),
                               tuple(pi.shape[:pi.ndim]), bool(pi.is_buffer)))
        if share_with is not None:
            assert share_with.n_params == self.n_params and share_with.n_buffers == self.n_buffers
            self.params, self.grads, self.buffers = share_with.params, share_with.grads, share_with.buffers
        else:
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=device)
            self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=device)
            self.buffers = torch.zeros(max(self.n_buffers, 4), dtype=torch.float32, device=device)
        self.lib.s2st_engine_bind(h, self.params.data_ptr(), self.grads.data_ptr(), self.buffers.data_ptr())
        # fast mode: bf16 copy of the parameter arena (refreshed by the engine every forward)
        self.params_bf16 = None
        if not precise:
            self.params_bf16 = torch.zeros(self.n_params, dtype=torch.bfloat16, device=device)
            self.lib.s2st_engine_bind_bf16.argtypes = [C.c_void_p, C.c_void_p]
            self.lib.s2st_engine_bind_bf16(h, self.params_bf16.data_ptr())
            import os as _os
            if _os.environ.get("S2ST_NO_WT", "0") != "1":
                self.params_bf16_t = torch.zeros(self.n_params, dtype=torch.bfloat16, device=device)
                self.lib.s2st_engine_bind_bf16_transposed.argtypes = [C.c_void_p, C.c_void_p]
                self.lib.s2st_engine_bind_bf16_transposed(h, self.params_bf16_t.data_ptr())
        self.workspace: Optional[torch.Tensor] = None
        self._outpool: Optional[torch.Tensor] = None
        # Sinusoidal tables, one per width, allocated ONCE at the largest row count the flags allow (source /
        # target positions, the generator's max_iter): prepared batches hold raw pointers into them, so a table is
        # never freed while the engine lives -- should one still have to grow, the old one is retired, not released
        # (its rows are a prefix of the new one, so outstanding pointers stay valid and equal).
        self._pe: Dict[int, torch.Tensor] = {}
        self._pe_retired: List[torch.Tensor] = []
        self._pe_rows = max(int(getattr(args, "max_source_positions", 3000)),
                            int(getattr(args, "max_target_positions", 2400)),
                            int(getattr(args, "max_iter", 6000))) + 2
        for d_ in {self.cfg.enc_dim, self.cfg.dec_dim} | ({self.cfg.asr_dim} if self.cfg.has_asr else set()) | \
                ({self.cfg.st_dim} if self.cfg.has_st else set()):
            self.pe(d_, 1)
        self._plan: Dict[tuple, int] = {}
        self._keep = None
        self.step_seed = 1

    def inference_twin(self) -> "Engine":
        """A second engine over the same weights (created once, kept): for decoding two batches as two interleaved chains."""
        return self.inference_twins(1)[0]

    def inference_twins(self, n: int) -> List["Engine"]:
        """``n`` further engines over the same weights (created once, kept): one per additional decode chain."""
        tws = self.__dict__.setdefault("_twins", [])
        while len(tws) < n:
            # (a decode chain never uses the engine's second stream: created all the same it would take a hardware queue
            # slot and shift every later stream's mapping -- runtime/streams.py)
            # (said to the engine itself -- s2st_engine_allow_side_stream -- not through the process environment: setenv
            #  races with the getenv calls other threads' launches make inside the library)
            tw = Engine(self.args, self.device, precise=bool(self.cfg.precise), share_with=self)
            f = tw.lib.s2st_engine_allow_side_stream
            f.argtypes = [C.c_void_p, C.c_int32]
            bd.check(f(tw.h, 0), "s2st_engine_allow_side_stream")
            tws.append(tw)
        return tws[:n]

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.s2st_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- named views ---------------------------------------------------------------------------
    def named_views(self):
        """[(name, param_view, grad_view | None, is_buffer)] in arena order."""
        out = []
        for name, off, n, shape, is_buf in self.infos:
            if is_buf:
                out.append((name, self.buffers[off:off + n].view(shape), None, True))
            else:
                out.append((name, self.params[off:off + n].view(shape), self.grads[off:off + n].view(shape), False))
        return out

    def pe(self, dim: int, rows: int) -> torch.Tensor:
        t = self._pe.get(dim)
        if t is None or t.shape[0] < rows:
            n = max(rows, self._pe_rows, 64)
            n = 1 << (n - 1).bit_length()
            if t is not None:
                self._pe_retired.append(t)  # batches prepared earlier still point into it
            t = sinusoidal_table(n + 2, dim).to(self.device)
            self._pe[dim] = t
        return t

    # -- batch preparation (host logic: lengths, positions; mirrors lengths_to_padding_mask +
    #    make_positions, fairseq/utils.py:254-264) ---------------------------------------------
    def prepare(self, sample: Dict, training: bool = True, want_attn: bool = False,
                with_loss: bool = True, seed: Optional[int] = None):
        a, dev = self.args, self.device
        ni = sample["net_input"]
        if self.cfg.text_input:
            return self._prepare_text(sample, training, want_attn, with_loss, seed)
        src = ni["src_speech"].to(dev, torch.float32).contiguous()
        B, S, _ = src.shape
        src_lens = ni["src_speech_lens"].cpu().long()
        k = self.cfg.conv_k
        E = conv_out_len(conv_out_len(S, k), k)
        # encoder lengths: floor((len - 1) / 2 + 1) twice (s2st_transformer.py:126-130)
        enc_lens = src_lens.clone()
        # CTC input lengths come from the FBANK lengths, also when the encoder ran on HuBERT frames
        # (s2st_loss.py:231-232; models/s2st_transformer.py front_end_sample keeps them)
        ctc_src = ni.get("ctc_src_speech_lens")
        ctc_lens = (ctc_src if ctc_src is not None else src_lens).cpu().long().clone()
        for _ in range(2):
            enc_lens = ((enc_lens.float() - 1) / 2 + 1).floor().long()
            ctc_lens = (ctc_lens - k + 2 * (k // 2)) // 2 + 1  # s2st_loss.py:231-232
        if self.cfg.has_ctc and with_loss and int(ctc_lens.max()) > E:
            # what F.ctc_loss tells the reference's user in this situation (--use-hubert with --ctc-weight > 0:
            # fbank-rate lengths against HuBERT-rate encoder frames)
            raise RuntimeError(f"Expected input_lengths to have value at most {E}, but got value "
                               f"{int(ctc_lens.max())} (CTC input lengths are derived from the fbank lengths, "
                               f"s2st_loss.py:231-232)")
        prev = ni["prev_output_tokens"].to(dev, torch.float32).contiguous()
        D = prev.shape[1]
        tgt_lens = sample["target_lengths"].cpu().long()

        def speech_pos(lens, T):
            t = torch.arange(T).unsqueeze(0)
            valid = t < lens.unsqueeze(1)
            return torch.where(valid, t + PAD + 1, torch.full_like(t, PAD)).to(torch.int32)

        def token_pos(tok):
            m = tok.ne(PAD).int()
            return ((torch.cumsum(m, dim=1) * m) + PAD).to(torch.int32)

        keep = {"src": src, "prev": prev}
        keep["enc_lens"] = enc_lens.to(torch.int32).to(dev)
        keep["ctc_lens"] = ctc_lens.to(torch.int32).to(dev)
        keep["enc_pos"] = speech_pos(enc_lens, E).contiguous().to(dev)
        keep["tgt_lens"] = tgt_lens.to(torch.int32).to(dev)
        keep["dec_pos"] = speech_pos(tgt_lens, D).contiguous().to(dev)
        b = Batch()
        b.B, b.S, b.D, b.E = B, S, D, E
        b.src, b.prev = src.data_ptr(), prev.data_ptr()
        b.enc_lens, b.enc_pos = keep["enc_lens"].data_ptr(), keep["enc_pos"].data_ptr()
        b.ctc_in_lens = keep["ctc_lens"].data_ptr()
        b.tgt_lens, b.dec_pos = keep["tgt_lens"].data_ptr(), keep["dec_pos"].data_ptr()
        b.pe_enc = self.pe(self.cfg.enc_dim, E + 2).data_ptr()
        b.pe_dec = self.pe(self.cfg.dec_dim, D + 2).data_ptr()
        if with_loss:
            keep["tgt"] = sample["tgt_speech"].to(dev, torch.float32).contiguous()
            b.tgt = keep["tgt"].data_ptr()
        Ls = Lt = 0
        # (text tensors are absent when only the encoder / mel decoder is wanted -- forward_encoder, AR generation:
        # the aux heads and the CTC loss are then skipped by the engine)
        if (self.cfg.has_asr or self.cfg.has_ctc) and sample.get("src_text") is not None:
            st = sample["src_text"].cpu().long().contiguous()
            Ls = st.shape[1]
            keep["src_txt"] = st.to(dev)
            keep["src_txt_lens"] = sample["src_text_len"].to(torch.int32).to(dev)
            b.src_txt, b.src_txt_lens = keep["src_txt"].data_ptr(), keep["src_txt_lens"].data_ptr()
        if self.cfg.has_asr and ni.get("prev_src_text_tokens") is not None:
            pt = ni["prev_src_text_tokens"].cpu().long().contiguous()
            keep["prev_src_txt"] = pt.to(dev)
            keep["src_txt_pos"] = token_pos(pt).contiguous().to(dev)
            b.prev_src_txt, b.src_txt_pos = keep["prev_src_txt"].data_ptr(), keep["src_txt_pos"].data_ptr()
            b.pe_asr = self.pe(self.cfg.asr_dim, Ls + 2).data_ptr()
        if (self.cfg.has_st or self.cfg.has_ctc_tgt) and sample.get("tgt_text") is not None:
            tt = sample["tgt_text"].cpu().long().contiguous()
            Lt = tt.shape[1]
            keep["tgt_txt"] = tt.to(dev)
            keep["tgt_txt_lens"] = sample["tgt_text_len"].to(torch.int32).to(dev)
            b.tgt_txt, b.tgt_txt_lens = keep["tgt_txt"].data_ptr(), keep["tgt_txt_lens"].data_ptr()
        if self.cfg.has_st and sample.get("tgt_text") is not None and ni.get("prev_tgt_text_tokens") is not None:
            pt = ni["prev_tgt_text_tokens"].cpu().long().contiguous()
            keep["prev_tgt_txt"] = pt.to(dev)
            keep["tgt_txt_pos"] = token_pos(pt).contiguous().to(dev)
            b.prev_tgt_txt, b.tgt_txt_pos = keep["prev_tgt_txt"].data_ptr(), keep["tgt_txt_pos"].data_ptr()
            b.pe_st = self.pe(self.cfg.st_dim, Lt + 2).data_ptr()
        if self.cfg.s2t_mode:
            # the model's own text decoder reads the SOURCE text (--test-type asr) or the TARGET text (st):
            # criterions/s2t_loss.py:88-92, s2t_transformer_me.py:300-305; the tokens ride in the source-text slots
            key = "src" if getattr(self, "s2t_test_type", "asr") == "asr" else "tgt"
            pt = ni.get(f"prev_{key}_text_tokens")
            if pt is not None:
                pt = pt.cpu().long().contiguous()
                Ls = pt.shape[1]
                keep["prev_src_txt"] = pt.to(dev)
                keep["src_txt_pos"] = token_pos(pt).contiguous().to(dev)
                b.prev_src_txt, b.src_txt_pos = keep["prev_src_txt"].data_ptr(), keep["src_txt_pos"].data_ptr()
                b.pe_asr = self.pe(self.cfg.dec_dim, Ls + 2).data_ptr()
                keep["src_txt_lens"] = sample[f"{key}_text_len"].to(torch.int32).to(dev)
                b.src_txt_lens = keep["src_txt_lens"].data_ptr()
                if sample.get(f"{key}_text") is not None:
                    keep["src_txt"] = sample[f"{key}_text"].cpu().long().contiguous().to(dev)
                    b.src_txt = keep["src_txt"].data_ptr()
            Lt = 0
        b.Ls, b.Lt = Ls, Lt
        if self.cfg.n_speakers > 0:
            spk = sample.get("speaker")
            if spk is None:
                spk = ni.get("speaker")
            if spk is not None:  # [B, 1] ids (s2st_dataset.py:386-390)
                keep["speaker"] = self._speaker_ids(spk, B)
                b.speaker = keep["speaker"].data_ptr()
        b.ntokens = int(sample["ntokens"])
        b.src_txt_ntokens = int(sample.get("src_txt_ntokens", 0))
        b.tgt_txt_ntokens = int(sample.get("tgt_txt_ntokens", 0))
        b.training, b.want_attn = int(training), int(want_attn)
        if seed is None:
            seed = self.step_seed
            self.step_seed += 1
        b.seed = seed
        return b, keep

    def _speaker_ids(self, spk, B: int) -> Optional[torch.Tensor]:
        """[B, 1] / [B] speaker ids -> device int64 [B], range-checked on the host like nn.Embedding would
        (an id outside the table is an out-of-bounds row read in the kernels)."""
        if spk is None:
            return None
        ids = spk.reshape(-1).to(torch.int64).cpu()
        if int(ids.numel()) != B:
            raise ValueError("sample['speaker'] must hold one id per utterance")
        if B and (int(ids.min()) < 0 or int(ids.max()) >= self.cfg.n_speakers):
            raise IndexError(f"speaker id out of range: ids span [{int(ids.min())}, {int(ids.max())}], the table has "
                             f"{self.cfg.n_speakers} rows (tasks/s2s_translation.py:156-160 sizes it by len(--speaker-to-id))")
        return ids.contiguous().to(self.device)

    def _prepare_text(self, sample, training, want_attn, with_loss, seed):
        """t2s_transformer: the encoder reads token ids -- ``sample["src_text"]`` / ``["src_text_len"]``, what the
        reference's t2s criterion passes as ``src_tokens`` / ``src_lengths`` (criterions/t2s_loss.py:110-121)."""
        dev = self.device
        ni = sample["net_input"]
        tok = sample["src_text"].cpu().long().contiguous()
        B, S = tok.shape
        lens = sample["src_text_len"].cpu().long()
        t = torch.arange(S).unsqueeze(0)
        pos = torch.where(t < lens.unsqueeze(1), t + PAD + 1, torch.full_like(t, PAD)).to(torch.int32)
        # (positions from the PAD mask of the tokens themselves, t2s_transformer.py:93-95: right-padded batches)
        prev = ni["prev_output_tokens"].to(dev, torch.float32).contiguous()
        D = prev.shape[1]
        tgt_lens = sample["target_lengths"].cpu().long()
        td = torch.arange(D).unsqueeze(0)
        dpos = torch.where(td < tgt_lens.unsqueeze(1), td + PAD + 1, torch.full_like(td, PAD)).to(torch.int32)
        keep = {"prev": prev, "src_txt": tok.to(dev), "src_txt_lens": lens.to(torch.int32).to(dev),
                "enc_lens": lens.to(torch.int32).to(dev), "enc_pos": pos.contiguous().to(dev),
                "tgt_lens": tgt_lens.to(torch.int32).to(dev), "dec_pos": dpos.contiguous().to(dev)}
        b = Batch()
        b.B, b.S, b.D, b.E, b.Ls, b.Lt = B, S, D, S, S, 0
        b.prev = prev.data_ptr()
        b.src_txt, b.src_txt_lens = keep["src_txt"].data_ptr(), keep["src_txt_lens"].data_ptr()
        b.enc_lens, b.enc_pos = keep["enc_lens"].data_ptr(), keep["enc_pos"].data_ptr()
        b.ctc_in_lens = keep["enc_lens"].data_ptr()
        b.tgt_lens, b.dec_pos = keep["tgt_lens"].data_ptr(), keep["dec_pos"].data_ptr()
        b.pe_enc = self.pe(self.cfg.enc_dim, S + 2).data_ptr()
        b.pe_dec = self.pe(self.cfg.dec_dim, D + 2).data_ptr()
        if with_loss:
            keep["tgt"] = sample["tgt_speech"].to(dev, torch.float32).contiguous()
            b.tgt = keep["tgt"].data_ptr()
        if self.cfg.n_speakers > 0:
            spk = sample.get("speaker")
            if spk is None:
                # t2s_transformer.py:107-109 calls self.embed_speaker(speaker) unconditionally once the table exists
                raise ValueError("a t2s_transformer built with --speaker-to-id needs sample['speaker']")
            keep["speaker"] = self._speaker_ids(spk, B)
            b.speaker = keep["speaker"].data_ptr()
        b.ntokens = int(sample["ntokens"])
        b.training, b.want_attn = int(training), int(want_attn)
        if seed is None:
            seed = self.step_seed
            self.step_seed += 1
        b.seed = seed
        return b, keep

    def forward(self, sample, training: bool = True, want_attn: bool = False,
                with_loss: bool = True, seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """``sample`` is a collater dict, or the ``(Batch, keep)`` pair returned by ``prepare``
        (device-resident inputs, reused across steps by the data loader / bench).

        After ``reserve()`` the tensor outputs (``post_feat_out``, ``encoder_out``, logits, ...) are slices of one
        output pool: they are valid until the NEXT forward of this engine; clone what has to live longer.
        ``stats`` (loss terms, counts) is its own tensor and stays valid."""
        if isinstance(sample, tuple):
            b, keep = sample
            b.training, b.want_attn = int(training), int(want_attn)
            if seed is None:
                seed = self.step_seed
                self.step_seed += 1
            b.seed = seed
            with_loss = with_loss and bool(b.tgt)
        else:
            b, keep = self.prepare(sample, training, want_attn, with_loss, seed)
        geo = (b.B, b.S, b.D, b.Ls, b.Lt, bool(b.tgt), b.training)
        need = self._plan.get(geo)
        if need is None:
            need = int(self.lib.s2st_engine_workspace_floats(self.h, C.byref(b)))
            self._plan[geo] = need
        if need < 0:
            raise bd.S2STHipError(f"workspace planning failed ({need})")
        self._grow_workspace(need, int(need * 1.05) + 4096)
        dev, c = self.device, self.cfg
        B, D, E = b.B, b.D, b.E
        pool = self._outpool
        cursor = [0]

        def buf(*shape):
            # outputs are carved from the pool reserved by reserve() when it is large enough (no
            # allocator traffic inside a training loop over many batch geometries)
            n = 1
            for d_ in shape:
                n *= d_
            n64 = (n + 63) // 64 * 64
            if pool is not None and cursor[0] + n64 <= pool.numel():
                t = pool[cursor[0]:cursor[0] + n].view(shape)
                cursor[0] += n64
                return t
            return torch.empty(shape, dtype=torch.float32, device=dev)

        o = {"post_feat_out": buf(B, D, c.out_dim), "feature_out": buf(B, D, c.out_dim),
             "eos_out": buf(B, D, 1), "encoder_out": buf(B, E, c.enc_dim),
             # the loss / logging scalars are read lazily, possibly after later forwards (LazyLog, update_freq > 1):
             # they get a tensor of their own, never a slice of the pool, which the next forward overwrites
             "stats": torch.empty(32, dtype=torch.float32, device=dev)}
        out = Outputs()
        out.post_feat, out.feat, out.eos = o["post_feat_out"].data_ptr(), o["feature_out"].data_ptr(), o["eos_out"].data_ptr()
        out.enc_out, out.stats = o["encoder_out"].data_ptr(), o["stats"].data_ptr()
        if want_attn:
            o["attn"] = buf(B, E, D)
            out.attn = o["attn"].data_ptr()
        if c.has_asr or c.has_ctc:
            o["tap0"] = buf(B, E, c.enc_dim)
            out.tap0 = o["tap0"].data_ptr()
        if c.has_st:
            o["tap1"] = buf(B, E, c.enc_dim)
            out.tap1 = o["tap1"].data_ptr()
        if c.has_asr:
            o["asr_logits"] = buf(B, b.Ls, c.src_vocab)
            out.asr_logits = o["asr_logits"].data_ptr()
        if c.s2t_mode and b.Ls > 0:  # the text decoder's logits over the TARGET dictionary
            o["asr_logits"] = buf(B, b.Ls, c.tgt_vocab)
            out.asr_logits = o["asr_logits"].data_ptr()
        if c.has_st:
            o["st_logits"] = buf(B, b.Lt, c.tgt_vocab)
            out.st_logits = o["st_logits"].data_ptr()
        if c.has_ctc and with_loss:
            o["ctc_lprobs"] = buf(B, E, c.src_vocab)
            out.ctc_lprobs = o["ctc_lprobs"].data_ptr()
        if self.params_bf16 is not None and getattr(self, "_ph_version", None) == self.params._version:
            self.lib.s2st_engine_bf16_is_fresh(self.h)  # written by the optimizer kernel, parameters untouched since
        self._ph_version = None
        if os.environ.get("S2ST_POISON_WORKSPACE"):  # debugging aid: a kernel that relies on a cleared workspace shows
            self.workspace.fill_(float("nan"))
        rc = self.lib.s2st_engine_forward(self.h, C.byref(b), C.byref(out), self.workspace.data_ptr(),
                                          self.workspace.numel(), bd.stream_ptr())
        bd.check(rc, "s2st_engine_forward")
        self._keep = (b, keep, o)  # inputs/outputs must outlive the backward
        o["encoder_lens"] = keep["enc_lens"]
        return o

    # -- AR decoding (config 5): encoder once, then one decoder step per output frame -------------
    def decode_begin(self, src: torch.Tensor, src_lens: torch.Tensor, max_steps: int,
                     speaker: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Runs the encoder (eval mode) and fills the decoding caches for ``max_steps`` frames.  ``src``: fbank / HuBERT
        features [B, S, in_dim], or -- for a text-input model (t2s_transformer; speech_generator_for_s2st.py:60-64) -- the
        token ids [B, S] of ``sample["src_text"]``.  ``speaker``: [B, 1] ids of a speaker-conditioned model (kept for the
        decoding steps)."""
        dev, c = self.device, self.cfg
        b = Batch()
        if c.text_input:
            tok = src.cpu().long().contiguous()
            if tok.dim() != 2:
                raise ValueError("a text-input model decodes from token ids [B, S]")
            B, S = tok.shape
            E = S
            enc_lens = src_lens.cpu().long().clone()
            keep = {"src_txt": tok.to(dev), "src_txt_lens": enc_lens.to(torch.int32).to(dev)}
            b.src_txt, b.src_txt_lens, b.Ls = keep["src_txt"].data_ptr(), keep["src_txt_lens"].data_ptr(), S
        else:
            src = src.to(dev, torch.float32).contiguous()
            B, S, _ = src.shape
            k = c.conv_k
            E = conv_out_len(conv_out_len(S, k), k)
            enc_lens = src_lens.cpu().long().clone()
            for _ in range(2):
                enc_lens = ((enc_lens.float() - 1) / 2 + 1).floor().long()
            keep = {"src": src}
            b.src = src.data_ptr()
        t = torch.arange(E).unsqueeze(0)
        enc_pos = torch.where(t < enc_lens.unsqueeze(1), t + PAD + 1, torch.full_like(t, PAD)).to(torch.int32)
        keep["enc_lens"] = enc_lens.to(torch.int32).to(dev)
        keep["enc_pos"] = enc_pos.contiguous().to(dev)
        b.B, b.S, b.D, b.E = B, S, 1, E
        b.enc_lens, b.enc_pos = keep["enc_lens"].data_ptr(), keep["enc_pos"].data_ptr()
        b.ctc_in_lens = keep["enc_lens"].data_ptr()
        b.pe_enc = self.pe(c.enc_dim, E + 2).data_ptr()
        b.pe_dec = self.pe(c.dec_dim, max_steps + 2).data_ptr()
        b.training, b.seed = 0, 0
        if c.n_speakers > 0 and speaker is not None:
            keep["speaker"] = self._speaker_ids(speaker, B)
            b.speaker = keep["speaker"].data_ptr()
        elif c.n_speakers > 0 and c.text_input:
            raise ValueError("a t2s_transformer built with --speaker-to-id needs sample['speaker']")
        # workspace: the encoder forward of this geometry (planned with the full schedule: a superset)
        geo = ("enc", B, S)
        need = self._plan.get(geo)
        if need is None:
            dry = Batch()
            for f, _ in Batch._fields_:
                setattr(dry, f, getattr(b, f))
            need = int(self.lib.s2st_engine_workspace_floats(self.h, C.byref(dry)))
            self._plan[geo] = need
        if need < 0:
            raise bd.S2STHipError(f"workspace planning failed ({need})")
        need = max(need, 64 << 20)
        self._grow_workspace(need, int(need * 1.05) + 4096)
        o = {"encoder_out": torch.empty(B, E, c.enc_dim, device=dev)}
        out = Outputs()
        out.enc_out = o["encoder_out"].data_ptr()
        if c.has_asr or c.has_ctc:
            o["tap0"] = torch.empty(B, E, c.enc_dim, device=dev)
            out.tap0 = o["tap0"].data_ptr()
        if c.has_st:
            o["tap1"] = torch.empty(B, E, c.enc_dim, device=dev)
            out.tap1 = o["tap1"].data_ptr()
        self.lib.s2st_engine_decode_state_floats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        self.lib.s2st_engine_decode_state_floats.restype = C.c_int64
        n_state = int(self.lib.s2st_engine_decode_state_floats(self.h, B, E, max_steps))
        state = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.lib.s2st_engine_decode_begin.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Outputs), C.c_void_p,
                                                      C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        bd.check(self.lib.s2st_engine_decode_begin(self.h, C.byref(b), C.byref(out), state.data_ptr(), n_state,
                                                   max_steps, self.workspace.data_ptr(), self.workspace.numel(),
                                                   bd.stream_ptr()), "s2st_engine_decode_begin")
        self._dec = dict(B=B, E=E, max_steps=max_steps, state=state, keep=keep, batch=b, outs=o)
        o["encoder_lens"] = keep["enc_lens"]
        return o

    def decode_row_map(self, row_map: Optional[torch.Tensor]):
        """After ``decode_begin`` of a MERGED batch: ``row_map`` [B] int32 (device) = every row's index inside its own batch."""
        f = self.lib.s2st_engine_decode_row_map
        f.argtypes = [C.c_void_p, C.c_void_p]
        self._dec["row_map"] = row_map  # (kept alive for the run)
        bd.check(f(self.h, bd.ptr(row_map)), "s2st_engine_decode_row_map")

    def decode_buffers(self, max_steps: int):
        """Whole-run buffers of an AR decode (round 4: nothing is allocated, filled or uploaded per step): outputs
        ``feat`` [steps, B, out_dim] / ``eos`` [steps, B] / ``attn`` [steps, B, E], the positions of every step, and the stop
        rule's device state (``finished``, ``out_lens`` = max_steps while running, ``klen`` = key lengths of the next step's
        self-attention, ``n_done`` [steps])."""
        d, dev, c = self._dec, self.device, self.cfg
        B, E = d["B"], d["E"]
        pos = (torch.arange(max_steps, dtype=torch.int32) + PAD + 1).view(-1, 1).expand(-1, B).contiguous().to(dev)
        bufs = dict(feat=torch.empty(max_steps, B, c.out_dim, device=dev), eos=torch.empty(max_steps, B, device=dev),
                    attn=torch.empty(max_steps, B, E, device=dev), pos=pos,
                    finished=torch.zeros(B, dtype=torch.int32, device=dev),
                    out_lens=torch.full((B,), max_steps, dtype=torch.int32, device=dev),
                    klen=torch.ones(B, dtype=torch.int32, device=dev),  # step 0: one key
                    n_done=torch.zeros(max_steps, dtype=torch.int32, device=dev),
                    bos=torch.zeros(B, c.out_dim, device=dev))
        d["bufs"] = bufs
        return bufs

    def decode_step_into(self, step: int, seed: int, thr: float, max_iter: int):
        """Step ``step`` of a buffered decode: input = the previous step's feature row block (the zero frame at step 0),
        outputs into the run's buffers, then the stop rule's update -- all stream-ordered, no host round trip."""
        d, c = self._dec, self.cfg
        b = d["bufs"]
        prev = b["bos"] if step == 0 else b["feat"][step - 1]
        f = self.lib.s2st_engine_decode_step
        f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        bd.check(f(self.h, step, prev.data_ptr(), b["pos"][step].data_ptr(), b["klen"].data_ptr(), seed,
                   b["feat"][step].data_ptr(), b["eos"][step].data_ptr(), b["attn"][step].data_ptr(),
                   self.workspace.data_ptr(), self.workspace.numel(), bd.stream_ptr()), "s2st_engine_decode_step")
        bd.call("s2st_decode_stop_update_i32", b["eos"][step], float(thr), step, int(max_iter), d["B"], b["finished"],
                b["out_lens"], b["klen"], b["n_done"])

    # ---- the step in its graph-replayable form (include/s2st_hip.h s2st_decode_replay; round 5) ------------------------
    def decode_replay_prepare(self, seed0: int, thr: float, max_iter: int, graph: bool = True) -> bool:
        """After ``decode_buffers``: set up this run's steps in the form whose step-dependent values live in device memory
        (step counter, prenet seeds, input frame, position row), so that ONE captured step is replayed for the whole run
        -- a HIP graph launch per step instead of ~55 kernel launches (config 5 was bound by the host's enqueue rate).
        ``graph=False``: the same calls enqueued directly every step (what the CPU tests run: the emulator has no graphs).
        False where the run's steps do not qualify (the caller keeps ``decode_step_into``)."""
        d, dev, c = self._dec, self.device, self.cfg
        lib = self.lib
        lib.s2st_engine_decode_replay_supported.argtypes = [C.c_void_p]
        lib.s2st_engine_decode_replay_supported.restype = C.c_int32
        if not int(lib.s2st_engine_decode_replay_supported(self.h)):
            return False
        B, E = d["B"], d["E"]
        r = dict(step=torch.zeros(1, dtype=torch.int32, device=dev), seeds=torch.zeros(8, dtype=torch.int64, device=dev),
                 cur_feat=torch.zeros(B, c.out_dim, device=dev), cur_eos=torch.zeros(B, device=dev),
                 cur_attn=torch.zeros(B, E, device=dev), pe_cur=torch.zeros(c.dec_dim, device=dev))
        st = DecodeReplay(*(r[k].data_ptr() for k in ("step", "seeds", "cur_feat", "cur_eos", "cur_attn", "pe_cur")))
        seed0 &= (1 << 64) - 1
        d["replay"] = dict(bufs=r, st=st, seed0=seed0, thr=float(thr), max_iter=int(max_iter), graph=None)
        lib.s2st_engine_decode_replay_begin.argtypes = [C.c_void_p, C.POINTER(DecodeReplay), C.c_uint64, C.c_void_p]
        lib.s2st_engine_decode_step_replay.argtypes = [C.c_void_p, C.POINTER(DecodeReplay), C.c_void_p, C.c_void_p, C.c_int64,
                                                       C.c_void_p]
        lib.s2st_engine_decode_replay_commit.argtypes = [C.c_void_p, C.POINTER(DecodeReplay), C.c_uint64, C.c_float, C.c_int32,
                                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_void_p, C.c_void_p]
        if graph and dev.type == "cuda":
            from . import streams
            g = torch.cuda.CUDAGraph()
            cap = streams.capture_stream(dev)
            cap.wait_stream(torch.cuda.current_stream())
            # (thread_local: the vocoder's phase-draw thread keeps uploading on its own stream while this thread captures)
            with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                self._decode_replay_enqueue()
            torch.cuda.current_stream().wait_stream(cap)
            d["replay"]["graph"] = g
        bd.check(lib.s2st_engine_decode_replay_begin(self.h, C.byref(st), seed0, bd.stream_ptr()),
                 "s2st_engine_decode_replay_begin")
        return True

    def _decode_replay_enqueue(self):
        d = self._dec
        rp, b = d["replay"], d["bufs"]
        bd.check(self.lib.s2st_engine_decode_step_replay(self.h, C.byref(rp["st"]), b["klen"].data_ptr(),
                                                         self.workspace.data_ptr(), self.workspace.numel(), bd.stream_ptr()),
                 "s2st_engine_decode_step_replay")
        bd.check(self.lib.s2st_engine_decode_replay_commit(self.h, C.byref(rp["st"]), rp["seed0"], rp["thr"], rp["max_iter"],
                                                           b["finished"].data_ptr(), b["out_lens"].data_ptr(),
                                                           b["klen"].data_ptr(), b["n_done"].data_ptr(), b["feat"].data_ptr(),
                                                           b["eos"].data_ptr(), b["attn"].data_ptr(), bd.stream_ptr()),
                 "s2st_engine_decode_replay_commit")

    def decode_replay_step(self):
        """The next step of a run ``decode_replay_prepare`` set up (same outputs as ``decode_step_into`` step by step)."""
        g = self._dec["replay"]["graph"]
        if g is not None:
            g.replay()
        else:
            self._decode_replay_enqueue()

    def decode_step(self, step: int, prev: torch.Tensor, seed: int, want_attn: bool = True,
                    self_klen: Optional[torch.Tensor] = None):
        """prev [B, out_dim] -> (feature_out [B, out_dim], eos_prob [B], attn [B, E] | None)."""
        d, dev, c = self._dec, self.device, self.cfg
        B, E = d["B"], d["E"]
        prev = prev.to(dev, torch.float32).contiguous()
        pos = torch.full((B,), step + PAD + 1, dtype=torch.int32, device=dev)
        feat = torch.empty(B, c.out_dim, device=dev)
        eos = torch.empty(B, device=dev)
        attn = torch.empty(B, E, device=dev) if want_attn else None
        klen = self_klen.to(dev, torch.int32).contiguous() if self_klen is not None else None
        self.lib.s2st_engine_decode_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int64, C.c_void_p]
        bd.check(self.lib.s2st_engine_decode_step(self.h, step, prev.data_ptr(), pos.data_ptr(), bd.ptr(klen), seed,
                                                  feat.data_ptr(),
                                                  eos.data_ptr(), bd.ptr(attn), self.workspace.data_ptr(),
                                                  self.workspace.numel(), bd.stream_ptr()), "s2st_engine_decode_step")
        d["last"] = (prev, pos, klen)
        return feat, eos, attn

    def postnet_eval(self, feat: torch.Tensor) -> torch.Tensor:
        """feat [B, D, out_dim] -> feat + postnet(feat) with BatchNorm running statistics."""
        dev = self.device
        feat = feat.to(dev, torch.float32).contiguous()
        B, D, _ = feat.shape
        out = torch.empty_like(feat)
        need = 64 * B * (D + 8) * max(self.cfg.postnet_dim, self.cfg.out_dim) + (8 << 20)
        self._grow_workspace(need, need)
        self.lib.s2st_engine_postnet_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                      C.c_void_p, C.c_int64, C.c_void_p]
        bd.check(self.lib.s2st_engine_postnet_eval(self.h, feat.data_ptr(), B, D, out.data_ptr(),
                                                   self.workspace.data_ptr(), self.workspace.numel(), bd.stream_ptr()),
                 "s2st_engine_postnet_eval")
        return out

    def _grow_workspace(self, need: int, alloc: int):
        """(Re)allocate the per-call workspace when ``need`` floats do not fit.  Growing is rare (``reserve`` sizes it
        up front) but must not pull the old buffer from under kernels of the previous call: those on the engine's second
        stream are unknown to torch's allocator, which would hand the memory out again at once."""
        if self.workspace is not None and self.workspace.numel() >= need:
            return
        if self.workspace is not None and self.workspace.is_cuda:
            torch.cuda.synchronize(self.device)
        self.workspace = None
        self.workspace = torch.empty(alloc, dtype=torch.float32, device=self.device)

    def reserve(self, batches, training: bool = True, want_attn: bool = False):
        """Size the activation workspace and the output pool for the largest of ``batches`` (prepared
        ``(Batch, keep)`` pairs) up front -- what a data loader with a max-tokens bound does once -- so a
        loop over many batch geometries performs no device allocation (hipMalloc / hipFree stall the
        queue)."""
        need, out = 0, 0
        c = self.cfg
        for b, _ in batches:
            b.training, b.want_attn = int(training), int(want_attn)
            geo = (b.B, b.S, b.D, b.Ls, b.Lt, bool(b.tgt), b.training)
            n = self._plan.get(geo)
            if n is None:
                n = int(self.lib.s2st_engine_workspace_floats(self.h, C.byref(b)))
                self._plan[geo] = n
            if n < 0:
                raise bd.S2STHipError(f"workspace planning failed ({n})")
            need = max(need, n)
            o = 2 * b.B * b.D * c.out_dim + b.B * b.D + 3 * b.B * b.E * c.enc_dim + b.B * b.E * b.D \
                + b.B * b.Ls * c.src_vocab + b.B * b.Lt * c.tgt_vocab + b.B * b.E * c.src_vocab + 32
            out = max(out, o + 64 * 16)
        self._grow_workspace(need, int(need * 1.05) + 4096)
        if self._outpool is None or self._outpool.numel() < out:
            self._outpool = torch.empty(out, dtype=torch.float32, device=self.device)

    def mark_bf16_fresh(self):
        """The optimizer kernel just wrote ``params_bf16`` together with the update: remember the parameter
        tensor's version so the next forward can skip its refresh pass -- unless something else (a state-dict
        load, a manual copy_) touches ``params`` in between, which bumps the version."""
        self._ph_version = self.params._version

    # ---- dropout-site log (test instrumentation: include/s2st_hip.h, s2st_engine_site_log) ----------------------------
    def site_log(self, on: bool = True):
        """Make every later forward record its dropout sites (seed, kind, p, element geometry, place)."""
        f = self.lib.s2st_engine_site_log
        f.argtypes = [C.c_void_p, C.c_int32]
        bd.check(f(self.h, 1 if on else 0), "s2st_engine_site_log")

    def dropout_sites(self) -> Dict[str, "DropoutSite"]:
        """The last forward's sites by name ``<ctx>/<kind><ordinal>`` (e.g. ``enc.L3/lin1``, ``dec.L0/attn1``, ``post/norm2``)."""
        f = self.lib.s2st_engine_site_log_get
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        f.restype = C.c_int32
        n = int(f(self.h, None, 0))
        recs = (DropoutSite * max(n, 1))()
        f(self.h, C.cast(recs, C.c_void_p), n)
        out = {}
        for r in recs[:n]:
            name = f"{r.ctx.decode()}/{SITE_KIND[r.kind]}{r.ordinal}"
            assert name not in out, name
            out[name] = r
        return out

    def dropout_keep_mask(self, site: "DropoutSite") -> torch.Tensor:
        """The site's keep decisions as a 0/1 float tensor in the ENGINE's element geometry (dims of the record: rows x N,
        or B x H x T x ld for attention probabilities), regenerated by ``s2st_dropout_f32`` over ones -- the same
        (seed, element index) hash every fused epilogue evaluates."""
        d = [int(v) for v in site.dims]
        shape = (d[0], d[1], d[2], d[4]) if site.kind == 2 else (d[0], d[1])
        n = 1
        for v in shape:
            n *= v
        ones = torch.ones(n, dtype=torch.float32, device=self.device)
        y = torch.empty_like(ones)
        bd.call("s2st_dropout_f32", ones, y, n, 1.0, float(site.p), int(site.seed), 0)
        keep = (y != 0).float()
        return keep.view(*shape)

    def side_stream(self):
        """torch view of the engine's second stream (None on the emulator / when disabled)."""
        if bd.is_emulator() or not torch.cuda.is_available():
            return None
        self.lib.s2st_engine_side_stream.argtypes = [C.c_void_p]
        self.lib.s2st_engine_side_stream.restype = C.c_void_p
        p = self.lib.s2st_engine_side_stream(self.h)
        return torch.cuda.ExternalStream(p) if p else None

    def num_segments(self) -> int:
        return int(self.lib.s2st_engine_num_segments(self.h))

    def segment_range(self, i: int) -> Tuple[int, int]:
        lo, hi = C.c_int64(), C.c_int64()
        bd.check(self.lib.s2st_engine_segment_range(self.h, i, C.byref(lo), C.byref(hi)), "segment_range")
        return int(lo.value), int(hi.value)

    def backward(self, gscale: float = 1.0,
                 on_segment: Optional[Callable[[int, int, int], None]] = None):
        """grads += gscale * dLoss/dparams.  ``on_segment(i, lo, hi)`` is called after tape
        segment i has been enqueued: gradients in arena range [lo, hi) are then final (used to
        overlap the gradient all-reduce with the rest of the backward)."""
        if on_segment is None:
            bd.check(self.lib.s2st_engine_backward(self.h, gscale, -1, bd.stream_ptr()), "s2st_engine_backward")
            return
        for i in range(self.num_segments()):
            bd.check(self.lib.s2st_engine_backward(self.h, gscale, i, bd.stream_ptr()), "s2st_engine_backward")
            lo, hi = self.segment_range(i)
            if hi > lo:
                on_segment(i, lo, hi)

    def adam_overlapped(self, exp_avg, exp_avg_sq, sumsq_parts, n_parts, gmul, gmul_dev, max_norm, lr, beta1, beta2, eps, wd,
                        step, gnorm_out, skipped, write_bf16: bool, n_chunks: int = 8):
        """``s2st_engine_adam_overlapped``: the fused scale / clip / Adam update in chunks on the engine's second stream;
        the next ``forward`` waits chunk by chunk.  Anything else that reads the parameters calls ``wait_optimizer``."""
        f = self.lib.s2st_engine_adam_overlapped
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_float, C.c_float,
                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                      C.c_void_p]
        for t in (exp_avg, exp_avg_sq, sumsq_parts, gnorm_out, skipped):
            bd.require_device(t)
        bd.check(f(self.h, exp_avg.data_ptr(), exp_avg_sq.data_ptr(), sumsq_parts.data_ptr(), int(n_parts), float(gmul),
                   bd.ptr(gmul_dev), float(max_norm), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step),
                   gnorm_out.data_ptr(), skipped.data_ptr(), 1 if write_bf16 else 0, int(n_chunks), bd.stream_ptr()),
                 "s2st_engine_adam_overlapped")

    def wait_optimizer(self):
        """The current stream waits for an overlapped update still in flight (no-op otherwise)."""
        f = self.lib.s2st_engine_wait_optimizer
        f.argtypes = [C.c_void_p, C.c_void_p]
        bd.check(f(self.h, bd.stream_ptr()), "s2st_engine_wait_optimizer")

    def zero_grad(self):
        self.grads.zero_()
