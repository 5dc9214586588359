"""Frozen HuBERT front end of config 4 (``--use-hubert true``) on the HIP path.

Mirrors what the reference calls on it: ``HubertModel.extract_features(source, padding_mask)``
-> ``(features [B, T', E], frame padding mask [B, T'])`` in eval mode with ``mask=False``
(fairseq/models/hubert/hubert.py:518-534, called from
examples/s2s_trans/models/s2st_transformer.py:245-252).  ``state_dict`` keys / shapes are the
reference ``HubertModel``'s for every tensor the forward reads; the pre-training-only tensors
(``mask_emb``, ``final_proj.*``, ``label_embs_concat``) are accepted and ignored by
``load_state_dict``.  All arithmetic runs in libs2st_hip.so (``s2st_hubert_*``); torch holds the
arenas.  The engine keeps conv weights in GEMM layout and the weight-normed positional conv as
its effective weight, so loading converts layouts (data movement only, done once: the module is
frozen).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from ..runtime import binding as bd
from ..runtime.engine import ParamInfo

BASE_CONV = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


class HubertConfigC(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_dim", C.c_int32 * 8), ("conv_k", C.c_int32 * 8),
                ("conv_stride", C.c_int32 * 8)] + [(n, C.c_int32) for n in (
                    "embed", "layers", "heads", "ffn", "conv_pos", "conv_pos_groups", "precise")]


class HubertFrontend:
    """hubert_base geometry by default (conv stack 7 layers, 12 x 768, 12 heads, ffn 3072, conv_pos 128/16)."""

    def __init__(self, device: torch.device, conv=None, embed=768, layers=12, heads=12, ffn=3072, conv_pos=128,
                 conv_pos_groups=16, precise: bool = False):
        self.device = device
        self.conv = list(conv or BASE_CONV)
        self.embed, self.layers, self.heads, self.ffn = embed, layers, heads, ffn
        self.conv_pos, self.groups, self.precise = conv_pos, conv_pos_groups, precise
        lib = self.lib = bd.lib()
        cfg = HubertConfigC()
        cfg.n_conv = len(self.conv)
        for i, (c, k, s) in enumerate(self.conv):
            cfg.conv_dim[i], cfg.conv_k[i], cfg.conv_stride[i] = c, k, s
        cfg.embed, cfg.layers, cfg.heads, cfg.ffn = embed, layers, heads, ffn
        cfg.conv_pos, cfg.conv_pos_groups, cfg.precise = conv_pos, conv_pos_groups, int(precise)
        lib.s2st_hubert_create.argtypes = [C.POINTER(HubertConfigC), C.POINTER(C.c_void_p)]
        lib.s2st_engine_destroy.argtypes = [C.c_void_p]
        lib.s2st_engine_destroy.restype = None
        lib.s2st_engine_num_params.argtypes = [C.c_void_p]
        lib.s2st_engine_param_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ParamInfo)]
        lib.s2st_engine_param_floats.argtypes = [C.c_void_p]
        lib.s2st_engine_param_floats.restype = C.c_int64
        lib.s2st_engine_bind.argtypes = [C.c_void_p] * 4
        lib.s2st_engine_bind_bf16.argtypes = [C.c_void_p, C.c_void_p]
        lib.s2st_hubert_out_frames.argtypes = [C.c_void_p, C.c_int32]
        lib.s2st_hubert_workspace_floats.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.s2st_hubert_workspace_floats.restype = C.c_int64
        lib.s2st_hubert_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_void_p]
        h = C.c_void_p()
        bd.check(lib.s2st_hubert_create(C.byref(cfg), C.byref(h)), "s2st_hubert_create")
        self.h = h
        self.n_params = int(lib.s2st_engine_param_floats(h))
        self.infos: List[Tuple[str, int, int, Tuple[int, ...]]] = []
        for i in range(lib.s2st_engine_num_params(h)):
            pi = ParamInfo()
            bd.check(lib.s2st_engine_param_info(h, i, C.byref(pi)), "param_info")
            self.infos.append((pi.name.decode(), int(pi.offset), int(pi.numel), tuple(pi.shape[:pi.ndim])))
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        lib.s2st_engine_bind(h, self.params.data_ptr(), None, None)
        self.params_bf16 = None
        if not precise:
            self.params_bf16 = torch.zeros(self.n_params, dtype=torch.bfloat16, device=device)
            lib.s2st_engine_bind_bf16(h, self.params_bf16.data_ptr())
        self.workspace: Optional[torch.Tensor] = None
        self._plan: Dict[Tuple[int, int], int] = {}
        self._extra: Dict[str, torch.Tensor] = {}  # reference tensors the forward does not read

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.s2st_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def eval(self):  # the reference calls hubert.eval() every forward (s2st_transformer.py:246)
        return self

    # -- parameters: reference names / layouts <-> engine arena ------------------------------------
    def _view(self, name):
        for n, off, numel, shape in self.infos:
            if n == name:
                return self.params[off:off + numel].view(shape)
        raise KeyError(name)

    def reference_shapes(self) -> Dict[str, Tuple[int, ...]]:
        s: Dict[str, Tuple[int, ...]] = {}
        for n, _, _, shape in self.infos:
            if n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight"):
                s[n] = (shape[0], shape[2], shape[1])  # engine [O][k][I] <- reference [O][I][k]
            elif n == "encoder.pos_conv.0.weight":
                Eg = self.embed // self.groups
                s["encoder.pos_conv.0.weight_g"] = (1, 1, self.conv_pos)
                s["encoder.pos_conv.0.weight_v"] = (self.embed, Eg, self.conv_pos)
            else:
                s[n] = shape
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        need = self.reference_shapes()
        missing = [k for k in need if k not in sd]
        if missing and strict:
            raise KeyError(f"missing HuBERT tensors: {missing[:5]}")
        dev = self.device
        for n, _, _, shape in self.infos:
            if n == "encoder.pos_conv.0.weight":
                g = sd["encoder.pos_conv.0.weight_g"].to(dev, torch.float32)
                v = sd["encoder.pos_conv.0.weight_v"].to(dev, torch.float32)
                self._extra["encoder.pos_conv.0.weight_g"] = g.clone()
                self._extra["encoder.pos_conv.0.weight_v"] = v.clone()
                # nn.utils.weight_norm(dim=2): w[:, :, k] = g[k] * v[:, :, k] / ||v[:, :, k]||_F
                # (wav2vec2.py:836).  A one-off host-side parameter fold of a frozen module.
                w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
                G, Eg = self.groups, self.embed // self.groups
                self._view(n).copy_(w.view(G, Eg, Eg, self.conv_pos).permute(0, 1, 3, 2))
            elif n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight"):
                self._view(n).copy_(sd[n].to(dev, torch.float32).permute(0, 2, 1))
            else:
                self._view(n).copy_(sd[n].to(dev, torch.float32).view(shape))
        for k, v in sd.items():
            if k not in need:
                self._extra[k] = v.detach().clone()
        self.invalidate_bf16()  # (explicit: the copies above also bump torch's version counter)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        for n, _, _, shape in self.infos:
            if n == "encoder.pos_conv.0.weight":
                continue
            v = self._view(n)
            if n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight"):
                v = v.permute(0, 2, 1).contiguous()
            out[n] = v.clone()
        out.update({k: v.clone() for k, v in self._extra.items()})
        return out

    # -- forward --------------------------------------------------------------------------------------
    def _to_device_async(self, t: torch.Tensor) -> torch.Tensor:
        """Small host tensor -> device without stalling the queue: a pageable H2D copy synchronises with
        everything already enqueued, and a fresh pin_memory() per call makes the caching host allocator
        grow (its blocks are still in flight).  A ring of reusable pinned staging buffers, each guarded by
        the event of its last copy, does neither."""
        if self.device.type != "cuda":
            return t
        ring = self.__dict__.setdefault("_pin_ring", {})
        key = (t.dtype, tuple(t.shape))
        slots = ring.setdefault(key, {"i": 0, "bufs": []})
        if len(slots["bufs"]) < 8:
            slots["bufs"].append([torch.empty(t.shape, dtype=t.dtype).pin_memory(), None])
            buf = slots["bufs"][-1]
        else:
            buf = slots["bufs"][slots["i"] % 8]
            slots["i"] += 1
            if buf[1] is not None:
                buf[1].synchronize()  # only waits if 8 later copies are still behind this one
        buf[0].copy_(t)
        out = buf[0].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        buf[1] = ev
        return out

    def out_frames(self, n_samples: int) -> int:
        return int(self.lib.s2st_hubert_out_frames(self.h, n_samples))

    @staticmethod
    def frame_padding_mask(padding_mask: torch.Tensor, n_frames: int) -> torch.Tensor:
        """hubert.py:400-410: a frame is padding iff all samples of its chunk are."""
        extra = padding_mask.size(1) % n_frames
        if extra > 0:
            padding_mask = padding_mask[:, :-extra]
        return padding_mask.view(padding_mask.size(0), n_frames, -1).all(-1)

    @staticmethod
    def _suffix_frame_mask(padding_mask: torch.Tensor, n_frames: int):
        """``frame_padding_mask`` for the masks a collater produces (padding = a suffix of every row) without the
        [B][T][chunk] boolean reduction over millions of samples (20 - 50 ms per 24 x 8 s batch on the host: it made the
        host-fed --use-hubert step twice as long as the device-resident one): with n_b valid samples, frame f's chunk
        [f c, (f + 1) c) is all padding iff f c >= n_b.  None when some row's padding is not a suffix."""
        import numpy as np
        pm = padding_mask.detach().cpu().numpy()
        B, N = pm.shape
        chunk = (N - N % n_frames) // n_frames
        if chunk <= 0 or pm.dtype != np.bool_:
            return None
        n_pad = np.count_nonzero(pm, axis=1)
        first = np.where(n_pad > 0, pm.argmax(axis=1), N)
        if not np.array_equal(first + n_pad, np.full(B, N)):
            return None
        pad_from = -(-first // chunk)  # ceil(n_b / chunk)
        return torch.from_numpy(np.arange(n_frames)[None, :] >= pad_from[:, None])

    def stage(self, source: torch.Tensor, padding_mask: Optional[torch.Tensor] = None):
        """Host side of a call, done once per batch: upload the waveform, turn the sample-level padding mask into
        frame counts (hubert.py:400-410).  Returns ``(wave_dev, frame_lens_dev int32, frame_pad_mask_host, T)``;
        ``self.last_frame_lens`` holds the host copy of the frame counts."""
        wave = source.to(self.device, torch.float32).contiguous()
        bd.require_device(wave)
        B, N = wave.shape
        T = self.out_frames(N)
        if T <= 0:
            raise ValueError(f"{N} samples are shorter than the conv stack's receptive field")
        if padding_mask is None:
            padding_mask = torch.zeros(B, N, dtype=torch.bool)
        fpm = self._suffix_frame_mask(padding_mask, T)
        if fpm is None:  # (not a suffix mask at the sample level: the general reduction decides)
            fpm = self.frame_padding_mask(padding_mask.cpu(), T)
        if not bool(((~fpm).long().cumsum(1)[:, -1:] == (~fpm).long().sum(1, keepdim=True)).all()) or \
                bool((fpm[:, :-1] & ~fpm[:, 1:]).any()):
            raise ValueError("padding must be a suffix of every utterance")
        self.last_frame_lens = (~fpm).sum(1).long()  # host copy: the encoder's length / position bookkeeping
        lens = self._to_device_async(self.last_frame_lens.to(torch.int32))
        return wave, lens, fpm, T

    def forward_into(self, wave: torch.Tensor, lens: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """Device side: the frozen forward of staged inputs into ``out`` [B, T, embed] (no host work besides the
        enqueue: what a training step repeats on a prepared batch)."""
        B, N = wave.shape
        key = (B, N)
        if key not in self._plan:
            n = int(self.lib.s2st_hubert_workspace_floats(self.h, B, N))
            if n < 0:
                raise bd.S2STHipError(f"s2st_hubert_workspace_floats failed with code {n}")
            self._plan[key] = n
        need = self._plan[key]
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.float32, device=self.device)
        # frozen weights: the engine's bf16 copy stays valid while nobody wrote the parameter tensor (or a view of it).
        # torch's version counter sees in-place ops on the tensor and its views; writers that bypass it (``.data``, raw
        # pointers: a C-side load, broadcast_ on a .data view) call invalidate_bf16().  The cast of the last refresh ran on
        # ONE stream: a forward on another stream waits for that cast's event before it reads the copy.
        cur = torch.cuda.current_stream() if self.device.type == "cuda" else None
        if self.params_bf16 is not None and getattr(self, "_ph_version", None) == self.params._version:
            ev = getattr(self, "_ph_event", None)
            if ev is not None and cur is not None and getattr(self, "_ph_stream", None) != cur.cuda_stream:
                cur.wait_event(ev)
            self.lib.s2st_engine_bf16_is_fresh(self.h)
            refreshed = False
        else:
            refreshed = True
        self._ph_version = self.params._version
        bd.check(self.lib.s2st_hubert_forward(self.h, wave.data_ptr(), lens.data_ptr(), B, N, out.data_ptr(),
                                              self.workspace.data_ptr(), self.workspace.numel(),
                                              C.c_void_p(bd.stream_ptr())), "s2st_hubert_forward")
        if refreshed and cur is not None:  # the refresh cast was enqueued by this forward, on this stream
            self._ph_event = torch.cuda.Event()
            self._ph_event.record(cur)
            self._ph_stream = cur.cuda_stream
        self._keep = (wave, lens)
        return out

    def invalidate_bf16(self):
        """The parameters were written behind torch's version counter: the next forward refreshes the bf16 copy."""
        self._ph_version = None

    def reserve(self, B: int, N: int):
        """Size the workspace for a [B, N] waveform batch up front (no allocation inside a training loop)."""
        n = int(self.lib.s2st_hubert_workspace_floats(self.h, B, N))
        if n < 0:
            raise bd.S2STHipError(f"s2st_hubert_workspace_floats failed with code {n}")
        self._plan[(B, N)] = n
        if self.workspace is None or self.workspace.numel() < n:
            self.workspace = torch.empty(n, dtype=torch.float32, device=self.device)

    def extract_features(self, source: torch.Tensor, padding_mask: Optional[torch.Tensor] = None,
                         mask: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        if mask:
            raise NotImplementedError("the frozen front end runs with mask=False (s2st_transformer.py:248)")
        wave, lens, fpm, T = self.stage(source, padding_mask)
        out = torch.empty(wave.shape[0], T, self.embed, dtype=torch.float32, device=self.device)
        self.forward_into(wave, lens, out)
        return out, self._to_device_async(fpm)
