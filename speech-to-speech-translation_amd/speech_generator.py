"""AR mel generation on the HIP path: the counterpart of
``fairseq/speech_generator_for_s2st.py:35-134`` (AutoRegressiveSpeechGenerator) with the same
constructor arguments, stop rule, outputs and post-processing.  The decoder steps, the eval-mode
post-net, the alignment argmax, the global-CMVN de-normalisation and the vocoder all run in
libs2st_hip.so; this file is the reference's host loop (step counting, finished flags, slicing)."""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch

from .runtime import binding as bd


class PendingHypos(list):
    """Finalized hypotheses whose vocoder outputs are still being computed on the generator's second stream."""
    _events = ()

    def wait(self):
        """Make the current stream wait for the vocoder launches; afterwards the list is an ordinary result."""
        if self._events:
            cur = torch.cuda.current_stream()
            for ev in self._events:
                cur.wait_event(ev)
            for h in self:  # (everything a second stream produced is read on this one from now on)
                for w in h.values():
                    if torch.is_tensor(w) and w.is_cuda:
                        w.record_stream(cur)
            self._events = ()
        return self


class SpeechGenerator:
    def __init__(self, model, vocoder, data_cfg=None):
        self.model, self.vocoder = model, vocoder
        self.gcmvn_mean = self.gcmvn_std = None
        stats = None
        if data_cfg is not None:
            stats = getattr(data_cfg, "global_cmvn_stats", None) if not isinstance(data_cfg, dict) else data_cfg.get("global_cmvn_stats")
        if stats is not None:  # speech_generator_for_s2st.py:18-28: {"mean": [...], "std": [...]}
            self.gcmvn_mean = torch.as_tensor(stats["mean"], dtype=torch.float32)
            self.gcmvn_std = torch.as_tensor(stats["std"], dtype=torch.float32)

    def gcmvn_denormalize(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, T, C] -> x * std + mean (speech_generator_for_s2st.py:18-28)."""
        if self.gcmvn_mean is None:
            return x
        dev = x.device
        y = torch.empty_like(x)
        bd.call("s2st_affine_cols_f32", x.contiguous(), self.gcmvn_std.to(dev), self.gcmvn_mean.to(dev), y,
                x.shape[0] * x.shape[1], x.shape[2])
        return y

    def get_waveform(self, feat):
        return None if self.vocoder is None else self.vocoder(feat).squeeze(0)

    # ``generate(..., defer_vocoder=True)``: the batch's vocoder launches go to a SECOND stream and the call returns a
    # ``PendingHypos`` -- the same list, whose "waveform" tensors are complete only after ``.wait()``.  A driver that calls
    # ``generate`` for the next batch before it waits (generate_waveform.py, bench.py) gets the Griffin-Lim iterations of
    # batch k -- few large launches -- beside the decoding steps of batch k + 1 -- thousands of small dependent ones that
    # leave most of the chip idle.  Host-side order is unchanged (batch k's phase draws precede batch k + 1's), so are
    # the results.  Measured on the bench batch: 118.7 -> 111.7 ms per batch only -- the decoding steps under a running
    # Griffin-Lim kernel take 17 us instead of 8 (shared HBM / L2, not queue arbitration: a high-priority decode queue,
    # fewer Griffin-Lim workgroups per CU and shorter-lived ones changed nothing; profiles/r04_z_*).
    defer_vocoder = False

    def _vocode(self, feats):
        if not (self.defer_vocoder and self.vocoder is not None and feats and feats[0].is_cuda):
            return self.get_waveforms(feats), None
        vs = getattr(self.vocoder, "_defer_stream", None)  # (one per vocoder: generators that share it share the stream)
        if vs is None:
            from .runtime import streams
            vs = streams.get("vocoder", feats[0].device, may_share=("phase-upload",))  # (one per device: see runtime/streams.py)
            try:
                self.vocoder._defer_stream = vs
            except AttributeError:
                pass
        cur = torch.cuda.current_stream()
        vs.wait_stream(cur)
        with torch.cuda.stream(vs):
            waves = self.get_waveforms(feats)
            ev = torch.cuda.Event()
            ev.record(vs)
        for f in feats:
            f.record_stream(vs)
        return waves, ev

    def get_waveforms(self, feats):
        """All utterances of a batch through the vocoder together (same random-phase draws, in the same
        order, as calling get_waveform per utterance like speech_generator.py:81-94 does)."""
        if self.vocoder is None:
            return [None] * len(feats)
        if hasattr(self.vocoder, "batch"):
            return [w.squeeze(0) for w in self.vocoder.batch(feats)]
        return [self.get_waveform(f) for f in feats]


class AutoRegressiveSpeechGenerator(SpeechGenerator):
    def __init__(self, model, vocoder, data_cfg=None, max_iter: int = 6000, eos_prob_threshold: float = 0.5,
                 input_text: bool = False, seed: int = 1):
        super().__init__(model, vocoder, data_cfg)
        self.max_iter, self.eos_prob_threshold, self.seed = max_iter, eos_prob_threshold, seed
        self.input_text = bool(input_text)

    @torch.no_grad()
    def generate(self, model, sample, has_targ: bool = False, **kwargs) -> List[Dict[str, Optional[torch.Tensor]]]:
        model.eval()
        eng = model.engine
        ni = sample["net_input"]
        if self.input_text != bool(eng.cfg.text_input):
            # speech_generator_for_s2st.py:60-64 feeds token ids (--input-text true) or fbank frames to whatever encoder
            # the model has; the mismatched pairs fail inside the reference's first encoder op -- say so
            raise ValueError("--input-text true goes with a text-input model (--arch t2s_transformer), and only with one")
        if self.input_text:
            # speech_generator_for_s2st.py:60-64: the encoder reads sample["src_text"] / ["src_text_len"]
            src, src_lens = sample["src_text"], sample["src_text_len"]
        else:
            src, src_lens = model._front_end(ni.get("src_speech"), ni.get("src_speech_lens"),
                                             ni.get("collated_audios_orig"), ni.get("padding_mask"))
        bsz = src.shape[0]
        self._enc = eng.decode_begin(src, src_lens, self.max_iter, speaker=sample.get("speaker"))
        self.defer_vocoder = bool(kwargs.get("defer_vocoder", False))
        finalized = PendingHypos(dict() for _ in range(bsz))
        try:
            self._decode_mel(model, sample, bsz, finalized)
            if has_targ:
                self._add_targets(model, sample, bsz, finalized)
        finally:
            self.defer_vocoder = False
        return finalized

    @torch.no_grad()
    def generate_two(self, model, sample_a, sample_b, has_targ: bool = False, **kwargs):
        """``generate_many`` for two batches (round 4's entry point)."""
        return tuple(self.generate_many(model, [sample_a, sample_b], has_targ, **kwargs))

    @torch.no_grad()
    def generate_many(self, model, samples, has_targ: bool = False, **kwargs):
        """Several batches decoded AT ONCE (round 4: two; round 5: any number of chains): the decoding steps of one batch are
        thousands of small dependent launches that leave most of the chip idle, and the other batches' steps do not depend on
        them.  Batch k > 0 runs on its own engine object over the same weights (``Engine.inference_twins``: own caches,
        workspace, bf16 copies) and its own stream; the host alternates the step loops.  Results and their order are those
        of ``generate`` called on the batches one after the other: a batch's post-processing -- its vocoder's draws from
        numpy's phase stream -- is held until the batch before it has done its own, and its run-ahead phase stream starts
        from the state the earlier batch's draws REALLY leave numpy in (vocoder._HostMTStream).  Returns the hypothesis lists
        (``PendingHypos``) in batch order."""
        model.eval()
        samples = list(samples)
        n = len(samples)
        dev = model.engine.device
        if n == 1 or getattr(model, "hubert", None) is not None:
            # (a frozen front end whose one workspace the chains would share -- checked BEFORE the twin engines, each with its
            # own bf16 arena, are built: ADVICE r4)
            return [self.generate(model, s, has_targ, **kwargs) for s in samples]
        # round 6, S2ST_DECODE_MERGE=1: the batches as ONE merged batch on one chain (built, bit-equal to sequential decoding,
        # measured NEUTRAL on the GPU: 759 utterances/s against 768 - 862 for three chains -- a step's cost follows its rows, the
        # attention launches read every row's caches; profiles/r06_infer_merged_ab.txt -- and the only multi-batch form a CPU
        # (emulator) run can take).  Default: round 5's chains.
        if os.environ.get("S2ST_DECODE_MERGE", "0" if dev.type == "cuda" else "1") != "0" and not self.input_text and (
                not model.engine.cfg.precise or model.engine.cfg.prenet_dropout == 0.0) and all(
                s.get("speaker") is None for s in samples) and sum(int(s["net_input"]["src_speech"].shape[0]) for s in samples) <= 256:
            return self._generate_merged(model, samples, has_targ, **kwargs)
        if dev.type != "cuda":  # (no second stream)
            return [self.generate(model, s, has_targ, **kwargs) for s in samples]
        engs = [model.engine] + model.engine.inference_twins(n - 1)
        from .runtime import streams
        cs = [streams.get(f"decode-chain-{k}", dev) for k in range(1, n)]  # (one per chain and device: runtime/streams.py)
        cur = torch.cuda.current_stream()
        streams = [cur] + cs[:n - 1]
        for st in streams[1:]:
            st.wait_stream(cur)
        self.defer_vocoder = bool(kwargs.get("defer_vocoder", False))
        if self.vocoder is not None and hasattr(self.vocoder, "set_inflight"):
            self.vocoder.set_inflight(n)  # (that many run-ahead phase streams may be pending at once)
        runs = []
        try:
            for eng, st, sample in zip(engs, streams, samples):
                with torch.cuda.stream(st):
                    ni = sample["net_input"]
                    if self.input_text != bool(eng.cfg.text_input):
                        raise ValueError("--input-text true goes with a text-input model (--arch t2s_transformer), and only with one")
                    if self.input_text:
                        src, src_lens = sample["src_text"], sample["src_text_len"]
                    else:
                        src, src_lens = model._front_end(ni.get("src_speech"), ni.get("src_speech_lens"),
                                                         ni.get("collated_audios_orig"), ni.get("padding_mask"))
                    bsz = src.shape[0]
                    eng.decode_begin(src, src_lens, self.max_iter, speaker=sample.get("speaker"))
                    fin = PendingHypos(dict() for _ in range(bsz))
                    # [generator, stream, hypotheses, sample, batch size, alive, held at "post"]
                    runs.append([self._decode_mel_steps(model, sample, bsz, fin, eng), st, fin, sample, bsz, True, False])
            alive = n
            while alive:
                for k, r in enumerate(runs):
                    if not r[5]:
                        continue
                    if r[6]:
                        if runs[k - 1][5]:
                            continue  # this batch's post-processing (its phase draws) comes after the previous batch's
                        r[6] = False
                    with torch.cuda.stream(r[1]):
                        try:
                            if next(r[0]) == "post" and k > 0:
                                r[6] = True
                        except StopIteration:
                            r[5] = False
                            alive -= 1
                            if has_targ:
                                self._add_targets(model, r[3], r[4], r[2])
            # the caller's stream takes in what the other chains produced
            for r in runs[1:]:
                ev = torch.cuda.Event()
                ev.record(r[1])
                r[2]._events = tuple(r[2]._events) + (ev,)
        finally:
            self.defer_vocoder = False
        return [r[2] for r in runs]

    def _generate_merged(self, model, samples, has_targ: bool = False, **kwargs):
        """Round 6: the batches of ``generate_many`` as ONE merged batch on ONE chain (<= 256 utterances: the skinny
        projections' row limit).  A decoding step costs ~55 launches whether it carries 64 rows or 192, and the launches of
        several chains mostly wait for each other's CUs (profiles/r05_infer_graph.txt: three chains overlap 1.2 x), so the rows
        ride together instead.  The hypotheses are those of one batch after the other, bit for bit: every kernel of the decode
        treats rows independently; the always-on Prenet dropout draws, for a row of batch k, the mask its own batch would
        have drawn (``Engine.decode_row_map``); a batch's post-net / vocoder see the steps up to ITS last stop; the phase draws
        follow batch order.  Sources are padded to the longest batch (padded encoder frames are masked keys, as always)."""
        eng = model.engine
        dev = eng.device
        srcs, lens, sizes = [], [], []
        for s in samples:
            ni = s["net_input"]
            src, sl = model._front_end(ni.get("src_speech"), ni.get("src_speech_lens"), ni.get("collated_audios_orig"),
                                       ni.get("padding_mask"))
            srcs.append(src.to(dev, torch.float32))
            lens.append(sl.to(dev))
            sizes.append(int(src.shape[0]))
        smax = max(int(x.shape[1]) for x in srcs)
        srcs = [x if x.shape[1] == smax else torch.nn.functional.pad(x, (0, 0, 0, smax - x.shape[1])) for x in srcs]
        src_all, lens_all = torch.cat(srcs, 0).contiguous(), torch.cat(lens, 0)
        eng.decode_begin(src_all, lens_all, self.max_iter)
        eng.decode_row_map(torch.cat([torch.arange(nb, dtype=torch.int32) for nb in sizes]).to(dev))
        self.defer_vocoder = bool(kwargs.get("defer_vocoder", False))
        if self.vocoder is not None and hasattr(self.vocoder, "set_inflight"):
            self.vocoder.set_inflight(len(samples))
        fins, groups, r0 = [], [], 0
        for s, nb in zip(samples, sizes):
            fin = PendingHypos(dict() for _ in range(nb))
            fins.append(fin)
            groups.append((slice(r0, r0 + nb), nb, fin, s if has_targ else None))
            r0 += nb
        try:
            for _ in self._decode_mel_steps(model, None, r0, None, eng, groups=groups):
                pass
        finally:
            self.defer_vocoder = False
        return fins

    def _decode_mel(self, model, sample, bsz: int, finalized: List[Dict]) -> None:
        """The AR loop + post-processing of speech_generator_for_s2st.py:70-122 over the caches ``decode_begin`` filled."""
        for _ in self._decode_mel_steps(model, sample, bsz, finalized, model.engine):
            pass

    def _decode_mel_steps(self, model, sample, bsz: int, finalized: List[Dict], eng, groups=None):
        """``_decode_mel`` as a generator that yields after every enqueued decoding step: ``generate_two`` alternates two of
        them (two batches on two engines / streams), ``_decode_mel`` just runs one to its end."""
        c = eng.cfg
        n_frames_per_step = model.args.n_frames_per_step
        out_dim = c.out_dim
        raw_dim = out_dim // n_frames_per_step
        dev = eng.device
        if self.vocoder is not None and hasattr(self.vocoder, "prefetch_phases"):
            # the vocoder's random initial phases: numpy's generator starts running ahead while the GPU decodes
            # (groups: the batches of a merged decode, each with its own run-ahead stream, in batch order)
            # (with targets a batch's draws are followed by its targets' before the next batch's: no run-ahead past the first)
            for nb in ([bsz] if groups is None else [g[1] for g in groups[:1 if groups[0][3] is not None else len(groups)]]):
                self.vocoder.prefetch_phases(nb * self.max_iter * n_frames_per_step)
        # The loop of speech_generator_for_s2st.py:83-103 without a host round trip per step (round 4): the stop rule --
        # finished flags, out_lens, the next step's key lengths -- is a one-workgroup kernel behind every step
        # (s2st_decode_stop_update_i32), outputs land in whole-run buffers, and the host learns the number of finished
        # utterances LAG steps late through pinned memory: it may enqueue up to LAG steps past the stop, which are then
        # dropped (they only wrote rows nobody reads).  Same outputs as the step-by-step form (tests/test_inference.py).
        LAG = 4
        bufs = eng.decode_buffers(self.max_iter)
        on_gpu = dev.type == "cuda"
        # (a persistent pinned buffer: a fresh pinned allocation waits for the whole device, i.e. for a deferred vocoder)
        pinned = eng.__dict__.get("_pinned_done")  # (per engine: two chains decode at once in generate_two)
        if pinned is None or pinned.numel() < self.max_iter:
            pinned = eng._pinned_done = torch.zeros(self.max_iter, dtype=torch.int32, pin_memory=on_gpu)
        pinned = pinned[:self.max_iter]
        pinned.zero_()
        events = []
        n_steps = self.max_iter
        # Round 5, measured and left OFF (profiles/r05_infer_graph.txt): the step as ONE HIP-graph launch.  A run's ~55
        # kernels per step are enqueued one by one (15 k launches per batch of 64 utterances); the engine's replay form keeps
        # everything that changes from step to step in device memory, so a step captured once per run can be replayed
        # (runtime/engine.py decode_replay_prepare).  The host's share of a step drops from 0.21 to 0.03 ms -- and the GPU's
        # grows from 0.46 to 0.52 ms (the runtime's graph nodes cost more than stream launches between dependent kernels),
        # which is what bounds config 5: 640 - 670 utterances/s against 725 - 780.  S2ST_DECODE_GRAPH=1: the graph; =direct:
        # the replay form's calls without a graph (what runs on the CPU emulator); default 0: the step-by-step calls.  Same
        # outputs in all three, bit for bit (tests/test_inference.py).
        mode = os.environ.get("S2ST_DECODE_GRAPH", "0")
        replay = mode != "0" and eng.decode_replay_prepare(self.seed * 1000003, self.eos_prob_threshold, self.max_iter,
                                                            graph=(mode != "direct"))
        for step in range(self.max_iter):
            if replay:
                eng.decode_replay_step()
            else:
                eng.decode_step_into(step, self.seed * 1000003 + step, self.eos_prob_threshold, self.max_iter)
            pinned[step:step + 1].copy_(bufs["n_done"][step:step + 1], non_blocking=True)
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
            look = step - LAG if on_gpu else step
            if look >= 0:
                if on_gpu:
                    events[look].synchronize()
                if int(pinned[look]) == bsz:
                    n_steps = look + 1
                    break
            yield step
        else:
            if on_gpu:
                torch.cuda.current_stream().synchronize()
            done = (pinned == bsz).nonzero()
            if done.numel():
                n_steps = int(done[0]) + 1
        yield "post"  # (generate_two holds the second batch here until the first one's post-processing is done)
        if groups is None:
            self._post(model, eng, bufs, n_steps, None, bsz, finalized)
        else:
            # a merged decode: batch k's own loop would have ended at the step its LAST utterance stopped (the merged loop ran on
            # for the other batches; those extra rows are the dropped run-ahead of batch k)
            ol = bufs["out_lens"].cpu()
            for gi, (rows, nb, fin, targ_sample) in enumerate(groups):
                if gi > 0 and targ_sample is not None and self.vocoder is not None and hasattr(self.vocoder, "prefetch_phases"):
                    self.vocoder.prefetch_phases(nb * self.max_iter * n_frames_per_step)
                self._post(model, eng, bufs, min(n_steps, int(ol[rows].max())), rows, nb, fin)
                if targ_sample is not None:  # (phase draws in the order of one batch after the other: prediction, then target)
                    self._add_targets(model, targ_sample, nb, fin)

    def _post(self, model, eng, bufs, n_steps: int, rows, bsz: int, finalized: List[Dict]) -> None:
        """speech_generator_for_s2st.py:104-122 over the whole-run buffers of a decode: ``rows`` = the batch's rows of the
        buffers (None: all of them; a slice when several batches were decoded as ONE merged batch, ``_generate_merged``)."""
        n_frames_per_step = model.args.n_frames_per_step
        raw_dim = eng.cfg.out_dim // n_frames_per_step
        dev = eng.device
        sel = (lambda t, dim: t) if rows is None else (lambda t, dim: t.narrow(dim, rows.start, rows.stop - rows.start))
        # (the reference's out_lens after its loop: steps at which the utterances stopped, max_iter for those still running;
        # an utterance that would have stopped only in a dropped run-ahead step is still running at the real stop)
        out_lens = sel(bufs["out_lens"], 0).cpu().long()
        out_lens = torch.where(out_lens > n_steps, torch.full_like(out_lens, self.max_iter), out_lens)
        feat = sel(bufs["feat"][:n_steps], 1).transpose(0, 1).contiguous()          # [B, steps, out_dim]
        eos_prob = sel(bufs["eos"][:n_steps], 1).transpose(0, 1).contiguous()        # [B, steps]
        attn = sel(bufs["attn"][:n_steps], 1).permute(1, 2, 0).contiguous()          # [B, E, steps]
        feat = eng.postnet_eval(feat)  # postnet(feat) + feat, BatchNorm in eval mode
        alignment = torch.empty(bsz, attn.shape[2], dtype=torch.long, device=dev)
        bd.call("s2st_argmax_dim1_f32", attn, alignment, bsz, attn.shape[1], attn.shape[2])
        feat = self.gcmvn_denormalize(feat.reshape(bsz, -1, raw_dim))
        eos_prob = eos_prob.repeat_interleave(n_frames_per_step, dim=1)
        attn = attn.repeat_interleave(n_frames_per_step, dim=2)
        alignment = alignment.repeat_interleave(n_frames_per_step, dim=1)
        out_lens = out_lens * n_frames_per_step
        lens = out_lens.tolist()
        waves, ev = self._vocode([feat[b, :l] for b, l in enumerate(lens)])
        if ev is not None:
            finalized._events = tuple(finalized._events) + (ev,)
        for b, l in enumerate(lens):
            finalized[b].update({"feature": feat[b, :l], "eos_prob": eos_prob[b, :l], "attn": attn[b, :, :l],
                                 "alignment": alignment[b, :l], "waveform": waves[b]})

    def _add_targets(self, model, sample, bsz: int, finalized: List[Dict]) -> None:
        """speech_generator_for_s2st.py:124-133."""
        eng = model.engine
        n_frames_per_step = model.args.n_frames_per_step
        raw_dim = eng.cfg.out_dim // n_frames_per_step
        assert sample["tgt_speech"].size(-1) == eng.cfg.out_dim
        tgt = self.gcmvn_denormalize(sample["tgt_speech"].to(eng.device, torch.float32).reshape(bsz, -1, raw_dim))
        tl = (sample["target_lengths"] * n_frames_per_step).tolist()
        twaves, ev = self._vocode([tgt[b, :l] for b, l in enumerate(tl)])
        if ev is not None:
            finalized._events = tuple(finalized._events) + (ev,)
        for b, l in enumerate(tl):
            finalized[b]["targ_feature"] = tgt[b, :l]
            finalized[b]["targ_waveform"] = twaves[b]
