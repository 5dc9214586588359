#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the s2st_transformer training step (fwd + bwd +
gradient all-reduce + clip + Adam) on synthetic Fisher-shaped fbank80 -> mel80 batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): base 12 enc / 6 dec, d=512, n-frames-per-step 4, recipe
flags of run_baseline.sh (pre-LN, 1-layer d=64 aux ASR/ST decoders, dropout .1/.1/.01,
pre/post-net .5) plus CTC; batches packed by the reference's batch_by_size rule at
max-tokens=20000.  One "step" = one optimizer update on one batch per rank (update-freq 1);
rank r takes batches r, r+W, ... (weak scaling: per-GPU work fixed).  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "speech-to-speech-translation_amd"

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBPS = 8000.0          # HBM3E, same table


def algorithmic_macs(sample, a):
    """Forward MACs of the GEMM-shaped work on VALID (un-padded) tokens: the per-utterance
    formula of SURVEY.md section 8(d) (subsample, encoder layers + scores, prenet, decoder
    layers + cross K/V + scores (causal half), heads, post-net, CTC proj, aux decoders)."""
    ni = sample["net_input"]
    S = ni["src_speech_lens"].double()
    E = (((S - 1) / 2 + 1).floor() - 1).div(2).add(1).floor()
    D = sample["target_lengths"].double()
    C, F, Cd, Fd = a.encoder_embed_dim, a.encoder_ffn_embed_dim, a.decoder_embed_dim, a.decoder_ffn_embed_dim
    P, out = a.prenet_dim, a.output_frame_dim * a.n_frames_per_step
    k = 5
    hub = str(getattr(a, "use_hubert", "false")) == "true"
    hub_macs = 0.0
    if hub:
        # frozen hubert_base forward (1x, no backward), SURVEY 8(d): per utterance of N samples: conv stack over its own
        # frame counts, projection, grouped positional conv, 12 layers (+ scores) at T' frames; the encoder then sees
        # T' frames of 768 features (subsample conv 0 is 768 -> 1024)
        N = (ni["padding_mask"].shape[1] - ni["padding_mask"].long().sum(1)).double()
        n, cin = N, 1
        for (cd, ck, cs) in [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2:
            n = ((n - ck) / cs).floor() + 1
            hub_macs += float((n * cd * ck * cin).sum())
            cin = cd
        T = n
        hub_macs += float((T * (512 * 768 + 768 * 48 * 128 + 12 * (4 * 768 * 768 + 2 * 768 * 3072)) + 12 * 2 * 768 * T * T).sum())
        S = T
        E = (((S - 1) / 2 + 1).floor() - 1).div(2).add(1).floor()
    in_dim = a.hubert_hidden if hub else a.input_feat_per_channel
    m = (S / 2 * (in_dim * k * 1024) + S / 4 * (512 * k * 2 * C)).sum()
    m += (E * a.encoder_transformer_layers * (4 * C * C + 2 * C * F) + a.encoder_transformer_layers * 2 * E * E * C).sum()
    m += (D * (out * P + (a.prenet_layers - 1) * P * P + P * Cd)).sum()
    L = a.decoder_transformer_layers
    m += (D * L * (4 * Cd * Cd + 2 * Cd * Cd + 2 * Cd * Fd) + E * L * 2 * C * Cd + L * 2 * Cd * (D * D / 2 + D * E)).sum()
    m += (D * Cd * (out + 1)).sum()
    pc = a.postnet_conv_dim
    m += (D * k * (out * pc + (a.postnet_layers - 2) * pc * pc + pc * out)).sum()
    if a.ctc_weight > 0:
        m += (E * C * a.src_vocab_size).sum()
    for on, d, nl, lens, V, first in ((a.asr_ce_weight > 0, a.asr_decoder_embed_dim, a.asr_decoder_layers,
                                       sample.get("src_text_len"), a.src_vocab_size, True),
                                      (a.st_ce_weight > 0, a.st_decoder_embed_dim, a.st_decoder_layers,
                                       sample.get("tgt_text_len"), a.tgt_vocab_size, False)):
        if on:
            Lt = lens.double()
            m += (nl * (Lt * (4 * d * d + 2 * d * d + 2 * d * Fd) + E * 2 * C * d + 2 * d * (Lt * Lt / 2 + Lt * E))).sum()
            m += (Lt * ((512 * d if first else 0) + d * 512 + 512 * V)).sum()
    # (frozen front end: forward only -- returned as fwd-equivalent MACs of a 3x fwd+bwd count)
    return float(m) + hub_macs / 3.0


_T0 = time.perf_counter()


def vlog(*a):
    if os.environ.get("S2ST_BENCH_VERBOSE"):
        print(f"[bench +{time.perf_counter() - _T0:7.2f}s]", *a, file=sys.stderr, flush=True)


def effective_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def cpu_leg(args):
    """Child process (the only place bench.py touches oracle/): the oracle's train step (torch CPU fp32, all usable
    host cores) -- forward + backward + clip + Adam -- on the utterances of ONE timed batch of the GPU run:
    3 warm-up + up to 10 timed steps, median; stops early when the time budget is used up and
    says how many steps were timed.  With --use-hubert the frozen front end (HuBERT oracle) is inside the step."""
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import s2st_amd  # noqa: F401
    import s2st_oracle as O
    C = importlib.import_module(PKG + ".configs")
    D = importlib.import_module(PKG + ".data")
    ncores = effective_cores()
    torch.set_num_threads(ncores)
    O.USE_TORCH_CTC = True  # the reference calls torch.nn.CTCLoss (s2st_loss.py:173-176)
    pa = C.recipe_args(args.config)
    hub = str(pa.use_hubert) == "true"
    ca = O.make_args(**vars(pa))
    ca._hubert_input = hub
    corpus = D.SyntheticFisherCorpus(n_utts=args.n_utts, seed=1234, with_audio=hub)
    ids = [int(i) for i in args.cpu_leg.split(",")]
    sub = corpus.collate_batch(ids)
    front = lambda smp: smp  # noqa: E731
    if hub:
        import hubert_oracle as HO
        geo = HO.HUBERT_CONFIGS["base"]
        hstate = HO.synth_state(geo)

        def front(smp):
            ni = smp["net_input"]
            with torch.no_grad():
                f, fpm = HO.extract_features(hstate, geo, ni["collated_audios_orig"], ni["padding_mask"])
            out = dict(smp)
            out["net_input"] = dict(ni, src_speech=f, src_speech_lens=(~fpm).long().sum(-1))
            return out
    torch.manual_seed(1)
    m = O.S2STModel(ca)
    m.train()
    opt = O.FairseqAdam(m.parameters())
    t_all = time.perf_counter()
    times, n_warm = [], 0
    for u in range(3 + 10):
        t1 = time.perf_counter()
        O.train_step(m, opt, front(sub), u, 1.5e-3, 4000, 1.0)
        dt1 = time.perf_counter() - t1
        if u < 3:
            n_warm += 1
        else:
            times.append(dt1)
            if len(times) >= 3 and (time.perf_counter() - t_all) > args.cpu_seconds:
                break
    fr = ca.n_frames_per_step * sub["ntokens"]
    med = statistics.median(times)
    print(json.dumps({"value": round(fr / med, 1), "unit": "mel-frames/s", "cores": ncores, "kind": "port",
                      "sample": f"oracle (torch CPU fp32, {ncores} threads) fwd+bwd+clip+Adam"
                                f"{' incl. the frozen HuBERT-base forward' if hub else ''} on "
                                f"{'ALL' if args.cpu_whole else 'the first'} {len(ids)} utterances of timed batch 0 "
                                f"({fr} mel frames, src up to {int(corpus.src_n_frames[ids].max())} frames), "
                                f"{n_warm} warm-up + {len(times)} timed steps, median {med:.2f} s/step"}))


# ================================================================================================
# --config infer_base (BASELINE.json configs[4]): AR mel decode + Griffin-Lim (64 iterations) on Fisher-shaped inputs,
# base geometry, 1 GPU: utterances / s, and the MCD of the GPU path's waveforms against the CPU path's on the same inputs
# ================================================================================================
INFER_MAX_TOKENS = 100000  # the recipe's generate_waveform batches by --max-tokens 100000 (run_baseline.sh:143-147):
                           # utterances x longest source of a length-ordered batch; the 64 utterances below are ONE batch
                           # (rounds 1 - 3 decoded them 16 at a time: four times the sequential decoding steps)
INFER_N_UTTS = 64     # 64 utterances of the synthetic Fisher-shaped distribution
INFER_GL_ITERS = int(os.environ.get("S2ST_BENCH_GL_ITERS", "64"))   # --spec-bwd-max-iter 64 (the env override: a diagnosis aid, never the reported workload)


def infer_cpu_leg(args):
    """Child process (the only place bench.py touches oracle/): the oracle's AR generator + Griffin-Lim vocoder (torch CPU
    fp32, all usable host cores) on the first utterances of batch 0, weights and initial phases read from the files the
    parent wrote; prints its rate and leaves the waveforms for the parent's MCD."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import s2st_amd  # noqa: F401
    import s2st_oracle as O
    import infer_oracle as IO
    C = importlib.import_module(PKG + ".configs")
    D = importlib.import_module(PKG + ".data")
    ncores = effective_cores()
    torch.set_num_threads(ncores)
    job = torch.load(args.cpu_leg[len("infer:"):], weights_only=False)
    pa = C.recipe_args("base_recipe", prenet_dropout=0.0)
    m = O.S2STModel(O.make_args(**vars(pa)))
    missing = m.load_state_dict(job["state"], strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k and "_float_tensor" not in k and "version" not in k], missing
    m.eval()
    corpus = D.SyntheticFisherCorpus(n_utts=INFER_N_UTTS, seed=1234)
    sub = corpus.collate_batch(job["ids"])
    ni = sub["net_input"]
    t0 = time.perf_counter()
    with torch.no_grad():
        fin = IO.ar_generate(m, ni["src_speech"], ni["src_speech_lens"], job["max_iter"], 2.0, pa.n_frames_per_step)
    t_dec = time.perf_counter() - t0
    waves = []
    t0 = time.perf_counter()
    for f, ang in zip(fin, job["angles"]):
        waves.append(IO.vocoder(f["feature"], ang, n_iter=INFER_GL_ITERS, **job["voc"]).numpy())
    t_voc = time.perf_counter() - t0
    np.savez(job["out"], **{f"wave.{i}": w for i, w in enumerate(waves)},
             **{f"feature.{i}": f["feature"].numpy() for i, f in enumerate(fin)})
    n = len(job["ids"])
    print(json.dumps({"value": round(n / (t_dec + t_voc), 4), "unit": "utterances/s", "cores": ncores, "kind": "port",
                      "sample": f"oracle (torch CPU fp32, {ncores} threads): AR decode of {n} utterances x {job['max_iter']} "
                                f"steps ({t_dec:.1f} s, the decoder re-run on the prefix each step) + Griffin-Lim "
                                f"{INFER_GL_ITERS} iterations per utterance ({t_voc:.1f} s)"}))


def infer_main(args):
    import numpy as np
    if int(os.environ.get("WORLD_SIZE", 1)) != 1 or args.gpus != 1:
        raise SystemExit("--config infer_base is a 1-GPU workload (BASELINE.json configs[4]): utterances are independent, "
                         "N GPUs = N replicas of this run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import s2st_amd  # noqa: F401
    C_ = importlib.import_module(PKG + ".configs")
    tasks = importlib.import_module(PKG + ".tasks")
    G = importlib.import_module(PKG + ".speech_generator")
    V = importlib.import_module(PKG + ".vocoder")
    M = importlib.import_module(PKG + ".metrics")
    D = importlib.import_module(PKG + ".data")
    bd = importlib.import_module(PKG + ".runtime.binding")
    a = C_.recipe_args("base_recipe")
    task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
    torch.manual_seed(1)
    model = task.build_model(a)  # random-init weights of the architecture (no checkpoints on the box)
    voc_kw = dict(sample_rate=24000, win_size=1200, hop_size=300, n_fft=2048, n_mels=80, f_min=20, f_max=8000)
    voc = V.GriffinLimVocoder(spec_bwd_max_iter=INFER_GL_ITERS, device=dev, **voc_kw)
    corpus = D.SyntheticFisherCorpus(n_utts=INFER_N_UTTS, seed=1234)
    order = np.argsort(-corpus.src_n_frames, kind="stable")  # length-ordered batches, like the task's iterator
    groups = [g.tolist() for g in D.batch_by_size(order, corpus.src_n_frames[order], INFER_MAX_TOKENS, -1, 1)]
    vlog("batches", [len(g) for g in groups])
    samples, iters = [], []
    for ix in groups:
        s_ = corpus.collate_batch(ix)
        s_["net_input"]["collated_audios_orig"] = None
        s_["net_input"]["padding_mask"] = None
        samples.append(s_)
        iters.append(int(s_["target_lengths"].max()))  # teacher length of the batch: deterministic work
    gens = [G.AutoRegressiveSpeechGenerator(model, voc, None, max_iter=it, eos_prob_threshold=2.0) for it in iters]

    # As generate_waveform.py drives it: batch k's hypotheses are collected after batch k + 1 has been enqueued, its vocoder
    # launches on the generator's second stream -- Griffin-Lim of one batch beside the decoding steps of the next
    # (S2ST_DEFER_VOCODER=0: strictly one after the other).  Every timed step decodes AND vocodes one batch; the last
    # batch's vocoder is waited for inside the timed region.
    DEFER = os.environ.get("S2ST_DEFER_VOCODER", "1") != "0"

    # S2ST_DECODE_CHAINS=<n> (default 3): consecutive batches are decoded n at a time (generate_many: batch k > 0 on a twin
    # engine and its own stream, the step loops alternated) -- one batch's decoding steps leave most of the chip idle, the
    # next batches' do not depend on them.  Same hypotheses as one batch after the other (tests/test_inference.py).
    CHAINS = importlib.import_module(PKG + ".runtime.streams").default_decode_chains()

    def run(gen_list, n_steps, first=0):
        held, n_u, n_f = [], 0, 0

        def collect():
            nonlocal n_u, n_f
            for h in held:
                h.wait()
                n_u += len(h)
                n_f += sum(int(f["feature"].shape[0]) for f in h)
            held.clear()

        i = first
        while i < first + n_steps:
            k = i % len(samples)
            nc = min(CHAINS, first + n_steps - i)
            while nc >= 2 and not all(gen_list[(i + q) % len(samples)] is gen_list[k] for q in range(nc)):
                nc -= 1
            if nc >= 2:
                fins = list(gen_list[k].generate_many(model, [samples[(i + q) % len(samples)] for q in range(nc)], defer_vocoder=DEFER))
                i += nc
            else:
                fins = [gen_list[k].generate(model, samples[k], defer_vocoder=DEFER)]
                i += 1
            collect()
            held.extend(fins)
        collect()
        return n_u, n_f

    # every batch geometry once, and one full group of chains: workspace sizes, code objects, twin engines, the pool's
    # streams and the pinned phase-draw ring are created before the clock starts
    run(gens, max(args.warmup, len(samples), 2 * CHAINS))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_utt, n_frames = run(gens, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vlog("timed region", dt, "s for", n_utt, "utterances")
    # the same pass with the initial phases drawn by the device's generator instead of numpy's (GriffinLim(phase_rng="device"):
    # same distribution, no host-side generator run) -- reported next to `value`, never as it
    voc_d = V.GriffinLimVocoder(spec_bwd_max_iter=INFER_GL_ITERS, device=dev, phase_rng="device", **voc_kw)
    gens_d = [G.AutoRegressiveSpeechGenerator(model, voc_d, None, max_iter=it, eos_prob_threshold=2.0) for it in iters]
    run(gens_d, max(4, len(samples)))  # (its vocoder's buffers and second stream exist before the clock starts)
    torch.cuda.synchronize()
    t0d = time.perf_counter()
    nd, _ = run(gens_d, args.steps)
    torch.cuda.synchronize()
    value_device_rng = nd / (time.perf_counter() - t0d)
    # ---- the same pass with EARLY STOPS: the headline's stop threshold is never reached (fixed work, SURVEY
    #      8(d)), which also means no utterance ever leaves the batch early and every run-ahead phase draw is used -- not what
    #      a trained checkpoint does.  Random-init weights give every (utterance, step) some stop probability: the threshold
    #      is set to the median over the utterances of their largest probability in the first 70 % of the steps, so about
    #      half of them stop somewhere in there and the rest later or never; the batch ends when its last utterance has.
    with torch.no_grad():
        probe = G.AutoRegressiveSpeechGenerator(model, None, None, max_iter=iters[0], eos_prob_threshold=2.0).generate(model, samples[0])
    nf = a.n_frames_per_step
    peak = torch.stack([f["eos_prob"][::nf][: max(1, int(0.7 * iters[0]))].max() for f in probe])
    thr_e = float(peak.median())
    gens_e = [G.AutoRegressiveSpeechGenerator(model, voc, None, max_iter=it, eos_prob_threshold=thr_e) for it in iters]
    run(gens_e, max(2, len(samples)))
    torch.cuda.synchronize()
    t0e = time.perf_counter()
    ne, nfe = run(gens_e, args.steps)
    torch.cuda.synchronize()
    dte = time.perf_counter() - t0e
    early = {"threshold": round(thr_e, 5), "utterances_per_s": round(ne / dte, 2), "mel_frames_per_s": round(nfe / dte, 1),
             "mean_decoded_fraction_of_the_fixed_work": round(nfe / max(n_frames, 1), 3),
             "note": "same batches, weights, chains and vocoder overlap as `value`, stop threshold = median over utterances of "
                     "their peak stop probability in the first 70 % of the steps (random-init weights): utterances leave at "
                     "different steps, Griffin-Lim runs on what was decoded"}
    vlog("early-stop leg", early)
    # ... and strictly one after the other (the vocoder on the decoder's stream), for the record
    t0s = time.perf_counter()
    for i in range(args.steps):
        gens[i % len(samples)].generate(model, samples[i % len(samples)])
    torch.cuda.synchronize()
    value_serial = args.steps * len(samples[0]["id"]) / (time.perf_counter() - t0s) if len(samples) == 1 else None
    # decode / vocoder split of one batch (un-timed above)
    tA = time.perf_counter()
    g0 = G.AutoRegressiveSpeechGenerator(model, None, None, max_iter=iters[0], eos_prob_threshold=2.0)
    f0 = g0.generate(model, samples[0])
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - tA
    tA = time.perf_counter()
    voc.batch([f["feature"] for f in f0])
    torch.cuda.synchronize()
    t_voc = time.perf_counter() - tA

    # ---- roofline: replay one pass over the batches with per-dispatch timing; the kernel with the largest share --------
    roofline = None
    if not args.no_roofline:
        import ctypes as C
        lib = bd.lib()
        lib.s2st_profile_enable.argtypes = [C.c_int32]
        lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
        lib.s2st_profile_report.restype = C.c_int64
        lib.s2st_profile_enable(1)
        for i in range(len(samples)):  # (one stream: the per-dispatch events belong to one queue)
            gens[i].generate(model, samples[i])
        torch.cuda.synchronize()
        lib.s2st_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        n = lib.s2st_profile_report(buf, len(buf))
        rows = {}
        for ln in buf.value.decode().splitlines() if n > 0 else []:
            tag, cnt, us, w1, w2 = ln.split("\t")
            rows[tag] = dict(n=int(cnt), us=float(us), work=float(w1), work2=float(w2))
        if os.environ.get("S2ST_BENCH_VERBOSE"):
            for tag, r in sorted(rows.items(), key=lambda kv: -kv[1]["us"])[:25]:
                vlog("  %-60s launches %7d  avg %7.2f us  total ms %8.3f" % (tag, r["n"], r["us"] / r["n"], r["us"] * 1e-3))
        if rows:
            tot_us = sum(r["us"] for r in rows.values())
            dom_tag, dom = max(rows.items(), key=lambda kv: kv[1]["us"])
            avg_us = dom["us"] / dom["n"]
            # (the skinny-M decode GEMM streams its weights once per launch: HBM / L2-bound, its work figure is bytes)
            is_mfma = dom_tag.startswith(("gemm", "flash_")) and dom["work"] > 0 and dom_tag != "gemm_skinny_kernel"
            if is_mfma:
                ach = dom["work"] / dom["n"] / (avg_us * 1e-6) / 1e12
                roofline = {"bound": "mfma", "kernel": dom_tag, "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 5), "traffic": None,
                            "gflop_per_launch": round(dom["work"] / dom["n"] / 1e9, 3)}
            else:
                ach = dom["work"] / dom["n"] / (avg_us * 1e-6) / 1e9 if dom["work"] > 0 else None
                roofline = {"bound": "hbm", "kernel": dom_tag, "achieved": None if ach is None else round(ach, 1),
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBPS, 5),
                            "traffic": None}
            roofline.update({"avg_launch_us": round(avg_us, 2), "launches_per_utterance": round(dom["n"] / INFER_N_UTTS, 1),
                             "share_of_kernel_time": round(dom["us"] / tot_us, 4),
                             "kernel_ms_per_utterance": round(tot_us * 1e-3 / INFER_N_UTTS, 3),
                             "launches_per_utterance_all_kernels": round(sum(r["n"] for r in rows.values()) / INFER_N_UTTS, 1),
                             "timing": "start/stop events attached to each dispatch, one pass over the %d batches" % len(samples)})

    # ---- CPU leg + MCD of the GPU path against the CPU path on the same inputs (Prenet dropout 0 on both sides: the
    #      reference's always-on Prenet dropout makes the decode a random variable; the timed workload above keeps it on) --
    cpu, mcd = None, None
    if args.cpu_seconds > 0:
        import subprocess
        import tempfile
        n_cpu = args.cpu_utts or 8
        # utterances around the corpus' median length (the CPU oracle re-runs its decoder on the whole prefix every step:
        # ~1 s per utterance here; `--cpu-utts 64` = all of them, the run committed under profiles/)
        mid = len(order) // 2
        ids = order[max(0, mid - n_cpu // 2): max(0, mid - n_cpu // 2) + n_cpu].tolist() if n_cpu < len(order) else order.tolist()
        sub = corpus.collate_batch(ids)
        sub["net_input"]["collated_audios_orig"] = None
        sub["net_input"]["padding_mask"] = None
        mi = int(sub["target_lengths"].max())
        a0 = C_.recipe_args("base_recipe", prenet_dropout=0.0)
        m0 = tasks.S2ST_TranslationTask.setup_task(a0, device=dev).build_model(a0)
        m0.load_state_dict(model.state_dict(), strict=True)
        rs = np.random.RandomState(5)
        F_ = voc_kw["n_fft"] // 2 + 1
        # (the reference's initial phases, vocoder.py:101-102, here from a seeded generator and shared with the CPU leg)
        angles = [np.angle(np.exp(2j * np.pi * rs.rand(F_, mi * a0.n_frames_per_step))).astype(np.float32) for _ in ids]
        g_ = G.AutoRegressiveSpeechGenerator(m0, None, None, max_iter=mi, eos_prob_threshold=2.0)
        fin0 = g_.generate(m0, sub)
        waves_gpu = voc.batch([f["feature"] for f in fin0], angles)
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory() as td:
            job = {"state": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "ids": ids, "max_iter": mi,
                   "angles": angles, "voc": voc_kw, "out": os.path.join(td, "cpu_out.npz")}
            jp = os.path.join(td, "job.pt")
            torch.save(job, jp)
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", "infer:" + jp],
                                   capture_output=True, text=True, timeout=10 * args.cpu_seconds + 300)
                for ln in r.stdout.splitlines():
                    if ln.startswith("{"):
                        cpu = json.loads(ln)
                if cpu is None:
                    vlog("cpu leg produced no result:", r.stderr[-600:])
                else:
                    z = np.load(job["out"])
                    y_cpu = [torch.from_numpy(z[f"wave.{i}"]).to(dev) for i in range(len(ids))]
                    y_gpu = [w.reshape(-1).float() for w in waves_gpu]
                    res = M.batch_mel_cepstral_distortion(y_gpu, y_cpu, voc_kw["sample_rate"], "path", device=dev)
                    ferr = max(float((f["feature"].cpu() - torch.from_numpy(z[f"feature.{i}"])).abs().max())
                               for i, f in enumerate(fin0))
                    mcd = {"mcd_gpu_vs_cpu": round(float(sum(float(d) for d, _ in res) / len(res)), 4),
                           "utterances": len(ids), "max_abs_feature_diff": round(ferr, 5),
                           "note": "MFCC distance along the DTW path between the GPU path's waveform (bf16 operands) and the "
                                   "CPU oracle's (fp32) for the same weights, inputs, Prenet dropout 0 and initial phases; "
                                   "0 = identical.  Both sides use this repository's restatements of two third-party tables "
                                   "that are absent from the image (librosa's Slaney mel filters, torchaudio's MFCC): the "
                                   "figure compares the two paths with each other, its absolute scale is unpinned"}
            except subprocess.TimeoutExpired:
                vlog("cpu leg timed out")

    value = n_utt / dt
    line = {"metric": "utterances/sec (AR mel decode + Griffin-Lim %d it) on Fisher-shaped inputs" % INFER_GL_ITERS,
            "value": round(value, 2), "unit": "utterances/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "generate_waveform-shaped inference: s2st_transformer base 12enc/6dec d512 nfps4 (random-init), "
                                   "%d Fisher-shaped utterances in length-ordered --max-tokens %d batches (%s utterances), key/value-cached "
                                   "AR decode for the batch's longest teacher length (stop threshold never reached: fixed work), "
                                   "Prenet dropout 0.5 on, post-net, Griffin-Lim %d iterations (n_fft 2048, hop 300: real FFTs in "
                                   "LDS) batched over the utterances" % (INFER_N_UTTS, INFER_MAX_TOKENS,
                                                                        "+".join(str(len(g)) for g in groups), INFER_GL_ITERS),
                       "name": "infer_base", "utterances_per_step": len(groups[0]), "mel_frames_per_s": round(n_frames / dt, 1),
                       "initial_phases": "numpy global generator on the host (the reference's draws, vocoder.py:101-102)",
                       "value_with_device_phase_rng": round(value_device_rng, 2),
                       "vocoder_overlap": ("batch k's Griffin-Lim on a second stream beside batch k + 1's decoding steps"
                                           if DEFER else "off"),
                       "decode_chains": CHAINS, "early_stop": early,
                       # roles of the package's stream pool for which no free hardware queue was found (runtime/streams.py)
                       "stream_collisions": importlib.import_module(PKG + ".runtime.streams").collisions(),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)"),
                       "value_without_vocoder_overlap": round(value_serial, 2) if value_serial else None,
                       "decode_steps_per_batch": iters, "batch0_decode_ms": round(t_dec * 1e3, 2),
                       "batch0_vocoder_alone_ms": round(t_voc * 1e3, 2),
                       "batch0_note": "vocoder alone = called by itself, its phase draws NOT run ahead under the decode as they "
                                      "are inside generate() (vocoder.GriffinLim.prefetch_phases)"}}
    if roofline:
        line["roofline"] = roofline
    if cpu:
        line["cpu_baseline"] = cpu
        line["config"]["x_over_cpu"] = round(value / cpu["value"], 1)
    if mcd:
        line["mcd"] = mcd
    print(json.dumps(line))


# The optimizer update runs in chunks on the engine's second stream and the next forward waits chunk by chunk (the same
# parameter trajectory, bit for bit: tests/test_full_size.py::test_overlapped_optimizer_update_gives_the_same_trajectory).
# Round 3 measured it neutral; with the nontemporal optimizer kernel of round 6 it is worth 0.02 - 0.11 ms per step on two
# boxes (profiles/r06_adam_overlap_ab.txt) and is the default of bench.py and train.py; S2ST_ADAM_OVERLAP=0 switches it off.
ADAM_OVERLAP = os.environ.get("S2ST_ADAM_OVERLAP", "1") != "0"


def spawn_ranks(args) -> None:
    """One rank per GPU through `python -m torch.distributed.run` (RCCL rendezvous on 127.0.0.1, a free port), with this
    command line.  Refuses -- non-zero exit, no JSON line -- when the node shows fewer than N devices, unless
    S2ST_BENCH_SHARE_GPU=1 asks for the one-GPU control-flow check (all ranks on cuda:0, gloo carries the bytes)."""
    import socket
    n = args.gpus
    share = os.environ.get("S2ST_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()  # (counts devices without initialising the runtime in this process)
    if have < n and not share:
        raise SystemExit(f"[bench] --gpus {n} but only {have} HIP device(s) are visible: refusing to report an N={n} line "
                         f"(S2ST_BENCH_SHARE_GPU=1 runs the {n}-rank code path on one device as a control-flow check)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or len(lines) != 1:
        sys.stdout.write(r.stdout)
        raise SystemExit(f"[bench] the {n}-rank run failed (exit code {r.returncode}, {len(lines)} result lines)")
    line = json.loads(lines[0])
    if line.get("n_gpus") != n or line.get("n_ranks_seen") != n:
        raise SystemExit(f"[bench] asked for {n} ranks, the run reports n_gpus={line.get('n_gpus')} / "
                         f"n_ranks_seen={line.get('n_ranks_seen')}")
    print(lines[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # ~1 s of timed work: long enough for an external utilisation sampler
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="base_recipe")
    ap.add_argument("--max-tokens", type=int, default=20000)
    ap.add_argument("--n-utts", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="time budget of the CPU-baseline leg (0 = skip)")
    ap.add_argument("--cpu-utts", type=int, default=0, help="CPU leg: first N utterances of timed batch 0 (0 = the "
                    "whole batch; default 8 for the HuBERT configuration, whose CPU front end is ~10x the model)")
    ap.add_argument("--cpu-whole", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs that put the other single-GPU BASELINE workloads (infer_base = configs[4], "
                         "base_recipe_hubert = configs[3] on one GPU) into config.other_configs of the default line")
    ap.add_argument("--grad-exchange-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: type the gradient ranges are all-reduced in (runtime/distributed.py; fp32 = the reference's)")
    ap.add_argument("--exchange-proxy", default=None, metavar="WGS,RANKS,GBPS",
                    help="ONE GPU only: per gradient bucket, launch the library's stand-in for the all-reduce's kernels on the "
                         "gradient-exchange stream (that many workgroups moving the bucket's 2 (N - 1) / N share at that pace): "
                         "what RCCL's kernels beside the backward would cost the step.  Reported under `exchange_proxy`, and "
                         "the line's `value` is then NOT the headline (config.name gets a suffix)")
    ap.add_argument("--timeline", default=None, help="write the per-dispatch timeline (stream, start, duration on the GPU "
                    "clock) of ONE replayed step to this file; tools/timeline.py summarises it")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_leg:
        return infer_cpu_leg(args) if args.cpu_leg.startswith("infer:") else cpu_leg(args)
    if args.config == "infer_base":
        if args.steps == 100:
            args.steps = 4  # (default: four passes over the corpus' batch(es))
        return infer_main(args)

    if args.exchange_proxy:
        if args.gpus != 1 or "WORLD_SIZE" in os.environ:
            raise SystemExit("--exchange-proxy is a ONE-GPU experiment (the stand-in for a collective that cannot run here)")
        os.environ["S2ST_EXCHANGE_PROXY"] = args.exchange_proxy  # read by runtime/distributed.GradReducer
        args.no_other_configs = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, as fresh child processes BEFORE this
        # process touches the GPU (the reference spawns its ranks itself too: fairseq/distributed/utils.py:334-369
        # `torch.multiprocessing.spawn` from `call_main`); the parent only relays the JSON line and the exit code.
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        # never an N = 1 line under an N = 8 label (or the reverse)
        raise SystemExit(f"[bench] WORLD_SIZE={world} contradicts --gpus {args.gpus}: launch with `--nproc-per-node "
                         f"{args.gpus}` (or run `python bench.py --gpus {args.gpus}` without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    # S2ST_BENCH_SHARE_GPU=1 (a check of the N > 1 code path on a one-GPU box, never a measurement): every rank drives
    # cuda:0 and gloo carries the gradient bytes through pinned host memory -- RCCL refuses two ranks on one device
    share = world > 1 and os.environ.get("S2ST_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

    import s2st_amd  # noqa: F401
    C_ = importlib.import_module(PKG + ".configs")
    tasks = importlib.import_module(PKG + ".tasks")
    trainer_mod = importlib.import_module(PKG + ".trainer")
    bd = importlib.import_module(PKG + ".runtime.binding")
    prefetch = importlib.import_module(PKG + ".runtime.prefetch")

    a = C_.recipe_args(args.config)  # named configurations live in the package (configs.py)
    a.grad_exchange_dtype = args.grad_exchange_dtype
    hub = str(a.use_hubert) == "true"
    task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
    torch.manual_seed(1)
    model = task.build_model(a)
    if hub:  # hubert_base_ls960.pt is not on the box: random-init weights of that architecture
        g_ = torch.Generator(device="cpu").manual_seed(2)
        model.hubert.params.copy_(torch.randn(model.hubert.n_params, generator=g_) * 0.02)
        for n_, off_, num_, shp_ in model.hubert.infos:
            if len(shp_) == 1 and n_.endswith(".weight"):
                model.hubert.params[off_:off_ + num_].fill_(1.0)  # norm gains
    criterion = task.build_criterion(a)
    trainer = trainer_mod.Trainer(a, task, model, criterion)
    eng = model.engine
    vlog('model built', eng.n_params, 'params')

    corpus = task.load_dataset("train", n_utts=args.n_utts, seed=1234, with_audio=hub)
    batches = corpus.batches(max_tokens=args.max_tokens, bsz_mult=8)
    # deterministic shuffle of the (length-sorted) batches, then deal round-robin to ranks
    import numpy as np
    order = np.random.RandomState(7).permutation(len(batches))
    need = (args.steps + args.warmup) + (1 if hub else 0)  # (+1: the batch whose front end the last timed step launches ahead)
    mine = [batches[order[(i * world + rank) % len(batches)]] for i in range(need)]
    samples = [corpus.collate_batch(ix) for ix in mine]
    if hub:
        for s_ in samples:
            s_["net_input"]["src_speech"] = None  # HuBERT mode: the collater hands over audio, no fbank tensor
    # device-resident inputs (features / staged waveforms, length and position vectors)
    prepared = [model.prepare_sample(s_, training=True) for s_ in samples]
    frames = [a.n_frames_per_step * s["ntokens"] for s in samples]
    macs = [algorithmic_macs(s, a) for s in samples]
    vlog('batches prepared', [(len(ix), int(corpus.src_n_frames[ix].max())) for ix in mine])

    # size workspace / output pool for the largest batch geometry up front (what a max-tokens data
    # loader knows): no device allocation inside the loop
    trainer.engine.reserve(prepared)
    if hub:
        wb = max((p_.hubert_io[0].shape for p_ in prepared), key=lambda sh: sh[0] * sh[1])
        for p_ in prepared:
            model.hubert.reserve(*p_.hubert_io[0].shape)
        vlog('largest waveform batch', tuple(wb))

    def step(i):
        # (the update overlaps the next step's forward, see ADAM_OVERLAP above; the last timed step's update is inside the
        #  timed region: the closing synchronize waits for every stream)
        if hub and i + 1 < len(prepared):
            # frozen HuBERT of the NEXT batch beside this step (it does not depend on the update): every timed step launches
            # exactly one front-end forward, as before -- for the batch after it instead of its own
            model.front_end_ahead(prepared[i + 1])
        return trainer.train_step([prepared[i]], overlap_optimizer=ADAM_OVERLAP)

    for i in range(args.warmup):
        step(i)
        if os.environ.get('S2ST_BENCH_VERBOSE'):
            torch.cuda.synchronize()
            vlog('warmup step', i, 'done')
    torch.cuda.synchronize()
    if os.environ.get('S2ST_BENCH_VERBOSE') and args.warmup > 0 and not hub:
        # host cost of enqueueing ONE step into an empty queue (no back-pressure) vs its GPU time
        th = time.perf_counter()
        step(args.warmup - 1)
        th1 = time.perf_counter() - th
        torch.cuda.synchronize()
        vlog('single step: host enqueue %.2f ms, until GPU done %.2f ms' % (th1 * 1e3, (time.perf_counter() - th) * 1e3))
        # forward / backward / optimizer split of one step on the caller's stream (un-profiled)
        eng_ = trainer.engine
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        eng_.zero_grad()
        ev[0].record()
        eng_.forward(prepared[args.warmup - 1], training=True, with_loss=True)
        ev[1].record()
        eng_.backward(1.0)
        ev[2].record()
        step(args.warmup - 1)
        ev[3].record()
        torch.cuda.synchronize()
        vlog('GPU time on the data-path stream: forward %.2f ms, backward %.2f ms, (next full step %.2f ms)' % (
            ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))
    if world > 1:
        torch.distributed.barrier()
    if (world > 1 or args.exchange_proxy) and trainer.reducer is not None and trainer.reducer.cuda:
        trainer.reducer.exposed_ms()  # (drop what the warm-up recorded)
        trainer.reducer.measure_exposed = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ev = []
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
        # per-step GPU time (one event record per step end: no synchronisation, not part of the metric) -- the straggler
        # statistic below is simulated from these
        e_ = torch.cuda.Event(enable_timing=True)
        e_.record()
        step_ev.append(e_)
    t_issue = time.perf_counter() - t0  # host time to enqueue the steps (GPU-bound if << dt)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    my_dt_local = dt
    vlog('timed region', dt, 'host issue time', t_issue)
    last_loss = float(trainer.criterion.last_outputs["stats"][16])  # (of the last TIMED step: the legs below run more)
    # N > 1: what the gradient exchange cost beyond what the backward hid (the compute stream's wait in
    # GradReducer.finish(), GPU clock), and the ranges it was issued in
    exchange = None
    proxy = None
    if args.exchange_proxy and trainer.reducer is not None and trainer.reducer.cuda:
        trainer.reducer.measure_exposed = False
        ex = trainer.reducer.exposed_ms()
        bk = trainer.reducer.last_buckets
        w_, n_, g_ = trainer.reducer.proxy
        proxy = {"workgroups": w_, "ranks_modelled": n_, "pace_gbps": g_,
                 "buckets_mib": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi in bk],
                 "bytes_moved_per_update": int(sum(hi - lo for lo, hi in bk) * 4 * 2 * (n_ - 1) / n_),
                 "exposed_ms": round(sum(ex) / max(len(ex), 1), 4), "exposed_ms_max": round(max(ex) if ex else 0.0, 4),
                 "note": "no collective ran: per bucket, a stand-in kernel (s2st_exchange_proxy_f32) on the gradient-exchange "
                         "stream read + wrote the bucket's 2 (N - 1) / N share on that many workgroups at that pace"}
    if world > 1 and trainer.reducer is not None and trainer.reducer.cuda:
        trainer.reducer.measure_exposed = False
        ex = trainer.reducer.exposed_ms()
        bk = trainer.reducer.last_buckets
        exchange = {"allreduce_exposed_ms": round(sum(ex) / max(len(ex), 1), 4),
                    "allreduce_exposed_ms_max": round(max(ex) if ex else 0.0, 4),
                    "buckets_mib": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi in bk],
                    "bytes_per_update": int(sum(hi - lo for lo, hi in bk) * (2 if trainer.reducer.exchange_dtype == "bf16" else 4)),
                    "dtype": "bf16" if trainer.reducer.exchange_dtype == "bf16" else "f32",
                    # RCCL's channel / CU budget and algorithm choices as this run saw them (unset = RCCL's defaults): the
                    # collective's kernels share the chip with the backward, so the first 8-GPU run should explain itself
                    "rccl_env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO",
                                                            "NCCL_NCHANNELS_PER_PEER", "RCCL_MSCCL_ENABLE", "NCCL_P2P_LEVEL",
                                                            "HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES") if k in os.environ},
                    "transport": ("gloo through pinned host memory (S2ST_BENCH_SHARE_GPU: a control-flow check, not a "
                                  "measurement)" if share else
                                  ("RCCL via the C ABI (s2st_allreduce_sum_f32)" if trainer.reducer.native is not None
                                   else "RCCL via torch.distributed (backend nccl)"))}
    if os.environ.get('S2ST_STALL_TRACE'):
        import ctypes as C_
        fn = eng.lib.s2st_engine_stall_report
        fn.restype, fn.argtypes = C_.c_int64, [C_.c_void_p, C_.c_int32]
        step(args.warmup)
        torch.cuda.synchronize()
        fn(eng.h, 0)
        step(args.warmup + 1)
        torch.cuda.synchronize()
        vlog('cross-stream waits of one step on the data-path stream: %d us in total' % fn(eng.h, 1))
    # PCIe-inclusive rate: the same steps fed from HOST batches (the collater's output), i.e. with the feature upload and the
    # per-batch index preparation of Engine.prepare inside the timed region -- through runtime/prefetch.DevicePrefetcher
    # (background thread, own stream, 3 batches ahead: what fairseq's BufferedIterator + pinned DataLoader do for the
    # reference).  Steady state: the clock starts once the prefetcher's queue is full.  Reported in `config`
    # (host_fed_ms_per_step), never as `value`.
    host_fed = None
    if not args.no_host_fed and world == 1:
        n_h = args.steps
        feed = [samples[i] for i in range(args.warmup, args.warmup + n_h)]
        pf = prefetch.DevicePrefetcher(feed, eng, depth=3, model=model if hub else None)
        t_fill = time.perf_counter()
        while not pf.q.full() and time.perf_counter() - t_fill < 5.0:
            time.sleep(0.001)
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        for smp in pf:
            trainer.train_step([smp], overlap_optimizer=ADAM_OVERLAP)
        torch.cuda.synchronize()
        th = time.perf_counter() - th0
        host_fed = th / n_h * 1e3
        vlog('host-fed (PCIe-inclusive), prefetched uploads: %.3f ms/step, %.0f mel-frames/s over %d steps' % (
            host_fed, sum(frames[args.warmup:args.warmup + n_h]) / th, n_h))
    if os.environ.get('S2ST_BENCH_VERBOSE'):
        n_h = min(args.steps, 10)
        feed = [samples[i] for i in range(args.warmup, args.warmup + n_h)]
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        for smp in feed:
            trainer.train_step([smp], overlap_optimizer=ADAM_OVERLAP)
        torch.cuda.synchronize()
        th = time.perf_counter() - th0
        vlog('host-fed (PCIe-inclusive), in-line uploads: %.3f ms/step over %d steps' % (th / n_h * 1e3, n_h))
        # host cost of preparing one batch (Engine.prepare: index vectors + uploads; --use-hubert: waveform staging + frame
        # mask too), main thread, idle GPU
        tp0 = time.perf_counter()
        for smp in feed:
            model.prepare_sample(smp, training=True) if hub else eng.prepare(smp, training=True, seed=0)
        torch.cuda.synchronize()
        vlog('batch preparation alone: %.3f ms per batch' % ((time.perf_counter() - tp0) / n_h * 1e3))
    step_ms = [step_ev[j - 1].elapsed_time(step_ev[j]) for j in range(1, len(step_ev))]
    if step_ms:
        vlog('per-step GPU ms (mel frames):', ' '.join('%.2f(%d)' % (step_ms[j - 1], frames[args.warmup + j])
                                                     for j in range(1, len(step_ev))))
    # how many ranks the communicator really joined (a SUM of ones over it), reported next to n_gpus
    n_ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if share else dev)
        torch.distributed.all_reduce(ones, op=torch.distributed.ReduceOp.SUM)
        n_ranks_seen = int(round(float(ones[0])))
    my_frames = float(sum(frames[args.warmup:args.warmup + args.steps]))
    my_flops = 3.0 * 2.0 * sum(macs[args.warmup:args.warmup + args.steps])  # fwd + bwd = 3 x fwd, 2 FLOP per MAC
    stat = torch.tensor([dt, my_frames, my_flops], dtype=torch.float64, device="cpu" if share else dev)
    if world > 1:
        tmax = stat[:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tot = stat[1:].clone()
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
        dt, total_frames, total_flops = float(tmax[0]), float(tot[0]), float(tot[1])
    else:
        total_frames, total_flops = my_frames, my_flops

    # ---- roofline leg: replay the timed steps with per-DISPATCH timing (include/s2st_hip.h s2st_profile_*) ------
    # Every launch of the dominant kernels carries its own start / stop event (hipExtLaunchKernelGGL): the elapsed
    # time of a pair is that dispatch's begin -> end on the stream it ran on -- the duration a rocprofv3 kernel trace
    # of this command reports (profiles/r02_*_kernel_stats.txt) -- so per-kernel sums cannot exceed the step.
    roofline = None
    n_replay = args.steps
    if not args.no_roofline and world > 1 and rank != 0:
        # the replay below contains the gradient all-reduce: every rank has to take the same steps
        for i in range(args.warmup, args.warmup + n_replay):
            step(i)
        torch.cuda.synchronize()
    if not args.no_roofline and rank == 0:
        import ctypes as C
        lib = bd.lib()
        lib.s2st_profile_enable.argtypes = [C.c_int32]
        lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
        lib.s2st_profile_report.restype = C.c_int64
        lib.s2st_profile_enable(1)
        for i in range(args.warmup, args.warmup + n_replay):
            step(i)
        torch.cuda.synchronize()
        lib.s2st_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        n = lib.s2st_profile_report(buf, len(buf))
        rows = {}
        for ln in buf.value.decode().splitlines() if n > 0 else []:
            tag, cnt, us, w1, w2 = ln.split("\t")
            rows[tag] = dict(n=int(cnt), us=float(us), work=float(w1), work2=float(w2))
        vlog('roofline leg done', sum(r["n"] for r in rows.values()), 'profiled launches')
        if os.environ.get('S2ST_BENCH_VERBOSE'):
            for tag, r in sorted(rows.items(), key=lambda kv: -kv[1]["us"]):
                vlog('  %-52s launches/step %6.1f  avg %7.2f us  ms/step %6.3f' % (
                    tag, r["n"] / n_replay, r["us"] / r["n"], r["us"] / n_replay * 1e-3))
        mfma = {t: r for t, r in rows.items() if t.startswith(("gemm_bf16", "flash_"))}
        gemm = {t: r for t, r in rows.items() if t.startswith("gemm_bf16")}
        alg = 3.0 * 2.0 * sum(macs[args.warmup:args.warmup + n_replay])  # valid (un-padded) tokens, fwd + bwd
        launched = sum(r["work"] for r in mfma.values())                  # as launched: padded rows / rectangles included
        useful = alg / launched                                           # algorithmic share of the launched FLOPs
        # HBM traffic per launch and kernel: rocprofv3 PMC passes of this command (they cannot run inside this process); the
        # committed measurement is used when -- and only when -- it was taken on the sources the loaded library was built from
        tj, traffic_src = None, None
        try:
            import glob
            hb = C.create_string_buffer(32)
            lib.s2st_source_hash.argtypes = [C.c_char_p, C.c_int32]
            lib.s2st_source_hash(hb, 32)
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
            for fn in cands:
                with open(fn) as f:
                    tj_ = json.load(f)
                if tj_.get("source_hash") == hb.value.decode() and tj_.get("config", "base_recipe") == args.config:
                    tj, traffic_src = tj_, "profiles/" + os.path.basename(fn)
                    break
            else:
                traffic_src = ("stale: no profiles/r*_pmc_traffic.json was measured on the sources the loaded library was built "
                               "from (%s; newest file: %s) -- tools/profile_round.sh re-measures" % (
                                   hb.value.decode(), os.path.basename(cands[0]) if cands else "none"))
        except Exception:
            pass

        def counted_bytes(tag):
            if tj is None or tag not in tj["kernels"]:
                return None
            return float(tj["kernels"][tag]["hbm_bytes_per_launch"])

        ridge = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)  # FLOP per byte at which the two roofs meet (312.5)

        def kernel_entry(tag, r):
            """One GEMM instantiation against BOTH roofs.  The binding roof follows from its arithmetic intensity --
            algorithmic FLOP per launch / HBM bytes per launch (the PMC-counted ones when a matching measurement exists, the
            bytes the launch has to move at least otherwise) -- against the ridge, not from a constant."""
            us = r["us"] / r["n"]
            alg_fl = useful * r["work"] / r["n"]
            minb = r["work2"] / r["n"]
            cb = counted_bytes(tag)
            inten = alg_fl / (cb if cb else max(minb, 1.0))
            tf = alg_fl / (us * 1e-6) / 1e12
            gb_alg = minb / (us * 1e-6) / 1e9
            return {"kernel": tag, "launches_per_step": round(r["n"] / n_replay, 1), "avg_launch_us": round(us, 2),
                    "ms_per_step": round(r["us"] / n_replay * 1e-3, 3),
                    "algorithmic_gflop_per_launch": round(alg_fl / 1e9, 3),
                    "as_launched_gflop_per_launch": round(r["work"] / r["n"] / 1e9, 3),
                    "algorithmic_bytes_per_launch": round(minb), "counted_hbm_bytes_per_launch": round(cb) if cb else None,
                    "intensity_flop_per_byte": round(inten, 1), "bound": "hbm" if inten < ridge else "mfma",
                    "achieved_tflops": round(tf, 2), "frac_mfma": round(tf / MFMA_BF16_PEAK_TFLOPS, 5),
                    "achieved_gbps_algorithmic": round(gb_alg, 1), "frac_hbm": round(gb_alg / HBM_PEAK_GBPS, 5),
                    "frac_hbm_counted": round(cb / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5) if cb else None}

        by_time = sorted(gemm.items(), key=lambda kv: -kv[1]["us"])
        top = [kernel_entry(t, r) for t, r in by_time[:3]]
        dom = top[0]
        gemm_us = sum(r["us"] for r in gemm.values())
        all_tf = useful * sum(r["work"] for r in gemm.values()) / (gemm_us * 1e-6) / 1e12
        # the whole step against its HBM traffic (every kernel of the matching PMC measurement x its launches per step)
        hbm_step = None
        if tj is not None:
            nsteps_pass = float(tj.get("steps_in_pass", 5))
            hbm_step = sum(v["hbm_bytes_per_launch"] * v["launches_fetch_pass"] / nsteps_pass for v in tj["kernels"].values())
        roofline = {
            "bound": dom["bound"],
            "kernel": dom["kernel"] + " (the GEMM instantiation with the largest share of the step in this run)",
            # `achieved` / `peak` / `frac` are the figures of the BINDING roof of that kernel; both fractions are listed
            "achieved": dom["achieved_gbps_algorithmic"] if dom["bound"] == "hbm" else dom["achieved_tflops"],
            "peak": HBM_PEAK_GBPS if dom["bound"] == "hbm" else MFMA_BF16_PEAK_TFLOPS,
            "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s",
            "frac": dom["frac_hbm"] if dom["bound"] == "hbm" else dom["frac_mfma"],
            "frac_mfma": dom["frac_mfma"], "frac_hbm": dom["frac_hbm"],
            "traffic": dom["counted_hbm_bytes_per_launch"], "traffic_unit": "B/launch", "traffic_source": traffic_src,
            "ridge_flop_per_byte": round(ridge, 1),
            "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"], "ms_per_step": dom["ms_per_step"],
            "algorithmic_gflop_per_launch": dom["algorithmic_gflop_per_launch"],
            "as_launched_gflop_per_launch": dom["as_launched_gflop_per_launch"],
            "min_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
            "algorithmic_share_of_launched_flops": round(useful, 4),
            "timing": "start/stop events attached to each dispatch (hipExtLaunchKernelGGL), %d replayed steps" % n_replay,
            # the three GEMM instantiations with the largest shares (the ring forms of the data path and the grouped
            # weight-gradient launch trade places from box to box): each with its own figures against both roofs
            "top_kernels": top,
            "all_gemm": {"launches_per_step": round(sum(r["n"] for r in gemm.values()) / n_replay, 1),
                         "ms_per_step": round(gemm_us / n_replay * 1e-3, 3), "achieved": round(all_tf, 2),
                         "frac": round(all_tf / MFMA_BF16_PEAK_TFLOPS, 5)},
            "algorithmic_gflop_per_step": round(alg / n_replay / 1e9, 1),
            "hbm_gb_per_step": round(hbm_step / 1e9, 2) if hbm_step else None,
            "step_intensity_flop_per_byte": round(alg / n_replay / hbm_step, 1) if hbm_step else None,
            "step_hbm_frac": round(hbm_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4) if hbm_step else None,
            "recompute_from": {"durations": "this run's per-dispatch events; the rocprofv3 kernel trace of the same command is "
                                            "profiles/<round tag>_kernel_stats_replayed_steps.txt",
                               "traffic": traffic_src},
            # the HBM-bound kernels of the step: bytes they have to move / their own dispatch time
            "hbm_kernels": {t: {"gbps": round(r["work"] / (r["us"] * 1e-6) / 1e9, 1),
                                "frac": round(r["work"] / (r["us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                "avg_launch_us": round(r["us"] / r["n"], 2),
                                "launches_per_step": round(r["n"] / n_replay, 1)}
                            for t, r in rows.items() if not t.startswith(("gemm_bf16", "flash_")) and r["work"] > 0},
        }

    if args.timeline and rank == 0 and world == 1:
        import ctypes as C
        lib = bd.lib()
        lib.s2st_profile_enable.argtypes = [C.c_int32]
        lib.s2st_profile_timeline.argtypes = [C.c_char_p, C.c_int64]
        lib.s2st_profile_timeline.restype = C.c_int64
        step(args.warmup)
        torch.cuda.synchronize()
        lib.s2st_profile_enable(1)
        step(args.warmup + 1)
        torch.cuda.synchronize()
        lib.s2st_profile_enable(0)
        buf = C.create_string_buffer(1 << 20)
        n = lib.s2st_profile_timeline(buf, len(buf))
        os.makedirs(os.path.dirname(os.path.abspath(args.timeline)), exist_ok=True)
        with open(args.timeline, "w") as f:
            f.write(buf.value.decode() if n > 0 else "")
        vlog("timeline of one step written to", args.timeline)

    # ---- CPU baseline leg: the oracle (torch fp32, all usable host cores) on one timed batch, in a child process
    #      with a hard time limit so a slow host can never stall the bench ----------------------------------------
    cpu = None
    if args.cpu_seconds > 0 and rank == 0 and world == 1:
        import subprocess
        n_cpu = args.cpu_utts or (8 if hub else 0)
        ids = mine[args.warmup] if n_cpu <= 0 else mine[args.warmup][:n_cpu]
        idx = ",".join(str(int(i)) for i in ids)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", idx, "--config", args.config,
                                "--n-utts", str(args.n_utts), "--cpu-seconds", str(args.cpu_seconds),
                                "--cpu-whole", str(int(n_cpu <= 0))],
                               capture_output=True, text=True, timeout=8 * args.cpu_seconds + 120)
            for ln in r.stdout.splitlines():
                if ln.startswith("{"):
                    cpu = json.loads(ln)
            if cpu is None:
                vlog("cpu leg produced no result:", r.stderr[-400:])
        except subprocess.TimeoutExpired:
            vlog("cpu leg timed out")

    # ---- the other single-GPU BASELINE workloads as short legs: so that configs[3] (on one GPU) and
    #      configs[4] carry a figure in the driver's own run.  Fresh child processes of this script (started while this
    #      process sits idle; never an exec), each printing its own line; summarised under config.other_configs ----------
    others = None
    # (under rocprofv3 the children would inherit the profiler's preload and its -d / -o target: three processes writing one
    #  run_results.db, the other-config legs inside the per-kernel accounting -- ADVICE r5; skipped there)
    profiled = any(("rocprof" in os.environ.get(k, "").lower()) for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB",
                                                                          "ROCPROFILER_LIBRARY_PATH")) or \
        any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ)
    if rank == 0 and world == 1 and args.config == "base_recipe" and not args.no_other_configs and not profiled:
        import subprocess
        others = {}
        legs = {"infer_base": ["--config", "infer_base", "--steps", "8", "--warmup", "2", "--cpu-seconds", "0", "--no-roofline"],
                "base_recipe_hubert": ["--config", "base_recipe_hubert", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0",
                                       "--no-host-fed", "--no-roofline"]}
        torch.cuda.empty_cache()
        for name, extra in legs.items():
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-other-configs"] + extra,
                                   capture_output=True, text=True, timeout=600)
                got = None
                for ln in r.stdout.splitlines():
                    if ln.startswith("{"):
                        got = json.loads(ln)
                if got is None:
                    others[name] = {"error": (r.stderr or "no line")[-300:]}
                else:
                    others[name] = {k: got[k] for k in ("metric", "value", "unit", "ms_per_step", "steps") if k in got}
                    others[name]["workload"] = got.get("config", {}).get("workload")
                    for k in ("mcd_vs_cpu", "early_stop"):
                        if k in got.get("config", {}):
                            others[name][k] = got["config"][k]
            except subprocess.TimeoutExpired:
                others[name] = {"error": "timed out"}

    # ---- 8-GPU pricing, term (a): straggler loss.  Batches are dealt round-robin (fairseq/data/iterators.py:518-548) and
    #      every update waits for the slowest rank: max / mean - 1 of the ranks' step times.  N > 1: measured per rank;
    #      N = 1: SIMULATED from this run's per-batch GPU times for W = 2, 4, 8 (update u of rank r takes batch u W + r) ----
    straggler = None
    if world > 1:
        mine_t = torch.tensor([my_dt_local / args.steps * 1e3], dtype=torch.float64, device="cpu" if share else dev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        torch.distributed.all_gather(allt, mine_t)
        ts = [float(t[0]) for t in allt]
        straggler = {"per_rank_ms_per_step": {"min": round(min(ts), 3), "mean": round(sum(ts) / len(ts), 3), "max": round(max(ts), 3)},
                     "straggler_loss": round(max(ts) / (sum(ts) / len(ts)) - 1.0, 4),
                     "note": "host time of each rank's own timed loop (it contains the waits for the other ranks' gradients)"}
    elif len(step_ms) >= 8:
        sim = {}
        for W in (2, 4, 8):
            nu = len(step_ms) // W
            if nu < 1:
                continue
            mx = [max(step_ms[u * W:(u + 1) * W]) for u in range(nu)]
            mean = sum(step_ms[:nu * W]) / (nu * W)
            sim["dp%d" % W] = round(sum(mx) / nu / mean - 1.0, 4)
        straggler = {"simulated_straggler_loss": sim,
                     "per_batch_gpu_ms": {"min": round(min(step_ms), 3), "mean": round(sum(step_ms) / len(step_ms), 3),
                                          "max": round(max(step_ms), 3)},
                     "note": "max over W consecutive batches / their mean - 1, averaged over the updates the timed batches give: what "
                             "round-robin dealing costs before any communication"}

    if rank == 0:
        value = total_frames / dt
        line = {
            "metric": "mel-frames/sec (fwd+bwd+opt) on Fisher-shaped fbank80->mel80", "value": round(value, 1),
            "unit": "mel-frames/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("s2st_transformer base 12enc/6dec d512 nfps4 + aux ASR/ST(1x64) + CTC, "
                                    "recipe dropouts, max-tokens=20000 Fisher-shaped batches, update-freq 1"
                                    if not hub else
                                    "frozen hubert_base front end on 16 kHz audio (--use-hubert true, random-init weights) "
                                    "+ s2st_transformer base 12enc/6dec d512 nfps4 + aux ASR/ST(1x64), CTC off (the "
                                    "reference fails with both, SURVEY B.7), recipe dropouts, max-tokens=20000 "
                                    "(fbank-frame cost) Fisher-shaped batches, update-freq 1"),
                       "name": args.config,
                       "global_batch_mel_frames": round(total_frames / args.steps, 1),
                       "parallelism": f"dp{world}", "gemm": "bf16 MFMA operands (bf16 copies of fp32 tensors), fp32 accumulate, fp32 master weights / residual stream / softmax / losses",
                       "final_loss": round(last_loss, 4),
                       "model_tflops": round(total_flops / dt / 1e12, 2),
                       "optimizer_update": ("scale + clip + Adam of every timed step inside the timed region; step k's update runs in "
                                            "16 chunks on the engine's second stream beside step k + 1's forward, which waits chunk by "
                                            "chunk (same trajectory bit for bit; S2ST_ADAM_OVERLAP=0: in line)" if ADAM_OVERLAP else
                                            "scale + clip + Adam of every timed step inside the timed region, in line"),
                       **({"host_fed_ms_per_step": round(host_fed, 3),
                           "host_fed_note": "the same steps fed from host batches through DevicePrefetcher (feature upload + "
                                            "Engine.prepare per batch inside the timed region, steady state); never `value`"}
                          if host_fed is not None else {})},
        }
        if roofline:
            line["roofline"] = roofline
        if cpu:
            line["cpu_baseline"] = cpu
            line["config"]["x_over_cpu"] = round(value / cpu["value"], 1)
        if exchange:
            line["gradient_exchange"] = exchange  # rank 0's view
        if straggler:
            line["straggler"] = straggler
        if proxy:
            line["exchange_proxy"] = proxy
            line["config"]["name"] = args.config + "+exchange_proxy"
        if others:
            line["config"]["other_configs"] = others
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
