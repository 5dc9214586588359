"""``generate_waveform``: the counterpart of ``examples/s2s_trans/generate_waveform.py:127-183`` (BASELINE configs[4]) for
the MI355X path.

    python -m s2st_amd.generate_waveform DATA --config-yaml config.yaml --gen-subset test_fisher --path CKPT.pt \
        --results-path OUT --max-tokens 50000 --spec-bwd-max-iter 64 --dump-waveforms --dump-features --dump-target

loads a checkpoint in the reference's ``.pt`` layout (model flags from ``cfg["model"]``, tensors by the names of SURVEY
Appendix A), builds the task's autoregressive generator (``task.build_generator_tts``: key/value-cached decoding on the
engine, Griffin-Lim vocoder with all utterances of a batch per GEMM), walks the split in length-ordered max-tokens
batches (no shuffle) and writes, per utterance id, what the reference's ``dump_result`` writes (:67-124):
``feat/<id>.npy`` (+ ``feat_tgt/``), ``attn/<id>.npy``, ``eos/<id>.npy``, ``wav_<rate>hz_<vocoder>/<id>.wav`` (+ ``_tgt``),
``plot/<id>.png``.  Differences, stated: waveforms are written as 16-bit PCM with the standard library (``soundfile`` is
not in the image; the reference's default subtype for wav is PCM_16 too) and only at the feature sample rate (the
reference resamples through torchaudio's sox effects when ``--output-sample-rate`` differs; asking for another rate is an
error here, not a silent no-op).  Returns a summary dict (utterances, seconds, files) so tests and ``bench.py --config
infer_base`` can drive it in-process.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import wave
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import checkpoint_utils
from .registry import TASKS
from .runtime import streams as _streams


def make_parser() -> argparse.ArgumentParser:
    """fairseq's ``options.get_speech_generation_parser`` + generate_waveform.py:28-44, by their fairseq names."""
    p = argparse.ArgumentParser(prog="s2st_amd.generate_waveform", allow_abbrev=False)
    a = p.add_argument
    a("data")
    a("--user-dir", default=None, help="accepted for command-line compatibility (this package IS the plugin)")
    a("--config-yaml", default="config.yaml")
    a("--task", default="s2s_translation")
    a("--path", required=True, help="checkpoint (reference .pt layout)")
    a("--gen-subset", default="test")
    a("--results-path", required=True)
    a("--max-tokens", type=int, default=None)
    a("--batch-size", "--max-sentences", type=int, default=None, dest="batch_size")
    a("--required-batch-size-multiple", type=int, default=1)
    a("--skip-invalid-size-inputs-valid-test", action="store_true")
    a("--num-shards", type=int, default=1)
    a("--shard-id", type=int, default=0)
    a("--num-workers", type=int, default=0)
    a("--seed", type=int, default=1)
    a("--max-target-positions", type=int, default=None, help="AR steps at most (default: the checkpoint's value)")
    a("--eos-prob-threshold", type=float, default=0.5)
    a("--vocoder", default="griffin_lim", choices=["griffin_lim"])
    a("--spec-bwd-max-iter", type=int, default=8)
    a("--gl-phase-rng", default="numpy", choices=["numpy", "device"],
      help="initial Griffin-Lim phases: numpy's global generator (the reference's draws, vocoder.py:101-102) or the device's "
           "counter-based generator (same distribution, no host work)")
    a("--dump-features", action="store_true")
    a("--dump-waveforms", action="store_true")
    a("--dump-attentions", action="store_true")
    a("--dump-eos-probs", action="store_true")
    a("--dump-plots", action="store_true")
    a("--dump-target", action="store_true")
    a("--output-sample-rate", default=None, type=int)
    a("--audio-format", default="wav", choices=["wav"])
    a("--precise-gemm", action="store_true", help="bf16x3 GEMMs (fp32-accurate; parity runs)")
    a("--max-batches", type=int, default=0, help="stop after this many batches (0: the whole split)")
    return p


def write_wav(path: str, wave_f32: np.ndarray, sample_rate: int) -> None:
    """16-bit PCM mono (what ``soundfile.write`` makes of a float array for .wav by default)."""
    x = np.clip(np.asarray(wave_f32, dtype=np.float64).reshape(-1), -1.0, 1.0 - 1.0 / 32768)
    pcm = np.round(x * 32768.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


def _to_np(x):
    return None if x is None else x.detach().cpu().numpy()


def dump_result(args, vocoder_name: str, sample_rate: int, sample_id, hypo, written: List[str]) -> None:
    """generate_waveform.py:67-124."""
    out_root = Path(args.results_path)

    def save(sub, name, fn):
        d = out_root / sub
        d.mkdir(exist_ok=True, parents=True)
        fn(str(d / name))
        written.append(str(d / name))

    if args.dump_features:
        save("feat", f"{sample_id}.npy", lambda p: np.save(p, _to_np(hypo["feature"])))
        if args.dump_target:
            save("feat_tgt", f"{sample_id}.npy", lambda p: np.save(p, _to_np(hypo["targ_feature"])))
    if args.dump_attentions:
        save("attn", f"{sample_id}.npy", lambda p: np.save(p, _to_np(hypo["attn"])))
    if args.dump_eos_probs:
        save("eos", f"{sample_id}.npy", lambda p: np.save(p, _to_np(hypo["eos_prob"])))
    if args.dump_plots:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        images = [_to_np(hypo["feature"]).T, _to_np(hypo["attn"])]
        names = ["output", "alignment"]
        if args.dump_target:
            images, names = [_to_np(hypo["targ_feature"]).T] + images, [f"target (idx={sample_id})"] + names
        fig, axes = plt.subplots(len(images) + 1, 1, figsize=(8, 2.2 * (len(images) + 1)))
        for ax, im, nm in zip(axes, images, names):
            ax.imshow(im, aspect="auto", origin="lower", interpolation="none")
            ax.set_title(nm, fontsize=8)
        axes[-1].plot(_to_np(hypo["eos_prob"]))
        axes[-1].set_title("eos prob", fontsize=8)
        fig.tight_layout()
        save("plot", f"{sample_id}.png", lambda p: fig.savefig(p))
        plt.close(fig)
    if args.dump_waveforms:
        ext = args.audio_format
        if hypo.get("waveform") is not None:
            save(f"{ext}_{sample_rate}hz_{vocoder_name}", f"{sample_id}.{ext}",
                 lambda p: write_wav(p, _to_np(hypo["waveform"]), sample_rate))
        if args.dump_target and hypo.get("targ_waveform") is not None:
            save(f"{ext}_{sample_rate}hz_{vocoder_name}_tgt", f"{sample_id}.{ext}",
                 lambda p: write_wav(p, _to_np(hypo["targ_waveform"]), sample_rate))


def main(argv: Optional[List[str]] = None, device: Optional[torch.device] = None, on_model_built=None,
         parser: Optional[argparse.ArgumentParser] = None, mtl: bool = False) -> Dict:
    """``mtl``: the flow of generate_waveform_mtl.py (see generate_waveform_mtl.py in this package): the task's own
    generator, source-transcript decoding + WER files, mel dumping only with --decode-target-mel."""
    args = (parser or make_parser()).parse_args(argv)
    decode_src = mtl and args.decode_source_text
    decode_mel = (not mtl) or args.decode_target_mel
    if not (args.dump_features or args.dump_waveforms or args.dump_attentions or args.dump_eos_probs or args.dump_plots
            or decode_src):
        raise SystemExit("nothing to do: pass at least one --dump-* flag (generate_waveform.py:128-129)")
    if args.max_tokens is None and args.batch_size is None:
        args.max_tokens = 8000  # :130-131
    if device is None:
        if not torch.cuda.is_available():
            raise SystemExit("s2st_amd.generate_waveform needs a HIP device (the product path has no CPU fallback)")
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    from . import criterions, models, tasks  # noqa: F401  (fill the registries)
    state = checkpoint_utils.load_checkpoint_to_cpu(args.path)
    margs = state["cfg"]["model"]
    margs = argparse.Namespace(**(vars(margs) if not isinstance(margs, dict) else margs))
    # generation-time flags override what the checkpoint carried (checkpoint_utils.load_model_ensemble_and_task +
    # generate_waveform.py:142-144: the task is set up from the command line, n_frames_per_step from the checkpoint)
    margs.data, margs.config_yaml = args.data, args.config_yaml
    margs.eos_prob_threshold = args.eos_prob_threshold
    margs.spec_bwd_max_iter = args.spec_bwd_max_iter
    margs.gl_phase_rng = args.gl_phase_rng
    margs.precise_gemm = bool(args.precise_gemm)
    margs.eval_inference = False
    margs.train_subset = None
    if args.max_target_positions is not None:
        margs.max_target_positions = args.max_target_positions
    for k in ("load_pretrained_encoder_from", "load_pretrained_decoder_from"):
        setattr(margs, k, None)  # the tensors come from the checkpoint itself
    task = TASKS[getattr(margs, "task", None) or args.task].setup_task(margs, device=device)
    if task.data_cfg is not None:
        margs.src_vocab_size, margs.tgt_vocab_size = len(task.source_dictionary), len(task.target_dictionary)
    model = task.build_model(margs)
    model.load_state_dict(state["model"], strict=True)
    if on_model_built is not None:
        on_model_built(model)
    dataset = task.load_dataset(args.gen_subset)
    sample_rate = task.sr
    if args.output_sample_rate not in (None, sample_rate):
        raise SystemExit(f"--output-sample-rate {args.output_sample_rate}: resampling (torchaudio sox effects in the "
                         f"reference, generate_waveform.py:148-156) is not available here; the features are {sample_rate} Hz")
    # (generate_waveform.py:158 / generate_waveform_mtl.py:164: the mtl task's build_generator IS its speech generator)
    generator = task.build_generator([model], margs) if mtl else task.build_generator_tts([model], margs)
    wer = src_f = hyp_f = None
    if decode_src:  # generate_waveform_mtl.py:183-185
        from .scoring import build_scorer
        wer = build_scorer(args.scoring, getattr(model, "src_dict", None))
        Path(args.results_path).mkdir(exist_ok=True, parents=True)
        src_f = open(os.path.join(args.results_path, "src_texts.txt"), "w")
        hyp_f = open(os.path.join(args.results_path, "hyps_src_texts.txt"), "w")
    itr = task.get_batch_iterator(dataset, max_tokens=args.max_tokens, max_sentences=args.batch_size,
                                  max_positions=(sys.maxsize, sys.maxsize),
                                  required_batch_size_multiple=args.required_batch_size_multiple, seed=args.seed,
                                  num_shards=args.num_shards, shard_id=args.shard_id).next_epoch_itr(shuffle=False)
    Path(args.results_path).mkdir(exist_ok=True, parents=True)
    ids = getattr(dataset, "ids", None)
    written: List[str] = []
    n_utt, n_frames, t_gen, n_batches = 0, 0, 0.0, 0
    def finish(sample, hypos):
        """What the reference's loop does with a batch's hypotheses (generate_waveform.py:172-183 / _mtl.py:196-205)."""
        nonlocal n_utt, n_frames
        if hasattr(hypos, "wait"):
            hypos.wait()  # (the vocoder ran on the generator's second stream, beside the next batch's decoding)
        if decode_src:  # :196-200
            for hypo in hypos:
                wer.add_string(hypo["src_texts"], hypo["hyps_src_texts"])
                src_f.write(hypo["src_texts"] + "\n")
                hyp_f.write(hypo["hyps_src_texts"] + "\n")
        for i, hypo in zip(sample["id"].tolist(), hypos):
            if decode_mel:
                dump_result(args, args.vocoder, sample_rate, ids[i] if ids is not None else i, hypo, written)
                n_frames += int(hypo["feature"].shape[0])
            n_utt += 1

    # one batch of overlap on the GPU: batch k's files are written after batch k + 1 has been enqueued, so that its
    # Griffin-Lim iterations run beside that batch's decoding steps (S2ST_DEFER_VOCODER=0: strictly one after the other)
    defer = device.type == "cuda" and os.environ.get("S2ST_DEFER_VOCODER", "1") != "0"
    # ... and two batches are DECODED at once (generate_two: the second on a twin engine and a second stream) where the
    # generator has that form: the decoding steps of one batch leave most of the chip idle (S2ST_DECODE_CHAINS=1: one by one)
    chains = _streams.default_decode_chains() if (defer and not mtl and hasattr(generator, "generate_many")) else 1

    def groups():
        buf, n = [], 0
        for sample in itr:
            if sample is None or len(sample) == 0:
                continue
            buf.append(sample)
            n += 1
            last = bool(args.max_batches and n >= args.max_batches)
            if len(buf) == chains or last:
                yield buf
                buf = []
            if last:
                return
        if buf:
            yield buf

    held = []
    t_all = time.perf_counter()
    for group in groups():
        t0 = time.perf_counter()
        if len(group) >= 2:
            hyps = list(generator.generate_many(model, group, has_targ=args.dump_target, defer_vocoder=defer))
        elif mtl:  # generate_waveform_mtl.py:195
            hyps = [generator.generate(model, group[0], has_targ=args.dump_target and decode_mel,
                                       decode_source_text=decode_src, decode_target_mel=decode_mel, defer_vocoder=defer)]
        else:
            hyps = [generator.generate(model, group[0], has_targ=args.dump_target, defer_vocoder=defer)]
        if device.type == "cuda" and not defer:
            torch.cuda.synchronize(device)
        t_gen += time.perf_counter() - t0
        for h in held:
            finish(*h)
        held = list(zip(group, hyps))
        if not defer:
            for h in held:
                finish(*h)
            held = []
        n_batches += len(group)
    for h in held:
        finish(*h)
    if defer:  # (generator time cannot be told from file writing when the two overlap: the loop's wall time)
        torch.cuda.synchronize(device)
        t_gen = time.perf_counter() - t_all
    out = {"utterances": n_utt, "mel_frames": n_frames, "generate_seconds": t_gen, "batches": n_batches,
           "files": written, "sample_rate": sample_rate}
    if decode_src:  # :207-210
        src_f.close()
        hyp_f.close()
        out["wer"] = wer.score()
        written += [src_f.name, hyp_f.name]
        print(f"WER: {wer.score()}")
    return out


def cli_main():
    r = main(sys.argv[1:])
    print(f"generated {r['utterances']} utterances ({r['mel_frames']} mel frames) in {r['generate_seconds']:.2f} s "
          f"of generator time; {len(r['files'])} files under the results path")


if __name__ == "__main__":
    cli_main()
