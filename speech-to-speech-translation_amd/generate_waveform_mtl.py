"""``generate_waveform_mtl``: the counterpart of ``examples/s2s_trans/generate_waveform_mtl.py:129-219`` -- generation with
the ``s2s_translation_mtl`` task's model.

    python -m s2st_amd.generate_waveform_mtl DATA --task s2s_translation_mtl --path CKPT.pt --gen-subset test \
        --results-path OUT --decode-source-text --decode-target-mel --dump-features --dump-waveforms

On top of ``generate_waveform``'s flags (:42-45): ``--decode-source-text`` (greedy CTC transcript of the SOURCE speech from
the encoder-tap head; writes ``src_texts.txt`` / ``hyps_src_texts.txt`` under the results path and prints the corpus WER,
:183-210), ``--decode-target-mel`` (the AR mel decoder + vocoder; without it nothing is dumped, :202-206),
``--decode-target-text`` (parsed and unused, as in the reference), ``--middle-layers`` (the checkpoint's value wins) and
``--scoring`` (``wer``).  The loop itself is ``generate_waveform.main(mtl=True)``.
"""
from __future__ import annotations

import sys
from typing import Dict, List, Optional

from . import generate_waveform as _gw


def make_parser():
    p = _gw.make_parser()
    p.prog = "s2st_amd.generate_waveform_mtl"
    p.set_defaults(task="s2s_translation_mtl")
    a = p.add_argument
    a("--middle-layers", default="6", type=str)
    a("--decode-source-text", action="store_true")
    a("--decode-target-text", action="store_true")
    a("--decode-target-mel", action="store_true")
    a("--scoring", default="wer", choices=["wer"])
    return p


def main(argv: Optional[List[str]] = None, device=None, on_model_built=None) -> Dict:
    return _gw.main(argv, device=device, on_model_built=on_model_built, parser=make_parser(), mtl=True)


def cli_main():
    r = main(sys.argv[1:])
    print(f"generated {r['utterances']} utterances ({r['mel_frames']} mel frames) in {r['generate_seconds']:.2f} s "
          f"of generator time; {len(r['files'])} files under the results path")


if __name__ == "__main__":
    cli_main()
