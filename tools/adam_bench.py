#!/usr/bin/env python3
"""Optimizer kernel alone on the base model's arena size: python tools/adam_bench.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s2st_amd  # noqa
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
dev = torch.device("cuda:0")
n = 56_500_000
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
ss = torch.ones(1, device=dev); gn = torch.zeros(1, device=dev)
ph = torch.zeros(n, dtype=torch.bfloat16, device=dev)
for with_ph in (None, ph):
    for _ in range(3):
        bd.call("s2st_adam_f32", p, g, m, v, n, ss, 1e-4, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 3, gn, with_ph, None, 0, 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        bd.call("s2st_adam_f32", p, g, m, v, n, ss, 1e-4, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 3, gn, with_ph, None, 0, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    gb = n * (32 + (2 if with_ph is not None else 0)) / 1e9
    print(f"variant={os.environ.get('S2ST_ADAM_VARIANT','0')} blocks={os.environ.get('S2ST_ADAM_BLOCKS','4096')} ph={'yes' if with_ph is not None else 'no'}: {dt*1e6:.0f} us, {gb/dt/1e3:.2f} TB/s")
