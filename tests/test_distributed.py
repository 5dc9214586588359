"""Data-parallel path on CPU: 2 ranks over gloo, each driving the emulator build.  The gradient
ranges all-reduced segment by segment + the device-side world/sum(sample_size) multiplier must
reproduce a single-process update over both batches (what DDP + Trainer.train_step do,
fairseq/trainer.py:838-843)."""
import importlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "speech-to-speech-translation_amd"


def _worker(rank, world, port, q, on_gpu=False, fast=False):
    try:
        _worker_body(rank, world, port, q, on_gpu, fast)
    except BaseException as e:  # the parent must not wait for a result that will never come
        import traceback
        q.put(("error", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))
        raise


def _worker_body(rank, world, port, q, on_gpu, fast):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import s2st_amd  # noqa: F401
    import s2st_oracle as O
    from synth_weights import load_synth
    from test_engine import NANO, nano_batches
    bd = importlib.import_module(PKG + ".runtime.binding")
    if on_gpu:  # the product library; both ranks share cuda:0 (RCCL refuses that, gloo through pinned host memory does not)
        torch.cuda.set_device(0)
        bd.load_library(bd.DEFAULT_LIB, emulator=False)
    else:
        bd.load_library(os.path.join(ROOT, "tests", "hipemu", "_build", "libs2st_emu.so"), emulator=True)
    tasks = importlib.import_module(PKG + ".tasks")
    tr = importlib.import_module(PKG + ".trainer")
    D = importlib.import_module(PKG + ".data")
    from test_engine import MICRO
    cfg = NANO if not fast else dict(MICRO, encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2,
                                     decoder_attention_heads=2)
    a = O.make_args(**cfg)
    a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = not fast, 1e-3, 1, 0.05
    a.grad_exchange_dtype = os.environ.get("S2ST_TEST_EXCHANGE", "fp32")
    task = tasks.S2ST_TranslationTask.setup_task(a, device=torch.device("cuda", 0) if on_gpu else torch.device("cpu"))
    model = task.build_model(a)
    load_synth(model, rank)  # deliberately different per rank: the Trainer must broadcast rank 0's
    crit = task.build_criterion(a)
    trainer = tr.Trainer(a, task, model, crit)
    trainer.reducer.min_bucket = 50_000  # several buckets even on the nano model
    assert trainer.reducer.exchange_dtype == a.grad_exchange_dtype
    mine = nano_batches()[rank]
    gnorms = []
    for u in range(3):
        # third update: rank 1's shard has run out (the sharded iterator hands it an empty batch)
        r = trainer.train_step([mine if (u < 2 or rank == 0) else {}])
        gnorms.append(float(r["gnorm"]))
    if on_gpu:
        assert trainer.reducer.staged
        two_streams = fast and os.environ.get("S2ST_NO_SIDE_STREAM", "0") in ("", "0")
        assert (trainer.reducer.extra_stream is not None) == two_streams  # bf16 mode: weight gradients on the second stream
        torch.cuda.synchronize()
    if rank == 0:
        q.put({n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()})
        q.put(gnorms if fast else float(r["gnorm"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_two_rank_bf16_gradient_exchange_is_bounded(where, monkeypatch):
    """``--grad-exchange-dtype bf16`` (SURVEY 8(e) prices it at half the wire time): every finished range
    is rounded to bf16 (``s2st_grad_pack_bf16_f32``), summed over the ranks in that type and widened back.  Same scenario as
    the fp32 test below, against the same single-process oracle: each rank's contribution carries one bf16 rounding (2^-9
    relative per element): the clipped gradient norm stays within 5e-3, and the parameters after three Adam updates within
    1.5e-2 of their scale -- measured 6.3e-3 on the emulator (the subsampler's second convolution: Adam divides by sqrt(v),
    so elements whose gradient is small against its rounding step move by whole learning-rate steps); the fp32 exchange is
    held to 2e-3 / 1e-3.  fp32 stays the default: it is what the reference exchanges."""
    monkeypatch.setenv("S2ST_TEST_EXCHANGE", "bf16")
    _two_rank_vs_oracle(where, gnorm_tol=5e-3, param_tol=1.5e-2)


@pytest.mark.parametrize("where", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_two_rank_update_equals_single_process(where):
    _two_rank_vs_oracle(where, gnorm_tol=2e-3, param_tol=1e-3)


def _two_rank_vs_oracle(where, gnorm_tol, param_tol):
    """Two processes, three updates (the third with an exhausted shard on rank 1), against the ORACLE's single-process
    update over both batches.  ``emu``: host gradients, emulator build.  ``hip``: both ranks drive the product library on
    the one GPU the test box has -- device gradients, the engine's two streams, the reducer's stream and events, the
    device-side sample-size exchange -- with gloo carrying the bytes (RCCL refuses two ranks on one device)."""
    import subprocess
    on_gpu = where == "hip"
    if on_gpu:
        assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    else:
        subprocess.check_call([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, on_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    params = q.get(timeout=900 if not on_gpu else 300)
    assert not (isinstance(params, tuple) and params[0] == "error"), params[1]
    gnorm = q.get(timeout=60)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: both batches, gradients summed, scaled by 1 / total sample size
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import s2st_oracle as O
    from synth_weights import load_synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_engine import NANO, nano_batches
    a = O.make_args(**NANO)
    m = O.S2STModel(a)
    load_synth(m, 0)
    m.train()
    opt = O.FairseqAdam(m.parameters())
    b0, b1 = nano_batches()
    for u in range(3):
        for p in m.parameters():
            p.grad = None
        ss = 0
        for s in ((b0, b1) if u < 2 else (b0,)):
            loss, n, _, _ = O.criterion_forward(m, s)
            loss.backward()  # accumulates
            ss += n
        with torch.no_grad():
            for p in m.parameters():
                if p.grad is not None:
                    p.grad.mul_(1.0 / ss)  # world / sum(sample_size) on gradients summed over ranks / world ...
        gn = O.clip_grad_norm_(list(m.parameters()), 0.05)
        opt.step(O.inverse_sqrt_lr(u, 1e-3, 1))
    assert abs(gnorm - float(gn)) < gnorm_tol * float(gn), (gnorm, float(gn))
    worst = max((float((torch.from_numpy(params[n]) - p.detach()).abs().max()) / (float(p.detach().abs().max()) + 1e-6), n)
                for n, p in m.named_parameters())
    print(f"[two ranks vs oracle, exchange {os.environ.get('S2ST_TEST_EXCHANGE', 'fp32')}] gnorm {gnorm:.6f} vs {float(gn):.6f}, "
          f"worst parameter {worst[1]} {worst[0]:.2e}")
    assert worst[0] < param_tol, worst


@pytest.mark.gpu
def test_two_ranks_in_bf16_mode_equal_one_process_accumulating_both_batches():
    """The benchmarked mode (bf16 operands: weight gradients on the engine's SECOND stream, grouped launches) through the
    two-process exchange on one GPU: every gradient range is reduced behind both engine streams.  Two ranks with one
    batch each must give the update of one process that accumulates both batches (update-freq 2): same kernels, sums in
    a different order."""
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True, True)) for r in range(2)]
    for p in procs:
        p.start()
    params = q.get(timeout=300)
    assert not (isinstance(params, tuple) and params[0] == "error"), params[1]
    gnorm = q.get(timeout=60)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import s2st_amd  # noqa: F401
    import s2st_oracle as O
    from synth_weights import load_synth
    from test_engine import MICRO, nano_batches
    bd = importlib.import_module(PKG + ".runtime.binding")
    bd.load_library(bd.DEFAULT_LIB, emulator=False)
    tasks = importlib.import_module(PKG + ".tasks")
    tr = importlib.import_module(PKG + ".trainer")
    cfg = dict(MICRO, encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2, decoder_attention_heads=2)
    a = O.make_args(**cfg)
    a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = False, 1e-3, 1, 0.05
    task = tasks.S2ST_TranslationTask.setup_task(a, device=torch.device("cuda", 0))
    model = task.build_model(a)
    load_synth(model, 0)
    trainer = tr.Trainer(a, task, model, task.build_criterion(a))
    b0, b1 = nano_batches()
    ref_gnorm = []
    for u in range(3):
        r = trainer.train_step([b0, b1] if u < 2 else [b0])
        ref_gnorm.append(float(r["gnorm"]))
    torch.cuda.synchronize()
    # (all dropouts are 0 in this configuration: the two runs differ only in the order of fp32 sums.)
    # Updates 1 and 2 (both ranks contribute): the gradient norms agree to summation order.
    assert isinstance(gnorm, list) and len(gnorm) == 3
    for u in (0, 1):
        assert abs(gnorm[u] - ref_gnorm[u]) < 1e-5 * ref_gnorm[u], (u, gnorm, ref_gnorm)
    # Update 3 (rank 1's shard exhausted: its dummy batch must count zero) is the first forward after the weights really
    # changed.  In bf16 mode its gradient norm has been observed on a few discrete values ~1e-3 apart, in EITHER arm and in
    # a single process too ("Reproducibility": an open issue of the bf16 schedule, independent of
    # the exchange tested here; the gradients that differ most are the mathematically zero ones -- key-projection biases,
    # convolution biases in front of BatchNorm).  A lost or doubled rank contribution moves the norm by O(1).
    assert abs(gnorm[2] - ref_gnorm[2]) < 5e-3 * ref_gnorm[2], (gnorm, ref_gnorm)
    noise_driven = lambda n: n.endswith("k_proj.bias") or (".postnet.convolutions." in n and n.endswith(".0.bias"))  # noqa: E731
    for n, p in model.named_parameters():
        if noise_driven(n):
            continue
        ref = p.detach().cpu()
        # three Adam updates of <= lr = 1e-3 each (the first with lr 0): entries whose third-step gradient is small follow
        # that difference through Adam's normalisation by up to ~1 lr; a lost or doubled contribution moves whole tensors
        assert float((torch.from_numpy(params[n]) - ref).abs().max()) < 2.5e-3, n
        assert float((torch.from_numpy(params[n]) - ref).abs().mean()) < 1e-4, n


@pytest.mark.gpu
def test_native_communicator_through_the_c_abi():
    """``s2st_comm_*`` / ``s2st_allreduce_sum_f32`` (include/s2st_hip.h): the library binds the RCCL copy the process
    already carries, creates a communicator from a unique id and runs an in-place SUM all-reduce on a side stream.  One
    GPU here, so one rank (the SUM is then the identity): this pins binding, call signature, stream ordering and error
    codes; N > 1 is the driver's multi-GPU bench (``S2ST_NATIVE_ALLREDUCE=1``)."""
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    import s2st_amd  # noqa: F401
    bd = importlib.import_module(PKG + ".runtime.binding")
    bd.load_library(bd.DEFAULT_LIB, emulator=False)
    dm = importlib.import_module(PKG + ".runtime.distributed")
    torch.cuda.set_device(0)
    comm = dm.NativeComm()
    assert comm.world == 1 and comm.rank == 0
    x = torch.randn(3_000_001, device="cuda:0")
    ref = x.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        comm.all_reduce_(x)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    # argument errors come back as codes, not crashes
    assert comm.lib.s2st_allreduce_sum_f32(None, x.data_ptr(), 4, None) == -4
    comm.close()


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_control_flow():
    """``bench.py --gpus 2`` exactly as the driver launches it (``python -m torch.distributed.run --nproc-per-node 2 ...``),
    with both ranks on the one GPU of this box (``S2ST_BENCH_SHARE_GPU=1``: gloo carries the gradient bytes, RCCL
    refuses two ranks per device): the N > 1 control flow -- rendezvous on 127.0.0.1, parameter broadcast, round-robin
    batches, range-wise exchange behind both engine streams, barriers around the timed region, max-over-ranks time,
    rank-0-only JSON line with the whole-job value, the exchange's exposure figure and bucket list -- runs every round.
    A control-flow check, not a measurement."""
    import json
    import socket
    import subprocess
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, S2ST_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--no-roofline", "--cpu-seconds", "0", "--n-utts", "512"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["parallelism"] == "dp2"
    ex = rec["gradient_exchange"]
    assert ex["allreduce_exposed_ms"] >= 0 and ex["bytes_per_update"] > 280e6 and ex["dtype"] == "f32"
    assert len(ex["buckets_mib"]) >= 3 and min(ex["buckets_mib"][:-1]) >= 16.0  # >= 16 MiB ranges (the tail may be smaller)
    assert abs(rec["ms_per_step"] * rec["steps"] * rec["value"] / 1e3 - rec["config"]["global_batch_mel_frames"] * rec["steps"]) \
        < 1e-3 * rec["config"]["global_batch_mel_frames"] * rec["steps"]
    assert rec["n_ranks_seen"] == 2
    # the driver's command shape is `python bench.py --gpus N` WITHOUT a launcher -- bench.py starts the
    # N ranks itself (fresh children, before the parent touches the GPU) and relays their one line ...
    bare = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-roofline",
            "--cpu-seconds", "0", "--n-utts", "512"]
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(bare, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec2 = json.loads(lines[0])
    assert rec2["n_gpus"] == 2 and rec2["n_ranks_seen"] == 2 and "gradient_exchange" in rec2
    assert rec2["config"]["global_batch_mel_frames"] == rec["config"]["global_batch_mel_frames"]  # same batches, same ranks
    # ... and refuses loudly -- non-zero exit, no result line -- when the node has fewer devices than ranks asked for
    if torch.cuda.device_count() < 2:
        env2 = {k: v for k, v in env.items() if k != "S2ST_BENCH_SHARE_GPU"}
        r = subprocess.run(bare, capture_output=True, text=True, timeout=300, env=env2, cwd=ROOT)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert "refusing" in r.stderr


def test_bench_refuses_more_ranks_than_devices():
    """(CPU) ``python bench.py --gpus 2`` on a box without two devices exits non-zero and prints no JSON line; so does a
    launcher whose WORLD_SIZE contradicts --gpus."""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two devices")
    env = {k: v for k, v in os.environ.items() if k not in ("S2ST_BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "contradicts" in r.stderr and "{" not in r.stdout
