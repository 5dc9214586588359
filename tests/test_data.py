"""On-disk data path against goldens produced by the reference's own dataset classes
(oracle/gen_golden_data.py): TSV manifest, byte-range reads out of an uncompressed zip and plain .npy, per-side
feature transforms (utterance / global CMVN, SpecAugment with numpy's global RNG), dictionaries with OOV, target
frame stacking, length-sorted indices and the collated batch -- byte / integer work, compared bit-exactly."""
import importlib
import os

import numpy as np
import pytest
import torch

from data_corpus import flatten_batch, make_corpus

PKG = "speech-to-speech-translation_amd"


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    return make_corpus(str(tmp_path_factory.mktemp("s2st_corpus")))


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "data_path.npz"))


def _load(corpus, split):
    D = importlib.import_module(PKG + ".data")
    cfg = D.S2STDataConfig(os.path.join(corpus, "config.yaml"))
    sd = D.Dictionary.load(os.path.join(corpus, cfg.src_vocab_filename))
    td = D.Dictionary.load(os.path.join(corpus, cfg.tgt_vocab_filename))
    ds = D.S2STDatasetCreator.from_tsv(corpus, cfg, split, sd, td, None, None, is_train_split=split.startswith("train"),
                                       epoch=1, seed=1, n_frames_per_step=4, speaker_to_id={"spk0": 0, "spk1": 1})
    return ds, sd, td


@pytest.mark.parametrize("split", ["train_tiny", "dev_tiny"])
def test_items_and_batch_equal_the_reference(corpus, golden, split):
    ds, sd, td = _load(corpus, split)
    assert len(sd) == int(golden["src_dict_len"]) and len(td) == int(golden["tgt_dict_len"])
    np.random.seed(11)
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        for k in ("src_speech", "tgt_speech", "src_text", "tgt_text"):
            ref = golden[f"{split}.item{i}.{k}"]
            got = getattr(it, k).numpy()
            assert got.dtype == ref.dtype and got.shape == ref.shape, (i, k)
            assert np.array_equal(got, ref), (i, k)
    assert np.array_equal(np.asarray(ds.ordered_indices()), golden[f"{split}.ordered_indices"])
    assert np.array_equal(np.asarray([ds.size(i) for i in range(len(ds))]), golden[f"{split}.sizes"])
    pick = golden[f"{split}.batch_pick"].tolist()
    got = flatten_batch(ds.collater([items[i] for i in pick]))
    want = {k[len(split) + 7:]: golden[k] for k in golden.files if k.startswith(split + ".batch.")}
    assert set(got) == set(want), set(got) ^ set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert np.array_equal(got[k], want[k]), k


def test_batches_feed_the_engine_schema(corpus):
    """max-tokens batching over the on-disk set yields the sample schema the engine consumes (same keys as the
    synthetic corpus' batches); items beyond max_positions are dropped like filter_indices_by_size does."""
    D = importlib.import_module(PKG + ".data")
    ds, _, _ = _load(corpus, "train_tiny")
    batches = ds.batches(max_tokens=150, bsz_mult=2, max_positions=(55, 2400))
    seen = sorted(int(i) for b in batches for i in b)
    assert seen == sorted(i for i in range(len(ds)) if ds.n_frames[i] <= 55)
    for b in batches:
        assert len(b) * max(ds.n_frames[int(i)] for i in b) <= 150
    s = ds.collate_batch(batches[0])
    syn = D.SyntheticFisherCorpus(n_utts=4, seed=1).collate_batch(range(2))
    assert set(syn) <= set(s) and set(syn["net_input"]) <= set(s["net_input"])
    assert s["net_input"]["prev_output_tokens"].shape == s["tgt_speech"].shape
    assert torch.equal(s["net_input"]["prev_output_tokens"][:, 1:], s["tgt_speech"][:, :-1])


def test_dictionary_file_format(tmp_path):
    D = importlib.import_module(PKG + ".data")
    p = tmp_path / "d.txt"
    p.write_text("a 5\nb c 3\na 9 #fairseq:overwrite\n")
    d = D.Dictionary.load(str(p))
    assert [d.bos(), d.pad(), d.eos(), d.unk()] == [0, 1, 2, 3] and d.symbols[4:6] == ["a", "b c"]
    assert d.index("a") == 6 and d.index("zzz") == d.unk()  # the overwrite row re-points the symbol to a new index
    assert d.encode_line("a  zzz\tb", add_if_not_exist=False).tolist() == [6, 3, 3, 2]
    p.write_text("a 5\na 6\n")
    with pytest.raises(RuntimeError):
        D.Dictionary.load(str(p))
    p.write_text("a\n")
    with pytest.raises(ValueError):
        D.Dictionary.load(str(p))


def test_on_disk_corpus_trains_through_the_task(backend, corpus):
    """data directory -> setup_task (dictionaries from the config) -> load_dataset -> max-tokens batches ->
    Trainer.train_step on the HIP path; the same batches through the CPU oracle give the same losses."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import s2st_oracle as O
    from synth_weights import load_synth
    from test_engine import NANO
    tasks = importlib.import_module(PKG + ".tasks")
    tr = importlib.import_module(PKG + ".trainer")
    a = O.make_args(**NANO)
    a.data, a.config_yaml, a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = corpus, "config.yaml", True, 1e-3, 2, 0.05
    task = tasks.S2ST_TranslationTask.setup_task(a, device=backend.device)
    assert len(task.source_dictionary) == 12 and len(task.target_dictionary) == 14
    a.src_vocab_size, a.tgt_vocab_size = len(task.source_dictionary), len(task.target_dictionary)
    ds = task.load_dataset("dev_tiny")  # eval split: no SpecAugment draws
    model = task.build_model(a)
    load_synth(model, 0)
    trainer = tr.Trainer(a, task, model, task.build_criterion(a))
    m = O.S2STModel(a)
    load_synth(m, 0)
    opt = O.FairseqAdam(m.parameters())
    batches = ds.batches(max_tokens=120, bsz_mult=2, max_positions=task.max_positions())
    assert len(batches) >= 2
    for u, b in enumerate(batches[:2]):
        s = ds.collate_batch(b)
        r = trainer.train_step([s])
        lo, gn, lr, log, _ = O.train_step(m, opt, s, u, 1e-3, 2, 0.05)
        backend.sync()
        assert abs(float(r["logs"][0]["loss"]) - float(lo)) < 5e-5 * float(lo)
        assert abs(float(r["gnorm"]) - float(gn)) < 2e-3 * float(gn)


def _batch_ends(batches):
    return np.cumsum([len(b) for b in batches]).astype(np.int64)


def test_native_batcher_equals_the_reference_cython_extension(golden_dir):
    """s2st_batch_by_size (C library) and the oracle restatement against outputs of the reference's own
    data_utils_fast extension (tests/golden/batcher.npz; integer work: exact), and -- when oracle/_ref holds that
    extension, i.e. wherever build() ran with the reference present -- against the extension itself on fresh
    random cases."""
    import sys
    D = importlib.import_module(PKG + ".data")
    import data_oracle as DO
    z = np.load(os.path.join(golden_dir, "batcher.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) >= 8
    for name in names:
        nt = z[name + ".num_tokens"]
        mt, ms, mult = (int(v) for v in z[name + ".args"])
        idx = np.arange(len(nt), dtype=np.int64)
        for fn in (D.batch_by_size, DO.batch_by_size):
            got = fn(idx, nt, mt, ms, mult)
            assert np.array_equal(_batch_ends(got), z[name + ".ends"]), (name, fn.__module__)
            assert np.array_equal(np.concatenate(got), idx)
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    try:
        sys.path.insert(0, ref_dir)
        import data_utils_fast as F
    except ImportError:
        return
    rs = np.random.RandomState(17)
    for _ in range(200):
        n = int(rs.randint(1, 300))
        nt = rs.randint(1, 800, size=n).astype(np.int64)
        if rs.rand() < 0.7:
            nt = np.sort(nt)[::-1].copy()
        mt = int(rs.choice([0, 800, 1500, 5000]))
        ms = int(rs.choice([-1, 0, 3, 16]))
        mult = int(rs.choice([1, 2, 8]))
        idx = rs.permutation(n).astype(np.int64)
        want = F.batch_by_size_vec(idx, nt, mt, ms, mult)
        got = D.batch_by_size(idx, nt, mt, ms, mult)
        assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    with pytest.raises(AssertionError):
        D.batch_by_size(np.arange(3), np.asarray([10, 2000, 5]), 1000, -1, 1)
    assert D.batch_by_size(np.arange(0), np.arange(0), 100, -1, 8) == []


def test_epoch_iterator_order_equals_the_reference(golden_dir):
    """Per-epoch shuffled, rank-sharded batch order and the resume position against the reference's
    EpochBatchIterator (tests/golden/epoch_iterator.npz): every rank must see exactly the reference's batches."""
    D = importlib.import_module(PKG + ".data")
    zb = np.load(os.path.join(golden_dir, "batcher.npz"))
    z = np.load(os.path.join(golden_dir, "epoch_iterator.npz"))
    nt = zb["case0.num_tokens"]
    mt, ms, mult = (int(v) for v in zb["case0.args"])
    batches = [b.tolist() for b in D.batch_by_size(np.arange(len(nt), dtype=np.int64), nt, mt, ms, mult)]
    assert len(batches) == int(z["n_batches"])

    class DS:
        def __getitem__(self, i):
            return int(i)

    def flat(seq):
        return (np.asarray([i for b in seq for i in (b or [])], dtype=np.int64),
                np.asarray([len(b) if b else 0 for b in seq], dtype=np.int64))

    state0 = np.random.get_state()[1].copy()
    for shards in (1, 3):
        for sid in range(shards):
            it = D.EpochBatchIterator(DS(), lambda items: list(items), batches, seed=3, num_shards=shards, shard_id=sid)
            for ep in (1, 2, 3):
                f, l = flat(list(it.next_epoch_itr(shuffle=True)))
                assert np.array_equal(f, z[f"s{shards}.{sid}.e{ep}.flat"]), (shards, sid, ep)
                assert np.array_equal(l, z[f"s{shards}.{sid}.e{ep}.lens"])
                assert it.end_of_epoch()
            st = it.state_dict()
            assert [st["epoch"], st["iterations_in_epoch"]] == z[f"s{shards}.{sid}.final_state"].tolist()
    assert np.array_equal(np.random.get_state()[1], state0)  # numpy_seed restores the global RNG
    it = D.EpochBatchIterator(DS(), lambda items: list(items), batches, seed=3, num_shards=3, shard_id=1)
    itr = it.next_epoch_itr(shuffle=True)
    next(itr), next(itr)
    st = it.state_dict()
    assert [st["epoch"], st["iterations_in_epoch"]] == z["resume.state"].tolist()
    it2 = D.EpochBatchIterator(DS(), lambda items: list(items), batches, seed=3, num_shards=3, shard_id=1)
    it2.load_state_dict(st)
    f, l = flat(list(it2.next_epoch_itr(shuffle=True)))
    assert np.array_equal(f, z["resume.flat"]) and np.array_equal(l, z["resume.lens"])


def test_task_batch_iterator_over_the_on_disk_set(corpus):
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import s2st_oracle as O
    from test_engine import NANO
    tasks = importlib.import_module(PKG + ".tasks")
    a = O.make_args(**NANO)
    a.data, a.config_yaml = corpus, "config.yaml"
    task = tasks.S2ST_TranslationTask.setup_task(a)
    ds = task.load_dataset("train_tiny")
    seen = []
    for sid in range(2):
        it = task.get_batch_iterator(ds, max_tokens=150, max_positions=task.max_positions(),
                                     required_batch_size_multiple=2, seed=5, num_shards=2, shard_id=sid)
        for s in it.next_epoch_itr(shuffle=True):
            if s:
                assert s["net_input"]["src_speech"].shape[0] * s["net_input"]["src_speech"].shape[1] <= 150
                seen += s["id"].tolist()
    assert sorted(seen) == list(range(len(ds)))
