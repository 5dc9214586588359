"""runtime/streams.py: the stream pool hands out streams that were MEASURED not to share a hardware queue with the streams
they run beside (a process gets four queues on this machine; two streams on one queue run one after the other)."""
import importlib

import pytest
import torch

PKG = "speech-to-speech-translation_amd"


@pytest.mark.gpu
def test_pool_streams_run_beside_each_other_and_the_caller():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    S = importlib.import_module(PKG + ".runtime.streams")
    dev = torch.device("cuda", 0)
    roles = ["test-role-a", "test-role-b", "test-role-c"]  # + the caller's stream: the four queues of a process
    got = [S.get(r, dev) for r in roles]
    assert S.get(roles[0], dev) is got[0]  # one stream per role, kept
    free = [r for r in roles if r not in S.collisions(dev)]
    # (other tests of this process may have taken queues already: what the pool reports free must BE free)
    cur = torch.cuda.current_stream(dev)
    for r, st in zip(roles, got):
        if r in free:
            assert not S.shares_queue(st, cur), r
    for i in range(len(roles)):
        for j in range(i + 1, len(roles)):
            if roles[i] in free and roles[j] in free:
                assert not S.shares_queue(got[i], got[j]), (roles[i], roles[j])
    # a role that may share says so: no collision is recorded for it whatever queue it lands on
    extra = S.get("test-role-d", dev, may_share=tuple(roles))
    assert isinstance(extra, torch.cuda.Stream)
    # and the probe itself: a stream shares a queue with itself
    assert S.shares_queue(got[0], got[0])


def test_default_decode_chains_env(monkeypatch):
    S = importlib.import_module(PKG + ".runtime.streams")
    monkeypatch.delenv("S2ST_DECODE_CHAINS", raising=False)
    assert S.default_decode_chains() == 3
    monkeypatch.setenv("S2ST_DECODE_CHAINS", "1")
    assert S.default_decode_chains() == 1
    monkeypatch.setenv("S2ST_DECODE_CHAINS", "0")
    assert S.default_decode_chains() == 1
