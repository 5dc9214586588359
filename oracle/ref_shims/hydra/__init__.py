"""Stub of `hydra` (absent in this container). Test infrastructure only."""


def main(*a, **k):
    def deco(f):
        return f
    return deco
