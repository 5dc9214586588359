def ba2int(*a, **k):
    raise RuntimeError("bitarray stub")


def int2ba(*a, **k):
    raise RuntimeError("bitarray stub")
