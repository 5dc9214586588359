"""Stub of `bitarray` (absent in this container). Test infrastructure only."""


class bitarray:
    pass


class util:
    pass
