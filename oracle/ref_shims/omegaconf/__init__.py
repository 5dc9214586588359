"""Stub of the `omegaconf` package: just enough surface for `import fairseq`
to succeed in a container where omegaconf is not installed.

TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py to import the
reference in this container).  Contains no reference code.
"""
from contextlib import contextmanager

MISSING = "???"


def II(x):
    return "${" + str(x) + "}"


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


class OmegaConf:
    @staticmethod
    def create(x=None, **kw):
        return DictConfig(x or {})

    @staticmethod
    def set_struct(cfg, flag):
        return None

    @staticmethod
    def is_config(x):
        return isinstance(x, (DictConfig, ListConfig))

    @staticmethod
    def to_container(x, **kw):
        return x

    @staticmethod
    def merge(*xs):
        out = DictConfig()
        for x in xs:
            out.update(x)
        return out


@contextmanager
def open_dict(cfg):
    yield cfg


class _Utils:
    @staticmethod
    def is_primitive_type(x):
        return isinstance(x, (int, float, str, bool, type(None)))


_utils = _Utils()
