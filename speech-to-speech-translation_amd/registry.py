"""Plugin registry mirroring fairseq's decorators (fairseq/tasks/__init__.py:63-98,
fairseq/models/__init__.py:137-207, fairseq/criterions/__init__.py + fairseq/registry.py:40-100).

When fairseq is importable, the four decorators below forward to fairseq's own -- task ``s2s_translation``, model and
architecture ``s2st_transformer``, criterion ``s2st_loss`` -- and the classes extend fairseq's bases
(``LegacyFairseqTask`` / ``BaseFairseqModel`` / ``FairseqCriterion``: fairseq's registries reject anything else), so
``fairseq_cli.train --user-dir <this package> --task s2s_translation --arch s2st_transformer --criterion s2st_loss``
resolves them through ``utils.import_user_module`` (fairseq/utils.py:462-507) exactly as it resolves
``examples/s2s_trans``.  A registration fairseq refuses (duplicate name: the reference's own plugin is loaded too) is
reported once on stderr, never swallowed silently.  Without fairseq (the GPU image) the local tables are what the
package's own ``train.py`` / trainer use, and the bases are plain ``object`` / ``torch.nn.Module``.
"""
import sys

import torch

TASKS, MODELS, ARCHS, CRITERIA = {}, {}, {}, {}
FAIRSEQ_REGISTERED = {"task": [], "model": [], "arch": [], "criterion": []}

try:  # fairseq is absent in the GPU image; present (with stub omegaconf / hydra) in the build container's tests
    from fairseq.tasks import LegacyFairseqTask as _FsTask, register_task as _fs_register_task
    from fairseq.models import (BaseFairseqModel as _FsModel, register_model as _fs_register_model,
                                register_model_architecture as _fs_register_arch)
    from fairseq.criterions import FairseqCriterion as _FsCriterion, register_criterion as _fs_register_criterion
    HAVE_FAIRSEQ = True
except Exception:  # ImportError, or a half-importable fairseq (missing omegaconf ...)
    HAVE_FAIRSEQ = False
    _FsTask, _FsModel, _FsCriterion = object, torch.nn.Module, torch.nn.Module

TaskBase, ModelBase, CriterionBase = _FsTask, _FsModel, _FsCriterion


def _forward(kind, name, fn):
    if not HAVE_FAIRSEQ:
        return
    try:
        fn()
        FAIRSEQ_REGISTERED[kind].append(name)
    except ValueError as e:  # "Cannot register duplicate ..."
        print(f"[s2st_amd] fairseq refused {kind} registration of {name!r}: {e}", file=sys.stderr)


def register_task(name):
    def deco(cls):
        TASKS[name] = cls
        _forward("task", name, lambda: _fs_register_task(name)(cls))
        return cls
    return deco


def register_model(name):
    def deco(cls):
        MODELS[name] = cls
        _forward("model", name, lambda: _fs_register_model(name)(cls))
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCHS[arch_name] = (model_name, fn)
        _forward("arch", arch_name, lambda: _fs_register_arch(model_name, arch_name)(fn))
        return fn
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        CRITERIA[name] = cls
        _forward("criterion", name, lambda: _fs_register_criterion(name)(cls))
        return cls
    return deco
