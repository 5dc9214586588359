"""Plugin registry mirroring fairseq's decorators (fairseq/tasks/__init__.py:63-98,
fairseq/models/__init__.py:137-207, fairseq/criterions/__init__.py + fairseq/registry.py).

When fairseq is importable the classes are registered there too, under the reference's
names (task ``s2s_translation``, model/arch ``s2st_transformer``, criterion ``s2st_loss``), so
``fairseq_cli.train --user-dir <this package>`` resolves them; without fairseq (this image)
the local tables below are what ``s2st_amd`` 's own trainer and tests use.
"""
TASKS, MODELS, ARCHS, CRITERIA = {}, {}, {}, {}


def _fairseq(kind):
    try:  # pragma: no cover - fairseq is absent in the build image
        import fairseq.tasks as ft
        import fairseq.models as fm
        import fairseq.criterions as fc
        return {"task": ft.register_task, "model": fm.register_model,
                "arch": fm.register_model_architecture, "criterion": fc.register_criterion}[kind]
    except Exception:
        return None


def register_task(name):
    def deco(cls):
        TASKS[name] = cls
        f = _fairseq("task")
        if f is not None:  # pragma: no cover
            try:
                f(name)(cls)
            except Exception:
                pass
        return cls
    return deco


def register_model(name):
    def deco(cls):
        MODELS[name] = cls
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCHS[arch_name] = (model_name, fn)
        return fn
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        CRITERIA[name] = cls
        return cls
    return deco
