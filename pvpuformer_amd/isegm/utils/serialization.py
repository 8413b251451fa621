"""Constructor-argument capture and model re-creation, API-compatible with the reference's
isegm/utils/serialization.py:7-112 (checkpoint ``config`` dicts written by either side load on the other)."""
import inspect
from copy import deepcopy
from functools import wraps
from importlib import import_module


def _classname(cls):
    module = cls.__module__
    # checkpoints must name the reference path so that both implementations can load them
    if module.startswith("pvpuformer_amd.isegm"):
        module = module[len("pvpuformer_amd."):]
    return f"{module}.{cls.__qualname__}"


def _default_params(cls):
    params = {}
    for klass in cls.mro():
        if klass.__module__ == "builtins" or "__init__" not in klass.__dict__:
            continue
        for name, p in inspect.signature(klass.__init__).parameters.items():
            if p.default is not p.empty and name not in params:
                params[name] = p
    return params


def serialize(init):
    names = list(inspect.signature(init).parameters)

    @wraps(init)
    def wrapped(self, *args, **kwargs):
        given = deepcopy(kwargs)
        for n, v in zip(names[1:], args):
            given[n] = v
        specified = set(given)
        for n, p in _default_params(self.__class__).items():
            given.setdefault(n, p.default)
        cfg = {"class": _classname(self.__class__), "params": {}}
        for n, v in given.items():
            kind = "builtin"
            if inspect.isclass(v):
                kind, v = "class", _classname(v)
            cfg["params"][n] = {"type": kind, "value": v, "specified": n in specified}
        self._config = cfg
        init(self, *args, **kwargs)
    return wrapped


def _class_from_str(path):
    parts = path.split(".")
    for split in range(len(parts) - 1, 0, -1):
        try:
            obj = import_module(".".join(parts[:split]))
        except ImportError:
            continue
        for a in parts[split:]:
            obj = getattr(obj, a)
        return obj
    raise ImportError(path)


def load_model(config, eval_ritm=False, **kwargs):
    cls = _class_from_str(config["class"])
    defaults = _default_params(cls)
    args = {}
    for n, p in config["params"].items():
        v = p["value"]
        if p["type"] == "class":
            v = _class_from_str(v)
        if n not in defaults and not p["specified"]:
            continue
        if n in defaults and not p["specified"] and defaults[n].default == v:
            continue
        args[n] = v
    args.update(kwargs)
    return cls(**args)
