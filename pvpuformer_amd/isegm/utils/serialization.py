"""Constructor-argument capture and model re-creation: the whole public surface of the reference's
isegm/utils/serialization.py:7-112 (``serialize``, ``load_model``, ``get_config_repr``, ``get_default_params``,
``get_classname``, ``get_class_from_str``); checkpoint ``config`` dicts written by either side load on the other."""
import inspect
from copy import deepcopy
from functools import wraps
from importlib import import_module


def get_classname(cls):
    module = cls.__module__
    # checkpoints must name the reference path so that both implementations can load them
    if module.startswith("pvpuformer_amd.isegm"):
        module = module[len("pvpuformer_amd."):]
    return f"{module}.{cls.__qualname__}"


def get_default_params(cls):
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
        for n, p in get_default_params(self.__class__).items():
            given.setdefault(n, p.default)
        cfg = {"class": get_classname(self.__class__), "params": {}}
        for n, v in given.items():
            kind = "builtin"
            if inspect.isclass(v):
                kind, v = "class", get_classname(v)
            cfg["params"][n] = {"type": kind, "value": v, "specified": n in specified}
        self._config = cfg
        init(self, *args, **kwargs)
    return wrapped


def get_class_from_str(path):
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
    cls = get_class_from_str(config["class"])
    defaults = get_default_params(cls)
    args = {}
    for n, p in config["params"].items():
        v = p["value"]
        if p["type"] == "class":
            v = get_class_from_str(v)
        if n not in defaults and not p["specified"]:
            continue
        if n in defaults and not p["specified"] and defaults[n].default == v:
            continue
        args[n] = v
    args.update(kwargs)
    if eval_ritm:                      # serialization.py:66-69: RITM checkpoints predate the flag (the VPU model rejects it)
        args["use_rgb_conv"] = True
    return cls(**args)


def get_config_repr(config):
    """Printable summary of a ``_config`` capture (serialization.py:72-83): one ``name = value`` line per constructor
    argument, class-valued arguments by their bare name, unspecified ones marked ``(default)``."""
    lines = [f'Model: {config["class"]}']
    for name, p in config['params'].items():
        value = p['value'].split('.')[-1] if p['type'] == 'class' else p['value']
        lines.append(f'{name:<22} = {str(value):<12}' + ('' if p['specified'] else ' (default)'))
    return '\n'.join(lines) + '\n'
