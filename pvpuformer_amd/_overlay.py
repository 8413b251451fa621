"""Name fall-through for mirror modules that shadow a module of the ``isegm`` tree the overlay sits on.

``pvpuformer_amd.install()`` registers every mirror module under its ``isegm.*`` name, so a mirror file hides the other
tree's file of the same name.  A mirror module offers the whole public surface of the file it shadows for the hot path
(``tests/test_overlay_reference_cpu.py`` checks that against the reference), and for everything else -- names added by a
newer reference, RITM-era helpers -- ``attach()`` gives the module a PEP-562 ``__getattr__``: the first lookup of a name the
mirror does not define executes the shadowed file (once, under the same dotted name and ``__package__``, so its relative
imports land in the overlay package) and serves the name from there.  Classes defined by the shadowed file therefore
pickle by a path that resolves again through the same fall-through."""
import importlib
import importlib.util
import os


def shadowed_file(other_root, rel):
    """``<other>/<a>/<b>.py`` or ``<other>/<a>/<b>/__init__.py`` for the mirror module ``a.b`` -- or None."""
    base = os.path.join(other_root, *rel.split("."))
    for cand, is_pkg in ((base + ".py", False), (os.path.join(base, "__init__.py"), True)):
        if os.path.isfile(cand):
            return cand, is_pkg
    return None, False


def attach(mod, fullname, path, is_pkg):
    state = {"module": None, "loading": False}

    def load():
        spec = importlib.util.spec_from_file_location(
            fullname, path, submodule_search_locations=list(mod.__path__) if is_pkg else None)
        shadow = importlib.util.module_from_spec(spec)
        state["loading"] = True
        try:
            spec.loader.exec_module(shadow)
        finally:
            state["loading"] = False
        return shadow

    def __getattr__(name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if is_pkg:                                   # `from package import submodule` asks the package first
            try:
                return importlib.import_module(fullname + "." + name)
            except ModuleNotFoundError as e:
                if e.name != fullname + "." + name:
                    raise
        if state["loading"]:                         # the shadowed file asks the mirror for a name while it is executing
            raise AttributeError(f"module {fullname!r} has no attribute {name!r}")
        if state["module"] is None:
            try:
                state["module"] = load()
            except Exception as e:
                raise AttributeError(f"module {fullname!r} (MI355X mirror) has no attribute {name!r}, and the file it "
                                     f"shadows, {path}, failed to load: {type(e).__name__}: {e}") from e
        try:
            return getattr(state["module"], name)
        except AttributeError:
            raise AttributeError(f"module {fullname!r} has no attribute {name!r} (neither the MI355X mirror nor {path})") from None

    mod.__getattr__ = __getattr__
    mod.__vpu_shadows__ = path
