"""Name -> class registries and `build_from_cfg`, the contract of det3d/utils/registry.py:6-78 and
det3d/models/registry.py:3-14: a config dict `{"type": "<registered name>" | <class>, **kwargs}` is turned into an
instance, missing keyword arguments are filled from `default_args`.  Behaviour kept: `@REG.register_module` decorator,
KeyError for duplicate or unknown names, TypeError for a non-class / a `type` that is neither str nor class, and the
`_module_dict` / `module_dict` / `name` / `get` accessors that reference-side code (and INTEGRATION.md) uses."""
import inspect


class Registry:
    """A named table of classes."""

    def __init__(self, name):
        self._name = str(name)
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (type(self).__name__, self._name, sorted(self._module_dict))

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        """Registered class or None."""
        return self._module_dict.get(key)

    def register_module(self, cls):
        """Class decorator: registers `cls` under its own __name__ and returns it unchanged."""
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        key = cls.__name__
        if key in self._module_dict:
            raise KeyError("%s is already registered in %s" % (key, self._name))
        self._module_dict[key] = cls
        return cls


def _resolve(kind, registry):
    if inspect.isclass(kind):
        return kind
    if not isinstance(kind, str):
        raise TypeError("type must be a str or valid type, but got %s" % type(kind))
    found = registry.get(kind)
    if found is None:
        raise KeyError("%s is not in the %s registry" % (kind, registry.name))
    return found


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate `cfg["type"]` (a registered name or a class) with the remaining keys; `default_args` only fills keys
    the config does not set."""
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError("cfg must be a dict with a 'type' key")
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError("default_args must be a dict or None")
    kwargs = {k: v for k, v in cfg.items() if k != "type"}
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return _resolve(cfg["type"], registry)(**kwargs)


READERS, BACKBONES, NECKS, TRACK, SECOND_STAGE = (Registry(n) for n in ("reader", "backbone", "neck", "track", "second_stage"))
