"""
Attribute-access configuration tree with typed "required" placeholders.

Mirrors the behaviour (not the code) of the reference's ``surreal.session.config``
(surreal/session/config.py:154-255): the three nested configs handed to every
Agent / Learner / Replay plugin are ``Config`` objects, filled from defaults by
``extend()``, where a default of ``'_int_'``, ``'_float_'``, ``'_num_'``, ``'_str_'``,
``'_bool_'``, ``'_list_'``, ``'_dict_'``, ``'_singleton_'``, ``'_object_'`` or
``'_enum[a, b]_'`` marks a key the user must supply with that type (config.py:24-53).
"""
import json
import re

_ENUM = re.compile(r'_enum\[(.*)\]_')
_RESERVED = frozenset(['keys', 'items', 'values', 'get', 'copy', 'update', 'extend',
                       'load_file', 'dump_file', 'to_dict'])

_TYPE_TESTS = {
    '_object_': (lambda v: True, 'filled'),
    '_singleton_': (lambda v: not isinstance(v, (list, dict)), 'a singleton (non-list/dict)'),
    '_list_': (lambda v: isinstance(v, list), 'a list'),
    '_dict_': (lambda v: isinstance(v, dict), 'a dict'),
    '_int_': (lambda v: isinstance(v, int), 'an integer'),
    '_float_': (lambda v: isinstance(v, float), 'a float'),
    '_num_': (lambda v: isinstance(v, (int, float)), 'a numeric value'),
    '_str_': (lambda v: isinstance(v, str), 'a string'),
    '_bool_': (lambda v: isinstance(v, bool), 'a boolean'),
}


class ConfigError(Exception):
    pass


def _placeholder(value):
    """-> (test, description) when `value` is a requirement marker, else None"""
    if not isinstance(value, str):
        return None
    low = value.lower()
    if low in _TYPE_TESTS:
        return _TYPE_TESTS[low]
    m = _ENUM.match(low)
    if m:
        options = [o.strip() for o in m.group(1).split(',') if o.strip()]
        if not options:
            raise ConfigError('_enum[...]_ cannot be empty')
        return (lambda v: v in options), 'an enum in [%s]' % m.group(1)
    return None


def _contains_placeholder(tree):
    return any(_placeholder(v) is not None or (isinstance(v, dict) and _contains_placeholder(v))
               for v in tree.values())


def _merge_defaults(cfg, defaults, path):
    for key, dflt in defaults.items():
        where = 'key "%s"' % '.'.join(path + [key])
        req = _placeholder(dflt)
        if key not in cfg:
            if req is not None:
                raise ConfigError('Required entry missing: %s must be %s.' % (where, req[1]))
            if isinstance(dflt, dict) and _contains_placeholder(dflt):
                raise ConfigError('Sub-dict under %s contains a required config: %s.'
                                  % (where, dflt))
            cfg[key] = dflt
            continue
        value = cfg[key]
        if req is not None:
            if _placeholder(value) is not None:       # still a placeholder after extension
                if value != dflt:
                    raise ConfigError('inherited %s: "%s" must match default "%s"'
                                      % (where, value, dflt))
            elif not req[0](value):
                raise ConfigError('Wrong type: %s must be %s.' % (where, req[1]))
        elif isinstance(dflt, dict):
            if not isinstance(value, dict):
                raise ConfigError(where + ' must have a sub-dict instead of a singleton')
            cfg[key] = _merge_defaults(value, dflt, path + [key])
        elif isinstance(value, dict):
            raise ConfigError(where + ' must be a singleton instead of a sub-dict')
    return cfg


class Config(dict):
    """dict whose keys are also attributes; nested dicts become Config recursively"""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        src = dict(d or {})
        src.update(kwargs)
        for k, v in src.items():
            self[k] = v

    def _wrap(self, value):
        if isinstance(value, (list, tuple)):
            return [Config(x) if isinstance(x, dict) and not isinstance(x, Config) else x
                    for x in value]
        if isinstance(value, dict) and not isinstance(value, Config):
            return Config(value)
        return value

    def __setitem__(self, name, value):
        if name in _RESERVED:
            raise ConfigError('"%s()" is a reserved method, cannot override' % name)
        super().__setitem__(name, self._wrap(value))

    def __setattr__(self, name, value):
        self[name] = value

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            if name.startswith('__') and name.endswith('__'):
                # protocol look-ups (pickle's __getstate__ / __reduce_ex__, copy's __deepcopy__): not config keys --
                # a Config travels to worker processes as the dict it is
                raise AttributeError(name)
            raise ConfigError('config key "%s" missing.' % name)

    def update(self, other=None, **kw):
        for k, v in dict(other or {}, **kw).items():
            self[k] = v

    def to_dict(self):
        def unwrap(v):
            if isinstance(v, Config):
                return v.to_dict()
            if isinstance(v, (list, tuple)):
                return type(v)(unwrap(x) for x in v)
            return v
        return {k: unwrap(v) for k, v in self.items()}

    def copy(self):
        return Config(self.to_dict())

    def extend(self, default_config):
        if not isinstance(default_config, dict):
            raise TypeError('default_config must be a dict')
        return _merge_defaults(self, default_config, [])

    @classmethod
    def load_file(cls, path):
        with open(path, 'r') as fp:
            if path.endswith('.json'):
                return cls(json.load(fp))
            import yaml
            return cls(yaml.safe_load(fp))

    def dump_file(self, path):
        with open(path, 'w') as fp:
            if path.endswith('.json'):
                json.dump(self.to_dict(), fp, indent=4)
            else:
                import yaml
                yaml.safe_dump(self.to_dict(), fp, indent=4, default_flow_style=False)


def extend_config(config, default_config):
    """`config` completed with `default_config`; raises ConfigError on unmet requirements"""
    return Config(_merge_defaults(Config(config), default_config, []))
