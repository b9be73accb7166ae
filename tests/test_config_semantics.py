"""CPU tier: Config / extend_config semantics (surreal/session/config.py:154-255) on the scenarios the
reference's own known-answer tests pin (test-old/test_config.py): placeholders for required values
('_dict_', '_list_', '_singleton_', '_str_', '_num_', '_int_', '_float_', '_bool_', '_object_',
'_enum[a, b]_'), recursive defaults, attribute access, reserved names, json / yaml round trips.
The same module run against the reference's own Config (under oracle/ref_shims.py) passes every case
except the yaml round trip, which the reference fails with today's PyYAML (yaml.load without a Loader)."""
import copy

import pytest

from surreal_amd.session import Config, ConfigError, extend_config

DEFAULTS = {
    'redis': {'replay': {'host': 'localhost', 'port': 6379},
              'ps': {'host': '_dict_', 'port': '_list_', 'single': '_singleton_'}},
    'log': {'files': ['f1.txt', 'f2.txt', 'f3.txt'],
            'outputs': [{'stdout1': 1, 'stdout2': 2}, {'stderr1': 10, 'stderr2': 20}]},
}
GOOD_PS = {'host': {'s': 2}, 'port': [1, 2], 'single': 'one-value'}
MERGED = {'log': DEFAULTS['log'], 'redis': {'replay': {'host': 'localhost', 'port': 6379}, 'ps': GOOD_PS}}
TYPED = {'redis': {'replay': {'host': '_str_', 'port': '_num_', 'catchall': '_object_'},
                   'ps': {'fport': '_float_', 'iport': '_int_', 'flag': '_bool_'}}}


def test_attribute_access_and_missing_key():
    C = Config(copy.deepcopy(DEFAULTS))
    assert C.redis.replay.host == 'localhost' and C.log.files[1] == 'f2.txt' and C.log.outputs[1].stderr2 == 20
    with pytest.raises(ConfigError):
        C.redis.ps.badkey


def test_defaults_fill_in_both_ways():
    user = {'redis': {'ps': copy.deepcopy(GOOD_PS)}}
    assert extend_config(copy.deepcopy(user), Config(copy.deepcopy(DEFAULTS))) == MERGED
    C2 = Config(copy.deepcopy(user))
    C2.extend(Config(copy.deepcopy(DEFAULTS)))
    assert C2 == C2.to_dict() == MERGED


@pytest.mark.parametrize('bad_redis', [
    {'ps': {'host': 3, 'port': [1, 2]}},                                         # '_dict_' got a scalar
    {'ps': {'host': {'s': 2}, 'port': {'t': 'x'}}},                              # '_list_' got a dict
    {'ps': {'host': {'s': 2}, 'port': [1, 2], 'single': {}}},                    # '_singleton_' got a dict
    {'ps': dict(GOOD_PS), 'replay': 'wrong single value'},                       # a dict default got a scalar
    {'ps': dict(GOOD_PS), 'replay': {'host': {}}},                               # a scalar default got a dict
    {'ps': {'port': [1, 2], 'single': 'one-value'}},                             # a required value is missing
])
def test_required_placeholders_reject(bad_redis):
    with pytest.raises(ConfigError):
        extend_config({'redis': copy.deepcopy(bad_redis)}, Config(copy.deepcopy(DEFAULTS)))


def typed(host='localhost', port=123, fport=1.23, iport=10, flag=False):
    return {'redis': {'replay': {'host': host, 'port': port, 'catchall': None},
                      'ps': {'fport': fport, 'iport': iport, 'flag': flag}}}


def test_typed_placeholders():
    ok = typed(port=13.23, fport=1.2e4)
    assert extend_config(copy.deepcopy(ok), Config(copy.deepcopy(TYPED))) == ok
    for bad in (typed(host=3), typed(iport=10.78)):
        with pytest.raises(ConfigError):
            extend_config(bad, Config(copy.deepcopy(TYPED)))


def test_enum_placeholder_and_subtree_extend():
    default = {'redis': {'replay': {'type': '_enum[uniform, priority, fifo]_'}}}
    user = {'redis': {'replay': {'type': 'fifo'}}}
    assert extend_config(copy.deepcopy(user), default) == user
    with pytest.raises(ConfigError):
        extend_config({'redis': {'replay': {'type': 'lifo'}}}, default)
    C = Config(copy.deepcopy(user))
    C.redis.replay.extend({'type': '_enum[uniform, priority, fifo]_', 'other': 'should be added'})
    assert C.redis.replay.type == 'fifo' and C.redis.replay.other == 'should be added'


def test_reserved_names_cannot_be_overridden():
    C = Config({'a': 3, 'b': 4})
    with pytest.raises(ConfigError):
        C.extend = 10
    with pytest.raises(ConfigError):
        C['keys'] = 10
    with pytest.raises(ConfigError):
        Config({'items': 1})
    with pytest.raises(ConfigError):                      # also deep inside lists of dicts
        Config({'a': {'d': [{'b': {'items': 100}}]}, 'c': 10})


@pytest.mark.parametrize('ext', ['json', 'yaml'])
def test_file_round_trip(tmp_path, ext):
    C = Config(copy.deepcopy(DEFAULTS))
    path = str(tmp_path / ('cfg.' + ext))
    C.dump_file(path)
    assert Config.load_file(path) == C
