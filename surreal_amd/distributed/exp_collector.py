"""
Replay side of the experience wire format (surreal/distributed/exp_collector.py:37-65): a chunk is
``(exp_list, ob_storage)``; every dict key that ends in ``_hash`` is replaced by the stored object
(suffix removed).  Objects already seen are reused through a weak-value map, so the overlapping
windows of one trajectory share their observation arrays in the replay as well.
"""
import weakref

from surreal_amd.utils import serializer as S


class ExperienceCollector(object):
    def __init__(self, exp_handler):
        """exp_handler(exp): called once per experience, e.g. ``replay._insert_wrapper``"""
        self._exp_handler = exp_handler
        # hash -> the stored object itself, weakly (as exp_collector.py:40-41): an observation stays
        # shared across chunks for exactly as long as some experience in the replay still holds it
        self._weakref_map = weakref.WeakValueDictionary()

    def recv(self, binary):
        exp, storage = S.deserialize(binary)
        experience_list = self._retrieve_storage(exp, storage)
        for e in experience_list:
            self._exp_handler(e)
        return len(experience_list)

    def _lookup(self, key, storage):
        obj = self._weakref_map.get(key)
        if obj is None:
            obj = storage[key]
            try:
                self._weakref_map[key] = obj
            except TypeError:         # not weakly referenceable (a list of frames, a scalar): shared
                pass                  # within this chunk through `storage`, not across chunks
        return obj

    def _retrieve_storage(self, exp, storage):
        """exp_collector.py:44-65: recurse through lists / dicts, strip the `_hash` suffix from
        keys, replace every hash string by its stored object.  (The reference looks EVERY string
        up and raises KeyError on a plain one; here a string that is not a known hash is kept.)"""
        if isinstance(exp, list):
            return [self._retrieve_storage(e, storage) for e in exp]
        if isinstance(exp, tuple):
            return tuple(self._retrieve_storage(e, storage) for e in exp)
        if isinstance(exp, dict):
            out = {}
            for key, value in exp.items():
                new_key = key[:-len('_hash')] if isinstance(key, str) and key.endswith('_hash') else key
                out[new_key] = self._retrieve_storage(value, storage)
            return out
        if isinstance(exp, str):
            if exp not in storage and exp not in self._weakref_map:
                return exp
            return self._lookup(exp, storage)
        return exp
