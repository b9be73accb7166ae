"""
Parameter blobs (surreal/distributed/module_dict.py:8-63): every module's ``state_dict()`` as numpy
float32 arrays, ``{module_name: {parameter_name: ndarray}}``, serialised in one piece.

Parameter names are this package's canonical ones (``actor.fc1.W`` ...).  The reference's names come
from torchx's layer classes, whose source is absent (unpinned); ``key_map``
translates at the boundary once they are known.
"""
import collections

import numpy as np
import torch

from surreal_amd.utils import serializer as S


class ModuleDict(object):
    def __init__(self, module_dict, key_map=None):
        if not isinstance(module_dict, dict):
            raise TypeError('module_dict must be a dict')
        for k, m in module_dict.items():
            if not isinstance(k, str):
                raise TypeError('Key "{}" must be string.'.format(k))
            if not (hasattr(m, 'state_dict') and hasattr(m, 'load_state_dict')):
                raise TypeError('"{}" must provide state_dict / load_state_dict.'.format(m))
        self._module_dict = module_dict
        self._to_wire = dict(key_map or {})
        self._from_wire = {v: k for k, v in self._to_wire.items()}

    def numpy_dict(self):
        out = {}
        for k, m in self._module_dict.items():
            sd = collections.OrderedDict()
            for key, value in m.state_dict().items():
                arr = value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value)
                sd[self._to_wire.get(key, key)] = arr
            out[k] = sd
        return out

    def dumps(self):
        return S.serialize(self.numpy_dict())

    def loads(self, binary):
        self.load(S.deserialize(binary))

    def load(self, numpy_dict):
        for k, m in self._module_dict.items():
            sd = collections.OrderedDict()
            for key, value in numpy_dict[k].items():
                sd[self._from_wire.get(key, key)] = np.asarray(value, dtype=np.float32)
            m.load_state_dict(sd)
