"""
Agent side of the experience wire format (surreal/distributed/exp_sender.py:10-98).

An experience is split into a *hashed* part -- the observations, which overlapping n-step windows
repeat many times -- and a plain part.  Every distinct observation object travels once per flushed
chunk, keyed by its content hash; experiences carry the hashes (``obs`` -> ``obs_hash``).  A chunk is
``serialize((exp_list, ob_storage))``.

The transport is a callable (``send_fn(binary)``): the reference pushes chunks through a ZeroMQ
socket (caraml, absent here); anything that moves bytes will do, including the in-process
``ExperienceCollector.recv``.
"""
from surreal_amd.utils import serializer as S

# which fields the reference's wrappers put into hash_dict (exp_sender_wrapper.py:59-61, 250-253)
HASHED_KEYS = ('obs', 'obs_next')


class ExpBuffer(object):
    """temporarily holds and de-duplicates experience (exp_sender.py:10-59)"""

    def __init__(self):
        self.exp_list = []
        self.ob_storage = {}

    def add(self, hash_dict, nonhash_dict):
        if not isinstance(hash_dict, dict) or not isinstance(nonhash_dict, dict):
            raise TypeError('hash_dict and nonhash_dict must be dicts')
        exp = {}
        for key, values in hash_dict.items():
            assert not key.endswith('_hash'), 'do not manually append `_hash`'
            exp[key + '_hash'] = self._hash_nested(values)
        exp.update(nonhash_dict)
        self.exp_list.append(exp)

    def flush(self):
        binary = S.serialize((self.exp_list, self.ob_storage))
        self.exp_list = []
        self.ob_storage = {}
        return binary

    def _hash_nested(self, values):
        if isinstance(values, list):
            return [self._hash_nested(v) for v in values]
        if isinstance(values, tuple):
            return tuple(self._hash_nested(v) for v in values)
        if isinstance(values, dict):
            return {k: self._hash_nested(v) for k, v in values.items()}
        if values is None:
            return None
        hsh = S.pyobj_hash(values)
        if hsh not in self.ob_storage:
            self.ob_storage[hsh] = values
        return hsh


class ExpSender(object):
    """exp_sender.py:62-98 with the socket replaced by ``send_fn``"""

    def __init__(self, *, send_fn, flush_iteration):
        if not isinstance(flush_iteration, int) or flush_iteration < 1:
            raise ValueError('flush_iteration must be a positive int')
        self._send_fn = send_fn
        self._exp_buffer = ExpBuffer()
        self._flush_iteration = flush_iteration
        self._count = 0

    def send(self, hash_dict, nonhash_dict):
        """returns the chunk's content hash when this call flushed, else None"""
        self._exp_buffer.add(hash_dict=hash_dict, nonhash_dict=nonhash_dict)
        self._count += 1
        if self._count % self._flush_iteration == 0:
            binary = self._exp_buffer.flush()
            self._send_fn(binary)
            return S.binary_hash(binary)
        return None

    def send_exp(self, exp):
        """an experience dict as the windowing wrappers emit it (the ``sink`` signature)"""
        hashed = {k: exp[k] for k in HASHED_KEYS if k in exp}
        plain = {k: v for k, v in exp.items() if k not in hashed}
        return self.send(hashed, plain)
