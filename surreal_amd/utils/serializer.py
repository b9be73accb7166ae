"""
Object (de)serialisation and content hashes shared by the parameter path and the experience wire
format (surreal/utils/serializer.py:11-70).

The reference's default serialiser is ``pyarrow.serialize`` -- an API pyarrow removed in 2.0; its
byte format cannot be produced (or read) by any pyarrow that still installs.  The reference's own
escape hatch is ``set_global_serializer`` (:26-33, with ``pickle.dumps / pickle.loads`` named in the
source as the alternative), so pickle is the default here and both ends must agree on it.
The content hash is the reference's: base64(md5(binary))[:16] (:55-66).
"""
import base64
import hashlib
import pickle

_SERIALIZER = pickle.dumps
_DESERIALIZER = pickle.loads


def set_global_serializer(serializer, deserializer):
    """call at the start of a script, on both ends of a connection (serializer.py:26-33)"""
    assert callable(serializer) and callable(deserializer)
    global _SERIALIZER, _DESERIALIZER
    _SERIALIZER = serializer
    _DESERIALIZER = deserializer


def serialize(obj):
    return _SERIALIZER(obj)


def deserialize(binary):
    return _DESERIALIZER(binary)


def binary_hash(binary):
    """16-character content key (serializer.py:55-66)"""
    return base64.b64encode(hashlib.md5(binary).digest())[:16].decode('utf-8')


def string_hash(s):
    assert isinstance(s, str)
    return binary_hash(s.encode('utf-8'))


def pyobj_hash(obj):
    return binary_hash(serialize(obj))
