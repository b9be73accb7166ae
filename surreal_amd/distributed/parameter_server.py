"""
Parameter path with the reference's protocol (surreal/distributed/parameter_server.py:20-303):

    learner  --publish(binary, info)-->  ParameterServer  <--request--  ParameterClient (agent)

``info = {time, iteration, message, hash}`` with ``hash = binary_hash(binary)``; requests are the
strings ``'info'``, ``'parameter'`` and ``'parameter:<last hash>'`` -- the last one gets
``(None, info)`` back when nothing changed, which is what keeps idle agents from re-downloading.

Transport is callables: ``ParameterPublisher(publish_fn=server.set_storage)`` and
``ParameterClient(request_fn=server.handle_request)`` wire the three objects together in one process;
a socket layer (the reference's is ZeroMQ through caraml, absent here) goes in the same two places
with ``serializer.serialize`` / ``deserialize`` around it.
"""
import time

from surreal_amd.utils import serializer as S
from .module_dict import ModuleDict


class ParameterPublisher(object):
    def __init__(self, publish_fn, module_dict):
        self._publish_fn = publish_fn
        self._module_dict = module_dict if isinstance(module_dict, ModuleDict) else ModuleDict(module_dict)

    def publish(self, iteration, message=''):
        binary = self._module_dict.dumps()
        info = {'time': time.time(), 'iteration': iteration, 'message': message,
                'hash': S.binary_hash(binary)}
        self._publish_fn((binary, info))
        return info


class ParameterServer(object):
    """storage + the request handler of parameter_server.py:175-209 (no process, no socket)"""

    def __init__(self):
        self.parameters = None
        self.param_info = None

    def set_storage(self, data):
        self.parameters, self.param_info = data

    def handle_request(self, request):
        if request == 'info':
            return None, self.param_info
        if isinstance(request, str) and request.startswith('parameter'):
            if self.parameters is None:
                return None, None
            if ':' in request:
                _, last_hash = request.split(':', 1)
                if last_hash == self.param_info['hash']:       # parameters did not change
                    return None, self.param_info
            return self.parameters, self.param_info
        raise ValueError('invalid request: ' + str(request))


class ParameterClient(object):
    def __init__(self, request_fn):
        """request_fn(request_str) -> (binary or None, info or None); raises TimeoutError when the
        server cannot be reached"""
        self._request_fn = request_fn
        self._last_hash = ''
        self.alive = False

    def fetch_parameter_with_info(self, force_update=False):
        try:
            response = self._request_fn('parameter' if force_update else 'parameter:' + self._last_hash)
        except TimeoutError:
            self.alive = False
            return None, None
        self.alive = True
        param, info = response
        if info is None:
            return None, None
        self._last_hash = info['hash']
        return param, info

    def fetch_info(self):
        try:
            _, info = self._request_fn('info')
        except TimeoutError:
            self.alive = False
            return None
        self.alive = True
        return info
