#!/usr/bin/env python
"""
TEST INFRASTRUCTURE.  Fixtures for the SURVEY.md section 8(f) formats, recorded from the reference's
own code run under oracle/ref_shims.py (tests/golden/wire/):

  exp_chunk.pkl        one flushed ExpBuffer chunk (surreal/distributed/exp_sender.py:10-59) of PPO
                       window experiences + the experiences ExperienceCollectorServer._retrieve_storage
                       (exp_collector.py:37-65) makes of it
  param_protocol.json  ParameterServer._handle_agent_request replies (parameter_server.py:175-209)
                       and U.binary_hash values (utils/serializer.py:55-66)
  module_dict.pkl      ModuleDict.dumps() of a small module (module_dict.py:21-35)
  ckpt/                a checkpoint folder written by surreal.utils.checkpoint.Checkpoint.save
                       (three saves, keep_history=2, keep_best=1)

The reference's default serialiser (pyarrow.serialize) no longer exists in pyarrow; the generator
installs pickle through the reference's own ``set_global_serializer`` hook (serializer.py:26-33).
It also checks the other direction here, where the reference is importable: a chunk / checkpoint
written by surreal_amd is consumed by the reference's code.
"""
import collections
import copy
import json
import os
import pickle
import shutil
import sys
import tempfile
import weakref

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shims  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'wire')


def make_experiences(n=5, n_step=3, D=4, A=2, seed=0):
    """what ExpSenderWrapperMultiStepMovingWindowWithInfo.send builds (exp_sender_wrapper.py:230-264):
    overlapping windows share observation objects"""
    rs = np.random.RandomState(seed)
    obs_seq = [collections.OrderedDict(low_dim=collections.OrderedDict(
        flat_inputs=rs.randn(D).astype(np.float32))) for _ in range(n + n_step)]
    exps = []
    for i in range(n):
        exps.append({
            'obs': [obs_seq[i + k] for k in range(n_step)],
            'obs_next': obs_seq[i + n_step],
            'actions': [rs.randn(A).astype(np.float32) for _ in range(n_step)],
            'onetime_infos': [],
            'persistent_infos': [[rs.randn(2 * A).astype(np.float32)] for _ in range(n_step)],
            'rewards': [float(rs.randn()) for _ in range(n_step)],
            'dones': [False] * (n_step - 1) + [bool(i == n - 1)],
            'infos': [{} for _ in range(n_step)],
            'n_step': n_step,
        })
    return exps


class TinyModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(3, 2)
        with torch.no_grad():
            self.fc.weight.copy_(torch.arange(6.0).view(2, 3) / 10)
            self.fc.bias.copy_(torch.tensor([0.5, -0.5]))


def main():
    ref_shims.install()
    import surreal.utils as U
    U.set_global_serializer(pickle.dumps, pickle.loads)
    from surreal.distributed.exp_sender import ExpBuffer
    from surreal.distributed.exp_collector import ExperienceCollectorServer
    from surreal.distributed.parameter_server import ParameterServer
    from surreal.distributed.module_dict import ModuleDict
    from surreal.utils.checkpoint import Checkpoint
    os.makedirs(OUT, exist_ok=True)

    # ---- experience chunk ---------------------------------------------------------------------
    exps = make_experiences()
    buf = ExpBuffer()
    for e in exps:
        buf.add(hash_dict={'obs': e['obs'], 'obs_next': e['obs_next']},
                nonhash_dict={k: v for k, v in e.items() if k not in ('obs', 'obs_next')})
    n_storage = len(buf.ob_storage)
    chunk = buf.flush()
    srv = object.__new__(ExperienceCollectorServer)
    srv._weakref_map = weakref.WeakValueDictionary()
    exp_list, storage = U.deserialize(chunk)
    unpacked = srv._retrieve_storage(copy.deepcopy(exp_list), storage)
    with open(os.path.join(OUT, 'exp_chunk.pkl'), 'wb') as fp:
        pickle.dump({'chunk': chunk, 'unpacked': unpacked, 'n_storage': n_storage,
                     'chunk_hash': U.binary_hash(chunk)}, fp)

    # ---- the batch the reference's learner would see of that chunk (aggregator.py:106-262) ------------
    from surreal.learner.aggregator import MultistepAggregatorWithInfo
    agg = MultistepAggregatorWithInfo(
        collections.OrderedDict(low_dim=collections.OrderedDict(flat_inputs=[4])), {'dim': [2], 'type': 'continuous'})
    bt = agg.aggregate(copy.deepcopy(unpacked))
    assert bt['onetime_infos'] is None and len(bt['persistent_infos']) == 1
    np.savez(os.path.join(OUT, 'exp_chunk_aggregated.npz'), obs=bt['obs']['low_dim']['flat_inputs'],
             obs_next=bt['obs_next']['low_dim']['flat_inputs'], actions=bt['actions'], rewards=bt['rewards'],
             dones=bt['dones'], persistent_infos0=bt['persistent_infos'][0])

    # ---- parameter protocol -------------------------------------------------------------------
    ps = object.__new__(ParameterServer)
    ps.parameters, ps.param_info = None, None
    proto = {'empty_parameter': list(ps._handle_agent_request('parameter')),
             'empty_info': list(ps._handle_agent_request('info'))}
    binary = b'\\x00\\x01surreal-parameters\\xff'
    info = {'time': 12.5, 'iteration': 7, 'message': 'm', 'hash': U.binary_hash(binary)}
    ps._set_storage((binary, info))

    def rep(r):
        p, i = ps._handle_agent_request(r)
        return [None if p is None else p.hex(), i]
    proto['info'] = rep('info')
    proto['parameter'] = rep('parameter')
    proto['parameter_same_hash'] = rep('parameter:' + info['hash'])
    proto['parameter_other_hash'] = rep('parameter:abc')
    proto['parameter_empty_hash'] = rep('parameter:')
    proto['binary_hex'] = binary.hex()
    proto['hashes'] = {'': U.binary_hash(b''), 'abc': U.binary_hash(b'abc'),
                       'string_hash(surreal)': U.string_hash('surreal')}
    with open(os.path.join(OUT, 'param_protocol.json'), 'w') as fp:
        json.dump(proto, fp, indent=1, sort_keys=True)

    # ---- ModuleDict ---------------------------------------------------------------------------
    TxModule = sys.modules['torchx.nn'].Module
    TinyTx = type('TinyTx', (TinyModule, TxModule), {})
    md = ModuleDict({'net': TinyTx()})
    with open(os.path.join(OUT, 'module_dict.pkl'), 'wb') as fp:
        fp.write(md.dumps())

    # ---- checkpoint folder --------------------------------------------------------------------
    class Tracked(object):
        pass
    t = Tracked()
    t.model = TinyModule()
    t.counter = 3
    ck_dir = os.path.join(OUT, 'ckpt')
    shutil.rmtree(ck_dir, ignore_errors=True)
    ck = Checkpoint(ck_dir, 'learner', tracked_obj=t, tracked_attrs=['model', 'counter'],
                    keep_history=2, keep_best=1)
    for step, score in ((10, 1.0), (20, 3.0), (30, 2.0)):
        t.counter = step
        with torch.no_grad():
            t.model.fc.bias.fill_(float(step))
        ck.save(score=score, global_steps=step)
    print('reference checkpoint files:', sorted(os.listdir(ck_dir)))

    # ---- the other direction: surreal_amd writes, the reference reads ---------------------------
    from surreal_amd.utils import serializer as S
    from surreal_amd.distributed import ExpBuffer as OurBuffer
    from surreal_amd.utils.checkpoint import Checkpoint as OurCheckpoint
    S.set_global_serializer(pickle.dumps, pickle.loads)
    ob = OurBuffer()
    for e in exps:
        ob.add({'obs': e['obs'], 'obs_next': e['obs_next']},
               {k: v for k, v in e.items() if k not in ('obs', 'obs_next')})
    ours = ob.flush()
    assert ours == chunk, 'surreal_amd chunk differs from the reference chunk byte for byte'
    srv2 = object.__new__(ExperienceCollectorServer)
    srv2._weakref_map = weakref.WeakValueDictionary()
    a, b = U.deserialize(ours)
    got = srv2._retrieve_storage(a, b)
    assert pickle.dumps(got) == pickle.dumps(unpacked)
    tmp = tempfile.mkdtemp()
    t2 = Tracked()
    t2.model, t2.counter = TinyModule(), 99
    with torch.no_grad():
        t2.model.fc.bias.fill_(42.0)
    OurCheckpoint(tmp, 'learner', tracked_obj=t2, tracked_attrs=['model', 'counter'], keep_history=2,
                  keep_best=0).save(global_steps=5)
    t3 = Tracked()
    t3.model, t3.counter = TinyModule(), 0
    rck = Checkpoint(tmp, 'learner', tracked_obj=t3, tracked_attrs=None)
    assert rck.restore(0, 'history', check_ckpt_exists=True)
    assert t3.counter == 99 and float(t3.model.fc.bias[0]) == 42.0
    shutil.rmtree(tmp)
    print('reference consumed the surreal_amd chunk and checkpoint; fixtures in', os.path.relpath(OUT, ROOT))


if __name__ == '__main__':
    main()
