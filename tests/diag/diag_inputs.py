"""diagnostic: are the regenerated inputs bit-identical across hosts; fp32 vs fp64 oracle"""
import copy, json, sys, os, hashlib, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as H
import ppo_oracle
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_clip'
g, case = H.load_golden(name)
batch, params, zstate = H.case_inputs(case)
def md5(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
print('obs', md5(batch['obs']['low_dim']['flat_inputs']), 'obs_next', md5(batch['obs_next']['low_dim']['flat_inputs']),
      'act', md5(batch['actions']), 'rew', md5(batch['rewards']), 'dones', md5(batch['dones']), 'pds', md5(batch['persistent_infos'][0]))
print('params', {k: md5(v) for k, v in params.items()})
print('z', {k: md5(v) for k, v in zstate.items()})
print('threads', torch.get_num_threads(), torch.__config__.show().split('\n')[3:6])
hyper = dict(case['hyper']); hyper['n_step'] = case['shape']['N']; hyper['epoch_baseline'] = 2; hyper['epoch_policy'] = 1
O = ppo_oracle.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
O.learn(copy.deepcopy(batch))
print('fp32', [(t['_val_loss'], t['grad_norm_critic']) for t in O.trace['value']])
print('ret md5', md5(O.trace['returns']), 'adv md5', md5(O.trace['advantages']), float(np.abs(O.trace['returns']-g['returns']).max()))
src = open(os.path.join(ROOT, 'oracle', 'ppo_oracle.py')).read().replace('torch.float32', 'torch.float64')
mod = types.ModuleType('o64'); exec(compile(src, 'o64', 'exec'), mod.__dict__)
torch.set_default_dtype(torch.float64)
O = mod.OraclePPOLearner(params, case['shape']['A'], case['shape']['B'], zstate=zstate, **hyper)
O.learn(copy.deepcopy(batch))
print('fp64', [(t['_val_loss'], t['grad_norm_critic']) for t in O.trace['value']])
print('golden', [(t['_val_loss'], t['grad_norm_critic']) for t in json.loads(str(g['value_trace_json']))[:2]])
