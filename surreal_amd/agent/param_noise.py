"""
Parameter-space exploration noise for the DDPG agents (surreal/agent/param_noise.py:5-74; Plappert et
al., arXiv 1706.01905).  It acts on the parameters an agent has just FETCHED -- the wire form
``{module: {key: np.ndarray}}`` -- in ``on_parameter_fetched`` (ddpg_agent.py:149-153): every
array gets ``N(0, sigma)`` added, drawn from numpy's global stream in iteration order, as in the
reference.

  normal            fixed sigma
  adaptive_normal   sigma is divided / multiplied by alpha at every fetch so that the distance
                    between the noisy and the clean policy's actions tracks target_stddev; the clean
                    policy is a second copy of the model loaded with the un-noised parameters, and the
                    distance is the L2 distance of ONE action pair measured every
                    compute_dist_interval-th act() (the reference overwrites, it does not accumulate,
                    then divides by the number of act() calls -- kept as is).
"""
import copy

import numpy as np


def _perturbed(params, sigma):
    """params with N(0, sigma) added to every array (new arrays; same nesting and order)"""
    for module in params:
        for key in params[module]:
            p = params[module][key]
            if type(p) is not np.ndarray:
                raise AssertionError('parameter noise expects numpy parameters, got %r' % type(p))
            params[module][key] = p + np.random.normal(0, sigma, size=tuple(p.shape))
    return params


class ParameterNoise(object):
    def apply(self, params):
        return params


class NormalParameterNoise(ParameterNoise):
    def __init__(self, sigma):
        self.sigma = sigma

    def apply(self, params):
        return _perturbed(params, self.sigma)

    def __repr__(self):
        return 'NormalParameterNoise(sigma={})'.format(self.sigma)


class AdaptiveNormalParameterNoise(ParameterNoise):
    def __init__(self, model_copy, module_dict_copy, target_stddev, compute_dist_interval=10,
                 alpha=1.04, sigma=0.01):
        self.sigma, self.alpha = sigma, alpha
        self.target_stddev = target_stddev
        self.compute_dist_interval = compute_dist_interval
        self.original_model = model_copy                 # the clean policy
        self.original_model_module_dict = module_dict_copy
        self.i = 0
        self.total_action_distance = 0.0

    def compute_action_distance(self, obs, modified_model_action):
        if self.i % self.compute_dist_interval == 0:
            clean, _ = self.original_model(obs, calculate_value=False)
            self.total_action_distance = float((((clean - modified_model_action) ** 2).sum()) ** 0.5)
        self.i += 1

    def apply(self, params):
        if self.i > 0:
            if self.total_action_distance / self.i > self.target_stddev:
                self.sigma /= self.alpha
            else:
                self.sigma *= self.alpha
        self.i = 0
        self.original_model_module_dict.load(copy.deepcopy(params))     # load() converts in place
        return _perturbed(params, self.sigma)

    def __repr__(self):
        return 'AdaptiveNormalParameterNoise(target={}, alpha={}, sigma={})'.format(
            self.target_stddev, self.alpha, self.sigma)
