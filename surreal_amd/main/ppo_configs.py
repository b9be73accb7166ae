"""
Default PPO hyper-parameters: the values that pin the reference's numerics
(surreal/main/ppo_configs.py:15-94), as functions returning fresh
Config trees so callers can edit them freely.
"""
from surreal_amd.session import (Config, BASE_LEARNER_CONFIG, BASE_ENV_CONFIG,
                                 LOCAL_SESSION_CONFIG)


def ppo_learner_config(**algo_overrides):
    cfg = Config({
        'model': {
            'convs': [],
            'actor_fc_hidden_sizes': [300, 200],
            'critic_fc_hidden_sizes': [300, 200],
            'cnn_feature_dim': 256,
            'use_layernorm': False,
        },
        'algo': {
            'use_z_filter': True,
            'use_r_filter': False,
            'gamma': 0.995,
            'n_step': 25,
            'stride': 20,
            'network': {
                'lr_actor': 1e-4, 'lr_critic': 1e-4,
                'clip_actor_gradient': True, 'actor_gradient_norm_clip': 5.0,
                'clip_critic_gradient': True, 'critic_gradient_norm_clip': 5.0,
                'actor_regularization': 0.0, 'critic_regularization': 0.0,
                'anneal': {'lr_scheduler': 'LinearWithMinLR', 'frames_to_anneal': 5e6,
                           'lr_update_frequency': 100, 'min_lr': 5e-5},
            },
            'ppo_mode': 'adapt',
            'advantage': {'norm_adv': True, 'lam': 0.97, 'reward_scale': 1.0},
            'rnn': {'if_rnn_policy': True, 'rnn_hidden': 100, 'rnn_layer': 1, 'horizon': 5},
            'consts': {'init_log_sig': -1.0, 'log_sig_range': 0.25, 'epoch_policy': 10,
                       'epoch_baseline': 10, 'adjust_threshold': (0.5, 2.0),
                       'kl_target': 0.015},
            'adapt_consts': {'kl_cutoff_coeff': 250, 'beta_init': 1.0,
                             'beta_range': (1 / 35.0, 35.0), 'scale_constant': 1.5},
            'clip_consts': {'clip_epsilon_init': 0.2, 'clip_range': (0.05, 0.3),
                            'scale_constant': 1.2},
        },
        'replay': {'batch_size': 64, 'memory_size': 96, 'sampling_start_size': 64,
                   'replay_shards': 1},
        'parameter_publish': {'exp_interval': 4096},
    })
    for k, v in algo_overrides.items():
        cfg.algo[k] = v
    cfg.extend(BASE_LEARNER_CONFIG)
    return cfg


def ppo_env_config(obs_dim, action_dim, env_name='synthetic:flat', pixel=None):
    """obs_spec / action_spec in the format make_env_config derives at run time
    (surreal/env/make_env.py:16-38, docs/env.md:48-77); pixel = (C, H, W) adds the camera0
    frames and sets pixel_input (ppo_configs.py:100,112)"""
    cfg = Config({
        'env_name': env_name,
        'action_repeat': 1,
        'pixel_input': False,
        'use_grayscale': False,
        'use_depth': False,
        'frame_stacks': 1,
        'limit_episode_length': 200,
        'stochastic_eval': True,
        'action_spec': {'dim': [action_dim], 'type': 'continuous'},
        'obs_spec': {'low_dim': {'flat_inputs': [obs_dim]}},
    })
    if pixel is not None:
        cfg.pixel_input = True
        cfg.obs_spec['pixel'] = {'camera0': [int(v) for v in pixel]}
    cfg.extend(BASE_ENV_CONFIG)
    return cfg


def ppo_session_config(folder='/tmp/surreal_amd'):
    cfg = Config({
        'folder': folder,
        'agent': {'fetch_parameter_mode': 'step', 'fetch_parameter_interval': 100, 'num_gpus': 0},
        'sender': {'flush_iteration': 3},
        'learner': {'num_gpus': 1},
        'replay': {'max_puller_queue': 3, 'max_prefetch_queue': 1},
        'checkpoint': {'learner': {'mode': 'history', 'periodic': 1000, 'min_interval': 15 * 60}},
    })
    cfg.extend(LOCAL_SESSION_CONFIG)
    return cfg
