"""Default DDPG hyper-parameters (surreal/main/ddpg_configs.py:16-98)."""
from surreal_amd.session import (Config, BASE_LEARNER_CONFIG, BASE_ENV_CONFIG,
                                 LOCAL_SESSION_CONFIG)


def ddpg_learner_config():
    cfg = Config({
        'model': {
            'convs': [],
            'actor_fc_hidden_sizes': [300, 200],
            'critic_fc_hidden_sizes': [400, 300],
            'use_layernorm': False,
            'conv_spec': {'out_channels': [16, 32], 'kernel_sizes': [8, 4], 'strides': [4, 2],
                          'hidden_output_dim': 200},
        },
        'algo': {
            'gamma': 0.99,
            'n_step': 3,
            'stride': 1,
            'network': {
                'lr_actor': 1e-4, 'lr_critic': 1e-3,
                'clip_actor_gradient': True, 'actor_gradient_value_clip': 1.0,
                'clip_critic_gradient': False, 'critic_gradient_value_clip': 5.0,
                'actor_regularization': 0.0, 'critic_regularization': 0.0,
                'use_action_regularization': False, 'use_double_critic': False,
                'target_update': {'type': 'hard', 'interval': 500},
            },
            'exploration': {
                'param_noise_type': None, 'param_noise_sigma': 0.05, 'param_noise_alpha': 1.15,
                'param_noise_target_stddev': 0.005,
                'noise_type': 'normal', 'max_sigma': 1.0, 'theta': 0.15, 'dt': 1e-3,
            },
        },
        'replay': {'batch_size': 512, 'memory_size': int(1000000 / 3), 'sampling_start_size': 3000,
                   'replay_shards': 3},
        'parameter_publish': {'min_publish_interval': 3},
    })
    cfg.extend(BASE_LEARNER_CONFIG)
    return cfg


def ddpg_env_config(obs_dim, action_dim, num_agents=1, env_name='synthetic:flat', pixel=None):
    """pixel = (C, H, W) adds the camera0 frames and sets pixel_input (ddpg_configs.py:100-112)"""
    obs_spec = {'low_dim': {'flat_inputs': [obs_dim]}}
    if pixel is not None:
        obs_spec = {'pixel': {'camera0': list(pixel)}, 'low_dim': {'flat_inputs': [obs_dim]}}
    cfg = Config({
        'env_name': env_name, 'num_agents': num_agents, 'action_repeat': 1, 'pixel_input': pixel is not None,
        'frame_stacks': 1, 'limit_episode_length': 0, 'stochastic_eval': True,
        'action_spec': {'dim': [action_dim], 'type': 'continuous'},
        'obs_spec': obs_spec,
    })
    cfg.extend(BASE_ENV_CONFIG)
    return cfg


def ddpg_session_config(folder='/tmp/surreal_amd_ddpg'):
    cfg = Config({
        'folder': folder,
        'agent': {'fetch_parameter_mode': 'step', 'fetch_parameter_interval': 200},
        'sender': {'flush_iteration': 100},
        'learner': {'prefetch_processes': 3, 'num_gpus': 1},
        'replay': {'max_puller_queue': 3, 'max_prefetch_queue': 1},
    })
    cfg.extend(LOCAL_SESSION_CONFIG)
    return cfg
