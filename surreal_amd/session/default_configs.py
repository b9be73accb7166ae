"""
Base config trees every algorithm config extends (placeholders mark what a concrete
algorithm must fill).  Same keys as the reference's surreal/session/default_configs.py:4-29,
31-55,57-148 for the entries the hot path reads; the ZMQ / tensorplex / kubernetes sections
of the reference's session config have no meaning here (in-process hand-off) and are kept
only as inert defaults so reference config dicts extend without error.
"""

BASE_LEARNER_CONFIG = {
    'model': '_dict_',
    'algo': {
        'n_step': 1,
        'gamma': '_float_',
        'use_batchnorm': False,
        'limit_training_episode_length': 0,
        'network': {'actor_regularization': 0.0, 'critic_regularization': 0.0},
    },
    'replay': {'batch_size': '_int_', 'replay_shards': 1},
    'parameter_publish': {'min_publish_interval': 0.3},
}

BASE_ENV_CONFIG = {
    'env_name': '_str_',
    'sleep_time': 0.0,
    'video': {'record_video': False, 'max_videos': 10, 'record_every': 10, 'save_folder': None},
    'eval_mode': {},
    'action_spec': {},
    'obs_spec': {},
    'frame_stacks': 1,
    'frame_stack_concatenate_on_env': True,
}

BASE_SESSION_CONFIG = {
    'folder': '_str_',
    'replay': {'max_puller_queue': 3, 'evict_interval': 0.0, 'tensorboard_display': True},
    'sender': {'flush_iteration': 1, 'flush_time': 0},
    'learner': {'num_gpus': 0, 'prefetch_processes': 2, 'max_prefetch_queue': 10,
                'max_preprocess_queue': 2},
    'agent': {'fetch_parameter_mode': 'step', 'fetch_parameter_interval': 100, 'num_gpus': 0},
    # default_configs.py:183-192: how often (in episodes) the env monitors report, and the
    # evaluator's throttle; the tensorplex transport itself is replaced by in-process recorders
    'tensorplex': {'update_schedule': {'training_env': 20, 'eval_env': 20, 'eval_env_sleep': 30,
                                       'agent': 20, 'learner': 20, 'learner_min_update_interval': 30}},
    'checkpoint': {
        'restore': False,
        'restore_folder': None,
        'learner': {'restore_target': 0, 'mode': 'history', 'keep_history': 2, 'keep_best': 0,
                    'periodic': 100000, 'min_interval': 0},
        'agent': {'restore_target': 0, 'mode': 'history', 'keep_history': 2, 'keep_best': 0,
                  'periodic': 100},
    },
}

LOCAL_SESSION_CONFIG = dict(BASE_SESSION_CONFIG)
