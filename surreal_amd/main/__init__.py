"""default configurations of the two algorithms (surreal/main/ppo_configs.py, ddpg_configs.py)"""
