from .config import Config, ConfigError, extend_config
from .default_configs import BASE_LEARNER_CONFIG, BASE_ENV_CONFIG, BASE_SESSION_CONFIG, LOCAL_SESSION_CONFIG
