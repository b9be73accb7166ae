from .base import Env, Wrapper, MaxStepWrapper, FrameStackWrapper, stack_sources
from .exp_sender_wrapper import (ExpSenderWrapperBase, ExpSenderWrapperSSAR,
                                 ExpSenderWrapperSSARNStepBootstrap,
                                 ExpSenderWrapperMultiStepMovingWindowWithInfo)
from .synthetic_env import SyntheticEnv, SyntheticVecEnv
from .adapters import (ObsTransform, GymAdapter, RobosuiteWrapper, DMControlAdapter, FilterWrapper,
                       ObservationConcatenationWrapper, TransposeWrapper, GrayscaleWrapper,
                       make_env, make_env_config, wrap_gym, wrap_robosuite, wrap_dm_control)
from .monitor import (EpisodeMonitor, ConsoleMonitor, TrainingTensorplexMonitor,
                      EvalTensorplexMonitor)
