from .base import Env, Wrapper, MaxStepWrapper, FrameStackWrapper
from .exp_sender_wrapper import (ExpSenderWrapperBase, ExpSenderWrapperSSAR,
                                 ExpSenderWrapperSSARNStepBootstrap,
                                 ExpSenderWrapperMultiStepMovingWindowWithInfo)
from .synthetic_env import SyntheticEnv, SyntheticVecEnv
