from .exp_sender import ExpBuffer, ExpSender
from .exp_collector import ExperienceCollector
from .module_dict import ModuleDict
from .parameter_server import ParameterPublisher, ParameterServer, ParameterClient
from .data_fetcher import (LearnerDataPrefetcher, PinnedBatchStager, SharedBatchStager, AggregationPool,
                           PooledDataPrefetcher, ppo_aggregate_factory)
