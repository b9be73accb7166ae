from .base import Replay, DeviceTable
from .fifo_replay import FIFOReplay
from .uniform_replay import UniformReplay
