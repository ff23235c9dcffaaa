"""optimum_quanto_amd: MI355X-native backend for the optimum-quanto quantized-linear hot path.

Public names follow ``optimum.quanto`` so that code written against the reference runs unchanged::

    from optimum_quanto_amd import quantize, freeze, qint4, QLinear
"""
__version__ = "0.1.0"

from .library import *
from .tensor import *
from .nn import *
from .model_api import *
from .models import *
from .checkpoint import *
from . import parallel  # noqa: E402,F401  (column shard over RCCL / gloo)
