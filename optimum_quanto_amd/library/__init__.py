from .extension import *
from .hip import QuantoHipError, quanto_hip
from . import ops  # noqa: F401  (defines / registers the quanto:: operators)
