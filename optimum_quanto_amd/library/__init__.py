from .extension import *
from .hip import QuantoHipError, quanto_hip
from . import ops  # noqa: F401  (defines / registers the quanto:: operators)
from . import plugin  # noqa: E402,F401

plugin.install()  # no-op unless optimum.quanto is already imported (plug-in mode)
