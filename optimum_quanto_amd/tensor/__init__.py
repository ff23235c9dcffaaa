from .base import *
from .dtypes import *
from .grouping import *
from .linear_function import *
from .packing import *
from .scale_search import *
from .weights import *
from .activations import *
