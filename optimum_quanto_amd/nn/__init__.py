from .conv import *
from .layernorm import *
from .linear import *
from .module import *
