from .conv import *
from .linear import *
from .module import *
