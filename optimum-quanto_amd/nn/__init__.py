from .linear import *
from .module import *
