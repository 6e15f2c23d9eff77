from .actor import Gaussian_FF_Actor
from .critic import FF_V
