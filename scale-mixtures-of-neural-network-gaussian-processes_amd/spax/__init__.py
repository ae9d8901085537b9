"""spax — host-side mirror of the reference library's hot-path surface (spax/__init__.py)."""
from . import kernels
from . import models
from . import likelihoods
from . import utils
from . import bijectors

from .base import *
