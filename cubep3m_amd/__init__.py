"""MI355X-native P3M gravity step behind cubep3m's `particle_mesh`."""
from .params import Params  # noqa: F401
