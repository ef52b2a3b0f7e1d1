from .activations import shifted_softplus
from .base import Dense
from .blocks import build_mlp
from .cutoff import CosineCutoff, cosine_cutoff
from .radial import BesselRBF, GaussianRBF, gaussian_rbf
from .scatter import scatter_add
from .utils import replicate_module
