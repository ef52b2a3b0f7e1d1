from .painn import PaiNN, PaiNNInteraction, PaiNNMixing
from .schnet import SchNet, SchNetInteraction
