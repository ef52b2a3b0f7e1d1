"""schnetpack_amd -- MI355X-native (gfx950) message-passing core behind SchNetPack's
``representation.{SchNet,PaiNN}`` / ``nn.{radial,cutoff,scatter}`` API.  See DESIGN.md."""
from . import properties  # noqa: F401

__version__ = "0.1.0"
