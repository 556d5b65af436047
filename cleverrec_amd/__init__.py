"""cleverrec_amd: MI355X (gfx950) embedding-training core behind CleverRec's class surface.
See DESIGN.md.  Importing the package needs no GPU; opening a Session does."""
from .config import read_configs  # noqa: F401
from .data import RankingData, synthetic  # noqa: F401

__version__ = "0.1.0"
