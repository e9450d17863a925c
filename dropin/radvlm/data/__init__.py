"""``radvlm.data``: this build's record format / instruction generators, with the rest of the reference's ``radvlm.data`` modules
(datasets, utils, create_llava_dataset ...) still importable from its own tree through the extended ``__path__``."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from radvlm_amd.data import *  # noqa: E402,F401,F403
