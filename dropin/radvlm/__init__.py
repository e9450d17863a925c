"""``radvlm`` namespace extension: adds this build's ``radvlm.data`` contract in front of, not instead of, the reference's package.

`dropin/` goes first on PYTHONPATH; ``pkgutil.extend_path`` appends every other ``radvlm`` directory found
on the path to this package's ``__path__``, so ``radvlm.data.datasets``, ``radvlm.evaluation`` ... keep resolving to the reference's own
tree.  The reference's ``radvlm/__init__.py`` (:1-7) is then not executed; the one name it defines, ``DATA_DIR``, is provided here with
the same error when the environment variable is unset -- raised on use instead of at import (training needs no data root).
"""
import os
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)


def __getattr__(name):
    if name == "DATA_DIR":
        value = os.environ.get("DATA_DIR")
        if value is None:
            raise EnvironmentError("The environment variable 'DATA_DIR' is not set.")
        return value
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
