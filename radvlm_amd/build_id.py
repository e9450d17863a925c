"""Identity of the kernel sources a measurement belongs to.

The GPU box receives a snapshot without .git, so profiles are tied to the code by a hash of the HIP sources and the C-ABI header
instead of a commit id: ``bench.py`` prints it, the PMC post-processing tools store it, and bench.py refuses PMC-derived numbers
whose hash differs from the sources it is running (stale evidence is reported as null, never as a number)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha256():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "radvlm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "radvlm_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
