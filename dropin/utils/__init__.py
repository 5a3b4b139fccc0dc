"""`utils` of the reference trainer -> textualdegremoval_amd.utils (see dropin/_alias.py)"""
import os
import sys

_here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_here, os.path.dirname(_here)):
    if _p not in sys.path:
        sys.path.append(_p)
from _alias import alias  # noqa: E402

alias(__name__, 'textualdegremoval_amd.utils')
