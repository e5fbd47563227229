"""Import name for the package kept in the directory `prime-rl-cse_amd/`.

A hyphen cannot appear in a Python module name, so this module loads that
directory as the package `prime_rl_cse_amd` and puts it in its own place:
    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd.loss import policy_loss
"""

import importlib.util
import sys
from pathlib import Path

_root = Path(__file__).resolve().parent / "prime-rl-cse_amd"
_spec = importlib.util.spec_from_file_location(
    __name__, _root / "__init__.py", submodule_search_locations=[str(_root)]
)
_pkg = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
