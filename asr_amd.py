"""Import alias: `import asr_amd` loads the package in ./end-to-end_asr_pytorch_amd (whose directory name, fixed by
the project layout, is not a valid Python identifier) and registers it under this module's name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "end-to-end_asr_pytorch_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
