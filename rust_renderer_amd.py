"""Import shim: the package directory is `rust-renderer_amd/` (not a valid identifier), so
`import rust_renderer_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rust-renderer_amd")
_spec = importlib.util.spec_from_file_location("rust_renderer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rust_renderer_amd"] = _mod
_spec.loader.exec_module(_mod)
