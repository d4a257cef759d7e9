"""Import aliases for the reference's package names.

The reference directory is `streammind/` while every absolute import in it says `videollama2.*` (46 files,
e.g. eval/video_score_stream_demo.py:19,32-38; pyproject names the distribution `videollama2`).  The top-level packages
`videollama2/` and `streammind/` of this repo call `install(__name__)`: the name (and every dotted name under it) then
resolves to the SAME module objects as `streammind_amd`, so `from videollama2.model.builder import load_pretrained_model`
or `from videollama2 import model_init` in a caller picks up the MI355X drop-in without an edit."""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import sys

TARGET = "streammind_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, alias: str):
        self.alias = alias

    def _real(self, fullname: str) -> str:
        return TARGET + fullname[len(self.alias):]

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + "."):
            return None
        try:
            real = importlib.util.find_spec(self._real(fullname))
        except ModuleNotFoundError:
            return None
        if real is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=real.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(self._real(spec.name))      # the real module object itself: no second copy

    def exec_module(self, module):
        pass


def install(alias: str) -> None:
    if not any(isinstance(f, _AliasFinder) and f.alias == alias for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder(alias))
    sys.modules[alias] = importlib.import_module(TARGET)
