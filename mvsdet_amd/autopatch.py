"""Two ways to get the HIP-backed functions into an unmodified reference run:

* import this module AFTER the reference's `projects.NeRF-Det.nerfdet.mvsdet` (e.g. as the last entry of
  mmengine's `custom_imports`): everything already imported is patched at once (`PATCHED`);
* `install_import_hook()` BEFORE the reference is imported (what `python -m mvsdet_amd.launch tools/test.py ...`
  does): a meta-path finder waits for a module named `*.nerfdet.mvsdet` and patches it right after its own
  loader has executed it -- the config and the reference source stay untouched.
"""
import importlib.abc
import importlib.util
import sys

from . import integration

TARGET_SUFFIX = ".nerfdet.mvsdet"


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        if all(hasattr(module, n) for n in ("MVSDet", "homo_warping", "backproject_Weigh")):
            module.__mvsdet_amd_originals__ = integration.patch_reference(module)


class ReferenceImportHook(importlib.abc.MetaPathFinder):
    """Delegates to the remaining finders, then wraps the loader of the one module we care about."""

    def find_spec(self, fullname, path=None, target=None):
        if not (fullname == TARGET_SUFFIX[1:] or fullname.endswith(TARGET_SUFFIX)):
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None:
                spec.loader = _PatchingLoader(spec.loader)
                return spec
        return None


def install_import_hook() -> ReferenceImportHook:
    for finder in sys.meta_path:
        if isinstance(finder, ReferenceImportHook):
            return finder
    hook = ReferenceImportHook()
    sys.meta_path.insert(0, hook)
    return hook


def remove_import_hook() -> None:
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, ReferenceImportHook)]


PATCHED = integration.apply_on_import()
