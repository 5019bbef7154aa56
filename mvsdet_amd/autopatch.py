"""Import me AFTER the reference's `projects.NeRF-Det.nerfdet.mvsdet` (e.g. as the last entry of mmengine's
`custom_imports`): rebinds the reference's hot-path functions to the HIP-backed mirrors."""
from . import integration

PATCHED = integration.apply_on_import()
