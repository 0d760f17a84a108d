"""MI355X-native DiffAb diffusion/denoise hot path; import name kept from the reference (diffab_pytorch/__init__.py:1)."""
from diffab_pytorch.diffab_pytorch import DiffAb  # noqa: F401
