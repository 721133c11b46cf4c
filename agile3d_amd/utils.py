"""``MinkowskiEngine.utils`` as the reference's callers use it (datasets/InterMultiObj3DSegDataset.py:67-71,129;
interactive_tool/interactive_segmentation_user.py): with ``import agile3d_amd as ME`` the calls
``ME.utils.sparse_quantize(...)`` and ``ME.utils.batched_coordinates(...)`` keep working unchanged."""
from .sparse import batched_coordinates, sparse_quantize  # noqa: F401

__all__ = ["sparse_quantize", "batched_coordinates"]
