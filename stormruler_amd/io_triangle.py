"""Triangle (2-D TetGen-format) mesh ingestion -> face graph: the 2-D case of :mod:`stormruler_amd.io_tetgen`, which
follows both branches of ``read_mesh_from_tetgen`` (source/Storm/Mallard/IoTetgen.hpp:44-235).  Kept under its
round-2 name for the callers that read the reference's own 2-D fixtures (``tests/_data/mesh/*.1.*``)."""
from .io_tetgen import read_triangle

__all__ = ["read_triangle"]
