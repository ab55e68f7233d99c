"""MI355X-native `pointnet2_ops` package: `_ext` (C-ABI backed), `pointnet2_utils`,
`pointnet2_modules` -- the call surface of the reference's
external/pointnet2_ops_lib/pointnet2_ops."""
from . import _ext  # noqa: F401
from . import pointnet2_utils  # noqa: F401
from . import pointnet2_modules  # noqa: F401

__version__ = "3.0.0"  # the reference's _version.py
