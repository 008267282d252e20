"""linear_amd -- MI355X-native hot path of `linear filter` (xp3i4/linear): minimizer index build,
seed lookup, anchor chaining and window extension into cords, as hand-written HIP kernels behind the
C ABI of include/linear_amd.h.  This package is only the thin Python plumbing over that ABI."""
from .api import Filter, LnrError  # noqa: F401
