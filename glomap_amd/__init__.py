"""MI355X-native implementation of GLOMAP's estimator hot path (RA -> GP -> BA).

The product is the C-ABI shared library ``glomap_amd/csrc/libgsfm.so`` (include/gsfm.h);
this package is the host-side mirror of the reference's estimator interface on top of it.
"""
from .flat import BaProblem, GpProblem, RaProblem  # noqa: F401
