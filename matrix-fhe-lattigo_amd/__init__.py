"""MI355X-native full-RNS polynomial-ring engine: host-side mirror of the reference's `ring` interface.

The product path is the HIP library matrix-fhe-lattigo_amd/lib/libringhip.so (C ABI: include/ringhip.h).  This
package only binds it (ctypes) and mirrors the names of ring.Ring / ring.SubRing / ring.BasisExtender so tests read
like the reference's.  There is no CPU fallback: importing works without a GPU (the library loads), any compute call
without a device fails loudly."""
from .ringhip import (  # noqa: F401
    RingHipError, Ring, SubRing, DevicePoly, PinnedBuffer, BasisExtender, Standard, ConjugateInvariant, Matrix3N, AutomorphismNTTIndex, OPS, lib, library_path,
)
from .schemes import Ciphertext, MatrixCKKSEvaluator, ckks_tensor_degree1, ckks_polymul  # noqa: F401,E402
from . import rlwe  # noqa: F401,E402
from . import ckks  # noqa: F401,E402
from . import rgsw  # noqa: F401,E402
