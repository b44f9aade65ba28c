"""CPU oracle for the ring hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The product
(matrix-fhe-lattigo_amd) never does."""
from .ring_oracle import *  # noqa: F401,F403
