"""CPU oracle for the face-detection hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (infercam_onnx_amd) never does.  See oracle/ufd_oracle.h.
"""
from .pyoracle import *  # noqa: F401,F403
