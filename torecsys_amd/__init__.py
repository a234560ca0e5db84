"""torecsys_amd -- MI355X (gfx950) native embedding-lookup + feature-interaction path for torecsys
CTR models: drop-in ``nn.Module``s (same names / signatures / tensor names / state_dict keys as
``torecsys.inputs`` and ``torecsys.layers``) over hand-written HIP kernels behind a C ABI
(``include/trs_abi.h`` -> ``torecsys_amd/libtrs_hip.so``).  No CPU path: modules raise off-GPU."""
from . import functional, fused, graph, inputs, layers, optim, staging  # noqa: F401
from .patching import patch, unpatch  # noqa: F401

__version__ = "0.1.0"
