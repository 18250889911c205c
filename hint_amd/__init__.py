"""hint_amd — MI355X-native (gfx950) implementation of HINT's recursive affine-coupling block.

Public surface mirrors /root/reference/hint.py (see hint_amd/hint.py); the compute lives in
hint_amd/csrc behind the C ABI of include/hint_amd.h.
"""
from .hint import (HierarchicalAffineCouplingBlock, HierarchicalAffineCouplingTree,  # noqa: F401
                   HintAmdError, linear_subnet_constructor, set_pack_cache, set_param_grad_mode)

from .flow import FixedOrthogonal, HintFlow  # noqa: F401,E402
from .train import FlowTrainer  # noqa: F401,E402
from .conditional import (AffineCoupling, ConditionalFlowTrainer, ConditionalHintFlow,  # noqa: F401,E402
                          ExternalAffineCoupling, F_fully_connected)

__version__ = "0.1.0"
