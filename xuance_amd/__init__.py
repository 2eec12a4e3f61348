"""xuance_amd -- MI355X-native engine for XuanCe's RL update loop (rollout buffer + PPO / DQN / QMIX learners).

Host-side classes mirror the reference's plugin interfaces (Buffer, Learner, Agent) and drive hand-written HIP
kernels for gfx950 through the C ABI in include/xrl_hip.h.  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
