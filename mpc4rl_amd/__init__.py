"""mpc4rl_amd — MI355X-native batched MPC-as-policy engine.

Drop-in for the hot path of MPC-Based-Reinforcement-Learning/mpc4rl (``rlmpc.mpc``): the OCP solve
(acados SQP in the reference, rlmpc/mpc/common/mpc.py:27-96,177-202) and the KKT sensitivities
``dV/dp``, ``dQ/dp``, ``du0*/dp`` (rlmpc/mpc/nlp.py:1341-1424) for a whole batch of instances in one
HIP kernel launch.  The arithmetic lives in ``csrc/`` behind the C ABI of ``include/mpcrl.h``;
PyTorch is used for device memory and streams only.  There is no CPU fallback: importing the solver
classes without ``libmpcrl_hip.so`` raises.
"""
from .problems import OcpDescription, cartpole_ocp, chain_mass_ocp, linear_system_ocp  # noqa: F401
from .batch import MPCBatch, SolveResult  # noqa: F401
from .envs import BatchedCartPoleSwingUpEnv, BatchedLinearSystemEnv  # noqa: F401
from .qlearning import BatchedQLearning  # noqa: F401
from .td3 import BatchedTD3, ContinuousCritic, DeviceReplayBuffer, MPCActor, MPCTD3Policy  # noqa: F401
from .config import cartpole_ocp_from_config, read_config, store_iterate, load_iterate  # noqa: F401
from .mpc import MPC, CartpoleMPC, ChainMassMPC, LinearSystemMPC  # noqa: F401

__all__ = ["OcpDescription", "cartpole_ocp", "linear_system_ocp", "chain_mass_ocp", "MPCBatch", "SolveResult", "MPC", "CartpoleMPC",
           "LinearSystemMPC", "ChainMassMPC"]
