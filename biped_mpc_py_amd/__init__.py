"""biped_mpc_py_amd -- MI355X-native batched HECTOR force-and-moment MPC.

Drop-in for the hot path of junhengl/biped_mpc_py (`solve_mpc`, REF:187-304), solved for whole
batches by hand-written HIP kernels (csrc/) behind a C ABI (include/bmpc.h).  Importing this package
does not load the shared library; the first solver call does, and raises if it is absent.
"""
from .params import MPC, Biped, pack_params                                  # noqa: F401
from .api import (BatchSolver, solve_mpc, solve_mpc_batch, get_contact_sequence,   # noqa: F401
                  phase_index, phase_indices, lowLevelControl, getFootPositionWorld, SolverStatusWarning,
                  close_cached_solvers, get_reference_trajectory, get_reference_foot_trajectory,
                  reference_trajectories_batch)
from . import sharding                                                        # noqa: F401
from ._lib import BmpcError                                                   # noqa: F401

__all__ = ["MPC", "Biped", "pack_params", "BatchSolver", "solve_mpc", "solve_mpc_batch",
           "get_contact_sequence", "phase_index", "phase_indices", "lowLevelControl", "getFootPositionWorld", "sharding",
           "BmpcError", "SolverStatusWarning", "close_cached_solvers",
           "get_reference_trajectory", "get_reference_foot_trajectory", "reference_trajectories_batch"]
