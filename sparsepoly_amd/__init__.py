"""sparsepoly_amd -- MI355X-native proximal coordinate-descent core for sparse
factorization machines; drop-in for the pcd / pbcd path of neonnnnn/sparsepoly
(reference ``sparsepoly/__init__.py:1-19`` exports the same estimator and
regularizer names)."""
from .regularizer import L1, L21, OmegaCS, OmegaTI, SquaredL12, SquaredL21
from .sparse_all_subsets import SparseAllSubsetsClassifier, SparseAllSubsetsRegressor
from .sparse_factorization_machines import (
    SparseFactorizationMachineClassifier,
    SparseFactorizationMachineRegressor,
)

__all__ = [
    "L1",
    "L21",
    "OmegaCS",
    "OmegaTI",
    "SquaredL12",
    "SquaredL21",
    "SparseAllSubsetsClassifier",
    "SparseAllSubsetsRegressor",
    "SparseFactorizationMachineClassifier",
    "SparseFactorizationMachineRegressor",
]
