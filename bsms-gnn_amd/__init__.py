"""bsms_gnn_amd: MI355X-native engine for the BSMS-GNN multi-level message-passing path.

Host-side mirror of the reference's `src/ops` + `src/models` interface over the C ABI of
libbsms_hip.so (include/bsms_hip.h).  Import name: `bsms_gnn_amd` (directory: `bsms-gnn_amd/`)."""
from . import _abi  # noqa: F401
from .graph import LevelData, LevelPlan, MeshBank, clear_plan_cache, collate_variable_meshes, concat_plans, plan_for  # noqa: F401
from .model import BSMS_Simulator, Normalizer, masked_rmse  # noqa: F401
from .ops import BSGMP, GMP, MLP, InferenceSession, Unpool, WeightedEdgeConv, degree, scatter_sum  # noqa: F401
from .dp import DataParallel, GradBuckets, global_masked_rmse  # noqa: F401
from .hierarchy import BistrideMultiLayerGraph, to_flat_edge  # noqa: F401
from .rollout import RolloutErrors, rank_slice, rollout_batch, rollout_dataset, rollout_errors, rollout_one_traj, rollout_rmse  # noqa: F401
from .step import FusedStep  # noqa: F401
from .trainer import DevicePrefetcher, FusedAdamW, Trainer, WarmupCosineDecay  # noqa: F401

__version__ = "0.1.0"
