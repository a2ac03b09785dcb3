"""scene-net_amd -- MI355X-native SCENE-Net GENEO forward path (voxelise -> GENEO bank -> 3D conv -> head).

Only what the hot path needs: `csrc/` (hand-written HIP for gfx950 behind the C ABI of
include/scenenet_hip.h) and the host-side mirror of the reference's operator interface.
Import as `scene_net_amd` (shim package at the repo root).
"""
from . import _hip
from ._hip import HipLibraryError, LIB_PATH
from .geneos import (GENEO_kernel_torch, arrow, cone_kernel, cylinder_kernel, cylinderv2, neg_sphere_kernel,
                     negSpherev2)
from .scene_net import GENEO_Layer, SCENE_Net, SCENE_Net_Class, SCENENetQuantile, SceneNet
from .transforms import ToFullDense, ToTensor, Voxelization
from .voxelization import (PointBatch, VoxelGrids, hist_on_voxel, prob_to_label, reg_on_voxel, voxelize_batch,
                           vxg_to_xyz)
from .pipeline import CapturedPipeline, ScenePipeline, shard_range
from .tiles import TS40KTiles, batch_to_device, pack_csr, point_predictions, split_tile
from .training import CapturedTrainingStep, allreduce_flat_grads
from .criterions import (BinaryDiceLoss, BinaryDiceLoss_BCE, FocalTverskyLoss, GENEO_Dice_BCE, GENEO_Dice_Loss, GENEO_Loss,
                         GENEO_Tversky_Loss, TverskyLoss, WeightedMSE)

__all__ = ["SceneNet", "SCENE_Net", "SCENENetQuantile", "SCENE_Net_Class", "cylinder_kernel", "cone_kernel",
           "neg_sphere_kernel", "GENEO_Layer", "GENEO_kernel_torch", "cylinderv2", "arrow", "negSpherev2", "Voxelization",
           "ToTensor", "ToFullDense", "hist_on_voxel", "reg_on_voxel", "prob_to_label", "vxg_to_xyz", "voxelize_batch",
           "PointBatch", "VoxelGrids", "ScenePipeline", "CapturedPipeline", "CapturedTrainingStep", "allreduce_flat_grads", "shard_range", "TS40KTiles", "batch_to_device", "pack_csr",
           "point_predictions", "split_tile", "HipLibraryError", "LIB_PATH", "WeightedMSE", "GENEO_Loss",
           "GENEO_Tversky_Loss", "GENEO_Dice_Loss", "GENEO_Dice_BCE", "TverskyLoss", "FocalTverskyLoss", "BinaryDiceLoss",
           "BinaryDiceLoss_BCE"]
