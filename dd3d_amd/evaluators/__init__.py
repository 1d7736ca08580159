"""Evaluator-side kernels (the step right after the forward path)."""
from dd3d_amd.evaluators.rotate_iou import d3_box_overlap, d3_box_overlap_kernel, image_box_overlap, rotate_iou_gpu_eval  # noqa: F401
from dd3d_amd.evaluators.formatting import format_boxes3d  # noqa: F401
from dd3d_amd.evaluators.kitti_3d_evaluator import KITTI3DEvaluator, convert_3d_box_to_kitti  # noqa: F401
from dd3d_amd.evaluators.nuscenes_evaluator import NuscenesEvaluator  # noqa: F401
