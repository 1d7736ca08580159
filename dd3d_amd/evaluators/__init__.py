"""Evaluator-side kernels (the step right after the forward path)."""
from dd3d_amd.evaluators.rotate_iou import d3_box_overlap, d3_box_overlap_kernel, image_box_overlap, rotate_iou_gpu_eval  # noqa: F401
