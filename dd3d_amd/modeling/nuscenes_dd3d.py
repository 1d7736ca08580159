"""``NuscenesDD3D`` meta-architecture (tridet/modeling/dd3d/nuscenes_dd3d.py:300-466), inference branch.

Adds to DD3D: attribute logits and speed predicted from the cls tower (fused into the cls predictor launch, see
engine.ForwardPlan._heads), gathered per candidate inside the select/decode kernel (NuscenesInference,
nuscenes_dd3d.py:268-296), and the cross-camera sample aggregation = camera->global transform + BEV rotated NMS +
500-per-sample cap (postprocessing.py:59-108) as the last kernel stage of the plan.
"""
from collections import OrderedDict

import torch
from torch import nn

from dd3d_amd.layers import Conv2d
from dd3d_amd.modeling.dd3d import DD3D
from dd3d_amd.registry import META_ARCH_REGISTRY
from dd3d_amd.structures import GenericBoxes3D

MAX_NUM_ATTRIBUTES = 3  # tridet/data/datasets/nuscenes/build.py (cycle / pedestrian / vehicle attribute groups)


def get_group_idxs(sample_tokens, num_images_per_sample, inverse=False):
    """tridet/modeling/dd3d/postprocessing.py:111-129: images of one sample share a token; returns sample_token -> list
    of batch indices in first-appearance order (or the inverse map).  ValueError when a group is not complete."""
    group_idxs = OrderedDict()
    for idx, token in enumerate(sample_tokens):
        group_idxs.setdefault(token, []).append(idx)
    if any(len(idxs) != num_images_per_sample for idxs in group_idxs.values()):
        raise ValueError("Group sizes does not match with 'num_images_per_sample'.")
    if inverse:
        return OrderedDict((i, token) for token, idxs in group_idxs.items() for i in idxs)
    return group_idxs


@META_ARCH_REGISTRY.register()
class NuscenesDD3D(DD3D):
    aggregates_samples = True  # engine: append the sample-level BEV stage when DO_POSTPROCESS

    def __init__(self, cfg):
        super().__init__(cfg)
        in_channels = self.backbone_output_shape[0].channels
        self.attr_logits = Conv2d(in_channels, MAX_NUM_ATTRIBUTES, kernel_size=3, stride=1, padding=1, bias=True)
        self.speed = Conv2d(in_channels, 1, kernel_size=3, stride=1, padding=1, bias=True)  # + relu, applied in the fused epilogue
        for m in (self.attr_logits, self.speed):
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)
        self.num_images_per_sample = cfg.DD3D.NUSC.INFERENCE.NUM_IMAGES_PER_SAMPLE
        assert self.num_images_per_sample == 6
        assert cfg.DATALOADER.TEST.NUM_IMAGES_PER_GROUP == 6
        self.max_num_dets_per_sample = cfg.DD3D.NUSC.INFERENCE.MAX_NUM_DETS_PER_SAMPLE  # evaluator limit: 500 per sample

    def _sample_groups(self, batched_inputs):
        if not self.postprocess_in_inference:
            return list(range(len(batched_inputs)))
        groups = get_group_idxs([x["sample_token"] for x in batched_inputs], self.num_images_per_sample)
        out = [0] * len(batched_inputs)
        for gi, idxs in enumerate(groups.values()):
            for i in idxs:
                out[i] = gi
        return out

    def _collect_extra(self, r, d, plan):
        r.pred_attributes = d[:, 20].to(torch.int64)
        r.pred_speeds = d[:, 21]  # (views of the image's private copy of its detection rows, like every other field: DD3D._instances)
        if plan.has_global_boxes:
            r.pred_boxes3d_global = GenericBoxes3D(d[:, 22:26], d[:, 26:29], d[:, 17:20])
