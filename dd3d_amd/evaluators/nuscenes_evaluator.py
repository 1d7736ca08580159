"""Result side of tridet/evaluators/nuscenes_evaluator.py: `NuscenesEvaluator.reset / process / build_nusc_detection` (:138-247)
with the same names and outputs; attribute naming on the host, the global velocity of every detection of the call in one launch
(dd3d_amd.evaluators.formatting).  `evaluate` (:249-330) drives the nuScenes devkit, which is outside the hot path.
"""
from collections import OrderedDict, defaultdict

from dd3d_amd.evaluators.formatting import format_boxes3d, xyxy_to_xywh
from dd3d_amd.modeling.nuscenes_dd3d import get_group_idxs

NUM_IMAGES_PER_SAMPLE = 6
# tridet/data/datasets/nuscenes/build.py:50-61 (CATEGORY_IDS order)
NUSCENES_DETECTION_CATEGORIES = ["barrier", "bicycle", "bus", "car", "construction_vehicle", "motorcycle", "pedestrian", "traffic_cone",
                                 "trailer", "truck"]
# nuscenes_evaluator.py:34-43
DEFAULT_ATTRIBUTES = {
    "car": "vehicle.moving", "bus": "vehicle.moving", "construction_vehicle": "vehicle.moving", "trailer": "vehicle.moving",
    "truck": "vehicle.moving", "bicycle": "cycle.with_rider", "motorcycle": "cycle.with_rider", "pedestrian": "pedestrian.moving"
}
# nuscenes_evaluator.py:58-64
VEH_ATTR_CLASSES = ("car", "bus", "construction_vehicle", "trailer", "truck")
PED_ATTR_CLASSES = ("pedestrian", )
CYC_ATTR_CLASSES = ("bicycle", "motorcycle")
VEH_ATTR_ID_TO_NAME = {0: "vehicle.moving", 1: "vehicle.parked", 2: "vehicle.stopped"}
PED_ATTR_ID_TO_NAME = {0: "pedestrian.moving", 1: "pedestrian.standing", 2: "pedestrian.sitting_lying_down"}
CYC_ATTR_ID_TO_NAME = {0: "cycle.with_rider", 1: "cycle.without_rider"}


def attribute_name(class_name, attr):
    """nuscenes_evaluator.py:185-193: the attribute id is taken modulo the size of the class family's table."""
    for classes, table in ((VEH_ATTR_CLASSES, VEH_ATTR_ID_TO_NAME), (PED_ATTR_CLASSES, PED_ATTR_ID_TO_NAME), (CYC_ATTR_CLASSES, CYC_ATTR_ID_TO_NAME)):
        if class_name in classes:
            return table[attr % len(table)]
    return ""


class NuscenesEvaluator:
    def __init__(self, nusc_root=None, dataset_name=None, output_dir=None):
        self._nusc_root = nusc_root
        self._dataset_name = dataset_name
        self._output_dir = output_dir
        self._only_make_submission_file = dataset_name == "nusc_test"
        self.reset()

    def reset(self):
        self._predictions_as_json = []
        self._nusc_sample_results = defaultdict(list)

    def process(self, inputs, outputs):
        sample_tokens = [x["sample_token"] for x in inputs]
        idx_to_token = get_group_idxs(sample_tokens, NUM_IMAGES_PER_SAMPLE, inverse=True)
        for token in set(sample_tokens):  # samples with no detections still get an entry
            self._nusc_sample_results[token]  # pylint: disable=pointless-statement
        for image_idx, (inp, out) in enumerate(zip(inputs, outputs)):
            inst = out["instances"]
            n = len(inst)
            glob = inst.pred_boxes3d_global.vectorize()
            conv = format_boxes3d(inst.pred_boxes3d.vectorize(), glob[:, :4], inst.pred_speeds)
            classes = inst.pred_classes.cpu().tolist()
            boxes = inst.pred_boxes.tensor.cpu().tolist()
            vec = inst.pred_boxes3d.vectorize().cpu().numpy()
            glob = glob.cpu().tolist()
            scores, scores_3d = inst.scores.cpu().tolist(), inst.scores_3d.cpu().tolist()
            attrs = inst.pred_attributes.cpu().tolist()
            token = idx_to_token[image_idx]
            for i in range(n):
                name = NUSCENES_DETECTION_CATEGORIES[classes[i]]
                self._predictions_as_json.append(OrderedDict(
                    category_id=int(classes[i]), category=name, bbox3d=vec[i].tolist(), bbox=xyxy_to_xywh(boxes[i]), score=float(scores[i]),
                    score_3d=float(scores_3d[i]), file_name=inp["file_name"], image_id=inp["image_id"]))
                self._nusc_sample_results[token].append({
                    "sample_token": token, "rotation": glob[i][:4], "translation": glob[i][4:7], "size": glob[i][7:],
                    "detection_name": name, "detection_score": scores_3d[i], "attribute_name": attribute_name(name, attrs[i]),
                    "velocity": [float(conv[i, 8]), float(conv[i, 9])]
                })

    @staticmethod
    def build_nusc_detection(sample_token, box3d_global, category, score, attribute=None, velocity=None):
        """nuscenes_evaluator.py:231-247."""
        v = box3d_global.vectorize().tolist()[0]
        return {
            "sample_token": sample_token, "rotation": v[:4], "translation": v[4:7], "size": v[7:], "detection_name": category,
            "detection_score": score.item(), "attribute_name": DEFAULT_ATTRIBUTES.get(category, "") if attribute is None else attribute,
            "velocity": [0., .0] if velocity is None else velocity
        }

    def evaluate(self):
        raise NotImplementedError("nuscenes_evaluator.py:249-330 runs the nuScenes devkit on _nusc_sample_results; outside the hot path")
