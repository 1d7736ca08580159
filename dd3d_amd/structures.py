"""Output / intermediate containers of the forward path (the drop-in boundary's types).

``Boxes3D`` / ``GenericBoxes3D`` follow tridet/structures/boxes3d.py:37-289; ``Instances`` / ``Boxes``
follow detectron2.structures [ext] (fields as attributes, ``__getitem__`` over every field, ``cat``);
``ShapeSpec`` follows detectron2.layers.ShapeSpec.  If detectron2 is importable its ``Instances`` /
``Boxes`` are used so evaluators receive the types they expect.
"""
from collections import namedtuple

import torch

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=[None, None, None, None])

# boxes3d.py:12-16
BOX3D_CORNER_MAPPING = [[1, 1, 1, 1, -1, -1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [1, 1, -1, -1, 1, 1, -1, -1]]

try:  # pragma: no cover
    from detectron2.structures import Boxes, Instances  # noqa: F401
except Exception:

    class Boxes:
        """[ext] detectron2.structures.Boxes: (n,4) float XYXY absolute."""
        def __init__(self, tensor):
            if tensor.numel() == 0:
                tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
            assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
            self.tensor = tensor

        def __len__(self):
            return self.tensor.shape[0]

        def __getitem__(self, item):
            if isinstance(item, int):
                return Boxes(self.tensor[item].view(1, -1))
            return Boxes(self.tensor[item])

        def to(self, *a, **k):
            return Boxes(self.tensor.to(*a, **k))

        def area(self):
            b = self.tensor
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        def nonempty(self, threshold=0.0):
            b = self.tensor
            return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

        @property
        def device(self):
            return self.tensor.device

        @classmethod
        def cat(cls, boxes_list):
            if len(boxes_list) == 0:
                return cls(torch.empty(0))
            return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

        def __repr__(self):
            return "Boxes(" + str(self.tensor) + ")"

    class Instances:
        """[ext] detectron2.structures.Instances: per-image container, fields as attributes."""
        def __init__(self, image_size, **kwargs):
            object.__setattr__(self, "_image_size", image_size)
            object.__setattr__(self, "_fields", {})
            for k, v in kwargs.items():
                self.set(k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            if name.startswith("_"):
                object.__setattr__(self, name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
            return self._fields[name]

        def set(self, name, value):
            n = len(value)
            if len(self._fields):
                assert len(self) == n, f"Adding a field of length {n} to a Instances of length {len(self)}"
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def remove(self, name):
            del self._fields[name]

        def get(self, name):
            return self._fields[name]

        def get_fields(self):
            return self._fields

        def to(self, *a, **k):
            ret = Instances(self._image_size)
            for kk, v in self._fields.items():
                ret.set(kk, v.to(*a, **k) if hasattr(v, "to") else v)
            return ret

        def __getitem__(self, item):
            if type(item) == int:
                if item >= len(self) or item < -len(self):
                    raise IndexError("Instances index out of range!")
                item = slice(item, None, len(self))
            ret = Instances(self._image_size)
            for k, v in self._fields.items():
                ret.set(k, v[item])
            return ret

        def __len__(self):
            for v in self._fields.values():
                return v.__len__()
            raise NotImplementedError("Empty Instances does not support __len__!")

        @staticmethod
        def cat(instance_lists):
            assert len(instance_lists) > 0
            if len(instance_lists) == 1:
                return instance_lists[0]
            ret = Instances(instance_lists[0].image_size)
            for k in instance_lists[0]._fields.keys():
                values = [i.get(k) for i in instance_lists]
                v0 = values[0]
                if isinstance(v0, torch.Tensor):
                    values = torch.cat(values, dim=0)
                elif hasattr(type(v0), "cat"):
                    values = type(v0).cat(values)
                else:
                    raise ValueError(f"Unsupported type {type(v0)} for concatenation")
                ret.set(k, values)
            return ret

        def __repr__(self):
            s = f"Instances(num_instances={len(self) if self._fields else 0}, image_height={self._image_size[0]}, "
            s += f"image_width={self._image_size[1]}, fields=[{', '.join(f'{k}: {v}' for k, v in self._fields.items())}])"
            return s


def quaternion_to_matrix(q):
    """[ext] pytorch3d.transforms.quaternion_to_matrix (w,x,y,z).  Host-side helper for ``corners``."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r), two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k), two_s * (j * k - i * r), two_s * (i * k - j * r), two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j)
        ), -1
    )
    return o.reshape(q.shape[:-1] + (3, 3))


class GenericBoxes3D:
    """tridet/structures/boxes3d.py:37-154: quat (w,x,y,z), tvec, size (W,L,H)."""
    def __init__(self, quat, tvec, size):
        self.quat = torch.as_tensor(quat, dtype=torch.float32).reshape(-1, 4)
        self._tvec = torch.as_tensor(tvec, dtype=torch.float32).reshape(-1, 3)
        self.size = torch.as_tensor(size, dtype=torch.float32).reshape(-1, 3)

    @property
    def tvec(self):
        return self._tvec

    @property
    def corners(self):
        """boxes3d.py:47-64: corner_i = R(q) (0.5 (l,w,h) * sign_i) + tvec, (l,w,h) = size[:, [1,0,2]]."""
        R = quaternion_to_matrix(self.quat)
        signs = 0.5 * self.quat.new_tensor(BOX3D_CORNER_MAPPING).T  # (8,3)
        c = self.size[:, [1, 0, 2]].unsqueeze(1) * signs.unsqueeze(0)
        return torch.einsum("nij,nkj->nki", R, c) + self.tvec.unsqueeze(1)

    def vectorize(self):
        """boxes3d.py:142-144."""
        return torch.cat([self.quat, self.tvec, self.size], dim=1)

    @classmethod
    def cat(cls, boxes_list, dim=0):
        if len(boxes_list) == 0:
            return cls(torch.empty(0), torch.empty(0), torch.empty(0))
        return cls(
            torch.cat([b.quat for b in boxes_list], dim), torch.cat([b.tvec for b in boxes_list], dim),
            torch.cat([b.size for b in boxes_list], dim)
        )

    def __getitem__(self, item):
        if isinstance(item, int):
            return GenericBoxes3D(self.quat[item].view(1, -1), self.tvec[item].view(1, -1), self.size[item].view(1, -1))
        return GenericBoxes3D(self.quat[item], self.tvec[item], self.size[item])

    def __len__(self):
        return self.quat.shape[0]

    def clone(self):
        return GenericBoxes3D(self.quat.clone(), self.tvec.clone(), self.size.clone())

    @property
    def device(self):
        return self.quat.device

    def to(self, *a, **k):
        return GenericBoxes3D(self.quat.to(*a, **k), self.tvec.to(*a, **k), self.size.to(*a, **k))


class Boxes3D(GenericBoxes3D):
    """tridet/structures/boxes3d.py:157-289: vision-based container; tvec = K^-1 [proj_ctr,1] * depth."""
    def __init__(self, quat, proj_ctr, depth, size, inv_intrinsics):
        self.quat = quat
        self.proj_ctr = proj_ctr
        self.depth = depth
        self.size = size
        self.inv_intrinsics = inv_intrinsics

    @property
    def tvec(self):
        pts = torch.nn.functional.pad(self.proj_ctr, (0, 1), value=1.0)
        ray = torch.matmul(self.inv_intrinsics, pts.unsqueeze(-1)).squeeze(-1)
        return ray * self.depth

    @classmethod
    def cat(cls, boxes_list, dim=0):
        if len(boxes_list) == 0:
            return cls(torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0), torch.empty(0))
        return cls(*[torch.cat([getattr(b, f) for b in boxes_list], dim) for f in cls._FIELDS])

    _FIELDS = ("quat", "proj_ctr", "depth", "size", "inv_intrinsics")

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes3D(
                self.quat[item].view(1, -1), self.proj_ctr[item].view(1, -1), self.depth[item].view(1, -1),
                self.size[item].view(1, -1), self.inv_intrinsics[item].view(1, 3, 3)
            )
        return Boxes3D(*[getattr(self, f)[item] for f in self._FIELDS])

    def __len__(self):
        return self.quat.shape[0]

    def clone(self):
        return Boxes3D(*[getattr(self, f).clone() for f in self._FIELDS])

    def to(self, *a, **k):
        return Boxes3D(*[getattr(self, f).to(*a, **k) for f in self._FIELDS])


class _Quat:
    """Minimal unit-quaternion (w, x, y, z) used by ``Pose`` (the reference uses pyquaternion, not installed here)."""
    def __init__(self, q):
        import numpy as np
        self.elements = np.asarray(q.elements if hasattr(q, "elements") else q, dtype=np.float64).reshape(4).copy()

    def __mul__(self, o):
        import numpy as np
        w1, x1, y1, z1 = self.elements
        w2, x2, y2, z2 = o.elements
        return _Quat(np.array([
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
        ]))

    @property
    def inverse(self):
        import numpy as np
        return _Quat(self.elements * np.array([1.0, -1.0, -1.0, -1.0]) / float(self.elements @ self.elements))

    @property
    def rotation_matrix(self):
        import numpy as np
        w, x, y, z = self.elements / np.linalg.norm(self.elements)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def rotate(self, v):
        import numpy as np
        return self.rotation_matrix @ np.asarray(v, dtype=np.float64)


class Pose:
    """SE(3) pose, the slice of tridet/structures/pose.py:6-164 the forward path needs on the host: a rotation quaternion
    (``.quat.elements`` = w,x,y,z) and a translation (``.tvec``); composition, inverse, 4x4 matrix.  Any object with those
    two attributes (e.g. the reference's own Pose) is accepted wherever a pose is expected."""
    def __init__(self, wxyz=(1.0, 0.0, 0.0, 0.0), tvec=(0.0, 0.0, 0.0)):
        import numpy as np
        self.quat = _Quat(wxyz)
        assert abs(1.0 - np.linalg.norm(self.quat.elements)) < 1.0e-3
        self.tvec = np.asarray(tvec, dtype=np.float64).reshape(3)

    def __mul__(self, other):
        if isinstance(other, Pose):
            return Pose((self.quat * other.quat).elements, self.quat.rotate(other.tvec) + self.tvec)
        return NotImplemented

    def inverse(self):
        qinv = self.quat.inverse
        return Pose(qinv.elements, qinv.rotate(-self.tvec))

    @property
    def matrix(self):
        import numpy as np
        m = np.eye(4)
        m[:3, :3] = self.quat.rotation_matrix
        m[:3, 3] = self.tvec
        return m

    @classmethod
    def from_yaw(cls, yaw_deg, tvec=(0.0, 0.0, 0.0)):
        import math
        h = math.radians(yaw_deg) / 2.0
        return cls((math.cos(h), 0.0, 0.0, math.sin(h)), tvec)
